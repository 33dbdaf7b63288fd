"""Device buffers for the tests through the HIP runtime the product library already uses (ctypes on
libamdhip64.so.7: the loader hands back the copy libcftrt.so is linked against, or torch's if torch was imported
first - never a second runtime in the process)."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so.7")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    return _hip


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with HIP error {rc}")


class DeviceArray:
    """n float32 zeros on the current device."""

    def __init__(self, n):
        self.n = int(n)
        self.ptr = C.c_void_p()
        _ok(hip().hipMalloc(C.byref(self.ptr), 4 * self.n), "hipMalloc")
        self.zero()

    def zero(self):
        _ok(hip().hipMemset(self.ptr, 0, 4 * self.n), "hipMemset")
        _ok(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")

    def data_ptr(self):
        return self.ptr.value

    def numpy(self):
        out = np.empty(self.n, np.float32)
        _ok(hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, 4 * self.n, 2), "hipMemcpy D2H")
        return out

    def upload(self, a):
        a = np.ascontiguousarray(a, np.float32).reshape(-1)
        assert a.size == self.n
        _ok(hip().hipMemcpy(self.ptr, a.ctypes.data_as(C.c_void_p), 4 * self.n, 1), "hipMemcpy H2D")

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def mem_free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    _ok(hip().hipMemGetInfo(C.byref(f), C.byref(t)), "hipMemGetInfo")
    return f.value
