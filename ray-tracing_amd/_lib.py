"""ctypes binding of libcftrt.so (include/cftrt.h).  No fallback: if the HIP library is missing the
import of this module fails loudly — there is no CPU path in the product."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CFTRT_LIB") or os.path.join(_HERE, "csrc", "libcftrt.so")   # CFTRT_LIB: experiments only

CFTRT_MAX_LEVELS = 256
CFTRT_STATIC_SCRATCH_BYTES = 256 << 20


class cftrt_material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("p0", C.c_double), ("p1", C.c_double),
                ("color", C.c_double * 3)]


class cftrt_light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("sph_begin", C.c_int32), ("sph_count", C.c_int32),
                ("tri_begin", C.c_int32), ("tri_count", C.c_int32), ("reserved", C.c_int32),
                ("v", C.c_double * 3), ("color", C.c_double * 3)]


class cftrt_camera(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("up", C.c_double * 3), ("sight", C.c_double * 3),
                ("focus_dist", C.c_double), ("aperture", C.c_double), ("fov_rad", C.c_double),
                ("aspect", C.c_double)]


class cftrt_stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("phong_rays", C.c_uint64),
                ("exact_second", C.c_uint64), ("exact_rescan", C.c_uint64), ("flops", C.c_double),
                ("flops_intersect", C.c_double), ("ms_total", C.c_double), ("ms_generate", C.c_double),
                ("ms_intersect", C.c_double), ("ms_shade", C.c_double), ("ms_shadow", C.c_double),
                ("ms_resolve", C.c_double), ("launches_intersect", C.c_uint32), ("levels", C.c_uint32),
                ("passes", C.c_uint32), ("reserved", C.c_uint32),
                ("rays_per_level", C.c_uint64 * CFTRT_MAX_LEVELS)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n not in ("rays_per_level", "reserved")}
        d["rays_per_level"] = [int(x) for x in self.rays_per_level[: max(self.levels, 1)]]
        return d


# every symbol include/cftrt.h declares: (name, restype, argtypes)
_P = C.c_void_p
_D = C.POINTER(C.c_double)
_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int32)
SYMBOLS = [
    ("cftrt_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("cftrt_ctx_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("cftrt_ctx_destroy", None, [_P]),
    ("cftrt_last_error", C.c_char_p, [_P]),
    ("cftrt_ctx_set_workspace_limit", C.c_int, [_P, C.c_uint64]),
    ("cftrt_ctx_set_pipeline", C.c_int, [_P, C.c_int]),
    ("cftrt_ctx_set_wave_cull", C.c_int, [_P, C.c_int]),
    ("cftrt_ctx_set_grid", C.c_int, [_P, C.c_int, C.c_double]),
    ("cftrt_ctx_set_variant", C.c_int, [_P, C.c_int]),
    ("cftrt_ctx_set_finish", C.c_int, [_P, C.c_int, C.c_uint32]),
    ("cftrt_ctx_set_shards", C.c_int, [_P, C.c_int]),
    ("cftrt_ctx_set_tile_lists", C.c_int, [_P, C.c_int]),
    ("cftrt_ctx_set_lds_grid", C.c_int, [_P, C.c_int, C.c_uint32]),
    ("cftrt_ctx_set_walk_classes", C.c_int, [_P, C.c_int, C.c_uint32, C.c_uint32]),
    ("cftrt_scene_create", C.c_int, [_P, C.POINTER(_P)]),
    ("cftrt_scene_destroy", None, [_P]),
    ("cftrt_scene_set_spheres", C.c_int, [_P, _D, _D, _D, _I, _I, C.c_int64]),
    ("cftrt_scene_set_triangles", C.c_int, [_P, _D, _D, _D, _D, _I, _I, C.c_int64]),
    ("cftrt_scene_set_materials", C.c_int, [_P, C.POINTER(cftrt_material), C.c_int64]),
    ("cftrt_scene_set_lights", C.c_int, [_P, C.POINTER(cftrt_light), C.c_int64, _D, _D, C.c_int64, _D, _D, _D,
                                         C.c_int64]),
    ("cftrt_scene_set_triangle_pairs", C.c_int, [_P, _I, C.c_int64, _I, C.c_int64]),
    ("cftrt_scene_set_precise_radius", C.c_int, [_P, C.c_double]),
    ("cftrt_scene_commit", C.c_int, [_P]),
    ("cftrt_render", C.c_int, [_P, C.POINTER(cftrt_camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                               C.c_uint32, C.c_uint64, _F, C.POINTER(cftrt_stats)]),
    ("cftrt_render_device", C.c_int, [_P, C.POINTER(cftrt_camera), C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_uint32, C.c_uint32, C.c_uint64, _P, C.POINTER(cftrt_stats)]),
    ("cftrt_render_device_static", C.c_int, [_P, C.POINTER(cftrt_camera), C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_uint32, C.c_uint32, C.c_uint64, _P, C.POINTER(cftrt_stats)]),
    ("cftrt_comm_unique_id", C.c_int, [C.POINTER(C.c_ubyte)]),
    ("cftrt_comm_init", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_ubyte)]),
    ("cftrt_reduce", C.c_int, [_P, _P, C.c_uint64, C.c_int]),
    ("cftrt_comm_destroy", C.c_int, [_P]),
    ("cftrt_generate", C.c_int, [_P, C.POINTER(cftrt_camera), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                                 C.c_uint64, C.c_int, _F]),
    ("cftrt_intersect", C.c_int, [_P, _F, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, _I, _D, _F,
                                  _F, _I]),
    ("cftrt_level_hits", C.c_int, [_P, C.POINTER(cftrt_camera), C.c_uint32, C.c_uint32, _F, C.c_uint64, C.c_uint64,
                                   C.c_uint64, C.c_uint32, _I, _D, _I]),
    ("cftrt_trace_paths", C.c_int, [_P, C.POINTER(cftrt_camera), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                                    C.c_uint32, C.c_uint64, C.c_int, _F, C.POINTER(cftrt_stats)]),
    ("cftrt_resolve_rgb8", C.c_int, [_P, _F, C.c_uint64, C.c_double, C.POINTER(C.c_uint8)]),
]

_lib = None


def load():
    """Loads libcftrt.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        if os.environ.get("CFTRT_LIB") and not hasattr(lib, name):
            continue              # (experiments only: an older build of the library for an A/B on one box)
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class CftrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"cftrt error {code}: {msg}")
        self.code = code
