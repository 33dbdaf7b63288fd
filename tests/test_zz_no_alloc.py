"""cftrt_render_device_static - the entry point that allocates nothing (include/cftrt.h) - against
cftrt_render_device on the same inputs, through the C ABI.  Both run the same kernels on the same path ids; only
the workspace differs (a file-scope device array instead of hipMalloc'ed buffers), so the accumulation images
must be equal bit for bit.  Two calls in a row, the second smaller: the array keeps the first call's contents."""
import ctypes as C

import numpy as np
import pytest

import hip_mem as hm   # device buffers through the HIP runtime libcftrt.so itself uses (no torch in this process)

pytestmark = pytest.mark.gpu
SEED = 0x5EED0002


def _pair(rt, world, cam, W, H, depth, spp_begin, spp_end, dev):
    a, b = hm.DeviceArray(H * W * 3), hm.DeviceArray(H * W * 3)      # zeroed and synchronised
    sa = world.render_device(cam, W, H, depth, a.data_ptr(), seed=SEED, spp_begin=spp_begin, spp_end=spp_end, device=dev)
    sb = world.render_device(cam, W, H, depth, b.data_ptr(), seed=SEED, spp_begin=spp_begin, spp_end=spp_end, device=dev,
                             static=True)
    return a.numpy(), b.numpy(), sa, sb


def test_static_entry_matches_render_device(rt):
    dev = rt.Device(0)            # a ctx of its own: its hipMalloc'ed workspace has never grown when the static call runs
    world, cam = rt.scenes.s500()
    world.commit(dev)
    # call 1: 640x360, 8 spp - one pass in either workspace
    a, b, sa, sb = _pair(rt, world, cam, 640, 360, 50, 0, 8, dev)
    assert a.any() and np.array_equal(a, b)
    assert int(sa.rays) == int(sb.rays) and int(sb.passes) == 1
    assert list(sa.rays_per_level[: int(sa.levels)]) == list(sb.rays_per_level[: int(sb.levels)])
    assert float(sb.ms_total) > 0.0 and float(sb.ms_intersect) == 0.0      # no per-kernel events in the static call
    # call 2, smaller input, other sample range: the array still holds call 1's queues and counters
    a, b, sa, sb = _pair(rt, world, cam, 320, 184, 50, 5, 8, dev)
    assert a.any() and np.array_equal(a, b)
    assert int(sa.rays) == int(sb.rays)
    # call 3: the headline frame size; one sample fills 199 of the 256 MiB, so 3 samples are 3 passes here against
    # one pass in the hipMalloc'ed workspace - k_resolve adds sample by sample, so the bits still agree
    a, b, sa, sb = _pair(rt, world, cam, 1920, 1080, 50, 0, 3, dev)
    assert int(sb.passes) == 3 and int(sa.passes) == 1
    assert np.array_equal(a, b)
    # no device memory is taken or returned by the static call (hipMemGetInfo before / after two more calls)
    acc = hm.DeviceArray(640 * 360 * 3)
    free0 = hm.mem_free_bytes()
    for k in range(2):
        world.render_device(cam, 640, 360, 50, acc.data_ptr(), seed=SEED, spp_begin=k, spp_end=k + 2, device=dev, static=True)
    assert hm.mem_free_bytes() == free0
    dev.close()


def test_static_entry_phong_motion_and_limits(rt):
    dev = rt.Device(0)
    # Phong queue (k_shadow) carved from the array too; then a two-scan scene (motion + sky)
    world, cam = rt.scenes.cornell_rs(point_light=True)
    world.commit(dev)
    a, b, sa, sb = _pair(rt, world, cam, 160, 120, 10, 0, 8, dev)
    assert int(sb.phong_rays) > 0 and np.array_equal(a, b)
    world2, cam2 = rt.scenes.scene_rs()
    world2.commit(dev)
    a, b, sa, sb = _pair(rt, world2, cam2, 320, 200, 10, 0, 4, dev)
    assert a.any() and np.array_equal(a, b)
    # one sample of a 3840x2160 image is 796 MB of workspace: refused, nothing rendered, the error says why
    big = hm.DeviceArray(16)   # never written: the call fails before any launch
    with pytest.raises(rt.CftrtError) as e:
        world2.render_device(cam2, 3840, 2160, 10, big.data_ptr(), seed=SEED, spp_begin=0, spp_end=1, device=dev, static=True)
    assert e.value.code == 4 and "static workspace" in str(e.value)
    assert not big.numpy().any()
    # null buffer / uncommitted scene: same status codes as cftrt_render_device
    L = rt.load()
    h = C.c_void_p()
    assert L.cftrt_scene_create(dev.h, C.byref(h)) == 0
    pod = cam2._pod()
    assert L.cftrt_render_device_static(h, C.byref(pod), 4, 4, 0, 1, 5, 0, big.ptr, None) == 3
    assert L.cftrt_render_device_static(h, C.byref(pod), 4, 4, 0, 1, 5, 0, None, None) == 1
    L.cftrt_scene_destroy(h)
    dev.close()
