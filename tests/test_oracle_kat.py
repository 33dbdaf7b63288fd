"""Pins the CPU oracle (oracle/oracle.cpp) against every known answer the reference holds for this path:
the reference's own unit tests (src/util.rs:440-500, src/object.rs:362-393), replayed assertion for
assertion, plus the hand-derived vectors of SURVEY.md §8c and the Random123 Philox4x32-10 KATs.
No GPU needed."""
import ctypes as C
import math

import numpy as np
import pytest

EPS = 1e-3  # util.rs:10, also approx's default epsilon for Vec3 (util.rs:310-312)


def close(a, b, eps=EPS):
    return np.all(np.abs(np.asarray(a, float) - np.asarray(b, float)) <= eps)


# ---------------------------------------------------------------- util.rs:440-487 test_vec3
def test_vec3(orc):
    op = orc.vec3_op
    assert abs(op(0, (1, 2, 3), (5, 40, 200))[0] - 685.0) <= EPS
    assert close(op(1, (1, 0, 0), (0, 1, 0)), (0, 0, 1))
    assert close(op(1, (0, 1, 0), (0, 0, 1)), (1, 0, 0))
    assert close(op(1, (0, 0, 1), (1, 0, 0)), (0, 1, 0))
    assert close(op(1, (0, 1, 0), (1, 0, 0)), (0, 0, -1))
    assert close(op(1, (0, 0, 1), (0, 1, 0)), (-1, 0, 0))
    assert close(op(1, (1, 0, 0), (0, 0, 1)), (0, -1, 0))
    assert close(op(2, (1, 2, 3), (10, 100, 1000)), (11, 102, 1003))
    assert close(op(8, (1, 2, 3), s=10.0), (11, 12, 13))
    assert close(op(2, (10, 100, 1000), (1, 2, 3)), (11, 102, 1003))
    assert close(op(3, (1, 2, 3), (10, 100, 1000)), (-9, -98, -997))
    assert close(op(9, (1, 2, 3), s=10.0), (-9, -8, -7))
    assert close(op(3, (10, 100, 1000), (1, 2, 3)), (9, 98, 997))
    assert close(op(10, (1, 2, 3), s=10.0), (9, 8, 7))
    assert close(op(4, (5, 6, 9), (1, 2, 3)), (5, 12, 27))
    assert close(op(5, (1, 1, 1), (1, 2, 3)), (1, 0.5, 1.0 / 3.0))
    assert close(op(6, (1, 2, 3), s=5.0), (5, 10, 15))
    assert close(op(7, (10, 20, 30), s=5.0), (2, 4, 6))
    assert close(op(11, (1, 2, 3), s=24.0), (24, 12, 8))
    assert close(op(6, (1, 2, 3), s=10.0), (10, 20, 30))
    assert close(op(7, (1, 2, 3), s=10.0), (0.1, 0.2, 0.3))
    acc = np.zeros(3)   # Sum: fold from (0,0,0), util.rs:350-354
    for v in [(1, 2, 3), (10, 20, 30), (100, 200, 300)]:
        acc = op(2, acc, v)
    assert close(acc, (111, 222, 333))


# ---------------------------------------------------------------- util.rs:489-492 test_min_max
def test_min_max(orc):
    L = orc.lib()

    def arr(*v):
        return (C.c_double * len(v))(*v)

    mx = L.orc_max_macro(arr(0.0, 10.0, 20.0, 30.0), 4)
    mn = L.orc_min_macro(arr(6.0, 9.0, 7.0), 3)
    assert abs(L.orc_min_macro(arr(mx, mn, 5.0), 3) - 5.0) <= EPS
    # NaN behaviour PhongModel::render relies on (basic.rs:68-70): min!(NaN, 1.) == 1.
    assert L.orc_min_macro(arr(float("nan"), 1.0), 2) == 1.0


# ---------------------------------------------------------------- util.rs:494-500 test_gen_point_in_sphere
def test_gen_point_in_sphere(orc):
    L = orc.lib()
    rng = np.random.default_rng(1)
    out = (C.c_double * 3)()
    zmin = 1.0
    for u0, u1 in rng.random((100000, 2)):
        L.orc_gen_point_in_sphere(5.0, u0, u1, out)
        assert out[0] * out[0] + out[1] * out[1] + out[2] * out[2] <= 25.0 + EPS
        zmin = min(zmin, out[2])
    assert zmin >= 0.0          # the quirk: always the +z hemisphere (util.rs:388,393)
    L.orc_gen_point_in_disk(2.0, 0.25, 0.5, out)   # theta = pi/4, r = .5  (util.rs:397-402)
    assert close(out[:], (math.cos(math.pi / 4), math.sin(math.pi / 4), 0.0), 1e-12)


# ---------------------------------------------------------------- object.rs:362-393 test_triangle
def _tri_hit(orc, ray_pos, ray_dir):
    out = (C.c_double * 14)()
    ok = orc.lib().orc_shape_hit_info(1, orc.d3((0, -1, 0)), orc.d3((1, 1, 0)), orc.d3((-1, 1, 0)), 0.0,
                                      orc.d3(ray_pos), orc.d3(ray_dir), 1, out)
    return ok, np.array(out[:])


def test_triangle(orc):
    L = orc.lib()
    ok, h = _tri_hit(orc, (0, 0, 1), (0, 0, -1))
    assert ok == 1
    dir_out, normal, pos = h[7:10], h[1:4], h[10:13]
    assert close(pos, EPS * dir_out + np.zeros(3))
    assert close(dir_out, (0, 0, 1))
    assert close(normal, (0, 0, 1))
    assert _tri_hit(orc, (3, 0, 1), (0, 0, -1))[0] == 0
    assert _tri_hit(orc, (3, 0, -1), (0, 0, 1))[0] == 0
    assert _tri_hit(orc, (0, 0, 1), (1, 0, 0))[0] == 0
    assert _tri_hit(orc, (0, 0, -1), (1, 0, 0))[0] == 0
    assert _tri_hit(orc, (0, 0, -1), (0, 0, -1))[0] == 0
    p = [orc.d3((0, -1, 0)), orc.d3((1, 1, 0)), orc.d3((-1, 1, 0))]
    assert L.orc_tri_is_in_plane(*p, orc.d3((1, 4, 0))) == 1
    assert L.orc_tri_is_in_plane(*p, orc.d3((1, 4, 0.1))) == 0
    assert L.orc_tri_contain(*p, orc.d3((0, 0, 0))) == 1
    assert L.orc_tri_contain(*p, orc.d3((3, 0, 0))) == 0
    assert L.orc_tri_contain(*p, orc.d3((0, 0, 0.1))) == 0
    assert L.orc_tri_contain(*p, orc.d3((1, 1, 0))) == 1


# ---------------------------------------------------------------- SURVEY §8c hand-derived vectors
def _sphere_hit(orc, c, r, pos, d):
    out = (C.c_double * 14)()
    ok = orc.lib().orc_shape_hit_info(0, orc.d3(c), orc.d3((0, 0, 0)), orc.d3((0, 0, 0)), float(r), orc.d3(pos),
                                      orc.d3(d), 1, out)
    return ok, np.array(out[:])


def test_sphere_known_answers(orc):
    ok, h = _sphere_hit(orc, (0, 0, 0), 1.0, (0, 0, 5), (0, 0, -1))      # (1)
    assert ok and h[0] == 4.0 and close(h[4:7], (0, 0, 1), 0) and close(h[1:4], (0, 0, 1), 0)
    assert h[13] == 0.0 and close(h[7:10], (0, 0, 1), 0) and close(h[10:13], (0, 0, 1.001), 1e-15)
    ok, h = _sphere_hit(orc, (0, 0, 0), 1.0, (0, 0, 0), (0, 0, 1))       # (2) from inside: far root, flipped
    assert ok and h[0] == 1.0 and close(h[1:4], (0, 0, -1), 0) and h[13] == 1.0
    ok, h = _sphere_hit(orc, (0, 0, 0), -1.0, (0, 0, 5), (0, 0, -1))     # negative radius flips the normal first
    assert ok and h[0] == 4.0 and h[13] == 1.0
    assert _sphere_hit(orc, (0, 0, 0), 1.0, (0, 0, 5), (0, 0, 1))[0] == 0    # behind: t2 < 0
    assert _sphere_hit(orc, (0, 0, 0), 1.0, (2, 0, 5), (0, 0, -1))[0] == 0   # delta < 0


def test_refract_and_schlick(orc):
    L = orc.lib()
    out = (C.c_double * 6)()
    assert L.orc_hitinfo_refract(orc.d3((0, 0, 1)), orc.d3((0, 0, 0)), orc.d3((0, 0, -1)), 1 / 1.5, out) == 1   # (3)
    assert close(out[3:6], (0, 0, -1), 1e-15) and close(out[0:3], (0, 0, -EPS), 1e-15)
    # (4) r0 = (1-ior)/(1+ior)^2 is NOT squared: -0.08 for ior 1.5
    assert abs(L.orc_hitinfo_reflect_prob(orc.d3((0, 0, 1)), orc.d3((0, 0, -1)), 1.5) - (-0.08)) < 1e-15
    g = (math.sqrt(1 - 1e-12), 0, -1e-6)   # grazing: cos -> 0, prob -> 1
    assert abs(L.orc_hitinfo_reflect_prob(orc.d3((0, 0, 1)), orc.d3(g), 1.5) - 1.0) < 1e-4
    # (5) leaving glass (ratio 1.5): total internal reflection iff |cos| < sqrt(1 - 1/2.25) = 0.745356
    for cosv, expect in [(0.74, 0), (0.75, 1)]:
        din = (math.sqrt(1 - cosv * cosv), 0, -cosv)
        assert L.orc_hitinfo_refract(orc.d3((0, 0, 1)), orc.d3((0, 0, 0)), orc.d3(din), 1.5, out) == expect


def test_sky(orc):
    out = (C.c_double * 3)()
    for z, col in [(1, (0.5, 0.7, 1.0)), (0, (0.75, 0.85, 1.0)), (-1, (1, 1, 1))]:   # (6)
        orc.lib().orc_sky_color(orc.d3((0, 0, z)), out)
        assert close(out[:], col, 1e-15)


def test_camera_basis(orc, rt):
    L = orc.lib()
    s, r, u, k = [(C.c_double * 5)() for _ in range(4)]
    cam = rt.scenes.cornell_rs()[1]                                          # (7)
    pod = orc.camera_pod(cam)
    L.orc_camera_basis(C.byref(pod), s, r, u, k)
    assert close(s[:3], (-1, 0, 0), 1e-15) and close(r[:3], (0, 1, 0), 1e-15) and close(u[:3], (0, 0, 1), 1e-15)
    assert abs(k[0] - 1.1157034787043882) < 1e-12 and abs(k[1] - 1.1157034787043882) < 1e-12
    assert close(k[2:5], (-0.2, 0, 0), 1e-12)
    cam = rt.scenes.scene_rs()[1]                                            # (8)
    pod = orc.camera_pod(cam)
    L.orc_camera_basis(C.byref(pod), s, r, u, k)
    assert close(s[:3], np.array((-13, 3, -2)) / math.sqrt(182), 1e-15)
    assert abs(k[0] - 3.5265396141692996) < 1e-12 and abs(k[1] - 1.6 * 3.5265396141692996) < 1e-12
    # the host mirror's Camera::new/look agrees with the oracle's
    assert close(cam.sight, s[:3], 0) and close(cam.up, u[:3], 0) and close(cam.right(), r[:3], 0)


def test_empty_world_sky_is_analytic(orc, rt):
    """(9) an empty world with a SkyLight renders the sky gradient of each primary ray."""
    w = rt.World.empty()
    w.add_light(rt.SkyLight())
    cam = rt.Camera.new((0, -5, 1), (0, 0, 1)).with_fov(60.0)
    ow = orc.OracleWorld(w)
    rgb, rays = ow.trace_paths(cam, 16, 8, 0, 128, 5, 9)
    d = orc.emit_rays(cam, 16, 8, 0, 128, 9)[:, 3:6]
    t = 0.5 * (d[:, 2] + 1.0)
    want = (1 - t)[:, None] * np.ones(3) + t[:, None] * np.array((0.5, 0.7, 1.0))
    assert close(rgb, want, 1e-15) and rays == 128


def test_mirror_chain_is_deterministic(orc, rt):
    """(10) Metal fuzz 0 draws nothing that matters: the same path twice, and under another seed
    (jitter off), gives the same colour."""
    w = rt.World.empty()
    w.add_obj(rt.Object(rt.Sphere((0, 0, 0), 1.0), rt.Metal(0.0, 0.9).with_color((0.9, 0.8, 0.7))))
    w.add_obj(rt.Object(rt.Sphere((0, 0, -101), 100.0), rt.Specular(0.5)))
    w.add_light(rt.SkyLight())
    cam = rt.Camera.new((0, -4, 0.5), (0, 0, 0))
    ow = orc.OracleWorld(w)
    a, _ = ow.trace_paths(cam, 8, 8, 0, 64, 10, 1, jitter=False)
    b, _ = ow.trace_paths(cam, 8, 8, 0, 64, 10, 2, jitter=False)
    assert np.array_equal(a, b) and a.max() > 0


# ---------------------------------------------------------------- RNG
def test_philox_kat(orc):
    """Random123 known-answer vectors for philox4x32-10."""
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        out = (C.c_uint32 * 4)()
        orc.lib().orc_philox((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
        assert tuple(out) == want


def test_scene_rng_matches_oracle_philox(orc, rt):
    from importlib import import_module
    sc = import_module("ray-tracing_amd.scenes")
    for i, seed in [(0, 0x5EED0001), (12345, 0x1234567890ABCDEF)]:
        out = (C.c_uint32 * 4)()
        orc.lib().orc_philox((C.c_uint32 * 4)(i, 0, 0, 0), (C.c_uint32 * 2)(seed & 0xffffffff, seed >> 32), out)
        assert sc._philox_word0(i, seed) == out[0]


# ---------------------------------------------------------------- golden fixtures regenerate bit-exactly
@pytest.mark.parametrize("name", ["scene_rs", "cornell", "s500"])
def test_golden_primary_hits(orc, rt, name):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"primary_{name}.npz"))
    world, cam = {"scene_rs": lambda: rt.scenes.scene_rs(motion=False), "cornell": rt.scenes.cornell_rs,
                  "s500": rt.scenes.s500}[name]()
    ow = orc.OracleWorld(world)
    ids, t, nrm, pt, fl = ow.hit(g["rays"].astype(np.float64), seed=0x5EED0002)
    assert np.array_equal(ids, g["ids"]) and np.array_equal(t, g["t"]) and np.array_equal(nrm, g["normal"])
    rays64 = orc.emit_rays(cam, 64, 40, 0, 64 * 40, 0x5EED0002, jitter=False)
    assert np.array_equal(rays64, g["rays64"])


def test_golden_images(orc, rt):
    import os
    for name, (fn, w, h) in {"cornell": (rt.scenes.cornell_rs, 40, 30), "s500": (rt.scenes.s500, 48, 27)}.items():
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"image_{name}.npz"))
        world, cam = fn()
        acc, rays = orc.OracleWorld(world).render(cam, w, h, int(g["depth"]), 0x5EED0002, 0, int(g["spp"]), nthreads=4)
        assert rays == int(g["rays"])
        assert np.array_equal(acc.astype(np.float32), g["accum"])


def test_rgb8(orc):
    acc = np.array([[0.0, 0.25, 1.0], [2.0, -1.0, 0.5]]) * 4
    out = orc.to_rgb8(acc, 4.0)
    assert out.tolist() == [[0, 127, 255], [255, 0, int(255.99 * math.sqrt(0.5))]]
