"""Host side without a GPU: the C-ABI library loads and exports every symbol include/cftrt.h declares, and
the host mirror flattens the reference's trait objects exactly as the reference evaluates them."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(rt):
    hdr = open(os.path.join(ROOT, "include", "cftrt.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cftrt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = rt.load()                      # dlopen, no GPU call
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == {n for n, _, _ in rt.SYMBOLS}


def test_struct_layouts_match_header(rt):
    L = __import__("importlib").import_module("ray-tracing_amd._lib")
    assert C.sizeof(L.cftrt_material) == 48
    assert C.sizeof(L.cftrt_light) == 72
    assert C.sizeof(L.cftrt_camera) == 104
    assert C.sizeof(L.cftrt_stats) == 5 * 8 + 8 * 8 + 4 * 4 + 8 * 256


def _closest_of_tris(orc, tris, pos, d):
    """closest hit over flattened triangles, first index wins ties"""
    best = None
    out = (C.c_double * 14)()
    for a, b, c in tris:
        if orc.lib().orc_shape_hit_info(1, orc.d3(a), orc.d3(b), orc.d3(c), 0.0, orc.d3(pos), orc.d3(d), 1, out):
            if best is None or out[0] < best[0]:
                best = (out[0], tuple(out[1:4]))
    return best


@pytest.mark.parametrize("shape_kind", ["cube", "square"])
def test_flatten_matches_reference_evaluation(rt, orc, shape_kind):
    """Cube::squares order / Square corner order (object.rs:147-162,215-230): the flattened triangles give
    the same hit as the oracle's un-flattened shape, bit for bit."""
    rng = np.random.default_rng(3)
    if shape_kind == "cube":
        args = ((0.1, -0.2, 0.3), (1.0, 0.5, 0.0), (-0.5, 1.0, 0.2), 1.7)
        shape, kind = rt.Cube(*args), 3
    else:
        args = ((0.0, 0.0, 0.99), (1.0, 0.0, 0.0), (0.0, -1.0, 0.0), 0.9)
        shape, kind = rt.Square(*args), 2
    tris = shape._flatten()[1]
    assert len(tris) == (12 if shape_kind == "cube" else 2)
    out = (C.c_double * 14)()
    nhit = 0
    for _ in range(400):
        pos = rng.uniform(-3, 3, 3)
        d = -pos + rng.uniform(-0.8, 0.8, 3)
        ok = orc.lib().orc_shape_hit_info(kind, orc.d3(args[0]), orc.d3(args[1]), orc.d3(args[2]), args[3],
                                          orc.d3(pos), orc.d3(d), 1, out)
        got = _closest_of_tris(orc, tris, pos, d)
        assert bool(ok) == (got is not None)
        if ok:
            nhit += 1
            assert out[0] == got[0] and tuple(out[1:4]) == got[1]
    assert nhit > (100 if shape_kind == "cube" else 20)


def test_world_flatten_tables(rt):
    w, cam = rt.scenes.cornell_rs()
    f = w.flatten()
    assert len(f.sph_radius) == 0 and len(f.tri_obj) == 12 and set(f.tri_obj) == {0}
    assert f.lights[0]["kind"] == 1 and f.lights[0]["tri_count"] == 2 and len(f.ltri_p0) == 2
    assert cam.fov == 45.0 and cam.aspect == 1.0 and cam.aperture == 0.0    # 45 *radians* (ray.rs:147)
    w, cam = rt.scenes.scene_rs()
    f = w.flatten()
    assert len(w.objects) == 488 and len(f.sph_radius) == 488          # 1 + 22*22 + 3 (scene.rs:25-72)
    assert f.sph_radius[0] == 1000.0 and tuple(f.sph_center[0]) == (0.0, 0.0, -1000.0)
    moving = (f.sph_moving != 0).any(axis=1)
    assert 120 < moving.sum() < 270 and np.all(f.sph_moving[moving] == (0.0, 0.0, 0.1))   # ~40 % (scene.rs:55)
    kinds = np.array([m[0] for m in f.materials[1:485]])
    assert 0.7 < (kinds == 0).mean() < 0.9 and 0.08 < (kinds == 1).mean() < 0.22 and (kinds == 2).mean() < 0.1
    assert np.array_equal(f.sph_obj, np.arange(488))
    w, _ = rt.scenes.s500()
    assert len(w.objects) == 500 and not (w.flatten().sph_moving != 0).any()


def test_s20k_counts(rt):
    w, _ = rt.scenes.s20k(n_spheres=1900, n_squares=50)
    f = w.flatten()
    assert len(f.sph_radius) == 1901 and len(f.tri_obj) == 100
    assert f.tri_obj.min() == 1901 and np.all(np.diff(f.tri_obj) >= 0)
    r = f.sph_radius[1:]
    assert r.min() >= 0.05 and r.max() <= 0.2


def test_builders(rt):
    m = rt.Metal(3.0, 0.9)
    assert m.fuzz == 1.0 and m.with_fuzz(3.0).fuzz == 3.0          # compose.rs:18 vs :26-29
    assert rt.Metal(0.3, 1.0).with_color((0.7, 0.6, 0.5)).with_fuzz(0.0)._row() == (1, 0.0, 1.0, (0.7, 0.6, 0.5))
    p = rt.PhongModel()
    assert (p.shininess, p.diffuse, p.color) == (1.0, 0.5, (1.0, 1.0, 1.0))          # basic.rs:18-24
    assert rt.Dielectric(1.5)._row() == (2, 1.5, 0.0, (1.0, 1.0, 1.0))
    o = rt.Object(rt.Sphere((0, 0, 0), 1), rt.LambertianModel(0.5)).moved((0, 0, 0.1))
    assert o.moving_to == (0.0, 0.0, 0.1)
    c = rt.Camera.new((13, -3, 2), (0, 0, 0)).with_fov(20.0)
    assert abs(c.fov - 20.0 / 180.0 * np.pi) < 1e-15
    with pytest.raises(TypeError):
        rt.Object(object(), rt.LambertianModel(1.0))          # user-defined Shape impl: no device counterpart
    with pytest.raises(TypeError):
        rt.World.empty().add_light(object())


def test_sample_range(rt):
    for spp, ws in [(64, 8), (50, 8), (5, 8), (1024, 3)]:
        r = [rt.multi.sample_range(spp, ws, k) for k in range(ws)]
        assert r[0][0] == 0 and r[-1][1] == spp
        assert all(r[k][1] == r[k + 1][0] for k in range(ws - 1))
        sizes = [e - b for b, e in r]
        assert max(sizes) - min(sizes) <= 1


def test_square_from_points_flattens_as_a_pair(rt, orc):
    """Square::from_points (object.rs:169-174): tri0 = (p0,p1,p2), tri1 = (p1,p2,p3), hit = tri0.or(tri1) (object.rs:192).
    The host mirror flags the pair (tri_first) for the device; the oracle's Square reproduces the rule: where both
    triangles are hit the square reports tri0's hit although tri1's is nearer."""
    p = [(0.0, 0.0, 0.0), (2.0, 0.0, 0.0), (0.0, 2.0, 0.0), (0.5, 0.5, 1.0)]     # tri1 rises above tri0
    w = rt.World.empty()
    w.add_obj(rt.Object(rt.Sphere((5, 5, 5), 0.1), rt.LambertianModel(0.5)))
    w.add_obj(rt.Object(rt.Square.from_points(*p), rt.LambertianModel(0.5)))
    w.add_obj(rt.Object(rt.Square((0, 0, -3), (1, 0, 0), (0, 1, 0), 2.0), rt.LambertianModel(0.5)))
    w.add_light(rt.LightShape(rt.Square.from_points(*[(x, y, z + 4.0) for x, y, z in p])))
    f = w.flatten()
    assert f.tri_first.tolist() == [-1, 0, -1, -1] and f.ltri_first.tolist() == [-1, 0]
    assert tuple(f.tri_p0[1]) == p[1] and tuple(f.tri_p1[1]) == p[2] and tuple(f.tri_p2[1]) == p[3]
    ow = orc.OracleWorld(w)
    # straight down through (0.6, 0.6): meets tri1 (z ~ 0.8) first, then tri0 (z = 0) - the square reports tri0
    ray = np.array([[0.6, 0.6, 5.0, 0.0, 0.0, -1.0]])
    ids, t, nrm, pt, fl = ow.hit(ray)
    assert ids[0] == 1 and t[0] == 5.0 and nrm[0].tolist() == [0.0, 0.0, 1.0]
    out = (C.c_double * 14)()
    assert orc.lib().orc_shape_hit_info(1, orc.d3(p[1]), orc.d3(p[2]), orc.d3(p[3]), 0.0, orc.d3(ray[0, :3]), orc.d3(ray[0, 3:]), 1, out)
    assert out[0] < 5.0                                  # tri1 alone is nearer
    # a ray that misses tri0 (parallel to its plane) and meets tri1 only, at (0.75, 0.75, 0.5)
    r = 2.0 ** -0.5
    ray2 = np.array([[3.0, 3.0, 0.5, -r, -r, 0.0]])
    ids, t, _, pt2, _ = ow.hit(ray2)
    assert ids[0] == 1 and np.allclose(pt2[0], (0.75, 0.75, 0.5))


def test_comm_entry_points_reject_bad_arguments_from_many_threads(rt):
    """No GPU here: the multi-GPU entry points still return codes (never crash) when eight threads hit them at once -
    the first call loads RCCL behind a function-local static, so its initialisation is what races."""
    import threading
    L = rt.load()
    out = []

    def work():
        buf = (C.c_ubyte * 128)()
        out.append((L.cftrt_comm_unique_id(None), L.cftrt_comm_init(None, 2, 0, buf), L.cftrt_reduce(None, None, 0, 0),
                    L.cftrt_comm_destroy(None), L.cftrt_comm_unique_id(buf) in (0, 2)))

    ts = [threading.Thread(target=work) for _ in range(8)]
    [t.start() for t in ts]
    [t.join(timeout=60) for t in ts]
    assert out == [(1, 1, 1, 1, True)] * 8
