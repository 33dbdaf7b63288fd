"""Runs BASELINE.json's five configs on one GPU (and the CPU restatement on C1/C2/C3-sample) and prints one
JSON object per config.  Not the driver's bench (bench.py is); fills the table in BASELINE.md §4."""
import hashlib, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("ray-tracing_amd")
import oracle_binding as ob

def gpu(name, world, cam, W, H, spp, depth, reps=2):
    cam = cam.with_sample_rate(spp)
    world.render(cam, W, H, depth, spp_begin=0, spp_end=min(spp, 2))   # warm-up (allocations)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        acc, st = world.render(cam, W, H, depth, return_stats=True)
        dt = time.perf_counter() - t0
        if best is None or st.ms_total < best[1].ms_total: best = (dt, st, acc)
    dt, st, acc = best
    f = world.flatten()
    r = dict(config=name, size=f"{W}x{H}x{spp} depth {depth}", n_spheres=int(len(f.sph_radius)), n_tris=int(len(f.tri_obj)),
             paths=int(st.paths), rays=int(st.rays), passes=int(st.passes), device_ms=st.ms_total, wall_ms_with_pcie=dt * 1e3,
             mrays_per_s=st.rays / st.ms_total / 1e3, intersect_ms=st.ms_intersect, shade_ms=st.ms_shade,
             shadow_ms=st.ms_shadow, intersect_tflops=st.flops_intersect / st.ms_intersect / 1e9,
             pct_fp32_peak=st.flops_intersect / st.ms_intersect / 1e9 / 157.3 * 100, levels=int(st.levels),
             exact_second=int(st.exact_second), exact_rescan=int(st.exact_rescan),
             image_sha1=hashlib.sha1(np.ascontiguousarray(acc).tobytes()).hexdigest()[:16], mean_rgb=[float(x) for x in acc.astype(np.float64).mean(axis=(0, 1)) / spp])
    print(json.dumps(r), flush=True)
    return r

def cpu(name, world, cam, W, H, spp, depth, threads=16):
    ow = ob.OracleWorld(world)
    t0 = time.perf_counter()
    acc, rays = ow.render(cam.with_sample_rate(spp), W, H, depth, rt.scenes.SAMPLE_SEED, 0, spp, nthreads=threads)
    dt = time.perf_counter() - t0
    r = dict(config=name + " [CPU restatement]", size=f"{W}x{H}x{spp} depth {depth}", rays=int(rays), seconds=dt,
             mrays_per_s=rays / dt / 1e6, threads=threads, mean_rgb=[float(x) for x in acc.astype(np.float64).mean(axis=(0, 1)) / spp])
    print(json.dumps(r), flush=True)
    return r

if os.environ.get("CFTRT_GRID"):   # "mode[,cells_per_prim]" (0 = the plain lists)
    g = os.environ["CFTRT_GRID"].split(",")
    rt.default_device().set_grid(int(g[0]), float(g[1]) if len(g) > 1 else 0.0)
if os.environ.get("CFTRT_TILE_LISTS"):
    rt.default_device().set_tile_lists(int(os.environ["CFTRT_TILE_LISTS"]))
if os.environ.get("CFTRT_FINISH"):        # "on[,below]"
    f = os.environ["CFTRT_FINISH"].split(",")
    rt.default_device().set_finish(int(f[0]), int(f[1]) if len(f) > 1 else 0)
if os.environ.get("CFTRT_LDS_GRID"):      # "on[,above]"
    f = os.environ["CFTRT_LDS_GRID"].split(",")
    rt.default_device().set_lds_grid(int(f[0]), int(f[1]) if len(f) > 1 else 0)
if os.environ.get("CFTRT_CLASSES"):       # "on[,cells[,above]]"
    f = [int(x) for x in os.environ["CFTRT_CLASSES"].split(",")] + [0, 0]
    rt.default_device().set_walk_classes(f[0], f[1], f[2])
if os.environ.get("CFTRT_VARIANT"):
    rt.default_device().set_variant(int(os.environ["CFTRT_VARIANT"]))
which = sys.argv[1:] or ["C1", "C2", "C2p", "C3", "C4", "C5", "cpu"]
if "C1" in which:
    w, c = rt.scenes.scene_rs(); gpu("C1 scene.rs (motion blur on)", w, c, 800, 500, 50, 10)
if "C2" in which:
    w, c = rt.scenes.cornell_rs(); gpu("C2 cornell.rs area light", w, c, 400, 300, 256, 10)
if "C2p" in which:
    w, c = rt.scenes.cornell_rs(point_light=True); gpu("C2 cornell Phong/point light (shadow kernel)", w, c, 400, 300, 256, 10)
if "C3" in which:
    w, c = rt.scenes.s500(); gpu("C3 S500", w, c, 1920, 1080, 64, 50, reps=3)
if "C5" in which:
    w, c = rt.scenes.s20k(); gpu("C5 S20k", w, c, 1920, 1080, 64, 50, reps=1)
if "C4" in which:
    w, c = rt.scenes.s500(width=3840, height=2160); gpu("C4 S500 4K 1024spp (1 GPU)", w, c, 3840, 2160, 1024, 50, reps=1)
if "cpu" in which:
    w, c = rt.scenes.scene_rs(); cpu("C1 scene.rs", w, c, 800, 500, 50, 10)
    w, c = rt.scenes.cornell_rs(); cpu("C2 cornell.rs", w, c, 400, 300, 256, 10)
