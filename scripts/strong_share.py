"""One rank's share of the fixed S500 frame under strong scaling (64 spp split over N ranks), timed on one GPU:
what a rank of an N-GPU run has to do before the reduce.  Not a scaling measurement (no second GPU here)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracing_amd")
if os.environ.get("CFTRT_FINISH"):        # "on[,below]"
    f = os.environ["CFTRT_FINISH"].split(",")
    rt.default_device().set_finish(int(f[0]), int(f[1]) if len(f) > 1 else 0)
w, c = rt.scenes.s500()
c = c.with_sample_rate(64)
w.render(c, 1920, 1080, 50, spp_begin=0, spp_end=2)
ns = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]   # (one N: for a kernel trace of that share)
for n in ns:
    b, e = rt.multi.sample_range(64, n, 0)
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        acc, st = w.render(c, 1920, 1080, 50, spp_begin=b, spp_end=e, return_stats=True)
        wall = (time.perf_counter() - t0) * 1e3
        if best is None or st.ms_total < best[0]: best = (st.ms_total, wall, st.rays, st.launches_intersect)
    print(f"N={n}: rank 0 renders samples [{b},{e}): device {best[0]:.2f} ms (wall with PCIe image copy {best[1]:.2f} ms), {best[2]} rays, "
          f"{best[3]} level launches -> {best[2] * n / best[0] / 1e3:.0f} Mrays/s if all {n} ranks took this long", flush=True)
