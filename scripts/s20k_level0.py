"""S20k level 0 on its own (depth 1), with and without the squares: what the triangles of the tile lists cost."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracing_amd")
for nsq in (500, 0):
    w, c = rt.scenes.s20k(n_squares=nsq)
    c = c.with_sample_rate(64)
    w.render(c, 1920, 1080, 1, spp_begin=0, spp_end=2)
    best = 1e9
    for _ in range(3):
        acc, st = w.render(c, 1920, 1080, 1, return_stats=True)
        best = min(best, st.ms_total)
    print("squares", nsq, "level 0 only (depth 1):", round(best, 2), "ms", flush=True)
