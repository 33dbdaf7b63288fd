"""S20k variants: without the squares, with them (as in scenes.s20k), and with them kept low (z 0.1 .. 0.3, so that
the grid's box stays as flat as the sphere layer): what the triangles cost in the tile lists and in the grid walk."""
import importlib, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracing_amd")
A = rt.api if hasattr(rt, "api") else rt

def with_squares(zlo, zhi, n=500, seed=12345):
    import random
    rd = random.Random(seed).random
    w, c = rt.scenes.s20k(n_squares=0)
    d = A.LambertianModel(1.0)
    for k in range(n):
        center = (-11.0 + 22.0 * rd(), -11.0 + 22.0 * rd(), zlo + (zhi - zlo) * rd())
        th, ph = 2.0 * math.pi * rd(), math.pi * rd()
        x = (math.cos(th) * math.sin(ph), math.sin(th) * math.sin(ph), math.cos(ph))
        up = (0.0, 0.0, 1.0) if abs(x[2]) < 0.9 else (1.0, 0.0, 0.0)
        y = A.v_unit(A.v_cross(x, up))
        w.add_obj(A.Object(A.Square(center, x, y, 0.5), d.with_color((0.5, 0.5, 0.5))))
    return w, c

def one_tiny_square():
    w, c = rt.scenes.s20k(n_squares=0)
    w.add_obj(A.Object(A.Square((0.013, 0.017, 0.2), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), 1e-3), A.LambertianModel(1.0).with_color((0.5, 0.5, 0.5))))
    return w, c

for name, (w, c) in (("no squares", rt.scenes.s20k(n_squares=0)), ("no squares + one 1e-3 square (the kernels with triangle code)", one_tiny_square()), ("squares z 0.3..1.0", with_squares(0.3, 1.0)),
                     ("squares z 0.1..0.3", with_squares(0.1, 0.3))):
    c = c.with_sample_rate(64)
    for depth in (1, 50):
        w.render(c, 1920, 1080, depth, spp_begin=0, spp_end=2)
        best = None
        for _ in range(2):
            acc, st = w.render(c, 1920, 1080, depth, return_stats=True)
            if best is None or st.ms_total < best.ms_total: best = st
        print(name, "| depth", depth, ":", round(best.ms_total, 2), "ms,", best.rays, "rays", flush=True)
