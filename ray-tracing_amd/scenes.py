"""Seeded scene definitions: the reference's two examples and the synthetic benchmark scenes.

The reference builds its scene from an unseeded `thread_rng()` (examples/scene.rs:34-35), so a seeded
restatement with the same distributions and draw order is the only reproducible form.  Draw i of a
scene is word 0 of Philox4x32-10(counter=(i,0,0,0), key=seed), as a 24-bit uniform in [0,1).
"""
import math

from . import api as A

SCENE_SEED = 0x5EED0001
SAMPLE_SEED = 0x5EED0002


def _philox_word0(i, seed):
    M0, M1, W0, W1, MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    c0, c1, c2, c3 = i & MASK, (i >> 32) & MASK, 0, 0
    k0, k1 = seed & MASK, (seed >> 32) & MASK
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0


def _sq(x):
    return x * x   # f64::powi(2)


class Rd:
    """`rng.gen::<f64>()` stand-in: sequential uniform draws in [0,1)."""

    def __init__(self, seed):
        self.seed, self.i = seed, 0

    def __call__(self):
        w = _philox_word0(self.i, self.seed)
        self.i += 1
        return (w >> 8) / 16777216.0


def _rtiow_spheres(world, rd, motion, lo=-11, hi=11):
    """The 22x22 grid of examples/scene.rs:36-62, same draw order:
    cx, cy, material, [colour x3 | colour x3 + fuzz | -], move."""
    d = A.LambertianModel(1.0)
    t = A.Dielectric(1.5)
    m = A.Metal(0.3, 1.0)
    for a in range(lo, hi):
        for b in range(lo, hi):
            center = (a + 0.9 * rd(), b + 0.9 * rd(), 0.2)
            choose = rd()
            if choose < 0.8:
                col = (_sq(rd()), _sq(rd()), _sq(rd()))
                obj = A.Object(A.Sphere(center, 0.2), d.with_color(col))
            elif choose < 0.95:
                col = ((1.0 + rd()) / 2.0, (1.0 + rd()) / 2.0, (1.0 + rd()) / 2.0)
                obj = A.Object(A.Sphere(center, 0.2), m.with_color(col).with_fuzz(rd() / 2.0))
            else:
                obj = A.Object(A.Sphere(center, 0.2), t)
            choose_move = rd()
            if motion and choose_move < 0.4:
                obj = obj.moved((0.0, 0.0, 0.1))
            world.add_obj(obj)


def _big_three(world):
    d = A.LambertianModel(1.0)
    t = A.Dielectric(1.5)
    m = A.Metal(0.3, 1.0)
    world.add_obj(A.Object(A.Sphere((0.0, 0.0, 1.0), 1.0), t))
    world.add_obj(A.Object(A.Sphere((-4.0, 0.0, 1.0), 1.0), d.with_color((0.4, 0.2, 0.1))))
    world.add_obj(A.Object(A.Sphere((4.0, 0.0, 1.0), 1.0), m.with_color((0.7, 0.6, 0.5)).with_fuzz(0.0)))


def scene_rs(seed=SCENE_SEED, motion=True, width=800, height=500, spp=50):
    """C1 — examples/scene.rs:25-80: 488 objects, SkyLight, thin-lens camera.  Returns (world, camera)."""
    world = A.World.empty()
    world.add_obj(A.Object(A.Sphere((0.0, 0.0, -1000.0), 1000.0), A.LambertianModel(1.0).with_color((0.5, 0.5, 0.5))))
    _rtiow_spheres(world, Rd(seed), motion)
    _big_three(world)
    world.add_light(A.SkyLight())
    camera = (A.Camera.new((13.0, -3.0, 2.0), (0.0, 0.0, 0.0)).with_focus_dist(10.0).with_aperture(0.1)
              .with_fov(20.0).with_aspect(width / height).with_sample_rate(spp))
    return world, camera


def cornell_rs(spp=5, point_light=False):
    """C2 — examples/cornell.rs:18-30: camera inside a Lambertian cube, square LightShape under the
    ceiling; default camera (fov 45 rad, aspect 1).  point_light=True is the Phong/PointLight variant
    that exercises the shadow kernel (BASELINE.json configs[1] "area/point light")."""
    world = A.World.empty()
    if point_light:
        mat = A.PhongModel().with_shininess(8.0).with_diffuse(0.8).with_color((0.9, 0.8, 0.7))
        world.add_obj(A.Object(A.Cube((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), 2.0), mat))
        world.add_obj(A.Object(A.Sphere((-0.3, 0.3, -0.6), 0.4), A.PhongModel().with_color((0.2, 0.5, 0.9))))
        world.add_light(A.PointLight((0.0, 0.0, 0.8)))
    else:
        world.add_obj(A.Object(A.Cube((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), 2.0), A.LambertianModel(0.8)))
        world.add_light(A.LightShape(A.Square((0.0, 0.0, 0.99), (1.0, 0.0, 0.0), (0.0, -1.0, 0.0), 0.9)))
    camera = A.Camera.new((0.8, 0.0, 0.0), (0.0, 0.0, 0.0)).with_sample_rate(spp)
    return world, camera


def s500(seed=SCENE_SEED, width=1920, height=1080, spp=64):
    """C3/C4 — SURVEY §8d "S500": the scene.rs generator without motion blur, padded from 488 to 500
    objects with 12 more r=0.2 spheres drawn by the same grid law (cells a=11, b=-11..0)."""
    world = A.World.empty()
    world.add_obj(A.Object(A.Sphere((0.0, 0.0, -1000.0), 1000.0), A.LambertianModel(1.0).with_color((0.5, 0.5, 0.5))))
    rd = Rd(seed)
    _rtiow_spheres(world, rd, motion=False)
    _big_three(world)
    # 12 extra spheres, same law, on the next grid column
    d = A.LambertianModel(1.0)
    for b in range(-11, 1):
        center = (11 + 0.9 * rd(), b + 0.9 * rd(), 0.2)
        col = (_sq(rd()), _sq(rd()), _sq(rd()))
        world.add_obj(A.Object(A.Sphere(center, 0.2), d.with_color(col)))
    world.add_light(A.SkyLight())
    camera = (A.Camera.new((13.0, -3.0, 2.0), (0.0, 0.0, 0.0)).with_focus_dist(10.0).with_aperture(0.1)
              .with_fov(20.0).with_aspect(width / height).with_sample_rate(spp))
    return world, camera


def s20k(seed=SCENE_SEED + 1, width=1920, height=1080, spp=64, n_spheres=19000, n_squares=500):
    """C5 — SURVEY §8d "S20k": 19 000 spheres r in [0.05,0.2] on a jittered 140x140 grid over the
    ground sphere + 500 Squares (1 000 triangles) of side 0.5 at random orientations; same material
    mix and camera as S500."""
    world = A.World.empty()
    world.add_obj(A.Object(A.Sphere((0.0, 0.0, -1000.0), 1000.0), A.LambertianModel(1.0).with_color((0.5, 0.5, 0.5))))
    rd = Rd(seed)
    d = A.LambertianModel(1.0)
    t = A.Dielectric(1.5)
    m = A.Metal(0.3, 1.0)

    def material():
        choose = rd()
        if choose < 0.8:
            return d.with_color((_sq(rd()), _sq(rd()), _sq(rd())))
        if choose < 0.95:
            return m.with_color(((1.0 + rd()) / 2.0, (1.0 + rd()) / 2.0, (1.0 + rd()) / 2.0)).with_fuzz(rd() / 2.0)
        return t

    grid = 140
    cell = 22.0 / grid
    cells = [(i, j) for i in range(grid) for j in range(grid)]
    stride = len(cells) / float(n_spheres)
    for k in range(n_spheres):
        i, j = cells[int(k * stride)]
        r = 0.05 + 0.15 * rd()
        center = (-11.0 + (i + 0.9 * rd()) * cell, -11.0 + (j + 0.9 * rd()) * cell, r)
        world.add_obj(A.Object(A.Sphere(center, r), material()))
    for k in range(n_squares):
        center = (-11.0 + 22.0 * rd(), -11.0 + 22.0 * rd(), 0.3 + 0.7 * rd())
        th, ph = 2.0 * math.pi * rd(), math.pi * rd()
        x = (math.cos(th) * math.sin(ph), math.sin(th) * math.sin(ph), math.cos(ph))
        up = (0.0, 0.0, 1.0) if abs(x[2]) < 0.9 else (1.0, 0.0, 0.0)
        y = A.v_unit(A.v_cross(x, up))
        world.add_obj(A.Object(A.Square(center, x, y, 0.5), material()))
    world.add_light(A.SkyLight())
    camera = (A.Camera.new((13.0, -3.0, 2.0), (0.0, 0.0, 0.0)).with_focus_dist(10.0).with_aperture(0.1)
              .with_fov(20.0).with_aspect(width / height).with_sample_rate(spp))
    return world, camera
