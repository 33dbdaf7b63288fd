"""Host-side mirror of the reference crate's public scene/camera API (cfeitong/ray-tracing), in the
reference's own names, over the C ABI of include/cftrt.h.

  reference                                   here
  Vec3 / vec3! / From<(T,T,T)>  util.rs       3-tuples of float (helpers below keep f64 op order)
  Camera::new().with_*()        ray.rs:59-152 Camera.new(from, to).with_*()
  Sphere/Triangle/Square/Cube   object.rs     same names; Square/Cube expand to triangles in the
                                              reference's face/triangle order (object.rs:147-162,215-230)
  Object::new(s, m).moved(d)    object.rs:25-40
  World::empty/add_obj/add_light object.rs:311-324
  LambertianModel/Metal/Dielectric/PhongModel/Specular/Transparent  material/*.rs
  SkyLight/LightShape/PointLight/ParallelLight                      light.rs
  (the examples' hand-rolled render loop, scene.rs:91-102)  ->  World.render(camera, w, h, depth)

The trait objects (`Box<dyn Shape>`, `Arc<dyn Material>`, `Arc<dyn LightSource>`) are flattened here
into the SoA arrays the device wants; user-defined impls have no device counterpart and are rejected.
"""
import ctypes as C
import math
import weakref

import numpy as np

from . import _lib as L

EPS = 1e-3  # util.rs:10


# ---- Vec3 helpers on 3-tuples; operation order as util.rs so the f64 results match the reference
def _v(x):
    x = tuple(float(c) for c in x)
    if len(x) != 3:
        raise ValueError("Vec3 needs 3 components")
    return x


def v_add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def v_sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def v_neg(a): return (-a[0], -a[1], -a[2])
def v_mul(a, s): return (a[0] * s, a[1] * s, a[2] * s)
def v_div(a, s): return (a[0] / s, a[1] / s, a[2] / s)
def v_dot(a, b): return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
def v_cross(a, b): return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
def v_len(a): return math.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])
def v_unit(a): return v_div(a, v_len(a))


# ---- shapes (object.rs)
class Sphere:
    def __init__(self, o, r):
        self.center = _v(o)
        self.radius = float(r)

    def _flatten(self):
        return [(self.center, self.radius)], []


class Triangle:
    def __init__(self, p0, p1, p2):
        self.p0, self.p1, self.p2 = _v(p0), _v(p1), _v(p2)

    def _flatten(self):
        return [], [(self.p0, self.p1, self.p2)]


class Square:
    """Square::new(center, x, y, len), object.rs:148-162 (x, y are NOT normalised here)."""

    def __init__(self, center, x, y, len):
        center, x, y = _v(center), _v(x), _v(y)
        self._ctor = (center, x, y, float(len))
        x2 = v_div(v_mul(x, float(len)), 2.0)
        y2 = v_div(v_mul(y, float(len)), 2.0)
        p0 = v_add(v_sub(center, x2), y2)
        p1 = v_sub(v_sub(center, x2), y2)
        p2 = v_sub(v_add(center, x2), y2)
        p3 = v_add(v_add(center, x2), y2)
        self.tri0 = Triangle(p0, p1, p2)
        self.tri1 = Triangle(p2, p3, p0)
        self.or_pair = False

    @classmethod
    def from_points(cls, p0, p1, p2, p3):  # object.rs:169-174
        """tri0 = (p0, p1, p2), tri1 = (p1, p2, p3).  The reference's Square::hit_info is `tri0.or(tri1)`
        (object.rs:192): tri0 wins whenever it is hit, even where tri1 is nearer.  For Square::new that never differs
        from the closest of the two (coplanar triangles that share only the diagonal); for from_points, whose triangles
        may overlap or leave the plane, it does - the pair is flagged (`or_pair`) and the device's exact evaluation
        ignores tri1 wherever tri0 is hit (cftrt_scene_set_triangle_pairs)."""
        sq = cls.__new__(cls)
        sq._ctor = None
        sq._points = (_v(p0), _v(p1), _v(p2), _v(p3))
        sq.tri0 = Triangle(p0, p1, p2)
        sq.tri1 = Triangle(p1, p2, p3)
        sq.or_pair = True
        return sq

    def _flatten(self):
        return [], [(self.tri0.p0, self.tri0.p1, self.tri0.p2), (self.tri1.p0, self.tri1.p1, self.tri1.p2)]


class Cube:
    """Cube::new(center, x, y, len); faces in the order of Cube::squares, object.rs:215-230."""

    def __init__(self, center, x, y, len):
        self.center, self.x, self.y, self.len = _v(center), _v(x), _v(y), float(len)

    def squares(self):
        x = v_unit(self.x)
        y = v_unit(self.y)
        z = v_unit(v_cross(x, y))
        c, l = self.center, self.len
        return [
            Square(v_add(c, v_mul(x, l / 2.0)), y, z, l),
            Square(v_sub(c, v_mul(x, l / 2.0)), v_neg(y), z, l),
            Square(v_add(c, v_mul(y, l / 2.0)), v_neg(x), z, l),
            Square(v_sub(c, v_mul(y, l / 2.0)), x, z, l),
            Square(v_add(c, v_mul(z, l / 2.0)), x, y, l),
            Square(v_sub(c, v_mul(z, l / 2.0)), x, v_neg(y), l),
        ]

    def _flatten(self):
        tris = []
        for sq in self.squares():
            tris += sq._flatten()[1]
        return [], tris


# ---- materials (material/basic.rs, material/compose.rs)
class _Material:
    kind = -1

    def _row(self):
        raise NotImplementedError

    def _copy(self):
        c = self.__class__.__new__(self.__class__)
        c.__dict__.update(self.__dict__)
        return c


class LambertianModel(_Material):
    kind = 0

    def __init__(self, albedo):
        self.albedo, self.color = float(albedo), (1.0, 1.0, 1.0)

    def with_color(self, color):
        c = self._copy(); c.color = _v(color); return c

    def _row(self): return (0, self.albedo, 0.0, self.color)


class Metal(_Material):
    kind = 1

    def __init__(self, fuzz, albedo):
        self.fuzz = 1.0 if fuzz > 1.0 else float(fuzz)   # compose.rs:18 (with_fuzz does not clamp)
        self.albedo, self.color = float(albedo), (1.0, 1.0, 1.0)

    def with_fuzz(self, fuzz):
        c = self._copy(); c.fuzz = float(fuzz); return c

    def with_albedo(self, albedo):
        c = self._copy(); c.albedo = float(albedo); return c

    def with_color(self, color):
        c = self._copy(); c.color = _v(color); return c

    def _row(self): return (1, self.fuzz, self.albedo, self.color)


class Dielectric(_Material):
    kind = 2

    def __init__(self, ior):
        self.ior = float(ior)

    def with_ior(self, ior):
        c = self._copy(); c.ior = float(ior); return c

    def _row(self): return (2, self.ior, 0.0, (1.0, 1.0, 1.0))


class PhongModel(_Material):
    kind = 3

    def __init__(self):
        self.shininess, self.diffuse, self.color = 1.0, 0.5, (1.0, 1.0, 1.0)

    def with_shininess(self, s):
        c = self._copy(); c.shininess = float(s); return c

    def with_diffuse(self, kd):
        c = self._copy(); c.diffuse = float(kd); return c

    def with_color(self, color):
        c = self._copy(); c.color = _v(color); return c

    def _row(self): return (3, self.shininess, self.diffuse, self.color)


class Specular(_Material):
    kind = 4

    def __init__(self, albedo):
        self.albedo = float(albedo)

    def with_albedo(self, albedo):
        c = self._copy(); c.albedo = float(albedo); return c

    def _row(self): return (4, self.albedo, 0.0, (1.0, 1.0, 1.0))


class Transparent(_Material):
    kind = 5

    def __init__(self, opacity, ior):
        self.opacity, self.ior, self.color = float(opacity), float(ior), (1.0, 1.0, 1.0)

    def with_ior(self, ior):
        c = self._copy(); c.ior = float(ior); return c

    def with_opacity(self, o):
        c = self._copy(); c.opacity = float(o); return c

    def with_color(self, color):
        c = self._copy(); c.color = _v(color); return c

    def _row(self): return (5, self.opacity, self.ior, self.color)


# ---- lights (light.rs)
class SkyLight:
    kind = 0


class LightShape:
    kind = 1

    def __init__(self, shape):
        if not hasattr(shape, "_flatten"):
            raise TypeError("LightShape needs one of Sphere/Triangle/Square/Cube")
        self.shape = shape


class PointLight:
    kind = 2

    def __init__(self, pos):
        self.pos, self.light_color = _v(pos), (1.0, 1.0, 1.0)

    def with_color(self, c):
        o = PointLight(self.pos); o.light_color = _v(c); return o


class ParallelLight:
    kind = 3

    def __init__(self, dir):
        self.dir, self.light_color = _v(dir), (1.0, 1.0, 1.0)

    def with_color(self, c):
        o = ParallelLight(self.dir); o.light_color = _v(c); return o


# ---- Object / Camera / World
class Object:
    def __init__(self, shape, material):
        if not hasattr(shape, "_flatten"):
            raise TypeError("user-defined Shape impls have no device counterpart")
        if not isinstance(material, _Material):
            raise TypeError("user-defined Material impls have no device counterpart")
        self.shape, self.material, self.moving_to = shape, material, (0.0, 0.0, 0.0)

    def moved(self, delta):
        self.moving_to = _v(delta)
        return self


class Camera:
    """Camera::new(from, to) + builder, ray.rs:59-152.  fov is stored in radians; the default 45. is
    therefore 45 radians, as in the reference (ray.rs:147)."""

    def __init__(self):
        raise TypeError("use Camera.new(from, to)")

    @classmethod
    def new(cls, frm, to):
        c = cls.__new__(cls)
        c.pos = _v(frm)
        c.up = (0.0, 0.0, 1.0)
        c.sight = (0.0, 0.0, 1.0)
        c.sample_rate, c.focus_dist, c.aperture, c.fov, c.aspect = 1, 1.0, 0.0, 45.0, 1.0
        c.look(_v(to))
        return c

    def _copy(self):
        c = Camera.__new__(Camera)
        c.__dict__.update(self.__dict__)
        return c

    def look(self, point):  # ray.rs:85-89
        self._to = _v(point)
        self.sight = v_unit(v_sub(_v(point), self.pos))
        right = self.right()
        self.up = v_unit(v_cross(right, self.sight))

    def right(self):  # ray.rs:97-99
        return v_unit(v_cross(self.sight, self.up))

    def with_sample_rate(self, rate):
        c = self._copy(); c.sample_rate = int(rate); return c

    def with_focus_dist(self, d):
        c = self._copy(); c.focus_dist = float(d); return c

    def with_aperture(self, a):
        c = self._copy(); c.aperture = float(a); return c

    def with_fov(self, deg):  # ray.rs:74-77
        c = self._copy(); c.fov = float(deg) / 180.0 * math.pi; return c

    def with_aspect(self, a):
        c = self._copy(); c.aspect = float(a); return c

    def _pod(self):
        pod = L.cftrt_camera()
        pod.pos[:] = self.pos
        pod.up[:] = self.up
        pod.sight[:] = self.sight
        pod.focus_dist, pod.aperture, pod.fov_rad, pod.aspect = self.focus_dist, self.aperture, self.fov, self.aspect
        return pod


def _dptr(a): return a.ctypes.data_as(C.POINTER(C.c_double))
def _fptr(a): return a.ctypes.data_as(C.POINTER(C.c_float))
def _iptr(a): return a.ctypes.data_as(C.POINTER(C.c_int32))


class Device:
    """One cftrt ctx: one GPU, one stream.  Not thread-safe (one per host thread / process)."""

    def __init__(self, index=0):
        self.lib = L.load()
        h = C.c_void_p()
        rc = self.lib.cftrt_ctx_create(int(index), C.byref(h))
        if rc != 0:
            raise L.CftrtError(rc, self.lib.cftrt_last_error(None).decode())
        self.h = h
        self.index = index
        self._worlds = weakref.WeakSet()   # worlds with a scene on this ctx: destroyed before the ctx
        # what the switches are set to (the C ABI has setters only): settings() / restore() for callers that change
        # them for a while
        self._set = {"pipeline": 0, "wave_cull": 1, "grid": 1, "tile_lists": 1, "shards": 1, "finish": (1, 1 << 20), "variant": 0,
                     "lds_grid": (1, 1 << 20), "walk_classes": (1, 0, 1 << 18)}

    def settings(self):
        return dict(self._set)

    def restore(self, st):
        self.set_pipeline(st["pipeline"]); self.set_wave_cull(st["wave_cull"]); self.set_grid(st["grid"])
        self.set_tile_lists(st["tile_lists"]); self.set_shards(st["shards"]); self.set_finish(*st["finish"])
        self.set_variant(st["variant"]); self.set_lds_grid(*st["lds_grid"])
        self.set_walk_classes(*st["walk_classes"])

    def check(self, rc):
        if rc != 0:
            raise L.CftrtError(rc, self.lib.cftrt_last_error(self.h).decode())

    def set_workspace_limit(self, nbytes):
        self.check(self.lib.cftrt_ctx_set_workspace_limit(self.h, int(nbytes)))

    def set_pipeline(self, split):
        """False/0 = fused k_trace_level per level (default), True/1 = k_intersect + k_shade with hit records."""
        self.check(self.lib.cftrt_ctx_set_pipeline(self.h, int(bool(split))))
        self._set["pipeline"] = int(bool(split))

    def set_wave_cull(self, on):
        """Wave-level sphere cull of the level kernels (default on); images are identical either way."""
        self.check(self.lib.cftrt_ctx_set_wave_cull(self.h, int(bool(on))))
        self._set["wave_cull"] = int(bool(on))

    def set_grid(self, mode, cells_per_prim=0.0):
        """Uniform grid (an extension to the reference's linear scan): 0 lists only, 1 (default) bounce rays walk it
        (primary rays too in scenes of more than 4096 primitives), 2 every ray; cells_per_prim > 0 sizes the grid
        of scenes committed afterwards.  Images are identical with any mode."""
        self.check(self.lib.cftrt_ctx_set_grid(self.h, int(mode), float(cells_per_prim)))
        self._set["grid"] = int(mode)

    def set_tile_lists(self, on):
        """Per-tile primitive lists for the primary rays of tiled frames (default on; same image either way)."""
        self.check(self.lib.cftrt_ctx_set_tile_lists(self.h, int(bool(on))))
        self._set["tile_lists"] = int(bool(on))

    def set_lds_grid(self, on, above=0):
        """Walk levels of sphere-only scenes whose grid fits keep it in LDS (default on, for levels of at least 2^20
        rays); same image either way."""
        self.check(self.lib.cftrt_ctx_set_lds_grid(self.h, int(bool(on)), int(above)))
        self._set["lds_grid"] = (int(bool(on)), int(above) if above else self._set["lds_grid"][1])

    def set_walk_classes(self, on, cells=0, above=0):
        """Survivors queued in two classes by the length of their next grid walk (default on; cells = 0: threshold from the
        grid's size); same image either way."""
        self.check(self.lib.cftrt_ctx_set_walk_classes(self.h, int(bool(on)), int(cells), int(above)))
        self._set["walk_classes"] = (int(bool(on)), int(cells), int(above) if above else self._set["walk_classes"][2])

    def set_shards(self, on):
        """Big trace levels append through 8 counters into a segmented queue (default on); same image either way."""
        self.check(self.lib.cftrt_ctx_set_shards(self.h, int(bool(on))))
        self._set["shards"] = int(bool(on))

    def set_finish(self, on, below=0):
        """Deep trace levels path by path in one launch (default on, below 2^20 live rays); same image either way."""
        self.check(self.lib.cftrt_ctx_set_finish(self.h, int(bool(on)), int(below)))
        self._set["finish"] = (int(bool(on)), int(below) if below else self._set["finish"][1])

    # -- multi-GPU without torch.distributed: RCCL through the C ABI
    def comm_unique_id(self):
        buf = (C.c_ubyte * 128)()
        rc = self.lib.cftrt_comm_unique_id(buf)
        if rc != 0:
            raise L.CftrtError(rc, self.lib.cftrt_last_error(None).decode())
        return bytes(buf)

    def comm_init(self, nranks, rank, unique_id):
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        self.check(self.lib.cftrt_comm_init(self.h, int(nranks), int(rank), buf))

    def reduce(self, accum_ptr, n_floats, root=0):
        self.check(self.lib.cftrt_reduce(self.h, C.c_void_p(accum_ptr), int(n_floats), int(root)))

    def comm_destroy(self):
        self.check(self.lib.cftrt_comm_destroy(self.h))

    def set_variant(self, v):
        self.check(self.lib.cftrt_ctx_set_variant(self.h, int(v)))
        self._set["variant"] = int(v)

    def close(self):
        if getattr(self, "h", None):
            for w in list(self._worlds):
                w._invalidate()
            self.lib.cftrt_ctx_destroy(self.h)
            self.h = None

    def generate(self, camera, width, height, pid0, n, seed, jitter=True):
        out = np.empty((n, 6), np.float32)
        pod = camera._pod()
        self.check(self.lib.cftrt_generate(self.h, C.byref(pod), width, height, pid0, n, seed, int(jitter), _fptr(out)))
        return out

    def resolve_rgb8(self, accum, divisor):
        accum = np.ascontiguousarray(accum, np.float32)
        npix = accum.size // 3
        out = np.empty(accum.shape, np.uint8)
        self.check(self.lib.cftrt_resolve_rgb8(self.h, _fptr(accum), npix, float(divisor),
                                               out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_device = None


def default_device():
    global _default_device
    if _default_device is None:
        _default_device = Device(0)
    return _default_device


class FlatScene:
    """The SoA arrays handed to cftrt_scene_set_* (also what the tests inspect)."""
    pass


class World:
    def __init__(self):
        self.objects = []
        self.lights = []
        self._scene = None
        self._dev = None

    @classmethod
    def empty(cls):
        return cls()

    def add_obj(self, obj):
        if not isinstance(obj, Object):
            raise TypeError("add_obj needs an Object")
        self.objects.append(obj)
        self._invalidate()

    def add_light(self, light):
        if not isinstance(light, (SkyLight, LightShape, PointLight, ParallelLight)):
            raise TypeError("user-defined LightSource impls have no device counterpart")
        self.lights.append(light)
        self._invalidate()

    def _invalidate(self):
        if self._scene is not None:
            if self._dev.h is not None:
                self._dev.lib.cftrt_scene_destroy(self._scene)
            self._scene = None

    # -- flattening: trait objects -> SoA primitive buffers + material table + light list
    def flatten(self):
        f = FlatScene()
        sc, sr, smv, sobj, smat = [], [], [], [], []
        tp0, tp1, tp2, tmv, tobj, tmat, tfirst = [], [], [], [], [], [], []
        mats = []
        for oi, o in enumerate(self.objects):
            mats.append(o.material._row())   # one table row per object (Arc<dyn Material> per object)
            sph, tri = o.shape._flatten()
            for c, r in sph:
                sc.append(c); sr.append(r); smv.append(o.moving_to); sobj.append(oi); smat.append(oi)
            pair = getattr(o.shape, "or_pair", False)   # Square.from_points: its second triangle is shadowed by its first
            for j, (a, b, c) in enumerate(tri):
                tfirst.append(len(tp0) - 1 if (pair and j == 1) else -1)
                tp0.append(a); tp1.append(b); tp2.append(c); tmv.append(o.moving_to); tobj.append(oi); tmat.append(oi)
        f.sph_center = np.array(sc, np.float64).reshape(-1, 3)
        f.sph_radius = np.array(sr, np.float64)
        f.sph_moving = np.array(smv, np.float64).reshape(-1, 3)
        f.sph_obj = np.array(sobj, np.int32)
        f.sph_mat = np.array(smat, np.int32)
        f.tri_p0 = np.array(tp0, np.float64).reshape(-1, 3)
        f.tri_p1 = np.array(tp1, np.float64).reshape(-1, 3)
        f.tri_p2 = np.array(tp2, np.float64).reshape(-1, 3)
        f.tri_moving = np.array(tmv, np.float64).reshape(-1, 3)
        f.tri_obj = np.array(tobj, np.int32)
        f.tri_mat = np.array(tmat, np.int32)
        f.tri_first = np.array(tfirst, np.int32)
        f.materials = mats
        lrows, lsc, lsr, lt0, lt1, lt2, ltfirst = [], [], [], [], [], [], []
        for l in self.lights:
            row = dict(kind=l.kind, sph_begin=0, sph_count=0, tri_begin=0, tri_count=0, v=(0.0, 0.0, 0.0),
                       color=(1.0, 1.0, 1.0))
            if l.kind == 1:
                sph, tri = l.shape._flatten()
                row.update(sph_begin=len(lsr), sph_count=len(sph), tri_begin=len(lt0), tri_count=len(tri))
                for c, r in sph:
                    lsc.append(c); lsr.append(r)
                pair = getattr(l.shape, "or_pair", False)
                for j, (a, b, c) in enumerate(tri):
                    ltfirst.append(len(lt0) - 1 if (pair and j == 1) else -1)
                    lt0.append(a); lt1.append(b); lt2.append(c)
            elif l.kind == 2:
                row.update(v=l.pos, color=l.light_color)
            elif l.kind == 3:
                row.update(v=l.dir, color=l.light_color)
            lrows.append(row)
        f.lights = lrows
        f.lsph_center = np.array(lsc, np.float64).reshape(-1, 3)
        f.lsph_radius = np.array(lsr, np.float64)
        f.ltri_p0 = np.array(lt0, np.float64).reshape(-1, 3)
        f.ltri_p1 = np.array(lt1, np.float64).reshape(-1, 3)
        f.ltri_p2 = np.array(lt2, np.float64).reshape(-1, 3)
        f.ltri_first = np.array(ltfirst, np.int32)
        return f

    def commit(self, device=None, precise_radius=None):
        """Uploads the flattened world to `device` (default: GPU 0).  Done lazily by render()."""
        dev = device or default_device()
        if self._scene is not None and self._dev is dev:
            return self._scene
        self._invalidate()
        lib = dev.lib
        f = self.flatten()
        h = C.c_void_p()
        dev.check(lib.cftrt_scene_create(dev.h, C.byref(h)))
        try:
            if precise_radius is not None:
                dev.check(lib.cftrt_scene_set_precise_radius(h, float(precise_radius)))
            dev.check(lib.cftrt_scene_set_spheres(h, _dptr(f.sph_center), _dptr(f.sph_radius), _dptr(f.sph_moving),
                                                  _iptr(f.sph_obj), _iptr(f.sph_mat), len(f.sph_radius)))
            dev.check(lib.cftrt_scene_set_triangles(h, _dptr(f.tri_p0), _dptr(f.tri_p1), _dptr(f.tri_p2),
                                                    _dptr(f.tri_moving), _iptr(f.tri_obj), _iptr(f.tri_mat),
                                                    len(f.tri_obj)))
            mt = (L.cftrt_material * max(len(f.materials), 1))()
            for i, (kind, p0, p1, col) in enumerate(f.materials):
                mt[i].kind, mt[i].p0, mt[i].p1 = kind, p0, p1
                mt[i].color[:] = col
            dev.check(lib.cftrt_scene_set_materials(h, mt, len(f.materials)))
            lt = (L.cftrt_light * max(len(f.lights), 1))()
            for i, r in enumerate(f.lights):
                lt[i].kind = r["kind"]
                lt[i].sph_begin, lt[i].sph_count = r["sph_begin"], r["sph_count"]
                lt[i].tri_begin, lt[i].tri_count = r["tri_begin"], r["tri_count"]
                lt[i].v[:] = r["v"]
                lt[i].color[:] = r["color"]
            dev.check(lib.cftrt_scene_set_lights(h, lt, len(f.lights), _dptr(f.lsph_center), _dptr(f.lsph_radius),
                                                 len(f.lsph_radius), _dptr(f.ltri_p0), _dptr(f.ltri_p1),
                                                 _dptr(f.ltri_p2), len(f.ltri_p0)))
            if (f.tri_first >= 0).any() or (f.ltri_first >= 0).any():
                dev.check(lib.cftrt_scene_set_triangle_pairs(h, _iptr(f.tri_first), len(f.tri_first),
                                                             _iptr(f.ltri_first), len(f.ltri_first)))
            dev.check(lib.cftrt_scene_commit(h))
        except Exception:
            lib.cftrt_scene_destroy(h)
            raise
        self._scene, self._dev = h, dev
        dev._worlds.add(self)
        return h

    # -- the batched entry point: the loop scene.rs:91-102 / cornell.rs:32-39 hand-roll
    def render(self, camera, width, height, depth, seed=0x5EED0002, spp_begin=0, spp_end=None, device=None,
               accum=None, return_stats=False):
        """Returns accum (H, W, 3) float32 = sum over samples [spp_begin, spp_end) of World::trace.
        Divide by the sample count (scene.rs:114) to get the image."""
        h = self.commit(device)
        dev = self._dev
        if spp_end is None:
            spp_end = spp_begin + camera.sample_rate
        if accum is None:
            accum = np.zeros((height, width, 3), np.float32)
        assert accum.dtype == np.float32 and accum.size == width * height * 3 and accum.flags.c_contiguous
        st = L.cftrt_stats()
        pod = camera._pod()
        dev.check(dev.lib.cftrt_render(h, C.byref(pod), width, height, spp_begin, spp_end, depth, seed,
                                       _fptr(accum), C.byref(st)))
        return (accum, st) if return_stats else accum

    def render_device(self, camera, width, height, depth, accum_ptr, seed=0x5EED0002, spp_begin=0, spp_end=None,
                      device=None, static=False):
        """Same, accumulating into a device buffer (e.g. torch tensor .data_ptr()) of W*H*3 float32.
        static=True: cftrt_render_device_static, the entry point that allocates nothing (256 MiB static workspace;
        one sample of the image must fit)."""
        h = self.commit(device)
        dev = self._dev
        if spp_end is None:
            spp_end = spp_begin + camera.sample_rate
        st = L.cftrt_stats()
        pod = camera._pod()
        fn = dev.lib.cftrt_render_device_static if static else dev.lib.cftrt_render_device
        dev.check(fn(h, C.byref(pod), width, height, spp_begin, spp_end, depth, seed, C.c_void_p(accum_ptr), C.byref(st)))
        return st

    # -- parity hooks
    def intersect(self, rays, seed=0, pid0=0, level=0, role=0, device=None, with_outward=False):
        """Ray::hit for n x 6 float32 rays -> (object id, t f64, shading normal, hit point[, outward flag])."""
        h = self.commit(device)
        dev = self._dev
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = len(rays)
        ids = np.empty(n, np.int32)
        t = np.empty(n, np.float64)
        nrm = np.empty((n, 3), np.float32)
        pt = np.empty((n, 3), np.float32)
        outward = np.zeros(n, np.int32)
        dev.check(dev.lib.cftrt_intersect(h, _fptr(rays), n, seed, pid0, level, role, _iptr(ids), _dptr(t),
                                          _fptr(nrm), _fptr(pt), _iptr(outward)))
        return (ids, t, nrm, pt, outward) if with_outward else (ids, t, nrm, pt)

    def level_hits(self, rays=None, camera=None, width=0, height=0, n=None, seed=0, pid0=0, level=0, device=None):
        """Ray::hit as the production level kernel finds it (cftrt_level_hits): one launch of the fused level kernel at
        trace level `level`, on caller rays (n x 6) or, with rays=None, on the primary rays it generates itself for
        whole samples of the image.  Returns (object ids, f64 distances), each n x scans."""
        h = self.commit(device)
        dev = self._dev
        if rays is not None:
            rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
            n = len(rays)
        pod = camera._pod() if camera is not None else None
        ids = np.empty((n, 2), np.int32)
        t = np.empty((n, 2), np.float64)
        scans = C.c_int32(0)
        dev.check(dev.lib.cftrt_level_hits(h, C.byref(pod) if pod is not None else None, int(width), int(height),
                                           _fptr(rays) if rays is not None else None, n, seed, pid0, level,
                                           _iptr(ids), _dptr(t), C.byref(scans)))
        k = scans.value
        return ids.reshape(-1)[: n * k].reshape(n, k).copy(), t.reshape(-1)[: n * k].reshape(n, k).copy()

    def trace_paths(self, camera, width, height, pid0, n, depth, seed=0x5EED0002, jitter=True, device=None):
        h = self.commit(device)
        dev = self._dev
        out = np.empty((n, 3), np.float32)
        st = L.cftrt_stats()
        pod = camera._pod()
        dev.check(dev.lib.cftrt_trace_paths(h, C.byref(pod), width, height, pid0, n, depth, seed, int(jitter),
                                            _fptr(out), C.byref(st)))
        return out, st

    def __del__(self):
        try:
            self._invalidate()
        except Exception:
            pass


def vec3_to_rgb(accum, divisor, device=None):
    """scene.rs:122-127 on the device: u8 = trunc(255.99*sqrt(clamp(c/divisor,0,1)))."""
    return (device or default_device()).resolve_rgb8(accum, divisor)
