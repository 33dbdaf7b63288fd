//! `src/gpu.rs` of `cft_ray_tracer` — the batched GPU entry point on `World`.
//!
//! UNVERIFIED: never compiled (no Rust toolchain in the build image).  It lives INSIDE the crate because it reads
//! private fields (`Sphere::center`, `Camera::fov`, `Metal::fuzz`, ...; `src/object.rs:258-261`, `src/ray.rs:47-56`,
//! `src/material/compose.rs:10-14`).  It adds, and changes nothing else:
//!   * one defaulted `gpu_desc` method on each of `Shape`, `Material`, `LightSource` (user impls keep the default
//!     `None` and their worlds stay on the CPU loop);
//!   * `World::flatten` — the arrays `cftrt_scene_set_*` take;
//!   * `World::render(&Camera, width, height, depth) -> Vec<Color>` — the loop `examples/scene.rs:91-102` and
//!     `examples/cornell.rs:32-39` hand-roll; `World::render_sharded` — the same over several GPUs.
//! The per-ray API (`Camera::emit_rays`, `World::trace`, `Ray::hit`) stays as it is.
//!
//! Cargo.toml of the crate gains:  cftrt-sys = { path = "../cftrt-sys", optional = true }   and   [features] gpu = ["cftrt-sys"]
use crate::{
    light::{LightShape, ParallelLight, PointLight, SkyLight},
    material::{Dielectric, LambertianModel, Metal, PhongModel, Specular, Transparent},
    object::{Cube, Sphere, Square, Triangle, World},
    ray::Camera,
    util::{Color, Vec3},
};
use cftrt_sys::*;
use std::ffi::CStr;

/// What a built-in shape is to the device: a sphere, or triangles in the order the reference tests them
/// (`Square::new`: tri0 = (p0,p1,p2), tri1 = (p2,p3,p0), `object.rs:147-162`; `Cube::squares`: +x, -x, +y, -y, +z, -z,
/// `object.rs:215-230`).  "First minimum wins" (`ray.rs:23-27`) depends on that order.
pub enum GpuShape {
    Sphere { center: Vec3, radius: f64 },
    Triangles(Vec<[Vec3; 3]>),
    /// The two triangles of a `Square::from_points` (`object.rs:169-174`): `hit_info` is `tri0.or(tri1)`
    /// (`object.rs:192`), so the second counts only where the first is not hit - flagged for the device with
    /// `cftrt_scene_set_triangle_pairs` (a `Square::new` is planar and shares only its diagonal: plain `Triangles`).
    OrPair([[Vec3; 3]; 2]),
}

// ---- the three defaulted trait methods (added to the trait definitions in object.rs:13, material.rs:12, light.rs:7)
//
//     pub trait Shape: Sync + Send       { ...; fn gpu_desc(&self) -> Option<GpuShape> { None } }
//     pub trait Material: Sync + Send    { ...; fn gpu_desc(&self) -> Option<cftrt_material> { None } }
//     pub trait LightSource: Sync + Send { ...; fn gpu_desc(&self) -> Option<(cftrt_light, Option<GpuShape>)> { None } }
//
// and these bodies go into the existing `impl Shape for ...` / `impl Material for ...` / `impl LightSource for ...`
// blocks of the built-ins (written here as inherent helpers so that this file is one unit):

fn v3(v: Vec3) -> [f64; 3] {
    [v.x, v.y, v.z]
}

impl Sphere {
    pub(crate) fn gpu_desc_impl(&self) -> Option<GpuShape> {
        Some(GpuShape::Sphere { center: self.center, radius: self.radius }) // a negative radius flips the normal (object.rs:288-292)
    }
}
impl Triangle {
    pub(crate) fn gpu_desc_impl(&self) -> Option<GpuShape> {
        Some(GpuShape::Triangles(vec![[self.p0, self.p1, self.p2]]))
    }
}
impl Square {
    pub(crate) fn gpu_desc_impl(&self) -> Option<GpuShape> {
        // hit_info returns tri0's hit even where tri1's is nearer (object.rs:190-196).  For a Square::new - tri1 =
        // (tri0.p2, p3, tri0.p0) in tri0's plane - that is the closest of the two, the device's ordinary rule; anything
        // else (from_points: tri1 = (p1, p2, p3), overlapping or out of plane) goes as a flagged pair.
        let t0 = [self.tri0.p0, self.tri0.p1, self.tri0.p2];
        let t1 = [self.tri1.p0, self.tri1.p1, self.tri1.p2];
        let coplanar = (self.tri0.normal().cross(self.tri1.normal())).len() <= 1e-9;
        let shares_diagonal = (self.tri1.p0 - self.tri0.p2).len() == 0.0 && (self.tri1.p2 - self.tri0.p0).len() == 0.0;
        if coplanar && shares_diagonal {
            Some(GpuShape::Triangles(vec![t0, t1]))
        } else {
            Some(GpuShape::OrPair([t0, t1]))
        }
    }
}
impl Cube {
    pub(crate) fn gpu_desc_impl(&self) -> Option<GpuShape> {
        let mut tris = Vec::with_capacity(12);
        for sq in self.squares() {
            // same faces, same order as hit_info iterates them (object.rs:233-254)
            tris.push([sq.tri0.p0, sq.tri0.p1, sq.tri0.p2]);
            tris.push([sq.tri1.p0, sq.tri1.p1, sq.tri1.p2]);
        }
        Some(GpuShape::Triangles(tris))
    }
}

fn mat(kind: i32, p0: f64, p1: f64, color: Color) -> Option<cftrt_material> {
    Some(cftrt_material { kind, reserved: 0, p0, p1, color: v3(color) })
}
impl LambertianModel {
    pub(crate) fn gpu_desc_impl(&self) -> Option<cftrt_material> {
        mat(CFTRT_MAT_LAMBERTIAN, self.s.albedo, 0.0, self.c)
    }
}
impl Metal {
    pub(crate) fn gpu_desc_impl(&self) -> Option<cftrt_material> {
        mat(CFTRT_MAT_METAL, self.fuzz, self.s.albedo, self.color)
    }
}
impl Dielectric {
    pub(crate) fn gpu_desc_impl(&self) -> Option<cftrt_material> {
        mat(CFTRT_MAT_DIELECTRIC, self.r.ior, 0.0, Vec3::new(1.0, 1.0, 1.0))
    }
}
impl PhongModel {
    pub(crate) fn gpu_desc_impl(&self) -> Option<cftrt_material> {
        mat(CFTRT_MAT_PHONG, self.shininess, self.diffuse, self.color)
    }
}
impl Specular {
    pub(crate) fn gpu_desc_impl(&self) -> Option<cftrt_material> {
        mat(CFTRT_MAT_SPECULAR, self.albedo, 0.0, Vec3::new(1.0, 1.0, 1.0))
    }
}
impl Transparent {
    pub(crate) fn gpu_desc_impl(&self) -> Option<cftrt_material> {
        mat(CFTRT_MAT_TRANSPARENT, self.opacity, self.ior, self.color)
    }
}

fn light(kind: i32, v: Vec3, color: Color) -> cftrt_light {
    cftrt_light { kind, sph_begin: 0, sph_count: 0, tri_begin: 0, tri_count: 0, reserved: 0, v: v3(v), color: v3(color) }
}
impl SkyLight {
    pub(crate) fn gpu_desc_impl(&self) -> Option<(cftrt_light, Option<GpuShape>)> {
        Some((light(CFTRT_LIGHT_SKY, Vec3::new(0.0, 0.0, 0.0), Vec3::new(1.0, 1.0, 1.0)), None))
    }
}
impl PointLight {
    pub(crate) fn gpu_desc_impl(&self) -> Option<(cftrt_light, Option<GpuShape>)> {
        Some((light(CFTRT_LIGHT_POINT, self.pos, self.light_color), None))
    }
}
impl ParallelLight {
    pub(crate) fn gpu_desc_impl(&self) -> Option<(cftrt_light, Option<GpuShape>)> {
        Some((light(CFTRT_LIGHT_PARALLEL, self.dir, self.light_color), None)) // dir is used un-normalised (light.rs:66-102)
    }
}
impl LightShape {
    pub(crate) fn gpu_desc_impl(&self) -> Option<(cftrt_light, Option<GpuShape>)> {
        let shape = self.shape.gpu_desc()?;
        Some((light(CFTRT_LIGHT_SHAPE, Vec3::new(0.0, 0.0, 0.0), Vec3::new(1.0, 1.0, 1.0)), Some(shape))) // colour fixed (1,1,1): light.rs:233-242
    }
}

/// The world as the arrays of `cftrt_scene_set_*`.  Object ids are indices into `world.objects` (non-decreasing in
/// both primitive lists); the tie-break "lowest object index wins" is what `min_by` gives the reference.
#[derive(Default)]
pub struct FlatWorld {
    pub sph_center: Vec<f64>,
    pub sph_radius: Vec<f64>,
    pub sph_moving: Vec<f64>,
    pub sph_obj: Vec<i32>,
    pub sph_mat: Vec<i32>,
    pub tri_p0: Vec<f64>,
    pub tri_p1: Vec<f64>,
    pub tri_p2: Vec<f64>,
    pub tri_moving: Vec<f64>,
    pub tri_obj: Vec<i32>,
    pub tri_mat: Vec<i32>,
    pub tri_first: Vec<i32>,  // per triangle: i - 1 for the second triangle of a Square::from_points, else -1
    pub ltri_first: Vec<i32>, // the same over the light-shape triangles
    pub materials: Vec<cftrt_material>,
    pub lights: Vec<cftrt_light>,
    pub lsph_center: Vec<f64>,
    pub lsph_radius: Vec<f64>,
    pub ltri_p0: Vec<f64>,
    pub ltri_p1: Vec<f64>,
    pub ltri_p2: Vec<f64>,
}

fn push3(dst: &mut Vec<f64>, v: Vec3) {
    dst.extend_from_slice(&[v.x, v.y, v.z]);
}

impl World {
    /// `None` as soon as one shape, material or light has no device form (a user impl).
    pub fn flatten(&self) -> Option<FlatWorld> {
        let mut f = FlatWorld::default();
        for (i, obj) in self.objects.iter().enumerate() {
            let m = obj.material.gpu_desc()?;
            // one row per object; identical rows could be merged, the table is tiny either way
            let mat_id = f.materials.len() as i32;
            f.materials.push(m);
            match obj.shape.gpu_desc()? {
                GpuShape::Sphere { center, radius } => {
                    push3(&mut f.sph_center, center);
                    f.sph_radius.push(radius);
                    push3(&mut f.sph_moving, obj.moving_to);
                    f.sph_obj.push(i as i32);
                    f.sph_mat.push(mat_id);
                }
                shape @ GpuShape::Triangles(_) | shape @ GpuShape::OrPair(_) => {
                    let (tris, pair) = match shape {
                        GpuShape::Triangles(t) => (t, false),
                        GpuShape::OrPair(t) => (t.to_vec(), true),
                        _ => unreachable!(),
                    };
                    for (j, t) in tris.into_iter().enumerate() {
                        f.tri_first.push(if pair && j == 1 { f.tri_obj.len() as i32 - 1 } else { -1 });
                        push3(&mut f.tri_p0, t[0]);
                        push3(&mut f.tri_p1, t[1]);
                        push3(&mut f.tri_p2, t[2]);
                        push3(&mut f.tri_moving, obj.moving_to);
                        f.tri_obj.push(i as i32);
                        f.tri_mat.push(mat_id);
                    }
                }
            }
        }
        for l in self.lights.iter() {
            let (mut row, shape) = l.gpu_desc()?;
            match shape {
                None => {}
                Some(GpuShape::Sphere { center, radius }) => {
                    row.sph_begin = f.lsph_radius.len() as i32;
                    row.sph_count = 1;
                    push3(&mut f.lsph_center, center);
                    f.lsph_radius.push(radius);
                }
                Some(shape @ GpuShape::Triangles(_)) | Some(shape @ GpuShape::OrPair(_)) => {
                    let (tris, pair) = match shape {
                        GpuShape::Triangles(t) => (t, false),
                        GpuShape::OrPair(t) => (t.to_vec(), true),
                        _ => unreachable!(),
                    };
                    row.tri_begin = (f.ltri_p0.len() / 3) as i32;
                    row.tri_count = tris.len() as i32;
                    for (j, t) in tris.into_iter().enumerate() {
                        f.ltri_first.push(if pair && j == 1 { (f.ltri_p0.len() / 3) as i32 - 1 } else { -1 });
                        push3(&mut f.ltri_p0, t[0]);
                        push3(&mut f.ltri_p1, t[1]);
                        push3(&mut f.ltri_p2, t[2]);
                    }
                }
            }
            f.lights.push(row);
        }
        Some(f)
    }
}

/// Error of the GPU path: the status code of `include/cftrt.h` and the text of `cftrt_last_error`.
#[derive(Debug)]
pub struct GpuError {
    pub code: i32,
    pub text: String,
}

unsafe fn check(ctx: *const cftrt_ctx, rc: i32) -> Result<(), GpuError> {
    if rc == CFTRT_OK {
        return Ok(());
    }
    let text = CStr::from_ptr(cftrt_last_error(ctx)).to_string_lossy().into_owned();
    Err(GpuError { code: rc, text })
}

fn ptr_or_null<T>(v: &[T]) -> *const T {
    if v.is_empty() { std::ptr::null() } else { v.as_ptr() }
}

/// One GPU's context and the committed scene; scenes are destroyed before their ctx (cftrt.h, "Ownership").
pub struct GpuScene {
    ctx: *mut cftrt_ctx,
    scene: *mut cftrt_scene,
}

impl Drop for GpuScene {
    fn drop(&mut self) {
        unsafe {
            if !self.scene.is_null() {
                cftrt_scene_destroy(self.scene);
            }
            if !self.ctx.is_null() {
                cftrt_ctx_destroy(self.ctx);
            }
        }
    }
}

impl GpuScene {
    pub fn upload(flat: &FlatWorld, device: i32) -> Result<GpuScene, GpuError> {
        unsafe {
            let mut g = GpuScene { ctx: std::ptr::null_mut(), scene: std::ptr::null_mut() };
            check(std::ptr::null(), cftrt_ctx_create(device, &mut g.ctx))?;
            check(g.ctx, cftrt_scene_create(g.ctx, &mut g.scene))?;
            check(g.ctx, cftrt_scene_set_spheres(g.scene, ptr_or_null(&flat.sph_center), ptr_or_null(&flat.sph_radius),
                ptr_or_null(&flat.sph_moving), ptr_or_null(&flat.sph_obj), ptr_or_null(&flat.sph_mat),
                flat.sph_radius.len() as i64))?;
            check(g.ctx, cftrt_scene_set_triangles(g.scene, ptr_or_null(&flat.tri_p0), ptr_or_null(&flat.tri_p1),
                ptr_or_null(&flat.tri_p2), ptr_or_null(&flat.tri_moving), ptr_or_null(&flat.tri_obj),
                ptr_or_null(&flat.tri_mat), flat.tri_obj.len() as i64))?;
            check(g.ctx, cftrt_scene_set_materials(g.scene, ptr_or_null(&flat.materials), flat.materials.len() as i64))?;
            check(g.ctx, cftrt_scene_set_lights(g.scene, ptr_or_null(&flat.lights), flat.lights.len() as i64,
                ptr_or_null(&flat.lsph_center), ptr_or_null(&flat.lsph_radius), flat.lsph_radius.len() as i64,
                ptr_or_null(&flat.ltri_p0), ptr_or_null(&flat.ltri_p1), ptr_or_null(&flat.ltri_p2),
                (flat.ltri_p0.len() / 3) as i64))?;
            if flat.tri_first.iter().chain(flat.ltri_first.iter()).any(|&x| x >= 0) {
                check(g.ctx, cftrt_scene_set_triangle_pairs(g.scene, ptr_or_null(&flat.tri_first), flat.tri_first.len() as i64,
                    ptr_or_null(&flat.ltri_first), flat.ltri_first.len() as i64))?;
            }
            check(g.ctx, cftrt_scene_commit(g.scene))?;
            Ok(g)
        }
    }

    /// accum[(h*W+w)*3+c] += sum over samples [spp_begin, spp_end) of World::trace — HOST buffer.
    pub fn render_into(&self, cam: &cftrt_camera, width: u32, height: u32, spp_begin: u32, spp_end: u32, depth: u32,
                       seed: u64, accum: &mut [f32]) -> Result<(), GpuError> {
        assert_eq!(accum.len(), width as usize * height as usize * 3);
        unsafe {
            check(self.ctx, cftrt_render(self.scene, cam, width, height, spp_begin, spp_end, depth, seed,
                                         accum.as_mut_ptr(), std::ptr::null_mut()))
        }
    }
}

fn camera_pod(c: &Camera) -> cftrt_camera {
    cftrt_camera {
        pos: v3(c.pos),
        up: v3(c.up()),
        sight: v3(c.sight()),
        focus_dist: c.focus_dist,
        aperture: c.aperture,
        fov_rad: c.fov, // stored in radians by with_fov; the default is 45 *radians* (ray.rs:147)
        aspect: c.aspect,
    }
}

impl World {
    /// The loop `scene.rs:91-102` / `cornell.rs:32-39` hand-roll:
    /// `result[h*width + w]` = SUM over `camera.sample_rate` samples of `self.trace(ray, depth)` (divide by the sample
    /// rate as `scene.rs:114` does); `h = 0` is the top row.  Worlds with a user-defined shape, material or light, and
    /// hosts without a usable GPU, take the crate's own per-ray loop.
    pub fn render(&self, camera: &Camera, width: u64, height: u64, depth: u64) -> Vec<Color> {
        match self.render_gpu(camera, width, height, depth, 0) {
            Ok(Some(img)) => img,
            Ok(None) => self.render_cpu(camera, width, height, depth),
            Err(e) => {
                eprintln!("cftrt: {} ({}); rendering on the CPU", e.text, e.code);
                self.render_cpu(camera, width, height, depth)
            }
        }
    }

    /// `Ok(None)`: the world has no device form.  The seed keys the Philox streams (the reference's `thread_rng` is
    /// unseeded; only the distributions are its own).
    pub fn render_gpu(&self, camera: &Camera, width: u64, height: u64, depth: u64, seed: u64)
                      -> Result<Option<Vec<Color>>, GpuError> {
        let flat = match self.flatten() {
            Some(f) => f,
            None => return Ok(None),
        };
        let g = GpuScene::upload(&flat, 0)?; // cached on the World behind a OnceCell in a long-lived host
        let mut acc = vec![0f32; (width * height * 3) as usize];
        g.render_into(&camera_pod(camera), width as u32, height as u32, 0, camera.sample_rate as u32, depth as u32,
                      seed, &mut acc)?;
        Ok(Some(acc.chunks(3).map(|c| Vec3::new(c[0] as f64, c[1] as f64, c[2] as f64)).collect()))
    }

    /// The crate's loop as the examples write it today, one thread (the examples' own thread pool stays theirs).
    fn render_cpu(&self, camera: &Camera, width: u64, height: u64, depth: u64) -> Vec<Color> {
        let mut raw = vec![Vec3::new(0.0, 0.0, 0.0); (width * height) as usize];
        for (w, h, ray) in camera.emit_rays(width, height) {
            raw[(h * width + w) as usize] += self.trace(&ray, depth); // scene.rs:95-97
        }
        raw
    }

    /// Several GPUs of one node from ONE process, one host thread per GPU: GPU g renders samples
    /// [g*spp/n, (g+1)*spp/n) of every pixel into its own device image (`cftrt_render_device`) and the images are
    /// summed into GPU 0's with `cftrt_reduce` (RCCL over xGMI).  The path id of (sample, pixel) keys the random
    /// stream, so the sum equals the one-GPU image up to f32 summation order.  Device buffers come from the HIP
    /// runtime (`hipMalloc` via a hip-sys crate of the host's choice): `alloc`/`download`/`free` stand for them.
    pub fn render_sharded<A, D, F>(&self, camera: &Camera, width: u64, height: u64, depth: u64, seed: u64, n_gpus: i32,
                                   alloc: A, download: D, free: F) -> Result<Option<Vec<Color>>, GpuError>
    where
        A: Fn(i32, usize) -> *mut std::os::raw::c_void + Sync,
        D: Fn(i32, *mut std::os::raw::c_void, &mut [f32]),
        F: Fn(i32, *mut std::os::raw::c_void) + Sync,
    {
        let flat = match self.flatten() {
            Some(f) => f,
            None => return Ok(None),
        };
        let cam = camera_pod(camera);
        let spp = camera.sample_rate as u32;
        let n_floats = (width * height * 3) as usize;
        let mut id = [0u8; CFTRT_COMM_ID_BYTES];
        unsafe { check(std::ptr::null(), cftrt_comm_unique_id(id.as_mut_ptr()))? };
        let mut out = vec![0f32; n_floats];
        // Phase 1, every GPU: upload the scene and take the image buffer.  Only when ALL of them succeeded does any rank
        // enter the communicator: a rank that failed here would leave the others waiting in ncclCommInitRank for ever.
        let setups: Vec<Result<(GpuScene, usize), GpuError>> = std::thread::scope(|s| {
            let handles: Vec<_> = (0..n_gpus)
                .map(|g| {
                    let (flat, alloc) = (&flat, &alloc);
                    s.spawn(move || -> Result<(GpuScene, usize), GpuError> {
                        let sc = GpuScene::upload(flat, g)?;
                        let buf = alloc(g, n_floats * 4); // zero-filled device memory on GPU g
                        if buf.is_null() {
                            return Err(GpuError { code: CFTRT_ERR_HIP, text: format!("device image allocation failed on GPU {}", g) });
                        }
                        Ok((sc, buf as usize))
                    })
                })
                .collect();
            handles.into_iter().map(|h| h.join().unwrap()).collect()
        });
        let release = |ranks: &[(GpuScene, usize)]| {
            for (g, (_, buf)) in ranks.iter().enumerate() {
                free(g as i32, *buf as *mut std::os::raw::c_void);
            }
        };
        let mut ranks: Vec<(GpuScene, usize)> = Vec::new();
        let mut first_err = None;
        for r in setups {
            match r {
                Ok(x) => ranks.push(x),
                Err(e) => { if first_err.is_none() { first_err = Some(e); } }
            }
        }
        if let Some(e) = first_err {
            release(&ranks); // (buffers of the ranks that did come up; their GpuScene drops free ctx and scene)
            return Err(e);
        }
        // Phase 2, every GPU: communicator, render, reduce into GPU 0's image.
        let results: Vec<Result<(), GpuError>> = std::thread::scope(|s| {
            let handles: Vec<_> = ranks
                .iter()
                .enumerate()
                .map(|(g, (sc, buf))| {
                    let (cam, id) = (&cam, &id);
                    let (g, buf) = (g as i32, *buf as *mut std::os::raw::c_void);
                    s.spawn(move || -> Result<(), GpuError> {
                        let (b, e) = (spp * g as u32 / n_gpus as u32, spp * (g as u32 + 1) / n_gpus as u32);
                        unsafe {
                            check(sc.ctx, cftrt_comm_init(sc.ctx, n_gpus, g, id.as_ptr()))?;
                            // from here on every rank makes the same collective calls, whatever its own render returned
                            let r = check(sc.ctx, cftrt_render_device(sc.scene, cam, width as u32, height as u32, b, e,
                                                                      depth as u32, seed, buf, std::ptr::null_mut()));
                            let red = check(sc.ctx, cftrt_reduce(sc.ctx, buf, n_floats as u64, 0));
                            let _ = cftrt_comm_destroy(sc.ctx);
                            r.and(red)
                        }
                    })
                })
                .collect();
            handles.into_iter().map(|h| h.join().unwrap()).collect()
        });
        if let Some(e) = results.into_iter().find_map(|r| r.err()) {
            release(&ranks);
            return Err(e);
        }
        let root = ranks[0].1;
        download(0, root as *mut std::os::raw::c_void, &mut out);
        release(&ranks);
        Ok(Some(out.chunks(3).map(|c| Vec3::new(c[0] as f64, c[1] as f64, c[2] as f64)).collect()))
    }
}

// The two examples rewritten to the one call: rust/examples/scene.rs and rust/examples/cornell.rs (text, unverified).
