//! Raw bindings to `libcftrt.so` (`include/cftrt.h`).  UNVERIFIED: written by hand against the header, never
//! compiled (the build image has no Rust toolchain); `scripts/check_rust_ffi.py` keeps it in step with the header.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_uchar, c_void};

pub const CFTRT_OK: c_int = 0;
pub const CFTRT_ERR_INVALID: c_int = 1;
pub const CFTRT_ERR_HIP: c_int = 2;
pub const CFTRT_ERR_STATE: c_int = 3;
pub const CFTRT_ERR_LIMIT: c_int = 4;

pub const CFTRT_MAT_LAMBERTIAN: i32 = 0;
pub const CFTRT_MAT_METAL: i32 = 1;
pub const CFTRT_MAT_DIELECTRIC: i32 = 2;
pub const CFTRT_MAT_PHONG: i32 = 3;
pub const CFTRT_MAT_SPECULAR: i32 = 4;
pub const CFTRT_MAT_TRANSPARENT: i32 = 5;

pub const CFTRT_LIGHT_SKY: i32 = 0;
pub const CFTRT_LIGHT_SHAPE: i32 = 1;
pub const CFTRT_LIGHT_POINT: i32 = 2;
pub const CFTRT_LIGHT_PARALLEL: i32 = 3;

pub const CFTRT_MAX_LEVELS: usize = 256;
pub const CFTRT_STATIC_SCRATCH_BYTES: u32 = 256u32 << 20;
pub const CFTRT_COMM_ID_BYTES: usize = 128;

#[repr(C)]
pub struct cftrt_ctx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct cftrt_scene {
    _p: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct cftrt_material {
    pub kind: i32,
    pub reserved: i32,
    pub p0: f64,
    pub p1: f64,
    pub color: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct cftrt_light {
    pub kind: i32,
    pub sph_begin: i32,
    pub sph_count: i32,
    pub tri_begin: i32,
    pub tri_count: i32,
    pub reserved: i32,
    pub v: [f64; 3],
    pub color: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct cftrt_camera {
    pub pos: [f64; 3],
    pub up: [f64; 3],
    pub sight: [f64; 3],
    pub focus_dist: f64,
    pub aperture: f64,
    pub fov_rad: f64,
    pub aspect: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct cftrt_stats {
    pub paths: u64,
    pub rays: u64,
    pub phong_rays: u64,
    pub exact_second: u64,
    pub exact_rescan: u64,
    pub flops: f64,
    pub flops_intersect: f64,
    pub ms_total: f64,
    pub ms_generate: f64,
    pub ms_intersect: f64,
    pub ms_shade: f64,
    pub ms_shadow: f64,
    pub ms_resolve: f64,
    pub launches_intersect: u32,
    pub levels: u32,
    pub passes: u32,
    pub reserved: u32,
    pub rays_per_level: [u64; 256],
}

extern "C" {
    // ---- context
    pub fn cftrt_device_count(out_n: *mut c_int) -> c_int;
    pub fn cftrt_ctx_create(device: c_int, out: *mut *mut cftrt_ctx) -> c_int;
    pub fn cftrt_ctx_destroy(ctx: *mut cftrt_ctx);
    pub fn cftrt_last_error(ctx: *const cftrt_ctx) -> *const c_char;
    pub fn cftrt_ctx_set_workspace_limit(ctx: *mut cftrt_ctx, bytes: u64) -> c_int;
    pub fn cftrt_ctx_set_pipeline(ctx: *mut cftrt_ctx, split: c_int) -> c_int;
    pub fn cftrt_ctx_set_wave_cull(ctx: *mut cftrt_ctx, on: c_int) -> c_int;
    pub fn cftrt_ctx_set_grid(ctx: *mut cftrt_ctx, mode: c_int, cells_per_prim: f64) -> c_int;
    pub fn cftrt_ctx_set_shards(ctx: *mut cftrt_ctx, on: c_int) -> c_int;
    pub fn cftrt_ctx_set_tile_lists(ctx: *mut cftrt_ctx, on: c_int) -> c_int;
    pub fn cftrt_ctx_set_lds_grid(ctx: *mut cftrt_ctx, on: c_int, above: u32) -> c_int;
    pub fn cftrt_ctx_set_walk_classes(ctx: *mut cftrt_ctx, on: c_int, cells: u32, above: u32) -> c_int;
    pub fn cftrt_ctx_set_finish(ctx: *mut cftrt_ctx, on: c_int, below: u32) -> c_int;
    pub fn cftrt_ctx_set_variant(ctx: *mut cftrt_ctx, variant: c_int) -> c_int;
    // ---- scene
    pub fn cftrt_scene_create(ctx: *mut cftrt_ctx, out: *mut *mut cftrt_scene) -> c_int;
    pub fn cftrt_scene_destroy(scene: *mut cftrt_scene);
    pub fn cftrt_scene_set_spheres(
        scene: *mut cftrt_scene,
        center: *const f64,
        radius: *const f64,
        moving_to: *const f64,
        obj_id: *const i32,
        mat_id: *const i32,
        n: i64,
    ) -> c_int;
    pub fn cftrt_scene_set_triangles(
        scene: *mut cftrt_scene,
        p0: *const f64,
        p1: *const f64,
        p2: *const f64,
        moving_to: *const f64,
        obj_id: *const i32,
        mat_id: *const i32,
        n: i64,
    ) -> c_int;
    pub fn cftrt_scene_set_materials(scene: *mut cftrt_scene, table: *const cftrt_material, n: i64) -> c_int;
    pub fn cftrt_scene_set_lights(
        scene: *mut cftrt_scene,
        lights: *const cftrt_light,
        n_lights: i64,
        lsph_center: *const f64,
        lsph_radius: *const f64,
        n_lsph: i64,
        ltri_p0: *const f64,
        ltri_p1: *const f64,
        ltri_p2: *const f64,
        n_ltri: i64,
    ) -> c_int;
    pub fn cftrt_scene_set_triangle_pairs(
        scene: *mut cftrt_scene,
        tri_first: *const i32,
        n_tri: i64,
        ltri_first: *const i32,
        n_ltri: i64,
    ) -> c_int;
    pub fn cftrt_scene_set_precise_radius(scene: *mut cftrt_scene, radius: f64) -> c_int;
    pub fn cftrt_scene_commit(scene: *mut cftrt_scene) -> c_int;
    // ---- the hot path
    pub fn cftrt_render(
        scene: *mut cftrt_scene,
        cam: *const cftrt_camera,
        width: u32,
        height: u32,
        spp_begin: u32,
        spp_end: u32,
        max_depth: u32,
        seed: u64,
        accum_host: *mut f32,
        stats: *mut cftrt_stats,
    ) -> c_int;
    pub fn cftrt_render_device(
        scene: *mut cftrt_scene,
        cam: *const cftrt_camera,
        width: u32,
        height: u32,
        spp_begin: u32,
        spp_end: u32,
        max_depth: u32,
        seed: u64,
        accum_device: *mut c_void,
        stats: *mut cftrt_stats,
    ) -> c_int;
    pub fn cftrt_render_device_static(
        scene: *mut cftrt_scene,
        cam: *const cftrt_camera,
        width: u32,
        height: u32,
        spp_begin: u32,
        spp_end: u32,
        max_depth: u32,
        seed: u64,
        accum_device: *mut c_void,
        stats: *mut cftrt_stats,
    ) -> c_int;
    // ---- multi-GPU: the one exchange step
    pub fn cftrt_comm_unique_id(id128: *mut c_uchar) -> c_int;
    pub fn cftrt_comm_init(ctx: *mut cftrt_ctx, nranks: c_int, rank: c_int, id128: *const c_uchar) -> c_int;
    pub fn cftrt_reduce(ctx: *mut cftrt_ctx, accum_device: *mut c_void, n_floats: u64, root: c_int) -> c_int;
    pub fn cftrt_comm_destroy(ctx: *mut cftrt_ctx) -> c_int;
    // ---- parity hooks
    pub fn cftrt_generate(
        ctx: *mut cftrt_ctx,
        cam: *const cftrt_camera,
        width: u32,
        height: u32,
        pid0: u64,
        n: u64,
        seed: u64,
        jitter: c_int,
        out_rays: *mut f32,
    ) -> c_int;
    pub fn cftrt_intersect(
        scene: *mut cftrt_scene,
        rays: *const f32,
        n: u64,
        seed: u64,
        pid0: u64,
        level: u32,
        role: u32,
        out_id: *mut i32,
        out_t: *mut f64,
        out_normal: *mut f32,
        out_point: *mut f32,
        out_outward: *mut i32,
    ) -> c_int;
    pub fn cftrt_level_hits(
        scene: *mut cftrt_scene,
        cam: *const cftrt_camera,
        width: u32,
        height: u32,
        rays: *const f32,
        n: u64,
        seed: u64,
        pid0: u64,
        level: u32,
        out_id: *mut i32,
        out_t: *mut f64,
        out_scans: *mut i32,
    ) -> c_int;
    pub fn cftrt_trace_paths(
        scene: *mut cftrt_scene,
        cam: *const cftrt_camera,
        width: u32,
        height: u32,
        pid0: u64,
        n: u64,
        max_depth: u32,
        seed: u64,
        jitter: c_int,
        out_rgb: *mut f32,
        stats: *mut cftrt_stats,
    ) -> c_int;
    pub fn cftrt_resolve_rgb8(
        ctx: *mut cftrt_ctx,
        accum_host: *const f32,
        npix: u64,
        divisor: f64,
        out_rgb8: *mut u8,
    ) -> c_int;
}
