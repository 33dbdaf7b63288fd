//! The render loops of the crate's two examples, rewritten to the one batched call (UNVERIFIED text: no Rust toolchain
//! in the build image).  Only the REPLACED part of each `main` is written out; the scene and camera construction above
//! it stays exactly as the crate has it and is not repeated here.
//!
//!   examples/scene.rs    lines 82-109 (the `raw` vector, the thread pool over 128-ray chunks, the mutex-guarded
//!                        accumulate, the timing print)        ->  `scene_render_and_report`
//!   examples/cornell.rs  lines 31-44  (the `Mutex<Vec<(Vec3, count)>>`, the emit_rays/for_each loop, the divide by
//!                        the per-pixel count)                 ->  `cornell_render`
//!
//! Both keep what follows in the examples (`vec3_to_rgb`, `ImageBuffer::from_fn`, `img.save("test.jpg")`).
//! The `use` lists lose `parking_lot::Mutex`, `threadpool::ThreadPool`, `util::ChunkIter`, `std::sync::{Arc, Mutex}`.
use std::time::Instant;

use raytracer::{object::World, Camera, Color};

/// scene.rs: `let raw = scene_render_and_report(&world, &camera);` then the existing
/// `raw.iter().map(|pixel| vec3_to_rgb(*pixel / SAMPLE_RATE as f64))`.
pub fn scene_render_and_report(world: &World, camera: &Camera, width: u64, height: u64, sample_rate: u64, depth: u64) -> Vec<Color> {
    let start = Instant::now();
    // World::render (src/gpu.rs): result[h * width + w] = SUM over the camera's samples of world.trace(ray, depth);
    // the GPU path when every shape, material and light has a device form, the crate's own loop otherwise
    let raw = world.render(camera, width, height, depth);
    let duration = Instant::now().duration_since(start);
    // the example's own report, "pixel" meaning primary sample as before
    println!(
        "total: {} seconds, {:} ns/pixel",
        duration.as_secs(),
        duration.as_nanos() / (width * height * sample_rate) as u128
    );
    raw
}

/// cornell.rs: every pixel receives exactly `sample_rate` samples (Camera::emit_rays yields that many per pixel), so
/// the example's per-pixel counter is the constant `sample_rate`: `vec3_to_rgb(pixel / SAMPLE_RATE as f64)`.
pub fn cornell_render(world: &World, camera: &Camera, width: u64, height: u64) -> Vec<Color> {
    world.render(camera, width, height, 10) // the example hard-codes depth 10
}

/// Several GPUs of one node (src/gpu.rs, World::render_sharded): the same image from sample ranges, one thread per GPU,
/// summed over xGMI.  `alloc` / `download` / `free` are the host's own HIP bindings for a zeroed device buffer.
pub fn scene_render_on_gpus<A, D, F>(world: &World, camera: &Camera, width: u64, height: u64, depth: u64, n_gpus: i32,
                                     alloc: A, download: D, free: F) -> Vec<Color>
where
    A: Fn(i32, usize) -> *mut std::os::raw::c_void + Sync,
    D: Fn(i32, *mut std::os::raw::c_void, &mut [f32]),
    F: Fn(i32, *mut std::os::raw::c_void) + Sync,
{
    match world.render_sharded(camera, width, height, depth, 0, n_gpus, alloc, download, free) {
        Ok(Some(img)) => img,
        _ => world.render(camera, width, height, depth),
    }
}
