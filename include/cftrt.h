/* cftrt.h — C ABI of libcftrt.so: the MI355X (gfx950) trace loop for the cft_ray_tracer scene API.
 *
 * The reference crate (cfeitong/ray-tracing) has NO FFI and no batched render entry point: its
 * examples hand-roll the render loop around the per-ray API
 *     Camera::emit_rays  (src/ray.rs:107-136)
 *     World::trace       (src/object.rs:326-355)
 *     Ray::hit           (src/ray.rs:18-28)
 *     accumulate/resolve (examples/scene.rs:82-127, examples/cornell.rs:31-57)
 * This header is the boundary a façade's `World::render(&Camera, w, h, depth)` binds to
 * (INTEGRATION.md shows the Rust `extern "C"` block).  Plain pointers and sizes only; every
 * function returns a status code and never aborts or throws across the boundary (the reference
 * panics on `unwrap`; SURVEY.md §8b "Errors").
 *
 * Ownership: the caller owns every host buffer; `cftrt_scene_set_*` copy; handles are freed only
 * by `*_destroy`; destroy every scene BEFORE the ctx it was created on (a scene keeps a pointer to
 * its ctx).  Threading: a ctx is bound to one device and one stream and is NOT thread-safe;
 * use one ctx (one host thread / one process) per GPU.  Scenes are immutable after commit.
 * Device and stream: every call makes the ctx's device current for its own work and restores the
 * caller's current device before it returns.  The work runs on the ctx's own non-blocking stream
 * and the render calls return after that stream has drained; work the CALLER has queued on other
 * streams for a buffer it passes in (e.g. clearing the accumulation image) must have completed
 * before the call - the library does not wait on foreign streams.
 */
#ifndef CFTRT_H
#define CFTRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFTRT_OK 0
#define CFTRT_ERR_INVALID 1   /* bad argument (null pointer, negative size, ids out of range) */
#define CFTRT_ERR_HIP 2       /* a HIP runtime call failed; text in cftrt_last_error */
#define CFTRT_ERR_STATE 3     /* call out of order (render before commit, set after commit) */
#define CFTRT_ERR_LIMIT 4     /* size beyond a documented limit */

typedef struct cftrt_ctx cftrt_ctx;
typedef struct cftrt_scene cftrt_scene;

/* Material table row.  Replaces the `Arc<dyn Material>` trait objects:
 *   kind 0 LambertianModel  p0=albedo              color   (src/material/compose.rs:90-119)
 *   kind 1 Metal            p0=fuzz  p1=albedo     color   (src/material/compose.rs:9-52)
 *   kind 2 Dielectric       p0=ior                         (src/material/compose.rs:54-88)
 *   kind 3 PhongModel       p0=shininess p1=diffuse color  (src/material/basic.rs:10-98)
 *   kind 4 Specular         p0=albedo                      (src/material/basic.rs:100-124)
 *   kind 5 Transparent      p0=opacity p1=ior      color   (src/material/basic.rs:126-184) */
typedef struct cftrt_material {
  int32_t kind;
  int32_t reserved;
  double p0, p1;
  double color[3];
} cftrt_material;

#define CFTRT_MAT_LAMBERTIAN 0
#define CFTRT_MAT_METAL 1
#define CFTRT_MAT_DIELECTRIC 2
#define CFTRT_MAT_PHONG 3
#define CFTRT_MAT_SPECULAR 4
#define CFTRT_MAT_TRANSPARENT 5

/* Light list row.  Replaces `Arc<dyn LightSource>` (src/light.rs):
 *   kind 0 SkyLight                                   (light.rs:151-191)
 *   kind 1 LightShape: shape = light spheres [sph_begin, sph_begin+sph_count) and light
 *          triangles [tri_begin, tri_begin+tri_count), Square/Cube pre-expanded in the
 *          reference's face/triangle order; color fixed (1,1,1)   (light.rs:193-243)
 *   kind 2 PointLight    v=pos, color                 (light.rs:104-149)
 *   kind 3 ParallelLight v=dir (used un-normalised), color  (light.rs:66-102) */
typedef struct cftrt_light {
  int32_t kind;
  int32_t sph_begin, sph_count;
  int32_t tri_begin, tri_count;
  int32_t reserved;
  double v[3];
  double color[3];
} cftrt_light;

#define CFTRT_LIGHT_SKY 0
#define CFTRT_LIGHT_SHAPE 1
#define CFTRT_LIGHT_POINT 2
#define CFTRT_LIGHT_PARALLEL 3

/* Camera POD: the state `Camera::new(from,to).with_*()` leaves behind (src/ray.rs:46-104,139-152).
 * `up` and `sight` are the values stored after `look()`; right = unit(sight x up) is derived. */
typedef struct cftrt_camera {
  double pos[3];
  double up[3];
  double sight[3];
  double focus_dist;
  double aperture;
  double fov_rad; /* radians; the reference's default is 45 *radians* (ray.rs:147) */
  double aspect;
} cftrt_camera;

#define CFTRT_MAX_LEVELS 256
/* Size of the static device workspace of cftrt_render_device_static (bytes, per GPU the library runs on). */
#define CFTRT_STATIC_SCRATCH_BYTES (256u << 20)

/* Filled by the render calls.  "Rays" follow BASELINE.md §2: every trace() level with depth>0
 * plus one ray per (Phong hit, light). */
typedef struct cftrt_stats {
  uint64_t paths;             /* primary samples rendered */
  uint64_t rays;              /* all bounces (+ Phong light rays) */
  uint64_t phong_rays;        /* subset of `rays` issued by the shadow kernel */
  uint64_t exact_second;      /* rays whose runner-up candidate needed an f64 evaluation */
  uint64_t exact_rescan;      /* rays that fell back to the full f64 scan */
  double flops;               /* algorithmic: sum over rays of 16*Nsph + 44*Ntri + 200 */
  double flops_intersect;     /* part of `flops` done by the intersect kernel: rays*(16 Nsph+44 Ntri) */
  double ms_total;            /* wall time of the call's device work (HIP events) */
  double ms_generate, ms_intersect, ms_shade, ms_shadow, ms_resolve;
  uint32_t launches_intersect;
  uint32_t levels;            /* trace levels that had at least one ray */
  uint32_t passes;
  uint32_t reserved;
  uint64_t rays_per_level[CFTRT_MAX_LEVELS];
} cftrt_stats;

/* ---- context ---------------------------------------------------------------------------- */
int cftrt_device_count(int* out_n);
int cftrt_ctx_create(int device, cftrt_ctx** out);
void cftrt_ctx_destroy(cftrt_ctx* ctx);
/* Last error text of this ctx (or of the failed create when ctx is NULL). Never NULL. */
const char* cftrt_last_error(const cftrt_ctx* ctx);
/* Cap on the path workspace (bytes of HBM the render calls may hold; default 40 GiB). */
int cftrt_ctx_set_workspace_limit(cftrt_ctx* ctx, uint64_t bytes);
/* Pipeline per trace level: 0 (default) = one fused launch (k_trace_level: intersect, shade and queue append),
 * 1 = split (k_intersect -> hit records in HBM -> k_shade), the form the kernels are profiled in separately.
 * Both produce the same image bit for bit.  In fused mode cftrt_stats.ms_intersect / flops_intersect /
 * launches_intersect describe the fused kernel and ms_shade is 0. */
int cftrt_ctx_set_pipeline(cftrt_ctx* ctx, int split);
/* Wave-level cull (default on): the level kernels skip spheres that no ray of a wave's 64-256 ray bundle can
 * reach (a conservative cone test per sphere and wave) before the per-ray cull.  Results are identical with it
 * off; the switch exists for that comparison (tests) and for profiling the plain linear scan. */
int cftrt_ctx_set_wave_cull(cftrt_ctx* ctx, int on);
/* Uniform grid - an EXTENSION: it has no counterpart in the reference, whose Ray::hit scans every object (src/ray.rs:18-28).
 * cftrt_scene_commit also registers the static spheres, the swept bounds of moving spheres and the static
 * triangles of a scene with >= 64 of them in a uniform grid; rays then test only the primitives of the cells they
 * cross and stop behind the nearest candidate.  The candidates go through the same exact f64 evaluation as those
 * of the list scan, so images are identical bit for bit with any mode.
 *   mode 0: lists only (the reference's linear scan);  1 (default): the primary rays keep the lists with the
 *   wave-level cull, every later trace level walks the grid - and the primary rays too once the scene lists more
 *   than 4096 primitives;  2: every ray walks.
 *   cells_per_prim > 0 sets the grid resolution for scenes committed afterwards (default 1); <= 0 keeps it. */
int cftrt_ctx_set_grid(cftrt_ctx* ctx, int mode, double cells_per_prim);
/* Sharded appends (default on): a trace level with >= 2^22 rays appends its survivors through 8 counters into a
 * queue of 8 compacted segments instead of one counter (a single word takes ~88 returning atomics per microsecond
 * chip-wide, one per 256 rays).  The image does not depend on the order of the queue: same image bit for bit; the
 * switch exists for that comparison and for profiling. */
int cftrt_ctx_set_shards(cftrt_ctx* ctx, int on);
/* Per-tile lists for the primary rays (default on) - an EXTENSION like the grid: when the image is rendered in whole
 * 32x8-pixel tiles (width a multiple of 32, height of 8), every render call first bins the bounding spheres of the
 * listed primitives, widened for the lens, into the tiles whose rays can reach them (one pass over the primitives,
 * ~40 us), and the level-0 launch scans a tile's list instead of the whole lists / instead of walking the grid.  The
 * lists are conservative supersets and feed the same cull and exact evaluation: same image bit for bit; the switch
 * exists for that comparison and for profiling.  Not used by cftrt_render_device_static (it would allocate) nor by
 * the split pipeline. */
int cftrt_ctx_set_tile_lists(cftrt_ctx* ctx, int on);
/* The grid in LDS (default on): in a scene of spheres only whose grid - cell records and sphere entries - fits in
 * 50 KB (S500: 35 KB), the trace levels that walk the grid with at least `above` rays (0 keeps the current value,
 * default 2^20) run a kernel instance that first copies the grid into the CU's LDS and reads it from there: the
 * walk is a chain of dependent, divergent loads, and through the vector memory path its waves wait most of the
 * time.  Same cells, same entries, same tests: same image bit for bit; the switch exists for that comparison and
 * for profiling. */
int cftrt_ctx_set_lds_grid(cftrt_ctx* ctx, int on, uint32_t above);
/* Two classes of rays in the queue (default on): a wave of a level that walks the grid runs each of its ray slots as
 * long as the slot's longest walk, so a trace level with at least `above` rays (0 keeps the current value, default
 * 2^18) whose successor walks the grid appends the survivors whose next ray will cross at least `cells` cells of the
 * grid's box (0: a threshold derived from the grid's size) from the top of each queue segment downwards and the
 * others from the bottom up, and the next launch takes the long part first, in waves that hold nothing else.  No
 * extra memory.  S500 -3.7 %, S20k -4 %.  Which wave traces a ray never affects the image: same image bit for bit;
 * the switch exists for that comparison and for profiling. */
int cftrt_ctx_set_walk_classes(cftrt_ctx* ctx, int on, uint32_t cells, uint32_t above);
/* Deep levels path by path (default on): once fewer than `below` rays are alive (0 keeps the current value,
 * default 2^20; the host looks every 4 levels) the remaining trace levels run in one launch in which every lane
 * follows one path to its end (k_trace_finish) instead of one launch per level.  Same image bit for bit; the switch
 * exists for that comparison and for profiling.  Scenes with PhongModel materials always run level by level. */
int cftrt_ctx_set_finish(cftrt_ctx* ctx, int on, uint32_t below);
/* Intersect-kernel variant: rays per lane of k_intersect / k_trace_level (1..4); 0 = the tuned default. */
int cftrt_ctx_set_variant(cftrt_ctx* ctx, int variant);

/* ---- scene (replaces World::empty/add_obj/add_light, src/object.rs:311-324) ------------- */
int cftrt_scene_create(cftrt_ctx* ctx, cftrt_scene** out);
void cftrt_scene_destroy(cftrt_scene* scene);
/* Spheres (src/object.rs:257-303).  center: n x 3, radius: n (negative flips the normal),
 * moving_to: n x 3 or NULL (Object::moved, object.rs:37-40), obj_id: index of the owning object in
 * world.objects (non-decreasing), mat_id: row of the material table. */
int cftrt_scene_set_spheres(cftrt_scene* scene, const double* center, const double* radius,
                            const double* moving_to, const int32_t* obj_id, const int32_t* mat_id,
                            int64_t n);
/* Triangles (src/object.rs:58-139); Square = 2 and Cube = 12 triangles sharing one obj_id, expanded
 * by the host in the reference's order (object.rs:147-162, 215-230).  p0,p1,p2: n x 3 each. */
int cftrt_scene_set_triangles(cftrt_scene* scene, const double* p0, const double* p1,
                              const double* p2, const double* moving_to, const int32_t* obj_id,
                              const int32_t* mat_id, int64_t n);
int cftrt_scene_set_materials(cftrt_scene* scene, const cftrt_material* table, int64_t n);
int cftrt_scene_set_lights(cftrt_scene* scene, const cftrt_light* lights, int64_t n_lights,
                           const double* lsph_center, const double* lsph_radius, int64_t n_lsph,
                           const double* ltri_p0, const double* ltri_p1, const double* ltri_p2,
                           int64_t n_ltri);
/* Square::from_points (src/object.rs:169-174): its two triangles (p0,p1,p2), (p1,p2,p3) need not share a plane, and
 * Square::hit_info is `tri0.hit_info(ray).or(tri1.hit_info(ray))` (object.rs:192): the second triangle counts only
 * where the first is NOT hit, whatever the two distances (Square::new squares are planar and share only the diagonal,
 * so for them the rule is the closest of the two - they need no entry here).  tri_first[i] = i - 1 where triangle i
 * (of cftrt_scene_set_triangles) is the SECOND triangle of such a square and triangle i - 1, of the same obj_id, its
 * first; -1 for every other triangle.  ltri_first likewise over the light-shape triangles of cftrt_scene_set_lights
 * (both triangles in one light).  Optional; n_tri / n_ltri must equal the triangle counts at commit (0 with NULL: no
 * pairs of that kind). */
int cftrt_scene_set_triangle_pairs(cftrt_scene* scene, const int32_t* tri_first, int64_t n_tri,
                                   const int32_t* ltri_first, int64_t n_ltri);
/* Spheres with |radius| >= this are tested in f64 for every ray (default 25). */
int cftrt_scene_set_precise_radius(cftrt_scene* scene, double radius);
/* Uploads the flattened scene to HBM.  No setter may be called afterwards. */
int cftrt_scene_commit(cftrt_scene* scene);

/* ---- the hot path ------------------------------------------------------------------------ */
/* Replaces the examples' render loop (scene.rs:91-102, cornell.rs:32-39):
 *   accum[(h*W+w)*3 + c] += sum over samples s in [spp_begin, spp_end) of World::trace(ray, max_depth)
 * accum is a caller-allocated HOST buffer of W*H*3 floats, row-major, h = 0 is the top row.
 * The path id of (sample s, pixel h*W+w) is s*W*H + h*W + w and keys the Philox stream, so the
 * result does not depend on how the sample range is split over calls or GPUs. */
int cftrt_render(cftrt_scene* scene, const cftrt_camera* cam, uint32_t width, uint32_t height,
                 uint32_t spp_begin, uint32_t spp_end, uint32_t max_depth, uint64_t seed,
                 float* accum_host, cftrt_stats* stats);
/* Same, accumulating into a DEVICE buffer (W*H*3 floats on the ctx's device), e.g. a torch tensor
 * that is then reduced over ranks with RCCL.  The call returns after its stream has drained; pending
 * work of the caller on that buffer must have finished before the call (see "Device and stream"). */
int cftrt_render_device(cftrt_scene* scene, const cftrt_camera* cam, uint32_t width,
                        uint32_t height, uint32_t spp_begin, uint32_t spp_end, uint32_t max_depth,
                        uint64_t seed, void* accum_device, cftrt_stats* stats);

/* The same render for callers that cannot allocate.  It has NO counterpart in the reference; it exists for hosts
 * that must not call hipMalloc at this point: hosts with a HIP graph capture in progress on ANOTHER stream (an
 * allocation would invalidate the capture; the call itself synchronises its own stream and reads level counters back,
 * so it cannot be captured into a graph), tight serving loops, harnesses that forbid allocation.  Same arguments and result as cftrt_render_device (for a sample range that cftrt_render_device
 * renders in one pass the images are identical bit for bit; k_resolve adds samples one by one, so the cut into
 * passes does not change the sum either), but the pass workspace (ray queues, radiance slots, level counters) is a
 * file-scope device array of CFTRT_STATIC_SCRATCH_BYTES (256 MiB) that every GPU the library runs on holds until the
 * process ends; the call makes no hipMalloc/hipFree and creates no event (cftrt_stats: ms_total only, the per-kernel
 * times stay 0).  Limit: one sample of the image must fit - width*height*92 bytes (148 with PhongModel materials,
 * +40/80 in the split pipeline) <= 256 MiB less the 732 KiB the library keeps for the level counters, the slack
 * of the segmented queues and the alignment of its carves (kStaticHeader + 4 KiB in cftrt.hip), i.e. up to 2.91 M
 * pixels (1920x1080 fits, 3840x2160 does not): CFTRT_ERR_LIMIT otherwise.  Larger sample counts run as several passes.  All calls on one GPU share that
 * GPU's array (whatever the ctx): calls that overlap in time on one GPU race; the library clears the counters it
 * needs at the start of every pass, nothing else must start at zero. */
int cftrt_render_device_static(cftrt_scene* scene, const cftrt_camera* cam, uint32_t width,
                               uint32_t height, uint32_t spp_begin, uint32_t spp_end,
                               uint32_t max_depth, uint64_t seed, void* accum_device,
                               cftrt_stats* stats);

/* ---- multi-GPU: the one exchange step of the path ------------------------------------------- */
/* The image shards by sample range: rank g renders samples [b_g, e_g) of every pixel into its own accumulation
 * buffer (cftrt_render_device) and the buffers are summed - where the reference adds every sample into one vector
 * under a mutex (examples/scene.rs:97).  These calls do that sum with RCCL (ncclReduce over xGMI) for hosts without
 * torch.distributed; librccl is loaded on first use (dlopen), a single-GPU host never needs it.
 *   rank 0: cftrt_comm_unique_id(id)  -> ship the 128 bytes to the other ranks by any host channel
 *   every rank (one process or host thread per GPU): cftrt_comm_init(ctx, nranks, rank, id)
 *   after the render calls: cftrt_reduce(ctx, accum_device, W*H*3, root) - in place, f32 sum into root's buffer,
 *   on the ctx's stream (ordered behind the renders); returns after the stream has drained. */
#define CFTRT_COMM_ID_BYTES 128
int cftrt_comm_unique_id(unsigned char* id128);
int cftrt_comm_init(cftrt_ctx* ctx, int nranks, int rank, const unsigned char* id128);
int cftrt_reduce(cftrt_ctx* ctx, void* accum_device, uint64_t n_floats, int root);
int cftrt_comm_destroy(cftrt_ctx* ctx);

/* ---- parity hooks (one kernel each, host buffers) ------------------------------------------ */
/* Camera::emit_rays for path ids [pid0, pid0+n): out_rays = n x 6 floats (pos, dir).
 * jitter = 0 disables the pixel/lens draws (pixel centre, lens centre). */
int cftrt_generate(cftrt_ctx* ctx, const cftrt_camera* cam, uint32_t width, uint32_t height,
                   uint64_t pid0, uint64_t n, uint64_t seed, int jitter, float* out_rays);
/* Ray::hit for caller-supplied rays (n x 6 floats).  The direction is used as given, like a
 * struct-literal Ray (ray.rs:277-289): a direction that is not unit (| |d|^2 - 1 | > 1e-5) skips the
 * f32 cull, which assumes unit directions, and is resolved by the exact f64 scan (slow: every
 * primitive per ray).  out_id = object index or -1, out_t = distance (f64, the exact evaluation),
 * out_normal = n x 3 shading normal after HitInfo::new's flip, out_point = n x 3 hit point,
 * out_outward = HitInfo::outward (1 where the flip happened, ray.rs:237-241).  Motion times are keyed
 * by (seed, pid0+i, level, role) like the render path.  Any out_* except out_id may be NULL. */
int cftrt_intersect(cftrt_scene* scene, const float* rays, uint64_t n, uint64_t seed,
                    uint64_t pid0, uint32_t level, uint32_t role, int32_t* out_id, double* out_t,
                    float* out_normal, float* out_point, int32_t* out_outward);
/* Ray::hit as the PRODUCTION level kernel finds it (cftrt_intersect above runs a one-ray-per-lane form of the same
 * code): ONE launch of the fused level kernel exactly as a render call issues it at trace level `level` - the
 * instance for the scene's kind, the rays per lane of cftrt_ctx_set_variant, wave cull, grid walk, per-tile lists as
 * the ctx has them - as the pass's last level (nothing is appended), returning the winner of every scan of every ray.
 *   rays != NULL: n caller rays (n x 6 floats, unit directions) are the level's input queue; path id of ray i =
 *     pid0 + i.  With level >= 1 and a grid the walk instances run, with level 0 the cone-cull / list instances;
 *     fewer than 2^17 (grid) / 2^19 rays with variant 0 run the tail kernel, as in a render call.
 *   rays == NULL: level must be 0 and [pid0, pid0 + n) whole samples of the width x height image of `cam`: the
 *     launch generates its rays itself (Camera::emit_rays; the rays cftrt_generate returns for the same ids), in
 *     32x8-pixel tiles with per-tile lists where a render call would use them.
 * *out_scans = scans per ray (2 for a scene with moving objects and lights: [0] the lights' `looked` scan, keyed by
 * the first looking light's index, [1] trace's own, keyed by the number of lights; else 1).  out_id[i * scans + k] =
 * object index or -1, out_t[i * scans + k] = the exact f64 distance (+inf on a miss).  Not in the split pipeline. */
int cftrt_level_hits(cftrt_scene* scene, const cftrt_camera* cam, uint32_t width, uint32_t height,
                     const float* rays, uint64_t n, uint64_t seed, uint64_t pid0, uint32_t level,
                     int32_t* out_id, double* out_t, int32_t* out_scans);
/* World::trace per path: out_rgb = n x 3 floats for path ids [pid0, pid0+n).  Whole samples of the image
 * (pid0 and n multiples of width*height) run exactly as cftrt_render runs them (tiled queue, per-tile lists). */
int cftrt_trace_paths(cftrt_scene* scene, const cftrt_camera* cam, uint32_t width, uint32_t height,
                      uint64_t pid0, uint64_t n, uint32_t max_depth, uint64_t seed, int jitter,
                      float* out_rgb, cftrt_stats* stats);
/* vec3_to_rgb (scene.rs:122-127): out[i] = trunc(255.99*sqrt(clamp(accum[i]/divisor, 0, 1))). */
int cftrt_resolve_rgb8(cftrt_ctx* ctx, const float* accum_host, uint64_t npix, double divisor,
                       uint8_t* out_rgb8);

#ifdef __cplusplus
}
#endif
#endif /* CFTRT_H */
