// cftrt.hip — host side of libcftrt.so: the C ABI of include/cftrt.h over the kernels in kernels.h.
// One ctx = one device + one stream.  No torch types; plain pointers and sizes.
#include "../../include/cftrt.h"
#include "kernels.h"

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_create_error = "";

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct cftrt_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  uint64_t ws_limit = 40ull << 30;
  int variant = 0;
  int wave_cull = 1;      // level kernels may skip spheres no ray of a wave's bundle can reach (exact either way)
  int grid = 1;           // uniform grid (an extension to the reference's linear scan): 0 off, 1 waves the cone cull does not serve, 2 every wave
  double grid_density = 1.;   // cells per primitive the grid is sized for (scenes committed afterwards)
  int split = 0;          // 0: fused k_trace_level per level (default); 1: k_intersect -> hit records -> k_shade
  int shards = 1;         // 1: levels with >= 2^22 rays append through CFT_SHARDS counters into a segmented queue
  int finish = 1;         // 1: once few rays are alive the remaining levels run path by path in one launch (k_trace_finish)
  uint32_t finish_below = 1u << 20;
  int classes = 1;        // 1: levels whose successor walks the grid append their survivors in two classes by walk length (kernels.h, CFT_NSEG)
  uint32_t classes_above = 1u << 18, classes_thr = 0;   // ... for levels with at least this many rays; threshold in cells (0: from the grid's size)
  int lds_grid = 1;       // 1: walk levels of sphere-only scenes whose grid fits keep it in LDS (k_trace_level<.., LISTS & 16>)
  uint32_t lds_grid_above = 1u << 20;   // ... for levels with at least this many rays
  int tile_lists = 1;     // 1: primary rays of tiled frames scan per-tile primitive lists built per render call (k_tile_bin)
  bool tiles_active = false;   // the lists in tile_off / tile_idx are those of the render call in progress
  // what the lists in tile_off / tile_idx were built for: a render call with the same committed scene, camera and image
  // size (a progressive host, the frames of a benchmark, the passes of a multi-GPU rank) takes them as they are
  uint64_t tiles_scene = 0;    // commit serial of the scene (0: no lists)
  unsigned char tiles_cam[96] = {0};   // the CamDev they were binned with
  int tiles_usable = 0;        // the build decided the lists are worth using (not a scene wrapped around the camera)
  int n_cu = 256;
  // grow-only workspace
  DevBuf qa[2], qb[2], qc[2], h0[1 + CFT_MAX_LOOK], h1[1 + CFT_MAX_LOOK], ht[1 + CFT_MAX_LOOK], rad, pqa, pqb, pqc, pqd, counts, accum_tmp, misc, tile_rng, tile_cnt, tile_off, tile_idx;
  uint32_t* host_counts = nullptr;   // pinned
  std::vector<hipEvent_t> events;
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
  unsigned char* static_scratch = nullptr;   // this device's copy of `scratch` (cftrt_render_device_static)
  void* comm = nullptr;                      // ncclComm_t of cftrt_comm_init (RCCL, loaded on demand)
  int comm_ranks = 0, comm_rank = 0;
};

// RCCL through dlopen: the library is only needed by hosts that combine several GPUs without torch.distributed
// (cftrt_comm_* / cftrt_reduce); a single-GPU host never loads it.
struct RcclApi {
  void* lib = nullptr;
  struct UniqueId { char internal[128]; };
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
// Loaded once per process, whichever thread asks first (cftrt_comm_* may be called from one host thread per GPU).
static RcclApi load_rccl() {
  RcclApi api;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (api.lib) break;
  }
  if (api.lib) {
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.Reduce = (decltype(api.Reduce))dlsym(api.lib, "ncclReduce");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.Reduce || !api.CommDestroy) { dlclose(api.lib); api.lib = nullptr; }
  }
  return api;
}
static const RcclApi* rccl_api(std::string* err) {
  static const RcclApi api = load_rccl();   // (a function-local static: initialised exactly once, thread-safe since C++11)
  if (!api.lib) { if (err) *err = "librccl.so.1 could not be loaded (dlopen) or lacks the nccl* entry points"; return nullptr; }
  return &api;
}

// Workspace of the entry point that allocates nothing (cftrt_render_device_static): ray queues, radiance and level
// counters of a pass live in this file-scope array instead of hipMalloc'ed buffers.  Every GPU the library runs on
// holds its own copy, at its own address, until the process ends.  In host code the NAME is only a stand-in for
// the symbol: the device address comes from hipGetSymbolAddress (cftrt_ctx_create, with the ctx's device current)
// and only that pointer is ever handed to kernels, memsets and copies.
static __device__ __align__(256) unsigned char scratch[CFTRT_STATIC_SCRATCH_BYTES];

// Entry points make the ctx's device current for their own calls and put the caller's device back on return.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) err = hipSetDevice(device);
    else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define DEVICE_GUARD(ctx) DeviceGuard dev_guard_((ctx)->device); HIP_TRY(ctx, dev_guard_.err)

struct cftrt_scene {
  cftrt_ctx* ctx = nullptr;
  bool committed = false;
  uint64_t serial = 0;                 // unique per committed scene of the process (keys the ctx's cached tile lists)
  double precise_radius = 25.0;
  std::vector<double> sph_c, sph_r, sph_mv;
  std::vector<int32_t> sph_obj, sph_mat;
  std::vector<double> tri_p, tri_mv;   // 9 per tri
  std::vector<int32_t> tri_obj, tri_mat;
  std::vector<cftrt_material> mats;
  std::vector<cftrt_light> lights;
  std::vector<double> lsph, ltri;      // 4 / 9 per prim
  std::vector<int32_t> tri_first, ltri_first;   // Square::from_points pairs (cftrt_scene_set_triangle_pairs); empty: none
  int n_look = 0;                      // lights that look (SkyLight, LightShape)
  bool multi_look = false;             // moving scene with several of them: one `looked` scan per light (split pipeline)
  std::vector<void*> dev_allocs;
  SceneDev dev;
};

#define HIP_TRY(ctx, expr)                                                                     \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
      return CFTRT_ERR_HIP;                                                                    \
    }                                                                                          \
  } while (0)

static int fail(cftrt_ctx* ctx, int code, const char* msg) {
  if (ctx) ctx->err = msg;
  return code;
}

static int ensure(cftrt_ctx* ctx, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return CFTRT_OK;
  if (b.p) {
    HIP_TRY(ctx, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
  }
  if (bytes == 0) bytes = 256;
  HIP_TRY(ctx, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return CFTRT_OK;
}

template <typename T>
static int upload(cftrt_scene* s, const std::vector<T>& v, const T** out) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(v.size() * sizeof(T), 256);
  HIP_TRY(s->ctx, hipMalloc(&p, bytes));
  s->dev_allocs.push_back(p);
  HIP_TRY(s->ctx, hipMemsetAsync(p, 0, bytes, s->ctx->stream));
  if (!v.empty()) HIP_TRY(s->ctx, hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s->ctx->stream));
  *out = (const T*)p;
  return CFTRT_OK;
}

extern "C" {

int cftrt_device_count(int* out_n) {
  if (!out_n) return CFTRT_ERR_INVALID;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
    *out_n = 0;
    return CFTRT_ERR_HIP;
  }
  *out_n = n;
  return CFTRT_OK;
}

int cftrt_ctx_create(int device, cftrt_ctx** out) {
  if (!out) return CFTRT_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = std::string("no HIP device: ") + hipGetErrorString(e);
    return CFTRT_ERR_HIP;
  }
  if (device < 0 || device >= n) {
    g_create_error = "device index out of range";
    return CFTRT_ERR_INVALID;
  }
  cftrt_ctx* ctx = new cftrt_ctx();
  ctx->device = device;
  DeviceGuard guard(device);
  if ((e = guard.err) != hipSuccess || (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
    g_create_error = std::string("ctx_create: ") + hipGetErrorString(e);
    delete ctx;
    return CFTRT_ERR_HIP;
  }
  // this device's address of the static workspace (each GPU has its own copy of the array); `device` is current here
  if ((e = hipGetSymbolAddress((void**)&ctx->static_scratch, HIP_SYMBOL(scratch))) != hipSuccess || !ctx->static_scratch) {
    g_create_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CFTRT_ERR_HIP;
  }
  (void)hipEventCreate(&ctx->ev_t0);
  (void)hipEventCreate(&ctx->ev_t1);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
  if ((e = hipHostMalloc((void**)&ctx->host_counts, (size_t)CFT_NSEG * CFT_SHARD_STRIDE * sizeof(uint32_t))) != hipSuccess) {
    g_create_error = std::string("hipHostMalloc: ") + hipGetErrorString(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CFTRT_ERR_HIP;
  }
  *out = ctx;
  return CFTRT_OK;
}

void cftrt_ctx_destroy(cftrt_ctx* ctx) {
  if (!ctx) return;
  DeviceGuard guard(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm) (void)cftrt_comm_destroy(ctx);
  DevBuf* bufs[] = {&ctx->qa[0], &ctx->qa[1], &ctx->qb[0], &ctx->qb[1], &ctx->qc[0], &ctx->qc[1], &ctx->rad, &ctx->pqa, &ctx->pqb, &ctx->pqc,
                    &ctx->pqd, &ctx->counts, &ctx->accum_tmp, &ctx->misc, &ctx->tile_rng, &ctx->tile_cnt, &ctx->tile_off, &ctx->tile_idx};
  for (DevBuf* b : bufs)
    if (b->p) (void)hipFree(b->p);
  for (int k = 0; k < 1 + CFT_MAX_LOOK; ++k)
    for (DevBuf* b : {&ctx->h0[k], &ctx->h1[k], &ctx->ht[k]})
      if (b->p) (void)hipFree(b->p);
  for (hipEvent_t ev : ctx->events) (void)hipEventDestroy(ev);
  if (ctx->ev_t0) (void)hipEventDestroy(ctx->ev_t0);
  if (ctx->ev_t1) (void)hipEventDestroy(ctx->ev_t1);
  if (ctx->host_counts) (void)hipHostFree(ctx->host_counts);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* cftrt_last_error(const cftrt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int cftrt_ctx_set_workspace_limit(cftrt_ctx* ctx, uint64_t bytes) {
  if (!ctx) return CFTRT_ERR_INVALID;
  if (bytes < (64ull << 20)) return fail(ctx, CFTRT_ERR_LIMIT, "workspace limit below 64 MiB");
  ctx->ws_limit = bytes;
  return CFTRT_OK;
}

int cftrt_ctx_set_wave_cull(cftrt_ctx* ctx, int on) {
  if (!ctx) return CFTRT_ERR_INVALID;
  ctx->wave_cull = on ? 1 : 0;
  return CFTRT_OK;
}

int cftrt_ctx_set_grid(cftrt_ctx* ctx, int mode, double cells_per_prim) {
  if (!ctx) return CFTRT_ERR_INVALID;
  if (mode < 0 || mode > 2) return fail(ctx, CFTRT_ERR_INVALID, "grid mode must be 0, 1 or 2");
  ctx->grid = mode;
  if (cells_per_prim > 0.) ctx->grid_density = cells_per_prim;
  return CFTRT_OK;
}

int cftrt_ctx_set_pipeline(cftrt_ctx* ctx, int split) {
  if (!ctx) return CFTRT_ERR_INVALID;
  if (split != 0 && split != 1) return fail(ctx, CFTRT_ERR_INVALID, "pipeline must be 0 (fused) or 1 (split)");
  ctx->split = split;
  return CFTRT_OK;
}

int cftrt_ctx_set_shards(cftrt_ctx* ctx, int on) {
  if (!ctx) return CFTRT_ERR_INVALID;
  ctx->shards = on ? 1 : 0;
  return CFTRT_OK;
}

int cftrt_ctx_set_tile_lists(cftrt_ctx* ctx, int on) {
  if (!ctx) return CFTRT_ERR_INVALID;
  ctx->tile_lists = on ? 1 : 0;
  return CFTRT_OK;
}

int cftrt_ctx_set_walk_classes(cftrt_ctx* ctx, int on, uint32_t cells, uint32_t above) {
  if (!ctx) return CFTRT_ERR_INVALID;
  ctx->classes = on ? 1 : 0;
  ctx->classes_thr = cells;
  if (above) ctx->classes_above = above;
  return CFTRT_OK;
}

int cftrt_ctx_set_lds_grid(cftrt_ctx* ctx, int on, uint32_t above) {
  if (!ctx) return CFTRT_ERR_INVALID;
  ctx->lds_grid = on ? 1 : 0;
  if (above) ctx->lds_grid_above = above;
  return CFTRT_OK;
}

int cftrt_ctx_set_finish(cftrt_ctx* ctx, int on, uint32_t below) {
  if (!ctx) return CFTRT_ERR_INVALID;
  ctx->finish = on ? 1 : 0;
  if (below) ctx->finish_below = below;
  return CFTRT_OK;
}

int cftrt_ctx_set_variant(cftrt_ctx* ctx, int variant) {
  if (!ctx) return CFTRT_ERR_INVALID;
  if (variant < 0 || variant > 4) return fail(ctx, CFTRT_ERR_INVALID, "unknown intersect variant");
  ctx->variant = variant;
  return CFTRT_OK;
}

// ------------------------------------------------------------------ scene
int cftrt_scene_create(cftrt_ctx* ctx, cftrt_scene** out) {
  if (!ctx || !out) return CFTRT_ERR_INVALID;
  cftrt_scene* s = new cftrt_scene();
  s->ctx = ctx;
  memset(&s->dev, 0, sizeof s->dev);
  *out = s;
  return CFTRT_OK;
}

void cftrt_scene_destroy(cftrt_scene* s) {
  if (!s) return;
  if (s->ctx->tiles_scene == s->serial) s->ctx->tiles_scene = 0;
  DeviceGuard guard(s->ctx->device);
  (void)hipStreamSynchronize(s->ctx->stream);
  for (void* p : s->dev_allocs) (void)hipFree(p);
  delete s;
}

static int check_ids(cftrt_scene* s, const int32_t* obj, const int32_t* mat, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    if (obj[i] < 0 || mat[i] < 0) return fail(s->ctx, CFTRT_ERR_INVALID, "negative obj_id or mat_id");
    if (i && obj[i] < obj[i - 1]) return fail(s->ctx, CFTRT_ERR_INVALID, "obj_id must be non-decreasing");
  }
  return CFTRT_OK;
}

int cftrt_scene_set_spheres(cftrt_scene* s, const double* center, const double* radius, const double* moving_to,
                            const int32_t* obj_id, const int32_t* mat_id, int64_t n) {
  if (!s) return CFTRT_ERR_INVALID;
  if (s->committed) return fail(s->ctx, CFTRT_ERR_STATE, "scene already committed");
  if (n < 0 || (n > 0 && (!center || !radius || !obj_id || !mat_id))) return fail(s->ctx, CFTRT_ERR_INVALID, "set_spheres: null pointer or negative n");
  if (n > (1 << 26)) return fail(s->ctx, CFTRT_ERR_LIMIT, "set_spheres: more than 2^26 spheres");
  if (int rc = check_ids(s, obj_id, mat_id, n)) return rc;
  s->sph_c.assign(center, center + 3 * n);
  s->sph_r.assign(radius, radius + n);
  if (moving_to) s->sph_mv.assign(moving_to, moving_to + 3 * n);
  else s->sph_mv.assign(3 * n, 0.0);
  s->sph_obj.assign(obj_id, obj_id + n);
  s->sph_mat.assign(mat_id, mat_id + n);
  return CFTRT_OK;
}

int cftrt_scene_set_triangles(cftrt_scene* s, const double* p0, const double* p1, const double* p2,
                              const double* moving_to, const int32_t* obj_id, const int32_t* mat_id, int64_t n) {
  if (!s) return CFTRT_ERR_INVALID;
  if (s->committed) return fail(s->ctx, CFTRT_ERR_STATE, "scene already committed");
  if (n < 0 || (n > 0 && (!p0 || !p1 || !p2 || !obj_id || !mat_id))) return fail(s->ctx, CFTRT_ERR_INVALID, "set_triangles: null pointer or negative n");
  if (n > (1 << 26)) return fail(s->ctx, CFTRT_ERR_LIMIT, "set_triangles: more than 2^26 triangles");
  if (int rc = check_ids(s, obj_id, mat_id, n)) return rc;
  s->tri_p.resize(9 * n);
  for (int64_t i = 0; i < n; ++i) {
    memcpy(&s->tri_p[9 * i], p0 + 3 * i, 24);
    memcpy(&s->tri_p[9 * i + 3], p1 + 3 * i, 24);
    memcpy(&s->tri_p[9 * i + 6], p2 + 3 * i, 24);
  }
  if (moving_to) s->tri_mv.assign(moving_to, moving_to + 3 * n);
  else s->tri_mv.assign(3 * n, 0.0);
  s->tri_obj.assign(obj_id, obj_id + n);
  s->tri_mat.assign(mat_id, mat_id + n);
  return CFTRT_OK;
}

int cftrt_scene_set_materials(cftrt_scene* s, const cftrt_material* table, int64_t n) {
  if (!s) return CFTRT_ERR_INVALID;
  if (s->committed) return fail(s->ctx, CFTRT_ERR_STATE, "scene already committed");
  if (n < 0 || (n > 0 && !table)) return fail(s->ctx, CFTRT_ERR_INVALID, "set_materials: null pointer or negative n");
  for (int64_t i = 0; i < n; ++i)
    if (table[i].kind < 0 || table[i].kind > 5) return fail(s->ctx, CFTRT_ERR_INVALID, "set_materials: unknown kind");
  s->mats.assign(table, table + n);
  return CFTRT_OK;
}

int cftrt_scene_set_lights(cftrt_scene* s, const cftrt_light* lights, int64_t n_lights, const double* lsph_center,
                           const double* lsph_radius, int64_t n_lsph, const double* ltri_p0, const double* ltri_p1,
                           const double* ltri_p2, int64_t n_ltri) {
  if (!s) return CFTRT_ERR_INVALID;
  if (s->committed) return fail(s->ctx, CFTRT_ERR_STATE, "scene already committed");
  if (n_lights < 0 || n_lsph < 0 || n_ltri < 0 || (n_lights > 0 && !lights) ||
      (n_lsph > 0 && (!lsph_center || !lsph_radius)) || (n_ltri > 0 && (!ltri_p0 || !ltri_p1 || !ltri_p2)))
    return fail(s->ctx, CFTRT_ERR_INVALID, "set_lights: null pointer or negative n");
  if (n_lights > 64) return fail(s->ctx, CFTRT_ERR_LIMIT, "set_lights: more than 64 lights");
  for (int64_t i = 0; i < n_lights; ++i) {
    const cftrt_light& L = lights[i];
    if (L.kind < 0 || L.kind > 3) return fail(s->ctx, CFTRT_ERR_INVALID, "set_lights: unknown kind");
    if (L.kind == CFTRT_LIGHT_SHAPE) {
      if (L.sph_begin < 0 || L.sph_count < 0 || L.sph_begin + (int64_t)L.sph_count > n_lsph || L.tri_begin < 0 ||
          L.tri_count < 0 || L.tri_begin + (int64_t)L.tri_count > n_ltri)
        return fail(s->ctx, CFTRT_ERR_INVALID, "set_lights: shape range out of bounds");
    }
  }
  s->lights.assign(lights, lights + n_lights);
  s->lsph.resize(4 * n_lsph);
  for (int64_t i = 0; i < n_lsph; ++i) {
    memcpy(&s->lsph[4 * i], lsph_center + 3 * i, 24);
    s->lsph[4 * i + 3] = lsph_radius[i];
  }
  s->ltri.resize(9 * n_ltri);
  for (int64_t i = 0; i < n_ltri; ++i) {
    memcpy(&s->ltri[9 * i], ltri_p0 + 3 * i, 24);
    memcpy(&s->ltri[9 * i + 3], ltri_p1 + 3 * i, 24);
    memcpy(&s->ltri[9 * i + 6], ltri_p2 + 3 * i, 24);
  }
  return CFTRT_OK;
}

int cftrt_scene_set_triangle_pairs(cftrt_scene* s, const int32_t* tri_first, int64_t n_tri, const int32_t* ltri_first,
                                   int64_t n_ltri) {
  if (!s) return CFTRT_ERR_INVALID;
  if (s->committed) return fail(s->ctx, CFTRT_ERR_STATE, "scene already committed");
  if (n_tri < 0 || n_ltri < 0 || (n_tri > 0 && !tri_first) || (n_ltri > 0 && !ltri_first))
    return fail(s->ctx, CFTRT_ERR_INVALID, "set_triangle_pairs: null pointer or negative n");
  for (int64_t i = 0; i < n_tri; ++i)
    if (tri_first[i] != -1 && (tri_first[i] != i - 1 || i == 0 || tri_first[i - 1] != -1))
      return fail(s->ctx, CFTRT_ERR_INVALID, "set_triangle_pairs: tri_first[i] must be -1 or i - 1 (the triangle right before, itself not a second one)");
  for (int64_t i = 0; i < n_ltri; ++i)
    if (ltri_first[i] != -1 && (ltri_first[i] != i - 1 || i == 0 || ltri_first[i - 1] != -1))
      return fail(s->ctx, CFTRT_ERR_INVALID, "set_triangle_pairs: ltri_first[i] must be -1 or i - 1 (the triangle right before, itself not a second one)");
  s->tri_first.assign(tri_first, tri_first + n_tri);
  s->ltri_first.assign(ltri_first, ltri_first + n_ltri);
  return CFTRT_OK;
}

int cftrt_scene_set_precise_radius(cftrt_scene* s, double radius) {
  if (!s) return CFTRT_ERR_INVALID;
  if (s->committed) return fail(s->ctx, CFTRT_ERR_STATE, "scene already committed");
  if (!(radius > 0.0)) return fail(s->ctx, CFTRT_ERR_INVALID, "precise radius must be positive");
  s->precise_radius = radius;
  return CFTRT_OK;
}

// Does the triangle v[0..8] (three corners) meet the box (centre c, half-extents h)?  Separating axes: the box's three
// (done by the caller's cell ranges, repeated here), the triangle's normal, and the nine edge x axis products
// (Akenine-Moller).  Conservative: "separated" needs a gap above 1e-9 of the sizes involved.
static bool tri_box_overlap(const double c[3], const double h[3], const double* v) {
  double p[3][3];
  for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) p[j][k] = v[3 * j + k] - c[k];
  double scale = h[0] + h[1] + h[2];
  for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) scale = std::max(scale, std::fabs(p[j][k]));
  const double tol = 1e-9 * scale;
  for (int k = 0; k < 3; ++k) {
    const double lo = std::min(p[0][k], std::min(p[1][k], p[2][k])), hi = std::max(p[0][k], std::max(p[1][k], p[2][k]));
    if (lo > h[k] + tol || hi < -h[k] - tol) return false;
  }
  const double e[3][3] = {{p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2]},
                          {p[2][0] - p[1][0], p[2][1] - p[1][1], p[2][2] - p[1][2]},
                          {p[0][0] - p[2][0], p[0][1] - p[2][1], p[0][2] - p[2][2]}};
  auto separated = [&](const double a[3]) {
    const double al = std::fabs(a[0]) + std::fabs(a[1]) + std::fabs(a[2]);
    if (al == 0.) return false;
    double lo = 1e300, hi = -1e300;
    for (int j = 0; j < 3; ++j) { const double d = a[0] * p[j][0] + a[1] * p[j][1] + a[2] * p[j][2]; lo = std::min(lo, d); hi = std::max(hi, d); }
    const double r = h[0] * std::fabs(a[0]) + h[1] * std::fabs(a[1]) + h[2] * std::fabs(a[2]);
    const double t = tol * al;
    return lo > r + t || hi < -r - t;
  };
  const double nrm[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2], e[0][0] * e[1][1] - e[0][1] * e[1][0]};
  if (separated(nrm)) return false;
  for (int j = 0; j < 3; ++j)
    for (int k = 0; k < 3; ++k) {
      double a[3] = {0., 0., 0.};   // unit axis k x edge j
      const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
      a[k1] = -e[j][k2]; a[k2] = e[j][k1];
      if (separated(a)) return false;
    }
  return true;
}

int cftrt_scene_commit(cftrt_scene* s) {
  if (!s) return CFTRT_ERR_INVALID;
  if (s->committed) return fail(s->ctx, CFTRT_ERR_STATE, "scene already committed");
  cftrt_ctx* ctx = s->ctx;
  DEVICE_GUARD(ctx);
  const int ns = (int)s->sph_r.size(), nt = (int)s->tri_obj.size();
  const int nm = (int)s->mats.size();
  for (int i = 0; i < ns; ++i)
    if (s->sph_mat[i] >= nm) return fail(ctx, CFTRT_ERR_INVALID, "sphere mat_id beyond the material table");
  for (int i = 0; i < nt; ++i)
    if (s->tri_mat[i] >= nm) return fail(ctx, CFTRT_ERR_INVALID, "triangle mat_id beyond the material table");
  // Square::from_points pairs: one entry per triangle (or none at all), the shadowing triangle in the same object
  if (!s->tri_first.empty() && (int)s->tri_first.size() != nt) return fail(ctx, CFTRT_ERR_INVALID, "set_triangle_pairs: n_tri differs from the triangle count");
  if (!s->ltri_first.empty() && s->ltri_first.size() * 9 != s->ltri.size()) return fail(ctx, CFTRT_ERR_INVALID, "set_triangle_pairs: n_ltri differs from the light-triangle count");
  int any_pairs = 0;
  for (size_t i = 0; i < s->tri_first.size(); ++i)
    if (s->tri_first[i] >= 0) {
      if (s->tri_obj[s->tri_first[i]] != s->tri_obj[i]) return fail(ctx, CFTRT_ERR_INVALID, "set_triangle_pairs: a pair spans two objects");
      any_pairs = 1;
    }
  for (size_t i = 0; i < s->ltri_first.size(); ++i) {
    if (s->ltri_first[i] < 0) continue;
    bool same = false;   // both triangles inside one light's range
    for (const cftrt_light& L : s->lights)
      if (L.kind == CFTRT_LIGHT_SHAPE && (int64_t)i >= L.tri_begin && (int64_t)i < L.tri_begin + (int64_t)L.tri_count && s->ltri_first[i] >= L.tri_begin) same = true;
    if (!same) return fail(ctx, CFTRT_ERR_INVALID, "set_triangle_pairs: a light-triangle pair spans two lights");
    any_pairs = 1;
  }
  std::vector<int32_t> tri_first(s->tri_first), ltri_first(s->ltri_first);
  tri_first.resize((size_t)nt, -1);
  ltri_first.resize(s->ltri.size() / 9, -1);
  s->n_look = 0;
  for (const cftrt_light& L : s->lights) if (L.kind == CFTRT_LIGHT_SKY || L.kind == CFTRT_LIGHT_SHAPE) s->n_look++;

  std::vector<float4> sphF, msphF, msphB, triF, mtriF;
  double rb_max = 0.;   // largest swept-bound radius of a moving sphere
  std::vector<double> msph_rb;   // swept-bound radii (for the grid)
  std::vector<float> sphR2;
  int prec_motion = 0;
  // scene centre: bbox centre of every non-precise primitive; the f32 lists are stored relative to it
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  auto grow = [&](const double* p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } };
  for (int i = 0; i < ns; ++i) if (std::fabs(s->sph_r[i]) < s->precise_radius) grow(&s->sph_c[3 * i]);
  for (int i = 0; i < nt; ++i) { grow(&s->tri_p[9 * i]); grow(&s->tri_p[9 * i + 3]); grow(&s->tri_p[9 * i + 6]); }
  float bcf[3] = {0.f, 0.f, 0.f};
  if (lo[0] <= hi[0]) for (int k = 0; k < 3; ++k) bcf[k] = (float)(0.5 * (lo[k] + hi[k]));
  const double bc[3] = {(double)bcf[0], (double)bcf[1], (double)bcf[2]};
  std::vector<int> sphF_gid, msph_gid, prec_gid, triF_gid, mtri_gid, prim_obj(ns + nt), prim_mat(ns + nt);
  std::vector<double> sphD(4 * (size_t)ns);
  int any_motion = 0;
  for (int i = 0; i < ns; ++i) {
    const double* c = &s->sph_c[3 * i];
    const double* mv = &s->sph_mv[3 * i];
    double r = s->sph_r[i];
    sphD[4 * i] = c[0]; sphD[4 * i + 1] = c[1]; sphD[4 * i + 2] = c[2]; sphD[4 * i + 3] = r;
    prim_obj[i] = s->sph_obj[i];
    prim_mat[i] = s->sph_mat[i];
    bool moving = mv[0] != 0. || mv[1] != 0. || mv[2] != 0.;
    if (moving) any_motion = 1;
    if (std::fabs(r) >= s->precise_radius) {
      prec_gid.push_back(i);
      if (moving) prec_motion = 1;
    } else if (moving) {
      float objbits;
      memcpy(&objbits, &s->sph_obj[i], 4);
      msphF.push_back(make_float4((float)(c[0] - bc[0]), (float)(c[1] - bc[1]), (float)(c[2] - bc[2]), (float)(r * r)));
      msphF.push_back(make_float4((float)mv[0], (float)mv[1], (float)mv[2], objbits));
      msph_gid.push_back(i);
      {
        // swept bound in the static list's format: the centre is c0 + t mv, t in [0, 1), evaluated in f32 by the
        // kernel from the f32 values stored above; a sphere of radius |r| + |mv|/2 (+ slack far above those
        // roundings) around c0 + mv/2 contains it for every t
        const float4 c0 = msphF[msphF.size() - 2], m4 = msphF[msphF.size() - 1];
        const double bx = (double)c0.x + 0.5 * m4.x, by = (double)c0.y + 0.5 * m4.y, bz = (double)c0.z + 0.5 * m4.z;
        const double ml = std::sqrt((double)m4.x * m4.x + (double)m4.y * m4.y + (double)m4.z * m4.z);
        const double cl = std::sqrt((double)c0.x * c0.x + (double)c0.y * c0.y + (double)c0.z * c0.z);
        const double rb = (std::fabs(r) + 0.5 * ml) * (1. + 1e-5) + 1e-5 * (cl + ml) + 1e-7;
        rb_max = std::max(rb_max, rb);
        msph_rb.push_back(rb);
        msphB.push_back(make_float4((float)(-2. * bx), (float)(-2. * by), (float)(-2. * bz),
                                    (float)(4. * (bx * bx + by * by + bz * bz - rb * rb))));
      }
    } else {
      const double cx = c[0] - bc[0], cy = c[1] - bc[1], cz = c[2] - bc[2];
      sphF.push_back(make_float4((float)(-2. * cx), (float)(-2. * cy), (float)(-2. * cz),
                                 (float)(4. * (cx * cx + cy * cy + cz * cz - r * r))));
      sphR2.push_back(16.0f * (float)(r * r));   // the candidate path works on 16 disc (exact scaling)
      sphF_gid.push_back(i);
    }
  }
  for (int i = 0; i < nt; ++i) {
    const double* p = &s->tri_p[9 * i];
    const double* mv = &s->tri_mv[3 * i];
    prim_obj[ns + i] = s->tri_obj[i];
    prim_mat[ns + i] = s->tri_mat[i];
    bool moving = mv[0] != 0. || mv[1] != 0. || mv[2] != 0.;
    float4 P0 = make_float4((float)(p[0] - bc[0]), (float)(p[1] - bc[1]), (float)(p[2] - bc[2]), 0.f);
    float4 E1 = make_float4((float)(p[3] - p[0]), (float)(p[4] - p[1]), (float)(p[5] - p[2]), 0.f);
    float4 E2 = make_float4((float)(p[6] - p[0]), (float)(p[7] - p[1]), (float)(p[8] - p[2]), 0.f);
    if (moving) {
      any_motion = 1;
      mtriF.push_back(P0); mtriF.push_back(E1); mtriF.push_back(E2);
      float objbits;
      memcpy(&objbits, &s->tri_obj[i], 4);
      mtriF.push_back(make_float4((float)mv[0], (float)mv[1], (float)mv[2], objbits));
      mtri_gid.push_back(ns + i);
    } else {
      triF.push_back(P0); triF.push_back(E1); triF.push_back(E2);
      triF_gid.push_back(ns + i);
    }
  }
  // Primitives much larger than the rest (the three r = 1 spheres of scene.rs among r = 0.2 ones, the walls of a room)
  // would stretch the grid's box and sit in a great many cells.  Each list is reordered so that they come first:
  // with the grid on, that short prefix is scanned linearly by every ray and only the rest goes into the grid.
  // Big = extent above 4x the median extent, at most 64 per scene (the largest ones).  List order is free: exact
  // ties are broken by object id, not by position.
  int nb_sph = 0, nb_msph = 0, nb_tri = 0;
  {
    const int nS = (int)sphF_gid.size(), nM = (int)msph_gid.size(), nT = (int)triF_gid.size();
    std::vector<double> ext((size_t)(nS + nM + nT));
    for (int i = 0; i < nS; ++i) ext[i] = 0.5 * std::sqrt((double)sphR2[i]);
    for (int i = 0; i < nM; ++i) ext[nS + i] = 2. * msph_rb[i];
    for (int i = 0; i < nT; ++i) {
      const float4 e1 = triF[3 * i + 1], e2 = triF[3 * i + 2];
      double m = 0.;
      const double ex[3][2] = {{e1.x, e2.x}, {e1.y, e2.y}, {e1.z, e2.z}};
      for (int k = 0; k < 3; ++k)
        m = std::max(m, std::max(0., std::max(ex[k][0], ex[k][1])) - std::min(0., std::min(ex[k][0], ex[k][1])));
      ext[nS + nM + i] = m;
    }
    if (ext.size() >= 64) {
      std::vector<double> sorted(ext);
      std::nth_element(sorted.begin(), sorted.begin() + sorted.size() / 2, sorted.end());
      double thr = 4. * sorted[sorted.size() / 2];
      std::vector<double> bigs;
      for (double e : ext) if (e > thr) bigs.push_back(e);
      if (bigs.size() > 64) {
        std::nth_element(bigs.begin(), bigs.end() - 64, bigs.end());
        thr = std::max(thr, *(bigs.end() - 64));
      }
      // stable partition of each list, big first
      auto order = [&](int base, int n, int* nb) {
        std::vector<int> o;
        for (int i = 0; i < n; ++i) if (ext[base + i] > thr) o.push_back(i);
        *nb = (int)o.size();
        for (int i = 0; i < n; ++i) if (!(ext[base + i] > thr)) o.push_back(i);
        return o;
      };
      auto apply = [](const std::vector<int>& o, auto& v, int per) {
        auto old = v;
        for (size_t i = 0; i < o.size(); ++i)
          for (int k = 0; k < per; ++k) v[per * i + k] = old[per * (size_t)o[i] + k];
      };
      std::vector<int> o = order(0, nS, &nb_sph);
      apply(o, sphF, 1); apply(o, sphR2, 1); apply(o, sphF_gid, 1);
      o = order(nS, nM, &nb_msph);
      apply(o, msphF, 2); apply(o, msphB, 1); apply(o, msph_rb, 1); apply(o, msph_gid, 1);
      o = order(nS + nM, nT, &nb_tri);
      apply(o, triF, 3); apply(o, triF_gid, 1);
    }
  }
  // The static triangles as scan ITEMS (kernels.h, tri_item_bounds): two neighbours of the list that are the two triangles
  // of a Square::new - (p0, p1, p2) and (p2, p3, p0) of one object with p3 = p0 + p2 - p1, the corner order of
  // object.rs:148-161 - become ONE parallelogram item (p0; ea = p1 - p0, eb = p3 - p0): one f32 test instead of two, whose
  // result names the triangle(s) to hand to the exact evaluation.  Everything else (and the second triangle of a
  // Square::from_points, which is no parallelogram half) stays a single-triangle item.
  std::vector<float4> triItem;
  std::vector<int> item_first;   // index in triF of each item's first triangle
  std::vector<char> item_quad;
  int nb_item = 0;
  {
    const int nT = (int)triF_gid.size();
    for (int i = 0; i < nT;) {
      bool quad = false;
      float4 EB = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i + 1 < nT && (i < nb_tri) == (i + 1 < nb_tri)) {
        const int ti = triF_gid[i] - ns, tj = triF_gid[i + 1] - ns;
        const double* P = &s->tri_p[9 * (size_t)ti];
        const double* Q = &s->tri_p[9 * (size_t)tj];
        const bool flagged = !s->tri_first.empty() && (s->tri_first[(size_t)ti] >= 0 || s->tri_first[(size_t)tj] >= 0);
        if (tj == ti + 1 && s->tri_obj[(size_t)ti] == s->tri_obj[(size_t)tj] && !flagged && !memcmp(Q, P + 6, 24) && !memcmp(Q + 6, P, 24)) {
          double dev = 0., scale = 0.;
          for (int k = 0; k < 3; ++k) {
            dev = std::max(dev, std::fabs(Q[3 + k] - (P[k] + P[6 + k] - P[3 + k])));
            scale = std::max(scale, std::max(std::fabs(P[3 + k] - P[k]), std::fabs(P[6 + k] - P[k])));
          }
          if (dev <= 1e-10 * scale) {
            quad = true;
            EB = make_float4((float)(Q[3] - P[0]), (float)(Q[4] - P[1]), (float)(Q[5] - P[2]), 0.f);
          }
        }
      }
      const int meta = (i << 1) | (quad ? 1 : 0);
      float4 P0 = triF[3 * (size_t)i];
      memcpy(&P0.w, &meta, 4);
      triItem.push_back(P0);
      triItem.push_back(triF[3 * (size_t)i + 1]);
      triItem.push_back(quad ? EB : triF[3 * (size_t)i + 2]);
      item_first.push_back(i);
      item_quad.push_back(quad ? 1 : 0);
      if (i < nb_tri) nb_item++;
      i += quad ? 2 : 1;
    }
  }
  msphF.push_back(make_float4(0.f, 0.f, 0.f, -1e30f));   // the prefetched extra entry of the moving list
  msphF.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
  while (msphB.size() % 4) msphB.push_back(make_float4(0.f, 0.f, 0.f, 1e30f));
  for (int k = 0; k < 4; ++k) msphB.push_back(make_float4(0.f, 0.f, 0.f, 1e30f));
  while (msphF.size() < 2 * msphB.size()) msphF.push_back(make_float4(0.f, 0.f, 0.f, 0.f));   // entries the pad bounds point at
  const int n_sphF_real = (int)sphF_gid.size();
  while (sphF.size() % 4) { sphF.push_back(make_float4(0.f, 0.f, 0.f, 1e30f)); sphR2.push_back(0.f); }   // never hit: 4 disc' = Q^2 - 1e30
  for (int k = 0; k < 4; ++k) { sphF.push_back(make_float4(0.f, 0.f, 0.f, 1e30f)); sphR2.push_back(0.f); }   // the prefetched extra batch
  // Uniform grid over the static spheres, the moving spheres' swept bounds and the static triangles (kernels.h,
  // grid_walk_ray).  Cubic cells sized for ~1 cell per primitive (measured optimum 0.5 .. 2), at most 128 per axis; every primitive is
  // registered in the cells its bound, inflated by 1e-3 of the box half-extent, overlaps.  Scenes with fewer than 64
  // such primitives, or whose primitives are large against any sensible cell (a Cornell box), get no grid.
  std::vector<uint2> gridCell;   // per cell: (first sphere entry << 8) | count, (first triangle entry << 8) | count
  std::vector<float4> gridSph;
  std::vector<float2> gridSphX;
  std::vector<uint32_t> gridTri;
  std::vector<float4> boundF;
  struct { int n[3]; double lo[3], hi[3], cs[3], far; } G = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, 0};
  {
    const int nS = n_sphF_real, nM = (int)msph_gid.size(), nT = (int)triF_gid.size(), nP = nS + nM + nT;
    struct Box { double lo[3], hi[3]; };
    std::vector<Box> bx((size_t)nP);
    auto in_grid = [&](int i) { return i < nS ? i >= nb_sph : (i < nS + nM ? i - nS >= nb_msph : i - nS - nM >= nb_tri); };
    const int nG = nP - nb_sph - nb_msph - nb_tri;   // primitives that go into the grid
    std::vector<double> cen(3 * (size_t)(nS + nM)), rad((size_t)(nS + nM));
    for (int i = 0; i < nS + nM; ++i) {
      const float4 e = i < nS ? sphF[i] : msphB[i - nS];
      const double c[3] = {-0.5 * e.x, -0.5 * e.y, -0.5 * e.z};
      const double r = i < nS ? 0.25 * std::sqrt((double)sphR2[i]) : msph_rb[i - nS];
      for (int k = 0; k < 3; ++k) { cen[3 * i + k] = c[k]; bx[i].lo[k] = c[k] - r; bx[i].hi[k] = c[k] + r; }
      rad[i] = r;
    }
    std::vector<double> triV(9 * (size_t)nT);   // corners of the static triangles as the kernels' f32 data give them
    for (int i = 0; i < nT; ++i) {
      const float4 p0 = triF[3 * i], e1 = triF[3 * i + 1], e2 = triF[3 * i + 2];
      const double v[3][3] = {{p0.x, p0.y, p0.z}, {(double)p0.x + e1.x, (double)p0.y + e1.y, (double)p0.z + e1.z},
                              {(double)p0.x + e2.x, (double)p0.y + e2.y, (double)p0.z + e2.z}};
      for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) triV[9 * (size_t)i + 3 * j + k] = v[j][k];
      Box& b = bx[nS + nM + i];
      for (int k = 0; k < 3; ++k) {
        b.lo[k] = std::min(v[0][k], std::min(v[1][k], v[2][k]));
        b.hi[k] = std::max(v[0][k], std::max(v[1][k], v[2][k]));
      }
    }
    // bounding spheres of the listed primitives, in list order (the per-tile lists of the primary rays: k_tile_bin;
    // also what the cone test of a tile list reads), with slack far above the f32 rounding of the stored values
    boundF.resize((size_t)nP);
    for (int i = 0; i < nS + nM; ++i) {
      const double cl = std::sqrt(cen[3 * i] * cen[3 * i] + cen[3 * i + 1] * cen[3 * i + 1] + cen[3 * i + 2] * cen[3 * i + 2]);
      boundF[i] = make_float4((float)cen[3 * i], (float)cen[3 * i + 1], (float)cen[3 * i + 2],
                              (float)((rad[i] + 1e-6 * (cl + rad[i]) + 1e-7) * (1. + 1e-6)));
    }
    for (int i = 0; i < nT; ++i) {
      const float4 p0 = triF[3 * i], e1 = triF[3 * i + 1], e2 = triF[3 * i + 2];
      const double v[3][3] = {{p0.x, p0.y, p0.z}, {(double)p0.x + e1.x, (double)p0.y + e1.y, (double)p0.z + e1.z},
                              {(double)p0.x + e2.x, (double)p0.y + e2.y, (double)p0.z + e2.z}};
      double g[3], r2 = 0.;
      for (int k = 0; k < 3; ++k) g[k] = (v[0][k] + v[1][k] + v[2][k]) / 3.;
      for (int j = 0; j < 3; ++j) {
        double d2 = 0.;
        for (int k = 0; k < 3; ++k) d2 += (v[j][k] - g[k]) * (v[j][k] - g[k]);
        r2 = std::max(r2, d2);
      }
      const double gl = std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]), r = std::sqrt(r2);
      boundF[(size_t)nS + nM + i] = make_float4((float)g[0], (float)g[1], (float)g[2], (float)((r + 1e-6 * (gl + r) + 1e-7) * (1. + 1e-6)));
    }
    if (nG >= 64) {
      double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
      for (int i = 0; i < nP; ++i) if (in_grid(i)) for (int k = 0; k < 3; ++k) { blo[k] = std::min(blo[k], bx[i].lo[k]); bhi[k] = std::max(bhi[k], bx[i].hi[k]); }
      double E = 0.;
      for (int k = 0; k < 3; ++k) E = std::max(E, std::max(std::fabs(blo[k]), std::fabs(bhi[k])));
      const double margin = 1e-3 * E;
      double ext[3], vol = 1.;
      for (int k = 0; k < 3; ++k) { blo[k] -= 2. * margin; bhi[k] += 2. * margin; ext[k] = bhi[k] - blo[k]; vol *= ext[k]; }
      double cell = std::cbrt(vol / (ctx->grid_density * nG));
      for (int attempt = 0; attempt < 4 && !G.n[0]; ++attempt, cell *= 2.) {
        int n[3];
        double cs[3];
        for (int k = 0; k < 3; ++k) {
          n[k] = (int)std::min(128., std::max(1., std::ceil(ext[k] / cell)));
          cs[k] = ext[k] / n[k];
        }
        const size_t nc = (size_t)n[0] * n[1] * n[2];
        std::vector<uint32_t> cntS(nc, 0), cntT(nc, 0);
        size_t total = 0;
        // two passes over the same enumeration: count, then fill
        // i < nS + nM: sphere i of the lists; otherwise item i - nS - nM of the static triangles (its one or two triangles)
        auto each_cell = [&](int i, auto&& f) {
          int a[3], b[3];
          const int t0 = i < nS + nM ? -1 : item_first[(size_t)(i - nS - nM)];
          const int t1 = t0 >= 0 && item_quad[(size_t)(i - nS - nM)] ? t0 + 1 : t0;
          for (int k = 0; k < 3; ++k) {
            const double lo_k = t0 < 0 ? bx[i].lo[k] : std::min(bx[nS + nM + t0].lo[k], bx[nS + nM + t1].lo[k]);
            const double hi_k = t0 < 0 ? bx[i].hi[k] : std::max(bx[nS + nM + t0].hi[k], bx[nS + nM + t1].hi[k]);
            a[k] = std::min(n[k] - 1, std::max(0, (int)std::floor((lo_k - margin - blo[k]) / cs[k])));
            b[k] = std::min(n[k] - 1, std::max(0, (int)std::floor((hi_k + margin - blo[k]) / cs[k])));
          }
          for (int z = a[2]; z <= b[2]; ++z)
            for (int y = a[1]; y <= b[1]; ++y)
              for (int x = a[0]; x <= b[0]; ++x) {
                if (i < nS + nM) {   // sphere against the cell box, not just its own box against it
                  const int c3[3] = {x, y, z};
                  double d2 = 0.;
                  for (int k = 0; k < 3; ++k) {
                    const double lo_k = blo[k] + c3[k] * cs[k], hi_k = lo_k + cs[k], c = cen[3 * i + k];
                    const double d = c < lo_k ? lo_k - c : (c > hi_k ? c - hi_k : 0.);
                    d2 += d * d;
                  }
                  const double rr = rad[i] + margin;
                  if (d2 > rr * rr) continue;
                } else {   // triangle against the cell box (widened by the margin), not just its own box against it
                  const int c3[3] = {x, y, z};
                  double bc3[3], bh[3];
                  for (int k = 0; k < 3; ++k) { bc3[k] = blo[k] + (c3[k] + 0.5) * cs[k]; bh[k] = 0.5 * cs[k] + margin; }
                  if (!tri_box_overlap(bc3, bh, &triV[9 * (size_t)t0]) && (t1 == t0 || !tri_box_overlap(bc3, bh, &triV[9 * (size_t)t1]))) continue;
                }
                f(((size_t)z * n[1] + y) * n[0] + x);
              }
        };
        const int nUnits = nS + nM + (int)item_first.size();
        auto unit_in_grid = [&](int i) { return i < nS + nM ? in_grid(i) : in_grid(nS + nM + item_first[(size_t)(i - nS - nM)]); };
        for (int i = 0; i < nUnits; ++i) if (unit_in_grid(i)) each_cell(i, [&](size_t c) { (i < nS + nM ? cntS : cntT)[c]++; total++; });
        if (total > 24 * (size_t)nG) continue;   // primitives much larger than the cells: coarser
        {   // the packed record holds 8 bits of count and 24 of offset
          uint32_t cmax = 0;
          size_t tS = 0, tT = 0;
          for (size_t c = 0; c < nc; ++c) { cmax = std::max(cmax, std::max(cntS[c], cntT[c])); tS += cntS[c]; tT += cntT[c]; }
          if (cmax > 255u) continue;
          if (tS >= (1u << 24) || tT >= (1u << 20)) break;   // (no grid: lists)
        }
        gridCell.resize(nc);
        std::vector<uint32_t> offS(nc), offT(nc);
        uint32_t oS = 0, oT = 0;
        for (size_t c = 0; c < nc; ++c) { offS[c] = oS; offT[c] = oT; gridCell[c] = make_uint2(oS << 8, oT << 12); oS += cntS[c]; oT += cntT[c]; }
        gridSph.resize((size_t)oS);
        gridSphX.resize((size_t)oS);
        gridTri.resize(oT);
        for (int i = 0; i < nUnits; ++i)
          if (unit_in_grid(i)) each_cell(i, [&](size_t c) {
            if (i < nS + nM) {
              const size_t e = (size_t)offS[c] + (gridCell[c].x++ & 255u);
              float ib;
              memcpy(&ib, &i, 4);
              gridSph[e] = i < nS ? sphF[i] : msphB[i - nS];
              gridSphX[e] = make_float2(i < nS ? sphR2[i] : 0.f, ib);
            } else {
              { gridTri[(size_t)offT[c] + ((gridCell[c].y >> 4) & 255u)] = (uint32_t)(i - nS - nM); gridCell[c].y += 16u; }
            }
          });
        for (int k = 0; k < 3; ++k) { G.n[k] = n[k]; G.lo[k] = blo[k]; G.hi[k] = bhi[k]; G.cs[k] = cs[k]; }
        G.far = 256. * E;
      }
    }
    if (gridSph.empty()) { gridSph.push_back(make_float4(0.f, 0.f, 0.f, 0.f)); gridSphX.push_back(make_float2(0.f, 0.f)); }
    if (gridTri.empty()) gridTri.push_back(0u);
    if (gridCell.empty()) gridCell.push_back(make_uint2(0, 0));
  }
  std::vector<MatDev> mats(nm);
  int n_phong = 0;
  for (int i = 0; i < nm; ++i) {
    const cftrt_material& m = s->mats[i];
    mats[i] = MatDev{m.kind, (float)m.p0, (float)m.p1, (float)m.color[0], (float)m.color[1], (float)m.color[2], 0.f, 0.f};
    if (m.kind == CFTRT_MAT_PHONG) n_phong++;
  }
  std::vector<LightDev> lights(s->lights.size());
  for (size_t i = 0; i < s->lights.size(); ++i) {
    const cftrt_light& L = s->lights[i];
    lights[i] = LightDev{L.kind, L.sph_begin, L.sph_count, L.tri_begin, L.tri_count, (float)L.v[0], (float)L.v[1],
                         (float)L.v[2], (float)L.color[0], (float)L.color[1], (float)L.color[2]};
    if (L.kind == CFTRT_LIGHT_SHAPE) { lights[i].r = lights[i].g = lights[i].b = 1.0f; }   // light.rs:198-204
  }
  SceneDev& D = s->dev;
  memset(&D, 0, sizeof D);
  int rc;
  if ((rc = upload(s, sphF, &D.sphF))) return rc;
  if ((rc = upload(s, sphF_gid, &D.sphF_gid))) return rc;
  if ((rc = upload(s, sphR2, &D.sphR2))) return rc;
  if ((rc = upload(s, msphF, &D.msphF))) return rc;
  if ((rc = upload(s, msphB, &D.msphB))) return rc;
  if ((rc = upload(s, msph_gid, &D.msph_gid))) return rc;
  if ((rc = upload(s, prec_gid, &D.prec_gid))) return rc;
  if ((rc = upload(s, triF, &D.triF))) return rc;
  if ((rc = upload(s, triF_gid, &D.triF_gid))) return rc;
  if ((rc = upload(s, triItem, &D.triItem))) return rc;
  if ((rc = upload(s, mtriF, &D.mtriF))) return rc;
  if ((rc = upload(s, mtri_gid, &D.mtri_gid))) return rc;
  if ((rc = upload(s, gridCell, &D.gridCell))) return rc;
  if ((rc = upload(s, gridSph, &D.gridSph))) return rc;
  if ((rc = upload(s, gridSphX, &D.gridSphX))) return rc;
  if ((rc = upload(s, gridTri, &D.gridTri))) return rc;
  {
    // the sphere part of the grid as one image for the walk instances that keep it in LDS (scene_dev.h): only when
    // it fits and no cell lists a triangle
    bool tri_entries = false;
    for (const uint2& c : gridCell) tri_entries = tri_entries || ((c.y >> 4) & 255u) != 0u;
    const size_t nent = gridSph.size(), ncell = gridCell.size();
    const size_t bytes = (nent * 24 + ncell * 8 + 15) & ~(size_t)15;
    if (G.n[0] > 0 && !tri_entries && bytes <= (size_t)CFT_LDS_GRID_BYTES) {
      std::vector<uint4> img(bytes / 16, make_uint4(0u, 0u, 0u, 0u));
      unsigned char* b = (unsigned char*)img.data();
      memcpy(b, gridSph.data(), nent * 16);
      memcpy(b + nent * 16, gridCell.data(), ncell * 8);
      memcpy(b + nent * 16 + ncell * 8, gridSphX.data(), nent * 8);
      if ((rc = upload(s, img, &D.gridImg))) return rc;
      D.gridImgBytes = (uint32_t)bytes;
      D.gridImgEnt = (uint32_t)nent;
    }
  }
  if (boundF.empty()) boundF.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
  if ((rc = upload(s, boundF, &D.boundF))) return rc;
  D.tile_off = nullptr;
  D.tile_idx = nullptr;
  {   // list-ordered copies for the exact evaluation of candidates (scene_dev.h)
    std::vector<double> lsD(4 * sphF_gid.size()), lmD(8 * msph_gid.size());
    std::vector<int4> listRec;
    for (size_t i = 0; i < sphF_gid.size(); ++i) memcpy(&lsD[4 * i], &sphD[4 * (size_t)sphF_gid[i]], 32);
    for (size_t i = 0; i < msph_gid.size(); ++i) {
      memcpy(&lmD[8 * i], &sphD[4 * (size_t)msph_gid[i]], 32);
      memcpy(&lmD[8 * i + 4], &s->sph_mv[3 * (size_t)msph_gid[i]], 24);
      lmD[8 * i + 7] = 0.;
    }
    for (const std::vector<int>* v : {&sphF_gid, &msph_gid, &triF_gid, &mtri_gid})
      for (int g : *v) listRec.push_back(make_int4(g, prim_obj[g], prim_mat[g], 0));
    if ((rc = upload(s, lsD, &D.lsD))) return rc;
    if ((rc = upload(s, lmD, &D.lmD))) return rc;
    if ((rc = upload(s, listRec, &D.listRec))) return rc;
    D.prec_g0 = -1;
    if (!prec_gid.empty()) {
      const int g = prec_gid[0];
      const double* mv = &s->sph_mv[3 * (size_t)g];
      if (mv[0] == 0. && mv[1] == 0. && mv[2] == 0.) {
        D.prec_g0 = g; D.prec_mat0 = prim_mat[g];
        D.prec_c0x = sphD[4 * (size_t)g]; D.prec_c0y = sphD[4 * (size_t)g + 1]; D.prec_c0z = sphD[4 * (size_t)g + 2];
        D.prec_r0 = sphD[4 * (size_t)g + 3];
      }
    }
  }
  if ((rc = upload(s, sphD, &D.sphD))) return rc;
  if ((rc = upload(s, s->sph_mv, &D.sphMvD))) return rc;
  if ((rc = upload(s, s->tri_p, &D.triD))) return rc;
  if ((rc = upload(s, s->tri_mv, &D.triMvD))) return rc;
  if ((rc = upload(s, tri_first, &D.tri_first))) return rc;
  if ((rc = upload(s, ltri_first, &D.ltri_first))) return rc;
  D.any_pairs = any_pairs;
  if ((rc = upload(s, prim_obj, &D.prim_obj))) return rc;
  if ((rc = upload(s, prim_mat, &D.prim_mat))) return rc;
  if ((rc = upload(s, mats, &D.mats))) return rc;
  if ((rc = upload(s, lights, &D.lights))) return rc;
  if ((rc = upload(s, s->lsph, &D.lsphD))) return rc;
  if ((rc = upload(s, s->ltri, &D.ltriD))) return rc;
  D.n_sphF = n_sphF_real;
  D.n_msph = (int)msph_gid.size();
  D.n_prec = (int)prec_gid.size();
  D.n_triF = (int)triF_gid.size();
  D.n_mtri = (int)mtri_gid.size();
  D.n_sph = ns;
  D.n_tri = nt;
  D.n_mat = nm;
  D.n_lights = (int)lights.size();
  D.any_motion = any_motion;
  s->multi_look = any_motion && s->n_look >= 2;
  if (s->multi_look && s->n_look > CFT_MAX_LOOK)
    return fail(ctx, CFTRT_ERR_LIMIT, "a scene with moving objects may hold at most 8 lights that look (SkyLight, LightShape)");
  D.prec_motion = prec_motion;
  D.n_phong_mats = n_phong;
  D.gnx = G.n[0]; D.gny = G.n[1]; D.gnz = G.n[2];
  D.nb_sph = nb_sph; D.nb_msph = nb_msph; D.nb_tri = nb_tri;
  D.n_item = (int)item_first.size(); D.nb_item = nb_item;
  D.grid_mode = 0;   // set per pass from the ctx (cftrt_ctx_set_grid)
  if (G.n[0]) {
    // the box as the kernel sees it: low corner rounded down, high corner up (the box only grows)
    D.g0x = std::nextafterf((float)G.lo[0], -INFINITY); D.g0y = std::nextafterf((float)G.lo[1], -INFINITY); D.g0z = std::nextafterf((float)G.lo[2], -INFINITY);
    D.g1x = std::nextafterf((float)G.hi[0], INFINITY); D.g1y = std::nextafterf((float)G.hi[1], INFINITY); D.g1z = std::nextafterf((float)G.hi[2], INFINITY);
    D.gcsx = (float)G.cs[0]; D.gcsy = (float)G.cs[1]; D.gcsz = (float)G.cs[2];
    D.gicx = (float)(1. / G.cs[0]); D.gicy = (float)(1. / G.cs[1]); D.gicz = (float)(1. / G.cs[2]);
    D.gfar4 = (float)(4. * G.far);
  }
  {
    // f32 error model of the cull.  eps = 2^-24 (unit roundoff); every constant carries
    // a factor >= 2 over the bound derived there.
    const double eps = 5.9604644775390625e-8;
    double rmax = 0., brad = 0.;
    for (int i = 0; i < ns; ++i) {
      if (std::fabs(s->sph_r[i]) >= s->precise_radius) continue;
      const double* c = &s->sph_c[3 * i];
      const double* mv = &s->sph_mv[3 * i];
      rmax = std::max(rmax, std::fabs(s->sph_r[i]));
      double dx = c[0] - bc[0], dy = c[1] - bc[1], dz = c[2] - bc[2];
      brad = std::max(brad, std::sqrt(dx * dx + dy * dy + dz * dz) + std::sqrt(mv[0] * mv[0] + mv[1] * mv[1] + mv[2] * mv[2]));
    }
    const double cmax = brad;   // largest |c'| of the f32 lists
    D.K1 = (float)(48. * eps * rmax);
    D.K2 = (float)(eps * (rmax * (72. * rmax + 24. * cmax)));
    D.K3 = (float)(12. * eps);
    D.K4 = (float)(eps * (24. * rmax + 6. * cmax));
    D.Kc = (float)(36. * eps);
    D.rmax2 = (float)(std::max(rmax, rb_max) * std::max(rmax, rb_max));   // the cull also runs on the moving spheres' swept bounds
    D.bcx = bcf[0]; D.bcy = bcf[1]; D.bcz = bcf[2];
    D.brad = (float)(brad * 1.000001 + 1e-6);
    double emax = 0., pmax = 0.;
    for (int i = 0; i < nt; ++i) {
      const double* p = &s->tri_p[9 * i];
      const double* mv = &s->tri_mv[3 * i];
      double e1 = std::sqrt((p[3] - p[0]) * (p[3] - p[0]) + (p[4] - p[1]) * (p[4] - p[1]) + (p[5] - p[2]) * (p[5] - p[2]));
      double e2 = std::sqrt((p[6] - p[0]) * (p[6] - p[0]) + (p[7] - p[1]) * (p[7] - p[1]) + (p[8] - p[2]) * (p[8] - p[2]));
      emax = std::max(emax, std::max(e1, e2));
      double qx = p[0] - bc[0], qy = p[1] - bc[1], qz = p[2] - bc[2];
      pmax = std::max(pmax, std::sqrt(qx * qx + qy * qy + qz * qz) + std::sqrt(mv[0] * mv[0] + mv[1] * mv[1] + mv[2] * mv[2]));
    }
    D.tKA = (float)(32. * eps * emax * emax);
    D.tKU1 = (float)(32. * eps * emax);
    D.tKU2 = (float)(32. * eps * emax * (pmax + emax));
    D.tKT1 = (float)(32. * eps * emax * emax);
    D.tKT2 = (float)(32. * eps * emax * emax * (pmax + emax));
    D.tKT3 = (float)(32. * eps * emax * emax);
  }
  if (getenv("CFTRT_DEBUG")) {
    size_t emptyc = 0;
    uint32_t cmaxS = 0, cmaxT = 0;
    for (const uint2& c : gridCell) {
      if (!(c.x & 255u) && !((c.y >> 4) & 255u)) emptyc++;
      cmaxS = std::max(cmaxS, c.x & 255u); cmaxT = std::max(cmaxT, (c.y >> 4) & 255u);
    }
    fprintf(stderr, "cftrt commit: lists sphF %d msph %d triF %d mtri %d prec %d | big prefix sph %d msph %d tri %d | grid %d x %d x %d, "
                    "%zu cells (%zu empty), %zu sphere entries, %zu triangle entries, max per cell %u / %u\n",
            D.n_sphF, D.n_msph, D.n_triF, D.n_mtri, D.n_prec, nb_sph, nb_msph, nb_tri, D.gnx, D.gny, D.gnz, gridCell.size(), emptyc,
            gridSph.size(), gridTri.size(), cmaxS, cmaxT);
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  static std::atomic<uint64_t> next_serial{1};
  s->serial = next_serial.fetch_add(1);
  s->committed = true;
  return CFTRT_OK;
}

// ------------------------------------------------------------------ camera
static void v_unit(double* v) {
  double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  v[0] /= l; v[1] /= l; v[2] /= l;
}
// Viewport constants of Camera::emit_rays (ray.rs:112-118), in f64.
static int make_cam(cftrt_ctx* ctx, const cftrt_camera* c, uint32_t width, uint32_t height, CamDev* out) {
  if (!c) return fail(ctx, CFTRT_ERR_INVALID, "null camera");
  if (width == 0 || height == 0) return fail(ctx, CFTRT_ERR_INVALID, "zero image size");
  const double* s = c->sight;
  const double* u = c->up;
  double right[3] = {s[1] * u[2] - s[2] * u[1], s[2] * u[0] - s[0] * u[2], s[0] * u[1] - s[1] * u[0]};
  v_unit(right);
  double vh = 2. * std::tan(c->fov_rad / 2.) * c->focus_dist;
  double vw = vh * c->aspect;
  double pw[3], ph[3], tl[3];
  for (int k = 0; k < 3; ++k) {
    pw[k] = vw / (double)width * right[k];
    ph[k] = vh / (double)height * u[k];
  }
  for (int k = 0; k < 3; ++k) {
    double center = c->pos[k] + c->focus_dist * s[k];
    double bias = 0.5 * (pw[k] - ph[k]);
    tl[k] = center - vw * right[k] / 2. + vh * u[k] / 2. + bias;
  }
  CamDev C;
  C.tlx = (float)tl[0]; C.tly = (float)tl[1]; C.tlz = (float)tl[2];
  C.pwx = (float)pw[0]; C.pwy = (float)pw[1]; C.pwz = (float)pw[2];
  C.phx = (float)ph[0]; C.phy = (float)ph[1]; C.phz = (float)ph[2];
  C.px = (float)c->pos[0]; C.py = (float)c->pos[1]; C.pz = (float)c->pos[2];
  C.rx = (float)right[0]; C.ry = (float)right[1]; C.rz = (float)right[2];
  C.ux = (float)u[0]; C.uy = (float)u[1]; C.uz = (float)u[2];
  C.lens_radius = (float)(c->aperture / 2.);
  C.width = width;
  C.height = height;
  *out = C;
  return CFTRT_OK;
}

// ------------------------------------------------------------------ the trace loop
static uint32_t grid_for(cftrt_ctx* ctx, uint64_t n, uint32_t per_cu) {
  uint64_t blocks = (n + CFT_BLOCK - 1) / CFT_BLOCK;
  uint64_t cap = (uint64_t)ctx->n_cu * per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks == 0) blocks = 1;
  return (uint32_t)blocks;
}


// k_intersect<R, NS>: R rays per lane (ctx->variant, 0 = default), NS scans per ray in one pass.
static void launch_intersect(cftrt_ctx* ctx, const SceneDev& S, QueueDev q, const uint32_t* counts, uint32_t level,
                             uint64_t pid_base, uint32_t k0, uint32_t k1, bool two_scans, uint32_t role0, uint32_t role1,
                             HitDev out0, HitDev out1, uint32_t* xstats, uint32_t P) {
  int R = ctx->variant ? ctx->variant : CFT_RAYS_PER_LANE;
  // few live rays (deep levels): fewer rays per lane spread them over more waves; below one wave per SIMD the
  // launch time is one wave's latency through the list, which R = 2 halves against R = 4
  if (!ctx->variant && P < (1u << 20)) R = 2;
  if (two_scans && R > 2) R = 2;
  const uint32_t g = grid_for(ctx, ((uint64_t)P + R - 1) / R, 8);
  hipStream_t st = ctx->stream;
#define CFT_LAUNCH(RR, NN) hipLaunchKernelGGL((k_intersect<RR, NN>), dim3(g), dim3(CFT_BLOCK), 0, st, S, q, counts, level, \
                                              pid_base, k0, k1, role0, role1, out0, out1, xstats)
  if (two_scans) {
    if (R == 1) CFT_LAUNCH(1, 2); else CFT_LAUNCH(2, 2);
  } else {
    switch (R) {
      case 1: CFT_LAUNCH(1, 1); break;
      case 3: CFT_LAUNCH(3, 1); break;
      case 4: CFT_LAUNCH(4, 1); break;
      default: CFT_LAUNCH(2, 1); break;
    }
  }
#undef CFT_LAUNCH
}

// k_trace_level<R, NS>: the fused pipeline's one launch per level (same R / NS choice as launch_intersect).
// gen != null (level 0): the launch generates the primary rays itself (no k_generate).
static void launch_trace_level(cftrt_ctx* ctx, const SceneDev& S, QueueDev q, uint32_t* counts, uint32_t level,
                               uint32_t max_depth, uint64_t pid_base, uint32_t k0, uint32_t k1, bool two_scans,
                               uint32_t role0, uint32_t role1, QueueDev qout, Rad3* rad, PhongQueueDev pq,
                               uint32_t* phong_counts, uint32_t* xstats, uint32_t P, const GenArgs* gen, uint32_t in_D,
                               uint32_t* out_D_io, double* dbg_t = nullptr, int32_t* dbg_prim = nullptr, uint32_t cls_thr = 0) {
  // *out_D_io on entry: nonzero = the caller wants the output queue sharded; on return: the segment stride used (0: dense)
  const bool want_shards = *out_D_io != 0;
  *out_D_io = 0;
  int R = ctx->variant ? ctx->variant : CFT_RAYS_PER_LANE;
  if (!ctx->variant && P < (1u << 20)) R = 2;
  if (two_scans && R > 2) R = 2;
  hipStream_t st = ctx->stream;
  // static spheres only (S500 and the like): kernels compiled without the moving-sphere and triangle lists
  const bool lean = !S.n_msph && !S.n_triF && !S.n_mtri;
  // measured on S500: 2^18 .. 2^20 within 1 % on the lists; with the grid the level kernel stays ahead down to ~2^17
  const uint32_t tail_below = S.grid_mode ? 1u << 17 : 1u << 19;
  // (with the grid no wave scans a whole list; short lists are still quicker split over the four waves)
  if (!gen && !in_D && !ctx->variant && P < tail_below && (!S.grid_mode || S.n_sphF + S.n_msph + S.n_triF <= 2048)) {
    // too few rays to fill the chip: the list is split over the 4 waves of a block (k_trace_tail), 128 rays per block
    uint64_t blocks = ((uint64_t)P + 127) / 128;
    blocks = std::max<uint64_t>(1, std::min<uint64_t>(blocks, (uint64_t)ctx->n_cu * 8));
#define CFT_TAIL(NN, LL) hipLaunchKernelGGL((k_trace_tail<2, NN, LL>), dim3((uint32_t)blocks), dim3(CFT_BLOCK), 0, st, S, q, \
                                            counts, level, max_depth, pid_base, k0, k1, role0, role1, qout, rad, pq,       \
                                            phong_counts, xstats, dbg_t, dbg_prim)
    if (two_scans) CFT_TAIL(2, 3);
    else if (lean) CFT_TAIL(1, 0);
    else CFT_TAIL(1, 3);
#undef CFT_TAIL
    return;
  }
  // Sphere-only scenes whose grid image fits in LDS (S500: 35 KB): the big walk levels run the instance that keeps the
  // grid there - one 1024-thread block per CU, 3 rays per lane (kernels.h, k_trace_level).
  // (also scenes with moving spheres, whose swept bounds are the grid's entries: scene.rs - two scans per level when
  // the scene has lights, 2 rays per lane and a smaller cap then)
  const bool spheres_only = !S.n_triF && !S.n_mtri;
  const bool ldsg = !gen && spheres_only && !ctx->variant && S.grid_mode && S.gridImg && ctx->lds_grid && P >= ctx->lds_grid_above &&
                    (two_scans ? S.gridImgBytes <= (uint32_t)CFT_LDS_GRID_BYTES2 : true);
  if (ldsg) R = two_scans ? 2 : 3;   // (one scan at 2 rays per lane: 2 % slower)
  const uint32_t g = ldsg ? (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)P + 1024 * R - 1) / (1024 * R), (uint64_t)ctx->n_cu))
                          : grid_for(ctx, ((uint64_t)P + R - 1) / R, 8u);   // 8 blocks a CU (4 and 16/32 measured slower, r03)
  const GenArgs no_gen{};
  // (experiments: unused dynamic LDS per block lowers the number of resident blocks per CU)
  static const uint32_t dyn_lds = getenv("CFTRT_EXTRA_LDS") ? (uint32_t)atoi(getenv("CFTRT_EXTRA_LDS")) : 0u;
  // sharded output: trips of this launch <= ceil(P / 64 R) + one partial trip per input segment; segment `it mod 8`
  // receives the survivors of every 8th trip
  uint32_t out_k = 1, out_D = 0;
  if (want_shards) {
    const uint64_t trips = ((uint64_t)P + 64u * R - 1) / (64u * R) + CFT_NSEG;
    out_k = CFT_SHARDS;
    out_D = (uint32_t)((trips + CFT_SHARDS - 1) / CFT_SHARDS) * 64u * R;
    *out_D_io = out_D;
  } else if (cls_thr) {
    // one segment with both classes: the short rays from slot 0 upwards, the long ones from slot out_D downwards
    out_D = (uint32_t)std::min<uint64_t>(((uint64_t)P + 64u * R - 1) / (64u * R) * (64u * R), 0xFFFFFF00ull);
    *out_D_io = out_D;
  }
  LevelArgs LA{S, q, counts, level, max_depth, pid_base, k0, k1, role0, role1, qout, rad, pq, phong_counts, xstats,
               gen ? *gen : no_gen, in_D, out_k, out_D, cls_thr, dbg_t, dbg_prim};
#define CFT_LAUNCH(RR, NN, LL)                                                                                          \
  do {                                                                                                                  \
    if (gen) hipLaunchKernelGGL((k_trace_level<RR, NN, LL, true>), dim3(g), dim3(CFT_BLOCK), dyn_lds, st, LA);           \
    else hipLaunchKernelGGL((k_trace_level<RR, NN, LL, false>), dim3(g), dim3(CFT_BLOCK), dyn_lds, st, LA);              \
  } while (0)
#define CFT_LAUNCH_R(LL)                                                    \
  if (two_scans) {                                                        \
    if (R == 1) CFT_LAUNCH(1, 2, (LL) | 3); else CFT_LAUNCH(2, 2, (LL) | 3);  \
  } else {                                                                \
    switch (R) {                                                          \
      case 1: CFT_LAUNCH(1, 1, LL); break;                                \
      case 3: CFT_LAUNCH(3, 1, LL); break;                                \
      case 4: CFT_LAUNCH(4, 1, LL); break;                                \
      default: CFT_LAUNCH(2, 1, LL); break;                               \
    }                                                                     \
  }
  // instances: with / without the moving-sphere and triangle lists, with / without the grid walk
  // (primary rays of a scene whose lists are short take the cone cull, or at worst the lists: the instance without
  // the walk has the registers for 4 rays per lane without spills)
  // (with per-tile lists the primary rays of long-list scenes take them instead of the walk)
  const bool tiles = gen && gen->tiled && S.tile_off && (R == 1 || R == 2 || R == 4);
  const bool walk = S.grid_mode && (level > 0 || (S.grid_mode == 2 && !tiles));
  // (no moving spheres, but triangles - cornell, S20k: without the moving-sphere code the pack spills half as much)
  const bool no_msph = !S.n_msph && !two_scans;
  if (ldsg && two_scans) hipLaunchKernelGGL((k_trace_level<2, 2, 1 | 4 | 16, false>), dim3(g), dim3(1024), dyn_lds, st, LA);
  else if (ldsg && lean) hipLaunchKernelGGL((k_trace_level<3, 1, 4 | 16, false>), dim3(g), dim3(1024), dyn_lds, st, LA);
  else if (ldsg) hipLaunchKernelGGL((k_trace_level<3, 1, 1 | 4 | 16, false>), dim3(g), dim3(1024), dyn_lds, st, LA);
  else if (walk) { if (lean) { CFT_LAUNCH_R(4) } else if (no_msph) { CFT_LAUNCH_R(6) } else { CFT_LAUNCH_R(7) } }
  else      { if (lean) { CFT_LAUNCH_R(0) } else if (no_msph) { CFT_LAUNCH_R(2) } else { CFT_LAUNCH_R(3) } }
#undef CFT_LAUNCH_R
#undef CFT_LAUNCH
}

// Survivors whose next ray crosses at least this many cells of the grid's box form the long class of the queue.
static uint32_t walk_class_threshold(const cftrt_ctx* ctx, const SceneDev& S) {
  if (ctx->classes_thr) return ctx->classes_thr;
  const int t = (S.gnx + S.gny + S.gnz) / 16;
  return (uint32_t)std::max(3, std::min(t, 16));
}

enum { CLS_GEN = 0, CLS_ISECT, CLS_SHADE, CLS_SHADOW, CLS_RESOLVE, CLS_N };

// Pairs of HIP events around launches on the ctx stream, summed per class after a sync.  An event record is a
// packet of its own on the queue (~6 us each, 0.65 ms per S500 frame with one pair per launch), so consecutive
// launches of the SAME class share one pair: the end event is only recorded when the class changes or the host
// is about to wait on the stream (flush).  The fused pipeline thus times each group of levels between two
// peeks as one span (launch-to-launch gaps of ~10 us included); the split pipeline alternates classes and keeps
// one pair per launch.
struct EvTimer {
  cftrt_ctx* ctx;
  bool timing = true;                           // false: count launches only (no event is created or recorded)
  size_t used = 0;
  bool open = false;                            // the last span's end event has not been recorded yet
  std::vector<std::pair<size_t, int>> spans;    // (start event index, class)
  uint32_t launches[CLS_N] = {0};
  int begin(int cls) {
    launches[cls] += 1;
    if (!timing) return 0;
    if (open && spans.back().second == cls) return 0;
    flush();
    if (used + 2 > ctx->events.size()) {
      size_t old = ctx->events.size();
      ctx->events.resize(old + 256);
      for (size_t i = old; i < ctx->events.size(); ++i)
        if (hipEventCreate(&ctx->events[i]) != hipSuccess) return -1;
    }
    spans.push_back({used, cls});
    (void)hipEventRecord(ctx->events[used], ctx->stream);
    used += 2;
    open = true;
    return 0;
  }
  void flush() {
    if (open) (void)hipEventRecord(ctx->events[spans.back().first + 1], ctx->stream);
    open = false;
  }
  void collect(double* ms_by_class, uint32_t* n_by_class) {   // after a stream sync
    flush();
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& sp : spans) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ctx->events[sp.first], ctx->events[sp.first + 1]) == hipSuccess) ms_by_class[sp.second] += ms;
    }
    for (int c = 0; c < CLS_N; ++c) { n_by_class[c] += launches[c]; launches[c] = 0; }
    spans.clear();
    used = 0;
  }
};

static const size_t kQueueSlack = (size_t)CFT_NSEG * 2 * 64 * 4 + 256;   // slots a sharded queue may need beyond the ray count
// Device buffers of one pass (P paths): the ray queues, the radiance slots, the Phong queue, the level counters.
struct PassBufs {
  QueueDev q[2];
  LookRecs hA;   // records of the `looked` scan(s): one set, or one per looking light (scene->multi_look)
  HitDev hB;     // records of trace's own scan (= hA.r[0] unless the scene has motion AND lights)
  PhongQueueDev pq;
  Rad3* rad;
  uint32_t* counts;
};
static const size_t kShardWords = 2 * CFTRT_MAX_LEVELS + 8 + 128 + 2 + 4 * 64;   // (+ walk counters of the diagnostic build)   // used words of shard 0's block
static_assert(kShardWords <= CFT_SHARD_STRIDE, "shard 0's counters, statistics and stamps must fit its block");
static const size_t kCountWords = (size_t)CFT_NSEG * CFT_SHARD_STRIDE;   // shard s of level l: word s * CFT_SHARD_STRIDE + l; the rest in shard 0's block:   // (+ 64 stamp sums of 64 bits of the diagnostic build)
static const size_t kStampWord = 2 * CFTRT_MAX_LEVELS + 2 + 8;   // even: 8-byte aligned behind a 256-byte aligned base
static_assert(kStampWord % 2 == 0, "stamp sums are 64-bit words");   // [0..L] level counts, [L+1..2L+1] phong counts, then: second evaluations, rescans, k_trace_finish's queue cursor
static_assert(CFTRT_MAX_LEVELS == CFTRT_MAX_LEVELS_DEV, "kernels.h and cftrt.h disagree on the level limit");

// A pass runs the split pipeline (k_intersect -> hit records -> k_shade) when the ctx asks for it, and always for a
// moving scene with several lights that look: each of those lights then gets a `looked` scan of its own.
static bool pass_split(const cftrt_ctx* ctx, const cftrt_scene* s) { return ctx->split || s->multi_look; }
static int pass_look_sets(const cftrt_scene* s) { return s->multi_look ? s->n_look : 1; }

static uint64_t bytes_per_path(const cftrt_ctx* ctx, const cftrt_scene* s) {
  const SceneDev& S = s->dev;
  const bool two_scans = S.any_motion && S.n_lights > 0;
  return 80 + 12 + (pass_split(ctx, s) ? 40ull * (pass_look_sets(s) + (two_scans ? 1 : 0)) : 0) + (S.n_phong_mats ? 56 : 0);
}

// From the ctx's grow-only hipMalloc'ed workspace.
static int bufs_from_ctx(cftrt_ctx* ctx, const cftrt_scene* s, uint32_t P, PassBufs* B) {
  const SceneDev& S = s->dev;
  const bool phong = S.n_phong_mats > 0;
  const bool two_scans = S.any_motion && S.n_lights > 0;
  const int nA = pass_look_sets(s);
  int rc;
  const size_t Q = (size_t)P + kQueueSlack;   // a sharded queue: CFT_SHARDS segments, each rounded up to whole trips
  for (int k = 0; k < 2; ++k) {
    if ((rc = ensure(ctx, ctx->qa[k], Q * 16))) return rc;
    if ((rc = ensure(ctx, ctx->qb[k], Q * 16))) return rc;
    if ((rc = ensure(ctx, ctx->qc[k], Q * 8))) return rc;
  }
  for (int k = 0; pass_split(ctx, s) && k < nA + (two_scans ? 1 : 0); ++k) {   // hit records: split pipeline only
    if ((rc = ensure(ctx, ctx->h0[k], (size_t)P * 16))) return rc;
    if ((rc = ensure(ctx, ctx->h1[k], (size_t)P * 16))) return rc;
    if ((rc = ensure(ctx, ctx->ht[k], (size_t)P * 8))) return rc;
  }
  if ((rc = ensure(ctx, ctx->rad, (size_t)P * 12))) return rc;
  if (phong) {
    if ((rc = ensure(ctx, ctx->pqa, (size_t)P * 16))) return rc;
    if ((rc = ensure(ctx, ctx->pqb, (size_t)P * 16))) return rc;
    if ((rc = ensure(ctx, ctx->pqc, (size_t)P * 16))) return rc;
    if ((rc = ensure(ctx, ctx->pqd, (size_t)P * 8))) return rc;
  }
  if ((rc = ensure(ctx, ctx->counts, kCountWords * 4))) return rc;
  for (int k = 0; k < 2; ++k) B->q[k] = QueueDev{(float4*)ctx->qa[k].p, (float4*)ctx->qb[k].p, (float2*)ctx->qc[k].p};
  memset(&B->hA, 0, sizeof B->hA);
  B->hA.n = nA;
  for (int k = 0; k < nA; ++k) B->hA.r[k] = HitDev{(float4*)ctx->h0[k].p, (float4*)ctx->h1[k].p, (double*)ctx->ht[k].p};
  B->hB = two_scans ? HitDev{(float4*)ctx->h0[nA].p, (float4*)ctx->h1[nA].p, (double*)ctx->ht[nA].p} : B->hA.r[0];
  B->pq = PhongQueueDev{(float4*)ctx->pqa.p, (float4*)ctx->pqb.p, (float4*)ctx->pqc.p, (uint2*)ctx->pqd.p};
  B->rad = (Rad3*)ctx->rad.p;
  B->counts = (uint32_t*)ctx->counts.p;
  return CFTRT_OK;
}

// Carved out of the static array (`base` = its device address on the ctx's device, from hipGetSymbolAddress).
// No allocation; fails if the pass does not fit.
static const size_t kStaticHeader = kCountWords * 4 + 2 * kQueueSlack * 40 + 4096;   // the counters, the queues' slack, and room for the 256-byte alignment of each carve
static int bufs_from_static(cftrt_ctx* ctx, const cftrt_scene* s, uint32_t P, PassBufs* B) {
  const SceneDev& S = s->dev;
  const bool phong = S.n_phong_mats > 0;
  const bool two_scans = S.any_motion && S.n_lights > 0;
  unsigned char* base = ctx->static_scratch;
  size_t off = 0;
  bool ok = true;
  auto carve = [&](size_t bytes) -> void* {
    off = (off + 255) & ~(size_t)255;
    void* p = base + off;
    off += bytes;
    if (off > (size_t)CFTRT_STATIC_SCRATCH_BYTES) ok = false;
    return p;
  };
  B->counts = (uint32_t*)carve(kCountWords * 4);
  const size_t Q = (size_t)P + kQueueSlack;
  for (int k = 0; k < 2; ++k) {
    B->q[k].a = (float4*)carve(Q * 16);
    B->q[k].b = (float4*)carve(Q * 16);
    B->q[k].c = (float2*)carve(Q * 8);
  }
  B->rad = (Rad3*)carve((size_t)P * 12);
  memset(&B->hA, 0, sizeof B->hA);
  B->hA.n = 1;
  B->hB = HitDev{nullptr, nullptr, nullptr};
  if (pass_split(ctx, s)) {
    B->hA.n = pass_look_sets(s);
    for (int k = 0; k < B->hA.n; ++k)
      B->hA.r[k] = HitDev{(float4*)carve((size_t)P * 16), (float4*)carve((size_t)P * 16), (double*)carve((size_t)P * 8)};
    B->hB = two_scans ? HitDev{(float4*)carve((size_t)P * 16), (float4*)carve((size_t)P * 16), (double*)carve((size_t)P * 8)} : B->hA.r[0];
  }
  B->pq = PhongQueueDev{nullptr, nullptr, nullptr, nullptr};
  if (phong)
    B->pq = PhongQueueDev{(float4*)carve((size_t)P * 16), (float4*)carve((size_t)P * 16), (float4*)carve((size_t)P * 16),
                          (uint2*)carve((size_t)P * 8)};
  if (!ok) return fail(ctx, CFTRT_ERR_LIMIT, "pass beyond the static workspace (CFTRT_STATIC_SCRATCH_BYTES)");
  return CFTRT_OK;
}

// Per-tile primitive lists for the primary rays of this render call (kernels.h, k_tile_bin): count, prefix, fill.
// One host round trip for the total (the index buffer is sized by it); ~40 us per call.  Leaves ctx->tiles_active false
// (the frame then renders as before) when the frame is not tiled, the lens is wide against the focus distance, or the
// scene lists fewer primitives than one round of the cone test handles anyway.
static int build_tile_lists(cftrt_scene* s, const CamDev& C, const cftrt_camera* cam) {
  cftrt_ctx* ctx = s->ctx;
  ctx->tiles_active = false;
  const SceneDev& S = s->dev;
  const uint32_t n = (uint32_t)(S.n_sphF + S.n_msph + S.n_triF);
  if (!ctx->tile_lists || pass_split(ctx, s) || n < 64 || C.width % 32u || C.height % 8u) return CFTRT_OK;
  const uint32_t ntx = C.width / 32u, nty = C.height / 8u, nt = ntx * nty;
  if (ntx > 65535u || nty > 65535u) return CFTRT_OK;
  {   // lens radius against the distance of the viewport plane (the blur a |1 - depth / fd| that k_tile_bin widens a primitive by stays small against it)
    const double lu = std::sqrt(cam->up[0] * cam->up[0] + cam->up[1] * cam->up[1] + cam->up[2] * cam->up[2]);
    const double ls = std::sqrt(cam->sight[0] * cam->sight[0] + cam->sight[1] * cam->sight[1] + cam->sight[2] * cam->sight[2]);
    const double fd = cam->focus_dist * ls;   // (up to the tilt of a non-orthogonal `up`: a factor the 1/8 absorbs)
    if (!(fd > 0.) || !(0.5 * std::fabs(cam->aperture) * (1. + lu) * 1.0001 < 0.125 * fd)) return CFTRT_OK;
  }
  static_assert(sizeof(CamDev) <= sizeof(ctx->tiles_cam), "tiles_cam holds a CamDev");
  if (ctx->tiles_scene == s->serial && !memcmp(ctx->tiles_cam, &C, sizeof C)) {   // same scene, camera and size: keep them
    ctx->tiles_active = ctx->tiles_usable != 0;
    return CFTRT_OK;
  }
  ctx->tiles_scene = 0;
  int rc;
  if ((rc = ensure(ctx, ctx->tile_rng, (size_t)n * sizeof(TileRange)))) return rc;
  if ((rc = ensure(ctx, ctx->tile_cnt, (size_t)nt * 4))) return rc;
  if ((rc = ensure(ctx, ctx->tile_off, (size_t)(nt + 1) * 4))) return rc;
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, hipMemsetAsync(ctx->tile_cnt.p, 0, (size_t)nt * 4, st));
  const uint32_t g = (uint32_t)std::min<uint64_t>(((uint64_t)n * 64 + CFT_BLOCK - 1) / CFT_BLOCK, (uint64_t)ctx->n_cu * 8);   // a wave per primitive
  hipLaunchKernelGGL(k_tile_bin, dim3(g), dim3(CFT_BLOCK), 0, st, C, S.boundF, (const float4*)S.triF, (uint32_t)(S.n_sphF + S.n_msph), n, S.bcx, S.bcy, S.bcz, ntx, nty,
                     (TileRange*)ctx->tile_rng.p, (uint32_t*)ctx->tile_cnt.p, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0);
  hipLaunchKernelGGL(k_tile_prefix, dim3(1), dim3(CFT_BLOCK), 0, st, (uint32_t*)ctx->tile_cnt.p, nt, (uint32_t*)ctx->tile_off.p);
  uint32_t total = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&total, (uint32_t*)ctx->tile_off.p + nt, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  memset(ctx->tiles_cam, 0, sizeof ctx->tiles_cam);
  memcpy(ctx->tiles_cam, &C, sizeof C);
  ctx->tiles_usable = 0;
  ctx->tiles_scene = s->serial;
  if ((uint64_t)total * 4 > (1ull << 30)) return CFTRT_OK;   // (a scene wrapped around the camera: not worth a GiB of lists)
  if ((rc = ensure(ctx, ctx->tile_idx, std::max<size_t>((size_t)total * 4, 256)))) return rc;
  hipLaunchKernelGGL(k_tile_bin, dim3(g), dim3(CFT_BLOCK), 0, st, C, S.boundF, (const float4*)S.triF, (uint32_t)(S.n_sphF + S.n_msph), n, S.bcx, S.bcy, S.bcz, ntx, nty,
                     (TileRange*)ctx->tile_rng.p, (uint32_t*)ctx->tile_cnt.p, (const uint32_t*)ctx->tile_off.p,
                     (uint32_t*)ctx->tile_idx.p, 1);
  HIP_TRY(ctx, hipGetLastError());
  if (getenv("CFTRT_DEBUG"))
    fprintf(stderr, "cftrt tile lists: %u primitives, %u x %u tiles, %u entries (%.1f per tile)\n", n, ntx, nty, total, (double)total / nt);
  ctx->tiles_usable = 1;
  ctx->tiles_active = true;
  return CFTRT_OK;
}

// Renders path ids [pid_base, pid_base + P) (P = npix * nsamples) into the radiance buffer and, when
// accum_dev != null, resolves them into accum_dev.
static int run_pass(cftrt_scene* s, const CamDev& C, uint64_t pid_base, uint32_t P, uint32_t nsamples, uint32_t npix,
                    uint32_t max_depth, uint64_t seed, int jitter, float* accum_dev, cftrt_stats* st, EvTimer& tm,
                    bool use_static = false) {
  cftrt_ctx* ctx = s->ctx;
  SceneDev S = s->dev;
  S.wave_cull = ctx->wave_cull;
  S.grid_mode = S.gnx ? ctx->grid : 0;
  static const int promote_above = getenv("CFTRT_GRID_PROMOTE") ? atoi(getenv("CFTRT_GRID_PROMOTE")) : 4096;
  if (S.grid_mode == 1 && S.n_sphF + S.n_msph + S.n_triF > promote_above) S.grid_mode = 2;   // long lists: every wave walks
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const bool phong = S.n_phong_mats > 0;
  const bool two_scans = S.any_motion && S.n_lights > 0;   // `looked` and trace each draw their own motion times
  int rc;
  PassBufs B;
  if ((rc = use_static ? bufs_from_static(ctx, s, P, &B) : bufs_from_ctx(ctx, s, P, &B))) return rc;
  const bool split = pass_split(ctx, s);
  const size_t ncount = kCountWords;
  uint32_t* counts = B.counts;
  uint32_t* phong_counts = counts + CFTRT_MAX_LEVELS + 1;
  uint32_t* xstats = counts + 2 * CFTRT_MAX_LEVELS + 2;   // [0] second exact evaluations, [1] f64 rescans
  hipStream_t st_ = ctx->stream;
  HIP_TRY(ctx, hipMemsetAsync(counts, 0, ncount * 4, st_));
  HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)counts, (int)P, 1, st_));   // counts[0] = P (no host staging buffer)

  QueueDev q[2] = {B.q[0], B.q[1]};
  const LookRecs hA = B.hA;
  const HitDev hB = B.hB;
  PhongQueueDev pq = B.pq;
  Rad3* rad = B.rad;

  const int tiled = (C.width % 32u == 0 && C.height % 8u == 0 && npix == C.width * C.height && P % npix == 0 && pid_base % npix == 0) ? 1 : 0;
  const GenArgs gen{C, jitter, tiled};
  if (tiled && ctx->tiles_active && !split && !use_static) {
    S.tile_off = (const uint32_t*)ctx->tile_off.p;
    S.tile_idx = (const uint32_t*)ctx->tile_idx.p;
  }
  if (split) {   // the fused pipeline generates inside its level-0 launch
    if (tm.begin(CLS_GEN)) return fail(ctx, CFTRT_ERR_HIP, "hipEventCreate failed");
    hipLaunchKernelGGL(k_generate, dim3(grid_for(ctx, P, 16)), dim3(CFT_BLOCK), 0, st_, C, pid_base, P, k0, k1, jitter, tiled, q[0]);
  }

  uint32_t ub = P;   // upper bound on the live rays of the current level (a ray spawns at most one ray)
  uint32_t levels_run = 0;
  uint32_t in_D = 0;   // segment stride of the current level's input queue (0: dense)
  // Motion key of a light's `looked` scan: the reference scans once per light that looks (SkyLight, LightShape),
  // keyed here by that light's index in world.lights (Point/Parallel lights never scan, light.rs:17-19).  A moving
  // scene with several such lights (s->multi_look) runs one scan per light, through the split pipeline.
  uint32_t looked_role = 0;
  std::vector<uint32_t> look_roles;
  for (size_t i = 0; i < s->lights.size(); ++i)
    if (s->lights[i].kind == CFTRT_LIGHT_SKY || s->lights[i].kind == CFTRT_LIGHT_SHAPE) look_roles.push_back((uint32_t)i);
  if (!look_roles.empty()) looked_role = look_roles[0];
  for (uint32_t level = 0; level < max_depth; ++level) {
    const uint32_t grid = grid_for(ctx, ub, 8);
    int in = level & 1, outq = in ^ 1;
    const uint32_t nl = (uint32_t)S.n_lights;
    if (!split) {
      if (tm.begin(CLS_ISECT)) return fail(ctx, CFTRT_ERR_HIP, "hipEventCreate failed");   // booked as the intersect class (ms_shade stays 0)
      // big levels append through CFT_SHARDS counters (one counter takes ~88 appends per microsecond)
      uint32_t out_D = (ctx->shards && ub >= (1u << 22) && level + 1 < max_depth) ? 1u : 0u;
      // survivors in two classes by the length of their next walk, when the next level walks the grid (kernels.h, CFT_NSEG)
      // (S20k -4 %; S500 -3.7 % with the instance that reads the grid from LDS and does not sort its rays - with the
      // sorting instance the next level gained what the class key costs this one)
      const uint32_t cls_thr = (S.grid_mode && ctx->classes && !ctx->variant && ub >= ctx->classes_above && level + 1 < max_depth) ? walk_class_threshold(ctx, S) : 0u;
      launch_trace_level(ctx, S, q[in], counts, level, max_depth, pid_base, k0, k1, two_scans, two_scans ? looked_role : nl, nl,
                         q[outq], rad, pq, phong_counts, xstats, ub, level == 0 ? &gen : nullptr, in_D, &out_D, nullptr, nullptr, cls_thr);
      in_D = out_D;
    } else {
      tm.begin(CLS_ISECT);
      if (s->multi_look) {   // one scan per looking light, then trace's own: each with its own motion times
        for (int k = 0; k < hA.n; ++k)
          launch_intersect(ctx, S, q[in], counts, level, pid_base, k0, k1, false, look_roles[(size_t)k], look_roles[(size_t)k], hA.r[k], hA.r[k], xstats, ub);
        launch_intersect(ctx, S, q[in], counts, level, pid_base, k0, k1, false, nl, nl, hB, hB, xstats, ub);
      } else {
        launch_intersect(ctx, S, q[in], counts, level, pid_base, k0, k1, two_scans, two_scans ? looked_role : nl, nl, hA.r[0], hB, xstats, ub);
      }
      tm.begin(CLS_SHADE);
      hipLaunchKernelGGL(k_shade, dim3(grid), dim3(CFT_BLOCK), 0, st_, S, q[in], hA, hB, counts, level, max_depth,
                         pid_base, k0, k1, q[outq], rad, pq, phong_counts);
    }
    if (phong) {
      tm.begin(CLS_SHADOW);
      const ShadowArgs SA{S, pq, phong_counts, level, pid_base, k0, k1, rad, xstats};
      hipLaunchKernelGGL(k_shadow, dim3(grid), dim3(CFT_BLOCK), 0, st_, SA);
    }
    levels_run = level + 1;
    // every 4 levels (every 16 once the launches are latency-bound: a peek idles the GPU for ~35 us) look at the
    // next level's count; stop when the queue has drained
    if ((level & (ub < (1u << 19) ? 15u : 3u)) == 3u && level + 1 < max_depth) {
      tm.flush();
      // (the next level's count: the sum of its counter blocks, by one wave, into pinned memory)
      hipLaunchKernelGGL(k_peek, dim3(1), dim3(64), 0, st_, (const uint32_t*)(counts + level + 1), ctx->host_counts);
      HIP_TRY(ctx, hipStreamSynchronize(st_));
      if (ctx->host_counts[0] == 0) break;
      ub = ctx->host_counts[0];
      if (!split && !phong && ctx->finish && !ctx->variant && ub < ctx->finish_below) {
        // few rays left: every remaining level in ONE launch, path by path (k_trace_finish); one lane per live ray,
        // at most what the chip holds at once (lanes refill from the queue)
        if (tm.begin(CLS_ISECT)) return fail(ctx, CFTRT_ERR_HIP, "hipEventCreate failed");
        const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)ub + CFT_BLOCK - 1) / CFT_BLOCK, (uint64_t)ctx->n_cu * 8));
        const FinishArgs FA{S, q[outq], counts, level + 1, max_depth, pid_base, k0, k1, two_scans ? looked_role : nl, nl,
                            two_scans ? 1 : 0, rad, xstats + 2, xstats, in_D};
        // the instance by scene kind, as for the level kernels: with a grid every ray of these (bounce) levels walks it
        const bool f_lean = !S.n_msph && !S.n_triF && !S.n_mtri, f_no_msph = !S.n_msph && !two_scans;
        if (!S.grid_mode) hipLaunchKernelGGL(k_trace_finish<7>, dim3(blocks), dim3(CFT_BLOCK), 0, st_, FA);
        else if (f_lean) hipLaunchKernelGGL((k_trace_finish<4 | 8>), dim3(blocks), dim3(CFT_BLOCK), 0, st_, FA);
        else if (f_no_msph) hipLaunchKernelGGL((k_trace_finish<6 | 8>), dim3(blocks), dim3(CFT_BLOCK), 0, st_, FA);
        else hipLaunchKernelGGL((k_trace_finish<7 | 8>), dim3(blocks), dim3(CFT_BLOCK), 0, st_, FA);
        levels_run = max_depth;
        break;
      }
    }
  }
  if (accum_dev) {
    tm.begin(CLS_RESOLVE);
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(ctx, npix, 16)), dim3(CFT_BLOCK), 0, st_, rad, npix, nsamples, accum_dev);
  }
  tm.flush();
  HIP_TRY(ctx, hipMemcpyAsync(ctx->host_counts, counts, ncount * 4, hipMemcpyDeviceToHost, st_));
  HIP_TRY(ctx, hipStreamSynchronize(st_));
  HIP_TRY(ctx, hipGetLastError());
  if (st) {
    const uint32_t* hc = ctx->host_counts;
    const uint32_t* hp = hc + CFTRT_MAX_LEVELS + 1;
    for (uint32_t l = 0; l < levels_run && l < CFTRT_MAX_LEVELS; ++l) {
      for (int sh = 1; sh < CFT_NSEG; ++sh) ctx->host_counts[l] += ctx->host_counts[sh * CFT_SHARD_STRIDE + l];
      st->rays_per_level[l] += hc[l];
      st->rays += hc[l];
      uint64_t pr = (uint64_t)hp[l] * (uint64_t)S.n_lights;
      st->phong_rays += pr;
      st->rays += pr;
      if (hc[l] && l + 1 > st->levels) st->levels = l + 1;
    }
#ifdef CFT_STAMPS
    if (getenv("CFTRT_STAMPS")) {
      static const char* names[8] = {"gen", "sort", "setup", "scan", "exact+hit", "shade", "append", "other"};
      for (uint32_t l = 0; l < 8 && l < levels_run; ++l) {
        const unsigned long long* sp = (const unsigned long long*)(hc + kStampWord) + 8 * l;
        unsigned long long tot = 0;
        for (int p = 0; p < 8; ++p) tot += sp[p];
        if (!tot) continue;
        const double trips = (double)hc[l] / 256.0;   // (R = 4)
        fprintf(stderr, "stamps level %u (%u rays): total %.0f cycles per 256 rays |", l, hc[l], (double)tot / trips);
        for (int p = 0; p < 8; ++p) fprintf(stderr, " %s %.1f%%", names[p], 100.0 * (double)sp[p] / (double)tot);
        fprintf(stderr, "\n");
        const unsigned long long* wk = (const unsigned long long*)(hc + kStampWord + 128 + 2) + 32 * l;
        if (l < 4 && wk[0])
          fprintf(stderr, "  walk: %llu rays, cells/ray %.1f, entries/ray %.1f, candidate-path entries/ray %.2f | cells hist <4 <8 <16 <32 <64 64+: %.3f %.3f %.3f %.3f %.3f %.3f | entries hist <4..128+: %.3f %.3f %.3f %.3f %.3f %.3f %.3f\n",
                  wk[0], (double)wk[1] / wk[0], (double)wk[2] / wk[0], (double)wk[3] / wk[0], (double)wk[8] / wk[0], (double)wk[9] / wk[0],
                  (double)wk[10] / wk[0], (double)wk[11] / wk[0], (double)wk[12] / wk[0], (double)wk[13] / wk[0], (double)wk[16] / wk[0],
                  (double)wk[17] / wk[0], (double)wk[18] / wk[0], (double)wk[19] / wk[0], (double)wk[20] / wk[0], (double)wk[21] / wk[0], (double)wk[22] / wk[0]);
        if (l < 4 && wk[3])   // of the entries that reach the candidate path: robust discriminant passes, a bound results, a hit is certified
          fprintf(stderr, "  candidate path: %.3f of its entries pass the robust discriminant, %.3f leave a bound, %.3f certify a hit\n",
                  (double)wk[28] / wk[3], (double)wk[29] / wk[3], (double)wk[30] / wk[3]);
        if (l < 4 && wk[24])   // lanes in each part of the walk's loop, per execution by a wave
          fprintf(stderr, "  walk loop per ray slot of a wave (%.0f): step %.1f executions with %.1f lanes | batch %.1f with %.1f lanes | candidate path %.1f with %.1f lanes\n",
                  (double)hc[l] / 64.0, (double)wk[24] / ((double)hc[l] / 64.0), (double)wk[1] / (double)wk[24], (double)wk[25] / ((double)hc[l] / 64.0),
                  (double)wk[27] / (double)(wk[25] ? wk[25] : 1), (double)wk[26] / ((double)hc[l] / 64.0), (double)wk[3] / (double)(wk[26] ? wk[26] : 1));
      }
    }
#endif
    if (getenv("CFTRT_DEBUG")) {   // rays of the long class per level / all rays of the level
      fprintf(stderr, "cftrt long-class rays per level:");
      for (uint32_t l = 0; l < levels_run && l < 12; ++l) {
        uint64_t lg = 0;
        for (int sh = CFT_SHARDS; sh < CFT_NSEG; ++sh) lg += ctx->host_counts[sh * CFT_SHARD_STRIDE + l];
        fprintf(stderr, " %llu/%u", (unsigned long long)lg, hc[l]);
      }
      fprintf(stderr, "\n");
    }
    st->exact_second += hc[2 * CFTRT_MAX_LEVELS + 2];
    st->exact_rescan += hc[2 * CFTRT_MAX_LEVELS + 3];
    st->paths += P;
    st->passes += 1;
  }
  return CFTRT_OK;
}

static int render_impl(cftrt_scene* s, const cftrt_camera* cam, uint32_t width, uint32_t height, uint32_t spp_begin,
                       uint32_t spp_end, uint32_t max_depth, uint64_t seed, float* accum_dev, cftrt_stats* stats,
                       bool use_static = false) {
  cftrt_ctx* ctx = s->ctx;
  if (!s->committed) return fail(ctx, CFTRT_ERR_STATE, "scene not committed");
  if (spp_end < spp_begin) return fail(ctx, CFTRT_ERR_INVALID, "spp_end < spp_begin");
  if (max_depth > CFTRT_MAX_LEVELS) return fail(ctx, CFTRT_ERR_LIMIT, "max_depth beyond CFTRT_MAX_LEVELS (256)");
  CamDev C;
  if (int rc = make_cam(ctx, cam, width, height, &C)) return rc;
  DEVICE_GUARD(ctx);
  const uint64_t npix = (uint64_t)width * height;
  if (npix > (1ull << 31)) return fail(ctx, CFTRT_ERR_LIMIT, "image beyond 2^31 pixels");
  // a pass renders whole samples of the image; the static workspace keeps 4 KiB + 256 B per buffer for its carving
  const uint64_t limit = use_static ? (uint64_t)CFTRT_STATIC_SCRATCH_BYTES - kStaticHeader - 16 * 256 : ctx->ws_limit;
  uint64_t spp_pass = limit / (bytes_per_path(ctx, s) * npix);
  spp_pass = std::min<uint64_t>(spp_pass, 0xFFFFFF00ull / npix);
  if (spp_pass == 0)
    return fail(ctx, CFTRT_ERR_LIMIT, use_static ? "one sample per pixel does not fit the static workspace (CFTRT_STATIC_SCRATCH_BYTES)"
                                                 : "one sample per pixel does not fit the workspace limit");
  cftrt_stats st;
  memset(&st, 0, sizeof st);
  EvTimer tm{ctx};
  tm.timing = !use_static;   // per-class kernel times need events created on demand: not in the call that allocates nothing
  double ms[CLS_N] = {0};
  uint32_t nl[CLS_N] = {0};
  HIP_TRY(ctx, hipEventRecord(ctx->ev_t0, ctx->stream));
  if (!use_static && spp_end > spp_begin)   // (allocates: not in the call that must not)
    if (int rc = build_tile_lists(s, C, cam)) return rc;
  for (uint32_t sb = spp_begin; sb < spp_end;) {
    uint32_t ns = (uint32_t)std::min<uint64_t>(spp_pass, spp_end - sb);
    uint64_t pid_base = (uint64_t)sb * npix;
    int rc = run_pass(s, C, pid_base, (uint32_t)(npix * ns), ns, (uint32_t)npix, max_depth, seed, 1, accum_dev, &st, tm,
                      use_static);
    if (rc) { ctx->tiles_active = false; return rc; }
    sb += ns;
    tm.collect(ms, nl);   // run_pass has synchronised: fold this pass's kernel spans
  }
  ctx->tiles_active = false;
  HIP_TRY(ctx, hipEventRecord(ctx->ev_t1, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  float msT = 0.f;
  (void)hipEventElapsedTime(&msT, ctx->ev_t0, ctx->ev_t1);
  st.ms_total = msT;
  st.ms_generate = ms[CLS_GEN];
  st.ms_intersect = ms[CLS_ISECT];
  st.ms_shade = ms[CLS_SHADE];
  st.ms_shadow = ms[CLS_SHADOW];
  st.ms_resolve = ms[CLS_RESOLVE];
  st.launches_intersect = nl[CLS_ISECT];
  const double per_ray_scan = 16.0 * s->dev.n_sph + 44.0 * s->dev.n_tri;
  st.flops_intersect = (double)st.rays * per_ray_scan;
  st.flops = (double)st.rays * (per_ray_scan + 200.0);
  if (stats) *stats = st;
  return CFTRT_OK;
}

int cftrt_render_device(cftrt_scene* s, const cftrt_camera* cam, uint32_t width, uint32_t height, uint32_t spp_begin,
                        uint32_t spp_end, uint32_t max_depth, uint64_t seed, void* accum_device, cftrt_stats* stats) {
  if (!s) return CFTRT_ERR_INVALID;
  if (!accum_device) return fail(s->ctx, CFTRT_ERR_INVALID, "null accumulation buffer");
  return render_impl(s, cam, width, height, spp_begin, spp_end, max_depth, seed, (float*)accum_device, stats);
}

int cftrt_render_device_static(cftrt_scene* s, const cftrt_camera* cam, uint32_t width, uint32_t height,
                               uint32_t spp_begin, uint32_t spp_end, uint32_t max_depth, uint64_t seed, void* accum_device,
                               cftrt_stats* stats) {
  if (!s) return CFTRT_ERR_INVALID;
  if (!accum_device) return fail(s->ctx, CFTRT_ERR_INVALID, "null accumulation buffer");
  return render_impl(s, cam, width, height, spp_begin, spp_end, max_depth, seed, (float*)accum_device, stats, true);
}

int cftrt_render(cftrt_scene* s, const cftrt_camera* cam, uint32_t width, uint32_t height, uint32_t spp_begin,
                 uint32_t spp_end, uint32_t max_depth, uint64_t seed, float* accum_host, cftrt_stats* stats) {
  if (!s) return CFTRT_ERR_INVALID;
  cftrt_ctx* ctx = s->ctx;
  if (!accum_host) return fail(ctx, CFTRT_ERR_INVALID, "null accumulation buffer");
  if (width == 0 || height == 0) return fail(ctx, CFTRT_ERR_INVALID, "zero image size");
  DEVICE_GUARD(ctx);
  size_t bytes = (size_t)width * height * 3 * sizeof(float);
  if (int rc = ensure(ctx, ctx->accum_tmp, bytes)) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->accum_tmp.p, accum_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  int rc = render_impl(s, cam, width, height, spp_begin, spp_end, max_depth, seed, (float*)ctx->accum_tmp.p, stats);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(accum_host, ctx->accum_tmp.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CFTRT_OK;
}

// ------------------------------------------------------------------ multi-GPU: the one exchange step
int cftrt_comm_unique_id(unsigned char* id128) {
  if (!id128) return CFTRT_ERR_INVALID;
  const RcclApi* R = rccl_api(&g_create_error);
  if (!R) return CFTRT_ERR_HIP;
  RcclApi::UniqueId id;
  const int rc = R->GetUniqueId(&id);
  if (rc != 0) { g_create_error = std::string("ncclGetUniqueId: ") + (R->GetErrorString ? R->GetErrorString(rc) : "error"); return CFTRT_ERR_HIP; }
  memcpy(id128, id.internal, 128);
  return CFTRT_OK;
}

int cftrt_comm_init(cftrt_ctx* ctx, int nranks, int rank, const unsigned char* id128) {
  if (!ctx) return CFTRT_ERR_INVALID;
  if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, CFTRT_ERR_INVALID, "comm_init: bad rank, size or id");
  if (ctx->comm) return fail(ctx, CFTRT_ERR_STATE, "comm_init: this ctx already has a communicator");
  const RcclApi* R = rccl_api(&ctx->err);
  if (!R) return CFTRT_ERR_HIP;
  DEVICE_GUARD(ctx);
  RcclApi::UniqueId id;
  memcpy(id.internal, id128, 128);
  const int rc = R->CommInitRank(&ctx->comm, nranks, id, rank);
  if (rc != 0) { ctx->comm = nullptr; ctx->err = std::string("ncclCommInitRank: ") + (R->GetErrorString ? R->GetErrorString(rc) : "error"); return CFTRT_ERR_HIP; }
  ctx->comm_ranks = nranks; ctx->comm_rank = rank;
  return CFTRT_OK;
}

int cftrt_reduce(cftrt_ctx* ctx, void* accum_device, uint64_t n_floats, int root) {
  if (!ctx) return CFTRT_ERR_INVALID;
  if (!ctx->comm) return fail(ctx, CFTRT_ERR_STATE, "reduce: no communicator (cftrt_comm_init)");
  if (!accum_device && n_floats) return fail(ctx, CFTRT_ERR_INVALID, "reduce: null buffer");
  if (root < 0 || root >= ctx->comm_ranks) return fail(ctx, CFTRT_ERR_INVALID, "reduce: root out of range");
  const RcclApi* R = rccl_api(&ctx->err);
  if (!R) return CFTRT_ERR_HIP;
  DEVICE_GUARD(ctx);
  // in place: every rank's buffer is summed into root's (ncclFloat32 = 7, ncclSum = 0), on the ctx's stream, so it is
  // ordered behind the render calls that filled the buffer
  const int rc = R->Reduce(accum_device, accum_device, (size_t)n_floats, 7, 0, root, ctx->comm, ctx->stream);
  if (rc != 0) { ctx->err = std::string("ncclReduce: ") + (R->GetErrorString ? R->GetErrorString(rc) : "error"); return CFTRT_ERR_HIP; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CFTRT_OK;
}

int cftrt_comm_destroy(cftrt_ctx* ctx) {
  if (!ctx) return CFTRT_ERR_INVALID;
  if (!ctx->comm) return CFTRT_OK;
  const RcclApi* R = rccl_api(&ctx->err);
  if (R) { DeviceGuard guard(ctx->device); (void)R->CommDestroy(ctx->comm); }
  ctx->comm = nullptr;
  ctx->comm_ranks = 0;
  return CFTRT_OK;
}

// ------------------------------------------------------------------ parity hooks
int cftrt_generate(cftrt_ctx* ctx, const cftrt_camera* cam, uint32_t width, uint32_t height, uint64_t pid0, uint64_t n,
                   uint64_t seed, int jitter, float* out_rays) {
  if (!ctx) return CFTRT_ERR_INVALID;
  if (!out_rays && n) return fail(ctx, CFTRT_ERR_INVALID, "null output");
  if (n > (1ull << 28)) return fail(ctx, CFTRT_ERR_LIMIT, "generate hook: more than 2^28 rays");
  CamDev C;
  if (int rc = make_cam(ctx, cam, width, height, &C)) return rc;
  if (n == 0) return CFTRT_OK;
  DEVICE_GUARD(ctx);
  int rc;
  if ((rc = ensure(ctx, ctx->qa[0], n * 16)) || (rc = ensure(ctx, ctx->qb[0], n * 16)) || (rc = ensure(ctx, ctx->qc[0], n * 8))) return rc;
  QueueDev q{(float4*)ctx->qa[0].p, (float4*)ctx->qb[0].p, (float2*)ctx->qc[0].p};
  hipLaunchKernelGGL(k_generate, dim3(grid_for(ctx, n, 16)), dim3(CFT_BLOCK), 0, ctx->stream, C, pid0, (uint32_t)n,
                     (uint32_t)seed, (uint32_t)(seed >> 32), jitter, 0, q);
  std::vector<float4> a(n), b(n);
  HIP_TRY(ctx, hipMemcpyAsync(a.data(), q.a, n * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(b.data(), q.b, n * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  for (uint64_t i = 0; i < n; ++i) {
    float* o = out_rays + 6 * i;
    o[0] = a[i].x; o[1] = a[i].y; o[2] = a[i].z; o[3] = a[i].w; o[4] = b[i].x; o[5] = b[i].y;
  }
  return CFTRT_OK;
}

int cftrt_intersect(cftrt_scene* s, const float* rays, uint64_t n, uint64_t seed, uint64_t pid0, uint32_t level,
                    uint32_t role, int32_t* out_id, double* out_t, float* out_normal, float* out_point,
                    int32_t* out_outward) {
  if (!s) return CFTRT_ERR_INVALID;
  cftrt_ctx* ctx = s->ctx;
  if (!s->committed) return fail(ctx, CFTRT_ERR_STATE, "scene not committed");
  if (n && (!rays || !out_id)) return fail(ctx, CFTRT_ERR_INVALID, "null rays or out_id");
  if (n > (1ull << 28)) return fail(ctx, CFTRT_ERR_LIMIT, "intersect hook: more than 2^28 rays");
  if (n == 0) return CFTRT_OK;
  DEVICE_GUARD(ctx);
  // layout in misc: rays (24n) | id (4n) | t (8n) | n (12n) | p (12n) | outward (4n) | counter
  size_t off_id = 24 * n, off_t = off_id + 4 * n;
  off_t = (off_t + 7) & ~(size_t)7;
  size_t off_n = off_t + 8 * n, off_p = off_n + 12 * n, off_o = off_p + 12 * n, off_c = off_o + 4 * n, total = off_c + 16;
  if (int rc = ensure(ctx, ctx->misc, total)) return rc;
  char* base = (char*)ctx->misc.p;
  HIP_TRY(ctx, hipMemcpyAsync(base, rays, 24 * n, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(base + off_c, 0, 16, ctx->stream));
  SceneDev S = s->dev;
  S.wave_cull = ctx->wave_cull;
  S.grid_mode = S.gnx ? ctx->grid : 0;
  if (S.grid_mode == 1 && S.n_sphF + S.n_msph + S.n_triF > 4096) S.grid_mode = 2;   // as the render path does
  hipLaunchKernelGGL(k_intersect_hook, dim3(grid_for(ctx, n, 8)), dim3(CFT_BLOCK), 0, ctx->stream, S,
                     (const float*)base, (uint32_t)n, pid0, level, role, (uint32_t)seed, (uint32_t)(seed >> 32),
                     (int32_t*)(base + off_id), (double*)(base + off_t), (float*)(base + off_n), (float*)(base + off_p),
                     (int32_t*)(base + off_o), (uint32_t*)(base + off_c));
  HIP_TRY(ctx, hipMemcpyAsync(out_id, base + off_id, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
  if (out_t) HIP_TRY(ctx, hipMemcpyAsync(out_t, base + off_t, 8 * n, hipMemcpyDeviceToHost, ctx->stream));
  if (out_normal) HIP_TRY(ctx, hipMemcpyAsync(out_normal, base + off_n, 12 * n, hipMemcpyDeviceToHost, ctx->stream));
  if (out_point) HIP_TRY(ctx, hipMemcpyAsync(out_point, base + off_p, 12 * n, hipMemcpyDeviceToHost, ctx->stream));
  if (out_outward) HIP_TRY(ctx, hipMemcpyAsync(out_outward, base + off_o, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  return CFTRT_OK;
}

int cftrt_level_hits(cftrt_scene* s, const cftrt_camera* cam, uint32_t width, uint32_t height, const float* rays, uint64_t n,
                     uint64_t seed, uint64_t pid0, uint32_t level, int32_t* out_id, double* out_t, int32_t* out_scans) {
  if (!s) return CFTRT_ERR_INVALID;
  cftrt_ctx* ctx = s->ctx;
  if (!s->committed) return fail(ctx, CFTRT_ERR_STATE, "scene not committed");
  if (n && (!out_id || !out_t)) return fail(ctx, CFTRT_ERR_INVALID, "level_hits: null output");
  if (n > (1ull << 26)) return fail(ctx, CFTRT_ERR_LIMIT, "level_hits hook: more than 2^26 rays");
  if (level >= CFTRT_MAX_LEVELS) return fail(ctx, CFTRT_ERR_LIMIT, "level beyond CFTRT_MAX_LEVELS (256)");
  if (pass_split(ctx, s)) return fail(ctx, CFTRT_ERR_STATE, "level_hits: the hook runs the fused level kernel (split pipeline in use)");
  const bool gen_mode = rays == nullptr;
  CamDev C;
  memset(&C, 0, sizeof C);
  const uint64_t npix = (uint64_t)width * height;
  if (gen_mode) {
    if (int rc = make_cam(ctx, cam, width, height, &C)) return rc;
    if (level != 0 || npix == 0 || npix > 0xFFFFFFFFull || pid0 % npix || n % npix)
      return fail(ctx, CFTRT_ERR_INVALID, "level_hits without rays: level 0 and whole samples of the image only");
  }
  SceneDev S = s->dev;
  const bool two_scans = S.any_motion && S.n_lights > 0;
  if (out_scans) *out_scans = two_scans ? 2 : 1;
  if (n == 0) return CFTRT_OK;
  DEVICE_GUARD(ctx);
  S.wave_cull = ctx->wave_cull;
  S.grid_mode = S.gnx ? ctx->grid : 0;
  if (S.grid_mode == 1 && S.n_sphF + S.n_msph + S.n_triF > 4096) S.grid_mode = 2;   // as run_pass does
  int rc;
  if (gen_mode && (rc = build_tile_lists(s, C, cam))) return rc;
  PassBufs B;
  if ((rc = bufs_from_ctx(ctx, s, (uint32_t)n, &B))) return rc;
  const int NS = two_scans ? 2 : 1;
  const size_t off_prim = (size_t)n * NS * 8;
  if ((rc = ensure(ctx, ctx->misc, off_prim + (size_t)n * NS * 4))) return rc;
  double* dbg_t = (double*)ctx->misc.p;
  int32_t* dbg_prim = (int32_t*)((char*)ctx->misc.p + off_prim);
  hipStream_t st_ = ctx->stream;
  uint32_t* counts = B.counts;
  HIP_TRY(ctx, hipMemsetAsync(counts, 0, kCountWords * 4, st_));
  HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)(counts + level), (int)n, 1, st_));
  HIP_TRY(ctx, hipMemsetAsync(dbg_prim, 0xFF, (size_t)n * NS * 4, st_));
  if (!gen_mode) {   // the caller's rays as a dense input queue: throughput 1, local path id = index
    std::vector<float4> a(n), b(n);
    std::vector<float2> c(n);
    for (uint64_t i = 0; i < n; ++i) {
      const float* r = rays + 6 * i;
      a[i] = make_float4(r[0], r[1], r[2], r[3]);
      b[i] = make_float4(r[4], r[5], 1.f, 1.f);
      uint32_t lp = (uint32_t)i;
      float lpf;
      memcpy(&lpf, &lp, 4);
      c[i] = make_float2(1.f, lpf);
    }
    const QueueDev& q = B.q[level & 1];
    HIP_TRY(ctx, hipMemcpyAsync(q.a, a.data(), n * 16, hipMemcpyHostToDevice, st_));
    HIP_TRY(ctx, hipMemcpyAsync(q.b, b.data(), n * 16, hipMemcpyHostToDevice, st_));
    HIP_TRY(ctx, hipMemcpyAsync(q.c, c.data(), n * 8, hipMemcpyHostToDevice, st_));
    HIP_TRY(ctx, hipStreamSynchronize(st_));   // (the staging vectors die at the end of this block)
  }
  const int tiled = gen_mode && C.width % 32u == 0 && C.height % 8u == 0 ? 1 : 0;
  const GenArgs gen{C, 1, tiled};
  if (tiled && ctx->tiles_active) {
    S.tile_off = (const uint32_t*)ctx->tile_off.p;
    S.tile_idx = (const uint32_t*)ctx->tile_idx.p;
  }
  uint32_t looked_role = 0;
  for (size_t i = 0; i < s->lights.size(); ++i)
    if (s->lights[i].kind == CFTRT_LIGHT_SKY || s->lights[i].kind == CFTRT_LIGHT_SHAPE) { looked_role = (uint32_t)i; break; }
  const uint32_t nl = (uint32_t)S.n_lights;
  uint32_t out_D = 0;
  // one launch of the level kernel exactly as a render call issues it at this level (instance by scene kind, rays per
  // lane, wave cull, grid walk / tile lists); the level is the last one (max_depth = level + 1), so nothing is appended
  launch_trace_level(ctx, S, B.q[level & 1], counts, level, level + 1, pid0, (uint32_t)seed, (uint32_t)(seed >> 32), two_scans,
                     two_scans ? looked_role : nl, nl, B.q[(level & 1) ^ 1], B.rad, B.pq, counts + CFTRT_MAX_LEVELS + 1,
                     counts + 2 * CFTRT_MAX_LEVELS + 2, (uint32_t)n, gen_mode ? &gen : nullptr, 0, &out_D, dbg_t, dbg_prim);
  ctx->tiles_active = false;
  std::vector<int32_t> prim((size_t)n * NS);
  HIP_TRY(ctx, hipMemcpyAsync(out_t, dbg_t, (size_t)n * NS * 8, hipMemcpyDeviceToHost, st_));
  HIP_TRY(ctx, hipMemcpyAsync(prim.data(), dbg_prim, (size_t)n * NS * 4, hipMemcpyDeviceToHost, st_));
  HIP_TRY(ctx, hipStreamSynchronize(st_));
  HIP_TRY(ctx, hipGetLastError());
  const int32_t ns = (int32_t)s->sph_obj.size();
  for (size_t i = 0; i < prim.size(); ++i) {
    const int32_t g = prim[i];
    out_id[i] = g < 0 ? -1 : (g < ns ? s->sph_obj[(size_t)g] : s->tri_obj[(size_t)(g - ns)]);
  }
  return CFTRT_OK;
}

int cftrt_trace_paths(cftrt_scene* s, const cftrt_camera* cam, uint32_t width, uint32_t height, uint64_t pid0, uint64_t n,
                      uint32_t max_depth, uint64_t seed, int jitter, float* out_rgb, cftrt_stats* stats) {
  if (!s) return CFTRT_ERR_INVALID;
  cftrt_ctx* ctx = s->ctx;
  if (!s->committed) return fail(ctx, CFTRT_ERR_STATE, "scene not committed");
  if (n && !out_rgb) return fail(ctx, CFTRT_ERR_INVALID, "null output");
  if (n > (1ull << 26)) return fail(ctx, CFTRT_ERR_LIMIT, "trace_paths hook: more than 2^26 paths");
  if (max_depth > CFTRT_MAX_LEVELS) return fail(ctx, CFTRT_ERR_LIMIT, "max_depth beyond CFTRT_MAX_LEVELS (256)");
  CamDev C;
  if (int rc = make_cam(ctx, cam, width, height, &C)) return rc;
  cftrt_stats st;
  memset(&st, 0, sizeof st);
  if (n == 0) { if (stats) *stats = st; return CFTRT_OK; }
  DEVICE_GUARD(ctx);
  EvTimer tm{ctx};
  if (max_depth == 0) {
    memset(out_rgb, 0, 12 * n);
    if (stats) *stats = st;
    return CFTRT_OK;
  }
  // Whole samples of the image run as the render calls run them: per-tile lists for the primary rays, the queue in
  // 32x8-pixel tiles (when the size allows), 24-byte level-0 rays, slot-numbered survivors.  Anything else is one
  // row-major pass over the n paths.
  const uint64_t npix = (uint64_t)width * height;
  const bool whole = npix <= 0xFFFFFFFFull && pid0 % npix == 0 && n % npix == 0;
  int rc = CFTRT_OK;
  if (whole) {
    if ((rc = build_tile_lists(s, C, cam))) return rc;
    rc = run_pass(s, C, pid0, (uint32_t)n, (uint32_t)(n / npix), (uint32_t)npix, max_depth, seed, jitter, nullptr, &st, tm);
    ctx->tiles_active = false;
  } else {
    rc = run_pass(s, C, pid0, (uint32_t)n, 1, (uint32_t)n, max_depth, seed, jitter, nullptr, &st, tm);
  }
  if (rc) return rc;
  std::vector<Rad3> rad(n);
  HIP_TRY(ctx, hipMemcpy(rad.data(), ctx->rad.p, n * 12, hipMemcpyDeviceToHost));
  for (uint64_t i = 0; i < n; ++i) {
    out_rgb[3 * i] = rad[i].x; out_rgb[3 * i + 1] = rad[i].y; out_rgb[3 * i + 2] = rad[i].z;
  }
  const double per_ray_scan = 16.0 * s->dev.n_sph + 44.0 * s->dev.n_tri;
  st.flops_intersect = (double)st.rays * per_ray_scan;
  st.flops = (double)st.rays * (per_ray_scan + 200.0);
  if (stats) *stats = st;
  return CFTRT_OK;
}

int cftrt_resolve_rgb8(cftrt_ctx* ctx, const float* accum_host, uint64_t npix, double divisor, uint8_t* out_rgb8) {
  if (!ctx) return CFTRT_ERR_INVALID;
  if (npix && (!accum_host || !out_rgb8)) return fail(ctx, CFTRT_ERR_INVALID, "null buffer");
  if (!(divisor > 0.0)) return fail(ctx, CFTRT_ERR_INVALID, "divisor must be positive");
  if (npix > (1ull << 31)) return fail(ctx, CFTRT_ERR_LIMIT, "image beyond 2^31 pixels");
  if (npix == 0) return CFTRT_OK;
  DEVICE_GUARD(ctx);
  size_t nf = 3 * npix;
  if (int rc = ensure(ctx, ctx->misc, nf * 4 + nf)) return rc;
  char* base = (char*)ctx->misc.p;
  HIP_TRY(ctx, hipMemcpyAsync(base, accum_host, nf * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_rgb8, dim3(grid_for(ctx, nf, 16)), dim3(CFT_BLOCK), 0, ctx->stream, (const float*)base,
                     (uint64_t)nf, (float)divisor, (uint8_t*)(base + nf * 4));
  HIP_TRY(ctx, hipMemcpyAsync(out_rgb8, base + nf * 4, nf, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  return CFTRT_OK;
}

}  // extern "C"
