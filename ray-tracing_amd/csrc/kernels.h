// kernels.h — the wavefront trace loop for gfx950: generate / trace level (intersect + shade) / shadow / resolve.
//
// What each kernel replaces in the reference (cfeitong/ray-tracing):
//   k_generate     Camera::emit_rays                      src/ray.rs:107-136
//   k_trace_level  one World::trace level for every live ray: intersect stage + shade stage + queue append
//   k_trace_tail   the same for levels with too few rays to fill the chip (list split over a block's waves)
//     intersect stage (isect_setup, scan_lists, isect_hit; alone: k_intersect)
//                  Ray::hit + Object::hit_by + Shape::hit  src/ray.rs:18-28, src/object.rs:42-55,99-138,272-302
//     shade stage (shade_one; alone: k_shade)
//                  World::trace body, HitInfo::new, Material::{scatter,render}, LightSource::looked
//                                                         src/object.rs:326-355, src/ray.rs:233-316,
//                                                         src/material/*.rs, src/light.rs:184-190,233-242
//   k_shadow       PhongModel::render + is_in_shadow      src/material/basic.rs:56-98, src/light.rs
//   k_resolve      raw[h*W+w] += p                        examples/scene.rs:97
//
// Numerics.  The linear scan culls in f32 - per wave (a cone around the wave's ray bundle), then per ray (7 packed
// FMAs per ray pair and sphere), then the cancellation-free discriminant r^2 - |oc - (oc.d)d|^2 with explicit
// error bounds - and the surviving candidates are re-evaluated in f64 with the reference's own formula,
// operation for operation, so the reported hit is the primitive and the f64 distance the reference computes
// for the same ray.  Built with -ffp-contract=off: every FMA in this file is explicit.
//
// Which primitives a ray tests: every one of the lists (the reference's linear scan), minus what the wave's cone
// rules out (scan_spheres_culled), or - an extension, cftrt_ctx_set_grid - those of the grid cells the ray crosses
// (grid_walk_ray).  All three feed the same candidate sets and the same exact finalisation, so the image does not
// depend on the choice (tests compare them bit for bit).
#pragma once
#include "device_math.h"
#include "scene_dev.h"

#define CFT_BLOCK 256
// A thread's index inside its 256-thread group.  Every kernel here runs 256-thread blocks (the mask folds away under
// their launch bounds) except the level-kernel instances that keep the grid in LDS: those run ONE 1024-thread block
// per CU - four such groups, each with its own park and sort rows, sharing the one copy of the grid.
#define CFT_TID (threadIdx.x & (CFT_BLOCK - 1u))
#define CFTRT_MAX_LEVELS_DEV 256   // = CFTRT_MAX_LEVELS of include/cftrt.h (static_assert in cftrt.hip)
#define CFT_SHADE_CHUNK 1024
#ifndef CFT_SHADE_UNROLL
#define CFT_SHADE_UNROLL 4   // ray sub-batches of a chunk in flight together (loads overlap)
#endif
#ifndef CFT_RAYS_PER_LANE
#define CFT_RAYS_PER_LANE 4
#endif

// Diagnostic build (-DCFT_STAMPS, never the shipped library): shader-clock stamps around the phases of a
// k_trace_level trip, summed per wave and added to 64-bit words behind the level counters (phase p of level l < 8 at
// stats + 8 + 2 (8 l + p)); CFTRT_STAMPS=1 makes the host print them.  Stamp values reach no output.
#ifdef CFT_STAMPS
static __device__ unsigned long long* g_walk_stats = nullptr;   // (diagnostic build: 32 counters of the grid walk, set per launch)
#define CFT_STAMP_DECL uint64_t st_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t st_last_ = __builtin_amdgcn_s_memtime()
#define CFT_STAMP(p) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); st_acc_[p] += now_ - st_last_; st_last_ = now_; } while (0)
#define CFT_STAMP_FLUSH(stats, level)                                                                   \
  do {                                                                                                  \
    if ((level) < 8u && (threadIdx.x & 63u) == 0u)                                                      \
      for (int p_ = 0; p_ < 8; ++p_)                                                                    \
        atomicAdd((unsigned long long*)((stats) + 8 + 2 * (8 * (level) + p_)), (unsigned long long)st_acc_[p_]); \
  } while (0)
#else
#define CFT_STAMP_DECL
#define CFT_STAMP(p)
#define CFT_STAMP_FLUSH(stats, level)
#endif

// The ray queue of a level may come in CFT_SHARDS compacted segments (segment s at slots [s D, s D + count[s])), each
// with its own append counter: one returning atomic per wave trip on ONE word saturates near 88 per microsecond
// chip-wide, which capped the big levels (S500 level 0: 518 k trips in 5.7 ms).  Counter s of level l is word
// counts[s * CFT_SHARD_STRIDE + l]; a dense queue is the case where only segment 0 is filled.
#define CFT_SHARDS 8
#define CFT_SHARD_STRIDE 1024   // words between the counter blocks of two shards (separate cache lines and channels)
// Two CLASSES of rays per segment (LevelArgs::cls_thr): a level kernel that walks the grid runs each slot of a wave as
// long as the slot's longest walk, so the few rays that cross many cells hold up the waves they are scattered over
// (S500: 8 % of the bounce rays cross four cells or more, and nearly every wave of 256 holds some).  The producing
// level therefore appends survivors whose next ray will cross at least cls_thr cells of the grid's box ("long",
// walk_key) from the TOP of its segment downwards and the others from the bottom up - both fit whatever the split,
// so the queue needs no more memory - and the consuming launch takes the long segments' trips first, in waves that
// hold nothing else.  Counter block s < CFT_SHARDS: short rays of segment s (and every ray of a queue without
// classes); block CFT_SHARDS + s: its long rays.  The order of the queue never affects the image.
#define CFT_NSEG (2 * CFT_SHARDS)
// Trip order t = 0 .. CFT_NSEG - 1 (long segments first): counter block and first slot of the part, D = segment stride
__device__ __forceinline__ uint32_t seg_block(uint32_t t) { return t < CFT_SHARDS ? t + CFT_SHARDS : t - CFT_SHARDS; }
__device__ __forceinline__ size_t seg_first_slot(uint32_t t, uint32_t D, uint32_t count) {
  return t < CFT_SHARDS ? (size_t)t * D + (D - count) : (size_t)(t - CFT_SHARDS) * D;
}

// constant-address-space views: a wave-uniform index then compiles to s_load (SGPR operands)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(4))) * c_f4p;
typedef float v16f __attribute__((ext_vector_type(16)));
typedef const v16f __attribute__((address_space(4))) * c_f16p;
typedef const int __attribute__((address_space(4))) * c_i32p;
typedef const double __attribute__((address_space(4))) * c_f64p;
#define AS_CONST(T, p) ((T)(uintptr_t)(p))

// ------------------------------------------------------------------ f64 exact evaluation
struct d3 {
  double x, y, z;
};
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator*(d3 a, double s) { return d3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ double ddot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double dlen2(d3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ d3 dcross(d3 a, d3 b) {
  return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ d3 dunit(d3 a) {
  double l = sqrt(dlen2(a));
  return d3{a.x / l, a.y / l, a.z / l};
}

// Sphere::hit_info, object.rs:273-286, same operation order.
__device__ __forceinline__ bool exact_sphere(d3 o, d3 d, d3 c, double r, double* t_out) {
  double a = dlen2(d);
  d3 oc = o - c;
  double b = 2. * ddot(oc, d);
  double cc = dlen2(oc) - r * r;
  const double bb = b * b;
  double delta = bb - 4. * a * cc;
  if (delta < 0.) return false;
  // Origin outside (cc > 0) and moving away (b > 0): both roots are negative and the reference returns None at its
  // `t2 < 0` test.  Decided here, before the square root and the division, whenever the rounded arithmetic cannot
  // turn -b + sqrt(delta) into 0: delta at least 1e-11 relative below b^2 puts sqrt(delta) strictly below b.  (Half
  // of all bounce rays leave the ground sphere upwards; a quarter of the primary rays look at the sky.)
  if (b > 0. && cc > 0. && delta < bb * 0.99999999999) return false;
  double sq = sqrt(delta);
  // The reference divides both roots by 2a and tests the quotients' signs; 2a > 0, so the signs are the
  // numerators' (a quotient could only differ by underflowing to -0, which needs |numerator| < 2^-1021) and only
  // the returned root is divided.
  double n1 = -b - sq, n2 = -b + sq;
  if (n2 < 0.) return false;
  *t_out = (n1 < 0. ? n2 : n1) / (2. * a);
  return true;
}
// Triangle::hit_info, object.rs:100-120.
__device__ __forceinline__ bool exact_tri(d3 o, d3 d, d3 p0, d3 p1, d3 p2, double* t_out) {
  const double EPS = 1e-3;
  d3 e1 = p1 - p0;
  d3 e2 = p2 - p0;
  d3 h = dcross(d, e2);
  double a = ddot(e1, h);
  if (-EPS < a && a < EPS) return false;
  double f = 1. / a;
  d3 s = o - p0;
  double u = f * ddot(s, h);
  if (u < 0. || u > 1.) return false;
  d3 q = dcross(s, e1);
  double v = f * ddot(d, q);
  if (v < 0. || u + v > 1.) return false;
  double t = f * ddot(e2, q);
  if (t > EPS) {
    *t_out = t;
    return true;
  }
  return false;
}

__device__ __forceinline__ d3 ld3(const double* p) { return d3{p[0], p[1], p[2]}; }

// Exact test of global prim g for this ray and scan (motion time keyed by the owning object).
__device__ __forceinline__ bool exact_prim(const SceneDev& S, int g, d3 o, d3 d, uint32_t scan_word,
                                           double* t_out, int* g_hit = nullptr) {
  if (g_hit) *g_hit = g;
  if (g < S.n_sph) {
    const double* sp = S.sphD + 4 * (size_t)g;
    d3 c = ld3(sp);
    if (S.any_motion) {
      d3 mv = ld3(S.sphMvD + 3 * (size_t)g);
      if (mv.x != 0. || mv.y != 0. || mv.z != 0.) {
        double tm = (double)motion_time(scan_word, (uint32_t)S.prim_obj[g]);
        c = c + mv * tm;
      }
    }
    return exact_sphere(o, d, c, sp[3], t_out);
  }
  // Triangles.  The two triangles of a Square::from_points (object.rs:169-174) are evaluated as the reference evaluates
  // the square, tri0.or(tri1) (object.rs:192): whichever of the two is asked for, the first triangle is tested first and
  // the second only where the first is not hit - so the second never counts where the first is hit, whatever the
  // distances, and the answer is the square's own hit.  One exact_tri call site in a loop of at most two rounds (a second
  // inlined copy cost every level kernel 30-50 bytes of scratch per lane).  *g_hit = the triangle actually hit.
  int ti = g - S.n_sph;
  int last = ti;
  if (S.any_pairs) {
    if (S.tri_first[ti] >= 0) ti -= 1;                                   // asked for the second: start at the first
    last = (ti + 1 < S.n_tri && S.tri_first[ti + 1] >= 0) ? ti + 1 : ti;   // a pair: its second triangle comes next
  }
  bool hit;
  for (;;) {
    const double* tp = S.triD + 9 * (size_t)ti;
    d3 p0 = ld3(tp), p1 = ld3(tp + 3), p2 = ld3(tp + 6);
    if (S.any_motion) {
      d3 mv = ld3(S.triMvD + 3 * (size_t)ti);
      if (mv.x != 0. || mv.y != 0. || mv.z != 0.) {
        double tm = (double)motion_time(scan_word, (uint32_t)S.prim_obj[S.n_sph + ti]);
        d3 delta = mv * tm;
        p0 = p0 + delta; p1 = p1 + delta; p2 = p2 + delta;
      }
    }
    hit = exact_tri(o, d, p0, p1, p2, t_out);
    if (hit || ti >= last) break;
    ti += 1;
  }
  if (g_hit) *g_hit = S.n_sph + ti;
  return hit;
}

// Ray::hit in f64 over every primitive (the fallback when the f32 cull cannot decide); lowest object index wins
// exact ties (min_by keeps the first minimum, ray.rs:23-27).  Done by the whole wave: `want` lanes ask for it (rarely: a third candidate within reach, or a grid
// walk that stopped early on a hit the exact arithmetic then refuses).  One lane alone needs ~90 ns per primitive -
// 1.8 ms in a 20 k primitive scene, and the launch waits for it - so the lanes that are active here take the
// requests one by one, each testing every nact-th primitive; the per-lane winners are then compared in the order
// (t, object id, prim id) - what a serial loop over ascending prim ids with a strict compare realises.  Must be reached by the requesting
// lanes together (any set of other lanes may be inactive: fewer helpers).
__device__ __forceinline__ void exact_scan_all_wave_call(const SceneDev& S, bool want, d3 o, d3 d, uint32_t scan_word,
                                                     double* T, int* G) {
  const unsigned long long act = __ballot(1);
  unsigned long long need = __ballot(want);
  const int lane = (int)(threadIdx.x & 63u);
  const int nact = __popcll(act), rank = __popcll(act & ((1ull << lane) - 1ull));
  const int n = S.n_sph + S.n_tri;
#define CFT_BCAST_D(x, src) __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), src), __builtin_amdgcn_readlane(__double2loint(x), src))
  while (need) {
    const int src = (int)__builtin_ctzll(need);
    need &= need - 1ull;
    const d3 ro = d3{CFT_BCAST_D(o.x, src), CFT_BCAST_D(o.y, src), CFT_BCAST_D(o.z, src)};
    const d3 rd = d3{CFT_BCAST_D(d.x, src), CFT_BCAST_D(d.y, src), CFT_BCAST_D(d.z, src)};
    const uint32_t rw = (uint32_t)__builtin_amdgcn_readlane((int)scan_word, src);
    double bt = __builtin_inf();
    int bg = 0x7fffffff, bobj = 0x7fffffff;
    for (int g = rank; g < n; g += nact) {
      double t;
      int gh;
      if (!exact_prim(S, g, ro, rd, rw, &t, &gh)) continue;
      const int obj = S.prim_obj[g];
      if (t < bt || (t == bt && obj < bobj)) { bt = t; bg = gh; bobj = obj; }   // g ascends: the first of equals stays
    }
    // the winners of the active lanes, one by one (uniform values; every lane ends with the same result)
    double wt = __builtin_inf();
    int wg = 0x7fffffff, wobj = 0x7fffffff;
    for (unsigned long long m = act; m; m &= m - 1ull) {
      const int l = (int)__builtin_ctzll(m);
      const double lt = CFT_BCAST_D(bt, l);
      const int lg = __builtin_amdgcn_readlane(bg, l), lobj = __builtin_amdgcn_readlane(bobj, l);
      if (lt < wt || (lt == wt && (lobj < wobj || (lobj == wobj && lg < wg)))) { wt = lt; wg = lg; wobj = lobj; }
    }
    if (lane == src) { *T = wt; *G = wt < __builtin_inf() ? wg : -1; }
  }
#undef CFT_BCAST_D
}

__device__ __forceinline__ void exact_scan_all_wave(const SceneDev& S, bool want, d3 o, d3 d, uint32_t scan_word,
                                                    double* T, int* G) {
  if (__ballot(want)) exact_scan_all_wave_call(S, want, o, d, scan_word, T, G);
}

// Geometric unit normal of prim g at point p (before HitInfo::new's flip):
// sphere (p-c)/|p-c|, negated for negative radius (object.rs:288-292); triangle unit(e1 x e2) (:123).
__device__ __forceinline__ d3 exact_normal(const SceneDev& S, int g, d3 p, uint32_t scan_word) {
  if (g < S.n_sph) {
    const double* sp = S.sphD + 4 * (size_t)g;
    d3 c = ld3(sp);
    if (S.any_motion) {
      d3 mv = ld3(S.sphMvD + 3 * (size_t)g);
      if (mv.x != 0. || mv.y != 0. || mv.z != 0.)
        c = c + mv * (double)motion_time(scan_word, (uint32_t)S.prim_obj[g]);
    }
    d3 n = dunit(p - c);
    if (sp[3] < 0.) n = d3{-n.x, -n.y, -n.z};
    return n;
  }
  const double* tp = S.triD + 9 * (size_t)(g - S.n_sph);
  d3 p0 = ld3(tp), p1 = ld3(tp + 3), p2 = ld3(tp + 6);
  return dunit(dcross(p1 - p0, p2 - p0));  // translation does not change the normal
}

// ------------------------------------------------------------------ f32 cull (hot loops)
// The scan keeps, per ray, the three smallest LOWER BOUNDS on the hit distance among all primitives
// the f32 arithmetic cannot rule out.  After the scan the best two are evaluated exactly (f64,
// reference formula); if the third bound could still win, the ray is rescanned entirely in f64.
// With conservative bounds this returns exactly the primitive and distance the f64 reference finds:
// the true winner w has t_w >= lo_w; if w is not among the two smallest bounds then lo3 <= lo_w <= t_w
// <= (any exact hit found) and the rescan fires.
struct Cand {
  float lo1, lo2, lo3;   // three smallest lower bounds, ascending (+inf when absent)
  int i1, i2;            // list indices of the first two (-1 when absent)
};
__device__ __forceinline__ void cand_init(Cand& c) {
  c.lo1 = c.lo2 = c.lo3 = __builtin_inff();
  c.i1 = c.i2 = -1;
}
__device__ __forceinline__ void cand_insert(Cand& c, float tlo, int idx) {
  // straight-line: masks + bitfield selects (nested ?: compiles to nested exec-mask branches)
  const int m1 = (tlo < c.lo1) ? -1 : 0, m2 = (tlo < c.lo2) ? -1 : 0;
  c.lo3 = __builtin_amdgcn_fmed3f(c.lo2, c.lo3, tlo);
  const int t2 = (m2 & idx) | (~m2 & c.i2);
  c.i2 = (m1 & c.i1) | (~m1 & t2);
  c.lo2 = __builtin_amdgcn_fmed3f(c.lo1, c.lo2, tlo);
  c.i1 = (m1 & idx) | (~m1 & c.i1);
  c.lo1 = fminf(c.lo1, tlo);
}

#define CFT_U4EPS 2.3841858e-7f   // 4 * 2^-24: slack for a 1-ulp sqrt/rcp and the final adds

// A sphere the fast path could not rule out (disc >= -E_ray): turn (q, disc) into a lower bound on its hit
// distance, or -1 when it certainly is no hit nearer than U.
//   q = oc.d,  disc = r^2 - |oc - q d|^2  (both f32, error model in DESIGN.md §4)
// thi (optional): an UPPER bound on the distance when the arithmetic can certify the hit (origin certainly outside
// and approaching with a certainly positive discriminant, or origin certainly inside); untouched otherwise.
__device__ __forceinline__ float sphere_lower_bound(const SceneDev& S, float q, float d, float U, float* thi = nullptr) {
  const float aq = fabsf(q);
  const float e = fmaf(aq, S.K1, S.K2);                 // |disc - true disc| <= e
  const float eq = fmaf(aq, S.K3, S.K4);                // |q - true q| <= eq
  const float io = fmaf(-q, q, d);                      // r^2 - |oc|^2 = disc - q^2
  // |io - true| <= e + |q^2 - q*^2| + rounding of the fma
  const float eio = e + fmaf(2.0f * aq + eq, eq, CFT_U4EPS * fmaf(aq, aq, fabsf(d)));
  // The commonest candidate of a bounce ray is the sphere it just left: origin certainly outside (io < -eio)
  // and moving away (q > eq), so both roots are negative.  Decided before the sqrt.
  if (q > eq && io < -eio) return -1.0f;
  const bool inside = io > eio;                         // r^2 - |oc|^2 certainly > 0
  const float arg = inside ? d - e : d + e;             // inside: disc > q^2 + e, so d - e > 0
  const float s = __builtin_amdgcn_sqrtf(fmaxf(arg, 0.0f));
  const float s_hi = fmaf(s, CFT_U4EPS, s), s_lo = fmaf(-s, CFT_U4EPS, s);
  // outside: t = t1 = -q - sqrt(disc) >= -q - s_hi ; inside: t = t2 = -q + sqrt(disc) >= -q + s_lo
  float tlo = inside ? (s_lo - q) : -(q + s_hi);
  tlo = tlo - eq - fabsf(tlo) * CFT_U4EPS;
  const bool possible = arg >= 0.0f && (inside || (s_hi - q + eq >= 0.0f));   // disc, t2 could be >= 0
  tlo = fmaxf(tlo, 0.0f);
  if (thi) {
    if (inside) {
      // t2 <= -q + eq + sqrt(d + e)
      const float sh = __builtin_amdgcn_sqrtf(d + e);
      const float h = fmaf(sh, CFT_U4EPS, sh) - q + eq;
      *thi = fmaf(fabsf(h), CFT_U4EPS, h);
    } else if (io < -eio && q < -eq && d - e > 0.0f) {
      // both roots positive; t1 <= -q + eq - sqrt(d - e)
      const float sl = __builtin_amdgcn_sqrtf(d - e);
      const float h = eq - q - fmaf(-sl, CFT_U4EPS, sl);
      *thi = fmaf(fabsf(h), CFT_U4EPS, h);
    }
  }
  return (possible && tlo <= U) ? tlo : -1.0f;
}

// Moller-Trumbore in f32 with explicit error margins (triangle = p0, e1, e2): lower bound on the hit
// distance, or -1.
__device__ __forceinline__ float tri_lower_bound(const SceneDev& S, float ox, float oy, float oz, float dx, float dy,
                                                 float dz, f3 p0, f3 e1, f3 e2, float U, float* thi = nullptr) {
  f3 d = mk3(dx, dy, dz);
  f3 h = cross3(d, e2);
  float a = dot3(e1, h);
  float r = -1.0f;
  if (fabsf(a) >= CFT_EPS - S.tKA) {                    // the reference rejects |a| < EPS (object.rs:105)
    float f = __builtin_amdgcn_rcpf(a);
    float af = fabsf(f);
    f3 s = mk3(ox - p0.x, oy - p0.y, oz - p0.z);
    float s1 = fabsf(s.x) + fabsf(s.y) + fabsf(s.z);    // >= |s|
    float u = f * dot3(s, h);
    f3 q = cross3(s, e1);
    float v = f * dot3(d, q);
    float t = f * dot3(e2, q);
    float mu = af * fmaf(s1, S.tKU1, S.tKU2);           // |u - true u|, |v - true v| <= mu
    float mt = af * fmaf(fabsf(t), S.tKT3, fmaf(s1, S.tKT1, S.tKT2));   // |t - true t| <= mt
    if (u >= -mu && v >= -mu && u <= 1.0f + mu && u + v <= 1.0f + 2.0f * mu && t + mt > CFT_EPS) {
      float tlo = fmaxf(t - mt, 0.0f);
      if (tlo <= U) r = tlo;
      // certainly a hit of the reference's test: |a| >= EPS, u, v, u + v inside by the margins, t > EPS
      if (thi && fabsf(a) >= CFT_EPS + S.tKA && u >= mu && v >= mu && u + v <= 1.0f - 2.0f * mu && t - mt > CFT_EPS)
        *thi = t + mt;
    }
  }
  return r;
}

// One ITEM of the static triangle scan: a single triangle (p0, e1, e2), or - quad - the two triangles of a Square::new
// as ONE parallelogram (p0; ea, eb): tri0 = (p0, p0 + ea, p0 + ea + eb) and tri1 = (p0 + ea + eb, p0 + eb, p0), the
// reference's corner order (object.rs:148-161).  Both triangles lie in the parallelogram's plane and their
// determinant e1.(d x e2) equals ea.(d x eb), so one Moller-Trumbore evaluation in the parallelogram's coordinates
// (point = p0 + u ea + v eb) serves both: tri0 is the half v <= u, tri1 the half v >= u (with tri0's own barycentrics
// (u', v') the point is p0 + u' ea + v' (ea + eb), i.e. u = u' + v', v = v': u' >= 0, v' >= 0, u' + v' <= 1 read
// u - v >= 0, v >= 0, u <= 1).  Same error margins as tri_lower_bound (|u, v - true| <= mu each, |t - true| <= mt; the
// constants cover edges up to the longest edge of any triangle, the diagonal included, and the <= 1e-10 relative by
// which a square's fourth corner may miss p0 + ea + eb - p1 in f64).  Out: *c0 / *c1 = the item's first / second
// triangle may be hit no farther than U (a single triangle only ever sets *c0), *tlo = the lower bound on that
// distance; *thi = an upper bound when one of them certainly is hit (untouched otherwise).
__device__ __forceinline__ void tri_item_bounds(const SceneDev& S, float ox, float oy, float oz, float dx, float dy, float dz,
                                                f3 p0, f3 e1, f3 e2, bool quad, float U, bool* c0, bool* c1, float* tlo,
                                                float* thi = nullptr) {
  const f3 d = mk3(dx, dy, dz);
  const f3 h = cross3(d, e2);
  const float a = dot3(e1, h);
  *c0 = false; *c1 = false; *tlo = 0.0f;
  if (fabsf(a) >= CFT_EPS - S.tKA) {                    // the reference rejects |a| < EPS (object.rs:105)
    const float f = __builtin_amdgcn_rcpf(a);
    const float af = fabsf(f);
    const f3 s = mk3(ox - p0.x, oy - p0.y, oz - p0.z);
    const float s1 = fabsf(s.x) + fabsf(s.y) + fabsf(s.z);
    const float u = f * dot3(s, h);
    const f3 q = cross3(s, e1);
    const float v = f * dot3(d, q);
    const float t = f * dot3(e2, q);
    const float mu = af * fmaf(s1, S.tKU1, S.tKU2);
    const float mt = af * fmaf(fabsf(t), S.tKT3, fmaf(s1, S.tKT1, S.tKT2));
    const float tl = fmaxf(t - mt, 0.0f);
    const bool reach = t + mt > CFT_EPS && tl <= U;
    const bool sure = thi && fabsf(a) >= CFT_EPS + S.tKA && t - mt > CFT_EPS;
    bool in0, in1 = false, sure_in;
    if (!quad) {
      in0 = u >= -mu && v >= -mu && u <= 1.0f + mu && u + v <= 1.0f + 2.0f * mu;
      sure_in = u >= mu && v >= mu && u + v <= 1.0f - 2.0f * mu;
    } else {
      const float w = u - v;
      in0 = v >= -mu && w >= -2.0f * mu && u <= 1.0f + mu;
      in1 = u >= -mu && w <= 2.0f * mu && v <= 1.0f + mu;
      sure_in = (v >= mu && w >= 2.0f * mu && u <= 1.0f - mu) || (u >= mu && w <= -2.0f * mu && v <= 1.0f - mu);
    }
    *c0 = in0 && reach; *c1 = in1 && reach; *tlo = tl;
    if (sure && sure_in) *thi = t + mt;
  }
}

// The static triangle items of the lists, 3 float4 each: (p0', meta bits), (e1 | ea), (e2 | eb); meta = (index of the
// item's first triangle in the triF list << 1) | quad.
struct TriItem { f3 p0, e1, e2; bool quad; int first; };
__device__ __forceinline__ TriItem tri_item(float4 a, float4 b, float4 c) {
  const int meta = __float_as_int(a.w);
  return TriItem{mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), (meta & 1) != 0, meta >> 1};
}
__device__ __forceinline__ TriItem tri_item(v4f a, v4f b, v4f c) {
  return tri_item(make_float4(a[0], a[1], a[2], a[3]), make_float4(b[0], b[1], b[2], b[3]), make_float4(c[0], c[1], c[2], c[3]));
}

// A path's radiance as it waits for k_resolve: 12 bytes (the resolve pass runs at the HBM's speed, a fourth float of
// padding was a quarter of its traffic).
struct Rad3 { float x, y, z; };

struct Closest {
  double t;   // exact (f64) distance, +inf on miss
  int g;      // global prim id, -1 on miss
};

// R rays per lane: the wave-uniform sphere operands (SGPRs) are fetched once for R tests, and one
// uniform branch per sphere serves all of them.  NS scans per ray: the reference scans the world up to
// twice per trace level (the lights' `looked`, then trace's own Ray::hit), each with fresh motion times;
// static primitives are tested once and feed both candidate sets, moving ones are tested per scan.
//
// Rays are held in PAIRS (slot r = element r&1 of pair r>>1) because the cull runs on v_pk_fma_f32: on
// gfx950 one v_pk_fma_f32 (two FMAs) issues in ~1.5 slots, a v_fma_f32 with an SGPR source in ~1.46
// (scripts/ubench_valu.hip), so the packed form is nearly twice as cheap per test.  An odd R leaves the
// last slot a dummy (thr4 = +inf never passes).
typedef float v2f __attribute__((ext_vector_type(2)));
template <int R, int NS>
struct RayPack {
  static constexpr int NP = (R + 1) / 2;
  // Registers held across the scan are the occupancy limit of k_intersect, so the pack carries only what
  // the scan reads: the origin as given is re-read from the ray queue by the epilogue, and the origin
  // relative to the scene centre is kept only as o4 = 4 o' (scaling by 4 is exact).
  v2f dx[NP], dy[NP], dz[NP];
  v2f o4x[NP], o4y[NP], o4z[NP];  // 4 * o', o' = origin relative to the scene centre S.bc (all f32 lists are relative to it)
  v2f od2[NP];                    // 2 * (o' . d)
  v2f thr4[NP];                   // fast-path threshold 4 (|o'|^2 - margin), folded into the cull's o4 chain
  float negE16[R];                // -16 * (margin of the robust discriminant): candidates have 16 disc >= negE16
  float U[R];                     // upper bound on the winner's distance (exact precise-sphere hit), f32 rounded up
  uint32_t word[NS][R];           // motion key of each scan
  Cand c[NS][R];
};
#define PK(P, f, r) ((P).f[(r) >> 1][(r) & 1])

template <int R, int NS>
__device__ __forceinline__ void pack_dummy_slot(RayPack<R, NS>& P, int r) {   // a slot that never is a candidate
  PK(P, dx, r) = 0.f; PK(P, dy, r) = 0.f; PK(P, dz, r) = 1.f;
  PK(P, o4x, r) = 0.f; PK(P, o4y, r) = 0.f; PK(P, o4z, r) = 0.f;
  PK(P, od2, r) = 0.f;
  PK(P, thr4, r) = __builtin_inff();
}

typedef const float __attribute__((address_space(4))) * c_f32p;

// The cull of one sphere against one ray pair, 7 v_pk_fma_f32.  The list stores C = -2 c' and
// k4 = 4 (|c'|^2 - r^2) (c' relative to the scene centre) as the SGPR pairs (Cx, Cy), (Cz, k4); the ray carries
// o4 = 4 o', od2 = 2 o'.d and thr4.  With Q = 2 q = od2 + C.d:
//     V = Q^2 - (thr4 + C.o4)      and      4 disc = Q^2 - k4 - C.o4 - 4 |o'|^2 = V - k4 + (thr4 - 4 |o'|^2)
// so "disc >= -margin" is "V >= k4" once thr4 = 4 (|o'|^2 - margin): the per-ray threshold rides in the
// chain's start value and the compare is against a scalar, which lets ONE v_cmp serve all R rays (max of V).
// The two chains are interleaved so that no instruction reads the result of the one before it.
__device__ __forceinline__ v2f pair_cull(v2f cxy, v2f czk, v2f dx, v2f dy, v2f dz, v2f od2, v2f o4x, v2f o4y, v2f o4z,
                                         v2f thr4) {
  v2f q, c, v;
  asm("v_pk_fma_f32 %0, %3, %7, %8 op_sel_hi:[0,1,1]\n\t"         // q  = Cz * dz + od2
      "v_pk_fma_f32 %1, %3, %11, %12 op_sel_hi:[0,1,1]\n\t"       // c  = Cz * o4z + thr4
      "v_pk_fma_f32 %0, %4, %6, %0 op_sel:[1,0,0]\n\t"            // q += Cy * dy
      "v_pk_fma_f32 %1, %4, %10, %1 op_sel:[1,0,0]\n\t"           // c += Cy * o4y
      "v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[0,1,1]\n\t"         // q += Cx * dx
      "v_pk_fma_f32 %1, %4, %9, %1 op_sel_hi:[0,1,1]\n\t"         // c += Cx * o4x
      "v_pk_fma_f32 %2, %0, %0, %1 neg_lo:[0,0,1] neg_hi:[0,0,1]"   // v  = q * q - c
      : "=&v"(q), "=&v"(c), "=&v"(v)
      : "s"(czk), "s"(cxy), "v"(dx), "v"(dy), "v"(dz), "v"(od2), "v"(o4x), "v"(o4y), "v"(o4z), "v"(thr4));
  return v;
}

// A sphere the cull could not rule out for ray slot r: recompute the cancellation-free form
// (q, disc = r^2 - |oc - q d|^2), turn it into a lower bound and insert it.  Everything is scaled by 4 (16 for
// squares), which is exact: 4 oc = 4 o' + 2 C is one rounding of the same real number as 4 (o' - c'), so
// q4 = 4 q and d16 = 16 disc bit for bit.
// ONCE: the caller may offer the same primitive more than once (the grid walk: a primitive sits in every cell it
// overlaps), and a primitive must not take two of the indexed slots.
template <bool ONCE>
__device__ __forceinline__ void cand_offer(Cand& c, float tlo, int idx) {
  if (!ONCE || (idx != c.i1 && idx != c.i2)) cand_insert(c, tlo, idx);
}

template <int R, int NS, bool ONCE = false>
__device__ __forceinline__ void sphere_candidate(const SceneDev& S, RayPack<R, NS>& P, int r, float Cx, float Cy,
                                                 float Cz, float r2x16, int idx, float* hi = nullptr) {
  const float dx = PK(P, dx, r), dy = PK(P, dy, r), dz = PK(P, dz, r);
  float ocx = fmaf(2.0f, Cx, PK(P, o4x, r)), ocy = fmaf(2.0f, Cy, PK(P, o4y, r)), ocz = fmaf(2.0f, Cz, PK(P, o4z, r));
  float q4 = fmaf(ocz, dz, fmaf(ocy, dy, ocx * dx));
  float lx = fmaf(-q4, dx, ocx), ly = fmaf(-q4, dy, ocy), lz = fmaf(-q4, dz, ocz);
  float d16 = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, r2x16)));
  if (d16 >= P.negE16[r]) {
    float thi = __builtin_inff();
    float tlo = sphere_lower_bound(S, 0.25f * q4, 0.0625f * d16, P.U[r], hi ? &thi : nullptr);
#ifdef CFT_STAMPS
    if (ONCE && g_walk_stats) {   // (diagnostic build: what becomes of the entries that reach the walk's candidate path)
      atomicAdd(&g_walk_stats[28], 1ull);
      if (tlo >= 0.0f) atomicAdd(&g_walk_stats[29], 1ull);
      if (thi < __builtin_inff()) atomicAdd(&g_walk_stats[30], 1ull);
    }
#endif
    if (tlo >= 0.0f) {
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        cand_offer<ONCE>(P.c[k][r], tlo, idx);
        if (hi) hi[k] = fminf(hi[k], thi);
      }
    }
  }
}

// One static sphere against R rays: cull (pair_cull), then the lanes that pass handle the candidate.
// (Queueing the passing lanes' sphere indices in LDS and draining them after the scan, every lane on its own
// sphere, was tried: 5 % fewer VALU instructions, no gain in time, +8 VGPRs - DESIGN.md §6.)
template <int R, int NS>
__device__ __forceinline__ void scan_sphere(const SceneDev& S, RayPack<R, NS>& P, v2f cxy, v2f czk, float r2x16,
                                            int idx) {
  constexpr int NP = RayPack<R, NS>::NP;
  v2f V[NP];
#pragma unroll
  for (int h = 0; h < NP; ++h)
    V[h] = pair_cull(cxy, czk, P.dx[h], P.dy[h], P.dz[h], P.od2[h], P.o4x[h], P.o4y[h], P.o4z[h], P.thr4[h]);
  // max over the slots: plain v_max / v_max3 (the values come out of FMAs; fmaxf would add canonicalising ops)
  float m;
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(V[0].x), "v"(V[0].y));
#pragma unroll
  for (int h = 1; h < NP; ++h) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(V[h].x), "v"(V[h].y));
  const float k4 = czk.y;
  if (__builtin_amdgcn_ballot_w64(m >= k4) != 0ull) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (V[r >> 1][r & 1] >= k4) sphere_candidate<R, NS>(S, P, r, cxy.x, cxy.y, czk.x, r2x16, idx);
  }
}


// Wave-level cull.  The 64 R rays of a wave often form a narrow bundle (primary rays of a pixel tile, mirror
// and refraction rays of neighbouring pixels): origins within rho of a reference origin, directions within an
// angle theta of a reference axis.  A ray of the bundle can only hit sphere (c, r) if the ray from the reference
// origin with the same direction hits the sphere inflated by rho, i.e. if angle(c - o_ref, axis) <= theta +
// asin(R'/L), R' = r + rho, L = |c - o_ref|:
//     v.a >= cos(theta) sqrt(L^2 - R'^2) - sin(theta) R'        (or L <= R')
// Lanes test 64 spheres at a time against that cone (vector loads, one lane per sphere), the ballot is walked
// bit by bit and only those spheres reach the per-ray cull.  Conservative by construction plus slack far above
// f32 rounding (0.1 % on R', 1e-5 (1 + L) absolute); `wave_cull` in the tests compares images with it off.
// Returns false (nothing done) when the bundle is wider than ~37 degrees: the caller scans the whole list.
typedef float v8f __attribute__((ext_vector_type(8)));
typedef const v8f __attribute__((address_space(4))) * c_f8p;

struct Bundle {
  float rox, roy, roz;   // reference origin (relative to the scene centre)
  float ax, ay, az;      // reference axis (a ray direction, unit)
  float cth, sth, rho;   // cos / sin of the half-angle (widened), radius of the origins around the reference
  bool live, narrow;     // any live ray in the wave; narrow enough for the cull to pay (< ~37 degrees)
};

template <int R, int NS>
__device__ __forceinline__ Bundle wave_bundle(const RayPack<R, NS>& P) {
  Bundle B;
  const float inf = __builtin_inff();
  const unsigned long long v0 = __builtin_amdgcn_ballot_w64(PK(P, thr4, 0) < inf);   // slot 0 fills first
  B.live = v0 != 0ull;
  B.narrow = false;
  B.rox = B.roy = B.roz = B.ax = B.ay = B.az = B.cth = B.sth = B.rho = 0.0f;
  if (!B.live) return B;
  const int lead = (int)__builtin_ctzll(v0);
#define CFT_RL(x) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lead))
  B.rox = 0.25f * CFT_RL(PK(P, o4x, 0)); B.roy = 0.25f * CFT_RL(PK(P, o4y, 0)); B.roz = 0.25f * CFT_RL(PK(P, o4z, 0));
  B.ax = CFT_RL(PK(P, dx, 0)); B.ay = CFT_RL(PK(P, dy, 0)); B.az = CFT_RL(PK(P, dz, 0));
#undef CFT_RL
  float d2max = 0.0f, cmin = 1.0f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (PK(P, thr4, r) < inf) {
      const float ex = fmaf(0.25f, PK(P, o4x, r), -B.rox), ey = fmaf(0.25f, PK(P, o4y, r), -B.roy), ez = fmaf(0.25f, PK(P, o4z, r), -B.roz);
      d2max = fmaxf(d2max, fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
      cmin = fminf(cmin, fmaf(PK(P, dz, r), B.az, fmaf(PK(P, dy, r), B.ay, PK(P, dx, r) * B.ax)));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    d2max = fmaxf(d2max, __shfl_xor(d2max, o));
    cmin = fminf(cmin, __shfl_xor(cmin, o));
  }
  B.cth = cmin - 4e-6f;                      // directions are unit to a few ulp
  B.narrow = B.cth > 0.8f;
  B.sth = __builtin_amdgcn_sqrtf(fmaxf(1.0f - B.cth * B.cth, 0.0f)) * 1.0001f + 1e-6f;   // (1-ulp square roots: the slack is 1e-4)
  B.rho = __builtin_amdgcn_sqrtf(d2max) * 1.0001f;
  return B;
}

// Can a ray of the bundle reach the sphere (centre c' relative to the scene centre, radius r)?
__device__ __forceinline__ bool bundle_reaches(const Bundle& B, float cx, float cy, float cz, float r) {
  const float vx = cx - B.rox, vy = cy - B.roy, vz = cz - B.roz;
  const float L2 = fmaf(vz, vz, fmaf(vy, vy, vx * vx));
  const float va = fmaf(vz, B.az, fmaf(vy, B.ay, vx * B.ax));
  const float slack = 1e-5f * (1.0f + __builtin_amdgcn_sqrtf(L2));   // (1-ulp square roots here: far inside the slack)
  const float Rp = (r + B.rho) * 1.001f + slack;
  const float h2 = fmaf(-Rp, Rp, L2);
  return h2 <= 0.0f || va >= B.cth * __builtin_amdgcn_sqrtf(fmaxf(h2, 0.0f)) - B.sth * Rp - slack;
}

template <int R, int NS>
__device__ __forceinline__ void scan_spheres_culled(const SceneDev& S, RayPack<R, NS>& P, const Bundle& B, int lo, int hi) {
  const int lane = (int)(threadIdx.x & 63u);
  for (int s0 = lo; s0 < hi; s0 += 64) {
    const int idx = s0 + lane;
    bool pass = false;
    if (idx < hi) {
      const float4 sp = S.sphF[idx];
      pass = bundle_reaches(B, -0.5f * sp.x, -0.5f * sp.y, -0.5f * sp.z, 0.25f * __builtin_amdgcn_sqrtf(S.sphR2[idx]));
    }
    unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
    if (m) {   // walk the set bits; the next sphere's scalar loads are in flight while the current one is tested
      int id = s0 + (int)__builtin_ctzll(m);
      m &= m - 1ull;
      v4f sp = AS_CONST(c_f4p, S.sphF)[id];
      float r2x16 = AS_CONST(c_f32p, S.sphR2)[id];
      while (true) {
        const int idn = m ? s0 + (int)__builtin_ctzll(m) : id;
        const v4f spn = AS_CONST(c_f4p, S.sphF)[idn];
        const float r2n = AS_CONST(c_f32p, S.sphR2)[idn];
        scan_sphere<R, NS>(S, P, v2f{sp[0], sp[1]}, v2f{sp[2], sp[3]}, r2x16, id);
        if (!m) break;
        m &= m - 1ull;
        id = idn; sp = spn; r2x16 = r2n;
      }
    }
  }
}

// A moving sphere whose swept bound the cull could not rule out for ray slot r (entry: 8 floats - centre at
// time 0, r^2, moving_to, object id bits): per scan, the centre at that scan's motion time (object.rs:42-55),
// the cancellation-free discriminant, the lower bound, the insert.
template <int R, int NS, bool ONCE = false>
__device__ __forceinline__ void msphere_candidate(const SceneDev& S, RayPack<R, NS>& P, int r, v8f cur, int idx,
                                                  float* hi = nullptr) {
  const uint32_t obj = __float_as_uint(cur[7]);
  const float dx = PK(P, dx, r), dy = PK(P, dy, r), dz = PK(P, dz, r);
  const float tx = 0.25f * PK(P, o4x, r), ty = 0.25f * PK(P, o4y, r), tz = 0.25f * PK(P, o4z, r);
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    float tm = motion_time(P.word[k][r], obj);
    float cx = fmaf(cur[4], tm, cur[0]), cy = fmaf(cur[5], tm, cur[1]), cz = fmaf(cur[6], tm, cur[2]);
    float ocx = tx - cx, ocy = ty - cy, ocz = tz - cz;
    float q = fmaf(ocz, dz, fmaf(ocy, dy, ocx * dx));
    float lx = fmaf(-q, dx, ocx), ly = fmaf(-q, dy, ocy), lz = fmaf(-q, dz, ocz);
    float d = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, cur[3])));
    if (16.0f * d >= P.negE16[r]) {
      float thi = __builtin_inff();
      float tlo = sphere_lower_bound(S, q, d, P.U[r], hi ? &thi : nullptr);
      if (tlo >= 0.0f) {
        cand_offer<ONCE>(P.c[k][r], tlo, idx);
        if (hi) hi[k] = fminf(hi[k], thi);
      }
    }
  }
}

// One moving sphere against R rays: the per-ray cull runs on the sphere's swept bound (an entry of S.msphB in the
// static list's format, so the 7 packed FMAs and the one compare serve it too); lanes that pass fetch the
// sphere itself and test it at their own motion times.  s = index in the moving list.
template <int R, int NS>
__device__ __forceinline__ void scan_msphere(const SceneDev& S, RayPack<R, NS>& P, v2f cxy, v2f czk, int s) {
  constexpr int NP = RayPack<R, NS>::NP;
  v2f V[NP];
#pragma unroll
  for (int h = 0; h < NP; ++h)
    V[h] = pair_cull(cxy, czk, P.dx[h], P.dy[h], P.dz[h], P.od2[h], P.o4x[h], P.o4y[h], P.o4z[h], P.thr4[h]);
  float m;
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(V[0].x), "v"(V[0].y));
#pragma unroll
  for (int h = 1; h < NP; ++h) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(V[h].x), "v"(V[h].y));
  const float k4 = czk.y;
  if (__builtin_amdgcn_ballot_w64(m >= k4) != 0ull) {
    const v8f cur = AS_CONST(c_f8p, S.msphF)[s];
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (V[r >> 1][r & 1] >= k4) msphere_candidate<R, NS>(S, P, r, cur, S.n_sphF + s);
  }
}

// ------------------------------------------------------------------ uniform grid (an extension: the reference scans linearly)
// Static spheres, the swept bounds of moving spheres and static triangles are also registered in a uniform grid
// over their bounding box (cftrt_scene_commit).  A wave whose bundle is too wide for the cone cull walks the grid
// instead of the lists: every lane takes its own ray through the cells it crosses (3D DDA) and tests the
// primitives registered there with the same f32 tests as the list scan, so the candidate sets, and with them the
// exact f64 result, are those of the full scan:
//  * the walk is the exact walk of the ray through a grid whose planes are displaced by rounding (a few ulp of
//    the box size per plane, plus the accumulated tMax rounding: < 1e-4 of the box half-extent E for origins
//    within 256 E); every primitive is registered with its bound inflated by 1e-3 E, so a primitive the ray hits is
//    met in one of the visited cells;
//  * the walk ends where it leaves the box, passes the ray's upper bound U (the exact precise-sphere hit), or
//    passes the upper bound of a candidate that certainly is a hit (the f32 error bounds can certify most hits).
//    In the last case the primitives not met are all farther than the exit of the last cell; that distance also
//    becomes the ray's third bound lo3, so the epilogue would rescan in f64 if the exact result did not beat it
//    (ray_epilogue) - the stop needs no trust in the certificate.
// Rays that start farther than 256 E from the scene centre are not walked: in the level kernels they go to the exact
// f64 rescan (lo3 = 0), in the one-ray form closest_hit the wave scans the lists.
// LISTS & 16 (walk instances of sphere-only scenes whose grid fits, k_trace_level): the grid's cell records and sphere
// entries are read from a copy in LDS.  A walk is a chain of dependent, divergent loads - cell record -> entries ->
// (16 r^2, index) of the entries that pass - and through the vector memory path each link is a round trip of several
// hundred cycles during which the lane's wave is parked (rocprofv3: 60 % of the wave cycles of the walk instances,
// VALU busy 0.65 at four waves per SIMD); from LDS a link costs tens of cycles.
#ifndef CFT_LDS_GRID_BYTES
#define CFT_LDS_GRID_BYTES (50 * 1024)    // one scan per level: beside 4 x 27 KB of park (3 rays per lane)
#define CFT_LDS_GRID_BYTES2 (38 * 1024)   // two scans per level (moving scenes with lights): beside 4 x 30 KB (2 rays per lane, 15 words each)
#endif
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef const v4f __attribute__((address_space(3)))* l_f4p;
typedef const float __attribute__((address_space(3)))* l_f32p;
typedef const uint32_t __attribute__((address_space(3)))* l_u32p;
struct LdsGrid {
  l_f4p sph;     // sphere entries
  l_u32p cell;   // cell records (2 words each)
  l_f32p x;      // (16 r^2, list index bits) per entry (2 words each)
};
#define CFT_NO_LDS_GRID LdsGrid{nullptr, nullptr, nullptr}

#ifndef CFT_WALK_ALWAYS
#define CFT_WALK_ALWAYS 2
#endif
#ifndef CFT_WALK_BATCH_PARKED
#define CFT_WALK_BATCH_PARKED 4
#endif
#ifndef CFT_WALK_BATCH
#define CFT_WALK_BATCH 2   // sphere entries of a cell tested per round of loads (measured: 2 beats 1 and 4 on S500 and S20k)
#endif
template <int R, int NS, int LISTS>
__device__ __forceinline__ void grid_walk_ray(const SceneDev& S, RayPack<R, NS>& P, int r, const LdsGrid lg = CFT_NO_LDS_GRID) {
  const float inf = __builtin_inff();
  constexpr bool LDSG = (LISTS & 16) != 0;
  auto cell_at = [&](uint32_t i) -> uint2 {
    if constexpr (LDSG) return make_uint2(lg.cell[2u * i], lg.cell[2u * i + 1u]);   // (one ds_read_b64)
    else return S.gridCell[i];
  };
  auto sph_at = [&](uint32_t e) -> float4 {
    if constexpr (LDSG) { const v4f a = lg.sph[e]; return make_float4(a[0], a[1], a[2], a[3]); }
    else return S.gridSph[e];
  };
  auto sphx_at = [&](uint32_t e) -> float2 {
    if constexpr (LDSG) return make_float2(lg.x[2u * e], lg.x[2u * e + 1u]);
    else return S.gridSphX[e];
  };
  // entries per round of loads: a one-ray pack (the instances that park the others) leaves registers for more
  constexpr int NB = R == 1 ? CFT_WALK_BATCH_PARKED : CFT_WALK_BATCH;
  const float dx = PK(P, dx, r), dy = PK(P, dy, r), dz = PK(P, dz, r);
  const float ox = 0.25f * PK(P, o4x, r), oy = 0.25f * PK(P, o4y, r), oz = 0.25f * PK(P, o4z, r);
  // reciprocals that stay finite (a zero component never reaches a plane: |t| ~ 1e30 * distance)
  const float ix_ = __builtin_amdgcn_rcpf(fabsf(dx) < 1e-30f ? copysignf(1e-30f, dx) : dx);
  const float iy_ = __builtin_amdgcn_rcpf(fabsf(dy) < 1e-30f ? copysignf(1e-30f, dy) : dy);
  const float iz_ = __builtin_amdgcn_rcpf(fabsf(dz) < 1e-30f ? copysignf(1e-30f, dz) : dz);
  // slab clip against the (inflated) box
  const float ax0 = (S.g0x - ox) * ix_, ax1 = (S.g1x - ox) * ix_;
  const float ay0 = (S.g0y - oy) * iy_, ay1 = (S.g1y - oy) * iy_;
  const float az0 = (S.g0z - oz) * iz_, az1 = (S.g1z - oz) * iz_;
  const float t0 = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fmaxf(fminf(az0, az1), 0.0f));
  const float texit = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1));
  const float tend = fminf(texit, fmaf(P.U[r], 1.0001f, 1e-4f));
  if (!(t0 <= tend)) return;   // misses the box, or the box lies beyond U
  // The walk runs from the entry point p = o + t0 d with distances counted from there, so that its rounding scales
  // with the box and not with how far away the ray started (a bounce off the ground sphere near the horizon).
  const float px = fmaf(dx, t0, ox), py = fmaf(dy, t0, oy), pz = fmaf(dz, t0, oz);
  const float tendr = (tend - t0) + fmaf(t0, 1e-6f, 1e-30f);
  int cx = (int)floorf((px - S.g0x) * S.gicx), cy = (int)floorf((py - S.g0y) * S.gicy), cz = (int)floorf((pz - S.g0z) * S.gicz);
  cx = min(max(cx, 0), S.gnx - 1); cy = min(max(cy, 0), S.gny - 1); cz = min(max(cz, 0), S.gnz - 1);
  // distance (from p) to the next plane of each axis, and between planes (signed like the direction: the step)
  float tmx = (fmaf((float)(cx + (dx >= 0.0f ? 1 : 0)), S.gcsx, S.g0x) - px) * ix_;
  float tmy = (fmaf((float)(cy + (dy >= 0.0f ? 1 : 0)), S.gcsy, S.g0y) - py) * iy_;
  float tmz = (fmaf((float)(cz + (dz >= 0.0f ? 1 : 0)), S.gcsz, S.g0z) - pz) * iz_;
  const float sdx = S.gcsx * ix_, sdy = S.gcsy * iy_, sdz = S.gcsz * iz_;
  const int stx = dx >= 0.0f ? 1 : -1, sty = dy >= 0.0f ? 1 : -1, stz = dz >= 0.0f ? 1 : -1;
  // |2 oc.d - true| <= ~24 eps (|o'| + |c'|): eQ = 64 eps (|o'| + brad)
  const float eQ = 3.9e-6f * fmaf(__builtin_amdgcn_sqrtf(fmaf(oz, oz, fmaf(oy, oy, ox * ox))), 1.000001f, S.brad);
  int guard = S.gnx + S.gny + S.gnz + 3;
  float hi[NS];        // per scan: smallest upper bound among the candidates that certainly are hits
#pragma unroll
  for (int k = 0; k < NS; ++k) hi[k] = inf;
  float tstop = inf;   // lower bound on the distance of every primitive the walk did not meet
  // Cell record: x = (first sphere entry << 8) | count, y = (first triangle entry << 12) | (count << 4).
  // The walk runs one cell AHEAD of the tests: the step
  // to the next cell does not depend on what the current cell holds, so the next record is requested before the
  // current cell's entries are tested and its latency hides under them (a wave has one ray per lane in flight and a
  // SIMD four such waves: the dependent record -> entries loads were what the walk waited for).
  uint2 cell = cell_at(__umul24(__umul24((uint32_t)cz, (uint32_t)S.gny) + (uint32_t)cy, (uint32_t)S.gnx) + (uint32_t)cx);   // < 2^21
#ifdef CFT_STAMPS
  uint32_t wk_cells = 0, wk_tests = 0, wk_pass = 0;
  // how many lanes are in each part of the loop when the wave runs it: executions are counted by the first active lane
  uint32_t wk_xstep = 0, wk_xbatch = 0, wk_xcand = 0, wk_batches = 0;
#define CFT_WK_LEADER() ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)__ballot(1)) - 1u)
#endif
  while (true) {
#ifdef CFT_STAMPS
    wk_cells += 1; wk_tests += (cell.x & 255u) + ((cell.y >> 4) & 255u);
    wk_xstep += CFT_WK_LEADER() ? 1u : 0u;
#endif
    const float tnr = fminf(tmx, fminf(tmy, tmz));   // exit of this cell, from p
    {   // step into the neighbour across the nearest plane (straight-line: three-way branches diverge)
      const bool ax = tmx <= tmy && tmx <= tmz, ay = !ax && tmy <= tmz, az = !ax && !ay;
      cx += ax ? stx : 0; cy += ay ? sty : 0; cz += az ? stz : 0;
      tmx += ax ? fabsf(sdx) : 0.0f; tmy += ay ? fabsf(sdy) : 0.0f; tmz += az ? fabsf(sdz) : 0.0f;
    }
    const bool inside = (unsigned)cx < (unsigned)S.gnx && (unsigned)cy < (unsigned)S.gny && (unsigned)cz < (unsigned)S.gnz;
    uint2 next = make_uint2(0u, 0u);
    if (inside && tnr <= tendr)
      next = cell_at(__umul24(__umul24((uint32_t)cz, (uint32_t)S.gny) + (uint32_t)cy, (uint32_t)S.gnx) + (uint32_t)cx);
    // The cell's sphere entries, NB at a time: the 16 B loads of a batch are in flight together (one after
    // the other they are a chain of dependent round trips: a cell of a 20 k-primitive scene holds up to 19 entries),
    // the cull of each is 8 FMAs, and only entries that pass go on to the (rare, divergent) candidate path.  A cell
    // visited holds 1.3 (S500) to 2.2 (S20k) entries on average: batches of 4 mostly test clamped repeats.
    for (uint32_t e0 = cell.x >> 8, ee = (cell.x >> 8) + (cell.x & 255u); e0 < ee; e0 += (uint32_t)NB) {
      float4 a4[NB];
#ifdef CFT_STAMPS
      wk_batches += 1; wk_xbatch += CFT_WK_LEADER() ? 1u : 0u;
#endif
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        // the first two of a batch are loaded whatever the cell holds (clamped repeats); the others only by lanes whose
        // cell has that many entries left (a divergent load costs the texture path by its active lanes)
        if (j < CFT_WALK_ALWAYS) a4[j] = sph_at(min(e0 + (uint32_t)j, ee - 1u));
        else if (e0 + (uint32_t)j < ee) a4[j] = sph_at(e0 + (uint32_t)j);
        else a4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      uint32_t pass = 0u;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        // the per-ray cull of scan_sphere, unpacked (same operations in the same order): q = 2 oc.d, pass iff the
        // LINE meets the sphere.  Two cheap rejects keep lanes out of the (divergent) candidate path:
        //  * behind: origin certainly outside (c + k4 = 4 (|oc|^2 - r^2) - 4 E + err > 0, the margin E inside thr4
        //    exceeds err) and certainly moving away (q > eQ): both roots negative;
        //  * seen (below): the primitive is already an indexed candidate (it sits in every cell it overlaps).
        const float4 a = a4[j];
        float q = fmaf(a.z, dz, PK(P, od2, r)), c = fmaf(a.z, PK(P, o4z, r), PK(P, thr4, r));
        q = fmaf(a.y, dy, q); c = fmaf(a.y, PK(P, o4y, r), c);
        q = fmaf(a.x, dx, q); c = fmaf(a.x, PK(P, o4x, r), c);
        const bool behind = q > eQ && c + a.w > 0.0f;
        if (e0 + (uint32_t)j < ee && fmaf(q, q, -c) >= a.w && !behind) pass |= 1u << j;
      }
      while (pass) {
#ifdef CFT_STAMPS
        wk_pass += 1; wk_xcand += CFT_WK_LEADER() ? 1u : 0u;
#endif
        const uint32_t e = e0 + (uint32_t)__builtin_ctz(pass);
        pass &= pass - 1u;
        const float4 a = sph_at(e);    // (again: the batch sits in registers under a constant index only)
        const float2 b = sphx_at(e);   // (16 r^2, list index)
        const int idx = __float_as_int(b.y);
        bool seen = true;
#pragma unroll
        for (int k = 0; k < NS; ++k) seen = seen && (idx == P.c[k][r].i1 || idx == P.c[k][r].i2);
        if (seen) continue;
        if (!(LISTS & 1) || idx < S.n_sphF) {
          sphere_candidate<R, NS, true>(S, P, r, a.x, a.y, a.z, b.x, idx, hi);
        } else {
          const float4 m0 = S.msphF[2 * (idx - S.n_sphF)], m1 = S.msphF[2 * (idx - S.n_sphF) + 1];
          msphere_candidate<R, NS, true>(S, P, r, v8f{m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w}, idx, hi);
        }
      }
    }
    if (LISTS & 2) {
      for (uint32_t e = cell.y >> 12, ee = (cell.y >> 12) + ((cell.y >> 4) & 255u); e < ee; ++e) {
        const int s = (int)S.gridTri[e];   // an item: one triangle, or the two of a square as one parallelogram
        const TriItem it = tri_item(S.triItem[3 * s], S.triItem[3 * s + 1], S.triItem[3 * s + 2]);
        float thi = inf, tlo;
        bool c0, c1;
        tri_item_bounds(S, ox, oy, oz, dx, dy, dz, it.p0, it.e1, it.e2, it.quad, P.U[r], &c0, &c1, &tlo, &thi);
        if (c0 || c1) {
#pragma unroll
          for (int k = 0; k < NS; ++k) {
            if (c0) cand_offer<true>(P.c[k][r], tlo, S.n_sphF + S.n_msph + it.first);
            if (c1) cand_offer<true>(P.c[k][r], tlo, S.n_sphF + S.n_msph + it.first + 1);
            hi[k] = fminf(hi[k], thi);
          }
        }
      }
    }
    if (!(tnr <= tendr)) break;                      // out of the box or beyond U: nothing left that could win
    const float tn = t0 + tnr;
    float best = hi[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) best = fmaxf(best, hi[k]);
    const float tcut = fmaf(tn, 0.9999f, -1e-4f);   // every primitive not met is farther than this
    if (tcut > best || --guard < 0) {
      // a certain hit nearer than everything not met: stop (guard: cannot happen; the rescan then decides)
      tstop = guard < 0 ? 0.0f : tcut;
      break;
    }
    if (!inside) break;
    cell = next;
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) P.c[k][r].lo3 = fminf(P.c[k][r].lo3, tstop);
#ifdef CFT_STAMPS
  if (g_walk_stats) {
    atomicAdd(&g_walk_stats[0], 1ull); atomicAdd(&g_walk_stats[1], (unsigned long long)wk_cells);
    atomicAdd(&g_walk_stats[2], (unsigned long long)wk_tests); atomicAdd(&g_walk_stats[3], (unsigned long long)wk_pass);
    // histogram of cells walked: bins 0-3, 4-7, 8-15, 16-31, 32-63, 64+
    const int bin = wk_cells < 4 ? 0 : wk_cells < 8 ? 1 : wk_cells < 16 ? 2 : wk_cells < 32 ? 3 : wk_cells < 64 ? 4 : 5;
    atomicAdd(&g_walk_stats[8 + bin], 1ull);
    const int tb = wk_tests < 4 ? 0 : wk_tests < 8 ? 1 : wk_tests < 16 ? 2 : wk_tests < 32 ? 3 : wk_tests < 64 ? 4 : wk_tests < 128 ? 5 : 6;
    atomicAdd(&g_walk_stats[16 + tb], 1ull);
    atomicAdd(&g_walk_stats[24], (unsigned long long)wk_xstep); atomicAdd(&g_walk_stats[25], (unsigned long long)wk_xbatch);
    atomicAdd(&g_walk_stats[26], (unsigned long long)wk_xcand); atomicAdd(&g_walk_stats[27], (unsigned long long)wk_batches);
  }
#endif
}

// One static triangle item (wave-uniform: its data sit in SGPRs) against the R rays of a lane.
template <int R, int NS>
__device__ __forceinline__ void scan_tri_item(const SceneDev& S, RayPack<R, NS>& P, const TriItem it) {
  const int base = S.n_sphF + S.n_msph + it.first;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    bool c0, c1;
    float tlo;
    tri_item_bounds(S, 0.25f * PK(P, o4x, r), 0.25f * PK(P, o4y, r), 0.25f * PK(P, o4z, r), PK(P, dx, r), PK(P, dy, r),
                    PK(P, dz, r), it.p0, it.e1, it.e2, it.quad, P.U[r], &c0, &c1, &tlo);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      if (c0) cand_insert(P.c[k][r], tlo, base);
      if (c1) cand_insert(P.c[k][r], tlo, base + 1);
    }
  }
}

// Moving triangles (4 float4 each: p0, e1, e2, (mv, object id)), once per scan with that scan's motion time.
template <int R, int NS>
__device__ __forceinline__ void scan_mtris(const SceneDev& S, RayPack<R, NS>& P, int base, int part, int nparts) {
  if (!S.n_mtri) return;
  c_f4p tp = AS_CONST(c_f4p, S.mtriF);
  for (int s = S.n_mtri * part / nparts; s < S.n_mtri * (part + 1) / nparts; ++s) {
    v4f a = tp[4 * s], b = tp[4 * s + 1], cc = tp[4 * s + 2], m = tp[4 * s + 3];
    const uint32_t obj = __float_as_uint(m.w);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float tm = motion_time(P.word[k][r], obj);
        float tlo = tri_lower_bound(S, 0.25f * PK(P, o4x, r), 0.25f * PK(P, o4y, r), 0.25f * PK(P, o4z, r), PK(P, dx, r),
                                    PK(P, dy, r), PK(P, dz, r),
                                    mk3(fmaf(m.x, tm, a.x), fmaf(m.y, tm, a.y), fmaf(m.z, tm, a.z)),
                                    mk3(b.x, b.y, b.z), mk3(cc.x, cc.y, cc.z), P.U[r]);
        if (tlo >= 0.0f) cand_insert(P.c[k][r], tlo, base + s);
      }
    }
  }
}

// ---- per-tile lists (primary rays of the default pipeline; k_tile_bin below builds them per render call).
// A wave of the level-0 launch holds 64 R consecutive queue slots = pixels of ONE 32x8 tile and one sample, so instead
// of putting every listed primitive through the cone test (scan_spheres_culled: n / 64 rounds per wave, which is what
// made the grid walk the cheaper way for primary rays above 4 096 primitives) it reads the tile's list: the primitives
// whose bounding sphere, widened for the lens, meets the tile's pyramid.  The list is a conservative superset of what
// any ray of the tile can hit, so everything downstream - per-ray cull, candidate sets, exact evaluation - is what the
// full scan would do for those primitives, and the image is the same bit for bit (tests: tile lists on / off).
// n0 .. n1 = the tile's range in S.tile_idx.  (narrow: also put each entry through the wave's cone test first; the
// level kernels pass false - on lists this tight the test costs more than it removes.)
template <int R, int NS, int LISTS>
__device__ __forceinline__ void scan_tile_list(const SceneDev& S, RayPack<R, NS>& P, const Bundle& B, bool narrow,
                                               uint32_t n0, uint32_t n1) {
  const uint32_t lane = threadIdx.x & 63u;
  const int nS = S.n_sphF, nSM = S.n_sphF + S.n_msph;
  for (uint32_t s0 = n0; s0 < n1; s0 += 64u) {
    int idx = -1;
    bool pass = false;
    if (s0 + lane < n1) {
      idx = (int)S.tile_idx[s0 + lane];
      pass = true;
      if (narrow) {
        const float4 b = S.boundF[idx];
        pass = bundle_reaches(B, b.x, b.y, b.z, b.w);
      }
    }
    unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
    while (m) {
      const int src = (int)__builtin_ctzll(m);
      m &= m - 1ull;
      const int id = __builtin_amdgcn_readlane(idx, src);   // wave-uniform: the primitive comes through scalar loads
      if (!(LISTS & 3) || id < nS) {
        const v4f sp = AS_CONST(c_f4p, S.sphF)[id];
        scan_sphere<R, NS>(S, P, v2f{sp[0], sp[1]}, v2f{sp[2], sp[3]}, AS_CONST(c_f32p, S.sphR2)[id], id);
      } else if ((LISTS & 1) && id < nSM) {
        const v4f sb = AS_CONST(c_f4p, S.msphB)[id - nS];
        scan_msphere<R, NS>(S, P, v2f{sb[0], sb[1]}, v2f{sb[2], sb[3]}, id - nS);
      } else if (LISTS & 2) {
        c_f4p tp = AS_CONST(c_f4p, S.triF);
        const int t = id - nSM;
        const v4f a = tp[3 * t], b = tp[3 * t + 1], cc = tp[3 * t + 2];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float tlo = tri_lower_bound(S, 0.25f * PK(P, o4x, r), 0.25f * PK(P, o4y, r), 0.25f * PK(P, o4z, r), PK(P, dx, r),
                                            PK(P, dy, r), PK(P, dz, r), mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z),
                                            mk3(cc.x, cc.y, cc.z), P.U[r]);
          if (tlo >= 0.0f) {
#pragma unroll
            for (int k = 0; k < NS; ++k) cand_insert(P.c[k][r], tlo, id);
          }
        }
      }
    }
  }
}

// (the LDS park of the level kernels, described at isect_setup below: words of slot r in the lane's own column)
#define CFT_PARKED_WALK(LISTS, R) (((LISTS) & 2) != 0 && (R) > 1)   // which walk instances walk one-ray packs (scan_lists)
template <int R, int NS> __device__ __forceinline__ void isect_park(const RayPack<R, NS>& P, float (*park)[CFT_BLOCK]);
template <int NS> struct ParkLayout {
  static constexpr int CAND = 3, WORD = 3 + 5 * NS, WORDS = (4 + 6 * NS) > 9 ? (3 + 6 * NS) : 9;
};
#define PARK(r, j) park[(r) * ParkLayout<NS>::WORDS + (j)][CFT_TID]

// f32 scan of every non-precise primitive for R rays (indices into [sphF | msph | triF | mtri]).
// part / nparts (wave-uniform): this wave scans the part-th slice of each list (nparts = 1: all of it).
// WCULL: try the wave-level cull first (the R-rays-per-lane kernels; every lane of the wave must be active).
// LISTS (compile time): 1 = the scene may hold moving spheres, 2 = triangles, 4 = the scene has a grid and the ctx
// uses it; a scene of static spheres only, or a pass on the plain lists, runs kernels without the other code (and
// its registers).
// tile >= 0 (level 0 of the default pipeline, with S.tile_off set): the wave's rays are pixels of that tile, and its
// list stands for the static / moving spheres and static triangles.
template <int R, int NS, bool WCULL = false, int LISTS = 3>
__device__ __forceinline__ void scan_lists(const SceneDev& S, RayPack<R, NS>& P, int part = 0, int nparts = 1, int tile = -1,
                                           float (*park)[CFT_BLOCK] = nullptr, const LdsGrid lg = CFT_NO_LDS_GRID) {
  const int n4_all = (S.n_sphF + 3) >> 2;
  const int lo4 = n4_all * part / nparts, hi4 = n4_all * (part + 1) / nparts;
  Bundle B = {};
  bool cull = false;
  // (a wave that takes its tile's list needs no bundle: the list already is what the tile's rays can reach, and the cone
  // test per entry costs more than the few entries it removes - level 0 of S20k 8.8 -> 8.6 ms without it, before
  // counting the bundle's own reductions)
  const bool by_tile = WCULL && !(LISTS & 8) && tile >= 0;
  if (WCULL && S.wave_cull && !by_tile) {
    B = wave_bundle<R, NS>(P);
    if (!B.live) return;          // no live ray in this wave
    cull = B.narrow;              // wider than ~37 degrees: scan the whole lists
  }
  // the grid instead of the lists (mode 1: for waves the cone cull does not serve; 2: for every wave - also what the
  // host makes of mode 1 once the lists are so long, 4 096 entries, that even one cone test per primitive and wave
  // costs more than the walks), unless a ray of the wave starts too far out for the walk's error bounds
  // LISTS & 8: the caller guarantees a grid and sends every wave through it (the level kernels' walk instances);
  // a ray that starts too far out is then not walked but handed to the exact f64 rescan (lo3 = 0), and the
  // instance carries no list-scan code at all.
  bool grid = (LISTS & 8) != 0;
  if (!(LISTS & 8) && (LISTS & 4) && nparts == 1 && S.grid_mode &&
      (S.grid_mode == 2 || !cull)) {
    bool far = false;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (PK(P, thr4, r) < __builtin_inff())
        far |= fmaxf(fabsf(PK(P, o4x, r)), fmaxf(fabsf(PK(P, o4y, r)), fabsf(PK(P, o4z, r)))) > S.gfar4;
    grid = __builtin_amdgcn_ballot_w64(far) == 0ull;
  }
  if (grid) {
    // moving triangles are not in the grid; the walk comes last (it may lower lo3 below lo2, which no insert may follow)
    // the few primitives too big for the grid (the head of each list): linear, as in the list scan
    for (int i = 0; i < S.nb_sph; ++i) {
      const v4f sp = AS_CONST(c_f4p, S.sphF)[i];
      scan_sphere<R, NS>(S, P, v2f{sp[0], sp[1]}, v2f{sp[2], sp[3]}, AS_CONST(c_f32p, S.sphR2)[i], i);
    }
    if (LISTS & 1)
      for (int i = 0; i < S.nb_msph; ++i) {
        const v4f sb = AS_CONST(c_f4p, S.msphB)[i];
        scan_msphere<R, NS>(S, P, v2f{sb[0], sb[1]}, v2f{sb[2], sb[3]}, i);
      }
    if (LISTS & 2) {
      c_f4p tp = AS_CONST(c_f4p, S.triItem);
      for (int s = 0; s < S.nb_item; ++s) scan_tri_item<R, NS>(S, P, tri_item(tp[3 * s], tp[3 * s + 1], tp[3 * s + 2]));
      scan_mtris<R, NS>(S, P, S.n_sphF + S.n_msph + S.n_triF, 0, 1);
    }
    if constexpr ((LISTS & 8) != 0 && CFT_PARKED_WALK(LISTS, R)) {
      // The walk instances of the level kernels (park given) walk their R slots one after the other, and a walk touches
      // only its own slot's candidate set, bounds and motion words.  So those leave the registers here: the candidate
      // sets go to the park now (where the epilogue wants them anyway), and each walk runs on a ONE-ray pack that
      // takes its slot's set from the park and puts it back - 5 live candidate words instead of 5 R, and the slot's
      // upper bound and margin recomputed from what the park and the pack hold (the R-ray pack kept all of that live
      // across every walk: the walk instances sat at the 128-register cap, and what they spilled cost whole percents).
      // Only the instances with triangle code walk this way (S20k -2.4 %); the sphere-only ones lost 2.8 % to it.
      isect_park<R, NS>(P, park);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (PK(P, thr4, r) < __builtin_inff()) {
          if (fmaxf(fabsf(PK(P, o4x, r)), fmaxf(fabsf(PK(P, o4y, r)), fabsf(PK(P, o4z, r)))) > S.gfar4) {
#pragma unroll
            for (int k = 0; k < NS; ++k) PARK(r, ParkLayout<NS>::CAND + 5 * k + 2) = 0.0f;   // too far out for the walk's error bounds: lo3 = 0, rescan
          } else {
            RayPack<1, NS> Q;
            Q.dx[0] = v2f{PK(P, dx, r), 0.f}; Q.dy[0] = v2f{PK(P, dy, r), 0.f}; Q.dz[0] = v2f{PK(P, dz, r), 1.f};
            Q.o4x[0] = v2f{PK(P, o4x, r), 0.f}; Q.o4y[0] = v2f{PK(P, o4y, r), 0.f}; Q.o4z[0] = v2f{PK(P, o4z, r), 0.f};
            Q.od2[0] = v2f{PK(P, od2, r), 0.f}; Q.thr4[0] = v2f{PK(P, thr4, r), __builtin_inff()};
            {   // ray_setup's margin and upper bound again (same operations on the same numbers)
              const float tx = 0.25f * PK(P, o4x, r), ty = 0.25f * PK(P, o4y, r), tz = 0.25f * PK(P, o4z, r);
              const float dist = __builtin_amdgcn_sqrtf(fmaf(tz, tz, fmaf(ty, ty, tx * tx))) * 1.000001f + S.brad;
              Q.negE16[0] = -16.0f * fmaf(dist, S.K1, S.K2);
              const float u = (float)__hiloint2double(__float_as_int(PARK(r, 1)), __float_as_int(PARK(r, 0)));
              Q.U[0] = (__float_as_int(PARK(r, 2)) >= 0) ? fmaf(fabsf(u), CFT_U4EPS, u) + 1e-30f : __builtin_inff();
              if (NS == 2 && S.prec_motion) Q.U[0] = __builtin_inff();
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
              Q.word[k][0] = (LISTS & 1) ? __float_as_uint(PARK(r, ParkLayout<NS>::WORD + k)) : 0u;
              Q.c[k][0].lo1 = PARK(r, ParkLayout<NS>::CAND + 5 * k + 0);
              Q.c[k][0].lo2 = PARK(r, ParkLayout<NS>::CAND + 5 * k + 1);
              Q.c[k][0].lo3 = PARK(r, ParkLayout<NS>::CAND + 5 * k + 2);
              Q.c[k][0].i1 = __float_as_int(PARK(r, ParkLayout<NS>::CAND + 5 * k + 3));
              Q.c[k][0].i2 = __float_as_int(PARK(r, ParkLayout<NS>::CAND + 5 * k + 4));
            }
            grid_walk_ray<1, NS, LISTS>(S, Q, 0, lg);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
              PARK(r, ParkLayout<NS>::CAND + 5 * k + 0) = Q.c[k][0].lo1;
              PARK(r, ParkLayout<NS>::CAND + 5 * k + 1) = Q.c[k][0].lo2;
              PARK(r, ParkLayout<NS>::CAND + 5 * k + 2) = Q.c[k][0].lo3;
              PARK(r, ParkLayout<NS>::CAND + 5 * k + 3) = __int_as_float(Q.c[k][0].i1);
              PARK(r, ParkLayout<NS>::CAND + 5 * k + 4) = __int_as_float(Q.c[k][0].i2);
            }
          }
        }
      return;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (PK(P, thr4, r) < __builtin_inff()) {
        if ((LISTS & 8) && fmaxf(fabsf(PK(P, o4x, r)), fmaxf(fabsf(PK(P, o4y, r)), fabsf(PK(P, o4z, r)))) > S.gfar4) {
#pragma unroll
          for (int k = 0; k < NS; ++k) P.c[k][r].lo3 = 0.0f;   // too far out for the walk's error bounds: rescan
        } else {
          grid_walk_ray<R, NS, LISTS>(S, P, r, lg);
        }
      }
    return;
  }
  if (LISTS & 8) return;   // (not reached)
  if (by_tile) {
    const uint32_t n0 = AS_CONST(const uint32_t __attribute__((address_space(4)))*, S.tile_off)[tile];
    const uint32_t n1 = AS_CONST(const uint32_t __attribute__((address_space(4)))*, S.tile_off)[tile + 1];
    bool any = false;
#pragma unroll
    for (int r = 0; r < R; ++r) any = any || PK(P, thr4, r) < __builtin_inff();
    if (__builtin_amdgcn_ballot_w64(any) != 0ull) scan_tile_list<R, NS, LISTS>(S, P, B, false, n0, n1);
    if (LISTS & 2) scan_mtris<R, NS>(S, P, S.n_sphF + S.n_msph + S.n_triF, part, nparts);
    return;
  }
  if (cull) {
    scan_spheres_culled<R, NS>(S, P, B, lo4 << 2, min(hi4 << 2, S.n_sphF));
  } else {
    // 4 spheres per s_load_dwordx16 (+ their r^2 in one s_load_dwordx4), fetched one batch ahead; both
    // arrays are padded on the host with never-hit spheres (k4 = +1e30) to a multiple of 4 plus one batch.
    c_f16p sp = AS_CONST(c_f16p, S.sphF);
    c_f4p rp = AS_CONST(c_f4p, S.sphR2);
    v16f cur = sp[lo4];
    v4f cr = rp[lo4];
    // (the first batch is waited for HERE: scalar loads return out of order, so the only wait there is is "all of them";
    // left to the compiler it sits at the head of the loop, right behind the prefetch of the next batch - needed on the
    // first trip only, paid on every one: the prefetch then covers nothing)
    asm volatile("" : : "s"(cur[0]), "s"(cr[0]));
    for (int s4 = lo4; s4 < hi4; ++s4) {
      v16f nxt = sp[s4 + 1];
      v4f nr = rp[s4 + 1];
      const int s = s4 << 2;
      scan_sphere<R, NS>(S, P, v2f{cur[0], cur[1]}, v2f{cur[2], cur[3]}, cr[0], s);
      scan_sphere<R, NS>(S, P, v2f{cur[4], cur[5]}, v2f{cur[6], cur[7]}, cr[1], s + 1);
      scan_sphere<R, NS>(S, P, v2f{cur[8], cur[9]}, v2f{cur[10], cur[11]}, cr[2], s + 2);
      scan_sphere<R, NS>(S, P, v2f{cur[12], cur[13]}, v2f{cur[14], cur[15]}, cr[3], s + 3);
      cur = nxt;
      cr = nr;
    }
  }
  int base = S.n_sphF;
  if ((LISTS & 1) && S.n_msph) {
    // moving spheres: the cull runs on their swept bounds (S.msphB, same format and padding as sphF)
    const int m4_all = (S.n_msph + 3) >> 2;
    const int mlo4 = m4_all * part / nparts, mhi4 = m4_all * (part + 1) / nparts;
    if (cull) {
      // the swept bound against the wave's cone, one lane per sphere; bound centre = -C/2, radius from k4
      const int lane = (int)(threadIdx.x & 63u);
      const int lo = mlo4 << 2, n = min(mhi4 << 2, S.n_msph);
      for (int s0 = lo; s0 < n; s0 += 64) {
        const int idx = s0 + lane;
        bool pass = false;
        if (idx < n) {
          const float4 c0 = S.msphF[2 * idx], mv = S.msphF[2 * idx + 1];
          const float hm = 0.5f * __builtin_amdgcn_sqrtf(fmaf(mv.z, mv.z, fmaf(mv.y, mv.y, mv.x * mv.x))) * 1.0001f;
          pass = bundle_reaches(B, fmaf(0.5f, mv.x, c0.x), fmaf(0.5f, mv.y, c0.y), fmaf(0.5f, mv.z, c0.z),
                                __builtin_amdgcn_sqrtf(c0.w) * 1.0001f + hm);
        }
        unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
        while (m) {
          const int id = s0 + (int)__builtin_ctzll(m);
          m &= m - 1ull;
          const v4f sb = AS_CONST(c_f4p, S.msphB)[id];
          scan_msphere<R, NS>(S, P, v2f{sb[0], sb[1]}, v2f{sb[2], sb[3]}, id);
        }
      }
    } else {
      c_f16p sp = AS_CONST(c_f16p, S.msphB);
      v16f cur = sp[mlo4];
      asm volatile("" : : "s"(cur[0]));   // (as above)
      for (int s4 = mlo4; s4 < mhi4; ++s4) {
        v16f nxt = sp[s4 + 1];
        const int s = s4 << 2;
        scan_msphere<R, NS>(S, P, v2f{cur[0], cur[1]}, v2f{cur[2], cur[3]}, s);
        scan_msphere<R, NS>(S, P, v2f{cur[4], cur[5]}, v2f{cur[6], cur[7]}, s + 1);
        scan_msphere<R, NS>(S, P, v2f{cur[8], cur[9]}, v2f{cur[10], cur[11]}, s + 2);
        scan_msphere<R, NS>(S, P, v2f{cur[12], cur[13]}, v2f{cur[14], cur[15]}, s + 3);
        cur = nxt;
      }
    }
  }
  base += S.n_msph;
  if (LISTS & 2) {
    c_f4p tp = AS_CONST(c_f4p, S.triItem);
    const int n = S.n_item * (part + 1) / nparts;
    const int lo = S.n_item * part / nparts;
    auto one = [&](int s) { scan_tri_item<R, NS>(S, P, tri_item(tp[3 * s], tp[3 * s + 1], tp[3 * s + 2])); };
    if (cull) {
      // bound of the item: a triangle - sphere around p0 + (e1 + e2)/3 through the farthest vertex; a parallelogram -
      // sphere around its centre p0 + (ea + eb)/2 through the corners (half the longer diagonal)
      const int lane = (int)(threadIdx.x & 63u);
      for (int s0 = lo; s0 < n; s0 += 64) {
        const int idx = s0 + lane;
        bool pass = false;
        if (idx < n) {
          const float4 p0 = S.triItem[3 * idx], e1 = S.triItem[3 * idx + 1], e2 = S.triItem[3 * idx + 2];
          const float k = (__float_as_int(p0.w) & 1) ? 0.5f : (1.0f / 3.0f);
          const float gx = (e1.x + e2.x) * k, gy = (e1.y + e2.y) * k, gz = (e1.z + e2.z) * k;
          const float r0 = fmaf(gz, gz, fmaf(gy, gy, gx * gx));
          const float r1 = fmaf(e1.z - gz, e1.z - gz, fmaf(e1.y - gy, e1.y - gy, (e1.x - gx) * (e1.x - gx)));
          const float r2 = fmaf(e2.z - gz, e2.z - gz, fmaf(e2.y - gy, e2.y - gy, (e2.x - gx) * (e2.x - gx)));
          pass = bundle_reaches(B, p0.x + gx, p0.y + gy, p0.z + gz, __builtin_amdgcn_sqrtf(fmaxf(r0, fmaxf(r1, r2))) * 1.001f);
        }
        unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
        while (m) {
          const int id = s0 + (int)__builtin_ctzll(m);
          m &= m - 1ull;
          one(id);
        }
      }
    } else {
      for (int s = lo; s < n; ++s) one(s);
    }
  }
  base += S.n_triF;
  if (LISTS & 2) scan_mtris<R, NS>(S, P, base, part, nparts);
}

__device__ __forceinline__ int list_to_gid(const SceneDev& S, int i) {
  if (i < S.n_sphF) return S.sphF_gid[i];
  if ((i -= S.n_sphF) < S.n_msph) return S.msph_gid[i];
  if ((i -= S.n_msph) < S.n_triF) return S.triF_gid[i];
  return S.mtri_gid[i - S.n_triF];
}

// Exact (f64) closest hit among the precise spheres.
__device__ __forceinline__ Closest precise_hit(const SceneDev& S, float ox, float oy, float oz, float dx, float dy,
                                               float dz, uint32_t word) {
  d3 o = d3{(double)ox, (double)oy, (double)oz};
  d3 d = d3{(double)dx, (double)dy, (double)dz};
  Closest r;
  r.t = __builtin_inf();
  r.g = -1;
  for (int k = 0; k < S.n_prec; ++k) {
    // g is wave-uniform: the sphere comes through scalar loads (precise primitives are spheres: g < n_sph)
    const int g = AS_CONST(c_i32p, S.prec_gid)[k];
    c_f64p sp = AS_CONST(c_f64p, S.sphD) + 4 * (size_t)g;
    d3 c = d3{sp[0], sp[1], sp[2]};
    const double rad = sp[3];
    if (S.prec_motion) {
      c_f64p mp = AS_CONST(c_f64p, S.sphMvD) + 3 * (size_t)g;
      const d3 mv = d3{mp[0], mp[1], mp[2]};
      if (mv.x != 0. || mv.y != 0. || mv.z != 0.) {
        const double tm = (double)motion_time(word, (uint32_t)AS_CONST(c_i32p, S.prim_obj)[g]);
        c = c + mv * tm;
      }
    }
    double t;
    if (exact_sphere(o, d, c, rad, &t)) {
      if (t < r.t || (t == r.t && S.prim_obj[g] < S.prim_obj[r.g])) { r.t = t; r.g = g; }
    }
  }
  return r;
}

// Before the scan: slot r of the pack (translated origin, scaled copies, margins, upper bound).
template <int R, int NS>
__device__ __forceinline__ void ray_setup(const SceneDev& S, RayPack<R, NS>& P, int r, float ox, float oy, float oz,
                                          float dx, float dy, float dz, const Closest& pre) {
  PK(P, dx, r) = dx; PK(P, dy, r) = dy; PK(P, dz, r) = dz;
  const float tx = ox - S.bcx, ty = oy - S.bcy, tz = oz - S.bcz;
  PK(P, o4x, r) = 4.0f * tx; PK(P, o4y, r) = 4.0f * ty; PK(P, o4z, r) = 4.0f * tz;
  PK(P, od2, r) = 2.0f * fmaf(tz, dz, fmaf(ty, dy, tx * dx));
  const float oo = fmaf(tz, tz, fmaf(ty, ty, tx * tx));
  // |o'| + |c'| <= dist: bound of every magnitude in the scan (S.brad = largest |c'| incl. motion)
  const float dist = __builtin_amdgcn_sqrtf(oo) * 1.000001f + S.brad;   // 1-ulp sqrt, covered by the factor
  P.negE16[r] = -16.0f * fmaf(dist, S.K1, S.K2);
  // cheap form: |4 disc' - 4 oo - 4 disc*| <= 4 Kc (dist^2 + rmax^2)  =>  pass iff  4 disc' >= 4 (oo - E)
  const float thr = fmaf(-S.Kc, fmaf(dist, dist, S.rmax2), oo);
  PK(P, thr4, r) = 4.0f * (thr - fabsf(thr) * CFT_U4EPS) - 1e-30f;
  const float u = (float)pre.t;              // round to nearest, then push up past the f64 value
  P.U[r] = (pre.g >= 0) ? fmaf(fabsf(u), CFT_U4EPS, u) + 1e-30f : __builtin_inff();
}

// The winner of a ray as the shading needs it.  geom: the primitive is a sphere whose centre (at this scan's motion time)
// and radius are at hand, and mat is its material row - no further table look-up by prim id.
struct Winner {
  double t;    // exact distance, +inf on a miss
  int g;       // global prim id, -1 on a miss
  int obj;     // owning object (valid with geom; tie-breaks)
  int mat;
  bool geom;
  d3 c;
  double rad;
};

// Exact test of LIST entry li (what the f32 scan hands over) for this ray and scan: the same arithmetic as exact_prim
// on the same numbers, fetched by list index in one round of loads (data and record together).
__device__ __forceinline__ bool exact_list_prim(const SceneDev& S, int li, d3 o, d3 d, uint32_t scan_word, double* t_out,
                                                Winner* w) {
  const int4 rec = S.listRec[li];
  w->g = rec.x; w->obj = rec.y; w->mat = rec.z;
  w->geom = false;
  if (li < S.n_sphF) {
    const double* sp = S.lsD + 4 * (size_t)li;
    w->c = ld3(sp); w->rad = sp[3];
    w->geom = true;
    return exact_sphere(o, d, w->c, w->rad, t_out);
  }
  if (li < S.n_sphF + S.n_msph) {
    const double* sp = S.lmD + 8 * (size_t)(li - S.n_sphF);
    d3 c = ld3(sp);
    const d3 mv = ld3(sp + 4);
    const double tm = (double)motion_time(scan_word, (uint32_t)rec.y);
    c = c + mv * tm;
    w->c = c; w->rad = sp[3];
    w->geom = true;
    return exact_sphere(o, d, c, w->rad, t_out);
  }
  return exact_prim(S, rec.x, o, d, scan_word, t_out, &w->g);   // triangles: by prim id (w->g: the triangle hit - a Square::from_points answers as a pair)
}

// After the scan: exact evaluation of the two best candidates, rescan if a third could still win.
// stats[0] += second evaluations, stats[1] += rescans.  pre_obj: object id of the precise-sphere hit `r` (if any).
__device__ __forceinline__ Winner ray_epilogue_w(const SceneDev& S, float ox, float oy, float oz, float dx, float dy,
                                                 float dz, uint32_t word, const Cand& c, Closest pre, uint32_t* stats,
                                                 bool* defer) {
  d3 o = d3{(double)ox, (double)oy, (double)oz};
  d3 d = d3{(double)dx, (double)dy, (double)dz};
  Winner r;
  r.t = pre.t; r.g = pre.g; r.obj = -1; r.mat = 0; r.geom = false;
  r.c = d3{0., 0., 0.}; r.rad = 0.;
  auto obj_of = [&](const Winner& w) { return w.obj >= 0 ? w.obj : S.prim_obj[w.g]; };   // (only on exact ties)
  // Exact ties go to the lower object index and, inside one object (two faces of a Cube, the two triangles of a Square),
  // to the lower prim id - the first minimum of the reference's scans (ray.rs:23-27, object.rs:192,246-253) and the
  // order of exact_scan_all_wave.  A total order: the result does not depend on which of two tied candidates the f32
  // scan happened to rank first (the per-tile lists are filled through atomics, in no fixed order).
  auto wins_tie = [&](const Winner& w, const Winner& cur) {
    const int a = w.obj, b = obj_of(cur);
    return a < b || (a == b && w.g < cur.g);
  };
  if (c.i1 >= 0) {
    Winner w;
    double t;
    if (exact_list_prim(S, c.i1, o, d, word, &t, &w) && (r.g < 0 || t < r.t || (t == r.t && wins_tie(w, r)))) { r = w; r.t = t; }
  }
  const bool second = c.i2 >= 0 && (double)c.lo2 <= r.t;
  if (second) {
    Winner w;
    double t;
    if (exact_list_prim(S, c.i2, o, d, word, &t, &w) && (r.g < 0 || t < r.t || (t == r.t && wins_tie(w, r)))) { r = w; r.t = t; }
  }
  const bool rescan = c.lo3 < __builtin_inff() && (double)c.lo3 <= r.t;   // lo3 = +inf unless a third candidate exists
  // the caller runs the rescan where the whole wave is convergent (lanes without a ray help too)
  *defer = rescan;
  if (stats) {
    unsigned long long m2 = __ballot(second), m3 = __ballot(rescan);
    if (m2 | m3) {
      const uint32_t leader = (uint32_t)__ffsll((long long)__ballot(1)) - 1u;   // first active lane
      if ((threadIdx.x & 63u) == leader) {
        if (m2) atomicAdd(&stats[0], (uint32_t)__popcll(m2));
        if (m3) atomicAdd(&stats[1], (uint32_t)__popcll(m3));
      }
    }
  }
  return r;
}

// The one-ray form's epilogue (parity hook, shadow rays, k_trace_finish): the winner as (t, prim id); the rescan runs
// here, among whatever lanes are active.
__device__ __forceinline__ Closest ray_epilogue(const SceneDev& S, float ox, float oy, float oz, float dx, float dy,
                                                float dz, uint32_t word, const Cand& c, Closest pre, uint32_t* stats) {
  bool rescan;
  const Winner w = ray_epilogue_w(S, ox, oy, oz, dx, dy, dz, word, c, pre, stats, &rescan);
  Closest r;
  r.t = w.t; r.g = w.g;
  const d3 o = d3{(double)ox, (double)oy, (double)oz};
  const d3 d = d3{(double)dx, (double)dy, (double)dz};
  exact_scan_all_wave(S, rescan, o, d, word, &r.t, &r.g);
  return r;
}

// Ray::hit for one ray (used by the parity hook, the shadow kernel and k_trace_finish).  LISTS as in scan_lists: 7 = every
// path (cone cull aside: one ray is no bundle), chosen at run time; with 8 the caller guarantees a grid and the ray walks it.
template <int LISTS = 7>
__device__ __forceinline__ Closest closest_hit(const SceneDev& S, float ox, float oy, float oz, float dx,
                                               float dy, float dz, uint32_t scan_word, uint32_t* stats,
                                               const LdsGrid lg = CFT_NO_LDS_GRID) {
  RayPack<1, 1> P;
  P.word[0][0] = scan_word;
  cand_init(P.c[0][0]);
  Closest pre = precise_hit(S, ox, oy, oz, dx, dy, dz, scan_word);
  ray_setup<1, 1>(S, P, 0, ox, oy, oz, dx, dy, dz, pre);
  pack_dummy_slot<1, 1>(P, 1);
  scan_lists<1, 1, false, LISTS>(S, P, 0, 1, -1, nullptr, lg);
  return ray_epilogue(S, ox, oy, oz, dx, dy, dz, scan_word, P.c[0][0], pre, stats);
}

// ------------------------------------------------------------------ generate
struct CamDev {   // viewport constants of Camera::emit_rays (ray.rs:112-118), computed in f64 on the host
  float tlx, tly, tlz;      // top_left
  float pwx, pwy, pwz;      // pw = vw/width * right
  float phx, phy, phz;      // ph = vh/height * up
  float px, py, pz;         // pos
  float rx, ry, rz;         // right
  float ux, uy, uz;         // up
  float lens_radius;        // aperture / 2
  uint32_t width, height;
};

// pix = pid mod (width * height) when the caller knows it (k_generate's tiled order), 0xFFFFFFFF otherwise; (pw_, ph_) =
// the pixel's column and row when the caller knows them too (no division then).
__device__ __forceinline__ void primary_ray(const CamDev& C, uint64_t pid, uint32_t k0, uint32_t k1, int jitter,
                                            f3* pos, f3* dir, uint32_t pix = 0xFFFFFFFFu, uint32_t pw_ = 0xFFFFFFFFu,
                                            uint32_t ph_ = 0u) {
  uint32_t w = pw_, h = ph_;
  if (pw_ == 0xFFFFFFFFu) {
    if (pix == 0xFFFFFFFFu) pix = (uint32_t)(pid % ((uint64_t)C.width * C.height));   // (a 64-bit division)
    w = pix % C.width; h = pix / C.width;   // image index h*WIDTH + w (scene.rs:97)
  }
  float urw = 0.5f, urh = 0.5f, uth = 0.0f, ur = 0.0f;
  if (jitter) {
    u4 r = philox4x32_10((uint32_t)pid, (uint32_t)(pid >> 32), CFT_LEVEL_CAMERA, CFT_PURPOSE_CAMERA, k0, k1);
    urw = u24f(r.x); urh = u24f(r.y); uth = u24f(r.z); ur = u24f(r.w);   // draw order rw, rh, theta, r
  }
  float fw = (float)w + (urw - 0.5f);
  float fh = (float)h + (urh - 0.5f);
  f3 to = mk3(fmaf(-fh, C.phx, fmaf(fw, C.pwx, C.tlx)), fmaf(-fh, C.phy, fmaf(fw, C.pwy, C.tly)),
              fmaf(-fh, C.phz, fmaf(fw, C.pwz, C.tlz)));
  float rdx, rdy;
  gen_point_in_disk(C.lens_radius, uth, ur, &rdx, &rdy);
  f3 from = mk3(fmaf(C.ux, rdy, fmaf(C.rx, rdx, C.px)), fmaf(C.uy, rdy, fmaf(C.ry, rdx, C.py)),
                fmaf(C.uz, rdy, fmaf(C.rz, rdx, C.pz)));
  *pos = from;
  *dir = unit3(to - from);   // Ray::new normalises (ray.rs:30-35)
}

struct GenArgs {   // level-0 launches that generate their own rays (k_trace_level<.., GEN>)
  CamDev cam;
  int jitter, tiled;
};

// one lane = one (pixel, sample).  Queue slot i holds local path id lp (global pid = pid_base + lp).  With
// `tiled` (whole samples of an image whose width is a multiple of 32 and height of 8) the slots of a sample run
// through 32x8-pixel tiles, so that the 256 (or 128) consecutive slots a wave of the level kernels takes are a
// compact pixel block: their primary rays form a narrow bundle, which the wave-level cull (wave_bundle) uses.
// Otherwise lp = i (row-major).  The order of the queue never affects the image: radiance is stored by lp.
// th (optional, tiled frames): the sample and the tile the slot belongs to, worked out once per trip by the caller (a
// trip's slots are pixels of one tile and one sample); without it they come out of i by three divisions per lane.
struct TileHint { uint32_t smp, tx, ty, slot0; };   // slot0 = the queue slot of the tile's first pixel in that sample; 0xFFFFFFFF: no hint
#define CFT_NO_TILE_HINT TileHint{0u, 0u, 0u, 0xFFFFFFFFu}
__device__ __forceinline__ void generate_slot(const CamDev& C, uint64_t pid_base, uint32_t i, uint32_t k0, uint32_t k1,
                                              int jitter, int tiled, const QueueDev& q, float4* a_out = nullptr,
                                              float4* b_out = nullptr, uint32_t* lp_out = nullptr, TileHint th = CFT_NO_TILE_HINT,
                                              bool lean = false) {
  uint32_t lp = i, pix = 0xFFFFFFFFu, pxw = 0xFFFFFFFFu, pxh = 0u;
  if (tiled && th.slot0 != 0xFFFFFFFFu) {
    const uint32_t wi = i - th.slot0;   // 0 .. 255 inside the tile
    pxw = th.tx * 32u + (wi & 31u);
    pxh = th.ty * 8u + (wi >> 5);
    pix = pxh * C.width + pxw;
    lp = th.smp * (C.width * C.height) + pix;
  } else if (tiled) {   // (pid_base is a multiple of npix here, so the pixel of path pid_base + lp is lp mod npix)
    const uint32_t npix = C.width * C.height;
    const uint32_t smp = i / npix, j = i - smp * npix;
    const uint32_t t = j >> 8, wi = j & 255u, ntx = C.width >> 5;
    const uint32_t ty = t / ntx, tx = t - ty * ntx;
    pix = (ty * 8u + (wi >> 5)) * C.width + tx * 32u + (wi & 31u);
    lp = smp * npix + pix;
  }
  f3 pos, dir;
  primary_ray(C, pid_base + lp, k0, k1, jitter, &pos, &dir, pix, pxw, pxh);
  const float4 a = make_float4(pos.x, pos.y, pos.z, dir.x), b = make_float4(dir.y, dir.z, 1.0f, 1.0f);
  q.a[i] = a;
  if (lean) {
    // (the fused level-0 launch, which reads the ray back itself: throughput 1 and the path id follow from the level
    // and the slot, so 24 of the 40 bytes are written and read back - that launch runs at the HBM's speed)
    // (dy, dz go into the float2 stream: dense 8-byte stores, where half-filled 16-byte slots would cost whole sectors)
    q.c[i] = make_float2(b.x, b.y);
  } else {
    q.b[i] = b;
    q.c[i] = make_float2(1.0f, __uint_as_float(lp));
  }
  if (a_out) { *a_out = a; *b_out = b; *lp_out = lp; }
}

// Alone: the split pipeline and the parity hook.  The default pipeline generates inside the level-0 launch
// (k_trace_level<.., GEN = true>).
__global__ __launch_bounds__(CFT_BLOCK) void k_generate(CamDev C, uint64_t pid_base, uint32_t n, uint32_t k0,
                                                        uint32_t k1, int jitter, int tiled, QueueDev q) {
  for (uint32_t i = blockIdx.x * CFT_BLOCK + threadIdx.x; i < n; i += gridDim.x * CFT_BLOCK)
    generate_slot(C, pid_base, i, k0, k1, jitter, tiled, q);
}

// ------------------------------------------------------------------ per-tile lists for the primary rays
// Which 32x8-pixel tiles can a primitive's bounding sphere (c, r) be hit from?  A primary ray runs from a lens point o,
// |o - pos| <= a, through a viewport point F = tl + fw pw - fh ph (primary_ray above).  With the pinhole ray
// P(s) = pos + s (F - pos), the real ray is X(s) = P(s) + (1 - s)(o - pos): within a |1 - s| of the pinhole ray at the
// same s, and s = depth / fd (depth along the viewport normal g, fd = (tl - pos).g; the lens lies in the plane
// through pos spanned by right and up, so o has depth 0): the blur vanishes in the plane of focus and grows with the
// distance from it.  X can touch the sphere only at depths within r of z, the depth of c, where |1 - s| is at most
// (|z - fd| + r) / fd, so the pinhole ray passes within
//     rho = r + a (|z - fd| + r) / fd
// of c.  The pinhole rays of tile column tx lie between the planes through pos that contain the viewport lines
// fw = 32 tx - 0.5 and fw = 32 (tx + 1) - 0.5 (pixel jitter is +-0.5); likewise the rows.  A sphere wholly in front of the
// camera meets that wedge only if it is not wholly on the far side of either plane: two signed plane distances
// against rho per column and row.  Spheres that reach behind the lens plane go to every tile; spheres wholly behind it
// to none (rays only go forward).  All in f64 from the f32 camera constants the generation uses, with slack (1e-4 rho,
// 1e-5 of the distances involved, 0.01 + rounding pixels on the boundaries) far above the f32 rounding of the rays.
struct TileRange { uint16_t tx0, tx1, ty0, ty1; };   // inclusive; tx0 > tx1: no tile

// One WAVE per primitive: the lanes share out the column and row boundaries (two plane tests each, no square root:
// d >= -rho  <=>  N.v >= 0 or (N.v)^2 <= rho^2 |N|^2), ballots give the first and last column / row, then the lanes
// share out the tiles of the range.  pass 0: ranges and per-tile counts; pass 1 (after the prefix sum): the lists.
__global__ __launch_bounds__(CFT_BLOCK) void k_tile_bin(CamDev C, const float4* boundF, const float4* triF, uint32_t n_sm, uint32_t n, float bcx, float bcy,
                                                        float bcz, uint32_t ntx, uint32_t nty, TileRange* ranges,
                                                        uint32_t* cnt, const uint32_t* off, uint32_t* idx, int pass) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * CFT_BLOCK + threadIdx.x) >> 6, nwaves = (gridDim.x * CFT_BLOCK) >> 6;
  const d3 pos = d3{(double)C.px, (double)C.py, (double)C.pz};
  const d3 A = d3{(double)C.tlx, (double)C.tly, (double)C.tlz} - pos;
  const d3 pw = d3{(double)C.pwx, (double)C.pwy, (double)C.pwz}, ph = d3{(double)C.phx, (double)C.phy, (double)C.phz};
  d3 g = dunit(dcross(pw, ph));
  if (ddot(A, g) < 0.) g = d3{-g.x, -g.y, -g.z};
  const double fd = ddot(A, g);
  // |rdx right + rdy up| <= a * (largest singular value of [right up]) for (rdx, rdy) in the disk of radius a
  const double rr = (double)C.rx * C.rx + (double)C.ry * C.ry + (double)C.rz * C.rz;
  const double uu = (double)C.ux * C.ux + (double)C.uy * C.uy + (double)C.uz * C.uz;
  const double ru = (double)C.rx * C.ux + (double)C.ry * C.uy + (double)C.rz * C.uz;
  const double tr = rr + uu, det = rr * uu - ru * ru;
  const double a = (double)C.lens_radius * sqrt(0.5 * (tr + sqrt(fmax(tr * tr - 4. * det, 0.)))) * 1.0001;
  const double lpos = sqrt(ddot(pos, pos));
  const double lpw = sqrt(ddot(pw, pw)), lph = sqrt(ddot(ph, ph));
  const double ltl = sqrt((double)C.tlx * C.tlx + (double)C.tly * C.tly + (double)C.tlz * C.tlz);
  const double mp = 0.01 + 32. * 5.97e-8 * (ltl + lpos + C.width * lpw + C.height * lph) / fmin(lpw, lph);
  for (uint32_t i = wave; i < n; i += nwaves) {
    TileRange t{1, 0, 1, 0};
    if (pass == 0) {
      const float4 b = boundF[i];
      const d3 v = d3{(double)b.x + (double)bcx, (double)b.y + (double)bcy, (double)b.z + (double)bcz} - pos;
      const double z = ddot(v, g), lv = sqrt(ddot(v, v));
      // A sphere is one point widened by rho; a TRIANGLE is its three corners widened by the blur alone: a ray that
      // touches it does so at a point X of it, the pinhole ray passes within a |z_X - fd| / fd of X, and that is
      // convex in X, so at most the largest of the corners' values - the pinhole ray meets the convex hull of three
      // balls of that radius, which lies beyond a plane when all three do.  (The bounding sphere of a triangle covers
      // three to four times the tiles the triangle does.)
      d3 vv[3] = {v, v, v};
      double zlo = z, zhi = z, lvv = lv;
      double rho = (double)b.w + a * (fabs(z - fd) + (double)b.w) / fd;   // (the host only bins when a < fd / 8)
      const int nv = i >= n_sm ? 3 : 1;
      if (nv == 3) {
        const float4 p0 = triF[3 * (size_t)(i - n_sm)], e1 = triF[3 * (size_t)(i - n_sm) + 1], e2 = triF[3 * (size_t)(i - n_sm) + 2];
        const d3 q0 = d3{(double)p0.x + (double)bcx, (double)p0.y + (double)bcy, (double)p0.z + (double)bcz} - pos;
        vv[0] = q0;
        vv[1] = q0 + d3{(double)e1.x, (double)e1.y, (double)e1.z};
        vv[2] = q0 + d3{(double)e2.x, (double)e2.y, (double)e2.z};
        rho = 0.; zlo = 1e300; zhi = -1e300; lvv = 0.;
        for (int j = 0; j < 3; ++j) {
          const double zj = ddot(vv[j], g);
          zlo = fmin(zlo, zj); zhi = fmax(zhi, zj);
          rho = fmax(rho, a * fabs(zj - fd) / fd);
          lvv = fmax(lvv, sqrt(ddot(vv[j], vv[j])));
        }
      }
      rho = rho * 1.0001 + 1e-5 * (1. + lvv + lpos);
      const double rho2 = rho * rho;
      if (zhi + rho <= 0.) {
        // wholly behind the lens plane: no tile
      } else if (zlo - rho <= 1e-6 * fd) {
        t = TileRange{0, (uint16_t)(ntx - 1), 0, (uint16_t)(nty - 1)};
      } else {
        // not_below(N): some corner is not farther than rho on the negative side of the plane with normal N through pos
        auto not_below = [&](d3 N) {
          const double nn = rho2 * ddot(N, N);
          bool ok = false;
          for (int j = 0; j < nv; ++j) { const double d = ddot(N, vv[j]); ok = ok || d >= 0. || d * d <= nn; }
          return ok;
        };
        uint32_t x0 = 0xFFFFFFFFu, x1 = 0u, y0 = 0xFFFFFFFFu, y1 = 0u;
        bool anyx = false, anyy = false;
        for (uint32_t k0 = 0; k0 < ntx; k0 += 64u) {
          const uint32_t tx = k0 + lane;
          bool ok = false;
          if (tx < ntx) {
            // column tx lies between the planes at fw = 32 tx - 0.5 and 32 (tx + 1) - 0.5 (+- the margin); a plane's
            // normal (A + fw pw) x ph is turned towards growing fw
            d3 NL = dcross(A + pw * (32. * tx - 0.5 - mp), ph), NR = dcross(A + pw * (32. * (tx + 1) - 0.5 + mp), ph);
            if (ddot(NL, pw) < 0.) NL = d3{-NL.x, -NL.y, -NL.z};
            if (ddot(NR, pw) > 0.) NR = d3{-NR.x, -NR.y, -NR.z};   // (turned towards falling fw: "not right of it")
            ok = not_below(NL) && not_below(NR);
          }
          const unsigned long long m = __ballot(ok);
          if (m) {
            if (!anyx) x0 = k0 + (uint32_t)__builtin_ctzll(m);
            x1 = k0 + 63u - (uint32_t)__builtin_clzll(m);
            anyx = true;
          }
        }
        for (uint32_t k0 = 0; k0 < nty && anyx; k0 += 64u) {
          const uint32_t ty = k0 + lane;
          bool ok = false;
          if (ty < nty) {
            d3 NT = dcross(A - ph * (8. * ty - 0.5 - mp), pw), NB = dcross(A - ph * (8. * (ty + 1) - 0.5 + mp), pw);
            if (ddot(NT, ph) > 0.) NT = d3{-NT.x, -NT.y, -NT.z};   // towards growing fh = falling ph
            if (ddot(NB, ph) < 0.) NB = d3{-NB.x, -NB.y, -NB.z};
            ok = not_below(NT) && not_below(NB);
          }
          const unsigned long long m = __ballot(ok);
          if (m) {
            if (!anyy) y0 = k0 + (uint32_t)__builtin_ctzll(m);
            y1 = k0 + 63u - (uint32_t)__builtin_clzll(m);
            anyy = true;
          }
        }
        if (anyx && anyy) t = TileRange{(uint16_t)x0, (uint16_t)x1, (uint16_t)y0, (uint16_t)y1};
      }
      if (lane == 0) ranges[i] = t;
    } else {
      t = ranges[i];
    }
    if (t.tx0 > t.tx1) continue;
    const uint32_t w = (uint32_t)t.tx1 - t.tx0 + 1u, h = (uint32_t)t.ty1 - t.ty0 + 1u;
    for (uint32_t j = lane; j < w * h; j += 64u) {
      const uint32_t tile = (t.ty0 + j / w) * ntx + t.tx0 + j % w;
      const uint32_t k = atomicAdd(&cnt[tile], 1u);
      if (pass) idx[off[tile] + k] = i;
    }
  }
}

// Exclusive prefix sum of the tile counts (one block; a frame has ~8 k tiles): off[0 .. n], off[n] = total; clears cnt
// for the second pass.
__global__ __launch_bounds__(CFT_BLOCK) void k_tile_prefix(uint32_t* cnt, uint32_t n, uint32_t* off) {
  __shared__ uint32_t part[CFT_BLOCK];
  const uint32_t per = (n + CFT_BLOCK - 1) / CFT_BLOCK;
  const uint32_t b = threadIdx.x * per, e = min(b + per, n);
  uint32_t sum = 0;
  for (uint32_t i = b; i < e; ++i) sum += cnt[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t t = 0; t < threadIdx.x; ++t) base += part[t];
  for (uint32_t i = b; i < e; ++i) {
    off[i] = base;
    base += cnt[i];
    cnt[i] = 0u;
  }
  if (threadIdx.x == CFT_BLOCK - 1) off[n] = base;
}

// ------------------------------------------------------------------ intersect
// Unit normal of prim g at hit point p for the hit record: differences in f64, normalised in f32
// (k_shade re-normalises, as HitInfo::new does).
__device__ __forceinline__ f3 hit_normal(const SceneDev& S, int g, d3 p, uint32_t word) {
  d3 n;
  if (g < S.n_sph) {
    const double* sp = S.sphD + 4 * (size_t)g;
    d3 c = ld3(sp);
    if (S.any_motion) {
      d3 mv = ld3(S.sphMvD + 3 * (size_t)g);
      if (mv.x != 0. || mv.y != 0. || mv.z != 0.) c = c + mv * (double)motion_time(word, (uint32_t)S.prim_obj[g]);
    }
    n = p - c;
    if (sp[3] < 0.) n = d3{-n.x, -n.y, -n.z};
  } else {
    const double* tp = S.triD + 9 * (size_t)(g - S.n_sph);
    d3 p0 = ld3(tp), p1 = ld3(tp + 3), p2 = ld3(tp + 6);
    n = dcross(p1 - p0, p2 - p0);
  }
  f3 nf = mk3((float)n.x, (float)n.y, (float)n.z);
  float s = fmaxf(fmaxf(fabsf(nf.x), fabsf(nf.y)), fabsf(nf.z));   // keep |n|^2 inside the f32 range
  nf = nf * __builtin_amdgcn_rcpf(s);                              // 1-ulp rcp / rsq: k_shade renormalises anyway
  return nf * __builtin_amdgcn_rsqf(dot3(nf, nf));
}

// ---- pieces shared by k_intersect and the fused k_trace_level
// LDS "park": what a ray needs only AFTER the scan lives in the lane's own LDS column (no barrier) instead of
// in registers, and the per-ray work after the scan runs as a rolled loop over the R slots that indexes the
// park dynamically - one copy of the f64 / shading code and its registers, not R.  Words of slot r:
//   [0..2] precise-sphere hit (t lo, t hi, prim)   [3 + 5k ..] candidate set of scan k   [3 + 5 NS + k] motion word k
// The fused kernel later reuses a slot's words for the surviving ray (9 words).

// Before the scan: load the lane's R rays of this level (ray i, i + stride, ...), draw the motion words, test the
// precise spheres, fill the pack.
// perm (optional): the queue index of each (slot, thread) when the wave has reordered its rays (wave_sort_rays).
// gen (level 0 of the default pipeline): the ray of a slot is generated here (and written to the queue) instead of loaded.
template <int R, int NS>
__device__ __forceinline__ void isect_setup(const SceneDev& S, const QueueDev& q, uint32_t n, uint32_t base,
                                            uint32_t stride, uint32_t level, uint64_t pid_base, uint32_t k0,
                                            uint32_t k1, uint32_t role0, uint32_t role1, RayPack<R, NS>& P,
                                            float (*park)[CFT_BLOCK], const uint16_t (*perm)[CFT_BLOCK] = nullptr,
                                            const GenArgs* gen = nullptr, TileHint th = CFT_NO_TILE_HINT, bool lean = false) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    // (with perm, base is the wave's first ray)
    const uint32_t i = perm ? (perm[r][CFT_TID] == 0xFFFFu ? 0xFFFFFFFFu : base + perm[r][CFT_TID]) : base + r * stride;
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 1.f;
    Closest pre;
    pre.t = __builtin_inf(); pre.g = -1;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      P.word[k][r] = 0;
      cand_init(P.c[k][r]);
    }
    if (i < n) {
      float4 a, b;
      uint32_t lp = 0;
      if (gen) generate_slot(gen->cam, pid_base, i, k0, k1, gen->jitter, gen->tiled, q, &a, &b, &lp, th, lean);
      else { a = q.a[i]; b = q.b[i]; }
      ox = a.x; oy = a.y; oz = a.z; dx = a.w; dy = b.x; dz = b.y;
      if (S.any_motion) {
        uint64_t pid = pid_base + (gen ? lp : __float_as_uint(q.c[i].y));
#pragma unroll
        for (int k = 0; k < NS; ++k)
          P.word[k][r] = philox4x32_10((uint32_t)pid, (uint32_t)(pid >> 32), level,
                                       CFT_PURPOSE_SCAN0 + (k == 0 ? role0 : role1), k0, k1).x;
      }
      pre = precise_hit(S, ox, oy, oz, dx, dy, dz, P.word[0][r]);
    }
    PARK(r, 0) = __int_as_float(__double2loint(pre.t));
    PARK(r, 1) = __int_as_float(__double2hiint(pre.t));
    PARK(r, 2) = __int_as_float(pre.g);
    ray_setup<R, NS>(S, P, r, ox, oy, oz, dx, dy, dz, pre);
    if (!(i < n)) { PK(P, thr4, r) = __builtin_inff(); P.negE16[r] = __builtin_inff(); }   // a dummy slot never is a candidate
    if (NS == 2 && S.prec_motion) P.U[r] = __builtin_inff();   // a moving precise sphere: no shared upper bound
  }
  if (R & 1) pack_dummy_slot<R, NS>(P, R);
}

// Right after the scan: the candidate sets and motion words join the park; the pack is dead from here on.
template <int R, int NS>
__device__ __forceinline__ void isect_park(const RayPack<R, NS>& P, float (*park)[CFT_BLOCK]) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const Cand& c = P.c[k][r];
      PARK(r, ParkLayout<NS>::CAND + 5 * k + 0) = c.lo1;
      PARK(r, ParkLayout<NS>::CAND + 5 * k + 1) = c.lo2;
      PARK(r, ParkLayout<NS>::CAND + 5 * k + 2) = c.lo3;
      PARK(r, ParkLayout<NS>::CAND + 5 * k + 3) = __int_as_float(c.i1);
      PARK(r, ParkLayout<NS>::CAND + 5 * k + 4) = __int_as_float(c.i2);
      PARK(r, ParkLayout<NS>::WORD + k) = __uint_as_float(P.word[k][r]);
    }
  }
}

// The hit record of slot r (any r: the park is indexed dynamically), scan k, for the ray (o, d) re-read from the
// queue: (t, prim, px, py), (pz, nx, ny, nz); prim = -1 on a miss.
// Called by every lane of the wave together (live = this lane has a ray in slot r): the rare f64 rescan in the middle
// is done by the whole wave for the lanes that need it.
template <int NS>
__device__ __forceinline__ void isect_hit(const SceneDev& S, bool live, int r, int k, float ox, float oy, float oz,
                                          float dx, float dy, float dz, float (*park)[CFT_BLOCK], uint32_t* stats,
                                          float4* h0, float4* h1, double* t64, int* mat_out) {
  *h0 = make_float4(__builtin_inff(), __int_as_float(-1), 0.f, 0.f);
  *h1 = make_float4(0.f, 0.f, 0.f, 0.f);
  *t64 = __builtin_inf();
  *mat_out = 0;
  uint32_t word = 0;
  Winner w;
  w.t = __builtin_inf(); w.g = -1; w.obj = -1; w.mat = 0; w.geom = false;
  w.c = d3{0., 0., 0.}; w.rad = 0.;
  bool rescan = false;
  if (live) {
    word = __float_as_uint(PARK(r, ParkLayout<NS>::WORD + k));
    Closest pre;
    if (k == 0 || !S.prec_motion) {
      pre.t = __hiloint2double(__float_as_int(PARK(r, 1)), __float_as_int(PARK(r, 0)));
      pre.g = __float_as_int(PARK(r, 2));
    } else {
      pre = precise_hit(S, ox, oy, oz, dx, dy, dz, word);
    }
    Cand cd;
    cd.lo1 = PARK(r, ParkLayout<NS>::CAND + 5 * k + 0);
    cd.lo2 = PARK(r, ParkLayout<NS>::CAND + 5 * k + 1);
    cd.lo3 = PARK(r, ParkLayout<NS>::CAND + 5 * k + 2);
    cd.i1 = __float_as_int(PARK(r, ParkLayout<NS>::CAND + 5 * k + 3));
    cd.i2 = __float_as_int(PARK(r, ParkLayout<NS>::CAND + 5 * k + 4));
    w = ray_epilogue_w(S, ox, oy, oz, dx, dy, dz, word, cd, pre, stats, &rescan);
  }
  const d3 o = d3{(double)ox, (double)oy, (double)oz};
  const d3 d = d3{(double)dx, (double)dy, (double)dz};
  if (__ballot(rescan)) {
    Closest c;
    c.t = w.t; c.g = w.g;
    exact_scan_all_wave_call(S, rescan, o, d, word, &c.t, &c.g);
    if (rescan) { w.t = c.t; w.g = c.g; w.geom = false; }
  }
  if (live && w.g >= 0) {
    if (!w.geom && w.g == S.prec_g0) {   // the (static) first precise sphere: its data sit in the kernel arguments
      w.c = d3{S.prec_c0x, S.prec_c0y, S.prec_c0z};
      w.rad = S.prec_r0; w.mat = S.prec_mat0;
      w.geom = true;
    }
    const d3 p = o + d * w.t;                  // point = pos + dir * t
    f3 nn;
    if (w.geom) {                              // hit_normal's sphere branch on the data at hand
      d3 n = p - w.c;
      if (w.rad < 0.) n = d3{-n.x, -n.y, -n.z};
      nn = mk3((float)n.x, (float)n.y, (float)n.z);
      const float sc = fmaxf(fmaxf(fabsf(nn.x), fabsf(nn.y)), fabsf(nn.z));
      nn = nn * __builtin_amdgcn_rcpf(sc);
      nn = nn * __builtin_amdgcn_rsqf(dot3(nn, nn));
      *mat_out = w.mat;
    } else {
      nn = hit_normal(S, w.g, p, word);
      *mat_out = S.prim_mat[w.g];
    }
    *h0 = make_float4((float)w.t, __int_as_float(w.g), (float)p.x, (float)p.y);
    *h1 = make_float4((float)p.z, nn.x, nn.y, nn.z);
    *t64 = w.t;
  }
}

// one lane = R live rays of trace level `level` (ray i, i + stride, ...); primitive lists are
// wave-uniform SGPR operands.  NS = 1: one scan with motion key `role0`.  NS = 2 (scenes with moving
// objects AND lights): the lights' `looked` scan (role0 -> out0) and trace's own (role1 -> out1) in one pass.
// This is the intersect stage of the SPLIT pipeline (k_intersect -> hit records -> k_shade); the default
// pipeline runs the same code fused with the shading (k_trace_level below).
template <int R, int NS>
__global__ __launch_bounds__(CFT_BLOCK) void k_intersect(SceneDev S, QueueDev q, const uint32_t* counts,
                                                         uint32_t level, uint64_t pid_base, uint32_t k0,
                                                         uint32_t k1, uint32_t role0, uint32_t role1, HitDev out0,
                                                         HitDev out1, uint32_t* stats) {
  __shared__ float park[ParkLayout<NS>::WORDS * R][CFT_BLOCK];
  const uint32_t n = counts[level];
  // a wave takes 64 R consecutive queue slots (slot r of lane l = wave base + 64 r + l); the trip count is
  // wave-uniform so that every lane stays active for the wave-level cull
  const uint32_t stride = 64u;
  const uint32_t wstep = gridDim.x * CFT_BLOCK * R;
  for (uint32_t wb = (blockIdx.x * CFT_BLOCK + (threadIdx.x & ~63u)) * R; wb < n; wb += wstep) {
    const uint32_t base = wb + (threadIdx.x & 63u);
    {
      RayPack<R, NS> P;
      isect_setup<R, NS>(S, q, n, base, stride, level, pid_base, k0, k1, role0, role1, P, park);
      scan_lists<R, NS, true>(S, P);
      isect_park<R, NS>(P, park);
    }
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
      if (!(wb + r * stride < n)) break;   // wave-uniform: the whole wave stays together for isect_hit
      const uint32_t i = base + r * stride;
      const bool live = i < n;
      const float4 a = q.a[live ? i : wb];
      const float4 b = q.b[live ? i : wb];
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        float4 h0, h1;
        double t64;
        int mat;
        isect_hit<NS>(S, live, r, k, a.x, a.y, a.z, a.w, b.x, b.y, park, stats, &h0, &h1, &t64, &mat);
        if (live) {
          if (k == 0) { out0.h0[i] = h0; out0.h1[i] = h1; out0.t[i] = t64; }
          else        { out1.h0[i] = h0; out1.h1[i] = h1; out1.t[i] = t64; }
        }
      }
    }
  }
}

// Parity hook: Ray::hit for caller rays; writes the f64 distance and the normal after HitInfo::new's
// flip (ray.rs:233-241), all decided in f64.
__global__ __launch_bounds__(CFT_BLOCK) void k_intersect_hook(SceneDev S, const float* rays, uint32_t n,
                                                              uint64_t pid0, uint32_t level, uint32_t role,
                                                              uint32_t k0, uint32_t k1, int32_t* out_id,
                                                              double* out_t, float* out_n, float* out_p,
                                                              int32_t* out_outward, uint32_t* stats) {
  // (whole waves run the loop together: the f64 scan below is done by the wave)
  for (uint32_t i0 = blockIdx.x * CFT_BLOCK + (threadIdx.x & ~63u); i0 < n; i0 += gridDim.x * CFT_BLOCK) {
    const uint32_t i = i0 + (threadIdx.x & 63u);
    const bool live = i < n;
    const float* r = rays + 6 * (size_t)(live ? i : i0);
    float ox = r[0], oy = r[1], oz = r[2], dx = r[3], dy = r[4], dz = r[5];
    uint64_t pid = pid0 + i;
    uint32_t scan_word =
        philox4x32_10((uint32_t)pid, (uint32_t)(pid >> 32), level, CFT_PURPOSE_SCAN0 + role, k0, k1).x;
    // The f32 cull and the grid walk assume |d| = 1 (the render path's rays are unit to ~1e-7: Ray::new normalises,
    // reflect / refract keep the length).  The hook takes the direction as given, like a struct-literal Ray
    // (ray.rs:277-289): a ray whose direction is not unit goes straight to the exact f64 scan, whose sphere formula
    // carries a = |d|^2 as the reference's does (object.rs:273-286).
    const float l2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const bool unit_dir = fabsf(l2 - 1.0f) <= 1e-5f;
    Closest c;
    c.t = __builtin_inf(); c.g = -1;
    if (live && unit_dir) c = closest_hit(S, ox, oy, oz, dx, dy, dz, scan_word, stats);
    const d3 o = d3{(double)ox, (double)oy, (double)oz};
    const d3 d = d3{(double)dx, (double)dy, (double)dz};
    exact_scan_all_wave(S, live && !unit_dir, o, d, scan_word, &c.t, &c.g);
    if (!live) continue;
    d3 nn = d3{0., 0., 0.}, p = d3{0., 0., 0.};
    int outward = 0;
    if (c.g >= 0) {
      p = o + d * c.t;
      nn = dunit(exact_normal(S, c.g, p, scan_word));
      d3 din = dunit(d);
      if (ddot(nn, din) > -1e-3) { nn = d3{-nn.x, -nn.y, -nn.z}; outward = 1; }   // ray.rs:237-241
    }
    if (out_outward) out_outward[i] = outward;
    out_id[i] = c.g >= 0 ? S.prim_obj[c.g] : -1;
    out_t[i] = c.t;
    out_n[3 * (size_t)i] = (float)nn.x; out_n[3 * (size_t)i + 1] = (float)nn.y; out_n[3 * (size_t)i + 2] = (float)nn.z;
    out_p[3 * (size_t)i] = (float)p.x; out_p[3 * (size_t)i + 1] = (float)p.y; out_p[3 * (size_t)i + 2] = (float)p.z;
  }
}

// ------------------------------------------------------------------ shade
struct HitInfoF {   // ray.rs:222-251 in f32
  f3 n, p, din, dout;
  bool outward;
};
__device__ __forceinline__ HitInfoF hitinfo_new(f3 norm, f3 p, f3 dir_in) {
  HitInfoF h;
  f3 n = unit3(norm);
  f3 din = unit3(dir_in);
  h.outward = dot3(n, din) > -CFT_EPS;
  if (h.outward) n = -n;
  f3 nu = unit3(n);                           // proj_to re-normalises its argument (util.rs:60-63)
  f3 proj = nu * dot3(din, nu);
  h.dout = unit3(din - proj * 2.0f);
  h.n = n; h.p = p; h.din = din;
  return h;
}
__device__ __forceinline__ f3 hi_pos(const HitInfoF& h) { return madd3(h.p, h.dout, CFT_EPS); }  // ray.rs:269-271
__device__ __forceinline__ bool hi_refract(const HitInfoF& h, float ratio, f3* pos, f3* dir) {   // ray.rs:293-308
  float c = dot3(h.din, h.n);
  float disc = 1.0f - (ratio * ratio) * (1.0f - c * c);
  if (disc > 0.0f) {
    f3 dd = (h.din - h.n * c) * ratio - h.n * sqrtf(disc);
    *pos = madd3(h.p, dd, CFT_EPS);
    *dir = dd;
    return true;
  }
  return false;
}
__device__ __forceinline__ float hi_reflect_prob(const HitInfoF& h, float ior) {  // ray.rs:312-316
  float r0 = (1.0f - ior) / ((1.0f + ior) * (1.0f + ior));   // NOT squared, as in the reference
  float m = 1.0f - fabsf(dot3(h.din, h.n));
  const float m2 = m * m;
  return r0 + (1.0f - r0) * (m * (m2 * m2));   // powi(5) = x * (x^2)^2
}

// Static hit of a LightShape's own shape (Shape::hit_info, no motion), f64: light shapes are few.
__device__ __forceinline__ bool light_shape_hit(const SceneDev& S, const LightDev& L, f3 pos, f3 dir, double* t_out) {
  d3 o = d3{(double)pos.x, (double)pos.y, (double)pos.z};
  d3 d = d3{(double)dir.x, (double)dir.y, (double)dir.z};
  double bt = __builtin_inf();
  bool any = false;
  for (int k = 0; k < L.sph_count; ++k) {
    const double* sp = S.lsphD + 4 * (size_t)(L.sph_begin + k);
    double t;
    if (exact_sphere(o, d, ld3(sp), sp[3], &t) && t < bt) { bt = t; any = true; }
  }
  bool prev_hit = false;
  for (int k = 0; k < L.tri_count; ++k) {
    const double* tp = S.ltriD + 9 * (size_t)(L.tri_begin + k);
    double t;
    // (LightShape(Square::from_points): tri0.or(tri1) - the second triangle, which follows its first, does not count
    // where the first was hit)
    if (S.any_pairs && prev_hit && S.ltri_first[L.tri_begin + k] >= 0) { prev_hit = false; continue; }
    prev_hit = exact_tri(o, d, ld3(tp), ld3(tp + 3), ld3(tp + 6), &t);
    if (prev_hit && t < bt) { bt = t; any = true; }
  }
  *t_out = bt;
  return any;
}

struct PhongQueueDev {   // Phong hits handed to k_shadow: 64 B per entry
  float4* a;   // px, py, pz, nx
  float4* b;   // ny, nz, doutx, douty
  float4* c;   // doutz, thr_r, thr_g, thr_b
  uint2* d;    // path id, material id
};

// World::trace for one (ray, hit) of this level (object.rs:326-355): lights' `looked`, then the material.
// H supplies the hit records lazily: A0() = (t, prim, ..) of the lights' `looked` scan, B0()/B1() = both words
// of trace's own scan (the same scan for a static scene).  Writes the path's radiance when it terminates,
// queues Phong hits for k_shadow, and returns whether the path continues (then *npos, *ndir, *thr are the next
// ray and the updated throughput).
template <typename H>
__device__ __forceinline__ bool shade_one(const SceneDev& S, const H& hit, f3 pos, f3 dir, f3* thr_io, uint32_t lpid,
                                          uint32_t level, uint32_t max_depth, uint64_t pid_base, uint32_t k0,
                                          uint32_t k1, Rad3* radiance, const PhongQueueDev& pq,
                                          uint32_t* phong_counts, f3* npos_out, f3* ndir_out) {
  f3 thr = *thr_io;
  f3 npos = mk3(0, 0, 0), ndir = mk3(0, 0, 0);
  bool alive = false;
  // the material row is two dependent loads away from the prim id: start them before the lights and HitInfo::new
  const float4 B0 = hit.B0();
  const int prim = __float_as_int(B0.y);
  const int mat_row = hit.mat(S, prim);
  const MatDev m = S.mats[mat_row];
  // ---- lights: LightSource::looked (object.rs:333-341)
  bool see = false;
  f3 col = mk3(0, 0, 0);
  int kl = 0;   // ordinal of the light among those that look: picks its scan's record where each light has its own
  for (int l = 0; l < S.n_lights; ++l) {
    LightDev L;   // wave-uniform row: scalar loads
    {
      static_assert(sizeof(LightDev) == 44, "LightDev is 11 dwords");
      c_i32p lp = AS_CONST(c_i32p, S.lights) + 11 * l;
      int w[11];
#pragma unroll
      for (int j = 0; j < 11; ++j) w[j] = lp[j];
      __builtin_memcpy(&L, w, sizeof L);
    }
    if (L.kind == 0) {                       // SkyLight::looked, light.rs:184-190
      const int primA = __float_as_int(hit.A0(kl).y);
      if (primA < 0) { see = true; col = col + sky_color_from(dir); }
      ++kl;
    } else if (L.kind == 1) {                // LightShape::looked, light.rs:233-242
      double tl;
      if (light_shape_hit(S, L, pos, dir, &tl)) {
        const int primA = __float_as_int(hit.A0(kl).y);
        if (primA < 0 || tl < hit.tA(kl)) { see = true; col = col + mk3(L.r, L.g, L.b); }   // f64 against f64 (light.rs:236)
      }
      ++kl;
    }
  }
  if (see) {
    radiance[lpid] = Rad3{thr.x * col.x, thr.y * col.y, thr.z * col.z};
  } else {
    if (prim < 0) {
      radiance[lpid] = Rad3{0.f, 0.f, 0.f};   // object.rs:354
    } else {
      const float4 B1 = hit.B1();
      HitInfoF h = hitinfo_new(mk3(B1.y, B1.z, B1.w), mk3(B0.z, B0.w, B1.x), dir);
      uint64_t pid = pid_base + lpid;
      f3 k = mk3(1, 1, 1);
      bool scattered = true;
      // Lambertian, Metal and Dielectric draw from the same Philox block (purpose = scatter of this level);
      // drawn once for the wave, not once per material branch
      u4 rnd = u4{0u, 0u, 0u, 0u};
      if (m.kind <= 2) rnd = philox4x32_10((uint32_t)pid, (uint32_t)(pid >> 32), level, CFT_PURPOSE_SCATTER, k0, k1);
      switch (m.kind) {
        case 0:     // LambertianModel, compose.rs:110-119: lobe radius 1, k = albedo * colour
        case 1: {   // Metal, compose.rs:42-52: lobe radius fuzz (p0), k = albedo (p1) * colour
          const float radius = m.kind == 0 ? 1.0f : m.p0, alb = m.kind == 0 ? m.p0 : m.p1;
          npos = hi_pos(h);
          ndir = unit3(h.dout + gen_point_in_sphere(radius, u24f(rnd.x), u24f(rnd.y)));
          k = mk3(alb * m.r, alb * m.g, alb * m.b);
        } break;
        case 2: {   // Dielectric, compose.rs:74-88
          bool refl = u24f(rnd.x) < hi_reflect_prob(h, m.p0);
          if (!refl) {
            float ratio = h.outward ? m.p0 : 1.0f / m.p0;
            refl = !hi_refract(h, ratio, &npos, &ndir);   // total reflection -> reflect
          }
          if (refl) { npos = hi_pos(h); ndir = h.dout; }
        } break;
        case 4: {   // Specular, basic.rs:120-124 (default scatter = reflect)
          npos = hi_pos(h); ndir = h.dout;
          k = mk3(m.p0, m.p0, m.p0);
        } break;
        case 5: {   // Transparent, basic.rs:168-184
          float ratio = h.outward ? m.p1 : 1.0f / m.p1;
          if (!hi_refract(h, ratio, &npos, &ndir)) { npos = hi_pos(h); ndir = h.dout; }
          float sc = 1.0f - m.p0;
          k = mk3(m.r * sc, m.g * sc, m.b * sc);
        } break;
        default: {  // PhongModel: terminal, handed to k_shadow
          scattered = false;
          uint32_t slot = atomicAdd(&phong_counts[level], 1u);
          pq.a[slot] = make_float4(h.p.x, h.p.y, h.p.z, h.n.x);
          pq.b[slot] = make_float4(h.n.y, h.n.z, h.dout.x, h.dout.y);
          pq.c[slot] = make_float4(h.dout.z, thr.x, thr.y, thr.z);
          pq.d[slot] = make_uint2(lpid, (uint32_t)mat_row);
        } break;
      }
      if (scattered) {
        thr = thr * k;
        if (level + 1 < max_depth) alive = true;
        else radiance[lpid] = Rad3{0.f, 0.f, 0.f};   // trace(.., 0) is black (object.rs:327-329)
      }
    }
  }
  *thr_io = thr; *npos_out = npos; *ndir_out = ndir;
  return alive;
}

struct HitFromRecords {   // the split pipeline: hit records written by k_intersect, loaded on demand
  const LookRecs& A;   // (the kernel's own argument: a wave-uniform k indexes it through scalar loads)
  HitDev B;
  uint32_t i;
  __device__ __forceinline__ double tA(int k) const { return A.r[k < A.n ? k : 0].t[i]; }
  __device__ __forceinline__ int mat(const SceneDev& S, int prim) const { return prim >= 0 ? S.prim_mat[prim] : 0; }
  __device__ __forceinline__ float4 A0(int k) const { return A.r[k < A.n ? k : 0].h0[i]; }
  __device__ __forceinline__ float4 B0() const { return B.h0[i]; }
  __device__ __forceinline__ float4 B1() const { return B.h1[i]; }
};
struct HitInRegisters {   // the fused pipeline
  float4 a0, b0, b1;
  double ta;   // exact distance of the lights' `looked` scan
  int mat_id;  // material row of trace's own hit (0 on a miss)
  __device__ __forceinline__ double tA(int) const { return ta; }   // (one `looked` scan serves every light: the host
  __device__ __forceinline__ int mat(const SceneDev&, int) const { return mat_id; }   // sends moving scenes with several looking lights through the split pipeline)
  __device__ __forceinline__ float4 A0(int) const { return a0; }
  __device__ __forceinline__ float4 B0() const { return b0; }
  __device__ __forceinline__ float4 B1() const { return b1; }
};

// SPLIT pipeline, shade stage: one lane = one (ray, hit) record of this level.  hitA = the lights' `looked`
// scan, hitB = trace's own scan (the same buffers for a static scene).
__global__ __launch_bounds__(CFT_BLOCK) void k_shade(SceneDev S, QueueDev qin, LookRecs hitA, HitDev hitB,
                                                     uint32_t* counts, uint32_t level, uint32_t max_depth,
                                                     uint64_t pid_base, uint32_t k0, uint32_t k1, QueueDev qout,
                                                     Rad3* radiance, PhongQueueDev pq, uint32_t* phong_counts) {
  // Survivors of a 1024-ray chunk are staged in LDS and appended to the next queue with ONE global
  // atomic per chunk (a returning atomic on one word saturates near 88/us chip-wide, which a
  // per-wave atomic would hit at ~2M waves per level), then copied out coalesced.
  __shared__ float4 s_a[CFT_SHADE_CHUNK];
  __shared__ float4 s_b[CFT_SHADE_CHUNK];
  __shared__ float2 s_c[CFT_SHADE_CHUNK];
  __shared__ uint32_t s_cnt, s_base;
  const uint32_t n = counts[level];
  const uint32_t nchunks = (n + CFT_SHADE_CHUNK - 1) / CFT_SHADE_CHUNK;
  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
   if (threadIdx.x == 0) s_cnt = 0;
   __syncthreads();
#pragma unroll CFT_SHADE_UNROLL
   for (uint32_t rr = 0; rr < CFT_SHADE_CHUNK / CFT_BLOCK; ++rr) {
    uint32_t i = chunk * CFT_SHADE_CHUNK + rr * CFT_BLOCK + threadIdx.x;
    bool alive = false;          // continues to the next level
    f3 npos = mk3(0, 0, 0), ndir = mk3(0, 0, 0), thr = mk3(0, 0, 0);
    uint32_t lpid = 0;
    if (i < n) {
      float4 a = qin.a[i];
      float4 b = qin.b[i];
      float2 c = qin.c[i];
      thr = mk3(b.z, b.w, c.x);
      lpid = __float_as_uint(c.y);
      alive = shade_one(S, HitFromRecords{hitA, hitB, i}, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), &thr, lpid, level,
                        max_depth, pid_base, k0, k1, radiance, pq, phong_counts, &npos, &ndir);
    }
    // ---- stage survivors in LDS: one LDS atomic per wave (ballot + prefix count)
    unsigned long long mask = __ballot(alive);
    if (mask) {
      uint32_t cnt = (uint32_t)__popcll(mask);
      uint32_t lane = threadIdx.x & 63u;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&s_cnt, cnt);
      base = __shfl(base, 0);
      if (alive) {
        uint32_t off = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        uint32_t sl = base + off;
        s_a[sl] = make_float4(npos.x, npos.y, npos.z, ndir.x);
        s_b[sl] = make_float4(ndir.y, ndir.z, thr.x, thr.y);
        s_c[sl] = make_float2(thr.z, __uint_as_float(lpid));
      }
    }
   }
   __syncthreads();
   const uint32_t nb = s_cnt;
   if (threadIdx.x == 0 && nb) s_base = atomicAdd(&counts[level + 1], nb);
   __syncthreads();
   const uint32_t gb = s_base;
   for (uint32_t k = threadIdx.x; k < nb; k += CFT_BLOCK) {
     qout.a[gb + k] = s_a[k];
     qout.b[gb + k] = s_b[k];
     qout.c[gb + k] = s_c[k];
   }
   __syncthreads();
  }
}

// ------------------------------------------------------------------ trace level (fused intersect + shade)
// Rays of one wave, reordered by how long their grid walk will be.  A wave walks its R slots one after the other
// and each slot as long as its longest ray, so with random slot membership a wave pays R x the maximum; sorted
// (slot 0 = the 64 shortest walks ... slot R-1 = the longest) it pays about the 1/R, 2/R ... quantiles.  The key is
// the number of cells between the ray's entry into the grid's box and its exit (0: misses the box), counted into 32
// bins in LDS; perm[slot][thread] receives the queue index (0xFFFFFFFF: no ray).  Wave-synchronous: the wave's own
// LDS rows only, no barrier.  The order of rays inside a wave never affects the image.
__device__ __forceinline__ uint32_t walk_key(const SceneDev& S, float4 a, float4 b) {
  const float ox = a.x - S.bcx, oy = a.y - S.bcy, oz = a.z - S.bcz, dx = a.w, dy = b.x, dz = b.y;
  const float ix_ = __builtin_amdgcn_rcpf(fabsf(dx) < 1e-30f ? copysignf(1e-30f, dx) : dx);
  const float iy_ = __builtin_amdgcn_rcpf(fabsf(dy) < 1e-30f ? copysignf(1e-30f, dy) : dy);
  const float iz_ = __builtin_amdgcn_rcpf(fabsf(dz) < 1e-30f ? copysignf(1e-30f, dz) : dz);
  const float ax0 = (S.g0x - ox) * ix_, ax1 = (S.g1x - ox) * ix_;
  const float ay0 = (S.g0y - oy) * iy_, ay1 = (S.g1y - oy) * iy_;
  const float az0 = (S.g0z - oz) * iz_, az1 = (S.g1z - oz) * iz_;
  const float t0 = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fmaxf(fminf(az0, az1), 0.0f));
  const float t1 = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1));
  if (!(t0 <= t1)) return 0u;
  const float cells = (t1 - t0) * fmaf(fabsf(dz), S.gicz, fmaf(fabsf(dy), S.gicy, fabsf(dx) * S.gicx));
  return 1u + (uint32_t)fminf(cells, 30.0f);
}

template <int R, bool ON> struct SortStore {   // LDS of wave_sort_rays; empty in kernels that do not sort
  uint16_t perm[R][CFT_BLOCK];   // queue index of (slot, thread) as an offset from the wave's first ray; 0xFFFF: none
  uint32_t hist[CFT_BLOCK / 64][32];
};
template <int R> struct SortStore<R, false> {};

template <int R>
__device__ __forceinline__ void wave_sort_rays(const SceneDev& S, const QueueDev& q, uint32_t n, uint32_t base,
                                               uint16_t (*perm)[CFT_BLOCK], uint32_t (*hist)[32]) {
  const uint32_t lane = threadIdx.x & 63u, wave = CFT_TID >> 6, col0 = CFT_TID & ~63u;
  if (lane < 32u) hist[wave][lane] = 0u;
#pragma unroll
  for (int r = 0; r < R; ++r) perm[r][CFT_TID] = (uint16_t)0xFFFFu;
  __builtin_amdgcn_wave_barrier();
  uint32_t key[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t i = base + r * 64u;
    key[r] = 32u;
    if (i < n) {
      key[r] = walk_key(S, q.a[i], q.b[i]);
      atomicAdd(&hist[wave][key[r]], 1u);
    }
  }
  __builtin_amdgcn_wave_barrier();
  {   // exclusive prefix over the 32 bins (lanes 0..31)
    const uint32_t h = lane < 32u ? hist[wave][lane] : 0u;
    uint32_t incl = h;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o);
      if (lane >= (uint32_t)o) incl += t;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 32u) hist[wave][lane] = incl - h;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (key[r] < 32u) {
      const uint32_t p = atomicAdd(&hist[wave][key[r]], 1u);   // the ray's rank: bins ascending, arrival order inside
      perm[p >> 6][col0 + (p & 63u)] = (uint16_t)(lane + r * 64u);
    }
  __builtin_amdgcn_wave_barrier();
}

// The default pipeline.  One launch per trace level: the intersect stage above, then World::trace's shading of
// each ray straight from registers, then the survivors are appended to the next level's queue.  Against the
// split pipeline this removes the hit records (32 B written + read per ray and scan) and the second read of
// the ray, and the remaining queue traffic hides under the scan of the other waves of the CU.
// Compaction: a lane's R survivors wait in the park (they take the place of the slot's candidate words);
// ballot + prefix popcount order them inside the wave and ONE global atomic per wave and loop trip (R x 64 rays)
// reserves their slots - 4x fewer atomics than one per wave and ray row, below the ~88/us a single counter
// sustains.
// GEN (level 0 only): the launch also is Camera::emit_rays - isect_setup generates the primary ray of each slot instead
// of loading it, and writes it to the queue (the 40 B a separate k_generate would write) for the slot loop after the
// scan to read back; the writes drain under the scan.  A wave that sorts its rays first (long-list scenes, where the
// primary rays walk the grid too) generates them into the queue before the sort and loads them like any other level.
// The arguments of k_trace_level, as ONE struct at the start of the kernarg segment.  The kernel does not use its
// parameter: each phase of a trip reads what it needs through the kernarg pointer (scalar loads that hit the scalar
// cache), taken afresh so that the compiler cannot keep a value across phases.  Passed as ~110 separate dwords the
// arguments were all loaded at the kernel's entry and, far more than the 100 SGPRs there are, lived in VGPR lanes:
// 11 % of the kernel's instructions were v_readlane / v_writelane moves of them (28 % of the set-up blocks).
struct LevelArgs {
  SceneDev S;
  QueueDev q;
  uint32_t* counts;
  uint32_t level, max_depth;
  uint64_t pid_base;
  uint32_t k0, k1, role0, role1;
  QueueDev qout;
  Rad3* radiance;
  PhongQueueDev pq;
  uint32_t* phong_counts;
  uint32_t* stats;
  GenArgs gen;
  uint32_t in_D, out_k, out_D;
  uint32_t cls_thr;   // > 0: survivors whose next ray crosses at least this many cells go to the long class (CFT_NSEG above)
  // parity hook (cftrt_level_hits; null in every render call): the winner of each scan of each ray as this kernel found
  // it, by local path id - exact distance dbg_t[NS lp + k], global prim id dbg_prim[NS lp + k] (-1: miss)
  double* dbg_t;
  int32_t* dbg_prim;
};
template <typename T>
__device__ __forceinline__ const T& kernel_args() {   // (the kernel's one parameter is a T: offset 0 of the segment)
  typedef const T __attribute__((address_space(4)))* P4;
  P4 p = (P4)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));   // a fresh view: nothing loaded through an earlier one is known to be the same
  return *(const T*)p;
}
__device__ __forceinline__ const LevelArgs& level_args() { return kernel_args<LevelArgs>(); }

// LISTS & 16 (with & 4; scenes of spheres only, static or moving, whose grid image fits CFT_LDS_GRID_BYTES): the instance runs as ONE block
// of 1024 threads per CU - the same 16 waves as four 256-thread blocks, each 256-thread group with its own park
// rows - and the block first copies the grid image into LDS, which every walk then reads instead of the vector
// memory path (grid_walk_ray).  R = 3 and no in-wave sort there: 4 x 27 KB of park leave 50 KB for the grid.
template <int R, int NS, int LISTS, bool GEN = false>
__global__ __launch_bounds__((LISTS & 16) ? 1024 : CFT_BLOCK, 4) void k_trace_level(LevelArgs) {
  constexpr uint32_t BLK = (LISTS & 16) ? 1024u : (uint32_t)CFT_BLOCK;
  __shared__ float park_[BLK / CFT_BLOCK][ParkLayout<NS>::WORDS * R][CFT_BLOCK];
  float (*const park)[CFT_BLOCK] = park_[threadIdx.x / CFT_BLOCK];
  // (the 4-ray instances serve the primary rays, which are not sorted; their park already fills the 40 KB a block
  // may take at 4 blocks per CU)
  // (two-scan instances: measured slower, they spill; the instance with the grid in LDS: its walks are cheap enough
  // that the sort no longer pays - same time with and without, and without it the sort rows' 8 KB go to the grid)
  constexpr bool SORT = (LISTS & 4) != 0 && NS == 1 && (LISTS & 16) == 0;
  __shared__ SortStore<R, SORT> ss_[BLK / CFT_BLOCK];
  SortStore<R, SORT>& ss = ss_[threadIdx.x / CFT_BLOCK];
  LdsGrid lg = CFT_NO_LDS_GRID;
  if constexpr ((LISTS & 16) != 0) {
    constexpr uint32_t CAP16 = (NS == 2 ? CFT_LDS_GRID_BYTES2 : CFT_LDS_GRID_BYTES) / 16;
    __shared__ uint4 lgrid[CAP16];
    const LevelArgs& A = level_args();
    const uint32_t n16 = min(A.S.gridImgBytes >> 4, CAP16);   // (the host launches this instance only when the image fits)
    for (uint32_t i = threadIdx.x; i < n16; i += BLK) lgrid[i] = A.S.gridImg[i];
    const uint32_t ncell = (uint32_t)(A.S.gnx * A.S.gny * A.S.gnz);
    lg.sph = (l_f4p)&lgrid[0];
    lg.cell = (l_u32p)&lgrid[0] + 4u * A.S.gridImgEnt;
    lg.x = (l_f32p)&lgrid[0] + 4u * A.S.gridImgEnt + 2u * ncell;
    __syncthreads();
  }
  // The input queue: CFT_SHARDS segments of stride in_D (a dense queue: segment 0 only), each in two parts - the long
  // class at its top, the short class (or every ray, in a queue without classes) at its bottom (CFT_NSEG above).  A trip
  // takes 64 R slots of ONE part; trips are numbered through the long parts, then the short ones.  The survivors of trip
  // `it` go to output segment it mod out_k (out_k = 1: dense), which therefore receives at most
  // ceil(trips / out_k) * 64 R rays <= out_D, whatever their split into classes.
  uint32_t trips_to[CFT_NSEG + 1];
  trips_to[0] = 0u;
  {
    const LevelArgs& A = level_args();
#pragma unroll
    for (int sh = 0; sh < CFT_NSEG; ++sh)
      trips_to[sh + 1] = trips_to[sh] + (AS_CONST(const uint32_t __attribute__((address_space(4)))*, A.counts)[seg_block((uint32_t)sh) * CFT_SHARD_STRIDE + A.level] + 64u * R - 1u) / (64u * R);
#ifdef CFT_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) g_walk_stats = A.level < 4u ? (unsigned long long*)(A.stats + 8 + 128 + 2) + 32 * A.level : nullptr;
#endif
  }
  const uint32_t stride = 64u;   // wave-contiguous slots, wave-uniform trip count (see k_intersect)
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t n = 0;   // slots of the current trip's segment
  auto slot_index = [&](int r, uint32_t base) -> uint32_t {
    if constexpr (SORT) return ss.perm[r][CFT_TID] == 0xFFFFu ? 0xFFFFFFFFu : base - (threadIdx.x & 63u) + ss.perm[r][CFT_TID];
    else return base + r * 64u < n ? base + r * 64u : 0xFFFFFFFFu;
  };
  CFT_STAMP_DECL;
  for (uint32_t it = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (BLK / 64u) + (threadIdx.x >> 6)));
       it < trips_to[CFT_NSEG]; it += gridDim.x * (BLK / 64u)) {
    uint32_t seg = 0;
#pragma unroll
    for (int sh = 1; sh < CFT_NSEG; ++sh) seg += it >= trips_to[sh] ? 1u : 0u;
    uint32_t seg_trip0 = 0;
#pragma unroll
    for (int sh = 1; sh < CFT_NSEG; ++sh) seg_trip0 = seg == (uint32_t)sh ? trips_to[sh] : seg_trip0;
    const uint32_t wb = (it - seg_trip0) * 64u * R;
    const uint32_t base = wb + lane;
    QueueDev q;   // the trip's segment of the input queue, indexed from 0
    // Level 0 of the fused pipeline (the launch generated the rays itself): the path id of a slot follows from the slot
    // (generate_slot's numbering) - lp_known; then only 24 bytes of a ray went to the queue and come back.
    bool lp_known = false;
    uint32_t lp0 = 0u, lp_slot0 = 0u, lp_w = 0u;   // tiled: path id and slot of the tile's first pixel, image width (0: lp = slot)
    auto slot_lp = [&](uint32_t i) -> uint32_t {
      const uint32_t wi = i - lp_slot0;
      return lp_w ? lp0 + (wi >> 5) * lp_w + (wi & 31u) : i;
    };
    {
      // ---- the rays of the trip: (generated,) sorted, set up, scanned, parked
      const LevelArgs& A = level_args();
      const SceneDev& S = A.S;
      n = AS_CONST(const uint32_t __attribute__((address_space(4)))*, A.counts)[seg_block(seg) * CFT_SHARD_STRIDE + A.level];
      const size_t qoff = seg_first_slot(seg, A.in_D, n);
      q = QueueDev{A.q.a + qoff, A.q.b + qoff, A.q.c + qoff};
      CFT_STAMP(7);   // (loop overhead, and the time before the first trip)
      if constexpr (GEN && SORT) {
#pragma unroll 1
        for (int r = 0; r < R; ++r)
          if (base + r * stride < n) generate_slot(A.gen.cam, A.pid_base, base + r * stride, A.k0, A.k1, A.gen.jitter, A.gen.tiled, q);
        // the wave reads its slots back in sorted order (other lanes' slots too), through the same L1
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      CFT_STAMP(0);
      if constexpr (SORT) {
        // bounce rays that will walk the grid are sorted by walk length inside the wave (primary rays take the cone cull)
        if (S.grid_mode && (A.level > 0 || S.grid_mode == 2)) {
          wave_sort_rays<R>(S, q, n, base, ss.perm, ss.hist);
        } else {
#pragma unroll
          for (int r = 0; r < R; ++r) ss.perm[r][CFT_TID] = base + r * stride < n ? (uint16_t)(lane + r * stride) : (uint16_t)0xFFFFu;
        }
      }
      CFT_STAMP(1);
      RayPack<R, NS> P;
      if constexpr (SORT) {
        isect_setup<R, NS>(S, q, n, wb, stride, A.level, A.pid_base, A.k0, A.k1, A.role0, A.role1, P, park, ss.perm);
      } else if constexpr (GEN) {
        // a tiled frame: the trip's slots are pixels of one tile of one sample - which ones, once for the wave
        TileHint th = CFT_NO_TILE_HINT;
        if (A.gen.tiled && 256 % (64 * R) == 0) {
          const uint32_t npix = A.gen.cam.width * A.gen.cam.height, ntx = A.gen.cam.width >> 5;
          const uint32_t wbu = (uint32_t)__builtin_amdgcn_readfirstlane((int)wb);
          th.smp = wbu / npix;
          const uint32_t t = (wbu - th.smp * npix) >> 8;
          th.ty = t / ntx;
          th.tx = t - th.ty * ntx;
          th.slot0 = th.smp * npix + (t << 8);
          lp_w = A.gen.cam.width;
          lp0 = th.smp * npix + th.ty * 8u * lp_w + th.tx * 32u;
          lp_slot0 = th.slot0;
          lp_known = true;
        } else if (!A.gen.tiled) {
          lp_known = true;   // (lp = slot)
        }
        isect_setup<R, NS>(S, q, n, base, stride, A.level, A.pid_base, A.k0, A.k1, A.role0, A.role1, P, park, nullptr, &A.gen, th, lp_known);
      } else {
        isect_setup<R, NS>(S, q, n, base, stride, A.level, A.pid_base, A.k0, A.k1, A.role0, A.role1, P, park);
      }
      CFT_STAMP(2);
      {
        // (the instances with the walk run the bounce levels, where nearly every wave walks the grid: without the
        // cone-cull code they keep more registers; the few narrow bundles there walk too)
        const LevelArgs& A2 = level_args();
        const SceneDev& S2 = A2.S;
        int tile = -1;   // level 0 of a tiled frame with per-tile lists: the trip's 64 R slots are pixels of one 32x8 tile
        if constexpr (GEN && (LISTS & 4) == 0 && 256 % (64 * R) == 0)
          if (S2.tile_off && A2.gen.tiled) tile = (int)((wb % (A2.gen.cam.width * A2.gen.cam.height)) >> 8);
        scan_lists<R, NS, (LISTS & 4) == 0, (LISTS & 4) ? (LISTS | 8) : LISTS>(S2, P, 0, 1, tile, park, lg);
      }
      CFT_STAMP(3);
      if constexpr (!((LISTS & 4) != 0 && CFT_PARKED_WALK(LISTS, R))) isect_park<R, NS>(P, park);   // (those walk instances park on their way)
    }
    CFT_STAMP(4);
    uint32_t alive_bits = 0, long_bits = 0;
    {
      // ---- the slots one by one: exact evaluation, hit record, shading
      const LevelArgs& A = level_args();
      const SceneDev& S = A.S;
      // the slot's ray is re-read from the queue (HBM: the queue is far bigger than the caches); the next slot's
      // loads are issued before this slot is processed
      uint32_t inext = slot_index(0, base);
      auto load_b = [&](uint32_t j) -> float4 {
        if (lp_known) { const float2 t = q.c[j]; return make_float4(t.x, t.y, 1.0f, 1.0f); }
        return q.b[j];
      };
      auto load_c = [&](uint32_t j) -> float2 { return lp_known ? make_float2(1.0f, __uint_as_float(slot_lp(j))) : q.c[j]; };
      float4 an = q.a[inext < n ? inext : wb], bn = load_b(inext < n ? inext : wb);
      float2 cn = load_c(inext < n ? inext : wb);
#pragma unroll 1
      for (int r = 0; r < R; ++r) {
        if (!(wb + r * stride < n)) break;   // wave-uniform: the whole wave stays together for isect_hit
        const uint32_t i = inext;
        const bool live = i < n;
        const float4 a = an, b = bn;
        const float2 c = cn;
        {
          inext = r + 1 < R ? slot_index(r + 1, base) : 0xFFFFFFFFu;
          const uint32_t j = inext < n ? inext : wb;
          an = q.a[j]; bn = load_b(j); cn = load_c(j);
        }
        HitInRegisters hit;
        isect_hit<NS>(S, live, r, 0, a.x, a.y, a.z, a.w, b.x, b.y, park, A.stats, &hit.a0, &hit.b1, &hit.ta, &hit.mat_id);
        hit.b0 = hit.a0;
        double tb = hit.ta;
        if (NS == 2) isect_hit<NS>(S, live, r, 1, a.x, a.y, a.z, a.w, b.x, b.y, park, A.stats, &hit.b0, &hit.b1, &tb, &hit.mat_id);
        if (A.dbg_t != nullptr && live) {   // (wave-uniform pointer test; never taken in a render call)
          const size_t w0 = (size_t)NS * __float_as_uint(c.y);
          A.dbg_t[w0] = hit.ta; A.dbg_prim[w0] = __float_as_int(hit.a0.y);
          if (NS == 2) { A.dbg_t[w0 + 1] = tb; A.dbg_prim[w0 + 1] = __float_as_int(hit.b0.y); }
        }
        CFT_STAMP(4);   // (diagnostic build: "park" = the exact evaluation and hit record of the slots; "epilogue" = shading)
        f3 thr = mk3(b.z, b.w, c.x), npos, ndir;
        const bool alive = live && shade_one(S, hit, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), &thr, __float_as_uint(c.y), A.level,
                                             A.max_depth, A.pid_base, A.k0, A.k1, A.radiance, A.pq, A.phong_counts, &npos, &ndir);
        if (alive) {   // the survivor takes over the slot's park words (path id: re-read from the queue below)
          alive_bits |= 1u << r;
          if (A.cls_thr != 0u && walk_key(S, make_float4(npos.x, npos.y, npos.z, ndir.x), make_float4(ndir.y, ndir.z, 0.f, 0.f)) >= A.cls_thr)
            long_bits |= 1u << r;
          PARK(r, 0) = npos.x; PARK(r, 1) = npos.y; PARK(r, 2) = npos.z; PARK(r, 3) = ndir.x;
          PARK(r, 4) = ndir.y; PARK(r, 5) = ndir.z; PARK(r, 6) = thr.x;  PARK(r, 7) = thr.y;
          PARK(r, 8) = thr.z;
        }
        CFT_STAMP(5);
      }
    }
    CFT_STAMP(5);
    // append the wave's survivors: slot by slot, lanes in order
    uint32_t total = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) total += (uint32_t)__popcll(__ballot((alive_bits >> r) & 1u));
    if (total) {   // wave-uniform
      const LevelArgs& A = level_args();
      const QueueDev qout = A.qout;
      const uint32_t leader = (uint32_t)__ffsll((long long)__ballot(1)) - 1u;
      uint32_t wbase = 0, lbase = 0;
      const uint32_t oseg = A.out_k > 1u ? it % A.out_k : 0u;
      // the long class (none unless cls_thr is set): reserved from the top of the segment downwards
      uint32_t ltotal = 0;
      if (A.cls_thr != 0u) {
#pragma unroll
        for (int r = 0; r < R; ++r) ltotal += (uint32_t)__popcll(__ballot((long_bits >> r) & 1u));
      }
      if (lane == leader) {
        if (total != ltotal) wbase = atomicAdd(&A.counts[oseg * CFT_SHARD_STRIDE + A.level + 1], total - ltotal);
        if (ltotal) lbase = atomicAdd(&A.counts[(CFT_SHARDS + oseg) * CFT_SHARD_STRIDE + A.level + 1], ltotal);
      }
      wbase = __shfl(wbase, (int)leader) + oseg * A.out_D;
      lbase = oseg * A.out_D + A.out_D - __shfl(lbase, (int)leader) - ltotal;
      // The survivor's path id is re-read from the input queue (a dependent load per slot) - except at level 0 (lp_known)
      // (the R path-id loads issued together ahead of an unrolled loop: slower, S500 17.8 -> 18.5 ms)
#pragma unroll 1
      for (int r = 0; r < R; ++r) {
        const bool lng = ((long_bits >> r) & 1u) != 0u;
        const unsigned long long ml = __ballot(lng);
        const unsigned long long m = __ballot((alive_bits >> r) & 1u) & ~ml;
        if ((alive_bits >> r) & 1u) {
          const uint32_t o = lng ? lbase + (uint32_t)__popcll(ml & ((1ull << lane) - 1ull)) : wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          qout.a[o] = make_float4(PARK(r, 0), PARK(r, 1), PARK(r, 2), PARK(r, 3));
          qout.b[o] = make_float4(PARK(r, 4), PARK(r, 5), PARK(r, 6), PARK(r, 7));
          float pidw;
          if (lp_known) {
            pidw = __uint_as_float(slot_lp(base + (uint32_t)r * 64u));
          } else {
            pidw = q.c[slot_index(r, base)].y;
          }
          qout.c[o] = make_float2(PARK(r, 8), pidw);
        }
        wbase += (uint32_t)__popcll(m);
        lbase += (uint32_t)__popcll(ml);
      }
    }
    CFT_STAMP(6);
  }
#ifdef CFT_STAMPS
  {
    const LevelArgs& A = level_args();
    CFT_STAMP_FLUSH(A.stats, A.level);
  }
#endif
}

// The same level step for launches that cannot fill the chip (the deep levels: 16 k - 300 k live rays against
// 65 k lanes x 4+ waves).  There the launch time is ONE wave's latency through the list, so the list is split
// over the four waves of a block: all four take the same 64 x R rays, each scans a quarter of every list, the
// candidate sets meet in the LDS park (a wave's top three bounds, merged best-two-first so that a third bound
// never displaces an indexed one on a tie), and wave 0 finishes, shades and appends as usual.
template <int R, int NS, int LISTS>
__global__ __launch_bounds__(CFT_BLOCK, 4) void k_trace_tail(SceneDev S, QueueDev q, uint32_t* counts, uint32_t level,
                                                          uint32_t max_depth, uint64_t pid_base, uint32_t k0,
                                                          uint32_t k1, uint32_t role0, uint32_t role1, QueueDev qout,
                                                          Rad3* radiance, PhongQueueDev pq, uint32_t* phong_counts,
                                                          uint32_t* stats, double* dbg_t, int32_t* dbg_prim) {
  __shared__ float park[ParkLayout<NS>::WORDS * R][CFT_BLOCK];
  const uint32_t n = counts[level];
  const uint32_t lane = threadIdx.x & 63u;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t stride = 64u;
  for (uint32_t wb = blockIdx.x * 64u * R; wb < n; wb += gridDim.x * 64u * R) {   // block-uniform: barriers inside
    const uint32_t base = wb + lane;
    {
      RayPack<R, NS> P;
      isect_setup<R, NS>(S, q, n, base, stride, level, pid_base, k0, k1, role0, role1, P, park);
      scan_lists<R, NS, true, LISTS>(S, P, wave, CFT_BLOCK / 64);
      isect_park<R, NS>(P, park);
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          Cand c;
          c.lo1 = PARK(r, ParkLayout<NS>::CAND + 5 * k + 0);
          c.lo2 = PARK(r, ParkLayout<NS>::CAND + 5 * k + 1);
          c.lo3 = PARK(r, ParkLayout<NS>::CAND + 5 * k + 2);
          c.i1 = __float_as_int(PARK(r, ParkLayout<NS>::CAND + 5 * k + 3));
          c.i2 = __float_as_int(PARK(r, ParkLayout<NS>::CAND + 5 * k + 4));
#pragma unroll
          for (int w = 1; w < CFT_BLOCK / 64; ++w) {
            float (*col)[CFT_BLOCK] = park;   // the other wave's column: same lane, thread index + 64 w
            const int t = (int)threadIdx.x + 64 * w;
            const int j = r * ParkLayout<NS>::WORDS + ParkLayout<NS>::CAND + 5 * k;
            const float l1 = col[j + 0][t], l2 = col[j + 1][t], l3 = col[j + 2][t];
            const int i1 = __float_as_int(col[j + 3][t]), i2 = __float_as_int(col[j + 4][t]);
            if (i1 >= 0) cand_insert(c, l1, i1);
            if (i2 >= 0) cand_insert(c, l2, i2);
            if (l3 < __builtin_inff()) cand_insert(c, l3, -1);   // lands third at best: two bounds <= it went in first
          }
          PARK(r, ParkLayout<NS>::CAND + 5 * k + 0) = c.lo1;
          PARK(r, ParkLayout<NS>::CAND + 5 * k + 1) = c.lo2;
          PARK(r, ParkLayout<NS>::CAND + 5 * k + 2) = c.lo3;
          PARK(r, ParkLayout<NS>::CAND + 5 * k + 3) = __int_as_float(c.i1);
          PARK(r, ParkLayout<NS>::CAND + 5 * k + 4) = __int_as_float(c.i2);
        }
      }
    }
    __syncthreads();   // the other waves may overwrite their columns only after the merge
    if (wave != 0) continue;
    uint32_t alive_bits = 0;
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
      if (!(wb + r * stride < n)) break;   // wave-uniform: the whole wave stays together for isect_hit
      const uint32_t i = base + r * stride;
      const bool live = i < n;
      const float4 a = q.a[live ? i : wb];
      const float4 b = q.b[live ? i : wb];
      const float2 c = q.c[live ? i : wb];
      HitInRegisters hit;
      isect_hit<NS>(S, live, r, 0, a.x, a.y, a.z, a.w, b.x, b.y, park, stats, &hit.a0, &hit.b1, &hit.ta, &hit.mat_id);
      hit.b0 = hit.a0;
      double tb = hit.ta;
      if (NS == 2) isect_hit<NS>(S, live, r, 1, a.x, a.y, a.z, a.w, b.x, b.y, park, stats, &hit.b0, &hit.b1, &tb, &hit.mat_id);
      if (dbg_t != nullptr && live) {   // (parity hook, see LevelArgs)
        const size_t w0 = (size_t)NS * __float_as_uint(c.y);
        dbg_t[w0] = hit.ta; dbg_prim[w0] = __float_as_int(hit.a0.y);
        if (NS == 2) { dbg_t[w0 + 1] = tb; dbg_prim[w0 + 1] = __float_as_int(hit.b0.y); }
      }
      f3 thr = mk3(b.z, b.w, c.x), npos, ndir;
      const bool alive = live && shade_one(S, hit, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), &thr, __float_as_uint(c.y), level,
                                           max_depth, pid_base, k0, k1, radiance, pq, phong_counts, &npos, &ndir);
      if (alive) {
        alive_bits |= 1u << r;
        PARK(r, 0) = npos.x; PARK(r, 1) = npos.y; PARK(r, 2) = npos.z; PARK(r, 3) = ndir.x;
        PARK(r, 4) = ndir.y; PARK(r, 5) = ndir.z; PARK(r, 6) = thr.x;  PARK(r, 7) = thr.y;
        PARK(r, 8) = thr.z;
      }
    }
    uint32_t total = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) total += (uint32_t)__popcll(__ballot((alive_bits >> r) & 1u));
    if (total) {   // wave-uniform
      const uint32_t leader = (uint32_t)__ffsll((long long)__ballot(1)) - 1u;
      uint32_t wbase = 0;
      if (lane == leader) wbase = atomicAdd(&counts[level + 1], total);
      wbase = __shfl(wbase, (int)leader);
#pragma unroll 1
      for (int r = 0; r < R; ++r) {
        const unsigned long long m = __ballot((alive_bits >> r) & 1u);
        if ((alive_bits >> r) & 1u) {
          const uint32_t o = wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          qout.a[o] = make_float4(PARK(r, 0), PARK(r, 1), PARK(r, 2), PARK(r, 3));
          qout.b[o] = make_float4(PARK(r, 4), PARK(r, 5), PARK(r, 6), PARK(r, 7));
          qout.c[o] = make_float2(PARK(r, 8), q.c[base + r * stride].y);
        }
        wbase += (uint32_t)__popcll(m);
      }
    }
  }
}

// The last levels of a pass, PATH by path.  Once few rays are alive (the host switches below ~2^20) a launch per level
// no longer fills the chip: each costs one wave's latency through setup, scan, exact evaluation and shading, whatever
// the ray count (S500: 38 launches of 29-74 us for 0.3 % of the rays).  Here a lane keeps one path in registers and
// follows it level after level - closest_hit (the one-ray form: grid walk or lists), World::trace's shading, next
// ray - with no queue, no append and no launch in between; a lane whose path ends takes the next ray of the input
// queue (one atomic per wave and refill), so the launch lasts as long as the queue feeds the lanes plus one longest
// path.  Every hit goes through the same exact evaluation and the same shade_one as in the level kernels: the image
// is the same bit for bit (cftrt_ctx_set_finish toggles it; tested).
// Rays traced per level are counted in LDS and added to counts[level] at the end (counts[level0] is already right).
// Not for scenes with PhongModel materials (their hits go through k_shadow level by level).
struct FinishArgs {   // (one struct, read per phase through the kernarg pointer: see LevelArgs)
  SceneDev S;
  QueueDev q;
  uint32_t* counts;
  uint32_t level0, max_depth;
  uint64_t pid_base;
  uint32_t k0, k1, role0, role1;
  int two_scans;
  Rad3* radiance;
  uint32_t* cursor;
  uint32_t* stats;
  uint32_t in_D;
};
// LISTS: what the one-ray scans may need (scan_lists): 7 = everything, decided at run time (grid off, or any scene);
// with 8 the scene has a grid and every ray walks it - instances without the list-scan code, and without the
// moving-sphere / triangle code where the scene has none, keep fewer registers (more resident waves: the kernel waits
// on latency, not on issue).  (An instance with the grid in LDS was no faster: the launch lasts as long as its longest path.)
#ifndef CFT_FINISH_WAVES
#define CFT_FINISH_WAVES 3   // (measured: S20k 44.5 -> 43.7 ms at 3 waves per SIMD, 168 VGPRs, a few spills; 4 spills 190 B per lane for the same time; S500 unchanged)
#endif
template <int LISTS>
__global__ __launch_bounds__(CFT_BLOCK, (LISTS & 8) ? CFT_FINISH_WAVES : 2) void k_trace_finish(FinishArgs) {
  __shared__ uint32_t hist[CFT_BLOCK / 64][CFTRT_MAX_LEVELS_DEV];
  const FinishArgs& A0 = kernel_args<FinishArgs>();
  uint32_t* const counts = A0.counts;
  const uint32_t level0 = A0.level0, in_D = A0.in_D;
  const QueueDev q = A0.q;
  // the input queue may be sharded: slot v of the concatenated segments sits at seg * in_D + (v - slots before seg)
  // (the segments' parts in trip order, CFT_NSEG above)
  uint32_t slots_to[CFT_NSEG + 1];
  slots_to[0] = 0u;
#pragma unroll
  for (int sh = 0; sh < CFT_NSEG; ++sh) slots_to[sh + 1] = slots_to[sh] + counts[seg_block((uint32_t)sh) * CFT_SHARD_STRIDE + level0];
  const uint32_t n = slots_to[CFT_NSEG];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint32_t l = lane; l < CFTRT_MAX_LEVELS_DEV; l += 64u) hist[wave][l] = 0u;
  __builtin_amdgcn_wave_barrier();
  bool have = false, drained = false;
  f3 pos = mk3(0, 0, 0), dir = mk3(0, 0, 1), thr = mk3(0, 0, 0);
  uint32_t lpid = 0, level = level0;
  const PhongQueueDev no_pq{nullptr, nullptr, nullptr, nullptr};
  while (true) {
    if (!drained) {   // wave-uniform
      uint32_t* const cursor = kernel_args<FinishArgs>().cursor;
      const unsigned long long idle = __ballot(!have);
      if (idle) {
        const uint32_t cnt = (uint32_t)__popcll(idle);
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(idle)) base = atomicAdd(cursor, cnt);
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(idle));
        if (!have) {
          const uint32_t v = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
          if (v < n) {
            size_t i = v;
#pragma unroll
            for (int sh = 0; sh < CFT_NSEG; ++sh)
              if (v >= slots_to[sh] && v < slots_to[sh + 1])
                i = seg_first_slot((uint32_t)sh, in_D, slots_to[sh + 1] - slots_to[sh]) + (v - slots_to[sh]);
            const float4 a = q.a[i], b = q.b[i];
            const float2 c = q.c[i];
            pos = mk3(a.x, a.y, a.z); dir = mk3(a.w, b.x, b.y); thr = mk3(b.z, b.w, c.x);
            lpid = __float_as_uint(c.y);
            level = level0;
            have = true;
          }
        }
        drained = base + cnt >= n;
      }
    }
    if (!__ballot(have)) break;   // the queue is empty and every path of this wave has ended
    if (have) {
      // rays per level: one LDS add per distinct level among the wave's paths (refilled lanes lag behind the others)
      for (unsigned long long todo = __ballot(1); todo;) {
        const int lead = (int)__builtin_ctzll(todo);
        const uint32_t lv = (uint32_t)__builtin_amdgcn_readlane((int)level, lead);
        const unsigned long long m = __ballot(level == lv);
        if (lane == (uint32_t)lead) hist[wave][lv] += (uint32_t)__popcll(m);
        todo &= ~m;
      }
      const FinishArgs& A = kernel_args<FinishArgs>();
      const SceneDev& S = A.S;
      const uint64_t pid_base = A.pid_base;
      const uint32_t k0 = A.k0, k1 = A.k1, role0 = A.role0, role1 = A.role1;
      const int two_scans = A.two_scans;
      uint32_t* const stats = A.stats;
      uint32_t wA = 0, wB = 0;
      if (S.any_motion) {
        const uint64_t pid = pid_base + lpid;
        wA = philox4x32_10((uint32_t)pid, (uint32_t)(pid >> 32), level, CFT_PURPOSE_SCAN0 + role0, k0, k1).x;
        wB = two_scans ? philox4x32_10((uint32_t)pid, (uint32_t)(pid >> 32), level, CFT_PURPOSE_SCAN0 + role1, k0, k1).x : wA;
      }
      const Closest cA = closest_hit<LISTS>(S, pos.x, pos.y, pos.z, dir.x, dir.y, dir.z, wA, stats);
      Closest cB = cA;
      if (two_scans) cB = closest_hit<LISTS>(S, pos.x, pos.y, pos.z, dir.x, dir.y, dir.z, wB, stats);
      HitInRegisters hit;
      hit.ta = cA.t;
      hit.a0 = make_float4((float)cA.t, __int_as_float(cA.g), 0.f, 0.f);
      hit.b0 = make_float4(__builtin_inff(), __int_as_float(-1), 0.f, 0.f);
      hit.b1 = make_float4(0.f, 0.f, 0.f, 0.f);
      hit.mat_id = cB.g >= 0 ? S.prim_mat[cB.g] : 0;
      if (cB.g >= 0) {   // the hit record as isect_hit builds it
        const d3 o = d3{(double)pos.x, (double)pos.y, (double)pos.z};
        const d3 d = d3{(double)dir.x, (double)dir.y, (double)dir.z};
        const d3 p = o + d * cB.t;
        const f3 nn = hit_normal(S, cB.g, p, wB);
        hit.b0 = make_float4((float)cB.t, __int_as_float(cB.g), (float)p.x, (float)p.y);
        hit.b1 = make_float4((float)p.z, nn.x, nn.y, nn.z);
      }
      f3 npos, ndir;
      {
        const FinishArgs& A2 = kernel_args<FinishArgs>();
        have = shade_one(A2.S, hit, pos, dir, &thr, lpid, level, A2.max_depth, A2.pid_base, A2.k0, A2.k1, A2.radiance, no_pq, nullptr, &npos, &ndir);
      }
      pos = npos; dir = ndir;
      level += 1;
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (uint32_t l = lane; l < CFTRT_MAX_LEVELS_DEV; l += 64u)
    if (l > level0 && hist[wave][l]) atomicAdd(&counts[l], hist[wave][l]);
}

// ------------------------------------------------------------------ shadow (PhongModel::render)
// one lane = one Phong hit; loops over the lights (basic.rs:57-94).  Every (hit, light) pair scans
// one ray against the world: the mirror ray (sky / shape lights: `looked`, then `is_in_shadow` on the
// same ray) or the ray towards the light (point / parallel lights).
struct ShadowArgs {   // (one struct, read per phase through the kernarg pointer: see LevelArgs)
  SceneDev S;
  PhongQueueDev pq;
  const uint32_t* phong_counts;
  uint32_t level;
  uint64_t pid_base;
  uint32_t k0, k1;
  Rad3* radiance;
  uint32_t* stats;
};
__global__ __launch_bounds__(CFT_BLOCK) void k_shadow(ShadowArgs) {
  const ShadowArgs& A0 = kernel_args<ShadowArgs>();
  const uint32_t n = A0.phong_counts[A0.level];
  const uint32_t nl = (uint32_t)A0.S.n_lights;
  const uint32_t level = A0.level;
  for (uint32_t i = blockIdx.x * CFT_BLOCK + threadIdx.x; i < n; i += gridDim.x * CFT_BLOCK) {
    const ShadowArgs& A = kernel_args<ShadowArgs>();
    const SceneDev& S = A.S;
    const PhongQueueDev pq = A.pq;
    const uint64_t pid_base = A.pid_base;
    const uint32_t k0 = A.k0, k1 = A.k1;
    uint32_t* const stats = A.stats;
    float4 a = pq.a[i], b = pq.b[i], c = pq.c[i];
    uint2 dd = pq.d[i];
    f3 p = mk3(a.x, a.y, a.z), nrm = mk3(a.w, b.x, b.y), dout = mk3(b.z, b.w, c.x);
    f3 thr = mk3(c.y, c.z, c.w);
    const MatDev m = S.mats[dd.y];
    uint64_t pid = pid_base + dd.x;
    f3 pos = madd3(p, dout, CFT_EPS);   // hit.pos()
    f3 sum = mk3(0, 0, 0);
    for (uint32_t j = 0; j < nl; ++j) {
      const LightDev L = S.lights[j];
      uint32_t wl = 0, ws = 0;
      if (S.any_motion) {
        wl = philox4x32_10((uint32_t)pid, (uint32_t)(pid >> 32), level, CFT_PURPOSE_SCAN0 + nl + 1 + 2 * j, k0, k1).x;
        ws = philox4x32_10((uint32_t)pid, (uint32_t)(pid >> 32), level, CFT_PURPOSE_SCAN0 + nl + 2 + 2 * j, k0, k1).x;
      }
      const bool mirror = L.kind == 0 || L.kind == 1;
      f3 lp = mk3(L.vx, L.vy, L.vz);
      f3 ldir = (L.kind == 2) ? unit3(pos - lp) : (L.kind == 3 ? lp : -dout);   // dir_at(hit)
      f3 rd = mirror ? dout : unit3(-ldir);                                      // the ray scanned
      Closest h1 = closest_hit(S, pos.x, pos.y, pos.z, rd.x, rd.y, rd.z, mirror ? wl : ws, stats);
      Closest h2 = h1;
      if (S.any_motion && mirror) h2 = closest_hit(S, pos.x, pos.y, pos.z, rd.x, rd.y, rd.z, ws, stats);
      f3 lcol = mk3(L.r, L.g, L.b);
      float inten = 1.0f;
      bool shadow;
      if (L.kind == 0) {            // SkyLight, light.rs:166-191
        if (h1.g < 0) { sum = sum + sky_color_from(dout); continue; }          // looked(reflect) -> return c
        lcol = sky_color_from(dout);
        shadow = h2.g >= 0;
      } else if (L.kind == 1) {     // LightShape, light.rs:207-243
        double tl;
        bool hs = light_shape_hit(S, L, pos, dout, &tl);
        if (hs && (h1.g < 0 || tl < h1.t)) { sum = sum + lcol; continue; }
        inten = hs ? 1.0f : 0.0f;
        shadow = h2.g >= 0 && hs && h2.t < tl;
      } else if (L.kind == 2) {     // PointLight, light.rs:111-135
        f3 dl = lp - pos;
        float l2 = dot3(dl, dl);
        inten = 1.0f / l2;
        float t = (float)h1.t;      // |point - hit.pos()|^2 = t^2 for the unit ray
        shadow = h1.g >= 0 && (t * t + CFT_EPS < l2);
      } else {                      // ParallelLight, light.rs:86-102
        shadow = h1.g >= 0;
      }
      float ratio2 = dot3(dout, -ldir);
      float ratio = powf(ratio2, m.p0);
      ratio = (ratio < 1.0f) ? ratio : 1.0f;     // min!(ratio, 1.): NaN -> 1
      ratio = (ratio > 0.0f) ? ratio : 0.0f;     // max!(ratio, 0.)
      float dn = dot3(nrm, -ldir);
      float di = (dn > 0.0f) ? dn : 0.0f;
      float w = shadow ? 0.1f : (ratio * 0.5f + di * 0.5f + 0.1f);
      sum = madd3(sum, lcol, inten * w);
    }
    f3 outc = mk3(m.p1 * sum.x * m.r, m.p1 * sum.y * m.g, m.p1 * sum.z * m.b);   // kd * c * color
    kernel_args<ShadowArgs>().radiance[dd.x] = Rad3{thr.x * outc.x, thr.y * outc.y, thr.z * outc.z};
  }
}

// The host's peek at a level's ray count: the sum of the level's counter blocks (kernels.h, CFT_NSEG), written to
// pinned host memory by one wave - the host then only waits for the stream (a strided device-to-host copy of the 16
// words took 40 - 160 us of idle GPU per peek).
__global__ __launch_bounds__(64) void k_peek(const uint32_t* counts_level, uint32_t* host_out) {
  uint32_t v = threadIdx.x < (uint32_t)CFT_NSEG ? counts_level[threadIdx.x * CFT_SHARD_STRIDE] : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (threadIdx.x == 0) {
    *host_out = v;
    __threadfence_system();
  }
}

// ------------------------------------------------------------------ resolve
// one lane = one pixel: the pass's samples are added to accum[pix] one by one, in sample order.  The running sum
// starts from the value already there, so the result does not depend on how a call's sample range was cut into
// passes (workspace limit, static workspace): one pass and many give the same bits.
__global__ __launch_bounds__(CFT_BLOCK) void k_resolve(const Rad3* radiance, uint32_t npix, uint32_t nsamples,
                                                       float* accum) {
  for (uint32_t p = blockIdx.x * CFT_BLOCK + threadIdx.x; p < npix; p += gridDim.x * CFT_BLOCK) {
    float r = accum[3 * (size_t)p], g = accum[3 * (size_t)p + 1], b = accum[3 * (size_t)p + 2];
    for (uint32_t s = 0; s < nsamples; ++s) {
      const Rad3 v = radiance[(size_t)s * npix + p];
      r += v.x; g += v.y; b += v.z;
    }
    accum[3 * (size_t)p] = r;
    accum[3 * (size_t)p + 1] = g;
    accum[3 * (size_t)p + 2] = b;
  }
}

// vec3_to_rgb, scene.rs:122-127
__global__ __launch_bounds__(CFT_BLOCK) void k_rgb8(const float* accum, uint64_t n, float div, uint8_t* out) {
  for (uint64_t i = blockIdx.x * (uint64_t)CFT_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * CFT_BLOCK) {
    float c = accum[i] / div;
    float m = (1.0f < c) ? 1.0f : c;
    float x = (0.0f > m) ? 0.0f : m;
    out[i] = (uint8_t)(255.99f * sqrtf(x));
  }
}
