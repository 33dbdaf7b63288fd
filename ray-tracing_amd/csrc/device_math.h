// device_math.h — f32 vector helpers, Philox4x32-10 and the reference's sampling helpers for gfx950.
// The whole library is built with -ffp-contract=off; every fused multiply-add below is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CFT_EPS 1e-3f  // util.rs:10 — the crate's single epsilon

struct f3 {
  float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
  return f3{fmaf(a.y, b.z, -a.z * b.y), fmaf(a.z, b.x, -a.x * b.z), fmaf(a.x, b.y, -a.y * b.x)};
}
// a + b*s
__device__ __forceinline__ f3 madd3(f3 a, f3 b, float s) {
  return f3{fmaf(b.x, s, a.x), fmaf(b.y, s, a.y), fmaf(b.z, s, a.z)};
}
// util.rs:48-50 (v / len).  1 / len is the hardware reciprocal square root (1 ulp) refined by one Newton step, so the
// result is unit to ~1 ulp like the IEEE sqrt + divide form, at a quarter of its instructions (the shading runs in
// f32 against an f64 reference either way; the cull's error model allows directions that are unit to a few ulp).
__device__ __forceinline__ f3 unit3(f3 a) {
  const float x = dot3(a, a);
  float y = __builtin_amdgcn_rsqf(x);
  y = y * fmaf(-0.5f * x, y * y, 1.5f);
  return a * y;
}

// ---- Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11), counter = (pid_lo, pid_hi, level, purpose)
struct u4 {
  uint32_t x, y, z, w;
};
__host__ __device__ __forceinline__ u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                     uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return u4{c0, c1, c2, c3};
}
#define CFT_PURPOSE_CAMERA 0u
#define CFT_PURPOSE_SCATTER 1u
#define CFT_PURPOSE_SCAN0 16u
#define CFT_LEVEL_CAMERA 0xFFFFFFFFu

// 24-bit uniform in [0,1): exact in f32
__host__ __device__ __forceinline__ float u24f(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }
__host__ __device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// per-object motion time for one scan (object.rs:42-46; one fresh draw per object per scan)
__host__ __device__ __forceinline__ float motion_time(uint32_t scan_word, uint32_t obj) {
  return u24f(lowbias32(scan_word ^ (obj * 0x9E3779B9u)));
}

// util.rs:384-395 — on the sphere surface, z >= 0, not uniform (reproduced as is): theta = 2 pi u_theta,
// phi = pi u_phi - pi/2, i.e. sin(phi) = -cos(pi u_phi), cos(phi) = sin(pi u_phi).
__device__ __forceinline__ f3 gen_point_in_sphere(float radius, float u_theta, float u_phi) {
  // v_sin_f32 / v_cos_f32 take their argument in revolutions - theta = 2 pi u_theta, pi u_phi = 2 pi (u_phi / 2), both
  // in [0, 1): no range reduction - and are good to ~1e-6 absolute: four instructions where two sincospif calls issue
  // ~50.  The scattered direction is f32 against the reference's f64 either way (same share of paths within 1e-3 of the
  // oracle's, tests/test_gpu_parity.py::test_per_path_radiance); once the walk's loads left the level kernels
  // VALU-bound (round 3) this is 1-2 % of a frame.  The camera's lens sample below keeps sincospif.
  const float st = __builtin_amdgcn_sinf(u_theta), ct = __builtin_amdgcn_cosf(u_theta);
  const float s1 = __builtin_amdgcn_sinf(0.5f * u_phi), c1 = __builtin_amdgcn_cosf(0.5f * u_phi);
  const float sp = -c1, cp = s1;
  return f3{radius * sp * ct, radius * sp * st, radius * cp};
}
// util.rs:397-402 — half disk
__device__ __forceinline__ void gen_point_in_disk(float radius, float u_theta, float u_r, float* x, float* y) {
  float st, ct;
  sincospif(u_theta, &st, &ct);
  float k = radius * u_r;
  *x = k * ct;
  *y = k * st;
}
// light.rs:156-163
__device__ __forceinline__ f3 sky_color_from(f3 dir) {
  float t = 0.5f * (dir.z + 1.0f);
  float v = 1.0f - t;
  return f3{fmaf(t, 0.5f, v), fmaf(t, 0.7f, v), fmaf(t, 1.0f, v)};
}
