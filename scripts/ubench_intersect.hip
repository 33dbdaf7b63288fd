// Micro-benchmark of inner-loop variants for the sphere scan (not part of the product).
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/ubench_intersect.hip -o scripts/ubench_intersect
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstdint>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef const v16f __attribute__((address_space(4))) * c_f16p;
typedef const v4f __attribute__((address_space(4))) * c_f4p;
#define AS_CONST(T, p) ((T)(uintptr_t)(p))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// mode 0: coherent primary rays; mode 1: incoherent secondary-like rays
__global__ void k_make_rays(float4* a, float2* b, uint32_t n, int mode) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float ox, oy, oz, dx, dy, dz;
  if (mode == 0) {
    uint32_t W = 1920; uint32_t x = i % W, y = (i / W) % 1080;
    ox = 13.f; oy = -3.f; oz = 2.f;
    float u = ((float)x / W - 0.5f) * 0.6f, v = (0.5f - (float)y / 1080.f) * 0.35f;
    float sx = -13.f, sy = 3.f, sz = -2.f; float il = rsqrtf(sx*sx+sy*sy+sz*sz); sx*=il; sy*=il; sz*=il;
    float rx = sy, ry = -sx, rz = 0.f; float ir = rsqrtf(rx*rx+ry*ry); rx*=ir; ry*=ir;
    float ux = ry*sz - rz*sy, uy = rz*sx - rx*sz, uz = rx*sy - ry*sx;
    dx = sx + u*rx + v*ux; dy = sy + u*ry + v*uy; dz = sz + u*rz + v*uz;
  } else {
    ox = -11.f + 22.f * u01(hash32(i * 7u + 1u)); oy = -11.f + 22.f * u01(hash32(i * 7u + 2u)); oz = 0.001f + 0.5f * u01(hash32(i * 7u + 3u));
    float th = 6.2831853f * u01(hash32(i * 7u + 4u)), z = -0.3f + 1.3f * u01(hash32(i * 7u + 5u));
    float r = sqrtf(fmaxf(0.f, 1.f - z * z)); dx = r * cosf(th); dy = r * sinf(th); dz = z;
  }
  float il = rsqrtf(dx*dx+dy*dy+dz*dz); dx*=il; dy*=il; dz*=il;
  a[i] = make_float4(ox, oy, oz, dx); b[i] = make_float2(dy, dz);
}

#define SLOW(b, disc, idx, best_t, best_i) { float sq = __builtin_amdgcn_sqrtf(disc); float t1 = -b - sq, t2 = sq - b; float t = t1 < 0.f ? t2 : t1; if (t2 >= 0.f && t < best_t) { best_t = t; best_i = idx; } }

__device__ __forceinline__ void test1(float ox,float oy,float oz,float dx,float dy,float dz,float cx,float cy,float cz,float r2,int idx,float& bt,int& bi) {
  float ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
  float b = fmaf(ocz, dz, fmaf(ocy, dy, ocx * dx));
  float lx = fmaf(-b, dx, ocx), ly = fmaf(-b, dy, ocy), lz = fmaf(-b, dz, ocz);
  float disc = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, r2)));
  if (disc >= 0.f) SLOW(b, disc, idx, bt, bi)
}

// A: 1 ray/lane, scalar operands x16 prefetch, branch per sphere (the product's loop)
__global__ __launch_bounds__(256) void k_A(const float4* ra, const float2* rb, uint32_t n, const float4* sph, int ns, float* out_t, int* out_i) {
  c_f16p sp = AS_CONST(c_f16p, sph); const int n4 = (ns + 3) >> 2;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float4 a = ra[i]; float2 bb = rb[i];
    float bt = __builtin_inff(); int bi = -1;
    v16f cur = sp[0];
    for (int s4 = 0; s4 < n4; ++s4) {
      v16f nxt = sp[s4 + 1]; int s = s4 << 2;
      test1(a.x,a.y,a.z,a.w,bb.x,bb.y,cur[0],cur[1],cur[2],cur[3],s,bt,bi);
      test1(a.x,a.y,a.z,a.w,bb.x,bb.y,cur[4],cur[5],cur[6],cur[7],s+1,bt,bi);
      test1(a.x,a.y,a.z,a.w,bb.x,bb.y,cur[8],cur[9],cur[10],cur[11],s+2,bt,bi);
      test1(a.x,a.y,a.z,a.w,bb.x,bb.y,cur[12],cur[13],cur[14],cur[15],s+3,bt,bi);
      cur = nxt;
    }
    out_t[i] = bt; out_i[i] = bi;
  }
}

// C: 2 rays/lane, scalar operands, one branch per sphere for both rays
__global__ __launch_bounds__(256) void k_C(const float4* ra, const float2* rb, uint32_t n, const float4* sph, int ns, float* out_t, int* out_i) {
  c_f16p sp = AS_CONST(c_f16p, sph); const int n4 = (ns + 3) >> 2;
  uint32_t half = (n + 1) / 2;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < half; i += gridDim.x * 256) {
    uint32_t j = i + half < n ? i + half : i;
    float4 a0 = ra[i]; float2 b0 = rb[i]; float4 a1 = ra[j]; float2 b1 = rb[j];
    float bt0 = __builtin_inff(), bt1 = __builtin_inff(); int bi0 = -1, bi1 = -1;
    v16f cur = sp[0];
    for (int s4 = 0; s4 < n4; ++s4) {
      v16f nxt = sp[s4 + 1]; int s = s4 << 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float cx = cur[4*k], cy = cur[4*k+1], cz = cur[4*k+2], r2 = cur[4*k+3];
        float ocx0 = a0.x - cx, ocy0 = a0.y - cy, ocz0 = a0.z - cz;
        float ocx1 = a1.x - cx, ocy1 = a1.y - cy, ocz1 = a1.z - cz;
        float q0 = fmaf(ocz0, b0.y, fmaf(ocy0, b0.x, ocx0 * a0.w));
        float q1 = fmaf(ocz1, b1.y, fmaf(ocy1, b1.x, ocx1 * a1.w));
        float lx0 = fmaf(-q0, a0.w, ocx0), ly0 = fmaf(-q0, b0.x, ocy0), lz0 = fmaf(-q0, b0.y, ocz0);
        float lx1 = fmaf(-q1, a1.w, ocx1), ly1 = fmaf(-q1, b1.x, ocy1), lz1 = fmaf(-q1, b1.y, ocz1);
        float d0 = fmaf(-lz0, lz0, fmaf(-ly0, ly0, fmaf(-lx0, lx0, r2)));
        float d1 = fmaf(-lz1, lz1, fmaf(-ly1, ly1, fmaf(-lx1, lx1, r2)));
        if (fmaxf(d0, d1) >= 0.f) {
          if (d0 >= 0.f) SLOW(q0, d0, s + k, bt0, bi0)
          if (d1 >= 0.f) SLOW(q1, d1, s + k, bt1, bi1)
        }
      }
      cur = nxt;
    }
    out_t[i] = bt0; out_i[i] = bi0; if (j != i) { out_t[j] = bt1; out_i[j] = bi1; }
  }
}

// D: 2 rays/lane with packed f32 math
__device__ __forceinline__ v2f pfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__global__ __launch_bounds__(256) void k_D(const float4* ra, const float2* rb, uint32_t n, const float4* sph, int ns, float* out_t, int* out_i) {
  c_f16p sp = AS_CONST(c_f16p, sph); const int n4 = (ns + 3) >> 2;
  uint32_t half = (n + 1) / 2;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < half; i += gridDim.x * 256) {
    uint32_t j = i + half < n ? i + half : i;
    float4 a0 = ra[i]; float2 b0 = rb[i]; float4 a1 = ra[j]; float2 b1 = rb[j];
    v2f ox = {a0.x, a1.x}, oy = {a0.y, a1.y}, oz = {a0.z, a1.z}, dx = {a0.w, a1.w}, dy = {b0.x, b1.x}, dz = {b0.y, b1.y};
    float bt0 = __builtin_inff(), bt1 = __builtin_inff(); int bi0 = -1, bi1 = -1;
    v16f cur = sp[0];
    for (int s4 = 0; s4 < n4; ++s4) {
      v16f nxt = sp[s4 + 1]; int s = s4 << 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v2f cx = {cur[4*k], cur[4*k]}, cy = {cur[4*k+1], cur[4*k+1]}, cz = {cur[4*k+2], cur[4*k+2]}, r2 = {cur[4*k+3], cur[4*k+3]};
        v2f ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
        v2f q = pfma(ocz, dz, pfma(ocy, dy, ocx * dx));
        v2f lx = pfma(-q, dx, ocx), ly = pfma(-q, dy, ocy), lz = pfma(-q, dz, ocz);
        v2f d = pfma(-lz, lz, pfma(-ly, ly, pfma(-lx, lx, r2)));
        if (fmaxf(d.x, d.y) >= 0.f) {
          if (d.x >= 0.f) SLOW(q.x, d.x, s + k, bt0, bi0)
          if (d.y >= 0.f) SLOW(q.y, d.y, s + k, bt1, bi1)
        }
      }
      cur = nxt;
    }
    out_t[i] = bt0; out_i[i] = bi0; if (j != i) { out_t[j] = bt1; out_i[j] = bi1; }
  }
}


// E: R rays/lane (template), scalar operands, one uniform (ballot) branch per sphere
template <int R>
__global__ __launch_bounds__(256) void k_E(const float4* ra, const float2* rb, uint32_t n, const float4* sph, int ns, float* out_t, int* out_i) {
  c_f16p sp = AS_CONST(c_f16p, sph); const int n4 = (ns + 3) >> 2;
  uint32_t per = (n + R - 1) / R;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < per; i += gridDim.x * 256) {
    float ox[R], oy[R], oz[R], dx[R], dy[R], dz[R], bt[R]; int bi[R]; uint32_t idx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      uint32_t j = i + r * per; if (j >= n) j = i; idx[r] = j;
      float4 a = ra[j]; float2 b = rb[j];
      ox[r] = a.x; oy[r] = a.y; oz[r] = a.z; dx[r] = a.w; dy[r] = b.x; dz[r] = b.y; bt[r] = __builtin_inff(); bi[r] = -1;
    }
    v16f cur = sp[0];
    for (int s4 = 0; s4 < n4; ++s4) {
      v16f nxt = sp[s4 + 1]; int s = s4 << 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float cx = cur[4*k], cy = cur[4*k+1], cz = cur[4*k+2], r2 = cur[4*k+3];
        float q[R], d[R]; float m = -1.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float ocx = ox[r] - cx, ocy = oy[r] - cy, ocz = oz[r] - cz;
          q[r] = fmaf(ocz, dz[r], fmaf(ocy, dy[r], ocx * dx[r]));
          float lx = fmaf(-q[r], dx[r], ocx), ly = fmaf(-q[r], dy[r], ocy), lz = fmaf(-q[r], dz[r], ocz);
          d[r] = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, r2)));
          m = fmaxf(m, d[r]);
        }
        if (__ballot(m >= 0.f) != 0ull) {
#pragma unroll
          for (int r = 0; r < R; ++r) if (d[r] >= 0.f) SLOW(q[r], d[r], s + k, bt[r], bi[r])
        }
      }
      cur = nxt;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) if (r == 0 || idx[r] != i) { out_t[idx[r]] = bt[r]; out_i[idx[r]] = bi[r]; }
  }
}

// G: 1 ray/lane, LDS broadcast operands
__global__ __launch_bounds__(256) void k_G(const float4* ra, const float2* rb, uint32_t n, const float4* sph, int ns, float* out_t, int* out_i) {
  __shared__ float4 s_sph[512];
  for (int k = threadIdx.x; k < ns; k += 256) s_sph[k] = sph[k];
  __syncthreads();
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float4 a = ra[i]; float2 bb = rb[i];
    float bt = __builtin_inff(); int bi = -1;
#pragma unroll 4
    for (int s = 0; s < ns; ++s) {
      float4 c = s_sph[s];
      test1(a.x,a.y,a.z,a.w,bb.x,bb.y,c.x,c.y,c.z,c.w,s,bt,bi);
    }
    out_t[i] = bt; out_i[i] = bi;
  }
}

// H: 1 ray/lane, branch-free selection (no divergence, more VALU)
__global__ __launch_bounds__(256) void k_H(const float4* ra, const float2* rb, uint32_t n, const float4* sph, int ns, float* out_t, int* out_i) {
  c_f16p sp = AS_CONST(c_f16p, sph); const int n4 = (ns + 3) >> 2;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float4 a = ra[i]; float2 bb = rb[i];
    float bt = __builtin_inff(); int bi = -1;
    v16f cur = sp[0];
    for (int s4 = 0; s4 < n4; ++s4) {
      v16f nxt = sp[s4 + 1]; int s = s4 << 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float cx = cur[4*k], cy = cur[4*k+1], cz = cur[4*k+2], r2 = cur[4*k+3];
        float ocx = a.x - cx, ocy = a.y - cy, ocz = a.z - cz;
        float q = fmaf(ocz, bb.y, fmaf(ocy, bb.x, ocx * a.w));
        float lx = fmaf(-q, a.w, ocx), ly = fmaf(-q, bb.x, ocy), lz = fmaf(-q, bb.y, ocz);
        float d = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, r2)));
        float sq = __builtin_amdgcn_sqrtf(fmaxf(d, 0.f));
        float t1 = -q - sq, t2 = sq - q; float t = t1 < 0.f ? t2 : t1;
        bool ok = d >= 0.f && t2 >= 0.f && t < bt;
        bt = ok ? t : bt; bi = ok ? s + k : bi;
      }
      cur = nxt;
    }
    out_t[i] = bt; out_i[i] = bi;
  }
}

// raw VALU throughput probes
__global__ __launch_bounds__(256) void k_fma(float* out, int iters, float s) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
    x0 = fmaf(x0, s, 1.f); x1 = fmaf(x1, s, 1.f); x2 = fmaf(x2, s, 1.f); x3 = fmaf(x3, s, 1.f);
    x4 = fmaf(x4, s, 1.f); x5 = fmaf(x5, s, 1.f); x6 = fmaf(x6, s, 1.f); x7 = fmaf(x7, s, 1.f);
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ __launch_bounds__(256) void k_pkfma(float* out, int iters, float s) {
  v2f x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  v2f ss = {s, s}, one = {1.f, 1.f};
  for (int i = 0; i < iters; ++i) {
    x0 = pfma(x0, ss, one); x1 = pfma(x1, ss, one); x2 = pfma(x2, ss, one); x3 = pfma(x3, ss, one);
    x4 = pfma(x4, ss, one); x5 = pfma(x5, ss, one); x6 = pfma(x6, ss, one); x7 = pfma(x7, ss, one);
  }
  v2f r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
}
__global__ __launch_bounds__(256) void k_dfma(double* out, int iters, double s) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
    x0 = fma(x0, s, 1.); x1 = fma(x1, s, 1.); x2 = fma(x2, s, 1.); x3 = fma(x3, s, 1.);
    x4 = fma(x4, s, 1.); x5 = fma(x5, s, 1.); x6 = fma(x6, s, 1.); x7 = fma(x7, s, 1.);
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <typename F> float time_ms(F f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main() {
  const uint32_t N = 1u << 23;   // 8.4M rays
  const int NS = 500;
  std::vector<float4> sph;
  sph.push_back(make_float4(4.f, 0.f, 1.f, 1.f)); sph.push_back(make_float4(0.f, 0.f, 1.f, 1.f)); sph.push_back(make_float4(-4.f, 0.f, 1.f, 1.f));
  srand(1);
  for (int a = -11; a < 11 && (int)sph.size() < NS; ++a) for (int b = -11; b < 11 && (int)sph.size() < NS; ++b)
    sph.push_back(make_float4(a + 0.9f * rand() / RAND_MAX, b + 0.9f * rand() / RAND_MAX, 0.2f, 0.04f));
  while ((int)sph.size() < NS) sph.push_back(make_float4(11.5f, -11.f + sph.size() % 20, 0.2f, 0.04f));
  while (sph.size() % 4) sph.push_back(make_float4(0, 0, 0, -1.f));
  for (int k = 0; k < 4; ++k) sph.push_back(make_float4(0, 0, 0, -1.f));
  float4 *d_sph, *ra; float2* rb; float* ot; int* oi; float *ot2; int* oi2;
  CK(hipMalloc(&d_sph, sph.size() * 16)); CK(hipMemcpy(d_sph, sph.data(), sph.size() * 16, hipMemcpyHostToDevice));
  CK(hipMalloc(&ra, (size_t)N * 16)); CK(hipMalloc(&rb, (size_t)N * 8)); CK(hipMalloc(&ot, (size_t)N * 4)); CK(hipMalloc(&oi, (size_t)N * 4));
  CK(hipMalloc(&ot2, (size_t)N * 4)); CK(hipMalloc(&oi2, (size_t)N * 4));
  const int grid = 256 * 8;
  std::vector<int> h0(N), h1(N);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k_make_rays, dim3((N + 255) / 256), dim3(256), 0, 0, ra, rb, N, mode);
    CK(hipDeviceSynchronize());
    printf("== mode %d (%s) rays %u spheres %d\n", mode, mode ? "incoherent" : "coherent primary", N, NS);
    hipLaunchKernelGGL(k_A, dim3(grid), dim3(256), 0, 0, ra, rb, N, d_sph, NS, ot, oi);
    CK(hipMemcpy(h0.data(), oi, (size_t)N * 4, hipMemcpyDeviceToHost));
    long hits = 0; for (uint32_t i = 0; i < N; ++i) hits += h0[i] >= 0;
    printf("   hit fraction %.3f\n", (double)hits / N);
    struct V { const char* name; void (*k)(const float4*, const float2*, uint32_t, const float4*, int, float*, int*); } vs[] = {
      {"A 1ray scalar branch", k_A}, {"C 2ray scalar", k_C}, {"E1 ballot", k_E<1>}, {"E2 ballot", k_E<2>}, {"E3 ballot", k_E<3>}, {"E4 ballot", k_E<4>}, {"E6 ballot", k_E<6>}};
    for (auto& v : vs) {
      float ms = time_ms([&] { hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, ra, rb, N, d_sph, NS, ot2, oi2); }, 5);
      CK(hipMemcpy(h1.data(), oi2, (size_t)N * 4, hipMemcpyDeviceToHost));
      long diff = 0; for (uint32_t i = 0; i < N; ++i) diff += h0[i] != h1[i];
      double tests = (double)N * NS;
      printf("   %-22s %8.3f ms  %7.2f Gtests/s  %6.2f TF(16 flop)  id diffs vs A: %ld\n", v.name, ms, tests / ms / 1e6, tests * 16 / ms / 1e9, diff);
    }
  }
  // raw throughput
  float* outf; double* outd; CK(hipMalloc(&outf, 2048 * 256 * 4 * 4)); CK(hipMalloc(&outd, 2048 * 256 * 8 * 4));
  for (int bpc : {4, 8}) {
    int g = 256 * bpc, iters = 20000;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(g), dim3(256), 0, 0, outf, iters, 0.999f); }, 3);
    printf("v_fma_f32    %d blocks/CU: %.2f TF\n", bpc, (double)g * 256 * iters * 8 * 2 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_pkfma, dim3(g), dim3(256), 0, 0, outf, iters, 0.999f); }, 3);
    printf("v_pk_fma_f32 %d blocks/CU: %.2f TF\n", bpc, (double)g * 256 * iters * 8 * 4 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_dfma, dim3(g), dim3(256), 0, 0, outd, iters, 0.999); }, 3);
    printf("v_fma_f64    %d blocks/CU: %.2f TF\n", bpc, (double)g * 256 * iters * 8 * 2 / ms / 1e9);
  }
  return 0;
}
