// Microbenchmark of k_intersect's fast path in isolation: what do the 7-FMA cull, the wave-uniform
// branch around the candidate path and the scalar operand fetch cost per (ray, sphere) test?
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o ubench_fastpath scripts/ubench_fastpath.hip
// Variants (R rays per lane):
//   N  no branch at all: count passes with a carry add
//   B  uniform branch on ballot(any) around a trivial body
//   S  like B, but the body is the product's robust recompute (12 VALU + compare) under the exec mask
// Two sphere lists: "scene" (S500-like grid, rays pass ~1 % of the tests) and "never" (k4 = 1e30).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v16f __attribute__((address_space(4))) * c_f16p;
typedef const v4f __attribute__((address_space(4))) * c_f4p;
#define AS_CONST(T, p) ((T)(uintptr_t)(p))

struct Ray { float tx, ty, tz, dx, dy, dz, o4x, o4y, o4z, od2, thr4, negE; };

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ float u01(uint32_t x) { return (x >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ Ray make_ray(uint32_t i) {
  // incoherent: origin anywhere over the field, random direction
  float ox = -11.f + 22.f * u01(hash(i * 6 + 0)), oz = -11.f + 22.f * u01(hash(i * 6 + 1)), oy = 0.05f + 1.5f * u01(hash(i * 6 + 2));
  float dx = 2.f * u01(hash(i * 6 + 3)) - 1.f, dy = 2.f * u01(hash(i * 6 + 4)) - 1.f, dz = 2.f * u01(hash(i * 6 + 5)) - 1.f;
  float il = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz + 1e-6f);
  Ray r;
  r.tx = ox; r.ty = oy; r.tz = oz; r.dx = dx * il; r.dy = dy * il; r.dz = dz * il;
  r.o4x = 4.f * ox; r.o4y = 4.f * oy; r.o4z = 4.f * oz;
  r.od2 = 2.f * (ox * r.dx + oy * r.dy + oz * r.dz);
  float oo = ox * ox + oy * oy + oz * oz;
  r.thr4 = 4.f * (oo - 1e-4f);
  r.negE = -1e-4f;
  return r;
}

template <int R, int MODE>
__global__ __launch_bounds__(256) void k_scan(const float* sphF, const float* sphR2, int n4, uint32_t nrays, uint32_t* out) {
  const uint32_t stride = gridDim.x * 256;
  uint32_t total = 0;
  for (uint32_t base = blockIdx.x * 256 + threadIdx.x; base < nrays; base += stride * R) {
    Ray P[R];
    uint32_t cnt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { P[r] = make_ray(base + r * stride); cnt[r] = 0; }
    c_f16p sp = AS_CONST(c_f16p, sphF);
    c_f4p rp = AS_CONST(c_f4p, sphR2);
    v16f cur = sp[0];
    v4f cr = rp[0];
    for (int s4 = 0; s4 < n4; ++s4) {
      v16f nxt = sp[s4 + 1];
      v4f nr = rp[s4 + 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float Cx = cur[4 * j], Cy = cur[4 * j + 1], Cz = cur[4 * j + 2], k4 = cur[4 * j + 3], r2 = cr[j];
        bool pass[R];
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float Q = fmaf(Cx, P[r].dx, fmaf(Cy, P[r].dy, fmaf(Cz, P[r].dz, P[r].od2)));
          const float CC = fmaf(Cx, P[r].o4x, fmaf(Cy, P[r].o4y, fmaf(Cz, P[r].o4z, k4)));
          const float D = fmaf(Q, Q, -CC);
          pass[r] = D >= P[r].thr4;
          any |= pass[r];
        }
        if (MODE == 0) {
#pragma unroll
          for (int r = 0; r < R; ++r) cnt[r] += pass[r] ? 1u : 0u;
        } else if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            if (MODE == 1) {
              cnt[r] += pass[r] ? 1u : 0u;
            } else if (pass[r]) {
              float ocx = fmaf(0.5f, Cx, P[r].tx), ocy = fmaf(0.5f, Cy, P[r].ty), ocz = fmaf(0.5f, Cz, P[r].tz);
              float q = fmaf(ocz, P[r].dz, fmaf(ocy, P[r].dy, ocx * P[r].dx));
              float lx = fmaf(-q, P[r].dx, ocx), ly = fmaf(-q, P[r].dy, ocy), lz = fmaf(-q, P[r].dz, ocz);
              float d = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, r2)));
              if (d >= P[r].negE) cnt[r] += 1u;
            }
          }
        }
      }
      cur = nxt;
      cr = nr;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) total += cnt[r];
  }
  if (total == 0xFFFFFFFFu) out[0] = total;   // keep the work alive
  atomicAdd(&out[1], total);
}

// T: the per-ray threshold is folded into the start of the o4 chain (no v_mov of k4 into a VGPR) and the
// R results are max-combined, so one v_cmp against the scalar k4 serves all R rays:
//   V = Q^2 - (thr4 + C.o4),  pass iff V >= k4
template <int R>
__global__ __launch_bounds__(256) void k_scan_fold(const float* sphF, const float* sphR2, int n4, uint32_t nrays, uint32_t* out) {
  const uint32_t stride = gridDim.x * 256;
  uint32_t total = 0;
  for (uint32_t base = blockIdx.x * 256 + threadIdx.x; base < nrays; base += stride * R) {
    Ray P[R];
    uint32_t cnt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { P[r] = make_ray(base + r * stride); cnt[r] = 0; }
    c_f16p sp = AS_CONST(c_f16p, sphF);
    c_f4p rp = AS_CONST(c_f4p, sphR2);
    v16f cur = sp[0];
    v4f cr = rp[0];
    for (int s4 = 0; s4 < n4; ++s4) {
      v16f nxt = sp[s4 + 1];
      v4f nr = rp[s4 + 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float Cx = cur[4 * j], Cy = cur[4 * j + 1], Cz = cur[4 * j + 2], k4 = cur[4 * j + 3], r2 = cr[j];
        float V[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float Q = fmaf(Cx, P[r].dx, fmaf(Cy, P[r].dy, fmaf(Cz, P[r].dz, P[r].od2)));
          const float CC = fmaf(Cx, P[r].o4x, fmaf(Cy, P[r].o4y, fmaf(Cz, P[r].o4z, P[r].thr4)));
          V[r] = fmaf(Q, Q, -CC);
        }
        float m = V[0];
#pragma unroll
        for (int r = 1; r < R; ++r) m = fmaxf(m, V[r]);
        if (__builtin_amdgcn_ballot_w64(m >= k4) != 0ull) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            if (V[r] >= k4) {
              float ocx = fmaf(0.5f, Cx, P[r].tx), ocy = fmaf(0.5f, Cy, P[r].ty), ocz = fmaf(0.5f, Cz, P[r].tz);
              float q = fmaf(ocz, P[r].dz, fmaf(ocy, P[r].dy, ocx * P[r].dx));
              float lx = fmaf(-q, P[r].dx, ocx), ly = fmaf(-q, P[r].dy, ocy), lz = fmaf(-q, P[r].dz, ocz);
              float d = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, r2)));
              if (d >= P[r].negE) cnt[r] += 1u;
            }
          }
        }
      }
      cur = nxt;
      cr = nr;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) total += cnt[r];
  }
  if (total == 0xFFFFFFFFu) out[0] = total;
  atomicAdd(&out[1], total);
}

// P: T with the 7 FMAs of a ray PAIR forced into v_pk_fma_f32 (the compiler packs them only sometimes; a
// v_fma_f32 with an SGPR source issues ~1.5x slower than one v_pk_fma_f32 doing two of them).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pair_disc(v2f cxy, v2f czk, v2f dx, v2f dy, v2f dz, v2f od2, v2f o4x, v2f o4y, v2f o4z, v2f thr) {
  v2f q, c, v;
  asm("v_pk_fma_f32 %0, %3, %7, %8 op_sel_hi:[0,1,1]\n\t"             // q  = Cz*dz + od2
      "v_pk_fma_f32 %1, %3, %11, %12 op_sel_hi:[0,1,1]\n\t"           // c  = Cz*o4z + thr
      "v_pk_fma_f32 %0, %4, %6, %0 op_sel:[1,0,0]\n\t"                // q += Cy*dy
      "v_pk_fma_f32 %1, %4, %10, %1 op_sel:[1,0,0]\n\t"               // c += Cy*o4y
      "v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[0,1,1]\n\t"             // q += Cx*dx
      "v_pk_fma_f32 %1, %4, %9, %1 op_sel_hi:[0,1,1]\n\t"             // c += Cx*o4x
      "v_pk_fma_f32 %2, %0, %0, %1 neg_lo:[0,0,1] neg_hi:[0,0,1]"       // v  = q*q - c
      : "=&v"(q), "=&v"(c), "=&v"(v)
      : "s"(czk), "s"(cxy), "v"(dx), "v"(dy), "v"(dz), "v"(od2), "v"(o4x), "v"(o4y), "v"(o4z), "v"(thr));
  return v;
}
// two ray pairs in one block, interleaved so that no instruction reads the result of the one before it
__device__ __forceinline__ void quad_disc(v2f cxy, v2f czk, const v2f* dx, const v2f* dy, const v2f* dz, const v2f* od2,
                                          const v2f* o4x, const v2f* o4y, const v2f* o4z, const v2f* thr, v2f* va, v2f* vb) {
  v2f q0, c0, q1, c1, v0, v1;
  asm("v_pk_fma_f32 %0, %6, %12, %14 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %2, %6, %13, %15 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %1, %6, %20, %22 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %3, %6, %21, %23 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %7, %10, %0 op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %2, %7, %11, %2 op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %1, %7, %18, %1 op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %3, %7, %19, %3 op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %0, %7, %8, %0 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %2, %7, %9, %2 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %1, %7, %16, %1 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %3, %7, %17, %3 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %4, %0, %0, %1 neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
      "v_pk_fma_f32 %5, %2, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]"
      : "=&v"(q0), "=&v"(c0), "=&v"(q1), "=&v"(c1), "=&v"(v0), "=&v"(v1)
      : "s"(czk), "s"(cxy), "v"(dx[0]), "v"(dx[1]), "v"(dy[0]), "v"(dy[1]), "v"(dz[0]), "v"(dz[1]), "v"(od2[0]), "v"(od2[1]),
        "v"(o4x[0]), "v"(o4x[1]), "v"(o4y[0]), "v"(o4y[1]), "v"(o4z[0]), "v"(o4z[1]), "v"(thr[0]), "v"(thr[1]));
  *va = v0; *vb = v1;
}
template <int R>   // R = 2 or 4
__global__ __launch_bounds__(256) void k_scan_pk(const float* sphF, const float* sphR2, int n4, uint32_t nrays, uint32_t* out) {
  const uint32_t stride = gridDim.x * 256;
  uint32_t total = 0;
  for (uint32_t base = blockIdx.x * 256 + threadIdx.x; base < nrays; base += stride * R) {
    Ray P[R];
    uint32_t cnt[R];
    v2f dx[R / 2], dy[R / 2], dz[R / 2], od2[R / 2], o4x[R / 2], o4y[R / 2], o4z[R / 2], thr[R / 2];
#pragma unroll
    for (int r = 0; r < R; ++r) { P[r] = make_ray(base + r * stride); cnt[r] = 0; }
#pragma unroll
    for (int h = 0; h < R / 2; ++h) {
      dx[h] = v2f{P[2 * h].dx, P[2 * h + 1].dx}; dy[h] = v2f{P[2 * h].dy, P[2 * h + 1].dy}; dz[h] = v2f{P[2 * h].dz, P[2 * h + 1].dz};
      od2[h] = v2f{P[2 * h].od2, P[2 * h + 1].od2}; thr[h] = v2f{P[2 * h].thr4, P[2 * h + 1].thr4};
      o4x[h] = v2f{P[2 * h].o4x, P[2 * h + 1].o4x}; o4y[h] = v2f{P[2 * h].o4y, P[2 * h + 1].o4y}; o4z[h] = v2f{P[2 * h].o4z, P[2 * h + 1].o4z};
    }
    c_f16p sp = AS_CONST(c_f16p, sphF);
    c_f4p rp = AS_CONST(c_f4p, sphR2);
    v16f cur = sp[0];
    v4f cr = rp[0];
    for (int s4 = 0; s4 < n4; ++s4) {
      v16f nxt = sp[s4 + 1];
      v4f nr = rp[s4 + 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float Cx = cur[4 * j], Cy = cur[4 * j + 1], Cz = cur[4 * j + 2], k4 = cur[4 * j + 3], r2 = cr[j];
        v2f V[R / 2];
        if (R == 2) V[0] = pair_disc(v2f{Cx, Cy}, v2f{Cz, k4}, dx[0], dy[0], dz[0], od2[0], o4x[0], o4y[0], o4z[0], thr[0]);
        else quad_disc(v2f{Cx, Cy}, v2f{Cz, k4}, dx, dy, dz, od2, o4x, o4y, o4z, thr, &V[0], &V[R / 2 - 1]);
        // max as median with +inf: v_med3_f32 is emitted as is (fmaxf would add canonicalising v_max's)
        float m = __builtin_amdgcn_fmed3f(V[0].x, V[0].y, __builtin_inff());
        if (R == 4) m = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(m, V[R / 2 - 1].x, __builtin_inff()), V[R / 2 - 1].y, __builtin_inff());
        if (__builtin_amdgcn_ballot_w64(m >= k4) != 0ull) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float v = (r & 1) ? V[r / 2].y : V[r / 2].x;
            if (v >= k4) {
              float ocx = fmaf(0.5f, Cx, P[r].tx), ocy = fmaf(0.5f, Cy, P[r].ty), ocz = fmaf(0.5f, Cz, P[r].tz);
              float q = fmaf(ocz, P[r].dz, fmaf(ocy, P[r].dy, ocx * P[r].dx));
              float lx = fmaf(-q, P[r].dx, ocx), ly = fmaf(-q, P[r].dy, ocy), lz = fmaf(-q, P[r].dz, ocz);
              float d = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, r2)));
              if (d >= P[r].negE) cnt[r] += 1u;
            }
          }
        }
      }
      cur = nxt;
      cr = nr;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) total += cnt[r];
  }
  if (total == 0xFFFFFFFFu) out[0] = total;
  atomicAdd(&out[1], total);
}

// G: one uniform branch per batch of 4 spheres: the 4 discriminants of a ray share its threshold, so their
// maximum is compared once (v_max3 + v_max + v_cmp instead of 4 v_cmp), and the scalar or/branch chain runs
// once per 4 spheres.
template <int R>
__global__ __launch_bounds__(256) void k_scan_batch(const float* sphF, const float* sphR2, int n4, uint32_t nrays, uint32_t* out) {
  const uint32_t stride = gridDim.x * 256;
  uint32_t total = 0;
  for (uint32_t base = blockIdx.x * 256 + threadIdx.x; base < nrays; base += stride * R) {
    Ray P[R];
    uint32_t cnt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { P[r] = make_ray(base + r * stride); cnt[r] = 0; }
    c_f16p sp = AS_CONST(c_f16p, sphF);
    c_f4p rp = AS_CONST(c_f4p, sphR2);
    v16f cur = sp[0];
    v4f cr = rp[0];
    for (int s4 = 0; s4 < n4; ++s4) {
      v16f nxt = sp[s4 + 1];
      v4f nr = rp[s4 + 1];
      float D[4][R];
      bool any = false;
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float Cx = cur[4 * j], Cy = cur[4 * j + 1], Cz = cur[4 * j + 2], k4 = cur[4 * j + 3];
          const float Q = fmaf(Cx, P[r].dx, fmaf(Cy, P[r].dy, fmaf(Cz, P[r].dz, P[r].od2)));
          const float CC = fmaf(Cx, P[r].o4x, fmaf(Cy, P[r].o4y, fmaf(Cz, P[r].o4z, k4)));
          D[j][r] = fmaf(Q, Q, -CC);
        }
        const float m = fmaxf(fmaxf(fmaxf(D[0][r], D[1][r]), D[2][r]), D[3][r]);
        any |= m >= P[r].thr4;
      }
      if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float Cx = cur[4 * j], Cy = cur[4 * j + 1], Cz = cur[4 * j + 2], r2 = cr[j];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            if (D[j][r] >= P[r].thr4) {
              float ocx = fmaf(0.5f, Cx, P[r].tx), ocy = fmaf(0.5f, Cy, P[r].ty), ocz = fmaf(0.5f, Cz, P[r].tz);
              float q = fmaf(ocz, P[r].dz, fmaf(ocy, P[r].dy, ocx * P[r].dx));
              float lx = fmaf(-q, P[r].dx, ocx), ly = fmaf(-q, P[r].dy, ocy), lz = fmaf(-q, P[r].dz, ocz);
              float d = fmaf(-lz, lz, fmaf(-ly, ly, fmaf(-lx, lx, r2)));
              if (d >= P[r].negE) cnt[r] += 1u;
            }
          }
        }
      }
      cur = nxt;
      cr = nr;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) total += cnt[r];
  }
  if (total == 0xFFFFFFFFu) out[0] = total;
  atomicAdd(&out[1], total);
}

template <typename F> float time_ms(F f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main(int argc, char** argv) {
  const uint32_t N = argc > 2 ? (uint32_t)atoi(argv[2]) : 1u << 23;   // small N: one wave's latency through the list
  const int NS = 500;
  int bpc = argc > 1 ? atoi(argv[1]) : 4;
  std::vector<float> scene, never, r2;
  srand(1);
  auto add = [&](std::vector<float>& v, float cx, float cy, float cz, float r) {
    v.push_back(-2 * cx); v.push_back(-2 * cy); v.push_back(-2 * cz); v.push_back(4 * (cx * cx + cy * cy + cz * cz - r * r));
  };
  for (int a = -11; a < 11; ++a) for (int b = -11; b < 11; ++b) {
    if ((int)scene.size() / 4 >= NS) break;
    add(scene, a + 0.9f * rand() / RAND_MAX, 0.2f, b + 0.9f * rand() / RAND_MAX, 0.2f); r2.push_back(0.04f);
  }
  while ((int)scene.size() / 4 < NS) { add(scene, 4.f * (scene.size() % 3) - 4.f, 1.f, 0.f, 1.f); r2.push_back(1.f); }
  for (int k = 0; k < 8; ++k) { scene.push_back(0); scene.push_back(0); scene.push_back(0); scene.push_back(1e30f); r2.push_back(0); }
  for (int k = 0; k < NS + 8; ++k) { never.push_back(0.1f * k); never.push_back(0.2f); never.push_back(-0.1f * k); never.push_back(1e30f); }
  float *d_scene, *d_never, *d_r2; uint32_t* d_out;
  CK(hipMalloc(&d_scene, scene.size() * 4)); CK(hipMemcpy(d_scene, scene.data(), scene.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_never, never.size() * 4)); CK(hipMemcpy(d_never, never.data(), never.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_r2, r2.size() * 4)); CK(hipMemcpy(d_r2, r2.data(), r2.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_out, 8)); CK(hipMemset(d_out, 0, 8));
  const int n4 = NS / 4;
  int grid = 256 * bpc;
  if ((uint32_t)grid * 256u > N) grid = (int)((N + 255) / 256);
  printf("rays %u spheres %d blocks/CU %d\n", N, NS, bpc);
  struct V { const char* name; void (*k)(const float*, const float*, int, uint32_t, uint32_t*); int R; } vs[] = {
    {"N1 no branch", k_scan<1, 0>, 1}, {"N2 no branch", k_scan<2, 0>, 2}, {"N4 no branch", k_scan<4, 0>, 4},
    {"B1 branch, trivial body", k_scan<1, 1>, 1}, {"B2 branch, trivial body", k_scan<2, 1>, 2}, {"B4 branch, trivial body", k_scan<4, 1>, 4},
    {"S1 branch, robust body", k_scan<1, 2>, 1}, {"S2 branch, robust body", k_scan<2, 2>, 2}, {"S4 branch, robust body", k_scan<4, 2>, 4},
    {"G1 batch branch, robust", k_scan_batch<1>, 1}, {"G2 batch branch, robust", k_scan_batch<2>, 2}, {"G3 batch branch, robust", k_scan_batch<3>, 3}, {"G4 batch branch, robust", k_scan_batch<4>, 4},
    {"T1 folded thr, robust", k_scan_fold<1>, 1}, {"T2 folded thr, robust", k_scan_fold<2>, 2}, {"T3 folded thr, robust", k_scan_fold<3>, 3}, {"T4 folded thr, robust", k_scan_fold<4>, 4},
    {"P2 folded thr, forced pk", k_scan_pk<2>, 2}, {"P4 folded thr, forced pk", k_scan_pk<4>, 4}};
  for (int list = 0; list < 2; ++list) {
    const float* L = list ? d_never : d_scene;
    printf("== list %s\n", list ? "never" : "scene");
    for (auto& v : vs) {
      CK(hipMemset(d_out, 0, 8));
      float ms = time_ms([&] { hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, L, d_r2, n4, N, d_out); }, 5);
      uint32_t h[2]; CK(hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost));
      double tests = (double)N * NS;
      printf("   %-26s %8.3f ms  %7.2f Gtests/s  %6.2f TF(16 flop)  pass/ray %.2f\n", v.name, ms, tests / ms / 1e6,
             tests * 16 / ms / 1e9, (double)h[1] / 6 / N);
    }
  }
  return 0;
}
