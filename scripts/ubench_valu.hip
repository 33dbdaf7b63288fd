// Raw VALU issue-rate probes for the instruction forms k_intersect's fast path uses (gfx950).
//   hipcc -O3 --offload-arch=gfx950 -o ubench_valu scripts/ubench_valu.hip
// Prints wave-instructions per cycle per SIMD assuming CLK_GHZ (argument 2, default 2.0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// 0: v_fma_f32 vgpr   1: v_fma_f32 sgpr src0   2: v_pk_fma vgpr   3: v_pk_fma sgpr-pair src0 (both halves)
// 4: v_pk_fma sgpr-pair src0 broadcast low (op_sel_hi:[0,1,1])   5: v_cmp_ge_f32_e64 -> sgpr pair
// 6: v_cmp_ge_f32_e32 -> vcc   7: mix as in the product: 7 pk_fma(sgpr bcast) + v_mov + 2 v_cmp_e64 + s_or
template <int MODE>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters, float s, v2f sp) {
  v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  v2f x = {s, s * 0.5f};
  unsigned long long m = 0;
  float f0 = s * 3.f, f1 = s * 5.f;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#define X(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##k.x) : "v"(x.x), "v"(x.y));
      REP8(X)
#undef X
    } else if (MODE == 1) {
#define X(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##k.x) : "s"(sp.x), "v"(x.y));
      REP8(X)
#undef X
    } else if (MODE == 2) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a##k) : "v"(x), "v"(x));
      REP8(X)
#undef X
    } else if (MODE == 3) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a##k) : "s"(sp), "v"(x));
      REP8(X)
#undef X
    } else if (MODE == 4) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a##k) : "s"(sp), "v"(x));
      REP8(X)
#undef X
    } else if (MODE == 5) {
#define X(k) { unsigned long long t; asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(t) : "v"(a##k.x), "v"(x.x)); m |= t; }
      REP8(X)
#undef X
    } else if (MODE == 6) {
#define X(k) asm volatile("v_cmp_ge_f32_e32 vcc, %0, %1" : : "v"(a##k.x), "v"(x.x) : "vcc");
      REP8(X)
#undef X
    } else if (MODE == 7) {
      unsigned long long t0, t1;
      asm volatile(
          "v_mov_b32 %2, %9\n"
          "v_pk_fma_f32 %0, %6, %7, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, %6, %8, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %0, %6, %7, %0 op_sel:[1,0,0]\n"
          "v_pk_fma_f32 %1, %6, %8, %1 op_sel:[1,0,0]\n"
          "v_pk_fma_f32 %0, %6, %7, %0 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %1, %6, %8, %1 op_sel_hi:[0,1,1]\n"
          "v_pk_fma_f32 %0, %0, %0, %1 neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
          "s_nop 0\n"
          "v_cmp_ge_f32_e64 %3, %10, %11\n"
          "v_cmp_ge_f32_e64 %4, %11, %10\n"
          "s_or_b64 %5, %3, %4\n"
          : "+v"(a0), "+v"(a1), "=v"(a2.x), "=s"(t0), "=s"(t1), "=s"(m)
          : "s"(sp), "v"(x), "v"(a3), "s"(sp.y), "v"(f0), "v"(f1));
    }
  }
  v2f r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y + (float)(m & 1);
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
  int bpc = argc > 1 ? atoi(argv[1]) : 8;
  double ghz = argc > 2 ? atof(argv[2]) : 2.0;
  float* out; CK(hipMalloc(&out, 256 * 16 * 256 * 4));
  const int iters = 20000, g = 256 * bpc;
  v2f sp = {0.999f, 1.001f};
  struct V { const char* name; void (*k)(float*, int, float, v2f); int per_iter; } vs[] = {
    {"v_fma_f32 vgpr", k_probe<0>, 8}, {"v_fma_f32 sgpr src0", k_probe<1>, 8}, {"v_pk_fma_f32 vgpr", k_probe<2>, 8},
    {"v_pk_fma_f32 sgpr pair", k_probe<3>, 8}, {"v_pk_fma_f32 sgpr bcast", k_probe<4>, 8},
    {"v_cmp_ge_f32_e64 -> sgpr", k_probe<5>, 8}, {"v_cmp_ge_f32_e32 -> vcc", k_probe<6>, 8},
    {"product mix (10 VALU + s_nop + s_or)", k_probe<7>, 1}};
  printf("blocks/CU %d, assuming %.2f GHz\n", bpc, ghz);
  for (auto& v : vs) {
    float ms = time_ms([&] { hipLaunchKernelGGL(v.k, dim3(g), dim3(256), 0, 0, out, iters, 0.999f, sp); }, 3);
    double wave_instr = (double)g * 4 * iters * v.per_iter;            // per launch
    double per_simd_per_s = wave_instr / (ms * 1e-3) / 1024.0;
    printf("  %-40s %8.3f ms   %6.2f cycles per %s per SIMD\n", v.name, ms, ghz * 1e9 / per_simd_per_s,
           v.per_iter == 1 ? "group" : "instruction");
  }
  return 0;
}
