// Issue-rate probes for the instruction forms of the per-ray fixed work (gfx950): Philox's 32x32->64 multiplies, f64
// arithmetic of the exact evaluation, transcendentals.  Same frame as ubench_valu.hip.
//   hipcc -O3 --offload-arch=gfx950 -o ubench_valu2 scripts/ubench_valu2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters, float s, unsigned u) {
  float a0 = (float)threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  unsigned i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  unsigned long long l0 = i0, l1 = i1, l2 = i2, l3 = i3, l4 = i4, l5 = i5, l6 = i6, l7 = i7;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  const double ds = s;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#define X(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##k) : "v"(s), "v"(s));
      REP8(X)
#undef X
    } else if (MODE == 1) {
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(l##k) : "v"(i##k), "v"(u) : "vcc");
      REP8(X)
#undef X
    } else if (MODE == 2) {
#define X(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i##k) : "v"(u));
      REP8(X)
#undef X
    } else if (MODE == 3) {
#define X(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(i##k) : "v"(u));
      REP8(X)
#undef X
    } else if (MODE == 4) {
#define X(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(i##k) : "v"(u));
      REP8(X)
#undef X
    } else if (MODE == 5) {
#define X(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(i##k) : "v"(u));
      REP8(X)
#undef X
    } else if (MODE == 6) {
#define X(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##k) : "v"(ds));
      REP8(X)
#undef X
    } else if (MODE == 7) {
#define X(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##k) : "v"(ds));
      REP8(X)
#undef X
    } else if (MODE == 8) {
#define X(k) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d##k) : "v"(ds));
      REP8(X)
#undef X
    } else if (MODE == 9) {
#define X(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##k));
      REP8(X)
#undef X
    } else if (MODE == 10) {
#define X(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##k));
      REP8(X)
#undef X
    } else if (MODE == 11) {
#define X(k) asm volatile("v_rsq_f64 %0, %0" : "+v"(d##k));
      REP8(X)
#undef X
    } else if (MODE == 12) {
#define X(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(d##k));
      REP8(X)
#undef X
    } else if (MODE == 13) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i##k) : "v"(u) : "vcc");
      REP8(X)
#undef X
    } else if (MODE == 14) {
#define X(k) asm volatile("v_sin_f32 %0, %0" : "+v"(a##k));
      REP8(X)
#undef X
    } else if (MODE == 15) {
#define X(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d##k) : "v"(a##k));
      REP8(X)
#undef X
    }
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(i0 ^ i1 ^ i2 ^ i3 ^ i4 ^ i5 ^ i6 ^ i7) +
            (float)(l0 ^ l1 ^ l2 ^ l3 ^ l4 ^ l5 ^ l6 ^ l7) + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  out[blockIdx.x * 256 + threadIdx.x] = r;
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
  float* out; CK(hipMalloc(&out, 256 * 16 * 256 * 4));
  const int iters = 10000, g = 256 * bpc;
  struct V { const char* name; void (*k)(float*, int, float, unsigned); } vs[] = {
    {"v_fma_f32", k_probe<0>}, {"v_mad_u64_u32", k_probe<1>}, {"v_mul_lo_u32", k_probe<2>}, {"v_mul_hi_u32", k_probe<3>},
    {"v_mul_u32_u24", k_probe<4>}, {"v_xor_b32", k_probe<5>}, {"v_add_f64", k_probe<6>}, {"v_mul_f64", k_probe<7>},
    {"v_fma_f64", k_probe<8>}, {"v_sqrt_f32", k_probe<9>}, {"v_rcp_f32", k_probe<10>}, {"v_rsq_f64", k_probe<11>},
    {"v_rcp_f64", k_probe<12>}, {"v_cndmask_b32", k_probe<13>}, {"v_sin_f32", k_probe<14>}, {"v_cvt_f64_f32", k_probe<15>}};
  float base = 0.f;
  printf("blocks/CU %d; cost relative to v_fma_f32 (all-VGPR)\n", bpc);
  for (auto& v : vs) {
    float ms = time_ms([&] { hipLaunchKernelGGL(v.k, dim3(g), dim3(256), 0, 0, out, iters, 0.999f, 0xD2511F53u); }, 3);
    if (base == 0.f) base = ms;
    printf("  %-16s %8.3f ms   x%.2f\n", v.name, ms, ms / base);
  }
  return 0;
}
