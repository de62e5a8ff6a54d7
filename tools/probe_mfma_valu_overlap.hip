// MFMA / VALU overlap inside ONE wave per SIMD: per loop iteration 8 x { v_mfma_f32_32x32x16_f16 , N fillers } ;
// variants: accumulator file (AGPR / arch VGPR), filler kind (v_fma_f32 / v_exp_f32 / v_cvt_pk / v_max3), N = 0..8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FILL_FMA(x) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
#define FILL_EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
#define FILL_MAX(x) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x));
#define FILL_CVT(x) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(x));

template <int ACC, int KIND, int N>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  half8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = 0.001f * (threadIdx.x + i);
  asm volatile("" : "+v"(a), "+v"(b));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (ACC == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[m & 3]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[m & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int n = 0; n < N; ++n) {
        if (KIND == 0) { FILL_FMA(f[n]) } else if (KIND == 1) { FILL_EXP(f[n]) } else if (KIND == 2) { FILL_MAX(f[n]) } else { FILL_CVT(f[n]) }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += f[i];
  for (int i = 0; i < 4; ++i) s += c[i][0];
  if (s == 12345.678f) out[0] = s;
}
template <int ACC, int KIND, int N>
void run(float* d, const char* an, const char* kn) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<ACC, KIND, N><<<256, 256>>>(d, 10);
  hipEventRecord(e0);
  k<ACC, KIND, N><<<256, 256>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("acc %s filler %-8s N=%d: %7.2f ns per MFMA slot\n", an, kn, N, ms * 1e6 / (iters * 8.0));
}
template <int ACC, int KIND>
void sweep(float* d, const char* an, const char* kn) {
  run<ACC, KIND, 0>(d, an, kn); run<ACC, KIND, 1>(d, an, kn); run<ACC, KIND, 2>(d, an, kn); run<ACC, KIND, 3>(d, an, kn);
  run<ACC, KIND, 4>(d, an, kn); run<ACC, KIND, 5>(d, an, kn); run<ACC, KIND, 6>(d, an, kn); run<ACC, KIND, 8>(d, an, kn);
}
int main() {
  float* d; hipMalloc(&d, 64);
  sweep<0, 0>(d, "AGPR", "fma"); sweep<1, 0>(d, "VGPR", "fma");
  sweep<0, 1>(d, "AGPR", "exp"); sweep<1, 1>(d, "VGPR", "exp");
  sweep<0, 2>(d, "AGPR", "max3"); sweep<0, 3>(d, "AGPR", "cvt_pk");
  return 0;
}
