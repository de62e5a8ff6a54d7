// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the LMD/LMD+
// stage-2 denoising path.  No CUDA compatibility layer: this code only targets
// MI355X.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LGD_WAVE 64

// Error codes returned through the C ABI (0 = ok).
#define LGD_OK 0
#define LGD_ERR_ARG (-1)
#define LGD_ERR_LAUNCH (-2)
#define LGD_ERR_UNSUPPORTED (-3)

static inline int lgd_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "[lgd_hip] launch error: %s (%d)\n", hipGetErrorString(e), (int)e);
    return LGD_ERR_LAUNCH;
  }
  return LGD_OK;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0] + red[1] + red[2] + red[3];
  return r;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
// d/dx silu(x) = s + x*s*(1-s), s = sigmoid(x)
__device__ __forceinline__ float silu_grad_f(float x) {
  float s = 1.f / (1.f + __expf(-x));
  return s * (1.f + x * (1.f - s));
}
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the fp16 resolution of every
// consumer): ~14 VALU ops instead of the device library's branchy erff — the GEGLU epilogue of the
// K = 320 feed-forward GEMM evaluates it once per output element and was VALU-bound on it.
__device__ __forceinline__ float erf_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.f - p * t * __expf(-ax * ax);
  return copysignf(e, x);
}
// GELU, erf form (torch.nn.functional.gelu default; attention.py:286-289 GEGLU, SAM's MLP), round 4: x * Phi(x) with
// Phi(x) = 0.5 + xc * P(2 xc^2 / 20.25 - 1), xc = clamp(x, -4.5, 4.5), P = degree-10 least-squares fit on Chebyshev nodes
// (coefficients <= 0.16 in magnitude: Horner in fp32 is well conditioned).  Max abs error of gelu against the erf form
// 1.2e-5 over [-6, 6] in fp32 arithmetic (beyond the clamp Phi is 1 - 3.4e-6 / 3.4e-6) — 40x below the fp16 resolution of
// the outputs it feeds.  No transcendental (the A&S erf above costs a v_rcp_f32 and a v_exp_f32, quarter rate each), and
// on PAIRS of values every step is one packed-fp32 instruction (v_pk_mul_f32 / v_pk_fma_f32): 8.5 VALU instructions per
// output in the GEGLU epilogue instead of ~24 issue slots — that epilogue's VALU work runs in series with the MFMAs of
// the short-K feed-forward GEMMs.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu2_f(f32x2_t x) {
  f32x2_t xc;
  xc[0] = __builtin_amdgcn_fmed3f(x[0], -4.5f, 4.5f);
  xc[1] = __builtin_amdgcn_fmed3f(x[1], -4.5f, 4.5f);
  const f32x2_t s = __builtin_elementwise_fma(xc * xc, (f32x2_t){2.f / 20.25f, 2.f / 20.25f}, (f32x2_t){-1.f, -1.f});
  constexpr float C[11] = {1.569049306e-01f, -7.719386027e-02f, 5.470118655e-02f, -4.010922254e-02f, 2.828396914e-02f,
                           -1.902089461e-02f, 1.143910392e-02f, -5.251618182e-03f, 2.716435037e-03f, -2.339980905e-03f,
                           9.806225403e-04f};
  f32x2_t p = {C[10], C[10]};
#pragma unroll
  for (int k = 9; k >= 0; --k) p = __builtin_elementwise_fma(p, s, (f32x2_t){C[k], C[k]});
  const f32x2_t phi = __builtin_elementwise_fma(xc, p, (f32x2_t){0.5f, 0.5f});
  return x * phi;
}
// scalar form on the A&S erf (1.5e-7): the element-wise kernels of the GRAD plans (geglu_fwd / geglu_bwd, whose backward
// differentiates exactly this function) and SAM's MLP keep it; only the fused GEGLU epilogue of the GEMM uses gelu2_f
__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.f + erf_f(x * 0.70710678118654752f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf = 0.5f * (1.f + erf_f(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
