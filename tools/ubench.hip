// GPU-box tool: instruction-rate microbenchmarks that decide the attention kernel design (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o llm-groundeddiffusion_amd/build/ubench
// Each kernel runs `iters` iterations of an unrolled block of independent instructions in every wave of a
// 256-CU x (waves/SIMD) grid and reports cycles per wave-instruction per SIMD (s_memtime around the loop).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define N_INST 32

template <int MODE>
__global__ void k_valu(float* out, long* cyc, int iters) {
  float x[N_INST];
#pragma unroll
  for (int i = 0; i < N_INST; ++i) x[i] = -0.001f * (threadIdx.x + i + 1);
  long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < N_INST; ++i) {
      if (MODE == 0) x[i] = __builtin_amdgcn_exp2f(x[i]);                 // v_exp_f32
      if (MODE == 1) x[i] = fmaf(x[i], 1.0001f, -0.5f);                    // v_fma_f32
      if (MODE == 2) asm volatile("v_exp_f16 %0, %0" : "+v"(x[i]));       // v_exp_f16 (low half)
      if (MODE == 3) x[i] = __builtin_amdgcn_rcpf(x[i]);                   // v_rcp_f32
    }
  }
  long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < N_INST; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// exp interleaved with fma 1:K to see whether the transcendental unit runs beside the main VALU
template <int K>
__global__ void k_mix(float* out, long* cyc, int iters) {
  float x[8], y[8 * K];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = -0.001f * (threadIdx.x + i + 1);
#pragma unroll
  for (int i = 0; i < 8 * K; ++i) y[i] = 0.001f * (threadIdx.x + i + 1);
  long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x[i] = __builtin_amdgcn_exp2f(x[i]);
#pragma unroll
      for (int j = 0; j < K; ++j) y[i * K + j] = fmaf(y[i * K + j], 1.0001f, -0.5f);
    }
  }
  long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 8 * K; ++i) s += y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
__global__ void k_mfma(float* out, long* cyc, int iters) {
  half8_t a8, b8;
  half4_t a4, b4;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.01f * (threadIdx.x % 7 + i)); b8[i] = (_Float16)(0.02f * (threadIdx.x % 5 + i)); }
#pragma unroll
  for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
  f32x4 c[8];
  f32x16 cc[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) cc[i][j] = 0.f;
  long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[i], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[i], 0, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) cc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, cc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) cc[i] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, cc[i], 0, 0, 0);
    }
  }
  long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += c[i][0];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += cc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// cross-lane exchange used by the softmax row max: ds_bpermute vs permlane swaps vs DPP
template <int MODE>
__global__ void k_xlane(float* out, long* cyc, int iters) {
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = 0.001f * (threadIdx.x + i + 1);
  const int idx = ((threadIdx.x & 63) ^ 16) << 2;
  long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) {
        float o = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx, __builtin_bit_cast(int, x[i])));
        x[i] = fmaxf(x[i], o) * 1.0001f;
      } else if (MODE == 1) {
        unsigned a = __builtin_bit_cast(unsigned, x[i]), b = a;
        auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        x[i] = fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1])) * 1.0001f;
      } else if (MODE == 2) {
        unsigned a = __builtin_bit_cast(unsigned, x[i]), b = a;
        auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        x[i] = fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1])) * 1.0001f;
      }
    }
  }
  long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// Can one SIMD run MFMA of one wave beside VALU (exp + fma) of another?  512-thread workgroups = 2 waves per SIMD.
//   MODE 0: all 8 waves MFMA only     MODE 1: all 8 waves VALU only
//   MODE 2: waves 0-3 MFMA, waves 4-7 VALU (ping-pong roles)     MODE 3: every wave both, interleaved by the compiler
//   MODE 4: as 2 with s_setprio 1 on the MFMA waves
// One iteration = 28 MFMA 16x16x32 (one attention key tile of a 32-query wave at d = 40) and / or 32 v_exp + 96 v_fma.
template <int MODE, int SHAPE = 0>
__global__ __launch_bounds__(512) void k_pp(float* out, long* cyc, int iters) {
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  half8_t a8, b8;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.01f * (threadIdx.x % 7 + i)); b8[i] = (_Float16)(0.02f * (threadIdx.x % 5 + i)); }
  f32x4 c[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x16 cc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) cc[i][j] = 0.f;
  float x[32], y[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) { x[i] = -0.001f * (threadIdx.x + i + 1); y[i] = 0.001f * (i + 1); }
  const bool do_m = MODE == 0 || MODE == 3 || ((MODE == 2 || MODE == 4) && wid < 4);
  const bool do_v = MODE == 1 || MODE == 3 || ((MODE == 2 || MODE == 4) && wid >= 4);
  if (MODE == 4 && wid < 4) __builtin_amdgcn_s_setprio(1);
  long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
      if (SHAPE == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 7; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[i], 0, 0, 0);
      } else {       // the same flops as 14 x 32x32x16
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
          for (int i = 0; i < 2; ++i) cc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, cc[i], 0, 0, 0);
      }
    }
    if (do_v) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        x[i] = __builtin_amdgcn_exp2f(x[i]);
        y[i] = fmaf(y[i], 1.0001f, -0.5f);
        y[i] = fmaf(y[i], 0.9999f, 0.5f);
        y[i] = fmaf(y[i], 1.0001f, -0.25f);
      }
    }
  }
  long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 7; ++i) s += c[i][0];
  s += cc[0][0] + cc[1][3];
#pragma unroll
  for (int i = 0; i < 32; ++i) s += x[i] + y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename F>
static void run_pp(const char* name, F launch) {
  const int iters = 4000, blocks = 256, threads = 512;
  float* out; long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(long) * blocks);
  launch(blocks, threads, out, cyc, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  launch(blocks, threads, out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %8.1f ns per loop iteration (8 waves/CU)\n", name, ms * 1e6 / iters);
  hipFree(out); hipFree(cyc);
}

template <typename F>
static void run(const char* name, F launch, int inst_per_iter, int waves_per_simd, double flops_per_inst) {
  const int iters = 2000;
  const int blocks = 256;
  const int threads = 64 * 4 * waves_per_simd;
  float* out; long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(long) * blocks);
  launch(blocks, threads, out, cyc, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  launch(blocks, threads, out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
  // s_memtime ticks at a fixed 100 MHz on this part; derive cycles from wall time instead, assume 2.4 GHz max
  const double n_inst_simd = (double)iters * inst_per_iter * waves_per_simd;   // wave-instructions per SIMD
  const double ns_per_inst = ms * 1e6 / n_inst_simd;
  printf("%-34s waves/SIMD %d: %7.2f ns per wave-instr per SIMD  (%.1f clk @2.4GHz)  memtime ticks/iter %.1f", name,
         waves_per_simd, ns_per_inst, ns_per_inst * 2.4, avg / iters);
  if (flops_per_inst > 0) printf("  chip %.0f TF/s", flops_per_inst * 1024.0 / ns_per_inst / 1e3);
  printf("\n");
  hipFree(out); hipFree(cyc);
}

#define L(K) [](int b, int t, float* o, long* c, int it) { hipLaunchKernelGGL(K, dim3(b), dim3(t), 0, 0, o, c, it); }

int main() {
  run_pp("ping-pong: all 8 waves 28 MFMA", L(k_pp<0>));
  run_pp("ping-pong: all 8 waves 32 exp + 96 fma", L(k_pp<1>));
  run_pp("ping-pong: waves 0-3 MFMA | waves 4-7 VALU", L(k_pp<2>));
  run_pp("ping-pong: every wave MFMA + VALU (compiler interleave)", L(k_pp<3>));
  run_pp("ping-pong: waves 0-3 MFMA (prio 1) | waves 4-7 VALU", L(k_pp<4>));
  run_pp("32x32x16: all 8 waves 14 MFMA", L((k_pp<0, 1>)));
  run_pp("32x32x16: waves 0-3 MFMA | waves 4-7 VALU", L((k_pp<2, 1>)));
  run_pp("32x32x16: every wave MFMA + VALU (compiler interleave)", L((k_pp<3, 1>)));
  run_pp("32x32x16: waves 0-3 MFMA (prio 1) | waves 4-7 VALU", L((k_pp<4, 1>)));
  for (int w : {1, 2, 4}) {
    run("v_exp_f32", L(k_valu<0>), N_INST, w, 0);
    run("v_fma_f32", L(k_valu<1>), N_INST, w, 0);
    run("v_exp_f16", L(k_valu<2>), N_INST, w, 0);
    run("v_rcp_f32", L(k_valu<3>), N_INST, w, 0);
    run("8 x (v_exp_f32 + 1 v_fma)", L(k_mix<1>), 8 * 2, w, 0);
    run("8 x (v_exp_f32 + 3 v_fma)", L(k_mix<3>), 8 * 4, w, 0);
    run("8 x (v_exp_f32 + 6 v_fma)", L(k_mix<6>), 8 * 7, w, 0);
  }
  for (int w : {1, 2}) {
    run("mfma_f32_16x16x32_f16", L(k_mfma<0>), 8, w, 2.0 * 16 * 16 * 32);
    run("mfma_f32_16x16x16_f16", L(k_mfma<1>), 8, w, 2.0 * 16 * 16 * 16);
    run("mfma_f32_32x32x16_f16", L(k_mfma<2>), 4, w, 2.0 * 32 * 32 * 16);
    run("mfma_f32_32x32x8_f16", L(k_mfma<3>), 4, w, 2.0 * 32 * 32 * 8);
  }
  for (int w : {1, 4}) {
    run("ds_bpermute + max + mul", L(k_xlane<0>), 8, w, 0);
    run("permlane32_swap + max + mul", L(k_xlane<1>), 8, w, 0);
    run("permlane16_swap + max + mul", L(k_xlane<2>), 8, w, 0);
  }
  return 0;
}
