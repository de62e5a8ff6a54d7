// Small HBM/latency-bound pieces of the denoising loop: boundary convolutions (4 <-> C channels,
// NCHW fp32 <-> channels-last fp16), elementwise gradient helpers, and the fused per-step update
// (classifier-free guidance + DDIM + frozen-mask blend).
#include "common.h"
#include "../../include/lgd_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------
// conv_in: NCHW fp32 (B,Cin<=8,L,L) -> [B][L*L][Cout] fp16.   w: [Cout][9*Cin] (ky,kx,ci).
// Workgroup = 32 pixels; weights transposed into LDS as [k][Cout] so that lanes run over output
// channels (coalesced stores, conflict-free LDS reads); the 9*Cin input patch of a pixel is
// broadcast.
// ---------------------------------------------------------------------------------------------
constexpr int CI_PIX = 32;
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x,
                                                       const half_t* __restrict__ w,
                                                       const float* __restrict__ bias,
                                                       half_t* __restrict__ y, int B, int Cin,
                                                       int L, int Cout) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  const int K = 9 * Cin;
  half_t* wT = reinterpret_cast<half_t*>(dyn_smem);                 // [K][Cout]
  float* patch = reinterpret_cast<float*>(dyn_smem + (size_t)((K * Cout * 2 + 15) & ~15));  // [32][K]
  const int HW = L * L;
  const long pix0 = (long)blockIdx.x * CI_PIX;
  for (int i = threadIdx.x; i < K * Cout; i += 256) {
    int co = i / K, k = i - co * K;
    wT[k * Cout + co] = w[i];
  }
  for (int i = threadIdx.x; i < CI_PIX * K; i += 256) {
    int p = i / K, k = i - p * K;
    long pix = pix0 + p;
    float v = 0.f;
    if (pix < (long)B * HW) {
      int b = (int)(pix / HW), rem = (int)(pix - (long)b * HW);
      int oy = rem / L, ox = rem - oy * L;
      int tap = k / Cin, ci = k - tap * Cin;
      int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
      if (iy >= 0 && iy < L && ix >= 0 && ix < L) v = x[(((long)b * Cin + ci) * L + iy) * L + ix];
    }
    patch[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int pp = wid; pp < CI_PIX; pp += 4) {
    long pix = pix0 + pp;
    if (pix >= (long)B * HW) break;
    const float* pt = patch + pp * K;
    for (int co = lane; co < Cout; co += 64) {
      float acc = bias ? bias[co] : 0.f;
      for (int k = 0; k < K; ++k) acc += pt[k] * (float)wT[k * Cout + co];
      y[pix * Cout + co] = (half_t)acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// conv_out: [B][L*L][Cin] fp16 -> NCHW fp32 (B,Cout<=8,L,L), 3x3 pad 1.  w: [Cout][9*Cin].
// One wave per output pixel; the 9*Cin reduction is spread over the lanes in 8-channel vectors and
// closed with shuffles.  Also used as conv_in's input gradient with flipped/transposed weights.
// ---------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void conv_out_kernel(const half_t* __restrict__ x,
                                                        const half_t* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ y, int B, int Cin, int L,
                                                        float out_scale) {
  const int lane = threadIdx.x & 63;
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int HW = L * L;
  if (pix >= (long)B * HW) return;
  const int b = (int)(pix / HW), rem = (int)(pix - (long)b * HW);
  const int oy = rem / L, ox = rem - oy * L;
  const int vpt = Cin / 8;  // vectors per tap
  const int nvec = 9 * vpt;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
  for (int v = lane; v < nvec; v += 64) {
    int tap = v / vpt, cv = v - tap * vpt;
    int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
    if (iy < 0 || iy >= L || ix < 0 || ix >= L) continue;
    half8_t hx = *reinterpret_cast<const half8_t*>(x + ((long)b * HW + iy * L + ix) * Cin + cv * 8);
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
      half8_t hw = *reinterpret_cast<const half8_t*>(w + (long)c * 9 * Cin + tap * Cin + cv * 8);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)hx[e] * (float)hw[e];
      acc[c] += s;
    }
  }
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = wave_sum(acc[c]);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < COUT; ++c)
      y[(((long)b * COUT + c) * L + oy) * L + ox] = (acc[c] + (bias ? bias[c] : 0.f)) * out_scale;
  }
}

// NCHW fp32 (B, C <= 8, HW) -> channels-last fp16 [B*HW][8]: channels 0..C-1 = fp16(x), C..2C-1 = fp16(x - fp16(x))
// when they fit, the rest zero: the 4-channel latents become an
// 8-channel map, the narrowest the implicit-GEMM convolution takes (one 16-byte vector per pixel and tap), so that
// conv_in runs on the matrix cores (K = 72) instead of conv_in_kernel's LDS-bound scalar loop.
__global__ __launch_bounds__(256) void nchw_to_nhwc8_kernel(const float* __restrict__ x, half_t* __restrict__ y,
                                                             int C, long HW, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const long b = i / HW, p = i - b * HW;
    half8_t o = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < C) {
        const float v = x[(b * C + c) * HW + p];
        const half_t hi = (half_t)v;
        o[c] = hi;
        // the spare channels carry the rounding remainder (the filter is duplicated over them by the caller):
        // hi + lo reproduces the fp32 latent to 2^-22, so the fp16 operand format costs conv_in no input precision
        if (2 * C <= 8 && c + C < 8) o[c + C] = (half_t)(v - (float)hi);
      }
    reinterpret_cast<half8_t*>(y)[i] = o;
  }
}

// ---------------------------------------------------------------------------------------------
// elementwise helpers (16-byte vectors, grid-stride)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_kernel(const half_t* a, const half_t* b, half_t* y,
                                                   long nvec) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += gridDim.x * 256L) {
    half8_t x0 = reinterpret_cast<const half8_t*>(a)[i];
    half8_t x1 = reinterpret_cast<const half8_t*>(b)[i];
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)x0[e] + (float)x1[e]);
    reinterpret_cast<half8_t*>(y)[i] = o;
  }
}
__global__ __launch_bounds__(256) void scale_kernel(const half_t* a, half_t* y, float alpha,
                                                     long nvec) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += gridDim.x * 256L) {
    half8_t x0 = reinterpret_cast<const half8_t*>(a)[i];
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)x0[e] * alpha);
    reinterpret_cast<half8_t*>(y)[i] = o;
  }
}

// y = x * sigmoid(1.702 x): the "quick GELU" of the CLIP text encoder's MLP ([ext] transformers CLIPMLP).
__global__ __launch_bounds__(256) void quick_gelu_kernel(const half_t* a, half_t* y, long nvec) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += gridDim.x * 256L) {
    half8_t x0 = reinterpret_cast<const half8_t*>(a)[i];
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (float)x0[e];
      o[e] = (half_t)(x / (1.f + __expf(-1.702f * x)));
    }
    reinterpret_cast<half8_t*>(y)[i] = o;
  }
}

// GEGLU forward on a stored pre-activation (the grad-enabled guidance pass keeps h for backward;
// the no-grad pass uses the fused GEMM epilogue instead).  h packed [16 value | 16 gate] blocks.
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const half_t* __restrict__ h,
                                                         half_t* __restrict__ y, long rows, int n) {
  const long nvec_row = n / 8;
  const long total = rows * nvec_row;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    long r = i / nvec_row;
    int j = (int)(i - r * nvec_row) * 8;
    long base = r * 2L * n + (j / 16) * 32 + (j % 16);
    half8_t v = *reinterpret_cast<const half8_t*>(h + base);
    half8_t g = *reinterpret_cast<const half8_t*>(h + base + 16);
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)v[e] * gelu_f((float)g[e]));
    *reinterpret_cast<half8_t*>(y + r * (long)n + j) = o;
  }
}

// GEGLU backward.  h packed as [.. 16 value | 16 gate ..] blocks (the layout the GEMM epilogue
// consumes); y[j] = v[j]*gelu(g[j]);  gv = gy*gelu(g), gg = gy*v*gelu'(g).
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const half_t* __restrict__ h,
                                                         const half_t* __restrict__ gy,
                                                         half_t* __restrict__ gh, long rows, int n) {
  const long nvec_row = n / 8;
  const long total = rows * nvec_row;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    long r = i / nvec_row;
    int j = (int)(i - r * nvec_row) * 8;           // output column (multiple of 8, inside a 16-block)
    long base = r * 2L * n + (j / 16) * 32 + (j % 16);
    half8_t v = *reinterpret_cast<const half8_t*>(h + base);
    half8_t g = *reinterpret_cast<const half8_t*>(h + base + 16);
    half8_t dy = *reinterpret_cast<const half8_t*>(gy + r * (long)n + j);
    half8_t ov, og;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float gf = (float)g[e], d = (float)dy[e];
      ov[e] = (half_t)(d * gelu_f(gf));
      og[e] = (half_t)(d * (float)v[e] * gelu_grad_f(gf));
    }
    *reinterpret_cast<half8_t*>(gh + base) = ov;
    *reinterpret_cast<half8_t*>(gh + base + 16) = og;
  }
}

__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const half_t* __restrict__ gy,
                                                              half_t* __restrict__ gx, int B, int H,
                                                              int W, int C) {
  const long nvec = C / 8;
  const long total = (long)B * H * W * nvec;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    long pix = i / nvec;
    int cv = (int)(i - pix * nvec);
    int b = (int)(pix / (H * W)), rem = (int)(pix - (long)b * H * W);
    int y = rem / W, x = rem - y * W;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        long src = ((long)b * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx;
        half8_t v = *reinterpret_cast<const half8_t*>(gy + src * C + cv * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
      }
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)acc[e];
    *reinterpret_cast<half8_t*>(gx + pix * C + cv * 8) = o;
  }
}

// Row softmax of scale*x (fp16 in/out, fp32 math), one wave per row; used by the single-head
// 512-channel attention of the VAE decoder mid block (the flash kernel covers head dims <= 160).
// (x and y may be the same buffer: every lane rewrites only elements it has read itself)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const half_t* x, half_t* y, long rows, int n,
                                                            float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const half_t* xr = x + row * n;
  half_t* yr = y + row * n;
  const int nvec = n / 8;
  float mx = -1.0e30f;
  for (int v = lane; v < nvec; v += 64) {
    half8_t h = *reinterpret_cast<const half8_t*>(xr + v * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)h[e] * scale);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int v = lane; v < nvec; v += 64) {
    half8_t h = *reinterpret_cast<const half8_t*>(xr + v * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += __expf((float)h[e] * scale - mx);
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int v = lane; v < nvec; v += 64) {
    half8_t h = *reinterpret_cast<const half8_t*>(xr + v * 8);
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)(__expf((float)h[e] * scale - mx) * inv);
    *reinterpret_cast<half8_t*>(yr + v * 8) = o;
  }
}

// per-step fused update, see lgd_hip.h
__global__ __launch_bounds__(256) void cfg_ddim_kernel(
    const float* __restrict__ eps, const float* __restrict__ x, float* __restrict__ x_out,
    const float* __restrict__ coef_table, const int32_t* __restrict__ dyn,
    const float* __restrict__ frozen_ref, const float* __restrict__ mask,
    float* __restrict__ hist, int B, int CHW, int HW) {
  const int step = dyn[0];
  const int frozen_steps = dyn[1];
  const float a_t = coef_table[step * 4 + 0], a_p = coef_table[step * 4 + 1];
  const float gs = coef_table[step * 4 + 2];
  const bool vpred = coef_table[step * 4 + 3] != 0.f;
  const float sa = sqrtf(a_t), sb = sqrtf(1.f - a_t), pa = sqrtf(a_p), pb = sqrtf(1.f - a_p);
  const long n = (long)B * CHW;
  const bool blend = frozen_ref && mask && step < frozen_steps;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    float eu = eps[i], ec = eps[n + i];
    float m = eu + gs * (ec - eu);
    float xv = x[i];
    float x0, e;
    if (vpred) {
      x0 = sa * xv - sb * m;
      e = sa * m + sb * xv;
    } else {
      e = m;
      x0 = (xv - sb * e) / sa;
    }
    float xn = pa * x0 + pb * e;
    if (blend) {
      int b = (int)(i / CHW);
      int p = (int)(i % HW);
      float mk = mask[(long)b * HW + p];
      xn = frozen_ref[(long)(step + 1) * n + i] * mk + xn * (1.f - mk);
    }
    x_out[i] = xn;
    if (hist) hist[(long)(step + 1) * n + i] = xn;
  }
}

// Classifier-free guidance + one step of a LINEAR MULTISTEP sampler (DPM-Solver++ 2M and anything else whose update
// is linear in the latents, the current data prediction and the previous one) + frozen-mask blend + history:
//     m  = eu + gs (ec - eu)                  model output under CFG
//     x0 = c0 x + c1 m                        data prediction (epsilon: 1/alpha_t, -sigma_t/alpha_t; v: alpha_t, -sigma_t)
//     x' = A x + B x0 + C x0_prev             first-order steps have C = 0 (x0_prev is not read)
//     x0_prev <- x0
// coef rows: fp32 [T][8] = {c0, c1, A, B, C, guidance_scale, -, -}.
__global__ __launch_bounds__(256) void cfg_multistep_kernel(
    const float* __restrict__ eps, const float* __restrict__ x, float* __restrict__ x_out, float* __restrict__ x0_prev,
    const float* __restrict__ coef_table, const int32_t* __restrict__ dyn, const float* __restrict__ frozen_ref,
    const float* __restrict__ mask, float* __restrict__ hist, int B, int CHW, int HW) {
  const int step = dyn[0];
  const int frozen_steps = dyn[1];
  const float* c = coef_table + step * 8;
  const float c0 = c[0], c1 = c[1], A = c[2], Bc = c[3], Cc = c[4], gs = c[5];
  const long n = (long)B * CHW;
  const bool blend = frozen_ref && mask && step < frozen_steps;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const float eu = eps[i], ec = eps[n + i];
    const float m = eu + gs * (ec - eu);
    const float xv = x[i];
    const float x0 = c0 * xv + c1 * m;
    float xn = A * xv + Bc * x0;
    if (Cc != 0.f) xn += Cc * x0_prev[i];
    x0_prev[i] = x0;
    if (blend) {
      const int b = (int)(i / CHW);
      const int p = (int)(i % HW);
      const float mk = mask[(long)b * HW + p];
      xn = frozen_ref[(long)(step + 1) * n + i] * mk + xn * (1.f - mk);
    }
    x_out[i] = xn;
    if (hist) hist[(long)(step + 1) * n + i] = xn;
  }
}

__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         const float* __restrict__ table,
                                                         const int32_t* __restrict__ dyn, int row_stride, int col,
                                                         long n, int reps) {
  const float s = table[(long)dyn[0] * row_stride + col];
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const float v = x[i] * s;
    for (int r = 0; r < reps; ++r) out[r * n + i] = v;
  }
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ g, float* __restrict__ x,
                                                    const float* __restrict__ coef_table,
                                                    const int32_t* __restrict__ step_idx, int col,
                                                    const float* __restrict__ active, long per_sample,
                                                    long n) {
  const float s = coef_table[(*step_idx) * 4 + col];
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const float a = active ? active[i / per_sample] : 1.f;   // per-image on/off (batched guidance)
    x[i] -= a * s * g[i];
  }
}

__global__ __launch_bounds__(256) void select_row_kernel(const float* __restrict__ table,
                                                          const int32_t* __restrict__ idx,
                                                          float* __restrict__ out, int n) {
  const long r = *idx;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = table[r * n + i];
}

inline int ew_blocks(long n) {
  long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int lgd_conv_in_f16(const float* x_nchw, const void* w, const float* bias, void* y, int B,
                               int Cin, int L, int Cout, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (Cin < 1 || Cin > 8 || Cout < 1) return LGD_ERR_ARG;
  const int K = 9 * Cin;
  size_t smem = (size_t)((K * Cout * 2 + 15) & ~15) + (size_t)CI_PIX * K * 4;
  if (smem > 64 * 1024) return LGD_ERR_UNSUPPORTED;
  long npix = (long)B * L * L;
  hipLaunchKernelGGL(conv_in_kernel, dim3((unsigned)((npix + CI_PIX - 1) / CI_PIX)), dim3(256), smem,
                     reinterpret_cast<hipStream_t>(stream), x_nchw, (const half_t*)w, bias,
                     (half_t*)y, B, Cin, L, Cout);
  return lgd_check_launch();
}

extern "C" int lgd_conv_out_f16(const void* x, const void* w, const float* bias, float* y_nchw, int B,
                                int Cin, int L, int Cout, float out_scale, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if ((Cin % 8) || Cin < 8) return LGD_ERR_ARG;
  long npix = (long)B * L * L;
  dim3 grid((unsigned)((npix + 3) / 4));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (Cout == 4)
    hipLaunchKernelGGL((conv_out_kernel<4>), grid, dim3(256), 0, st, (const half_t*)x,
                       (const half_t*)w, bias, y_nchw, B, Cin, L, out_scale);
  else if (Cout == 8)
    hipLaunchKernelGGL((conv_out_kernel<8>), grid, dim3(256), 0, st, (const half_t*)x,
                       (const half_t*)w, bias, y_nchw, B, Cin, L, out_scale);
  else
    return LGD_ERR_UNSUPPORTED;
  return lgd_check_launch();
}

extern "C" int lgd_add_f16(const void* a, const void* b, void* y, int64_t n, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (n % 8) return LGD_ERR_ARG;
  hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (const half_t*)a, (const half_t*)b,
                     (half_t*)y, (long)(n / 8));
  return lgd_check_launch();
}

extern "C" int lgd_scale_f16(const void* x, void* y, float alpha, int64_t n, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (n % 8) return LGD_ERR_ARG;
  hipLaunchKernelGGL(scale_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (const half_t*)x, (half_t*)y, alpha,
                     (long)(n / 8));
  return lgd_check_launch();
}

extern "C" int lgd_nchw_to_nhwc8_f16(const float* x, void* y, int B, int C, int HW, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (B < 1 || C < 1 || C > 8 || HW < 1) return LGD_ERR_ARG;
  const long total = (long)B * HW;
  hipLaunchKernelGGL(nchw_to_nhwc8_kernel, dim3(ew_blocks(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     (half_t*)y, C, (long)HW, total);
  return lgd_check_launch();
}

extern "C" int lgd_quick_gelu_f16(const void* x, void* y, int64_t n, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (n % 8) return LGD_ERR_ARG;
  hipLaunchKernelGGL(quick_gelu_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (const half_t*)x, (half_t*)y, (long)(n / 8));
  return lgd_check_launch();
}

extern "C" int lgd_geglu_fwd_f16(const void* h, void* y, int64_t rows, int n, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (n % 16) return LGD_ERR_ARG;
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(ew_blocks(rows * (n / 8))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (const half_t*)h, (half_t*)y, (long)rows, n);
  return lgd_check_launch();
}

extern "C" int lgd_geglu_bwd_f16(const void* h, const void* gy, void* gh, int64_t rows, int n,
                                 void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (n % 16) return LGD_ERR_ARG;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(ew_blocks(rows * (n / 8))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (const half_t*)h, (const half_t*)gy,
                     (half_t*)gh, (long)rows, n);
  return lgd_check_launch();
}

extern "C" int lgd_upsample2x_bwd_f16(const void* gy, void* gx, int B, int H, int W, int C,
                                      void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (C % 8) return LGD_ERR_ARG;
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(ew_blocks((long)B * H * W * (C / 8))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (const half_t*)gy, (half_t*)gx, B, H, W,
                     C);
  return lgd_check_launch();
}

extern "C" int lgd_softmax_rows_f16(const void* x, void* y, int64_t rows, int n, float scale,
                                    void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (n % 8 || rows < 1) return LGD_ERR_ARG;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (const half_t*)x, (half_t*)y, (long)rows, n,
                     scale);
  return lgd_check_launch();
}

extern "C" int lgd_cfg_ddim_step_f32(const float* eps, const float* x, float* x_out,
                                     const float* coef_table, const int32_t* dyn,
                                     const float* frozen_ref, const float* mask, float* hist, int B,
                                     int C, int HW, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3(ew_blocks((long)B * C * HW)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), eps, x, x_out, coef_table, dyn, frozen_ref,
                     mask, hist, B, C * HW, HW);
  return lgd_check_launch();
}

extern "C" int lgd_cfg_multistep_step_f32(const float* eps, const float* x, float* x_out, float* x0_prev,
                                          const float* coef_table, const int32_t* dyn, const float* frozen_ref,
                                          const float* mask, float* hist, int B, int C, int HW, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (!eps || !x || !x_out || !x0_prev || !coef_table || !dyn) return LGD_ERR_ARG;
  hipLaunchKernelGGL(cfg_multistep_kernel, dim3(ew_blocks((long)B * C * HW)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), eps, x, x_out, x0_prev, coef_table, dyn, frozen_ref,
                     mask, hist, B, C * HW, HW);
  return lgd_check_launch();
}

extern "C" int lgd_scale_rows_f32(const float* x, float* out, const float* table, const int32_t* dyn, int row_stride,
                                  int col, int64_t n, int reps, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (!x || !out || !table || !dyn || n < 1 || reps < 1 || col < 0 || col >= row_stride) return LGD_ERR_ARG;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(ew_blocks(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, out,
                     table, dyn, row_stride, col, (long)n, reps);
  return lgd_check_launch();
}

extern "C" int lgd_axpy_f32(const float* g, float* x, const float* coef_table,
                            const int32_t* step_idx, int col, const float* active, int64_t per_sample,
                            int64_t n, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_blocks(n)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), g, x, coef_table, step_idx, col, active,
                     (long)(per_sample > 0 ? per_sample : n), (long)n);
  return lgd_check_launch();
}

extern "C" int lgd_select_row_f32(const float* table, const int32_t* idx, float* out, int n,
                                  void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(select_row_kernel, dim3(ew_blocks(n)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), table, idx, out, n);
  return lgd_check_launch();
}
