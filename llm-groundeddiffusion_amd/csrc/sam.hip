// SAM mask refinement (reference models/sam.py:25-55 calls the Hugging Face `SamModel`):
// the pieces of the ViT image encoder that are not plain GEMM / LayerNorm / attention calls.
//
//  * sam_relpos_qkv_kernel — window partition + decomposed relative-position bias, folded into the attention
//    operands.  SAM adds  rel_h[q, kh] + rel_w[q, kw]  (rel_h[q, kh] = q . Rh[qh - kh + S - 1], likewise for w;
//    [ext] transformers SamVisionAttention.get_decomposed_rel_pos) to the scaled logits.  That bias is an inner
//    product of a query-side vector with a one-hot key-side vector, so it is appended to the head dimension:
//        Q' = [ q | rel_h(q, 0..S-1) / scale | rel_w(q, 0..S-1) / scale | 0 ]      (DA columns)
//        K' = [ k | onehot_S(kh)            | onehot_S(kw)              | 0 ]
//        V' = [ v | 0 ]
//    and  scale * Q'.K' = scale * q.k + rel_h[q, kh] + rel_w[q, kw]  comes out of the ordinary flash-attention
//    kernel (lgd_attn_fwd_f16 with d = DA) without materialising the S^2 x S^2 bias.  The same kernel gathers the
//    tokens into window order; positions of the zero padding SAM applies AFTER the first LayerNorm (SamVisionLayer
//    .window_partition) carry k = b_k, v = b_v (the projection of a zero vector) and stay visible as keys.
//  * sam_window_merge_kernel — the inverse gather: window order -> raster order, padding and extra columns dropped.
//  * act_kernel — exact (erf) GELU and ReLU on fp16 vectors (encoder MLP / mask-decoder MLPs).
#include "common.h"
#include "../../include/lgd_hip.h"

namespace {

struct RelposArgs {
  const half_t* qkv;      // [B*Hs*Ws][3*C], C = NH*d, raster token order
  const float* qkv_bias;  // [3*C]
  const float* rel_h;     // [2*S-1][d]
  const float* rel_w;     // [2*S-1][d]
  half_t *qa, *ka, *va;   // [B*nwy*nwx*S*S][NH*DA]
  int B, Hs, Ws, S, nwy, nwx, NH, d, DA;
  float inv_scale;
};

__global__ __launch_bounds__(256) void sam_relpos_qkv_kernel(const RelposArgs a) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  const int C = a.NH * a.d, S = a.S;
  float* s_q = reinterpret_cast<float*>(dyn_smem);  // [C]
  float* s_rel = s_q + C;                            // [NH][2S]
  const long wt = blockIdx.x;                        // window-token index
  const int ix = (int)(wt % S);
  const int iy = (int)((wt / S) % S);
  const long win = wt / ((long)S * S);
  const int wx = (int)(win % a.nwx);
  const int wy = (int)((win / a.nwx) % a.nwy);
  const int b = (int)(win / ((long)a.nwx * a.nwy));
  const int y = wy * S + iy, x = wx * S + ix;
  const bool valid = y < a.Hs && x < a.Ws;
  const half_t* src = a.qkv + (((long)b * a.Hs + y) * a.Ws + x) * 3L * C;

  for (int i = threadIdx.x; i < C; i += 256) s_q[i] = valid ? (float)src[i] : a.qkv_bias[i];
  __syncthreads();
  for (int i = threadIdx.x; i < a.NH * 2 * S; i += 256) {
    const int h = i / (2 * S), j = i - h * 2 * S;
    const float* tab = j < S ? a.rel_h + (long)(iy - j + S - 1) * a.d : a.rel_w + (long)(ix - (j - S) + S - 1) * a.d;
    const float* q = s_q + h * a.d;
    float acc = 0.f;
    for (int c = 0; c < a.d; ++c) acc = fmaf(q[c], tab[c], acc);
    s_rel[i] = acc * a.inv_scale;
  }
  __syncthreads();
  const long row = wt * (long)a.NH * a.DA;
  for (int i = threadIdx.x; i < a.NH * a.DA; i += 256) {
    const int h = i / a.DA, c = i - h * a.DA;
    half_t q, k, v;
    if (c < a.d) {
      const int e = h * a.d + c;
      q = (half_t)s_q[e];
      k = valid ? src[C + e] : (half_t)a.qkv_bias[C + e];
      v = valid ? src[2 * C + e] : (half_t)a.qkv_bias[2 * C + e];
    } else if (c < a.d + 2 * S) {
      const int j = c - a.d;
      q = (half_t)s_rel[h * 2 * S + j];
      k = (half_t)((j < S ? j == iy : (j - S) == ix) ? 1.f : 0.f);
      v = (half_t)0.f;
    } else {
      q = k = v = (half_t)0.f;
    }
    a.qa[row + i] = q;
    a.ka[row + i] = k;
    a.va[row + i] = v;
  }
}

// oa [B*nwy*nwx*S*S][NH*DA] (window order) -> out [B*Hs*Ws][NH*d] (raster order); one workgroup per output token
__global__ __launch_bounds__(256) void sam_window_merge_kernel(const half_t* __restrict__ oa, half_t* __restrict__ out,
                                                                int B, int Hs, int Ws, int S, int nwy, int nwx, int NH,
                                                                int d, int DA) {
  const long t = blockIdx.x;
  const int x = (int)(t % Ws), y = (int)((t / Ws) % Hs), b = (int)(t / ((long)Ws * Hs));
  const int wy = y / S, iy = y - wy * S, wx = x / S, ix = x - wx * S;
  const long wt = ((((long)b * nwy + wy) * nwx + wx) * S + iy) * S + ix;
  const half_t* src = oa + wt * (long)NH * DA;
  half_t* dst = out + t * (long)NH * d;
  const int dv = d / 8;
  for (int i = threadIdx.x; i < NH * dv; i += 256) {
    const int h = i / dv, s = i - h * dv;
    reinterpret_cast<half8_t*>(dst + h * d)[s] = reinterpret_cast<const half8_t*>(src + h * DA)[s];
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void act_kernel(const half_t* a, half_t* y, long nvec) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += gridDim.x * 256L) {
    half8_t x0 = reinterpret_cast<const half8_t*>(a)[i];
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (float)x0[e];
      o[e] = (half_t)(MODE == LGD_ACT_GELU ? gelu_f(x) : fmaxf(x, 0.f));
    }
    reinterpret_cast<half8_t*>(y)[i] = o;
  }
}

}  // namespace

extern "C" int lgd_sam_relpos_qkv_f16(const void* qkv, const float* qkv_bias, const float* rel_h, const float* rel_w,
                                      int B, int Hs, int Ws, int window, int NH, int d, int DA, float scale, void* qa,
                                      void* ka, void* va, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (B < 1 || Hs < 1 || Ws < 1 || NH < 1 || d < 8 || (d % 8) || (DA % 8) || window < 0 || scale <= 0.f) return LGD_ERR_ARG;
  if (window == 0 && Hs != Ws) return LGD_ERR_ARG;  // global attention: one window = the whole (square) grid
  RelposArgs a;
  a.S = window ? window : Hs;
  if (DA < d + 2 * a.S) return LGD_ERR_ARG;
  a.qkv = (const half_t*)qkv; a.qkv_bias = qkv_bias; a.rel_h = rel_h; a.rel_w = rel_w;
  a.qa = (half_t*)qa; a.ka = (half_t*)ka; a.va = (half_t*)va;
  a.B = B; a.Hs = Hs; a.Ws = Ws; a.NH = NH; a.d = d; a.DA = DA;
  a.nwy = (Hs + a.S - 1) / a.S; a.nwx = (Ws + a.S - 1) / a.S;
  a.inv_scale = 1.f / scale;
  const long blocks = (long)B * a.nwy * a.nwx * a.S * a.S;
  const size_t lds = (size_t)(NH * d + NH * 2 * a.S) * sizeof(float);
  if (blocks > 0x7fffffffL || lds > 64 * 1024) return LGD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(sam_relpos_qkv_kernel, dim3((unsigned)blocks), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), a);
  return lgd_check_launch();
}

extern "C" int lgd_sam_window_merge_f16(const void* oa, void* out, int B, int Hs, int Ws, int window, int NH, int d,
                                        int DA, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (B < 1 || Hs < 1 || Ws < 1 || NH < 1 || d < 8 || (d % 8) || (DA % 8) || DA < d || window < 0) return LGD_ERR_ARG;
  if (window == 0 && Hs != Ws) return LGD_ERR_ARG;
  const int S = window ? window : Hs;
  const long blocks = (long)B * Hs * Ws;
  if (blocks > 0x7fffffffL) return LGD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(sam_window_merge_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const half_t*)oa, (half_t*)out, B, Hs, Ws, S, (Hs + S - 1) / S, (Ws + S - 1) / S, NH, d, DA);
  return lgd_check_launch();
}

extern "C" int lgd_act_f16(const void* x, void* y, int64_t n, int mode, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (n < 0 || (n % 8)) return LGD_ERR_ARG;
  long b = (n / 8 + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (mode == LGD_ACT_GELU)
    hipLaunchKernelGGL(act_kernel<LGD_ACT_GELU>, dim3((unsigned)b), dim3(256), 0, st, (const half_t*)x, (half_t*)y, (long)(n / 8));
  else if (mode == LGD_ACT_RELU)
    hipLaunchKernelGGL(act_kernel<LGD_ACT_RELU>, dim3((unsigned)b), dim3(256), 0, st, (const half_t*)x, (half_t*)y, (long)(n / 8));
  else
    return LGD_ERR_ARG;
  return lgd_check_launch();
}
