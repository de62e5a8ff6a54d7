// BoxDiff energy and its gradient on the cross-attention probability maps — one launch, one workgroup per image.
//
// Reference: utils/boxdiff.py:20-101 (_compute_max_attention_per_index), :104-118 (_compute_loss), :121-196
// (compute_ca_loss_boxdiff / add_ca_loss_per_attn_map_to_loss_boxdiff), scaled by amp_loss_scale at :224.  There it is a
// Python loop over objects x phrase tokens with a mask build, a 3x3 smoothing conv, two torch.topk and two max
// reductions each, on top of a cat / mean over the five 16x16 cross-attention maps and a token soft-max, plus the
// autograd graph through all of it.
//
// Structure (HW = side^2 <= 1024 positions, T text tokens, n_maps x H (layer, head) maps per image):
//   phase 1  one wave per position: mean over the n_maps x H maps (lane = token), x100, soft-max over tokens 1 .. T-2
//            (:34-36); keeps the row maximum, the row sum and the probabilities of the ITEM tokens in LDS;
//   phase 2  per item (object, phrase token), whole workgroup: reflect-padded 3x3 smoothing (:71-76), inner- / outer-box
//            top-k means by exact ranking (:80-87), corner terms from the row / column maxima (:89-99), and the gradient
//            of all of it on the unsmoothed probability image (fixed summation order: no atomics);
//   phase 3  one wave per position again: soft-max backward over the tokens and the 1 / (n_maps H) fan-out of the mean,
//            written to every (layer, head) gradient map.
// Python's `max(0, 1 - v)` of (int, tensor) (:107-109) keeps the tensor only if `tensor > 0`; a top-k of k = 0 elements
// (a box of fewer than 1 / P pixels) has mean NaN and therefore drops out — reproduced by skipping terms with k = 0.
#include "common.h"
#include "../../include/lgd_hip.h"

namespace {

constexpr int BD_MAXHW = 1024;
constexpr int BD_MAXSIDE = 32;

struct BoxDiffArgs {
  const float* const* maps;
  float* const* gmaps;
  const int32_t* items;    // [n_items][8] = {token, mask_id, k_fg, k_bg, -, -, -, -}
  const float* masks;      // [n_masks][3][HW]: box mask | corner_x[side], corner_y[side] | gt_x[side], gt_y[side]
  const float* smooth;     // 9 weights (row-major dy, dx) or nullptr
  const int32_t* groups;   // [n_samples][2] = {first item, item count}
  float* loss;             // [n_samples]
  int n_maps, side, H, T, max_items;
  float loss_scale, grad_scale;
};

__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ __launch_bounds__(256) void boxdiff_energy_kernel(BoxDiffArgs a) {
  extern __shared__ float lds[];
  const int side = a.side, HW = side * side, T = a.T, H = a.H;
  const int smp = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int first = a.groups[smp * 2], count = a.groups[smp * 2 + 1];
  float* s_m = lds;                    // [HW] row maximum of 100 * mean map over tokens 1 .. T-2
  float* s_z = s_m + HW;               // [HW] row sum of exp
  float* s_c = s_z + HW;               // [HW] sum_i g_i s_i of the soft-max backward
  float* s_sm = s_c + HW;              // [HW] smoothed image of the current item
  float* s_gsm = s_sm + HW;            // [HW] gradient on the smoothed image
  float* s_v = s_gsm + HW;             // [HW] image * mask
  float* s_w = s_v + HW;               // [HW] image * (1 - mask)
  float* s_img = s_w + HW;             // [max_items][HW] probabilities of the item tokens
  float* s_gimg = s_img + (long)a.max_items * HW;   // [max_items][HW] gradient on them
  __shared__ float s_red[4];
  __shared__ float s_line[2 * BD_MAXSIDE];          // column maxima | row maxima
  __shared__ int s_arg[2 * BD_MAXSIDE];
  const float inv_n = 1.f / (float)(a.n_maps * H);
  const int j0 = lane, j1 = lane + 64;
  const bool ok0 = j0 >= 1 && j0 <= T - 2, ok1 = j1 >= 1 && j1 <= T - 2;

  // ---- phase 1: mean over (layer, head), x100, soft-max over the tokens
  for (int pos = wave; pos < HW; pos += 4) {
    float a0 = 0.f, a1 = 0.f;
    for (int k = 0; k < a.n_maps; ++k) {
      const float* base = a.maps[k] + ((long)smp * H * HW + pos) * T;
      for (int h = 0; h < H; ++h) {
        const float* row = base + (long)h * HW * T;
        if (j0 < T) a0 += row[j0];
        if (j1 < T) a1 += row[j1];
      }
    }
    const float x0 = a0 * inv_n * 100.f, x1 = a1 * inv_n * 100.f;
    const float m = wave_max(fmaxf(ok0 ? x0 : -INFINITY, ok1 ? x1 : -INFINITY));
    const float e0 = ok0 ? __expf(x0 - m) : 0.f, e1 = ok1 ? __expf(x1 - m) : 0.f;
    const float z = wave_sum(e0 + e1);
    if (lane == 0) { s_m[pos] = m; s_z[pos] = z; }
    for (int it = 0; it < count; ++it) {
      const int tok = a.items[(first + it) * 8 + 0];
      const float e = tok < 64 ? __shfl(e0, tok, 64) : __shfl(e1, tok - 64, 64);
      if (lane == 0) s_img[(long)it * HW + pos] = e / z;
    }
  }
  __syncthreads();

  // ---- phase 2: the items of this image, one after the other
  float total = 0.f;                                   // meaningful in thread 0
  for (int it = 0; it < count; ++it) {
    const int32_t* item = a.items + (first + it) * 8;
    const int k_fg = item[2], k_bg = item[3];
    const float* M = a.masks + (long)item[1] * 3 * HW;
    const float* cmask = M + HW;                       // corner_x[side] | corner_y[side]
    const float* gt = M + 2 * HW;                      // gt_x[side] | gt_y[side]
    const float* img = s_img + (long)it * HW;
    float* gimg = s_gimg + (long)it * HW;
    for (int i = tid; i < HW; i += 256) {
      float v;
      if (a.smooth) {
        const int y = i / side, x = i - y * side;
        v = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) v += a.smooth[dy * 3 + dx] * img[refl(y + dy - 1, side) * side + refl(x + dx - 1, side)];
      } else {
        v = img[i];
      }
      const float m = M[i];
      s_sm[i] = v;
      s_v[i] = v * m;
      s_w[i] = v * (1.f - m);
      s_gsm[i] = 0.f;
    }
    __syncthreads();
    // inner- / outer-box top-k by exact ranking (index tie-break), as csrc/energy.hip
    float fg_sum = 0.f, bg_sum = 0.f;
    unsigned sel_f = 0, sel_b = 0;                     // selection bits of this thread's positions (<= 4)
    for (int i = tid, n = 0; i < HW; i += 256, ++n) {
      const float vi = s_v[i], wi = s_w[i];
      int rf = 0, rb = 0;
      for (int j = 0; j < HW; ++j) {
        const float vj = s_v[j], wj = s_w[j];
        rf += (vj > vi) || (vj == vi && j < i);
        rb += (wj > wi) || (wj == wi && j < i);
      }
      if (rf < k_fg) { fg_sum += vi; sel_f |= 1u << n; }
      if (rb < k_bg) { bg_sum += wi; sel_b |= 1u << n; }
    }
    fg_sum = block_sum_256(fg_sum, s_red);
    bg_sum = block_sum_256(bg_sum, s_red);
    const float l_fg = k_fg > 0 ? 1.f - fg_sum / (float)k_fg : 0.f;     // max(0, 1 - mean) with Python's max
    const float l_bg = k_bg > 0 ? bg_sum / (float)k_bg : 0.f;
    const bool on_fg = k_fg > 0 && l_fg > 0.f, on_bg = k_bg > 0 && l_bg > 0.f;
    for (int i = tid, n = 0; i < HW; i += 256, ++n) {
      const float m = M[i];
      float g = 0.f;
      if (on_fg && (sel_f >> n & 1u)) g -= m / (float)k_fg;
      if (on_bg && (sel_b >> n & 1u)) g += (1.f - m) / (float)k_bg;
      s_gsm[i] = g;
    }
    // corner terms: column maxima (thread x < side) and row maxima (thread side + y)
    if (tid < 2 * side) {
      const bool col = tid < side;
      const int q = col ? tid : tid - side;
      float best = -INFINITY;
      int arg = 0;
      for (int r = 0; r < side; ++r) {
        const float v = col ? s_sm[r * side + q] : s_sm[q * side + r];
        if (v > best) { best = v; arg = r; }
      }
      s_line[tid] = best;
      s_arg[tid] = arg;
    }
    __syncthreads();
    float dist = 0.f;
    if (tid < 2 * side) dist = fabsf(s_line[tid] - gt[tid]) * cmask[tid] / (float)side;     // .mean() over the side
    dist = block_sum_256(dist, s_red);
    if (tid < side) {                                  // one writer per column, then one per row: no conflicts
      const float d = s_line[tid] - gt[tid];
      s_gsm[s_arg[tid] * side + tid] += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * cmask[tid] / (float)side;
    }
    __syncthreads();
    if (tid >= side && tid < 2 * side) {
      const float d = s_line[tid] - gt[tid];
      s_gsm[(tid - side) * side + s_arg[tid]] += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * cmask[tid] / (float)side;
    }
    __syncthreads();
    if (tid == 0) total += (on_fg ? l_fg : 0.f) + (on_bg ? l_bg : 0.f) + dist;
    // back through the reflect-padded smoothing: gather form (every input position sums the outputs that read it)
    for (int i = tid; i < HW; i += 256) {
      float g;
      if (a.smooth) {
        const int y = i / side, x = i - y * side;
        g = 0.f;
        for (int yo = max(y - 1, 0); yo <= min(y + 1, side - 1); ++yo)
          for (int dy = 0; dy < 3; ++dy) {
            if (refl(yo + dy - 1, side) != y) continue;
            for (int xo = max(x - 1, 0); xo <= min(x + 1, side - 1); ++xo)
              for (int dx = 0; dx < 3; ++dx)
                if (refl(xo + dx - 1, side) == x) g += a.smooth[dy * 3 + dx] * s_gsm[yo * side + xo];
          }
      } else {
        g = s_gsm[i];
      }
      gimg[i] = g;
    }
    __syncthreads();
  }
  if (tid == 0) a.loss[smp] = a.loss_scale * total;
  if (!a.gmaps) return;

  // ---- phase 3: soft-max backward over the tokens, fan-out to every (layer, head) map
  for (int i = tid; i < HW; i += 256) {
    float c = 0.f;
    for (int it = 0; it < count; ++it) c += s_gimg[(long)it * HW + i] * s_img[(long)it * HW + i];
    s_c[i] = c;
  }
  __syncthreads();
  const float gs = 100.f * a.loss_scale * a.grad_scale * inv_n;
  for (int pos = wave; pos < HW; pos += 4) {
    float a0 = 0.f, a1 = 0.f;
    for (int k = 0; k < a.n_maps; ++k) {
      const float* base = a.maps[k] + ((long)smp * H * HW + pos) * T;
      for (int h = 0; h < H; ++h) {
        const float* row = base + (long)h * HW * T;
        if (j0 < T) a0 += row[j0];
        if (j1 < T) a1 += row[j1];
      }
    }
    const float m = s_m[pos], iz = 1.f / s_z[pos], c = s_c[pos];
    const float p0 = ok0 ? __expf(a0 * inv_n * 100.f - m) * iz : 0.f;
    const float p1 = ok1 ? __expf(a1 * inv_n * 100.f - m) * iz : 0.f;
    float g0 = 0.f, g1 = 0.f;
    for (int it = 0; it < count; ++it) {
      const int tok = a.items[(first + it) * 8 + 0];
      const float g = s_gimg[(long)it * HW + pos];
      if (tok == j0) g0 += g;
      if (tok == j1) g1 += g;
    }
    const float d0 = gs * p0 * (g0 - c), d1 = gs * p1 * (g1 - c);
    for (int k = 0; k < a.n_maps; ++k) {
      float* base = a.gmaps[k] + ((long)smp * H * HW + pos) * T;
      for (int h = 0; h < H; ++h) {
        float* row = base + (long)h * HW * T;
        if (ok0) row[j0] = d0;
        if (ok1) row[j1] = d1;
      }
    }
  }
}

}  // namespace

extern "C" int lgd_boxdiff_energy_f32(const float* const* maps, float* const* gmaps, int n_maps, int side,
                                      const int32_t* items, const float* masks, const float* smooth,
                                      const int32_t* groups, int n_samples, int max_items, int H, int T,
                                      float loss_scale, float grad_scale, float* loss, void* stream) {
  (void)hipGetLastError();
  if (!maps || !items || !masks || !groups || !loss || n_maps < 1 || n_samples < 1 || H < 1 || max_items < 0) return LGD_ERR_ARG;
  if (side < 2 || side > BD_MAXSIDE || side * side > BD_MAXHW || T < 3 || T > 128) return LGD_ERR_UNSUPPORTED;
  const size_t bytes = (size_t)(7 + 2 * (size_t)max_items) * side * side * sizeof(float);
  if (bytes > 150 * 1024) return LGD_ERR_UNSUPPORTED;     // item images live in LDS: HW = 256 allows 70 items per image
  // the attribute is per DEVICE and a failure must not stick: set it whenever the launch needs more than the default
  // 64 KB (a host call per launch of an eager-only kernel that runs once per denoising step)
  if (bytes > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(boxdiff_energy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          150 * 1024) != hipSuccess)
    return LGD_ERR_LAUNCH;
  BoxDiffArgs a{maps, gmaps, items, masks, smooth, groups, loss, n_maps, side, H, T, max_items, loss_scale, grad_scale};
  hipLaunchKernelGGL(boxdiff_energy_kernel, dim3(n_samples), dim3(256), bytes, reinterpret_cast<hipStream_t>(stream), a);
  return lgd_check_launch();
}
