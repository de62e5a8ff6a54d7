// Cross-attention energy of LMD/LMD+ and its gradient on the probability maps — one launch.
//
// Reference (Python loops over keys x objects x tokens, two torch.topk + a mask build each, plus
// the autograd graph through them): utils/guidance.py:91-148 (add_ca_loss_per_attn_map_to_loss:
// item kind 0 = max-based branch :131-145, kind 2 = ratio-based branch :118-130 — the branch the
// signature defaults to and generation/backward_guidance.py:99-112 therefore runs), :150-242
// (add_ref_ca_loss_per_attn_map_to_lossv2, kind 1), :244-286 (compute_ca_lossv3), scaled by
// loss_scale at models/pipelines.py:48.
//
// grid = (heads, n_groups); one workgroup evaluates, for one head, all items that touch the same column
// (map, image, token) of <= 4096 spatial positions (round 5: 64x64 guidance keys), one after the other, accumulating their map gradients in
// LDS and storing the column once: no atomics, so the gradient (and through it the data-dependent
// iteration count of the guidance loop) is bit-reproducible however many terms share a token column (a
// phrase with several boxes contributes one reference term per box on top of its box term).
// top-k is done by exact ranking (count of greater elements, index tie-break) from LDS — k-independent,
// deterministic, O(HW^2/256) per thread (HW = 256: 256 comparisons per position; HW = 4096: 16 positions x 4096 each).
// Every normalisation of the reference (1/len(tokens), 1/(n_obj*n_keys), mean over heads for the
// reference term, loss_scale) is folded into the per-item coefficients by the host.
#include "common.h"
#include "../../include/lgd_hip.h"

namespace {

constexpr int E_MAXHW = 4096;     // 3 x 16 KB of LDS: every cross-attention map of SD 1.x at 512^2 (the 64x64 level included)
constexpr float REF_EPS = 1e-5f;  // guidance.py:150 (eps=1e-5)

__global__ __launch_bounds__(256) void ca_energy_kernel(
    const float* const* __restrict__ maps, float* const* __restrict__ gmaps,
    const int32_t* __restrict__ map_hw, const int32_t* __restrict__ items,
    const float* __restrict__ coefs, const float* __restrict__ masks,
    const float* __restrict__ refs, long refs_step_stride, const int32_t* __restrict__ dyn,
    const int32_t* __restrict__ groups, int H, int T, int max_hw, float gscale,
    float* __restrict__ partial) {
  __shared__ float s_v[E_MAXHW];   // A * M      (fg) / A*M (ref)
  __shared__ float s_w[E_MAXHW];   // A * (1-M)  (bg) / R*M (ref)
  __shared__ float s_g[E_MAXHW];   // d loss / d A of this column, summed over the group's items
  __shared__ float s_red[4];
  const int h = blockIdx.x;
  const int first = groups[blockIdx.y * 2], count = groups[blockIdx.y * 2 + 1];
  const int tid = threadIdx.x;
  // all items of a group share the map, the image and the token
  const int32_t* it0 = items + first * 8;
  const int map_id = it0[0], tok = it0[2], smp = it0[7];
  const int HW = map_hw[map_id];
  // maps are [n_samples][H][HW][T]; item `smp` selects the image of the batch
  const long img = ((long)smp * H + h) * HW * T + tok;
  const float* A = maps[map_id] + img;
  float* G = gmaps ? gmaps[map_id] + img : nullptr;
  for (int i = tid; i < HW; i += 256) s_g[i] = 0.f;     // thread t owns positions t, t+256, ... throughout

  for (int item = first; item < first + count; ++item) {
    const int32_t* it = items + item * 8;
    const int kind = it[1], mask_id = it[3], k_fg = it[4], k_bg = it[5], ref_id = it[6];
    const float c_fg = coefs[item * 4 + 0], c_bg = coefs[item * 4 + 1], c_ref = coefs[item * 4 + 2];
    const float* M = masks + (long)mask_id * max_hw;
    __syncthreads();                                     // s_v / s_w of the previous item are no longer read
    if (kind == 0) {
      for (int i = tid; i < HW; i += 256) {
        float a = A[(long)i * T], m = M[i];
        s_v[i] = a * m;
        s_w[i] = a * (1.f - m);
      }
      __syncthreads();
      float fg_sum = 0.f, bg_sum = 0.f;
      for (int i = tid; i < HW; i += 256) {
        const float vi = s_v[i], wi = s_w[i];
        int rf = 0, rb = 0;
        for (int j = 0; j < HW; ++j) {
          const float vj = s_v[j], wj = s_w[j];
          rf += (vj > vi) || (vj == vi && j < i);
          rb += (wj > wi) || (wj == wi && j < i);
        }
        float g = 0.f;
        const float m = M[i];
        if (rf < k_fg) { fg_sum += vi; g -= c_fg / (float)k_fg * m; }
        if (rb < k_bg) { bg_sum += wi; g += c_bg / (float)k_bg * (1.f - m); }
        s_g[i] += g * gscale;
      }
      fg_sum = block_sum_256(fg_sum, s_red);
      bg_sum = block_sum_256(bg_sum, s_red);
      if (tid == 0)
        partial[item * H + h] = c_fg * (1.f - fg_sum / (float)k_fg) + c_bg * (bg_sum / (float)k_bg);
    } else if (kind == 2) {
      // ratio-based term (guidance.py:124-126): r = sum(A*M) / sum(A) per head, term (1 - r)^2, the mean over
      // heads and the 1/len(tokens), 1/(n_obj*n_keys), loss_scale factors are folded into coefs[3] by the host.
      // d term / d A_i = -2 c (1 - r) (M_i * sum(A) - sum(A*M)) / sum(A)^2.  No epsilon, as in the reference.
      const float c_rat = coefs[item * 4 + 3];
      float sa = 0.f, sm = 0.f;
      for (int i = tid; i < HW; i += 256) {
        const float a = A[(long)i * T];
        sa += a;
        sm += a * M[i];
      }
      sa = block_sum_256(sa, s_red);
      sm = block_sum_256(sm, s_red);
      const float r = sm / sa;
      const float gk = -2.f * c_rat * (1.f - r) / (sa * sa) * gscale;
      for (int i = tid; i < HW; i += 256) s_g[i] += gk * (M[i] * sa - sm);
      if (tid == 0) partial[item * H + h] = c_rat * (1.f - r) * (1.f - r);
    } else {
      const float* R = refs + (long)dyn[0] * refs_step_stride + ((long)ref_id * H + h) * max_hw;
      float sa = 0.f, sr = 0.f;
      for (int i = tid; i < HW; i += 256) {
        float m = M[i];
        float am = A[(long)i * T] * m, rm = R[i] * m;
        s_v[i] = am;
        s_w[i] = rm;
        sa += am;
        sr += rm;
      }
      sa = block_sum_256(sa, s_red);
      sr = block_sum_256(sr, s_red);
      const float ia = 1.f / (sa + REF_EPS), ir = 1.f / (sr + REF_EPS);
      float l1 = 0.f, dot = 0.f;  // dot = sum_j sign_j * A_j M_j
      for (int i = tid; i < HW; i += 256) {
        float df = s_v[i] * ia - s_w[i] * ir;
        float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        l1 += fabsf(df);
        dot += sg * s_v[i];
      }
      l1 = block_sum_256(l1, s_red);
      dot = block_sum_256(dot, s_red);
      for (int i = tid; i < HW; i += 256) {
        float m = M[i];
        if (m != 0.f) {
          float df = s_v[i] * ia - s_w[i] * ir;
          float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
          s_g[i] += gscale * c_ref * m * (sg * ia - dot * ia * ia);
        }
      }
      if (tid == 0) partial[item * H + h] = c_ref * l1;
    }
  }
  if (G)
    for (int i = tid; i < HW; i += 256) G[(long)i * T] = s_g[i];
}

// loss[b] = sum of the partial terms of the items that belong to image b (grid = n_samples)
__global__ __launch_bounds__(256) void energy_sum_kernel(const float* __restrict__ partial,
                                                          const int32_t* __restrict__ items, int n_items,
                                                          int H, float* __restrict__ loss) {
  __shared__ float s_red[4];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < n_items * H; i += 256)
    if (items[(i / H) * 8 + 7] == b) s += partial[i];
  s = block_sum_256(s, s_red);
  if (threadIdx.x == 0) loss[b] = s;
}

}  // namespace

extern "C" int lgd_ca_energy_f32(const float* const* maps, float* const* gmaps,
                                 const int32_t* map_hw, const int32_t* items, const float* coefs,
                                 const float* masks, const float* refs, int64_t refs_step_stride,
                                 const int32_t* dyn, const int32_t* groups, int n_groups, int n_items,
                                 int n_samples, int H, int T, int max_hw, float grad_scale, float* partial,
                                 float* loss, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (n_items < 0 || n_groups < 0 || n_samples < 1 || H < 1 || max_hw > E_MAXHW) return LGD_ERR_ARG;
  if (n_items > 0 && (n_groups < 1 || !groups)) return LGD_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n_items > 0)
    hipLaunchKernelGGL(ca_energy_kernel, dim3(H, n_groups), dim3(256), 0, st, maps, gmaps, map_hw,
                       items, coefs, masks, refs, (long)refs_step_stride, dyn, groups, H, T, max_hw,
                       grad_scale, partial);
  hipLaunchKernelGGL(energy_sum_kernel, dim3(n_samples), dim3(256), 0, st, partial, items, n_items, H,
                     loss);
  return lgd_check_launch();
}
