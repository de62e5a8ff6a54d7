// Attention input-gradients for the backward-guidance pass (torch.autograd.grad at
// models/pipelines.py:56 through attention_processor.py:201-233,447 / :355-357).
//
// Self-attention: flash-style recompute in two kernels, both reusing the transposed-MFMA layout of
// attn.hip (lane&15 = query or key, 4 consecutive contraction rows per lane):
//   dq kernel : workgroup = 64 queries, loops over key tiles.  Also produces
//               delta[q] = sum_d dO[q,d] O[q,d] for the second kernel.
//   dkv kernel: workgroup = 64 keys, loops over query tiles.
// With P = softmax(S), S = scale Q K^T:  dV = P^T dO;  dP = dO V^T;  dS = P o (dP - delta);
// dQ = scale dS K;  dK = scale dS^T Q.
//
// Cross-attention (77 text tokens): only dQ is needed (text K/V are constants), but the
// probability map is itself an output that receives a gradient from the energy, so the kernel
// takes gP in addition to gO:  dP_total = gP + gO V^T.  cross_attn_bwd_mfma_kernel does this on the
// matrix cores with all (<= 96) keys of a query in registers; the older VALU kernel (one wave per query
// row) remains as the fallback for 96 < Sk <= 128.
#include "common.h"
#include "../../include/lgd_hip.h"
#include <stdlib.h>

namespace {

constexpr int T64 = 64;
constexpr int TR_LD = T64 + 8;  // halfs per row of a transposed tile

// Tile staging, split into a global->register half and a register->LDS half so that the loads of
// tile t+1 are in flight while tile t is multiplied.  A tile is 64 rows x DP; a thread owns
// (row pair, 8-column segment) items, which lets the transposed image be written as 4-byte
// (row pair) stores.  Rows past `nrows` and columns past `d` are zero.
template <int DP, int NT = 256>
struct PairTile {
  static constexpr int KSEG = DP / 8;
  static constexpr int ITEMS = (T64 / 2) * KSEG;
  static constexpr int IT = (ITEMS + NT - 1) / NT;
  uint4 v[IT][2];

  __device__ __forceinline__ void load(const half_t* __restrict__ src, long ld, int row0, int nrows, int d) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int idx = threadIdx.x + i * NT;
      const int pair = idx & 31, seg = idx >> 5;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int row = row0 + pair * 2 + r;
        const bool ok = idx < ITEMS && row < nrows && seg * 8 < d;
        v[i][r] = ok ? *reinterpret_cast<const uint4*>(src + (long)row * ld + seg * 8) : make_uint4(0, 0, 0, 0);
      }
    }
  }
  template <bool RM, bool TR>
  __device__ __forceinline__ void store(half_t* rm, half_t* tr) const {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int idx = threadIdx.x + i * NT;
      if (idx >= ITEMS) continue;
      const int pair = idx & 31, seg = idx >> 5;
      if (RM) {
        *reinterpret_cast<uint4*>(rm + (pair * 2) * (DP + 16) + seg * 8) = v[i][0];
        *reinterpret_cast<uint4*>(rm + (pair * 2 + 1) * (DP + 16) + seg * 8) = v[i][1];
      }
      if (TR) {
        // (row 2p, row 2p+1) halves of column seg*8+e packed into one dword, by integer ops on the
        // loaded words (no address-taken register arrays: those end up in scratch)
        const uint32_t w0[4] = {v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w};
        const uint32_t w1[4] = {v[i][1].x, v[i][1].y, v[i][1].z, v[i][1].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t even = (w0[j] & 0xffffu) | (w1[j] << 16);
          const uint32_t odd = (w0[j] >> 16) | (w1[j] & 0xffff0000u);
          *reinterpret_cast<uint32_t*>(tr + (seg * 8 + 2 * j) * TR_LD + pair * 2) = even;
          *reinterpret_cast<uint32_t*>(tr + (seg * 8 + 2 * j + 1) * TR_LD + pair * 2) = odd;
        }
      }
    }
  }
};

struct AttnBwdArgs {
  const half_t* q; long ldq, q_bs;
  const half_t* k; long ldk, k_bs;
  const half_t* v; long ldv, v_bs;
  const half_t* o; long ldo, o_bs;
  const half_t* go; long ldgo, go_bs;
  const float* lse;
  float* delta;
  half_t* gq; long ldgq, gq_bs;
  half_t* gk; long ldgk, gk_bs;
  half_t* gv; long ldgv, gv_bs;
  int B, H, Sq, Sk, d;
  int sk_grad;          // dK / dV are wanted for the first sk_grad keys only (<= Sk; dQ always sums over all Sk keys)
  float scale, scale_log2;
};

// A operand from a transposed LDS tile with the permuted contraction order used throughout:
// element j of lane group g <-> contraction index 32c + (j<4 ? g*4+j : 16+g*4+j-4).
__device__ __forceinline__ half8_t tr_frag(const half_t* tr, int row, int c, int g) {
  const half_t* p = tr + row * TR_LD + c * 32 + g * 4;
  half4_t lo = *reinterpret_cast<const half4_t*>(p);
  half4_t hi = *reinterpret_cast<const half4_t*>(p + 16);
  return (half8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// dQ.  Each wave owns QT x 16 queries (K / V / K^T fragments read from LDS feed QT MFMAs each);
// the next key tile is fetched into registers while the current one is multiplied.
// NW waves per workgroup (8: 256 queries share each staged key tile, as in the forward kernel); DB: two LDS stages and
// one barrier per key tile instead of store-after-barrier with two.
template <int DP, int QT, int NDT, int NW = 4, bool DB = false>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dq_kernel(const AttnBwdArgs a) {
  constexpr int K_LD = DP + 16;
  constexpr int NDC = DP / 32;   // NDT: 16-row output tiles that hold data (3 of DP/16 = 4 for d = 40)
  constexpr int STAGE = 2 * T64 * K_LD + DP * TR_LD;   // halfs: K, V row-major + K transposed
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  half_t* const smem = reinterpret_cast<half_t*>(dyn_smem);
  half_t* Ks = smem;
  half_t* Vs = Ks + T64 * K_LD;
  half_t* Kt = Vs + T64 * K_LD;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int d = a.d;
  const int q0 = blockIdx.x * (16 * NW * QT) + wid * (16 * QT);
  const half_t* Qb = a.q + (long)b * a.q_bs + (long)h * d;
  const half_t* Kb = a.k + (long)b * a.k_bs + (long)h * d;
  const half_t* Vb = a.v + (long)b * a.v_bs + (long)h * d;
  const half_t* Ob = a.o + (long)b * a.o_bs + (long)h * d;
  const half_t* GOb = a.go + (long)b * a.go_bs + (long)h * d;

  half8_t qf[QT][NDC], dof[QT][NDC];
  float delta[QT], nlse[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qrow = q0 + qt * 16 + c16;
    const bool q_ok = qrow < a.Sq;
    float dl = 0.f;
#pragma unroll
    for (int dc = 0; dc < NDC; ++dc) {
      const int dd = dc * 32 + g * 8;
      if (q_ok && dd < d) {
        qf[qt][dc] = *reinterpret_cast<const half8_t*>(Qb + (long)qrow * a.ldq + dd);
        dof[qt][dc] = *reinterpret_cast<const half8_t*>(GOb + (long)qrow * a.ldgo + dd);
        half8_t of = *reinterpret_cast<const half8_t*>(Ob + (long)qrow * a.ldo + dd);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)dof[qt][dc][e] * (float)of[e];
      } else {
        qf[qt][dc] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
        dof[qt][dc] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    const long stat_idx = ((long)b * a.H + h) * a.Sq + qrow;
    if (q_ok && g == 0) a.delta[stat_idx] = dl;
    delta[qt] = dl;
    nlse[qt] = q_ok ? -a.lse[stat_idx] : 0.f;
  }

  f32x4 dq[QT][NDT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) dq[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float sl2 = a.scale_log2;
  const int n_tiles = (a.Sk + T64 - 1) / T64;
  PairTile<DP, 64 * NW> kreg, vreg;
  kreg.load(Kb, a.ldk, 0, a.Sk, d);
  vreg.load(Vb, a.ldv, 0, a.Sk, d);
  kreg.template store<true, true>(Ks, Kt);
  vreg.template store<true, false>(Vs, nullptr);
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    const int kv0 = t * T64;
    if constexpr (DB) {
      Ks = smem + (t & 1) * STAGE;
      Vs = Ks + T64 * K_LD;
      Kt = Vs + T64 * K_LD;
    }
    if (t + 1 < n_tiles) {
      kreg.load(Kb, a.ldk, kv0 + T64, a.Sk, d);
      vreg.load(Vb, a.ldv, kv0 + T64, a.Sk, d);
    }
    f32x4 ds[QT][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      half8_t kf[NDC], vf[NDC];
#pragma unroll
      for (int dc = 0; dc < NDC; ++dc) {
        kf[dc] = *reinterpret_cast<const half8_t*>(Ks + (kt * 16 + c16) * K_LD + dc * 32 + g * 8);
        vf[dc] = *reinterpret_cast<const half8_t*>(Vs + (kt * 16 + c16) * K_LD + dc * 32 + g * 8);
      }
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dc = 0; dc < NDC; ++dc) {
          sv = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[dc], qf[qt][dc], sv, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[dc], dof[qt][dc], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(sv[r], sl2, nlse[qt]));
          ds[qt][kt][r] = pv * (dp[r] - delta[qt]);
        }
      }
    }
    if (kv0 + T64 > a.Sk) {  // ragged last tile: padded keys carry no gradient
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kv0 + kt * 16 + g * 4 + r >= a.Sk) ds[qt][kt][r] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      half8_t dsf[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dsf[qt][r] = (half_t)ds[qt][2 * c][r];
          dsf[qt][4 + r] = (half_t)ds[qt][2 * c + 1][r];
        }
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        half8_t ktf = tr_frag(Kt, dt * 16 + c16, c, g);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          dq[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ktf, dsf[qt], dq[qt][dt], 0, 0, 0);
      }
    }
    if constexpr (DB) {
      // the other stage was last read in iteration t-1, which every wave left through the barrier below
      if (t + 1 < n_tiles) {
        half_t* nK = smem + ((t + 1) & 1) * STAGE;
        kreg.template store<true, true>(nK, nK + 2 * T64 * K_LD);
        vreg.template store<true, false>(nK + T64 * K_LD, nullptr);
      }
      __syncthreads();
    } else {
      __syncthreads();
      if (t + 1 < n_tiles) {
        kreg.template store<true, true>(Ks, Kt);
        vreg.template store<true, false>(Vs, nullptr);
        __syncthreads();
      }
    }
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qrow = q0 + qt * 16 + c16;
    if (qrow < a.Sq) {
      half_t* out = a.gq + (long)b * a.gq_bs + (long)qrow * a.ldgq + (long)h * d;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const int dv = dt * 16 + g * 4;
        if (dv < d) {
          half4_t o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (half_t)(dq[qt][dt][r] * a.scale);
          *reinterpret_cast<half4_t*>(out + dv) = o;
        }
      }
    }
  }
}

// dK, dV.  Each wave owns KT x 16 keys; loops over query tiles (Q, dO row-major and transposed,
// lse and delta in LDS), next tile prefetched into registers.  Padded query rows are all-zero in
// Q and dO, so they contribute nothing whatever their recomputed probability is.
template <int DP, int KT, int NDT, int NW = 4, bool DB = false>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dkv_kernel(const AttnBwdArgs a) {
  constexpr int K_LD = DP + 16;
  constexpr int NDC = DP / 32;
  constexpr int STAGE = 2 * T64 * K_LD + 2 * DP * TR_LD + 4 * T64;   // halfs (the two float[64] tables = 4*64 halfs)
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  half_t* const smem = reinterpret_cast<half_t*>(dyn_smem);
  half_t* Qs = smem;
  half_t* Gs = Qs + T64 * K_LD;          // dO row-major
  half_t* Qt = Gs + T64 * K_LD;          // Q transposed [d][q]
  half_t* Gt = Qt + DP * TR_LD;          // dO transposed [dv][q]
  float* s_lse = reinterpret_cast<float*>(Gt + DP * TR_LD);  // -lse
  float* s_delta = s_lse + T64;
  auto set_stage = [&](int st) {
    Qs = smem + st * STAGE;
    Gs = Qs + T64 * K_LD;
    Qt = Gs + T64 * K_LD;
    Gt = Qt + DP * TR_LD;
    s_lse = reinterpret_cast<float*>(Gt + DP * TR_LD);
    s_delta = s_lse + T64;
  };

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int d = a.d;
  const int k0 = blockIdx.x * (16 * NW * KT) + wid * (16 * KT);
  const half_t* Qb = a.q + (long)b * a.q_bs + (long)h * d;
  const half_t* Kb = a.k + (long)b * a.k_bs + (long)h * d;
  const half_t* Vb = a.v + (long)b * a.v_bs + (long)h * d;
  const half_t* GOb = a.go + (long)b * a.go_bs + (long)h * d;
  const float* lse_b = a.lse + ((long)b * a.H + h) * a.Sq;
  const float* delta_b = a.delta + ((long)b * a.H + h) * a.Sq;

  half8_t kf[KT][NDC], vf[KT][NDC];
#pragma unroll
  for (int k2 = 0; k2 < KT; ++k2) {
    const int krow = k0 + k2 * 16 + c16;
#pragma unroll
    for (int dc = 0; dc < NDC; ++dc) {
      const int dd = dc * 32 + g * 8;
      if (krow < a.Sk && dd < d) {
        kf[k2][dc] = *reinterpret_cast<const half8_t*>(Kb + (long)krow * a.ldk + dd);
        vf[k2][dc] = *reinterpret_cast<const half8_t*>(Vb + (long)krow * a.ldv + dd);
      } else {
        kf[k2][dc] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
        vf[k2][dc] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
  }
  f32x4 dk[KT][NDT], dv[KT][NDT];
#pragma unroll
  for (int k2 = 0; k2 < KT; ++k2)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      dk[k2][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dv[k2][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  const float sl2 = a.scale_log2;
  const int n_tiles = (a.Sq + T64 - 1) / T64;
  PairTile<DP, 64 * NW> qreg, greg;
  float st_l = 0.f, st_d = 0.f;
  auto load_stats = [&](int q0) {
    if (tid < T64) {
      const int qq = q0 + tid;
      st_l = qq < a.Sq ? -lse_b[qq] : 0.f;
      st_d = qq < a.Sq ? delta_b[qq] : 0.f;
    }
  };
  auto store_all = [&]() {
    qreg.template store<true, true>(Qs, Qt);
    greg.template store<true, true>(Gs, Gt);
    if (tid < T64) { s_lse[tid] = st_l; s_delta[tid] = st_d; }
  };
  qreg.load(Qb, a.ldq, 0, a.Sq, d);
  greg.load(GOb, a.ldgo, 0, a.Sq, d);
  load_stats(0);
  store_all();
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    if constexpr (DB) set_stage(t & 1);
    if (t + 1 < n_tiles) {
      qreg.load(Qb, a.ldq, (t + 1) * T64, a.Sq, d);
      greg.load(GOb, a.ldgo, (t + 1) * T64, a.Sq, d);
      load_stats((t + 1) * T64);
    }
    f32x4 p[KT][4], ds[KT][4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      half8_t qa[NDC], ga[NDC];
#pragma unroll
      for (int dc = 0; dc < NDC; ++dc) {
        qa[dc] = *reinterpret_cast<const half8_t*>(Qs + (qt * 16 + c16) * K_LD + dc * 32 + g * 8);
        ga[dc] = *reinterpret_cast<const half8_t*>(Gs + (qt * 16 + c16) * K_LD + dc * 32 + g * 8);
      }
      const float4 nl = *reinterpret_cast<const float4*>(s_lse + qt * 16 + g * 4);
      const float4 dl = *reinterpret_cast<const float4*>(s_delta + qt * 16 + g * 4);
#pragma unroll
      for (int k2 = 0; k2 < KT; ++k2) {
        f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dc = 0; dc < NDC; ++dc) {
          sv = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[dc], kf[k2][dc], sv, 0, 0, 0);   // D[q][key]
          dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ga[dc], vf[k2][dc], dp, 0, 0, 0);   // D[q][key]
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float nlr = r == 0 ? nl.x : (r == 1 ? nl.y : (r == 2 ? nl.z : nl.w));
          const float dlr = r == 0 ? dl.x : (r == 1 ? dl.y : (r == 2 ? dl.z : dl.w));
          const float pv = __builtin_amdgcn_exp2f(fmaf(sv[r], sl2, nlr));
          p[k2][qt][r] = pv;
          ds[k2][qt][r] = pv * (dp[r] - dlr);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      half8_t pf[KT], dsf[KT];
#pragma unroll
      for (int k2 = 0; k2 < KT; ++k2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pf[k2][r] = (half_t)p[k2][2 * c][r];
          pf[k2][4 + r] = (half_t)p[k2][2 * c + 1][r];
          dsf[k2][r] = (half_t)ds[k2][2 * c][r];
          dsf[k2][4 + r] = (half_t)ds[k2][2 * c + 1][r];
        }
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        half8_t gtf = tr_frag(Gt, dt * 16 + c16, c, g);
        half8_t qtf = tr_frag(Qt, dt * 16 + c16, c, g);
#pragma unroll
        for (int k2 = 0; k2 < KT; ++k2) {
          dv[k2][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gtf, pf[k2], dv[k2][dt], 0, 0, 0);   // dV^T[dv][key]
          dk[k2][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qtf, dsf[k2], dk[k2][dt], 0, 0, 0);  // dK^T[j][key]
        }
      }
    }
    if constexpr (DB) {
      if (t + 1 < n_tiles) {
        set_stage((t + 1) & 1);
        store_all();
      }
      __syncthreads();
    } else {
      __syncthreads();
      if (t + 1 < n_tiles) {
        store_all();
        __syncthreads();
      }
    }
  }
#pragma unroll
  for (int k2 = 0; k2 < KT; ++k2) {
    const int krow = k0 + k2 * 16 + c16;
    if (krow < a.sk_grad) {
      half_t* outk = a.gk + (long)b * a.gk_bs + (long)krow * a.ldgk + (long)h * d;
      half_t* outv = a.gv + (long)b * a.gv_bs + (long)krow * a.ldgv + (long)h * d;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const int j = dt * 16 + g * 4;
        if (j < d) {
          half4_t ok_, ov_;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            ok_[r] = (half_t)(dk[k2][dt][r] * a.scale);
            ov_[r] = (half_t)dv[k2][dt][r];
          }
          *reinterpret_cast<half4_t*>(outk + j) = ok_;
          *reinterpret_cast<half4_t*>(outv + j) = ov_;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Cross-attention dQ, VALU fallback (96 < Sk <= 128; the 77-token case runs cross_attn_bwd_mfma_kernel).
// grid = (ceil(Sq/64), H, B): a wave walks 16 query rows; K and V of the head are staged in LDS as fp16
// [Sk][d+2] (odd dword stride: lanes walk rows conflict-free).
// ---------------------------------------------------------------------------------------------
struct CrossBwdArgs {
  const half_t* q; long ldq, q_bs;
  const half_t* k; long ldk, k_bs;
  const half_t* v; long ldv, v_bs;
  const half_t* go; long ldgo, go_bs;
  const float* gp;
  half_t* gq; long ldgq, gq_bs;
  int B, H, Sq, Sk, d;
  float scale;
};

constexpr int XB_MAXSK = 128;
constexpr int XB_MAXD = 192;
constexpr int XB_ROWS = 16;  // query rows per wave

__global__ __launch_bounds__(256) void cross_attn_bwd_kernel(const CrossBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  const int d = a.d, Sk = a.Sk;
  const int ld = d + 2;  // halfs; (d+2)/2 dwords is odd for every d % 8 == 0 -> conflict-free rows
  half_t* Ks = reinterpret_cast<half_t*>(dyn_smem);
  half_t* Vs = Ks + Sk * ld;
  float* s_q = reinterpret_cast<float*>(Vs + Sk * ld);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const half_t* Kb = a.k + (long)b * a.k_bs + (long)h * d;
  const half_t* Vb = a.v + (long)b * a.v_bs + (long)h * d;
  const int nseg = d / 8;
  for (int i = tid; i < Sk * nseg; i += 256) {
    int r = i / nseg, sg = i - r * nseg;
    uint4 kk = *reinterpret_cast<const uint4*>(Kb + (long)r * a.ldk + sg * 8);
    uint4 vv = *reinterpret_cast<const uint4*>(Vb + (long)r * a.ldv + sg * 8);
    uint32_t* kd = reinterpret_cast<uint32_t*>(Ks + r * ld + sg * 8);
    uint32_t* vd = reinterpret_cast<uint32_t*>(Vs + r * ld + sg * 8);
    kd[0] = kk.x; kd[1] = kk.y; kd[2] = kk.z; kd[3] = kk.w;
    vd[0] = vv.x; vd[1] = vv.y; vd[2] = vv.z; vd[3] = vv.w;
  }
  float* qv = s_q + wid * 2 * XB_MAXD;  // q row (fp32)
  float* gov = qv + XB_MAXD;            // dO row
  float* s_ds = s_q + 4 * 2 * XB_MAXD + wid * XB_MAXSK;
  for (int qi = 0; qi < XB_ROWS; ++qi) {
    const int qrow = (blockIdx.x * 4 + wid) * XB_ROWS + qi;
    const bool ok = qrow < a.Sq;
    if (ok) {
      const half_t* qp = a.q + (long)b * a.q_bs + (long)qrow * a.ldq + (long)h * d;
      const half_t* gp_ = a.go ? a.go + (long)b * a.go_bs + (long)qrow * a.ldgo + (long)h * d : nullptr;
      for (int c = lane; c < d; c += 64) {
        qv[c] = (float)qp[c];
        gov[c] = gp_ ? (float)gp_[c] : 0.f;
      }
    }
    __syncthreads();
    // each lane owns keys lane and lane+64
    float s[2], dp[2], p[2];
    float mx = -1.0e30f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int key = lane + r * 64;
      s[r] = -1.0e30f;
      dp[r] = 0.f;
      if (ok && key < Sk) {
        float acc = 0.f, accv = 0.f;
        const half2_t* kr = reinterpret_cast<const half2_t*>(Ks + key * ld);
        const half2_t* vr = reinterpret_cast<const half2_t*>(Vs + key * ld);
        for (int c2 = 0; c2 < d / 2; ++c2) {
          half2_t kk = kr[c2], vv = vr[c2];
          acc += qv[2 * c2] * (float)kk[0] + qv[2 * c2 + 1] * (float)kk[1];
          accv += gov[2 * c2] * (float)vv[0] + gov[2 * c2 + 1] * (float)vv[1];
        }
        s[r] = acc * a.scale;
        dp[r] = accv;
        if (a.gp) dp[r] += a.gp[(((long)b * a.H + h) * a.Sq + qrow) * Sk + key];
        mx = fmaxf(mx, s[r]);
      }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      p[r] = (ok && lane + r * 64 < Sk) ? __expf(s[r] - mx) : 0.f;
      sum += p[r];
    }
    sum = wave_sum(sum);
    const float inv = ok ? 1.f / sum : 0.f;
    float dot = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      p[r] *= inv;
      dot += p[r] * dp[r];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int key = lane + r * 64;
      if (key < Sk) s_ds[key] = p[r] * (dp[r] - dot) * a.scale;
    }
    __syncthreads();
    if (ok) {
      half_t* out = a.gq + (long)b * a.gq_bs + (long)qrow * a.ldgq + (long)h * d;
      for (int c = lane; c < d; c += 64) {
        float acc = 0.f;
        for (int key = 0; key < Sk; ++key) acc += s_ds[key] * (float)Ks[key * ld + c];
        out[c] = (half_t)acc;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Cross-attention dQ on the matrix cores (Sk <= 96, i.e. the 77 text tokens).  grid =
// (ceil(Sq/256), H, B); K, V (row-major) and K^T of the head are staged in LDS once per workgroup,
// then every wave walks 16-query tiles.  All <= 96 keys of a query sit in registers at once (6
// accumulator tiles, lane = query lane&15, keys kt*16 + (lane>>4)*4 + r), so the softmax is exact,
// not online:
//   S^T = K Q^T, dP^T = V gO^T (+ gP read from the fp32 map gradient), P = softmax(scale S),
//   dS = P o (dP - sum_k P dP) * scale, dQ^T = K^T dS^T (permuted contraction order of tr_frag).
// ---------------------------------------------------------------------------------------------
constexpr int XM_KEYS = 96;
constexpr int XM_LD = XM_KEYS + 8;  // halfs per row of the transposed K image

template <int DP>
__global__ __launch_bounds__(256) void cross_attn_bwd_mfma_kernel(const CrossBwdArgs a) {
  constexpr int K_LD = DP + 16;
  constexpr int NDC = DP / 32, NDT = DP / 16, KSEG = DP / 8;
  constexpr int NKT = XM_KEYS / 16;
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  half_t* Ks = reinterpret_cast<half_t*>(dyn_smem);
  half_t* Vs = Ks + XM_KEYS * K_LD;
  half_t* Kt = Vs + XM_KEYS * K_LD;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int d = a.d, Sk = a.Sk;
  const half_t* Kb = a.k + (long)b * a.k_bs + (long)h * d;
  const half_t* Vb = a.v + (long)b * a.v_bs + (long)h * d;

  // ---- stage K, V, K^T (zero padded to 96 keys x DP)
  for (int idx = tid; idx < XM_KEYS * KSEG; idx += 256) {
    const int row = idx / KSEG, seg = idx - row * KSEG;
    const bool ok = row < Sk && seg * 8 < d;
    const uint4 kk = ok ? *reinterpret_cast<const uint4*>(Kb + (long)row * a.ldk + seg * 8) : make_uint4(0, 0, 0, 0);
    const uint4 vv = ok ? *reinterpret_cast<const uint4*>(Vb + (long)row * a.ldv + seg * 8) : make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(Ks + row * K_LD + seg * 8) = kk;
    *reinterpret_cast<uint4*>(Vs + row * K_LD + seg * 8) = vv;
    const half_t* e = reinterpret_cast<const half_t*>(&kk);
#pragma unroll
    for (int j = 0; j < 8; ++j) Kt[(seg * 8 + j) * XM_LD + row] = e[j];
  }
  __syncthreads();

  const float sl2 = a.scale * 1.4426950408889634f;
  const int q_blk = blockIdx.x * 256;
  for (int it = 0; it < 4; ++it) {
    const int qrow = q_blk + it * 64 + wid * 16 + c16;
    if (q_blk + it * 64 + wid * 16 >= a.Sq) break;  // wave-uniform
    const bool q_ok = qrow < a.Sq;
    half8_t qf[NDC], gof[NDC];
#pragma unroll
    for (int dc = 0; dc < NDC; ++dc) {
      const int dd = dc * 32 + g * 8;
      qf[dc] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
      gof[dc] = qf[dc];
      if (q_ok && dd < d) {
        qf[dc] = *reinterpret_cast<const half8_t*>(a.q + (long)b * a.q_bs + (long)qrow * a.ldq + (long)h * d + dd);
        if (a.go)
          gof[dc] = *reinterpret_cast<const half8_t*>(a.go + (long)b * a.go_bs + (long)qrow * a.ldgo + (long)h * d + dd);
      }
    }
    f32x4 sv[NKT], dp[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dc = 0; dc < NDC; ++dc) {
        half8_t kf = *reinterpret_cast<const half8_t*>(Ks + (kt * 16 + c16) * K_LD + dc * 32 + g * 8);
        half8_t vf = *reinterpret_cast<const half8_t*>(Vs + (kt * 16 + c16) * K_LD + dc * 32 + g * 8);
        s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[dc], s, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, gof[dc], t, 0, 0, 0);
      }
      sv[kt] = s;
      dp[kt] = t;
    }
    // ---- exact softmax over the (masked) keys of query c16
    float mx = -1.0e30f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kt * 16 + g * 4 + r >= Sk) sv[kt][r] = -1.0e30f;
        mx = fmaxf(mx, sv[kt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float nm = -mx * sl2;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sv[kt][r], sl2, nm));
        sv[kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    if (a.gp && q_ok) {
      const float* gpr = a.gp + (((long)b * a.H + h) * a.Sq + qrow) * Sk;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + g * 4 + r;
          if (key < Sk) dp[kt][r] += gpr[key];
        }
    }
    float dot = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sv[kt][r] *= inv;
        dot += sv[kt][r] * dp[kt][r];
      }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    // ---- dQ^T = K^T dS^T
    f32x4 dq[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NKT / 2; ++c) {
      half8_t dsf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dsf[r] = (half_t)(sv[2 * c][r] * (dp[2 * c][r] - dot) * a.scale);
        dsf[4 + r] = (half_t)(sv[2 * c + 1][r] * (dp[2 * c + 1][r] - dot) * a.scale);
      }
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const half_t* pk = Kt + (dt * 16 + c16) * XM_LD + c * 32 + g * 4;
        half4_t lo = *reinterpret_cast<const half4_t*>(pk);
        half4_t hi = *reinterpret_cast<const half4_t*>(pk + 16);
        half8_t ktf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ktf, dsf, dq[dt], 0, 0, 0);
      }
    }
    if (q_ok) {
      half_t* out = a.gq + (long)b * a.gq_bs + (long)qrow * a.ldgq + (long)h * d;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const int dv = dt * 16 + g * 4;
        if (dv < d) {
          half4_t o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (half_t)dq[dt][r];
          *reinterpret_cast<half4_t*>(out + dv) = o;
        }
      }
    }
  }
}

template <int DP>
int launch_cross_bwd_mfma(const CrossBwdArgs& a, hipStream_t st) {
  const size_t smem = (size_t)(2 * XM_KEYS * (DP + 16) + DP * XM_LD) * 2;
  static const bool attr_set = [smem] {      // thread-safe one-time initialisation (lane threads launch concurrently)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_attn_bwd_mfma_kernel<DP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    return true;
  }();
  (void)attr_set;
  hipLaunchKernelGGL((cross_attn_bwd_mfma_kernel<DP>), dim3((a.Sq + 255) / 256, a.H, a.B), dim3(256), smem, st, a);
  return lgd_check_launch();
}

template <int DP, int NQ, int NK, int NDT, int NW = 4, bool DB = false>
int launch_bwd_nt(const AttnBwdArgs& a, hipStream_t st) {
  constexpr int K_LD = DP + 16;
  constexpr int NST = DB ? 2 : 1;
  const size_t smem_dq = (size_t)NST * (2 * T64 * K_LD + DP * TR_LD) * 2;
  const size_t smem_dkv = (size_t)NST * ((2 * T64 * K_LD + 2 * DP * TR_LD) * 2 + 2 * T64 * 4);
  static const bool attr_set = [smem_dq, smem_dkv] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<DP, NK, NDT, NW, DB>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_dkv);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<DP, NQ, NDT, NW, DB>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_dq);
    return true;
  }();
  (void)attr_set;
  hipLaunchKernelGGL((attn_bwd_dq_kernel<DP, NQ, NDT, NW, DB>), dim3((a.Sq + 16 * NW * NQ - 1) / (16 * NW * NQ), a.H, a.B),
                     dim3(64 * NW), smem_dq, st, a);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<DP, NK, NDT, NW, DB>), dim3((a.sk_grad + 16 * NW * NK - 1) / (16 * NW * NK), a.H, a.B),
                     dim3(64 * NW), smem_dkv, st, a);
  return lgd_check_launch();
}

template <int DP>
int launch_bwd(const AttnBwdArgs& a, hipStream_t st) {
  // two 16-row tiles per wave once there are enough 128-row workgroups to fill the chip
  const long wgs = (long)((a.Sq < a.Sk ? a.Sq : a.Sk) / 128) * a.H * a.B;
  // LGD_ATTN_BWD (tools, A/B): 0 = single-buffered 4-wave kernels of round 1, 1 = double-buffered, 2 = double-buffered 8 waves.
  // Measured (tools/attn_bwd_quick.py, same box): d = 40, S = 4096: 607 -> 587 (1) -> 547 us (2); with the fuser's
  // 4126 keys 679 -> 653 -> 655; d = 80 and d = 160 unchanged; d = 64 at S = 9216 (SD2.1): 1067 -> 1199 us with two stages
  // (79 KB of LDS per workgroup halves the resident workgroups) — so only the narrow-head case takes the new variants.
  static const int env = [] { const char* e = getenv("LGD_ATTN_BWD"); return e ? atoi(e) : -2; }();
  const bool narrow = env == -2;           // default: new variants for d <= 48 only
  const int mode = narrow ? 0 : env;
  if constexpr (DP == 64) {
    if (a.d <= 48) {  // d = 40: the fourth 16-row tile of dQ / dK / dV would be all padding
      if (wgs >= 512) {
        if ((narrow || mode == 2) && wgs >= 1024) return launch_bwd_nt<DP, 2, 2, 3, 8, true>(a, st);
        if (narrow || mode >= 1) return launch_bwd_nt<DP, 2, 2, 3, 4, true>(a, st);
        return launch_bwd_nt<DP, 2, 2, 3>(a, st);
      }
      return launch_bwd_nt<DP, 1, 1, 3>(a, st);
    }
  }
  if constexpr (DP <= 96) {
    if (wgs >= 512) {
      // (8-wave double-buffered kernels exist for DP = 64 only: at DP = 96 the dK/dV kernel spilled 92 registers)
      if constexpr (DP == 64)
        if (mode == 2 && wgs >= 1024) return launch_bwd_nt<DP, 2, 2, DP / 16, 8, true>(a, st);
      if (mode >= 1) return launch_bwd_nt<DP, 2, 2, DP / 16, 4, true>(a, st);
      return launch_bwd_nt<DP, 2, 2, DP / 16>(a, st);
    }
  }
  return launch_bwd_nt<DP, 1, 1, DP / 16>(a, st);
}

}  // namespace

extern "C" int lgd_attn_bwd_keys_f16(const void* q, int64_t ldq, int64_t q_bs, const void* k,
                                     int64_t ldk, int64_t k_bs, const void* v, int64_t ldv, int64_t v_bs,
                                     const void* o, int64_t ldo, int64_t o_bs, const void* go,
                                     int64_t ldgo, int64_t go_bs, const float* lse, float* delta, void* gq,
                                     int64_t ldgq, int64_t gq_bs, void* gk, int64_t ldgk, int64_t gk_bs,
                                     void* gv, int64_t ldgv, int64_t gv_bs, int B, int H, int Sq, int Sk,
                                     int Sk_grad, int d, float scale, void* stream);

extern "C" int lgd_attn_bwd_f16(const void* q, int64_t ldq, int64_t q_bs, const void* k,
                                int64_t ldk, int64_t k_bs, const void* v, int64_t ldv, int64_t v_bs,
                                const void* o, int64_t ldo, int64_t o_bs, const void* go,
                                int64_t ldgo, int64_t go_bs, const float* lse, float* delta, void* gq,
                                int64_t ldgq, int64_t gq_bs, void* gk, int64_t ldgk, int64_t gk_bs,
                                void* gv, int64_t ldgv, int64_t gv_bs, int B, int H, int Sq, int Sk,
                                int d, float scale, void* stream) {
  return lgd_attn_bwd_keys_f16(q, ldq, q_bs, k, ldk, k_bs, v, ldv, v_bs, o, ldo, o_bs, go, ldgo, go_bs, lse, delta, gq, ldgq,
                               gq_bs, gk, ldgk, gk_bs, gv, ldgv, gv_bs, B, H, Sq, Sk, Sk, d, scale, stream);
}

extern "C" int lgd_attn_bwd_keys_f16(const void* q, int64_t ldq, int64_t q_bs, const void* k,
                                     int64_t ldk, int64_t k_bs, const void* v, int64_t ldv, int64_t v_bs,
                                     const void* o, int64_t ldo, int64_t o_bs, const void* go,
                                     int64_t ldgo, int64_t go_bs, const float* lse, float* delta, void* gq,
                                     int64_t ldgq, int64_t gq_bs, void* gk, int64_t ldgk, int64_t gk_bs,
                                     void* gv, int64_t ldgv, int64_t gv_bs, int B, int H, int Sq, int Sk,
                                     int Sk_grad, int d, float scale, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (B < 1 || H < 1 || Sq < 1 || Sk < 1 || Sk_grad < 1 || Sk_grad > Sk || d < 8 || (d % 8)) return LGD_ERR_ARG;
  if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 8) || (ldgo % 8) || (ldgq % 4) || (ldgk % 4) ||
      (ldgv % 4) || !lse || !delta)
    return LGD_ERR_ARG;
  AttnBwdArgs a;
  a.q = (const half_t*)q; a.ldq = ldq; a.q_bs = q_bs;
  a.k = (const half_t*)k; a.ldk = ldk; a.k_bs = k_bs;
  a.v = (const half_t*)v; a.ldv = ldv; a.v_bs = v_bs;
  a.o = (const half_t*)o; a.ldo = ldo; a.o_bs = o_bs;
  a.go = (const half_t*)go; a.ldgo = ldgo; a.go_bs = go_bs;
  a.lse = lse; a.delta = delta;
  a.gq = (half_t*)gq; a.ldgq = ldgq; a.gq_bs = gq_bs;
  a.gk = (half_t*)gk; a.ldgk = ldgk; a.gk_bs = gk_bs;
  a.gv = (half_t*)gv; a.ldgv = ldgv; a.gv_bs = gv_bs;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.d = d;
  a.sk_grad = Sk_grad;
  a.scale = scale;
  a.scale_log2 = scale * 1.4426950408889634f;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d <= 32) return launch_bwd<32>(a, st);
  if (d <= 64) return launch_bwd<64>(a, st);
  if (d <= 96) return launch_bwd<96>(a, st);
  if (d <= 128) return launch_bwd<128>(a, st);
  if (d <= 160) return launch_bwd<160>(a, st);
  return LGD_ERR_UNSUPPORTED;
}

extern "C" int lgd_cross_attn_bwd_f16(const void* q, int64_t ldq, int64_t q_bs, const void* k,
                                      int64_t ldk, int64_t k_bs, const void* v, int64_t ldv,
                                      int64_t v_bs, const void* go, int64_t ldgo, int64_t go_bs,
                                      const float* gp, void* gq, int64_t ldgq, int64_t gq_bs, int B,
                                      int H, int Sq, int Sk, int d, float scale, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (B < 1 || H < 1 || Sq < 1 || Sk < 1 || Sk > XB_MAXSK || d < 8 || (d % 8) || d > XB_MAXD ||
      (ldq % 1) || (ldk % 8) || (ldv % 8))
    return LGD_ERR_ARG;
  CrossBwdArgs a;
  a.q = (const half_t*)q; a.ldq = ldq; a.q_bs = q_bs;
  a.k = (const half_t*)k; a.ldk = ldk; a.k_bs = k_bs;
  a.v = (const half_t*)v; a.ldv = ldv; a.v_bs = v_bs;
  a.go = (const half_t*)go; a.ldgo = ldgo; a.go_bs = go_bs;
  a.gp = gp;
  a.gq = (half_t*)gq; a.ldgq = ldgq; a.gq_bs = gq_bs;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.d = d; a.scale = scale;
  hipStream_t st_ = reinterpret_cast<hipStream_t>(stream);
  if (Sk <= XM_KEYS && d <= 160 && (ldq % 8) == 0 && (ldgo % 8) == 0 && (ldgq % 4) == 0) {
    if (d <= 32) return launch_cross_bwd_mfma<32>(a, st_);
    if (d <= 64) return launch_cross_bwd_mfma<64>(a, st_);
    if (d <= 96) return launch_cross_bwd_mfma<96>(a, st_);
    if (d <= 128) return launch_cross_bwd_mfma<128>(a, st_);
    return launch_cross_bwd_mfma<160>(a, st_);
  }
  const int ld = d + 2;
  size_t smem = (size_t)(2 * Sk * ld + 2) * 2 + (size_t)4 * 2 * XB_MAXD * 4 + (size_t)4 * XB_MAXSK * 4;
  smem = (smem + 15) & ~(size_t)15;
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_attn_bwd_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr_set;
  hipLaunchKernelGGL(cross_attn_bwd_kernel, dim3((Sq + 4 * XB_ROWS - 1) / (4 * XB_ROWS), H, B), dim3(256), smem,
                     reinterpret_cast<hipStream_t>(stream), a);
  return lgd_check_launch();
}
