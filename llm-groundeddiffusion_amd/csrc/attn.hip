// Flash-style scaled-dot-product attention for gfx950 — forward.
//
// Two kernels: attn_fwd_kernel (cross-attention WITH probability-map capture: exact two-pass softmax) and
// attn_self_kernel (everything without map capture: self-attention, GLIGEN fuser attention, plain
// cross-attention).  Both: 4 waves per workgroup, each wave owns 16 (or 2x16) query rows; K/V are
// consumed in tiles of 64 keys staged in LDS (register-prefetched one tile ahead).
//
// MFMA layout trick (no LDS round trip for P): both products are issued transposed,
//     S^T = K Q^T   (A := K rows,  B := Q rows)   -> lane holds query (lane&15), 4 keys per tile
//     O^T = V^T P^T (A := V^T rows, B := P^T)      -> lane holds query (lane&15), 4 dv per tile
// so the softmax statistics of a query live in the lanes {q, q+16, q+32, q+48} (two xor-shuffles
// for the row max), the rescale factor is a per-lane scalar, and the S^T accumulators convert to
// the P^T operand in registers.  The contraction index of the second MFMA is permuted the same way
// on both operands: element j of lane group g is key 32c + (j<4 ? g*4+j : 16+g*4+j-4), which is
// why V is staged transposed ([dv][key]) and read as two 8-byte pieces.
//
// Map capture (attention_processor.py:440-480): two passes over the keys — pass 1 row max / row sum,
// pass 2 normalised probabilities, which are written to the fp32 map and fed to the PV product.
#include <atomic>
#include "common.h"
#include "../../include/lgd_hip.h"
#include "attn_w4.h"
#include <stdlib.h>
#include <string.h>

namespace {

int g_attn32_nw = -1;     // waves per workgroup of the 32x32x16 kernel: 8 (256 queries per workgroup) or 4
int attn32_nw() {
  if (g_attn32_nw < 0) { const char* e = getenv("LGD_ATTN32_NW"); g_attn32_nw = e ? atoi(e) : 8; }
  return g_attn32_nw;
}
int g_attn32_var = 0;      // tools: 0 = fragment prefetch 2 slots ahead, pinned slot order; 1 = 4 ahead; 2 = compiler's order
int attn32_var() { return g_attn32_var; }
int g_attn32 = -1;
int attn32_mode() {
  if (g_attn32 < 0) { const char* e = getenv("LGD_ATTN32"); g_attn32 = e ? atoi(e) : 1; }
  return g_attn32;
}

// round-4 kernel for d = 40 (attn_w4.hip): 1 = default, 0 = never (A/B timing), 2 = for every size (tests).  Read by
// every lane thread at launch time: an atomic whose first reader takes LGD_ATTN_W4 once (no torn lazy initialisation)
std::atomic<int> g_attn_w4{-1};
int attn_w4_mode() {
  int v = g_attn_w4.load(std::memory_order_relaxed);
  if (v < 0) {
    static const int env = [] { const char* e = getenv("LGD_ATTN_W4"); return e ? atoi(e) : 1; }();
    int expect = -1;
    g_attn_w4.compare_exchange_strong(expect, env, std::memory_order_relaxed);
    v = g_attn_w4.load(std::memory_order_relaxed);
  }
  return v;
}

constexpr int KV_T = 64;         // keys per tile
constexpr int VT_LD = KV_T + 8;  // halfs per row of the transposed V tile (ds_read_b64: conflict-free)
// Row-major K tiles use DP + 16 halfs per row: the only padding <= 48 for which the 4 hardware lane
// groups of a ds_read_b128 fragment read ({0-3,12-15,20-27}, ...) each hit 16 distinct 16-B slots
// (DP + 8 is 2-way conflicted: 9 % of wave cycles in the PMC profile).
constexpr float NEG_BIG = -1.0e30f;

struct AttnArgs {
  const half_t* q; long ldq, q_bs;
  const half_t* k; long ldk, k_bs;
  const half_t* v; long ldv, v_bs;
  half_t* o; long ldo, o_bs;
  float* lse;
  float* probs; int tok; int cond_only;
  int causal;        // attn_fwd_kernel only: key j attends to query i iff j <= i (CLIP text encoder)
  int B, H, Sq, Sk, d;
  float scale_log2;  // scale * log2(e)
};

// Map-capture kernel (cross-attention with a saved probability map).  Attention WITHOUT map capture
// goes through attn_self_kernel below.
template <int DP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
  constexpr int K_LD = DP + 16;
  constexpr int NDC = DP / 32;  // 32-wide chunks of the head dim (QK^T contraction)
  constexpr int NDT = DP / 16;  // 16-row tiles of dv (O^T rows)
  constexpr int KSEG = DP / 8;  // 16-byte segments per K row
  constexpr int K_IT = (KV_T * KSEG + 255) / 256;
  constexpr int V_ITEMS = (KV_T / 2) * KSEG;  // (key pair, segment)
  constexpr int V_IT = (V_ITEMS + 255) / 256;

  __shared__ __attribute__((aligned(16))) half_t smem[KV_T * K_LD + DP * VT_LD];
  half_t* Ks = smem;
  half_t* Vt = smem + KV_T * K_LD;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 64 + wid * 16;
  const int d = a.d;
  const half_t* Qb = a.q + (long)b * a.q_bs + (long)h * d;
  const half_t* Kb = a.k + (long)b * a.k_bs + (long)h * d;
  const half_t* Vb = a.v + (long)b * a.v_bs + (long)h * d;

  // ---- Q fragments (B operand of S^T): lane -> query c16, head-dim elements dc*32 + g*8..+8
  half8_t qf[NDC];
  {
    const int qrow = q0 + c16;
#pragma unroll
    for (int dc = 0; dc < NDC; ++dc) {
      const int dd = dc * 32 + g * 8;
      if (qrow < a.Sq && dd < d)
        qf[dc] = *reinterpret_cast<const half8_t*>(Qb + (long)qrow * a.ldq + dd);
      else
        qf[dc] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  uint4 k_reg[K_IT], v_reg[V_IT][2];
  auto load_tile = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
      int idx = tid + i * 256;
      int row = idx / KSEG, seg = idx - row * KSEG;
      bool ok = (idx < KV_T * KSEG) && (kv0 + row < a.Sk) && (seg * 8 < d);
      k_reg[i] = ok ? *reinterpret_cast<const uint4*>(Kb + (long)(kv0 + row) * a.ldk + seg * 8)
                    : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
      int idx = tid + i * 256;
      int pair = idx & 31, seg = idx >> 5;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        int row = pair * 2 + r;
        bool ok = (idx < V_ITEMS) && (kv0 + row < a.Sk) && (seg * 8 < d);
        v_reg[i][r] = ok ? *reinterpret_cast<const uint4*>(Vb + (long)(kv0 + row) * a.ldv + seg * 8)
                         : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
      int idx = tid + i * 256;
      if (idx < KV_T * KSEG) {
        int row = idx / KSEG, seg = idx - row * KSEG;
        *reinterpret_cast<uint4*>(Ks + row * K_LD + seg * 8) = k_reg[i];
      }
    }
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
      int idx = tid + i * 256;
      if (idx < V_ITEMS) {
        int pair = idx & 31, seg = idx >> 5;
        const half_t* e0 = reinterpret_cast<const half_t*>(&v_reg[i][0]);
        const half_t* e1 = reinterpret_cast<const half_t*>(&v_reg[i][1]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          half2_t pr = {e0[e], e1[e]};
          *reinterpret_cast<half2_t*>(Vt + (seg * 8 + e) * VT_LD + pair * 2) = pr;
        }
      }
    }
  };

  // S^T for the current LDS tile, scaled to the log2 domain and masked.
  const int key_end = a.causal ? min(a.Sk, q0 + c16 + 1) : a.Sk;   // first key this lane's query does not see
  auto compute_s = [&](int kv0, f32x4 (&s)[4]) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dc = 0; dc < NDC; ++dc) {
        half8_t kf =
            *reinterpret_cast<const half8_t*>(Ks + (kt * 16 + c16) * K_LD + dc * 32 + g * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[dc], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = kv0 + kt * 16 + g * 4 + r;
        s[kt][r] = key < key_end ? acc[r] * a.scale_log2 : NEG_BIG;
      }
    }
  };

  f32x4 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;

  auto pv = [&](const f32x4 (&p)[4]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      half8_t pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pf[r] = (half_t)p[2 * c][r];
        pf[4 + r] = (half_t)p[2 * c + 1][r];
      }
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const half_t* vrow = Vt + (dt * 16 + c16) * VT_LD + c * 32 + g * 4;
        half4_t lo = *reinterpret_cast<const half4_t*>(vrow);
        half4_t hi = *reinterpret_cast<const half4_t*>(vrow + 16);
        half8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, oacc[dt], 0, 0, 0);
      }
    }
  };

  const int n_tiles = (a.Sk + KV_T - 1) / KV_T;

  {
    // ---- pass 1: exact row max and row sum
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
      if (t + 1 < n_tiles) load_tile((t + 1) * KV_T);
      f32x4 s[4];
      compute_s(t * KV_T, s);
      float mx = NEG_BIG;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float m_new = fmaxf(m_run, mx);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += exp2f(s[kt][r] - m_new);
      l_run = l_run * exp2f(m_run - m_new) + sum;
      m_run = m_new;
      __syncthreads();
      if (t + 1 < n_tiles) {
        store_tile();
        __syncthreads();
      }
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv_l = 1.f / l_run;
    // ---- pass 2: normalised probabilities -> map + PV
    const int qrow = q0 + c16;
    const bool store_b = !a.cond_only || b >= a.B / 2;
    const int bp = a.cond_only ? b - a.B / 2 : b;
    const int Tp = a.tok >= 0 ? 1 : a.Sk;
    float* prow = a.probs ? a.probs + (((long)bp * a.H + h) * a.Sq + qrow) * Tp : nullptr;
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
      if (t + 1 < n_tiles) load_tile((t + 1) * KV_T);
      f32x4 s[4];
      compute_s(t * KV_T, s);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = exp2f(s[kt][r] - m_run) * inv_l;
          s[kt][r] = p;
          int key = t * KV_T + kt * 16 + g * 4 + r;
          if (prow && store_b && qrow < a.Sq && key < a.Sk) {
            if (a.tok < 0) prow[key] = p;
            else if (key == a.tok) prow[0] = p;
          }
        }
      pv(s);
      __syncthreads();
      if (t + 1 < n_tiles) {
        store_tile();
        __syncthreads();
      }
    }
    l_run = 1.f;  // already normalised
  }

  // ---- epilogue: lane owns query q0+c16, dv = dt*16 + g*4 + r
  const int qrow = q0 + c16;
  if (qrow < a.Sq) {
    const float inv = 1.f / l_run;
    half_t* orow = a.o + (long)b * a.o_bs + (long)qrow * a.ldo + (long)h * d;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int dv = dt * 16 + g * 4;
      if (dv < d) {
        half4_t o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)(oacc[dt][r] * inv);
        *reinterpret_cast<half4_t*>(orow + dv) = o;
      }
    }
  }
}

// max over the four lanes {q, q+16, q+32, q+48} that share a query in the 16x16 accumulator layout, delivered to all
// four, without LDS (the ds_bpermute pair this replaces cost two LDS round trips per query tile and key tile):
// v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of the other, v_permlane32_swap
// the upper half of one with the lower half of the other.  Inline asm for the reason given at half_pair_max.
__device__ __forceinline__ float quad_row_max(float mx) {
  float a = mx, b = mx;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  a = fmaxf(a, b);
  b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

// ---------------------------------------------------------------------------------------------
// Self-attention forward (no map capture): the kernel the UNet spends most of its attention time
// in (S = 4096, d = 40 at the 64x64 level).  Same transposed-product layout as attn_fwd_kernel, but
//   * each wave owns QT x 16 queries, so every K / V^T fragment read from LDS feeds QT MFMAs and
//     the tile staging cost is split over QT x 64 queries per workgroup;
//   * two LDS stages, one barrier per key tile; staging pointers and predicates are hoisted, the
//     head-dim padding (columns d..DP of K, rows d..DP of V^T) is written once, not per tile;
//   * softmax pared down to max3 / fma / v_exp_f32 / packed convert: the reference exponent m_ref
//     is only raised when a row's tile max exceeds it by more than 2^8 (wave-uniform vote), the
//     row sum is produced by the PV MFMA from a row of ones at V^T row d (ONES, needs d < DP).
// NDT: 16-row tiles of V^T / O^T actually multiplied (d = 40 with the ones row needs 3, not DP/16 = 4).
// NW: waves per workgroup (4 or 8).  With 8 waves the K / V^T tile of a key block is staged once for 8 x QT x 16
// queries, so the per-wave share of the staging loads, the transposing LDS stores and the LDS footprint halves.
template <int DP, bool ONES, int QT, int NDT = DP / 16, int NW = 4>
__global__ __launch_bounds__(64 * NW) void attn_self_kernel(const AttnArgs a) {
  constexpr int NT = 64 * NW;
  constexpr int K_LD = DP + 16;
  constexpr int NDC = DP / 32;
  constexpr int KSEG = DP / 8;
  constexpr int K_IT = (KV_T * KSEG + NT - 1) / NT;
  constexpr int V_ITEMS = (KV_T / 2) * KSEG;
  constexpr int V_IT = (V_ITEMS + NT - 1) / NT;
  // V^T rows: 80 halfs = ten 16-byte slots (like the K rows): with that stride the four hardware lane groups of a
  // ds_read_b128 fragment read hit 16 distinct slots.  Inside each 32-key block the keys are stored in the order the
  // P^T operand wants them (position 8 g + 4 a + j holds key 16 a + 4 g + j), so a fragment is ONE 16-byte read
  // (it was a ds_read2_b64: half the LDS rate, 2-way conflicted under its 32-bank rule — 30 % of the LDS cycles)
  constexpr int VT_LD = KV_T + 16;
  constexpr int STAGE = KV_T * K_LD + NDT * 16 * VT_LD;

  __shared__ __attribute__((aligned(16))) half_t smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (16 * QT * NW) + wid * (16 * QT);
  const int d = a.d;
  const half_t* Qb = a.q + (long)b * a.q_bs + (long)h * d;
  const half_t* Kb = a.k + (long)b * a.k_bs + (long)h * d;
  const half_t* Vb = a.v + (long)b * a.v_bs + (long)h * d;

  // ---- one-time LDS fill: zeros everywhere (padding columns / rows must not hold NaN bit
  // patterns), ones at V^T row d of both stages.
  for (int i = tid; i < 2 * STAGE / 8; i += NT)
    reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (ONES && tid < 2 * VT_LD) {
    const int st = tid / VT_LD, col = tid - st * VT_LD;
    smem[st * STAGE + KV_T * K_LD + d * VT_LD + col] = (half_t)1.f;        // V^T row d  -> row sums
    if (col < KV_T) smem[st * STAGE + col * K_LD + d] = (half_t)1.f;         // K column d -> -m_ref term
  }

  // ---- Q fragments
  half8_t qf[QT][NDC];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qrow = q0 + qt * 16 + c16;
#pragma unroll
    for (int dc = 0; dc < NDC; ++dc) {
      const int dd = dc * 32 + g * 8;
      if (qrow < a.Sq && dd < d)
        qf[qt][dc] = *reinterpret_cast<const half8_t*>(Qb + (long)qrow * a.ldq + dd);
      else
        qf[qt][dc] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
      if (ONES) {  // scores come out of the MFMA already in the log2 domain
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[qt][dc][e] = (half_t)((float)qf[qt][dc][e] * a.scale_log2);
      }
    }
  }
  // ONES: head-dim slot d of Q carries -m_ref (K holds 1 there), so the MFMA result is already
  // s * scale * log2(e) - m_ref.  The slot lives in fragment dc_m, lane group g_m, element 0.
  const int dc_m = d >> 5, g_m = (d & 31) >> 3;
  auto set_ref = [&](int qt, float m) {
    if (g == g_m) {
#pragma unroll
      for (int dc = 0; dc < NDC; ++dc)
        if (dc == dc_m) qf[qt][dc][0] = (half_t)(-m);
    }
  };

  // ---- hoisted staging coordinates
  const half_t* kp[K_IT];
  int k_row[K_IT], k_dst[K_IT];
  bool k_use[K_IT];
#pragma unroll
  for (int i = 0; i < K_IT; ++i) {
    const int idx = tid + i * NT;
    const int row = idx / KSEG, seg = idx - row * KSEG;
    k_use[i] = (idx < KV_T * KSEG) && (seg * 8 < d);
    k_row[i] = row;
    k_dst[i] = row * K_LD + seg * 8;
    kp[i] = Kb + (long)row * a.ldk + seg * 8;
  }
  const half_t* vp[V_IT];
  int v_row[V_IT], v_dst[V_IT];
  bool v_use[V_IT];
#pragma unroll
  for (int i = 0; i < V_IT; ++i) {
    const int idx = tid + i * NT;
    const int pair = idx & 31, seg = idx >> 5;
    v_use[i] = (idx < V_ITEMS) && (seg * 8 < d);
    v_row[i] = pair * 2;
    const int key = pair * 2;
    const int pos = (key & ~31) | (((key >> 2) & 3) << 3) | (((key >> 4) & 1) << 2) | (key & 3);
    v_dst[i] = KV_T * K_LD + (seg * 8) * VT_LD + pos;
    vp[i] = Vb + (long)(pair * 2) * a.ldv + seg * 8;
  }

  uint4 k_reg[K_IT], v_reg[V_IT][2];
  auto load_tile = [&](int kv0) {
    const bool full = kv0 + KV_T <= a.Sk;
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
      const bool ok = k_use[i] && (full || kv0 + k_row[i] < a.Sk);
      k_reg[i] = ok ? *reinterpret_cast<const uint4*>(kp[i]) : make_uint4(0, 0, 0, 0);
      kp[i] += (long)KV_T * a.ldk;
    }
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const bool ok = v_use[i] && (full || kv0 + v_row[i] + r < a.Sk);
        v_reg[i][r] = ok ? *reinterpret_cast<const uint4*>(vp[i] + (long)r * a.ldv)
                         : make_uint4(0, 0, 0, 0);
      }
      vp[i] += (long)KV_T * a.ldv;
    }
  };
  auto store_tile = [&](int stage) {
    half_t* base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < K_IT; ++i)
      if (k_use[i]) *reinterpret_cast<uint4*>(base + k_dst[i]) = k_reg[i];
#pragma unroll
    for (int i = 0; i < V_IT; ++i)
      if (v_use[i]) {
        const half_t* e0 = reinterpret_cast<const half_t*>(&v_reg[i][0]);
        const half_t* e1 = reinterpret_cast<const half_t*>(&v_reg[i][1]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          half2_t pr = {e0[e], e1[e]};
          *reinterpret_cast<half2_t*>(base + v_dst[i] + e * VT_LD) = pr;
        }
      }
  };

  f32x4 oacc[QT][NDT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) oacc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_ref[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { m_ref[qt] = NEG_BIG; l_run[qt] = 0.f; }
  const float sc = a.scale_log2;
  const int n_tiles = (a.Sk + KV_T - 1) / KV_T;

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    if (t + 1 < n_tiles) load_tile((t + 1) * KV_T);
    const half_t* Ks = smem + (t & 1) * STAGE;
    const half_t* Vt = Ks + KV_T * K_LD;
    const int kv0 = t * KV_T;
    // ---- S^T = K Q^T (raw, unscaled)
    f32x4 s[QT][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      half8_t kf[NDC];
#pragma unroll
      for (int dc = 0; dc < NDC; ++dc)
        kf[dc] = *reinterpret_cast<const half8_t*>(Ks + (kt * 16 + c16) * K_LD + dc * 32 + g * 8);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dc = 0; dc < NDC; ++dc)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[dc], qf[qt][dc], acc, 0, 0, 0);
        s[qt][kt] = acc;
      }
    }
    if (kv0 + KV_T > a.Sk) {  // ragged last tile
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kv0 + kt * 16 + g * 4 + r >= a.Sk) s[qt][kt][r] = NEG_BIG;
    }
    // ---- tile max per query, reference update vote
    float mxs[QT];
    bool raise = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = fmaxf(fmaxf(s[qt][0][0], s[qt][0][1]), s[qt][0][2]);
      mx = fmaxf(fmaxf(mx, s[qt][0][3]), s[qt][1][0]);
      mx = fmaxf(fmaxf(mx, s[qt][1][1]), s[qt][1][2]);
      mx = fmaxf(fmaxf(mx, s[qt][1][3]), s[qt][2][0]);
      mx = fmaxf(fmaxf(mx, s[qt][2][1]), s[qt][2][2]);
      mx = fmaxf(fmaxf(mx, s[qt][2][3]), s[qt][3][0]);
      mx = fmaxf(fmaxf(mx, s[qt][3][1]), s[qt][3][2]);
      mx = fmaxf(mx, s[qt][3][3]);
      mx = quad_row_max(mx);
      if (ONES) {
        mxs[qt] = mx;  // already relative to m_ref
        raise = raise || (mx > 8.f) || (t == 0);
      } else {
        mxs[qt] = mx * sc;
        raise = raise || (mxs[qt] > m_ref[qt] + 8.f);
      }
    }
    if (__builtin_amdgcn_ballot_w64(raise) != 0) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        if (ONES) {
          // new reference, rounded to fp16 so that the Q slot holds it exactly (first tile: the
          // slot is still 0, i.e. the scores are absolute)
          const float m_old = t == 0 ? 0.f : m_ref[qt];
          const float m_abs = mxs[qt] + m_old;
          const float m_new = (float)(half_t)(t == 0 ? m_abs : fmaxf(m_old, m_abs));
          const float shift = m_new - m_old;
          // first tile: the accumulators are still zero and the shift is the absolute reference, which may be far
          // below -128 (exp2 -> +inf, 0 * inf = NaN): nothing to rescale
          const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-shift);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[qt][dt][r] *= alpha;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[qt][kt][r] -= shift;
          m_ref[qt] = m_new;
          set_ref(qt, m_new);
        } else {
          const float m_new = fmaxf(m_ref[qt], mxs[qt]);
          const float alpha = __builtin_amdgcn_exp2f(m_ref[qt] - m_new);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[qt][dt][r] *= alpha;
          l_run[qt] *= alpha;
          m_ref[qt] = m_new;
        }
      }
    }
    // ---- P = exp2(.), packed to the fp16 B operand of the PV product
    half8_t pf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const float nm = -m_ref[qt];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p;
          if (ONES) {
            p = __builtin_amdgcn_exp2f(s[qt][kt][r]);
          } else {
            p = __builtin_amdgcn_exp2f(fmaf(s[qt][kt][r], sc, nm));
            l_run[qt] += p;
          }
          pf[qt][kt >> 1][(kt & 1) * 4 + r] = (half_t)p;
        }
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const half8_t vf = *reinterpret_cast<const half8_t*>(Vt + (dt * 16 + c16) * VT_LD + c * 32 + g * 8);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qt][c], oacc[qt][dt], 0, 0, 0);
      }
    // ---- next tile into the other stage (its last readers finished before the previous barrier)
    if (t + 1 < n_tiles) store_tile((t + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: lane owns query q0 + qt*16 + c16, dv = dt*16 + g*4 + r
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l;
    if (ONES) {
      const int dt_l = d >> 4, g_l = (d & 15) >> 2;
      float lv = 0.f;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
        if (dt == dt_l) lv = oacc[qt][dt][0];
      l = __shfl(lv, g_l * 16 + c16, 64);
    } else {
      l = l_run[qt];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
    const int qrow = q0 + qt * 16 + c16;
    if (qrow < a.Sq) {
      const float inv = 1.f / l;
      half_t* orow = a.o + (long)b * a.o_bs + (long)qrow * a.ldo + (long)h * d;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const int dv = dt * 16 + g * 4;
        if (dv < d) {
          half4_t o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (half_t)(oacc[qt][dt][r] * inv);
          *reinterpret_cast<half4_t*>(orow + dv) = o;
        }
      }
      if (a.lse && g == 0) a.lse[((long)b * a.H + h) * a.Sq + qrow] = m_ref[qt] + log2f(l);
    }
  }
}

// max over the two lanes (l, l + 32) that share a query in the 32x32 accumulator layout, delivered to both.
// Written as inline asm on purpose: with __builtin_amdgcn_permlane32_swap(x, x) hipcc (ROCm 7.2) drops the fmaxf of the
// two results — it folds the swap of two copies of one value into the identity, also behind an opaque register copy —
// and every lane then sees the lower half's value only (found with spiked keys in the upper half: fp16 inf in P).
// s_nop 1 = the two wait states between a VALU write of an operand and v_permlane32_swap reading it.
__device__ __forceinline__ float half_pair_max(float mx) {
  float a = mx, b = mx;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

// ---------------------------------------------------------------------------------------------
// Self-attention forward, 32x32x16 MFMA edition (round 3) for head dims with a spare slot (d + 2 <= DK).
//
// Why a second kernel.  Measured on MI355X (tools/ubench.hip, profiles/r03_ubench.txt): a SIMD does NOT run the VALU
// instructions of one wave beside the MFMAs of another — a wave doing 28 MFMA 16x16x32 next to a wave doing 32 v_exp +
// 96 v_fma takes the SUM of the two (511 ns vs 253 + 244), for 32x32x16 as well — and v_exp_f32 costs 3.6 plain VALU
// slots.  attn_self_kernel above is therefore the sum MFMA (241 ns per 32-query x 64-key wave tile) + softmax VALU
// (~225 ns) + stalls = 566 ns at d = 40, whatever the occupancy.  The only overlap the hardware offers is INSIDE one
// wave's instruction stream: independent VALU instructions issued between that wave's own MFMAs (the guide's "<= 5
// fillers per 32x32x16 gap"; 16x16x32 hides almost nothing: 7 % vs 17-36 % measured).  Hence:
//   * 32x32x16 MFMAs for both products.  S^T = K Q^T per 32 keys x 32 queries: the contraction is padded to DK = 48
//     for d = 40 (three k-steps of 16) instead of 64; O^T = V^T P^T in 32-row tiles of dv (two for d = 40 incl. the
//     row of ones that produces the row sums).  A lane owns ONE query (lane & 31) and 16 of the 32 keys of a block
//     (rows (r&3) + 8(r>>2) + 4(lane>>5)): the row max is 15 max3 + one permlane32 swap (no LDS bpermute), and the
//     S^T accumulators convert to the P^T operand in registers — V^T is staged with the key order of each 16-key group
//     permuted to match (position 8h + 4a + c holds key 8a + 4h + c), so that a V^T fragment is one ds_read_b128.
//   * software pipelining across key tiles inside the wave: S(t+1) = K(t+1) Q^T is issued in the same basic block as
//     P(t) = exp2(S(t)) and O += V^T(t) P(t); the two chains are independent, so the exponentials sit between the MFMAs.
//     The reference-raise decision for tile t+1 (wave-uniform vote, rare) is taken after PV(t) completed, so a rescale
//     covers everything accumulated at the old reference exactly once (S(t+1) is shifted explicitly).
//   * K(t+1) and V(t) are staged in the same iteration (V lags K by one tile): one barrier per key tile, two stages each.
//   * ragged key counts cost nothing in the loop: head-dim slot d+1 of Q holds -30000 and K holds 1 there for padded
//     keys (0 for real ones), next to slot d = (-m_ref, 1) of the running-reference trick.
template <int DK, int NDT, int NW, int PF = 2, bool PIN = true, int OCC = 1>
__global__ __launch_bounds__(64 * NW, OCC) void attn_self32_kernel(const AttnArgs a) {
  constexpr int NT = 64 * NW;
  constexpr int NKS = DK / 16;                  // k-steps of the QK^T contraction
  constexpr int K_LDB = DK * 2 + 16;            // bytes per K row: an odd multiple of 16 -> conflict-free ds_read_b128
  static_assert(((K_LDB / 16) & 1) == 1, "K row stride must be an odd multiple of 16 bytes");
  constexpr int VT_LDB = KV_T * 2 + 16;         // bytes per V^T row (64 keys): 144 = 9 x 16
  constexpr int VROWS = NDT * 32;
  constexpr int K_STAGE = KV_T * K_LDB;         // bytes
  constexpr int V_STAGE = VROWS * VT_LDB;
  constexpr int KSEG = DK / 8;                  // 16-byte segments per K row that may hold data
  constexpr int K_IT = (KV_T * KSEG + NT - 1) / NT;
  constexpr int V_ITEMS = (KV_T / 2) * KSEG;    // (key pair, 8-wide dv segment)
  constexpr int V_IT = (V_ITEMS + NT - 1) / NT;
  constexpr float MASKV = 30000.f;

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * K_STAGE + 2 * V_STAGE];
  unsigned char* const Kst = smem;
  unsigned char* const Vst = smem + 2 * K_STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int qi = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (32 * NW) + wid * 32;
  const int d = a.d;
  const half_t* Qb = a.q + (long)b * a.q_bs + (long)h * d;
  const half_t* Kb = a.k + (long)b * a.k_bs + (long)h * d;
  const half_t* Vb = a.v + (long)b * a.v_bs + (long)h * d;

  // ---- one-time LDS fill: zeros; ones at K column d (both stages) and at V^T row d
  for (int i = tid; i < (int)sizeof(smem) / 16; i += NT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  for (int i = tid; i < 2 * KV_T; i += NT) {
    const int st = i / KV_T, row = i - st * KV_T;
    *reinterpret_cast<half_t*>(Kst + st * K_STAGE + row * K_LDB + d * 2) = (half_t)1.f;
    *reinterpret_cast<half_t*>(Vst + st * V_STAGE + d * VT_LDB + row * 2) = (half_t)1.f;
  }

  // ---- Q^T fragments (B operand of S^T): lane -> query qi, head-dim elements 16 s + 8 hh .. + 8, pre-scaled to the
  // log2 domain; slot d carries -m_ref (K holds 1 there), slot d+1 carries -MASKV (K holds 1 there for padded keys)
  half8_t qf[NKS];
  {
    const int qrow = q0 + qi;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const int dd = s * 16 + hh * 8;
      if (qrow < a.Sq && dd < d) {
        qf[s] = *reinterpret_cast<const half8_t*>(Qb + (long)qrow * a.ldq + dd);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = (half_t)((float)qf[s][e] * a.scale_log2);
      } else {
        qf[s] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
  }
  const int s_m = d >> 4, h_m = (d >> 3) & 1;
#pragma unroll
  for (int s = 0; s < NKS; ++s)
    if (s == s_m && hh == h_m) qf[s][1] = (half_t)(-MASKV);
  auto set_ref = [&](float m) {
#pragma unroll
    for (int s = 0; s < NKS; ++s)
      if (s == s_m && hh == h_m) qf[s][0] = (half_t)(-m);
  };

  // ---- staging coordinates (hoisted)
  const half_t* kp[K_IT];
  int k_row[K_IT], k_dst[K_IT];
  bool k_use[K_IT];
#pragma unroll
  for (int i = 0; i < K_IT; ++i) {
    const int idx = tid + i * NT;
    const int row = idx / KSEG, seg = idx - row * KSEG;
    k_use[i] = (idx < KV_T * KSEG) && (seg * 8 < d);
    k_row[i] = row;
    k_dst[i] = row * K_LDB + seg * 16;
    kp[i] = Kb + (long)row * a.ldk + seg * 8;
  }
  const half_t* vp[V_IT];
  int v_row[V_IT], v_dst[V_IT];
  bool v_use[V_IT];
#pragma unroll
  for (int i = 0; i < V_IT; ++i) {
    const int idx = tid + i * NT;
    const int pair = idx & 31, seg = idx >> 5;
    const int key = pair * 2;
    // position of `key` inside its 16-key group: 8 h' + 4 a + c for key = 8 a + 4 h' + c
    const int pos = (key & ~15) | (((key >> 2) & 1) << 3) | (((key >> 3) & 1) << 2) | (key & 3);
    v_use[i] = (idx < V_ITEMS) && (seg * 8 < d);
    v_row[i] = key;
    v_dst[i] = (seg * 8) * VT_LDB + pos * 2;
    vp[i] = Vb + (long)key * a.ldv + seg * 8;
  }
  uint4 k_reg[K_IT], v_reg[V_IT][2];
  auto load_k = [&](int kv0) {
    const bool full = kv0 + KV_T <= a.Sk;
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
      const bool ok = k_use[i] && (full || kv0 + k_row[i] < a.Sk);
      k_reg[i] = ok ? *reinterpret_cast<const uint4*>(kp[i]) : make_uint4(0, 0, 0, 0);
      kp[i] += (long)KV_T * a.ldk;
    }
  };
  auto load_v = [&](int kv0) {
    const bool full = kv0 + KV_T <= a.Sk;
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const bool ok = v_use[i] && (full || kv0 + v_row[i] + r < a.Sk);
        v_reg[i][r] = ok ? *reinterpret_cast<const uint4*>(vp[i] + (long)r * a.ldv) : make_uint4(0, 0, 0, 0);
      }
      vp[i] += (long)KV_T * a.ldv;
    }
  };
  auto store_k = [&](int stage, int kv0) {
    unsigned char* base = Kst + stage * K_STAGE;
#pragma unroll
    for (int i = 0; i < K_IT; ++i)
      if (k_use[i]) *reinterpret_cast<uint4*>(base + k_dst[i]) = k_reg[i];
    if (kv0 + KV_T > a.Sk && tid < KV_T && kv0 + tid >= a.Sk)      // ragged last tile: flag the padded keys
      *reinterpret_cast<half_t*>(base + tid * K_LDB + (d + 1) * 2) = (half_t)1.f;
  };
  auto store_v = [&](int stage) {
    unsigned char* base = Vst + stage * V_STAGE;
#pragma unroll
    for (int i = 0; i < V_IT; ++i)
      if (v_use[i]) {
        const half_t* e0 = reinterpret_cast<const half_t*>(&v_reg[i][0]);
        const half_t* e1 = reinterpret_cast<const half_t*>(&v_reg[i][1]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          half2_t pr = {e0[e], e1[e]};
          *reinterpret_cast<half2_t*>(base + v_dst[i] + e * VT_LDB) = pr;
        }
      }
  };

  // ---- the two products
  const unsigned k_lane = qi * K_LDB + hh * 16;       // + stage + kb * 32 * K_LDB + s * 32
  const unsigned v_lane = qi * VT_LDB + hh * 16;      // + stage + dt * 32 * VT_LDB + kb * 64 + u * 32
  auto qk = [&](int stage, f32x16 (&s)[2]) {
    const unsigned char* base = Kst + stage * K_STAGE + k_lane;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const half8_t kf = *reinterpret_cast<const half8_t*>(base + kb * 32 * K_LDB + ks * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], acc, 0, 0, 0);
      }
      s[kb] = acc;
    }
  };
  f32x16 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  auto exp_pv = [&](int stage, const f32x16 (&s)[2]) {
    const unsigned char* base = Vst + stage * V_STAGE + v_lane;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        half8_t pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (half_t)__builtin_amdgcn_exp2f(s[kb][8 * u + j]);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const half8_t vf = *reinterpret_cast<const half8_t*>(base + dt * 32 * VT_LDB + kb * 64 + u * 32);
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[dt], 0, 0, 0);
        }
      }
  };
  // row max of a score tile (relative to the current reference) -> every lane of the query's pair
  auto tile_max = [&](const f32x16 (&s)[2]) {
    float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[0][r]), s[0][r + 1]);
    mx = fmaxf(fmaxf(mx, s[0][15]), s[1][0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[1][r]), s[1][r + 1]);
    mx = fmaxf(mx, s[1][15]);
    return half_pair_max(mx);
  };

  // MFMA slots of one key tile: NQ = 2 NKS slots of S(t+1) (k-step major, the two 32-key blocks alternating), then
  // NP = 4 NDT slots of O += V^T(t) P(t) (exp group g = 2 kb + u major, dv tile minor)
  constexpr int NQ = 2 * NKS, NP = 4 * NDT, NM = NQ + NP;
  constexpr int NE_SLOTS = NQ + 2 * NDT;        // the 32 exponentials are spread over the slots in front of PV(g = 2)
  auto pipelined_tile = [&](int kstage, int vstage, const f32x16 (&sc)[2], f32x16 (&sn)[2]) {
    const unsigned char* kbase = Kst + kstage * K_STAGE + k_lane;
    const unsigned char* vbase = Vst + vstage * V_STAGE + v_lane;
    auto frag = [&](int slot) -> half8_t {
      if (slot < NQ) {
        const int ks = slot >> 1, kb = slot & 1;
        return *reinterpret_cast<const half8_t*>(kbase + kb * 32 * K_LDB + ks * 32);
      }
      const int p = slot - NQ, g = p / NDT, dt = p - g * NDT;
      return *reinterpret_cast<const half8_t*>(vbase + dt * 32 * VT_LDB + g * 32);      // kb * 64 + u * 32 = 32 g
    };
    half8_t pf[4];
    half8_t fr[PF + 1];                 // fragment ring: slot s multiplies fr[0]; fr[PF] is requested PF slots ahead
#pragma unroll
    for (int i = 0; i < PF; ++i) fr[i] = frag(i);
    float mx = NEG_BIG;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sn[0][r] = 0.f; sn[1][r] = 0.f; }
#pragma unroll
    for (int slot = 0; slot < NM; ++slot) {
      if (slot + PF < NM) fr[PF] = frag(slot + PF);
      // exponentials due by the end of this slot (group g complete before PV slot NQ + g NDT)
      const int e0 = slot < NE_SLOTS ? (32 * slot) / NE_SLOTS : 32;
      const int e1 = slot + 1 < NE_SLOTS ? (32 * (slot + 1)) / NE_SLOTS : 32;
#pragma unroll
      for (int e = 0; e < 32; ++e)
        if (e >= e0 && e < e1) pf[e >> 3][e & 7] = (half_t)__builtin_amdgcn_exp2f(sc[e >> 4][e & 15]);
      if (slot < NQ) {
        const int ks = slot >> 1, kb = slot & 1;
        sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[0], qf[ks], sn[kb], 0, 0, 0);
      } else {
        const int p = slot - NQ, g = p / NDT, dt = p - g * NDT;
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[0], pf[g], oacc[dt], 0, 0, 0);
      }
      // row max of S(t+1) (complete since slot NQ - 1): 16 max3 spread over the slots behind the exponentials
      if (slot >= NE_SLOTS) {
        const int p0 = (16 * (slot - NE_SLOTS)) / (NM - NE_SLOTS), p1 = (16 * (slot + 1 - NE_SLOTS)) / (NM - NE_SLOTS);
#pragma unroll
        for (int pr = 0; pr < 16; ++pr)
          if (pr >= p0 && pr < p1) mx = fmaxf(fmaxf(mx, sn[pr >> 3][2 * (pr & 7)]), sn[pr >> 3][2 * (pr & 7) + 1]);
      }
#pragma unroll
      for (int i = 0; i < PF; ++i) fr[i] = fr[i + 1];
      if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
    return half_pair_max(mx);
  };

  float m_ref = 0.f;
  const int n_tiles = (a.Sk + KV_T - 1) / KV_T;

  // ---- prologue: K(0) staged, S(0) computed and referenced to its own row max; K(1), V(0) on their way
  load_k(0);
  store_k(0, 0);
  load_v(0);
  if (n_tiles > 1) load_k(KV_T);
  __syncthreads();
  f32x16 s_cur[2], s_nxt[2];
  qk(0, s_cur);
  {
    const float mx = tile_max(s_cur);
    m_ref = (float)(half_t)mx;             // rounded to fp16 so that the Q slot holds it exactly
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_cur[kb][r] -= m_ref;
    set_ref(m_ref);
  }

  for (int t = 0; t < n_tiles; ++t) {
    const bool more = t + 1 < n_tiles;
    // [A] stage K(t+1) and V(t): K stage (t+1)&1 last held K(t-1) (read in iteration t-2), V stage t&1 last held
    // V(t-2) (read in iteration t-2); the barrier of iteration t-1 lies in between
    if (more) store_k((t + 1) & 1, (t + 1) * KV_T);
    store_v(t & 1);
    __syncthreads();
    if (t + 2 < n_tiles) load_k((t + 2) * KV_T);
    if (more) load_v((t + 1) * KV_T);
    // [B] S(t+1) = K(t+1) Q^T   and   [C] P(t) = exp2(S(t)), O += V^T(t) P(t): independent chains, issued as ONE pinned
    // sequence of MFMA slots — every slot = {fragment read two slots ahead, a few exponentials of P(t), one MFMA} — so
    // that the softmax VALU work sits in the issue gaps of this wave's own MFMAs (the only MFMA/VALU overlap this
    // chip has).  (The last iteration multiplies a stale K stage into s_nxt, which nobody reads.)
    const float mx_nxt = pipelined_tile((t + 1) & 1, t & 1, s_cur, s_nxt);
    // [D] reference of tile t+1 (PV(t) is complete: a rescale covers everything accumulated so far exactly once)
    if (more) {
      const float mx = mx_nxt;
      if (__builtin_amdgcn_ballot_w64(mx > 8.f) != 0) {
        const float m_new = (float)(half_t)fmaxf(m_ref, mx + m_ref);
        const float shift = m_new - m_ref;
        const float alpha = __builtin_amdgcn_exp2f(-shift);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s_nxt[kb][r] -= shift;
        m_ref = m_new;
        set_ref(m_new);
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) s_cur[kb] = s_nxt[kb];
    }
  }

  // ---- epilogue.  Row sum = O^T row d (the row of ones): tile d>>5, lane half ((d&31)>>2)&1, register (d&3) + 4((d&31)>>3)
  float l;
  {
    const int dl = d & 31, dt_l = d >> 5, h_l = (dl >> 2) & 1, r_l = (dl & 3) + 4 * (dl >> 3);
    float lv = 0.f;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (dt == dt_l && r == r_l) lv = oacc[dt][r];
    l = __shfl(lv, h_l * 32 + qi, 64);
  }
  const int qrow = q0 + qi;
  if (qrow < a.Sq) {
    const float inv = 1.f / l;
    half_t* orow = a.o + (long)b * a.o_bs + (long)qrow * a.ldo + (long)h * d;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int dv = dt * 32 + g4 * 8 + hh * 4;
        if (dv < d) {
          half4_t o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (half_t)(oacc[dt][g4 * 4 + r] * inv);
          *reinterpret_cast<half4_t*>(orow + dv) = o;
        }
      }
    if (a.lse && hh == 0) a.lse[((long)b * a.H + h) * a.Sq + qrow] = m_ref + log2f(l);
  }
}

template <int DP, bool SAVE_P>
void launch_attn_dp(const AttnArgs& a, hipStream_t st) {
  if constexpr (SAVE_P) {
    dim3 grid((a.Sq + 63) / 64, a.H, a.B);
    hipLaunchKernelGGL((attn_fwd_kernel<DP>), grid, dim3(256), 0, st, a);
  } else {
    // round-3 kernel (32x32x16 MFMA, in-wave software pipelining) for the narrow heads with a spare slot, once there
    // are enough 256-query blocks to fill the chip; LGD_ATTN32=0 keeps the 16x16x32 kernel (A/B timing, tools)
    if constexpr (DP == 64 || DP == 96) {
      // g_attn32 (lgd_set_option("attn32", v); initial value from LGD_ATTN32 in the environment): 0 = never (the
      // 16x16x32 kernel: A/B timing), 1 = default, 2 = for every problem size (tests)
      const int a32 = attn32_mode();
      const int dk = ((a.d + 2 + 15) / 16) * 16;
      // measured (tools/attn_quick.py, B = 16): d = 80 (DK = 96) 97 -> 80 us; d = 40 (DK = 48) 570 -> 630 us — at two
      // waves per SIMD (170 VGPRs) its stalls are not covered the way the 16x16x32 kernel's four waves cover theirs —
      // so by default only the DK = 96 heads take this kernel
      // round 6: the threshold was 512 blocks of 256 queries ("fill the chip twice"); measured (tools/attn80_ab.py) the
      // 8-image calls (256 blocks) run 70.3 -> 42.0 us on this kernel (305 -> 512 TF/s; with the fuser's 1054 keys 73.7 ->
      // 45.9) and the 4-image calls (128 blocks) 34.7 -> 33.2 us with 128-query workgroups
      const long blocks256 = (long)((a.Sq + 255) / 256) * a.H * a.B;
      if (a32 && a.d % 8 == 0 && (a32 == 2 || (dk == 96 && blocks256 >= 128))) {
        const int var = attn32_var();
        auto go = [&](auto kern, int nw) {
          dim3 g32((a.Sq + 32 * nw - 1) / (32 * nw), a.H, a.B);
          hipLaunchKernelGGL(kern, g32, dim3(64 * nw), 0, st, a);
        };
        if (attn32_nw() == 4 || (a32 != 2 && blocks256 < 256)) {
          if (dk == 48) { go(&attn_self32_kernel<48, 2, 4>, 4); return; }
          if (dk == 96) { go(&attn_self32_kernel<96, 3, 4>, 4); return; }
        } else if (var == 1) {
          if (dk == 48) { go(&attn_self32_kernel<48, 2, 8, 4, true>, 8); return; }
          if (dk == 96) { go(&attn_self32_kernel<96, 3, 8, 4, true>, 8); return; }
        } else if (var == 2) {
          // (the <= 128-VGPR, two-workgroups-per-CU build of the dk = 48 kernel spilled 76 registers and is gone)
          if (dk == 48) { go(&attn_self32_kernel<48, 2, 8, 2, false>, 8); return; }
          if (dk == 96) { go(&attn_self32_kernel<96, 3, 8, 2, false>, 8); return; }
        } else {
          if (dk == 48) { go(&attn_self32_kernel<48, 2, 8>, 8); return; }
          if (dk == 96) { go(&attn_self32_kernel<96, 3, 8>, 8); return; }
        }
      }
    }
    // two query tiles per wave once there are enough 128-query blocks to fill the chip
    const bool qt2 = (long)((a.Sq + 127) / 128) * a.H * a.B >= 1024 && DP <= 96;
    dim3 grid(qt2 ? (a.Sq + 127) / 128 : (a.Sq + 63) / 64, a.H, a.B);
    if constexpr (DP <= 96) {
      // 8-wave workgroups (256 queries share each staged K / V^T tile) pay where the head is narrow: measured
      // +15 % at d = 40 (S = 4096), +4 % at d = 80, -15 % at d = 64 (S = 9216).  LGD_ATTN_NW=4 / 8 overrides (tools).
      static const int nw_env = [] { const char* e = getenv("LGD_ATTN_NW"); return e ? atoi(e) : 0; }();
      const bool nw8 = nw_env ? nw_env == 8 : ((DP == 64 && a.d < 48) || DP == 96);
      if (qt2 && nw8 && (long)((a.Sq + 255) / 256) * a.H * a.B >= 512) {
        dim3 g8((a.Sq + 255) / 256, a.H, a.B);
        if (DP == 64 && a.d < 48) hipLaunchKernelGGL((attn_self_kernel<DP, true, 2, DP == 64 ? 3 : DP / 16, 8>), g8, dim3(512), 0, st, a);
        else if (a.d < DP) hipLaunchKernelGGL((attn_self_kernel<DP, true, 2, DP / 16, 8>), g8, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((attn_self_kernel<DP, false, 2, DP / 16, 8>), g8, dim3(512), 0, st, a);
        return;
      }
      if (qt2) {
        if (DP == 64 && a.d < 48) hipLaunchKernelGGL((attn_self_kernel<DP, true, 2, DP == 64 ? 3 : DP / 16>), grid, dim3(256), 0, st, a);
        else if (a.d < DP) hipLaunchKernelGGL((attn_self_kernel<DP, true, 2>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_self_kernel<DP, false, 2>), grid, dim3(256), 0, st, a);
        return;
      }
    }
    if constexpr (DP == 160) {
      // round 6: the 16x16 level (S = 256, d = 160): ONE workgroup per (image, head) — eight waves x two query tiles = all
      // 256 queries share each staged K / V^T tile, which the 64-query workgroups stage four times over.  A workgroup of
      // either form is a chain of four or five load -> stage -> barrier round trips of ~4.5 us (not MFMA work), so the wide
      // form wins only where the narrow one needs two rounds of workgroups: measured (tools/attn160_ab.py, 40 launches in
      // one graph) B = 16: 33.6 -> 23.8 us (160 -> 226 TF/s), with the fuser's 286 keys 38.0 -> 26.8; B = 8: 18.4 -> 21.8,
      // B = 4: 14.5 -> 20.7 — hence the (image, head) count in the condition.  LGD_ATTN160=0: old form.  Same per-row
      // arithmetic and key order: bit-identical outputs.
      static const int wide = [] { const char* e = getenv("LGD_ATTN160"); return e ? atoi(e) : 1; }();
      if (wide && a.d == DP && a.Sq >= 256 && (long)a.H * a.B * ((a.Sq + 63) / 64) > 256) {
        dim3 g8((a.Sq + 255) / 256, a.H, a.B);
        hipLaunchKernelGGL((attn_self_kernel<DP, false, 2, DP / 16, 8>), g8, dim3(512), 0, st, a);
        return;
      }
    }
    if (DP == 64 && a.d < 48) hipLaunchKernelGGL((attn_self_kernel<DP, true, 1, DP == 64 ? 3 : DP / 16>), grid, dim3(256), 0, st, a);
    else if (a.d < DP) hipLaunchKernelGGL((attn_self_kernel<DP, true, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn_self_kernel<DP, false, 1>), grid, dim3(256), 0, st, a);
  }
}

template <bool SAVE_P>
int launch_attn(const AttnArgs& a, hipStream_t st) {
  const int d = a.d;
  if constexpr (!SAVE_P) {
    // d = 40 (SD1.x 64x64 level): the one-wave-per-SIMD kernel of attn_w4.hip once a launch has enough 256-query
    // workgroups to occupy the chip (it holds ONE workgroup per CU); smaller problems keep the 4-waves-per-SIMD kernel
    const int w4 = attn_w4_mode();
    if (w4 && d == 40) {
      AttnW4Args w;
      w.q = a.q; w.ldq = a.ldq; w.q_bs = a.q_bs; w.k = a.k; w.ldk = a.ldk; w.k_bs = a.k_bs;
      w.v = a.v; w.ldv = a.ldv; w.v_bs = a.v_bs; w.o = a.o; w.ldo = a.ldo; w.o_bs = a.o_bs; w.lse = a.lse;
      w.B = a.B; w.H = a.H; w.Sq = a.Sq; w.Sk = a.Sk; w.d = a.d; w.scale_log2 = a.scale_log2;
      const long wgs = (long)((a.Sq + 255) / 256) * a.H * a.B;
      if (lgd_attn_w4_supported(w) && (w4 == 2 || (wgs >= 256 && a.Sk >= 256))) return lgd_attn_w4_launch(w, st);
    }
  }
  if (d <= 32) launch_attn_dp<32, SAVE_P>(a, st);
  else if (d <= 64) launch_attn_dp<64, SAVE_P>(a, st);
  else if (d <= 96) launch_attn_dp<96, SAVE_P>(a, st);
  else if (d <= 128) launch_attn_dp<128, SAVE_P>(a, st);
  else if (d <= 160) launch_attn_dp<160, SAVE_P>(a, st);
  else if (d <= 192) launch_attn_dp<192, SAVE_P>(a, st);  // SAM global attention: 64 + 2 x 64 bias columns
  else return LGD_ERR_UNSUPPORTED;
  return lgd_check_launch();
}

bool bad_view(int64_t ld, int d) { return (ld % 8) != 0 || (d % 8) != 0; }

}  // namespace

void lgd_gn_set_fused_hw(int hw);                          // norm.hip
void lgd_gn_set_slab(int on);                              // norm.hip
void lgd_ln_set_stream(int on);                            // norm.hip
void lgd_gn_set_apply_wgs(int n);                          // norm.hip

extern "C" int lgd_set_option(const char* name, int value) {
  if (!name) return LGD_ERR_ARG;
  if (!strcmp(name, "gn_fused") && value >= 0 && value <= 4096) { lgd_gn_set_fused_hw(value); return LGD_OK; }
  if (!strcmp(name, "gn_slab") && (value == 0 || value == 1)) { lgd_gn_set_slab(value); return LGD_OK; }
  if (!strcmp(name, "ln_stream") && (value == 0 || value == 1)) { lgd_ln_set_stream(value); return LGD_OK; }
  if (!strcmp(name, "gn_apply_wgs") && value >= 64 && value <= 8192) { lgd_gn_set_apply_wgs(value); return LGD_OK; }
  if (!strcmp(name, "attn32")) { g_attn32 = value; return LGD_OK; }
  if (!strcmp(name, "attn_w4") && value >= 0 && value <= 2) { g_attn_w4.store(value, std::memory_order_relaxed); return LGD_OK; }
  if (!strcmp(name, "attn_w4_pipe") && (value == 0 || value == 1)) { lgd_attn_w4_set_pipe(value); return LGD_OK; }
  if (!strcmp(name, "attn32_nw") && (value == 4 || value == 8)) { g_attn32_nw = value; return LGD_OK; }
  if (!strcmp(name, "attn32_var") && value >= 0 && value <= 2) { g_attn32_var = value; return LGD_OK; }
  return LGD_ERR_ARG;
}

extern "C" int lgd_attn_fwd_f16(const void* q, int64_t ldq, int64_t q_bs, const void* k,
                                int64_t ldk, int64_t k_bs, const void* v, int64_t ldv, int64_t v_bs,
                                void* o, int64_t ldo, int64_t o_bs, float* lse, int B, int H, int Sq,
                                int Sk, int d, float scale, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (B < 1 || H < 1 || Sq < 1 || Sk < 1 || d < 8) return LGD_ERR_ARG;
  if (bad_view(ldq, d) || bad_view(ldk, d) || bad_view(ldv, d) || (ldo % 4)) return LGD_ERR_ARG;
  AttnArgs a;
  a.q = (const half_t*)q; a.ldq = ldq; a.q_bs = q_bs;
  a.k = (const half_t*)k; a.ldk = ldk; a.k_bs = k_bs;
  a.v = (const half_t*)v; a.ldv = ldv; a.v_bs = v_bs;
  a.o = (half_t*)o; a.ldo = ldo; a.o_bs = o_bs;
  a.lse = lse; a.probs = nullptr; a.tok = -1; a.cond_only = 0; a.causal = 0;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.d = d;
  a.scale_log2 = scale * 1.4426950408889634f;
  return launch_attn<false>(a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int lgd_cross_attn_fwd_f16(const void* q, int64_t ldq, int64_t q_bs, const void* k,
                                      int64_t ldk, int64_t k_bs, const void* v, int64_t ldv,
                                      int64_t v_bs, void* o, int64_t ldo, int64_t o_bs, float* probs,
                                      int tok, int cond_only, int B, int H, int Sq, int Sk, int d,
                                      float scale, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (B < 1 || H < 1 || Sq < 1 || Sk < 1 || d < 8) return LGD_ERR_ARG;
  if (bad_view(ldq, d) || bad_view(ldk, d) || bad_view(ldv, d) || (ldo % 4)) return LGD_ERR_ARG;
  if (cond_only && (B % 2)) return LGD_ERR_ARG;  // attention_processor.py:475
  if (tok >= Sk) return LGD_ERR_ARG;
  AttnArgs a;
  a.q = (const half_t*)q; a.ldq = ldq; a.q_bs = q_bs;
  a.k = (const half_t*)k; a.ldk = ldk; a.k_bs = k_bs;
  a.v = (const half_t*)v; a.ldv = ldv; a.v_bs = v_bs;
  a.o = (half_t*)o; a.ldo = ldo; a.o_bs = o_bs;
  a.lse = nullptr; a.probs = probs; a.tok = tok; a.cond_only = cond_only; a.causal = 0;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.d = d;
  a.scale_log2 = scale * 1.4426950408889634f;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (probs) return launch_attn<true>(a, st);
  return launch_attn<false>(a, st);
}

extern "C" int lgd_attn_causal_fwd_f16(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk,
                                       int64_t k_bs, const void* v, int64_t ldv, int64_t v_bs, void* o,
                                       int64_t ldo, int64_t o_bs, int B, int H, int S, int d, float scale,
                                       void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (B < 1 || H < 1 || S < 1 || d < 8) return LGD_ERR_ARG;
  if (bad_view(ldq, d) || bad_view(ldk, d) || bad_view(ldv, d) || (ldo % 4)) return LGD_ERR_ARG;
  AttnArgs a;
  a.q = (const half_t*)q; a.ldq = ldq; a.q_bs = q_bs;
  a.k = (const half_t*)k; a.ldk = ldk; a.k_bs = k_bs;
  a.v = (const half_t*)v; a.ldv = ldv; a.v_bs = v_bs;
  a.o = (half_t*)o; a.ldo = ldo; a.o_bs = o_bs;
  a.lse = nullptr; a.probs = nullptr; a.tok = -1; a.cond_only = 0; a.causal = 1;
  a.B = B; a.H = H; a.Sq = S; a.Sk = S; a.d = d;
  a.scale_log2 = scale * 1.4426950408889634f;
  return launch_attn<true>(a, reinterpret_cast<hipStream_t>(stream));   // exact two-pass softmax kernel
}
