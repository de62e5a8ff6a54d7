// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950.
//
//   C[m][n] = epilogue( sum_k A(m,k) * W[n][k] ),  fp16 operands, fp32 accumulate.
//
// One kernel covers every dense contraction of the SD UNet (see include/lgd_hip.h for the
// reference call sites): nn.Linear, 1x1 conv, 3x3 conv (stride 1/2, nearest-2x upsample folded into
// the gather, channel-concat of two sources folded into the K loop) and their dgrad forms.
//
// Two main loops share one epilogue:
//   gemm_dma_kernel  (the one every UNet contraction runs on, K % 64 == 0): operands go global -> LDS by
//                    LDS-DMA, see its header further down;
//   gemm_kernel      (fallback for any K): register-staged, described here.
// Structure (per workgroup, 256 threads = 4 waves in a 2x2 grid):
//   * tile BM x BN x 64; each wave owns (BM/2)x(BN/2) as MI x NI MFMA 16x16x32 tiles;
//   * gemm_kernel: operands are register-staged: global -> VGPR (issued before the MFMAs of the current
//     K-tile) -> LDS after the barrier, so the HBM/L2 latency hides under the MFMA phase; LDS rows are
//     padded to 80 halfs (160 B): with that stride each of the 4 hardware lane groups of a ds_read_b128
//     fragment read touches 16 distinct 16-B slots (72 halfs is 2-way conflicted);
//   * the MFMA is issued "swapped" (A := weight rows, B := activation rows), which makes each lane
//     own 4 consecutive output channels of one pixel -> one 8-byte store, and 4-wide bias /
//     residual loads in the epilogue;
//   * split-K (grid.z) writes fp32 partials to a workspace, a second launch reduces and applies
//     the same epilogue (deterministic, no atomics) — needed for the 8x8 / 16x16 levels where
//     M = B*HW is 128..512 and the weight stream must be spread over all 256 CUs.
#include "common.h"
#include "../../include/lgd_hip.h"
#include <utility>
#include <stdlib.h>

namespace {

constexpr int BK = 64;
constexpr int LDS_LD = BK + 16;  // halfs per LDS row (160 B: the 4 ds_read_b128 lane groups hit 16 distinct 16-B slots)

struct GemmArgs {
  LgdGemmDesc d;
  int cin;       // c0 + c1
  int k_per_split;  // multiple of BK
  int group_m;      // gemm_pipe_kernel: output tiles are walked in groups of group_m row panels (1 = rows of tiles)
#ifdef LGD_GEMM_ABLATION
  int stagger;      // tools: workgroups of the second residency slot start late by stagger x 64 x 127 cycles
#endif
};

__device__ __forceinline__ float4 ld_bias4(const float* p, int n) {
  return *reinterpret_cast<const float4*>(p + n);
}

// LGD_EPI_ROWNORM: the LayerNorm of the A rows, folded behind the contraction (see include/lgd_hip.h): accumulator
// columns n_in..n_in+3 of row m become rstd (acc - mean colsum).
__device__ __forceinline__ f32x4 rownorm4(const LgdGemmDesc& d, int m, int n_in, f32x4 v) {
  if (d.epi & LGD_EPI_ROWNORM) {
    const float2 st = *reinterpret_cast<const float2*>(d.rowstat + (long)m * 2);
    const float4 cs = *reinterpret_cast<const float4*>(d.colsum + n_in);
    v[0] = st.y * (v[0] - st.x * cs.x); v[1] = st.y * (v[1] - st.x * cs.y);
    v[2] = st.y * (v[2] - st.x * cs.z); v[3] = st.y * (v[3] - st.x * cs.w);
  }
  return v;
}

// The epilogue arithmetic on 4 consecutive output channels (n..n+3) of row m: bias, GEGLU, alpha, residual.
// For GEGLU `v` holds the value accumulators and `g` the gate accumulators of the same columns.
template <bool GEGLU>
__device__ __forceinline__ f32x4 epilogue_value4(const LgdGemmDesc& d, long r_off, int m, int n_out, f32x4 v, f32x4 g,
                                                 float4 bv, float4 bg, bool res_ready, half4_t res_pre) {
  // bv / bg: summed biases of the value / gate columns (zeros when absent); n_out: output column;
  // res_pre: the fp16 residual of these 4 outputs when the caller already fetched it (res_ready).
  v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
  if (GEGLU) {
    g[0] += bg.x; g[1] += bg.y; g[2] += bg.z; g[3] += bg.w;
#ifdef LGD_GEMM_ABLATION                 // tools: epi bit 16 = GEGLU without the GELU arithmetic
    if (d.epi & (1 << 16)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] * g[r];
    } else
#endif
    {
      const f32x2_t g01 = gelu2_f((f32x2_t){g[0], g[1]}), g23 = gelu2_f((f32x2_t){g[2], g[3]});
      v[0] *= g01[0]; v[1] *= g01[1]; v[2] *= g23[0]; v[3] *= g23[1];
    }
  }
  const float alpha = d.alpha;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] *= alpha;
  if (res_ready) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += (float)res_pre[r];
  } else if (d.res) {
    if (d.epi & LGD_EPI_RES_F32) {
      const float* rp = reinterpret_cast<const float*>(d.res) + r_off + (long)m * d.ldr + n_out;
      float4 rv = *reinterpret_cast<const float4*>(rp);
      v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
    } else {
      const half_t* rp = reinterpret_cast<const half_t*>(d.res) + r_off + (long)m * d.ldr + n_out;
      half4_t rv = *reinterpret_cast<const half4_t*>(rp);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
    }
  }
  return v;
}

// Applies the epilogue to 4 consecutive output channels (n..n+3) of row m and stores them.
template <bool GEGLU>
__device__ __forceinline__ void epilogue_store4(const LgdGemmDesc& d, long c_off, long r_off, int m,
                                                int n_out, f32x4 v, f32x4 g, float4 bv, float4 bg,
                                                bool res_ready = false,
                                                half4_t res_pre = (half4_t){0, 0, 0, 0}) {
  v = epilogue_value4<GEGLU>(d, r_off, m, n_out, v, g, bv, bg, res_ready, res_pre);
  if (d.epi & LGD_EPI_OUT_F32) {
    float* cp = reinterpret_cast<float*>(d.c) + c_off + (long)m * d.ldc + n_out;
    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    half_t* cp = reinterpret_cast<half_t*>(d.c) + c_off + (long)m * d.ldc + n_out;
    half4_t o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
    *reinterpret_cast<half4_t*>(cp) = o;
  }
}

// Sum of the (optional) two bias vectors at columns n..n+3.
__device__ __forceinline__ float4 ld_bias_sum4(const LgdGemmDesc& d, int n) {
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (d.bias) b = ld_bias4(d.bias, n);
  if (d.bias2) {
    float4 b2 = ld_bias4(d.bias2, n);
    b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
  }
  return b;
}

// Epilogue shared by both main-loop variants: lane owns pixel m = .. + (lane&15) and the 4
// consecutive channels n = .. + (lane>>4)*4 + r of every 16x16 accumulator tile.
template <int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& ga, f32x4 (&acc)[NI][MI], int m0, int n0,
                                              int wm, int wn, int lane, int batch, int split,
                                              long c_off, long r_off, int* lds_flag = nullptr) {
  const LgdGemmDesc& d = ga.d;
  const int m_l = lane & 15;
  const int n_l = (lane >> 4) * 4;
  if (d.splits > 1) {
    float* ws = ga.d.ws + ((long)batch * d.splits + split) * (long)d.M * d.N;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      int m = m0 + wm * 16 * MI + mi * 16 + m_l;
      if (m >= d.M) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        int n = n0 + wn * 16 * NI + ni * 16 + n_l;
        if (n >= d.N) continue;
        f32x4 v = acc[ni][mi];
        *reinterpret_cast<float4*>(ws + (long)m * d.N + n) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    if (!d.cnt || !lds_flag) return;      // combined by splitk_reduce_kernel (second launch)
    // ---- in-launch combine ("last arriver reduces"; the hand-off recipe of the CDNA guide, counter form): every
    // wave drains its partial stores, the workgroup meets, ONE lane publishes with an agent-scope release (the asm
    // wait restates the wait behind buffer_wbl2 where the compiler cannot drop it) and takes a ticket; the workgroup
    // that draws splits - 1 acquires (one lane, agent scope: drops this CU's stale L1 lines), re-zeroes the counter and
    // sums ALL partials — its own included — from memory in split order, so the result is bit-identical to the reduce
    // kernel and independent of arrival order.  Correct for any placement of a tile's splits over CUs / XCDs.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int32_t* cnt = d.cnt + (long)batch * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      *lds_flag = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (*lds_flag != d.splits - 1) return;
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const float* w0 = ga.d.ws + (long)batch * d.splits * (long)d.M * d.N;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + wm * 16 * MI + mi * 16 + m_l;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * 16 * NI + ni * 16 + n_l;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < d.M && n < d.N) {
          for (int s2 = 0; s2 < d.splits; ++s2) {
            const float4 t = *reinterpret_cast<const float4*>(w0 + (long)s2 * d.M * d.N + (long)m * d.N + n);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
          }
        }
        acc[ni][mi] = v;
      }
    }
  }
  const bool geglu = d.epi & LGD_EPI_GEGLU;
  if (geglu) {
    if constexpr (NI % 2 == 0) {
#pragma unroll
      for (int ni = 0; ni < NI; ni += 2) {
        const int n_in = n0 + wn * 16 * NI + ni * 16 + n_l;
        if (n_in >= d.N) continue;
        const int n_out = (n0 + wn * 16 * NI + ni * 16) / 2 + n_l;
        const float4 bv = ld_bias_sum4(d, n_in);
        float4 bg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias) bg = ld_bias4(d.bias, n_in + 16);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int m = m0 + wm * 16 * MI + mi * 16 + m_l;
          if (m < d.M)
            epilogue_store4<true>(d, c_off, r_off, m, n_out, rownorm4(d, m, n_in, acc[ni][mi]),
                                  rownorm4(d, m, n_in + 16, acc[ni + 1][mi]), bv, bg);
        }
      }
    }
    return;
  }
  // fp16 residual (every resnet conv2 / attention out-projection / feed-forward down-projection):
  // all MI x NI fetches are issued before the first one is consumed, so the epilogue pays one
  // memory round trip, not MI x NI of them.  Out-of-range lanes fetch a clamped, valid address.
  const bool res16 = d.res && !(d.epi & LGD_EPI_RES_F32);
  constexpr int NCH = NI <= 5 ? NI : (NI % 5 == 0 ? 5 : 4);  // ni columns per prefetch round (bounds the live registers)
  static_assert(NI % NCH == 0, "NI must split into equal prefetch rounds");
#pragma unroll
  for (int nc = 0; nc < NI; nc += NCH) {
    half4_t rpre[NCH][MI];
    if (res16) {
      const half_t* rb = reinterpret_cast<const half_t*>(d.res) + r_off;
#pragma unroll
      for (int nj = 0; nj < NCH; ++nj) {
        int n = n0 + wn * 16 * NI + (nc + nj) * 16 + n_l;
        if (n >= d.N) n = 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          int m = m0 + wm * 16 * MI + mi * 16 + m_l;
          if (m >= d.M) m = d.M - 1;
          rpre[nj][mi] = *reinterpret_cast<const half4_t*>(rb + (long)m * d.ldr + n);
        }
      }
    }
#pragma unroll
    for (int nj = 0; nj < NCH; ++nj) {
      const int ni = nc + nj;
      const int n = n0 + wn * 16 * NI + ni * 16 + n_l;
      if (n >= d.N) continue;
      const float4 bv = ld_bias_sum4(d, n);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * 16 * MI + mi * 16 + m_l;
        if (m < d.M)
          epilogue_store4<false>(d, c_off, r_off, m, n, rownorm4(d, m, n, acc[ni][mi]), acc[ni][mi], bv, bv, res16,
                                 res16 ? rpre[nj][mi] : (half4_t){0, 0, 0, 0});
      }
    }
  }
}

// Epilogue of the 8-wave kernels through LDS (round 3): the MFMA fragment layout gives a lane 4 consecutive channels
// of one pixel, so a direct store instruction writes 16 rows x 32 bytes — measured on MI355X, a 65536 x 960 fp16
// output leaves the chip at 2.1 TB/s that way, and every short-K GEMM (q/k/v/out projections, feed-forward) sat on
// that floor whatever its tile (tools/gemm_epi_probe.py).  Here the finished fp16 values of a group of wave rows are
// parked in ONE free stage of the operand ring (row stride + 16 B), then written out as whole rows, 16 bytes per lane:
// full 128-byte lines per request.  The arithmetic (bias, GEGLU, alpha, residual, single rounding to fp16) is
// epilogue_value4, unchanged; the residual is still fetched in the fragment layout (reads of 32-byte pieces are served
// by L1 / L2 at 4 TB/s; it is the WRITES of partial lines that were slow).
//   `lds`: a STAGE_B-byte region nobody else touches (the stage consumed last: no DMA in flight into it, all fragment
//   reads of it retired before the K loop's last barrier) on ENTRY, for one-tile and persistent workgroups alike; on
//   EXIT a persistent workgroup closes the last round with a barrier (see the end of the round loop).
//   Bare s_barrier + lgkmcnt(0): a __syncthreads() would drain the DMA queue of a persistent workgroup.
// The round-3 form with runtime flags: kept for the 256 x 256 tile, whose 128 accumulators leave no room for six
// specialisations' worth of live parameters (the specialised form spilled 6-10 registers there).
template <int MI, int NI, int WM, int WN, int STAGE_B, bool GEGLU, bool RES_PREFETCH>
__device__ __forceinline__ void gemm_epilogue_lds_generic(const GemmArgs& ga, f32x4 (&acc)[NI][MI], int m0, int n0, int wm,
                                                  int wn, int lane, int tid, long c_off, long r_off,
                                                  unsigned char* lds) {
  const LgdGemmDesc& d = ga.d;
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = WM * 16 * MI, BN = WN * 16 * NI;
  constexpr int BN_OUT = GEGLU ? BN / 2 : BN;
  constexpr int NI_OUT = GEGLU ? NI / 2 : NI;           // output column tiles of a wave
  constexpr int CROW = BN_OUT * 2 + 16;                 // bytes per staged row
  // rounds over the 16-row tiles (mi) of EVERY wave: all waves work in both phases of a round, and a round still
  // covers whole output rows (16 MI_R rows of each of the WM wave rows)
  constexpr int R = (BM * CROW <= STAGE_B) ? 1 : ((BM / 2) * CROW <= STAGE_B) ? 2 : 4;
  static_assert(MI % R == 0 && (BM / R) * CROW <= STAGE_B, "staged rows must fit one ring stage");
  constexpr int MI_R = MI / R;
  constexpr int WROWS = 16 * MI_R;                      // rows of one wave row per round
  constexpr int RROWS = WM * WROWS;
  constexpr int CPR = BN_OUT / 8;                       // 16-byte chunks per row
  const int m_l = lane & 15, n_l = (lane >> 4) * 4;
  const int n0_out = GEGLU ? n0 / 2 : n0;
  const int n_total_out = GEGLU ? d.N / 2 : d.N;
  const bool res16 = d.res && !(d.epi & LGD_EPI_RES_F32);
  half_t* cbase = reinterpret_cast<half_t*>(d.c) + c_off;
  // biases of this wave's columns: all loads in flight at once
  float4 bv[NI_OUT], bg[NI_OUT];
  // folded LayerNorm (LGD_EPI_ROWNORM): column sums with the biases, row statistics of all MI row tiles — one batch
  // of loads, one memory round trip, before the first accumulator is touched
  const bool rn = d.epi & LGD_EPI_ROWNORM;
  float4 cv[NI_OUT], cg[NI_OUT];
  float2 st[MI];
#pragma unroll
  for (int no = 0; no < NI_OUT; ++no) {
    const int n_out = n0_out + wn * 16 * NI_OUT + no * 16 + n_l;
    const int n_in = GEGLU ? n0 + wn * 16 * NI + no * 32 + n_l : n_out;
    bv[no] = make_float4(0.f, 0.f, 0.f, 0.f);
    bg[no] = bv[no];
    cv[no] = bv[no];
    cg[no] = bv[no];
#ifdef LGD_GEMM_ABLATION                 // tools: epi bit 18 = no parameter loads in the epilogue
    if (d.epi & (1 << 18)) continue;
#endif
    if (n_out < n_total_out) {
      bv[no] = ld_bias_sum4(d, n_in);
      if (GEGLU && d.bias) bg[no] = ld_bias4(d.bias, n_in + 16);
      if (rn) {
        cv[no] = *reinterpret_cast<const float4*>(d.colsum + n_in);
        if (GEGLU) cg[no] = *reinterpret_cast<const float4*>(d.colsum + n_in + 16);
      }
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int m = m0 + wm * 16 * MI + mi * 16 + m_l;
    if (m >= d.M) m = d.M - 1;
    st[mi] = rn ? *reinterpret_cast<const float2*>(d.rowstat + (long)m * 2) : make_float2(0.f, 1.f);
  }
  auto rownorm = [&](f32x4 v, const float2 s2, const float4 c) {
    if (rn) {
      v[0] = s2.y * (v[0] - s2.x * c.x); v[1] = s2.y * (v[1] - s2.x * c.y);
      v[2] = s2.y * (v[2] - s2.x * c.z); v[3] = s2.y * (v[3] - s2.x * c.w);
    }
    return v;
  };
#pragma unroll
  for (int r = 0; r < R; ++r) {
    // residual fetches of the round first (one memory round trip for MI_R x NI_OUT pieces)
    half4_t rpre[NI_OUT][MI_R];
    if (res16 && !GEGLU) {
      const half_t* rb = reinterpret_cast<const half_t*>(d.res) + r_off;
#pragma unroll
      for (int no = 0; no < NI_OUT; ++no) {
        int n = n0_out + wn * 16 * NI_OUT + no * 16 + n_l;
        if (n >= n_total_out) n = 0;
#pragma unroll
        for (int mj = 0; mj < MI_R; ++mj) {
          int m = m0 + wm * 16 * MI + (r * MI_R + mj) * 16 + m_l;
          if (m >= d.M) m = d.M - 1;
          rpre[no][mj] = *reinterpret_cast<const half4_t*>(rb + (long)m * d.ldr + n);
        }
      }
    }
#pragma unroll
    for (int no = 0; no < NI_OUT; ++no) {
      const int n_out = n0_out + wn * 16 * NI_OUT + no * 16 + n_l;
      const int nn = n_out < n_total_out ? n_out : 0;
#pragma unroll
      for (int mj = 0; mj < MI_R; ++mj) {
        const int mi = r * MI_R + mj;
        int m = m0 + wm * 16 * MI + mi * 16 + m_l;
        if (m >= d.M) m = d.M - 1;
        f32x4 v;
        if constexpr (GEGLU) {
          v = epilogue_value4<true>(d, r_off, m, nn, rownorm(acc[2 * no][mi], st[mi], cv[no]), rownorm(acc[2 * no + 1][mi], st[mi], cg[no]),
                                    bv[no], bg[no], false, (half4_t){0, 0, 0, 0});
        } else {
          v = epilogue_value4<false>(d, r_off, m, nn, rownorm(acc[no][mi], st[mi], cv[no]), acc[no][mi], bv[no], bv[no], res16,
                                     res16 ? rpre[no][mj] : (half4_t){0, 0, 0, 0});
        }
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
        *reinterpret_cast<half4_t*>(lds + (wm * WROWS + mj * 16 + m_l) * CROW + (wn * 16 * NI_OUT + no * 16 + n_l) * 2) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // whole rows, 16 bytes per lane
#pragma unroll
    for (int i = 0; i < (RROWS * CPR + NT - 1) / NT; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / CPR, ch = idx - row * CPR;
      const int wr = row / WROWS;
      const int m = m0 + wr * 16 * MI + r * WROWS + (row - wr * WROWS), n = n0_out + ch * 8;
#ifdef LGD_GEMM_ABLATION                 // tools: epi bit 17 = no global stores (one dummy lane keeps the data path alive)
      if ((d.epi & (1 << 17)) && idx != 0) continue;
#endif
      if (idx < RROWS * CPR && m < d.M && n < n_total_out) {
        const uint4 val = *reinterpret_cast<const uint4*>(lds + row * CROW + ch * 16);
        *reinterpret_cast<uint4*>(cbase + (long)m * d.ldc + n) = val;
      }
    }
    // Between rounds the staging rows are rewritten; behind the LAST round of a persistent workgroup (RES_PREFETCH ==
    // !PERSIST) the next output tile's K loop issues LDS-DMA into this very stage before its first barrier — every
    // wave's staged-row reads must have retired before any wave gets there.
    if (r + 1 < R || !RES_PREFETCH) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

// RES16 (an fp16 residual is added; never with GEGLU) and RN (LGD_EPI_ROWNORM) are COMPILE-TIME here (round 4): with
// runtime flags every 4-output piece carried the dispatch of all residual forms and the row-norm select — ~94
// instructions and six uniform branches per piece, 1500 per wave against the 2560 MFMA cycles of a K = 320 main loop
// (the epilogue was 45-60 % of the short-K GEMMs, tools/gemm_abl.sh).  The call site picks one of six specialisations
// once per tile; an fp32 residual or GEGLU + residual take the register-layout epilogue instead.
template <int MI, int NI, int WM, int WN, int STAGE_B, bool GEGLU, bool RES_PREFETCH, bool RES16, bool RN>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmArgs& ga, f32x4 (&acc)[NI][MI], int m0, int n0, int wm,
                                                  int wn, int lane, int tid, long c_off, long r_off,
                                                  unsigned char* lds) {
  static_assert(!(GEGLU && RES16), "GEGLU outputs take no residual here");
  const LgdGemmDesc& d = ga.d;
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = WM * 16 * MI, BN = WN * 16 * NI;
  constexpr int BN_OUT = GEGLU ? BN / 2 : BN;
  constexpr int NI_OUT = GEGLU ? NI / 2 : NI;           // output column tiles of a wave
  constexpr int CROW = BN_OUT * 2 + 16;                 // bytes per staged row
  // rounds over the 16-row tiles (mi) of EVERY wave: all waves work in both phases of a round, and a round still
  // covers whole output rows (16 MI_R rows of each of the WM wave rows)
  constexpr int R = (BM * CROW <= STAGE_B) ? 1 : ((BM / 2) * CROW <= STAGE_B) ? 2 : 4;
  static_assert(MI % R == 0 && (BM / R) * CROW <= STAGE_B, "staged rows must fit one ring stage");
  constexpr int MI_R = MI / R;
  constexpr int WROWS = 16 * MI_R;                      // rows of one wave row per round
  constexpr int RROWS = WM * WROWS;
  constexpr int CPR = BN_OUT / 8;                       // 16-byte chunks per row
  const int m_l = lane & 15, n_l = (lane >> 4) * 4;
  const int n0_out = GEGLU ? n0 / 2 : n0;
  const int n_total_out = GEGLU ? d.N / 2 : d.N;
  half_t* cbase = reinterpret_cast<half_t*>(d.c) + c_off;
  const f32x2_t alpha2 = {d.alpha, d.alpha};
  // parameters of this wave's columns (biases, column sums of the folded LayerNorm) and rows (clamped index, row
  // statistics): one batch of loads, one memory round trip, before the first accumulator is touched
  // LEAN (the 256 x 256 tile: 128 accumulators): column sums are fetched per column block inside the round loop and the
  // residual's column select is recomputed — hoisted they spill
  constexpr bool LEAN = MI * NI >= 32;
  // LEAN2 (the phase-split tiles: 128 / 160 accumulators): biases too are fetched per column block inside the round loop,
  // row statistics per round, the copy-out loop stays rolled — hoisted / unrolled they spilled up to 99 registers
  constexpr bool LEAN2 = MI * NI >= 32 && WN == 4;       // the phase-split tiles (2 x 4 waves)
  float4 bv[NI_OUT], bg[NI_OUT], cv[NI_OUT], cg[NI_OUT];
  int ncol[NI_OUT];                                     // clamped output column of the residual fetch
#pragma unroll
  for (int no = 0; no < NI_OUT; ++no) {
    const int n_out = n0_out + wn * 16 * NI_OUT + no * 16 + n_l;
    const int n_in = GEGLU ? n0 + wn * 16 * NI + no * 32 + n_l : n_out;
    const bool ok = n_out < n_total_out;
    ncol[no] = ok ? n_out : 0;
    bv[no] = make_float4(0.f, 0.f, 0.f, 0.f);
    bg[no] = bv[no];
    cv[no] = bv[no];
    cg[no] = bv[no];
#ifdef LGD_GEMM_ABLATION                 // tools: epi bit 18 = no parameter loads in the epilogue
    if (d.epi & (1 << 18)) continue;
#endif
    if (ok && !LEAN2) {
      bv[no] = ld_bias_sum4(d, n_in);
      if (GEGLU && d.bias) bg[no] = ld_bias4(d.bias, n_in + 16);
      if constexpr (RN && !LEAN) {
        cv[no] = *reinterpret_cast<const float4*>(d.colsum + n_in);
        if (GEGLU) cg[no] = *reinterpret_cast<const float4*>(d.colsum + n_in + 16);
      }
    }
  }
  int mrow[MI];
  float2 st[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int m = m0 + wm * 16 * MI + mi * 16 + m_l;
    if (m >= d.M) m = d.M - 1;
    mrow[mi] = m;
    st[mi] = make_float2(0.f, 1.f);
    if constexpr (RN && !LEAN2) st[mi] = *reinterpret_cast<const float2*>(d.rowstat + (long)m * 2);
  }
  // rstd (v - mean colsum) + bias on the two halves of a piece
  auto normed = [&](const f32x4& v, const float2 s2, const float4 c, const float4 b, f32x2_t& lo, f32x2_t& hi) {
    lo = (f32x2_t){v[0], v[1]};
    hi = (f32x2_t){v[2], v[3]};
    if constexpr (RN) {
      const f32x2_t mean = {s2.x, s2.x}, rstd = {s2.y, s2.y};
      lo = rstd * (lo - mean * (f32x2_t){c.x, c.y});
      hi = rstd * (hi - mean * (f32x2_t){c.z, c.w});
    }
    lo += (f32x2_t){b.x, b.y};
    hi += (f32x2_t){b.z, b.w};
  };
  unsigned char* lw = lds + (wm * WROWS + m_l) * CROW + (wn * 16 * NI_OUT + n_l) * 2;   // this lane's staging origin
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if constexpr (RN && LEAN2) {
#pragma unroll
      for (int mj = 0; mj < MI_R; ++mj)
        st[r * MI_R + mj] = *reinterpret_cast<const float2*>(d.rowstat + (long)mrow[r * MI_R + mj] * 2);
    }
    // residual fetches of the round first (one memory round trip for MI_R x NI_OUT pieces)
    half4_t rpre[NI_OUT][MI_R];
    if constexpr (RES16) {
      const half_t* rb = reinterpret_cast<const half_t*>(d.res) + r_off;
#pragma unroll
      for (int mj = 0; mj < MI_R; ++mj) {
        const half_t* rrow = rb + (long)mrow[r * MI_R + mj] * d.ldr;
#pragma unroll
        for (int no = 0; no < NI_OUT; ++no) {
          int nc = ncol[no];
          if constexpr (LEAN) {
            const int n_out = n0_out + wn * 16 * NI_OUT + no * 16 + n_l;
            nc = n_out < n_total_out ? n_out : 0;
          }
          rpre[no][mj] = *reinterpret_cast<const half4_t*>(rrow + nc);
        }
      }
    }
#pragma unroll
    for (int no = 0; no < NI_OUT; ++no) {
      if constexpr ((RN && LEAN) || LEAN2) {
        const int n_out = n0_out + wn * 16 * NI_OUT + no * 16 + n_l;
        const int n_in = GEGLU ? n0 + wn * 16 * NI + no * 32 + n_l : n_out;
        if (n_out < n_total_out) {
          if constexpr (RN) {
            cv[no] = *reinterpret_cast<const float4*>(d.colsum + n_in);
            if (GEGLU) cg[no] = *reinterpret_cast<const float4*>(d.colsum + n_in + 16);
          }
          if constexpr (LEAN2) {
            bv[no] = ld_bias_sum4(d, n_in);
            if (GEGLU && d.bias) bg[no] = ld_bias4(d.bias, n_in + 16);
          }
        }
      }
#pragma unroll
      for (int mj = 0; mj < MI_R; ++mj) {
        const int mi = r * MI_R + mj;
        f32x2_t lo, hi;
        if constexpr (GEGLU) {
          f32x2_t glo, ghi;
          normed(acc[2 * no][mi], st[mi], cv[no], bv[no], lo, hi);
          normed(acc[2 * no + 1][mi], st[mi], cg[no], bg[no], glo, ghi);
#ifdef LGD_GEMM_ABLATION                 // tools: epi bit 16 = GEGLU without the GELU arithmetic
          if (d.epi & (1 << 16)) { lo *= glo; hi *= ghi; } else
#endif
          { lo *= gelu2_f(glo); hi *= gelu2_f(ghi); }
        } else {
          normed(acc[no][mi], st[mi], cv[no], bv[no], lo, hi);
        }
        lo *= alpha2;
        hi *= alpha2;
        if constexpr (RES16) {
          const half4_t rr = rpre[no][mj];
          lo += (f32x2_t){(float)rr[0], (float)rr[1]};
          hi += (f32x2_t){(float)rr[2], (float)rr[3]};
        }
        const half4_t o = {(half_t)lo[0], (half_t)lo[1], (half_t)hi[0], (half_t)hi[1]};
        *reinterpret_cast<half4_t*>(lw + mj * 16 * CROW + no * 32) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // whole rows, 16 bytes per lane (LEAN2: a rolled loop — unrolled, the five chunks' 64-bit addresses were all computed
    // ahead of the barrier above and spilled)
#pragma unroll LEAN2 ? 1 : (RROWS * CPR + NT - 1) / NT
    for (int i = 0; i < (RROWS * CPR + NT - 1) / NT; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / CPR, ch = idx - row * CPR;
      const int wr = row / WROWS;
      const int m = m0 + wr * 16 * MI + r * WROWS + (row - wr * WROWS), n = n0_out + ch * 8;
#ifdef LGD_GEMM_ABLATION                 // tools: epi bit 17 = no global stores (one dummy lane keeps the data path alive)
      if ((d.epi & (1 << 17)) && idx != 0) continue;
#endif
      if (idx < RROWS * CPR && m < d.M && n < n_total_out) {
        const uint4 val = *reinterpret_cast<const uint4*>(lds + row * CROW + ch * 16);
        *reinterpret_cast<uint4*>(cbase + (long)m * d.ldc + n) = val;
      }
    }
    // Between rounds the staging rows are rewritten; behind the LAST round of a persistent workgroup (RES_PREFETCH ==
    // !PERSIST) the next output tile's K loop issues LDS-DMA into this very stage before its first barrier — every
    // wave's staged-row reads must have retired before any wave gets there.
    if (r + 1 < R || !RES_PREFETCH) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

template <int MI, int NI>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs ga) {
  constexpr int BM = 32 * MI;
  constexpr int BN = 32 * NI;
  constexpr int A_IT = BM * 8 / 256;  // 16-byte segments per thread for the A tile
  constexpr int B_IT = BN * 8 / 256;
  static_assert(A_IT >= 1 && B_IT >= 1, "tile too small");

  __shared__ __attribute__((aligned(16))) half_t smem[(BM + BN) * LDS_LD];
  half_t* As = smem;
  half_t* Bs = smem + BM * LDS_LD;

  const LgdGemmDesc& d = ga.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  // ---- block coordinates: x -> (m tile, n tile), z -> (batch, split)
  const int n_tiles_n = (d.N + BN - 1) / BN;
  const int tile_m = blockIdx.x / n_tiles_n;
  const int tile_n = blockIdx.x - tile_m * n_tiles_n;
  const int zz = blockIdx.z;
  const int batch = zz / d.splits;
  const int split = zz - batch * d.splits;
  const int b_o = batch / d.nb_i, b_i = batch - b_o * d.nb_i;
  const long a_off = b_o * d.a_bs_o + b_i * d.a_bs_i;
  const long w_off = b_o * d.w_bs_o + b_i * d.w_bs_i;
  const long c_off = b_o * d.c_bs_o + b_i * d.c_bs_i;
  const long r_off = b_o * d.r_bs_o + b_i * d.r_bs_i;

  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int k_beg = split * ga.k_per_split;
  int k_end = k_beg + ga.k_per_split;
  if (k_end > d.K) k_end = d.K;
  const int nk = (k_end - k_beg + BK - 1) / BK;

  const half_t* A0 = reinterpret_cast<const half_t*>(d.a0) + a_off;
  const half_t* A1 = d.a1 ? reinterpret_cast<const half_t*>(d.a1) + a_off : nullptr;
  const half_t* W = reinterpret_cast<const half_t*>(d.w) + w_off;

  // ---- per-thread staging coordinates
  const int kseg = tid & 7;       // which 8-half segment of the 64-wide K tile
  const int rbase = tid >> 3;     // row within a 32-row pass
  const int cin = ga.cin;
  const bool conv = d.taps == 9;

  // A rows of this thread: pixel decomposition (conv) or plain row
  int a_iy0[A_IT], a_ix0[A_IT];
  long a_row[A_IT];  // plain: row index; conv: b*hin*win
  bool a_ok[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int m = m0 + rbase + 32 * i;
    a_ok[i] = m < d.M;
    if (conv) {
      int hw = d.hout * d.wout;
      int b = m / hw;
      int rem = m - b * hw;
      int oy = rem / d.wout;
      int ox = rem - oy * d.wout;
      a_iy0[i] = oy * d.stride - 1;
      a_ix0[i] = ox * d.stride - 1;
      a_row[i] = (long)b * d.hin * d.win;
    } else {
      a_iy0[i] = 0; a_ix0[i] = 0;
      a_row[i] = m;
    }
  }
  // running (tap, channel) of this thread's K segment
  int k_cur = k_beg + kseg * 8;
  int tap = conv ? k_cur / cin : 0;
  int ch = k_cur - tap * cin;

  uint4 a_reg[A_IT], b_reg[B_IT];

  auto load_tile = [&]() {
    // A
    const bool k_ok = k_cur < k_end;
    int ky = 0, kx = 0;
    if (conv) { ky = tap / 3; kx = tap - ky * 3; }
    const bool src1 = ch >= d.c0;
    const half_t* src = src1 ? A1 : A0;
    const long ld = src1 ? d.lda1 : d.lda0;
    const int cc = src1 ? ch - d.c0 : ch;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      bool ok = a_ok[i] && k_ok;
      long row = a_row[i];
      if (conv) {
        int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
        if (d.ups) {
          // ups=1: nearest-2x upsampled input; ups=2: zero-inserted input (only even coordinates
          // carry data) — the input-gradient of a stride-2 convolution.
          ok = ok && iy >= 0 && ix >= 0 && iy < 2 * d.hin && ix < 2 * d.win;
          if (d.ups == 2) ok = ok && !((iy | ix) & 1);
          iy >>= 1; ix >>= 1;
        } else {
          ok = ok && iy >= 0 && ix >= 0 && iy < d.hin && ix < d.win;
        }
        row += (long)iy * d.win + ix;
      }
      if (ok) a_reg[i] = *reinterpret_cast<const uint4*>(src + row * ld + cc);
      else a_reg[i] = make_uint4(0, 0, 0, 0);
    }
    // B (weights)
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      int n = n0 + rbase + 32 * i;
      bool ok = (n < d.N) && k_ok;
      if (ok) b_reg[i] = *reinterpret_cast<const uint4*>(W + (long)n * d.ldw + k_cur);
      else b_reg[i] = make_uint4(0, 0, 0, 0);
    }
    // advance to the next K tile
    k_cur += BK;
    ch += BK;
    if (conv) {
      while (ch >= cin) { ch -= cin; ++tap; }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      *reinterpret_cast<uint4*>(As + (rbase + 32 * i) * LDS_LD + kseg * 8) = a_reg[i];
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      *reinterpret_cast<uint4*>(Bs + (rbase + 32 * i) * LDS_LD + kseg * 8) = b_reg[i];
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fk = (lane >> 4) * 8;

  if (nk > 0) {
    load_tile();
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) load_tile();
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
        half8_t af[MI], bf[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          af[mi] = *reinterpret_cast<const half8_t*>(
              As + (wm * 16 * MI + mi * 16 + frow) * LDS_LD + kk * 32 + fk);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          bf[ni] = *reinterpret_cast<const half8_t*>(
              Bs + (wn * 16 * NI + ni * 16 + frow) * LDS_LD + kk * 32 + fk);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] =
                __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ni], af[mi], acc[ni][mi], 0, 0, 0);
      }
      __syncthreads();
      if (kt + 1 < nk) {
        store_tile();
        __syncthreads();
      }
    }
  }

  gemm_epilogue<MI, NI>(ga, acc, m0, n0, wm, wn, lane, batch, split, c_off, r_off, reinterpret_cast<int*>(smem));
}

// 128 B of zeros: the source of every conv-padding / out-of-range segment of the LDS-DMA variant
// (an LDS-DMA lane cannot be predicated into writing zeros, so it is pointed here instead).
__device__ __attribute__((aligned(128))) unsigned int g_zero_line[32];

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// Main-loop variant 2: operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4), no staging
// VGPRs and no ds_write pass; two LDS stages, one barrier per K tile.
//   * a wave-instruction moves 64 x 16 B = 8 tile rows of 128 B; the DMA's LDS image is lane-linear,
//     so LDS rows are unpadded [row][64 halfs] and the bank spread comes from an XOR swizzle applied on
//     the SOURCE side: the lane that fills physical 16-B slot p of row r fetches logical segment
//     p ^ ((r>>1)&7).  The 8 lanes of a row still cover one full 128-B line, so HBM/L2 traffic is
//     unchanged; the fragment reads (16 rows x 4 segments per ds_read_b128) hit 16 distinct slots of
//     the 256-B bank row in each of the 4 hardware lane groups -> conflict-free.
//   * requires every K tile to be full (K % 64 == 0) — true for every UNet contraction; the entry
//     point routes other shapes to the register-staged variant.
//   * WM waves along M x 2 along N (WM = 2: 256 threads, WM = 4: 512 threads).  The 8-wave 256-row
//     tiles exist because the vector-memory path (64 B/clk/CU) bounds this loop before the matrix
//     cores do: a 128x160 tile moves 1 B per 71 flop, right at the ridge (4069 flop/clk/CU / 64
//     B/clk/CU); 256x320 moves 1 B per 142 flop.
template <int MI, int NI, int WM>
__global__ __launch_bounds__(128 * WM) void gemm_dma_kernel(const GemmArgs ga) {
  constexpr int NT = 128 * WM;       // threads
  constexpr int RP = NT / 8;         // tile rows filled per pass (one 16-B segment per thread)
  constexpr int BM = WM * 16 * MI;
  constexpr int BN = 32 * NI;
  static_assert(BM % RP == 0 && BN % RP == 0, "tile rows must be a multiple of the staging pass");
  constexpr int A_IT = BM / RP;
  constexpr int B_IT = BN / RP;
  constexpr int STAGE = (BM + BN) * BK;  // halfs per stage

  __shared__ __attribute__((aligned(1024))) half_t smem[2 * STAGE];

  const LgdGemmDesc& d = ga.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;

  // ---- block coordinates. Workgroups are dealt round-robin to the 8 XCDs; remap so that each XCD
  // walks a contiguous range of tiles (n fastest): the tiles that share A rows share one L2.
  const int n_tiles_n = (d.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / n_tiles_n;
  const int tile_n = bid - tile_m * n_tiles_n;
  const int zz = blockIdx.z;
  const int batch = zz / d.splits;
  const int split = zz - batch * d.splits;
  const int b_o = batch / d.nb_i, b_i = batch - b_o * d.nb_i;
  const long a_off = b_o * d.a_bs_o + b_i * d.a_bs_i;
  const long w_off = b_o * d.w_bs_o + b_i * d.w_bs_i;
  const long c_off = b_o * d.c_bs_o + b_i * d.c_bs_i;
  const long r_off = b_o * d.r_bs_o + b_i * d.r_bs_i;

  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int k_beg = split * ga.k_per_split;
  int k_end = k_beg + ga.k_per_split;
  if (k_end > d.K) k_end = d.K;
  const int nk = (k_end - k_beg) / BK;

  const half_t* A0 = reinterpret_cast<const half_t*>(d.a0) + a_off;
  const half_t* A1 = d.a1 ? reinterpret_cast<const half_t*>(d.a1) + a_off : nullptr;
  const half_t* W = reinterpret_cast<const half_t*>(d.w) + w_off;
  const half_t* zero = reinterpret_cast<const half_t*>(g_zero_line);

  // ---- per-thread staging coordinates
  const int rrow = tid >> 3;                       // row within an RP-row pass
  const int kseg = (tid & 7) ^ ((rrow >> 1) & 7);  // logical 8-half segment this lane fetches
  const int cin = ga.cin;
  const bool conv = d.taps == 9;

  int a_iy0[A_IT], a_ix0[A_IT];
  long a_row[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int m = m0 + rrow + RP * i;
    if (m >= d.M) m = d.M - 1;  // rows past M are computed on valid data and never stored
    if (conv) {
      int hw = d.hout * d.wout;
      int b = m / hw;
      int rem = m - b * hw;
      int oy = rem / d.wout;
      int ox = rem - oy * d.wout;
      a_iy0[i] = oy * d.stride - 1;
      a_ix0[i] = ox * d.stride - 1;
      a_row[i] = (long)b * d.hin * d.win;
    } else {
      a_iy0[i] = 0; a_ix0[i] = 0;
      a_row[i] = m;
    }
  }
  const half_t* w_row[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    int n = n0 + rrow + RP * i;
    if (n >= d.N) n = d.N - 1;
    w_row[i] = W + (long)n * d.ldw + k_beg + kseg * 8;
  }
  const int k_first = k_beg + kseg * 8;
  int tap = conv ? k_first / cin : 0;
  int ch = k_first - tap * cin;

  // Source pointers are re-derived only when the K walk enters a new tap or crosses from the first
  // to the second (concatenated) source; inside a run they just advance by one K tile.
  const half_t* a_ptr[A_IT];
  bool a_ok[A_IT];
  bool rederive = true;
  auto issue_tile = [&](int stage) {
    half_t* As = smem + stage * STAGE;
    half_t* Bs = As + BM * BK;
    if (rederive) {
      int ky = 0, kx = 0;
      if (conv) { ky = tap / 3; kx = tap - ky * 3; }
      const bool src1 = ch >= d.c0;
      const half_t* src = src1 ? A1 : A0;
      const long ld = src1 ? d.lda1 : d.lda0;
      const int cc = src1 ? ch - d.c0 : ch;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        bool ok = true;
        long row = a_row[i];
        if (conv) {
          int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
          if (d.ups) {
            ok = iy >= 0 && ix >= 0 && iy < 2 * d.hin && ix < 2 * d.win;
            if (d.ups == 2) ok = ok && !((iy | ix) & 1);
            iy >>= 1; ix >>= 1;
          } else {
            ok = iy >= 0 && ix >= 0 && iy < d.hin && ix < d.win;
          }
          row += (long)iy * d.win + ix;
        }
        a_ok[i] = ok;
        a_ptr[i] = src + row * ld + cc;
      }
      rederive = false;
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const half_t* p = a_ok[i] ? a_ptr[i] : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)p, (lds_ptr_t)(As + (RP * i + 8 * wid) * BK), 16, 0, 0);
      a_ptr[i] += BK;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)w_row[i], (lds_ptr_t)(Bs + (RP * i + 8 * wid) * BK), 16, 0, 0);
      w_row[i] += BK;
    }
    const int ch_prev = ch;
    ch += BK;
    if (ch_prev < d.c0 && ch >= d.c0) rederive = true;
    if (ch >= cin) {
      rederive = true;
      do { ch -= cin; ++tap; } while (ch >= cin);
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fg = lane >> 4;
  const int fsw = (frow >> 1) & 7;

  if (nk > 0) issue_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed (this wave's DMAs: vmcnt; the other waves': barrier), and every wave has
    // finished reading the other stage (it was consumed in iteration kt-1).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) issue_tile((kt + 1) & 1);
    const half_t* As = smem + (kt & 1) * STAGE;
    const half_t* Bs = As + BM * BK;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      const int slot = ((kk * 4 + fg) ^ fsw) * 8;
      half8_t af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        af[mi] = *reinterpret_cast<const half8_t*>(As + (wm * 16 * MI + mi * 16 + frow) * BK + slot);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        bf[ni] = *reinterpret_cast<const half8_t*>(Bs + (wn * 16 * NI + ni * 16 + frow) * BK + slot);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
  }

  gemm_epilogue<MI, NI>(ga, acc, m0, n0, wm, wn, lane, batch, split, c_off, r_off, reinterpret_cast<int*>(smem));
}

// ---- pieces of the pipelined main loop that must not be left to the compiler's own wait insertion --------
// LDS fragment reads and every wait are inline asm: hipcc's s_waitcnt pass is conservative across the loop
// back edge (it emits lgkmcnt(0) behind freshly issued prefetch reads, serialising them with the MFMAs they
// were meant to overlap) and treats a pending LDS-DMA as a hazard for every LDS read it can see.
template <int OFF>
__device__ __forceinline__ void lds_read128(half8_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int BASE, int STRIDE, int N, int... I>
__device__ __forceinline__ void lds_read_frags(half8_t (&f)[N], unsigned addr, std::integer_sequence<int, I...>) {
  (lds_read128<BASE + I * STRIDE>(f[I], addr), ...);
}
// Wait for this wave's LDS reads (and, with VM >= 0, until at most VM vector-memory operations are in flight).
// The fragments are named read-write so that no MFMA consuming them can be scheduled above the wait.
template <int VM, int NA, int NB>
__device__ __forceinline__ void wait_frags(half8_t (&a)[NA], half8_t (&b)[NB]) {
  static_assert(NA <= 8 && NB <= 10, "operand count");
  if constexpr (VM >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
  if constexpr (NA == 4 && NB == 5)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]),
                 "+v"(b[2]), "+v"(b[3]), "+v"(b[4]));
  else if constexpr (NA == 8 && NB == 8)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]),
                 "+v"(a[6]), "+v"(a[7]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]),
                 "+v"(b[6]), "+v"(b[7]));
  else if constexpr (NA == 4 && NB == 8)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]),
                 "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
  else if constexpr (NA == 4 && NB == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]),
                 "+v"(b[2]), "+v"(b[3]));
  else if constexpr (NA == 4 && NB == 2)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]));
  else if constexpr (NA == 2 && NB == 5)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                 "+v"(b[4]));
  else if constexpr (NA == 2 && NB == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
  else if constexpr (NA == 2 && NB == 2)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
  else if constexpr (NA == 1 && NB == 5)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]));
  else if constexpr (NA == 1 && NB == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
  else if constexpr (NA == 1 && NB == 2)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]));
  else
    static_assert(NA == 0, "add the operand list for this tile");
}

// Main-loop variant 3: 8 waves (WM x WN), one workgroup per CU, 256-row class tiles, THREE LDS stages.
//   The 4-wave loop above is bound by the CU's vector-memory path (a 128x160 tile needs 58 B/clk/CU of
//   global->LDS traffic at full MFMA rate, the path delivers ~64) and drains the DMA queue at every K tile
//   (vmcnt(0) + barrier).  Here a 256x160 tile needs 42 B/clk/CU (256x128: 48), the DMA of K tile kt+2 is
//   issued while tile kt is multiplied and is only waited for (counted vmcnt) one full K tile later, so a
//   DMA has a whole K tile (~2.5k cycles) of flight time and nothing in the loop ever waits for vmcnt(0);
//   the barrier is a bare s_barrier (a __syncthreads() would drain the LDS-DMA queue).
//   Per K tile and wave (fragment reads run one k-half ahead of the MFMAs that consume them):
//       wait: frags(kt, half 0) landed          | read frags(kt, half 1)
//       MFMA half 0, the T DMA instructions of tile kt+2 interleaved between MFMA groups
//       wait: vmcnt(T) = tile kt+1 landed (own share), lgkmcnt(0)   | s_barrier
//       read frags(kt+1, half 0)                | MFMA half 1
//   Every wave issues the same number T of DMA instructions per tile (a 160-row B tile = 20 row groups over 8
//   waves: two whole groups each plus one half-masked instruction), and tiles past the end of K are "loaded"
//   from the zero line, so the loop body has no tail case and vmcnt(T) always means "tile kt+1 landed".
//   LDS image and source-side swizzle are those of gemm_dma_kernel (128-B rows, conflict-free ds_read_b128).
//   Hazards: stage (kt+2)%3 held tile kt-1, whose last fragment reads were waited for (lgkmcnt(0)) before the
//   barrier of iteration kt-1 by every wave; tile kt+1 is read only after the barrier of iteration kt, in
//   front of which every wave waited for its own share of that tile's DMAs.
//   Requires cin % 64 == 0 and c0 % 64 == 0 (a K tile never straddles a tap or a source; true for every
//   UNet contraction), checked by the launcher.
// ABL (tools only; results are wrong by design): 1 = no DMA in the loop, 2 = no fragment reads in the loop,
// 4 = no MFMA, 8 = no barrier, 16 = no epilogue — the cost of each ingredient by removal.
// CM ("chunk-major"): 3x3 stride-1 single-source convolutions walk K as (channel chunk, tap) instead of the
// weight layout's (tap, channel): the nine taps of one 64-channel chunk touch the same 6 x 66 pixel halo of
// the input, 1/5..1/20 of the map's bytes, so the re-reads hit the XCD's L2 instead of falling out of it
// between taps (the weight rows are simply visited in a different order; partial sums are order-free).
// Two-stage rings of the 128-row tiles are 64-72 KB: TWO workgroups share a CU (launch bound 2 => at most 128 VGPRs), so
// that one's epilogue (GEGLU arithmetic + stores: 60 % of the K = 320 feed-forward GEMM, tools/gemm_abl.sh ABL=16) runs
// beside the other's MFMAs — waves of ONE workgroup move through the phases in lock-step and cannot overlap them.
template <int MI, int NI, int NS>
constexpr int pipe_occupancy() { return (NS == 2 && MI * NI <= 10) ? 2 : 1; }

template <int MI, int NI, int WM, int WN, int NS, bool CM, bool PERSIST = false, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN, (pipe_occupancy<MI, NI, NS>())) void gemm_pipe_kernel(const GemmArgs ga) {
  constexpr int NW = WM * WN;
  static_assert(NW == 8 || NW == 4, "eight waves, or four (one per SIMD, the whole register file each)");
  constexpr int BM = WM * 16 * MI;
  constexpr int BN = WN * 16 * NI;
  constexpr int GA = BM / 8, GB = BN / 8;            // 8-row groups (one DMA wave-instruction each)
  static_assert(GA % NW == 0, "A rows");
  static_assert(GB % NW == 0 || GB % NW == NW / 2, "B rows: whole groups per wave plus at most a half group");
  static_assert(NS != 2 || GB % NW == 0, "the two-stage issue path has no half-group instruction");
  constexpr int A_IT = GA / NW, B_FULL = GB / NW;
  constexpr bool B_HALF = (GB % NW) != 0;
  constexpr int B_IT = B_FULL + (B_HALF ? 1 : 0);
  constexpr int T_DMA = A_IT + B_IT;                         // DMA instructions per wave per K tile
  // TWO-STAGE RING (round 4, the 256 x 256 tile: a stage is 64 KB).  Tile kt sits in stage kt % 2; the stage of tile kt
  // is free for tile kt + 2 once every wave's last fragment reads of it (k-half 1, requested at the top of iteration
  // kt) have retired — they are waited for in front of the barrier in the middle of iteration kt — so ALL of tile
  // kt + 2's DMA instructions go beside k-half 1 of iteration kt and have until the middle of iteration kt + 1 to
  // land: one whole K tile of flight time, which on a tile of this size is as many cycles (128 MFMAs per wave) as
  // the two K tiles a three-stage ring gives the smaller tiles.  The middle wait is then vmcnt(0).
  constexpr bool TWO = NS == 2;
  constexpr int T1 = TWO ? 0 : (T_DMA + 1) / 2, T2 = T_DMA - T1;   // issued beside k-half 0 / k-half 1
  constexpr int STAGE = (BM + BN) * BK;                      // halfs per stage
  constexpr int STAGE_B = STAGE * 2;                         // bytes
  static_assert(NS >= 2 && NS * STAGE_B <= 160 * 1024, "LDS");
  static_assert(!(TWO && PERSIST), "the two-stage ring has no tile-crossing form");
  constexpr int D = TWO ? 2 : NS - 1;                        // K tiles in flight behind the prologue
  constexpr int NMMA = MI * NI;                              // MFMAs per k-half

  extern __shared__ __attribute__((aligned(1024))) half_t smem[];

  const LgdGemmDesc& d = ga.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid - wm * WN;

#ifdef LGD_GEMM_ABLATION
  if (ga.stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
    for (int i = 0; i < ga.stagger; ++i) __builtin_amdgcn_s_sleep(127);
#endif
  const int n_tiles_n = (d.N + BN - 1) / BN;
  // PERSISTENT TILE LOOP: the launcher may start fewer workgroups than output tiles; workgroup b then computes
  // tiles b, b + gridDim.x, ... and keeps its DMA ring running across tile boundaries: the first D K tiles of the
  // next output tile are fetched while the last K tiles of the current one are multiplied and its epilogue runs,
  // so only the first tile of a workgroup pays the pipeline-fill latency.  (With gridDim.x == number of tiles this
  // is the one-tile-per-workgroup kernel.)  Consecutive workgroup ids land on consecutive XCDs, and b + i*gridDim.x
  // keeps b's XCD when gridDim.x % 8 == 0, so the XCD-contiguous tile remap below holds for every tile of a workgroup.
  const int total_tiles = ((d.M + BM - 1) / BM) * n_tiles_n;
  // GROUPED WALK (round 4): with the plain order (n fastest) every row panel streams the WHOLE weight matrix; once that
  // no longer fits the XCD's 4 MB L2 (feed-forward layers: 5120 x 640, 10240 x 1280, 1280 x 5120 weights = 6.5-26 MB)
  // each panel re-fetches it from the Infinity Cache (PMC: 265 MB read per launch for 27 MB of operands).  Walking
  // group_m row panels per weight column block keeps that block AND the group's A panels L2-resident.
  auto tile_origin = [&](int v, int& m0_, int& n0_) {
    const int q = total_tiles >> 3, r = total_tiles & 7;
    const int xcd = v & 7, idx = v >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    int tile_m, tile_n;
    if (ga.group_m > 1) {
      const int per_group = ga.group_m * n_tiles_n;
      const int g = bid / per_group, in = bid - g * per_group;
      const int n_tiles_m = total_tiles / n_tiles_n;
      int rows = n_tiles_m - g * ga.group_m;
      if (rows > ga.group_m) rows = ga.group_m;
      tile_n = in / rows;
      tile_m = g * ga.group_m + (in - tile_n * rows);
    } else {
      tile_m = bid / n_tiles_n;
      tile_n = bid - tile_m * n_tiles_n;
    }
    m0_ = tile_m * BM;
    n0_ = tile_n * BN;
  };
  const int zz = blockIdx.z;
  const int batch = zz / d.splits;
  const int split = zz - batch * d.splits;
  const int b_o = batch / d.nb_i, b_i = batch - b_o * d.nb_i;
  const long a_off = b_o * d.a_bs_o + b_i * d.a_bs_i;
  const long w_off = b_o * d.w_bs_o + b_i * d.w_bs_i;
  const long c_off = b_o * d.c_bs_o + b_i * d.c_bs_i;
  const long r_off = b_o * d.r_bs_o + b_i * d.r_bs_i;

  const int k_beg = split * ga.k_per_split;
  int k_end = k_beg + ga.k_per_split;
  if (k_end > d.K) k_end = d.K;
  const int nk = (k_end - k_beg) / BK;

  const half_t* A0 = reinterpret_cast<const half_t*>(d.a0) + a_off;
  const half_t* A1 = d.a1 ? reinterpret_cast<const half_t*>(d.a1) + a_off : nullptr;
  const half_t* W = reinterpret_cast<const half_t*>(d.w) + w_off;
  const half_t* zero = reinterpret_cast<const half_t*>(g_zero_line);

  // ---- staging: wave w fills the 8-row groups w, w+8, ... of A and of B; lane -> (row in group, 16-B slot).
  // The swizzle term (row>>1)&7 of row = 8*(8i + w) + lrow does not depend on i, so a lane fetches the same
  // logical K segment in every group it fills (as in gemm_dma_kernel).
  const int lrow = lane >> 3;
  const int kseg = (lane & 7) ^ ((((wid & 1) << 2) + (lrow >> 1)) & 7);
  const int cin = ga.cin;
  // persistent workgroups serve plain (taps == 1) contractions only (the launcher sees to it): a compile-time `false`
  // removes the convolution's gather state from the tile-crossing kernels, which spilled 62-170 registers with it
  // so do the two-stage (256 x 256) workgroups, which also keep NO per-row issue state at all: a wave of theirs issues 16
  // DMA instructions per K tile, and 16 row pointers + 16 weight-row pointers on top of 256 accumulators and 128
  // fragment registers do not fit; their source addresses are recomputed per instruction (a 64-bit multiply-add each,
  // against 8 MFMAs per DMA instruction)
  const bool conv = (PERSIST || TWO) ? false : d.taps == 9;
  int n0_iss = 0;
  // K position of the next tile to issue.  Tap-major: (tap0, ch0) follow the weight layout.  Chunk-major:
  // tile t of the split's range is (chunk = t / 9, tap = t % 9).
  int tap0 = 0, ch0 = 0;
  int kt_issue = 0;

  int a_iy0[A_IT], a_ix0[A_IT];   // tap-major conv: top-left input coordinate of the 3x3 window
  long a_row[A_IT];               // tap-major: first row of the image (conv) / row index (plain)
  const half_t* a_cen[A_IT];      // chunk-major: pointer to the window's centre pixel, this lane's segment
  int a_mask[A_IT];               // chunk-major: bit `tap` set = that tap's pixel lies inside the image
  // the half-masked instruction: waves 2g and 2g+1 fill rows 0-3 / 4-7 of group 8*B_FULL + g
  const int b_grp_last = B_FULL * NW + (wid >> 1);
  const bool b_half_on = (lane >> 5) == (wid & 1);
  const half_t* w_row[B_IT];
  // issue side of one output tile: K walk back to the start of the split's range, row pointers of the tile
  int m0_iss = 0;                 // first row of the tile the issue side works on (persistent: rows are recomputed from it)
  auto setup_issue = [&](int m0, int n0) {
  m0_iss = m0;
  n0_iss = n0;
  if constexpr (TWO) { kt_issue = 0; return; }
  if (CM) { const int t = k_beg / BK; ch0 = (t / 9) * BK; tap0 = t - (t / 9) * 9; }
  else { tap0 = conv ? k_beg / cin : 0; ch0 = k_beg - tap0 * cin; }
  kt_issue = 0;
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int m = m0 + (i * NW + wid) * 8 + lrow;
    if (m >= d.M) m = d.M - 1;  // rows past M are computed on valid data and never stored
    a_iy0[i] = 0; a_ix0[i] = 0; a_row[i] = m; a_cen[i] = nullptr; a_mask[i] = 0;
    if (conv) {
      int hw = d.hout * d.wout;
      int b = m / hw;
      int rem = m - b * hw;
      int oy = rem / d.wout;
      int ox = rem - oy * d.wout;
      if (CM) {
        a_cen[i] = A0 + ((long)b * d.hin * d.win + (long)oy * d.win + ox) * d.lda0 + kseg * 8;
        int msk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
          if (iy >= 0 && ix >= 0 && iy < d.hin && ix < d.win) msk |= 1 << t;
        }
        a_mask[i] = msk;
      } else {
        a_iy0[i] = oy * d.stride - 1;
        a_ix0[i] = ox * d.stride - 1;
        a_row[i] = (long)b * d.hin * d.win;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const bool last = B_HALF && i == B_FULL;
    const int grp = last ? b_grp_last : i * NW + wid;
    // the half group's index has its own parity, hence its own swizzle term
    const int ks = last ? ((lane & 7) ^ ((((b_grp_last & 1) << 2) + (lrow >> 1)) & 7)) : kseg;
    int n = n0 + grp * 8 + lrow;
    if (n >= d.N) n = d.N - 1;
    w_row[i] = W + (long)n * d.ldw + (CM ? tap0 * cin + ch0 : k_beg) + ks * 8;
  }
  };

  const half_t* a_ptr[A_IT];
  bool a_ok[A_IT];
  // tap-major: (re)derive the A source pointers of the tile about to be issued — at tap / source changes only
  auto derive = [&]() {
    if constexpr (TWO) return;
    int ky = 0, kx = 0;
    if (conv) { ky = tap0 / 3; kx = tap0 - ky * 3; }
    const bool src1 = ch0 >= d.c0;
    const half_t* src = src1 ? A1 : A0;
    const long ld = src1 ? d.lda1 : d.lda0;
    const int cc = (src1 ? ch0 - d.c0 : ch0) + kseg * 8;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      bool ok = true;
      long row = a_row[i];
      if constexpr (PERSIST) {      // plain contraction: the row index is cheaper to recompute than to keep (2 registers each)
        int m = m0_iss + (i * NW + wid) * 8 + lrow;
        row = m >= d.M ? d.M - 1 : m;
      }
      if (conv) {
        int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
        if (d.ups) {
          ok = iy >= 0 && ix >= 0 && iy < 2 * d.hin && ix < 2 * d.win;
          if (d.ups == 2) ok = ok && !((iy | ix) & 1);
          iy >>= 1; ix >>= 1;
        } else {
          ok = iy >= 0 && ix >= 0 && iy < d.hin && ix < d.win;
        }
        row += (long)iy * d.win + ix;
      }
      a_ok[i] = ok;
      a_ptr[i] = src + row * ld + cc;
    }
  };
  // chunk-major: offset (halfs) from a window's centre pixel to the tile's tap and channel chunk (uniform)
  long cm_off = 0;
  auto cm_offset = [&]() {
    const int ky = tap0 / 3, kx = tap0 - ky * 3;
    cm_off = ((long)(ky - 1) * d.win + (kx - 1)) * d.lda0 + ch0;
  };
  // One DMA instruction of the tile being issued (j = 0..T_DMA-1), into stage offset `so` (halfs).
  auto issue_one = [&](int j, int so, bool live) {
    half_t* As = smem + so;
    half_t* Bs = As + BM * BK;
    if constexpr (TWO) {
      const int kcur = k_beg + kt_issue * BK + kseg * 8;
      if (j < A_IT) {
        int m = m0_iss + (j * NW + wid) * 8 + lrow;
        if (m >= d.M) m = d.M - 1;
        const half_t* p = live ? A0 + (long)m * d.lda0 + kcur : zero;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)p, (lds_ptr_t)(As + (j * NW + wid) * 8 * BK), 16, 0, 0);
      } else {
        const int i = j - A_IT;
        int n = n0_iss + (i * NW + wid) * 8 + lrow;
        if (n >= d.N) n = d.N - 1;
        const half_t* p = live ? W + (long)n * d.ldw + kcur : zero;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)p, (lds_ptr_t)(Bs + (i * NW + wid) * 8 * BK), 16, 0, 0);
      }
      return;
    }
    if (j < A_IT) {
      const half_t* p;
      if (CM) p = (((a_mask[j] >> tap0) & 1) && live) ? a_cen[j] + cm_off : zero;
      else p = (a_ok[j] && live) ? a_ptr[j] : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)p, (lds_ptr_t)(As + (j * NW + wid) * 8 * BK), 16, 0, 0);
    } else {
      const int i = j - A_IT;
      const half_t* p = live ? w_row[i] : zero;
      if (B_HALF && i == B_FULL) {
        if (b_half_on)
          __builtin_amdgcn_global_load_lds((glb_ptr_t)p, (lds_ptr_t)(Bs + b_grp_last * 8 * BK), 16, 0, 0);
      } else {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)p, (lds_ptr_t)(Bs + (i * NW + wid) * 8 * BK), 16, 0, 0);
      }
    }
  };
  // after a tile has been issued: advance the K walk and the source pointers (uniform control flow)
  auto advance = [&]() {
    ++kt_issue;
    if constexpr (TWO) return;
    if (CM) {
      int wstep = cin;
      if (++tap0 == 9) { tap0 = 0; ch0 += BK; wstep = BK - 8 * cin; }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) w_row[i] += wstep;
      cm_offset();
    } else {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) w_row[i] += BK;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) a_ptr[i] += BK;
      const int prev = ch0;
      ch0 += BK;
      bool red = prev < d.c0 && ch0 >= d.c0;
      if (ch0 >= cin) { ch0 -= cin; ++tap0; red = true; }
      if (red && kt_issue < nk) derive();
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fg = lane >> 4;
  const int fsw = (frow >> 1) & 7;
  // byte addresses (LDS) of this lane's fragment rows, k-half 0 / 1, relative to a stage
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned a_b0 = lds0 + ((wm * 16 * MI + frow) * BK + ((0 * 4 + fg) ^ fsw) * 8) * 2;
  const unsigned a_b1 = lds0 + ((wm * 16 * MI + frow) * BK + ((1 * 4 + fg) ^ fsw) * 8) * 2;
  constexpr int B_OFF = BM * BK * 2;       // bytes from the stage base to the B rows
  const unsigned b_b0 = a_b0 - (wm * 16 * MI) * BK * 2 + B_OFF + (wn * 16 * NI) * BK * 2;
  const unsigned b_b1 = a_b1 - (wm * 16 * MI) * BK * 2 + B_OFF + (wn * 16 * NI) * BK * 2;
  using seqA = std::make_integer_sequence<int, MI>;
  using seqB = std::make_integer_sequence<int, NI>;
  constexpr int FSTR = 16 * BK * 2;        // bytes between consecutive 16-row fragments

  half8_t af0[MI], bf0[NI], af1[MI], bf1[NI];
  // MFMAs g0..g1-1 of a k-half (linear index = ni*MI + mi)
  auto mma_range = [&](half8_t (&af)[MI], half8_t (&bf)[NI], int g0, int g1) {
#pragma unroll
    for (int g = 0; g < NMMA; ++g)
      if (g >= g0 && g < g1) {
        const int ni = g / MI, mi = g % MI;
        if constexpr (ABL & 4) {
          asm volatile("" : "+v"(acc[ni][mi]) : "v"(bf[ni]), "v"(af[mi]));
        } else {
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
      }
  };
  // `cnt` DMA instructions starting at j0, each followed by its share of the k-half's MFMAs
  auto dma_and_mma = [&](int j0, int cnt, unsigned so_b, bool live, half8_t (&af)[MI], half8_t (&bf)[NI]) {
#pragma unroll
    for (int j = 0; j < T_DMA; ++j)
      if (j < cnt) {
        if constexpr (!(ABL & 1)) issue_one(j0 + j, (int)(so_b >> 1), live);
        __builtin_amdgcn_sched_barrier(0);
        mma_range(af, bf, (NMMA * j) / cnt, (NMMA * (j + 1)) / cnt);
        __builtin_amdgcn_sched_barrier(0);
      }
    if (cnt == 0) mma_range(af, bf, 0, NMMA);
  };

  // the issue side runs D K tiles ahead of the MFMA side and crosses into the workgroup's next output tile on its
  // own: `begin_k_tile` is called before each K tile is issued
  int v_issue = blockIdx.x;
  bool issue_dead = false;           // nothing left to fetch: the remaining ring slots are fed from the zero line
  auto begin_k_tile = [&]() {
    if constexpr (!PERSIST) {
      issue_dead = kt_issue >= nk;     // one tile per workgroup: nothing behind the last K tile
      return;
    }
    if (kt_issue >= nk && !issue_dead) {
      v_issue += gridDim.x;
      if (v_issue < total_tiles) {
        int m0n, n0n;
        tile_origin(v_issue, m0n, n0n);
        setup_issue(m0n, n0n);
        if (CM) cm_offset(); else derive();
      } else {
        issue_dead = true;
      }
    }
  };

  if (nk > 0) {
    int m0, n0;
    tile_origin(blockIdx.x, m0, n0);
    setup_issue(m0, n0);
    // ---- prologue: K tiles 0..D-1 in flight, tile 0 landed, its first fragments requested
    if (CM) cm_offset(); else derive();
#pragma unroll
    for (int t = 0; t < D; ++t) {
      begin_k_tile();
#pragma unroll
      for (int j = 0; j < T_DMA; ++j) issue_one(j, t * STAGE, !issue_dead);
      advance();
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * T_DMA) : "memory");
    __builtin_amdgcn_s_barrier();
    lds_read_frags<0, FSTR>(af0, a_b0, seqA{});
    lds_read_frags<0, FSTR>(bf0, b_b0, seqB{});
    // byte offsets of the stages of K tile kt, kt+1 and of the one being issued (kt+D); the ring keeps turning
    // across output tiles
    unsigned so_cur = 0, so_nxt = STAGE_B, so_iss = (D % NS) * STAGE_B;   // two stages: the issue stage IS the current one
    auto next_stage = [](unsigned x) { x += STAGE_B; return x == NS * STAGE_B ? 0u : x; };
    auto k_loop = [&]() {
      for (int kt = 0; kt < nk; ++kt) {
        begin_k_tile();
        const bool live = !issue_dead;
        wait_frags<-1>(af0, bf0);                                  // frags(kt, half 0)
        if constexpr (!(ABL & 2)) {
          lds_read_frags<0, FSTR>(af1, a_b1 + so_cur, seqA{});     // frags(kt, half 1)
          lds_read_frags<0, FSTR>(bf1, b_b1 + so_cur, seqB{});
        }
        dma_and_mma(0, T1, so_iss, live, af0, bf0);                // first part of K tile kt+D beside k-half 0
        // K tile kt+1 landed (own share: tiles kt+2..kt+D-1 and the T1 instructions just issued may be in
        // flight; stores of a previous epilogue only make the count more conservative); own reads retired
        wait_frags<(D - 2) * T_DMA + T1>(af1, bf1);
        if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
        if constexpr (!(ABL & 2)) {
          // frags(kt+1, half 0).  A persistent workgroup requests the NEXT output tile's first fragments only behind
          // the epilogue (below): held across it they cost 36 registers on top of the accumulators, the epilogue's own
          // and the issue side's state — the 256-row tiles spilled 54-131 registers for them
          if (!PERSIST || kt + 1 < nk) {
            lds_read_frags<0, FSTR>(af0, a_b0 + so_nxt, seqA{});
            lds_read_frags<0, FSTR>(bf0, b_b0 + so_nxt, seqB{});
          }
        }
        dma_and_mma(T1, T2, so_iss, live, af1, bf1);               // rest of K tile kt+D beside k-half 1
        advance();
        so_cur = so_nxt; so_nxt = next_stage(so_nxt); so_iss = next_stage(so_iss);
      }
    };
    // fp16 outputs without split-K leave through LDS as whole rows (gemm_epilogue_lds); the stage consumed last is free
    const bool lds_epi = d.splits == 1 && !(d.epi & LGD_EPI_OUT_F32) && !(d.N & 7) && !(d.ldc & 7) && !(c_off & 7) &&
                         !(reinterpret_cast<uintptr_t>(d.c) & 15) && !((d.epi & LGD_EPI_GEGLU) && (NI & 1)) &&
                         !(d.res && (d.epi & (LGD_EPI_RES_F32 | LGD_EPI_GEGLU)));   // fp32 / GEGLU residuals: register-layout epilogue
    auto epilogue = [&](int m0_, int n0_) {
      if constexpr (ABL & 16) {       // tools: no epilogue at all (every accumulator stays live: no MFMA is dead code)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(acc[ni][mi]));
        return;
      }
      if (lds_epi) {
        unsigned char* free_stage = reinterpret_cast<unsigned char*>(smem) + so_iss;
        // one specialisation per (GEGLU, fp16 residual, folded LayerNorm): no per-piece dispatch inside
#define LGD_EPI_LDS(G, RS, RNORM) \
  gemm_epilogue_lds<MI, NI, WM, WN, STAGE_B, G, !PERSIST, RS, RNORM>(ga, acc, m0_, n0_, wm, wn, lane, tid, c_off, r_off, free_stage)
        const bool rn_ = d.epi & LGD_EPI_ROWNORM;
        if constexpr (MI * NI >= 32) {
          if (d.epi & LGD_EPI_GEGLU)
            gemm_epilogue_lds_generic<MI, NI, WM, WN, STAGE_B, true, !PERSIST>(ga, acc, m0_, n0_, wm, wn, lane, tid, c_off, r_off, free_stage);
          else
            gemm_epilogue_lds_generic<MI, NI, WM, WN, STAGE_B, false, !PERSIST>(ga, acc, m0_, n0_, wm, wn, lane, tid, c_off, r_off, free_stage);
        } else if (d.epi & LGD_EPI_GEGLU) {
          if constexpr (NI % 2 == 0) {
            if (rn_) LGD_EPI_LDS(true, false, true); else LGD_EPI_LDS(true, false, false);
          }
        } else if (d.res) {
          if (rn_) LGD_EPI_LDS(false, true, true); else LGD_EPI_LDS(false, true, false);
        } else {
          if (rn_) LGD_EPI_LDS(false, false, true); else LGD_EPI_LDS(false, false, false);
        }
#undef LGD_EPI_LDS
      } else {
        // the 256 x 256 tile has no register-layout epilogue (128 accumulators + its residual prefetch spilled 239
        // registers): the launcher admits it only where the LDS epilogue applies
        if constexpr (!(TWO && MI * NI >= 32))
          gemm_epilogue<MI, NI>(ga, acc, m0_, n0_, wm, wn, lane, batch, split, c_off, r_off, reinterpret_cast<int*>(smem));
      }
    };
    if constexpr (PERSIST) {
      for (int v = blockIdx.x; v < total_tiles; v += gridDim.x) {
        tile_origin(v, m0, n0);
        k_loop();
        // epilogue of this output tile; the next tile's first D K tiles are already on their way into the ring
        epilogue(m0, n0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // first fragments of the next output tile (its first K tile landed before the last barrier of the K loop)
        if constexpr (!(ABL & 2)) {
          lds_read_frags<0, FSTR>(af0, a_b0 + so_cur, seqA{});
          lds_read_frags<0, FSTR>(bf0, b_b0 + so_cur, seqB{});
        }
      }
      // the reads of the (non-existent) K tile behind the last one and the zero-line DMAs behind the last real tiles
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      k_loop();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      epilogue(m0, n0);
    }
  } else {
    for (int v = blockIdx.x; v < total_tiles; v += gridDim.x) {
      int m0, n0;
      tile_origin(v, m0, n0);
      if constexpr (!(TWO && MI * NI >= 32))
        gemm_epilogue<MI, NI>(ga, acc, m0, n0, wm, wn, lane, batch, split, c_off, r_off, reinterpret_cast<int*>(smem));
    }
  }
}

// Main-loop variant 4 (round 6): PHASE-SPLIT 256-row tiles (256 x 256, 256 x 320), eight waves as 2 (M) x 4 (N).
//   The lock-step loops above let both waves of a SIMD want the matrix pipe, the LDS and the barrier at the same
//   moments (PMC, 256 x 128 family: MFMA pipe 29-47 % busy, waits 31-42 %).  Here the K tile is cut into FOUR phases
//   (one quadrant of the wave's 128 x (64|80) output each, both k-halves), every phase is a LOAD interval (fragment
//   ds_reads of that quadrant + one quarter of a future K tile's LDS-DMA) and an MFMA interval (16-24 back-to-back
//   MFMAs under s_setprio 1), separated by bare s_barriers, and the two wave groups (waves 0-3 = rows 0-127 = one wave
//   per SIMD, waves 4-7 = rows 128-255 = the other wave of each SIMD) run ONE interval apart: while one group
//   multiplies, the other reads and issues.  One fragment register set suffices (a wave never reads while it
//   multiplies), the LDS latency hides behind the partner's MFMAs instead of a software prefetch, and the matrix pipe
//   sees one MFMA stream at a time.
//   LDS: two stages, each [A_q0 | A_q1 | B_q0 | B_q1]: A_q(h) = the m-tiles h*MI/2.. of BOTH groups, B_q0 / B_q1 = the
//   first NI0 / last NI1 n-tiles of all four wave columns.  A quarter is read in exactly ONE phase (p0: A_q0 + B_q0,
//   p1: B_q1, p2: A_q1; B_q0 stays in registers for p3), so it can be re-filled two phases later — the DMA stream never
//   pauses although there are only two stages (the round-4 two-stage ring issued during half of each K tile):
//       (kt, p0) issues B_q1 of tile kt+1      (kt, p1) A_q1 of tile kt+1
//       (kt, p2) issues B_q0 of tile kt+2      (kt, p3) A_q0 of tile kt+2
//   i.e. every quarter flies for 4-5 phases (> one K tile), and `s_waitcnt vmcnt(T)` (T = DMA instructions per wave per
//   K tile: "everything older than the last four quarters has landed") at the end of the load intervals of p3 / p0 /
//   p1 is the only wait on the queue — it never drains.  Hazards (G0 = the early group, G1 one interval later):
//     RAW  a quarter is read one phase after the vmcnt that retires it: every wave's wait sits in front of the barrier
//          that closes its load interval, G0's next load interval and G1's lie behind G1's / G0's NEXT barrier;
//     WAR  G1's reads of phase p retire (lgkmcnt(0)) at the start of its MFMA interval = global interval 2p+2; the
//          earliest DMA into that quarter is G0's load interval of phase p+2 = global interval 2p+4.
//   DMA = buffer_load_dwordx4 ... lds through a raw buffer descriptor: per-lane voffset is ONE register per operand
//   (row order inside a quarter is chosen so that the rows of consecutive instructions differ by a wave-uniform
//   stride), everything else — K position, tap, row group — is the SCALAR soffset; lanes of rows past M / N, of padded
//   convolution taps and whole tiles past K get an out-of-range voffset / a zero-length descriptor and the hardware
//   writes zeros (no zero line, no per-instruction 64-bit address arithmetic: the two-stage rings spent as many scalar as
//   vector issue slots on it).  Plain single-source contractions and chunk-major 3x3 stride-1 convolutions.
// ABL (tools only, -DLGD_GEMM_ABLATION; results wrong by design): 1 = no DMA in the loop, 2 = no fragment reads, 4 = no MFMA,
// 16 = no epilogue, 32 = no s_setprio, 64 = no stagger (both groups in the same interval).
template <int MI, int NI, bool CM, int ABL = 0>
__global__ __launch_bounds__(512, 1) void gemm_phase_kernel(const GemmArgs ga) {
  constexpr int WM = 2, WN = 4;
  static_assert(MI == 8, "128 rows per wave group");
  constexpr int MH = MI / 2;                                  // m-tiles per A quarter and group
  constexpr int NI0 = (NI + 1) / 2, NI1 = NI - NI0;           // n-tiles of a wave in B_q0 / B_q1
  constexpr int BM = WM * 16 * MI, BN = WN * 16 * NI;
  constexpr int AQ_B = 2 * 16 * MH * 128;                     // bytes of an A quarter (both groups)
  constexpr int BQ0_B = 64 * NI0 * 128, BQ1_B = 64 * NI1 * 128;
  constexpr int OFF_A1 = AQ_B, OFF_B0 = 2 * AQ_B, OFF_B1 = 2 * AQ_B + BQ0_B;
  constexpr int STAGE_B = 2 * AQ_B + BQ0_B + BQ1_B;           // = (BM + BN) * 128
  static_assert(STAGE_B == (BM + BN) * BK * 2 && 2 * STAGE_B <= 160 * 1024, "LDS");
  constexpr int NA = 2;                                       // DMA instructions per wave per A quarter (one per group)
  constexpr int T_DMA = 2 * NA + NI;                          // per wave per K tile
  constexpr unsigned OOB = 0x80000000u;                       // >= num_records of every descriptor below
  constexpr unsigned NREC = 0x80000000u;

  extern __shared__ __attribute__((aligned(1024))) half_t smem[];
  unsigned char* lds = reinterpret_cast<unsigned char*>(smem);

  const LgdGemmDesc& d = ga.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wn = wid & 3;                     // wave group (M half) / wave column

  const int n_tiles_n = (d.N + BN - 1) / BN;
  const int total_tiles = ((d.M + BM - 1) / BM) * n_tiles_n;
  int m0, n0;
  {
    const int q = total_tiles >> 3, r = total_tiles & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    int tile_m, tile_n;
    if (ga.group_m > 1) {
      const int per_group = ga.group_m * n_tiles_n;
      const int g = bid / per_group, in = bid - g * per_group;
      const int n_tiles_m = total_tiles / n_tiles_n;
      int rows = n_tiles_m - g * ga.group_m;
      if (rows > ga.group_m) rows = ga.group_m;
      tile_n = in / rows;
      tile_m = g * ga.group_m + (in - tile_n * rows);
    } else {
      tile_m = bid / n_tiles_n;
      tile_n = bid - tile_m * n_tiles_n;
    }
    // integer divisions run on the vector ALU: pin the (uniform) results to SGPRs, or under register pressure everything
    // derived from them — base pointers, descriptors — stays in VGPRs and every DMA becomes a waterfall loop
    m0 = __builtin_amdgcn_readfirstlane(tile_m * BM);
    n0 = __builtin_amdgcn_readfirstlane(tile_n * BN);
  }
  // one matrix per launch (the launcher sees to it): blockIdx.z is the split, no batch offsets — their 64-bit products
  // were computed on the vector ALU and dragged the base pointers, hence every descriptor, into VGPRs
  const int split = blockIdx.z;
  constexpr int batch = 0;
  constexpr long a_off = 0, w_off = 0, c_off = 0, r_off = 0;

  const int k_beg = split * ga.k_per_split;
  int k_end = k_beg + ga.k_per_split;
  if (k_end > d.K) k_end = d.K;
  const int nk = __builtin_amdgcn_readfirstlane((k_end - k_beg) / BK);
  const int cin = ga.cin;

  // ---- DMA side.  Lane -> (row in the 8-row group, physical 16-B slot); the logical K segment it fetches carries the
  // source-side swizzle of gemm_dma_kernel (the swizzle term of quarter row (8j + w) * 8 + lrow does not depend on j).
  const int lrow = lane >> 3;
  const int kseg = (lane & 7) ^ ((((wid & 1) << 2) + (lrow >> 1)) & 7);
  const unsigned lda2 = (unsigned)d.lda0 * 2u, ldw2 = (unsigned)d.ldw * 2u;
  // A quarter rows are [group][m-tile][16]: instruction j of wave w fills group j, rows h*64 + 8w + lrow of its 128
  const int arow = wid * 8 + lrow;                            // tile row of (j = 0, h = 0)
  // B quarter rows are [n-tile][wave column][16]: instruction j fills n-tile j, wave column w >> 1, rows 8 (w & 1) + lrow
  const int bcol = (wid >> 1) * 16 * NI + (wid & 1) * 8 + lrow;   // tile column of (quarter 0, j = 0)
  // convolution: the centre pixel of output row m is input pixel m (stride 1, same size), the tap is a uniform offset;
  // the descriptor's base sits (win + 1) pixels in front of the map so that every tap offset is >= 0
  const half_t* a_base = reinterpret_cast<const half_t*>(d.a0) + a_off - (CM ? (long)(d.win + 1) * d.lda0 : 0L);
  const half_t* w_base = reinterpret_cast<const half_t*>(d.w) + w_off;
  const unsigned voffA = (unsigned)(m0 + arow) * lda2 + kseg * 16;
  const unsigned voffB = (unsigned)(n0 + bcol) * ldw2 + kseg * 16;
  // validity bits: A (h, j, tap) -> amask[h] bit 9j + tap (plain: tap 0); B n-tile i of the wave column -> bit i
  unsigned amask[2] = {0u, 0u};
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int m = m0 + j * 128 + h * 16 * MH + arow;
      if (m >= d.M) continue;
      if (CM) {
        const int ox = m % d.wout, oy = (m / d.wout) % d.hout;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
          if (iy >= 0 && ix >= 0 && iy < d.hin && ix < d.win) amask[h] |= 1u << (9 * j + t);
        }
      } else {
        amask[h] |= 1u << (9 * j);
      }
    }
  unsigned bmask = 0;
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (n0 + bcol + i * 16 < d.N) bmask |= 1u << i;

  // K position of a tile of the split's range: byte offsets into A and W, and the convolution tap
  const int t0 = k_beg / BK;
  auto k_pos = [&](int kt, unsigned& sa, unsigned& sw, int& tap) {
    if (CM) {
      const int t = t0 + kt, ch = __builtin_amdgcn_readfirstlane(t / 9);
      tap = t - ch * 9;
      const int ky = tap / 3, kx = tap - ky * 3;
      sa = (unsigned)(ky * d.win + kx) * lda2 + (unsigned)ch * (BK * 2);
      sw = (unsigned)(tap * cin + ch * BK) * 2u;
    } else {
      tap = 0;
      sa = sw = (unsigned)(k_beg + kt * BK) * 2u;
    }
  };
  // one A quarter (h) / B quarter (q) of the tile at (sa | sw, tap), `live` = the tile exists, into stage offset so
  auto dma_a = [&](int h, unsigned so, bool live, unsigned sa, int tap, bool inloop = true) {
    if ((ABL & 1) && inloop) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a_base), 0, live ? NREC : 0u, 0x00020000);
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const unsigned vo = ((amask[h] >> (9 * j + tap)) & 1u) ? voffA : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + so + (h ? OFF_A1 : 0) + (j * 8 + wid) * 1024), 16, vo,
                                               sa + (unsigned)(j * 128 + h * 16 * MH) * lda2, 0, 0);
    }
  };
  auto dma_b = [&](int q, unsigned so, bool live, unsigned sw, bool inloop = true) {
    if ((ABL & 1) && inloop) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(w_base), 0, live ? NREC : 0u, 0x00020000);
#pragma unroll
    for (int j = 0; j < (q ? NI1 : NI0); ++j) {
      const int i = (q ? NI0 : 0) + j;
      const unsigned vo = ((bmask >> i) & 1u) ? voffB : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + so + (q ? OFF_B1 : OFF_B0) + (j * 8 + wid) * 1024), 16, vo,
                                               sw + (unsigned)(i * 16) * ldw2, 0, 0);
    }
  };

  // ---- MFMA side: fragment addresses (bytes, stage 0) of this lane, k-half 0 / 1
  const int frow = lane & 15, fg = lane >> 4, fsw = (frow >> 1) & 7;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned a_ad0 = lds0 + (grp * 16 * MH + frow) * 128 + ((0 + fg) ^ fsw) * 16;
  unsigned a_ad1 = lds0 + (grp * 16 * MH + frow) * 128 + ((4 + fg) ^ fsw) * 16;
  unsigned b_ad0 = lds0 + OFF_B0 + (wn * 16 + frow) * 128 + ((0 + fg) ^ fsw) * 16;
  unsigned b_ad1 = lds0 + OFF_B0 + (wn * 16 + frow) * 128 + ((4 + fg) ^ fsw) * 16;
  using seqA = std::make_integer_sequence<int, MH>;
  using seqB0 = std::make_integer_sequence<int, NI0>;
  using seqB1 = std::make_integer_sequence<int, NI1>;

  f32x4 acc[NI][MI];
  half8_t af[2][MH], bf0[2][NI0], bf1[2][NI1];
  if constexpr (ABL & 2) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < MH; ++i) af[kk][i] = (half8_t){1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
      for (int i = 0; i < NI0; ++i) bf0[kk][i] = (half8_t){1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
      for (int i = 0; i < NI1; ++i) bf1[kk][i] = (half8_t){1, 1, 1, 1, 1, 1, 1, 1};
    }
  }

  // MFMA interval of a phase: k-half outer, so that the two MFMAs of one accumulator are NI_*MH issues apart
#define LGD_PH_MMA(BF, NB, NOFF, MOFF)                                                                              \
  {                                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    if constexpr (!(ABL & 32)) __builtin_amdgcn_s_setprio(1);                                                       \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                                \
      _Pragma("unroll") for (int ni = 0; ni < NB; ++ni)                                                             \
        _Pragma("unroll") for (int mi = 0; mi < MH; ++mi)                                                           \
          if constexpr (ABL & 4) asm volatile("" : "+v"(acc[NOFF + ni][MOFF + mi]) : "v"(BF[kk][ni]), "v"(af[kk][mi]));                \
          else acc[NOFF + ni][MOFF + mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(BF[kk][ni], af[kk][mi], acc[NOFF + ni][MOFF + mi], 0, 0, 0); \
    if constexpr (!(ABL & 32)) __builtin_amdgcn_s_setprio(0);                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
#define LGD_PH_BAR()                        \
  {                                         \
    __builtin_amdgcn_sched_barrier(0);      \
    __builtin_amdgcn_s_barrier();           \
    __builtin_amdgcn_sched_barrier(0);      \
  }

  if (nk > 0) {
    // ---- prologue: tile 0 whole, the first two quarters of tile 1
    {
      unsigned sa, sw; int tap;
      k_pos(0, sa, sw, tap);
      dma_b(0, 0, true, sw, false);
      dma_a(0, 0, true, sa, tap, false);
      dma_b(1, 0, true, sw, false);
      dma_a(1, 0, true, sa, tap, false);
      k_pos(1, sa, sw, tap);
      dma_b(0, STAGE_B, nk > 1, sw, false);
      dma_a(0, STAGE_B, nk > 1, sa, tap, false);
    }
    // accumulators are born here, behind the set-up and the prologue's issue (160 registers that were live across both
    // made the 256 x 320 kernel spill its set-up values)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T_DMA) : "memory");   // B_q0, A_q0 of tile 0 landed (own share)
    LGD_PH_BAR();
    if (grp == 1 && !(ABL & 64)) LGD_PH_BAR();                     // the late group starts one interval behind
    unsigned so_cur = 0, so_nxt = STAGE_B;
    int sdelta = STAGE_B;
    for (int kt = 0; kt < nk; ++kt) {
      unsigned sa1, sw1, sa2, sw2; int tap1, tap2;
      k_pos(kt + 1, sa1, sw1, tap1);
      k_pos(kt + 2, sa2, sw2, tap2);
      const bool live1 = kt + 1 < nk, live2 = kt + 2 < nk;
      // ---- phase 0: A_q0 x B_q0
      if constexpr (!(ABL & 2)) lds_read_frags<0, 64 * 128>(bf0[0], b_ad0, seqB0{});
      if constexpr (!(ABL & 2)) lds_read_frags<0, 64 * 128>(bf0[1], b_ad1, seqB0{});
      if constexpr (!(ABL & 2)) lds_read_frags<0, 16 * 128>(af[0], a_ad0, seqA{});
      if constexpr (!(ABL & 2)) lds_read_frags<0, 16 * 128>(af[1], a_ad1, seqA{});
      dma_b(1, so_nxt, live1, sw1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T_DMA) : "memory");
      LGD_PH_BAR();
      LGD_PH_MMA(bf0, NI0, 0, 0);
      LGD_PH_BAR();
      // ---- phase 1: A_q0 x B_q1
      if constexpr (!(ABL & 2)) lds_read_frags<BQ0_B, 64 * 128>(bf1[0], b_ad0, seqB1{});
      if constexpr (!(ABL & 2)) lds_read_frags<BQ0_B, 64 * 128>(bf1[1], b_ad1, seqB1{});
      dma_a(1, so_nxt, live1, sa1, tap1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T_DMA) : "memory");
      LGD_PH_BAR();
      LGD_PH_MMA(bf1, NI1, NI0, 0);
      LGD_PH_BAR();
      // ---- phase 2: A_q1 x B_q1
      if constexpr (!(ABL & 2)) lds_read_frags<OFF_A1, 16 * 128>(af[0], a_ad0, seqA{});
      if constexpr (!(ABL & 2)) lds_read_frags<OFF_A1, 16 * 128>(af[1], a_ad1, seqA{});
      dma_b(0, so_cur, live2, sw2);
      LGD_PH_BAR();
      LGD_PH_MMA(bf1, NI1, NI0, MH);
      LGD_PH_BAR();
      // ---- phase 3: A_q1 x B_q0 (no reads)
      dma_a(0, so_cur, live2, sa2, tap2);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T_DMA) : "memory");
      LGD_PH_BAR();
      LGD_PH_MMA(bf0, NI0, 0, MH);
      LGD_PH_BAR();
      // next K tile: the other stage
      a_ad0 += sdelta; a_ad1 += sdelta; b_ad0 += sdelta; b_ad1 += sdelta;
      sdelta = -sdelta;
      const unsigned t = so_cur; so_cur = so_nxt; so_nxt = t;
    }
    if (grp == 0 && !(ABL & 64)) LGD_PH_BAR();
    // zero-filling DMAs of the tiles past K are still writing LDS: drain before the epilogue reuses it
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    LGD_PH_BAR();
  }
  else {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#undef LGD_PH_MMA
#undef LGD_PH_BAR

  if constexpr (ABL & 16) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(acc[ni][mi]));
    return;
  }
  if (d.splits > 1) {
    // fp32 partials of this split, combined by splitk_reduce_kernel
    const int m_l = lane & 15, n_l = (lane >> 4) * 4;
    float* ws = d.ws + ((long)batch * d.splits + split) * (long)d.M * d.N;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + grp * 16 * MI + mi * 16 + m_l;
      if (m >= d.M) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * 16 * NI + ni * 16 + n_l;
        if (n >= d.N) continue;
        const f32x4 v = acc[ni][mi];
        *reinterpret_cast<float4*>(ws + (long)m * d.N + n) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    return;
  }
  // one split: fp16 rows through LDS (the launcher admits nothing else); both stages are free
#define LGD_EPI_LDS(G, RS, RNORM) \
  gemm_epilogue_lds<MI, NI, WM, WN, (MI * NI > 32 ? STAGE_B : 2 * STAGE_B), G, true, RS, RNORM>(ga, acc, m0, n0, grp, wn, lane, tid, c_off, r_off, lds)
  const bool rn_ = d.epi & LGD_EPI_ROWNORM;
  if (d.epi & LGD_EPI_GEGLU) {
    if constexpr (NI % 2 == 0) {
      if (rn_) LGD_EPI_LDS(true, false, true); else LGD_EPI_LDS(true, false, false);
    }
  } else if (d.res) {
    if (rn_) LGD_EPI_LDS(false, true, true); else LGD_EPI_LDS(false, true, false);
  } else {
    if (rn_) LGD_EPI_LDS(false, false, true); else LGD_EPI_LDS(false, false, false);
  }
#undef LGD_EPI_LDS
}

// Sums the split-K partials and applies the epilogue. One thread per 4 output channels.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs ga) {
  const LgdGemmDesc& d = ga.d;
  const bool geglu = d.epi & LGD_EPI_GEGLU;
  const int n_out_cols = geglu ? d.N / 2 : d.N;
  const long groups_per_row = n_out_cols / 4;
  const long total = (long)d.M * groups_per_row;
  const int batch = blockIdx.z;
  const int b_o = batch / d.nb_i, b_i = batch - b_o * d.nb_i;
  const long c_off = b_o * d.c_bs_o + b_i * d.c_bs_i;
  const long r_off = b_o * d.r_bs_o + b_i * d.r_bs_i;
  const float* ws = d.ws + (long)batch * d.splits * (long)d.M * d.N;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    int m = (int)(idx / groups_per_row);
    int gcol = (int)(idx - (long)m * groups_per_row) * 4;  // output column
    int n_in = geglu ? (gcol / 16) * 32 + (gcol % 16) : gcol;
    f32x4 v = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < d.splits; ++s) {
      const float* p = ws + (long)s * d.M * d.N + (long)m * d.N + n_in;
      float4 t = *reinterpret_cast<const float4*>(p);
      v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      if (geglu) {
        float4 u = *reinterpret_cast<const float4*>(p + 16);
        g[0] += u.x; g[1] += u.y; g[2] += u.z; g[3] += u.w;
      }
    }
    const float4 bv = ld_bias_sum4(d, n_in);
    v = rownorm4(d, m, n_in, v);
    if (geglu) {
      float4 bg = make_float4(0.f, 0.f, 0.f, 0.f);
      if (d.bias) bg = ld_bias4(d.bias, n_in + 16);
      epilogue_store4<true>(d, c_off, r_off, m, gcol, v, rownorm4(d, m, n_in + 16, g), bv, bg);
    } else {
      epilogue_store4<false>(d, c_off, r_off, m, gcol, v, g, bv, bv);
    }
  }
}

template <int MI, int NI, int WM = 2>
int launch_gemm(const GemmArgs& ga, hipStream_t st, bool dma) {
  constexpr int BM = WM * 16 * MI, BN = 32 * NI;
  const LgdGemmDesc& d = ga.d;
  long tiles = (long)((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
  dim3 grid((unsigned)tiles, 1, (unsigned)(d.nb_o * d.nb_i * d.splits));
  if constexpr (WM == 2) {
    if (dma) hipLaunchKernelGGL((gemm_dma_kernel<MI, NI, 2>), grid, dim3(256), 0, st, ga);
    else hipLaunchKernelGGL((gemm_kernel<MI, NI>), grid, dim3(256), 0, st, ga);
  } else {
    if (!dma) return LGD_ERR_ARG;  // the 8-wave tiles exist only with the LDS-DMA main loop
    hipLaunchKernelGGL((gemm_dma_kernel<MI, NI, WM>), grid, dim3(128 * WM), 0, st, ga);
  }
  return lgd_check_launch();
}

template <int MI, int NI, int WM, int WN, int NS, bool CM>
int launch_gemm_pipe_cm(const GemmArgs& ga, hipStream_t st) {
  constexpr int BM = WM * 16 * MI, BN = WN * 16 * NI;
  constexpr int SMEM = NS * (BM + BN) * BK * 2;
  const LgdGemmDesc& d = ga.d;
  // function-local statics with initialisers: C++11 guarantees one thread runs the initialiser while the others wait
  // (several lane threads launch GEMMs concurrently, lgd_amd/lanes.py)
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_kernel<MI, NI, WM, WN, NS, CM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    return true;
  }();
  (void)attr_set;
  long tiles = (long)((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
  // persistent launch: as many workgroups as the chip holds at once (a multiple of 8, one share per XCD); each walks
  // tiles b, b + grid, ... with its DMA ring running across tile boundaries.  LGD_GEMM_PERSIST=0 = one tile per
  // workgroup (A/B timing).  Split-K / batched launches already spread over blockIdx.z and keep one tile each.
  static const long resident = [] {
    // no persistent form of the chunk-major (3x3 convolution) kernels, nor of the 256 x 160 tile: with 80 accumulator
    // registers the tile-crossing state does not fit 256 VGPRs (58 spilled even after the trims above); measured gain
    // of persistence on that tile's K <= 640 shapes was +4..9 % of ~2.5 % of the step
    if constexpr (CM || (MI == 4 && NI == 5) || NS == 2) return 0L;
    else {
      const char* e = getenv("LGD_GEMM_PERSIST");
      int per_cu = 0, cus = 0, dev = 0;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_kernel<MI, NI, WM, WN, NS, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(
          &per_cu, reinterpret_cast<const void*>(&gemm_pipe_kernel<MI, NI, WM, WN, NS, false, true>), 64 * WM * WN, SMEM);
      return (e && e[0] == '0') ? 0L : (long)(cus / 8) * 8 * (per_cu > 0 ? per_cu : 1);
    }
  }();
  // measured (tools/gemm_ab.py, LGD_GEMM_PERSIST=0 vs 1): +4..9 % where K <= 640 (5-10 K tiles per output tile: the
  // pipeline fill is a visible share of a tile), neutral to -8 % from K = 1280 up — enabled for short K walks only
  const bool persist = resident > 0 && d.nb_o * d.nb_i * d.splits == 1 && tiles > resident && d.K <= 10 * BK && d.taps == 1;
  dim3 grid((unsigned)(persist ? resident : tiles), 1, (unsigned)(d.nb_o * d.nb_i * d.splits));
#ifdef LGD_GEMM_ABLATION
  static const int abl = [] { const char* e = getenv("LGD_GEMM_ABL"); return e ? atoi(e) : 0; }();
  if (abl) {
    auto go = [&](auto kern) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
      hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), SMEM, st, ga);
    };
    switch (abl) {
      case 1: go(&gemm_pipe_kernel<MI, NI, WM, WN, NS, CM, false, 1>); break;
      case 2: go(&gemm_pipe_kernel<MI, NI, WM, WN, NS, CM, false, 2>); break;
      case 3: go(&gemm_pipe_kernel<MI, NI, WM, WN, NS, CM, false, 3>); break;
      case 4: go(&gemm_pipe_kernel<MI, NI, WM, WN, NS, CM, false, 4>); break;
      case 6: go(&gemm_pipe_kernel<MI, NI, WM, WN, NS, CM, false, 6>); break;
      case 8: go(&gemm_pipe_kernel<MI, NI, WM, WN, NS, CM, false, 8>); break;
      case 11: go(&gemm_pipe_kernel<MI, NI, WM, WN, NS, CM, false, 11>); break;
      case 16: go(&gemm_pipe_kernel<MI, NI, WM, WN, NS, CM, false, 16>); break;
      case 17: go(&gemm_pipe_kernel<MI, NI, WM, WN, NS, CM, false, 17>); break;
      case 20: go(&gemm_pipe_kernel<MI, NI, WM, WN, NS, CM, false, 20>); break;
      default: break;
    }
    return lgd_check_launch();
  }
#endif
  if constexpr (!CM && !(MI == 4 && NI == 5) && NS != 2) {     // see `resident`: no such instantiations
    if (persist) {
      hipLaunchKernelGGL((gemm_pipe_kernel<MI, NI, WM, WN, NS, false, true>), grid, dim3(64 * WM * WN), SMEM, st, ga);
      return lgd_check_launch();
    }
  }
  hipLaunchKernelGGL((gemm_pipe_kernel<MI, NI, WM, WN, NS, CM>), grid, dim3(64 * WM * WN), SMEM, st, ga);
  return lgd_check_launch();
}

// chunk-major K order for plain 3x3 convolutions (stride 1, no upsampling fold, one source); LGD_GEMM_NO_CM=1
// in the environment keeps the weight layout's tap-major order (A/B timing of the two walks).
template <int MI, int NI, int WM, int WN, int NS>
int launch_gemm_pipe(const GemmArgs& ga, hipStream_t st) {
  const LgdGemmDesc& d = ga.d;
  static const int no_cm = [] { const char* e = getenv("LGD_GEMM_NO_CM"); return (e && e[0] == '1') ? 1 : 0; }();
  if constexpr (NS == 2) {          // the two-stage 256 x 256 tile: plain single-source contractions only
    if (d.taps != 1 || d.c1 > 0) return LGD_ERR_ARG;
    if constexpr (MI * NI >= 32) {  // 256 x 256: the LDS epilogue only (one split, fp16 rows of whole 16-byte pieces)
      const long c_off_max = (long)(d.nb_o - 1) * d.c_bs_o + (long)(d.nb_i - 1) * d.c_bs_i;
      if (d.splits != 1 || (d.epi & LGD_EPI_OUT_F32) || (d.N & 7) || (d.ldc & 7) || (c_off_max & 7) || (d.c_bs_o & 7) ||
          (d.c_bs_i & 7) || (reinterpret_cast<uintptr_t>(d.c) & 15) || d.K < BK ||
          (d.res && (d.epi & (LGD_EPI_RES_F32 | LGD_EPI_GEGLU))))
        return LGD_ERR_ARG;
    }
    return launch_gemm_pipe_cm<MI, NI, WM, WN, NS, false>(ga, st);
  } else {
  const bool cm = d.taps == 9 && d.stride == 1 && d.ups == 0 && d.c1 == 0 && !no_cm;
  return cm ? launch_gemm_pipe_cm<MI, NI, WM, WN, NS, true>(ga, st) : launch_gemm_pipe_cm<MI, NI, WM, WN, NS, false>(ga, st);
  }
}

// Phase-split 256-row tiles (gemm_phase_kernel): plain single-source contractions and 3x3 stride-1 same-size
// convolutions; one split leaves through the LDS epilogue (fp16 rows of whole 16-byte pieces), several splits write fp32
// partials for splitk_reduce_kernel.  Everything else is the caller's business (LGD_ERR_ARG; ops._table_tile_applies).
template <int MI, int NI>
int launch_gemm_phase(const GemmArgs& ga, hipStream_t st) {
  constexpr int BM = 32 * MI, BN = 64 * NI;
  constexpr int SMEM = 2 * (BM + BN) * BK * 2;
  const LgdGemmDesc& d = ga.d;
  const bool conv = d.taps == 9;
  if (d.c1 > 0 || d.K < BK || d.nb_o * d.nb_i != 1) return LGD_ERR_ARG;
  if (conv && (d.stride != 1 || d.ups != 0 || d.hin != d.hout || d.win != d.wout)) return LGD_ERR_ARG;
  // 32-bit byte offsets inside one buffer descriptor (2 GB each)
  if (((long)d.M + 2L * d.win + 2) * d.lda0 * 2 + 2L * d.K >= (1L << 31) || (long)d.N * d.ldw * 2 >= (1L << 31)) return LGD_ERR_ARG;
  if (d.splits == 1) {
    const long c_off_max = (long)(d.nb_o - 1) * d.c_bs_o + (long)(d.nb_i - 1) * d.c_bs_i;
    if ((d.epi & LGD_EPI_OUT_F32) || (d.N & 7) || (d.ldc & 7) || (c_off_max & 7) || (d.c_bs_o & 7) || (d.c_bs_i & 7) ||
        (reinterpret_cast<uintptr_t>(d.c) & 15) || ((d.epi & LGD_EPI_GEGLU) && (NI & 1)) ||
        (d.res && (d.epi & (LGD_EPI_RES_F32 | LGD_EPI_GEGLU))))
      return LGD_ERR_ARG;
  } else if (d.cnt) {
    return LGD_ERR_ARG;
  }
  long tiles = (long)((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
  dim3 grid((unsigned)tiles, 1, (unsigned)(d.nb_o * d.nb_i * d.splits));
#ifdef LGD_GEMM_ABLATION
  static const int abl = [] { const char* e = getenv("LGD_GEMM_ABL"); return e ? atoi(e) : 0; }();
  if (abl) {
    auto go = [&](auto kern) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
      hipLaunchKernelGGL(kern, grid, dim3(512), SMEM, st, ga);
    };
#define LGD_PH_ABL(A) case A: if (conv) go(&gemm_phase_kernel<MI, NI, true, A>); else go(&gemm_phase_kernel<MI, NI, false, A>); break;
    switch (abl) {
      LGD_PH_ABL(1) LGD_PH_ABL(2) LGD_PH_ABL(3) LGD_PH_ABL(4) LGD_PH_ABL(6) LGD_PH_ABL(16) LGD_PH_ABL(17) LGD_PH_ABL(20) LGD_PH_ABL(32) LGD_PH_ABL(64) LGD_PH_ABL(96)
      default: break;
    }
#undef LGD_PH_ABL
    return lgd_check_launch();
  }
#endif
  if (conv) {
    static const bool attr_set = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_phase_kernel<MI, NI, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
      return true;
    }();
    (void)attr_set;
    hipLaunchKernelGGL((gemm_phase_kernel<MI, NI, true>), grid, dim3(512), SMEM, st, ga);
  } else {
    static const bool attr_set = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_phase_kernel<MI, NI, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
      return true;
    }();
    (void)attr_set;
    hipLaunchKernelGGL((gemm_phase_kernel<MI, NI, false>), grid, dim3(512), SMEM, st, ga);
  }
  return lgd_check_launch();
}

}  // namespace

extern "C" int lgd_abi_version(void) { return LGD_ABI_VERSION; }

extern "C" int lgd_gemm_f16(const LgdGemmDesc* desc, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (!desc) return LGD_ERR_ARG;
  GemmArgs ga;
  ga.d = *desc;
  LgdGemmDesc& d = ga.d;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d.M <= 0 || d.N <= 0 || d.K <= 0) return LGD_ERR_ARG;
  if (d.taps != 1 && d.taps != 9) return LGD_ERR_ARG;
  ga.cin = d.c0 + d.c1;
  if (d.K != d.taps * ga.cin) return LGD_ERR_ARG;
  if ((d.c0 % 8) || (d.c1 % 8) || (d.N % 4) || (d.ldw % 8) || (d.lda0 % 8) || (d.lda1 % 8) ||
      (d.ldc % 4))
    return LGD_ERR_ARG;
  if (d.c1 > 0 && !d.a1) return LGD_ERR_ARG;
  if (d.res && (d.ldr % 4)) return LGD_ERR_ARG;
  if (d.nb_o < 1) d.nb_o = 1;
  if (d.nb_i < 1) d.nb_i = 1;
  if (d.splits < 1) d.splits = 1;
  const bool geglu = d.epi & LGD_EPI_GEGLU;
  if (geglu && (d.N % 32)) return LGD_ERR_ARG;
  if (d.splits > 1 && !d.ws) return LGD_ERR_ARG;
  // folded LayerNorm: statistics are indexed by the row of ONE matrix, the columns by the stored weight row
  if ((d.epi & LGD_EPI_ROWNORM) && (!d.rowstat || !d.colsum || d.taps != 1 || d.nb_o * d.nb_i != 1 || d.c1 > 0))
    return LGD_ERR_ARG;
  // K range of each split, multiple of BK
#ifdef LGD_GEMM_ABLATION
  { const char* e = getenv("LGD_GEMM_STAGGER"); ga.stagger = e ? atoi(e) : 0; }
#endif
  {
    // grouped tile walk of the pipelined kernels: plain contractions whose weights exceed what an XCD's L2 keeps beside
    // the A panels; the group is as many 256-row A panels as fit ~3 MB (LGD_GEMM_GROUP_M forces a size, 1 = off)
    static const int forced = [] { const char* e = getenv("LGD_GEMM_GROUP_M"); return e ? atoi(e) : 0; }();
    ga.group_m = 1;
    const long w_bytes = (long)d.N * d.K * 2;
    if (d.taps == 1 && d.nb_o * d.nb_i == 1 && w_bytes > (5L << 19)) {
      long g = (3L << 20) / (256L * d.K * 2);
      ga.group_m = forced > 0 ? forced : (int)(g < 1 ? 1 : g > 8 ? 8 : g);
    }
  }
  int ktiles = (d.K + BK - 1) / BK;
  int tps = (ktiles + d.splits - 1) / d.splits;
  ga.k_per_split = tps * BK;
  // drop empty trailing splits
  d.splits = (ktiles + tps - 1) / tps;

  int tile = d.tile;
  if (tile == 0) {
    // heuristic: the largest tile that still yields >= ~2 workgroups per CU-pair
    long batches = (long)d.nb_o * d.nb_i * d.splits;
    auto wgs = [&](int bm, int bn) {
      return batches * ((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn);
    };
    if (d.M <= 32) tile = 21;
    else if (!geglu && d.N % 160 == 0 && wgs(128, 160) >= 384) tile = 22;
    else if (!geglu && d.N % 160 == 0 && wgs(64, 160) >= 256) tile = 23;
    else if (wgs(128, 128) >= 384 && d.N % 128 == 0) tile = 17;
    else if (wgs(64, 128) >= 256 && d.N % 128 == 0) tile = 19;
    else tile = 20;
  }
  // tile codes 33.. = 8-wave three-stage pipelined main loop (K % 64 == 0 only)
  if (tile > 32) {
    if ((d.K % BK) || (ga.cin % BK) || (d.c0 % BK)) return LGD_ERR_ARG;
    int rc;
    switch (tile) {
      case 33: rc = geglu ? LGD_ERR_ARG : launch_gemm_pipe<4, 5, 4, 2, 3>(ga, st); break;  // 256x160, 3 stages
      case 34: rc = launch_gemm_pipe<4, 4, 4, 2, 3>(ga, st); break;                         // 256x128, 3
      case 35: rc = launch_gemm_pipe<4, 2, 4, 2, 4>(ga, st); break;                         // 256x64,  4
      case 37: rc = geglu ? LGD_ERR_ARG : launch_gemm_pipe<2, 5, 4, 2, 4>(ga, st); break;  // 128x160, 4
      case 38: rc = launch_gemm_pipe<2, 4, 4, 2, 4>(ga, st); break;                         // 128x128, 4
      case 39: rc = launch_gemm_pipe<2, 2, 4, 2, 5>(ga, st); break;                         // 128x64,  5
      case 40: rc = geglu ? LGD_ERR_ARG : launch_gemm_pipe<1, 5, 4, 2, 5>(ga, st); break;  // 64x160,  5
      case 41: rc = launch_gemm_pipe<1, 4, 4, 2, 5>(ga, st); break;                         // 64x128,  5
      case 42: rc = launch_gemm_pipe<1, 2, 4, 2, 6>(ga, st); break;                         // 64x64,   6
      case 44: rc = launch_gemm_pipe<4, 8, 4, 2, 2>(ga, st); break;                         // 256x256, 2 stages, eight waves of 64x128
      case 45: rc = launch_gemm_pipe<2, 4, 4, 2, 2>(ga, st); break;                         // 128x128, 2 stages, two workgroups per CU
      case 46: rc = launch_gemm_phase<8, 4>(ga, st); break;                                 // 256x256, phase-split (round 6)
      case 47: rc = geglu ? LGD_ERR_ARG : launch_gemm_phase<8, 5>(ga, st); break;          // 256x320, phase-split
      default: return LGD_ERR_ARG;
    }
    if (rc) return rc;
    if (d.splits > 1 && !d.cnt) {
      int n_out = geglu ? d.N / 2 : d.N;
      long total = (long)d.M * (n_out / 4);
      int blocks = (int)((total + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      dim3 grid(blocks, 1, d.nb_o * d.nb_i);
      hipLaunchKernelGGL(splitk_reduce_kernel, grid, dim3(256), 0, st, ga);
      rc = lgd_check_launch();
    }
    return rc;
  }
  // tile codes 1..7 = register-staged main loop; 16 + code = LDS-DMA main loop (17..23, and the 8-wave 25/26)
  bool dma = tile > 16;
  if (dma) tile -= 16;
  if (dma && (d.K % BK)) dma = false;
  int rc;
  switch (tile) {
    case 1: rc = launch_gemm<4, 4>(ga, st, dma); break;
    case 2: rc = launch_gemm<4, 2>(ga, st, dma); break;
    case 3: rc = launch_gemm<2, 4>(ga, st, dma); break;
    case 4: rc = launch_gemm<2, 2>(ga, st, dma); break;
    case 5: rc = launch_gemm<1, 4>(ga, st, dma); break;
    case 6: rc = geglu ? LGD_ERR_ARG : launch_gemm<4, 5>(ga, st, dma); break;  // 128x160: N = 320 k exactly
    case 7: rc = geglu ? LGD_ERR_ARG : launch_gemm<2, 5>(ga, st, dma); break;  // 64x160
    case 9: rc = geglu ? LGD_ERR_ARG : launch_gemm<4, 10, 4>(ga, st, dma); break;  // 256x320, 8 waves
    case 10: rc = launch_gemm<4, 4, 4>(ga, st, dma); break;                         // 256x128, 8 waves
    default: return LGD_ERR_ARG;
  }
  if (rc) return rc;
  if (d.splits > 1 && !d.cnt) {
    int n_out = geglu ? d.N / 2 : d.N;
    long total = (long)d.M * (n_out / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    dim3 grid(blocks, 1, d.nb_o * d.nb_i);
    hipLaunchKernelGGL(splitk_reduce_kernel, grid, dim3(256), 0, st, ga);
    rc = lgd_check_launch();
  }
  return rc;
}
