// Self-attention forward for 40-wide heads (the SD1.x 64x64 level: S = 4096 queries and keys, 8 heads) — round 4.
// Replaces F.scaled_dot_product_attention at attention_processor.py:355-357 (no probability map requested) for d = 40.
//
// Why another kernel.  At d = 40 the flash forward is bound by INSTRUCTION ISSUE, not by the matrix pipe: per
// 32-query x 64-key tile a wave needs 14 MFMA 32x32x16 (448 matrix-pipe cycles) but the round-3 kernels issued 213
// other instructions beside them (softmax 72, register staging of K / V^T with its masks and waits ~110, fragment
// reads 14, control) — 15 per MFMA gap against the ~5 a gap can hide (MI355X_MICROARCH.md, cycle constants).  This
// kernel removes the other instructions instead of re-ordering them:
//   * ONE wave per SIMD (4-wave workgroup, 256 queries, up to 512 registers per lane) and TWO 32-query blocks per wave:
//     every K / V fragment read from LDS feeds two MFMAs, all per-tile overhead is amortised over 28 MFMAs;
//   * K and V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4), three wave-instructions per wave and tile,
//     no staging registers, no ds_write, no exec masks; V stays in its natural [key][dv] layout and is transposed by
//     the LDS read itself: ds_read_b64_tr_b16 hands lane n of a 16-lane group column n of a 4-row x 16-column block
//     (probed on MI355X: tools/probe_tr.hip), which is exactly the K-contiguous A operand of O^T = V^T P^T;
//   * the running reference and the row sum cost no instruction: the DMA copies a constant 16-byte line behind every
//     key's 40 values — (1, 0, ...) for a valid key, so that K's slot 40 meets -m_ref in Q's slot 40 (the MFMA emits
//     S - m_ref) and V's column 40 makes row dv = 40 of the PV product the row sum; (0, 1, ...) for the padded keys of
//     a ragged last tile, whose slot 41 meets -30000 in Q (they drop out of row max and sum; their V rows are zeros);
//   * software pipeline over key tiles inside the wave: S(t+1) = K(t+1) Q^T next to P(t) = exp2(S(t)), then
//     O += V(t)^T P(t) next to the row max of S(t+1); the reference is raised only on a wave-uniform vote (row max
//     more than 2^8 above it), after PV(t) is complete, so a rescale covers everything accumulated exactly once.
// LDS image: K and V tiles 64 rows x 112 B (40 values, the constant line, one line of zeros: 7 slots of 16 B, an odd
// count -> the four lane groups of a ds_read_b128 fragment read hit 16 distinct slots), two stages each.
#include "common.h"
#include "../../include/lgd_hip.h"
#include "attn_w4.h"
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <type_traits>
#include <utility>

namespace {

typedef short short4_t __attribute__((ext_vector_type(4)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) short4_t* lds_s4_ptr;

// constant 16-byte lines the DMA copies behind a key's 40 values:
//   [0..7]  "valid key":  (1, 0, 0, ...)  -> K: slot 40 = 1 (times Q's -m_ref), V: column 40 = 1 (row sum)
//   [8..15] "padded key": (0, 1, 0, ...)  -> K: slot 41 = 1 (times Q's -30000: the key drops out of max and sum)
//   [16..23] zeros
__device__ __attribute__((aligned(64))) const half_t g_w4_lines[24] = {(half_t)1.f, 0, 0, 0, 0, 0, 0, 0,
                                                                        0, (half_t)1.f, 0, 0, 0, 0, 0, 0,
                                                                        0, 0, 0, 0, 0, 0, 0, 0};

constexpr int D = 40;
constexpr int ROWB = 112;                     // bytes per LDS row of a K or V tile: 7 slots of 16 B (odd -> the four
                                              // lane groups of a ds_read_b128 fragment read hit 16 distinct slots)
constexpr int SEGS = 7;
constexpr int TILE_B = 64 * ROWB;             // 7168
constexpr int T_INSTR = TILE_B / 1024;        // 7 DMA wave-instructions per tile
constexpr int N_PER_WAVE = 4;                 // every wave issues V, V, K, K per key tile (wave 3: one of each is padding), so
                                              // that "vmcnt(4)" means "everything but this iteration's DMAs has landed"
constexpr int K_STAGES = 4;                   // iteration t multiplies K(t+1) (fragments read at the end of t-1), reads the
                                              // fragments of K(t+2) and issues the DMA of K(t+4) into K(t+1)'s stage
constexpr int V_STAGES = 3;                   // iteration t multiplies V(t) and issues V(t+2) into V(t-1)'s stage
constexpr int V0_B = K_STAGES * TILE_B;       // byte offset of V stage 0
constexpr int DUMMY_B = V0_B + V_STAGES * TILE_B;   // 1 KiB nobody reads: target of the padding DMA instructions
constexpr int LDS_B = DUMMY_B + 1024 + 256;   // K stages, V stages, dummy, zero tail
constexpr float PADV = 30000.f;

__device__ __forceinline__ float pair_max(float mx) {
  // max over the two lanes (l, l + 32) of a query; inline asm: see half_pair_max in attn.hip
  float x = mx, y = mx;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  return fmaxf(x, y);
}

// MFMAs are inline asm with ASM-OWNED accumulator registers: the output accumulators O^T live in a[0:63] (query block
// qb, dv block dvb -> a[16 (2 qb + dvb) ..+15]) and the Q fragments in a[64:87] (a[64 + 4 (3 qb + ks) ..+3]) for the
// whole kernel; the score accumulators are ordinary compiler-allocated arch VGPRs (the softmax VALU reads them).
// hipcc puts either all or none of a function's MFMA results into AGPRs and then copies between the files (768
// v_accvgpr moves per two key tiles in the first, builtin-based build of this kernel; operands constrained to "a"
// were copied in front of every asm statement in the second).  The compiler itself uses no AGPR here: the kernel stays
// below 256 arch VGPRs without spilling (tests/test_kernel_resources.py checks both), and the one-time claim below
// makes the registers part of the kernel's allocation.  What hipcc does NOT do for an asm MFMA is the hazard
// bookkeeping: the code keeps >= 16 other MFMAs plus explicit wait states between an MFMA and the first VALU read of
// its result, two wait states between a VALU write of a P fragment and the MFMA that reads it (in the slot schedule
// a fragment is written at least one slot before its MFMA; the unpipelined last tile uses mfma_pv_fresh), and wait
// states between a VALU write of an AGPR and the next MFMA that reads it.
#define LGD_W4_AGPR_CLOBBERS                                                                                          \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17",  \
  "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33",      \
  "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49",      \
  "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65",      \
  "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81",      \
  "a82", "a83", "a84", "a85", "a86", "a87"
constexpr int QF0 = 64;                       // first AGPR of the Q fragments
template <int QI>                             // S = K-fragment x Q fragment QI (= 3 qb + ks), fresh accumulator
__device__ __forceinline__ void mfma_qk_first(f32x16& d, const half8_t& kf) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%2:%3], 0" : "=&v"(d) : "v"(kf), "n"(QF0 + 4 * QI), "n"(QF0 + 4 * QI + 3));
}
template <int QI>
__device__ __forceinline__ void mfma_qk_acc(f32x16& d, const half8_t& kf) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%2:%3], %0" : "+v"(d) : "v"(kf), "n"(QF0 + 4 * QI), "n"(QF0 + 4 * QI + 3));
}
template <int OI>                             // O^T block OI (= 2 qb + dvb) += V^T fragment x P fragment
__device__ __forceinline__ void mfma_pv(const half8_t& vf, const half8_t& pf) {
  asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(vf), "v"(pf), "n"(16 * OI), "n"(16 * OI + 15));
}
// the same behind two wait states: for a P fragment the VALU has just written (a VALU write of a VGPR needs two wait
// states before an MFMA reads it; hipcc inserts them for MFMAs it can see, not for an asm statement)
template <int OI>
__device__ __forceinline__ void mfma_pv_fresh(const half8_t& vf, const half8_t& pf) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(vf), "v"(pf), "n"(16 * OI), "n"(16 * OI + 15));
}
template <int R>
__device__ __forceinline__ void agpr_write(float v) { asm volatile("v_accvgpr_write_b32 a[%1], %0" ::"v"(v), "n"(R)); }
template <int R>
__device__ __forceinline__ float agpr_read() { float v; asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(R)); return v; }
template <int R0, int N>
__device__ __forceinline__ void agpr_scale(float alpha) {
  if constexpr (N > 0) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a[%2]\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a[%2], %0" : "=&v"(t) : "v"(alpha), "n"(R0));
    agpr_scale<R0 + 1, N - 1>(alpha);
  }
}
template <int R0, int N>
__device__ __forceinline__ void agpr_fill(float v) {
  if constexpr (N > 0) { agpr_write<R0>(v); agpr_fill<R0 + 1, N - 1>(v); }
}
// LDS-DMA in asm: hipcc would otherwise wait vmcnt(0) in front of every LDS read it can see while a DMA is pending
__device__ __forceinline__ void dma16(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory");
}

template <class F, int... I>
__device__ __forceinline__ void for_seq(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

// ABL (tools only; results are wrong by design): 1 = no DMA in the loop, 2 = no barrier / DMA wait, 4 = no exponentials,
// 8 = no V fragment reads, 16 = no MFMA, 32 = no row max — the cost of each ingredient by removal.
// PIPE = true: ONE wave per SIMD, the wave software-pipelines S(t+1) against P(t) / PV(t) itself (28-slot schedule below).
// PIPE = false: TWO waves per SIMD (two workgroups per CU, <= 256 registers per lane): a wave runs QK^T, softmax, PV of
// a key tile one after the other and the SIMD's other wave fills the pipe it leaves idle.
template <int ABL, bool PIPE>
__global__ __launch_bounds__(256, PIPE ? 1 : 2) void attn_w4_kernel(const AttnW4Args a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 256 + wid * 64;
  const half_t* Qb = a.q + (long)b * a.q_bs + (long)h * D;
  const half_t* Kb = a.k + (long)b * a.k_bs + (long)h * D;
  const half_t* Vb = a.v + (long)b * a.v_bs + (long)h * D;
  const int n_tiles = (a.Sk + 63) >> 6;
  const int rem_last = a.Sk - (n_tiles - 1) * 64;          // valid keys of the last tile (1..64)
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  // V fragment reads of the second dv block run past a row's 56 halfs into the next row / stage (output rows nobody
  // stores): whatever they find must at least not fault; zeros until a tile lands.
  for (int i = tid; i < LDS_B / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // ---- DMA side.  A tile = 7 wave-instructions of 1 KiB (64 lanes x 16 B, lane-linear in LDS); wave w issues pieces
  // w and w + 4 of the V tile, then pieces w and w + 4 of the K tile (piece 7 does not exist: wave 3 issues a padding
  // instruction from the zero line into the dummy KiB instead, so every wave has the same count in flight).
  // slot n: 0 = V piece w, 1 = V piece w + 4, 2 = K piece w, 3 = K piece w + 4
  const char* src[N_PER_WAVE];
  unsigned inc[N_PER_WAVE];     // bytes per key tile (0 for the constant lines)
  unsigned dst[N_PER_WAVE];     // LDS byte address inside stage 0 (wave-uniform)
  const bool pad_hi = wid == 3;   // slots 1 and 3 are padding
#pragma unroll
  for (int n = 0; n < N_PER_WAVE; ++n) {
    const bool isk = n >= 2;
    const int j = wid + 4 * (n & 1);
    const int x = j * 64 + lane;
    const int r = x / SEGS, sg = x - r * SEGS;
    const half_t* base = isk ? Kb + (long)r * a.ldk : Vb + (long)r * a.ldv;
    const bool pad = (n & 1) && pad_hi;
    if (sg < 5 && !pad) { src[n] = reinterpret_cast<const char*>(base + sg * 8); inc[n] = (unsigned)(128 * (isk ? a.ldk : a.ldv)); }
    else { src[n] = reinterpret_cast<const char*>(g_w4_lines + ((sg == 5 && !pad) ? 0 : 16)); inc[n] = 0; }
    dst[n] = lds0 + (pad ? DUMMY_B : (isk ? 0 : V0_B) + j * 1024);
  }
  unsigned smul[N_PER_WAVE];    // bytes per ring stage (0 for a padding slot: it always writes the dummy KiB)
#pragma unroll
  for (int n = 0; n < N_PER_WAVE; ++n) smul[n] = ((n & 1) && pad_hi) ? 0u : (unsigned)TILE_B;
  // ring positions of the next V / K tile to be issued
  int tv_next = 0, tk_next = 0;
  unsigned vs_next = 0, ks_next = 0;
  const int t_ragged = rem_last < 64 ? n_tiles - 1 : n_tiles;          // the tile with padded keys, if any
  // One DMA instruction (slot n: 0, 1 = V pieces, 2, 3 = K pieces) of the next V / K tile.  STEADY = the tile is an
  // interior one (exists, no padded keys): no condition at all — with ONE wave per SIMD every taken branch is an exposed
  // instruction-fetch bubble, so the steady-state loop below is branch-free apart from the (not taken) vote.
  auto issue_one = [&](auto N_, auto STEADY) {
    constexpr int n = decltype(N_)::value;
    constexpr bool isk = n >= 2;
    const unsigned stage = isk ? ks_next : vs_next;
    if constexpr (decltype(STEADY)::value) {
      dma16(src[n], dst[n] + stage * smul[n]);
    } else {
      const int T = isk ? tk_next : tv_next;
      const bool pad = (n & 1) && pad_hi;
      const char* p = src[n];
      if (T == t_ragged) {                                                 // ragged last tile (uniform branch)
        const int x = (wid + 4 * (n & 1)) * 64 + lane;
        const int r = x / SEGS, sg = x - r * SEGS;
        if (r >= rem_last && !pad)                                         // padded key: zeros, K slot 41 = 1
          p = reinterpret_cast<const char*>(g_w4_lines + ((sg == 5 && isk) ? 8 : 16));
      }
      const bool live = T < n_tiles;
      if (!live) p = reinterpret_cast<const char*>(g_w4_lines + 16);      // past the last tile: padding instruction
      dma16(p, live ? dst[n] + stage * smul[n] : lds0 + DUMMY_B);
    }
    src[n] += inc[n];
  };
  auto advance_v = [&]() { ++tv_next; vs_next = vs_next == V_STAGES - 1 ? 0 : vs_next + 1; };
  auto advance_k = [&]() { ++tk_next; ks_next = ks_next == K_STAGES - 1 ? 0 : ks_next + 1; };
  using I0_ = std::integral_constant<int, 0>;
  using I1_ = std::integral_constant<int, 1>;
  using I2_ = std::integral_constant<int, 2>;
  using I3_ = std::integral_constant<int, 3>;
  auto issue_v = [&]() { issue_one(I0_{}, std::false_type{}); issue_one(I1_{}, std::false_type{}); advance_v(); };
  auto issue_k = [&]() { issue_one(I2_{}, std::false_type{}); issue_one(I3_{}, std::false_type{}); advance_k(); };

  // ---- claim the asm-owned AGPRs (the clobber list makes them part of the kernel's register allocation) and clear O
  asm volatile("" ::: LGD_W4_AGPR_CLOBBERS);
  {
    float z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));          // a register, not a literal: v_accvgpr_write takes no constant here
    agpr_fill<0, 64>(z);
  }

  // ---- Q^T fragments (B operand of S^T = K Q^T), pre-scaled to the log2 domain, into a[64:87].  Head dims 40..47
  // (k-step 2, upper lane half): slot 40 = -m_ref (K holds 1 there for valid keys), slot 41 = -PADV (K holds 1 there
  // for padded keys).  q2d0 = dword 0 of the k-step-2 fragment as loaded (the one dword set_ref rewrites).
  float q2d0[2];
  auto load_q = [&](auto QB) {
    constexpr int qb = decltype(QB)::value;
    int qrow = q0 + qb * 32 + qi;
    if (qrow >= a.Sq) qrow = a.Sq - 1;
    f32x4 w[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int dd = ks * 16 + hh * 8;
      half8_t f = (half8_t){0, (half_t)(-PADV), 0, 0, 0, 0, 0, 0};
      if (dd < D) {
        f = *reinterpret_cast<const half8_t*>(Qb + (long)qrow * a.ldq + dd);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (half_t)((float)f[e] * a.scale_log2);
      }
      w[ks] = __builtin_bit_cast(f32x4, f);
    }
    q2d0[qb] = w[2][0];
    agpr_write<QF0 + 12 * qb + 0>(w[0][0]); agpr_write<QF0 + 12 * qb + 1>(w[0][1]);
    agpr_write<QF0 + 12 * qb + 2>(w[0][2]); agpr_write<QF0 + 12 * qb + 3>(w[0][3]);
    agpr_write<QF0 + 12 * qb + 4>(w[1][0]); agpr_write<QF0 + 12 * qb + 5>(w[1][1]);
    agpr_write<QF0 + 12 * qb + 6>(w[1][2]); agpr_write<QF0 + 12 * qb + 7>(w[1][3]);
    agpr_write<QF0 + 12 * qb + 8>(w[2][0]); agpr_write<QF0 + 12 * qb + 9>(w[2][1]);
    agpr_write<QF0 + 12 * qb + 10>(w[2][2]); agpr_write<QF0 + 12 * qb + 11>(w[2][3]);
  };
  load_q(std::integral_constant<int, 0>{});
  load_q(std::integral_constant<int, 1>{});
  float m_ref[2] = {0.f, 0.f};
  auto set_ref = [&](auto QB, float m) {         // m is a multiple of fp16's spacing: the slot holds it exactly
    constexpr int qb = decltype(QB)::value;
    m_ref[qb] = m;
    const half2_t slot = {(half_t)(-m), (half_t)(-PADV)};
    agpr_write<QF0 + 12 * qb + 8>(hh ? __builtin_bit_cast(float, slot) : q2d0[qb]);
    asm volatile("s_nop 3");                      // VALU write of an AGPR -> MFMA read
  };

  // ---- fragment addresses (stage 0)
  const unsigned char* k_lane = smem + qi * ROWB + hh * 16;
  const int j16 = lane & 15;
  const unsigned char* v_lane = smem + V0_B + (4 * hh + (j16 >> 2)) * ROWB + (16 * ((lane >> 4) & 1) + 4 * (j16 & 3)) * 2;

  f32x16 sA[2][2], sB[2][2];
  half8_t kf[6];                                   // K fragments (ks, kb) -> kf[2 ks + kb] of the tile QK^T multiplies next
  auto read_k = [&](unsigned stage) {
    const unsigned char* kbase = k_lane + stage * TILE_B;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) kf[2 * ks + kb] = *reinterpret_cast<const half8_t*>(kbase + kb * 32 * ROWB + ks * 32);
  };
  // MFMA number M (0..11) of S = K Q^T: k-step major, the four (key block, query block) chains alternating
  auto qk_mfma = [&](auto M_, f32x16 (&s)[2][2]) {
    constexpr int M = decltype(M_)::value, ks = M / 4, kb = (M / 2) % 2, qb = M % 2;
    if constexpr (ks == 0) mfma_qk_first<3 * qb + ks>(s[qb][kb], kf[2 * ks + kb]);
    else mfma_qk_acc<3 * qb + ks>(s[qb][kb], kf[2 * ks + kb]);
  };
  // step STEP (0..15) of the running row max over the 32 scores a lane holds of one query block
  auto max_step = [&](auto STEP_, float& mx, const f32x16 (&s)[2]) {
    constexpr int STEP = decltype(STEP_)::value;
    if constexpr (STEP == 0) mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
    else if constexpr (STEP < 7) mx = fmaxf(fmaxf(mx, s[0][2 * STEP + 1]), s[0][2 * STEP + 2]);
    else if constexpr (STEP == 7) mx = fmaxf(fmaxf(mx, s[0][15]), s[1][0]);
    else if constexpr (STEP < 15) mx = fmaxf(fmaxf(mx, s[1][2 * (STEP - 8) + 1]), s[1][2 * (STEP - 8) + 2]);
    else mx = fmaxf(mx, s[1][15]);
  };
  auto tile_max = [&](const f32x16 (&s)[2]) {
    float mx;
    for_seq([&](auto I) { max_step(I, mx, s); }, std::make_integer_sequence<int, 16>{});
    return pair_max(mx);
  };

  if constexpr (PIPE) {
    // ---- prologue: K(0..3), V(0), V(1) on their way; S(0) referenced to its own row max; fragments of K(1) requested
    issue_v(); issue_v();
    issue_k(); issue_k(); issue_k(); issue_k();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_k(0);
    for_seq([&](auto I) { qk_mfma(I, sA); }, std::make_integer_sequence<int, 12>{});
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sA[0][0]), "+v"(sA[0][1]), "+v"(sA[1][0]), "+v"(sA[1][1]));
    auto first_ref = [&](auto QB) {
      constexpr int qb = decltype(QB)::value;
      const float m = (float)(half_t)tile_max(sA[qb]);
  #pragma unroll
      for (int kb = 0; kb < 2; ++kb)
  #pragma unroll
        for (int r = 0; r < 16; ++r) sA[qb][kb][r] -= m;
      set_ref(QB, m);
    };
    first_ref(std::integral_constant<int, 0>{});
    first_ref(std::integral_constant<int, 1>{});
    if (n_tiles > 1) read_k(1);
    unsigned kr_stage = 2;                 // stage of K(t+2), whose fragments iteration t requests
    unsigned vr_stage = 0;                 // stage of V(t)

    // One key tile = 28 MFMA slots (12 of S(t+1) = K(t+1) Q^T, 16 of O += V(t)^T P(t)), every slot = {one MFMA, its share
    // of the softmax VALU work, at most two LDS fragment requests}, pinned in this order (sched_barrier): the VALU and
    // LDS instructions sit in the issue gaps of the wave's own MFMAs.  What a gap hides was measured
    // (tools/probe_mfma_valu_overlap.hip, one wave per SIMD): 15.3 ns per MFMA with up to 3 v_exp_f32 or 5 plain VALU
    // instructions beside it, +3.4 ns for every further exponential — so the 64 exponentials of a tile are dealt out
    // three per slot in the order the PV MFMAs need them (k-step 0 by slot 12, 1 by 16, 2 by 20, 3 by 24), the row max of
    // S(t+1) two steps per slot from slot 12 on, the V fragments six slots ahead of their MFMAs and the K fragments of
    // the next iteration in the last six slots.
    // `sc` holds S(t) (referenced), `sn` receives S(t+1); the caller alternates the two register blocks.
    auto tile = [&](f32x16 (&sc)[2][2], f32x16 (&sn)[2][2], bool rd_k, auto STEADY) {
      // V(t+2) -> stage of V(t-1) (read in iteration t-1), K(t+4) -> stage of K(t+1) (fragments read at the end of
      // iteration t-1): the barrier at the end of iteration t-1 lies in between.  The four DMA instructions sit behind
      // the MFMAs of slots 1, 8, 15, 22 (an LDS-DMA instruction occupies the wave's issue for 60+ cycles).
      const unsigned char* vbase = v_lane + vr_stage * TILE_B;
      const unsigned char* kbase = k_lane + kr_stage * TILE_B;
      half8_t vf[2][4], pf[2][4];
      float pe[64];
      float mx[2] = {0.f, 0.f};
      if constexpr (ABL & 8)
        for (int i = 0; i < 8; ++i) vf[i & 1][i >> 1] = kf[i % 6];
      // V^T fragment (dvb, ku), A operand of O^T = V^T P^T: two transposing reads (key rows 16 ku + 4 hh + 0..3 and + 8
      // of the natural [key][dv] tile), in the key order the P registers have
      auto read_v = [&](auto DVB, auto KU) {
        constexpr int dvb = decltype(DVB)::value, ku = decltype(KU)::value;
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(uintptr_t)(unsigned)(uintptr_t)(vbase + ku * 16 * ROWB + dvb * 64));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(uintptr_t)(unsigned)(uintptr_t)(vbase + ku * 16 * ROWB + 8 * ROWB + dvb * 64));
        const short8_t both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        vf[dvb][ku] = __builtin_bit_cast(half8_t, both);
      };
      // exponential E (0..63), in the order the PV MFMAs need them: k-step ku = E / 16, query block (E % 16) / 8
      auto exp_one = [&](auto E_) {
        constexpr int E = decltype(E_)::value, ku = E / 16, qb = (E % 16) / 8, j = E % 8;
        if constexpr (ABL & 4) pe[E] = sc[qb][ku / 2][8 * (ku % 2) + j];
        else pe[E] = __builtin_amdgcn_exp2f(sc[qb][ku / 2][8 * (ku % 2) + j]);
        if constexpr (j % 2 == 1) {
          pf[qb][ku][j - 1] = (half_t)pe[E - 1];
          pf[qb][ku][j] = (half_t)pe[E];
          // the packs happen HERE, not in front of the MFMA that reads the fragment (hipcc sinks them there otherwise,
          // and an asm MFMA gets no wait states behind a VALU write of its operand)
          if constexpr (j == 7) asm volatile("" : "+v"(pf[qb][ku]));
        }
      };
      auto slot = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        if constexpr (ABL & 16) {
          if constexpr (S < 12) asm volatile("" : "+v"(sn[S % 2][(S / 2) % 2]) : "v"(kf[S / 2]));
          else asm volatile("" ::"v"(vf[((S - 12) / 2) % 2][(S - 12) / 4]), "v"(pf[(S - 12) % 2][(S - 12) / 4]));
        } else if constexpr (S < 12) {
          qk_mfma(S_, sn);
        } else {
          constexpr int M = S - 12, ku = M / 4, dvb = (M / 2) % 2, qb = M % 2;
          mfma_pv<2 * qb + dvb>(vf[dvb][ku], pf[qb][ku]);
        }
        // three exponentials per slot (slots 0..21: 64 of them, 66 places)
        if constexpr (3 * S < 64) exp_one(std::integral_constant<int, 3 * S>{});
        if constexpr (3 * S + 1 < 64) exp_one(std::integral_constant<int, 3 * S + 1>{});
        if constexpr (3 * S + 2 < 64) exp_one(std::integral_constant<int, 3 * S + 2>{});
        // V fragments: k-step ku's two fragments in slots 4 ku + 4, 4 ku + 5 (their MFMAs start at slot 12 + 4 ku)
        if constexpr (S >= 4 && S < 20 && (S % 4) < 2 && !(ABL & 8))
          read_v(std::integral_constant<int, S % 4>{}, std::integral_constant<int, (S - 4) / 4>{});
        // row max of S(t+1), complete since slot 11 (the asm MFMAs' results need no explicit wait states by now: the last
        // one is >= 1 MFMA = 64+ cycles back at slot 12... the first step reads the chain that finished at slot 8)
        if constexpr (S >= 12 && !(ABL & 32)) {
          constexpr int q = (S - 12) / 8, st0 = 2 * ((S - 12) % 8);
          max_step(std::integral_constant<int, st0>{}, mx[q], sn[q]);
          max_step(std::integral_constant<int, st0 + 1>{}, mx[q], sn[q]);
          asm volatile("" : "+v"(mx[q]));          // keeps the steps in this slot (they would sink to the vote)
        }
        // fragments of K(t+2) for the next iteration (landed before the previous barrier)
        if constexpr (S >= 22) {
          if (decltype(STEADY)::value || rd_k)
            kf[S - 22] = *reinterpret_cast<const half8_t*>(kbase + ((S - 22) & 1) * 32 * ROWB + ((S - 22) >> 1) * 32);
        }
        if constexpr (!(ABL & 1)) {
          if constexpr (S == 1) issue_one(I0_{}, STEADY);
          if constexpr (S == 8) { issue_one(I1_{}, STEADY); advance_v(); }
          if constexpr (S == 15) issue_one(I2_{}, STEADY);
          if constexpr (S == 22) { issue_one(I3_{}, STEADY); advance_k(); }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      for_seq(slot, std::make_integer_sequence<int, 28>{});
      kr_stage = kr_stage == K_STAGES - 1 ? 0 : kr_stage + 1;
      vr_stage = vr_stage == V_STAGES - 1 ? 0 : vr_stage + 1;
      mx[0] = pair_max(mx[0]);
      mx[1] = pair_max(mx[1]);
      // reference of tile t+1: raised on a wave-uniform vote only (row max more than 2^8 above it); PV(t) is complete in
      // issue order, so the rescale covers everything accumulated at the old reference exactly once (S(t+1) is shifted)
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(fmaxf(mx[0], mx[1]) > 8.f) != 0, 0)) {
        asm volatile("s_nop 15\n\ts_nop 3");          // the last PV MFMAs have written their accumulators
        auto raise = [&](auto QB) {
          constexpr int qb = decltype(QB)::value;
          const float m_new = (float)(half_t)fmaxf(m_ref[qb], mx[qb] + m_ref[qb]);
          const float shift = m_new - m_ref[qb];
          agpr_scale<32 * qb, 32>(__builtin_amdgcn_exp2f(-shift));
  #pragma unroll
          for (int kb = 0; kb < 2; ++kb)
  #pragma unroll
            for (int r = 0; r < 16; ++r) sn[qb][kb][r] -= shift;
          set_ref(QB, m_new);
        };
        raise(std::integral_constant<int, 0>{});
        raise(std::integral_constant<int, 1>{});
      }
      // Everything but this iteration's four DMA instructions has landed: V(t+1) and K(t+3) of the previous iteration's
      // issue.  Behind the barrier every wave is done with V(t)'s fragments and holds K(t+2)'s in registers.
      if constexpr (!(ABL & 2)) {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    };
    // the last key tile: nothing to multiply ahead
    auto last_tile = [&](f32x16 (&sc)[2][2]) {
      const unsigned char* vbase = v_lane + vr_stage * TILE_B;
  #pragma unroll
      for (int ku = 0; ku < 4; ++ku) {
        half8_t pfq[2];
  #pragma unroll
        for (int qb = 0; qb < 2; ++qb)
  #pragma unroll
          for (int j = 0; j < 8; ++j) pfq[qb][j] = (half_t)__builtin_amdgcn_exp2f(sc[qb][ku >> 1][8 * (ku & 1) + j]);
        half8_t vfd[2];
  #pragma unroll
        for (int dvb = 0; dvb < 2; ++dvb) {
          const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(uintptr_t)(unsigned)(uintptr_t)(vbase + ku * 16 * ROWB + dvb * 64));
          const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(uintptr_t)(unsigned)(uintptr_t)(vbase + ku * 16 * ROWB + 8 * ROWB + dvb * 64));
          const short8_t both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          vfd[dvb] = __builtin_bit_cast(half8_t, both);
        }
        mfma_pv_fresh<0>(vfd[0], pfq[0]); mfma_pv_fresh<2>(vfd[0], pfq[1]);
        mfma_pv<1>(vfd[1], pfq[0]); mfma_pv<3>(vfd[1], pfq[1]);
      }
    };
    int t = 0;
    // steady state: both iterations issue interior tiles only (K(t+4), K(t+5) exist and have no padded keys)
    for (; t + 5 < t_ragged; t += 2) {
      tile(sA, sB, true, std::true_type{});
      tile(sB, sA, true, std::true_type{});
    }
    for (; t + 2 < n_tiles; t += 2) {
      tile(sA, sB, t + 2 < n_tiles, std::false_type{});
      tile(sB, sA, t + 3 < n_tiles, std::false_type{});
    }
    if (t + 1 < n_tiles) {          // two tiles left
      tile(sA, sB, false, std::false_type{});
      last_tile(sB);
    } else {
      last_tile(sA);
    }
  } else {
    // ---- two waves per SIMD: plain per-tile sequence.  K(0..2), V(0), V(1) on their way; iteration t issues V(t+2)
    // (stage of V(t-1)) and K(t+3) (stage of K(t-1)) and waits, at its end, for the previous iteration's DMAs.
    issue_v(); issue_v();
    issue_k(); issue_k(); issue_k();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned kst = 0, vst = 0;
    for (int t = 0; t < n_tiles; ++t) {
      issue_v();
      issue_k();
      read_k(kst);
      for_seq([&](auto I) { qk_mfma(I, sA); }, std::make_integer_sequence<int, 12>{});
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sA[0][0]), "+v"(sA[0][1]), "+v"(sA[1][0]), "+v"(sA[1][1]));
      float mx[2] = {tile_max(sA[0]), tile_max(sA[1])};
      // reference: the first tile's own row max, afterwards raised on a wave-uniform vote only (row max more than 2^8
      // above it); PV(t-1) is complete, so the rescale covers everything accumulated at the old reference exactly once
      if (__builtin_expect(t == 0 || __builtin_amdgcn_ballot_w64(fmaxf(mx[0], mx[1]) > 8.f) != 0, 0)) {
        asm volatile("s_nop 15\n\ts_nop 3");
        auto raise = [&](auto QB) {
          constexpr int qb = decltype(QB)::value;
          const float m_new = (float)(half_t)(t == 0 ? mx[qb] : fmaxf(m_ref[qb], mx[qb] + m_ref[qb]));
          const float shift = m_new - m_ref[qb];
          agpr_scale<32 * qb, 32>(__builtin_amdgcn_exp2f(-shift));
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sA[qb][kb][r] -= shift;
          set_ref(QB, m_new);
        };
        raise(std::integral_constant<int, 0>{});
        raise(std::integral_constant<int, 1>{});
      }
      const unsigned char* vbase = v_lane + vst * TILE_B;
#pragma unroll
      for (int ku = 0; ku < 4; ++ku) {
        half8_t pfq[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int j = 0; j < 8; ++j) pfq[qb][j] = (half_t)__builtin_amdgcn_exp2f(sA[qb][ku >> 1][8 * (ku & 1) + j]);
        half8_t vfd[2];
#pragma unroll
        for (int dvb = 0; dvb < 2; ++dvb) {
          const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(uintptr_t)(unsigned)(uintptr_t)(vbase + ku * 16 * ROWB + dvb * 64));
          const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(uintptr_t)(unsigned)(uintptr_t)(vbase + ku * 16 * ROWB + 8 * ROWB + dvb * 64));
          const short8_t both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          vfd[dvb] = __builtin_bit_cast(half8_t, both);
        }
        mfma_pv_fresh<0>(vfd[0], pfq[0]); mfma_pv_fresh<2>(vfd[0], pfq[1]);
        mfma_pv<1>(vfd[1], pfq[0]); mfma_pv<3>(vfd[1], pfq[1]);
      }
      kst = kst == K_STAGES - 1 ? 0 : kst + 1;
      vst = vst == V_STAGES - 1 ? 0 : vst + 1;
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the padding instructions behind the last tile
  asm volatile("s_nop 15\n\ts_nop 3");              // the last PV MFMAs have written their accumulators

  // ---- epilogue.  Row sum = O^T row dv = 40: dv block 1, register 4 of the lanes with hh = 0
  auto store_o = [&](auto QB) {
    constexpr int qb = decltype(QB)::value;
    constexpr int O0 = 32 * qb;
    const float l = __shfl(agpr_read<O0 + 16 + 4>(), qi, 64);
    const int qrow = q0 + qb * 32 + qi;
    const float inv = 1.f / l;
    half_t* orow = a.o + (long)b * a.o_bs + (long)qrow * a.ldo + (long)h * D;
    // dv = 32 dvb + 8 g4 + 4 hh + r  <->  register 16 dvb + 4 g4 + r
    const half4_t o0 = {(half_t)(agpr_read<O0 + 0>() * inv), (half_t)(agpr_read<O0 + 1>() * inv), (half_t)(agpr_read<O0 + 2>() * inv), (half_t)(agpr_read<O0 + 3>() * inv)};
    const half4_t o1 = {(half_t)(agpr_read<O0 + 4>() * inv), (half_t)(agpr_read<O0 + 5>() * inv), (half_t)(agpr_read<O0 + 6>() * inv), (half_t)(agpr_read<O0 + 7>() * inv)};
    const half4_t o2 = {(half_t)(agpr_read<O0 + 8>() * inv), (half_t)(agpr_read<O0 + 9>() * inv), (half_t)(agpr_read<O0 + 10>() * inv), (half_t)(agpr_read<O0 + 11>() * inv)};
    const half4_t o3 = {(half_t)(agpr_read<O0 + 12>() * inv), (half_t)(agpr_read<O0 + 13>() * inv), (half_t)(agpr_read<O0 + 14>() * inv), (half_t)(agpr_read<O0 + 15>() * inv)};
    const half4_t o4 = {(half_t)(agpr_read<O0 + 16>() * inv), (half_t)(agpr_read<O0 + 17>() * inv), (half_t)(agpr_read<O0 + 18>() * inv), (half_t)(agpr_read<O0 + 19>() * inv)};
    if (qrow < a.Sq) {
      *reinterpret_cast<half4_t*>(orow + 0 + hh * 4) = o0;
      *reinterpret_cast<half4_t*>(orow + 8 + hh * 4) = o1;
      *reinterpret_cast<half4_t*>(orow + 16 + hh * 4) = o2;
      *reinterpret_cast<half4_t*>(orow + 24 + hh * 4) = o3;
      *reinterpret_cast<half4_t*>(orow + 32 + hh * 4) = o4;
      if (a.lse && hh == 0) a.lse[((long)b * a.H + h) * a.Sq + qrow] = m_ref[qb] + log2f(l);
    }
  };
  store_o(std::integral_constant<int, 0>{});
  store_o(std::integral_constant<int, 1>{});
}

}  // namespace

// lane threads launch concurrently (lanes.py): the option is an atomic whose first reader takes LGD_W4_PIPE once
static std::atomic<int> g_w4_pipe{-1};
void lgd_attn_w4_set_pipe(int v) { g_w4_pipe.store(v ? 1 : 0, std::memory_order_relaxed); }
static int w4_pipe() {
  int v = g_w4_pipe.load(std::memory_order_relaxed);
  if (v < 0) {
    static const int env = [] { const char* e = getenv("LGD_W4_PIPE"); return e ? (atoi(e) ? 1 : 0) : 1; }();
    int expect = -1;
    g_w4_pipe.compare_exchange_strong(expect, env, std::memory_order_relaxed);
    v = g_w4_pipe.load(std::memory_order_relaxed);
  }
  return v;
}

int lgd_attn_w4_supported(const AttnW4Args& a) {
  // K / V rows travel by global_load_lds_dwordx4 and Q by 16-byte loads: leading dimensions in units of 8 halves are not
  // enough, the BASE pointers (and the per-image strides) must be 16-byte aligned too; O leaves as 8-byte half4 stores.
  // Anything else falls back to the round-3 kernel (attn.hip), which only needs 2-byte alignment.
  auto al = [](const void* p, uintptr_t n) { return (reinterpret_cast<uintptr_t>(p) & (n - 1)) == 0; };
  return a.d == D && a.Sq >= 1 && a.Sk >= 1 && (a.ldq % 8) == 0 && (a.ldk % 8) == 0 && (a.ldv % 8) == 0 && (a.ldo % 4) == 0 &&
         al(a.q, 16) && al(a.k, 16) && al(a.v, 16) && al(a.o, 8) && (a.q_bs % 8) == 0 && (a.k_bs % 8) == 0 &&
         (a.v_bs % 8) == 0 && (a.o_bs % 4) == 0;
}

int lgd_attn_w4_launch(const AttnW4Args& a, hipStream_t st) {
  dim3 grid((a.Sq + 255) / 256, a.H, a.B);
#ifdef LGD_W4_ABLATION
  static const int abl = [] { const char* e = getenv("LGD_W4_ABL"); return e ? atoi(e) : 0; }();
  switch (abl) {
    case 1: hipLaunchKernelGGL((attn_w4_kernel<1, true>), grid, dim3(256), 0, st, a); return lgd_check_launch();
    case 2: hipLaunchKernelGGL((attn_w4_kernel<2, true>), grid, dim3(256), 0, st, a); return lgd_check_launch();
    case 3: hipLaunchKernelGGL((attn_w4_kernel<3, true>), grid, dim3(256), 0, st, a); return lgd_check_launch();
    case 4: hipLaunchKernelGGL((attn_w4_kernel<4, true>), grid, dim3(256), 0, st, a); return lgd_check_launch();
    case 8: hipLaunchKernelGGL((attn_w4_kernel<8, true>), grid, dim3(256), 0, st, a); return lgd_check_launch();
    case 16: hipLaunchKernelGGL((attn_w4_kernel<16, true>), grid, dim3(256), 0, st, a); return lgd_check_launch();
    case 32: hipLaunchKernelGGL((attn_w4_kernel<32, true>), grid, dim3(256), 0, st, a); return lgd_check_launch();
    case 36: hipLaunchKernelGGL((attn_w4_kernel<36, true>), grid, dim3(256), 0, st, a); return lgd_check_launch();
    case 47: hipLaunchKernelGGL((attn_w4_kernel<47, true>), grid, dim3(256), 0, st, a); return lgd_check_launch();
    case 31: hipLaunchKernelGGL((attn_w4_kernel<31, true>), grid, dim3(256), 0, st, a); return lgd_check_launch();
    default: break;
  }
#endif
  // variant (lgd_set_option("attn_w4_pipe", v); initial value from LGD_W4_PIPE): 1 = one wave per SIMD with the in-wave
  // software pipeline (default), 0 = two waves per SIMD.  Measured equal within 2 % on MI355X (B = 16: 497 vs 500 us,
  // B = 8: 264 vs 259 us) — both sit at the SUM of their MFMA and softmax-VALU time, see DESIGN.md.
  if (w4_pipe()) hipLaunchKernelGGL((attn_w4_kernel<0, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((attn_w4_kernel<0, false>), grid, dim3(256), 0, st, a);
  return lgd_check_launch();
}
