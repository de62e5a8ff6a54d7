// GroupNorm(+SiLU) and LayerNorm, forward and input-gradient, for channels-last fp16 maps.
// All HBM-bound: 16-byte vector loads, fp32 statistics, wavefront-shuffle reductions.
#include "common.h"
#include "../../include/lgd_hip.h"

namespace {

// ------------------------------------------------------------------------------------------
// GroupNorm statistics, pass 1: each workgroup reduces a chunk of pixels of one image for all
// groups.  grid = (nchunk, B).  part[b][chunk][g] = {sum, sumsq}.
// A thread owns a fixed 8-channel vector column (tid + j*256 < C/8) and walks pixels, so loads of
// a wave are contiguous within a pixel row.
// ------------------------------------------------------------------------------------------
constexpr int GN_MAXC = 4096;   // SDXL-refiner up blocks normalise 1536 + 1536 channels

// Deterministic per-group reduction: every thread deposits its 8 per-channel partial sums (two
// quantities a, b) in LDS at [pixel lane][channel]; thread g < G then adds the channels of group g
// over all pixel lanes in a fixed order.  (No float atomics: results are bit-reproducible.)
constexpr int GN_PASS_C = 2048;   // channels covered by one pass of 256 8-channel vectors
constexpr int GN_UNROLL = 4;      // pixels a thread has in flight per loop trip (apply passes)
constexpr int GN_UNROLL_S = 8;    // same, forward statistics pass (one tensor, no stores)

__device__ __forceinline__ void gn_group_reduce(float* s_a, float* s_b, const float (&a)[8],
                                                const float (&b)[8], bool active, int slot, int c_lo,
                                                int c_n, int pl, int cpg, int G, float& acc_a,
                                                float& acc_b) {
  // slot = plane * (c_n) + (channel - c_lo) of this thread's first channel
  if (active) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { s_a[slot + e] = a[e]; s_b[slot + e] = b[e]; }
  }
  __syncthreads();
  // T = 2^k <= 256 / G consecutive lanes share a group: lane j adds elements j, j + T, ... of the group's
  // (channel, pixel-lane) list, a butterfly over the T lanes closes the sum, the group's total goes to thread g through
  // LDS.  Fixed order, no atomics: bit-reproducible; the serial version (one thread per group walking 40-80 LDS
  // words) was ~10 % of the statistics kernels on the small maps.
  __shared__ float s_tot[64][2];
  int T = 1;
  while (T * 2 * G <= 256 && T < 64) T *= 2;
  {
    const int g = threadIdx.x / T, j = threadIdx.x - g * T;
    float pa = 0.f, pb = 0.f;
    if (g < G) {
      int lo = g * cpg, hi = lo + cpg;
      if (lo < c_lo) lo = c_lo;
      if (hi > c_lo + c_n) hi = c_lo + c_n;
      const int n_e = hi > lo ? (hi - lo) * pl : 0;
      for (int e = j; e < n_e; e += T) {
        const int c = lo + e / pl, p_ = e - (e / pl) * pl;
        pa += s_a[p_ * c_n + (c - c_lo)];
        pb += s_b[p_ * c_n + (c - c_lo)];
      }
    }
    for (int o = 1; o < T; o <<= 1) {
      pa += __shfl_xor(pa, o, 64);
      pb += __shfl_xor(pb, o, 64);
    }
    if (g < G && j == 0) { s_tot[g][0] = pa; s_tot[g][1] = pb; }
  }
  __syncthreads();
  if (threadIdx.x < G) {
    acc_a += s_tot[threadIdx.x][0];
    acc_b += s_tot[threadIdx.x][1];
  }
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const half_t* __restrict__ x0,
                                                        const half_t* __restrict__ x1, int c0,
                                                        int c1, int HW, int G, float* part,
                                                        int nchunk) {
  __shared__ float s_a[GN_PASS_C], s_b[GN_PASS_C];
  const int C = c0 + c1;
  const int cpg = C / G;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int p_per = (HW + nchunk - 1) / nchunk;
  const int p_beg = chunk * p_per;
  int p_end = p_beg + p_per;
  if (p_end > HW) p_end = HW;
  const int nvec = C / 8;
  // thread -> (vector column, pixel lane): consecutive threads read consecutive channels of one
  // pixel; when C/8 < 256 the spare threads take further pixels of the chunk.
  const int vs = nvec < 256 ? nvec : 256;
  const int pl = 256 / vs;
  const int plane = threadIdx.x / vs;
  const int n_pass = (nvec + vs - 1) / vs;       // uniform trip count: the reduction has barriers
  float acc_s = 0.f, acc_q = 0.f;                // thread g < G: running sums of group g
  for (int pass = 0; pass < n_pass; ++pass) {
    const int v = threadIdx.x % vs + pass * vs;
    const bool active = v < nvec && plane < pl;
    const int c_lo = pass * vs * 8;
    const int c_n = (nvec - pass * vs < vs ? nvec - pass * vs : vs) * 8;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (active) {
      const int c = v * 8;
      const bool second = c >= c0;
      const half_t* src = second ? x1 : x0;
      const int cc = second ? c - c0 : c;
      const int ld = second ? c1 : c0;
      // GN_UNROLL_S independent 16-byte loads in flight per thread (zeros past the chunk end)
      for (int p = p_beg + plane; p < p_end; p += GN_UNROLL_S * pl) {
        half8_t h[GN_UNROLL_S];
#pragma unroll
        for (int u = 0; u < GN_UNROLL_S; ++u) {
          const int pp = p + u * pl;
          h[u] = pp < p_end ? *reinterpret_cast<const half8_t*>(src + ((long)b * HW + pp) * ld + cc)
                            : (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < GN_UNROLL_S; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float f = (float)h[u][e];
            s[e] += f;
            q[e] += f * f;
          }
      }
    }
    gn_group_reduce(s_a, s_b, s, q, active, plane * c_n + (v * 8 - c_lo), c_lo, c_n, pl, cpg, G, acc_s, acc_q);
  }
  if (threadIdx.x < G) {
    float* o = part + (((long)b * nchunk + chunk) * G + threadIdx.x) * 2;
    o[0] = acc_s;
    o[1] = acc_q;
  }
}

// pass 2: finalise statistics (every workgroup re-reduces the tiny partial table of its image),
// build per-channel scale/shift in LDS, normalise (+SiLU) a chunk of pixels.
__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* __restrict__ x0,
                                                        const half_t* __restrict__ x1, int c0,
                                                        int c1, int HW, int G, float eps,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int silu,
                                                        half_t* __restrict__ y,
                                                        const float* __restrict__ part, int nchunk,
                                                        float* stats, int napply) {
  __shared__ float s_mean[64], s_rstd[64];
  const int C = c0 + c1;
  const int cpg = C / G;
  const int b = blockIdx.y;
  // partial table [nchunk][G][2] -> (mean, rstd): 256 threads = (G groups) x (256/G slices), fixed
  // summation order (deterministic)
  __shared__ float s_ps[256], s_pq[256];
  {
    const int gi = threadIdx.x % G, sl = threadIdx.x / G, nsl = 256 / G;
    float s = 0.f, q = 0.f;
    if (sl < nsl)
      for (int ch = sl; ch < nchunk; ch += nsl) {
        const float* p = part + (((long)b * nchunk + ch) * G + gi) * 2;
        s += p[0];
        q += p[1];
      }
    s_ps[threadIdx.x] = s;
    s_pq[threadIdx.x] = q;
  }
  __syncthreads();
  if (threadIdx.x < G) {
    float s = 0.f, q = 0.f;
    for (int sl = 0; sl < 256 / G; ++sl) { s += s_ps[sl * G + threadIdx.x]; q += s_pq[sl * G + threadIdx.x]; }
    float n = (float)HW * cpg;
    float mean = s / n;
    float var = q / n - mean * mean;
    if (var < 0.f) var = 0.f;
    float rstd = rsqrtf(var + eps);
    s_mean[threadIdx.x] = mean;
    s_rstd[threadIdx.x] = rstd;
    if (stats && blockIdx.x == 0) {
      stats[((long)b * G + threadIdx.x) * 2 + 0] = mean;
      stats[((long)b * G + threadIdx.x) * 2 + 1] = rstd;
    }
  }
  __syncthreads();
  // A thread owns a fixed 8-channel column (scale/shift in registers) and walks the block's pixels
  // GN_UNROLL at a time: no per-element index division, no LDS traffic in the streaming loop.
  const int p_per = (HW + napply - 1) / napply;
  const int p_beg = blockIdx.x * p_per;
  int p_end = p_beg + p_per;
  if (p_end > HW) p_end = HW;
  const int nvec = C / 8;
  const int vs = nvec < 256 ? nvec : 256;
  const int pl = 256 / vs;
  const int plane = threadIdx.x / vs;
  const int n_pass = (nvec + vs - 1) / vs;
  for (int pass = 0; pass < n_pass; ++pass) {
    const int v = threadIdx.x % vs + pass * vs;
    if (v >= nvec || plane >= pl) continue;
    const int c = v * 8;
    float sa[8], sb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (c + e) / cpg;
      sa[e] = s_rstd[g] * gamma[c + e];
      sb[e] = beta[c + e] - s_mean[g] * sa[e];
    }
    const bool second = c >= c0;
    const half_t* src = second ? x1 : x0;
    const int cc = second ? c - c0 : c;
    const int ld = second ? c1 : c0;
    for (int p = p_beg + plane; p < p_end; p += GN_UNROLL * pl) {
      half8_t h[GN_UNROLL];
#pragma unroll
      for (int u = 0; u < GN_UNROLL; ++u) {
        const int pp = p + u * pl;
        if (pp < p_end) h[u] = *reinterpret_cast<const half8_t*>(src + ((long)b * HW + pp) * ld + cc);
      }
#pragma unroll
      for (int u = 0; u < GN_UNROLL; ++u) {
        const int pp = p + u * pl;
        if (pp < p_end) {
          half8_t o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float f = (float)h[u][e] * sa[e] + sb[e];
            if (silu) f = silu_f(f);
            o[e] = (half_t)f;
          }
          *reinterpret_cast<half8_t*>(y + ((long)b * HW + pp) * C + c) = o;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// GroupNorm(+SiLU) in ONE launch for maps whose (image, group set) slab fits the register file of a workgroup
// (round 4): grid = (G / kg, B); a workgroup owns kg whole groups of one image (kg = smallest count whose channels
// are a multiple of 8, so every 16-byte vector belongs to one workgroup), reads its slab ONCE into registers
// (MAXP pixels x 8 channels per thread), reduces mean, then the CENTRED squares (two-pass variance — no
// E[x^2] - mean^2 cancellation), and writes the normalised (+SiLU) slab.  The two-launch form reads the map
// twice and, at 16x16 and 8x8, both of its launches sit on the launch floor.  Fixed summation order, no atomics.
// ------------------------------------------------------------------------------------------
template <int MAXP>
__global__ __launch_bounds__(256) void gn_fused_kernel(const half_t* __restrict__ x0, const half_t* __restrict__ x1,
                                                        int c0, int c1, int HW, int G, float eps,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int silu, half_t* __restrict__ y, float* stats, int kg) {
  __shared__ float s_part[256 * 8];
  __shared__ float s_ch[256];
  __shared__ float s_g[8][2];
  const int tid = threadIdx.x;
  const int C = c0 + c1, cpg = C / G;
  const int b = blockIdx.y, g_lo = blockIdx.x * kg;
  const int W = kg * cpg, nv = W / 8, pl = 256 / nv;
  const int vcol = tid % nv, plane = tid / nv;
  const bool active = plane < pl;
  const int c = g_lo * cpg + vcol * 8;                 // first of this thread's 8 channels
  const bool second = c >= c0;
  const half_t* src = second ? x1 : x0;
  const int cc = second ? c - c0 : c;
  const int ld = second ? c1 : c0;
  half8_t h[MAXP];
#pragma unroll
  for (int u = 0; u < MAXP; ++u) {
    const int pp = plane + u * pl;
    h[u] = (active && pp < HW) ? *reinterpret_cast<const half8_t*>(src + ((long)b * HW + pp) * ld + cc)
                               : (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
  }
  // per-channel totals over the pixel lanes, then per-group totals: thread j < W owns channel j of the slab
  auto group_sum = [&](const float (&v)[8], int slot) {
#pragma unroll
    for (int e = 0; e < 8; ++e) s_part[tid * 8 + e] = active ? v[e] : 0.f;
    __syncthreads();
    if (tid < W) {
      float t = 0.f;
      for (int p = 0; p < pl; ++p) t += s_part[p * W + tid];
      s_ch[tid] = t;
    }
    __syncthreads();
    if (tid < kg) {
      float t = 0.f;
      for (int j = 0; j < cpg; ++j) t += s_ch[tid * cpg + j];
      s_g[tid][slot] = t / ((float)HW * cpg);
    }
    __syncthreads();
  };
  float a[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = 0.f;
#pragma unroll
  for (int u = 0; u < MAXP; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += (float)h[u][e];
  group_sum(a, 0);
  float mean[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { mean[e] = s_g[(vcol * 8 + e) / cpg][0]; a[e] = 0.f; }
#pragma unroll
  for (int u = 0; u < MAXP; ++u) {
    const bool ok = plane + u * pl < HW;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float dlt = (float)h[u][e] - mean[e];
      a[e] += ok ? dlt * dlt : 0.f;
    }
  }
  group_sum(a, 1);                                      // s_g[g][1] = variance
  if (tid < kg && stats) {
    stats[((long)b * G + g_lo + tid) * 2 + 0] = s_g[tid][0];
    stats[((long)b * G + g_lo + tid) * 2 + 1] = rsqrtf(s_g[tid][1] + eps);
  }
  if (!active) return;
  float sa[8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float rstd = rsqrtf(s_g[(vcol * 8 + e) / cpg][1] + eps);
    sa[e] = rstd * gamma[c + e];
    sb[e] = beta[c + e] - mean[e] * sa[e];
  }
#pragma unroll
  for (int u = 0; u < MAXP; ++u) {
    const int pp = plane + u * pl;
    if (pp < HW) {
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)h[u][e] * sa[e] + sb[e];
        if (silu) f = silu_f(f);
        o[e] = (half_t)f;
      }
      *reinterpret_cast<half8_t*>(y + ((long)b * HW + pp) * C + c) = o;
    }
  }
}

// option "gn_fused": largest map (pixels) the one-launch GroupNorm takes; 0 = always two launches
int g_gn_fused_hw = 256;

// ------------------------------------------------------------------------------------------
// Slab reduction shared by the one-launch GroupNorm BACKWARD below (round 6): a workgroup owns the (image, kg groups)
// slab; a thread folds its 8 channels into the (at most two, cpg >= 8) groups its vector touches and deposits the two
// pairs in LDS at [column][pixel lane]; one wave per column adds the pixel lanes (lane-strided, then a butterfly);
// thread t < kg adds the columns of group t.  Fixed order, no atomics: bit-reproducible.
// (A FORWARD slab kernel for the 64x64 / 32x32 maps — 320 KB slabs in the registers of 1024-thread workgroups — was
// built and measured in the same round: correct, but 0.4-0.7x of the two-launch form at 64x64 and +-5 % at 32x32: a
// workgroup streams its slab at ~19 GB/s whatever the batch, i.e. the per-CU limit on outstanding misses, and
// G / kg x B = 32 .. 128 workgroups cannot replace the ~1000 of the two-launch kernels.  Removed; profiles/HISTORY.md.)
// ------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void gn_slab_reduce(float* s_col, float* s_tot, float (*s_g)[2], bool active, int plane,
                                               int vcol, int nv, int pl, int cpg, int kg, float lo_a, float lo_b,
                                               float hi_a, float hi_b) {
  const int ncol = 2 * nv;
  if (active) {
    s_col[(2 * vcol + 0) * pl + plane] = lo_a;
    s_col[(2 * vcol + 1) * pl + plane] = hi_a;
    s_col[(ncol + 2 * vcol + 0) * pl + plane] = lo_b;
    s_col[(ncol + 2 * vcol + 1) * pl + plane] = hi_b;
  }
  __syncthreads();
  {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = w; c < ncol; c += NT / 64) {
      float pa = 0.f, pb = 0.f;
      for (int i = lane; i < pl; i += 64) {
        pa += s_col[c * pl + i];
        pb += s_col[(ncol + c) * pl + i];
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        pa += __shfl_xor(pa, o, 64);
        pb += __shfl_xor(pb, o, 64);
      }
      if (lane == 0) { s_tot[c] = pa; s_tot[64 + c] = pb; }
    }
  }
  __syncthreads();
  if (threadIdx.x < kg) {
    const int t = threadIdx.x;
    const int v_lo = (t * cpg) / 8, v_hi = ((t + 1) * cpg - 1) / 8;
    float a = 0.f, b = 0.f;
    for (int v = v_lo; v <= v_hi; ++v) {
      const int which = t - (8 * v) / cpg;          // 0: the group the vector starts in, 1: the next one
      a += s_tot[2 * v + which];
      b += s_tot[64 + 2 * v + which];
    }
    s_g[t][0] = a;
    s_g[t][1] = b;
  }
  __syncthreads();
}

// GroupNorm backward (w.r.t. x) in ONE launch: x and gy of the slab are read once into registers (running pointers and
// per-vector conversion pinned by empty asm statements: left alone hipcc materialises every address and every fp32
// conversion up front and spills), S1 = sum dxhat and S2 = sum dxhat * xhat reduced as above, dx written from the
// registers.  The two-launch form read both tensors twice and, on the 8x8 / 16x16 maps of the guidance backward (4
// images), paid two launch floors: measured 13.6 -> 9.5 us (8x8, C = 1280), 14.8 -> 11.7 (16x16, C = 1280), 23.4 -> 11.1
// (8x8, C = 2560), 26.1 -> 21.0 (16x16, C = 2560).  Slabs of more than 96 KB (x + gy) stay on the two-launch kernels:
// there a workgroup is bound by its CU's outstanding misses (32x32, C = 640: 18.1 -> 25.5 us).
template <int NT, int MAXP, bool SILU>
__global__ __launch_bounds__(NT) void gn_bwd_slab_kernel(const half_t* __restrict__ gy, const half_t* __restrict__ x0,
                                                          const half_t* __restrict__ x1, int c0, int c1, int HW, int G,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ stats, half_t* gx0,
                                                          half_t* gx1, int accumulate, int kg) {
  __shared__ float s_col[4 * NT];
  __shared__ float s_tot[128];
  __shared__ float s_g[8][2];
  const int tid = threadIdx.x;
  const int C = c0 + c1, cpg = C / G;
  const int b = blockIdx.y, g_lo = blockIdx.x * kg;
  const int W = kg * cpg, nv = W / 8, pl = NT / nv;
  const int vcol = tid % nv, plane = tid / nv;
  const bool active = plane < pl;
  const int c = g_lo * cpg + vcol * 8;
  const bool second = c >= c0;
  const int ld = second ? c1 : c0;
  const long xoff = (long)b * HW * ld + (second ? c - c0 : c);
  const half_t* src = (second ? x1 : x0) + xoff;
  const half_t* gsrc = gy + (long)b * HW * C + c;
  half8_t hx[MAXP], hg[MAXP];
  {
    const half_t* lp = src + (long)plane * ld;           // ONE running pointer per tensor, pinned by the empty asm below
    const half_t* gp = gsrc + (long)plane * C;
    const long lstep = (long)pl * ld, gstep = (long)pl * C;
#pragma unroll
    for (int u = 0; u < MAXP; ++u) {
      const int pp = plane + u * pl;
      const bool ok = active && pp < HW;
      hx[u] = *reinterpret_cast<const half8_t*>(ok ? lp : src);      // branch-free: a valid address either way (masked where used)
      hg[u] = *reinterpret_cast<const half8_t*>(ok ? gp : gsrc);
      lp += lstep;
      gp += gstep;
      asm volatile("" : "+v"(lp), "+v"(gp));
    }
  }
  const int gl = (8 * vcol) / cpg;
  const int e0 = (gl + 1) * cpg - 8 * vcol;
  // xhat = x * ra + rb;  z = gm * xhat + bt
  float ra[8], rb[8], gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int g = g_lo + (e < e0 ? gl : gl + 1);
    const bool in = active && g < G;                    // the last vector of the slab never reaches past its groups; guard anyway
    const float mean = in ? stats[((long)b * G + g) * 2] : 0.f;
    const float rstd = in ? stats[((long)b * G + g) * 2 + 1] : 0.f;
    ra[e] = rstd;
    rb[e] = -mean * rstd;
    gm[e] = active ? gamma[c + e] : 0.f;
    bt[e] = active ? beta[c + e] : 0.f;
  }
  float lo_a = 0.f, lo_b = 0.f, hi_a = 0.f, hi_b = 0.f;
  {
    float a1[8], a2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
#pragma unroll
    for (int u = 0; u < MAXP; ++u) {
      __builtin_amdgcn_sched_barrier(0);                  // convert vector by vector (register pressure)
      asm volatile("" : "+v"(hx[u]), "+v"(hg[u]));
      if (!(active && plane + u * pl < HW)) hg[u] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};   // zero gradient: contributes nothing
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (float)hx[u][e] * ra[e] + rb[e];
        float dz = (float)hg[u][e];
        if constexpr (SILU) dz *= silu_grad_f(gm[e] * xh + bt[e]);
        const float dxh = dz * gm[e];
        a1[e] += dxh;
        a2[e] += dxh * xh;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (e < e0) { lo_a += a1[e]; lo_b += a2[e]; }
      else { hi_a += a1[e]; hi_b += a2[e]; }
    }
  }
  gn_slab_reduce<NT>(s_col, s_tot, s_g, active, plane, vcol, nv, pl, cpg, kg, lo_a, lo_b, hi_a, hi_b);
#pragma unroll
  for (int u = 0; u < MAXP; ++u) asm volatile("" : "+v"(hx[u]), "+v"(hg[u]));     // keep the slab fp16 across the reduction
  if (!active) return;
  const float inv_n = 1.f / ((float)HW * cpg);
  const float m1_lo = s_g[gl][0] * inv_n, m2_lo = s_g[gl][1] * inv_n;
  const float m1_hi = e0 < 8 ? s_g[gl + 1][0] * inv_n : 0.f, m2_hi = e0 < 8 ? s_g[gl + 1][1] * inv_n : 0.f;
  half_t* dst = (second ? gx1 : gx0) + xoff + (long)plane * ld;
  const long dstep = (long)pl * ld;
#pragma unroll
  for (int u = 0; u < MAXP; ++u, dst += dstep) {
    const int pp = plane + u * pl;
    asm volatile("" : "+v"(dst));
    if (pp < HW) {
      half8_t ho;
      if (accumulate) ho = *reinterpret_cast<const half8_t*>(dst);
      half8_t o;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(hx[u]), "+v"(hg[u]));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (float)hx[u][e] * ra[e] + rb[e];
        float dz = (float)hg[u][e];
        if constexpr (SILU) dz *= silu_grad_f(gm[e] * xh + bt[e]);
        const float dxh = dz * gm[e];
        float dx = ra[e] * (dxh - (e < e0 ? m1_lo : m1_hi) - xh * (e < e0 ? m2_lo : m2_hi));
        if (accumulate) dx += (float)ho[e];
        o[e] = (half_t)dx;
      }
      *reinterpret_cast<half8_t*>(dst) = o;
    }
  }
}

// option "ln_stream": 1 = (default) statistics-only LayerNorm runs ln_stats_kernel, 0 = the row kernels
int g_ln_stream = 1;

// option "gn_slab": 1 = the one-launch backward takes every slab of <= 96 KB it can hold (default), 0 = two launches
int g_gn_slab = 1;

// slab geometry of a GroupNorm problem: kg groups per workgroup (smallest count whose channels fill whole 16-byte
// vectors); false when the slab kernels cannot take it (a vector would straddle three groups, too many columns)
bool gn_slab_geometry(int C, int G, int& kg, int& nv) {
  const int cpg = C / G;
  kg = 1;
  while ((kg * cpg) % 8) kg *= 2;
  nv = kg * cpg / 8;
  return cpg >= 8 && kg <= 8 && (G % kg) == 0 && nv <= 32;
}

// ------------------------------------------------------------------------------------------
// GroupNorm backward (w.r.t. x).  With xhat = (x-mean)*rstd, z = gamma*xhat+beta, y = act(z):
//   dz = gy * act'(z);  dxhat = dz*gamma
//   dx = rstd * (dxhat - mean_g(dxhat) - xhat * mean_g(dxhat*xhat))
// pass 1 accumulates per (b, g): S1 = sum dxhat, S2 = sum dxhat*xhat (same chunking as forward).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(
    const half_t* __restrict__ gy, const half_t* __restrict__ x0, const half_t* __restrict__ x1,
    int c0, int c1, int HW, int G, const float* __restrict__ gamma, const float* __restrict__ beta,
    int silu, const float* __restrict__ stats, float* part, int nchunk) {
  __shared__ float s_a[GN_PASS_C], s_b[GN_PASS_C];
  const int C = c0 + c1;
  const int cpg = C / G;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int p_per = (HW + nchunk - 1) / nchunk;
  const int p_beg = chunk * p_per;
  int p_end = p_beg + p_per;
  if (p_end > HW) p_end = HW;
  const int nvec = C / 8;
  const int vs = nvec < 256 ? nvec : 256;
  const int pl = 256 / vs;
  const int plane = threadIdx.x / vs;
  const int n_pass = (nvec + vs - 1) / vs;
  float acc_1 = 0.f, acc_2 = 0.f;
  for (int pass = 0; pass < n_pass; ++pass) {
    const int v = threadIdx.x % vs + pass * vs;
    const bool active = v < nvec && plane < pl;
    const int c_lo = pass * vs * 8;
    const int c_n = (nvec - pass * vs < vs ? nvec - pass * vs : vs) * 8;
    float a1[8], a2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    if (active) {
      const int c = v * 8;
      const bool second = c >= c0;
      const half_t* src = second ? x1 : x0;
      const int cc = second ? c - c0 : c;
      const int ld = second ? c1 : c0;
      float mean[8], rstd[8], gm[8], bt[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int g = (c + e) / cpg;
        mean[e] = stats[((long)b * G + g) * 2];
        rstd[e] = stats[((long)b * G + g) * 2 + 1];
        gm[e] = gamma[c + e];
        bt[e] = beta[c + e];
      }
      for (int p = p_beg + plane; p < p_end; p += GN_UNROLL * pl) {
        half8_t hx[GN_UNROLL], hg[GN_UNROLL];
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
          const int pp = p + u * pl;
          if (pp < p_end) {
            hx[u] = *reinterpret_cast<const half8_t*>(src + ((long)b * HW + pp) * ld + cc);
            hg[u] = *reinterpret_cast<const half8_t*>(gy + ((long)b * HW + pp) * C + c);
          } else {
            hg[u] = (half8_t){0, 0, 0, 0, 0, 0, 0, 0};  // zero gradient: contributes nothing
            hx[u] = hg[u];
          }
        }
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float xh = ((float)hx[u][e] - mean[e]) * rstd[e];
            float dz = (float)hg[u][e];
            if (silu) dz *= silu_grad_f(gm[e] * xh + bt[e]);
            float dxh = dz * gm[e];
            a1[e] += dxh;
            a2[e] += dxh * xh;
          }
      }
    }
    gn_group_reduce(s_a, s_b, a1, a2, active, plane * c_n + (v * 8 - c_lo), c_lo, c_n, pl, cpg, G, acc_1, acc_2);
  }
  if (threadIdx.x < G) {
    float* o = part + (((long)b * nchunk + chunk) * G + threadIdx.x) * 2;
    o[0] = acc_1;
    o[1] = acc_2;
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(
    const half_t* __restrict__ gy, const half_t* __restrict__ x0, const half_t* __restrict__ x1,
    int c0, int c1, int HW, int G, const float* __restrict__ gamma, const float* __restrict__ beta,
    int silu, const float* __restrict__ stats, half_t* gx0, half_t* gx1,
    const float* __restrict__ part, int nchunk, int accumulate, int napply) {
  __shared__ float s_m1[64], s_m2[64];
  const int C = c0 + c1;
  const int cpg = C / G;
  const int b = blockIdx.y;
  __shared__ float s_ps[256], s_pq[256];
  {
    const int gi = threadIdx.x % G, sl = threadIdx.x / G, nsl = 256 / G;
    float s = 0.f, q = 0.f;
    if (sl < nsl)
      for (int ch = sl; ch < nchunk; ch += nsl) {
        const float* p = part + (((long)b * nchunk + ch) * G + gi) * 2;
        s += p[0];
        q += p[1];
      }
    s_ps[threadIdx.x] = s;
    s_pq[threadIdx.x] = q;
  }
  __syncthreads();
  if (threadIdx.x < G) {
    float s1 = 0.f, s2 = 0.f;
    for (int sl = 0; sl < 256 / G; ++sl) { s1 += s_ps[sl * G + threadIdx.x]; s2 += s_pq[sl * G + threadIdx.x]; }
    float n = (float)HW * cpg;
    s_m1[threadIdx.x] = s1 / n;
    s_m2[threadIdx.x] = s2 / n;
  }
  __syncthreads();
  const int p_per = (HW + napply - 1) / napply;
  const int p_beg = blockIdx.x * p_per;
  int p_end = p_beg + p_per;
  if (p_end > HW) p_end = HW;
  const int nvec = C / 8;
  const int vs = nvec < 256 ? nvec : 256;
  const int pl = 256 / vs;
  const int plane = threadIdx.x / vs;
  const int n_pass = (nvec + vs - 1) / vs;
  for (int pass = 0; pass < n_pass; ++pass) {
    const int v = threadIdx.x % vs + pass * vs;
    if (v >= nvec || plane >= pl) continue;
    const int c = v * 8;
    float mean[8], rstd[8], gm[8], bt[8], m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (c + e) / cpg;
      mean[e] = stats[((long)b * G + g) * 2];
      rstd[e] = stats[((long)b * G + g) * 2 + 1];
      gm[e] = gamma[c + e];
      bt[e] = beta[c + e];
      m1[e] = s_m1[g];
      m2[e] = s_m2[g];
    }
    const bool second = c >= c0;
    const half_t* src = second ? x1 : x0;
    half_t* dst = second ? gx1 : gx0;
    const int cc = second ? c - c0 : c;
    const int ld = second ? c1 : c0;
    for (int p = p_beg + plane; p < p_end; p += 2 * pl) {
      half8_t hx[2], hg[2], ho[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pp = p + u * pl;
        if (pp < p_end) {
          const long off = ((long)b * HW + pp) * ld + cc;
          hx[u] = *reinterpret_cast<const half8_t*>(src + off);
          hg[u] = *reinterpret_cast<const half8_t*>(gy + ((long)b * HW + pp) * C + c);
          if (accumulate) ho[u] = *reinterpret_cast<const half8_t*>(dst + off);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pp = p + u * pl;
        if (pp < p_end) {
          half8_t o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float xh = ((float)hx[u][e] - mean[e]) * rstd[e];
            float dz = (float)hg[u][e];
            if (silu) dz *= silu_grad_f(gm[e] * xh + bt[e]);
            float dxh = dz * gm[e];
            float dx = rstd[e] * (dxh - m1[e] - xh * m2[e]);
            if (accumulate) dx += (float)ho[u][e];
            o[e] = (half_t)dx;
          }
          *reinterpret_cast<half8_t*>(dst + ((long)b * HW + pp) * ld + cc) = o;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (C <= 64*8*MAXV halfs).
// ------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 5;  // C up to 2560

__global__ __launch_bounds__(256) void layernorm_kernel(
    const half_t* __restrict__ x, long ldx, half_t* __restrict__ y, long ldy, int rows, int C,
    float eps, const float* __restrict__ gamma, const float* __restrict__ beta, float* stats,
    int rpb, long x_bs, long y_bs) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int bb = row / rpb, rr = row - bb * rpb;
  const half_t* xr = x + bb * x_bs + (long)rr * ldx;
  half_t* yr = y + bb * y_bs + (long)rr * ldy;
  const int nvec = C / 8;
  half8_t h[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int v = lane + j * 64;
    if (v < nvec) {
      h[j] = *reinterpret_cast<const half8_t*>(xr + v * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)h[j][e];
    }
  }
  const float mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int v = lane + j * 64;
    if (v < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float dlt = (float)h[j][e] - mean;
        q += dlt * dlt;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / C + eps);
  if (stats && lane == 0) {
    stats[(long)row * 2] = mean;
    stats[(long)row * 2 + 1] = rstd;
  }
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int v = lane + j * 64;
    if (v < nvec) {
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int c = v * 8 + e;
        o[e] = (half_t)(((float)h[j][e] - mean) * rstd * gamma[c] + beta[c]);
      }
      *reinterpret_cast<half8_t*>(yr + v * 8) = o;
    }
  }
}

// Same arithmetic as layernorm_kernel, restructured for memory-level parallelism: a wave owns ROWS
// consecutive rows and issues all their loads before the first reduction; gamma / beta of the lane's
// channel vectors are fetched once per wave.  MAXV = 16-byte vectors per lane (C <= 512 * MAXV).
template <int MAXV, int ROWS>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(
    const half_t* __restrict__ x, long ldx, half_t* __restrict__ y, long ldy, int rows, int C,
    float eps, const float* __restrict__ gamma, const float* __restrict__ beta, float* stats,
    int rpb, long x_bs, long y_bs) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
  if (row0 >= rows) return;
  const int nvec = C / 8;
  float gm[MAXV][8], bt[MAXV][8];
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int v = lane + j * 64;
    if (v < nvec && y) {                       // y == nullptr: statistics only
      const float4 g0 = *reinterpret_cast<const float4*>(gamma + v * 8);
      const float4 g1 = *reinterpret_cast<const float4*>(gamma + v * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(beta + v * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(beta + v * 8 + 4);
      gm[j][0] = g0.x; gm[j][1] = g0.y; gm[j][2] = g0.z; gm[j][3] = g0.w;
      gm[j][4] = g1.x; gm[j][5] = g1.y; gm[j][6] = g1.z; gm[j][7] = g1.w;
      bt[j][0] = b0.x; bt[j][1] = b0.y; bt[j][2] = b0.z; bt[j][3] = b0.w;
      bt[j][4] = b1.x; bt[j][5] = b1.y; bt[j][6] = b1.z; bt[j][7] = b1.w;
    }
  }
  half8_t h[ROWS][MAXV];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int row = row0 + r;
    const int bb = row / rpb, rr = row - bb * rpb;
    const half_t* xr = x + bb * x_bs + (long)rr * ldx;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int v = lane + j * 64;
      h[r][j] = (row < rows && v < nvec) ? *reinterpret_cast<const half8_t*>(xr + v * 8)
                                         : (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  // the ROWS rows of a wave go through every phase together: their butterfly reductions are independent chains that
  // interleave (one row after the other cost ROWS x two dependent 6-step shuffle chains per wave)
  float s[ROWS], q[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    s[r] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) s[r] += (float)h[r][j][e];          // lanes past nvec hold zeros
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) s[r] += __shfl_xor(s[r], o, 64);
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    s[r] /= C;                                                         // mean
    q[r] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
      if (lane + j * 64 < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dlt = (float)h[r][j][e] - s[r];
          q[r] += dlt * dlt;
        }
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) q[r] += __shfl_xor(q[r], o, 64);
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int row = row0 + r;
    if (row >= rows) break;  // wave-uniform
    const float mean = s[r];
    const float rstd = rsqrtf(q[r] / C + eps);
    if (stats && lane == 0) {
      stats[(long)row * 2] = mean;
      stats[(long)row * 2 + 1] = rstd;
    }
    if (!y) continue;                          // statistics only (uniform)
    const int bb = row / rpb, rr = row - bb * rpb;
    half_t* yr = y + bb * y_bs + (long)rr * ldy;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int v = lane + j * 64;
      if (v < nvec) {
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)h[r][j][e] - mean) * rstd * gm[j][e] + bt[j][e]);
        *reinterpret_cast<half8_t*>(yr + v * 8) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// LayerNorm STATISTICS ONLY (round 6; the other half of those LayerNorms rides in the consuming GEMM's epilogue,
// LGD_EPI_ROWNORM), as a stream: L = 8 / 16 / 32 / 64 lanes share a row (C = 320 / 640 / 1280 / 2560: five 16-byte
// vectors per lane, lane i of the group reads vectors i, i + L, ...: L x 16 contiguous bytes per instruction and
// group), so a wave-instruction covers 64 / L rows and every lane does useful work (layernorm_rows_kernel: one wave per
// row = 40 of 64 lanes at C = 320, and two 64-lane butterflies per 640-byte row).  R row sets per wave are loaded
// before the first reduction; the in-group sums are DPP steps inside a 16-lane row (quad permutes, half-mirror,
// mirror) plus at most two cross-row shuffles.  Same two-pass arithmetic (mean, then centred squares) as the row kernels.
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
  return v + __builtin_bit_cast(float, t);
}
template <int L>
__device__ __forceinline__ float lane_group_sum(float v) {
  v = dpp_add<0xB1>(v);                                  // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);                                  // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);                                 // row_half_mirror: lane i <-> 7 - i of its 8
  if constexpr (L >= 16) v = dpp_add<0x140>(v);          // row_mirror: lane i <-> 15 - i of its 16
  if constexpr (L >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (L >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

template <int L, int R>
__global__ __launch_bounds__(256) void ln_stats_kernel(const half_t* __restrict__ x, long ldx, int rows, int C, float eps,
                                                        float* __restrict__ stats, int rpb, long x_bs) {
  constexpr int V = 5;                                   // vectors per lane: C <= 8 * 5 * L
  constexpr int RPW = 64 / L;                            // rows per wave-instruction
  const int lane = threadIdx.x & 63, li = lane % L, lr = lane / L;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (RPW * R) + lr;
  const int nvec = C / 8;
  half8_t h[R][V];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = row0 + r * RPW;
    const int rc = row < rows ? row : rows - 1;          // a valid address either way; rows past the end are not stored
    const int bb = rc / rpb, rr = rc - bb * rpb;
    const half_t* xr = x + bb * x_bs + (long)rr * ldx;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int v = li + j * L;
      h[r][j] = v < nvec ? *reinterpret_cast<const half8_t*>(xr + v * 8) : (half8_t){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  const float inv_c = 1.f / C;
  float mean[R], var[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)h[r][j][e];              // lanes past nvec hold zeros
    mean[r] = s;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) mean[r] = lane_group_sum<L>(mean[r]) * inv_c;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < V; ++j) asm volatile("" : "+v"(h[r][j]));      // keep the rows fp16 between the passes (registers)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j)
      if (li + j * L < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dlt = (float)h[r][j][e] - mean[r];
          q += dlt * dlt;
        }
      }
    var[r] = q;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) var[r] = lane_group_sum<L>(var[r]) * inv_c;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = row0 + r * RPW;
    if (li == 0 && row < rows) *reinterpret_cast<float2*>(stats + (long)row * 2) = make_float2(mean[r], rsqrtf(var[r] + eps));
  }
}

// dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat))
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const half_t* __restrict__ gy, long ldgy, const half_t* __restrict__ x, long ldx, half_t* gx,
    long ldgx, int rows, int C, const float* __restrict__ gamma, const float* __restrict__ stats,
    int rpb, long gy_bs, long x_bs, long gx_bs, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int bb = row / rpb, rr = row - bb * rpb;
  const half_t* gr = gy + bb * gy_bs + (long)rr * ldgy;
  const half_t* xr = x + bb * x_bs + (long)rr * ldx;
  half_t* dr = gx + bb * gx_bs + (long)rr * ldgx;
  const float mean = stats[(long)row * 2], rstd = stats[(long)row * 2 + 1];
  const int nvec = C / 8;
  float dxh[LN_MAXV][8], xh[LN_MAXV][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int v = lane + j * 64;
    if (v < nvec) {
      half8_t hx = *reinterpret_cast<const half8_t*>(xr + v * 8);
      half8_t hg = *reinterpret_cast<const half8_t*>(gr + v * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = ((float)hx[e] - mean) * rstd;
        float d = (float)hg[e] * gamma[v * 8 + e];
        xh[j][e] = a;
        dxh[j][e] = d;
        s1 += d;
        s2 += d * a;
      }
    }
  }
  s1 = wave_sum(s1) / C;
  s2 = wave_sum(s2) / C;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int v = lane + j * 64;
    if (v < nvec) {
      half8_t o;
      if (accumulate) o = *reinterpret_cast<const half8_t*>(dr + v * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float dx = rstd * (dxh[j][e] - s1 - xh[j][e] * s2);
        if (accumulate) dx += (float)o[e];
        o[e] = (half_t)dx;
      }
      *reinterpret_cast<half8_t*>(dr + v * 8) = o;
    }
  }
}

// Workgroups per image of the apply passes: about 1024 workgroups in total, at least one full
// GN_UNROLL trip of pixels per thread.
int g_gn_apply_wgs = 1024;        // option "gn_apply_wgs" (tools): workgroups per launch the apply passes aim at
int gn_apply_blocks(int B, int HW, int C) {
  const int nvec = C / 8;
  const int pl = nvec < 256 ? 256 / nvec : 1;
  int px = GN_UNROLL * pl;
  const int want = (int)(((long)HW * B + g_gn_apply_wgs - 1) / g_gn_apply_wgs);
  if (px < want) px = want;
  int n = (HW + px - 1) / px;
  return n < 1 ? 1 : n;
}

}  // namespace

void lgd_gn_set_fused_hw(int hw) { g_gn_fused_hw = hw; }
void lgd_ln_set_stream(int on) { g_ln_stream = on; }        // lgd_set_option("ln_stream", 0 | 1) (attn.hip)
void lgd_gn_set_slab(int on) { g_gn_slab = on; }
void lgd_gn_set_apply_wgs(int n) { g_gn_apply_wgs = n; }            // lgd_set_option("gn_slab", 0 | 1) (attn.hip)    // lgd_set_option("gn_fused", hw) (attn.hip)

extern "C" int lgd_groupnorm_f16(const void* x0, const void* x1, int c0, int c1, int B, int HW,
                                 int G, float eps, const float* gamma, const float* beta, int silu,
                                 void* y, float* part, int nchunk, float* stats, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  const int C = c0 + c1;
  if (G > 64 || C > GN_MAXC || (C % G) || (c0 % 8) || (c1 % 8) || nchunk < 1) return LGD_ERR_ARG;
  if (c1 > 0 && !x1) return LGD_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {
    // one launch when a workgroup can hold its (image, kg groups) slab in registers
    const int cpg = C / G;
    int kg = 1;
    while ((kg * cpg) % 8) kg *= 2;
    const int W = kg * cpg, nv = W / 8;
    if (HW <= g_gn_fused_hw && kg <= 8 && (G % kg) == 0 && W <= 256) {
      const int pl = 256 / nv, npx = (HW + pl - 1) / pl;
      if (npx <= 32) {
#define GN_FUSED(P)                                                                                                 \
  hipLaunchKernelGGL(gn_fused_kernel<P>, dim3(G / kg, B), dim3(256), 0, st, (const half_t*)x0, (const half_t*)x1,    \
                     c0, c1, HW, G, eps, gamma, beta, silu, (half_t*)y, stats, kg)
        if (npx <= 4) GN_FUSED(4);
        else if (npx <= 8) GN_FUSED(8);
        else if (npx <= 16) GN_FUSED(16);
        else GN_FUSED(32);
#undef GN_FUSED
        return lgd_check_launch();
      }
    }
  }
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, B), dim3(256), 0, st, (const half_t*)x0,
                     (const half_t*)x1, c0, c1, HW, G, part, nchunk);
  const int napply = gn_apply_blocks(B, HW, C);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(napply, B), dim3(256), 0, st, (const half_t*)x0,
                     (const half_t*)x1, c0, c1, HW, G, eps, gamma, beta, silu, (half_t*)y, part,
                     nchunk, stats, napply);
  return lgd_check_launch();
}

extern "C" int lgd_groupnorm_bwd_f16(const void* gy, const void* x0, const void* x1, int c0, int c1,
                                     int B, int HW, int G, const float* gamma, const float* beta,
                                     int silu, const float* stats, void* gx0, void* gx1, float* part,
                                     int nchunk, int accumulate, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  const int C = c0 + c1;
  if (G > 64 || C > GN_MAXC || (C % G) || (c0 % 8) || (c1 % 8) || nchunk < 1) return LGD_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {
    int kg, nv;
    if (g_gn_slab && gn_slab_geometry(C, G, kg, nv)) {
#define GN_BWD_SLAB_(NT, P, S)                                                                                      \
  hipLaunchKernelGGL((gn_bwd_slab_kernel<NT, P, S>), dim3(G / kg, B), dim3(NT), 0, st, (const half_t*)gy,             \
                     (const half_t*)x0, (const half_t*)x1, c0, c1, HW, G, gamma, beta, stats, (half_t*)gx0,           \
                     (half_t*)gx1, accumulate, kg)
#define GN_BWD_SLAB(NT, P) do { if (silu) GN_BWD_SLAB_(NT, P, true); else GN_BWD_SLAB_(NT, P, false); } while (0)
      const long slab_bytes = 2L * HW * (8 * nv) * 2;             // x and gy of one workgroup
      if (slab_bytes <= 96 * 1024) {
        if (HW <= 8 * (256 / nv)) { GN_BWD_SLAB(256, 8); return lgd_check_launch(); }
        if (HW <= 11 * (512 / nv)) { GN_BWD_SLAB(512, 11); return lgd_check_launch(); }
      }
#undef GN_BWD_SLAB
#undef GN_BWD_SLAB_
    }
  }
  hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(nchunk, B), dim3(256), 0, st, (const half_t*)gy,
                     (const half_t*)x0, (const half_t*)x1, c0, c1, HW, G, gamma, beta, silu, stats,
                     part, nchunk);
  const int napply = gn_apply_blocks(B, HW, C);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(napply, B), dim3(256), 0, st, (const half_t*)gy,
                     (const half_t*)x0, (const half_t*)x1, c0, c1, HW, G, gamma, beta, silu, stats,
                     (half_t*)gx0, (half_t*)gx1, part, nchunk, accumulate, napply);
  return lgd_check_launch();
}

extern "C" int lgd_layernorm_f16(const void* x, int64_t ldx, void* y, int64_t ldy, int rows, int C,
                                 float eps, const float* gamma, const float* beta, float* stats,
                                 int rows_per_batch, int64_t x_bs, int64_t y_bs, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if ((C % 8) || C > 64 * 8 * LN_MAXV || rows < 1) return LGD_ERR_ARG;
  if (!y && (!stats || C > (g_ln_stream ? 64 * 5 * 8 : 192 * 8))) return LGD_ERR_ARG;      // statistics-only form: stats required
  if (rows_per_batch < 1) rows_per_batch = rows;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const half_t* xp = (const half_t*)x;
  half_t* yp = (half_t*)y;
  const int nvec = C / 8;
  // statistics only: the streaming kernel (lane groups per row) once the map is large enough to keep every CU streaming
  // (measured, 40 launches in one graph: 65536 x 320 16.4 -> 10.0 us, 32768 x 320 8.7 -> 4.9, 16384 x 640 7.1 -> 5.0;
  // below ~8 M elements the one-wave-per-row kernels with their 4x more workgroups win: 4096 x 1280 3.7 vs 4.8 us)
  if (!y && g_ln_stream && ((long)rows * C >= (8L << 20) || nvec > 192)) {
#define LGD_LN_STATS(L, R)                                                                                         \
  hipLaunchKernelGGL((ln_stats_kernel<L, R>), dim3((rows + 4 * (64 / L) * R - 1) / (4 * (64 / L) * R)), dim3(256), 0, st, xp, \
                     (long)ldx, rows, C, eps, stats, rows_per_batch, (long)x_bs)
    if (nvec <= 40) LGD_LN_STATS(8, 4);
    else if (nvec <= 80) LGD_LN_STATS(16, 4);
    else if (nvec <= 160) LGD_LN_STATS(32, 4);
    else LGD_LN_STATS(64, 2);
#undef LGD_LN_STATS
    return lgd_check_launch();
  }
#define LGD_LN_LAUNCH(MAXV, ROWS)                                                                   \
  hipLaunchKernelGGL((layernorm_rows_kernel<MAXV, ROWS>), dim3((rows + 4 * ROWS - 1) / (4 * ROWS)), \
                     dim3(256), 0, st, xp, (long)ldx, yp, (long)ldy, rows, C, eps, gamma, beta,      \
                     stats, rows_per_batch, (long)x_bs, (long)y_bs)
  if (nvec <= 64) LGD_LN_LAUNCH(1, 4);
  else if (nvec <= 128) LGD_LN_LAUNCH(2, 4);
  else if (nvec <= 192) LGD_LN_LAUNCH(3, 2);
  else
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, xp, (long)ldx, yp,
                       (long)ldy, rows, C, eps, gamma, beta, stats, rows_per_batch, (long)x_bs,
                       (long)y_bs);
#undef LGD_LN_LAUNCH
  return lgd_check_launch();
}

extern "C" int lgd_layernorm_bwd_f16(const void* gy, int64_t ldgy, const void* x, int64_t ldx,
                                     void* gx, int64_t ldgx, int rows, int C, const float* gamma,
                                     const float* stats, int rows_per_batch, int64_t gy_bs,
                                     int64_t x_bs, int64_t gx_bs, int accumulate, void* stream) {
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if ((C % 8) || C > 64 * 8 * LN_MAXV || rows < 1 || !stats) return LGD_ERR_ARG;
  if (rows_per_batch < 1) rows_per_batch = rows;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st,
                     (const half_t*)gy, (long)ldgy, (const half_t*)x, (long)ldx, (half_t*)gx,
                     (long)ldgx, rows, C, gamma, stats, rows_per_batch, (long)gy_bs, (long)x_bs,
                     (long)gx_bs, accumulate);
  return lgd_check_launch();
}
