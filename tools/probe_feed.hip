// GPU-box probe (round 4): how many bytes per clock one CU can pull from its XCD's L2 into the workgroup, by path:
//   0 = global_load_lds_dwordx4 (LDS-DMA, what gemm_pipe_kernel uses)      1 = global_load_dwordx4 into registers only
//   2 = global_load_dwordx4 into registers + ds_write_b128 (the classic staging)
// One workgroup per CU (LDS sized to forbid a second), NW waves, each instruction = 64 lanes x 16 B in the GEMM's
// operand pattern (8 lanes cover a 128-byte K segment of a row, 8 rows per instruction, row stride `ld` bytes), DEPTH
// instructions in flight per wave.  Source: one window of WINDOW_KB (default 16 MB: beyond an XCD's 4 MB L2, served by
// the Infinity Cache; 1024 = every XCD's L2 holds the whole window) that all workgroups walk.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_feed tools/probe_feed.hip && /tmp/probe_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
using glb_ptr_t = const __attribute__((address_space(1))) void*;
using lds_ptr_t = __attribute__((address_space(3))) void*;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void feed(const char* src, long window, int ld, int iters, float* sink, long* cyc) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const long boff = (long)blockIdx.x * 65536;
  const int lrow = lane >> 3, seg = lane & 7;
  f4 acc = {0, 0, 0, 0};
  f4 r[DEPTH];
  const long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      // wave `wid`, instruction (it, d): 8 rows of a 256-row tile, walking K 64 halfs per iteration
      const long row = (long)((d * nw + wid) * 8 + lrow);
      const char* p = src + (boff + row * ld + ((long)it * 128) % ld + seg * 16) % window;
      char* l = smem + ((d * nw + wid) * 1024);
      if (MODE == 0) __builtin_amdgcn_global_load_lds((glb_ptr_t)p, (lds_ptr_t)l, 16, 0, 0);
      else r[d] = *reinterpret_cast<const f4*>(p);
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE >= 1) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        if (MODE == 2) *reinterpret_cast<f4*>(smem + (d * nw + wid) * 1024 + lane * 16) = r[d];
        else acc += r[d];
      }
    }
  }
  const long t1 = clock64();
  if (MODE == 2) { __syncthreads(); acc += *reinterpret_cast<f4*>(smem + threadIdx.x * 16); }
  if (MODE == 0) { __syncthreads(); acc += *reinterpret_cast<f4*>(smem + threadIdx.x * 16); }
  if (acc[0] == 12345.f) sink[0] = acc[1];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int DEPTH>
void run(const char* name, const char* src, long window, int ld, int nw, float* sink, long* cyc) {
  const int iters = 2000, grid = 256;
  const int smem = 96 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&feed<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  feed<MODE, DEPTH><<<grid, 64 * nw, smem>>>(src, window, ld, 100, sink, cyc);
  (void)hipEventRecord(e0);
  feed<MODE, DEPTH><<<grid, 64 * nw, smem>>>(src, window, ld, iters, sink, cyc);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes_per_cu = (double)iters * DEPTH * nw * 1024;
  printf("%-28s ld %6d  %d waves x %2d in flight: %6.1f GB/s per CU, %5.2f TB/s chip  (%.1f B/clk/CU @2.4 GHz)\n", name, ld, nw, DEPTH,
         bytes_per_cu / (ms * 1e-3) / 1e9, bytes_per_cu * grid / (ms * 1e-3) / 1e12, bytes_per_cu / (ms * 1e-3) / 2.4e9);
}

int main() {
  const long window = (getenv("WINDOW_KB") ? atol(getenv("WINDOW_KB")) : 16384L) << 10;
  char* src; float* sink; long* cyc;
  (void)hipMalloc(&src, window + (1 << 20)); (void)hipMemset(src, 0, window + (1 << 20));
  (void)hipMalloc(&sink, 64); (void)hipMalloc(&cyc, 256 * 8);
  for (int ld : {640, 2560, 23040}) {
    for (int nw : {4, 8}) {
      run<0, 4>("LDS-DMA", src, window, ld, nw, sink, cyc);
      run<0, 8>("LDS-DMA", src, window, ld, nw, sink, cyc);
      run<1, 4>("global_load -> VGPR", src, window, ld, nw, sink, cyc);
      run<1, 8>("global_load -> VGPR", src, window, ld, nw, sink, cyc);
      run<2, 8>("global_load -> VGPR -> LDS", src, window, ld, nw, sink, cyc);
    }
  }
  return 0;
}
