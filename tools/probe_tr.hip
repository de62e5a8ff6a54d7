#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half_t;
typedef half_t half4_t __attribute__((ext_vector_type(4)));
typedef short short4_t __attribute__((ext_vector_type(4)));
__global__ void probe(short4_t* out) {
  __shared__ half_t lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (half_t)(float)i;   // value = linear index (exact up to 2048)
  __syncthreads();
  const int lane = threadIdx.x;
  // hypothesis: 16-lane group reads a 4(row) x 16(col) block row-wise; lane j supplies row j/4, cols 4(j%4)..+3; row stride 64 halfs
  const int g = lane >> 4, j = lane & 15;
  const half_t* p = lds + (g * 4 + j / 4) * 64 + 4 * (j % 4);
  short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(p));
  out[lane] = v;
}
int main() {
  short4_t* d; (void)hipMalloc(&d, 64 * 8);
  probe<<<1, 64>>>(d);
  unsigned short h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) { _Float16 f; __builtin_memcpy(&f, &h[l*4+e], 2); printf(" %5d", (int)(float)f); } printf("\n"); }
  return 0;
}
