// Probe of ds_read_b64_tr_b16 (gfx950): which LDS elements land in which lane / element, for a [row][16 x 16-bit] image.
// Each 16-lane group reads a block of 4 rows x 16 columns; lane 4q + p supplies the address of row q, columns 4p .. 4p+3
// (guide T10).  LDS element value = row * 16 + col.  Prints, for lanes 0..15 of group 0, the 4 received values.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 16];
  for (int i = threadIdx.x; i < 64 * 16; i += 64) lds[i] = (short)i;  // value = row * 16 + col
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, l = lane & 15, q = l >> 2, p = l & 3;
  // group g reads rows 8g .. 8g+3
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (8 * g + q) * 16 + 4 * p));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int lane = 0; lane < 64; lane += 1) {
    if ((lane & 15) < 16 && (lane >> 4) < 2) {
      printf("lane %2d:", lane);
      for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[lane * 4 + e] / 16, h[lane * 4 + e] % 16);
      printf("\n");
    }
  }
  return 0;
}
