// Micro-benchmark: store throughput of ONE CU (global_store_dwordx4), contiguous 1 KB per instruction vs the epilogue
// pattern (8 lines of 128 bytes at a pixel stride), as a function of the number of waves storing.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(1024) void st(char* __restrict__ buf, size_t per_wave_bytes, int iters, int stride,
                                           unsigned long long* ticks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  char* base = buf + ((size_t)blockIdx.x * nw + wave) * per_wave_bytes;
  const u32x4 v = {1u, 2u, 3u, (unsigned)lane};
  // MODE 0: lane -> 16 contiguous bytes (1 KB per instruction); MODE 1: 8 lanes per 128-byte line, lines `stride` apart
  const size_t loff = MODE == 0 ? (size_t)lane * 16 : (size_t)(lane >> 3) * stride + (lane & 7) * 16;
  const size_t step = MODE == 0 ? 1024 : (size_t)8 * stride;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) *reinterpret_cast<u32x4*>(base + (size_t)it * step + loff) = v;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 256;  // 16 = the burst of one epilogue (16 KB per wave)
  const size_t per_wave = (size_t)256 * 8 * 512;  // room for MODE 1 at stride 512
  char* buf; unsigned long long* ticks;
  hipMalloc(&buf, per_wave * 256 * 16); hipMalloc(&ticks, 8 * 1024);
  for (int blocks : {8, 256})
    for (int waves : {1, 4, 8, 16})
      for (int mode : {0, 1, 2}) {
        const int stride = mode == 2 ? 512 : 128;
        for (int rep = 0; rep < 2; ++rep) {
          if (mode == 0) hipLaunchKernelGGL(st<0>, dim3(blocks), dim3(waves * 64), 0, 0, buf, per_wave, iters, stride, ticks);
          else hipLaunchKernelGGL(st<1>, dim3(blocks), dim3(waves * 64), 0, 0, buf, per_wave, iters, stride, ticks);
        }
        hipDeviceSynchronize();
        unsigned long long h[256];
        hipMemcpy(h, ticks, 8 * blocks, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
        printf("blocks %3d waves %2d mode %d (stride %3d): %6.1f B/tick/CU, %5.0f ticks per store per wave\n", blocks, waves, mode,
               stride, (double)iters * waves * 1024 / avg, avg / iters);
      }
  return 0;
}
