// Micro-benchmark: how many bytes per cycle can ONE CU pull from an L2-resident buffer with global_load_dwordx4,
// as a function of the number of waves issuing and of the loads each keeps in flight?  (Design input for the movers of
// conv_mfma_sp.hip.)  Build: hipcc --offload-arch=gfx950 -O3 l2_stream.hip -o l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(1024) void stream(const char* __restrict__ buf, size_t bytes, int iters, unsigned* sink,
                                               unsigned long long* ticks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // every block streams the same `bytes` (weights-like sharing): wave w takes pieces w, w + nw, ...
  const size_t npieces = bytes / 1024;
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  size_t p = wave;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      v[i] = *reinterpret_cast<const u32x4*>(buf + (p & (npieces - 1)) * 1024 + lane * 16);  // npieces is a power of two
      p += nw;
    }
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) acc ^= v[i];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc.x == 0x12345678u) sink[0] = acc.y;
  (void)npieces;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const size_t bytes = 2u << 20;  // 2 MB: L2-resident, far larger than the 32 KB L1
  char* buf; unsigned* sink; unsigned long long* ticks;
  hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&sink, 4); hipMalloc(&ticks, 8 * 1024);
  const int iters = 64;
  for (int blocks : {8, 256}) {
    for (int waves : {1, 2, 4, 8, 12, 16}) {
      for (int depth : {4, 16}) {
        auto k = depth == 4 ? stream<4> : stream<16>;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(waves * 64), 0, 0, buf, bytes, iters, sink, ticks);  // warm
        hipLaunchKernelGGL(k, dim3(blocks), dim3(waves * 64), 0, 0, buf, bytes, iters, sink, ticks);
        hipDeviceSynchronize();
        unsigned long long h[256];
        hipMemcpy(h, ticks, 8 * blocks, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
        const double moved = (double)iters * depth * waves * 1024;
        printf("blocks %3d waves %2d depth %2d: %8.0f ticks, %6.1f B/tick/CU, %5.0f ticks per load per wave\n", blocks, waves,
               depth, avg, moved / avg, avg / (iters * depth));
      }
    }
  }
  return 0;
}
