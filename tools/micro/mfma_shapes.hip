// Sustained rate and clock of the two bf16 MFMA shapes under a full-chip load with random operands:
//   v_mfma_f32_16x16x32_bf16 (8192 MACs, 4 + 4 operand registers)  vs  v_mfma_f32_32x32x16_bf16 (16384 MACs, 4 + 4).
// The kernels of this repo use the 16x16x32 shape; the chip is power-limited under them (tools/mfma_clock.py: 1.85 GHz).
// Half the operand-register traffic per MAC might buy clock.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shapes mfma_shapes.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const bf16x8* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x * 8 + i) & 4095]; b[i] = src[(threadIdx.x * 8 + 4 + i) & 4095]; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  float acc_out = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + u) & 3], b[i & 3], c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc_out += c[i][0] + c[i][3];
  } else {
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[i & 3], c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_out += c[i][0] + c[i][15];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 512 + threadIdx.x] = acc_out;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  (void)lane;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  bf16x8* src; float* out; unsigned long long* clk;
  hipMalloc(&src, 4096 * sizeof(bf16x8)); hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&clk, 16);
  unsigned short h[4096 * 8];
  srand(1);
  for (int i = 0; i < 4096 * 8; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));  // random bf16 in +-[0.03, 0.06]
  hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  for (int rep = 0; rep < 2; ++rep)
    for (int shape : {16, 32}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      for (int l = 0; l < 5; ++l) {
        if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(cus), dim3(512), 0, 0, src, out, iters, clk);
        else hipLaunchKernelGGL(k<32>, dim3(cus), dim3(512), 0, 0, src, out, iters, clk);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
      const double macs_per = shape == 16 ? 8192.0 * 32 : 16384.0 * 16;  // MFMAs per iteration and wave x MACs
      const double flops = 2.0 * macs_per * iters * 8.0 * cus * 5;
      printf("{\"shape\": \"%s\", \"iters\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"clock_ghz\": %.3f}\n", shape == 16 ? "16x16x32" : "32x32x16", iters, ms,
             flops / (ms * 1e-3) / 1e12, (double)c[0] / ((double)c[1] / 100e6) / 1e9);
    }
  return 0;
}
