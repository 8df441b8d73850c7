// Matrix-pipe time of one full-precision 16x16x32 product step under a full-chip load, random operands, for the arithmetic
// schemes of tools/emulate_fp8_cross.py:
//   A  bf16x3            3 x v_mfma_f32_16x16x32_bf16 per step (the shipped path)
//   B  f16 + fp6         1 x v_mfma_f32_16x16x32_f16 + 1 x v_mfma_scale_f32_16x16x128_f8f6f4 (fp6 e2m3) per step: the two cross
//                        terms of ONE step use K = 64 of the instruction's 128
//   C  f16 + fp6 paired  2 x f16 + 1 x fp6 per TWO steps (cross terms of two steps stacked along K)
//   D  f16 + fp8 paired  2 x f16 + 1 x f8f6f4 (fp8 e4m3) per two steps
// Reports ns per step and wave, the in-kernel clock, and steps/s of the chip (8 waves per CU).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_mix mfma_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MIX>
__global__ __launch_bounds__(512) void k(const int* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* clk) {
  // operands: 4 + 4 sixteen-byte fragments (bit patterns are finite in bf16 and in fp16) and 2 + 2 fp6 / fp8 fragments
  union { int w[4]; bf16x8 b; f16x8 h; } a[4], b[4];
  i32x8 qa[2], qb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[i].w[j] = src[(threadIdx.x * 37 + i * 4 + j) & 4095]; b[i].w[j] = src[(threadIdx.x * 53 + 64 + i * 4 + j) & 4095]; }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { qa[i][j] = src[4096 + ((threadIdx.x * 11 + i * 8 + j) & 4095)]; qb[i][j] = src[4096 + ((threadIdx.x * 13 + 32 + i * 8 + j) & 4095)]; }
  const int sc = 127;  // E8M0 scale 2^0
  f32x4 c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    // one iteration = 16 steps per wave: 8 accumulator chains x 2 steps (products of a chain issue round-robin over the chains)
    if constexpr (MIX == 0) {
#pragma unroll
      for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + u) & 3].b, b[(i + 2 * u) & 3].b, c[i], 0, 0, 0);
    } else if constexpr (MIX == 1) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + u) & 3].h, b[(i + 2 * u) & 3].h, c[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qa[(i + u) & 1], qb[i & 1], c[i], 2, 2, 0, sc, 0, sc);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + u) & 3].h, b[(i + 2 * u) & 3].h, c[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        c[i] = MIX == 2 ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qa[i & 1], qb[(i >> 1) & 1], c[i], 2, 2, 0, sc, 0, sc)
                        : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qa[i & 1], qb[(i >> 1) & 1], c[i], 0, 0, 0, sc, 0, sc);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float acc_out = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc_out += c[i][0] + c[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = acc_out;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  int* src; float* out; unsigned long long* clk;
  hipMalloc(&src, 8192 * 4); hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&clk, 16);
  static int h[8192];
  srand(1);
  for (int i = 0; i < 4096; ++i) {  // two 16-bit values in +-[0.03, 0.06] as bf16, ~+-[1, 2] as fp16: finite in both readings
    const unsigned lo = 0x3c00u + (rand() & 0x3ff) + ((rand() & 1u) << 15), hi = 0x3c00u + (rand() & 0x3ff) + ((rand() & 1u) << 15);
    h[i] = (int)(lo | (hi << 16));
  }
  for (int i = 4096; i < 8192; ++i) h[i] = (int)(((unsigned)rand() << 16) ^ (unsigned)rand()) & 0x77777777;  // random fp6 / fp8 bits (fp8: no NaN pattern 0x7f / 0xff)
  hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const char* names[4] = {"A bf16x3", "B f16+fp6 (K=64 of 128 used)", "C f16+fp6 paired", "D f16+fp8 paired"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mix = 0; mix < 4; ++mix) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      for (int l = 0; l < 5; ++l) {
        if (mix == 0) hipLaunchKernelGGL(k<0>, dim3(cus), dim3(512), 0, 0, src, out, iters, clk);
        else if (mix == 1) hipLaunchKernelGGL(k<1>, dim3(cus), dim3(512), 0, 0, src, out, iters, clk);
        else if (mix == 2) hipLaunchKernelGGL(k<2>, dim3(cus), dim3(512), 0, 0, src, out, iters, clk);
        else hipLaunchKernelGGL(k<3>, dim3(cus), dim3(512), 0, 0, src, out, iters, clk);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
      const double steps_wave = 16.0 * iters * 5;  // per wave over the five launches
      const double clock = (double)c[0] / ((double)c[1] / 100e6) / 1e9;
      printf("{\"mix\": \"%s\", \"ms\": %.3f, \"ns_per_step_and_wave\": %.2f, \"cycles_per_step_and_simd\": %.1f, \"clock_ghz\": %.3f, \"algorithmic_tflops\": %.1f}\n",
             names[mix], ms, ms * 1e6 / steps_wave, ms * 1e6 / steps_wave * clock / 2.0, clock,
             2.0 * 8192.0 * steps_wave * 8.0 * cus / (ms * 1e-3) / 1e12);
    }
  return 0;
}
