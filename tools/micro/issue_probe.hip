// How fast does ONE wave per SIMD issue the matrix instructions of the FL arithmetic, and what slows it?
//   mode 0  16 x v_mfma_f32_16x16x32_f16 per slot, register operands, 16 accumulators          (the pipe's own rate)
//   mode 1  12 x f16 + 4 x v_mfma_scale_f32_16x16x128_f8f6f4 (fp6) per slot, register operands
//   mode 2  as 1, the B fragments of a slot read from LDS one slot ahead (3 ds_read_b128 per slot)
//   mode 3  as 2, plus a partner wave per SIMD that runs packed-fp16 VALU work and ds_write_b128 (a stand-in for a mover)
//   mode 4  as 2, plus a partner wave that only stores to LDS
//   mode 5  as 2, plus a partner wave that only runs VALU work
//   mode 6  as 2, the three reads one behind every fourth matrix instruction
//   mode 7 / 8  as 2 with the matrix instructions as inline assembly, accumulators tied in arch VGPRs / in AccVGPRs
// Prints cycles per matrix instruction (s_memtime of wave 0 of block 0) and the in-kernel clock.
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip ; run: ./issue_probe [slots]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const int* __restrict__ src, float* __restrict__ out, int slots, unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < 16 * 1024; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = src[i & 8191];
  __syncthreads();
  const bool partner = wid >= 4;
  if (partner) {
    if (MODE < 3 || MODE >= 6) return;
    f16x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(f16x2, src[(tid + i) & 8191]);
    u32x4 w = {1u, 2u, 3u, 4u};
    char* dst = lds + 32 * 1024 + (wid - 4) * 4096 + lane * 16;
    for (int s = 0; s < 2 * slots; ++s) {
      if (MODE == 3 || MODE == 5) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = v[i] * v[(i + 1) & 7] + v[(i + 3) & 7];
      }
      if (MODE == 3 || MODE == 4) {
        w[0] = __builtin_bit_cast(unsigned, v[0]) + s;
        *reinterpret_cast<u32x4*>(dst) = w;
        *reinterpret_cast<u32x4*>(dst + 1024) = w;
        *reinterpret_cast<u32x4*>(dst + 2048) = w;
      }
    }
    float a = 0.f;
    for (int i = 0; i < 8; ++i) a += (float)v[i][0];
    out[blockIdx.x * 512 + tid] = a;
    return;
  }
  u32x4 wa[12];
  i32x8 wq[4];
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) wa[i][j] = (unsigned)src[(tid * 37 + i * 4 + j) & 4095];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) wq[i][j] = j == 6 ? 127 : src[4096 + ((tid * 11 + i * 8 + j) & 4095)];
  f32x4 c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* base = lds + (wid * 8192) + lane * 16;
  u32x4 af = *reinterpret_cast<const u32x4*>(base);
  i32x8 xq;
  {
    const u32x4 a = *reinterpret_cast<const u32x4*>(base + 1024), b = *reinterpret_cast<const u32x4*>(base + 2048);
    xq = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], 127, 0};
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < slots; ++s) {
    u32x4 afn = af;
    i32x8 xn = xq;
    if (MODE >= 2 && MODE != 6) {  // (modes 7 / 8: as mode 2)
      const char* p = base + ((s & 1) ? 3072 : 0);
      afn = *reinterpret_cast<const u32x4*>(p);
      const u32x4 a = *reinterpret_cast<const u32x4*>(p + 1024), b = *reinterpret_cast<const u32x4*>(p + 2048);
      xn = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], 127, 0};
    }
    asm volatile("" ::: "memory");
    u32x4 ra = {0u, 0u, 0u, 0u}, rb = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (MODE == 7) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(wa[i]), "v"(af));  // accumulator tied, arch VGPRs
      else if (MODE == 8) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c[i]) : "v"(wa[i]), "v"(af));  // ... in AccVGPRs
      else
      c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wa[i]), __builtin_bit_cast(f16x8, af), c[i], 0, 0, 0);
      if (MODE == 6 && (i == 1 || i == 5 || i == 9)) {  // mode 6: the three reads of mode 2, one behind every fourth instruction
        const char* p = base + ((s & 1) ? 3072 : 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i == 1) afn = *reinterpret_cast<const u32x4*>(p);
        if (i == 5) ra = *reinterpret_cast<const u32x4*>(p + 1024);
        if (i == 9) rb = *reinterpret_cast<const u32x4*>(p + 2048);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE == 6) xn = i32x8{(int)ra[0], (int)ra[1], (int)ra[2], (int)ra[3], (int)rb[0], (int)rb[1], 127, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE == 0) c[12 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wa[i]), __builtin_bit_cast(f16x8, af), c[12 + i], 0, 0, 0);
      else if (MODE == 7 || MODE == 8) {
        typedef int i32x6 __attribute__((ext_vector_type(6)));
        const i32x6 w6 = __builtin_shufflevector(wq[i], wq[i], 0, 1, 2, 3, 4, 5), x6 = __builtin_shufflevector(xq, xq, 0, 1, 2, 3, 4, 5);
        if (MODE == 7) asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(c[12 + i]) : "v"(w6), "v"(x6), "v"(wq[i][6]), "v"(xq[6]));
        else asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+a"(c[12 + i]) : "v"(w6), "v"(x6), "v"(wq[i][6]), "v"(xq[6]));
      }
      else c[12 + i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wq[i], xq, c[12 + i], 2, 2, 0, wq[i][6], 0, xq[6]);
    }
    af = afn;
    xq = xn;
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) a += c[i][0] + c[i][3];
  out[blockIdx.x * 512 + tid] = a;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}


// kernel 2: `NR` ds_read_b128 per slot of 16 f16 instructions (each read becomes the B operand of 16 / NR instructions of the
// NEXT slot), `WPS` such waves per SIMD, `NV` extra v_mov per slot
template <int NR, int WPS, int NV>
__global__ __launch_bounds__(256 * WPS) void k2(const int* __restrict__ src, float* __restrict__ out, int slots, unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < 16 * 1024; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = src[i & 8191];
  __syncthreads();
  u32x4 wa[16];  // (distinct operands: identical products would be merged by the compiler)
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) wa[i][j] = (unsigned)src[(tid * 37 + i * 4 + j) & 4095];
  f32x4 c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* base = lds + (wid * 4096) + lane * 16;
  constexpr int NB = NR > 0 ? NR : 1;
  u32x4 b[NB], bn[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) b[i] = *reinterpret_cast<const u32x4*>(base + i * 1024);
  unsigned mv[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < slots; ++s) {
#pragma unroll
    for (int i = 0; i < NB; ++i) bn[i] = NR > 0 ? *reinterpret_cast<const u32x4*>(base + ((s & 1) ? 2048 : 0) + (i & 1) * 1024) : b[i];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wa[i]), __builtin_bit_cast(f16x8, b[i % NB]), c[i], 0, 0, 0);
      if (i < NV) asm volatile("v_mov_b32 %0, %1" : "=v"(mv[i & 7]) : "v"(mv[(i + 1) & 7]));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i] = bn[i];
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float a = (float)mv[0];
#pragma unroll
  for (int i = 0; i < 16; ++i) a += c[i][0] + c[i][3];
  out[blockIdx.x * 1024 + tid] = a;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int NR, int WPS, int NV>
void run2(int blocks, int slots, const int* src, float* out, unsigned long long* clk) {
  unsigned long long h[2];
  k2<NR, WPS, NV><<<blocks, 256 * WPS>>>(src, out, 64, clk);
  hipDeviceSynchronize();
  k2<NR, WPS, NV><<<blocks, 256 * WPS>>>(src, out, slots, clk);
  hipDeviceSynchronize();
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("{\"reads_per_16\": %d, \"waves_per_simd\": %d, \"vmov_per_16\": %d, \"blocks\": %d, \"cycles_per_slot_and_wave\": %.1f, \"simd_cycles_per_matrix_instruction\": %.2f, \"clock_ghz\": %.3f}\n",
         NR, WPS, NV, blocks, (double)h[0] / slots, (double)h[0] / ((double)slots * 16.0 * WPS), (double)h[0] / ((double)h[1] * 10.0));
}

template <int MODE>
void run(const char* name, int blocks, int slots, const int* src, float* out, unsigned long long* clk) {
  unsigned long long h[2];
  k<MODE><<<blocks, 512>>>(src, out, 64, clk);
  hipDeviceSynchronize();
  k<MODE><<<blocks, 512>>>(src, out, slots, clk);
  hipDeviceSynchronize();
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("{\"mode\": \"%s\", \"blocks\": %d, \"cycles_per_matrix_instruction\": %.2f, \"clock_ghz\": %.3f}\n", name, blocks,
         (double)h[0] / ((double)slots * 16.0), (double)h[0] / ((double)h[1] * 10.0) );
}

int main(int argc, char** argv) {
  const int slots = argc > 1 ? atoi(argv[1]) : 20000;
  int* src; float* out; unsigned long long* clk;
  hipMalloc(&src, 8192 * 4); hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&clk, 16);
  static int h[8192];
  srand(1);
  for (int i = 0; i < 8192; ++i) {
    const unsigned lo = 0x3c00u + (rand() & 0x3ff) + ((rand() & 1u) << 15), hi = 0x3c00u + (rand() & 0x3ff) + ((rand() & 1u) << 15);
    h[i] = (int)(lo | (hi << 16));
  }
  hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  for (int blocks : {256}) {
    run<0>("0 f16 only, registers", blocks, slots, src, out, clk);
    run<1>("1 12 f16 + 4 fp6, registers", blocks, slots, src, out, clk);
    run<2>("2 + fragments from LDS a slot ahead", blocks, slots, src, out, clk);
    run<3>("3 + partner wave: VALU + LDS stores", blocks, slots, src, out, clk);
    run<4>("4 + partner wave: LDS stores only", blocks, slots, src, out, clk);
    run<5>("5 + partner wave: VALU only", blocks, slots, src, out, clk);
    run<6>("6 as 2, reads interleaved with the matrix instructions", blocks, slots, src, out, clk);
    run<7>("7 as 2, inline assembly, accumulators tied in arch VGPRs", blocks, slots, src, out, clk);
    run<8>("8 as 2, inline assembly, accumulators in AccVGPRs", blocks, slots, src, out, clk);
  }
  return 0;
}
