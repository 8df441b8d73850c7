// Operand layout probe for v_mfma_scale_f32_16x16x128_f8f6f4 with FP6 (e2m3) operands - the instruction DESIGN.md section 9
// plans the cross terms on.  The guides give the C/D layout and the rates, not the A / B / scale packing; this checks, on
// the chip, the packing a kernel would use:
//   lane l supplies row (A) / column (B) l & 15 and K block g = l >> 4 (32 consecutive k of the 128);
//   element t = 0..31 of the block sits at bits [6 t, 6 t + 6) of the lane's 192-bit operand (registers 0-5 of the 8);
//   code c = (E << 3 | M) with sign in bit 5: value (-1)^s * (E ? 2^(E-1) (1 + M/8) : M/8);
//   the lane's block is multiplied by 2^(scale byte 0 of the scale register - 127);
//   D[i][j], i = 4 (l >> 4) + r, j = l & 15 in register r of lane l.
// Test 1: random codes and scales against a host sum under that packing.  Test 2 (only informative if test 1 fails): one-hot
// A (group ga, element ta) x one-hot B (gb, tb) for all 128 x 128 pairs -> which positions of A meet which positions of B.
// Build: hipcc --offload-arch=gfx950 -O3 -o fp6_probe fp6_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void run(const int* __restrict__ a, const int* __restrict__ b, const int* __restrict__ sa, const int* __restrict__ sb,
                    float* __restrict__ d, int ntests) {
  const int l = threadIdx.x;
  for (int t = 0; t < ntests; ++t) {
    i32x8 va, vb;
#pragma unroll
    for (int j = 0; j < 8; ++j) { va[j] = a[(t * 64 + l) * 8 + j]; vb[j] = b[(t * 64 + l) * 8 + j]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, c, 2, 2, 0, sa[t * 64 + l], 0, sb[t * 64 + l]);
#pragma unroll
    for (int r = 0; r < 4; ++r) d[(t * 64 + l) * 4 + r] = c[r];
  }
}

static double e2m3(int c) {
  const int s = (c >> 5) & 1, E = (c >> 3) & 3, M = c & 7;
  const double v = E ? ldexp(1.0 + M / 8.0, E - 1) : M / 8.0;
  return s ? -v : v;
}
static void put(int* op, int t, int code) {  // element t of a lane's operand (8 ints, 6 used)
  const int bit = 6 * t;
  unsigned long long* w = nullptr; (void)w;
  for (int k = 0; k < 6; ++k)
    if ((code >> k) & 1) op[(bit + k) >> 5] |= 1 << ((bit + k) & 31);
}

int main() {
  const int NPAIR = 128 * 128, NT = 1 + NPAIR;
  int *ha = (int*)calloc((size_t)NT * 64 * 8, 4), *hb = (int*)calloc((size_t)NT * 64 * 8, 4), *hsa = (int*)malloc((size_t)NT * 64 * 4), *hsb = (int*)malloc((size_t)NT * 64 * 4);
  float* hd = (float*)malloc((size_t)NT * 64 * 4 * 4);
  static int ca[16][128], cb[16][128], sca[16][4], scb[16][4];
  srand(7);
  for (int i = 0; i < NT * 64; ++i) hsa[i] = hsb[i] = 127;
  for (int i = 0; i < 16; ++i) {
    for (int k = 0; k < 128; ++k) { ca[i][k] = rand() & 63; cb[i][k] = rand() & 63; }
    for (int g = 0; g < 4; ++g) { sca[i][g] = 124 + rand() % 7; scb[i][g] = 124 + rand() % 7; }
  }
  for (int l = 0; l < 64; ++l) {  // test 0: random, hypothesised packing; garbage in the scale register's other bytes
    for (int t = 0; t < 32; ++t) { put(ha + l * 8, t, ca[l & 15][32 * (l >> 4) + t]); put(hb + l * 8, t, cb[l & 15][32 * (l >> 4) + t]); }
    hsa[l] = sca[l & 15][l >> 4] | 0x5a3c9100;
    hsb[l] = scb[l & 15][l >> 4] | 0x17e2b400;
  }
  for (int p = 0; p < NPAIR; ++p) {  // tests 1..: one-hot 1.0 (code 8) at A (row 3, group ga, element ta) and B (column 5, gb, tb)
    const int pa = p >> 7, pb = p & 127;
    put(ha + ((size_t)(1 + p) * 64 + (pa >> 5) * 16 + 3) * 8, pa & 31, 8);
    put(hb + ((size_t)(1 + p) * 64 + (pb >> 5) * 16 + 5) * 8, pb & 31, 8);
  }
  int *da, *db, *dsa, *dsb; float* dd;
  hipMalloc(&da, (size_t)NT * 64 * 32); hipMalloc(&db, (size_t)NT * 64 * 32); hipMalloc(&dsa, (size_t)NT * 256); hipMalloc(&dsb, (size_t)NT * 256); hipMalloc(&dd, (size_t)NT * 1024);
  hipMemcpy(da, ha, (size_t)NT * 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, hb, (size_t)NT * 64 * 32, hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa, (size_t)NT * 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, (size_t)NT * 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(run, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd, NT);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  hipMemcpy(hd, dd, (size_t)NT * 1024, hipMemcpyDeviceToHost);
  double worst = 0.0, big = 0.0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double want = 0.0;
      for (int k = 0; k < 128; ++k) want += e2m3(ca[i][k]) * ldexp(1.0, sca[i][k >> 5] - 127) * e2m3(cb[j][k]) * ldexp(1.0, scb[j][k >> 5] - 127);
      const double got = hd[((i >> 2) * 16 + j) * 4 + (i & 3)];
      worst = fmax(worst, fabs(got - want));
      big = fmax(big, fabs(want));
    }
  printf("test 1 (random codes, per-lane scales, hypothesised packing): max |got - want| = %.3g of max |want| = %.3g -> %s\n", worst, big,
         worst <= 1e-5 * big ? "PACKING CONFIRMED" : "MISMATCH");
  int identity = 0, other = 0, misplaced = 0;
  for (int p = 0; p < NPAIR; ++p) {
    const float* d = hd + (size_t)(1 + p) * 256;
    double sum = 0.0;
    for (int q = 0; q < 256; ++q) sum += fabs(d[q]);
    const double at = d[((3 >> 2) * 16 + 5) * 4 + (3 & 3)];  // D[3][5]
    if (sum != 0.0) {
      if (at != 1.0 || sum != 1.0) ++misplaced;
      if ((p >> 7) == (p & 127)) ++identity; else { if (other < 8) printf("  A position %d meets B position %d\n", p >> 7, p & 127); ++other; }
    }
  }
  printf("test 2 (one-hot pairs): %d of 128 identical positions meet, %d other pairs meet, %d with the product not exactly at D[3][5]\n", identity, other, misplaced);
  return 0;
}
