// Semantics probe for gfx950's one-instruction FP6 converters, the way DESIGN.md section 9 would use them in a mover lane that
// holds a whole 32-channel block:
//   v_cvt_scalef32_pk32_fp6_f16     32 fp16 (16 registers)            -> 32 e2m3 codes (6 registers), one f32 scale
//   v_cvt_scalef32_2xpk16_fp6_f32   2 x 16 f32 (16 + 16 registers)    -> 32 e2m3 codes, one f32 scale
// Questions (the guides do not say): is element t of the input at bits [6 t, 6 t + 6) of the output - the order the scaled MFMA
// reads (tools/micro/fp6_probe.hip)?  Does the scale divide or multiply?  Rounding?  Saturation above 7.5?
// Method: random inputs over six binades around the grid, scales 2^-2 .. 2^3, both hypotheses for the scale, host model
// round-to-nearest-even with saturation.
// Build: hipcc --offload-arch=gfx950 -O3 -o cvt_fp6_probe cvt_fp6_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));

// (a 6-element vector type is padded to 32 bytes: results go out through a plain pointer, 6 words per lane)
__global__ void k16(const f16x32* __restrict__ a, unsigned* __restrict__ o, float s) {
  const u32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(a[threadIdx.x], s);
  for (int i = 0; i < 6; ++i) o[threadIdx.x * 6 + i] = r[i];
}
__global__ void k32(const f32x16* __restrict__ a, const f32x16* __restrict__ b, unsigned* __restrict__ o, float s) {
  const u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a[threadIdx.x], b[threadIdx.x], s);
  for (int i = 0; i < 6; ++i) o[threadIdx.x * 6 + i] = r[i];
}

static unsigned model(float x, float mult) {  // e2m3 code of x * mult, round to nearest even, saturating
  const float v = fminf(fabsf(x * mult), 7.5f);
  unsigned code;
  if (v < 1.f) code = (unsigned)rintf(v * 8.f);
  else { unsigned b; memcpy(&b, &v, 4); b += 0x7ffffu + ((b >> 20) & 1u); code = (b >> 20) - 1008u; }
  return code | (signbit(x) ? 32u : 0u);
}
static unsigned elem(const unsigned* r, int t) {
  unsigned c = 0;
  for (int bit = 0; bit < 6; ++bit) c |= ((r[(6 * t + bit) >> 5] >> ((6 * t + bit) & 31)) & 1u) << bit;
  return c;
}

int main() {
  srand(3);
  static _Float16 h16[64][32];
  static float h32[64][32], ha[64][16], hb[64][16];
  for (int l = 0; l < 64; ++l)
    for (int t = 0; t < 32; ++t) {
      const float v = ldexpf(1.f + (rand() & 1023) / 1024.f, rand() % 6 - 3) * ((rand() & 1) ? -1.f : 1.f);
      h16[l][t] = (_Float16)v;
      h32[l][t] = (float)h16[l][t];  // (same values on both converters)
      (t < 16 ? ha[l][t] : hb[l][t - 16]) = h32[l][t];
    }
  f16x32* d16; f32x16 *da, *db; unsigned* dout;
  hipMalloc(&d16, sizeof(h16)); hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, 64 * 24);
  hipMemcpy(d16, h16, sizeof(h16), hipMemcpyHostToDevice); hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  static unsigned out[64][6];
  for (int which = 0; which < 2; ++which)
    for (int e = -2; e <= 3; ++e) {
      const float s = ldexpf(1.f, e);
      if (which == 0) hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, d16, dout, s);
      else hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, da, db, dout, s);
      if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
      hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
      int bad_div = 0, bad_mul = 0, bad_div_il = 0;
      for (int l = 0; l < 64; ++l)
        for (int t = 0; t < 32; ++t) {
          const unsigned got = elem(out[l], t);
          bad_div += got != model(h32[l][t], 1.f / s);
          bad_mul += got != model(h32[l][t], s);
          // the two-source form might interleave its sources: output element t from (t even ? first : second)[t / 2]
          bad_div_il += got != model(h32[l][(t & 1) * 16 + (t >> 1)], 1.f / s);
        }
      printf("%s scale 2^%d: mismatches of 2048 -  element t at bits 6t, x / scale: %d   x * scale: %d   (sources interleaved, x / scale: %d)\n",
             which == 0 ? "pk32_fp6_f16 " : "2xpk16_fp6_f32", e, bad_div, bad_mul, bad_div_il);
      if (bad_div && bad_mul && e == 0) {
        for (int t = 0; t < 8; ++t) printf("   lane 0 element %d: x = %g -> code %u (model %u)\n", t, h32[0][t], elem(out[0], t), model(h32[0][t], 1.f));
      }
    }
  return 0;
}
