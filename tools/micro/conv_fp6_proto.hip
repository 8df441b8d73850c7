// Numerics prototype of the arithmetic DESIGN.md section 9 proposes (NOT product code, not speed-representative: operands come
// straight from global memory, one wave per 16 pixels):  a 3x3 convolution layer, 64 -> 64 channels, computed on the chip as
//     fp16 main product (v_mfma_f32_16x16x32_f16)  +  both cross terms of TWO taps as one block-scaled FP6 instruction
//     (v_mfma_scale_f32_16x16x128_f8f6f4, e2m3, one E8M0 scale per 32-channel block)
// from activations in the "FL" line format a mover wave / a producer epilogue would write, next to the shipped split-bf16
// arithmetic (3 bf16 MFMAs per product) on the same data, both against an fp64 host sum.
//   SP line (today, 128 B per pixel and 32-channel chunk):  [8 bf16 hi] x 4 slots | [8 bf16 lo] x 4 slots
//   FL line (same 128 B):  [8 fp16 main] x 4 slots | 24 B fp6 of main | 24 B fp6 of the remainder x - main | 2 scale bytes
// The conversion SP -> FL is written the way a lane quad would do it (a lane owns slot s: 8 channels; the block maximum over
// the quad by two shuffles; e2m3 codes packed to 6 bytes per lane).  Element t = 8 s + j of a block is channel 8 s + j.
// Cross-term operand of a tap pair (A, B): k-groups 0 / 1 carry q(main) / q(remainder) of tap A's pixel against
// q(w_lo) / q(w_hi) of tap A, k-groups 2 / 3 the same for tap B (the ninth tap pairs with zeros).
// Build: hipcc --offload-arch=gfx950 -O3 -o conv_fp6_proto conv_fp6_proto.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 32, W = 32, C = 64, CO = 64, NCH = C / 32;

// ---- e2m3 with a power-of-two block scale: the largest 2^-e with amax * 2^-e <= 7.5; code = sign | E | M, round to nearest even
__host__ __device__ inline int block_exp(float amax) {  // e: real value = stored value * 2^e
  if (!(amax > 0.f)) return -127;                        // empty block: any scale (all codes are zero)
  int e;
  frexpf(amax, &e);                                      // amax = f * 2^e, f in [0.5, 1)  ->  amax * 2^-(e-3) in [4, 8)
  e -= 3;
  if (ldexpf(amax, -e) > 7.5f) ++e;
  return e < -127 ? -127 : e;
}
__host__ __device__ inline unsigned e2m3_code(float x, int e) {
  const float v = fminf(fabsf(ldexpf(x, -e)), 7.5f);
  unsigned code;
  if (v < 1.f) {
    code = (unsigned)rintf(v * 8.f);                     // subnormals k / 8; 8 is the code of 1.0
  } else {
    unsigned b;
    memcpy(&b, &v, 4);
    b += 0x7ffffu + ((b >> 20) & 1u);                    // round to nearest even at 3 mantissa bits
    code = (b >> 20) - 1008u;                            // ((exponent field - 126) << 3) | mantissa
  }
  return code | (x < 0.f ? 32u : 0u);
}
__host__ __device__ inline double e2m3_value(unsigned c) {
  const int E = (c >> 3) & 3, M = c & 7;
  const double v = E ? ldexp(1.0 + M / 8.0, E - 1) : M / 8.0;
  return (c & 32) ? -v : v;
}

// ---- SP -> FL, one lane per (pixel, chunk, slot): what a mover lane quad does with the two 16-byte pieces it holds
__global__ void sp_to_fl(const unsigned char* __restrict__ sp, unsigned char* __restrict__ fl, int nlines) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, line = tid >> 2, s = tid & 3;
  if (line >= nlines) return;
  const unsigned short* hi = reinterpret_cast<const unsigned short*>(sp + (size_t)line * 128 + s * 16);
  const unsigned short* lo = reinterpret_cast<const unsigned short*>(sp + (size_t)line * 128 + 64 + s * 16);
  float m[8], r[8], am = 0.f, ar = 0.f;
  _Float16 mh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = __uint_as_float((unsigned)hi[j] << 16) + __uint_as_float((unsigned)lo[j] << 16);  // exact: 16 significant bits
    mh[j] = (_Float16)x;
    m[j] = (float)mh[j];
    r[j] = x - m[j];
    am = fmaxf(am, fabsf(m[j]));
    ar = fmaxf(ar, fabsf(r[j]));
  }
  am = fmaxf(am, __shfl_xor(am, 1)); am = fmaxf(am, __shfl_xor(am, 2));  // the quad = the 32 channels of the block
  ar = fmaxf(ar, __shfl_xor(ar, 1)); ar = fmaxf(ar, __shfl_xor(ar, 2));
  const int em = block_exp(am), er = block_exp(ar);
  unsigned long long qm = 0, qr = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    qm |= (unsigned long long)e2m3_code(m[j], em) << (6 * j);
    qr |= (unsigned long long)e2m3_code(r[j], er) << (6 * j);
  }
  unsigned char* o = fl + (size_t)line * 128;
  memcpy(o + s * 16, mh, 16);
  memcpy(o + 64 + s * 6, &qm, 6);
  memcpy(o + 88 + s * 6, &qr, 6);
  if (s == 0) { o[112] = (unsigned char)(127 + em); o[113] = (unsigned char)(127 + er); }
}

// ---- SP -> FL, one lane per (pixel, chunk) - a whole 32-channel block in a lane, so that the conversions are the hardware's:
// v_cvt_scalef32_pk32_fp6_f16 takes 32 fp16 and one f32 scale (code of x / scale, RNE, saturating; element t at bits 6 t:
// tools/micro/cvt_fp6_probe.hip).  The remainder is exact in fp16 whenever it is not below fp16's subnormal step (2^-24).
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
__global__ __launch_bounds__(256) void sp_to_fl_lane(const unsigned char* __restrict__ sp, unsigned char* __restrict__ fl, int nlines) {
  const int line = blockIdx.x * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  const u32x4* in = reinterpret_cast<const u32x4*>(sp + (size_t)line * 128);
  u32x4 hi[4], lo[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) { hi[s] = in[s]; lo[s] = in[4 + s]; }
  f16x32 m, r;
  float am = 0.f, ar = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned h = hi[s][j >> 1], l = lo[s][j >> 1];
      const float x = __uint_as_float((j & 1) ? (h & 0xffff0000u) : (h << 16)) + __uint_as_float((j & 1) ? (l & 0xffff0000u) : (l << 16));
      const _Float16 mh = (_Float16)x;
      const float rem = x - (float)mh;
      m[s * 8 + j] = mh;
      r[s * 8 + j] = (_Float16)rem;
      am = fmaxf(am, fabsf((float)mh));
      ar = fmaxf(ar, fabsf(rem));
    }
  const int em = block_exp(am), er = block_exp(ar);
  const u32x6 qm = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(m, ldexpf(1.f, em));
  const u32x6 qr = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(r, ldexpf(1.f, er));
  unsigned* o = reinterpret_cast<unsigned*>(fl + (size_t)line * 128);
  const unsigned* mw = reinterpret_cast<const unsigned*>(&m);
#pragma unroll
  for (int s = 0; s < 4; ++s) reinterpret_cast<u32x4*>(o)[s] = u32x4{mw[4 * s], mw[4 * s + 1], mw[4 * s + 2], mw[4 * s + 3]};
  reinterpret_cast<u32x4*>(o)[4] = u32x4{qm[0], qm[1], qm[2], qm[3]};
  reinterpret_cast<u32x4*>(o)[5] = u32x4{qm[4], qm[5], qr[0], qr[1]};
  reinterpret_cast<u32x4*>(o)[6] = u32x4{qr[2], qr[3], qr[4], qr[5]};
  reinterpret_cast<u32x4*>(o)[7] = u32x4{(unsigned)(127 + em) | ((unsigned)(127 + er) << 8), 0u, 0u, 0u};
}

// ---- the layer: one wave per (row y, 16-pixel segment); A = weights (rows = output channels), B = activations (columns = pixels)
template <int SCHEME>  // 0: split bf16 x 3 from SP lines, 1: fp16 + paired fp6 from FL lines
__global__ __launch_bounds__(64) void conv(const unsigned char* __restrict__ act, const unsigned char* __restrict__ wmain,
                                           const unsigned char* __restrict__ wcross, float* __restrict__ out) {
  const int l = threadIdx.x, lr = l & 15, g = l >> 4;
  const int y = blockIdx.x / (W / 16), x0 = (blockIdx.x % (W / 16)) * 16;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto line = [&](int tap, int c) -> const unsigned char* {  // the lane's pixel under `tap`, chunk c; nullptr outside the image
    const int yy = y + tap / 3 - 1, xx = x0 + lr + tap % 3 - 1;
    return (tap < 9 && yy >= 0 && yy < H && xx >= 0 && xx < W) ? act + ((size_t)(yy * W + xx) * NCH + c) * 128 : nullptr;
  };
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  for (int c = 0; c < NCH; ++c) {
    if constexpr (SCHEME == 0) {
      for (int tap = 0; tap < 9; ++tap) {
        const unsigned char* p = line(tap, c);
        const u32x4 xh = p ? *reinterpret_cast<const u32x4*>(p + g * 16) : z4, xl = p ? *reinterpret_cast<const u32x4*>(p + 64 + g * 16) : z4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {  // weight images: [tap][chunk][hi | lo][k-group 4][64 channels] slots of 16 bytes
          const unsigned char* wp = wmain + ((((size_t)(tap * NCH + c) * 2) * 4 + g) * CO + t * 16 + lr) * 16;
          const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wp), wl = *reinterpret_cast<const bf16x8*>(wp + (size_t)4 * CO * 16);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, __builtin_bit_cast(bf16x8, xh), acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, __builtin_bit_cast(bf16x8, xl), acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, __builtin_bit_cast(bf16x8, xh), acc[t], 0, 0, 0);
        }
      }
    } else {
      for (int pr = 0; pr < 5; ++pr) {
        const int tapA = 2 * pr, tapB = 2 * pr + 1;
        const unsigned char* pa = line(tapA, c);
        const unsigned char* pb = line(tapB, c);
        const u32x4 xa = pa ? *reinterpret_cast<const u32x4*>(pa + g * 16) : z4, xb = pb ? *reinterpret_cast<const u32x4*>(pb + g * 16) : z4;
        // cross operand: k-groups 0 / 1 of tap A's pixel, 2 / 3 of tap B's: block q(main) (even group) or q(remainder) (odd)
        const unsigned char* pq = g < 2 ? pa : pb;
        i32x8 xq = {0, 0, 0, 0, 0, 0, 0, 0};
        int xs = 127;
        if (pq) {
          const unsigned long long* q = reinterpret_cast<const unsigned long long*>(pq + 64 + 24 * (g & 1));
          const unsigned long long q0 = q[0], q1 = q[1], q2 = q[2];
          xq[0] = (int)q0; xq[1] = (int)(q0 >> 32); xq[2] = (int)q1; xq[3] = (int)(q1 >> 32); xq[4] = (int)q2; xq[5] = (int)(q2 >> 32);
          xs = pq[112 + (g & 1)];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // main images: [tap][chunk][k-group 4][64] slots of 16 bytes (fp16); cross images: [pair][chunk][group 4][64] records of
          // 32 bytes: 24-byte block, scale byte - group 0 / 1: q(w_lo) / q(w_hi) of tap A, 2 / 3: of tap B
          const f16x8 wa = *reinterpret_cast<const f16x8*>(wmain + (((size_t)(tapA * NCH + c) * 4 + g) * CO + t * 16 + lr) * 16);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, __builtin_bit_cast(f16x8, xa), acc[t], 0, 0, 0);
          if (tapB < 9) {
            const f16x8 wb = *reinterpret_cast<const f16x8*>(wmain + (((size_t)(tapB * NCH + c) * 4 + g) * CO + t * 16 + lr) * 16);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, __builtin_bit_cast(f16x8, xb), acc[t], 0, 0, 0);
          }
          const unsigned char* wr = wcross + (((size_t)(pr * NCH + c) * 4 + g) * CO + t * 16 + lr) * 32;
          const unsigned long long* q = reinterpret_cast<const unsigned long long*>(wr);
          const unsigned long long q0 = q[0], q1 = q[1], q2 = q[2];
          const i32x8 wq = {(int)q0, (int)(q0 >> 32), (int)q1, (int)(q1 >> 32), (int)q2, (int)(q2 >> 32), 0, 0};
          acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wq, xq, acc[t], 2, 2, 0, (int)wr[24], 0, xs);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(size_t)(y * W + x0 + lr) * CO + t * 16 + 4 * g + r] = acc[t][r];
}

static unsigned short bf16_rne(float f) { unsigned b; memcpy(&b, &f, 4); b += 0x7fffu + ((b >> 16) & 1u); return (unsigned short)(b >> 16); }
static float bf16_f(unsigned short h) { unsigned b = (unsigned)h << 16; float f; memcpy(&f, &b, 4); return f; }
static float frand() { return (float)((rand() & 0xffffff) / 16777216.0); }
static float nrand() { return sqrtf(-2.f * logf(frand() + 1e-9f)) * cosf(6.2831853f * frand()); }

int main() {
  srand(11);
  std::vector<float> x((size_t)H * W * C), w((size_t)CO * C * 9);
  for (auto& v : x) { const float n = nrand(); v = n > 0.f ? n * expf(1.5f * nrand()) : 0.f; }  // post-ReLU, two decades of spread
  for (auto& v : w) v = nrand() * 0.05f * expf(0.7f * nrand());
  // SP lines of the input (hi = bf16(x), lo = bf16(x - hi)); the arithmetic sees x' = hi + lo on both paths
  std::vector<unsigned char> sp((size_t)H * W * NCH * 128);
  std::vector<double> xe((size_t)H * W * C);
  for (int p = 0; p < H * W; ++p)
    for (int ch = 0; ch < C; ++ch) {
      const float v = x[(size_t)p * C + ch];
      const unsigned short h = bf16_rne(v), lo = bf16_rne(v - bf16_f(h));
      unsigned short* ln = reinterpret_cast<unsigned short*>(&sp[((size_t)p * NCH + ch / 32) * 128]);
      ln[ch % 32] = h; ln[32 + ch % 32] = lo;
      xe[(size_t)p * C + ch] = (double)v;
    }
  // weight images
  std::vector<unsigned char> wb((size_t)9 * NCH * 2 * 4 * CO * 16), wf((size_t)9 * NCH * 4 * CO * 16), wc((size_t)5 * NCH * 4 * CO * 32, 0);
  for (int tap = 0; tap < 9; ++tap)
    for (int c = 0; c < NCH; ++c)
      for (int co = 0; co < CO; ++co) {
        float whf[32], wlf[32];
        for (int k = 0; k < 32; ++k) {
          const float v = w[((size_t)co * C + c * 32 + k) * 9 + tap];
          const unsigned short h = bf16_rne(v), lo = bf16_rne(v - bf16_f(h));
          unsigned short* ph = reinterpret_cast<unsigned short*>(&wb[((((size_t)(tap * NCH + c) * 2 + 0) * 4 + k / 8) * CO + co) * 16]);
          unsigned short* pl = reinterpret_cast<unsigned short*>(&wb[((((size_t)(tap * NCH + c) * 2 + 1) * 4 + k / 8) * CO + co) * 16]);
          ph[k % 8] = h; pl[k % 8] = lo;
          const _Float16 hh = (_Float16)v;
          memcpy(&wf[(((size_t)(tap * NCH + c) * 4 + k / 8) * CO + co) * 16 + (k % 8) * 2], &hh, 2);
          whf[k] = (float)hh; wlf[k] = v - (float)hh;
        }
        float ah = 0.f, al = 0.f;
        for (int k = 0; k < 32; ++k) { ah = fmaxf(ah, fabsf(whf[k])); al = fmaxf(al, fabsf(wlf[k])); }
        const int eh = block_exp(ah), el = block_exp(al);
        unsigned char* rl = &wc[(((size_t)((tap / 2) * NCH + c) * 4 + 2 * (tap & 1) + 0) * CO + co) * 32];  // q(w_lo): meets q(x main)
        unsigned char* rh = &wc[(((size_t)((tap / 2) * NCH + c) * 4 + 2 * (tap & 1) + 1) * CO + co) * 32];  // q(w_hi): meets q(x remainder)
        for (int k = 0; k < 32; ++k)
          for (int bit = 0; bit < 6; ++bit) {
            if ((e2m3_code(wlf[k], el) >> bit) & 1) rl[(6 * k + bit) >> 3] |= 1 << ((6 * k + bit) & 7);
            if ((e2m3_code(whf[k], eh) >> bit) & 1) rh[(6 * k + bit) >> 3] |= 1 << ((6 * k + bit) & 7);
          }
        rl[24] = (unsigned char)(127 + el); rh[24] = (unsigned char)(127 + eh);
      }
  for (int c = 0; c < NCH; ++c)  // the ninth tap's partner: zero blocks, scale 2^0
    for (int gq = 2; gq < 4; ++gq)
      for (int co = 0; co < CO; ++co) wc[(((size_t)(4 * NCH + c) * 4 + gq) * CO + co) * 32 + 24] = 127;
  unsigned char *dsp, *dfl, *dwb, *dwf, *dwc; float *o0, *o1;
  hipMalloc(&dsp, sp.size()); hipMalloc(&dfl, sp.size()); hipMalloc(&dwb, wb.size()); hipMalloc(&dwf, wf.size()); hipMalloc(&dwc, wc.size());
  hipMalloc(&o0, (size_t)H * W * CO * 4); hipMalloc(&o1, (size_t)H * W * CO * 4);
  hipMemcpy(dsp, sp.data(), sp.size(), hipMemcpyHostToDevice); hipMemcpy(dwb, wb.data(), wb.size(), hipMemcpyHostToDevice);
  hipMemcpy(dwf, wf.data(), wf.size(), hipMemcpyHostToDevice); hipMemcpy(dwc, wc.data(), wc.size(), hipMemcpyHostToDevice);
  const int nlines = H * W * NCH;
  hipLaunchKernelGGL(sp_to_fl, dim3((nlines * 4 + 255) / 256), dim3(256), 0, 0, dsp, dfl, nlines);
  {  // the lane-per-block form with the hardware converters must write the same lines (bytes 0 - 113)
    unsigned char* dfl2;
    hipMalloc(&dfl2, sp.size());
    hipMemset(dfl2, 0, sp.size());
    hipLaunchKernelGGL(sp_to_fl_lane, dim3((nlines + 255) / 256), dim3(256), 0, 0, dsp, dfl2, nlines);
    std::vector<unsigned char> a(sp.size()), b(sp.size());
    hipMemcpy(a.data(), dfl, a.size(), hipMemcpyDeviceToHost); hipMemcpy(b.data(), dfl2, b.size(), hipMemcpyDeviceToHost);
    long long diff = 0;
    for (int ln = 0; ln < nlines; ++ln)
      for (int i = 0; i < 114; ++i) diff += a[(size_t)ln * 128 + i] != b[(size_t)ln * 128 + i];
    printf("SP -> FL by a lane quad (manual e2m3 pack) vs by one lane per block (v_cvt_scalef32_pk32_fp6_f16): %lld differing bytes of %d lines\n", diff, nlines);
  }
  hipLaunchKernelGGL(conv<0>, dim3(H * (W / 16)), dim3(64), 0, 0, dsp, dwb, (const unsigned char*)nullptr, o0);
  hipLaunchKernelGGL(conv<1>, dim3(H * (W / 16)), dim3(64), 0, 0, dfl, dwf, dwc, o1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  std::vector<float> h0((size_t)H * W * CO), h1(h0.size());
  std::vector<unsigned char> fl(sp.size());
  hipMemcpy(h0.data(), o0, h0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(fl.data(), dfl, fl.size(), hipMemcpyDeviceToHost);
  // 1. the FL lines: main + decoded remainder against x' = hi + lo; decoded q(main) against main
  double wr = 0.0, wm = 0.0;
  for (int ln = 0; ln < nlines; ++ln) {
    const unsigned char* f = &fl[(size_t)ln * 128];
    const unsigned short* s = reinterpret_cast<const unsigned short*>(&sp[(size_t)ln * 128]);
    double am = 0.0;
    for (int k = 0; k < 32; ++k) { _Float16 m; memcpy(&m, f + 2 * k, 2); am = fmax(am, fabs((double)(float)m)); }
    for (int k = 0; k < 32; ++k) {
      _Float16 m; memcpy(&m, f + 2 * k, 2);
      unsigned cm = 0, cr = 0;
      for (int bit = 0; bit < 6; ++bit) { cm |= ((f[64 + ((6 * k + bit) >> 3)] >> ((6 * k + bit) & 7)) & 1u) << bit; cr |= ((f[88 + ((6 * k + bit) >> 3)] >> ((6 * k + bit) & 7)) & 1u) << bit; }
      const double xp = (double)bf16_f(s[k]) + (double)bf16_f(s[32 + k]);
      const double rem = e2m3_value(cr) * ldexp(1.0, f[113] - 127), qm = e2m3_value(cm) * ldexp(1.0, f[112] - 127);
      if (am > 0.0) { wr = fmax(wr, fabs((double)(float)m + rem - xp) / am); wm = fmax(wm, fabs(qm - (double)(float)m) / am); }
    }
  }
  printf("FL lines: max |main + q(rem) - x'| / block max = %.3g (remainder kept to 2^-12 * 2^-4);  max |q(main) - main| / block max = %.3g (e2m3: 2^-4 of a value, 2^-6.9 of the block)\n", wr, wm);
  // 2. the layer against an fp64 sum over the fp32 inputs
  double n0 = 0, n1 = 0, nr = 0, m0 = 0, m1 = 0, mr = 0;
  for (int yy = 0; yy < H; ++yy)
    for (int xx = 0; xx < W; ++xx)
      for (int co = 0; co < CO; ++co) {
        double ref = 0.0;
        for (int tap = 0; tap < 9; ++tap) {
          const int y2 = yy + tap / 3 - 1, x2 = xx + tap % 3 - 1;
          if (y2 < 0 || y2 >= H || x2 < 0 || x2 >= W) continue;
          for (int ch = 0; ch < C; ++ch) ref += xe[(size_t)(y2 * W + x2) * C + ch] * (double)w[((size_t)co * C + ch) * 9 + tap];
        }
        const double d0 = h0[(size_t)(yy * W + xx) * CO + co] - ref, d1 = h1[(size_t)(yy * W + xx) * CO + co] - ref;
        n0 += d0 * d0; n1 += d1 * d1; nr += ref * ref;
        m0 = fmax(m0, fabs(d0)); m1 = fmax(m1, fabs(d1)); mr = fmax(mr, fabs(ref));
      }
  printf("layer 3x3 64 -> 64 at %dx%d against fp64:   split bf16 x 3: max-rel %.3g rel-L2 %.3g   |   fp16 + paired fp6: max-rel %.3g rel-L2 %.3g\n",
         H, W, m0 / mr, sqrt(n0 / nr), m1 / mr, sqrt(n1 / nr));
  return 0;
}
