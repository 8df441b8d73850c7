// 3x3 stride-1 tap-convolution over SP-format activations for the SHALLOW layers: few input channels, large images, all
// weights resident in LDS (9 * Cin * Cout * 4 bytes <= ~150 KB) - the first residual block (conv1 + skip convolution,
// conv2 + 1x1 shortcut; reference UNet_model_superres.py:153-172), ups.2.conv, and up_convs.2 with the fused output
// projection (:197-207, :377).
//
// Why a second 3x3 kernel: on these layers a K-step of the wave-specialised kernel (conv_mfma_sp.hip) is short (one or
// two 32-channel chunks per 16x16 patch), its movers re-stage 36-72 KB of weights for every patch next to 41 KB of window,
// have one step of loads in flight against HBM latency, and every patch ends in an epilogue: 150-300 TFLOP/s where the deep
// layers reach 400 (phase timeline: 9.1 k ticks per step for 3.5 k ticks of MFMA).  Here
//   * the weights are copied into LDS once per block and stay (no ring, no counters, no mover waves);
//   * operands go global -> registers (an SP slot is an MFMA B-operand register group): a wave owns a strip of 16 x RB
//     output pixels and walks down its input rows; the x - 1 / x + 1 taps of a row are DPP lane shifts of the row's own
//     fragment (lanes 0 / 15 take the strip's outer pixels from a two-lane edge load), rows are prefetched a row pair ahead
//     and across pass boundaries;
//   * two output rows share every weight-fragment read (0.33 LDS reads per MFMA);
//   * per-image epilogue vectors (bias, time embedding rows) are staged in LDS at kernel start.
// All eight waves of a block are equal; the only barrier is the one after the weight copy.
// Flavours (template): NT channel tiles per wave (2: 32 channels of one group; 4: the 64 weight channels of the fused conv1 +
// skip pair, whose 16 input channels leave half of an MFMA's K empty: it multiplies TWO taps per MFMA, 5 instead of 9), HAS2 (the block's 1x1 shortcut input as one more tap in the last K-pass), DUAL, FUSE.
#include <stdio.h>
#include <stdlib.h>

#include "conv_epilogue.h"
#include "mfma_policy.h"

namespace {

constexpr int RB = 4;       // output rows per strip (two row pairs)
constexpr int NPOST = 64;  // images whose per-image epilogue vectors are staged in LDS (later images read them from memory)

struct RowOp { PolicyBF16X3::Frag c, e; };  // c: pixel x0 + lr of the row; e: lanes 0 / 15 hold pixels x0 - 1 / x0 + 16

__device__ __forceinline__ bf16x8 dpp_shift(const bf16x8& edge, const bf16x8& own, bool left) {
  const u32x4 e = __builtin_bit_cast(u32x4, edge), r = __builtin_bit_cast(u32x4, own);
  u32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j)  // row_shr:1 (x - 1): lane lr takes lane lr - 1, lane 0 keeps `edge`; row_shl:1 (x + 1): lane 15 keeps it
    o[j] = left ? (unsigned)__builtin_amdgcn_update_dpp((int)e[j], (int)r[j], 0x111, 0xf, 0xf, false)
                : (unsigned)__builtin_amdgcn_update_dpp((int)e[j], (int)r[j], 0x101, 0xf, 0xf, false);
  return __builtin_bit_cast(bf16x8, o);
}

template <int NT, bool HAS2, bool DUAL, bool FUSE>
__global__ __launch_bounds__(512, 1) void conv3x3_direct_sp_kernel(TapConv d, int nck, unsigned w_gimage, unsigned w2_gimage) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  using Frag = typename P::Frag;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kg = lane >> 4;
  const int CW = DUAL ? 2 * d.Cout : d.Cout;  // resident weight channels
  // bytes of one operand image of the 3x3 weights: [chunk][tap][k-group][CW] slots; the 16-channel pair flavour packs TWO taps
  // into the K = 32 of an MFMA (k-groups 0-1: tap 2j, k-groups 2-3: tap 2j + 1): [tap pair 5][k-group][CW]
  const int img = DUAL ? 5 * 4 * CW * 16 : nck * 36 * CW * 16;
  const int img2 = HAS2 ? 4 * CW * 16 : 0;    // ... of the 1x1 shortcut weights (one K-chunk): [k-group][CW]
  char* sW = smem;
  char* sW2 = smem + 2 * img;
  float* sBias = reinterpret_cast<float*>(sW2 + 2 * img2);  // [CW] bias (+ bias2)   (fused projection: + fuse_w[4][32] + fuse_b[4])
  float* sPost = sBias + (FUSE ? 32 + 128 + 4 : CW);        // [N][Cout] post_add rows, then [N][Cout] post2 rows
  const int npost = min(d.N, NPOST);
  float* sPost2 = sPost + npost * d.Cout;
  {
    auto copy = [&](char* dst, const char* src, int bytes) __attribute__((always_inline)) {  // 8 loads in flight per thread
      for (int o0 = tid * 16; o0 < bytes; o0 += 8 * 512 * 16) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int o = o0 + u * 512 * 16;
          if (o < bytes) v[u] = *reinterpret_cast<const u32x4*>(src + o);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int o = o0 + u * 512 * 16;
          if (o < bytes) *reinterpret_cast<u32x4*>(dst + o) = v[u];
        }
      }
    };
    const char* w = reinterpret_cast<const char*>(d.w);
    if constexpr (DUAL) {
      for (int i = tid; i < 2 * 20 * CW; i += 512) {
        const int im = i >= 20 * CW, idx = im ? i - 20 * CW : i;
        const int ch = idx % CW, kgp = (idx / CW) & 3, j = idx / (4 * CW), tap = 2 * j + (kgp >> 1);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (tap < 9) v = *reinterpret_cast<const u32x4*>(w + (size_t)im * w_gimage + ((size_t)(tap * 4 + (kgp & 1)) * CW + ch) * 16);
        *reinterpret_cast<u32x4*>(sW + (size_t)im * img + (size_t)idx * 16) = v;
      }
    } else {
      copy(sW, w, img);
      copy(sW + img, w + w_gimage, img);
    }
    if constexpr (HAS2) {
      const char* w2 = reinterpret_cast<const char*>(d.w2);
      copy(sW2, w2, img2);
      copy(sW2 + img2, w2 + w2_gimage, img2);
    }
    for (int i = tid; i < CW; i += 512) sBias[i] = (d.bias ? d.bias[i] : 0.f) + ((HAS2 && d.bias2) ? d.bias2[i] : 0.f);
    if constexpr (FUSE) {
      for (int i = tid; i < 128; i += 512) sBias[32 + i] = d.fuse_w[min(i >> 5, d.fuse_dim - 1) * d.Cout + (i & 31)];
      if (tid < 4) sBias[160 + tid] = d.fuse_b[min(tid, d.fuse_dim - 1)];
    } else {
      for (int i = tid; i < npost * d.Cout; i += 512) {
        const int n = i / d.Cout, c = i - n * d.Cout;
        sPost[i] = d.post_add ? d.post_add[(size_t)n * d.post_cs + c] : 0.f;
        sPost2[i] = (!DUAL && d.out2) ? d.post2[(size_t)n * d.post2_cs + c] : 0.f;
      }
    }
  }
  __syncthreads();

  // strips of 16 x RB output pixels x one 16 * NT channel group; blocks of an XCD (blockIdx % 8) take a contiguous eighth
  const int gx = d.W >> 4, gy = d.H / RB;
  const int groups = DUAL ? 1 : d.Cout / (16 * NT);
  const int total = d.N * gy * gx * groups;
  const int xcd = blockIdx.x & 7, member = blockIdx.x >> 3, members = gridDim.x >> 3;
  const int t_lo = (int)((long long)total * xcd / 8), t_hi = (int)((long long)total * (xcd + 1) / 8);
  const int stride = members * 8, first = member * 8 + wave;
  const int my_items = first < t_hi - t_lo ? (t_hi - t_lo - first + stride - 1) / stride : 0;
  const int S = my_items * nck;  // passes: one K-chunk of one strip each
  if (S == 0) return;
  const int half = drs_sp_group_bytes(d.in_cs), half2 = HAS2 ? drs_sp_group_bytes(d.in2_cs) : 0;
  const int pixb = d.in_cs * 4;
  const char* zero = reinterpret_cast<const char*>(d.zero_line) + kg * 16;
  auto strip_of = [&](int it, int& n, int& yb, int& x0, int& n0) __attribute__((always_inline)) {
    int q = t_lo + first + it * stride;
    n0 = (q % groups) * 16 * NT; q /= groups;
    x0 = (q % gx) * 16; q /= gx;
    yb = (q % gy) * RB;
    n = q / gy;
  };
  // one input row of a pass: window row wr (0 .. RB + 1) of chunk c of the strip
  auto load_row = [&](RowOp& r, int n, int yb, int x0, int c, int wr) __attribute__((always_inline)) {
    const int iy = yb - 1 + wr;
    const int kgc = DUAL ? (kg & 1) : kg;  // pair flavour (16 channels): k-groups 2-3 carry the same channels as 0-1
    const bool ok = iy >= 0 && iy < d.H && c * 32 + kgc * 8 < d.Cin;
    const char* base = reinterpret_cast<const char*>(d.in) +
                       ((((long long)n * d.H + iy) * d.W + x0 + lr) * d.in_cs + d.in_co) * 4 + c * 128 + kgc * 16;
    const char* p = ok ? base : zero;
    r.c = Frag{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(ok ? p + half : zero)};
    if (lr == 0 || lr == 15) {
      const int ex = lr == 0 ? x0 - 1 : x0 + 16;
      const bool eok = ok && ex >= 0 && ex < d.W;
      const char* pe = eok ? (lr == 0 ? base - pixb : base + pixb) : zero;
      r.e = Frag{*reinterpret_cast<const bf16x8*>(pe), *reinterpret_cast<const bf16x8*>(eok ? pe + half : zero)};
    }
  };
  // the shortcut input at output row r of the strip (centre tap only, one K-chunk)
  auto load_x2 = [&](Frag& f, int n, int yb, int x0, int r) __attribute__((always_inline)) {
    const bool ok = kg * 8 < d.Cin2;
    const char* base = reinterpret_cast<const char*>(d.in2) +
                       ((((long long)n * d.H2 + yb + r) * d.W2 + x0 + lr) * d.in2_cs + d.in2_co) * 4 + kg * 16;
    const char* p = ok ? base : zero;
    f = Frag{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(ok ? p + half2 : zero)};
  };
  const char* wlane = sW + ((size_t)kg * CW + lr) * 16;
  const char* w2lane = sW2 + ((size_t)kg * CW + lr) * 16;

  f32x4 acc[RB][NT];
  RowOp R[RB + 2], nx[4];
  Frag X2[RB], nx2[2];
  int n = 0, yb = 0, x0 = 0, n0 = 0;
  strip_of(0, n, yb, x0, n0);
#pragma unroll
  for (int i = 0; i < 4; ++i) load_row(nx[i], n, yb, x0, 0, i);
  if constexpr (HAS2) {
    if (nck == 1) {
      load_x2(nx2[0], n, yb, x0, 0);
      load_x2(nx2[1], n, yb, x0, 1);
    }
  }
  for (int s = 0; s < S; ++s) {
    const int it = s / nck, c = s - it * nck;
    const bool last_pass = c == nck - 1;
    if (c == 0) {
#pragma unroll
      for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) R[i] = nx[i];
    if constexpr (HAS2) { X2[0] = nx2[0]; X2[1] = nx2[1]; }
    // coordinates of the next pass (its first rows are fetched under this pass's last row pair)
    int n1 = n, yb1 = yb, x01 = x0, n01 = n0;
    const int c1 = c + 1 == nck ? 0 : c + 1;
    if (c1 == 0 && s + 1 < S) strip_of(it + 1, n1, yb1, x01, n01);
    const char* wc = wlane + (size_t)c * 36 * CW * 16 + (size_t)n0 * 16;
#pragma unroll
    for (int p = 0; p < RB / 2; ++p) {
      // ---- prefetch: the next row pair of this pass, or the first four rows of the next pass ----
      if (p + 1 < RB / 2) {
        load_row(R[2 * p + 4], n, yb, x0, c, 2 * p + 4);
        load_row(R[2 * p + 5], n, yb, x0, c, 2 * p + 5);
        if constexpr (HAS2) {
          if (last_pass) {
            load_x2(X2[2 * p + 2], n, yb, x0, 2 * p + 2);
            load_x2(X2[2 * p + 3], n, yb, x0, 2 * p + 3);
          }
        }
      } else if (s + 1 < S) {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_row(nx[i], n1, yb1, x01, c1, i);
        if constexpr (HAS2) {
          if (c1 == nck - 1) {
            load_x2(nx2[0], n1, yb1, x01, 0);
            load_x2(nx2[1], n1, yb1, x01, 1);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- output rows 2p, 2p + 1: window rows 2p + ky and 2p + 1 + ky; every weight fragment serves both ----
      auto tap_operand = [&](const RowOp& r, int kx) __attribute__((always_inline)) {
        if (kx == 1) return r.c;
        return Frag{dpp_shift(r.e.hi, r.c.hi, kx == 0), dpp_shift(r.e.lo, r.c.lo, kx == 0)};
      };
      if constexpr (DUAL) {
        // two taps per MFMA: lanes of k-groups 0-1 take tap 2j's pixel, lanes of k-groups 2-3 tap 2j + 1's (the ninth tap
        // shares its MFMA with zero weights)
        auto pick = [&](const Frag& a, const Frag& b) __attribute__((always_inline)) {
          const u32x4 ah = __builtin_bit_cast(u32x4, a.hi), al = __builtin_bit_cast(u32x4, a.lo);
          const u32x4 bh = __builtin_bit_cast(u32x4, b.hi), bl = __builtin_bit_cast(u32x4, b.lo);
          u32x4 h, l;
#pragma unroll
          for (int q = 0; q < 4; ++q) { h[q] = kg < 2 ? ah[q] : bh[q]; l[q] = kg < 2 ? al[q] : bl[q]; }
          return Frag{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, l)};
        };
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int ta = 2 * j, tb = 2 * j + 1 < 9 ? 2 * j + 1 : 2 * j;
          const Frag a0 = pick(tap_operand(R[2 * p + ta / 3], ta % 3), tap_operand(R[2 * p + tb / 3], tb % 3));
          const Frag a1 = pick(tap_operand(R[2 * p + 1 + ta / 3], ta % 3), tap_operand(R[2 * p + 1 + tb / 3], tb % 3));
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const Frag wf = P::load(wlane, (size_t)img, (size_t)(j * 4 * CW + t * 16) * 16);
            acc[2 * p][t] = P::mma(wf, a0, acc[2 * p][t]);
            acc[2 * p + 1][t] = P::mma(wf, a1, acc[2 * p + 1][t]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const RowOp& ra = R[2 * p + ky];
        const RowOp& rb = R[2 * p + 1 + ky];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const Frag a0 = tap_operand(ra, kx), a1 = tap_operand(rb, kx);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const Frag wf = P::load(wc, (size_t)img, (size_t)((ky * 3 + kx) * 4 * CW + t * 16) * 16);
            acc[2 * p][t] = P::mma(wf, a0, acc[2 * p][t]);
            acc[2 * p + 1][t] = P::mma(wf, a1, acc[2 * p + 1][t]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      if constexpr (HAS2) {
        if (last_pass) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const Frag wf = P::load(w2lane, (size_t)img2, (size_t)(n0 + t * 16) * 16);
            acc[2 * p][t] = P::mma(wf, X2[2 * p], acc[2 * p][t]);
            acc[2 * p + 1][t] = P::mma(wf, X2[2 * p + 1], acc[2 * p + 1][t]);
          }
        }
      }
    }
    if (last_pass) {
      auto lds8 = [&](const float* q, float (&v)[8]) __attribute__((always_inline)) {
        const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      };
      auto zero8 = [&](float (&v)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      };
      if constexpr (DUAL) {
        // out = relu(main + b_main) + post_add + (skip + b_skip): tiles t (main) and t + 2 (skip) of the same lane
        float bm[8], bs[8];
        lds8(sBias + kg * 8, bm);
        lds8(sBias + d.Cout + kg * 8, bs);
        f32x4 comb[RB][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) comb[r][t][j] = drs_maxf(acc[r][t][j] + bm[t * 4 + j], 0.f) + (acc[r][t + 2][j] + bs[t * 4 + j]);
        SpEpiConst kc;
        if (n < npost) lds8(sPost + n * d.Cout + kg * 8, kc.post);
        else if (d.post_add) lds8(d.post_add + (size_t)n * d.post_cs + kg * 8, kc.post);
        else zero8(kc.post);
#pragma unroll
        for (int j = 0; j < 8; ++j) { kc.bias[j] = 0.f; kc.post2[j] = 0.f; }
        TapConv de = d;
        de.relu_pre = 0;
        tile_epilogue_sp_pre<RB, false>(de, comb, kc, n, 0, yb, x0, 0, lr, kg);
      } else if constexpr (FUSE) {
        FuseEpiConst kc;
        kc.b0 = *reinterpret_cast<const float4*>(sBias + kg * 4);
        kc.b1 = *reinterpret_cast<const float4*>(sBias + 16 + kg * 4);
        const int m = min(lr >> 2, 3);  // (row lr of the projection operand = channel lr >> 2: fuse_epilogue_mfma_pre)
        const float4 w0 = *reinterpret_cast<const float4*>(sBias + 32 + m * 32 + kg * 4);
        const float4 w1 = *reinterpret_cast<const float4*>(sBias + 32 + m * 32 + 16 + kg * 4);
        kc.w8[0] = w0.x; kc.w8[1] = w0.y; kc.w8[2] = w0.z; kc.w8[3] = w0.w;
        kc.w8[4] = w1.x; kc.w8[5] = w1.y; kc.w8[6] = w1.z; kc.w8[7] = w1.w;
        const float4 fb = *reinterpret_cast<const float4*>(sBias + 160);
        kc.fb[0] = fb.x; kc.fb[1] = fb.y; kc.fb[2] = fb.z; kc.fb[3] = fb.w;
        fuse_epilogue_mfma_pre<RB>(d, acc, kc, n, n0, yb, x0, 0, lr, kg);
      } else {
        SpEpiConst kc;
        lds8(sBias + n0 + kg * 8, kc.bias);
        zero8(kc.post); zero8(kc.post2);
        if (n < npost) {
          lds8(sPost + n * d.Cout + n0 + kg * 8, kc.post);
          lds8(sPost2 + n * d.Cout + n0 + kg * 8, kc.post2);
        } else {
          if (d.post_add) lds8(d.post_add + (size_t)n * d.post_cs + n0 + kg * 8, kc.post);
          if (d.out2) lds8(d.post2 + (size_t)n * d.post2_cs + n0 + kg * 8, kc.post2);
        }
        tile_epilogue_sp_pre<RB, true>(d, acc, kc, n, n0, yb, x0, 0, lr, kg);
      }
    }
    n = n1; yb = yb1; x0 = x01; n0 = n01;
  }
}

// ---- folded projection, taps in the M dimension -------------------------------------------------------------------------
// output o up_convs.2[att half] is a 3x3 convolution Cin -> fuse_dim <= 4 (TapConv::proj).  With so few outputs the 16 rows of
// an MFMA tile hold the THREE KERNEL ROWS at once: row 4 ky + o = W[ky][kx][o][:], one tile per kernel column kx.  An input row
// is then multiplied ONCE (three column-shifted fragments x one tile = 9 MFMAs, against 27 when every output row multiplies
// its three input rows): T_wr[4 ky + o][p] = what input row wr gives to output row wr - ky, and an output row is the sum of
// three lane groups of three consecutive T's (k-group ky of T_{y + ky}): twelve lane reads per row.  A wave owns 16 x RBP
// output pixels and streams their RBP + 2 input rows through a short register ring; weights (6 KB, Cin == 32) stay in LDS /
// registers; three waves per SIMD.
constexpr int RBP = 8;  // output rows per strip of the projection kernel
template <bool GATE>
__global__ __launch_bounds__(256, GATE ? 2 : 3) void conv3x3_proj_sp_kernel(TapConv d, unsigned w_gimage) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  using Frag = typename P::Frag;
  constexpr int NR = RBP + 2, PF = 3;  // input rows of a strip; rows in flight ahead of the one being multiplied
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kg = lane >> 4;
  const int img = 3 * 4 * 16 * 16;  // one operand image (Cin == 32: one K-chunk): [kx][k-group][16 rows] slots
  char* sW = smem;
  float* sFb = reinterpret_cast<float*>(smem + 2 * img);
  {
    const char* w = reinterpret_cast<const char*>(d.w);
    for (int o = tid * 16; o < img; o += 256 * 16) {
      *reinterpret_cast<u32x4*>(sW + o) = *reinterpret_cast<const u32x4*>(w + o);
      *reinterpret_cast<u32x4*>(sW + img + o) = *reinterpret_cast<const u32x4*>(w + w_gimage + o);
    }
    if (tid < 4) sFb[tid] = d.fuse_b ? d.fuse_b[min(tid, d.fuse_dim - 1)] : 0.f;
    if (GATE && tid < 36) sFb[4 + tid] = d.bias ? d.bias[tid] : 0.f;
  }
  const float* sTab = sFb + 4;
  (void)sTab;
  __syncthreads();
  const int gx = d.W >> 4, gy = d.H / RBP;
  const int total = d.N * gy * gx;
  const int xcd = blockIdx.x & 7, member = blockIdx.x >> 3, members = gridDim.x >> 3;
  const int t_lo = (int)((long long)total * xcd / 8), t_hi = (int)((long long)total * (xcd + 1) / 8);
  const int stride = members * 4, first = member * 4 + wave;
  const int half = drs_sp_group_bytes(d.in_cs), pixb = d.in_cs * 4;
  const char* zero = reinterpret_cast<const char*>(d.zero_line) + kg * 16;
  const char* wlane = sW + ((size_t)kg * 16 + lr) * 16;
  const size_t plane = (size_t)d.OH * d.OW;
  const float fbk = sFb[kg];
  Frag wf[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) wf[kx] = P::load(wlane, (size_t)img, (size_t)(kx * 4 * 16) * 16);
  for (int q0 = t_lo + first; q0 < t_hi; q0 += stride) {
    int q = q0;
    const int x0 = (q % gx) * 16; q /= gx;
    const int yb = (q % gy) * RBP;
    const int n = q / gy;
    // An input row is multiplied ONCE and is dead afterwards: the rows stream through a ring of PF + 1 register sets, PF rows
    // ahead of the MFMAs (the scheduling barriers keep the compiler from hoisting every load of the strip to its top)
    RowOp R[PF + 1];
    float psv[PF + 1][3];
    auto load_row = [&](int slot, int wr) __attribute__((always_inline)) {
      const int iy = yb - 1 + wr;
      const bool ok = iy >= 0 && iy < d.H;
      const char* base = reinterpret_cast<const char*>(d.in) +
                         ((((long long)n * d.H + iy) * d.W + x0 + lr) * d.in_cs + d.in_co) * 4 + kg * 16;
      const char* p = ok ? base : zero;
      R[slot].c = Frag{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(ok ? p + half : zero)};
      if (lr == 0 || lr == 15) {
        const int ex = lr == 0 ? x0 - 1 : x0 + 16;
        const bool eok = ok && ex >= 0 && ex < d.W;
        const char* pe = eok ? (lr == 0 ? base - pixb : base + pixb) : zero;
        R[slot].e = Frag{*reinterpret_cast<const bf16x8*>(pe), *reinterpret_cast<const bf16x8*>(eok ? pe + half : zero)};
      }
      if constexpr (GATE) {
        // the attention gate: input pixel (iy, ix) was multiplied by psi[n][iy / 2][ix / 2] (reference :105-106, nearest 2x); the map
        // is linear, so the factor goes onto the 4 accumulators of the pixel's kernel column instead of its 32 channels
        const float* pr = d.gate + ((size_t)n * (d.H >> 1) + (min(max(iy, 0), d.H - 1) >> 1)) * (d.W >> 1);
        psv[slot][0] = pr[max(x0 + lr - 1, 0) >> 1];
        psv[slot][1] = pr[(x0 + lr) >> 1];
        psv[slot][2] = pr[min(x0 + lr + 1, d.W - 1) >> 1];
      }
    };
    f32x4 t[NR];
#pragma unroll
    for (int i = 0; i < PF; ++i) load_row(i, i);
#pragma unroll
    for (int wr = 0; wr < NR; ++wr) {
      if (wr + PF < NR) load_row((wr + PF) % (PF + 1), wr + PF);
      __builtin_amdgcn_sched_barrier(0);
      const RowOp& r = R[wr % (PF + 1)];
      f32x4 tk[GATE ? 3 : 1];
#pragma unroll
      for (int j = 0; j < (GATE ? 3 : 1); ++j) tk[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const Frag a = kx == 1 ? r.c : Frag{dpp_shift(r.e.hi, r.c.hi, kx == 0), dpp_shift(r.e.lo, r.c.lo, kx == 0)};
        tk[GATE ? kx : 0] = P::mma(wf[kx], a, tk[GATE ? kx : 0]);
      }
      if constexpr (GATE) {
        const float* sc = psv[wr % (PF + 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[wr][j] = sc[0] * tk[0][j] + sc[1] * tk[1][j] + sc[2] * tk[GATE ? 2 : 0][j];
      } else {
        t[wr] = tk[0];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float* o = d.fuse_out + ((size_t)n * d.fuse_dim + min(kg, d.fuse_dim - 1)) * plane + (size_t)yb * d.OW + x0 + lr;
#pragma unroll
    for (int y = 0; y < RBP; ++y) {
      float sv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        sv[j] = __shfl(t[y][j], lr, 64) + __shfl(t[y + 1][j], lr + 16, 64) + __shfl(t[y + 2][j], lr + 32, 64);
      float v = kg == 0 ? sv[0] : kg == 1 ? sv[1] : kg == 2 ? sv[2] : sv[3];
      if constexpr (GATE) {  // the folded `result` bias: a constant per (row class, column class) of the output pixel
        const int oy = yb + y, ox = x0 + lr;
        const int rc = oy == 0 ? 0 : (oy == d.H - 1 ? 2 : 1), cc = ox == 0 ? 0 : (ox == d.W - 1 ? 2 : 1);
        v += sTab[(rc * 3 + cc) * 4 + kg];
      }
      if (kg < d.fuse_dim) o[(size_t)y * d.OW] = v + fbk;
    }
  }
}

size_t direct_lds_bytes(const TapConv& d) {
  const int nck = (d.Cin + 31) / 32, CW = d.dual ? 2 * d.Cout : d.Cout;
  size_t b = (d.dual ? (size_t)2 * 5 * 4 * CW * 16 : (size_t)2 * nck * 36 * CW * 16) + (d.in2 ? (size_t)2 * 4 * CW * 16 : 0);
  b += d.fuse_out ? (32 + 128 + 4) * 4 : (size_t)(CW + 2 * (d.N < NPOST ? d.N : NPOST) * d.Cout) * 4;
  return b;
}

template <int NT, bool HAS2, bool DUAL, bool FUSE>
int direct_launch(const TapConv& d, hipStream_t s) {
  auto kern = conv3x3_direct_sp_kernel<NT, HAS2, DUAL, FUSE>;
  int num_cu = 0;
  {
    const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(kern), 160 * 1024, &num_cu);
    if (rc) return rc;
  }
  const int nck = (d.Cin + 31) / 32, CW = DUAL ? 2 * d.Cout : d.Cout;
  const unsigned w_gimage = (unsigned)((size_t)nck * 9 * 4 * CW * 16);
  const unsigned w2_gimage = HAS2 ? (unsigned)((size_t)((d.Cin2 + 31) / 32) * 4 * d.Cout * 16) : 0u;
  const int blocks = num_cu / 8 * 8;  // one block per CU
  DRS_LAUNCH(kern, dim3((unsigned)blocks), dim3(512), direct_lds_bytes(d), s, d, nck, w_gimage, w2_gimage);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

bool std3x3(const TapConv& d) {
  if (d.mode != 0 || d.ntaps != 9 || d.wtaps_total != 9 || d.in_stride != 1 || d.out_scale != 1 || d.out_oy || d.out_ox) return false;
  for (int i = 0; i < 9; ++i)
    if (d.dy[i] != i / 3 - 1 || d.dx[i] != i % 3 - 1 || d.wtap[i] != i) return false;
  return true;
}

// W'[4 ky + o][ci][kx] = sum_co fw[o][co] * w[co][cin_off + ci][ky * 3 + kx]: the 1x1 projection behind a bare 3x3 convolution
// folded into its weights (fp32 contraction at pack time, the same class of re-association as the BatchNorm fold and the
// composite stage), in the row order of conv3x3_proj_sp_kernel: a 16-row, 3-"tap" (kernel column) layer for the pack kernel
__global__ __launch_bounds__(256) void fold_proj_kernel(const float* __restrict__ w, int cin_total, int cin_off, int Cmid, int Cin,
                                                        const float* __restrict__ fw, int fuse_dim, float* __restrict__ dst) {
  const int total = 16 * Cin * 3;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int row = i / (Cin * 3), rem = i - row * Cin * 3, ci = rem / 3, kx = rem - ci * 3;
    const int ky = row >> 2, o = row & 3;
    float a = 0.f;
    if (ky < 3 && o < fuse_dim)
      for (int co = 0; co < Cmid; ++co) a += fw[(size_t)o * Cmid + co] * w[((size_t)co * cin_total + cin_off + ci) * 9 + ky * 3 + kx];
    dst[i] = a;
  }
}

// The gated form: the attention block's `result` convolution (1x1, BatchNorm folded: reference :84-87,107) sits between the gate
// and the att-half and is linear too: att = psi * (Wr' x) + br', so output o up_convs.2[att] o result is ONE 3x3 convolution of
// psi * x.  dst2[row][ci][kx] = sum_m dst1[row][m][kx] * Wr'[m][ci] (dst1: fold_proj_kernel's rows); the constant br' reaches an
// output pixel through the taps that stay inside the image only: tab[row class 3][column class 3][4] (first / interior / last).
__global__ __launch_bounds__(256) void fold_result_kernel(const float* __restrict__ dst1, int C, const float* __restrict__ wr,
                                                          const float* __restrict__ br, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ rmean,
                                                          const float* __restrict__ rvar, float eps, float* __restrict__ dst2,
                                                          float* __restrict__ tab) {
  const int total = 16 * C * 3;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int row = i / (C * 3), rem = i - row * C * 3, ci = rem / 3, kx = rem - ci * 3;
    float a = 0.f;
    for (int m = 0; m < C; ++m) {
      const float sc = gamma[m] / sqrtf(rvar[m] + eps);
      a += dst1[((size_t)row * C + m) * 3 + kx] * (sc * wr[(size_t)m * C + ci]);
    }
    dst2[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < 36) {
    const int o = threadIdx.x & 3, cc = (threadIdx.x >> 2) % 3, rc = threadIdx.x / 12;
    float a = 0.f;
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        if ((rc == 0 && ky == 0) || (rc == 2 && ky == 2) || (cc == 0 && kx == 0) || (cc == 2 && kx == 2)) continue;
        for (int m = 0; m < C; ++m) {
          const float sc = gamma[m] / sqrtf(rvar[m] + eps);
          const float bm = ((br ? br[m] : 0.f) - rmean[m]) * sc + beta[m];
          a += dst1[((size_t)(4 * ky + o) * C + m) * 3 + kx] * bm;
        }
      }
    tab[threadIdx.x] = a;
  }
}

}  // namespace

int drs_launch_fold_result(const float* dst1, int C, const float* wr, const float* br, const float* gamma, const float* beta,
                           const float* rmean, const float* rvar, float eps, float* dst2, float* tab, hipStream_t s) {
  DRS_REQUIRE(dst1 && wr && gamma && beta && rmean && rvar && dst2 && tab && C > 0, DRS_ERR_ARG, "fold_result: bad arguments");
  DRS_LAUNCH(fold_result_kernel, dim3((16 * C * 3 + 255) / 256), dim3(256), 0, s, dst1, C, wr, br, gamma, beta, rmean, rvar, eps, dst2, tab);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

int drs_launch_fold_proj(const float* w, int cin_total, int cin_off, int Cmid, int Cin, const float* fw, int fuse_dim, float* dst,
                         hipStream_t s) {
  DRS_REQUIRE(w && fw && dst && fuse_dim >= 1 && fuse_dim <= 4 && Cin > 0 && Cmid > 0, DRS_ERR_ARG, "fold_proj: bad arguments");
  DRS_LAUNCH(fold_proj_kernel, dim3((16 * Cin * 3 + 255) / 256), dim3(256), 0, s, w, cin_total, cin_off, Cmid, Cin, fw, fuse_dim, dst);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// The folded-projection kernel (TapConv::proj; conv3x3_proj_sp_kernel): 16-row x 3-column weight image, planar fp32 output.
bool drs_conv3x3_direct_sp_proj_supported(const TapConv& d, int impl) {
  static const bool env = !(getenv("DRS_FOLD_PROJ") && atoi(getenv("DRS_FOLD_PROJ")) == 0);
  if (!env || impl != DRS_IMPL_MFMA_BF16X3 || !d.proj) return false;
  if (!d.in || !d.in_sp || !d.zero_line || !std3x3(d) || !d.fuse_out || d.fuse_dim < 1 || d.fuse_dim > 4 || d.Cout != 16) return false;
  if (d.out || d.out2 || d.in2 || d.dual || d.in_add || d.res || d.post_add || d.relu_pre || d.relu_post || d.sigmoid) return false;
  if ((d.in_co & 31) || (d.in_cs & 31) || d.Cin != 32) return false;
  if ((d.W & 15) || (d.H % RBP) || d.H < 64 || d.TH != d.H || d.TW != d.W || d.OH != d.H || d.OW != d.W) return false;
  if (d.gate && ((d.H | d.W) & 1)) return false;
  return true;
}

// Eligibility (shape only, never the batch size: a forward must not change its arithmetic with the batch): what
// drs_tapconv_sp_supported takes, 32 output channels (the first / last level of every variant: one channel group per wave;
// measured slower than the wave-specialised kernel on the 64-channel layers at 128x128), images of at least 64 rows, and
// weights + epilogue vectors of NPOST images within LDS.  DRS_D3K=0 sends these layers back to the wave-specialised
// kernel; bits: 1 plain, 2 with shortcut input, 4 conv1 + skip pair, 8 fused projection.
bool drs_conv3x3_direct_sp_supported(const TapConv& d, int impl) {
  static const int env = getenv("DRS_D3K") ? atoi(getenv("DRS_D3K")) : 15;
  if (!env || impl != DRS_IMPL_MFMA_BF16X3) return false;
  if (d.proj) return false;  // (its own test: drs_conv3x3_direct_sp_proj_supported)
  if (!(env & (d.dual ? 4 : d.fuse_out ? 8 : d.in2 ? 2 : 1))) return false;
  if (!drs_tapconv_sp_supported(d, impl) || !std3x3(d)) return false;
  if ((d.W & 15) || (d.H % RB) || d.H < 64 || d.TH != d.H || d.TW != d.W || d.OH != d.H || d.OW != d.W) return false;
  if (d.in2 && (d.Cin2 > 32 || d.H2 != d.H || d.W2 != d.W)) return false;
  if (d.Cout != 32) return false;
  if (d.dual && (d.Cin != 16 || d.in_cs != 16)) return false;  // (the pair flavour packs two taps of 16 channels into one MFMA)
  TapConv worst = d;
  worst.N = NPOST;
  return direct_lds_bytes(worst) <= 156 * 1024;
}

int drs_launch_conv3x3_direct_sp(const TapConv& d, hipStream_t s) {
  if (d.proj) {
    DRS_REQUIRE(drs_conv3x3_direct_sp_proj_supported(d, DRS_IMPL_MFMA_BF16X3), DRS_ERR_SHAPE, "conv3x3_direct_sp: unsupported folded-projection layer");
    int num_cu = 0;
    const unsigned w_gimage = (unsigned)(3 * 4 * 16 * 16);
    const size_t lds = (size_t)2 * w_gimage + 256;
    const void* kern = d.gate ? reinterpret_cast<const void*>(conv3x3_proj_sp_kernel<true>)
                              : reinterpret_cast<const void*>(conv3x3_proj_sp_kernel<false>);
    {
      const int rc = drs_kernel_prepare(kern, 0, &num_cu);
      if (rc) return rc;
    }
    const long long strips = (long long)d.N * (d.H / RBP) * (d.W >> 4);
    long long blocks = (long long)num_cu * (d.gate ? 2 : 3);
    if (blocks * 4 > strips) blocks = (strips + 3) / 4;
    blocks = (blocks + 7) / 8 * 8;
    if (d.gate) DRS_LAUNCH(conv3x3_proj_sp_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, s, d, w_gimage);
    else DRS_LAUNCH(conv3x3_proj_sp_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, s, d, w_gimage);
    DRS_CHECK_HIP(hipGetLastError());
    return DRS_OK;
  }
  if (d.dual) return direct_launch<4, false, true, false>(d, s);
  if (d.fuse_out) return direct_launch<2, false, false, true>(d, s);
  return d.in2 ? direct_launch<2, true, false, false>(d, s) : direct_launch<2, false, false, false>(d, s);
}
