// 3x3 stride-2 convolution over SP-format activations (split bf16 hi | lo, drs_common.h): the three `downs` layers of
// the encoder (reference UNet_model_superres.py:366; nn.Conv2d(c, c, 3, stride 2, padding 1)).
//
// These layers are HBM-bound (4.8 GFLOP against 170 / 84 / 42 MB), and the tap-list flavour of the lock-step kernel ran
// them at 25 / 18 / 10 % of the HBM rate: a 17 x 33-pixel window (72 KB) staged through registers into LDS for every
// 8 x 16 output patch, 36 KB of weights re-staged per patch and K-chunk, one barrier-separated phase after the other.
// Here nothing is staged: an SP slot (8 channels of a pixel, hi or lo half) IS an MFMA B-operand register group, so a wave
// loads the taps of its 16 output pixels straight from global memory into operand registers (16 bytes per lane, the four
// k-group lanes of a pixel read one 64-byte half line; the left tap is the neighbour lane's right tap: a DPP shift), the block's weights stay in LDS for the whole kernel (all of them for 32 / 64 channels, one 32-channel output
// group per block for 128), and the only synchronisation is the barrier after the weight copy.  The loads of step s + 1
// (one 32-channel K-chunk of one 16-pixel row segment) are in flight while step s is multiplied (two register sets).
// Blocks of one XCD (blockIdx % 8) walk a contiguous eighth of the output rows, so the vertical re-use (input row 2y + 1
// serves output rows y and y + 1) hits that XCD's L2.
// Round 4 built the alternative the round-3 notes asked for on the 64- / 128-channel levels (window staged in LDS with whole-line
// loads, a block = 4 x 16 or 8 x 16 output pixels x ALL output channels, the weights streamed through LDS 64 output channels
// and one K-chunk at a time; parity-green): downs.1 33 -> 31 us, downs.2 38 -> 33 (4-row tiles) / 42 (8-row tiles, 128 blocks).
// Every tile then pulls the layer's whole weight image from L2 (151 MB per launch), 110 - 146 KB per CU and step for 54 - 108
// MFMAs per wave: the same bytes per step as the deep 3x3 kernel moves for 432, i.e. L2-bound at ~5 us per step.  With resident
// weights (this file) the input is re-read per channel group instead; both sit at 115 - 195 MB of L2 -> CU traffic for a 47 MB
// layer, and 256 independent CUs cannot share a weight stream.  Not kept.
#include <stdio.h>
#include <stdlib.h>

#include "conv_epilogue.h"
#include "mfma_policy.h"

namespace {

// the block's weights -> LDS: [operand image][chunk][tap][k-group][GC] slots; 8 loads in flight per thread (a dependent
// load -> store chain per slot would cost a memory round trip per 8 KB of the up to 147 KB)
__device__ __forceinline__ void copy_weights(char* smem, const TapConv& d, int nck, int GC, int n0, unsigned w_gimage, int tid) {
  const int slots = nck * 36 * GC;  // per operand image; a multiple of 512
  const char* wsrc = reinterpret_cast<const char*>(d.w);
  for (int i0 = tid; i0 < 2 * slots; i0 += 512 * 8) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 512;
      if (i < 2 * slots) {
        const int im = i >= slots, idx = im ? i - slots : i;
        const int r = idx / GC, j = idx - r * GC;
        v[u] = *reinterpret_cast<const u32x4*>(wsrc + (size_t)im * w_gimage + ((size_t)r * d.Cout + n0 + j) * 16);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 512;
      if (i < 2 * slots) *reinterpret_cast<u32x4*>(smem + (size_t)i * 16) = v[u];
    }
  }
}

template <int NT>  // 16 * NT output channels per block
__global__ __launch_bounds__(512, 1) void conv_s2_sp_kernel(TapConv d, int nck, unsigned w_gimage) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  constexpr int GC = 16 * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kg = lane >> 4;
  const int ngroups = d.Cout / GC;
  const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
  const int grp = j8 % ngroups, member = j8 / ngroups, members = nb8 / ngroups;  // (the launcher makes nb8 a multiple of ngroups)
  const int n0 = grp * GC;
  const int img = nck * 36 * GC * 16;  // bytes of one operand image of this block's weights: [chunk][tap][k-group][GC] slots
  copy_weights(smem, d, nck, GC, n0, w_gimage, tid);
  __syncthreads();

  // this XCD's rows, this wave's row segments
  const int bw = (d.OW + 15) / 16;
  const int rows = d.N * d.OH;
  const int r_lo = (int)((long long)rows * xcd / 8), r_hi = (int)((long long)rows * (xcd + 1) / 8);
  const int Q = (r_hi - r_lo) * bw;            // wave items of this XCD (per output-channel group)
  const int stride = members * 8, first = member * 8 + wave;
  const int my_items = first < Q ? (Q - first + stride - 1) / stride : 0;
  const int S = my_items * nck;
  if (S == 0) return;
  const int half = drs_sp_group_bytes(d.in_cs);  // bytes from a group's hi half to its lo half
  const char* zero = reinterpret_cast<const char*>(d.zero_line) + kg * 16;
  const char* wbase = smem + ((size_t)kg * GC + lr) * 16;

  // Operand registers of a step: per kernel row the centre and right taps of every lane (pixels 2x, 2x + 1) and, in lanes
  // lr == 0 only, the left tap of the segment's first pixel (2x0 - 1).  The left tap of every other lane is the right tap
  // of its left neighbour (2x - 1 = 2(x - 1) + 1): one DPP row shift at compute time instead of a third of the loads.
  struct Taps { typename P::Frag c[3], r[3], e[3]; };
  Taps fa, fb;
  auto issue = [&](int s, Taps& f) __attribute__((always_inline)) {
    const int it = s / nck, c = s - it * nck;
    const int q = first + it * stride;
    const int row = r_lo + q / bw, xb = q - (q / bw) * bw;
    const int n = row / d.OH, y = row - n * d.OH;
    const int px = min(xb * 16 + lr, d.OW - 1);
    const int iy0 = 2 * y - 1, ix0 = 2 * px - 1;
    const char* base = reinterpret_cast<const char*>(d.in) +
                       ((((long long)n * d.H + iy0) * d.W + ix0) * d.in_cs + d.in_co) * 4 + c * 128 + kg * 16;
    const int rowb = d.W * d.in_cs * 4, pixb = d.in_cs * 4;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const bool row_ok = ky > 0 || iy0 >= 0;  // padding 1: only the top row / left column can miss
      const char* pc = row_ok ? base + ky * rowb + pixb : zero;
      const char* pr = row_ok ? base + ky * rowb + 2 * pixb : zero;
      f.c[ky] = typename P::Frag{*reinterpret_cast<const bf16x8*>(pc), *reinterpret_cast<const bf16x8*>(row_ok ? pc + half : zero)};
      f.r[ky] = typename P::Frag{*reinterpret_cast<const bf16x8*>(pr), *reinterpret_cast<const bf16x8*>(row_ok ? pr + half : zero)};
      if (lr == 0) {
        const bool ok = row_ok && ix0 >= 0;
        const char* pe = ok ? base + ky * rowb : zero;
        f.e[ky] = typename P::Frag{*reinterpret_cast<const bf16x8*>(pe), *reinterpret_cast<const bf16x8*>(ok ? pe + half : zero)};
      }
    }
  };
  auto shift_in = [&](const bf16x8& edge, const bf16x8& right) __attribute__((always_inline)) {
    const u32x4 e = __builtin_bit_cast(u32x4, edge), r = __builtin_bit_cast(u32x4, right);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j)  // row_shr:1: lane lr takes lane lr - 1 of its 16-lane row; lane 0 keeps `edge`
      o[j] = (unsigned)__builtin_amdgcn_update_dpp((int)e[j], (int)r[j], 0x111, 0xf, 0xf, false);
    return __builtin_bit_cast(bf16x8, o);
  };
  f32x4 acc[1][NT];
  // bias of this lane's channels (SP order: 8 consecutive channels per tile pair), loaded ONCE: a load inside the step
  // loop would be waited for behind the prefetched operand loads of the next step (one in-order counter)
  float bias8[NT / 2][8];
#pragma unroll
  for (int pr = 0; pr < NT / 2; ++pr) {
    const float4 a = d.bias ? *reinterpret_cast<const float4*>(d.bias + n0 + pr * 32 + kg * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b = d.bias ? *reinterpret_cast<const float4*>(d.bias + n0 + pr * 32 + kg * 8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    bias8[pr][0] = a.x; bias8[pr][1] = a.y; bias8[pr][2] = a.z; bias8[pr][3] = a.w;
    bias8[pr][4] = b.x; bias8[pr][5] = b.y; bias8[pr][6] = b.z; bias8[pr][7] = b.w;
  }
  // epilogue of one row segment: bias (+ ReLU / per-image add if the descriptor asks), split, full-line stores (the lane
  // pair lr / lr ^ 8 exchanges halves so that lanes lr < 8 write the hi slots and lanes lr >= 8 the lo slots of 8 pixels)
  auto store_row = [&](int n, int y, int x0) __attribute__((always_inline)) {
    const bool lo = lr < 8;
    const int pl = lr & 7;
#pragma unroll
    for (int pr = 0; pr < NT / 2; ++pr) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = acc[0][2 * pr][j] + bias8[pr][j]; v[4 + j] = acc[0][2 * pr + 1][j] + bias8[pr][4 + j]; }
      if (d.relu_pre || d.relu_post) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = drs_maxf(v[j], 0.f);
      }
      u32x4 H, L;
      drs_sp_split8(v, H, L);
      const u32x4 got = drs_dpp_swap8(lo ? L : H);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tx = x0 + pl + 8 * h;
        if (tx < d.OW) {
          const size_t opix = ((size_t)n * d.OH + y) * d.OW + tx;
          char* gp = reinterpret_cast<char*>(d.out) + (opix * d.out_cs + d.out_co + n0 + pr * 32) * 4 + (lo ? 0 : 64) + kg * 16;
          drs_store16(gp, (h == 0) ? (lo ? H : got) : (lo ? got : L));
        }
      }
    }
  };
  auto compute = [&](int s, const Taps& f) __attribute__((always_inline)) {
    const int it = s / nck, c = s - it * nck;
    if (c == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const char* wc = wbase + (size_t)c * 36 * GC * 16;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const typename P::Frag left{shift_in(f.e[ky].hi, f.r[ky].hi), shift_in(f.e[ky].lo, f.r[ky].lo)};
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const typename P::Frag& a = kx == 0 ? left : (kx == 1 ? f.c[ky] : f.r[ky]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const typename P::Frag wf = P::load(wc, (size_t)img, (size_t)((ky * 3 + kx) * 4 * GC + t * 16) * 16);
          acc[0][t] = P::mma(wf, a, acc[0][t]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the weight-fragment reads near their MFMAs (registers)
    }
    if (c == nck - 1) {
      const int q = first + it * stride;
      const int row = r_lo + q / bw, xb = q - (q / bw) * bw;
      const int n = row / d.OH, y = row - n * d.OH;
      store_row(n, y, xb * 16);
    }
  };
  issue(0, fa);
  for (int s = 0; s < S; s += 2) {
    if (s + 1 < S) issue(s + 1, fb);
    compute(s, fa);
    if (s + 2 < S) issue(s + 2, fa);
    if (s + 1 < S) compute(s + 1, fb);
  }
}


// ---- ConvTranspose2d(k = 3, s = 2, p = 1, output_padding = 1) of UpConvBlock.transform (reference :185, :206) -------------
// out[2 iy - 1 + ky][2 ix - 1 + kx] += in[iy][ix] * w[ky][kx]: output phase (py, px) of input pixel (y, x) takes ky = 1 from
// row y (py = 0) or ky = 0 from row y + 1 and ky = 2 from row y (py = 1), the same in x: every one of the nine taps feeds
// exactly one of the four phases, and the operands are the four pixels (y, x), (y, x + 1), (y + 1, x), (y + 1, x + 1).
// Same structure as the stride-2 kernel above: weights resident in LDS, operands global -> registers (two rows of 16
// pixels per K-chunk; the x + 1 operand is the right neighbour lane: a DPP shift, lanes lr == 15 load the segment's
// 17th pixel), four accumulator sets, 2x2 output pixels per lane stored as SP lines.  A wave item is TWO input rows: three
// fragment rows feed both, and every weight-fragment read from LDS serves two rows of MFMAs (0.33 instead of 0.67 reads per
// MFMA: with 36 reads per 54 MFMAs the LDS, not the matrix pipe, set the pace; 75 -> 68 us at 64x64).  The lock-step kernel ran these
// layers (128 -> 128 at 64x64, 64 -> 64 at 128x128) at 86 / 110 us; 335 MB at 5.7 TB/s would be 59 us for the larger.
template <int NT>
__global__ __launch_bounds__(512, 1) void convt_sp_kernel(TapConv d, int nck, unsigned w_gimage) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  constexpr int GC = 16 * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kg = lane >> 4;
  const int ngroups = d.Cout / GC;
  const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
  const int grp = j8 % ngroups, member = j8 / ngroups, members = nb8 / ngroups;
  const int n0 = grp * GC;
  const int img = nck * 36 * GC * 16;
  copy_weights(smem, d, nck, GC, n0, w_gimage, tid);
  __syncthreads();

  const int bw = d.W / 16;  // (W is a multiple of 16: a lane's right neighbour is always a real pixel or the edge load)
  const int rows = d.N * (d.H >> 1);  // a wave item = TWO input rows x 16 pixels (H is even): every weight fragment read serves both
  const int r_lo = (int)((long long)rows * xcd / 8), r_hi = (int)((long long)rows * (xcd + 1) / 8);
  const int Q = (r_hi - r_lo) * bw;
  const int stride = members * 8, first = member * 8 + wave;
  const int my_items = first < Q ? (Q - first + stride - 1) / stride : 0;
  const int S = my_items * nck;
  if (S == 0) return;
  const int half = drs_sp_group_bytes(d.in_cs);
  const char* zero = reinterpret_cast<const char*>(d.zero_line) + kg * 16;
  const char* wbase = smem + ((size_t)kg * GC + lr) * 16;

  struct Rows { typename P::Frag a[3], e[3]; };  // a[i]: pixel (y + i, x), i = 0..2; e[i] (lanes lr == 15): pixel (y + i, x0 + 16)
  Rows fa, fb;
  auto issue = [&](int s, Rows& f) __attribute__((always_inline)) {
    const int it = s / nck, c = s - it * nck;
    const int q = first + it * stride;
    const int row = r_lo + q / bw, xb = q - (q / bw) * bw;
    const int hh = d.H >> 1;
    const int n = row / hh, y = 2 * (row - n * hh);
    const int px = xb * 16 + lr;
    const char* base = reinterpret_cast<const char*>(d.in) +
                       ((((long long)n * d.H + y) * d.W + px) * d.in_cs + d.in_co) * 4 + c * 128 + kg * 16;
    const int rowb = d.W * d.in_cs * 4, pixb = d.in_cs * 4;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const bool row_ok = y + dy < d.H;  // output_padding: the last odd output row / column sees zeros beyond the image
      const char* p = row_ok ? base + dy * rowb : zero;
      f.a[dy] = typename P::Frag{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(row_ok ? p + half : zero)};
      if (lr == 15) {
        const bool ok = row_ok && px + 1 < d.W;
        const char* pe = ok ? base + dy * rowb + pixb : zero;
        f.e[dy] = typename P::Frag{*reinterpret_cast<const bf16x8*>(pe), *reinterpret_cast<const bf16x8*>(ok ? pe + half : zero)};
      }
    }
  };
  auto shift_in = [&](const bf16x8& edge, const bf16x8& own) __attribute__((always_inline)) {
    const u32x4 e = __builtin_bit_cast(u32x4, edge), r = __builtin_bit_cast(u32x4, own);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j)  // row_shl:1: lane lr takes lane lr + 1 of its 16-lane row; lane 15 keeps `edge`
      o[j] = (unsigned)__builtin_amdgcn_update_dpp((int)e[j], (int)r[j], 0x101, 0xf, 0xf, false);
    return __builtin_bit_cast(bf16x8, o);
  };
  f32x4 acc[2][4][NT];  // [input row of the pair][output phase][channel tile]
  float bias8[NT / 2][8];
#pragma unroll
  for (int pr = 0; pr < NT / 2; ++pr) {
    const float4 a = d.bias ? *reinterpret_cast<const float4*>(d.bias + n0 + pr * 32 + kg * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b = d.bias ? *reinterpret_cast<const float4*>(d.bias + n0 + pr * 32 + kg * 8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    bias8[pr][0] = a.x; bias8[pr][1] = a.y; bias8[pr][2] = a.z; bias8[pr][3] = a.w;
    bias8[pr][4] = b.x; bias8[pr][5] = b.y; bias8[pr][6] = b.z; bias8[pr][7] = b.w;
  }
  auto store_item = [&](int r, int n, int y, int x0) __attribute__((always_inline)) {
    const bool lo = lr < 8;
    const int pl = lr & 7;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
      for (int pr = 0; pr < NT / 2; ++pr) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = acc[r][ph][2 * pr][j] + bias8[pr][j]; v[4 + j] = acc[r][ph][2 * pr + 1][j] + bias8[pr][4 + j]; }
        u32x4 H, L;
        drs_sp_split8(v, H, L);
        const u32x4 got = drs_dpp_swap8(lo ? L : H);
        const size_t orow = ((size_t)n * d.OH + 2 * y + (ph >> 1)) * d.OW;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const size_t opix = orow + 2 * (x0 + pl + 8 * h) + (ph & 1);
          char* gp = reinterpret_cast<char*>(d.out) + (opix * d.out_cs + d.out_co + n0 + pr * 32) * 4 + (lo ? 0 : 64) + kg * 16;
          drs_store16(gp, (h == 0) ? (lo ? H : got) : (lo ? got : L));
        }
      }
  };
  auto compute = [&](int s, const Rows& f) __attribute__((always_inline)) {
    const int it = s / nck, c = s - it * nck;
    if (c == 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][ph][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const char* wc = wbase + (size_t)c * 36 * GC * 16;
    // input row r of the pair takes tap row dy from fragment row r + dy: three fragment rows serve both
    typename P::Frag right[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) right[i] = typename P::Frag{shift_in(f.e[i].hi, f.a[i].hi), shift_in(f.e[i].lo, f.a[i].lo)};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            if ((ky == 0 ? 1 : 0) != dy || (kx == 0 ? 1 : 0) != dx) continue;
            const int ph = (ky == 1 ? 0 : 1) * 2 + (kx == 1 ? 0 : 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const typename P::Frag wf = P::load(wc, (size_t)img, (size_t)((ky * 3 + kx) * 4 * GC + t * 16) * 16);
#pragma unroll
              for (int r = 0; r < 2; ++r) acc[r][ph][t] = P::mma(wf, dx == 0 ? f.a[r + dy] : right[r + dy], acc[r][ph][t]);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (c == nck - 1) {
      const int q = first + it * stride;
      const int row = r_lo + q / bw, xb = q - (q / bw) * bw;
      const int hh = d.H >> 1;
      const int n = row / hh, y = 2 * (row - n * hh);
      store_item(0, n, y, xb * 16);
      store_item(1, n, y + 1, xb * 16);
    }
  };
  issue(0, fa);
  for (int s = 0; s < S; s += 2) {
    if (s + 1 < S) issue(s + 1, fb);
    compute(s, fa);
    if (s + 2 < S) issue(s + 2, fa);
    if (s + 1 < S) compute(s + 1, fb);
  }
}

template <int NT, bool CONVT = false>
int s2_launch(const TapConv& d, int nck, unsigned w_gimage, hipStream_t s) {
  auto kern = CONVT ? convt_sp_kernel<2> : conv_s2_sp_kernel<NT>;
  const size_t lds = (size_t)2 * nck * 36 * 16 * NT * 16;
  int num_cu = 0;
  {
    const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(kern), 160 * 1024, &num_cu);
    if (rc) return rc;
  }
  const int ngroups = d.Cout / (16 * NT);
  int per_xcd = num_cu / 8;                     // one block per CU; blocks of an XCD split into the channel groups
  per_xcd = per_xcd / ngroups * ngroups;
  if (per_xcd < ngroups) per_xcd = ngroups;
  DRS_LAUNCH(kern, dim3((unsigned)(8 * per_xcd)), dim3(512), lds, s, d, nck, w_gimage);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// NT for this layer (0 = the kernel does not take it): all output channels in one block if their weights fit into LDS
int s2_tiles(const TapConv& d) {
  const int nck = d.Cin / 32;
  const size_t budget = 156 * 1024;
  if (d.Cout % 64 == 0 && d.Cout <= 64 && (size_t)2 * nck * 36 * 64 * 16 <= budget) return 4;
  if ((size_t)2 * nck * 36 * 32 * 16 <= budget) return 2;
  return 0;
}

}  // namespace

bool drs_conv_s2_sp_supported(const TapConv& d, int impl) {
  static const int env = getenv("DRS_S2K") ? atoi(getenv("DRS_S2K")) : 1;
  if (!env || impl != DRS_IMPL_MFMA_BF16X3) return false;
  if (!d.in || !d.in_sp || !d.out || !d.out_sp || !d.zero_line || d.mode != 0 || d.ntaps != 9 || d.wtaps_total != 9) return false;
  if (d.in_stride != 2 || d.out_scale != 1 || d.out_oy || d.out_ox) return false;
  for (int i = 0; i < 9; ++i)
    if (d.dy[i] != i / 3 - 1 || d.dx[i] != i % 3 - 1 || d.wtap[i] != i) return false;
  if (d.in2 || d.out2 || d.fuse_out || d.dual || d.gate || d.in_add || d.res || d.sigmoid || d.out_nchw) return false;
  if ((d.H & 1) || (d.W & 1) || d.OH != d.H / 2 || d.OW != d.W / 2 || d.TH != d.OH || d.TW != d.OW) return false;
  if (d.Cin % 32 || d.Cout % 32 || (d.in_cs & 31) || (d.in_co & 31) || (d.out_cs & 31) || (d.out_co & 31)) return false;
  if (d.post_add || d.bias2) return false;
  return s2_tiles(d) != 0;
}

int drs_launch_conv_s2_sp(const TapConv& d, hipStream_t s) {
  const int nck = d.Cin / 32;
  const unsigned w_gimage = (unsigned)((size_t)nck * 9 * 4 * d.Cout * 16);  // bytes of the hi image in the packed weights
  return s2_tiles(d) == 4 ? s2_launch<4>(d, nck, w_gimage, s) : s2_launch<2>(d, nck, w_gimage, s);
}

bool drs_convt_sp_supported(const TapConv& d, int impl) {
  static const int env = getenv("DRS_S2K") ? atoi(getenv("DRS_S2K")) : 1;
  if (!env || impl != DRS_IMPL_MFMA_BF16X3) return false;
  if (!d.in || !d.in_sp || !d.out || !d.out_sp || !d.zero_line || d.mode != DRS_TAPMODE_CONVT || d.ntaps != 9 || d.wtaps_total != 9) return false;
  if (d.in_stride != 1 || d.out_scale != 2 || d.out_oy || d.out_ox) return false;
  for (int i = 0; i < 9; ++i)
    if (d.wtap[i] != i) return false;
  if (d.in2 || d.out2 || d.fuse_out || d.dual || d.gate || d.in_add || d.res || d.sigmoid || d.out_nchw) return false;
  if (d.post_add || d.bias2 || d.relu_pre || d.relu_post) return false;
  if ((d.W & 15) || (d.H & 1) || d.OH != 2 * d.H || d.OW != 2 * d.W || d.TH != d.H || d.TW != d.W) return false;
  if (d.Cin % 32 || d.Cout % 32 || (d.in_cs & 31) || (d.in_co & 31) || (d.out_cs & 31) || (d.out_co & 31)) return false;
  return (size_t)2 * (d.Cin / 32) * 36 * 32 * 16 <= 156 * 1024;
}

int drs_launch_convt_sp(const TapConv& d, hipStream_t s) {
  const int nck = d.Cin / 32;
  const unsigned w_gimage = (unsigned)((size_t)nck * 9 * 4 * d.Cout * 16);
  return s2_launch<2, true>(d, nck, w_gimage, s);  // (two rows per wave: 32 channels per block keep the accumulators at 64 registers)
}
