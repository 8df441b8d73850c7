// Wave-specialised 3x3 stride-1 tap-convolution over SP-format activations in the "FL" arithmetic: fp16 main product +
// both cross terms of a PAIR of taps as ONE block-scaled FP6 instruction.  Same layers, same HBM formats, same item / step /
// counter protocol as tapconv_sp_kernel (conv_mfma_sp.hip: conv1 / conv2 (+ fused 1x1 shortcut) of the residual blocks,
// ups.*.conv, the att-halves of up_convs.*; reference UNet_model_superres.py:153-172,197-207,377); what changes is the
// arithmetic between the window image and the accumulators:
//   split bf16 (conv_mfma_sp.hip):  x ~ xh + xl, w ~ wh + wl (bf16 each);  w x ~ wh xh + wh xl + wl xh      3 x 16 pipe cycles
//   here:  x = xm + xr, w = wm + wl with xm = fp16(x), wm = fp16(w) (11 bits), remainders ~2^-12 of the values:
//          w x ~ wm xm  [v_mfma_f32_16x16x32_f16, 16 cycles]
//              + q(wl) q(xm) + q(wm) q(xr)   [e2m3 codes, one power-of-two scale per 32-channel block and pixel / weight row]
//          and the cross terms of TWO taps fill the K = 128 of one v_mfma_scale_f32_16x16x128_f8f6f4 (fp6 runs at 4x the bf16
//          rate: 16 cycles): 9 + 5 instructions per row, tile and 32-channel chunk instead of 27.
//   Accuracy: the remainders sit three bits lower than bf16's, so 4-bit cross terms land at ~2^-16.
// Nothing else in the network changes: the inputs are SP tensors ([32 x bf16 hi | 32 x bf16 lo] per pixel and 32-channel group,
// x' = hi + lo carries 16 significant bits); the MOVER waves convert a window to the FL line while it passes through their
// registers, with VALU time they did not use:
//   FL line (128 bytes per window pixel and chunk, 16-byte slots, rotated by the pixel index like the SP window image):
//     slots 0-3  fp16 main xm, k-group = slot                        (the B operand of the fp16 instruction, as is)
//     slot 4 | 6 e2m3 codes of xm (bytes 0-15 | 16-23), then the block's E8M0 scale in byte 8 of slot 6
//     slot 5 | 7 the same for the remainder xr = x' - xm
//   a lane of a cross-term fragment reads slot 4 + (kg & 1) and slot 6 + (kg & 1) of its pixel: 8 consecutive registers =
//   6 registers of codes + the scale register of the instruction; k-groups 0 / 1 take the pair's first tap, 2 / 3 its second.
//   Mover lane = HALF a pixel (16 channels): 4 x 16-byte loads, packed-fp16 two-sum (xm = hi + lo rounded once, xr exact),
//   block maximum over the lane pair by one DPP step, v_cvt_scalef32_pk32_fp6_f16 for the codes (hardware RNE, saturating).
// Weights ("FL images", drs_launch_fl_repack below, derived from the packed split-bf16 images so that BatchNorm folding, the
// SP output-row permutation and partial packs stay where they are):
//     main  [chunk][tap 9][k-group 4][Cout] x 16 bytes (8 x fp16)      = the geometry of ONE split-bf16 image
//     cross [chunk][row 36][Cout] x 16 bytes: pairs 0-3: row = pair * 8 + half * 4 + group, pair 4: 32 + half * 2 + group;
//           group = 2 * (second tap of the pair) + (0: q(wl), meets q(xm) | 1: q(wm), meets q(xr)); half 0 = code bytes 0-15,
//           half 1 = code bytes 16-23 + the row block's scale in byte 8
//     tap pairs (kernel column c, row k): 0 = (c0k0 | c0k1), 1 = (c1k0 | c1k1), 2 = (c0k2 | c1k2), 3 = (c2k0 | c2k1), 4 = (c2k2 | -)
//     second input (the block's 1x1 shortcut): main [chunk][k-group 4][Cout], cross [chunk][half * 2 + group][Cout]
// Ring slot of kernel column j in LDS (1 KB rows of 64 channels): 12 main rows (ky * 4 + k-group), then the cross rows of
// the pairs COMPLETED by that column: column 0: pair 0; column 1: pairs 1, 2; column 2: pairs 3, 4 -> 20 + 28 + 24 KB.
//   LDS = 2 x 41 KB windows + 72 KB ring + counters + zero block + 2 epilogue-constant slots = 155.6 KB.
// 64 output channels per item only (launches with fewer items than CUs keep the split-bf16 kernel's 32-channel groups).
#include <stdio.h>
#include <stdlib.h>

#include "conv_epilogue.h"
#include "mfma_policy.h"
#include "sp_sync.h"

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef DRS_FL_TIMELINE  // in-kernel stage timing of block 0 (tools/build_variant.sh tl -DDRS_FL_TIMELINE): consumer waves 0 / 4, mover 0
__device__ unsigned long long drs_fl_tl[64];
#define FL_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl[i] += t_ - tl_last; tl_last = t_; } while (0)
#else
#define FL_STAMP(i) do { } while (0)
#endif

// The movers' polls: LDS counters guarding LDS slots - the relaxed form (sp_sync.h).  The acquire form also drains the wave's
// vector-memory counter: a mover would wait at every poll for the global loads it has just issued for LATER use.
#ifdef DRS_FL_ACQPOLL
#define FL_MPOLL sp_poll
#else
#define FL_MPOLL sp_poll_lds
#endif

struct FlGeom {
  static constexpr int IW = 18;
  static constexpr int WBUF = 41 * 1024;                     // one window buffer (328 lines of 128 bytes)
  static constexpr int RING0 = 0, RING1 = 20 * 1024, RING2 = 48 * 1024, RING = 72 * 1024;
  static constexpr int CROSS = 12 * 1024;                    // cross rows of a ring slot follow its 12 main rows
  static constexpr int EPI = 768;
  static constexpr int LDS = 2 * WBUF + RING + 64 + 64 + 2 * EPI;
};

__device__ __forceinline__ void fl_bump_prio(sp_flag_ptr f, unsigned step_base, int lane) {  // (sp_bump_prio of conv_mfma_sp.hip)
  unsigned old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifndef DRS_FL_NO_PRIO
  const unsigned rank = (unsigned)__builtin_amdgcn_readfirstlane((int)old) - step_base;
  if (rank >= 4) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(0);
#else
  (void)old; (void)step_base;
#endif
}

// ---- SP -> FL for HALF a pixel (16 channels of one 32-channel block) -------------------------------------------------------
// h0 / h1: the two bf16 hi slots, l0 / l1: the two lo slots.  Results: the two fp16 main slots; the lane's half of both code
// blocks (3 dwords each); the E8M0 scale bytes of the two blocks (common to the lane pair: the maximum goes over both lanes).
// Packed fp16 arithmetic: hi and lo are exact in fp16 (8 significant bits each; below fp16's subnormal step lo is truncated,
// above its range cvt_pkrtz saturates at 65504), m = hi + lo is x' rounded ONCE to fp16, t = m - hi and r = lo - t are exact
// (Fast2Sum, |hi| >= |lo|).  Block scale: the power of two that puts the block maximum into [3.75, 7.5] (e2m3's top binade).
struct FlHalf { u32x4 m0, m1; unsigned qm[3], qr[3]; unsigned sm, sr; bool over; };  // over: the block's maximum is not finite / at fp16's edge

__device__ __forceinline__ unsigned fl_block_exp(const f16x2 (&v)[8], bool& over) {  // biased (E8M0) exponent of the lane pair's block scale
  f16x2 mx = v[0], mn = v[0];
#pragma unroll
  // (IEEE maximum / minimum: hipcc folds the chains into v_pk_maximum3_f16 / v_pk_minimum3_f16 - 8 instructions instead of 15 -
  //  and a NaN element reaches the result, where `over` catches it)
  for (int d = 1; d < 8; ++d) { mx = __builtin_elementwise_maximum(mx, v[d]); mn = __builtin_elementwise_minimum(mn, v[d]); }
  const f16x2 am = __builtin_elementwise_maximum(mx, -mn);  // |.| maxima of the even / odd elements (non-negative: ordered as integers)
  const unsigned ab = __builtin_bit_cast(unsigned, am);
  unsigned a16 = max(ab & 0xffffu, ab >> 16);
  a16 = max(a16, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a16, 0xB1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]: the pair's other lane
  over = a16 >= 0x7bffu;  // 65504 (where cvt_pkrtz saturates), inf, NaN: fp16 does not hold this block
  const unsigned f = a16 >> 10, mant = a16 & 0x3ffu;
  // amax = (1 + mant / 1024) 2^(f - 15); scale 2^(f - 17) puts it into [4, 8); above 7.5 (mant > 0.875 * 1024) one more
  return max(f, 1u) + (127u - 17u) + (mant > 0x380u ? 1u : 0u);
}
__device__ __forceinline__ u32x6 fl_codes16(const f16x2 (&v)[8], unsigned eb) {  // e2m3 codes of 16 values: dwords 0-2 of the result
  f16x16 lo;
#pragma unroll
  for (int d = 0; d < 8; ++d) { lo[2 * d] = v[d][0]; lo[2 * d + 1] = v[d][1]; }
  const f16x32 src = __builtin_shufflevector(lo, lo, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, -1, -1, -1, -1, -1, -1, -1, -1,
                                             -1, -1, -1, -1, -1, -1, -1, -1);
  return __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(src, __uint_as_float(eb << 23));
}
__device__ __forceinline__ void fl_convert_half(const u32x4& h0, const u32x4& h1, const u32x4& l0, const u32x4& l1, FlHalf& o) {
  const unsigned hd[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
  const unsigned ld[8] = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
  f16x2 m[8], r[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const f16x2 hp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(hd[d] << 16), __uint_as_float(hd[d] & 0xffff0000u)));
    const f16x2 lp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(ld[d] << 16), __uint_as_float(ld[d] & 0xffff0000u)));
    m[d] = hp + lp;
    const f16x2 t = m[d] - hp;
    r[d] = lp - t;
  }
  o.sm = fl_block_exp(m, o.over);
  // The remainders' scale from their own maximum.  (|r| <= 2^-11 |m| element by element, so 2^-11 of the main scale would hold
  // them too, ~25 instructions cheaper per half pixel: measured on the trained-like fixture of tests/test_gpu_fl.py it costs
  // accuracy - forward max-rel 7.8e-5 -> 1.05e-4, the largest remainder is on average well under its bound - and buys
  // nothing measurable: DRS_FL_RSCALE_BOUND.)
#ifdef DRS_FL_RSCALE_BOUND
  o.sr = o.sm > 12u ? o.sm - 11u : 1u;
#else
  bool over_r;
  o.sr = fl_block_exp(r, over_r);
#endif
  const u32x6 cm = fl_codes16(m, o.sm), cr = fl_codes16(r, o.sr);
#pragma unroll
  for (int j = 0; j < 3; ++j) { o.qm[j] = cm[j]; o.qr[j] = cr[j]; }
  o.m0 = u32x4{__builtin_bit_cast(unsigned, m[0]), __builtin_bit_cast(unsigned, m[1]), __builtin_bit_cast(unsigned, m[2]), __builtin_bit_cast(unsigned, m[3])};
  o.m1 = u32x4{__builtin_bit_cast(unsigned, m[4]), __builtin_bit_cast(unsigned, m[5]), __builtin_bit_cast(unsigned, m[6]), __builtin_bit_cast(unsigned, m[7])};
}
// Stores of one converted half pixel.  line = the pixel's 128-byte line in the window buffer; a_*: byte offsets inside the line
// (the rotation applied): the two main slots, and per code block the dword the lane's first code word goes to (half 0: dword 0
// of slot 4 / 5; half 1: dword 3) and the 12-byte piece (half 0: dwords 0-2 of slot 4 / 5; half 1: slot 6 / 7: two code words +
// the scale).  (Half 0 writes its first word twice, with the same value: one instruction shape for both halves.)
__device__ __forceinline__ void fl_store_half(char* line, const FlHalf& f, bool h, int a_m0, int a_m1, int a_w4, int a_p46, int a_w5, int a_p57) {
  *reinterpret_cast<u32x4*>(line + a_m0) = f.m0;
  *reinterpret_cast<u32x4*>(line + a_m1) = f.m1;
  *reinterpret_cast<unsigned*>(line + a_w4) = f.qm[0];
  *reinterpret_cast<u32x3*>(line + a_p46) = u32x3{h ? f.qm[1] : f.qm[0], h ? f.qm[2] : f.qm[1], h ? f.sm : f.qm[2]};
  *reinterpret_cast<unsigned*>(line + a_w5) = f.qr[0];
  *reinterpret_cast<u32x3*>(line + a_p57) = u32x3{h ? f.qr[1] : f.qr[0], h ? f.qr[2] : f.qr[1], h ? f.sr : f.qr[2]};
}

template <bool HAS2>
__global__ __launch_bounds__(768, 1) void tapconv_fl_kernel(TapConv d, MfmaGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = FlGeom;
  constexpr int IW = G::IW, WBUF = G::WBUF, BNB = 64, RPW = 4, NT = 2, BN = 32, TH = 16, TW = 16;
  char* sWin = smem;                 // [buffer 2][window pixel][rotated slot 8] x 16 bytes (FL lines)
  char* sW = smem + 2 * WBUF;        // weight ring: three column slots (header)
  sp_flag_ptr sCR = (sp_flag_ptr)(sW + G::RING);  // counters: as tapconv_sp_kernel
  sp_flag_ptr sCL = sCR + 3;
  sp_flag_ptr sWL = sCR + 6;
  sp_flag_ptr sWR = sCR + 8;
  char* sZero = sW + G::RING + 64;   // 64 zero bytes: the absent half of a half pair (codes 0, scale 2^-127)
  float* sEpi = reinterpret_cast<float*>(sW + G::RING + 128);
  constexpr int EC = BNB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7 consumers, 8..11 movers
  const bool mover = wid >= 8;
  const int lr = lane & 15, kg = lane >> 4;

  // persistent blocks, XCD-aware item order (as tapconv_sp_kernel)
  const int ngroups = d.Cout / BNB;
  const int nitems = d.N * g.tiles_y * g.tiles_x * ngroups;
  const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
  const int per = (nitems + 7) >> 3;
  const int lo_item = xcd * per, hi_item = min(nitems, lo_item + per);
  const int span = hi_item - lo_item - j8;
  const int my_items = span > 0 ? (span + nb8 - 1) / nb8 : 0;
  const int nck = g.nchunks + (HAS2 ? g.nchunks2 : 0);
  const int S = my_items * nck;
  if (S == 0) return;
  auto step_kind = [&](int c_, bool& second_, int& cc_) __attribute__((always_inline)) {
    second_ = HAS2 && c_ >= g.nchunks;  // the one-tap steps of the second input follow the 3x3 steps (see tapconv_sp_kernel)
    cc_ = second_ ? c_ - g.nchunks : c_;
  };
  auto item_of = [&](int ordinal, int& n_, int& ty0_, int& tx0_, int& n0_) __attribute__((always_inline)) {
    int it = lo_item + ordinal * nb8 + j8;
    n0_ = (it % ngroups) * BNB;
    it /= ngroups;
    tx0_ = (it % g.tiles_x) * TW;
    it /= g.tiles_x;
    ty0_ = (it % g.tiles_y) * TH;
    n_ = it / g.tiles_y;
  };

#ifdef DRS_FL_TIMELINE
  unsigned long long tl[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_amdgcn_s_memtime();
  const unsigned long long tl_begin = tl_last;
#endif
  if (tid < 10) __hip_atomic_store(sCR + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (tid >= 64 && tid < 80) reinterpret_cast<unsigned*>(sZero)[tid - 64] = 0u;
  sp_wait_lds();
  sp_barrier();

  if (mover) {
    // ===================== movers =====================
    // Protocol of conv_sp_movers.inc (counters, bounded polls, the ring-hit rule and the poll behind a filling step), with two
    // changes: the window passes through fl_convert_half on its way into LDS, and its loads are issued a whole step AHEAD
    // (right behind the conversion of the previous window, whose registers they take over) - the conversion sits where the
    // loads of the next step used to start, and a one-tap step no longer exposes a memory round trip.
    const int pw = wid - 8;
    __builtin_amdgcn_s_setprio(3);
    const char* zero = reinterpret_cast<const char*>(d.zero_line) + (lane & 7) * 16;  // (every load below stays inside its 256 bytes)
    const int hh = lane & 1, lp = lane >> 1;  // half of the pixel, pixel inside a round of 32
    // A single wave issues an instruction every ~5 cycles whatever its kind: the loop below is written for FEW instructions -
    // scalar bases advanced by additions, per-lane offsets and LDS destinations computed once, compile-time counts.
    // Rounds R = pw + 4 i, i < 3, of the 18 x 18 window pixels, pixel p = 32 R + lp (11 rounds hold the 324 pixels: the twelfth,
    // mover 3's last, is empty - every mover runs the same three rounds).  32 R = 0 mod 8: the rotation of a lane's pixel is
    // the same in every round - the destination offsets are lane constants.  A one-tap step (the second input: window origin
    // = patch origin, no halo) goes the same way: its 16 x 16 patch pixels are window pixels (py, px) with py, px < 16.
    constexpr int NR = 3;
    const int rot = lp & 7, lpb = lp * 128;
    const int a_m0 = lpb + ((2 * hh + rot) & 7) * 16, a_m1 = lpb + ((2 * hh + 1 + rot) & 7) * 16;
    const int a_w4 = lpb + ((4 + rot) & 7) * 16 + (hh ? 12 : 0), a_p46 = lpb + ((4 + 2 * hh + rot) & 7) * 16;
    const int a_w5 = lpb + ((5 + rot) & 7) * 16 + (hh ? 12 : 0), a_p57 = lpb + ((5 + 2 * hh + rot) & 7) * 16;
    // byte offsets of the lane's half line from the window origin, per round and input (32 bits: a tensor spans < 2 GB).  Pixels
    // beyond the window (the last rounds) / beyond the patch (one-tap steps) take the origin's: loaded, never stored.
    int off1[NR], off2[NR];
    unsigned keep1 = 0u, keep2 = 0u;  // bit i: round i's pixel is stored (3x3 steps / one-tap steps)
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int p = (pw + 4 * i) * 32 + lp;
      const int py = (p * 3641) >> 16, px = p - py * IW;  // p / 18 (exact for p < 1024)
      const bool in1 = py < 18, in2 = py < 16 && px < 16;
      off1[i] = (in1 ? ((py * d.W + px) * d.in_cs) * 4 : 0) + hh * 32;
      off2[i] = HAS2 ? (in2 ? ((py * d.W2 + px) * d.in2_cs) * 4 : 0) + hh * 32 : 0;
      keep1 |= in1 ? 1u << i : 0u;
      keep2 |= in2 ? 1u << i : 0u;
    }
    // scalar strides of the two inputs; the origins have the window's (-1, -1) folded in
    const unsigned pix1 = (unsigned)d.in_cs * 4u, row1 = (unsigned)d.W * pix1;
    const unsigned long long img1 = (unsigned long long)d.H * row1;
    const char* in1o = reinterpret_cast<const char*>(d.in) + (long long)d.in_co * 4 - (long long)row1 - (long long)pix1;
    const unsigned pix2 = HAS2 ? (unsigned)d.in2_cs * 4u : 0u, row2 = HAS2 ? (unsigned)d.W2 * pix2 : 0u;
    const unsigned long long img2 = HAS2 ? (unsigned long long)d.H2 * row2 : 0ull;
    const char* in2o = HAS2 ? reinterpret_cast<const char*>(d.in2) + (long long)d.in2_co * 4 : nullptr;
    // Weights.  Piece i of this mover in column `col` = the 1 KB row idx = 4 i + pw of the column's ring slot: 3 main rows
    // (ky = i, k-group pw: memory rows (i * 3 + col) * 4 + pw, i.e. 12 rows apart) and 2 / 4 / 3 cross rows (4 rows apart from
    // first_cross_row(col) + pw).  So: per-lane offsets for the 3 + 4 strides, computed once; per step one scalar base for the
    // chunk; per column two scalar additions.  A one-tap step has rows 0-3 main (k-group) and 4 cross rows (half * 2 + group):
    // this mover's main row pw and cross row pw, which go where the 3x3 step's first main row and first cross row go.
    const unsigned row_b = (unsigned)d.Cout * 16u;  // bytes of one operand row (all output channels) in memory
    const unsigned lane16 = (unsigned)lane * 16u;
    unsigned vo_m[3], vo_c[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) vo_m[i] = lane16 + (unsigned)i * 12u * row_b;
#pragma unroll
    for (int j = 0; j < 4; ++j) vo_c[j] = lane16 + (unsigned)j * 4u * row_b;
    const char* wg = reinterpret_cast<const char*>(d.w_fl) + (size_t)pw * row_b;
    const char* wg2 = HAS2 ? reinterpret_cast<const char*>(d.w2_fl) + (size_t)pw * row_b : nullptr;
    const size_t chunk_b = (size_t)36 * row_b, chunk2_b = (size_t)4 * row_b;
    char* dlane = sW + pw * 1024 + lane * 16;  // this lane's 16 bytes inside row idx = pw of slot 0
    u32x4 wrA[6], wrB[7], ww[NR][4];
    // step bookkeeping: chunk inside the item, item ordinal and coordinates
    struct Step { int c, ord, n, ty0, tx0, n0; };
    auto advance = [&](Step& s) __attribute__((always_inline)) {
      if (++s.c == nck) s.c = 0;
      if (s.c == 0) item_of(++s.ord, s.n, s.ty0, s.tx0, s.n0);
    };
    // window loads of a step, rounds [r0, r1): 4 x 16 bytes per round (hi slots 2 hh, 2 hh + 1, lo slots 2 hh, 2 hh + 1 of the pixel)
    auto load_window = [&](const Step& s, int r0, int r1) __attribute__((always_inline)) {
      const bool sec = HAS2 && s.c >= g.nchunks;
      const int cc = sec ? s.c - g.nchunks : s.c;
      const char* base;
      bool inside;
      if (sec) {
        base = in2o + (unsigned long long)s.n * img2 + (unsigned long long)((unsigned)s.ty0 * row2 + (unsigned)s.tx0 * pix2) + (unsigned)(cc * 128);
        inside = s.ty0 + 16 <= d.H2 && s.tx0 + 16 <= d.W2;
      } else {
        base = in1o + (unsigned long long)s.n * img1 + (unsigned long long)((unsigned)s.ty0 * row1 + (unsigned)s.tx0 * pix1) + (unsigned)(cc * 128);
        inside = s.ty0 >= 1 && s.tx0 >= 1 && s.ty0 + 17 <= d.H && s.tx0 + 17 <= d.W;
      }
      if (inside) {  // every stored pixel lies inside the image
#pragma unroll
        for (int i = 0; i < NR; ++i)
          if (i >= r0 && i < r1) {
            const char* src = base + (unsigned)(sec ? off2[i] : off1[i]);
            ww[i][0] = *reinterpret_cast<const u32x4*>(src);
            ww[i][1] = *reinterpret_cast<const u32x4*>(src + 16);
            ww[i][2] = *reinterpret_cast<const u32x4*>(src + 64);
            ww[i][3] = *reinterpret_cast<const u32x4*>(src + 80);
          }
      } else {
        int lpx = lp;  // opaque: what is derived from it is recomputed here, not kept in registers across the loop
        asm volatile("" : "+v"(lpx));
        const int oy = sec ? s.ty0 : s.ty0 - 1, ox = sec ? s.tx0 : s.tx0 - 1;
        const int Hh = sec ? d.H2 : d.H, Wd = sec ? d.W2 : d.W, ext = sec ? 16 : 18;
#pragma unroll
        for (int i = 0; i < NR; ++i)
          if (i >= r0 && i < r1) {
            const int p = (pw + 4 * i) * 32 + lpx;
            const int py = (p * 3641) >> 16, px = p - py * IW;
            const bool ok = py < ext && px < ext && (unsigned)(oy + py) < (unsigned)Hh && (unsigned)(ox + px) < (unsigned)Wd;
            const char* src = ok ? base + (unsigned)(sec ? off2[i] : off1[i]) : zero;
            ww[i][0] = *reinterpret_cast<const u32x4*>(src);
            ww[i][1] = *reinterpret_cast<const u32x4*>(src + 16);
            ww[i][2] = *reinterpret_cast<const u32x4*>(src + 64);
            ww[i][3] = *reinterpret_cast<const u32x4*>(src + 80);
          }
      }
    };
    auto store_window = [&](bool sec, char* buf) __attribute__((always_inline)) {
      const unsigned keep = (HAS2 && sec) ? keep2 : keep1;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        FlHalf f;
#ifdef DRS_FL_COPYMOVER  // speed experiment (wrong numbers): what the kernel does when its movers only copy
        f.m0 = ww[i][0]; f.m1 = ww[i][1];
        f.qm[0] = ww[i][2][0]; f.qm[1] = ww[i][2][1]; f.qm[2] = ww[i][2][2]; f.qr[0] = ww[i][3][0]; f.qr[1] = ww[i][3][1]; f.qr[2] = ww[i][3][2];
        f.sm = 120u; f.sr = 110u;
#else
        fl_convert_half(ww[i][0], ww[i][1], ww[i][2], ww[i][3], f);
#endif
        if ((keep >> i) & 1u) fl_store_half(buf + (pw + 4 * i) * 4096, f, hh != 0, a_m0, a_m1, a_w4, a_p46, a_w5, a_p57);
#ifndef DRS_FL_COPYMOVER
        if (f.over && d.fault) {  // (never in a healthy network: reported by drs_unet_check_faults as DRS_ERR_RANGE, with the place)
          atomicOr(d.fault, 2u);
          d.fault[1] = ((unsigned)d.Cin << 16) | (unsigned)d.Cout;
          d.fault[2] = ((unsigned)d.TH << 16) | (sec ? 1u : 0u);
          d.fault[3] = ((unsigned)blockIdx.x << 16) | ((unsigned)(pw + 4 * i) << 8) | (unsigned)lane;
          d.fault[4] = f.sm;
        }
#endif
        __builtin_amdgcn_sched_barrier(0);  // one round's temporaries at a time
      }
    };

    // Order of a step k (loads in issue order; the in-order vector-memory counter gives the waits).  Everything a step's FIRST
    // stores need was requested during the previous step: its column 0 and its whole window.
    //   poll CR0, store col0(k) | col1(k) [7 | 0], col2(k) [6 | 0] | poll WR, convert + store window k | window of k + 1 [12] |
    //   poll CR1, store col1 | poll CR2, store col2 | col0(k + 1) [5 | 2]      (counts: 3x3 step | one-tap step; a ring hit loads
    //   no weights)
    auto load_col0 = [&](const Step& s, bool hit) __attribute__((always_inline)) {
      const bool sec = HAS2 && s.c >= g.nchunks;
      const int cc = sec ? s.c - g.nchunks : s.c;
      if (hit) return;
      if (sec) {
        const char* w = wg2 + (size_t)cc * chunk2_b + (size_t)s.n0 * 16;
        wrA[0] = *reinterpret_cast<const u32x4*>(w + vo_m[0]);
        wrA[3] = *reinterpret_cast<const u32x4*>(w + (size_t)g.w2_gimage + vo_c[0]);
      } else {
        const char* mb = wg + (size_t)cc * chunk_b + (size_t)s.n0 * 16;  // memory row (i * 3 + 0) * 4 + pw
        const char* cb = mb + (size_t)g.w_gimage;
#pragma unroll
        for (int i = 0; i < 3; ++i) wrA[i] = *reinterpret_cast<const u32x4*>(mb + vo_m[i]);
#pragma unroll
        for (int j = 0; j < 2; ++j) wrA[3 + j] = *reinterpret_cast<const u32x4*>(cb + vo_c[j]);
      }
    };
    Step cur = {-1, -1, 0, 0, 0, 0};
    advance(cur);
    load_window(cur, 0, 3);
    load_col0(cur, false);
    int ring_n0 = -1;
    bool prev_fill = false;
    for (int k = 0; k < S; ++k) {
      const bool ring_hit = nck == 1 && cur.n0 == ring_n0;  // a one-chunk layer keeps its three columns in the ring
      ring_n0 = cur.n0;
      const bool second = HAS2 && cur.c >= g.nchunks;
      const int cc = second ? cur.c - g.nchunks : cur.c;
      const bool w3 = !ring_hit && !second;  // the step brings a full set of 3x3 weights
      const bool w1 = !ring_hit && second;   // ... the two rows of a one-tap step
      // the chunk's weights (this mover's row offset folded in)
      const char* wcur = second ? wg2 + (size_t)cc * chunk2_b + (size_t)cur.n0 * 16 : wg + (size_t)cc * chunk_b + (size_t)cur.n0 * 16;
      const size_t gim = second ? (size_t)g.w2_gimage : (size_t)g.w_gimage;
      FL_STAMP(9);
      // epilogue constants of a new item (mover 3: its last window round is empty)
      const bool epi_step = cur.c == 0 && pw == 3;
      u32x4 ev = {0u, 0u, 0u, 0u};
      if (epi_step) {
        const float* src = nullptr;
        const float* src2 = nullptr;
        const int which = lane / (EC / 4), o = (lane % (EC / 4)) * 4;
        if (which == 0) { src = d.bias ? d.bias + cur.n0 + o : nullptr; src2 = (HAS2 && d.bias2) ? d.bias2 + cur.n0 + o : nullptr; }
        else if (which == 1) src = d.post_add ? d.post_add + (size_t)cur.n * d.post_cs + cur.n0 + o : nullptr;
        else if (which == 2) src = d.out2 ? d.post2 + (size_t)cur.n * d.post2_cs + cur.n0 + o : nullptr;
        if (src) {
          const float4 a = *reinterpret_cast<const float4*>(src);
          float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
          if (src2) b = *reinterpret_cast<const float4*>(src2);
          ev = u32x4{__float_as_uint(a.x + b.x), __float_as_uint(a.y + b.y), __float_as_uint(a.z + b.z), __float_as_uint(a.w + b.w)};
        }
      }
      FL_STAMP(0);
      if (k >= 1) FL_MPOLL(sCR, 8u * (unsigned)k, d.fault);  // every consumer holds column 0 of step k - 1 in registers
      FL_STAMP(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // column 0, the window (requested a step ago) and the constants have landed
      FL_STAMP(2);
      if (epi_step) {
        float* slot = sEpi + (cur.ord & 1) * (G::EPI / 4);
        if (lane < 3 * (EC / 4)) *reinterpret_cast<u32x4*>(slot + lane * 4) = ev;
      }
      if (w3) {
#pragma unroll
        for (int i = 0; i < 5; ++i) *reinterpret_cast<u32x4*>(dlane + G::RING0 + (i < 3 ? 4 * i : 12 + 4 * (i - 3)) * 1024) = wrA[i];
      } else if (w1) {
        *reinterpret_cast<u32x4*>(dlane + G::RING0) = wrA[0];
        *reinterpret_cast<u32x4*>(dlane + G::RING0 + 12 * 1024) = wrA[3];
      }
      sp_wait_lds();
      if (lane == 0) sp_bump(sCL);
      // ---- columns 1 and 2: in flight during the conversion ----
      if (w3) {
        const char* mb1 = wcur + (size_t)4 * row_b;
        const char* cb1 = wcur + gim + (size_t)8 * row_b;
        const char* mb2 = wcur + (size_t)8 * row_b;
        const char* cb2 = wcur + gim + (size_t)24 * row_b;
#pragma unroll
        for (int i = 0; i < 3; ++i) wrB[i] = *reinterpret_cast<const u32x4*>(mb1 + vo_m[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) wrB[3 + j] = *reinterpret_cast<const u32x4*>(cb1 + vo_c[j]);
#pragma unroll
        for (int i = 0; i < 3; ++i) wrA[i] = *reinterpret_cast<const u32x4*>(mb2 + vo_m[i]);
#pragma unroll
        for (int j = 0; j < 3; ++j) wrA[3 + j] = *reinterpret_cast<const u32x4*>(cb2 + vo_c[j]);
      }
      FL_STAMP(3);
      if (k >= 2) FL_MPOLL(sWR + (k & 1), 8u * (unsigned)(k >> 1), d.fault);  // the consumers left the buffer in step k - 2
      FL_STAMP(4);
      store_window(second, sWin + (k & 1) * WBUF);
      sp_wait_lds();
      if (lane == 0) sp_bump(sWL + (k & 1));
      FL_STAMP(5);
      // the NEXT step's window: a step ahead of its conversion (behind the last step: the same one once more, never used)
      Step nxt = cur;
      if (k + 1 < S) advance(nxt);
      load_window(nxt, 0, 3);
      FL_STAMP(6);
      // (the rule behind a filling step: conv_sp_movers.inc)
      const bool after_fill = prev_fill;
      prev_fill = w3;
      if (k >= 1 && (w3 || after_fill)) FL_MPOLL(sCR + 1, 8u * (unsigned)k, d.fault);
      if (w3) {
        asm volatile("s_waitcnt vmcnt(18)" ::: "memory");  // column 1 has landed (behind it: 6 pieces of column 2, 12 window loads)
#pragma unroll
        for (int i = 0; i < 7; ++i) *reinterpret_cast<u32x4*>(dlane + G::RING1 + (i < 3 ? 4 * i : 12 + 4 * (i - 3)) * 1024) = wrB[i];
      }
      sp_wait_lds();
      if (lane == 0) sp_bump(sCL + 1);
      FL_STAMP(7);
      if (k >= 1 && (w3 || after_fill)) FL_MPOLL(sCR + 2, 8u * (unsigned)k, d.fault);
      if (w3) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<u32x4*>(dlane + G::RING2 + (i < 3 ? 4 * i : 12 + 4 * (i - 3)) * 1024) = wrA[i];
      }
      sp_wait_lds();
      if (lane == 0) sp_bump(sCL + 2);
      // column 0 of the NEXT step, into the registers column 2 has just left: in flight while the consumers of this step's
      // predecessor are still a third of a step from releasing slot 0
      if (k + 1 < S) load_col0(nxt, nck == 1 && nxt.n0 == cur.n0);
      FL_STAMP(8);
      cur = nxt;
    }
  } else {
    // ===================== consumers =====================
    const int rw = wid & 3;         // rows [4 rw, 4 rw + 4) of the patch
    const int ng = (wid >> 2) & 1;  // channels [32 ng, 32 ng + 32) of the item's 64
    // main fragment of window pixel p = B + q + lr (B per wave, q compile time), slot kg: tm[q & 7] + q * 128
    const int B = rw * RPW * IW;
    int tm[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) tm[j] = (B + lr) * 128 + ((kg + B + j + lr) & 7) * 16;
    // Cross-term fragments: a lane reads slots 4 + (kg & 1) and 6 + (kg & 1) of its pixel; k-groups 0 / 1 take the pixel of the
    // pair's first tap, 2 / 3 of its second.  For kg < 2 slot 4 + kg is table entry (q + 4) & 7; for kg >= 2 the pixel lies
    // `delta` further and the slot is kg + 2: entry (q + 2 + delta) & 7, plus delta lines.  delta = 18 (next window row): the
    // same entry, (q + 20) & 7 = (q + 4) & 7, for all lanes; the second slot two entries further.
    const bool upper = kg >= 2;
    const int dl = upper ? IW * 128 : 0;
    const char* wlane = sW + (kg * BNB + ng * BN + lr) * 16;  // this lane's origin inside an operand row group
    f32x4 acc[RPW][NT];
    u32x4 wm[3][NT];
    int c = -1, ord = -1, n = 0, ty0 = 0, tx0 = 0, n0 = 0;
    auto mm16 = [](const u32x4& w, const u32x4& a, const f32x4& cacc) __attribute__((always_inline)) {
      return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), cacc, 0, 0, 0);
    };
    auto mm6 = [](const i32x8& w, const i32x8& x, const f32x4& cacc) __attribute__((always_inline)) {  // e2m3 x e2m3, scales = register 6 of each
      return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, x, cacc, 2, 2, 0, w[6], 0, x[6]);
    };
    auto ld8 = [](const char* p0, const char* p1) __attribute__((always_inline)) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(p0), b = *reinterpret_cast<const u32x4*>(p1);
      return i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
    };
    auto mfrag = [&](const char* buf, int q) __attribute__((always_inline)) {
      return *reinterpret_cast<const u32x4*>(buf + tm[q & 7] + q * 128);
    };
    auto wmain = [&](int slot, int ky, int t) __attribute__((always_inline)) {
      return *reinterpret_cast<const u32x4*>(wlane + slot + (ky * 4 * BNB + t * 16) * 16);
    };
    // cross-term window fragments of output row r; q = window offset of the pair's first tap
    auto xq_row = [&](const char* buf, int q) __attribute__((always_inline)) {  // second tap: one window row below
      const char* b = buf + dl;
      return ld8(b + tm[(q + 4) & 7] + q * 128, b + tm[(q + 6) & 7] + q * 128);
    };
    auto xq_col = [&](const char* buf, int q) __attribute__((always_inline)) {  // second tap: the next pixel of the row
      const int a0 = upper ? tm[(q + 3) & 7] + 128 : tm[(q + 4) & 7];
      const int a1 = upper ? tm[(q + 5) & 7] + 128 : tm[(q + 6) & 7];
      return ld8(buf + a0 + q * 128, buf + a1 + q * 128);
    };
    auto xq_half = [&](const char* buf, int q) __attribute__((always_inline)) {  // no second tap: zeros
      const char* p0 = upper ? sZero : buf + tm[(q + 4) & 7] + q * 128;
      const char* p1 = upper ? sZero + 16 : buf + tm[(q + 6) & 7] + q * 128;
      return ld8(p0, p1);
    };
    // cross-term weight fragments: a full pair = 8 rows (half * 4 + group) at ring offset `off`; a half pair = 4 rows (half * 2 + group)
    auto wq_full = [&](int off, int t) __attribute__((always_inline)) {
      const char* p = wlane + off + t * 256;
      return ld8(p, p + 4 * 1024);
    };
    auto wq_half = [&](int off, int t) __attribute__((always_inline)) {
      const char* p = wlane + off + t * 256;
      return ld8(upper ? sZero : p, upper ? sZero + 16 : p + 2 * 1024);
    };
    // cross-term products of one pair over the wave's four rows: the fragment of row r + 1 is requested before the
    // instructions of row r (the compiler barrier keeps the request order; left alone the scheduler hoists every read of a
    // column to its top and spills)
    // (Releasing a ring slot behind a COUNTED wait - only the reads issued before the release, with the next fragment already
    //  requested - was measured: no gain, and the fragment held across the release cost registers the step loop does not have.)
#define FL_CROSS(XQ, W) { const i32x8 x0_ = XQ(0); FL_CROSS_P(XQ, W, x0_) }
#ifdef DRS_FL_X_NOCROSS  // speed experiment (wrong numbers): no cross-term fragments / instructions
#undef FL_CROSS
#define FL_CROSS(XQ, W) { }
#define FL_CROSS_P(XQ, W, X0) { (void)X0; }
#else
#define FL_CROSS_P(XQ, W, X0)                                                  \
    {                                                                          \
      i32x8 xq_ = X0;                                                          \
      _Pragma("unroll") for (int r = 0; r < RPW; ++r) {                        \
        i32x8 xn_ = xq_;                                                       \
        if (r + 1 < RPW) xn_ = XQ(r + 1);                                      \
        asm volatile("" ::: "memory");                                         \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[r][t] = mm6(W[t], xq_, acc[r][t]); \
        xq_ = xn_;                                                             \
      }                                                                        \
    }
#endif
    // fp16 products of kernel column `col` (weights wm[ky]); PRE: the main weight fragments of the next column (ring offset
    // `next`) replace wm[ky] as soon as the last window row that needs the old ones has been issued
    auto main_col = [&](const char* buf, int col, bool pre, int next, unsigned ltarget) __attribute__((always_inline)) {
      u32x4 af = mfrag(buf, col);
#pragma unroll
      for (int wr = 0; wr < RPW + 2; ++wr) {
        u32x4 afn = af;
        if (wr + 1 < RPW + 2) afn = mfrag(buf, (wr + 1) * IW + col);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int r = wr - ky;
          if (r >= 0 && r < RPW) {
#pragma unroll
#ifndef DRS_FL_X_NOMAIN  // speed experiment (wrong numbers): no fp16 instructions (their fragments are still read)
            for (int t = 0; t < NT; ++t) acc[r][t] = mm16(wm[ky][t], af, acc[r][t]);
#else
            for (int t = 0; t < NT; ++t) acc[r][t][0] += __uint_as_float(wm[ky][t][0] ^ af[t]);
#endif
          }
        }
        if (pre) {
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
            if (wr == ky + RPW - 1) {
              if (ky == 0) sp_poll_lds(sCL + col + 1, ltarget, d.fault);
#pragma unroll
              for (int t = 0; t < NT; ++t) wm[ky][t] = wmain(next, ky, t);
            }
        }
        af = afn;
      }
    };
    for (int k = 0; k < S; ++k) {
      if (++c == nck) c = 0;
      if (c == 0) {
        item_of(++ord, n, ty0, tx0, n0);
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      bool second;
      int cc_unused;
      step_kind(c, second, cc_unused);
      const char* buf = sWin + (k & 1) * WBUF;
      const unsigned ltarget = 4u * (unsigned)(k + 1);
      FL_STAMP(9);
      sp_poll_lds(sWL + (k & 1), 4u * (unsigned)((k >> 1) + 1), d.fault);  // window k is in its buffer
      FL_STAMP(0);
      sp_poll_lds(sCL, ltarget, d.fault);                                   // ... and ring slot 0 of step k
      FL_STAMP(1);
      if (second) {  // one tap at the window origin: ring slot 0 holds 4 main rows (as ky = 0) + a half pair (where pair 0 goes)
        i32x8 wq[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { wm[0][t] = wmain(G::RING0, 0, t); wq[t] = wq_half(G::RING0 + G::CROSS, t); }
        sp_wait_lds();
        fl_bump_prio(sCR, 8u * (unsigned)k, lane);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          const u32x4 af = mfrag(buf, r * IW);
          const i32x8 xq = xq_half(buf, r * IW);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][t] = mm16(wm[0][t], af, acc[r][t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][t] = mm6(wq[t], xq, acc[r][t]);
        }
        sp_wait_lds();
        if (lane == 0) {
          sp_bump(sCR + 1);
          sp_bump(sCR + 2);
          sp_bump(sWR + (k & 1));
        }
      } else {
        // ---- column 0: taps (0, k); pair 0 = (c0k0 | c0k1) ----
        {
          i32x8 wq[NT];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int t = 0; t < NT; ++t) wm[ky][t] = wmain(G::RING0, ky, t);
#pragma unroll
          for (int t = 0; t < NT; ++t) wq[t] = wq_full(G::RING0 + G::CROSS, t);
          sp_wait_lds();
          fl_bump_prio(sCR, 8u * (unsigned)k, lane);  // slot 0 is in registers
          FL_STAMP(2);
          main_col(buf, 0, true, G::RING1, ltarget);
          FL_STAMP(3);
#define XQ0(r_) xq_row(buf, (r_) * IW)
          FL_CROSS(XQ0, wq)
          FL_STAMP(4);
        }
        // ---- column 1: pair 1 = (c1k0 | c1k1), pair 2 = (c0k2 | c1k2) ----
        // (one pair's weight fragments at a time: the slot is released behind the LAST read - the movers refill it a step
        //  ahead, half a step of delay costs nothing, 16 registers do)
        {
          i32x8 wq[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) wq[t] = wq_full(G::RING1 + G::CROSS, t);
          main_col(buf, 1, true, G::RING2, ltarget);
          FL_STAMP(5);
#define XQ1(r_) xq_row(buf, (r_) * IW + 1)
#define XQ2(r_) xq_col(buf, ((r_) + 2) * IW)
          FL_CROSS(XQ1, wq)
#pragma unroll
          for (int t = 0; t < NT; ++t) wq[t] = wq_full(G::RING1 + G::CROSS + 8192, t);
          sp_wait_lds();
          fl_bump_prio(sCR + 1, 8u * (unsigned)k, lane);
          FL_CROSS(XQ2, wq)
          FL_STAMP(6);
        }
        // ---- column 2: pair 3 = (c2k0 | c2k1), pair 4 = (c2k2 | -) ----
        {
          i32x8 wq[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) wq[t] = wq_full(G::RING2 + G::CROSS, t);
          main_col(buf, 2, false, 0, 0u);
          FL_STAMP(7);
#define XQ3(r_) xq_row(buf, (r_) * IW + 2)
#define XQ4(r_) xq_half(buf, ((r_) + 2) * IW + 2)
          FL_CROSS(XQ3, wq)
#pragma unroll
          for (int t = 0; t < NT; ++t) wq[t] = wq_half(G::RING2 + G::CROSS + 8192, t);
          sp_wait_lds();
          fl_bump_prio(sCR + 2, 8u * (unsigned)k, lane);
          FL_CROSS(XQ4, wq)
        }
        sp_wait_lds();  // the last window fragment has been read: the buffer may be refilled (for step k + 2)
        if (lane == 0) sp_bump(sWR + (k & 1));
        FL_STAMP(8);
      }
#ifdef DRS_FL_X_NOEPI  // speed experiment (wrong numbers): what the item epilogues cost
      if (c == nck - 1 && k + 1 == S) {
#else
      if (c == nck - 1) {
#endif
        int lr_e = lr, kg_e = kg;  // (opaque copies: conv_mfma_sp.hip)
        asm volatile("" : "+v"(lr_e), "+v"(kg_e));
        const float* ek = sEpi + (ord & 1) * (G::EPI / 4);
        auto lds8 = [&](const float* p, float (&v)[8]) __attribute__((always_inline)) {
          const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
          v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        };
        SpEpiConst kc;
        const float* e0 = ek + ng * BN + kg_e * 8;
        lds8(e0, kc.bias);
        lds8(e0 + EC, kc.post);
        lds8(e0 + 2 * EC, kc.post2);
        tile_epilogue_sp_pre<RPW, true>(d, acc, kc, n, n0 + ng * BN, ty0, tx0, rw, lr_e, kg_e);
      }
    }
  }
#ifdef DRS_FL_TIMELINE
  if (blockIdx.x == 0 && (wid == 0 || wid == 4 || wid == 8) && lane == 0) {
    const int o = wid == 0 ? 0 : (wid == 4 ? 16 : 32);
    for (int i = 0; i < 10; ++i) drs_fl_tl[o + i] = tl[i];
    drs_fl_tl[o + 10] = (unsigned long long)S;
    drs_fl_tl[o + 11] = __builtin_amdgcn_s_memtime() - tl_begin;
  }
#endif
}

// ---- FL images from packed split-bf16 images -------------------------------------------------------------------------------
// One thread per (chunk, tap, output row): the row's 32 weights of the chunk = hi + lo of the split-bf16 slots (16 significant
// bits of the folded fp32 weight: more than fp16 main + 4-bit remainder keep), main = fp16 (RNE), wl = w - main; e2m3 codes of
// wl and of main by the hardware converter, each with the row block's own power-of-two scale.
__device__ __forceinline__ unsigned fl_scale_exp_f32(float amax) {  // biased exponent of the scale that puts amax into [3.75, 7.5]
  const unsigned b = __float_as_uint(amax);
  const unsigned f = (b >> 23) & 0xffu, mant = b & 0x7fffffu;
  return max(f, 3u) - 2u + (mant > 0x700000u ? 1u : 0u);
}
__global__ void fl_repack_kernel(const char* __restrict__ sp, size_t sp_image, char* __restrict__ dst, int Cout, int nchunks, int taps,
                                 unsigned* __restrict__ flag, unsigned* __restrict__ wmax_bits) {
  const long long total = (long long)nchunks * taps * Cout;
  const size_t main_bytes = (size_t)nchunks * taps * 4 * Cout * 16;
  float lmax = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout), tap = (int)((i / Cout) % taps), c = (int)(i / ((long long)Cout * taps));
    float w[32], wl[32], wmf[32];
    float am = 0.f, al = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t s = (((size_t)c * taps + tap) * 4 + q) * Cout + co;
      const u32x4 h = *reinterpret_cast<const u32x4*>(sp + s * 16), l = *reinterpret_cast<const u32x4*>(sp + sp_image + s * 16);
      f16x8 m8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned hw = h[j >> 1], lw = l[j >> 1];
        const float x = __uint_as_float((j & 1) ? (hw & 0xffff0000u) : (hw << 16)) + __uint_as_float((j & 1) ? (lw & 0xffff0000u) : (lw << 16));
        const _Float16 mh = (_Float16)x;
        w[q * 8 + j] = x;
        wmf[q * 8 + j] = (float)mh;
        wl[q * 8 + j] = x - (float)mh;
        m8[j] = mh;
        am = fmaxf(am, fabsf((float)mh));
        al = fmaxf(al, fabsf(x - (float)mh));
      }
      *reinterpret_cast<f16x8*>(dst + s * 16) = m8;
    }
    const unsigned em = fl_scale_exp_f32(am), el = fl_scale_exp_f32(al);
    f32x16 ae, ao;  // the converter interleaves its two sources: element t = (t even ? first : second)[t / 2]
#pragma unroll
    for (int j = 0; j < 16; ++j) { ae[j] = wl[2 * j]; ao[j] = wl[2 * j + 1]; }
    const u32x6 ql = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(ae, ao, __uint_as_float(el << 23));
#pragma unroll
    for (int j = 0; j < 16; ++j) { ae[j] = wmf[2 * j]; ao[j] = wmf[2 * j + 1]; }
    const u32x6 qm = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(ae, ao, __uint_as_float(em << 23));
    // rows of the tap inside the chunk's cross image
    int row0, hstride;  // row of (half 0, group 0 of this tap), rows between the halves
    if (taps == 9) {
      const int col = tap % 3, ky = tap / 3;
      int pair, second;
      if (ky == 2) { pair = col == 2 ? 4 : 2; second = col == 1 ? 1 : 0; }
      else { pair = col == 0 ? 0 : (col == 1 ? 1 : 3); second = ky; }
      if (pair == 4) { row0 = 32; hstride = 2; }
      else { row0 = pair * 8 + 2 * second; hstride = 4; }
    } else {
      row0 = 0; hstride = 2;
    }
    const int rows = taps == 9 ? 36 : 4;
    char* cross = dst + main_bytes + ((size_t)c * rows) * Cout * 16;
    auto put = [&](int group, const u32x6& q, unsigned e) {
      *reinterpret_cast<u32x4*>(cross + ((size_t)(row0 + group) * Cout + co) * 16) = u32x4{q[0], q[1], q[2], q[3]};
      *reinterpret_cast<u32x4*>(cross + ((size_t)(row0 + hstride + group) * Cout + co) * 16) = u32x4{q[4], q[5], e, 0u};
    };
    put(0, ql, el);  // q(wl): meets q(x main)
    put(1, qm, em);  // q(w main): meets q(x remainder)
    (void)w;
    lmax = fmaxf(lmax, am);
  }
  // range of the layer's folded weights (non-negative floats order like their bit patterns)
  if (lmax > 0.f) atomicMax(wmax_bits, __float_as_uint(lmax));
  if (lmax > 60000.f) atomicOr(flag, 1u);
}
// a layer whose LARGEST weight is tiny would keep most of its weights in fp16's subnormal range
__global__ void fl_range_kernel(unsigned* __restrict__ flag, const unsigned* __restrict__ wmax_bits) {
  if (__uint_as_float(*wmax_bits) < 0.0009765625f) atomicOr(flag, 1u);
}

}  // namespace

size_t drs_fl_image_bytes(int Cout, int Cin, int taps) {  // = the two split-bf16 images of the layer
  return (size_t)2 * drs_cdiv(Cin, 32) * taps * 4 * Cout * 16;
}

int drs_launch_fl_repack(const void* sp_images, void* dst, int Cout, int Cin, int taps, unsigned* flag, hipStream_t s) {
  DRS_REQUIRE(taps == 9 || taps == 1, DRS_ERR_SHAPE, "fl_repack: taps=%d", taps);
  DRS_REQUIRE(flag != nullptr, DRS_ERR_ARG, "fl_repack: null flag");
  const int nchunks = drs_cdiv(Cin, 32);
  const size_t image = (size_t)nchunks * taps * 4 * Cout * 16;
  const long long total = (long long)nchunks * taps * Cout;
  int blocks = (int)((total + 127) / 128);
  if (blocks > 4096) blocks = 4096;
  // flag[0]: the layer's range flag, flag[1]: scratch for the largest |weight| (both zeroed by the caller)
  DRS_LAUNCH(fl_repack_kernel, dim3(blocks), dim3(128), 0, s, (const char*)sp_images, image, (char*)dst, Cout, nchunks, taps, flag, flag + 1);
  DRS_LAUNCH(fl_range_kernel, dim3(1), dim3(1), 0, s, flag, (const unsigned*)(flag + 1));
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// Eligibility: what tapconv_sp_kernel<HAS2, 64> takes (plain flavour: SP in, SP out, optional second input / second output),
// full 32-channel chunks, FL images present.
bool drs_tapconv_fl_supported(const TapConv& d, int impl) {
  if (!d.w_fl || !drs_tapconv_sp_supported(d, impl)) return false;
  if (d.dual || d.fuse_out || !d.out_sp) return false;
  if (d.Cout % 64 != 0 || d.Cin % 32 != 0 || (d.in_cs & 31)) return false;
  if (d.in2 && (!d.w2_fl || d.Cin2 % 32 != 0 || (d.in2_cs & 31))) return false;
  return true;
}

int drs_launch_tapconv_fl(const TapConv& d, const MfmaGeom& g, hipStream_t s) {
  DRS_REQUIRE(g.IH == 18 && g.IW == 18, DRS_ERR_SHAPE, "tapconv_fl: geometry");
  int num_cu = 0;
  auto k0 = tapconv_fl_kernel<false>;
  auto k1 = tapconv_fl_kernel<true>;
  {
    const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(d.in2 ? k1 : k0), 160 * 1024, &num_cu);
    if (rc) return rc;
  }
  static_assert(FlGeom::LDS <= 160 * 1024, "LDS budget");
  const long long nitems = (long long)d.N * g.tiles_x * g.tiles_y * (d.Cout / 64);
  // (No 32-channel-group flavour for launches with few items, unlike drs_launch_tapconv_sp: which arithmetic a layer runs in
  //  must not depend on the batch size - a batch's forward equals the forwards of its images, tests/test_gpu_parity.py.)
  long long blocks = num_cu;
  if (blocks > nitems) blocks = nitems;
  blocks = (blocks + 7) / 8 * 8;
  if (d.in2) DRS_LAUNCH(k1, dim3((unsigned)blocks), dim3(768), (size_t)FlGeom::LDS, s, d, g);
  else DRS_LAUNCH(k0, dim3((unsigned)blocks), dim3(768), (size_t)FlGeom::LDS, s, d, g);
  DRS_CHECK_HIP(hipGetLastError());
#ifdef DRS_FL_TIMELINE
  {
    unsigned long long h[64];
    hipEvent_t e0, e1;
    float ms = 0.f;
    DRS_CHECK_HIP(hipEventCreate(&e0)); DRS_CHECK_HIP(hipEventCreate(&e1));
    DRS_CHECK_HIP(hipEventRecord(e0, s));
    if (d.in2) DRS_LAUNCH(k1, dim3((unsigned)blocks), dim3(768), (size_t)FlGeom::LDS, s, d, g);  // timed repeat (same result)
    else DRS_LAUNCH(k0, dim3((unsigned)blocks), dim3(768), (size_t)FlGeom::LDS, s, d, g);
    DRS_CHECK_HIP(hipEventRecord(e1, s));
    DRS_CHECK_HIP(hipStreamSynchronize(s));
    DRS_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    DRS_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(drs_fl_tl), sizeof(h)));
    const double sc = h[10] ? 1.0 / (double)h[10] : 0.0;
    fprintf(stderr, "fl kernel Cin=%d Cout=%d TH=%d in2=%d: %.1f us, %llu steps/block, wave0 alive %llu ticks (100 MHz: %.1f us), %.0f ticks/step\n",
            d.Cin, d.Cout, d.TH, d.in2 ? d.Cin2 : 0, ms * 1e3, h[10], h[11], h[11] * 0.01, h[11] * sc);
    for (int w = 0; w < 2; ++w) {
      const unsigned long long* t = h + 16 * w;
      fprintf(stderr, "   C%d (ticks/step): epi+loop %.1f pollWL %.1f pollCL0 %.1f rd0 %.1f main0 %.1f x0 %.1f main1 %.1f x1x2 %.1f main2 %.1f x3x4 %.1f\n", 4 * w,
              t[9] * sc, t[0] * sc, t[1] * sc, t[2] * sc, t[3] * sc, t[4] * sc, t[5] * sc, t[6] * sc, t[7] * sc, t[8] * sc);
    }
    const unsigned long long* t = h + 32;
    fprintf(stderr, "   M0 (ticks/step): loop %.1f ldcol0 %.1f pollCR0 %.1f vm0 %.1f stcol0 %.1f pollWR %.1f convert %.1f ld12+win %.1f col1 %.1f col2 %.1f\n",
            t[9] * sc, t[0] * sc, t[1] * sc, t[2] * sc, t[3] * sc, t[4] * sc, t[5] * sc, t[6] * sc, t[7] * sc, t[8] * sc);
  }
#endif
  return DRS_OK;
}
