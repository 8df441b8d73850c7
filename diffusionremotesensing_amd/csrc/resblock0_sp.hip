// The first residual block of the encoder as ONE launch (reference ResConvBlock.forward, UNet_model_superres.py:153-172,
// block 0 with its skip convolution :353-355):
//   h   = relu(BN1(conv1(x))) + relu(time_mlp(t)) + skip(x)        16 -> 32 channels, 3x3 + 3x3
//   out = relu(BN2(conv2(h)) + BNs(shortcut(x)))                    32 -> 32 channels, 3x3 + 1x1
// Run as two launches (conv3x3_direct_sp.hip: the pair flavour, then conv2 with the shortcut as one more tap) h crosses
// HBM twice at the highest resolution: 134 MB written and read back of the 600 MB the two launches move, and both run at a
// third of the MFMA rate.  Here a GROUP of four waves owns a patch of 8 rows x 16 columns:
//   phase A  h on the patch's 10 x 18 window (the halo is recomputed: 41 % more conv1 work), 12 MFMA pixel blocks of 16
//            consecutive window pixels, three per wave.  The group first stages the 12 x 20 pixels of x around the patch in
//            LDS (coalesced 64-byte pixels, zeros outside the image: the tile IS conv1's zero padding) and every operand
//            is a ds_read at lane address + tap offset: a lane reads the pixel of ITS tap - two taps of 16 channels share
//            the K = 32 of an MFMA, k-groups 0-1 tap 2j, k-groups 2-3 tap 2j + 1, as in the pair flavour of the direct
//            kernel.  (Gathering the same operands from global memory, 30 scattered loads per wave and patch, cost 2.6 k of
//            a wave's 13 k cycles per patch in the timeline build.)  The next patch's tile travels global ->
//            registers during phase A and registers -> LDS once every wave of the group has left phase A.
//            The result is split into bf16 hi | lo - the rounding the SP tensor between the two launches had - and written
//            into an LDS window in the rotated pixel-major layout of the wave-specialised kernel (slot s of pixel p at
//            position (s + p) & 7).  Window pixels outside the image are conv2's zero padding: zeros, not conv1 of padded x.
//   phase B  conv2 from the LDS window (shifted taps are shifted LDS addresses: no lane shuffles), the 1x1 shortcut of x as
//            one more tap, ReLU, SP store.  A wave owns two rows of the patch.
// All weights (41 + 37 + 4 KB) stay in LDS for the lifetime of the persistent block.  A block is TWO such groups (waves
// 0-3 and 4-7: waves w and w + 4 share a SIMD) with a window and an x tile each, synchronised by three LDS counters per group, never by a
// workgroup barrier, and started half a patch apart: a third of a patch's cycles are epilogues (bias / ReLU / hi | lo split,
// window and global stores) in which the wave issues no MFMA - measured with all eight waves in lockstep on one 16 x 16
// patch: 7.2 k of 17.7 k cycles per patch with the matrix pipe idle - and the other group's MFMA phase fills them.
#include <stdio.h>
#include <stdlib.h>

#include "conv_epilogue.h"
#include "mfma_policy.h"
#include "sp_sync.h"

namespace {

constexpr int TW_ = 16, TH_ = 8, IW = TW_ + 2, IH = TH_ + 2, NPIX = IW * IH, NBLK_A = (NPIX + 15) / 16;  // 180 window pixels, 12 pixel blocks
constexpr int IMG1 = 5 * 4 * 64 * 16, IMG2 = 36 * 32 * 16, IMGS = 4 * 32 * 16, WIN = NPIX * 128;
constexpr int NPOST = 16;  // images whose time-embedding rows are staged in LDS (later images read them from memory)
constexpr int XPLANE = (TH_ + 4) * (TW_ + 4) * 16 + 64, XT = 4 * XPLANE;  // x tile: four slot planes of 240 pixels (+ 64 bytes: bank shift)

#ifdef DRS_SP_TIMELINE  // tools/build_tl.sh: per-phase s_memtime sums of waves 0 and 7 of block 0
__device__ unsigned long long drs_rb0_tl[32];
#define RB_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl[i] += t_ - tl_last; tl_last = t_; } while (0)
#else
#define RB_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(512, 1) void resblock0_kernel(ResBlock0Desc d, unsigned w1_gimage, unsigned w2_gimage,
                                                           unsigned ws_gimage, int debug) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  using Frag = typename P::Frag;
  // The three products of a split-bf16 multiply-accumulate are a dependent chain on one accumulator, and a dependent MFMA
  // issues a full latency (8 passes) after its predecessor: the chains of a step are emitted ROUND-ROBIN (product m of
  // every chain, then product m + 1), never chain by chain (measured: 26-31 cycles per MFMA with the compiler's order of
  // P::mma calls, a wave alone on its SIMD).
  auto mfma = [](const bf16x8& w, const bf16x8& x, const f32x4& c) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
  };
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kg = lane >> 4;
  char* sW1 = smem;               // [image 2][tap pair 5][k-group 4][64 channels: conv1 | skip] slots of 16 bytes
  char* sW2 = sW1 + 2 * IMG1;     // [image 2][tap 9][k-group 4][32]
  char* sWs = sW2 + 2 * IMG2;     // [image 2][k-group 4][32]
  char* sWinAll = sWs + 2 * IMGS; // two h windows (one per group): pixel p = py * 18 + px at p * 128, operand slot s at ((s + p) & 7) * 16
  char* sXAll = sWinAll + 2 * WIN;  // two x tiles (one per group)
  float* sB1 = reinterpret_cast<float*>(sXAll + 2 * XT);  // [64] conv1 | skip biases
  float* sB2 = sB1 + 64;                              // [32] conv2 + shortcut bias
  unsigned* sCntRaw = reinterpret_cast<unsigned*>(sB2 + 32);  // [group 2][window written, window read], [group 2] x tile written: monotonic counters
  float* sPost = sB2 + 32 + 8;                        // [min(N, NPOST)][32] relu(time_mlp(t)) rows
  const int npost = min(d.N, NPOST);
  {
    const char* w1 = reinterpret_cast<const char*>(d.w1);
    for (int i = tid; i < 2 * 20 * 64; i += 512) {  // tap-major packed source -> two taps per K = 32 (the ninth pairs with zeros)
      const int im = i >= 20 * 64, idx = im ? i - 20 * 64 : i;
      const int ch = idx % 64, kgp = (idx / 64) & 3, j = idx / (4 * 64), tap = 2 * j + (kgp >> 1);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (tap < 9) v = *reinterpret_cast<const u32x4*>(w1 + (size_t)im * w1_gimage + ((size_t)(tap * 4 + (kgp & 1)) * 64 + ch) * 16);
      *reinterpret_cast<u32x4*>(sW1 + (size_t)im * IMG1 + (size_t)idx * 16) = v;
    }
    auto copy = [&](char* dst, const char* src, int bytes) __attribute__((always_inline)) {
      for (int o = tid * 16; o < bytes; o += 512 * 16) *reinterpret_cast<u32x4*>(dst + o) = *reinterpret_cast<const u32x4*>(src + o);
    };
    const char* w2 = reinterpret_cast<const char*>(d.w2);
    const char* wsc = reinterpret_cast<const char*>(d.ws);
    copy(sW2, w2, IMG2);
    copy(sW2 + IMG2, w2 + w2_gimage, IMG2);
    copy(sWs, wsc, IMGS);
    copy(sWs + IMGS, wsc + ws_gimage, IMGS);
    for (int i = tid; i < 64; i += 512) sB1[i] = d.b1[i];
    for (int i = tid; i < 32; i += 512) sB2[i] = d.b2[i] + d.bs[i];
    if (tid < 8) sCntRaw[tid] = 0u;
    for (int i = tid; i < npost * 32; i += 512) sPost[i] = d.temb[(size_t)(i >> 5) * d.temb_cs + (i & 31)];
  }
  __syncthreads();

  // patches: blocks of an XCD (blockIdx % 8) take a contiguous eighth (neighbouring patches share their halo in that L2)
  const int grp = wave >> 2, gw = wave & 3;  // group, wave inside the group
  char* sWin = sWinAll + grp * WIN;
  const sp_flag_ptr cWritten = (sp_flag_ptr)(sCntRaw + 2 * grp), cRead = (sp_flag_ptr)(sCntRaw + 2 * grp + 1);
  const int gx = d.W / TW_, gy = d.H / TH_;
  const int total = d.N * gy * gx;
  const int xcd = blockIdx.x & 7, member = blockIdx.x >> 3, members = gridDim.x >> 3;
  const int t_lo = (int)((long long)total * xcd / 8), t_hi = (int)((long long)total * (xcd + 1) / 8);
  const char* zero = reinterpret_cast<const char*>(d.zero_line);
  const char* xb = reinterpret_cast<const char*>(d.x);
  char* sX = sXAll + grp * XT;  // this group's x tile
  const sp_flag_ptr cTile = (sp_flag_ptr)(sCntRaw + 4 + grp);
  // ---- x tile: 12 x 20 pixels around the patch (origin (ty0 - 2, tx0 - 2)), zeros outside the image, as four planes of
  // 16-byte slots [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15][pixel] (plane stride padded: conflict-free for the loader, whose
  // 16 consecutive lanes are 4 pixels x 4 slots, and for the operand reads, whose 16 lanes are 16 pixels of one slot).
  // The group's 256 lanes move it as 16-byte pieces e = 256 i + 64 gw + lane: pixel e >> 2, slot e & 3 - global reads of
  // whole 64-byte pixels, 1280 consecutive bytes per tile row.
  constexpr int XPIX = (TH_ + 4) * (TW_ + 4), XW = TW_ + 4;
  int tx0 = 0, ty0 = 0, n = 0;
  const int qstep_ = 2 * (int)(gridDim.x >> 3);
  // (patch coordinates advance by a fixed step: no division per patch)
  const int step_x = qstep_ % gx, step_y = (qstep_ / gx) % gy, step_n = qstep_ / (gx * gy);
  int bx = 0, by = 0;  // patch indices of (tx0, ty0)
  auto place_first = [&](int q) __attribute__((always_inline)) {
    bx = q % gx; by = (q / gx) % gy; n = q / (gx * gy);
    tx0 = bx * TW_; ty0 = by * TH_;
  };
  auto place_next = [&]() __attribute__((always_inline)) {
    bx += step_x;
    if (bx >= gx) { bx -= gx; ++by; }
    by += step_y;
    if (by >= gy) { by -= gy; ++n; }
    n += step_n;
    tx0 = bx * TW_; ty0 = by * TH_;
  };
  // this lane's four pieces of a tile: byte offset from the tile's first pixel, validity of the piece index
  unsigned goff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = i * 256 + gw * 64 + lane, pix = min(e >> 2, XPIX - 1);
    const int py = (pix * 3277) >> 16, px = pix - py * XW;  // pix / 20 (exact for pix < 1024)
    goff[i] = (unsigned)((py * d.W + px) * 64 + (e & 3) * 16);
  }
  u32x4 xr[4];
  auto tile_load = [&]() __attribute__((always_inline)) {
    const char* ximg = xb + (size_t)n * d.H * d.W * 64;
    if (ty0 >= 2 && ty0 + TH_ + 2 <= d.H && tx0 >= 2 && tx0 + TW_ + 2 <= d.W) {  // (wave-uniform) every pixel inside the image
      typedef const __attribute__((address_space(1))) char* gptr;
      const unsigned long long gi = (unsigned long long)(ximg + ((long long)(ty0 - 2) * d.W + (tx0 - 2)) * 64);
      // (scalar base + lane offset: not one 64-bit lane address per piece; the value is wave-uniform, said explicitly)
      gptr g = (gptr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(gi >> 32)) << 32) |
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gi));
      asm volatile("" : "+s"(g));
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[i] = *(const __attribute__((address_space(1))) u32x4*)(g + goff[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = i * 256 + gw * 64 + lane, pix = e >> 2;
        const int py = (pix * 3277) >> 16, px = pix - py * XW;
        const int y = ty0 - 2 + py, x = tx0 - 2 + px;
        const bool ok = pix < XPIX && y >= 0 && y < d.H && x >= 0 && x < d.W;
        xr[i] = *reinterpret_cast<const u32x4*>(ok ? ximg + ((size_t)y * d.W + x) * 64 + (e & 3) * 16 : zero);
      }
    }
  };
  auto tile_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = i * 256 + gw * 64 + lane;
      if ((e >> 2) < XPIX) *reinterpret_cast<u32x4*>(sX + (e & 3) * XPLANE + (e >> 2) * 16) = xr[i];
    }
    sp_release(cTile, lane);  // (in order behind the stores)
  };
  // this lane's operand addresses in the tile: window pixel of phase-A block b (+ 1, + 1: the tap offsets are -1 .. 1) and
  // the lane's tap of pair j: k-groups 0-1 multiply tap 2j, k-groups 2-3 tap 2j + 1 (pair 4: tap 8 and zero weights)
  constexpr int MAXB = NBLK_A / 4;  // pixel blocks of a wave in phase A: blocks gw, gw + 4, gw + 8
  static_assert(MAXB * 4 == NBLK_A, "three blocks per wave");
  int wp[MAXB], relL[MAXB], tapL[5];
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int p = min((gw + 4 * b) * 16 + lr, NPIX - 1);
    const int py = (p * 3641) >> 16, px = p - py * IW;  // p / 18 (exact for p < 1024)
    wp[b] = p;
    relL[b] = ((py + 1) * XW + px + 1) * 16 + (kg & 1) * XPLANE;
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int tap = min(2 * j + (kg >> 1), 8);
    tapL[j] = ((tap / 3 - 1) * XW + tap % 3 - 1) * 16;
  }
  auto operand = [&](int b, int j) __attribute__((always_inline)) {
    const char* p = sX + relL[b] + tapL[j];
    return Frag{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(p + 2 * XPLANE)};
  };
  const int q0 = t_lo + 2 * member + grp, qstep = 2 * members;  // the block's two groups take neighbouring patches
#ifdef DRS_SP_TIMELINE
  unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long tl_begin = __builtin_amdgcn_s_memtime();
  unsigned long long tl_last = tl_begin, tl_n = 0;
  if ((debug & 1) && grp == 1) return;  // one group alone: uncontended phase times (the output is incomplete)
#endif
  if (q0 < t_hi) {
    place_first(q0);
    tile_load();
    tile_store();
    if (q0 + qstep < t_hi) {  // the second patch's tile waits in registers for the first patch's phase A to end
      place_next();
      tile_load();
      // (landed before the loop is entered: loads still pending on the entry edge make the compiler drain the memory counter
      // at the first register write of EVERY iteration - behind the previous patch's output stores)
      asm volatile("" :: "v"(xr[0]), "v"(xr[1]), "v"(xr[2]), "v"(xr[3]));
    }
  }
  // the second group starts half a patch late (one MFMA phase): from then on its epilogues meet the first group's MFMA phases
  if (grp == 1) { __builtin_amdgcn_s_sleep(64); __builtin_amdgcn_s_sleep(32); }
  unsigned k = 0;  // patches done by this group
  u32x4 kept[4] = {};  // the previous patch's stored registers (see the output epilogue)
  int cbx = q0 % gx, cby = (q0 / gx) % gy, cn = q0 / (gx * gy);  // the CURRENT patch ((tx0, ty0, n) run ahead: tile loads)
  for (int q = q0; q < t_hi; q += qstep, ++k) {
#ifdef DRS_SP_TIMELINE
    ++tl_n;
#endif
    RB_STAMP(7);
    const int cty0 = cby * TH_, ctx0 = cbx * TW_;
    sp_poll_lds(cTile, 4u * (k + 1), d.fault);  // this patch's x tile is complete
    const bool more = q + qstep < t_hi;  // (its tile is in registers: loaded before the previous patch's output stores)
    // ================= phase A: h on the 10 x 18 window =================
    f32x4 acc[MAXB][4];
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (operands and weight fragments one step ahead of their MFMAs; the order is pinned: left alone the scheduler sinks the
    // reads behind the first MFMAs of a step to save registers and the wave waits for the LDS in every step)
    // A step = one tap pair x TWO channel tiles: six accumulator chains, so a dependent MFMA follows its predecessor at a
    // distance of six (at a distance of three - one tile per step - the loop ran at 23 cycles per MFMA instead of 16).
    const char* w1lane = sW1 + ((size_t)kg * 64 + lr) * 16;
    Frag a[2][MAXB], wfa[2][2];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) a[0][b] = operand(b, 0);
#pragma unroll
    for (int u = 0; u < 2; ++u) wfa[0][u] = P::load(w1lane, (size_t)IMG1, (size_t)(u * 16) * 16);
#pragma unroll
    for (int st = 0; st < 10; ++st) {  // step = (tap pair j, tile pair h): tiles 2h, 2h + 1
      const int j = st >> 1, h = st & 1;
      if (st + 1 < 10) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
          wfa[(st + 1) & 1][u] = P::load(w1lane, (size_t)IMG1, (size_t)(((st + 1) >> 1) * 4 * 64 + (((st + 1) & 1) * 2 + u) * 16) * 16);
      }
      if (h == 0 && j + 1 < 5) {
#pragma unroll
        for (int b = 0; b < MAXB; ++b) a[(j + 1) & 1][b] = operand(b, j + 1);
      }
      {
        const Frag (&wf)[2] = wfa[st & 1];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int b = 0; b < MAXB; ++b) acc[b][2 * h + u] = mfma(wf[u].lo, a[j & 1][b].hi, acc[b][2 * h + u]);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int b = 0; b < MAXB; ++b) acc[b][2 * h + u] = mfma(wf[u].hi, a[j & 1][b].lo, acc[b][2 * h + u]);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int b = 0; b < MAXB; ++b) acc[b][2 * h + u] = mfma(wf[u].hi, a[j & 1][b].hi, acc[b][2 * h + u]);
      }
      if (h == 0 && j + 1 < 5) __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * MAXB, 0);
      else if (st + 1 < 10) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 18, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" :: "v"(kept[0]), "v"(kept[1]), "v"(kept[2]), "v"(kept[3]));  // (end of their allocation)
    RB_STAMP(0);  // phase A operand reads + MFMA
    // the shortcut operand of phase B (the centre pixels of this wave's two rows; k-groups 2-3 meet zero weights)
    Frag xs[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const char* p = sX + ((2 + 2 * gw + r) * XW + 2 + lr) * 16 + (kg & 1) * XPLANE;
      xs[r] = Frag{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(p + 2 * XPLANE)};
    }
    // h = relu(conv1 + b1) + temb[n] + (skip + bs): tiles 0, 1 (conv1) and 2, 3 (skip) of the same lane are the same channels
    float bm[8], bs[8], te[8];
    {
      const float4 m0 = *reinterpret_cast<const float4*>(sB1 + kg * 8), m1 = *reinterpret_cast<const float4*>(sB1 + kg * 8 + 4);
      const float4 s0 = *reinterpret_cast<const float4*>(sB1 + 32 + kg * 8), s1 = *reinterpret_cast<const float4*>(sB1 + 32 + kg * 8 + 4);
      float4 e0, e1;
      if (cn < npost) {
        e0 = *reinterpret_cast<const float4*>(sPost + cn * 32 + kg * 8); e1 = *reinterpret_cast<const float4*>(sPost + cn * 32 + kg * 8 + 4);
      } else {
        const float* gp = d.temb + (size_t)cn * d.temb_cs + kg * 8;
        e0 = *reinterpret_cast<const float4*>(gp); e1 = *reinterpret_cast<const float4*>(gp + 4);
      }
      bm[0] = m0.x; bm[1] = m0.y; bm[2] = m0.z; bm[3] = m0.w; bm[4] = m1.x; bm[5] = m1.y; bm[6] = m1.z; bm[7] = m1.w;
      bs[0] = s0.x; bs[1] = s0.y; bs[2] = s0.z; bs[3] = s0.w; bs[4] = s1.x; bs[5] = s1.y; bs[6] = s1.z; bs[7] = s1.w;
      te[0] = e0.x; te[1] = e0.y; te[2] = e0.z; te[3] = e0.w; te[4] = e1.x; te[5] = e1.y; te[6] = e1.z; te[7] = e1.w;
    }
    RB_STAMP(1);  // last MFMAs + epilogue constants
    sp_poll_lds(cRead, 4u * k, d.fault);  // every wave of the group has the previous patch's window rows in registers
    RB_STAMP(2);  // wait 1
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
      float v[8];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          v[t * 4 + i] = (drs_maxf(acc[b][t][i] + bm[t * 4 + i], 0.f) + (acc[b][t + 2][i] + bs[t * 4 + i])) + te[t * 4 + i];  // (the two-launch path's order)
      u32x4 hh, hl;
      drs_sp_split8(v, hh, hl);
      int p = wp[b];
      asm volatile("" : "+v"(p));  // (coordinates recomputed here: nothing held across the phases)
      const int py = (p * 3641) >> 16, px = p - py * IW;
      const int y = cty0 - 1 + py, x = ctx0 - 1 + px;
      const bool inside = y >= 0 && y < d.H && x >= 0 && x < d.W;  // outside: conv2's zero padding
      if (!inside) { hh = u32x4{0u, 0u, 0u, 0u}; hl = u32x4{0u, 0u, 0u, 0u}; }
      if ((gw + 4 * b) * 16 + lr < NPIX) {
        char* wpix = sWin + p * 128;
        const int s0 = ((kg + p) & 7) * 16;
        *reinterpret_cast<u32x4*>(wpix + s0) = hh;
        *reinterpret_cast<u32x4*>(wpix + (s0 ^ 64)) = hl;  // slot + 4 (mod 8)
      }
    }
    sp_release(cWritten, lane);  // (the LDS executes a wave's operations in order: the add lands behind the stores)
    sp_poll_lds(cWritten, 4u * (k + 1), d.fault);  // the window is complete: every wave of the group is past phase A ...
    if (more) tile_store();                        // ... and its x tile may be replaced
    RB_STAMP(3);  // window stores + wait 2 + tile stores
    // ================= phase B: out = relu(conv2(h) + shortcut(x) + b) on rows 2 * gw, 2 * gw + 1 =================
    f32x4 o[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int t = 0; t < 2; ++t) o[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* w2lane = sW2 + ((size_t)kg * 32 + lr) * 16;
    auto win = [&](int row, int kx) __attribute__((always_inline)) {
      const int p = (2 * gw + row) * IW + lr + kx;
      const char* wpix = sWin + p * 128;
      const int s0 = ((kg + p) & 7) * 16;
      return Frag{*reinterpret_cast<const bf16x8*>(wpix + s0), *reinterpret_cast<const bf16x8*>(wpix + (s0 ^ 64))};
    };
    // (the four window rows of the wave's two output rows are read once: row r + 1 is tap row ky + 1 of output row r and
    //  tap row ky of output row r + 1; rows 2 and 3 are fetched under the MFMAs of tap rows 0 and 1: three rows live)
    Frag wr_[4][3];
    auto win_row = [&](int row) __attribute__((always_inline)) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) wr_[row][kx] = win(row, kx);
    };
    win_row(0);
    win_row(1);
    {
      const char* wslane = sWs + ((size_t)kg * 32 + lr) * 16;
      Frag wfb[2][2];  // [step parity][channel tile]
      wfb[0][0] = P::load(w2lane, (size_t)IMG2, 0);
      wfb[0][1] = P::load(w2lane, (size_t)IMG2, (size_t)16 * 16);
#pragma unroll
      for (int st = 0; st < 10; ++st) {  // steps: the nine taps of conv2, then the shortcut; four accumulator chains each
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (st + 1 < 9) wfb[(st + 1) & 1][t] = P::load(w2lane, (size_t)IMG2, (size_t)((st + 1) * 4 * 32 + t * 16) * 16);
          else if (st + 1 < 10) wfb[(st + 1) & 1][t] = P::load(wslane, (size_t)IMGS, (size_t)(t * 16) * 16);
        }
        if (st == 0) win_row(2);
        if (st == 3) {
          win_row(3);
          sp_release(cRead, lane);  // (behind the reads, in order: the next window may be written once all four waves are here)
        }
        const int ky = st < 9 ? st / 3 : 0, kx = st < 9 ? st % 3 : 0;
        const Frag& x0f = st < 9 ? wr_[ky][kx] : xs[0];
        const Frag& x1f = st < 9 ? wr_[ky + 1][kx] : xs[1];
        const Frag (&wf)[2] = wfb[st & 1];
#pragma unroll
        for (int t = 0; t < 2; ++t) { o[0][t] = mfma(wf[t].lo, x0f.hi, o[0][t]); o[1][t] = mfma(wf[t].lo, x1f.hi, o[1][t]); }
#pragma unroll
        for (int t = 0; t < 2; ++t) { o[0][t] = mfma(wf[t].hi, x0f.lo, o[0][t]); o[1][t] = mfma(wf[t].hi, x1f.lo, o[1][t]); }
#pragma unroll
        for (int t = 0; t < 2; ++t) { o[0][t] = mfma(wf[t].hi, x0f.hi, o[0][t]); o[1][t] = mfma(wf[t].hi, x1f.hi, o[1][t]); }
        if (st == 0 || st == 3) __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);  // next weights + a window row
        else if (st + 1 < 10) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);     // next weights
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the tile after the next one: into registers BEFORE this patch's output stores (one in-order memory counter: loads
    // issued behind the stores would return behind them, and registers the stores read cannot be reloaded before they left)
    if (q + 2 * qstep < t_hi) {
      place_next();
      tile_load();
    }
    RB_STAMP(4);  // window reads, phase B MFMA (incl. the shortcut), next tile loads issued
    {
      // out = relu(acc + b), split into bf16 hi | lo, stored as full 128-byte lines (lanes lr < 8 write the hi slots of
      // pixels lr and lr + 8, lanes lr >= 8 their lo slots: conv_epilogue.h, tile_epilogue_sp_pre).  The stored registers
      // stay allocated until the next patch's phase A is over (`kept`): a register a store in flight reads cannot be
      // rewritten before the store has left, and the compiler guards that with a full drain of the memory counter -
      // which was the first thing the next patch did (1.5 k cycles per patch).
      const float4 b0 = *reinterpret_cast<const float4*>(sB2 + kg * 8), b1 = *reinterpret_cast<const float4*>(sB2 + kg * 8 + 4);
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      const bool lo = lr < 8;
      const size_t pix0 = ((size_t)cn * d.H + cty0 + 2 * gw) * d.W + ctx0 + (lr & 7);
      char* g = reinterpret_cast<char*>(d.out) + pix0 * 128 + (lo ? 0 : 64) + kg * 16;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = drs_maxf(o[r][0][i] + bb[i], 0.f); v[4 + i] = drs_maxf(o[r][1][i] + bb[4 + i], 0.f); }
        u32x4 H, L;
        drs_sp_split8(v, H, L);
        const u32x4 got = drs_dpp_swap8(lo ? L : H);  // lr < 8 receives the partner's hi, lr >= 8 the partner's lo
        kept[2 * r] = lo ? H : got;
        kept[2 * r + 1] = lo ? got : L;
        char* gr = g + (size_t)r * d.W * 128;
        drs_store16(gr, kept[2 * r]);
        drs_store16(gr + 8 * 128, kept[2 * r + 1]);
      }
    }
    RB_STAMP(5);  // epilogue B
    cbx += step_x;
    if (cbx >= gx) { cbx -= gx; ++cby; }
    cby += step_y;
    if (cby >= gy) { cby -= gy; ++cn; }
    cn += step_n;
  }
#ifdef DRS_SP_TIMELINE
  if (blockIdx.x == 0 && (wave == 0 || wave == 7) && lane == 0) {
    const int o = wave == 0 ? 0 : 16;
    for (int i = 0; i < 8; ++i) drs_rb0_tl[o + i] = tl[i];
    drs_rb0_tl[o + 8] = tl_n;
    drs_rb0_tl[o + 9] = __builtin_amdgcn_s_memtime() - tl_begin;
  }
#endif
}

size_t resblock0_lds(int N) { return (size_t)2 * (IMG1 + IMG2 + IMGS) + 2 * WIN + 2 * XT + (size_t)(64 + 32 + 8 + 32 * (N < NPOST ? N : NPOST)) * 4; }

}  // namespace

// Shape gate: the 16 -> 32 -> 32 block on images that split into 16 x 16 patches.  DRS_RB0=0 keeps the two launches.
bool drs_resblock0_supported(int Cin, int Cout, int H, int W) {
  static const bool env = !(getenv("DRS_RB0") && atoi(getenv("DRS_RB0")) == 0);
  return env && Cin == 16 && Cout == 32 && H >= 32 && W >= 32 && H % TH_ == 0 && W % TW_ == 0;
}

int drs_launch_resblock0(const ResBlock0Desc& d, hipStream_t s) {
  DRS_REQUIRE(drs_resblock0_supported(16, 32, d.H, d.W), DRS_ERR_SHAPE, "resblock0: H=%d W=%d", d.H, d.W);
  DRS_REQUIRE(d.x && d.w1 && d.b1 && d.temb && d.w2 && d.b2 && d.ws && d.bs && d.out && d.zero_line, DRS_ERR_ARG, "resblock0: null pointer");
  if (d.N == 0) return DRS_OK;
  int num_cu = 0;
  {
    const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(resblock0_kernel), 160 * 1024, &num_cu);
    if (rc) return rc;
  }
  // global operand images (drs_launch_pack_conv_mfma): [chunk 1][tap][k-group 4][channels] slots, one image per bf16 half
  const unsigned w1_gimage = 9u * 4u * 64u * 16u, w2_gimage = 9u * 4u * 32u * 16u, ws_gimage = 4u * 32u * 16u;
  const int blocks = num_cu / 8 * 8;  // one block per CU
  DRS_LAUNCH(resblock0_kernel, dim3((unsigned)blocks), dim3(512), resblock0_lds(d.N), s, d, w1_gimage, w2_gimage, ws_gimage, 0);
  DRS_CHECK_HIP(hipGetLastError());
#ifdef DRS_SP_TIMELINE
  {
    unsigned long long h[32];
    const int dbg = getenv("DRS_RB0_DEBUG") ? atoi(getenv("DRS_RB0_DEBUG")) : 0;
    hipEvent_t e0, e1;
    float ms = 0.f;
    DRS_CHECK_HIP(hipEventCreate(&e0)); DRS_CHECK_HIP(hipEventCreate(&e1));
    DRS_CHECK_HIP(hipEventRecord(e0, s));
    DRS_LAUNCH(resblock0_kernel, dim3((unsigned)blocks), dim3(512), resblock0_lds(d.N), s, d, w1_gimage, w2_gimage, ws_gimage, dbg);  // timed repeat
    DRS_CHECK_HIP(hipEventRecord(e1, s));
    DRS_CHECK_HIP(hipStreamSynchronize(s));
    DRS_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    DRS_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(drs_rb0_tl), sizeof(h)));
    for (int w = 0; w < 2; ++w) {
      const unsigned long long* t = h + 16 * w;
      const double sc = t[8] ? 1.0 / (double)t[8] : 0.0;
      fprintf(stderr, "resblock0 %dx%d wave %d: %.1f us, %llu patches, alive %llu ticks (%.0f / patch) | top %.0f  A-mfma %.0f  A-epi %.0f  bar1 %.0f  "
              "winstore+bar2 %.0f  B-mfma %.0f  B-epi %.0f\n", d.H, d.W, w ? 7 : 0, ms * 1e3, t[8], t[9], t[9] * sc, t[7] * sc, t[0] * sc, t[1] * sc,
              t[2] * sc, t[3] * sc, t[4] * sc, t[5] * sc);
    }
  }
#endif
  return DRS_OK;
}
