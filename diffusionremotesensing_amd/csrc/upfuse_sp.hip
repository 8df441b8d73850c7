// ups.i.transform composed with the x-half of up_convs.i: ONE stride-2 transposed convolution of the low-resolution
// UpConvBlock output h instead of ConvTranspose2d(Cc, Cc, 3, 2, 1, 1) -> cat -> the first Cc input channels of
// Conv2d(Cc + Ch, Ch, 3, padding 1) (reference UNet_model_superres.py:197-207,320-322,376-377; UpFuseDesc in drs_common.h).
//
// Per axis, ConvT(k3, s2, p1, op1) followed by conv(k3, p1) reads 3 low-resolution taps at even outputs and 2 at odd ones:
//   u[2m] = w1 x[m],  u[2m+1] = w2 x[m] + w0 x[m+1],  y[o] = v0 u[o-1] + v1 u[o] + v2 u[o+1]
//   y[2m]   = v0w2 x[m-1] + (v0w0 + v1w1 + v2w2) x[m] + v2w0 x[m+1]
//   y[2m+1] = (v0w1 + v1w2) x[m] + (v1w0 + v2w1) x[m+1]
// (pair (kv, kw) belongs to tap t of phase p iff p + kv - kw == 2 (t - 1)), in 2-D 9 + 6 + 6 + 4 = 25 taps per 2 x 2 output
// pixels = 6.25 taps x Cc per output pixel against 2.25 x Cc x (Cc / Ch) + 9 x Cc today: 26.8 instead of 58.0 GFLOP per
// stage at B = 16, 256 x 256, the 4 x Cc-channel high-resolution tensor (67 / 134 / 268 MB) is neither written nor read,
// and the ConvTranspose launch disappears.  The composite weights U[py][px][ty][tx] = sum_c V[:, c] W[:, c] over the pairs
// are an fp32 contraction at pack time (the same class of re-association as the BatchNorm fold).  What the composite gets
// wrong is the output's first row and column: the 3x3 convolution pads u with ZEROS at row / column -1, the composite
// continues the transposed convolution there (u[-1] = w0 x[0]); and the ConvTranspose bias reaches an output pixel through
// the taps that stay inside the image only.  Both are per-edge fp32 vectors (upfuse_edges_kernel: a 1-D composite over the
// first row / column of h, 0.1 % of the layer's work) added in the epilogue of the border tiles.
//
// Kernel structure = tapconv_sp_kernel's (conv_mfma_sp.hip): 8 consumer + 4 mover waves on one CU, persistent blocks in
// XCD-aware item order, 18 x 18 low-resolution window double-buffered in LDS in the rotated pixel-major operand image,
// weights streamed through a ring of LDS slots, monotonic LDS counters instead of barriers.  Differences:
//   item   = 16 x 16 low-resolution cells (32 x 32 output pixels) x 32 output channels; K-step = one 32-channel chunk;
//   ring   = 3 slots of one "column group" (tx, px) = 5 taps ((py0: ty 0,1,2), (py1: ty 1,2)) x 32 x 32 x (hi, lo) = 20 KB;
//            a step streams the 5 groups (tx0,px0) (tx1,px0) (tx1,px1) (tx2,px0) (tx2,px1) = 100 KB, group q = 5 k + g sits
//            in slot q % 3;
//   waves  : consumer (rw, px) owns cell rows 4 rw .. 4 rw + 3 and the output columns of x-phase px; the px = 0 waves
//            multiply groups 0, 1, 3 (15 taps), the px = 1 waves groups 2, 4 (10 taps); waves w and w + 4 (one of each)
//            share a SIMD, so every SIMD carries the same 600 MFMAs per step and its two waves are never in the same phase;
//   accumulators: [4 cell rows][2 y-phases][2 channel tiles] x f32x4 = 64 registers; a group is two passes of the 3x3
//            kernel's column schedule (3 taps over 6 window rows, then 2 taps over 5), 0.35 LDS fragment reads per MFMA;
//   epilogue: + bias + att-half partial sums (`res`, SP) + edge vectors -> SP store (+ second output x + temb), or the
//            fused `output` projection on the matrix pipe (stage 2).
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "conv_epilogue.h"
#include "mfma_policy.h"
#include "sp_sync.h"

namespace {

#ifdef DRS_SP_TIMELINE
__device__ unsigned long long drs_uf_tl[64];
#define UF_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl[i] += t_ - tl_last; tl_last = t_; } while (0)
#else
#define UF_STAMP(i) do { } while (0)
#endif

struct UfGeom {
  static constexpr int IW = 18, NPIX = 18 * 18;
  static constexpr int NBLK = (NPIX + 7) / 8;  // window pieces of 8 pixels = 1 KB
  static constexpr int WBUF = NBLK * 1024;
  static constexpr int TAPS = 5, NGRP = 5, NSLOT = 3;
  static constexpr int W_IMAGE = TAPS * 4 * 32 * 16;  // one operand image (hi or lo) of a group: [tap][k-group][32 channels] slots
  static constexpr int SLOT = 2 * W_IMAGE;            // 20 KB
  static constexpr int CONST = 2048;                  // per-launch epilogue constants: bias[Ch <= 256] | fuse_w[4][32] | fuse_b[4]
  static constexpr int LDS = 2 * WBUF + NSLOT * SLOT + 64 + CONST;
};

// column groups (tx, px) and their taps (py, ty), in streaming order
__host__ __device__ constexpr int uf_group_tx(int g) { return g == 0 ? 0 : (g <= 2 ? 1 : 2); }
__host__ __device__ constexpr int uf_group_px(int g) { return (g == 2 || g == 4) ? 1 : 0; }
__host__ __device__ constexpr int uf_tap_py(int j) { return j >= 3 ? 1 : 0; }
__host__ __device__ constexpr int uf_tap_ty(int j) { return j >= 3 ? j - 2 : j; }
// pair (kv, kw) of (3x3 convolution tap, transposed-convolution tap) contributes to tap t of output phase p
__host__ __device__ constexpr bool uf_pair(int p, int t, int kv, int kw) { return p + kv - kw == 2 * (t - 1); }
// SP output-row permutation (drs_sp_cout_perm of the pack kernels): MFMA row nn of a 32-channel group carries logical channel
__host__ __device__ constexpr int uf_perm(int nn) { return ((nn & 15) >> 2) * 8 + (nn >> 4) * 4 + (nn & 3); }

template <bool FUSE>
__global__ __launch_bounds__(768, 1) void upfuse_sp_kernel(UpFuseDesc d, int tiles_y, int tiles_x, int nck, int debug) {
  constexpr int NT = 2;  // channel tiles a consumer multiplies
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  using G = UfGeom;
  constexpr int IW = G::IW, NBLK = G::NBLK, WBUF = G::WBUF, W_IMAGE = G::W_IMAGE, SLOT = G::SLOT;
  constexpr int TH = 16, TW = 16, RPW = 4;
  char* sWin = smem;           // [buffer 2][window pixel][rotated operand slot 8] x 16 bytes
  char* sW = smem + 2 * WBUF;  // [ring slot 3][image 2][tap 5][k-group 4][32] operand slots
  sp_flag_ptr sCR = (sp_flag_ptr)(sW + G::NSLOT * SLOT);  // CR[3]: consumer waves that hold the group in ring slot s in registers
  sp_flag_ptr sCL = sCR + 3;                              // CL[3]: mover waves whose part of the group in ring slot s has landed
  sp_flag_ptr sWL = sCR + 6;                              // WL[2]: mover waves whose part of window buffer b has landed
  sp_flag_ptr sWR = sCR + 8;                              // WR[2]: consumer waves that have finished reading window buffer b

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7 consumers, 8..11 movers
  const bool mover = wid >= 8;
  const int lr = lane & 15, kg = lane >> 4;

  // persistent blocks, XCD-aware item order (see tapconv_mfma_kernel)
  const int ngroups = d.Ch >> 5;
  const int nitems = d.N * tiles_y * tiles_x * ngroups;
  const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
  const int per = (nitems + 7) >> 3;
  const int lo_item = xcd * per, hi_item = min(nitems, lo_item + per);
  const int span = hi_item - lo_item - j8;
  const int my_items = span > 0 ? (span + nb8 - 1) / nb8 : 0;
  const int S = my_items * nck;
  if (S == 0) return;
  auto item_of = [&](int ordinal, int& n_, int& ty0_, int& tx0_, int& n0_) __attribute__((always_inline)) {
    int it = lo_item + ordinal * nb8 + j8;
    n0_ = (it % ngroups) * 32;
    it /= ngroups;
    tx0_ = (it % tiles_x) * TW;
    it /= tiles_x;
    ty0_ = (it % tiles_y) * TH;
    n_ = it / tiles_y;
  };
  if (tid < 10) __hip_atomic_store(sCR + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  // epilogue constants that do not depend on the item: staged once (a global load at the start of every item epilogue is a
  // memory round trip with the matrix pipe of its SIMD idle)
  float* sConst = reinterpret_cast<float*>(sW + G::NSLOT * SLOT + 64);
  for (int i = tid; i < d.Ch; i += 768) sConst[i] = d.bias[i];
  if constexpr (FUSE) {
    if (tid < 128) sConst[256 + tid] = d.fuse_w[(size_t)min(tid >> 5, d.fuse_dim - 1) * d.Ch + (tid & 31)];
    if (tid < 4) sConst[384 + tid] = d.fuse_b[min(tid, d.fuse_dim - 1)];
  }
  sp_wait_lds();
  sp_barrier();
  int c = -1, ord = -1, n = 0, ty0 = 0, tx0 = 0, n0 = 0;  // current step: chunk, item ordinal, item coordinates (cells)
#ifdef DRS_SP_TIMELINE
  unsigned long long tl[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_amdgcn_s_memtime();
  const unsigned long long tl_begin = tl_last;
#endif

  if (mover) {
    // ===================== movers =====================
    // global -> registers -> LDS (conv_sp_movers.inc has the measurements behind this form).  Step k uses window buffer
    // k & 1 and the ring slots q % 3 of its groups q = 5 k + g.  A mover iteration:
    //   loads of groups 0, 1 | per group g: wait CR[slot] (the four consumers of the group that used the slot before hold
    //   it in registers) -> counted vmcnt -> store -> CL[slot]; the loads of group g + 2 follow into the freed registers
    //   between groups 3 and 4: the WINDOW OF STEP k + 1 (loads after group 3's, store once WR[(k+1) & 1] says the consumers
    //   have left that buffer in step k - 1): it lands a whole step before its first reader polls WL.
    //   consumer (rw, px) of step k: wait WL[k & 1]; for its groups: wait CL[slot] -> read 3 taps -> MFMA pass py0 (reads the
    //   other 2 taps under its tail) -> CR[slot] -> MFMA pass py1;  after the last group WR[k & 1] -> epilogue of the item
    // A slot's CR counts 4 releases per group (the four waves of the group's x-phase), its CL 4 landings (the four movers).
    // Every wait is for an event whose own prerequisites lie strictly earlier in this order: no cycle.  Releases are plain
    // LDS adds issued behind the reads / writes they publish (in-order LDS queue: sp_sync.h), never a queue drain.
    const int pw = wid - 8;
    __builtin_amdgcn_s_setprio(3);
    const char* zero = reinterpret_cast<const char*>(d.zero_line) + (lane & 15) * 16;
    // this lane's role inside a window piece (8 pixels x 128 bytes): lanes 8*px .. 8*px + 7 fetch the 8 operand slots of
    // pixel px = ONE 128-byte line, rotated by px: LDS slot s of the pixel holds operand slot (s - px) & 7
    const int l_px = lane >> 3, l_c = ((lane & 7) - l_px) & 7, l_img = l_c >> 2, l_kg = l_c & 3;
    const int l_off1 = l_img * 64 + l_kg * 16;
    constexpr int NPW = (NBLK + 3) / 4;  // window pieces per mover wave (at most): blocks pw + 4*i
    static_assert(NBLK == 4 * (NPW - 1) + 1, "piece distribution: the last round holds block NBLK - 1 only, owned by mover 0");
    int off1[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int p = (pw + 4 * i) * 8 + l_px;
      const int py = (p * 3641) >> 16, px = p - py * IW;  // p / 18 (exact for p < 1024)
      off1[i] = ((py * d.LW + px) * d.in_cs) * 4 + l_off1;
    }
    const bool tail_ok = (NBLK - 1) * 8 + l_px < G::NPIX;  // the last block is half empty
    constexpr int WPC = 5;                                 // weight pieces (1 KB) per mover wave and group
    u32x4 wr[2][WPC], ww[NPW];
    const int nwin = pw == 0 ? NPW : NPW - 1;
    const char* wg = reinterpret_cast<const char*>(d.w);
    const unsigned my_piece = (unsigned)(pw * WPC) * 1024u + (unsigned)lane * 16u;
    // window of step (n_, ty_, tx_, chunk c_) -> registers
    auto win_load = [&](int n_, int ty_, int tx_, int c_) __attribute__((always_inline)) {
      if (ty_ >= 1 && ty_ + TH + 1 <= d.LH && tx_ >= 1 && tx_ + TW + 1 <= d.LW) {
        // fast path: every window pixel inside the image
        const char* base = reinterpret_cast<const char*>(d.in) +
                           ((((long long)n_ * d.LH + (ty_ - 1)) * d.LW + (tx_ - 1)) * d.in_cs + d.in_co) * 4 + c_ * 128;
#pragma unroll
        for (int i = 0; i < NPW - 1; ++i) ww[i] = *reinterpret_cast<const u32x4*>(base + (unsigned)off1[i]);
        if (pw == 0) ww[NPW - 1] = *reinterpret_cast<const u32x4*>(tail_ok ? base + (unsigned)off1[NPW - 1] : zero);
      } else {
        // border tiles: the window offsets of the fast path hold for every pixel inside the image, the others read the zero
        // line.  32-bit validity arithmetic per piece on an OPAQUE copy of the lane coordinate (conv_sp_movers.inc, round 4:
        // the 64-bit per-lane address chain cost ~400 cycles per piece, and everything derived from the lane id is
        // loop-invariant - hoisted out of the step loop it occupied two dozen registers).
        const char* base = reinterpret_cast<const char*>(d.in) +
                           ((((long long)n_ * d.LH + (ty_ - 1)) * d.LW + (tx_ - 1)) * d.in_cs + d.in_co) * 4 + c_ * 128;
        int lpx = l_px;
        asm volatile("" : "+v"(lpx));
#pragma unroll
        for (int i = 0; i < NPW; ++i)
          if (i < nwin) {
            const int p = (pw + 4 * i) * 8 + lpx;
            const int py = (p * 3641) >> 16, px = p - py * IW;
            const int iy = ty_ - 1 + py, ix = tx_ - 1 + px;
            const bool ok = p < G::NPIX && (unsigned)iy < (unsigned)d.LH && (unsigned)ix < (unsigned)d.LW;
            ww[i] = *reinterpret_cast<const u32x4*>(ok ? base + (unsigned)off1[i] : zero);
          }
      }
    };
    // registers -> window buffer kk & 1, once the consumers have left it (step kk - 2); `after` as in store_grp
    auto win_store = [&](int kk, int after) __attribute__((always_inline)) {
      UF_STAMP(0);
      if (kk >= 2) sp_poll(sWR + (kk & 1), 8u * (unsigned)(kk >> 1), d.fault);
      UF_STAMP(4);
      sp_wait_vm(after);
      UF_STAMP(5);
      char* buf = sWin + (kk & 1) * WBUF + lane * 16;
#pragma unroll
      for (int i = 0; i < NPW; ++i)
        if (i < nwin) *reinterpret_cast<u32x4*>(buf + (pw + 4 * i) * 1024) = ww[i];
      sp_release(sWL + (kk & 1), lane);
      UF_STAMP(6);
    };
    {  // the first window; every later one is fetched and stored a whole step ahead, in the shadow of the previous step
      int n_, ty_, tx_, n0_;
      item_of(0, n_, ty_, tx_, n0_);
      win_load(n_, ty_, tx_, 0);
      win_store(0, 0);
    }
    for (int k = 0; k < S; ++k) {
      if (++c == nck) c = 0;
      if (c == 0) item_of(++ord, n, ty0, tx0, n0);
      const char* gsrc = wg + ((size_t)(n0 >> 5) * nck + c) * (size_t)(G::NGRP * SLOT) + my_piece;
      auto load_grp = [&](int g, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WPC; ++i) wr[set][i] = *reinterpret_cast<const u32x4*>(gsrc + (unsigned)(g * SLOT + i * 1024));
      };
      // store group g (registers `set`) once its ring slot is free; `after` = vector-memory operations issued after its loads
      auto store_grp = [&](int g, int set, int after) __attribute__((always_inline)) {
        const unsigned q = 5u * (unsigned)k + (unsigned)g, fill = q / 3u, slot = q - 3u * fill;
        UF_STAMP(0);
        if (fill > 0) sp_poll(sCR + slot, 4u * fill, d.fault);
        UF_STAMP(1);
        sp_wait_vm(after);
        UF_STAMP(2);
        char* dst = sW + slot * SLOT + my_piece;
#pragma unroll
        for (int i = 0; i < WPC; ++i) *reinterpret_cast<u32x4*>(dst + i * 1024) = wr[set][i];
        sp_release(sCL + slot, lane);
        UF_STAMP(3);
      };
      const bool more = k + 1 < S;
      const int nw2 = more ? nwin : 0;  // window loads of step k + 1, issued between the loads of groups 3 and 4
      load_grp(0, 0);
      load_grp(1, 1);
      store_grp(0, 0, WPC);
      load_grp(2, 0);
      store_grp(1, 1, WPC);
      load_grp(3, 1);
      if (more) {
        int c2 = c + 1, n2 = n, ty2 = ty0, tx2 = tx0, n02 = n0;
        if (c2 == nck) { c2 = 0; item_of(ord + 1, n2, ty2, tx2, n02); }
        win_load(n2, ty2, tx2, c2);
      }
      store_grp(2, 0, WPC + nw2);
      load_grp(4, 0);
      store_grp(3, 1, nw2 + WPC);
      if (more) win_store(k + 1, WPC);
      store_grp(4, 0, 0);
    }
  } else {
    // ===================== consumers =====================
    const int rw = wid & 3;   // cell rows [4 rw, 4 rw + 4) of the tile
    const int px_wave = wid >> 2;  // x-phase of this wave's output columns
    // fragment addresses: window pixel p = B + q + lr with B = rw*RPW*18 (per wave) and q = wr*18 + tx (compile time);
    // slot address = p * 128 + ((c + p) & 7) * 16, c = image * 4 + k-group (conv_mfma_sp.hip)
    const int B = rw * RPW * IW;
    // One lane table per residue q & 7; the lo image sits 4 slots further in the rotation, and (x + 4) & 7 == x ^ 4, so the
    // lo address of residue j IS the hi address of residue j ^ 4: 8 registers serve both images.  The tables are rebuilt
    // at the top of every step from opaque copies of the lane coordinates: their live range then ends before the item
    // epilogue, which would otherwise push them into scratch (reloads inside the MFMA loop).
    int tab[8];
    const char* wlane = sW + (kg * 32 + lr) * 16;  // this lane's origin inside a ring slot
    f32x4 acc[RPW][2][NT];                          // [cell row][y-phase][channel tile]
    typename P::Frag wf[3][NT];
    auto win_frag = [&](const char* buf, int q) __attribute__((always_inline)) {  // q: compile-time window offset
      return typename P::Frag{*reinterpret_cast<const bf16x8*>(buf + tab[q & 7] + q * 128),
                              *reinterpret_cast<const bf16x8*>(buf + tab[(q & 7) ^ 4] + q * 128)};
    };
    // one column group: taps (py0: ty 0,1,2 | py1: ty 1,2) of input column tx for this wave's x-phase
    auto do_group = [&](const char* buf, int tx, unsigned q) __attribute__((always_inline)) {
      const unsigned fill = q / 3u, slot = q - 3u * fill;
      UF_STAMP(0);
      sp_poll_lds(sCL + slot, 4u * (fill + 1u), d.fault);
      UF_STAMP(1);
      const char* sb = wlane + slot * SLOT;
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int t = 0; t < NT; ++t) wf[ty][t] = P::load(sb, W_IMAGE, (size_t)(ty * 2048 + t * 256));
      // pass py0: 3 taps over window rows 0..5.  wf[0] (ty 0) is last used by row 3, wf[1] by row 4: the taps of pass py1
      // (ty 1 -> wf[0], ty 2 -> wf[1]) are read under the tail of this pass.
#pragma unroll
      for (int wr = 0; wr < RPW + 2; ++wr) {
        const typename P::Frag af = win_frag(buf, wr * IW + tx);
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const int r = wr - ty;
          if (r >= 0 && r < RPW) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[r][0][t] = P::mma(wf[ty][t], af, acc[r][0][t]);
          }
        }
        if (wr == RPW - 1) {
#pragma unroll
          for (int t = 0; t < NT; ++t) wf[0][t] = P::load(sb, W_IMAGE, (size_t)(3 * 2048 + t * 256));
        }
        if (wr == RPW) {
#pragma unroll
          for (int t = 0; t < NT; ++t) wf[1][t] = P::load(sb, W_IMAGE, (size_t)(4 * 2048 + t * 256));
        }
      }
      UF_STAMP(2);
      sp_release(sCR + slot, lane);  // all five taps have been read: the ring slot may be refilled
      UF_STAMP(3);
      // pass py1: ty 1 (wf[0]) and ty 2 (wf[1]) over window rows 1..5
#pragma unroll
      for (int wr = 1; wr < RPW + 2; ++wr) {
        const typename P::Frag af = win_frag(buf, wr * IW + tx);
        if (wr - 1 < RPW) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[wr - 1][1][t] = P::mma(wf[0][t], af, acc[wr - 1][1][t]);
        }
        if (wr - 2 >= 0) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[wr - 2][1][t] = P::mma(wf[1][t], af, acc[wr - 2][1][t]);
        }
      }
    };
    // the whole step loop once per x-phase (compile-time px: straight-line group sequences, no accumulator copies
    // between the two roles' code paths)
    auto run = [&](auto PXC) __attribute__((always_inline)) {
    constexpr int px = decltype(PXC)::value;
    for (int k = 0; k < S; ++k) {
      if (++c == nck) c = 0;
      if (c == 0) {
        item_of(++ord, n, ty0, tx0, n0);
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
          for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[r][py][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      {
        int lr_s = lr, kg_s = kg;
        asm volatile("" : "+v"(lr_s), "+v"(kg_s));
#pragma unroll
        for (int j = 0; j < 8; ++j) tab[j] = (B + lr_s) * 128 + ((kg_s + B + j + lr_s) & 7) * 16;
      }
      const char* buf = sWin + (k & 1) * WBUF;
      UF_STAMP(7);
      sp_poll_lds(sWL + (k & 1), 4u * (unsigned)((k >> 1) + 1), d.fault);  // window k is in its buffer
      UF_STAMP(4);
      const unsigned q0 = 5u * (unsigned)k;
      if constexpr (px == 0) {
        do_group(buf, 0, q0);
        do_group(buf, 1, q0 + 1u);
        do_group(buf, 2, q0 + 3u);
      } else {
        do_group(buf, 1, q0 + 2u);
        do_group(buf, 2, q0 + 4u);
      }
      UF_STAMP(5);
      sp_release(sWR + (k & 1), lane);  // the last window fragment has been read: the buffer may be refilled (for step k + 2)
      UF_STAMP(6);
      if (c == nck - 1) {
        // ---------------- item epilogue ----------------
        // lane (lr, kg) holds, per (cell row r, y-phase py), the 8 consecutive logical channels n0 + kg*8 .. +7 (tile 0:
        // +0..3, tile 1: +4..7, SP output-row permutation) of output pixel (2 (ty0 + 4 rw + r) + py, 2 (tx0 + lr) + px)
        int lr_e = lr, kg_e = kg;
        asm volatile("" : "+v"(lr_e), "+v"(kg_e));
        const int OH = 2 * d.LH, OW = 2 * d.LW;
        const int c8 = n0 + kg_e * 8;
        const bool lo = lr_e < 8;
        const int pl = lr_e & 7;
        const int mx = tx0 + lr_e;
        const bool own_ok = mx < d.LW;
        const int ox_own = 2 * min(mx, d.LW - 1) + px;
        const bool ok0 = tx0 + pl < d.LW, ok1 = tx0 + pl + 8 < d.LW;
        const int myb = ty0 + rw * RPW;
        auto load8 = [&](const float* p, float (&v)[8]) __attribute__((always_inline)) {
          const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
          v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        };
        float bias8[8];
        load8(sConst + c8, bias8);
        // att-half partial sums of this lane's pixels, added into the accumulators BEFORE the first store of the item (one
        // in-order counter for loads and stores: a load issued behind a store is complete only once that store is).  All
        // sixteen loads are in flight together (the fragment registers of the step loop are free by now): left to itself the
        // compiler issues them pair by pair with a full wait after each, eight memory round trips per item.
#ifdef DRS_SP_TIMELINE
        if (d.res && !(debug & 2)) {
#else
        if (d.res) {
#endif
          u32x4 rh[RPW][2], rl[RPW][2];
#pragma unroll
          for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              const int oy = 2 * min(myb + r, d.LH - 1) + py;
              const char* g = reinterpret_cast<const char*>(d.res) +
                              ((((size_t)n * OH + oy) * OW + ox_own) * d.res_cs + d.res_co + n0) * 4 + kg_e * 16;
              rh[r][py] = *reinterpret_cast<const u32x4*>(g);
              rl[r][py] = *reinterpret_cast<const u32x4*>(g + 64);
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              float rv[8];
              drs_sp_join8(rh[r][py], rl[r][py], rv);
#pragma unroll
              for (int j = 0; j < 4; ++j) { acc[r][py][0][j] += rv[j]; acc[r][py][1][j] += rv[4 + j]; }
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        // edge vectors of border tiles, also into the accumulators before the first store (a load behind a store would wait
        // for that store to reach memory: eight drains per item in the waves that own an image column).  First / last
        // output row: wave-uniform, at most two row-phases of a wave.  First / last output column: one lane of a border
        // tile's waves, all eight row-phases: fetched four at a time.
        if (d.eh) {
#pragma unroll
          for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              const int oy = 2 * (myb + r) + py;
              if (myb + r < d.LH && (oy == 0 || oy == OH - 1)) {
                float e8[8];
                load8(d.eh + (((size_t)n * 2 + (oy ? 1 : 0)) * OW + ox_own) * d.Ch + c8, e8);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[r][py][0][j] += e8[j]; acc[r][py][1][j] += e8[4 + j]; }
              }
            }
          if (own_ok && (ox_own == 0 || ox_own == OW - 1)) {
            const float* evp = d.ev + ((size_t)n * 2 + (ox_own ? 1 : 0)) * OH * d.Ch + c8;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              float e8[4][8];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int r = half * 2 + (q >> 1), py = q & 1;
                const int oy = min(2 * (myb + r) + py, OH - 1);
                load8(evp + (size_t)oy * d.Ch, e8[q]);
              }
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int r = half * 2 + (q >> 1), py = q & 1;
                const int oy = 2 * (myb + r) + py;
                // rows 0 / OH-1 took the row vectors, and ev HAS NO VALUE there (the edge kernel does not write those rows: whatever
                // the workspace held): selected, not multiplied by 0 - 0 x NaN is NaN (found in round 5, once a previous test's
                // unbounded chain had left NaNs in the allocator's memory; finite garbage x 0 had hidden it since round 3)
                const bool k = myb + r < d.LH && oy > 0 && oy < OH - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[r][py][0][j] += k ? e8[q][j] : 0.f; acc[r][py][1][j] += k ? e8[q][4 + j] : 0.f; }
              }
            }
          }
        }
        float post2_8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) post2_8[j] = 0.f;
        typename P::Frag wfr;
        float fb[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (FUSE) {
          // A operand of the projection: row 4 * j = fuse_w[j][this lane's 8 channels], every other row zero: output j of
          // pixel lr lands in register 0 of the lane of k-group j, and all fuse_dim planes go out in ONE store instruction
          // (and come in, for the projected att-half, in one load): a store costs its wave ~250 cycles, whatever it carries
          float w8[8];
          load8(sConst + 256 + min(lr_e >> 2, 3) * 32 + c8, w8);
          const float keep = ((lr_e & 3) == 0 && (lr_e >> 2) < d.fuse_dim) ? 1.f : 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) w8[j] *= keep;
          u32x4 h, l;
          drs_sp_split8(w8, h, l);
          wfr = typename P::Frag{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, l)};
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[j] = sConst[384 + j];
        } else {
          if (d.out2) load8(d.post2 + (size_t)n * d.post2_cs + c8, post2_8);
        }
        // Every load of the epilogue is consumed HERE, unconditionally: a value that is only used inside the `row is inside
        // the image` branches below leaves its load formally pending on the other path, and the compiler then drains the
        // whole vector-memory counter (the item's STORES included) in front of the next step's first fragment read.
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(bias8[j]), "v"(post2_8[j]));
        UF_STAMP(8);
        const int lane_b = (lo ? 0 : 64) + kg_e * 16;
        // fused projection on top of the projected att-half already in fuse_out: its values, loaded before the first store
        float prev[FUSE ? RPW : 1][2];  // (this lane's plane: k-group kg_e)
        float fbk = 0.f;
        if constexpr (FUSE) {
          const size_t plane = (size_t)OH * OW;
          fbk = kg_e == 0 ? fb[0] : kg_e == 1 ? fb[1] : kg_e == 2 ? fb[2] : fb[3];
#pragma unroll
          for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              prev[r][py] = 0.f;
              if (d.fuse_acc && own_ok && kg_e < d.fuse_dim && myb + r < d.LH)
                prev[r][py] = d.fuse_out[((size_t)n * d.fuse_dim + kg_e) * plane + (size_t)(2 * (myb + r) + py) * OW + ox_own];
            }
#pragma unroll
          for (int r = 0; r < RPW; ++r)  // (consumed unconditionally, like the constants above: no load may stay formally pending)
            asm volatile("" :: "v"(prev[r][0]), "v"(prev[r][1]));
          asm volatile("" :: "v"(fbk));
          asm volatile("" :: "v"(wfr.hi), "v"(wfr.lo));
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          const int my = myb + r;
          if (my < d.LH) {
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              const int oy = 2 * my + py;
              float v[8];
#pragma unroll
              for (int j = 0; j < 4; ++j) { v[j] = acc[r][py][0][j] + bias8[j]; v[4 + j] = acc[r][py][1][j] + bias8[4 + j]; }
              if constexpr (FUSE) {
                u32x4 h, l;
                drs_sp_split8(v, h, l);
                const typename P::Frag vf{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, l)};
                const f32x4 y = P::mma(wfr, vf, f32x4{0.f, 0.f, 0.f, 0.f});  // register 0 of the lane of k-group j: output j of pixel lr
#ifdef DRS_SP_TIMELINE
                if (debug & 1) { asm volatile("" :: "v"(y)); } else
#endif
                if (own_ok && kg_e < d.fuse_dim) {
                  const size_t plane = (size_t)OH * OW;
                  d.fuse_out[((size_t)n * d.fuse_dim + kg_e) * plane + (size_t)oy * OW + ox_own] = y[0] + fbk + prev[r][py];
                }
              } else {
                // SP stores in full 128-byte lines: lanes lr < 8 write hi slots, lanes lr >= 8 lo slots, of cells pl and pl + 8
                const size_t pix0 = ((size_t)n * OH + oy) * OW + 2 * (tx0 + pl) + px;
                auto put = [&](float* base, int cs, int co, const float (&w8)[8]) __attribute__((always_inline)) {
                  u32x4 H, L;
                  drs_sp_split8(w8, H, L);
                  const u32x4 got = drs_dpp_swap8(lo ? L : H);  // lr < 8 receives the partner's hi, lr >= 8 the partner's lo
                  char* g = reinterpret_cast<char*>(base) + (pix0 * cs + co + n0) * 4 + lane_b;
#ifdef DRS_SP_TIMELINE
                  if (debug & 1) { asm volatile("" :: "v"(H), "v"(L), "v"(got)); return; }
#endif
                  if (ok0) drs_store16(g, lo ? H : got);
                  if (ok1) drs_store16(g + (size_t)16 * cs * 4, lo ? got : L);
                };
                if (d.out) put(d.out, d.out_cs, d.out_co, v);
                if (d.out2) {
                  float p2[8];
#pragma unroll
                  for (int j = 0; j < 8; ++j) p2[j] = v[j] + post2_8[j];
                  put(d.out2, d.out2_cs, d.out2_co, p2);
                }
              }
            }
          }
        }
        UF_STAMP(9);
      }
    }
    };
    if (px_wave == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
  }
#ifdef DRS_SP_TIMELINE
  if (blockIdx.x == 0 && (wid == 0 || wid == 4 || wid == 8) && lane == 0) {
    const int o = wid == 0 ? 0 : (wid == 4 ? 16 : 32);
    for (int i = 0; i < 10; ++i) drs_uf_tl[o + i] = tl[i];
    drs_uf_tl[o + 10] = S;
    drs_uf_tl[o + 11] = __builtin_amdgcn_s_memtime() - tl_begin;
  }
#endif
}

// ---- pack: composite operand image ------------------------------------------------------------------------------------
// dst: [Ch/32][Cc/32][group 5][image 2][tap 5][k-group 4][32 rows][8 x bf16]; row nn of a 32-channel group carries logical
// channel uf_perm(nn).  One thread per 16-byte slot (8 input channels).
__global__ void upfuse_pack_kernel(const float* __restrict__ v_w, const float* __restrict__ t_w, int Cc, int Ch,
                                   char* __restrict__ dst) {
  const int nck = Cc >> 5;
  const size_t nslots = (size_t)(Ch >> 5) * nck * 5 * 5 * 4 * 32;
  const int cinv = Cc + Ch;
  for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += (size_t)gridDim.x * blockDim.x) {
    const int nn = (int)(s & 31);
    const int q = (int)((s >> 5) & 3);
    const int j = (int)((s >> 7) % 5);
    const int g = (int)((s / 640) % 5);
    const int ck = (int)((s / 3200) % nck);
    const int cgi = (int)(s / ((size_t)3200 * nck));
    const int co = cgi * 32 + uf_perm(nn);
    const int tx = uf_group_tx(g), px = uf_group_px(g), py = uf_tap_py(j), ty = uf_tap_ty(j);
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int ci0 = ck * 32 + q * 8;
    for (int kvy = 0; kvy < 3; ++kvy)
      for (int kwy = 0; kwy < 3; ++kwy) {
        if (!uf_pair(py, ty, kvy, kwy)) continue;
        for (int kvx = 0; kvx < 3; ++kvx)
          for (int kwx = 0; kwx < 3; ++kwx) {
            if (!uf_pair(px, tx, kvx, kwx)) continue;
            const float* vp = v_w + (size_t)co * cinv * 9 + kvy * 3 + kvx;
            const float* wp = t_w + (size_t)ci0 * Cc * 9 + kwy * 3 + kwx;
            for (int c = 0; c < Cc; ++c) {
              const float vv = vp[(size_t)c * 9];
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] += vv * wp[((size_t)e * Cc + c) * 9];
            }
          }
      }
    const size_t base = (((size_t)cgi * nck + ck) * 5 + g) * UfGeom::SLOT + (size_t)j * 2048 + q * 512 + nn * 16;
    PolicyBF16X3::cvt_store(dst, UfGeom::W_IMAGE, base, x);
  }
}

// aux (fp32): rt [5][Cc][Ch] (first-ROW paths: kvy = kwy = 0, pairs over x), rl [6][Cc][Ch] (first-COLUMN paths: kvx = kwx = 0,
// pairs over y; entry 5 = (py 0, ty 1) without the pair (0, 0): output row 0, whose row -1 paths are already in rt),
// bt [9][Ch] = sum_c V[co][c][kv] b_t[c], bias [Ch] = b_v + sum over the nine taps of bt.
__global__ void upfuse_aux_kernel(const float* __restrict__ v_w, const float* __restrict__ v_b, const float* __restrict__ t_w,
                                  const float* __restrict__ t_b, int Cc, int Ch, float* __restrict__ aux) {
  const int cinv = Cc + Ch;
  const size_t nr = (size_t)11 * Cc * Ch;
  float* rt = aux;
  float* bt = aux + nr;
  float* bias = bt + 9 * Ch;
  const size_t total = nr + (size_t)9 * Ch;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    if (i < nr) {
      const int co = (int)(i % Ch);
      const int ci = (int)((i / Ch) % Cc);
      const int jj = (int)(i / ((size_t)Ch * Cc));  // 0..4 rt, 5..10 rl
      const bool top = jj < 5;
      const int j = top ? jj : (jj - 5 == 5 ? 1 : jj - 5);
      const bool variant = jj == 10;
      const int p = uf_tap_py(j), t = uf_tap_ty(j);  // (phase, tap) along the free axis
      float acc = 0.f;
      for (int kv = 0; kv < 3; ++kv)
        for (int kw = 0; kw < 3; ++kw) {
          if (!uf_pair(p, t, kv, kw)) continue;
          if (variant && kv == 0 && kw == 0) continue;
          const int vt = top ? kv : kv * 3;  // top: (kvy 0, kvx kv); left: (kvy kv, kvx 0)
          const int wt = top ? kw : kw * 3;
          const float* vp = v_w + (size_t)co * cinv * 9 + vt;
          const float* wp = t_w + (size_t)ci * Cc * 9 + wt;
          for (int c = 0; c < Cc; ++c) acc += vp[(size_t)c * 9] * wp[(size_t)c * 9];
        }
      rt[i] = acc;
    } else {
      const size_t r = i - nr;
      const int co = (int)(r % Ch), kv = (int)(r / Ch);
      float acc = 0.f;
      for (int c = 0; c < Cc; ++c) acc += v_w[((size_t)co * cinv + c) * 9 + kv] * t_b[c];
      bt[r] = acc;
    }
  }
  (void)bias;
}
__global__ void upfuse_bias_kernel(const float* __restrict__ v_b, int Ch, const float* __restrict__ bt, float* __restrict__ bias) {
  const int co = blockIdx.x * blockDim.x + threadIdx.x;
  if (co >= Ch) return;
  float b = v_b ? v_b[co] : 0.f;
  for (int kv = 0; kv < 9; ++kv) b += bt[kv * Ch + co];
  bias[co] = b;
}

// ---- edge vectors -----------------------------------------------------------------------------------------------------
// eh[n][0 | 1][ox][co]: everything the composite needs added in output rows 0 and OH-1 (all columns, corners included);
// ev[n][0 | 1][oy][co]: the same for output columns 0 and OW-1, rows 1 .. OH-2 (zero in rows 0 and OH-1).
//   bias part: minus bt[kvy][kvx] for every tap whose transposed-convolution position lies outside the image;
//   data part: minus the composite's paths through row -1 (rt, output row 0) and column -1 (rl, output column 0).
__device__ __forceinline__ float uf_sp_value(const char* base, int ci) {  // element ci of a pixel's SP channel vector
  const char* g = base + (ci >> 5) * 128 + (ci & 31) * 2;
  return (float)*reinterpret_cast<const __bf16*>(g) + (float)*reinterpret_cast<const __bf16*>(g + 64);
}
// MFMA form.  One wave = (kind: 0 = first row | 1 = first column, image n, segment of 16 cells = 32 output positions along
// the edge, group of 32 output channels): a 1-D composite convolution with the weights as A operand (edge operand image:
// [kind][chunk][tap 7][k-group][Ch][8 x bf16], hi image then lo image; taps 0..4 = (p0: t 0,1,2), (p1: t 1,2); row blocks:
// tap 5 = rl0, tap 6 = rl[(0,2)] for the corner's column -1 paths) and SP slots of h straight from global memory as B
// operand (cells m-1, m, m+1 of the edge; lanes outside the image read zeros).  Operands of chunk c + 1 are in flight
// while chunk c is multiplied.  256 waves per stage, one per CU.
struct UfEdgeFrags { PolicyBF16X3::Frag w[7][2], b[3], cb[2]; };
__global__ __launch_bounds__(64) void upfuse_edges_mfma_kernel(UpFuseEdgeDesc d, const char* __restrict__ wimg, size_t img_bytes,
                                                               const char* __restrict__ zero16) {
  using P = PolicyBF16X3;
  const int OH = 2 * d.LH, OW = 2 * d.LW, Ch = d.Ch;
  const int nck = d.Cc >> 5;
  const int lane = threadIdx.x, lr = lane & 15, kg = lane >> 4;
  const int cgs = Ch >> 5;
  const int cg = blockIdx.x % cgs, seg = blockIdx.x / cgs, kind = blockIdx.y, n = blockIdx.z;
  const int L = kind ? d.LH : d.LW;
  const int m0 = seg * 16;
  if (m0 >= L) return;
  const bool corner = kind == 0 && seg == 0;
  const int NTAP = corner ? 7 : 5;
  auto pixel = [&](int y, int x) {
    return reinterpret_cast<const char*>(d.in) + ((((size_t)n * d.LH + y) * d.LW + x) * d.in_cs + d.in_co) * 4 + kg * 16;
  };
  const char* bsrc[3];
  bool bok[3];
#pragma unroll
  for (int sft = 0; sft < 3; ++sft) {
    const int m = m0 + lr + sft - 1;
    bok[sft] = m >= 0 && m < L;
    bsrc[sft] = bok[sft] ? pixel(kind ? m : 0, kind ? 0 : m) : zero16;
  }
  // corner operands: pixel (0, 0) and pixel (1, 0), lane lr == 0 only
  const bool c0ok = corner && lr == 0, c1ok = corner && lr == 0 && d.LH > 1;
  const char* csrc0 = c0ok ? pixel(0, 0) : zero16;
  const char* csrc1 = c1ok ? pixel(1, 0) : zero16;
  // weights: lane (lr, kg) -> row cg*32 + t*16 + lr of k-group kg
  const char* wbase = wimg + ((size_t)kind * nck * 7 * 4 * Ch + (size_t)kg * Ch + cg * 32 + lr) * 16;
  auto issue = [&](int c, UfEdgeFrags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int tap = 0; tap < 7; ++tap)
      if (tap < NTAP)
#pragma unroll
        for (int t = 0; t < 2; ++t) f.w[tap][t] = P::load(wbase, img_bytes, (((size_t)c * 7 + tap) * 4 * Ch + t * 16) * 16);
#pragma unroll
    for (int sft = 0; sft < 3; ++sft) {
      const char* p = bsrc[sft] + (bok[sft] ? c * 128 : 0);
      f.b[sft] = typename P::Frag{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(p + (bok[sft] ? 64 : 0))};
    }
    if (corner) {
      const char* p0 = csrc0 + (c0ok ? c * 128 : 0);
      const char* p1 = csrc1 + (c1ok ? c * 128 : 0);
      f.cb[0] = typename P::Frag{*reinterpret_cast<const bf16x8*>(p0), *reinterpret_cast<const bf16x8*>(p0 + (c0ok ? 64 : 0))};
      f.cb[1] = typename P::Frag{*reinterpret_cast<const bf16x8*>(p1), *reinterpret_cast<const bf16x8*>(p1 + (c1ok ? 64 : 0))};
    }
  };
  f32x4 acc[2][2], accc[2];  // [phase][tile]; corner term (even phase, position 0)
#pragma unroll
  for (int t = 0; t < 2; ++t) acc[0][t] = acc[1][t] = accc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mult = [&](const UfEdgeFrags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      acc[0][t] = P::mma(f.w[0][t], f.b[0], acc[0][t]);  // even position 2m: cells m-1, m, m+1
      acc[0][t] = P::mma(f.w[1][t], f.b[1], acc[0][t]);
      acc[0][t] = P::mma(f.w[2][t], f.b[2], acc[0][t]);
      acc[1][t] = P::mma(f.w[3][t], f.b[1], acc[1][t]);  // odd position 2m+1: cells m, m+1
      acc[1][t] = P::mma(f.w[4][t], f.b[2], acc[1][t]);
      if (corner) {
        accc[t] = P::mma(f.w[5][t], f.cb[0], accc[t]);
        accc[t] = P::mma(f.w[6][t], f.cb[1], accc[t]);
      }
    }
  };
  UfEdgeFrags fa, fb;
  issue(0, fa);
  for (int c = 0; c < nck; c += 2) {
    if (c + 1 < nck) issue(c + 1, fb);
    mult(fa);
    if (c + 2 < nck) issue(c + 2, fa);
    if (c + 1 < nck) mult(fb);
  }
  auto biasdelta4 = [&](int oy, int ox, int c0, float (&v)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = 0.f;
    for (int kvy = 0; kvy < 3; ++kvy)
      for (int kvx = 0; kvx < 3; ++kvx)
        if ((oy == 0 && kvy == 0) || (oy == OH - 1 && kvy == 2) || (ox == 0 && kvx == 0) || (ox == OW - 1 && kvx == 2)) {
          const float4 b = *reinterpret_cast<const float4*>(d.bt + (kvy * 3 + kvx) * Ch + c0);
          v[0] -= b.x; v[1] -= b.y; v[2] -= b.z; v[3] -= b.w;
        }
  };
  const int m = m0 + lr;
  if (m >= L) return;
#pragma unroll
  for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int pos = 2 * m + ph;
      const int c0 = cg * 32 + t * 16 + kg * 4;
      float bd[4], o[4];
      if (kind == 0) {
        const int ox = pos;
        biasdelta4(0, ox, c0, bd);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = bd[j] - acc[ph][t][j] - ((ph == 0 && corner) ? accc[t][j] : 0.f);  // (accc is zero but in lane 0)
        *reinterpret_cast<float4*>(d.eh + (((size_t)n * 2 + 0) * OW + ox) * Ch + c0) = make_float4(o[0], o[1], o[2], o[3]);
        if (ox != 0) {
          biasdelta4(OH - 1, ox, c0, bd);
          *reinterpret_cast<float4*>(d.eh + (((size_t)n * 2 + 1) * OW + ox) * Ch + c0) = make_float4(bd[0], bd[1], bd[2], bd[3]);
        }
      } else {
        const int oy = pos;
        if (oy == 0) continue;  // the row wave owns output row 0
        biasdelta4(oy, 0, c0, bd);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = bd[j] - acc[ph][t][j];
        if (oy == OH - 1) {
          *reinterpret_cast<float4*>(d.eh + (((size_t)n * 2 + 1) * OW + 0) * Ch + c0) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
          *reinterpret_cast<float4*>(d.ev + (((size_t)n * 2 + 0) * OH + oy) * Ch + c0) = make_float4(o[0], o[1], o[2], o[3]);
          biasdelta4(oy, OW - 1, c0, bd);
          *reinterpret_cast<float4*>(d.ev + (((size_t)n * 2 + 1) * OH + oy) * Ch + c0) = make_float4(bd[0], bd[1], bd[2], bd[3]);
        }
      }
    }
}

// fp32 edge weights (aux: rt | rl) -> the MFMA operand image of upfuse_edges_mfma_kernel
__global__ void upfuse_edge_pack_kernel(const float* __restrict__ aux, int Cc, int Ch, char* __restrict__ dst, size_t img_bytes) {
  const int nck = Cc >> 5;
  const size_t mat = (size_t)Cc * Ch;
  const size_t nslots = (size_t)2 * nck * 7 * 4 * Ch;
  for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(s % Ch);
    const int q = (int)((s / Ch) & 3);
    const int tap = (int)((s / ((size_t)Ch * 4)) % 7);
    const int ck = (int)((s / ((size_t)Ch * 28)) % nck);
    const int kind = (int)(s / ((size_t)Ch * 28 * nck));
    const float* src = nullptr;  // [ci][co] matrix
    if (tap < 5) src = aux + (size_t)(kind ? 5 + tap : tap) * mat;
    else if (kind == 0) src = aux + (size_t)(5 + (tap == 5 ? 5 : 2)) * mat;  // rl0, rl[(0, 2)]
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = src ? src[(size_t)(ck * 32 + q * 8 + e) * Ch + co] : 0.f;
    PolicyBF16X3::cvt_store(dst, img_bytes, s * 16, x);
  }
}

// `output` folded into up_convs.i (x-half): a 32-channel layer whose logical channel 8 j is output j (MFMA row 4 j of the first
// channel tile under uf_perm), everything else zero; the pack kernels above then build its composite image, edge weights and bias
__global__ __launch_bounds__(256) void upfuse_fold_proj_kernel(const float* __restrict__ v_w, const float* __restrict__ v_b,
                                                               const float* __restrict__ fw, const float* __restrict__ fb,
                                                               int fuse_dim, int Cc, int Ch, float* __restrict__ dst_w,
                                                               float* __restrict__ dst_b) {
  const int cinv = Cc + Ch;
  const int total = 32 * cinv * 9;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int row = i / (cinv * 9), rem = i - row * cinv * 9, c = rem / 9, kv = rem - c * 9;
    const int j = row >> 3;
    float a = 0.f;
    if ((row & 7) == 0 && j < fuse_dim && c < Cc)
      for (int co = 0; co < Ch; ++co) a += fw[(size_t)j * Ch + co] * v_w[((size_t)co * cinv + c) * 9 + kv];
    dst_w[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < 32) {
    const int row = threadIdx.x, j = row >> 3;
    float a = 0.f;
    if ((row & 7) == 0 && j < fuse_dim) {
      a = fb ? fb[j] : 0.f;
      if (v_b)
        for (int co = 0; co < Ch; ++co) a += fw[(size_t)j * Ch + co] * v_b[co];
    }
    dst_b[row] = a;
  }
}

// NCHW fp32 -> SP channels-last (operator-level entry / tests only)
__global__ void nchw_to_sp_kernel(const float* __restrict__ src, char* __restrict__ dst, int N, int C, int H, int W) {
  const int64_t hw = (int64_t)H * W, slots = (int64_t)N * hw * (C / 8);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (int64_t)gridDim.x * blockDim.x) {
    const int sl = (int)(i % (C / 8));
    const int64_t pix = i / (C / 8);
    const int n = (int)(pix / hw);
    const int64_t r = pix % hw;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = src[((int64_t)n * C + sl * 8 + j) * hw + r];
    PolicyBF16X3::cvt_store(dst, 64, (size_t)((pix * C + (sl >> 2) * 32) * 4 + (sl & 3) * 16), x);
  }
}

}  // namespace

bool drs_upfuse_supported(int Cc, int Ch, int LH, int LW) {
  static const int env = getenv("DRS_UPFUSE") ? atoi(getenv("DRS_UPFUSE")) : 1;
  return env && Cc % 32 == 0 && Ch % 32 == 0 && Cc >= 32 && Ch >= 32 && Ch <= 256 && LH > 8 && LW > 8;
}
size_t drs_upfuse_weight_bytes(int Cc, int Ch) { return (size_t)(Ch / 32) * (Cc / 32) * UfGeom::NGRP * UfGeom::SLOT; }
size_t drs_upfuse_aux_floats(int Cc, int Ch) { return (size_t)11 * Cc * Ch + (size_t)10 * Ch; }

int drs_launch_upfuse_pack(const float* v_w, const float* v_b, const float* t_w, const float* t_b, int Cc, int Ch, void* dst_w,
                           float* dst_aux, void* dst_edge, hipStream_t s) {
  DRS_REQUIRE(Cc % 32 == 0 && Ch % 32 == 0, DRS_ERR_SHAPE, "upfuse_pack: Cc=%d Ch=%d", Cc, Ch);
  const size_t nslots = drs_upfuse_weight_bytes(Cc, Ch) / 32;
  int blocks = (int)((nslots + 127) / 128);
  DRS_LAUNCH(upfuse_pack_kernel, dim3(blocks), dim3(128), 0, s, v_w, t_w, Cc, Ch, (char*)dst_w);
  DRS_CHECK_HIP(hipGetLastError());
  const size_t total = (size_t)11 * Cc * Ch + (size_t)9 * Ch;
  blocks = (int)((total + 255) / 256);
  DRS_LAUNCH(upfuse_aux_kernel, dim3(blocks), dim3(256), 0, s, v_w, v_b, t_w, t_b, Cc, Ch, dst_aux);
  DRS_CHECK_HIP(hipGetLastError());
  float* bt = dst_aux + (size_t)11 * Cc * Ch;
  DRS_LAUNCH(upfuse_bias_kernel, dim3((Ch + 127) / 128), dim3(128), 0, s, v_b, Ch, bt, bt + 9 * Ch);
  DRS_CHECK_HIP(hipGetLastError());
  {
    const size_t img = drs_upfuse_edge_image_bytes(Cc, Ch) / 2;
    DRS_LAUNCH(upfuse_edge_pack_kernel, dim3((unsigned)((img / 16 + 255) / 256)), dim3(256), 0, s, dst_aux, Cc, Ch,
                       (char*)dst_edge, img);
    DRS_CHECK_HIP(hipGetLastError());
  }
  return DRS_OK;
}

int drs_launch_upfuse_fold_proj(const float* v_w, const float* v_b, const float* fw, const float* fb, int fuse_dim, int Cc, int Ch,
                                float* dst_w, float* dst_b, hipStream_t s) {
  DRS_REQUIRE(v_w && fw && dst_w && dst_b && fuse_dim >= 1 && fuse_dim <= 4, DRS_ERR_ARG, "upfuse_fold_proj: bad arguments");
  DRS_LAUNCH(upfuse_fold_proj_kernel, dim3((32 * (Cc + Ch) * 9 + 255) / 256), dim3(256), 0, s, v_w, v_b, fw, fb, fuse_dim, Cc, Ch,
             dst_w, dst_b);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

size_t drs_upfuse_edge_image_bytes(int Cc, int Ch) { return (size_t)2 * 2 * (Cc / 32) * 7 * 4 * Ch * 16; }

int drs_launch_upfuse_edges(const UpFuseEdgeDesc& d, hipStream_t s) {
  if ((size_t)d.N * d.LH * d.LW == 0) return DRS_OK;
  DRS_REQUIRE(d.Ch % 32 == 0 && d.Cc % 32 == 0 && d.wimg && d.zero_line, DRS_ERR_SHAPE, "upfuse_edges: Cc=%d Ch=%d", d.Cc, d.Ch);
  const int segs = drs_cdiv(d.LH > d.LW ? d.LH : d.LW, 16);
  DRS_LAUNCH(upfuse_edges_mfma_kernel, dim3(segs * (d.Ch / 32), 2, d.N), dim3(64), 0, s, d,
                     reinterpret_cast<const char*>(d.wimg), drs_upfuse_edge_image_bytes(d.Cc, d.Ch) / 2,
                     reinterpret_cast<const char*>(d.zero_line));
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

int drs_launch_upfuse(const UpFuseDesc& d, hipStream_t s) {
  DRS_REQUIRE(d.in && d.w && d.bias && d.zero_line && (d.out || d.out2 || d.fuse_out), DRS_ERR_ARG, "upfuse: null tensor");
  DRS_REQUIRE(d.Cc % 32 == 0 && d.Ch % 32 == 0 && d.Ch <= 256 && (d.in_cs & 31) == 0 && (d.in_co & 31) == 0, DRS_ERR_SHAPE,
              "upfuse: channels");
  DRS_REQUIRE(!d.res || ((d.res_cs & 31) == 0 && (d.res_co & 31) == 0), DRS_ERR_SHAPE, "upfuse: res slice");
  DRS_REQUIRE(!d.out || ((d.out_cs & 31) == 0 && (d.out_co & 31) == 0), DRS_ERR_SHAPE, "upfuse: out slice");
  DRS_REQUIRE(!d.out2 || (d.post2 && (d.out2_cs & 31) == 0 && (d.out2_co & 31) == 0 && (d.post2_cs & 3) == 0),
              DRS_ERR_SHAPE, "upfuse: out2");
  DRS_REQUIRE(!d.proj, DRS_ERR_ARG, "upfuse: the folded projection runs on drs_launch_upfuse_proj (upfuse_proj_sp.hip)");
  DRS_REQUIRE(!d.fuse_out || (d.Ch == 32 && d.fuse_dim >= 1 && d.fuse_dim <= 4 && d.fuse_w && d.fuse_b && !d.out && !d.out2),
              DRS_ERR_SHAPE, "upfuse: fused projection needs Ch == 32, fuse_dim <= 4 and no wide output");
  DRS_REQUIRE(!d.fuse_acc || (d.fuse_out && !d.res), DRS_ERR_ARG, "upfuse: fuse_acc takes the att-half from fuse_out, not from res");
  DRS_REQUIRE((d.eh == nullptr) == (d.ev == nullptr), DRS_ERR_ARG, "upfuse: edge vectors");
  if ((size_t)d.N * d.LH * d.LW == 0) return DRS_OK;
  const int tiles_y = drs_cdiv(d.LH, 16), tiles_x = drs_cdiv(d.LW, 16), nck = d.Cc / 32;
  const long long nitems = (long long)d.N * tiles_y * tiles_x * (d.Ch / 32);
  int num_cu = 0;
  const void* kern = d.fuse_out ? reinterpret_cast<const void*>(upfuse_sp_kernel<true>)
                                : reinterpret_cast<const void*>(upfuse_sp_kernel<false>);
  static_assert(UfGeom::LDS <= 160 * 1024, "LDS budget");
  {
    const int rc = drs_kernel_prepare(kern, 160 * 1024, &num_cu);
    if (rc) return rc;
  }
  long long blocks = num_cu;  // one 12-wave block per CU
  if (blocks > nitems) blocks = nitems;
  blocks = (blocks + 7) / 8 * 8;
  static const int dbg = getenv("DRS_DEBUG_FLAGS") ? atoi(getenv("DRS_DEBUG_FLAGS")) : 0;  // timeline builds: 1 no stores, 2 no residual loads
  if (d.fuse_out)
    DRS_LAUNCH(upfuse_sp_kernel<true>, dim3((unsigned)blocks), dim3(768), UfGeom::LDS, s, d, tiles_y, tiles_x, nck, dbg);
  else
    DRS_LAUNCH(upfuse_sp_kernel<false>, dim3((unsigned)blocks), dim3(768), UfGeom::LDS, s, d, tiles_y, tiles_x, nck, dbg);
  DRS_CHECK_HIP(hipGetLastError());
#ifdef DRS_SP_TIMELINE
  {
    unsigned long long h[64];
    hipEvent_t e0, e1;
    float ms = 0.f;
    DRS_CHECK_HIP(hipEventCreate(&e0)); DRS_CHECK_HIP(hipEventCreate(&e1));
    DRS_CHECK_HIP(hipEventRecord(e0, s));
    if (d.fuse_out)  // timed repeat (same result)
      DRS_LAUNCH(upfuse_sp_kernel<true>, dim3((unsigned)blocks), dim3(768), UfGeom::LDS, s, d, tiles_y, tiles_x, nck, dbg);
    else
      DRS_LAUNCH(upfuse_sp_kernel<false>, dim3((unsigned)blocks), dim3(768), UfGeom::LDS, s, d, tiles_y, tiles_x, nck, dbg);
    DRS_CHECK_HIP(hipEventRecord(e1, s));
    DRS_CHECK_HIP(hipStreamSynchronize(s));
    DRS_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    DRS_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(drs_uf_tl), sizeof(h)));
    const double sc = h[10] ? 1.0 / (double)h[10] : 0.0;
    fprintf(stderr, "upfuse Cc=%d Ch=%d LH=%d fuse=%d: %.1f us, %llu steps/block, wave0 alive %llu ticks = %.2f GHz, %.0f ticks/step\n", d.Cc, d.Ch,
            d.LH, d.fuse_out ? 1 : 0, ms * 1e3, h[10], h[11], h[11] / (ms * 1e6), h[11] * sc);
    for (int w = 0; w < 2; ++w) {
      const unsigned long long* t = h + 16 * w;
      fprintf(stderr, "   C%d (px%d): top %.0f WLwait %.0f | grp: pre %.0f CLwait %.0f pass0 %.0f rel %.0f | pass1+tail %.0f WRrel %.0f | epi: loads %.0f rest %.0f\n", 4 * w, w,
              t[7] * sc, t[4] * sc, t[0] * sc, t[1] * sc, t[2] * sc, t[3] * sc, t[5] * sc, t[6] * sc, t[8] * sc, t[9] * sc);
    }
    const unsigned long long* t = h + 32;
    fprintf(stderr, "   M0: loads/other %.0f CRwait %.0f vmwait %.0f store %.0f | WRwait %.0f vmwait %.0f winstore %.0f\n", t[0] * sc, t[1] * sc,
            t[2] * sc, t[3] * sc, t[4] * sc, t[5] * sc, t[6] * sc);
  }
#endif
  return DRS_OK;
}

int drs_launch_nchw_to_sp(const float* src, float* dst, int N, int C, int H, int W, hipStream_t s) {
  DRS_REQUIRE(C % 32 == 0, DRS_ERR_SHAPE, "nchw_to_sp: C=%d", C);
  const int64_t slots = (int64_t)N * H * W * (C / 8);
  if (slots == 0) return DRS_OK;
  int64_t b = (slots + 255) / 256;
  if (b > 8192) b = 8192;
  DRS_LAUNCH(nchw_to_sp_kernel, dim3((unsigned)b), dim3(256), 0, s, src, reinterpret_cast<char*>(dst), N, C, H, W);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
