// Wave-specialised 3x3 stride-1 tap-convolution for the wide layers (Cin % KC == 0).  Flavours (template arguments):
//   BNB = 64  Cout % 64 == 0: two channel groups x 4 row-waves per block (described below);
//   BNB = 32  one channel group x 8 row-waves of 2 rows (Cout = 32: up_convs.2, with FUSE = the fused output projection);
//   DUAL      TapConv::dual: conv1 + skip convolution of a residual block from ONE 64-channel operand image, 8 row-waves
//             x 4 channel tiles, the epilogue adds a lane's tiles t (main, ReLU) and t + 2 (skip);
//   HAS2      the block's 1x1 shortcut input as extra one-tap K-chunks.
//
// Same GEMM view, operand slots, MFMA schedule and epilogue as tapconv_mfma_kernel<.., CONV3X3, .., NWG = 2>
// (conv_mfma.hip); what changes is WHO moves the operands and WHEN.  In that kernel every wave alternates between
// "issue the loads of the next chunk" (the wave sits in the vector-memory queue for a memory round trip), "convert +
// store to LDS" and "multiply", all 8 waves of a CU in lock step: the memory system idles while the matrix cores run
// and vice versa (phase timeline).  Here a block is 12 waves on one CU:
//   * 8 consumer waves (4 row-waves x 2 channel groups, 2 per SIMD) read fragments from LDS and issue MFMAs;
//   * 4 mover waves (1 per SIMD) only issue LDS-DMA (global_load_lds_dwordx4, no registers): the chunk's WEIGHTS (the
//     packed image is already in operand format) and the RAW fp32 input window of the next step, in full 128-byte
//     lines, into a staging area.
// Operand conversion of the window (fp32 -> bf16 hi/lo planes, zero padding of the halo) is one short LDS -> LDS pass
// of the consumer waves between two steps (5-6 quads per thread, each element converted once).
// LDS (bf16x3): raw window staging 41 KB + window operand planes 42.5 KB + ONE weight buffer of 72 KB = 155.5 KB.
// The weight buffer is a ring of the 3 kernel columns: the consumers copy a column's 6 fragments to registers before
// they multiply with it, so its LDS slot is free again long before the next chunk needs it.
// Step k (one K-chunk of one 16x16 patch x 64 channels): two block-wide barriers + three LDS counters:
//   Y0(k): consumers finished step k-1                   | raw window k and weight column 0 of k have landed
//          C: convert the window, read column 0          | M: nothing (the staging area is being read)
//   Y1(k): operand planes complete                       | column 1 of k has landed
//          C: MFMA column 0, read column 1, bump F1,     | M: bump L2 once column 2 of k has landed, then DMA column 0
//             MFMA column 1, wait L2, read column 2,     |    of k+1 and the raw window k+1 (the long, memory-bound
//             bump F2, MFMA column 2, epilogue           |    burst), wait F1 -> DMA column 1, wait F2 -> DMA column 2
//   F1 / F2 count consumer waves that hold column 1 / 2 in registers (ring slot free), L2 counts mover waves whose part
//   of column 2 has landed.  The movers sit in the vector-memory issue queue for most of a step; no barrier falls inside
//   that burst, so the consumers never wait for it.
// Barriers are raw s_barrier + explicit counted s_waitcnt: a __syncthreads() would drain the DMAs in flight.  The
// consumers never wait on the vector-memory counter inside the loop, so the epilogue's stores drain in the background
// (all of its loads are issued before its first store).
// Measured (DRS_WS_TIMELINE build, up_convs.1, ticks per step): conversion 2050, MFMA phase 8180 for 432 MFMAs per
// SIMD (16 cycles each = 6912: the matrix pipe is the bound inside that phase), epilogue share 690, barrier waits 800.
#include <stdio.h>
#include <stdlib.h>

#include "conv_epilogue.h"
#include "mfma_policy.h"

namespace {

__device__ __forceinline__ void ws_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void ws_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// wait until at most n of this wave's vector-memory operations are outstanding (n is wave-uniform)
__device__ __forceinline__ void ws_wait_vm(int n) {
#define DRS_WS_CASE(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break;
  switch (n) {
    DRS_WS_CASE(1) DRS_WS_CASE(2) DRS_WS_CASE(3) DRS_WS_CASE(4) DRS_WS_CASE(5) DRS_WS_CASE(6) DRS_WS_CASE(7)
    DRS_WS_CASE(8) DRS_WS_CASE(9) DRS_WS_CASE(10) DRS_WS_CASE(11) DRS_WS_CASE(12) DRS_WS_CASE(13) DRS_WS_CASE(14)
    DRS_WS_CASE(15) DRS_WS_CASE(16) DRS_WS_CASE(17) DRS_WS_CASE(18) DRS_WS_CASE(19) DRS_WS_CASE(20) DRS_WS_CASE(21)
    DRS_WS_CASE(22) DRS_WS_CASE(23) DRS_WS_CASE(24) DRS_WS_CASE(25) DRS_WS_CASE(26) DRS_WS_CASE(27) DRS_WS_CASE(28)
    DRS_WS_CASE(29) DRS_WS_CASE(30) DRS_WS_CASE(31)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef DRS_WS_CASE
}

#ifdef DRS_WS_TIMELINE
__device__ unsigned long long drs_ws_tl[48];
#define WS_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl[i] += t_ - tl_last; tl_last = t_; } while (0)
#else
#define WS_STAMP(i) do { } while (0)
#endif

typedef __attribute__((address_space(1))) void* ws_gptr;
typedef __attribute__((address_space(3))) unsigned* ws_flag_ptr;

// spin until the LDS counter *f reaches `target`.  A protocol error must never leave waves spinning on the GPU (a hung
// wave can take the whole node down): after 2^24 polls (seconds; a legitimate wait is
// microseconds) the wave traps: a GPU fault reported from this kernel means a protocol timeout, not a bad address.
__device__ __forceinline__ void ws_poll(ws_flag_ptr f, unsigned target) {
  unsigned spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    __builtin_amdgcn_s_sleep(4);
    if (++spins > (1u << 24)) __builtin_trap();
  }
}
typedef __attribute__((address_space(3))) void* ws_lptr;

template <class P, int BNB_ = 64>
struct WsGeom {
  static constexpr int KC = 4 * P::SLOT_CH;
  static constexpr int IW = 18, NPIX = 18 * 18;
  static constexpr int QPP = KC / 4;                        // 16-byte fp32 quads per pixel and chunk
  static constexpr int NQUAD = NPIX * QPP;                  // quads of one raw window
  static constexpr int NPIECE = (NQUAD + 63) / 64;          // 1 KB DMA pieces of one raw window
  static constexpr int STAGE = NPIECE * 1024;               // bytes of the raw staging area
  static constexpr int APS = (NPIX * 16 + 255) / 256 * 256 + 64;  // k-group plane stride: 64 B skew = conflict-free quad writes
  static constexpr int A_IMAGE = 4 * APS;
  static constexpr int BNB = BNB_;  // output channels per block: 64 = 2 channel groups x 4 row-waves, 32 = 1 x 8
  static constexpr int W_IMAGE = 9 * 4 * BNB * 16;
  static constexpr int LDS = STAGE + P::IMAGES * (A_IMAGE + W_IMAGE) + 16;  // + the three counters
};

// DUAL: conv1 + skip convolution of a residual block in one pass (TapConv::dual): 64 weight channels [main | skip], 8
// row-waves x 2 rows, 4 channel tiles per wave, so both halves of an output pixel meet in one lane
template <class P, bool HAS2, int BNB_ = 64, bool FUSE = false, bool DUAL = false>
__global__ __launch_bounds__(768, 1) void tapconv_ws_kernel(TapConv d, MfmaGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = WsGeom<P, BNB_>;
  constexpr int KC = G::KC, IW = G::IW, QPP = G::QPP, NQUAD = G::NQUAD, NPIECE = G::NPIECE;
  constexpr int APS = G::APS, A_IMAGE = G::A_IMAGE, W_IMAGE = G::W_IMAGE, BNB = G::BNB;
  constexpr int NT = DUAL ? 4 : 2, BN = 32, TH = 16, TW = 16;
  constexpr int NRW = DUAL ? 8 : 8 * BN / G::BNB;  // row-waves per channel group: 4 (64 channels per block) or 8 (32, dual)
  constexpr int RPW = TH / NRW;         // patch rows per consumer wave: 4 or 2
  constexpr int WPI = P::IMAGES * 12 / 4;        // weight pieces per mover wave and kernel column
  constexpr int W2PI = (P::IMAGES * 4 + 3) / 4;  // ... of the second input's single tap
  constexpr int NPW = (NPIECE + 3) / 4;          // raw window pieces per mover wave (at most)
  constexpr int CV_ITERS = (NQUAD + 511) / 512;  // conversion quads per consumer thread
  char* sStage = smem;                           // [window pixel][quad] raw fp32
  char* sA = smem + G::STAGE;                    // [image][kgroup(4)][window pixel] operand slots
  char* sW = sA + P::IMAGES * A_IMAGE;           // [image][kx(3)][ky(3)][kgroup(4)][BNB] operand slots
  ws_flag_ptr sFlag = (ws_flag_ptr)(sW + P::IMAGES * W_IMAGE);  // consumer waves that have read column 1 so far
  ws_flag_ptr sFlag2 = sFlag + 1;                                // ... column 2
  ws_flag_ptr sLand2 = sFlag + 2;                                // mover waves whose part of column 2 has landed

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7 consumers, 8..11 movers
  const bool mover = wid >= 8;
  const int lr = lane & 15, kg = lane >> 4;

  // persistent blocks, XCD-aware item order (see tapconv_mfma_kernel)
  const int ngroups = DUAL ? 1 : d.Cout / BNB;
  const int wcout = DUAL ? 2 * d.Cout : d.Cout;  // channels of the packed weight image
  const int nitems = d.N * g.tiles_y * g.tiles_x * ngroups;
  const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
  const int per = (nitems + 7) >> 3;
  const int lo_item = xcd * per, hi_item = min(nitems, lo_item + per);
  const int span = hi_item - lo_item - j8;
  const int my_items = span > 0 ? (span + nb8 - 1) / nb8 : 0;
  const int nck = g.nchunks + (HAS2 ? g.nchunks2 : 0);
  const int S = my_items * nck;
  if (S == 0) return;
  auto item_of = [&](int ordinal, int& n_, int& ty0_, int& tx0_, int& n0_) {
    int it = lo_item + ordinal * nb8 + j8;
    n0_ = (it % ngroups) * BNB;
    it /= ngroups;
    tx0_ = (it % g.tiles_x) * TW;
    it /= g.tiles_x;
    ty0_ = (it % g.tiles_y) * TH;
    n_ = it / g.tiles_y;
  };

  // ---- mover state -----------------------------------------------------------------------------------------------
  const int pw = wid - 8;
  const char* wg = reinterpret_cast<const char*>(d.w);
  const size_t w_chunk = (size_t)9 * 4 * wcout * 16;
  const size_t w_gimage = DUAL ? 2 * (size_t)g.w_gimage : (size_t)g.w_gimage;  // (the geometry was sized for d.Cout channels)
  int vm_issued = 0;               // vector-memory operations this wave has issued so far (program order)
  int end_col[3] = {0, 0, 0};      // issue count right after the DMA group that fills weight ring slot j
  int end_win = 0;                 // ... after the raw window group
  auto issue_col = [&](int col, int c_, int n0_) {
    const bool second = HAS2 && c_ >= g.nchunks;
    if constexpr (BNB == 64) {
      if (!second) {
  #pragma unroll
        for (int i = 0; i < WPI; ++i) {
          const int idx = pw * WPI + i;  // (image, ky, k-group) piece of this wave
          const int im = idx / 12, ky = (idx % 12) >> 2, kq = idx & 3;
          const char* src = wg + (size_t)im * w_gimage + (size_t)c_ * w_chunk +
                            ((size_t)((ky * 3 + col) * 4 + kq) * wcout + n0_ + lane) * 16;
          char* dst = sW + (size_t)im * W_IMAGE + (size_t)((col * 3 + ky) * 4 + kq) * BNB * 16;
          __builtin_amdgcn_global_load_lds((ws_gptr)src, (ws_lptr)dst, 16, 0, 0);
        }
        vm_issued += WPI;
      } else if (col == 0) {  // second input: one tap, stored where (kx 0, ky 0) lives
        const int cc = c_ - g.nchunks;
  #pragma unroll
        for (int i = 0; i < W2PI; ++i) {
          const int idx = pw * W2PI + i;
          const int im = idx >> 2, kq = idx & 3;
          if (im < P::IMAGES) {
            const char* src = reinterpret_cast<const char*>(d.w2) + (size_t)im * g.w2_gimage +
                              ((size_t)(cc * 4 + kq) * wcout + n0_ + lane) * 16;
            char* dst = sW + (size_t)im * W_IMAGE + (size_t)kq * BNB * 16;
            __builtin_amdgcn_global_load_lds((ws_gptr)src, (ws_lptr)dst, 16, 0, 0);
            vm_issued += 1;
          }
        }
      }
    } else {
      // 32 channels per block: a 1 KB piece = two consecutive k-group rows of 32 channels
      constexpr int RPP = 2, WPIECES = P::IMAGES * 12 / RPP, WPI32 = (WPIECES + 3) / 4;
      constexpr int W2PIECES = P::IMAGES * 4 / RPP, W2PI32 = (W2PIECES + 3) / 4;
      const int prow = lane >> 5, pco = lane & 31;
      if (!second) {
#pragma unroll
        for (int i = 0; i < WPI32; ++i) {
          const int idx = pw * WPI32 + i;
          if (idx < WPIECES) {
            const int row0 = idx * RPP, im = row0 / 12, ky = (row0 % 12) >> 2, kq0 = row0 & 3;
            const char* src = wg + (size_t)im * w_gimage + (size_t)c_ * w_chunk +
                              ((size_t)((ky * 3 + col) * 4 + kq0 + prow) * d.Cout + n0_ + pco) * 16;
            char* dst = sW + (size_t)im * W_IMAGE + (size_t)((col * 3 + ky) * 4 + kq0) * BNB * 16;
            __builtin_amdgcn_global_load_lds((ws_gptr)src, (ws_lptr)dst, 16, 0, 0);
            vm_issued += 1;
          }
        }
      } else if (col == 0) {
        const int cc = c_ - g.nchunks;
#pragma unroll
        for (int i = 0; i < W2PI32; ++i) {
          const int idx = pw * W2PI32 + i;
          if (idx < W2PIECES) {
            const int row0 = idx * RPP, im = row0 >> 2, kq0 = row0 & 3;
            const char* src = reinterpret_cast<const char*>(d.w2) + (size_t)im * g.w2_gimage +
                              ((size_t)(cc * 4 + kq0 + prow) * d.Cout + n0_ + pco) * 16;
            char* dst = sW + (size_t)im * W_IMAGE + (size_t)kq0 * BNB * 16;
            __builtin_amdgcn_global_load_lds((ws_gptr)src, (ws_lptr)dst, 16, 0, 0);
            vm_issued += 1;
          }
        }
      }
    }
    end_col[col] = vm_issued;
  };
  // raw fp32 window of a step -> staging, 128-byte lines; pieces [i0, i1) of this wave's share
  auto issue_win = [&](int c_, int n_, int ty_, int tx_, int i0, int i1) {
    const bool second = HAS2 && c_ >= g.nchunks;
    const int cc = second ? c_ - g.nchunks : c_;
    const int nq = min(QPP, ((second ? d.Cin2 : d.Cin) - cc * KC) >> 2);  // real quads of this chunk
    int lane_o = lane;  // opaque: the per-piece address arithmetic is recomputed here, not hoisted (and spilled)
    asm volatile("" : "+v"(lane_o));
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int j = pw + 4 * i;  // piece (wave-uniform)
      if (i >= i0 && i < i1 && j < NPIECE) {
        const int e = min(j * 64 + lane_o, NQUAD - 1);
        // a last chunk with fewer than KC channels (Cin = 16 on the 256x256 level): its missing quads fetch a duplicate
        // of the last real one - finite data that meets the zero padding of the packed weights
        const int p = e / QPP, quad = min(e % QPP, nq - 1), py = p / IW, px = p % IW;
        const float* src;
        if (!second) {
          const int iy = min(max(ty_ - 1 + py, 0), d.H - 1), ix = min(max(tx_ - 1 + px, 0), d.W - 1);
          src = d.in + ((size_t)n_ * d.H * d.W + (size_t)(iy * d.W + ix)) * d.in_cs + d.in_co + cc * KC + quad * 4;
        } else {
          const int iy = min(ty_ + py, d.H2 - 1), ix = min(tx_ + px, d.W2 - 1);
          src = d.in2 + ((size_t)n_ * d.H2 * d.W2 + (size_t)(iy * d.W2 + ix)) * d.in2_cs + d.in2_co + cc * KC + quad * 4;
        }
        __builtin_amdgcn_global_load_lds((ws_gptr)src, (ws_lptr)(sStage + j * 1024), 16, 0, 0);
        vm_issued += 1;
      }
    }
    end_win = vm_issued;
  };

  // ---- consumer state --------------------------------------------------------------------------------------------
  const int rw = (BNB == 64 && !DUAL) ? (wid & 3) : (wid & 7);         // row-wave: rows [rw*RPW, rw*RPW + RPW) of the patch
  const int ng = (BNB == 64 && !DUAL) ? ((wid >> 2) & 1) : 0;          // channel group: channels [ng*BN, ng*BN + BN) of the block's BNB
  const char* win = sA + kg * APS + (rw * RPW * IW + lr) * 16;           // this lane's window origin
  const char* wbase = sW + ((size_t)kg * BNB + ng * NT * 16 + lr) * 16;  // this lane's weight origin
  f32x4 acc[RPW][NT];
  typename P::Frag wf[3][NT];
  auto read_wf = [&](int col) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        wf[ky][t] = P::load(wbase, W_IMAGE, (size_t)(((col * 3 + ky) * 4 * BNB) + t * 16) * 16);
  };
  auto mma_col = [&](int col) {
#pragma unroll
    for (int wr = 0; wr < RPW + 2; ++wr) {
      const typename P::Frag af = P::load(win, A_IMAGE, (size_t)(wr * IW + col) * 16);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int r = wr - ky;
        if (r >= 0 && r < RPW) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][t] = P::mma(wf[ky][t], af, acc[r][t]);
        }
      }
    }
  };
  // ---- window conversion: raw staging -> operand planes, the 512 consumer threads (the movers, youngest waves of their
  //      SIMDs, would finish last and stretch the phase) -------------------------------------------
  int cv_src[CV_ITERS], cv_dst[CV_ITERS], cv_yx[CV_ITERS];  // chunk-independent per-thread quad descriptors
#pragma unroll
  for (int it = 0; it < CV_ITERS; ++it) {
    const int e = min(tid + it * 512, NQUAD - 1);
    const int p = e / QPP, quad = e % QPP;
    cv_src[it] = e * 16;
    cv_dst[it] = P::IMAGES == 2 ? (quad >> 1) * APS + p * 16 + (quad & 1) * 8 : quad * APS + p * 16;
    cv_yx[it] = ((p / IW) << 8) | (p % IW);
  }
  auto convert = [&](bool second, int ty_, int tx_) {
    // patches whose window lies inside the image need no zero padding (wave-uniform test)
    const bool interior = second ? (ty_ + TH <= d.H2 && tx_ + TW <= d.W2)
                                 : (ty_ >= 1 && ty_ + TH + 1 <= d.H && tx_ >= 1 && tx_ + TW + 1 <= d.W);
    f32x4 v[CV_ITERS];
#pragma unroll
    for (int it = 0; it < CV_ITERS; ++it) v[it] = *reinterpret_cast<const f32x4*>(sStage + cv_src[it]);  // all reads in flight
    if (!interior) {
#pragma unroll
      for (int it = 0; it < CV_ITERS; ++it) {
        const int py = cv_yx[it] >> 8, px = cv_yx[it] & 255;
        bool ok;
        if (!second) {
          const int iy = ty_ - 1 + py, ix = tx_ - 1 + px;
          ok = iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
        } else {
          ok = py < TH && px < TW && ty_ + py < d.H2 && tx_ + px < d.W2;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[it][j] = ok ? v[it][j] : 0.f;
      }
    }
#pragma unroll
    for (int it = 0; it < CV_ITERS; ++it) {
      if (it < CV_ITERS - 1 || tid + it * 512 < NQUAD) {
        if constexpr (P::IMAGES == 2) {
          uint2 h, l;
          { const uint2 s_ = drs_split2(v[it][0], v[it][1]); h.x = s_.x; l.x = s_.y; }
          { const uint2 s_ = drs_split2(v[it][2], v[it][3]); h.y = s_.x; l.y = s_.y; }
          *reinterpret_cast<uint2*>(sA + cv_dst[it]) = h;
          *reinterpret_cast<uint2*>(sA + cv_dst[it] + A_IMAGE) = l;
        } else {
          *reinterpret_cast<f32x4*>(sA + cv_dst[it]) = v[it];
        }
      }
    }
  };

#ifdef DRS_WS_TIMELINE
  unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_amdgcn_s_memtime();
  const unsigned long long tl_begin = tl_last;
#endif
  if (tid < 3) __hip_atomic_store(sFlag + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  ws_wait_lds();
  ws_barrier();
  int c = -1, ord = -1, n = 0, ty0 = 0, tx0 = 0, n0 = 0;  // current step: chunk, item ordinal, item coordinates
  if (mover) {
    // ===================== movers =====================
    {
      int n_, ty_, tx_, n0_;
      item_of(0, n_, ty_, tx_, n0_);
      issue_win(0, n_, ty_, tx_, 0, NPW);
      issue_col(0, 0, n0_);
      issue_col(1, 0, n0_);
      issue_col(2, 0, n0_);
    }
    for (int k = 0; k < S; ++k) {
      if (++c == nck) c = 0;
      if (c == 0) item_of(++ord, n, ty0, tx0, n0);
      WS_STAMP(0);
      ws_wait_vm(k ? vm_issued - max(end_win, end_col[0]) : 0);  // raw window k and column 0 of k have landed
      WS_STAMP(1);
      ws_barrier();  // Y0
      WS_STAMP(2);
      ws_wait_vm(vm_issued - end_col[1]);  // column 1 of k
      WS_STAMP(3);
      ws_barrier();  // Y1
      WS_STAMP(4);
      // column 2 of k was issued last in the previous burst and is read late in the step: certified by a counter
      ws_wait_vm(vm_issued - end_col[2]);
      if (lane == 0) __hip_atomic_fetch_add(sLand2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (k + 1 < S) {
        int c1 = c + 1, n1 = n, ty1 = ty0, tx1 = tx0, n01 = n0;
        if (c1 == nck) {
          c1 = 0;
          item_of(ord + 1, n1, ty1, tx1, n01);
        }
        issue_col(0, c1, n01);
        issue_win(c1, n1, ty1, tx1, 0, NPW);
        WS_STAMP(5);
        const unsigned target = 8u * (unsigned)(k + 1);
        ws_poll(sFlag, target);  // every consumer wave has read column 1 of k
        WS_STAMP(6);
        issue_col(1, c1, n01);
        ws_poll(sFlag2, target);
        issue_col(2, c1, n01);
      }
      WS_STAMP(7);
    }
  } else {
    // ===================== consumers =====================
    for (int k = 0; k < S; ++k) {
      if (++c == nck) c = 0;
      if (c == 0) {
        item_of(++ord, n, ty0, tx0, n0);
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const bool second = HAS2 && c >= g.nchunks;
      WS_STAMP(0);
      ws_wait_lds();
      ws_barrier();  // Y0
      WS_STAMP(1);
      convert(second, ty0, tx0);
      read_wf(0);
      WS_STAMP(2);
      ws_wait_lds();
      ws_barrier();  // Y1
      WS_STAMP(3);
      if (second) {  // second input: one tap, window origin; columns 1 and 2 are empty
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          const typename P::Frag af = P::load(win, A_IMAGE, (size_t)(r * IW) * 16);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][t] = P::mma(wf[0][t], af, acc[r][t]);
        }
        ws_wait_lds();
        if (lane == 0) {
          __hip_atomic_fetch_add(sFlag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(sFlag2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {
        mma_col(0);
        read_wf(1);
        WS_STAMP(4);
        ws_wait_lds();
        if (lane == 0) __hip_atomic_fetch_add(sFlag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // column 1 is in registers: its ring slot may be refilled
        WS_STAMP(5);
        mma_col(1);
        ws_poll(sLand2, 4u * (unsigned)(k + 1));
        read_wf(2);
        ws_wait_lds();
        if (lane == 0) __hip_atomic_fetch_add(sFlag2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        mma_col(2);
      }
      WS_STAMP(6);
      if (c == nck - 1) {
        // opaque copies of the lane coordinates: everything the epilogue derives from them is computed here, not hoisted
        // out of the step loop (where it would be spilled and reloaded with vmcnt(0) waits between the stores)
        int lr_e = lr, kg_e = kg;
        asm volatile("" : "+v"(lr_e), "+v"(kg_e));
        if constexpr (DUAL) {
          // out = relu(main + b_main) + post_add + (skip + b_skip): tiles t (main) and t + 2 (skip) of the same lane
          f32x4 comb[RPW][2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const float4 b1 = *reinterpret_cast<const float4*>(d.bias + t * 16 + kg_e * 4);
            const float4 b2 = *reinterpret_cast<const float4*>(d.bias + d.Cout + t * 16 + kg_e * 4);
            const float bm[4] = {b1.x, b1.y, b1.z, b1.w}, bs[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
              for (int j = 0; j < 4; ++j) comb[r][t][j] = drs_maxf(acc[r][t][j] + bm[j], 0.f) + (acc[r][t + 2][j] + bs[j]);
          }
          TapConv de = d;
          de.bias = nullptr; de.bias2 = nullptr; de.relu_pre = 0;
          tile_epilogue<RPW, 2, false, RPW>(de, comb, n, 0, ty0, tx0, rw, lr_e, kg_e, d.out_oy, d.out_ox);
        } else if constexpr (FUSE)
          fuse_epilogue<RPW, NT>(d, acc, n, n0 + ng * BN, ty0, tx0, rw, lr_e, kg_e);
        else
          tile_epilogue<RPW, NT, false, RPW>(d, acc, n, n0 + ng * BN, ty0, tx0, rw, lr_e, kg_e, d.out_oy, d.out_ox);
      }
      WS_STAMP(7);
    }
  }
  if (mover) ws_wait_vm(0);
#ifdef DRS_WS_TIMELINE
  if (blockIdx.x == 0 && (wid == 0 || wid == 4 || wid == 8) && lane == 0) {
    for (int i = 0; i < 8; ++i) drs_ws_tl[(wid == 0 ? 0 : (wid == 4 ? 32 : 16)) + i] = tl[i];
    drs_ws_tl[wid == 0 ? 8 : (wid == 4 ? 40 : 24)] = S;
    if (wid == 0) drs_ws_tl[9] = __builtin_amdgcn_s_memtime() - tl_begin;
  }
#endif
}

bool ws_std3x3(const TapConv& d) {
  if (d.mode != 0 || d.ntaps != 9 || d.wtaps_total != 9 || d.in_stride != 1 || d.out_scale != 1) return false;
  for (int i = 0; i < 9; ++i)
    if (d.dy[i] != i / 3 - 1 || d.dx[i] != i % 3 - 1 || d.wtap[i] != i) return false;
  return true;
}

template <class P, bool HAS2, int BNB = 64, bool FUSE = false, bool DUAL = false>
int ws_launch(const TapConv& d, const MfmaGeom& g, hipStream_t s) {
  auto kern = tapconv_ws_kernel<P, HAS2, BNB, FUSE, DUAL>;
  constexpr size_t kLds = WsGeom<P, BNB>::LDS;
  int num_cu = 0;
  {
    const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(kern), 160 * 1024, &num_cu);
    if (rc) return rc;
  }
  static_assert(kLds <= 160 * 1024, "LDS budget");
  const long long nitems = (long long)d.N * g.tiles_x * g.tiles_y * (DUAL ? 1 : d.Cout / BNB);
  long long blocks = num_cu;  // one 12-wave block per CU
  if (blocks > nitems) blocks = nitems;
  blocks = (blocks + 7) / 8 * 8;
  DRS_LAUNCH(kern, dim3((unsigned)blocks), dim3(768), kLds, s, d, g);
  DRS_CHECK_HIP(hipGetLastError());
#ifdef DRS_WS_TIMELINE
  {
    unsigned long long h[48];
    hipEvent_t e0, e1;
    float ms = 0.f;
    DRS_CHECK_HIP(hipEventCreate(&e0)); DRS_CHECK_HIP(hipEventCreate(&e1));
    DRS_CHECK_HIP(hipEventRecord(e0, s));
    DRS_LAUNCH(kern, dim3((unsigned)blocks), dim3(768), kLds, s, d, g);  // timed repeat (same result)
    DRS_CHECK_HIP(hipEventRecord(e1, s));
    DRS_CHECK_HIP(hipStreamSynchronize(s));
    DRS_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    DRS_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(drs_ws_tl), sizeof(h)));
    const double sc = h[8] ? 1.0 / (double)h[8] : 0.0;
    fprintf(stderr, "ws kernel %.1f us, block 0 wave 0 alive %llu ticks (%.2f GHz if it spans the launch), %lld blocks\n", ms * 1e3, h[9],
            h[9] / (ms * 1e6), blocks);
    fprintf(stderr, "ws Cin=%d Cout=%d TH=%d in2=%d S=%llu | C: epi>Y0 %.0f cvt+rd0 %.0f >Y1 %.0f col0+rd1 %.0f flag %.0f col1,2 %.0f epi %.0f | "
            "M: vmwait %.0f Y0 %.0f cvt+dma2 %.0f Y1 %.0f dma0+win %.0f poll %.0f dma1 %.0f\n",
            d.Cin, d.Cout, d.TH, d.in2 ? d.Cin2 : 0, h[8], (h[0] + h[1]) * sc, h[2] * sc, h[3] * sc, h[4] * sc, h[5] * sc, h[6] * sc, h[7] * sc,
            h[17] * sc, h[18] * sc, h[19] * sc, h[20] * sc, h[21] * sc, h[22] * sc, h[23] * sc);
    fprintf(stderr, "   wave4 C: epi>Y0 %.0f cvt+rd0 %.0f >Y1 %.0f col0+rd1 %.0f flag %.0f col1,2 %.0f epi %.0f\n", (h[32] + h[33]) * sc,
            h[34] * sc, h[35] * sc, h[36] * sc, h[37] * sc, h[38] * sc, h[39] * sc);
  }
#endif
  return DRS_OK;
}

}  // namespace

// Eligibility of the wave-specialised kernel (3x3 stride 1 on 16-row patches; the caller has established that geometry).
// DRS_WS: 0 off, bit 0: layers with Cout % 64 == 0; bit 2 (4) adds
// the 32-channel flavour (Cout % 64 != 0, incl. the fused output projection of up_convs.2).  Default 5.
bool drs_tapconv_ws_supported(const TapConv& d, int impl) {
  static const int env = getenv("DRS_WS") ? atoi(getenv("DRS_WS")) : 5;
  if (!env) return false;
  if (impl != DRS_IMPL_MFMA_BF16X3 && impl != DRS_IMPL_MFMA_F32) return false;
  const int KC = impl == DRS_IMPL_MFMA_F32 ? 16 : 32;
  if (!d.in || !ws_std3x3(d) || d.shared_cu || d.gate || d.in_add) return false;
  if (d.in_sp || d.in2_sp || d.out_sp || d.res_sp || d.out2) return false;  // SP-format tensors: conv_mfma_sp.hip
  if (d.dual)  // conv1 + skip in one pass: split-bf16 only, one 32-channel output group, partial K-chunk allowed
    return (env & 4) && impl == DRS_IMPL_MFMA_BF16X3 && d.Cout == 32 && d.Cin % 4 == 0 && d.TH > 8 && !d.in2 && !d.fuse_out &&
           !d.res && d.bias && !(d.in_cs & 3) && !(d.in_co & 3);
  // (partial K-chunks work - the movers duplicate the last real quad - but the 16-channel layers of the 256x256 level
  //  are HBM-bound and measured 1.3 % slower here than on the lock-step kernel)
  if (d.Cout % 32 != 0 || d.Cin % KC != 0 || d.TH <= 8) return false;
  // the 32-channel flavour pays for split-bf16 only (exact-fp32 training step: 37.6 vs 38.1 steps/s)
  const bool narrow_ok = (env & 4) && impl == DRS_IMPL_MFMA_BF16X3;
  if (d.Cout % 64 != 0 && !narrow_ok) return false;
  if (d.fuse_out && (d.Cout != 32 || d.in2 || !narrow_ok)) return false;
  if ((d.in_cs & 3) || (d.in_co & 3)) return false;
  if (d.in2 && (d.Cin2 % KC != 0 || (d.in2_cs & 3) || (d.in2_co & 3))) return false;
  return true;
}

template <class P>
static int ws_dispatch(const TapConv& d, const MfmaGeom& g, hipStream_t s) {
  if (d.dual) return ws_launch<P, false, 64, false, true>(d, g, s);
  if (d.fuse_out) return ws_launch<P, false, 32, true>(d, g, s);
  if (d.Cout % 64 == 0) return d.in2 ? ws_launch<P, true>(d, g, s) : ws_launch<P, false>(d, g, s);
  return d.in2 ? ws_launch<P, true, 32>(d, g, s) : ws_launch<P, false, 32>(d, g, s);
}

int drs_launch_tapconv_ws(const TapConv& d, const MfmaGeom& g, int impl, hipStream_t s) {
  DRS_REQUIRE(g.IH == 18 && g.IW == 18, DRS_ERR_SHAPE, "tapconv_ws: geometry");
  if (impl == DRS_IMPL_MFMA_F32) return ws_dispatch<PolicyF32>(d, g, s);
  return ws_dispatch<PolicyBF16X3>(d, g, s);
}
