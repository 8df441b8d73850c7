// Wave-specialised 3x3 stride-1 tap-convolution over SP-format activations (split bf16 hi | lo, drs_common.h): the
// default kernel of every wide 3x3 layer of the eval split-bf16 plan (conv1 / conv2 (+ fused 1x1 shortcut) of the residual
// blocks, ups.*.conv, up_convs.*; reference UNet_model_superres.py:153-172,197-207,377).
//
// Same GEMM view, MFMA schedule and weight ring as tapconv_ws_kernel (conv_mfma_ws.hip); what is gone is every
// conversion: the producers stored the operand halves, so the mover waves copy the input window STRAIGHT INTO THE OPERAND
// IMAGE (whole 128-byte lines: 8 pixels x [4 hi slots | 4 lo slots] per instruction, global -> registers a step ahead ->
// ds_write_b128; LDS-DMA was measured and is 3x slower per wave, see the mover section), and the consumer waves only read
// fragments and issue MFMAs.  Zero padding costs nothing either: a lane whose window pixel lies outside the image (or
// whose channels lie beyond Cin) takes a line of zeros as its source address.
//   block = 12 waves on one CU: 8 consumer waves (2 per SIMD) + 4 mover waves (1 per SIMD)
//   LDS   = 2 window buffers x 41 KB (double-buffered: window k+1 lands while window k is multiplied)
//           + weight ring of the 3 kernel columns (72 KB at 64 output channels per block) + 10 counters
//           + 2 slots of per-item epilogue constants = 156 KB
//   step k (one 32-channel K-chunk of one 16x16 patch): NO block-wide barrier, ten monotonic LDS counters ("landed" /
//   "released" per window buffer and per weight ring slot; protocol at the mover loop).  Waves drift apart by up to a
//   step: while one wave of a SIMD waits for fragments or writes its tile out, its partner's MFMAs keep the pipe busy.
// In steady state the kernel runs within ~10 % of what the matrix pipe sustains at the clock the chip's power limit
// allows under this load (tools/micro/cons_loop.hip); the shallow 32-channel layers, where its
// per-patch costs dominate, go to conv3x3_direct_sp.hip instead.
// Window image in LDS: like memory, 128 bytes per window pixel (18 x 18 window, row-major) = the pixel's 8 operand slots
// [hi k-groups 0-3 | lo k-groups 0-3], ROTATED by the pixel index: slot c of pixel p sits at position (c + p) & 7.  The
// DMA reads whole lines (8 consecutive lanes = one pixel's 128 bytes: full coalescing; without the rotation a fragment
// read would hit 4-way bank conflicts, with a pixel-minor layout every lane of a DMA would touch a different line and the
// instruction takes 320 instead of 140 cycles to issue).  A fragment read (16 consecutive pixels of one k-group) touches
// 16 different 16-byte bank groups in each of the hardware's ds_read_b128 lane groups (MI355X_MICROARCH.md): conflict-free.
// Lane addresses come from 2 x 8 per-wave tables (one per residue of the window offset mod 8) plus immediates.
// Flavours (template arguments): BNB = 64 (2 channel groups x 4 row-waves) / 32 (1 x 8 row-waves); HAS2 = the block's 1x1
// shortcut input as extra one-tap K-chunks; FUSE = up_convs.2 with the fused `output` projection (fp32 NCHW result);
// DUAL = conv1 + skip convolution of the first residual block from one 64-channel operand image; F32OUT = SP-format input,
// fp32 channels-last output through the general epilogue (residual accumulate included): the data-gradient convolutions of
// the training step, whose inputs (dZ of a BatchNorm backward) are written in SP form by bn_bwd_apply_kernel and whose
// outputs feed fp32 consumers (train_bwd.inc).
#include <stdio.h>
#include <stdlib.h>

#include "conv_epilogue.h"
#include "mfma_policy.h"
#include "sp_sync.h"

namespace {

#ifdef DRS_SP_TIMELINE
__device__ unsigned long long drs_sp_tl[48];
#define SP_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl[i] += t_ - tl_last; tl_last = t_; } while (0)
#else
#define SP_STAMP(i) do { } while (0)
#endif

// consumer release of a ring slot + instruction priority for what follows.  The two consumer waves of a SIMD share its
// matrix pipe and the hardware arbitrates "oldest first": left alone, the older wave runs a column at full speed while the
// younger one crawls, then blocks one step ahead at the ring and idles while the younger one runs ALONE (LDS latencies
// exposed: 63 % pipe occupancy measured).  The counter value returned by the release tells how many of the 8 consumers
// were here before this wave: the second half to arrive (the waves that are behind) raise their priority, the first half
// lower it, so the partners stay within a column of each other and one multiplies while the other waits for fragments.
__device__ __forceinline__ void sp_bump_prio(sp_flag_ptr f, unsigned step_base, int lane) {
  unsigned old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  const unsigned rank = (unsigned)__builtin_amdgcn_readfirstlane((int)old) - step_base;
  if (rank >= 4) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(0);
}

template <int BNB_>
struct SpGeom {
  static constexpr int IW = 18, NPIX = 18 * 18;
  static constexpr int NBLK = (NPIX + 7) / 8;      // pixel blocks of 8 = 1 KB DMA pieces per window
  static constexpr int WBUF = NBLK * 1024;         // bytes of one window buffer
  static constexpr int BNB = BNB_;
  static constexpr int W_IMAGE = 9 * 4 * BNB * 16;  // one operand image (hi or lo) of a chunk's weights
  static constexpr int EPI = 768;                   // bytes of one slot of per-item epilogue constants (two slots)
  static constexpr int LDS = 2 * WBUF + 2 * W_IMAGE + 64 + 2 * EPI;
};

template <bool HAS2, int BNB_, bool FUSE, bool DUAL, bool F32OUT = false>
__global__ __launch_bounds__(768, 1) void tapconv_sp_kernel(TapConv d, MfmaGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  using G = SpGeom<BNB_>;
  constexpr int KC = 32, IW = G::IW, NBLK = G::NBLK, WBUF = G::WBUF, W_IMAGE = G::W_IMAGE, BNB = G::BNB;
  constexpr int NT = DUAL ? 4 : 2, BN = 32, TH = 16, TW = 16;
  constexpr int NRW = DUAL ? 8 : 8 * BN / BNB;  // row-waves per channel group: 4 (64 channels per block) or 8 (32, dual)
  constexpr int RPW = TH / NRW;                  // patch rows per consumer wave: 4 or 2
  char* sWin = smem;                             // [buffer 2][window pixel][rotated operand slot 8] x 16 bytes
  char* sW = smem + 2 * WBUF;                    // [image][kx(3)][ky(3)][k-group(4)][BNB] operand slots
  // ten monotonic counters, no barrier inside the step loop (the waves of a SIMD drift to complementary phases: one
  // multiplies while its partner waits for fragments or writes its tile out):
  sp_flag_ptr sCR = (sp_flag_ptr)(sW + 2 * W_IMAGE);  // CR[3]: consumer waves that hold weight column j of their step in registers
  sp_flag_ptr sCL = sCR + 3;                          // CL[3]: mover waves whose part of weight column j has landed
  sp_flag_ptr sWL = sCR + 6;                          // WL[2]: mover waves whose part of window buffer b has landed
  sp_flag_ptr sWR = sCR + 8;                          // WR[2]: consumer waves that have finished reading window buffer b
  // per-item epilogue constants, staged by one mover wave with the item's first step (slot = item ordinal & 1): bias (+ bias2) |
  // post_add | post2 of the block's channels (fused projection: bias | fuse_w rows | fuse_b).  A global load in the
  // consumers' epilogue would expose a memory round trip per item on every SIMD.
  float* sEpi = reinterpret_cast<float*>(sW + 2 * W_IMAGE + 64);
  constexpr int EC = DUAL ? 64 : BNB;                 // floats per constant vector

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
  // K-step order of an item with a second input (the block's 1x1 shortcut): its one-tap steps FOLLOW the 3x3 steps.  They
  // cost 3 - 3.7 us each for 0.6 us of MFMAs (a mover starts the loads of step k + 1 when it is done with step k, so a short
  // step exposes a memory round trip).  Interleaving them with the 3x3 steps (L S L S ..., round 4) made it worse - conv2 +
  // shortcut of block 1: 74 -> 93 us - because every 3x3 window then lands in the SAME buffer and its 41 KB store can only
  // start when the previous 3x3 step ends: the double buffering is gone.  The one-tap steps FIRST (their load chains under the
  // previous item's epilogue): 84 -> 89 us in the build that had the switch - no gain either.  Third attempt: interleaved, with the
  // one-tap windows in the weight ring's column-1 / -2 slots (free during a one-tap step) so that the 3x3 windows keep
  // alternating buffers - parity-green, 68 / 64 / 63 -> 79 / 70 / 70 us, and the 64- and 32-channel flavours would then sum their
  // K-chunks in different orders (batch independence 7e-6 instead of exact).  The appended order stays.
  auto step_kind = [&](int c_, bool& second_, int& cc_) __attribute__((always_inline)) {
    second_ = HAS2 && c_ >= g.nchunks;
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

#ifdef DRS_SP_TIMELINE
  unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_amdgcn_s_memtime();
  const unsigned long long tl_begin = tl_last;
#endif
  if (tid < 10) __hip_atomic_store(sCR + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  sp_wait_lds();
  sp_barrier();
  // Experiment (DRS_DEBUG_FLAGS bits 8-11 = unit u): block b starts (b >> 3 & 7) * u * ~0.25 us late, so the CUs of an XCD
  // reach their item epilogues - 64 KB of stores per CU, issued by all CUs of the chip within the same microsecond - at
  // eight different times.
  {
    const int u = (g.debug >> 8) & 15;
    if (u) {
      const int ph = (blockIdx.x >> 3) & 7;
      for (int i = 0; i < ph * u; ++i) __builtin_amdgcn_s_sleep(8);
    }
  }
  int c = -1, ord = -1, n = 0, ty0 = 0, tx0 = 0, n0 = 0;  // current step: chunk, item ordinal, item coordinates

  if (mover) {
#define SP_NCONS 8
#define SP_MOVER_PRIO 3
#include "conv_sp_movers.inc"
#undef SP_NCONS
#undef SP_MOVER_PRIO
  } else {
    // ===================== consumers =====================
    const int rw = (BNB == 64 && !DUAL) ? (wid & 3) : (wid & 7);  // row-wave: rows [rw*RPW, rw*RPW + RPW) of the patch
    const int ng = (BNB == 64 && !DUAL) ? ((wid >> 2) & 1) : 0;   // channel group: channels [ng*BN, ng*BN + BN) of the block's BNB
    // fragment addresses: window pixel p = B + q + lr with B = rw*RPW*18 (per wave) and q = wr*18 + kx (compile time);
    // slot address = p * 128 + ((c + p) & 7) * 16, c = image * 4 + k-group.  The rotation depends on q only through q & 7:
    // one lane table per residue (hi and lo image), the rest (q * 128) is an immediate.
    const int B = rw * RPW * IW;
    int tab_hi[8], tab_lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int rot = (kg + B + j + lr) & 7;
      tab_hi[j] = (B + lr) * 128 + rot * 16;
      tab_lo[j] = (B + lr) * 128 + (rot ^ 4) * 16;
    }
    const char* wbase = sW + ((size_t)kg * BNB + ng * NT * 16 + lr) * 16;  // this lane's weight origin
    f32x4 acc[RPW][NT];
    typename P::Frag wf[3][NT];
    auto read_wf = [&](int col) __attribute__((always_inline)) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          wf[ky][t] = P::load(wbase, W_IMAGE, (size_t)(((col * 3 + ky) * 4 * BNB) + t * 16) * 16);
    };
    auto win_frag = [&](const char* buf, int q) __attribute__((always_inline)) {  // q: compile-time window offset
      return typename P::Frag{*reinterpret_cast<const bf16x8*>(buf + tab_hi[q & 7] + q * 128),
                              *reinterpret_cast<const bf16x8*>(buf + tab_lo[q & 7] + q * 128)};
    };
    // MFMAs of kernel column `col`.  PRE: the weight fragments of column col + 1 are read into wf[ky] as soon as the last
    // window row that needs the old wf[ky] has been issued (row ky + RPW - 1), behind a poll of that column's "landed"
    // counter: the fragment-read latency of the next column hides under the tail of this one.
    auto mma_col = [&](const char* buf, int col, bool pre, unsigned ltarget) __attribute__((always_inline)) {
#pragma unroll
      for (int wr = 0; wr < RPW + 2; ++wr) {
        const typename P::Frag af = win_frag(buf, wr * IW + col);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int r = wr - ky;
          if (r >= 0 && r < RPW) {
#ifdef DRS_SP_TIMELINE
            if (g.debug & 4) continue;
#endif
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[r][t] = P::mma(wf[ky][t], af, acc[r][t]);
          }
        }
        if (pre) {
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
            if (wr == ky + RPW - 1) {
              if (ky == 0) sp_poll_lds(sCL + col + 1, ltarget, d.fault);
#pragma unroll
              for (int t = 0; t < NT; ++t)
                wf[ky][t] = P::load(wbase, W_IMAGE, (size_t)((((col + 1) * 3 + ky) * 4 * BNB) + t * 16) * 16);
            }
        }
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
      const unsigned ltarget = 4u * (unsigned)(k + 1);  // four movers per column and step
      SP_STAMP(7);
      sp_poll_lds(sWL + (k & 1), 4u * (unsigned)((k >> 1) + 1), d.fault);  // window k is in its buffer
      sp_poll_lds(sCL, ltarget, d.fault);                                   // ... and weight column 0 of k
      SP_STAMP(0);
      read_wf(0);
      sp_wait_lds();
      sp_bump_prio(sCR, 8u * (unsigned)k, lane);  // column 0 is in registers: its ring slot may be refilled
      SP_STAMP(1);
      if (second) {  // second input: one tap, window origin; columns 1 and 2 are empty
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          const typename P::Frag af = win_frag(buf, r * IW);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][t] = P::mma(wf[0][t], af, acc[r][t]);
        }
        sp_wait_lds();
        if (lane == 0) {
          sp_bump(sCR + 1);
          sp_bump(sCR + 2);
          sp_bump(sWR + (k & 1));
        }
      } else {
        mma_col(buf, 0, true, ltarget);  // ... and the fragments of column 1 under its tail
        SP_STAMP(2);
        sp_wait_lds();
        sp_bump_prio(sCR + 1, 8u * (unsigned)k, lane);
        SP_STAMP(3);
        mma_col(buf, 1, true, ltarget);  // ... and of column 2
        SP_STAMP(4);
        sp_wait_lds();
        sp_bump_prio(sCR + 2, 8u * (unsigned)k, lane);
        SP_STAMP(5);
        mma_col(buf, 2, false, 0u);
        sp_wait_lds();  // the last window fragment has been read: the buffer may be refilled (for step k + 2)
        if (lane == 0) sp_bump(sWR + (k & 1));
        SP_STAMP(6);
      }
#ifdef DRS_SP_TIMELINE
      if (c == nck - 1 && (g.debug & 64)) continue;  // timing experiment: no epilogue
#endif
      if (c == nck - 1) {
        // opaque copies of the lane coordinates: everything the epilogue derives from them is computed here, not hoisted
        // out of the step loop (where it would be spilled and reloaded between the stores)
        int lr_e = lr, kg_e = kg;
        asm volatile("" : "+v"(lr_e), "+v"(kg_e));
#include "conv_sp_item_epilogue.inc"
      }
    }
  }
#ifdef DRS_SP_TIMELINE
  if (blockIdx.x == 0 && (wid == 0 || wid == 4 || wid == 8) && lane == 0) {
    for (int i = 0; i < 8; ++i) drs_sp_tl[(wid == 0 ? 0 : (wid == 4 ? 32 : 16)) + i] = tl[i];
    drs_sp_tl[wid == 0 ? 8 : (wid == 4 ? 40 : 24)] = S;
    if (wid == 0) drs_sp_tl[9] = __builtin_amdgcn_s_memtime() - tl_begin;
  }

#endif
}

bool sp_std3x3(const TapConv& d) {
  if (d.mode != 0 || d.ntaps != 9 || d.wtaps_total != 9 || d.in_stride != 1 || d.out_scale != 1) return false;
  for (int i = 0; i < 9; ++i)
    if (d.dy[i] != i / 3 - 1 || d.dx[i] != i % 3 - 1 || d.wtap[i] != i) return false;
  return true;
}

template <bool HAS2, int BNB, bool FUSE = false, bool DUAL = false, bool F32OUT = false>
int sp_launch(const TapConv& d, const MfmaGeom& g, hipStream_t s) {
  auto kern = tapconv_sp_kernel<HAS2, BNB, FUSE, DUAL, F32OUT>;
  constexpr size_t kLds = SpGeom<BNB>::LDS;
  static_assert(kLds <= 160 * 1024, "LDS budget");
  int num_cu = 0;
  {
    const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(kern), 160 * 1024, &num_cu);
    if (rc) return rc;
  }
  const long long nitems = (long long)d.N * g.tiles_x * g.tiles_y * (DUAL ? 1 : d.Cout / BNB);
  long long blocks = num_cu;  // one 12-wave block per CU
#ifdef DRS_SP_TIMELINE
  if (getenv("DRS_SP_MAXBLOCKS")) blocks = atoi(getenv("DRS_SP_MAXBLOCKS"));  // experiment: fewer CUs
#endif
  if (blocks > nitems) blocks = nitems;
  blocks = (blocks + 7) / 8 * 8;
  DRS_LAUNCH(kern, dim3((unsigned)blocks), dim3(768), kLds, s, d, g);
  DRS_CHECK_HIP(hipGetLastError());
#ifdef DRS_SP_TIMELINE
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
    DRS_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(drs_sp_tl), sizeof(h)));
    const double sc = h[8] ? 1.0 / (double)h[8] : 0.0;
    fprintf(stderr, "sp kernel Cin=%d Cout=%d TH=%d in2=%d BNB=%d fuse=%d dual=%d: %.1f us, %llu steps/block, wave0 alive %llu ticks = %.2f GHz, %.0f ticks/step\n",
            d.Cin, d.Cout, d.TH, d.in2 ? d.Cin2 : 0, BNB, (int)FUSE, (int)DUAL, ms * 1e3, h[8], h[9], h[9] / (ms * 1e6), h[9] * sc);
    fprintf(stderr, "   C0: epi>Y %.0f Y %.0f rd0 %.0f col0 %.0f L1+rd1 %.0f col1 %.0f L2+rd2 %.0f col2 %.0f\n", h[7] * sc, h[0] * sc, h[1] * sc,
            h[2] * sc, h[3] * sc, h[4] * sc, h[5] * sc, h[6] * sc);
    fprintf(stderr, "   C4: epi>Y %.0f Y %.0f rd0 %.0f col0 %.0f L1+rd1 %.0f col1 %.0f L2+rd2 %.0f col2 %.0f\n", h[39] * sc, h[32] * sc, h[33] * sc,
            h[34] * sc, h[35] * sc, h[36] * sc, h[37] * sc, h[38] * sc);
    fprintf(stderr, "   M : loads %.0f CR0poll %.0f col0 %.0f WR+win %.0f CR1poll %.0f col1+CR2poll %.0f col2 %.0f\n", (h[23] + h[16]) * sc,
            h[17] * sc, h[18] * sc, h[19] * sc, h[20] * sc, h[21] * sc, h[22] * sc);
  }
#endif
  return DRS_OK;
}

}  // namespace

// Eligibility: 3x3 stride 1 on 16-row patches, SP-format input(s), SP-format output (or the fused fp32 projection).
// Layers it does not take (8-row patches, gates, residuals ...) run on the lock-step kernel (tapconv_mfma_kernel<.., SP>).
bool drs_tapconv_sp_supported(const TapConv& d, int impl) {
  if (impl != DRS_IMPL_MFMA_BF16X3) return false;
  if (!d.in || !d.in_sp || !sp_std3x3(d) || !d.zero_line) return false;
  if (d.gate || d.in_add || d.res || d.sigmoid || d.out_nchw || d.TH <= 8) return false;
  if ((d.in_co & 31) || !(d.in_cs == 16 || (d.in_cs & 31) == 0)) return false;
  if (!(d.Cin % 32 == 0 || (d.Cin == 16 && d.in_cs == 16))) return false;
  if (d.dual)
    return d.Cout == 32 && !d.in2 && !d.fuse_out && !d.out2 && d.bias && d.out && d.out_sp && !(d.out_co & 31) && !(d.out_cs & 31);
  if (d.Cout % 32 != 0) return false;
  if (d.in2 && (!d.in2_sp || !d.w2 || (d.in2_co & 31) || !(d.in2_cs == 16 || (d.in2_cs & 31) == 0) ||
                !(d.Cin2 % 32 == 0 || (d.Cin2 == 16 && d.in2_cs == 16)) || d.H2 != d.TH || d.W2 != d.TW))
    return false;
  if (d.fuse_out) return d.Cout == 32 && !d.in2 && !d.out2 && !d.out_sp && d.fuse_dim <= 4 && !d.post_add && !d.relu_pre && !d.relu_post;
  if ((!d.out && !d.out2) || !d.out_sp || (d.out_co & 31) || (d.out_cs & 31)) return false;  // (out2 alone: a producer whose consumers only read x + temb)
  if (d.out2 && (!d.post2 || (d.out2_co & 31) || (d.out2_cs & 31) || (d.post2_cs & 3))) return false;
  if (d.post_add && (d.post_cs & 3)) return false;
  return true;
}

// SP-format input, fp32 channels-last output (general epilogue: bias, ReLUs, per-image add, fp32 residual): 3x3 stride 1 on
// 16-row patches, no second input / second output / fused projection.
bool drs_tapconv_sp_f32out_supported(const TapConv& d, int impl) {
  if (impl != DRS_IMPL_MFMA_BF16X3) return false;
  if (!d.in || !d.in_sp || !sp_std3x3(d) || !d.zero_line || d.TH <= 8) return false;
  if (d.gate || d.in_add || d.sigmoid || d.out_nchw || d.dual || d.in2 || d.fuse_out || d.out2 || d.out_sp || d.res_sp) return false;
  if ((d.in_co & 31) || (d.in_cs & 31) || d.Cin % 32 != 0 || d.Cout % 32 != 0) return false;
  if (!d.out || (d.out_cs & 3) || (d.out_co & 3)) return false;
  if (d.res && ((d.res_cs & 3) || (d.res_co & 3))) return false;
  if (d.post_add && (d.post_cs & 3)) return false;
  return true;
}

int drs_launch_tapconv_sp(const TapConv& d, const MfmaGeom& g, hipStream_t s) {
  DRS_REQUIRE(g.IH == 18 && g.IW == 18, DRS_ERR_SHAPE, "tapconv_sp: geometry");
  if (!d.out_sp && !d.fuse_out && !d.dual)
    return d.Cout % 64 == 0 ? sp_launch<false, 64, false, false, true>(d, g, s) : sp_launch<false, 32, false, false, true>(d, g, s);
  if (d.dual) return sp_launch<false, 64, false, true>(d, g, s);
  if (d.fuse_out) return sp_launch<false, 32, true>(d, g, s);
  // (32-channel groups on the levels with ONE 64-channel item per CU - twice the items, so that an item's stores could meet
  //  the next item's first step - measured in round 4: bottleneck conv2 65 -> 85 us, ups.0.conv 54 -> 64 us.  The window is
  //  then fetched twice per patch and a step has half the MFMAs over the same mover work.)
  if (d.Cout % 64 == 0) {
    // fewer 64-channel items than half the CUs (small batches; the 16 x 16 level of configs[3]: 128 items): 32-channel groups
    // double the items - every CU then has one, and a step moves 78 KB instead of 113.  (With an item per CU or more the
    // 64-channel groups win: see the note above.)
    int num_cu = 0;
    {
      const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(tapconv_sp_kernel<false, 64, false, false, false>), 160 * 1024, &num_cu);
      if (rc) return rc;
    }
    const long long items64 = (long long)d.N * g.tiles_x * g.tiles_y * (d.Cout / 64);
    if (2 * items64 <= num_cu) return d.in2 ? sp_launch<true, 32>(d, g, s) : sp_launch<false, 32>(d, g, s);
    return d.in2 ? sp_launch<true, 64>(d, g, s) : sp_launch<false, 64>(d, g, s);
  }
  return d.in2 ? sp_launch<true, 32>(d, g, s) : sp_launch<false, 32>(d, g, s);
}
