// Wave-specialised 3x3 stride-1 tap-convolution over SP-format activations, ONE consumer wave per SIMD: the successor of
// tapconv_sp_kernel (conv_mfma_sp.hip; same LDS images, same movers, same counter protocol, same epilogues).
//
// What the phase timeline of that kernel showed (tools/build_tl.sh): inside a kernel column its two consumer waves per SIMD
// keep the matrix pipe 96 % busy, but everything BETWEEN columns - draining the LDS queue, the release with its returned
// rank, the poll of the next column's counter, the first fragment reads - is exposed, because both waves of a SIMD reach
// those points together: 2.4 k of 11 k ticks per step (6.9 k = the MFMAs back to back).  tools/micro/cons_loop.hip: the
// fragment reads + MFMAs alone run at 92 % of the MFMA rate in either structure.  So here a SIMD has one consumer wave
// with twice the register tile (8 rows x 32 channels; 4 rows for the 32-channel / dual flavours) and nothing is ever
// waited for where it is issued:
//   * window fragments run PF rows ahead of their MFMAs through a register ring, across column and step boundaries;
//   * the weight fragments of column j + 1 replace those of column j row by row under its tail (as before);
//   * a counter is PEEKED two row groups before it is needed (plain LDS load, no wait) and only checked then;
//   * releases are relaxed LDS atomics without return: the LDS executes a wave's operations in order, so an add issued
//     after the fragment reads is performed after them - no lgkmcnt(0) drain anywhere in the steady state.
// Fragment reads per MFMA drop from 0.33 to 0.22 on the way (8-row tile).  Block = 4 consumer + 4 mover waves (512 threads,
// 256 registers per lane).
#include <stdio.h>
#include <stdlib.h>

#include "conv_epilogue.h"
#include "mfma_policy.h"

namespace {


__device__ __forceinline__ void sp_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void sp_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// wait until at most n of this wave's vector-memory operations are outstanding (n is wave-uniform)
__device__ __forceinline__ void sp_wait_vm(int n) {
#define DRS_SP_CASE(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break;
  switch (n) {
    DRS_SP_CASE(1) DRS_SP_CASE(2) DRS_SP_CASE(3) DRS_SP_CASE(4) DRS_SP_CASE(5) DRS_SP_CASE(6) DRS_SP_CASE(7)
    DRS_SP_CASE(8) DRS_SP_CASE(9) DRS_SP_CASE(10) DRS_SP_CASE(11) DRS_SP_CASE(12) DRS_SP_CASE(13) DRS_SP_CASE(14)
    DRS_SP_CASE(15) DRS_SP_CASE(16) DRS_SP_CASE(17) DRS_SP_CASE(18) DRS_SP_CASE(19) DRS_SP_CASE(20) DRS_SP_CASE(21)
    DRS_SP_CASE(22) DRS_SP_CASE(23) DRS_SP_CASE(24) DRS_SP_CASE(25) DRS_SP_CASE(26) DRS_SP_CASE(27) DRS_SP_CASE(28)
    DRS_SP_CASE(29) DRS_SP_CASE(30) DRS_SP_CASE(31)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef DRS_SP_CASE
}

#ifdef DRS_SP_TIMELINE
__device__ unsigned long long drs_sp1_tl[48];
#define SP_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl[i] += t_ - tl_last; tl_last = t_; } while (0)
#else
#define SP_STAMP(i) do { } while (0)
#endif

typedef __attribute__((address_space(1))) void* sp_gptr;
typedef __attribute__((address_space(3))) void* sp_lptr;
typedef __attribute__((address_space(3))) unsigned* sp_flag_ptr;

// spin until the LDS counter *f reaches `target`.  A protocol error must never leave waves spinning on the GPU: after
// ~2^22 polls (about a second) the wave traps and the launch fails loudly.
__device__ __forceinline__ void sp_poll(sp_flag_ptr f, unsigned target) {
  unsigned spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > (1u << 22)) __builtin_trap();
  }
}
__device__ __forceinline__ void sp_bump(sp_flag_ptr f) {
  __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// relaxed forms for the consumers (see the header): peek = issue the counter load, no wait; release = add without return
#ifdef DRS_SP_TIMELINE
#define SP_FREE sp_free_run    // timing experiments (DRS_DEBUG_FLAGS & 16): counters ignored, results wrong
#define SP_MUTE sp_mute        // ... & 32: the consumers do not even touch the counters
#else
#define SP_FREE false
#define SP_MUTE false
#endif
__device__ __forceinline__ unsigned sp_peek(sp_flag_ptr f) {
  return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void sp_check(sp_flag_ptr f, unsigned peeked, unsigned target) {
  if ((unsigned)__builtin_amdgcn_readfirstlane((int)peeked) < target) sp_poll(f, target);
  asm volatile("" ::: "memory");  // the fragment reads it guards stay behind it
}
__device__ __forceinline__ void sp_release(sp_flag_ptr f, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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

template <bool HAS2, int BNB_, bool FUSE, bool DUAL>
__global__ __launch_bounds__(512, 1) void tapconv_sp1_kernel(TapConv d, MfmaGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  using G = SpGeom<BNB_>;
  constexpr int KC = 32, IW = G::IW, NBLK = G::NBLK, WBUF = G::WBUF, W_IMAGE = G::W_IMAGE, BNB = G::BNB;
  constexpr int NT = DUAL ? 4 : 2, BN = 32, TH = 16, TW = 16;
  constexpr int NCONS = 4;                                // consumer waves (one per SIMD); waves 4..7 are the movers
  constexpr int NRW = (BNB == 64 && !DUAL) ? 2 : 4;       // row-waves per channel group
  constexpr int RPW = TH / NRW;                            // patch rows per consumer wave: 8 or 4
  char* sWin = smem;                             // [buffer 2][window pixel][rotated operand slot 8] x 16 bytes
  char* sW = smem + 2 * WBUF;                    // [image][kx(3)][ky(3)][k-group(4)][BNB] operand slots
  // ten monotonic counters, no barrier inside the step loop (the waves of a SIMD drift to complementary phases: one
  // multiplies while its partner waits for fragments or writes its tile out):
  sp_flag_ptr sCR = (sp_flag_ptr)(sW + 2 * W_IMAGE);  // CR[3]: consumer waves that hold weight column j of their step in registers
  sp_flag_ptr sCL = sCR + 3;                          // CL[3]: mover waves whose part of weight column j has landed
  sp_flag_ptr sWL = sCR + 6;                          // WL[2]: mover waves whose part of window buffer b has landed
  sp_flag_ptr sWR = sCR + 8;                          // WR[2]: consumer waves that have finished reading window buffer b
  float* sEpi = reinterpret_cast<float*>(sW + 2 * W_IMAGE + 64);  // per-item epilogue constants (conv_sp_movers.inc)
  constexpr int EC = DUAL ? 64 : BNB;

#ifdef DRS_SP_TIMELINE
  const bool sp_free_run = (g.debug & 16) != 0, sp_mute = (g.debug & 32) != 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..3 consumers, 4..7 movers
  const bool mover = wid >= NCONS;
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
  int c = -1, ord = -1, n = 0, ty0 = 0, tx0 = 0, n0 = 0;  // current step: chunk, item ordinal, item coordinates

  if (mover) {
#define SP_NCONS 4
#define SP_MOVER_PRIO 0  // (raised priority, what the two-consumer kernel needs, costs this one 0.9 k ticks per step)
#include "conv_sp_movers.inc"
#undef SP_NCONS
#undef SP_MOVER_PRIO
  } else {
    // ===================== consumers =====================
    const int rw = NRW == 2 ? (wid & 1) : (wid & 3);  // row-wave: rows [rw*RPW, rw*RPW + RPW) of the patch
    const int ng = NRW == 2 ? (wid >> 1) : 0;         // channel group: channels [ng*BN, ng*BN + BN) of the block's BNB
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
    constexpr int NWR = RPW + 2;             // window rows of a column
    constexpr int NB = RPW == 8 ? 5 : 3;     // fragment ring (NWR % NB == 0: the ring phase is the same in every column)
    constexpr int PF = RPW == 8 ? 3 : 2;     // rows the window fragments run ahead of their MFMAs
    static_assert(NWR % NB == 0 && PF < NB && PF + 2 <= NWR - 1, "ring geometry");
    f32x4 acc[RPW][NT];
    typename P::Frag wf[3][NT], af[NB];
    auto w_frag = [&](int col, int ky, int t) __attribute__((always_inline)) {
      return P::load(wbase, W_IMAGE, (size_t)(((col * 3 + ky) * 4 * BNB) + t * 16) * 16);
    };
    auto win_frag = [&](const char* buf, int q) __attribute__((always_inline)) {  // q: compile-time window offset
      return typename P::Frag{*reinterpret_cast<const bf16x8*>(buf + tab_hi[q & 7] + q * 128),
                              *reinterpret_cast<const bf16x8*>(buf + tab_lo[q & 7] + q * 128)};
    };
    bool primed = false;  // column 0's weight fragments and the first PF window rows of this step are already in registers
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
      const char* buf = sWin + (k & 1) * WBUF;
      const char* nbuf = sWin + ((k + 1) & 1) * WBUF;
      const unsigned ltarget = 4u * (unsigned)(k + 1);  // four movers per column and step
      SP_STAMP(7);
      if (second) {
        // second input (1x1 shortcut): one tap at the window origin, columns 1 and 2 are empty.  Short and rare: not pipelined.
        if (!SP_FREE) sp_poll(sWL + (k & 1), 4u * (unsigned)((k >> 1) + 1));
        if (!SP_FREE) sp_poll(sCL, ltarget);
#pragma unroll
        for (int t = 0; t < NT; ++t) wf[0][t] = w_frag(0, 0, t);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          const typename P::Frag a = win_frag(buf, r * IW);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][t] = P::mma(wf[0][t], a, acc[r][t]);
        }
        if (!SP_MUTE) sp_release(sCR, lane);
        if (!SP_MUTE) sp_release(sCR + 1, lane);
        if (!SP_MUTE) sp_release(sCR + 2, lane);
        if (!SP_MUTE) sp_release(sWR + (k & 1), lane);
        primed = false;
      } else {
        // is step k + 1 one this step can prime (a 3x3 step of this block)?
        const int c1 = c + 1 == nck ? 0 : c + 1;
        const bool nreg = k + 1 < S && !(HAS2 && c1 >= g.nchunks);
        if (!primed) {
          if (!SP_FREE) sp_poll(sWL + (k & 1), 4u * (unsigned)((k >> 1) + 1));  // window k is in its buffer
          if (!SP_FREE) sp_poll(sCL, ltarget);                                   // ... and weight column 0 of k
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[ky][t] = w_frag(0, ky, t);
#pragma unroll
          for (int i = 0; i < PF; ++i) af[i] = win_frag(buf, i * IW);
        }
        SP_STAMP(0);
        unsigned peek_w = 0, peek_c = 0;
#pragma unroll
        for (int col = 0; col < 3; ++col) {
#pragma unroll
          for (int wr = 0; wr < NWR; ++wr) {
            // ---- (a) window row PF groups ahead: this column, the next one, or column 0 of the next step ----
            {
              const int pr = wr + PF;
              if (pr < NWR) {
                af[pr % NB] = win_frag(buf, pr * IW + col);
              } else if (col < 2) {
                af[pr % NB] = win_frag(buf, (pr - NWR) * IW + col + 1);
              } else if (nreg) {
                if (pr == NWR) if (!SP_FREE) sp_check(sWL + ((k + 1) & 1), peek_w, 4u * (unsigned)(((k + 1) >> 1) + 1));
                af[pr % NB] = win_frag(nbuf, (pr - NWR) * IW);
              }
              // the last fragment read of this window buffer has been issued: the buffer may be refilled (for step k + 2)
              if (col == 2 && pr == NWR - 1) if (!SP_MUTE) sp_release(sWR + (k & 1), lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- (b) MFMAs of this window row: term-major over its (up to) 3 * NT accumulators ----
            {
              const typename P::Frag a = af[wr % NB];
#ifdef DRS_SP_TIMELINE
              if (!(g.debug & 4))
#endif
#pragma unroll
              for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                  const int r = wr - ky;
                  if (r >= 0 && r < RPW) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                      acc[r][t] = term == 0   ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][t].lo, a.hi, acc[r][t], 0, 0, 0)
                                  : term == 1 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][t].hi, a.lo, acc[r][t], 0, 0, 0)
                                              : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][t].hi, a.hi, acc[r][t], 0, 0, 0);
                  }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- (c) counters and the next column's weight fragments ----
            // all three kernel rows of this column are in registers once row 2 has used wf[2]: release its ring slot
            if (wr == 2) if (!SP_MUTE) sp_release(sCR + col, lane);
            if (col < 2 || nreg) {
              sp_flag_ptr ncl = sCL + (col == 2 ? 0 : col + 1);
              const unsigned ntarget = col == 2 ? ltarget + 4u : ltarget;
              if (wr == RPW - 3) peek_c = SP_MUTE ? 0u : sp_peek(ncl);
              if (wr == RPW - 1) if (!SP_FREE) sp_check(ncl, peek_c, ntarget);
#pragma unroll
              for (int ky = 0; ky < 3; ++ky)
                if (wr == ky + RPW - 1) {  // row ky + RPW - 1 was the last user of wf[ky]
#pragma unroll
                  for (int t = 0; t < NT; ++t) wf[ky][t] = w_frag(col == 2 ? 0 : col + 1, ky, t);
                }
            }
            if (col == 2 && nreg && wr == NWR - PF - 2) peek_w = SP_MUTE ? 0u : sp_peek(sWL + ((k + 1) & 1));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        primed = nreg;
        SP_STAMP(1);
      }
      if (c == nck - 1) {
        // opaque copies of the lane coordinates: everything the epilogue derives from them is computed here, not hoisted
        // out of the step loop (where it would be spilled and reloaded between the stores)
        int lr_e = lr, kg_e = kg;
        asm volatile("" : "+v"(lr_e), "+v"(kg_e));
        // (this variant keeps the epilogue that loads its constants from memory: with the primed fragment registers of the next
        // item live across it, the LDS-constant form of conv_sp_item_epilogue.inc spills 59 registers here)
        if constexpr (DUAL) {
          // out = relu(main + b_main) + post_add + (skip + b_skip): tiles t (main) and t + 2 (skip) of the same lane;
          // channel of (tile t, register j) in SP order: kg*8 + t*4 + j
          f32x4 comb[RPW][2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const float4 b1 = *reinterpret_cast<const float4*>(d.bias + kg_e * 8 + t * 4);
            const float4 b2 = *reinterpret_cast<const float4*>(d.bias + d.Cout + kg_e * 8 + t * 4);
            const float bm[4] = {b1.x, b1.y, b1.z, b1.w}, bs[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
              for (int j = 0; j < 4; ++j) comb[r][t][j] = fmaxf(acc[r][t][j] + bm[j], 0.f) + (acc[r][t + 2][j] + bs[j]);
          }
          TapConv de = d;
          de.bias = nullptr; de.bias2 = nullptr; de.relu_pre = 0;
          tile_epilogue_sp<RPW, 2, false>(de, comb, n, 0, ty0, tx0, rw, lr_e, kg_e, d.out_oy, d.out_ox);
        } else if constexpr (FUSE)
          fuse_epilogue_mfma<RPW>(d, acc, n, n0 + ng * BN, ty0, tx0, rw, lr_e, kg_e);
        else
          tile_epilogue_sp<RPW, NT, true>(d, acc, n, n0 + ng * BN, ty0, tx0, rw, lr_e, kg_e, d.out_oy, d.out_ox);
        SP_STAMP(2);
      }
    }
  }
#ifdef DRS_SP_TIMELINE
  if (blockIdx.x == 0 && (wid == 0 || wid == NCONS) && lane == 0) {
    for (int i = 0; i < 8; ++i) drs_sp1_tl[(wid == 0 ? 0 : 16) + i] = tl[i];
    drs_sp1_tl[wid == 0 ? 8 : 24] = S;
    if (wid == 0) drs_sp1_tl[9] = __builtin_amdgcn_s_memtime() - tl_begin;
  }
#endif
}

template <bool HAS2, int BNB, bool FUSE = false, bool DUAL = false>
int sp1_launch(const TapConv& d, const MfmaGeom& g, hipStream_t s) {
  auto kern = tapconv_sp1_kernel<HAS2, BNB, FUSE, DUAL>;
  constexpr size_t kLds = SpGeom<BNB>::LDS;
  static_assert(kLds <= 160 * 1024, "LDS budget");
  int num_cu = 0;
  {
    const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(kern), 160 * 1024, &num_cu);
    if (rc) return rc;
  }
  const long long nitems = (long long)d.N * g.tiles_x * g.tiles_y * (DUAL ? 1 : d.Cout / BNB);
  long long blocks = num_cu;  // one 8-wave block per CU
  if (blocks > nitems) blocks = nitems;
  blocks = (blocks + 7) / 8 * 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), kLds, s, d, g);
  DRS_CHECK_HIP(hipGetLastError());
#ifdef DRS_SP_TIMELINE
  {
    unsigned long long h[48];
    hipEvent_t e0, e1;
    float ms = 0.f;
    DRS_CHECK_HIP(hipEventCreate(&e0)); DRS_CHECK_HIP(hipEventCreate(&e1));
    DRS_CHECK_HIP(hipEventRecord(e0, s));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), kLds, s, d, g);  // timed repeat (same result)
    DRS_CHECK_HIP(hipEventRecord(e1, s));
    DRS_CHECK_HIP(hipStreamSynchronize(s));
    DRS_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    DRS_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(drs_sp1_tl), sizeof(h)));
    const double sc = h[8] ? 1.0 / (double)h[8] : 0.0;
    fprintf(stderr, "sp1 kernel Cin=%d Cout=%d TH=%d in2=%d BNB=%d fuse=%d dual=%d: %.1f us, %llu steps/block, wave0 alive %llu ticks = %.2f GHz, %.0f ticks/step\n",
            d.Cin, d.Cout, d.TH, d.in2 ? d.Cin2 : 0, BNB, (int)FUSE, (int)DUAL, ms * 1e3, h[8], h[9], h[9] / (ms * 1e6), h[9] * sc);
    fprintf(stderr, "   C : top %.0f prime %.0f columns %.0f epilogue %.0f\n", h[7] * sc, h[0] * sc, h[1] * sc, h[2] * sc);
    fprintf(stderr, "   M : loads %.0f CR0poll %.0f col0 %.0f WR+win %.0f CR1poll %.0f col1+CR2poll %.0f col2 %.0f\n", (h[23] + h[16]) * sc,
            h[17] * sc, h[18] * sc, h[19] * sc, h[20] * sc, h[21] * sc, h[22] * sc);
  }
#endif
  return DRS_OK;
}

}  // namespace

// Same eligibility as tapconv_sp_kernel (drs_tapconv_sp_supported, conv_mfma_sp.hip).
int drs_launch_tapconv_sp1(const TapConv& d, const MfmaGeom& g, hipStream_t s) {
  DRS_REQUIRE(g.IH == 18 && g.IW == 18, DRS_ERR_SHAPE, "tapconv_sp1: geometry");
  DRS_REQUIRE(!d.dual, DRS_ERR_SHAPE, "tapconv_sp1: the dual flavour stays on tapconv_sp_kernel (registers)");
  if (d.fuse_out) return sp1_launch<false, 32, true>(d, g, s);
  if (d.Cout % 64 == 0) return d.in2 ? sp1_launch<true, 64>(d, g, s) : sp1_launch<false, 64>(d, g, s);
  return d.in2 ? sp1_launch<true, 32>(d, g, s) : sp1_launch<false, 32>(d, g, s);
}
