// Wave-specialised 3x3 kernel, pipelined variant (bf16x3 only, opt-in DRS_WS=2): conv_mfma_ws.hip with the window
// conversion taken off the consumers' critical path.
//   * The raw fp32 window is converted IN PLACE: a 16-byte quad of 4 channels becomes {4 x bf16 hi, 4 x bf16 lo} in its
//     own 16 bytes, so the raw staging area doubles as the operand image and LDS holds TWO window buffers
//     (2 x 43 KB + 72 KB weight ring).  One pad slot per pixel pair (pair stride 272 bytes, laid out through the
//     DMA's per-lane source address) makes the fragment reads - the lane's 32-byte k-group block - conflict-free.
//   * 8 consumer waves (MFMA only) + 8 mover waves (16 waves, 128 registers each); a mover fetches its pieces of window k+1 by LDS-DMA, waits on its
//     OWN vector-memory counter until they have landed and converts exactly those pieces (one quad per lane and piece):
//     no cross-wave hand-off.  While the consumers multiply window k, window k+1 is fetched AND converted.
//   * One block-wide barrier per step; LDS counters F0/F1/F2 (consumer waves that hold weight column 0/1/2 in
//     registers -> its ring slot may be refilled).
#include <stdio.h>
#include <stdlib.h>

#include "conv_epilogue.h"
#include "mfma_policy.h"

namespace {

__device__ __forceinline__ void w3_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void w3_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void w3_wait_vm(int n) {
#define DRS_W3_CASE(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break;
  switch (n) {
    DRS_W3_CASE(1) DRS_W3_CASE(2) DRS_W3_CASE(3) DRS_W3_CASE(4) DRS_W3_CASE(5) DRS_W3_CASE(6) DRS_W3_CASE(7)
    DRS_W3_CASE(8) DRS_W3_CASE(9) DRS_W3_CASE(10) DRS_W3_CASE(11) DRS_W3_CASE(12) DRS_W3_CASE(13) DRS_W3_CASE(14)
    DRS_W3_CASE(15) DRS_W3_CASE(16) DRS_W3_CASE(17) DRS_W3_CASE(18) DRS_W3_CASE(19) DRS_W3_CASE(20) DRS_W3_CASE(21)
    DRS_W3_CASE(22) DRS_W3_CASE(23) DRS_W3_CASE(24) DRS_W3_CASE(25) DRS_W3_CASE(26) DRS_W3_CASE(27) DRS_W3_CASE(28)
    DRS_W3_CASE(29) DRS_W3_CASE(30) DRS_W3_CASE(31) DRS_W3_CASE(32) DRS_W3_CASE(33) DRS_W3_CASE(34) DRS_W3_CASE(35)
    DRS_W3_CASE(36) DRS_W3_CASE(37) DRS_W3_CASE(38) DRS_W3_CASE(39) DRS_W3_CASE(40) DRS_W3_CASE(41) DRS_W3_CASE(42)
    DRS_W3_CASE(43) DRS_W3_CASE(44) DRS_W3_CASE(45) DRS_W3_CASE(46) DRS_W3_CASE(47) DRS_W3_CASE(48)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef DRS_W3_CASE
}

#ifdef DRS_WS_TIMELINE
__device__ unsigned long long drs_w3_tl[64];
#define W3_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl[i] += t_ - tl_last; tl_last = t_; } while (0)
#define W3_DECL unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_amdgcn_s_memtime()
#define W3_DUMP(slot) do { if (blockIdx.x == 0 && lane == 0) { for (int i_ = 0; i_ < 8; ++i_) drs_w3_tl[(slot) * 8 + i_] = tl[i_]; drs_w3_tl[56] = S; } } while (0)
#else
#define W3_STAMP(i) do { } while (0)
#define W3_DECL do { } while (0)
#define W3_DUMP(slot) do { } while (0)
#endif

typedef __attribute__((address_space(1))) void* w3_gptr;
typedef __attribute__((address_space(3))) void* w3_lptr;
typedef __attribute__((address_space(3))) unsigned* w3_flag;

// spin until *f >= target; a protocol error must not hang the GPU: trap after ~2^22 polls (seconds)
__device__ __forceinline__ void w3_poll(w3_flag f, unsigned target) {
  unsigned spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > (1u << 22)) __builtin_trap();
  }
}
__device__ __forceinline__ void w3_bump(w3_flag f, int lane) {
  if (lane == 0) __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// NMOV mover waves; CCONV: the CONSUMERS convert window k+1 (their 5-6 quads each) between the MFMAs of columns 1 and 2
// of step k - the movers are pure LDS-DMA then - instead of every mover converting the pieces it fetched
template <bool HAS2, int NMOV, bool CCONV>
__global__ __launch_bounds__(512 + 64 * NMOV, 1) void tapconv_ws3_kernel(TapConv d, MfmaGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  // window image: 16-byte slots, 17 per pixel PAIR (8 quads of the even pixel, 8 of the odd one, 1 pad): the pair stride
  // of 272 bytes makes the 32-byte k-group blocks of 16 consecutive pixels fall into 16 different bank groups
  constexpr int KC = 32, IW = 18, NPIX = IW * IW, NSLOT = NPIX / 2 * 17, NPIECE = (NSLOT + 63) / 64;
  constexpr int WBUF = NSLOT * 16, BNB = 64, W_IMAGE = 9 * 4 * BNB * 16;
  constexpr int RPW = 4, NT = 2, BN = 32, TH = 16, TW = 16;
  constexpr int NPW = (NPIECE + NMOV - 1) / NMOV;          // window pieces per mover wave (at most)
  char* sWB = smem;                       // two window buffers, raw fp32 -> {hi, lo} in place
  char* sW = smem + 2 * WBUF;             // [image][kx][ky][kgroup][BNB] operand slots
  w3_flag sF = (w3_flag)(sW + 2 * W_IMAGE);  // F0, F1, F2, LW (mover waves whose window pieces have landed)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7 consumers, 8..15 movers
  const int lr = lane & 15, kg = lane >> 4;

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
  auto item_of = [&](int ordinal, int& n_, int& ty0_, int& tx0_, int& n0_) {
    int it = lo_item + ordinal * nb8 + j8;
    n0_ = (it % ngroups) * BNB;
    it /= ngroups;
    tx0_ = (it % g.tiles_x) * TW;
    it /= g.tiles_x;
    ty0_ = (it % g.tiles_y) * TH;
    n_ = it / g.tiles_y;
  };
  // step descriptor of step k+1 from the one of step k
  auto next_step = [&](int c_, int ord_, int& c1, int& ord1, int& n1, int& ty1, int& tx1, int& n01) {
    c1 = c_ + 1; ord1 = ord_;
    if (c1 == nck) {
      c1 = 0; ord1 = ord_ + 1;
      item_of(ord1, n1, ty1, tx1, n01);
    }
  };
  if (tid < 8) __hip_atomic_store(sF + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  w3_wait_lds();
  w3_barrier();

  int c = -1, ord = -1, n = 0, ty0 = 0, tx0 = 0, n0 = 0;

  if (wid >= 8) {
    // ============================== mover waves: DMA + in-place conversion of their own pieces ==============================
    const int pw = wid - 8;
    // the movers are the youngest waves of their SIMDs: without a raised priority the consumers' dense MFMA streams
    // starve their conversion arithmetic (6.2 k instead of ~1.5 k ticks per window)
    if (g.debug & 16) __builtin_amdgcn_s_setprio(3); else if (g.debug & 32) __builtin_amdgcn_s_setprio(1);
    const char* wg = reinterpret_cast<const char*>(d.w);
    const size_t w_chunk = (size_t)9 * 4 * d.Cout * 16;
    int vm_issued = 0, end_win = 0;
    // raw window of a step: pieces pw, pw + 4, ... (1 KB each, full 128-byte lines, k-group blocks swizzled)
    auto issue_win = [&](int buf, int c_, int n_, int ty_, int tx_) {
      const bool second = HAS2 && c_ >= g.nchunks;
      const int cc = second ? c_ - g.nchunks : c_;
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        const int j = pw + NMOV * i;
        if (j < NPIECE) {
          const int s_raw = j * 64 + lane_o, s = min(s_raw, NSLOT - 1);
          const int pr = s / 17, r = s - pr * 17;
          const int p = 2 * pr + (r >= 8 ? 1 : 0), quad = r & 7;  // r == 16: the pad slot (fetches a duplicate)
          const int py = p / IW, px = p - py * IW;
          const float* src;
          if (!second) {
            const int iy = min(max(ty_ - 1 + py, 0), d.H - 1), ix = min(max(tx_ - 1 + px, 0), d.W - 1);
            src = d.in + ((size_t)n_ * d.H * d.W + (size_t)(iy * d.W + ix)) * d.in_cs + d.in_co + cc * KC + quad * 4;
          } else {
            const int iy = min(ty_ + py, d.H2 - 1), ix = min(tx_ + px, d.W2 - 1);
            src = d.in2 + ((size_t)n_ * d.H2 * d.W2 + (size_t)(iy * d.W2 + ix)) * d.in2_cs + d.in2_co + cc * KC + quad * 4;
          }
          // tail of the last piece: lanes beyond the image are switched off, nothing is written past the buffer
          if (s_raw < NSLOT) __builtin_amdgcn_global_load_lds((w3_gptr)src, (w3_lptr)(sWB + buf * WBUF + j * 1024), 16, 0, 0);
          vm_issued += 1;
        }
      }
      end_win = vm_issued;
    };
    // in-place conversion of this wave's own pieces (they have landed: the caller waited on the wave's own counter).
    // Branch-free: per-piece slot offsets / window coordinates are precomputed, halo and tail handling is a mask.
    int cv_off[NPW], cv_yx[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int s = min((pw + NMOV * i) * 64 + lane, NSLOT - 1);
      const int pr = s / 17, r = s - pr * 17;
      const int p = 2 * pr + (r >= 8 ? 1 : 0), py = p / IW;
      cv_off[i] = s * 16;
      cv_yx[i] = (py << 8) | (p - py * IW);
    }
    auto convert_win = [&](int buf, int c_, int ty_, int tx_) {
      const bool second = HAS2 && c_ >= g.nchunks;
      // valid window coordinates: [ylo, yhi) x [xlo, xhi)
      const int ylo = second ? 0 : 1 - ty_, xlo = second ? 0 : 1 - tx_;
      const int yhi = second ? min(TH, d.H2 - ty_) : d.H + 1 - ty_, xhi = second ? min(TW, d.W2 - tx_) : d.W + 1 - tx_;
      char* wb = sWB + buf * WBUF;
      constexpr int NFULL = NPIECE / NMOV;  // pieces every mover wave owns (one more if pw + NMOV * NFULL < NPIECE)
      const bool has_last = pw + NMOV * NFULL < NPIECE;
      f32x4 v[NPW];
#pragma unroll
      for (int i = 0; i < NPW; ++i)
        if (i < NFULL || has_last) v[i] = *reinterpret_cast<const f32x4*>(wb + cv_off[i]);
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        if (i < NFULL || has_last) {
          const int py = cv_yx[i] >> 8, px = cv_yx[i] & 255;
          const bool ok = (unsigned)(py - ylo) < (unsigned)(yhi - ylo) && (unsigned)(px - xlo) < (unsigned)(xhi - xlo);
          typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
          typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
          bf16x4 h, l;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const float x = ok ? v[i][jj] : 0.f;
            h[jj] = (__bf16)x;
            l[jj] = (__bf16)(x - (float)h[jj]);
          }
          *reinterpret_cast<bf16x8v*>(wb + cv_off[i]) = bf16x8v{h[0], h[1], h[2], h[3], l[0], l[1], l[2], l[3]};
        }
      }
    };
    auto issue_col = [&](int col, int c_, int n0_) {
      const bool second = HAS2 && c_ >= g.nchunks;
      if (!second) {
#pragma unroll
        for (int i = 0; i < 24 / NMOV; ++i) {
          const int idx = pw * (24 / NMOV) + i;  // (image, ky, k-group) piece
          const int im = idx / 12, ky = (idx % 12) >> 2, kq = idx & 3;
          const char* src = wg + (size_t)im * g.w_gimage + (size_t)c_ * w_chunk +
                            ((size_t)((ky * 3 + col) * 4 + kq) * d.Cout + n0_ + lane) * 16;
          char* dst = sW + (size_t)im * W_IMAGE + (size_t)((col * 3 + ky) * 4 + kq) * BNB * 16;
          __builtin_amdgcn_global_load_lds((w3_gptr)src, (w3_lptr)dst, 16, 0, 0);
        }
        vm_issued += 24 / NMOV;
      } else if (col == 0) {
        const int cc = c_ - g.nchunks;
#pragma unroll
        for (int i = 0; i < 8 / NMOV; ++i) {
          const int idx = pw * (8 / NMOV) + i;
          const int im = idx >> 2, kq = idx & 3;
          const char* src = reinterpret_cast<const char*>(d.w2) + (size_t)im * g.w2_gimage +
                            ((size_t)(cc * 4 + kq) * d.Cout + n0_ + lane) * 16;
          char* dst = sW + (size_t)im * W_IMAGE + (size_t)kq * BNB * 16;
          __builtin_amdgcn_global_load_lds((w3_gptr)src, (w3_lptr)dst, 16, 0, 0);
        }
        vm_issued += 8 / NMOV;
      }
    };
    {
      int n_, ty_, tx_, n0_;
      item_of(0, n_, ty_, tx_, n0_);
      issue_win(0, 0, n_, ty_, tx_);
      issue_col(0, 0, n0_);
      issue_col(1, 0, n0_);
      issue_col(2, 0, n0_);
      w3_wait_vm(vm_issued - end_win);
      if constexpr (CCONV) w3_bump(sF + 3, lane); else convert_win(0, 0, ty_, tx_);
      w3_wait_vm(0);
      w3_wait_lds();
    }
    W3_DECL;
    for (int k = 0; k < S; ++k) {
      if (++c == nck) c = 0;
      if (c == 0) item_of(++ord, n, ty0, tx0, n0);
      W3_STAMP(0);
      w3_barrier();  // Y(k)
      W3_STAMP(1);
      if (k + 1 < S) {
        int c1, ord1, n1 = n, ty1 = ty0, tx1 = tx0, n01 = n0;
        next_step(c, ord, c1, ord1, n1, ty1, tx1, n01);
        const int buf = (k + 1) & 1;
        issue_win(buf, c1, n1, ty1, tx1);  // the long, memory-bound burst
        W3_STAMP(2);
        w3_poll(sF + 0, 8u * (unsigned)(k + 1));
        issue_col(0, c1, n01);
        w3_wait_vm(vm_issued - end_win);    // this wave's window pieces have landed
        W3_STAMP(3);
        if constexpr (CCONV) w3_bump(sF + 3, lane); else convert_win(buf, c1, ty1, tx1);
        W3_STAMP(4);
        w3_poll(sF + 1, 8u * (unsigned)(k + 1));
        issue_col(1, c1, n01);
        w3_poll(sF + 2, 8u * (unsigned)(k + 1));
        issue_col(2, c1, n01);
        W3_STAMP(5);
        w3_wait_vm(0);
        w3_wait_lds();
        W3_STAMP(6);
      }
    }
    if (wid == 8) W3_DUMP(2);
    return;
  }

  // ============================== consumers ==============================
  const int rw = wid & 3, ng = wid >> 2;
  const int pbase = rw * RPW * IW + lr;                                  // this lane's first window pixel
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
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  // byte offset of this lane's k-group block of window pixel pbase (even offsets) / pbase + 1 (odd offsets)
  const int offA = (pbase >> 1) * 272 + (pbase & 1) * 128 + kg * 32;
  const int offB = ((pbase + 1) >> 1) * 272 + ((pbase + 1) & 1) * 128 + kg * 32;
  auto a_frag = [&](const char* wb, int pofs) {  // window pixel pbase + pofs (compile-time pofs), this lane's k-group
    const char* a = wb + ((pofs & 1) ? offB + ((pofs - 1) >> 1) * 272 : offA + (pofs >> 1) * 272);
    const u32x2 h0 = *reinterpret_cast<const u32x2*>(a), l0 = *reinterpret_cast<const u32x2*>(a + 8);
    const u32x2 h1 = *reinterpret_cast<const u32x2*>(a + 16), l1 = *reinterpret_cast<const u32x2*>(a + 24);
    typename P::Frag f;
    f.hi = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
    f.lo = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
    return f;
  };
  auto mma_col = [&](const char* wb, int col) {
#pragma unroll
    for (int wr = 0; wr < RPW + 2; ++wr) {
      const typename P::Frag af = a_frag(wb, wr * IW + col);
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
  // ---- consumers' share of the in-place conversion (CCONV): slots tid, tid + 512, ... of the next window ----
  constexpr int CVI = (NSLOT + 511) / 512;
  f32x4 cv[CVI];
  auto cv_load = [&](int buf) {
    int t_o = tid;  // opaque: the slot arithmetic is redone here instead of living in registers across the MFMA loop
    asm volatile("" : "+v"(t_o));
#pragma unroll
    for (int i = 0; i < CVI; ++i) {
      const int s = t_o + 512 * i;
      if (i < CVI - 1 || s < NSLOT) cv[i] = *reinterpret_cast<const f32x4*>(sWB + buf * WBUF + s * 16);
    }
  };
  auto cv_store = [&](int buf, int c_, int ty_, int tx_) {
    const bool second = HAS2 && c_ >= g.nchunks;
    const int ylo = second ? 0 : 1 - ty_, xlo = second ? 0 : 1 - tx_;
    const int yhi = second ? min(TH, d.H2 - ty_) : d.H + 1 - ty_, xhi = second ? min(TW, d.W2 - tx_) : d.W + 1 - tx_;
    int t_o = tid;
    asm volatile("" : "+v"(t_o));
#pragma unroll
    for (int i = 0; i < CVI; ++i) {
      const int s = t_o + 512 * i;
      if (i < CVI - 1 || s < NSLOT) {
        const int pr = s / 17, r = s - pr * 17;
        const int p = 2 * pr + (r >= 8 ? 1 : 0), py = p / IW, px = p - py * IW;
        const bool ok = (unsigned)(py - ylo) < (unsigned)(yhi - ylo) && (unsigned)(px - xlo) < (unsigned)(xhi - xlo);
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
        bf16x4 h, l;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float x = ok ? cv[i][jj] : 0.f;
          h[jj] = (__bf16)x;
          l[jj] = (__bf16)(x - (float)h[jj]);
        }
        *reinterpret_cast<bf16x8v*>(sWB + buf * WBUF + s * 16) = bf16x8v{h[0], h[1], h[2], h[3], l[0], l[1], l[2], l[3]};
      }
    }
  };
  if constexpr (CCONV) {  // window 0
    int n_, ty_, tx_, n0_;
    item_of(0, n_, ty_, tx_, n0_);
    w3_poll(sF + 3, (unsigned)NMOV);
    cv_load(0);
    cv_store(0, 0, ty_, tx_);
  }
  W3_DECL;
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
    const char* wb = sWB + (k & 1) * WBUF;
    W3_STAMP(0);
    w3_wait_lds();
    w3_barrier();  // Y(k): window k converted, the three weight columns of step k landed
    W3_STAMP(1);
    read_wf(0);
    w3_wait_lds();
    w3_bump(sF + 0, lane);
    int c1 = 0, ord1 = 0, n1 = n, ty1 = ty0, tx1 = tx0, n01 = n0;
    const bool conv_next = CCONV && k + 1 < S;
    if (conv_next) next_step(c, ord, c1, ord1, n1, ty1, tx1, n01);
    if (second) {
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const typename P::Frag af = a_frag(wb, r * IW);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = P::mma(wf[0][t], af, acc[r][t]);
      }
      w3_wait_lds();
      w3_bump(sF + 1, lane);
      w3_bump(sF + 2, lane);
    } else {
      mma_col(wb, 0);
      read_wf(1);
      W3_STAMP(2);
      w3_wait_lds();
      w3_bump(sF + 1, lane);
      W3_STAMP(3);
      mma_col(wb, 1);
      read_wf(2);
      W3_STAMP(4);
      w3_wait_lds();
      w3_bump(sF + 2, lane);
    }
    if (conv_next) {  // window k+1 has landed by now (the movers' burst started at the barrier): fetch my quads ...
      w3_poll(sF + 3, (unsigned)NMOV * (unsigned)(k + 2));
      cv_load((k + 1) & 1);
    }
    if (!second) mma_col(wb, 2);
    if (conv_next) cv_store((k + 1) & 1, c1, ty1, tx1);  // ... and convert them in the shadow of column 2's MFMAs
    W3_STAMP(5);
    if (c == nck - 1) {
      int lr_e = lr, kg_e = kg;
      asm volatile("" : "+v"(lr_e), "+v"(kg_e));
      tile_epilogue<RPW, NT, false, RPW>(d, acc, n, n0 + ng * BN, ty0, tx0, rw, lr_e, kg_e, d.out_oy, d.out_ox);
    }
    W3_STAMP(6);
  }
  if (wid == 0) W3_DUMP(0);
  if (wid == 4) W3_DUMP(1);
}

template <bool HAS2, int NMOV, bool CCONV>
int ws3_launch(const TapConv& d, const MfmaGeom& g, hipStream_t s) {
  auto kern = tapconv_ws3_kernel<HAS2, NMOV, CCONV>;
  static bool attr_done = false;
  static int num_cu = 0;
  constexpr size_t kLds = 2 * (18 * 18 / 2 * 17 * 16) + 2 * 9 * 4 * 64 * 16 + 64;
  static_assert(kLds <= 160 * 1024, "LDS budget");
  if (!attr_done) {
    DRS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024));
    int dev = 0;
    DRS_CHECK_HIP(hipGetDevice(&dev));
    DRS_CHECK_HIP(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    attr_done = true;
  }
  const long long nitems = (long long)d.N * g.tiles_x * g.tiles_y * (d.Cout / 64);
  long long blocks = num_cu;
  if (blocks > nitems) blocks = nitems;
  blocks = (blocks + 7) / 8 * 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512 + 64 * NMOV), kLds, s, d, g);
  DRS_CHECK_HIP(hipGetLastError());
#ifdef DRS_WS_TIMELINE
  {
    unsigned long long h[64];
    DRS_CHECK_HIP(hipStreamSynchronize(s));
    DRS_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(drs_w3_tl), sizeof(h)));
    const double sc = h[56] ? 1.0 / (double)h[56] : 0.0;
    fprintf(stderr, "ws3 Cin=%d Cout=%d TH=%d in2=%d S=%llu\n", d.Cin, d.Cout, d.TH, d.in2 ? d.Cin2 : 0, h[56]);
    for (int w = 0; w < 2; ++w)
      fprintf(stderr, "   C wave %d: >Y %.0f rd0 %.0f col0+rd1 %.0f bump %.0f col1+rd2 %.0f col2 %.0f epi %.0f\n", w * 4,
              h[w * 8 + 0] * sc, h[w * 8 + 1] * sc, h[w * 8 + 2] * sc, h[w * 8 + 3] * sc, h[w * 8 + 4] * sc, h[w * 8 + 5] * sc, h[w * 8 + 6] * sc);
    fprintf(stderr, "   mover: prev %.0f Y %.0f window issue %.0f F0+col0+land %.0f convert %.0f F1,F2+cols %.0f land %.0f\n", h[16] * sc,
            h[17] * sc, h[18] * sc, h[19] * sc, h[20] * sc, h[21] * sc, h[22] * sc);
  }
#endif
  return DRS_OK;
}

}  // namespace

int drs_launch_tapconv_ws3(const TapConv& d, const MfmaGeom& g, hipStream_t s) {
  DRS_REQUIRE(g.IH == 18 && g.IW == 18, DRS_ERR_SHAPE, "tapconv_ws3: geometry");
  static const int variant = getenv("DRS_WS3") ? atoi(getenv("DRS_WS3")) : 0;
  if (variant == 1)  // 4 pure-DMA movers, the consumers convert window k+1 next to column 2's MFMAs (measured slower)
    return d.in2 ? ws3_launch<true, 4, true>(d, g, s) : ws3_launch<false, 4, true>(d, g, s);
  return d.in2 ? ws3_launch<true, 8, false>(d, g, s) : ws3_launch<false, 8, false>(d, g, s);  // 8 movers convert what they fetch
}
