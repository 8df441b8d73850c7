// LDS-tiled implicit-GEMM tap-convolution on the gfx950 matrix cores.
//
// GEMM view (per launch):  D[cout][pixel] = sum_{tap, ci} W[tap][cout][ci] * X[pixel + tap][ci]
//   M = Cout (A operand = weights), N = output positions (B operand = activations), K = taps x Cin.
// Weights are the MFMA A operand so that the accumulator of a lane holds 4 CONSECUTIVE output channels of one
// pixel (C/D layout: col = lane&15 -> pixel, row = 4*(lane>>4)+reg -> cout): the epilogue reads bias / residual
// and writes the channels-last output as 16-byte vectors, full 128-byte lines per store instruction.
//
// Block = 256 threads = 4 waves; output patch = (4*RPW) rows x 16 columns of logical output positions,
// BN output channels.  Wave w owns rows [w*RPW, (w+1)*RPW) of the patch (one 16-pixel MFMA column block each) and
// all BN channels.  K loop over chunks of KC input channels:
//   stage  : the input window (patch + halo of all taps) of KC channels -> LDS, converted to the policy's operand
//            type; the chunk's weights for all taps and BN channels -> LDS (pre-converted at pack time).  The
//            global loads of chunk c+1 are issued before chunk c is multiplied and live in registers meanwhile;
//   compute: MFMA over the window shifted by each tap - the window is re-read from LDS, never from HBM.
// Three compute schedules share the staging and the epilogue:
//   GENERIC  runtime tap list (1x1, 2x2 s2, 3x3 s2, single ConvTranspose phases);
//   CONV3X3  3x3 stride 1: each window-row fragment is read from LDS once and feeds the (up to 3) output rows it
//            belongs to - half the activation LDS reads of the generic schedule;
//   CONVT    all 4 output phases of ConvTranspose2d(k3, s2, p1, op1) from ONE staged window: 4 accumulator sets,
//            each of the 9 weight taps goes to the phase it belongs to (reference UpConvBlock.transform).
// LDS layout: "slot" = the 16 bytes a lane feeds to one MFMA operand register group (8 x 16-bit or 4 x f32 = the
// lane's K-group).  Activations: [image][kgroup(4)][window pixel] slots, k-groups 2,3 shifted by 128 B; weights:
// [image][tap][kgroup(4)][BN] slots.  16 lanes with consecutive pixels (or channels) read 256 contiguous bytes:
// conflict-free ds_read_b128; the staging lane order makes the ds_write_b128 conflict-free as well.
// The footprint is sized for 2 blocks per CU.
#include <stdlib.h>

#include "conv_epilogue.h"
#include <vector>

#include "mfma_policy.h"

enum { MODE_GENERIC = 0, MODE_CONV3X3 = 1, MODE_CONVT = 2, MODE_CONV3X3_FUSE = 3 };

template <int MODE, int RPW, int NWG>
struct ModeTraits {
  // window slots per thread: generic stride 1 <= 384 px, stride 2 <= 576 px, conv3x3 18x18, convT 9x17
  static constexpr int A_ITERS = (MODE == MODE_CONV3X3 || MODE == MODE_CONV3X3_FUSE)
                                     ? ((4 * RPW + 2) * 18 + 64 * NWG - 1) / (64 * NWG)
                                     : (MODE == MODE_CONVT ? 3 : (RPW == 4 ? 6 : 9)) / NWG;
  static constexpr int NACC = MODE == MODE_CONVT ? 4 : 1;
};

// NWG = 2: 512 threads, two groups of 4 waves share ONE staged input window and each compute their own BN output
// channels (BNB = 64 per block): half the activation loads / conversions / LDS writes per MFMA for layers with
// Cout % 64 == 0, at one block per CU (117 KB of LDS).
// SP (compile-time, so that an instantiation carries ONE staging path and ONE epilogue: the kernel sits at the 256-register
// limit): 0 = fp32 tensors; 1 = the activation operands (in, in2) are in SP format (TapConv::in_sp), fp32 output;
// 2 = operands and output in SP format
template <class P, int BN, int RPW, int MODE, bool HAS2, int NWG, int SP = 0>
__global__ __launch_bounds__(256 * NWG, NWG == 1 ? 2 : 1) void tapconv_mfma_kernel(TapConv d, MfmaGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KC = 4 * P::SLOT_CH;
  constexpr int TH = 4 * RPW, TW = 16, NT = BN / 16;
  constexpr int NTHR = 256 * NWG, BNB = BN * NWG, PPI = 64 * NWG;  // threads, block channels, window pixels per pass
  constexpr int A_ITERS = ModeTraits<MODE, RPW, NWG>::A_ITERS;
  constexpr int NACC = ModeTraits<MODE, RPW, NWG>::NACC;
  constexpr int W_ITERS = (DRS_MAX_TAPS * 4 * BNB + NTHR - 1) / NTHR;
  constexpr int V4 = P::SLOT_CH / 4;  // float4 loads per activation slot
  char* sA = smem;
  char* sW = smem + (size_t)P::IMAGES * g.a_image;
  // tap tables live in LDS: indexing the by-value kernel argument with a runtime tap would go through scratch
  int* sTapOff = reinterpret_cast<int*>(smem + (size_t)P::IMAGES * (g.a_image + g.w_image));  // window slot offset
  int* sTapW = sTapOff + 16;                                                                   // weight tap index

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = NWG == 1 ? (tid >> 6) : ((tid >> 6) & 3);  // row-wave inside its group
  const int ng = NWG == 1 ? 0 : (tid >> 8);                   // channel group of this wave
  const int lr = lane & 15, kg = lane >> 4;
  if (tid < DRS_MAX_TAPS) {
    int dyv = 0, dxv = 0, wt = 0;
#pragma unroll
    for (int i = 0; i < DRS_MAX_TAPS; ++i)
      if (i == tid) { dyv = d.dy[i]; dxv = d.dx[i]; wt = d.wtap[i]; }
    sTapOff[tid] = ((dyv - g.dy_min) * g.IW + (dxv - g.dx_min)) * 16;
    sTapW[tid] = wt;
  }
  __syncthreads();
  // Persistent blocks: the grid is 2 blocks per CU (a multiple of 8) and every block walks over its share of the
  // (patch, channel-group) items.  XCD-aware order: hardware deals consecutive block ids round-robin over the 8 XCDs,
  // so the blocks with equal blockIdx % 8 (one XCD = one private L2) sweep one contiguous eighth of the items, a run
  // of consecutive patches at a time: neighbouring halos and the layer's weights are served by that L2.  The loop
  // below is software-pipelined ACROSS items: the loads of the next item's first chunk are in flight while the
  // current item is multiplied and written out, so HBM reads, MFMA and HBM writes overlap instead of alternating.
  const int ngroups = d.Cout / BNB;
  const int nitems = d.N * g.tiles_y * g.tiles_x * ngroups;
  const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
  const int per = (nitems + 7) >> 3;  // items per XCD range
  const int lo_item = xcd * per, hi_item = min(nitems, lo_item + per);
  const int span = hi_item - lo_item - j8;
  const int my_items = span > 0 ? (span + nb8 - 1) / nb8 : 0;
  const int nck = g.nchunks + (HAS2 ? g.nchunks2 : 0);  // K-chunks per item: main input, then the optional second input
  const int S = my_items * nck;            // steps (item x chunk) of this block
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
  const int npix = g.IH * g.IW;
  const int wslots = d.ntaps * 4 * BNB;

  f32x4 acc[NACC][RPW][NT];

  // ---- per-thread staging descriptors (chunk independent).  Every address is clamped to a legal one and validity is
  //      a bit mask, so the loads below are straight-line code (no exec-masked branches, no serialising waits). ----
  // Lane order inside a group of 16 staging lanes (4 window pixels x 4 k-groups): lanes 0-7 carry k-groups 0 and 2,
  // lanes 8-15 k-groups 1 and 3, so that the 8 lanes of one ds_write_b128 pass hit 8 different 16-byte bank slots
  // (k-groups 2,3 sit 128 B further) while a wave still covers whole 128-byte pixel rows in global memory.
  const int ai = tid & 15;
  const int aq = ((ai & 1) << 1) | (ai >> 3);  // k-group of all of this thread's slots
  const int ap0 = (tid >> 4) * 4 + ((ai & 7) >> 1);
  // window coordinates of slot pixel p: py = p / IW through a 16-bit reciprocal (exact for p < 1024, IW <= 64)
  auto win_yx = [&](int it, int& py, int& px) __attribute__((always_inline)) {
    const int p = min(ap0 + it * PPI, npix - 1);
    py = (p * g.iw_magic) >> 16;
    px = p - py * g.IW;
  };
  int w_goff[W_ITERS];  // byte offset of the slot inside one chunk of one global weight image, channel group 0
#pragma unroll
  for (int it = 0; it < W_ITERS; ++it) {
    const int s = min(tid + it * NTHR, wslots - 1);
    const int nn = s % BNB, q = (s / BNB) & 3, tap = s / (BNB * 4);
    w_goff[it] = ((sTapW[tap] * 4 + q) * d.Cout + nn) * 16;
  }
  const bool has_add = SP == 0 && d.in_add != nullptr;
  const char* wg = reinterpret_cast<const char*>(d.w);
  const size_t w_chunk = (size_t)d.wtaps_total * 4 * d.Cout * 16;
  const int a_qoff = aq * g.a_plane + (aq >> 1) * 128;  // byte offset of this thread's k-group plane

  float4 areg[A_ITERS][V4];
  float4 addreg[V4];
  u32x4 wreg[W_ITERS][P::IMAGES];
  unsigned a_ok = 0;  // bit it: the slot held in areg[it] lies inside the image

  bool held_second = false;  // the registers hold a chunk of the second input
  int held_cc = 0;           // its chunk index within its input
  auto load_step = [&](int k) __attribute__((always_inline)) {  // global -> registers for step k, everything issued back to back
    const int c = k % nck;
    int ln, lty0, ltx0, ln0;
    item_of(k / nck, ln, lty0, ltx0, ln0);
    held_second = HAS2 && c >= g.nchunks;
    held_cc = held_second ? c - g.nchunks : c;
    int a_base[A_ITERS];  // element offset of channel 0 of the (clamped) pixel inside the image
    a_ok = 0;
    if (!held_second) {
      const int iy0 = lty0 * d.in_stride + g.dy_min, ix0 = ltx0 * d.in_stride + g.dx_min;
      const float* in_n = d.in + (size_t)ln * d.H * d.W * d.in_cs;
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        int py, px;
        win_yx(it, py, px);
        const int iy = iy0 + py, ix = ix0 + px;
        const bool ok = (ap0 + it * PPI) < npix && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
        const int iyc = min(max(iy, 0), d.H - 1), ixc = min(max(ix, 0), d.W - 1);
        a_base[it] = (iyc * d.W + ixc) * d.in_cs + d.in_co;
        a_ok |= (ok ? 1u : 0u) << it;
      }
      if constexpr (P::IMAGES == 2 && SP != 0) {
        // SP format: the hi and lo operand slots of this thread's k-group, no conversion
        const int aqc = (held_cc * KC + aq * P::SLOT_CH < d.Cin) ? aq : 0;
        const int half = drs_sp_group_bytes(d.in_cs);
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
          const char* gp = reinterpret_cast<const char*>(in_n + a_base[it]) + held_cc * 128 + aqc * 16;
          areg[it][0] = *reinterpret_cast<const float4*>(gp);
          areg[it][1] = *reinterpret_cast<const float4*>(gp + half);
        }
      } else {
#pragma unroll
      for (int v = 0; v < V4; ++v) {
        const int ch = min(held_cc * KC + aq * P::SLOT_CH + 4 * v, d.Cin - 4);
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) areg[it][v] = *reinterpret_cast<const float4*>(in_n + a_base[it] + ch);
        if (has_add) addreg[v] = *reinterpret_cast<const float4*>(d.in_add + (size_t)ln * d.in_add_cs + ch);
      }
      }
#pragma unroll
      for (int it = 0; it < W_ITERS; ++it)
#pragma unroll
        for (int im = 0; im < P::IMAGES; ++im)
          wreg[it][im] = *reinterpret_cast<const u32x4*>(wg + (size_t)im * g.w_gimage + (size_t)held_cc * w_chunk +
                                                         (size_t)ln0 * 16 + w_goff[it]);
    } else {
      // second input: logical position (ty, tx) of the patch, embedded at the top-left of the LDS window image
      const float* in_n = d.in2 + (size_t)ln * d.H2 * d.W2 * d.in2_cs;
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        int py, px;
        win_yx(it, py, px);
        const int iy = lty0 + py, ix = ltx0 + px;
        const bool ok = (ap0 + it * PPI) < npix && py < TH && px < TW && iy < d.H2 && ix < d.W2;
        const int iyc = min(iy, d.H2 - 1), ixc = min(ix, d.W2 - 1);
        a_base[it] = (iyc * d.W2 + ixc) * d.in2_cs + d.in2_co;
        a_ok |= (ok ? 1u : 0u) << it;
      }
      if constexpr (P::IMAGES == 2 && SP != 0) {
        const int aqc = (held_cc * KC + aq * P::SLOT_CH < d.Cin2) ? aq : 0;
        const int half = drs_sp_group_bytes(d.in2_cs);
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
          const char* gp = reinterpret_cast<const char*>(in_n + a_base[it]) + held_cc * 128 + aqc * 16;
          areg[it][0] = *reinterpret_cast<const float4*>(gp);
          areg[it][1] = *reinterpret_cast<const float4*>(gp + half);
        }
      } else {
#pragma unroll
      for (int v = 0; v < V4; ++v) {
        const int ch = min(held_cc * KC + aq * P::SLOT_CH + 4 * v, d.Cin2 - 4);
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) areg[it][v] = *reinterpret_cast<const float4*>(in_n + a_base[it] + ch);
      }
      }
      {  // 1 tap: 4 * BNB weight slots, all in the first staging pass
        const int s2 = min(tid, 4 * BNB - 1);
        const size_t off = ((size_t)held_cc * 4 * d.Cout + (size_t)(s2 / BNB) * d.Cout + ln0 + (s2 % BNB)) * 16;
#pragma unroll
        for (int im = 0; im < P::IMAGES; ++im)
          wreg[0][im] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(d.w2) + (size_t)im * g.w2_gimage + off);
      }
    }
  };
  auto store_chunk = [&](int c) __attribute__((always_inline)) {  // registers -> LDS (operand conversion happens here)
    if constexpr (P::IMAGES == 2 && SP != 0) {
      // SP format: the registers hold the operand slots themselves
      const bool ch_ok = held_cc * KC + aq * P::SLOT_CH < (held_second ? d.Cin2 : d.Cin);
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        const int p = ap0 + it * PPI;
        const bool ok = ch_ok && ((a_ok >> it) & 1u);
        float4 h = areg[it][0], l = areg[it][1];  // (by value: a conditional on the array elements would take their address)
        if (!ok) { h = z; l = z; }
        if (p < npix) {
          *reinterpret_cast<float4*>(sA + (size_t)a_qoff + (size_t)p * 16) = h;
          *reinterpret_cast<float4*>(sA + (size_t)g.a_image + (size_t)a_qoff + (size_t)p * 16) = l;
        }
      }
    } else {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      const int p = ap0 + it * PPI;
      const bool pix_ok = (a_ok >> it) & 1u;
      float x[P::SLOT_CH];
#pragma unroll
      for (int v = 0; v < V4; ++v) {
        const bool ok = pix_ok && (held_cc * KC + aq * P::SLOT_CH + 4 * v < (held_second ? d.Cin2 : d.Cin));
        float4 a = areg[it][v];
        if (has_add && !held_second) {  // per-(n, ci) input add applies to in-image pixels only (zero padding stays zero)
          a.x += addreg[v].x; a.y += addreg[v].y; a.z += addreg[v].z; a.w += addreg[v].w;
        }
        x[4 * v] = ok ? a.x : 0.f; x[4 * v + 1] = ok ? a.y : 0.f; x[4 * v + 2] = ok ? a.z : 0.f; x[4 * v + 3] = ok ? a.w : 0.f;
      }
      if (p < npix) P::cvt_store(sA, g.a_image, (size_t)a_qoff + (size_t)p * 16, x);
    }
    }
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it)
      if (tid + it * NTHR < (held_second ? 4 * BNB : wslots)) {
#pragma unroll
        for (int im = 0; im < P::IMAGES; ++im)
          *reinterpret_cast<u32x4*>(sW + (size_t)im * g.w_image + (size_t)(tid + it * NTHR) * 16) = wreg[it][im];
      }
  };
  const int kg_off = kg * g.a_plane + (kg >> 1) * 128;  // fragment reads: this lane's k-group plane
  auto a_frag = [&](int wrow, int wcol) {               // window pixel (wrow, wcol) of this lane's k-group
    return P::load(sA, g.a_image, (size_t)kg_off + (size_t)(wrow * g.IW + wcol) * 16);
  };
  auto w_frag = [&](int tap, int t) {
    return P::load(sW, g.w_image, ((size_t)(tap * 4 + kg) * BNB + (ng * NT + t) * 16 + lr) * 16);
  };

  if (!(g.debug & 4)) load_step(0);
  int n = 0, ty0 = 0, tx0 = 0, n0 = 0;
  for (int k = 0; k < S; ++k) {
    const int c = k % nck;
    if (c == 0) {
      item_of(k / nck, n, ty0, tx0, n0);
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[a][r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (k) __syncthreads();  // everyone finished reading the previous step's LDS image
    if (!(g.debug & 2)) store_chunk(c);
    __syncthreads();
    if (k + 1 < S && !(g.debug & 4)) load_step(k + 1);  // next step's loads fly while this one is multiplied / written
    if (!(g.debug & 1)) {
    if (HAS2 && c >= g.nchunks) {  // second input: one tap, stride 1, window origin
      typename P::Frag wf[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[t] = w_frag(0, t);
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const typename P::Frag af = a_frag(wave * RPW + r, lr);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[0][r][t] = P::mma(wf[t], af, acc[0][r][t]);
      }
    } else
    if constexpr (MODE == MODE_GENERIC || MODE == MODE_CONV3X3_FUSE) {
      for (int tap = 0; tap < d.ntaps; ++tap) {
        typename P::Frag wf[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) wf[t] = w_frag(tap, t);
        const int toff = sTapOff[tap];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          const int py = (wave * RPW + r) * d.in_stride;
          const int px = lr * d.in_stride;
          const typename P::Frag af = P::load(sA, g.a_image, (size_t)kg_off + (size_t)(py * g.IW + px) * 16 + toff);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[0][r][t] = P::mma(wf[t], af, acc[0][r][t]);
        }
      }
    } else if constexpr (MODE == MODE_CONV3X3) {
      // window row wr (0 .. RPW+1) of this wave feeds output row r = wr - ky with the weights of kernel row ky
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        typename P::Frag wf[3][NT];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int t = 0; t < NT; ++t) wf[ky][t] = w_frag(ky * 3 + kx, t);
#pragma unroll
        for (int wr = 0; wr < RPW + 2; ++wr) {
          const typename P::Frag af = a_frag(wave * RPW + wr, lr + kx);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int r = wr - ky;
            if (r >= 0 && r < RPW) {
#pragma unroll
              for (int t = 0; t < NT; ++t) acc[0][r][t] = P::mma(wf[ky][t], af, acc[0][r][t]);
            }
          }
        }
      }
    } else {  // MODE_CONVT: out[2*iy - 1 + ky][2*ix - 1 + kx] += in[iy][ix] * w[ky][kx]
      // output phase (py, px): py = 0 takes ky = 1 from input row t; py = 1 takes ky = 0 from row t+1 and ky = 2 from row t
#pragma unroll
      for (int dyi = 0; dyi < 2; ++dyi)
#pragma unroll
        for (int dxi = 0; dxi < 2; ++dxi) {
          typename P::Frag af[RPW];
#pragma unroll
          for (int r = 0; r < RPW; ++r) af[r] = a_frag(wave * RPW + r + dyi, lr + dxi);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              if (((ky == 0) ? 1 : 0) != dyi || ((kx == 0) ? 1 : 0) != dxi) continue;
              const int ph = ((ky == 1) ? 0 : 1) * 2 + ((kx == 1) ? 0 : 1);
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                const typename P::Frag wf = w_frag(ky * 3 + kx, t);
#pragma unroll
                for (int r = 0; r < RPW; ++r) acc[ph][r][t] = P::mma(wf, af[r], acc[ph][r][t]);
              }
            }
        }
    }
    }  // !(debug & 1)

    if (c == nck - 1 && !(g.debug & 8)) {  // item complete
      if constexpr (MODE == MODE_CONVT) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
          if constexpr (SP == 2) tile_epilogue_sp<RPW, NT, false, true>(d, acc[ph], n, n0 + ng * BN, ty0, tx0, wave, lr, kg, ph >> 1, ph & 1);
          else tile_epilogue<RPW, NT, false>(d, acc[ph], n, n0 + ng * BN, ty0, tx0, wave, lr, kg, ph >> 1, ph & 1);
        }
      } else {
        if constexpr (MODE == MODE_CONV3X3_FUSE)
          fuse_epilogue<RPW, NT>(d, acc[0], n, n0 + ng * BN, ty0, tx0, wave, lr, kg);
        else if constexpr (SP == 2)
          tile_epilogue_sp<RPW, NT>(d, acc[0], n, n0 + ng * BN, ty0, tx0, wave, lr, kg, d.out_oy, d.out_ox);
        else
          tile_epilogue<RPW, NT, false>(d, acc[0], n, n0 + ng * BN, ty0, tx0, wave, lr, kg, d.out_oy, d.out_ox);
      }
    }
  }
}

// ---- host side --------------------------------------------------------------------------------------------------
static constexpr int kLdsLimit = 160 * 1024;

static int slot_ch(int impl) { return impl == DRS_IMPL_MFMA_F32 ? 4 : 8; }
static int images(int impl) { return impl == DRS_IMPL_MFMA_BF16X3 ? 2 : 1; }

static bool is_std3x3(const TapConv& d) {
  if (d.mode != 0 || d.ntaps != 9 || d.in_stride != 1 || d.out_scale != 1) return false;
  for (int i = 0; i < 9; ++i)
    if (d.dy[i] != i / 3 - 1 || d.dx[i] != i % 3 - 1 || d.wtap[i] != i) return false;
  return true;
}

static int nwg_of(const TapConv& d, int mode) {
  static const int env = getenv("DRS_NWG") ? atoi(getenv("DRS_NWG")) : 2;
  return (env == 2 && mode == MODE_CONV3X3 && d.Cout % 64 == 0 && !d.shared_cu) ? 2 : 1;
}

static bool geom(const TapConv& d, int impl, MfmaGeom* g, int* bn, int* rpw, int* mode, size_t* lds) {
  *mode = d.mode == DRS_TAPMODE_CONVT ? MODE_CONVT : (is_std3x3(d) ? MODE_CONV3X3 : MODE_GENERIC);
  if (*mode == MODE_CONV3X3 && d.fuse_out) *mode = MODE_CONV3X3_FUSE;
  *bn = 32;  // BN = 64 would need > 256 VGPRs with the prefetch registers live across the epilogue
  // images of at most 8 rows (the 64x64 generation model's bottleneck): 8-row patches waste half as many MFMAs as
  // 16-row ones; the 3x3 schedule only exists for 16 rows, so those layers take the generic tap list
  const bool small = d.in && d.TH <= 8 && d.in_stride == 1;
  if (small && *mode == MODE_CONV3X3 && !d.in2) *mode = MODE_GENERIC;
  *rpw = (*mode == MODE_CONVT || d.in_stride != 1 || (small && *mode == MODE_GENERIC)) ? 2 : 4;
  // (measured: 32 x 16 patches with 8 rows per wave spill 60-125 VGPRs at 512 threads and run 5 % slower)
  const int TH = 4 * *rpw, TW = 16;
  if (*mode == MODE_CONVT) {
    g->dy_min = 0; g->dx_min = 0; g->IH = TH + 1; g->IW = TW + 1;
  } else {
    int dy0 = 1 << 30, dy1 = -(1 << 30), dx0 = 1 << 30, dx1 = -(1 << 30);
    for (int i = 0; i < d.ntaps; ++i) {
      dy0 = d.dy[i] < dy0 ? d.dy[i] : dy0; dy1 = d.dy[i] > dy1 ? d.dy[i] : dy1;
      dx0 = d.dx[i] < dx0 ? d.dx[i] : dx0; dx1 = d.dx[i] > dx1 ? d.dx[i] : dx1;
    }
    g->dy_min = dy0; g->dx_min = dx0;
    g->IH = (TH - 1) * d.in_stride + (dy1 - dy0) + 1;
    g->IW = (TW - 1) * d.in_stride + (dx1 - dx0) + 1;
  }
  const int nwg = nwg_of(d, *mode);
  const int a_iters = (*mode == MODE_CONV3X3 || *mode == MODE_CONV3X3_FUSE)
                          ? ((4 * *rpw + 2) * 18 + 64 * nwg - 1) / (64 * nwg)
                          : (*mode == MODE_CONVT ? 3 : (*rpw == 4 ? 6 : 9)) / nwg;
  if (g->IH * g->IW > a_iters * 64 * nwg) return false;
  g->tiles_x = drs_cdiv(d.TW, TW);
  g->tiles_y = drs_cdiv(d.TH, TH);
  const int KC = 4 * slot_ch(impl);
  g->nchunks = drs_cdiv(d.Cin, KC);
  g->a_plane = (g->IH * g->IW * 16 + 255) / 256 * 256;
  g->a_image = 4 * g->a_plane + 128;
  g->w_image = d.ntaps * 4 * *bn * nwg * 16;
  g->w_gimage = g->nchunks * d.wtaps_total * 4 * d.Cout * 16;
  g->nchunks2 = d.in2 ? drs_cdiv(d.Cin2, KC) : 0;
  g->iw_magic = (65536 + g->IW - 1) / g->IW;
  g->w2_gimage = g->nchunks2 * 4 * d.Cout * 16;
  static const int dbg = getenv("DRS_DEBUG_FLAGS") ? atoi(getenv("DRS_DEBUG_FLAGS")) : 0;
  g->debug = dbg;
  *lds = (size_t)images(impl) * ((size_t)g->a_image + g->w_image) + 128;  // + tap tables
  return *lds <= (size_t)kLdsLimit;
}

bool drs_tapconv_mfma_supported(const TapConv& d, int impl) {
  if (impl != DRS_IMPL_MFMA_F32 && impl != DRS_IMPL_MFMA_F16 && impl != DRS_IMPL_MFMA_BF16X3) return false;
  if (d.Cout % 32 != 0 || d.Cin % 4 != 0) return false;
  if (d.out_nchw || d.sigmoid) return false;
  if (d.ntaps < 1) return false;
  if (d.fuse_out && (d.Cout != 32 || d.fuse_dim > 4 || (d.in && !is_std3x3(d)) || d.res || d.gate || d.post_add ||
                     d.relu_pre || d.relu_post))
    return false;
  if ((d.in_sp || d.in2_sp || d.out_sp || d.res_sp || d.out2) && impl != DRS_IMPL_MFMA_BF16X3) return false;
  // SP instantiations: input(s) and output all in SP format (the plan's SP layers are); fp32 in -> SP out does not occur.
  // An fp32 OUTPUT from SP inputs (w_g, w_x: their results feed fp32 consumers) is the SP kernel with the fp32 epilogue.
  if (d.out_sp && !d.in_sp) return false;
  if (d.in2 && d.in2_sp != d.in_sp) return false;
  if ((d.out2 || d.dual) && !drs_tapconv_sp_supported(d, impl) && !(d.dual && drs_tapconv_ws_supported(d, impl)) &&
      !(d.out2 && !d.dual && drs_tapconv_sp8_supported(d, impl)))
    return false;  // second outputs / fused pairs come from the wave-specialised kernels only (plan.hip has the fallbacks)
  if (d.in_sp && (d.in_add || (d.in_co & 31) || (d.in_cs != 16 && (d.in_cs & 31)))) return false;
  if (d.in2_sp && ((d.in2_co & 31) || (d.in2_cs != 16 && (d.in2_cs & 31)))) return false;
  if (d.out_sp && ((d.out_co & 31) || (d.out_cs & 31) || d.fuse_out || (d.out2 && ((d.out2_co & 31) || (d.out2_cs & 31))))) return false;
  if (d.res_sp && ((d.res_co & 31) || (d.res_cs & 31))) return false;
  if (d.in == nullptr) return true;  // shape-only probe (weight packing): spatial details decided per launch
  if ((d.in_cs & 3) || (d.in_co & 3) || (d.out_cs & 3) || (d.out_co & 3)) return false;
  if (d.res && ((d.res_cs & 3) || (d.res_co & 3))) return false;
  if (d.post_add && (d.post_cs & 3)) return false;
  if (d.in_add && (d.in_add_cs & 3)) return false;
  if (d.in_stride != 1 && d.in_stride != 2) return false;
  if (d.mode == DRS_TAPMODE_CONVT && (d.ntaps != 9 || d.in_stride != 1 || d.out_scale != 2)) return false;
  if (d.mode == DRS_TAPMODE_CONVT && d.out_sp && (d.gate || d.relu_pre || d.relu_post || d.post_add || d.res || d.bias2)) return false;
  if (d.in2 && (!is_std3x3(d) || d.fuse_out || !d.w2 || (d.Cin2 & 3) || (d.in2_cs & 3) || (d.in2_co & 3) ||
                d.H2 != d.TH || d.W2 != d.TW))
    return false;
  MfmaGeom g; int bn, rpw, mode; size_t lds;
  return geom(d, impl, &g, &bn, &rpw, &mode, &lds);
}

template <class P, int BN, int RPW, int MODE, bool HAS2 = false, int NWG = 1, int SP = 0>
static int launch_t(const TapConv& d, const MfmaGeom& g, size_t lds, hipStream_t s) {
  auto kern = tapconv_mfma_kernel<P, BN, RPW, MODE, HAS2, NWG, SP>;
  int num_cu = 0;
  {
    const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(kern), kLdsLimit, &num_cu);
    if (rc) return rc;
  }
  // persistent grid: 2 blocks per CU (what the LDS footprint admits), a multiple of the 8 XCDs, never more than items
  const long long nitems = (long long)d.N * g.tiles_x * g.tiles_y * (d.Cout / (BN * NWG));
  static const int per_cu_env = getenv("DRS_BLOCKS_PER_CU") ? atoi(getenv("DRS_BLOCKS_PER_CU")) : 2;
  const int per_cu = NWG == 2 ? 1 : (d.shared_cu ? (d.shared_cu == 2 ? 2 : 1) : per_cu_env);
  long long blocks = (long long)num_cu * per_cu;
  if (blocks > nitems) blocks = nitems;
  blocks = (blocks + 7) / 8 * 8;
  dim3 grid((unsigned)blocks);
  DRS_LAUNCH(kern, grid, dim3(256 * NWG), lds, s, d, g);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

template <class P, int SP>
static int launch_p(const TapConv& d, const MfmaGeom& g, int bn, int rpw, int mode, size_t lds, hipStream_t s) {
  if (mode == MODE_CONVT) return launch_t<P, 32, 2, MODE_CONVT, false, 1, SP>(d, g, lds, s);
  if (bn == 32 && mode == MODE_CONV3X3 && nwg_of(d, mode) == 2) {
    if (d.in2) return launch_t<P, 32, 4, MODE_CONV3X3, true, 2, SP>(d, g, lds, s);
    return launch_t<P, 32, 4, MODE_CONV3X3, false, 2, SP>(d, g, lds, s);
  }
  if (bn == 32 && mode == MODE_CONV3X3 && d.in2) return launch_t<P, 32, 4, MODE_CONV3X3, true, 1, SP>(d, g, lds, s);
  if (bn == 32 && mode == MODE_CONV3X3) return launch_t<P, 32, 4, MODE_CONV3X3, false, 1, SP>(d, g, lds, s);
  if constexpr (SP != 2) {
    if (bn == 32 && mode == MODE_CONV3X3_FUSE) return launch_t<P, 32, 4, MODE_CONV3X3_FUSE, false, 1, SP>(d, g, lds, s);
  }
  if (bn == 32 && rpw == 4) return launch_t<P, 32, 4, MODE_GENERIC, false, 1, SP>(d, g, lds, s);
  if (bn == 32 && rpw == 2) return launch_t<P, 32, 2, MODE_GENERIC, false, 1, SP>(d, g, lds, s);
  DrsErr::set("tapconv_mfma: no kernel for BN=%d RPW=%d mode=%d", bn, rpw, mode);
  return DRS_ERR_SHAPE;
}

int drs_launch_tapconv_mfma(const TapConv& d, int impl, hipStream_t s) {
  DRS_REQUIRE(d.in && d.w && (d.out || d.fuse_out || d.out2), DRS_ERR_ARG, "tapconv_mfma: null tensor");
  DRS_REQUIRE(drs_tapconv_mfma_supported(d, impl), DRS_ERR_SHAPE, "tapconv_mfma: unsupported shape");
  if ((size_t)d.N * d.TH * d.TW == 0) return DRS_OK;
  if (drs_tapconv_sp8_supported(d, impl)) return drs_launch_tapconv_sp8(d, s);
  MfmaGeom g; int bn, rpw, mode; size_t lds;
  geom(d, impl, &g, &bn, &rpw, &mode, &lds);
  if (mode == MODE_GENERIC && drs_down_sp_supported(d, impl)) return drs_launch_down_sp(d, s);
  if (mode == MODE_GENERIC && drs_conv_s2_sp_supported(d, impl)) return drs_launch_conv_s2_sp(d, s);
  if (mode == MODE_CONVT && drs_convt_sp_supported(d, impl)) return drs_launch_convt_sp(d, s);
  if ((mode == MODE_CONV3X3 || mode == MODE_CONV3X3_FUSE) && drs_conv3x3_direct_sp_supported(d, impl))
    return drs_launch_conv3x3_direct_sp(d, s);
  if ((mode == MODE_CONV3X3 || mode == MODE_CONV3X3_FUSE) && drs_tapconv_sp_supported(d, impl)) {
    if (drs_tapconv_fl_supported(d, impl)) {
      const int rc = drs_launch_tapconv_fl(d, g, s);
      if (rc != DRS_FL_DECLINED) return rc;
    }
    return drs_launch_tapconv_sp(d, g, s);
  }
  if (mode == MODE_CONV3X3 && drs_tapconv_sp_f32out_supported(d, impl)) return drs_launch_tapconv_sp(d, g, s);
  if ((mode == MODE_CONV3X3 || mode == MODE_CONV3X3_FUSE) && drs_tapconv_ws_supported(d, impl))
    return drs_launch_tapconv_ws(d, g, impl, s);
  DRS_REQUIRE(!d.dual, DRS_ERR_SHAPE, "tapconv_mfma: the fused conv1 + skip op needs the wave-specialised kernel");
  if (impl == DRS_IMPL_MFMA_F32) return launch_p<PolicyF32, 0>(d, g, bn, rpw, mode, lds, s);
  if (impl == DRS_IMPL_MFMA_F16) return launch_p<PolicyF16, 0>(d, g, bn, rpw, mode, lds, s);
  if (d.in_sp) return d.out_sp ? launch_p<PolicyBF16X3, 2>(d, g, bn, rpw, mode, lds, s)
                               : launch_p<PolicyBF16X3, 1>(d, g, bn, rpw, mode, lds, s);
  return launch_p<PolicyBF16X3, 0>(d, g, bn, rpw, mode, lds, s);
}

// ---- weight packing for the MFMA kernels ------------------------------------------------------------------------
// dst image layout: [chunk][tap][kgroup(4)][Cout][SLOT_CH] elements, ci = chunk*KC + kgroup*SLOT_CH + j, zero-padded
// beyond Cin; BatchNorm (eval) folded like pack_conv_kernel.
// One layer's packing job.  The jobs of a whole plan (37 forward images after every optimizer step, as many data-gradient images
// per backward) are queued on the host and run as a few batched launches: blockIdx.y = job (drs_pack_queue_*, below).
struct PackJob {
  const float *w, *b, *gamma, *beta, *rmean, *rvar;
  char* dst_w;
  float* dst_b;
  size_t image_bytes;
  float eps;
  int Cout, Cin, taps, transposed, nchunks, cout_src, flip_taps, co_off, partial, perm, cin_total, cin_off;
};
constexpr int kPackBatch = 24;  // jobs per launch (the descriptors travel as kernel arguments: 24 x 120 bytes)
struct PackBatch { PackJob job[kPackBatch]; };

template <class P>
__global__ void pack_conv_mfma_kernel(PackBatch batch) {
  const PackJob& e = batch.job[blockIdx.y];
  const float* __restrict__ w = e.w;
  const float* __restrict__ b = e.b;
  const float* __restrict__ gamma = e.gamma;
  const float* __restrict__ beta = e.beta;
  const float* __restrict__ rmean = e.rmean;
  const float* __restrict__ rvar = e.rvar;
  char* __restrict__ dst_w = e.dst_w;
  float* __restrict__ dst_b = e.dst_b;
  const float eps = e.eps;
  const int Cout = e.Cout, Cin = e.Cin, taps = e.taps, transposed = e.transposed, nchunks = e.nchunks, cout_src = e.cout_src,
            flip_taps = e.flip_taps, co_off = e.co_off, partial = e.partial, perm = e.perm, cin_total = e.cin_total, cin_off = e.cin_off;
  const size_t image_bytes = e.image_bytes;
  // cin_total > 0: the source has cin_total input channels of which [cin_off, cin_off + Cin) are packed (one half of a
  // convolution over a channel concatenation); not for transposed sources
  // cout_src < Cout: the source has only cout_src output channels; the rest of the image is zero (a 16-channel
  // output padded to the kernel's 32-channel tile)
  constexpr int KC = 4 * P::SLOT_CH;
  const size_t nslots = (size_t)nchunks * taps * 4 * Cout;
  for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += (size_t)gridDim.x * blockDim.x) {
    // co_off / partial: the source fills channels [co_off, co_off + cout_src) of a wider image; with `partial` the other
    // channels are left alone (a second layer packed into the same image)
    // perm (layers that store their output in SP format): MFMA row 4*kg + j of n-tile t of a 32-channel group carries
    // logical channel kg*8 + t*4 + j, so that a lane's accumulators of a tile pair are 8 consecutive channels
    int co_phys = (int)(s % Cout);
    if (perm) {
      const int nn = co_phys & 31;
      co_phys = (co_phys & ~31) + ((nn & 15) >> 2) * 8 + (nn >> 4) * 4 + (nn & 3);
    }
    const int co = co_phys - co_off;
    if (partial && (co < 0 || co >= cout_src)) continue;
    const int q = (int)((s / Cout) & 3);
    const int tap = (int)((s / ((size_t)Cout * 4)) % taps);
    const int tap_src = flip_taps ? taps - 1 - tap : tap;  // data gradients convolve with the spatially flipped kernel
    const int c = (int)(s / ((size_t)Cout * 4 * taps));
    float sc = 1.f;
    if (gamma && co >= 0 && co < cout_src) sc = gamma[co] / sqrtf(rvar[co] + eps);
    float x[P::SLOT_CH];
#pragma unroll
    for (int j = 0; j < P::SLOT_CH; ++j) {
      const int ci = c * KC + q * P::SLOT_CH + j;
      float v = 0.f;
      if (ci < Cin && co >= 0 && co < cout_src) {
        const size_t src = transposed ? (((size_t)ci * cout_src + co) * taps + tap_src)
                                      : (((size_t)co * (cin_total > 0 ? cin_total : Cin) + cin_off + ci) * taps + tap_src);
        v = w[src];
        if (gamma) v *= sc;
      }
      x[j] = v;
    }
    P::cvt_store(dst_w, image_bytes, s * 16, x);
  }
  if (blockIdx.x == 0 && dst_b) {  // (dst_b null: a second operand image of a layer whose bias slot another job of the batch writes)
    for (int cot = threadIdx.x; cot < Cout; cot += blockDim.x) {
      const int co = cot - co_off;
      const bool mine = co >= 0 && co < cout_src;
      if (partial && !mine) continue;
      float bb = (b && mine) ? b[co] : 0.f;
      if (gamma && mine) {
        const float sc = gamma[co] / sqrtf(rvar[co] + eps);
        bb = (bb - rmean[co]) * sc + beta[co];
      }
      dst_b[cot] = bb;
    }
  }
}

size_t drs_pack_conv_mfma_bytes(int Cout, int Cin, int taps, int impl) {
  const int KC = 4 * slot_ch(impl);
  return (size_t)images(impl) * drs_cdiv(Cin, KC) * taps * 4 * Cout * 16;
}

// ---- packing queue (per host thread) -------------------------------------------------------------------------------
// Between drs_pack_queue_begin and drs_pack_queue_flush every drs_launch_pack_conv_mfma call only records its job; the
// flush launches the recorded jobs of each operand policy kPackBatch at a time.  Jobs that write into the same image
// (`partial`) touch disjoint channels, so their order inside a launch does not matter.
namespace {
struct PackQueue {
  bool open = false;
  std::vector<PackJob> jobs[3];  // by policy: fp32, fp16, split bf16
  std::vector<int> blocks[3];
};
thread_local PackQueue g_pack_queue;

template <class P>
int pack_launch(const PackJob* jobs, const int* blocks, int n, hipStream_t s) {
  for (int i0 = 0; i0 < n; i0 += kPackBatch) {
    PackBatch batch = {};
    const int m = n - i0 < kPackBatch ? n - i0 : kPackBatch;
    int bx = 1;
    for (int i = 0; i < m; ++i) { batch.job[i] = jobs[i0 + i]; bx = blocks[i0 + i] > bx ? blocks[i0 + i] : bx; }
    // (every job of the launch gets the largest job's block count: the kernel's slot loop is grid-strided, extra blocks exit)
    DRS_LAUNCH(pack_conv_mfma_kernel<P>, dim3(bx, m), dim3(256), 0, s, batch);
  }
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
int pack_dispatch(int pol, const PackJob* jobs, const int* blocks, int n, hipStream_t s) {
  if (n == 0) return DRS_OK;
  if (pol == 0) return pack_launch<PolicyF32>(jobs, blocks, n, s);
  if (pol == 1) return pack_launch<PolicyF16>(jobs, blocks, n, s);
  return pack_launch<PolicyBF16X3>(jobs, blocks, n, s);
}
}  // namespace

void drs_pack_queue_begin() {
  PackQueue& q = g_pack_queue;
  q.open = true;
  for (int p = 0; p < 3; ++p) { q.jobs[p].clear(); q.blocks[p].clear(); }
}
void drs_pack_queue_abandon() {
  PackQueue& q = g_pack_queue;
  q.open = false;
  for (int p = 0; p < 3; ++p) { q.jobs[p].clear(); q.blocks[p].clear(); }
}
int drs_pack_queue_flush(hipStream_t s) {
  PackQueue& q = g_pack_queue;
  q.open = false;
  for (int p = 0; p < 3; ++p) {
    const int rc = pack_dispatch(p, q.jobs[p].data(), q.blocks[p].data(), (int)q.jobs[p].size(), s);
    q.jobs[p].clear();
    q.blocks[p].clear();
    if (rc) return rc;
  }
  return DRS_OK;
}

int drs_launch_pack_conv_mfma(const float* w, const float* b, const float* gamma, const float* beta, const float* rmean,
                              const float* rvar, float eps, void* dst_w, float* dst_b, int Cout, int Cin, int taps,
                              int transposed, int impl, hipStream_t s, int cout_src, int flip_taps, int co_off, int partial,
                              int perm, int cin_total, int cin_off) {
  if (cout_src <= 0) cout_src = Cout;
  const int KC = 4 * slot_ch(impl);
  const int nchunks = drs_cdiv(Cin, KC);
  const size_t image = (size_t)nchunks * taps * 4 * Cout * 16;
  const size_t nslots = image / 16;
  int blocks = (int)((nslots + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  const PackJob job{w, b, gamma, beta, rmean, rvar, (char*)dst_w, dst_b, image, eps, Cout, Cin, taps, transposed, nchunks,
                    cout_src, flip_taps, co_off, partial, perm, cin_total, cin_off};
  const int pol = impl == DRS_IMPL_MFMA_F32 ? 0 : (impl == DRS_IMPL_MFMA_F16 ? 1 : 2);
  PackQueue& q = g_pack_queue;
  if (q.open) {
    q.jobs[pol].push_back(job);
    q.blocks[pol].push_back(blocks);
    return DRS_OK;
  }
  return pack_dispatch(pol, &job, &blocks, 1, s);
}
