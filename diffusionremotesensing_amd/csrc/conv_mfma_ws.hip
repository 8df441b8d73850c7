// Wave-specialised 3x3 stride-1 tap-convolution for the wide layers (Cout % 64 == 0, Cin % KC == 0).
//
// Same GEMM view, LDS slot layout, MFMA schedule and epilogue as tapconv_mfma_kernel<.., CONV3X3, .., NWG = 2>
// (conv_mfma.hip); what changes is WHO moves the operands and WHEN.  In that kernel every wave alternates between
// "issue the loads of the next chunk" (the wave sits in the vector-memory queue for a memory round trip), "convert +
// store to LDS" and "multiply", all 8 waves of a CU in lock step: the memory system idles while the matrix cores run
// and vice versa (phase timeline).  Here a block is 12 waves on one CU:
//   * 8 consumer waves (4 row-waves x 2 channel groups, 2 per SIMD) only read fragments from LDS and issue MFMAs;
//   * 4 producer waves (1 per SIMD) only move data: the chunk's WEIGHTS arrive by LDS-DMA (global_load_lds_dwordx4:
//     the packed weight image is already in operand format, so no registers and no conversion), the input WINDOW is
//     loaded to registers two steps ahead, converted to the operand format and stored one step ahead into the other
//     of two window buffers.
// LDS budget (bf16x3): 2 window buffers x 42 KB + ONE weight buffer of 72 KB = 157 KB.  The weight buffer is a ring
// of the 3 kernel columns: the consumers copy a column's 6 fragments to registers before they multiply with it, so
// its LDS slot is free again two thirds of a step before the next chunk needs it - the producers refill it then.
// Step k (one 32-channel chunk of one patch), three block-wide barriers Y1..Y3, all 12 waves:
//   Y1(k): consumers have read column 0 of k and finished step k-1 | window k stored, column 1 of k landed
//          C: MFMA column 0, read column 1          P: store window k+1, DMA column 0 of k+1
//   Y2(k): consumers have read column 1                            | column 2 of k landed
//          C: MFMA column 1, read column 2          P: DMA column 1 of k+1, load window k+2 to registers
//   Y3(k): consumers have read column 2                            | column 0 of k+1 landed
//          C: MFMA column 2, epilogue, read column 0 of k+1        P: DMA column 2 of k+1
// Barriers are raw s_barrier + explicit counted s_waitcnt: a __syncthreads() would drain the DMAs in flight.
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
// wait until at most n of this wave's vector-memory operations are outstanding (n is wave-uniform, 0..31)
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
__device__ unsigned long long drs_ws_tl[32];
#define WS_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl[i] += t_ - tl_last; tl_last = t_; } while (0)
#else
#define WS_STAMP(i) do { } while (0)
#endif

typedef __attribute__((address_space(1))) void* ws_gptr;
typedef __attribute__((address_space(3))) void* ws_lptr;

template <class P, bool HAS2>
__global__ __launch_bounds__(768, 1) void tapconv_ws_kernel(TapConv d, MfmaGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KC = 4 * P::SLOT_CH;
  constexpr int RPW = 4, NT = 2, BN = 32, BNB = 64, TH = 16, TW = 16;
  constexpr int A_ITERS = 6;               // window slots per producer thread: 18 x 18 px x 4 k-groups / 256
  constexpr int V4 = P::SLOT_CH / 4;       // float4 loads per window slot
  constexpr int NLOAD = A_ITERS * V4;      // vector loads of one window per producer thread
  constexpr int WPI = P::IMAGES * 12 / 4;  // 1 KB DMA pieces per producer wave and kernel column
  // the geometry is fixed (18 x 18 window, 64 channels per block): compile-time LDS offsets fold into the ds_read /
  // ds_write immediates instead of living in one address register per fragment
  constexpr int IW = 18, NPIX = 18 * 18;
  constexpr int A_PLANE = (NPIX * 16 + 255) / 256 * 256, A_IMAGE = 4 * A_PLANE + 128, W_IMAGE = 9 * 4 * BNB * 16;
  constexpr int a_buf = P::IMAGES * A_IMAGE;  // bytes of one window buffer
  char* sA = smem;                          // [buffer(2)][image][kgroup(4)][window pixel] slots
  char* sW = smem + 2 * (size_t)a_buf;      // [image][kx(3)][ky(3)][kgroup(4)][BNB] slots

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7 consumers, 8..11 producers
  const int lr = lane & 15, kg = lane >> 4;

  // persistent blocks, XCD-aware item order (see tapconv_mfma_kernel)
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

  if (wid >= 8) {
    // =============================== producers ===============================
    const int pt = tid - 512, pw = wid - 8;
    constexpr int npix = NPIX;
    const int ai = pt & 15;
    const int aq = ((ai & 1) << 1) | (ai >> 3);  // k-group of all of this thread's slots (conflict-free ds_write order)
    const int ap0 = (pt >> 4) * 4 + ((ai & 7) >> 1);
    const int a_qoff = aq * A_PLANE + (aq >> 1) * 128;
    const bool has_add = d.in_add != nullptr;
    const char* wg = reinterpret_cast<const char*>(d.w);
    const size_t w_chunk = (size_t)d.wtaps_total * 4 * d.Cout * 16;

    float4 areg[A_ITERS][V4];
    float4 addreg[V4];
    unsigned a_ok = 0;
    bool held_second = false;
    int vm_issued = 0;  // vector-memory operations this wave has issued so far (program order)

    auto load_window = [&](int k) {
      const int c = k % nck;
      int ln, lty0, ltx0, ln0;
      item_of(k / nck, ln, lty0, ltx0, ln0);
      held_second = HAS2 && c >= g.nchunks;
      const int cc = held_second ? c - g.nchunks : c;
      int a_base[A_ITERS];
      a_ok = 0;
      const float* in_n;
      if (!held_second) {
        in_n = d.in + (size_t)ln * d.H * d.W * d.in_cs;
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
          const int p = min(ap0 + it * 64, npix - 1);
          const int py = p / IW, px = p - py * IW;
          const int iy = lty0 - 1 + py, ix = ltx0 - 1 + px;
          const bool ok = (ap0 + it * 64) < npix && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
          const int iyc = min(max(iy, 0), d.H - 1), ixc = min(max(ix, 0), d.W - 1);
          a_base[it] = (iyc * d.W + ixc) * d.in_cs + d.in_co + cc * KC + aq * P::SLOT_CH;
          a_ok |= (ok ? 1u : 0u) << it;
        }
      } else {
        in_n = d.in2 + (size_t)ln * d.H2 * d.W2 * d.in2_cs;
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
          const int p = min(ap0 + it * 64, npix - 1);
          const int py = p / IW, px = p - py * IW;
          const int iy = lty0 + py, ix = ltx0 + px;
          const bool ok = (ap0 + it * 64) < npix && py < TH && px < TW && iy < d.H2 && ix < d.W2;
          const int iyc = min(iy, d.H2 - 1), ixc = min(ix, d.W2 - 1);
          a_base[it] = (iyc * d.W2 + ixc) * d.in2_cs + d.in2_co + cc * KC + aq * P::SLOT_CH;
          a_ok |= (ok ? 1u : 0u) << it;
        }
      }
#pragma unroll
      for (int v = 0; v < V4; ++v) {
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) areg[it][v] = *reinterpret_cast<const float4*>(in_n + a_base[it] + 4 * v);
      }
      vm_issued += NLOAD;
      if (has_add && !held_second) {
#pragma unroll
        for (int v = 0; v < V4; ++v)
          addreg[v] = *reinterpret_cast<const float4*>(d.in_add + (size_t)ln * d.in_add_cs + cc * KC + aq * P::SLOT_CH + 4 * v);
        vm_issued += V4;
      }
    };
    auto store_window = [&](int buf) {
      char* dst = sA + (size_t)buf * a_buf;
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        const int p = ap0 + it * 64;
        const bool ok = (a_ok >> it) & 1u;
        float x[P::SLOT_CH];
#pragma unroll
        for (int v = 0; v < V4; ++v) {
          float4 a = areg[it][v];
          if (has_add && !held_second) {
            a.x += addreg[v].x; a.y += addreg[v].y; a.z += addreg[v].z; a.w += addreg[v].w;
          }
          x[4 * v] = ok ? a.x : 0.f; x[4 * v + 1] = ok ? a.y : 0.f; x[4 * v + 2] = ok ? a.z : 0.f; x[4 * v + 3] = ok ? a.w : 0.f;
        }
        if (p < npix) P::cvt_store(dst + a_qoff, A_IMAGE, (size_t)p * 16, x);
      }
    };
    // DMA kernel column `col` of step k's weights into ring slot `col`; returns the issue count after the group
    int w_c = 0, w_n0 = 0;  // chunk and first output channel of the step whose weights are being fetched
    auto set_w = [&](int k) {
      int ln, lty0, ltx0;
      w_c = k % nck;
      item_of(k / nck, ln, lty0, ltx0, w_n0);
    };
    auto issue_w = [&](int col) {
      const int c = w_c, ln0 = w_n0;
      const bool second = HAS2 && c >= g.nchunks;
      if (!second) {
#pragma unroll
        for (int i = 0; i < WPI; ++i) {
          const int idx = pw * WPI + i;  // (image, ky, k-group) piece of this wave
          const int im = idx / 12, ky = (idx % 12) >> 2, kq = idx & 3;
          const char* src = wg + (size_t)im * g.w_gimage + (size_t)c * w_chunk +
                            ((size_t)((ky * 3 + col) * 4 + kq) * d.Cout + ln0 + lane) * 16;
          char* dst = sW + (size_t)im * W_IMAGE + (size_t)((col * 3 + ky) * 4 + kq) * BNB * 16;
          __builtin_amdgcn_global_load_lds((ws_gptr)src, (ws_lptr)dst, 16, 0, 0);
        }
        vm_issued += WPI;
      } else if (col == 0) {  // second input: one tap = IMAGES x 4 pieces, stored where (kx 0, ky 0) lives
        const int cc = c - g.nchunks;
#pragma unroll
        for (int i = 0; i < (P::IMAGES * 4 + 3) / 4; ++i) {
          const int idx = pw * ((P::IMAGES * 4 + 3) / 4) + i;
          const int im = idx >> 2, kq = idx & 3;
          const char* src = reinterpret_cast<const char*>(d.w2) + (size_t)im * g.w2_gimage +
                            ((size_t)(cc * 4 + kq) * d.Cout + ln0 + lane) * 16;
          char* dst = sW + (size_t)im * W_IMAGE + (size_t)kq * BNB * 16;
          __builtin_amdgcn_global_load_lds((ws_gptr)src, (ws_lptr)dst, 16, 0, 0);
        }
        vm_issued += (P::IMAGES * 4 + 3) / 4;
      }
      return vm_issued;
    };

    int end_col[3];  // issue count right after the DMA group that fills ring slot j
#ifdef DRS_WS_TIMELINE
    unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_amdgcn_s_memtime();
#endif
    set_w(0);
    end_col[0] = issue_w(0);
    end_col[1] = issue_w(1);
    end_col[2] = issue_w(2);
    load_window(0);
    store_window(0);
    if (S > 1) load_window(1);
    ws_wait_vm(vm_issued - end_col[2]);  // all three columns of step 0 landed
    ws_wait_lds();
    ws_barrier();  // P0
    WS_STAMP(0);
    for (int k = 0; k < S; ++k) {
      const bool more = k + 1 < S;
      ws_wait_vm(vm_issued - end_col[1]);  // column 1 of k landed
      ws_wait_lds();                       // window k stored
      WS_STAMP(1);
      ws_barrier();                        // Y1
      WS_STAMP(2);
      if (more) {
        store_window((k + 1) & 1);
        set_w(k + 1);
        end_col[0] = issue_w(0);
      }
      WS_STAMP(3);
      ws_wait_vm(vm_issued - end_col[2]);  // column 2 of k landed
      WS_STAMP(1);
      ws_barrier();                        // Y2
      WS_STAMP(4);
      if (more) end_col[1] = issue_w(1);
      if (k + 2 < S) load_window(k + 2);
      WS_STAMP(5);
      ws_wait_vm(vm_issued - end_col[0]);  // column 0 of k+1 landed
      WS_STAMP(1);
      ws_barrier();                        // Y3
      WS_STAMP(6);
      if (more) end_col[2] = issue_w(2);
      WS_STAMP(7);
    }
    ws_wait_vm(0);
#ifdef DRS_WS_TIMELINE
    if (blockIdx.x == 0 && wid == 8 && lane == 0) {
      for (int i = 0; i < 8; ++i) drs_ws_tl[16 + i] = tl[i];
      drs_ws_tl[24] = S;
    }
#endif
    return;
  }

  // =============================== consumers ===============================
  const int rw = wid & 3;   // row-wave: rows [rw*RPW, rw*RPW + RPW) of the patch
  const int ng = wid >> 2;  // channel group: channels [ng*BN, ng*BN + BN) of the block's BNB
  const int kg_off = kg * A_PLANE + (kg >> 1) * 128 + (rw * RPW * IW + lr) * 16;  // this lane's window origin
  const char* wbase = sW + ((size_t)kg * BNB + ng * NT * 16 + lr) * 16;                   // this lane's weight origin
  f32x4 acc[RPW][NT];
  typename P::Frag wf[3][NT];
  auto read_wf = [&](int col) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        wf[ky][t] = P::load(wbase, W_IMAGE, (size_t)(((col * 3 + ky) * 4 * BNB) + t * 16) * 16);
  };
  auto mma_col = [&](const char* win, int col) {
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
  ws_barrier();  // P0
#ifdef DRS_WS_TIMELINE
  unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_amdgcn_s_memtime();
#endif
  read_wf(0);
  int n = 0, ty0 = 0, tx0 = 0, n0 = 0;
  for (int k = 0; k < S; ++k) {
    const int c = k % nck;
    if (c == 0) {
      item_of(k / nck, n, ty0, tx0, n0);
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const char* win = sA + (size_t)(k & 1) * a_buf + kg_off;
    WS_STAMP(0);
    ws_wait_lds();
    ws_barrier();  // Y1
    WS_STAMP(1);
    if (HAS2 && c >= g.nchunks) {  // second input: one tap, window origin; columns 1 and 2 are empty
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const typename P::Frag af = P::load(win, A_IMAGE, (size_t)(r * IW) * 16);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = P::mma(wf[0][t], af, acc[r][t]);
      }
      ws_wait_lds();
      ws_barrier();  // Y2
      ws_barrier();  // Y3
    } else {
      mma_col(win, 0);
      read_wf(1);
      WS_STAMP(2);
      ws_wait_lds();
      ws_barrier();  // Y2
      WS_STAMP(3);
      mma_col(win, 1);
      read_wf(2);
      WS_STAMP(4);
      ws_wait_lds();
      ws_barrier();  // Y3
      WS_STAMP(5);
      mma_col(win, 2);
      WS_STAMP(6);
    }
    if (c == nck - 1) {
      // opaque copies of the lane coordinates: everything the epilogue derives from them is computed here, not hoisted
      // out of the step loop (where it would be spilled and reloaded with vmcnt(0) waits between the stores)
      int lr_e = lr, kg_e = kg;
      asm volatile("" : "+v"(lr_e), "+v"(kg_e));
      tile_epilogue<RPW, NT, false, RPW>(d, acc, n, n0 + ng * BN, ty0, tx0, rw, lr_e, kg_e, d.out_oy, d.out_ox);
    }
    if (k + 1 < S) read_wf(0);
    WS_STAMP(7);
  }
#ifdef DRS_WS_TIMELINE
  if (blockIdx.x == 0 && wid == 0 && lane == 0) {
    for (int i = 0; i < 8; ++i) drs_ws_tl[i] = tl[i];
    drs_ws_tl[8] = S;
  }
#endif
}

bool ws_std3x3(const TapConv& d) {
  if (d.mode != 0 || d.ntaps != 9 || d.wtaps_total != 9 || d.in_stride != 1 || d.out_scale != 1) return false;
  for (int i = 0; i < 9; ++i)
    if (d.dy[i] != i / 3 - 1 || d.dx[i] != i % 3 - 1 || d.wtap[i] != i) return false;
  return true;
}

template <class P, bool HAS2>
int ws_launch(const TapConv& d, const MfmaGeom& g, size_t lds, hipStream_t s) {
  auto kern = tapconv_ws_kernel<P, HAS2>;
  static bool attr_done = false;
  static int num_cu = 0;
  if (!attr_done) {
    DRS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024));
    int dev = 0;
    DRS_CHECK_HIP(hipGetDevice(&dev));
    DRS_CHECK_HIP(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    attr_done = true;
  }
  const long long nitems = (long long)d.N * g.tiles_x * g.tiles_y * (d.Cout / 64);
  long long blocks = num_cu;  // one 12-wave block per CU
  if (blocks > nitems) blocks = nitems;
  blocks = (blocks + 7) / 8 * 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(768), lds, s, d, g);
  DRS_CHECK_HIP(hipGetLastError());
#ifdef DRS_WS_TIMELINE
  {
    unsigned long long h[32];
    DRS_CHECK_HIP(hipStreamSynchronize(s));
    DRS_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(drs_ws_tl), sizeof(h)));
    const double sc = h[8] ? 1.0 / (double)h[8] : 0.0;
    fprintf(stderr, "ws Cin=%d Cout=%d TH=%d S=%llu | C: preY1 %.0f Y1 %.0f col0 %.0f Y2 %.0f col1 %.0f Y3 %.0f col2 %.0f epi+rd %.0f | P: wait %.0f Y1 %.0f st+dma0 %.0f Y2 %.0f dma1+ld %.0f Y3 %.0f dma2 %.0f\n",
            d.Cin, d.Cout, d.TH, h[8], h[0] * sc, h[1] * sc, h[2] * sc, h[3] * sc, h[4] * sc, h[5] * sc, h[6] * sc, h[7] * sc,
            h[17] * sc, h[18] * sc, h[19] * sc, h[20] * sc, h[21] * sc, h[22] * sc, h[23] * sc);
  }
#endif
  return DRS_OK;
}

}  // namespace

// Eligibility of the wave-specialised kernel; `g` must come from the CONV3X3 / NWG = 2 geometry of conv_mfma.hip.
bool drs_tapconv_ws_supported(const TapConv& d, int impl) {
  static const int env = getenv("DRS_WS") ? atoi(getenv("DRS_WS")) : 0;
  if (!env) return false;
  if (impl != DRS_IMPL_MFMA_BF16X3 && impl != DRS_IMPL_MFMA_F32) return false;
  const int KC = impl == DRS_IMPL_MFMA_F32 ? 16 : 32;
  if (!d.in || !ws_std3x3(d) || d.fuse_out || d.shared_cu || d.gate) return false;
  if (d.Cout % 64 != 0 || d.Cin % KC != 0 || d.TH <= 8) return false;
  if (d.in2 && (d.Cin2 % KC != 0)) return false;
  return true;
}

int drs_launch_tapconv_ws(const TapConv& d, const MfmaGeom& g, int impl, hipStream_t s) {
  const size_t lds = (size_t)(impl == DRS_IMPL_MFMA_F32 ? 1 : 2) * (2 * (size_t)g.a_image + g.w_image);
  DRS_REQUIRE(lds <= 160 * 1024, DRS_ERR_SHAPE, "tapconv_ws: %zu bytes of LDS", lds);
  DRS_REQUIRE(g.IH == 18 && g.IW == 18 && g.w_image == 9 * 4 * 64 * 16 && g.a_plane == 5376 && g.a_image == 4 * 5376 + 128,
              DRS_ERR_SHAPE, "tapconv_ws: geometry");
  if (impl == DRS_IMPL_MFMA_F32)
    return d.in2 ? ws_launch<PolicyF32, true>(d, g, lds, s) : ws_launch<PolicyF32, false>(d, g, lds, s);
  return d.in2 ? ws_launch<PolicyBF16X3, true>(d, g, lds, s) : ws_launch<PolicyBF16X3, false>(d, g, lds, s);
}
