// The wave-specialised SP 3x3 kernel (conv_mfma_sp.hip) for 8 x 8 images: the bottleneck level of a 64 x 64 model (BASELINE
// configs[4]: class-conditional generation, 128 rows per guided step; conv1 / conv2 (+ 1x1 shortcut) of the bottleneck block and
// ups.0.conv, reference UNet_model_generation.py / UNet_model_superres.py:153-172,197-207).
//
// conv_mfma_sp.hip works on 16 x 16 patches: an 8 x 8 image would fill a quarter of one, so those layers ran on the lock-step
// kernel's generic tap list at 80 - 110 TFLOP/s (configs[4]: 266 of a step's 1090 us).  Here an item is FOUR images x 32 output
// channels: an image is four MFMA pixel blocks of 2 rows x 8 columns, 16 blocks per item - the same 16 x 16 x 32 output tile
// as the 32-channel flavour of the big kernel, so block structure, protocol and weight ring are its own:
//   12 waves on one CU: 8 consumers (wave w: image w >> 1 of the item, blocks 2 (w & 1) and + 1) + 4 movers
//   window = 4 x (10 x 10) pixels with their zero border (50 KB), double-buffered, rotated pixel-major operand image
//   weights through the ring of the 3 kernel columns (37 KB at 32 output channels); ten monotonic LDS counters, no barrier
// Output addressing needs no special case: an 8 x 8 image, row-major, IS a 4 x 16 image whose row b is pixel block b.
// Per-image epilogue vectors (time embedding) come straight from memory: a wave's two blocks belong to one image, and these
// launches have one or two items per CU.
#include <stdio.h>
#include <stdlib.h>

#include "conv_epilogue.h"
#include "mfma_policy.h"
#include "sp_sync.h"

namespace {

constexpr int kImgs = 4, kWp = 10, kIpix = kWp * kWp, kNpix = kImgs * kIpix;  // window: 4 images x 10 x 10 pixels
constexpr int kNblk = (kNpix + 7) / 8;                                          // 50 pieces of 8 pixels (1 KB)
constexpr int kWbuf = kNblk * 1024;
constexpr int kBnb = 32, kWImage = 9 * 4 * kBnb * 16;                           // one operand image (hi or lo) of a chunk's weights
constexpr int kLds = 2 * kWbuf + 2 * kWImage + 64;

template <bool HAS2>
__global__ __launch_bounds__(768, 1) void tapconv_sp8_kernel(TapConv d, int nchunks, int nchunks2, unsigned w_gimage,
                                                             unsigned w2_gimage) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  constexpr int KC = 32, NT = 2, RPW = 2, BNB = kBnb, W_IMAGE = kWImage, WBUF = kWbuf;
  char* sWin = smem;
  char* sW = smem + 2 * WBUF;  // [image][kx(3)][ky(3)][k-group(4)][BNB] operand slots
  sp_flag_ptr sCR = (sp_flag_ptr)(sW + 2 * W_IMAGE);  // CR[3] | CL[3] | WL[2] | WR[2]: see conv_mfma_sp.hip
  sp_flag_ptr sCL = sCR + 3;
  sp_flag_ptr sWL = sCR + 6;
  sp_flag_ptr sWR = sCR + 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7 consumers, 8..11 movers
  const int lr = lane & 15, kg = lane >> 4;
  const int ngroups = d.Cout / BNB;
  const int nquads = (d.N + kImgs - 1) / kImgs;
  const int nitems = nquads * ngroups;
  const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
  const int per = (nitems + 7) >> 3;
  const int lo_item = xcd * per, hi_item = min(nitems, lo_item + per);
  const int span = hi_item - lo_item - j8;
  const int my_items = span > 0 ? (span + nb8 - 1) / nb8 : 0;
  const int nck = nchunks + (HAS2 ? nchunks2 : 0);
  const int S = my_items * nck;
  if (S == 0) return;
  auto item_of = [&](int ordinal, int& quad_, int& n0_) __attribute__((always_inline)) {
    const int it = lo_item + ordinal * nb8 + j8;
    n0_ = (it % ngroups) * BNB;
    quad_ = it / ngroups;
  };
  if (tid < 10) __hip_atomic_store(sCR + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  sp_wait_lds();
  sp_barrier();
  int c = -1, ord = -1, quad = 0, n0 = 0;

  if (wid >= 8) {
    // ===================== movers: global -> registers (a step ahead) -> LDS =====================
    // Protocol of conv_sp_movers.inc: loads of column 0, window, column 1; wait CR0 -> store column 0 -> CL0; wait WR -> store
    // window -> WL; wait CR1 -> store column 1 -> CL1; (column 2 loaded into column 0's registers) wait CR2 -> store -> CL2.
    const int pw = wid - 8;
    __builtin_amdgcn_s_setprio(3);
    const char* zero = reinterpret_cast<const char*>(d.zero_line) + (lane & 15) * 16;
    const int l_px = lane >> 3, l_c = ((lane & 7) - l_px) & 7, l_img = l_c >> 2, l_kg = l_c & 3;
    const int half1 = drs_sp_group_bytes(d.in_cs), half2 = HAS2 ? drs_sp_group_bytes(d.in2_cs) : 0;
    const int l_off1 = l_img * half1 + l_kg * 16, l_off2 = l_img * half2 + l_kg * 16;
    constexpr int NPW = (kNblk + 3) / 4;                 // window pieces per mover wave (at most): blocks pw + 4 i
    const int nwin = (kNblk - pw + 3) / 4;               // 13, 13, 12, 12
    constexpr int WPC = 3;                               // weight pieces (1 KB = two k-group rows of 32 channels) per wave and column
    u32x4 wr[2][WPC], ww[NPW];
    const unsigned lane_w = (unsigned)(((lane >> 5) * d.Cout + (lane & 31)) * 16);
    const char* wg = reinterpret_cast<const char*>(d.w);
    const size_t w_chunk = (size_t)9 * 4 * d.Cout * 16;
    auto piece = [&](int col, bool second, int cc, int n0_, int i, const char*& src, char*& dst) __attribute__((always_inline)) {
      const int row0 = (pw * (second ? 1 : WPC) + i) * 2;
      const int im = second ? row0 >> 2 : row0 / 12, ky = second ? 0 : (row0 % 12) >> 2, kq = row0 & 3;
      const char* base = second ? reinterpret_cast<const char*>(d.w2) + (size_t)cc * 4 * d.Cout * 16 + (size_t)n0_ * 16
                                : wg + (size_t)cc * w_chunk + (size_t)n0_ * 16;
      const unsigned soff = second ? (unsigned)im * w2_gimage + (unsigned)(kq * d.Cout * 16)
                                   : (unsigned)im * w_gimage + (unsigned)(((ky * 3 + col) * 4 + kq) * d.Cout * 16);
      src = base + soff + lane_w;
      dst = sW + im * W_IMAGE + (((second ? 0 : col) * 3 + ky) * 4 + kq) * BNB * 16 + lane * 16;
    };
    bool prev_fill = false;  // the previous step stored weight columns 1 / 2
    for (int k = 0; k < S; ++k) {
      if (++c == nck) c = 0;
      if (c == 0) item_of(++ord, quad, n0);
      const bool second = HAS2 && c >= nchunks;
      const int cc = second ? c - nchunks : c;
      int np[3];
#pragma unroll
      for (int col = 0; col < 3; ++col) np[col] = second ? (col == 0 ? 1 : 0) : WPC;
      auto load_col = [&](int col) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WPC; ++i)
          if (i < np[col]) {
            const char* src; char* dst;
            piece(col, second, cc, n0, i, src, dst);
            wr[col & 1][i] = *reinterpret_cast<const u32x4*>(src);
          }
      };
      auto store_col = [&](int col) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WPC; ++i)
          if (i < np[col]) {
            const char* src; char* dst;
            piece(col, second, cc, n0, i, src, dst);
            *reinterpret_cast<u32x4*>(dst) = wr[col & 1][i];
          }
        sp_wait_lds();
        if (lane == 0) sp_bump(sCL + col);
      };
      load_col(0);
      {
        // window pixel p of the item: image p / 100 of the quad, position (p % 100) / 10 - 1, % 10 - 1 inside it (the ring of
        // border pixels, images beyond the batch and channels beyond Cin read the zero line); the second input (1x1 shortcut)
        // is gathered in the same geometry and multiplied at the centre tap
        const int cin = second ? d.Cin2 : d.Cin, cs = second ? d.in2_cs : d.in_cs, co = second ? d.in2_co : d.in_co;
        const char* base = reinterpret_cast<const char*>(second ? d.in2 : d.in) + (((size_t)quad * kImgs * 64) * cs + co) * 4 + cc * 128;
        const int loff = second ? l_off2 : l_off1;
        const bool ch_ok = cc * KC + l_kg * 8 < cin;
        int lpx = l_px;  // (opaque: nothing derived from the lane id is hoisted out of the step loop - conv_sp_movers.inc)
        asm volatile("" : "+v"(lpx));
#pragma unroll
        for (int i = 0; i < NPW; ++i)
          if (i < nwin) {
            const int p = (pw + 4 * i) * 8 + lpx;
            const int img = (p * 41) >> 12, w = p - img * kIpix;      // p / 100 (exact for p < 512)
            const int py = (w * 205) >> 11, px = w - py * kWp;        // w / 10 (exact for w < 100)
            const int iy = py - 1, ix = px - 1;
            const bool ok = ch_ok && p < kNpix && quad * kImgs + img < d.N && (unsigned)iy < 8u && (unsigned)ix < 8u;
            const unsigned o = (unsigned)(((img * 64 + iy * 8 + ix) * cs) * 4 + loff);
            ww[i] = *reinterpret_cast<const u32x4*>(ok ? base + o : zero);
          }
      }
      load_col(1);
      if (k >= 1) sp_poll(sCR, 8u * (unsigned)k, d.fault);
      sp_wait_vm(nwin + np[1]);
      store_col(0);
      load_col(2);
      if (k >= 2) sp_poll(sWR + (k & 1), 8u * (unsigned)(k >> 1), d.fault);
      sp_wait_vm(np[1] + np[2]);
      {
        char* buf = sWin + (k & 1) * WBUF + lane * 16;
#pragma unroll
        for (int i = 0; i < NPW; ++i)
          if (i < nwin) *reinterpret_cast<u32x4*>(buf + (pw + 4 * i) * 1024) = ww[i];
        sp_wait_lds();
        if (lane == 0) sp_bump(sWL + (k & 1));
      }
      // (right behind a step that stored columns 1 / 2 a one-tap step polls too: the landed counters count bumps, not movers -
      //  conv_sp_movers.inc has the story)
      const bool after_fill = prev_fill;
      prev_fill = np[1] != 0 || np[2] != 0;
      if (k >= 1 && (np[1] || after_fill)) sp_poll(sCR + 1, 8u * (unsigned)k, d.fault);
      sp_wait_vm(np[2]);
      store_col(1);
      if (k >= 1 && (np[2] || after_fill)) sp_poll(sCR + 2, 8u * (unsigned)k, d.fault);
      sp_wait_vm(0);
      store_col(2);
    }
  } else {
    // ===================== consumers =====================
    const int rw = wid;            // image rw >> 1 of the item, pixel blocks 2 (rw & 1) + {0, 1} of it
    const int img = rw >> 1;
    // window pixel of (block r, lane lr) at tap (ky, kx): img * 100 + (2 blk + (lr >> 3) + ky) * 10 + (lr & 7) + kx
    int pr[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) pr[r] = img * kIpix + (2 * ((rw & 1) * 2 + r) + (lr >> 3)) * kWp + (lr & 7);
    auto win_frag = [&](const char* buf, int r, int q) __attribute__((always_inline)) {  // q = ky * 10 + kx: compile time
      const int p = pr[r] + q;
      const int pos = (kg + p) & 7;  // slot c of pixel p sits at position (c + p) & 7; the lo half is 4 slots further (mod 8)
      const char* line = buf + p * 128;
      return typename P::Frag{*reinterpret_cast<const bf16x8*>(line + (pos << 4)), *reinterpret_cast<const bf16x8*>(line + ((pos ^ 4) << 4))};
    };
    const char* wbase = sW + ((size_t)kg * BNB + lr) * 16;
    f32x4 acc[RPW][NT];
    typename P::Frag wf[3][NT];
    auto read_wf = [&](int col) __attribute__((always_inline)) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          wf[ky][t] = P::load(wbase, W_IMAGE, (size_t)(((col * 3 + ky) * 4 * BNB) + t * 16) * 16);
    };
    for (int k = 0; k < S; ++k) {
      if (++c == nck) c = 0;
      if (c == 0) {
        item_of(++ord, quad, n0);
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const bool second = HAS2 && c >= nchunks;
      const char* buf = sWin + (k & 1) * WBUF;
      const unsigned ltarget = 4u * (unsigned)(k + 1);
      sp_poll_lds(sWL + (k & 1), 4u * (unsigned)((k >> 1) + 1), d.fault);
      sp_poll_lds(sCL, ltarget, d.fault);
      read_wf(0);
      sp_wait_lds();
      if (lane == 0) sp_bump(sCR);
      if (second) {  // one tap, the window's centre
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          const typename P::Frag af = win_frag(buf, r, kWp + 1);
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
#pragma unroll
        for (int col = 0; col < 3; ++col) {
          if (col > 0) {
            sp_poll_lds(sCL + col, ltarget, d.fault);
            read_wf(col);
            sp_wait_lds();
            if (lane == 0) sp_bump(sCR + col);
          }
#pragma unroll
          for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              const typename P::Frag af = win_frag(buf, r, ky * kWp + col);
#pragma unroll
              for (int t = 0; t < NT; ++t) acc[r][t] = P::mma(wf[ky][t], af, acc[r][t]);
            }
        }
        sp_wait_lds();
        if (lane == 0) sp_bump(sWR + (k & 1));
      }
      if (c == nck - 1) {
        // epilogue: the quad's outputs as rows of 16 pixels (an 8 x 8 image = four such rows); this wave: image n, rows gb, gb + 1
        const int n = quad * kImgs + img;
        const int cg = n0;
        SpEpiConst kc;
        auto ld8 = [&](const float* p_, float (&v)[8]) __attribute__((always_inline)) {
          if (p_) {
            const float4 a = *reinterpret_cast<const float4*>(p_), b = *reinterpret_cast<const float4*>(p_ + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
          }
        };
        const int nn = min(n, d.N - 1);
        ld8(d.bias ? d.bias + cg + kg * 8 : nullptr, kc.bias);
        if (HAS2 && d.bias2) {
          float b2[8];
          ld8(d.bias2 + cg + kg * 8, b2);
#pragma unroll
          for (int j = 0; j < 8; ++j) kc.bias[j] += b2[j];
        }
        ld8(d.post_add ? d.post_add + (size_t)nn * d.post_cs + cg + kg * 8 : nullptr, kc.post);
        ld8(d.out2 ? d.post2 + (size_t)nn * d.post2_cs + cg + kg * 8 : nullptr, kc.post2);  // second output: value + post2 (x + temb)
        TapConv de = d;
        de.OH = d.N * 4; de.OW = 16; de.TH = d.N * 4; de.TW = 16;
        tile_epilogue_sp_pre<RPW, true>(de, acc, kc, 0, cg, n * 4 + (rw & 1) * 2, 0, 0, lr, kg);
      }
    }
  }
}

bool sp8_std3x3(const TapConv& d) {
  if (d.mode != 0 || d.ntaps != 9 || d.wtaps_total != 9 || d.in_stride != 1 || d.out_scale != 1) return false;
  for (int i = 0; i < 9; ++i)
    if (d.dy[i] != i / 3 - 1 || d.dx[i] != i % 3 - 1 || d.wtap[i] != i) return false;
  return true;
}

template <bool HAS2>
int sp8_launch(const TapConv& d, hipStream_t s) {
  auto kern = tapconv_sp8_kernel<HAS2>;
  static_assert(kLds <= 160 * 1024, "LDS budget");
  int num_cu = 0;
  {
    const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(kern), 160 * 1024, &num_cu);
    if (rc) return rc;
  }
  const int nchunks = d.Cin / 32, nchunks2 = HAS2 ? d.Cin2 / 32 : 0;
  const unsigned w_gimage = (unsigned)((size_t)nchunks * 9 * 4 * d.Cout * 16);
  const unsigned w2_gimage = (unsigned)((size_t)nchunks2 * 4 * d.Cout * 16);
  const long long nitems = (long long)((d.N + kImgs - 1) / kImgs) * (d.Cout / kBnb);
  long long blocks = num_cu;
  if (blocks > nitems) blocks = nitems;
  blocks = (blocks + 7) / 8 * 8;
  DRS_LAUNCH(kern, dim3((unsigned)blocks), dim3(768), kLds, s, d, nchunks, nchunks2, w_gimage, w2_gimage);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

}  // namespace

// 3x3 stride 1 over 8 x 8 SP images -> SP output; optional 1x1 second input of the same size.  DRS_SP8=0 keeps the lock-step kernel.
bool drs_tapconv_sp8_supported(const TapConv& d, int impl) {
  static const int env = getenv("DRS_SP8") ? atoi(getenv("DRS_SP8")) : 1;
  if (!env || impl != DRS_IMPL_MFMA_BF16X3) return false;
  if (!d.in || !d.in_sp || !sp8_std3x3(d) || !d.zero_line || !d.out || !d.out_sp) return false;
  if (d.H != 8 || d.W != 8 || d.TH != 8 || d.TW != 8 || d.OH != 8 || d.OW != 8) return false;
  if (d.gate || d.in_add || d.res || d.sigmoid || d.out_nchw || d.dual || d.fuse_out) return false;
  if (d.out2 && (!d.post2 || (d.out2_co & 31) || (d.out2_cs & 31) || (d.post2_cs & 3))) return false;
  if (d.Cin % 32 || d.Cout % 32 || (d.in_cs & 31) || (d.in_co & 31) || (d.out_cs & 31) || (d.out_co & 31)) return false;
  if (d.in2 && (!d.in2_sp || !d.w2 || d.Cin2 % 32 || (d.in2_cs & 31) || (d.in2_co & 31) || d.H2 != 8 || d.W2 != 8)) return false;
  if (d.post_add && (d.post_cs & 3)) return false;
  if ((long long)d.N * 64 * (d.in_cs > d.out_cs ? d.in_cs : d.out_cs) * 4 >= (1LL << 31)) return false;  // 32-bit lane offsets
  return true;
}

int drs_launch_tapconv_sp8(const TapConv& d, hipStream_t s) {
  if ((long long)d.N == 0) return DRS_OK;
  return d.in2 ? sp8_launch<true>(d, s) : sp8_launch<false>(d, s);
}
