// LDS-tiled implicit-GEMM tap-convolution on the gfx950 matrix cores.
//
// GEMM view (per launch):  D[cout][pixel] = sum_{tap, ci} W[tap][cout][ci] * X[pixel + tap][ci]
//   M = Cout (A operand = weights), N = output positions (B operand = activations), K = taps x Cin.
// Weights are the MFMA A operand so that the accumulator of a lane holds 4 CONSECUTIVE output channels of one
// pixel (C/D layout: col = lane&15 -> pixel, row = 4*(lane>>4)+reg -> cout): the epilogue reads bias / residual
// and writes the channels-last output as 16-byte vectors.
//
// Block = 256 threads = 4 waves; output patch = (4*RPW) rows x 16 columns of logical output positions,
// BN output channels.  Wave w owns rows [w*RPW, (w+1)*RPW) of the patch (one 16-pixel MFMA column block each) and
// all BN channels.  K loop over chunks of KC input channels:
//   stage  : the input window (patch + halo of all taps) of KC channels -> LDS, converted to the policy's operand
//            type; the chunk's weights for all taps and BN channels -> LDS (pre-converted at pack time);
//   compute: for every tap, MFMA over the window shifted by (dy, dx) - the 3x3 window is re-read from LDS, never
//            from HBM.
// LDS layout: "slot" = the 16 bytes a lane feeds to one MFMA operand register group (8 x 16-bit or 4 x f32 = the
// lane's K-group).  Activations: [image][kgroup(4)][window pixel] slots; weights: [image][tap][kgroup(4)][BN]
// slots.  16 lanes with consecutive pixels (or channels) read 256 contiguous bytes: conflict-free ds_read_b128.
// No intra-block software pipeline: the footprint is sized for 2 blocks per CU, whose stage/compute phases overlap.
#include <stdlib.h>

#include "drs_common.h"

#include "mfma_policy.h"

template <class P, int BN, int RPW>
__global__ __launch_bounds__(256, 2) void tapconv_mfma_kernel(TapConv d, MfmaGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KC = 4 * P::SLOT_CH;
  constexpr int TH = 4 * RPW, TW = 16, NT = BN / 16;
  constexpr int A_ITERS = (RPW == 4) ? 6 : 9;            // window slots per thread (stride 1: <= 384 px, stride 2: <= 576 px)
  constexpr int W_ITERS = (DRS_MAX_TAPS * 4 * BN + 255) / 256;
  constexpr int V4 = P::SLOT_CH / 4;                     // float4 loads per activation slot
  char* sA = smem;
  char* sW = smem + (size_t)P::IMAGES * g.a_image;
  // tap tables live in LDS: indexing the by-value kernel argument with a runtime tap would go through scratch
  int* sTapOff = reinterpret_cast<int*>(smem + (size_t)P::IMAGES * (g.a_image + g.w_image));  // window slot offset
  int* sTapW = sTapOff + 16;                                                                   // weight tap index

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  // XCD-aware patch order: hardware deals consecutive block ids round-robin over the 8 XCDs; give each XCD (= each
  // private L2) a contiguous run of patches so neighbouring halos and the layer's weights hit in L2.
  int bid = blockIdx.x;
  {
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  }
  const int tile_x = bid % g.tiles_x;
  bid /= g.tiles_x;
  const int tile_y = bid % g.tiles_y;
  const int n = bid / g.tiles_y;
  const int n0 = blockIdx.y * BN;
  const int ty0 = tile_y * TH, tx0 = tile_x * TW;
  const int iy0 = ty0 * d.in_stride + g.dy_min, ix0 = tx0 * d.in_stride + g.dx_min;
  const int nslots = g.IH * g.IW * 4;
  const int wslots = d.ntaps * 4 * BN;

  f32x4 acc[RPW][NT];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- per-thread staging descriptors (chunk independent).  Every address is clamped to a legal one and validity is
  //      a bit mask, so the loads below are straight-line code (no exec-masked branches, no serialising waits). ----
  // activation slot s = (window pixel p, k-group q), q fastest: 4 lanes read 4*SLOT_CH consecutive channels of a pixel
  int a_base[A_ITERS];  // element offset of channel 0 of the (clamped) pixel inside image n
  unsigned a_ok = 0;    // bit it: the slot's pixel lies inside the image
  const int aq = tid & 3;  // 256 % 4 == 0: the k-group of a thread's slots does not depend on `it`
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int s = tid + it * 256;
    const int p = s >> 2;
    const int py = p / g.IW, px = p - py * g.IW;
    const int iy = iy0 + py, ix = ix0 + px;
    const bool ok = s < nslots && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
    const int iyc = min(max(iy, 0), d.H - 1), ixc = min(max(ix, 0), d.W - 1);
    a_base[it] = (iyc * d.W + ixc) * d.in_cs + d.in_co;
    a_ok |= (ok ? 1u : 0u) << it;
  }
  int w_goff[W_ITERS];  // byte offset of the slot inside one chunk of one global weight image (clamped)
#pragma unroll
  for (int it = 0; it < W_ITERS; ++it) {
    const int s = min(tid + it * 256, wslots - 1);
    const int nn = s % BN, q = (s / BN) & 3, tap = s / (BN * 4);
    w_goff[it] = ((sTapW[tap] * 4 + q) * d.Cout + n0 + nn) * 16;
  }
  const float* in_n = d.in + (size_t)n * d.H * d.W * d.in_cs;
  const bool has_add = d.in_add != nullptr;
  const float* addp = has_add ? d.in_add + (size_t)n * d.in_add_cs : nullptr;
  const char* wg = reinterpret_cast<const char*>(d.w);
  const size_t w_chunk = (size_t)d.wtaps_total * 4 * d.Cout * 16;

  float4 areg[A_ITERS][V4];
  float4 addreg[V4];
  u32x4 wreg[W_ITERS][P::IMAGES];

  auto load_chunk = [&](int c) {  // global -> registers, everything issued back to back
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const int ch = min(c * KC + aq * P::SLOT_CH + 4 * v, d.Cin - 4);
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) areg[it][v] = *reinterpret_cast<const float4*>(in_n + a_base[it] + ch);
      if (has_add) addreg[v] = *reinterpret_cast<const float4*>(addp + ch);
    }
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it)
#pragma unroll
      for (int im = 0; im < P::IMAGES; ++im)
        wreg[it][im] = *reinterpret_cast<const u32x4*>(wg + (size_t)im * g.w_gimage + (size_t)c * w_chunk + w_goff[it]);
  };
  auto store_chunk = [&](int c) {  // registers -> LDS (operand conversion happens here)
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      const int s = tid + it * 256;
      const bool pix_ok = (a_ok >> it) & 1u;
      float x[P::SLOT_CH];
#pragma unroll
      for (int v = 0; v < V4; ++v) {
        const bool ok = pix_ok && (c * KC + aq * P::SLOT_CH + 4 * v < d.Cin);
        float4 a = areg[it][v];
        if (has_add) {  // per-(n, ci) input add applies to in-image pixels only (zero padding stays zero)
          a.x += addreg[v].x; a.y += addreg[v].y; a.z += addreg[v].z; a.w += addreg[v].w;
        }
        x[4 * v] = ok ? a.x : 0.f; x[4 * v + 1] = ok ? a.y : 0.f; x[4 * v + 2] = ok ? a.z : 0.f; x[4 * v + 3] = ok ? a.w : 0.f;
      }
      if (s < nslots) P::cvt_store(sA, g.a_image, (size_t)aq * g.a_plane + (size_t)(s >> 2) * 16, x);
    }
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it)
      if (tid + it * 256 < wslots) {
#pragma unroll
        for (int im = 0; im < P::IMAGES; ++im)
          *reinterpret_cast<u32x4*>(sW + (size_t)im * g.w_image + (size_t)(tid + it * 256) * 16) = wreg[it][im];
      }
  };

  if (!(g.debug & 4)) load_chunk(0);
  for (int c = 0; c < g.nchunks; ++c) {
    if (c) __syncthreads();  // everyone finished reading the previous chunk's LDS image
    if (!(g.debug & 2)) store_chunk(c);
    __syncthreads();
    if (c + 1 < g.nchunks && !(g.debug & 4)) load_chunk(c + 1);  // next chunk's loads fly while this one is multiplied
    if (!(g.debug & 1))
      for (int tap = 0; tap < d.ntaps; ++tap) {
        typename P::Frag wf[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
          wf[t] = P::load(sW, g.w_image, ((size_t)(tap * 4 + kg) * BN + t * 16 + lr) * 16);
        const int toff = sTapOff[tap];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          const int py = (wave * RPW + r) * d.in_stride;
          const int px = lr * d.in_stride;
          const typename P::Frag af = P::load(sA, g.a_image, (size_t)kg * g.a_plane + (size_t)(py * g.IW + px) * 16 + toff);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][t] = P::mma(wf[t], af, acc[r][t]);
        }
      }
  }

  // ---- epilogue ----
  // MFMA layout: lane (lr, kg) holds channels t*16 + kg*4 .. +3 of pixel lr for every n-tile t.  Pairs of n-tiles are
  // exchanged between lanes lr and lr^8 (one DPP row rotate) so that each store instruction covers 8 pixels x 32
  // channels = full 128-byte lines: pass h=0 writes pixels 0..7 of the row, pass h=1 pixels 8..15; lanes lr < 8 carry
  // the even n-tile of the pair, lanes lr >= 8 the odd one.  Residual / gate loads use the same mapping and are all
  // issued before the arithmetic (addresses clamped, stores predicated).
  if (g.debug & 8) return;
  constexpr int NP = NT / 2;
  const bool lo = lr < 8;
  const int pl = lr & 7;
  f32x4 val[RPW][2][NP];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      f32x4 mine, theirs;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float send = lo ? acc[r][2 * pr + 1][j] : acc[r][2 * pr][j];
        theirs[j] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128, 0xf, 0xf, false));
        mine[j] = lo ? acc[r][2 * pr][j] : acc[r][2 * pr + 1][j];
      }
      val[r][0][pr] = lo ? mine : theirs;   // pass 0: pixel pl     (lo lanes: own even tile; hi lanes: partner's odd tile)
      val[r][1][pr] = lo ? theirs : mine;   // pass 1: pixel pl + 8 (lo lanes: partner's even tile; hi lanes: own odd tile)
    }
  const int csel = (lo ? 0 : 16) + kg * 4;  // channel offset inside a pair of n-tiles
  bool valid[RPW][2];
  size_t opix[RPW][2];
  int oyx[RPW][2][2];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ty = ty0 + wave * RPW + r, tx = tx0 + pl + 8 * h;
      valid[r][h] = ty < d.TH && tx < d.TW;
      const int oy = min(ty, d.TH - 1) * d.out_scale + d.out_oy, ox = min(tx, d.TW - 1) * d.out_scale + d.out_ox;
      oyx[r][h][0] = oy; oyx[r][h][1] = ox;
      opix[r][h] = ((size_t)n * d.OH + oy) * d.OW + ox;
    }
  float4 bias4[NP], post4[NP];
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) {
    const int co = n0 + pr * 32 + csel;
    bias4[pr] = d.bias ? *reinterpret_cast<const float4*>(d.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    post4[pr] = d.post_add ? *reinterpret_cast<const float4*>(d.post_add + (size_t)n * d.post_cs + co)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float gv[RPW][2];
  float4 res4[RPW][2][NP];
  if (d.gate) {
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        gv[r][h] = d.gate[((size_t)n * (d.OH >> 1) + (oyx[r][h][0] >> 1)) * (d.OW >> 1) + (oyx[r][h][1] >> 1)];
  }
  if (d.res) {
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const size_t rp = d.res_bstride_zero ? ((size_t)oyx[r][h][0] * d.OW + oyx[r][h][1]) : opix[r][h];
#pragma unroll
        for (int pr = 0; pr < NP; ++pr)
          res4[r][h][pr] = *reinterpret_cast<const float4*>(d.res + rp * d.res_cs + d.res_co + n0 + pr * 32 + csel);
      }
  }
  float4 fw[4][NP];
  if (d.fuse_out) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int pr = 0; pr < NP; ++pr)
        fw[j][pr] = *reinterpret_cast<const float4*>(d.fuse_w + (size_t)min(j, d.fuse_dim - 1) * d.Cout + n0 + pr * 32 + csel);
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float fz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) {
        f32x4 v = val[r][h][pr];
        if (d.gate) v *= gv[r][h];
        v[0] += bias4[pr].x; v[1] += bias4[pr].y; v[2] += bias4[pr].z; v[3] += bias4[pr].w;
        if (d.relu_pre) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        v[0] += post4[pr].x; v[1] += post4[pr].y; v[2] += post4[pr].z; v[3] += post4[pr].w;
        if (d.res) {
          v[0] += res4[r][h][pr].x; v[1] += res4[r][h][pr].y; v[2] += res4[r][h][pr].z; v[3] += res4[r][h][pr].w;
        }
        if (d.relu_post) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (d.out && valid[r][h])
          *reinterpret_cast<float4*>(d.out + opix[r][h] * d.out_cs + d.out_co + n0 + pr * 32 + csel) =
              make_float4(v[0], v[1], v[2], v[3]);
        if (d.fuse_out) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            fz[j] += v[0] * fw[j][pr].x + v[1] * fw[j][pr].y + v[2] * fw[j][pr].z + v[3] * fw[j][pr].w;
        }
      }
      if (d.fuse_out) {  // y[j] = sum over the pixel's 32 channels = 8 lanes: lr^8 (n-tile of the pair) x 4 k-groups
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          fz[j] += __shfl_xor(fz[j], 8);
          fz[j] += __shfl_xor(fz[j], 16);
          fz[j] += __shfl_xor(fz[j], 32);
        }
        if (valid[r][h] && lo && kg == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < d.fuse_dim)
              d.fuse_out[(((size_t)n * d.fuse_dim + j) * d.OH + oyx[r][h][0]) * d.OW + oyx[r][h][1]] = fz[j] + d.fuse_b[j];
        }
      }
    }
}

// ---- host side --------------------------------------------------------------------------------------------------
static constexpr int kLdsLimit = 160 * 1024;

static int slot_ch(int impl) { return impl == DRS_IMPL_MFMA_F32 ? 4 : 8; }
static int images(int impl) { return impl == DRS_IMPL_MFMA_BF16X3 ? 2 : 1; }
static int pick_bn(const TapConv& d, int impl) {
  if (impl == DRS_IMPL_MFMA_BF16X3) return 32;
  return d.Cout % 64 == 0 ? 64 : 32;
}
static int pick_rpw(const TapConv& d) { return d.in_stride == 1 ? 4 : 2; }

static bool geom(const TapConv& d, int impl, MfmaGeom* g, int* bn, int* rpw, size_t* lds) {
  *bn = pick_bn(d, impl);
  *rpw = pick_rpw(d);
  int dy0 = 1 << 30, dy1 = -(1 << 30), dx0 = 1 << 30, dx1 = -(1 << 30);
  for (int i = 0; i < d.ntaps; ++i) {
    dy0 = d.dy[i] < dy0 ? d.dy[i] : dy0; dy1 = d.dy[i] > dy1 ? d.dy[i] : dy1;
    dx0 = d.dx[i] < dx0 ? d.dx[i] : dx0; dx1 = d.dx[i] > dx1 ? d.dx[i] : dx1;
  }
  const int TH = 4 * *rpw, TW = 16;
  g->dy_min = dy0; g->dx_min = dx0;
  g->IH = (TH - 1) * d.in_stride + (dy1 - dy0) + 1;
  g->IW = (TW - 1) * d.in_stride + (dx1 - dx0) + 1;
  g->tiles_x = drs_cdiv(d.TW, TW);
  g->tiles_y = drs_cdiv(d.TH, TH);
  const int KC = 4 * slot_ch(impl);
  g->nchunks = drs_cdiv(d.Cin, KC);
  g->a_plane = (g->IH * g->IW * 16 + 255) / 256 * 256;
  g->a_image = 4 * g->a_plane;
  g->w_image = d.ntaps * 4 * *bn * 16;
  g->w_gimage = g->nchunks * d.wtaps_total * 4 * d.Cout * 16;
  *lds = (size_t)images(impl) * ((size_t)g->a_image + g->w_image) + 128;  // + tap tables
  static const int dbg = getenv("DRS_DEBUG_FLAGS") ? atoi(getenv("DRS_DEBUG_FLAGS")) : 0;
  g->debug = dbg;
  return *lds <= (size_t)kLdsLimit;
}

bool drs_tapconv_mfma_supported(const TapConv& d, int impl) {
  if (impl != DRS_IMPL_MFMA_F32 && impl != DRS_IMPL_MFMA_F16 && impl != DRS_IMPL_MFMA_BF16X3) return false;
  if (d.Cout % 32 != 0 || d.Cin % 4 != 0) return false;
  if (d.out_nchw || d.sigmoid) return false;
  if (d.ntaps < 1) return false;
  if (d.fuse_out && (d.Cout != 32 || d.fuse_dim > 4)) return false;
  if (d.in == nullptr) return true;  // shape-only probe (weight packing): spatial details decided per launch
  if ((d.in_cs & 3) || (d.in_co & 3) || (d.out_cs & 3) || (d.out_co & 3)) return false;
  if (d.res && ((d.res_cs & 3) || (d.res_co & 3))) return false;
  if (d.post_add && (d.post_cs & 3)) return false;
  if (d.in_add && (d.in_add_cs & 3)) return false;
  if (d.in_stride != 1 && d.in_stride != 2) return false;
  MfmaGeom g; int bn, rpw; size_t lds;
  return geom(d, impl, &g, &bn, &rpw, &lds);
}

template <class P, int BN, int RPW>
static int launch_t(const TapConv& d, const MfmaGeom& g, size_t lds, hipStream_t s) {
  auto kern = tapconv_mfma_kernel<P, BN, RPW>;
  static bool attr_done = false;  // per instantiation
  if (!attr_done) {
    DRS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kLdsLimit));
    attr_done = true;
  }
  dim3 grid((unsigned)((size_t)d.N * g.tiles_x * g.tiles_y), d.Cout / BN);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, d, g);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

template <class P>
static int launch_p(const TapConv& d, const MfmaGeom& g, int bn, int rpw, size_t lds, hipStream_t s) {
  if constexpr (P::IMAGES == 1) {  // the split-bf16 policy always runs BN = 32 (LDS budget for 2 blocks per CU)
    if (bn == 64 && rpw == 4) return launch_t<P, 64, 4>(d, g, lds, s);
    if (bn == 64 && rpw == 2) return launch_t<P, 64, 2>(d, g, lds, s);
  }
  if (bn == 32 && rpw == 4) return launch_t<P, 32, 4>(d, g, lds, s);
  if (bn == 32 && rpw == 2) return launch_t<P, 32, 2>(d, g, lds, s);
  DrsErr::set("tapconv_mfma: no kernel for BN=%d RPW=%d", bn, rpw);
  return DRS_ERR_SHAPE;
}

int drs_launch_tapconv_mfma(const TapConv& d, int impl, hipStream_t s) {
  DRS_REQUIRE(d.in && d.w && d.out, DRS_ERR_ARG, "tapconv_mfma: null tensor");
  DRS_REQUIRE(drs_tapconv_mfma_supported(d, impl), DRS_ERR_SHAPE, "tapconv_mfma: unsupported shape");
  if ((size_t)d.N * d.TH * d.TW == 0) return DRS_OK;
  MfmaGeom g; int bn, rpw; size_t lds;
  static const bool no_ws = getenv("DRS_NO_WS") != nullptr;
  if (!no_ws && drs_tapconv_mfma_ws_geom(d, impl, &g, &bn, &lds)) return drs_launch_tapconv_mfma_ws(d, impl, g, bn, lds, s);
  geom(d, impl, &g, &bn, &rpw, &lds);
  if (impl == DRS_IMPL_MFMA_F32) return launch_p<PolicyF32>(d, g, bn, rpw, lds, s);
  if (impl == DRS_IMPL_MFMA_F16) return launch_p<PolicyF16>(d, g, bn, rpw, lds, s);
  return launch_p<PolicyBF16X3>(d, g, bn, rpw, lds, s);
}

// ---- weight packing for the MFMA kernels ------------------------------------------------------------------------
// dst image layout: [chunk][tap][kgroup(4)][Cout][SLOT_CH] elements, ci = chunk*KC + kgroup*SLOT_CH + j, zero-padded
// beyond Cin; BatchNorm (eval) folded like pack_conv_kernel.
template <class P>
__global__ void pack_conv_mfma_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                      char* __restrict__ dst_w, float* __restrict__ dst_b, int Cout, int Cin, int taps,
                                      int transposed, int nchunks, size_t image_bytes) {
  constexpr int KC = 4 * P::SLOT_CH;
  const size_t nslots = (size_t)nchunks * taps * 4 * Cout;
  for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(s % Cout);
    const int q = (int)((s / Cout) & 3);
    const int tap = (int)((s / ((size_t)Cout * 4)) % taps);
    const int c = (int)(s / ((size_t)Cout * 4 * taps));
    float sc = 1.f;
    if (gamma) sc = gamma[co] / sqrtf(rvar[co] + eps);
    float x[P::SLOT_CH];
#pragma unroll
    for (int j = 0; j < P::SLOT_CH; ++j) {
      const int ci = c * KC + q * P::SLOT_CH + j;
      float v = 0.f;
      if (ci < Cin) {
        const size_t src = transposed ? (((size_t)ci * Cout + co) * taps + tap) : (((size_t)co * Cin + ci) * taps + tap);
        v = w[src];
        if (gamma) v *= sc;
      }
      x[j] = v;
    }
    P::cvt_store(dst_w, image_bytes, s * 16, x);
  }
  if (blockIdx.x == 0) {
    for (int co = threadIdx.x; co < Cout; co += blockDim.x) {
      float bb = b ? b[co] : 0.f;
      if (gamma) {
        const float sc = gamma[co] / sqrtf(rvar[co] + eps);
        bb = (bb - rmean[co]) * sc + beta[co];
      }
      dst_b[co] = bb;
    }
  }
}

size_t drs_pack_conv_mfma_bytes(int Cout, int Cin, int taps, int impl) {
  const int KC = 4 * slot_ch(impl);
  return (size_t)images(impl) * drs_cdiv(Cin, KC) * taps * 4 * Cout * 16;
}

int drs_launch_pack_conv_mfma(const float* w, const float* b, const float* gamma, const float* beta, const float* rmean,
                              const float* rvar, float eps, void* dst_w, float* dst_b, int Cout, int Cin, int taps,
                              int transposed, int impl, hipStream_t s) {
  const int KC = 4 * slot_ch(impl);
  const int nchunks = drs_cdiv(Cin, KC);
  const size_t image = (size_t)nchunks * taps * 4 * Cout * 16;
  const size_t nslots = image / 16;
  int blocks = (int)((nslots + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
#define DRS_PACK(P)                                                                                                   \
  hipLaunchKernelGGL(pack_conv_mfma_kernel<P>, dim3(blocks), dim3(256), 0, s, w, b, gamma, beta, rmean, rvar, eps,    \
                     (char*)dst_w, dst_b, Cout, Cin, taps, transposed, nchunks, image)
  if (impl == DRS_IMPL_MFMA_F32) DRS_PACK(PolicyF32);
  else if (impl == DRS_IMPL_MFMA_F16) DRS_PACK(PolicyF16);
  else DRS_PACK(PolicyBF16X3);
#undef DRS_PACK
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
