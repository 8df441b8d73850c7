// Persistent, wave-specialised, double-buffered implicit-GEMM tap-convolution (stride-1 flavours).
//
// Same GEMM view, LDS layouts and operand policies as conv_mfma.hip, but every phase is overlapped by construction:
//   * one 512-thread block per CU (8 waves, 2 per SIMD): waves 0-3 are CONSUMERS (LDS fragment reads + MFMA +
//     epilogue), waves 4-7 are PRODUCERS (global loads, operand conversion, LDS stores);
//   * two LDS stages: while the consumers multiply step k out of stage k&1, the producers fill stage (k+1)&1 with
//     step k+1 and already have the global loads of step k+2 in flight in registers; ONE barrier per step;
//   * blocks are persistent: a block walks over its share of (patch, channel-group) items, so the epilogue stores of
//     one item overlap with the producers' loads of the next, and block start-up is paid once per CU;
//   * XCD-aware item order: the 32 blocks that share an XCD (blockIdx % 8) sweep one contiguous eighth of the
//     patches, 32 consecutive patches at a time, so halo pixels and the layer's weights are served by that XCD's L2.
// A "step" is one K-chunk (KC input channels x all taps) of one item.
#include <stdlib.h>

#include "mfma_policy.h"

template <class P, int BN>
__global__ __launch_bounds__(512, 2) void tapconv_mfma_ws_kernel(TapConv d, MfmaGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KC = 4 * P::SLOT_CH;
  constexpr int RPW = 4, TH = 16, TW = 16, NT = BN / 16, NP = NT / 2;
  constexpr int A_ITERS = 6;  // window slots per producer thread (<= 384 window pixels x 4 k-groups / 256)
  constexpr int W_ITERS = (DRS_MAX_TAPS * 4 * BN + 255) / 256;
  constexpr int V4 = P::SLOT_CH / 4;
  const int stage_bytes = P::IMAGES * (g.a_image + g.w_image);
  int* sTapOff = reinterpret_cast<int*>(smem + 2 * (size_t)stage_bytes);
  int* sTapW = sTapOff + 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  if (tid < DRS_MAX_TAPS) {
    int dyv = 0, dxv = 0, wt = 0;
#pragma unroll
    for (int i = 0; i < DRS_MAX_TAPS; ++i)
      if (i == tid) { dyv = d.dy[i]; dxv = d.dx[i]; wt = d.wtap[i]; }
    sTapOff[tid] = ((dyv - g.dy_min) * g.IW + (dxv - g.dx_min)) * 16;
    sTapW[tid] = wt;
  }
  __syncthreads();

  // ---- this block's share of the items ----
  const int ngroups = d.Cout / BN;
  const int nitems = d.N * g.tiles_y * g.tiles_x * ngroups;
  const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
  const int per = (nitems + 7) >> 3;                       // items per XCD range
  const int lo_item = xcd * per, hi_item = min(nitems, lo_item + per);
  const int my_items = (hi_item - lo_item - j8 + nb8 - 1) > 0 ? (hi_item - lo_item - j8 + nb8 - 1) / nb8 : 0;
  const int S = my_items * g.nchunks;                      // steps of this block
  const int nslots = g.IH * g.IW * 4;
  const int wslots = d.ntaps * 4 * BN;

  auto item_of = [&](int ordinal, int& n, int& ty0, int& tx0, int& n0) {
    int it = lo_item + ordinal * nb8 + j8;
    const int ng = it % ngroups;
    it /= ngroups;
    const int tile_x = it % g.tiles_x;
    it /= g.tiles_x;
    const int tile_y = it % g.tiles_y;
    n = it / g.tiles_y;
    ty0 = tile_y * TH;
    tx0 = tile_x * TW;
    n0 = ng * BN;
  };

  if (producer) {
    // =========================================== PRODUCERS ===========================================
    const int ptid = tid - 256;
    const int aq = ptid & 3;
    int a_py[A_ITERS], a_px[A_ITERS];
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      const int p = min((ptid + it * 256) >> 2, g.IH * g.IW - 1);
      a_py[it] = p / g.IW;
      a_px[it] = p - a_py[it] * g.IW;
    }
    int w_soff[W_ITERS];  // byte offset of the slot inside one chunk of one weight image, without the channel group
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it) {
      const int s = min(ptid + it * 256, wslots - 1);
      const int nn = s % BN, q = (s / BN) & 3, tap = s / (BN * 4);
      w_soff[it] = ((sTapW[tap] * 4 + q) * d.Cout + nn) * 16;
    }
    const char* wg = reinterpret_cast<const char*>(d.w);
    const size_t w_chunk = (size_t)d.wtaps_total * 4 * d.Cout * 16;
    const bool has_add = d.in_add != nullptr;

    float4 areg[A_ITERS][V4];
    float4 addreg[V4];
    u32x4 wreg[W_ITERS][P::IMAGES];
    unsigned a_ok = 0;  // validity of the slots held in areg
    int held_c = 0;     // chunk index of the data held in registers

    auto load_step = [&](int k) {  // global -> registers for step k
      const int c = k % g.nchunks;
      int n, ty0, tx0, n0;
      item_of(k / g.nchunks, n, ty0, tx0, n0);
      const int iy0 = ty0 * d.in_stride + g.dy_min, ix0 = tx0 * d.in_stride + g.dx_min;
      const float* in_n = d.in + (size_t)n * d.H * d.W * d.in_cs;
      a_ok = 0;
      held_c = c;
      int base[A_ITERS];
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        const int iy = iy0 + a_py[it], ix = ix0 + a_px[it];
        const bool ok = (ptid + it * 256) < nslots && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
        const int iyc = min(max(iy, 0), d.H - 1), ixc = min(max(ix, 0), d.W - 1);
        base[it] = (iyc * d.W + ixc) * d.in_cs + d.in_co;
        a_ok |= (ok ? 1u : 0u) << it;
      }
#pragma unroll
      for (int v = 0; v < V4; ++v) {
        const int ch = min(c * KC + aq * P::SLOT_CH + 4 * v, d.Cin - 4);
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) areg[it][v] = *reinterpret_cast<const float4*>(in_n + base[it] + ch);
        if (has_add) addreg[v] = *reinterpret_cast<const float4*>(d.in_add + (size_t)n * d.in_add_cs + ch);
      }
#pragma unroll
      for (int it = 0; it < W_ITERS; ++it)
#pragma unroll
        for (int im = 0; im < P::IMAGES; ++im)
          wreg[it][im] = *reinterpret_cast<const u32x4*>(wg + (size_t)im * g.w_gimage + (size_t)c * w_chunk +
                                                         (size_t)n0 * 16 + w_soff[it]);
    };
    auto store_step = [&](int stage) {  // registers -> LDS stage (operand conversion happens here)
      char* sA = smem + (size_t)stage * stage_bytes;
      char* sW = sA + (size_t)P::IMAGES * g.a_image;
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        const int s = ptid + it * 256;
        const bool pix_ok = (a_ok >> it) & 1u;
        float x[P::SLOT_CH];
#pragma unroll
        for (int v = 0; v < V4; ++v) {
          const bool ok = pix_ok && (held_c * KC + aq * P::SLOT_CH + 4 * v < d.Cin);
          float4 a = areg[it][v];
          if (has_add) {  // per-(n, ci) input add: in-image pixels only, zero padding stays zero
            a.x += addreg[v].x; a.y += addreg[v].y; a.z += addreg[v].z; a.w += addreg[v].w;
          }
          x[4 * v] = ok ? a.x : 0.f; x[4 * v + 1] = ok ? a.y : 0.f; x[4 * v + 2] = ok ? a.z : 0.f; x[4 * v + 3] = ok ? a.w : 0.f;
        }
        if (s < nslots) P::cvt_store(sA, g.a_image, (size_t)aq * g.a_plane + (size_t)(s >> 2) * 16, x);
      }
#pragma unroll
      for (int it = 0; it < W_ITERS; ++it)
        if (ptid + it * 256 < wslots) {
#pragma unroll
          for (int im = 0; im < P::IMAGES; ++im)
            *reinterpret_cast<u32x4*>(sW + (size_t)im * g.w_image + (size_t)(ptid + it * 256) * 16) = wreg[it][im];
        }
    };

    if (S > 0) {
      load_step(0);
      store_step(0);
      if (S > 1) load_step(1);
    }
    __syncthreads();
    for (int k = 0; k < S; ++k) {
      if (k + 1 < S) {
        store_step((k + 1) & 1);
        if (k + 2 < S) load_step(k + 2);
      }
      __syncthreads();
    }
  } else {
    // =========================================== CONSUMERS ===========================================
    const int lr = lane & 15, kg = lane >> 4;
    const bool lo = lr < 8;
    const int pl = lr & 7;
    const int csel = (lo ? 0 : 16) + kg * 4;
    f32x4 acc[RPW][NT];
    __syncthreads();  // stage 0 filled
    for (int k = 0; k < S; ++k) {
      const int c = k % g.nchunks;
      if (c == 0) {
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const char* sA = smem + (size_t)(k & 1) * stage_bytes;
      const char* sW = sA + (size_t)P::IMAGES * g.a_image;
      if (!(g.debug & 1))
        for (int tap = 0; tap < d.ntaps; ++tap) {
          typename P::Frag wf[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t)
            wf[t] = P::load(sW, g.w_image, ((size_t)(tap * 4 + kg) * BN + t * 16 + lr) * 16);
          const int toff = sTapOff[tap];
#pragma unroll
          for (int r = 0; r < RPW; ++r) {
            const typename P::Frag af =
                P::load(sA, g.a_image, (size_t)kg * g.a_plane + (size_t)((wave * RPW + r) * g.IW + lr) * 16 + toff);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[r][t] = P::mma(wf[t], af, acc[r][t]);
          }
        }
      if (c == g.nchunks - 1 && !(g.debug & 8)) {
        // ---- epilogue of the item (same lane exchange / full-line stores as conv_mfma.hip) ----
        int n, ty0, tx0, n0;
        item_of(k / g.nchunks, n, ty0, tx0, n0);
        float4 bias4[NP], post4[NP];
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
          const int co = n0 + pr * 32 + csel;
          bias4[pr] = d.bias ? *reinterpret_cast<const float4*>(d.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
          post4[pr] = d.post_add ? *reinterpret_cast<const float4*>(d.post_add + (size_t)n * d.post_cs + co)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 fw[4][NP];
        if (d.fuse_out) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pr = 0; pr < NP; ++pr)
              fw[j][pr] = *reinterpret_cast<const float4*>(d.fuse_w + (size_t)min(j, d.fuse_dim - 1) * d.Cout + n0 +
                                                           pr * 32 + csel);
        }
        // rows are processed in groups so that all residual / gate loads of a group are in flight together while
        // the register footprint stays bounded (BN = 64 keeps 2 rows live, BN = 32 all 4)
        constexpr int RG = (NT == 2) ? 4 : 2;
#pragma unroll
        for (int rg = 0; rg < RPW; rg += RG) {
          bool valid[RG][2];
          size_t opix[RG][2];
          int oyx[RG][2][2];
          float gv[RG][2];
          float4 res4[RG][2][NP];
#pragma unroll
          for (int rr = 0; rr < RG; ++rr)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int ty = ty0 + wave * RPW + rg + rr, tx = tx0 + pl + 8 * h;
              valid[rr][h] = ty < d.TH && tx < d.TW;
              const int oy = min(ty, d.TH - 1) * d.out_scale + d.out_oy, ox = min(tx, d.TW - 1) * d.out_scale + d.out_ox;
              oyx[rr][h][0] = oy; oyx[rr][h][1] = ox;
              opix[rr][h] = ((size_t)n * d.OH + oy) * d.OW + ox;
              if (d.gate) gv[rr][h] = d.gate[((size_t)n * (d.OH >> 1) + (oy >> 1)) * (d.OW >> 1) + (ox >> 1)];
              if (d.res) {
                const size_t rp = d.res_bstride_zero ? ((size_t)oy * d.OW + ox) : opix[rr][h];
#pragma unroll
                for (int pr = 0; pr < NP; ++pr)
                  res4[rr][h][pr] = *reinterpret_cast<const float4*>(d.res + rp * d.res_cs + d.res_co + n0 + pr * 32 + csel);
              }
            }
#pragma unroll
          for (int rr = 0; rr < RG; ++rr) {
            const int r = rg + rr;
            f32x4 val[2][NP];
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
              f32x4 mine, theirs;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float send = lo ? acc[r][2 * pr + 1][j] : acc[r][2 * pr][j];
                theirs[j] = __builtin_bit_cast(
                    float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128, 0xf, 0xf, false));
                mine[j] = lo ? acc[r][2 * pr][j] : acc[r][2 * pr + 1][j];
              }
              val[0][pr] = lo ? mine : theirs;
              val[1][pr] = lo ? theirs : mine;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              float fz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int pr = 0; pr < NP; ++pr) {
                f32x4 v = val[h][pr];
                if (d.gate) v *= gv[rr][h];
                v[0] += bias4[pr].x; v[1] += bias4[pr].y; v[2] += bias4[pr].z; v[3] += bias4[pr].w;
                if (d.relu_pre) {
#pragma unroll
                  for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                v[0] += post4[pr].x; v[1] += post4[pr].y; v[2] += post4[pr].z; v[3] += post4[pr].w;
                if (d.res) {
                  v[0] += res4[rr][h][pr].x; v[1] += res4[rr][h][pr].y; v[2] += res4[rr][h][pr].z; v[3] += res4[rr][h][pr].w;
                }
                if (d.relu_post) {
#pragma unroll
                  for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                if (d.out && valid[rr][h])
                  *reinterpret_cast<float4*>(d.out + opix[rr][h] * d.out_cs + d.out_co + n0 + pr * 32 + csel) =
                      make_float4(v[0], v[1], v[2], v[3]);
                if (d.fuse_out) {
#pragma unroll
                  for (int j = 0; j < 4; ++j)
                    fz[j] += v[0] * fw[j][pr].x + v[1] * fw[j][pr].y + v[2] * fw[j][pr].z + v[3] * fw[j][pr].w;
                }
              }
              if (d.fuse_out) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  fz[j] += __shfl_xor(fz[j], 8);
                  fz[j] += __shfl_xor(fz[j], 16);
                  fz[j] += __shfl_xor(fz[j], 32);
                }
                if (valid[rr][h] && lo && kg == 0) {
#pragma unroll
                  for (int j = 0; j < 4; ++j)
                    if (j < d.fuse_dim)
                      d.fuse_out[(((size_t)n * d.fuse_dim + j) * d.OH + oyx[rr][h][0]) * d.OW + oyx[rr][h][1]] =
                          fz[j] + d.fuse_b[j];
                }
              }
            }
          }
        }
      }
      __syncthreads();
    }
  }
}

// ---- host side --------------------------------------------------------------------------------------------------
static constexpr int kLdsLimit = 160 * 1024;

bool drs_tapconv_mfma_ws_geom(const TapConv& d, int impl, MfmaGeom* g, int* bn, size_t* lds) {
  if (d.in_stride != 1) return false;
  const int slot_ch = impl == DRS_IMPL_MFMA_F32 ? 4 : 8;
  const int images = impl == DRS_IMPL_MFMA_BF16X3 ? 2 : 1;
  *bn = (impl != DRS_IMPL_MFMA_BF16X3 && d.Cout % 64 == 0) ? 64 : 32;
  int dy0 = 1 << 30, dy1 = -(1 << 30), dx0 = 1 << 30, dx1 = -(1 << 30);
  for (int i = 0; i < d.ntaps; ++i) {
    dy0 = d.dy[i] < dy0 ? d.dy[i] : dy0; dy1 = d.dy[i] > dy1 ? d.dy[i] : dy1;
    dx0 = d.dx[i] < dx0 ? d.dx[i] : dx0; dx1 = d.dx[i] > dx1 ? d.dx[i] : dx1;
  }
  g->dy_min = dy0; g->dx_min = dx0;
  g->IH = 15 + (dy1 - dy0) + 1;
  g->IW = 15 + (dx1 - dx0) + 1;
  if (g->IH * g->IW > 384) return false;
  g->tiles_x = drs_cdiv(d.TW, 16);
  g->tiles_y = drs_cdiv(d.TH, 16);
  g->nchunks = drs_cdiv(d.Cin, 4 * slot_ch);
  g->a_plane = (g->IH * g->IW * 16 + 255) / 256 * 256;
  g->a_image = 4 * g->a_plane;
  g->w_image = d.ntaps * 4 * *bn * 16;
  g->w_gimage = g->nchunks * d.wtaps_total * 4 * d.Cout * 16;
  static const int dbg = getenv("DRS_DEBUG_FLAGS") ? atoi(getenv("DRS_DEBUG_FLAGS")) : 0;
  g->debug = dbg;
  *lds = 2 * (size_t)images * ((size_t)g->a_image + g->w_image) + 128;
  return *lds <= (size_t)kLdsLimit;
}

template <class P, int BN>
static int launch_ws(const TapConv& d, const MfmaGeom& g, size_t lds, hipStream_t s) {
  auto kern = tapconv_mfma_ws_kernel<P, BN>;
  static bool attr_done = false;
  static int num_cu = 0;
  if (!attr_done) {
    DRS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kLdsLimit));
    int dev = 0;
    DRS_CHECK_HIP(hipGetDevice(&dev));
    DRS_CHECK_HIP(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    attr_done = true;
  }
  const int nitems = d.N * g.tiles_x * g.tiles_y * (d.Cout / BN);
  int grid = (num_cu / 8) * 8;  // one persistent block per CU, a multiple of the 8 XCDs
  if (grid < 8) grid = 8;
  while (grid > 8 && grid / 2 >= nitems) grid /= 2;  // tiny problems: do not launch idle blocks
  grid = (grid / 8) * 8;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, d, g);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

int drs_launch_tapconv_mfma_ws(const TapConv& d, int impl, const MfmaGeom& g, int bn, size_t lds, hipStream_t s) {
  if (impl == DRS_IMPL_MFMA_BF16X3) return launch_ws<PolicyBF16X3, 32>(d, g, lds, s);
  if (impl == DRS_IMPL_MFMA_F16) return bn == 64 ? launch_ws<PolicyF16, 64>(d, g, lds, s) : launch_ws<PolicyF16, 32>(d, g, lds, s);
  return bn == 64 ? launch_ws<PolicyF32, 64>(d, g, lds, s) : launch_ws<PolicyF32, 32>(d, g, lds, s);
}
