// Backward-pass kernels of the UNet training step (reference loop body train_diffusion_superres.py:388-393:
// loss.backward() through Residual_Attention_UNet_superres).  First correct versions: fp32, LDS-tiled, float atomics
// for the cross-block sums.  Data gradients (dgrad) do not live here: they are tap-convolutions with re-packed
// weights and run on the forward kernels (conv_mfma.hip / conv_direct.hip).
#include "conv_epilogue.h"  // drs_sp_split4
#include "drs_common.h"

template <typename T>
__device__ __forceinline__ T sum16(T (*red)[64], int cl) {  // the 16 row-group partials of channel column cl
  T v = 0;
#pragma unroll
  for (int g = 0; g < 16; ++g) v += red[g][cl];
  return v;
}

static inline unsigned grid1d(long long total, int per_block, int cap) {
  long long b = (total + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of a tap convolution ("correlation" of two channels-last tensors):
//   dW[tap][a][b] = sum_{n, ty, tx} A[n][ty*sa + ay(tap)][tx*sa + ax(tap)][a] * B[n][ty*sb + by(tap)][tx*sb + bx(tap)][b]
// with zero outside either tensor.  Regular conv: A = layer input (sa = stride, offsets = tap offsets), B = dY (sb = 1);
// transposed conv: A = layer input at (ty, tx), B = dY at (2*ty - 1 + ky, 2*tx - 1 + kx).
// Optional: a_add[n][a] added to in-image A pixels (UpConvBlock adds the time embedding before its conv),
// a_gate[n][y/2][x/2] multiplied into A (the attention gate multiplies x before the `result` conv).
// Output index = out_transposed ? (a*Cb + b)*T + tap : (b*Ca + a)*T + tap  (torch ConvTranspose2d / Conv2d layout).
// grid = (position chunks, taps, (Ca/64)*(Cb/64) tiles); block 256 threads; thread = 4x4 block of (a, b).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad_kernel(WgradDesc d, int chunk) {
  constexpr int TP = 16;  // positions per LDS tile
  __shared__ float sA[TP][64 + 1];
  __shared__ float sB[TP][64 + 1];
  const int tap = blockIdx.y;
  const int tiles_b = (d.Cb + 63) / 64;
  const int a0 = (blockIdx.z / tiles_b) * 64, b0 = (blockIdx.z % tiles_b) * 64;
  const int ta = (threadIdx.x >> 4) * 4, tb = (threadIdx.x & 15) * 4;  // this thread's 4x4 block inside the 64x64 tile
  const long long P = (long long)d.N * d.TH * d.TW;
  const long long p_begin = (long long)blockIdx.x * chunk, p_end = min(P, p_begin + chunk);
  float acc[4][4] = {};
  for (long long p0 = p_begin; p0 < p_end; p0 += TP) {
    // stage TP positions x 64 channels of A and of B (zero outside the image / channel range / chunk)
    for (int i = threadIdx.x; i < TP * 64; i += 256) {
      const int pp = i >> 6, c = i & 63;
      const long long p = p0 + pp;
      float va = 0.f, vb = 0.f;
      if (p < p_end) {
        const int tx = (int)(p % d.TW), ty = (int)((p / d.TW) % d.TH), n = (int)(p / ((long long)d.TW * d.TH));
        const int ya = ty * d.sa + d.ay[tap], xa = tx * d.sa + d.ax[tap];
        const int yb = ty * d.sb + d.by[tap], xb = tx * d.sb + d.bx[tap];
        if (a0 + c < d.Ca && ya >= 0 && ya < d.AH && xa >= 0 && xa < d.AW) {
          va = d.A[(((long long)n * d.AH + ya) * d.AW + xa) * d.a_cs + d.a_co + a0 + c];
          if (d.a_add) va += d.a_add[(long long)n * d.a_add_cs + a0 + c];
          if (d.a_gate) va *= d.a_gate[((long long)n * (d.AH >> 1) + (ya >> 1)) * (d.AW >> 1) + (xa >> 1)];
        }
        if (b0 + c < d.Cb && yb >= 0 && yb < d.BH && xb >= 0 && xb < d.BW)
          vb = d.B[(((long long)n * d.BH + yb) * d.BW + xb) * d.b_cs + d.b_co + b0 + c];
      }
      sA[pp][c] = va;
      sB[pp][c] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < TP; ++pp) {
      float av[4], bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { av[j] = sA[pp][ta + j]; bv[j] = sB[pp][tb + j]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int a = a0 + ta + i, b = b0 + tb + j;
      if (a < d.Ca && b < d.Cb) {
        const long long idx = d.out_transposed ? ((long long)a * d.Cb + b) * d.T_total + d.wtap[tap]
                                               : ((long long)b * d.Ca + a) * d.T_total + d.wtap[tap];
        atomicAdd(&d.dW[idx], acc[i][j]);
      }
    }
}

int drs_launch_wgrad(const WgradDesc& d, hipStream_t s) {
  const long long P = (long long)d.N * d.TH * d.TW;
  if (P == 0) return DRS_OK;
  if (d.partial && drs_wgrad_mfma_bf16_supported(d)) return drs_launch_wgrad_mfma_bf16(d, d.partial, d.partial_bytes, s);
  if (d.partial && drs_wgrad_mfma_supported(d)) return drs_launch_wgrad_mfma(d, d.partial, d.partial_bytes, s);
  if (d.dbias) {  // the direct kernel has no fused bias gradient
    int rc = drs_launch_colsum(d.B, d.b_cs, d.b_co, d.Cb, (long long)d.N * d.BH * d.BW, (long long)d.BH * d.BW, 0, 0, d.dbias, s);
    if (rc) return rc;
  }
  const int tiles = ((d.Ca + 63) / 64) * ((d.Cb + 63) / 64);
  // enough position chunks to fill the chip (~2048 blocks in all), at least 256 positions each
  long long chunks = 2048 / ((long long)d.ntaps * tiles);
  if (chunks < 1) chunks = 1;
  long long chunk = (P + chunks - 1) / chunks;
  if (chunk < 256) chunk = 256;
  chunk = (chunk + 15) / 16 * 16;
  chunks = (P + chunk - 1) / chunk;
  DRS_LAUNCH(wgrad_kernel, dim3((unsigned)chunks, d.ntaps, tiles), dim3(256), 0, s, d, (int)chunk);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Column sums of a channels-last slice: out[c] += sum_p t[p][c]   (bias gradients), or per image:
// out[n*out_stride + c] += sum over the image's pixels (time-embedding gradients).  fp32 partials + float atomics.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ t, int cs, int co, int C, long long npix,
                                                     long long pix_per_image, int per_image, int out_stride,
                                                     float* __restrict__ out, int rows_per_block) {
  __shared__ float red[256];
  const int lanes_c = C < 256 ? C : 256;            // threads along channels
  const int rows = 256 / lanes_c;                    // pixel rows handled concurrently
  const int c = threadIdx.x % lanes_c, row = threadIdx.x / lanes_c;
  const long long p_begin = (long long)blockIdx.x * rows_per_block;
  const long long p_end = min(npix, p_begin + rows_per_block);
  for (int cb = 0; cb < C; cb += lanes_c) {
    float s = 0.f;
    if (row < rows && cb + c < C)
      for (long long p = p_begin + row; p < p_end; p += rows) s += t[p * cs + co + cb + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (row == 0 && cb + c < C) {
      for (int r = 1; r < rows; ++r) s += red[r * lanes_c + c];
      const long long n = per_image ? p_begin / pix_per_image : 0;
      atomicAdd(&out[n * out_stride + cb + c], s);
    }
    __syncthreads();
  }
}
// float4 variant (C, cs, co multiples of 4; C <= 1024): thread = (pixel row, channel quad), 4 independent loads in flight
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ t, int cs, int co, int C, long long npix,
                                                      long long pix_per_image, int per_image, int out_stride,
                                                      float* __restrict__ out, int rows_per_block) {
  __shared__ float red[256][4];
  const int qn = C >> 2;
  const int rows = 256 / qn;
  const int q = threadIdx.x % qn, row = threadIdx.x / qn;
  const long long p_begin = (long long)blockIdx.x * rows_per_block;
  const long long p_end = min(npix, p_begin + rows_per_block);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (row < rows) {
    const float* base = t + co + q * 4;
    long long p = p_begin + row;
    for (; p + 3LL * rows < p_end; p += 4LL * rows) {
      const float4 a = *reinterpret_cast<const float4*>(base + p * cs);
      const float4 b = *reinterpret_cast<const float4*>(base + (p + rows) * cs);
      const float4 c2 = *reinterpret_cast<const float4*>(base + (p + 2LL * rows) * cs);
      const float4 d = *reinterpret_cast<const float4*>(base + (p + 3LL * rows) * cs);
      s[0] += (a.x + b.x) + (c2.x + d.x); s[1] += (a.y + b.y) + (c2.y + d.y);
      s[2] += (a.z + b.z) + (c2.z + d.z); s[3] += (a.w + b.w) + (c2.w + d.w);
    }
    for (; p < p_end; p += rows) {
      const float4 a = *reinterpret_cast<const float4*>(base + p * cs);
      s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[threadIdx.x][j] = s[j];
  __syncthreads();
  if (threadIdx.x < qn) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = 0.f;
      for (int r = 0; r < rows; ++r) v += red[r * qn + threadIdx.x][j];
      const long long n = per_image ? p_begin / pix_per_image : 0;
      atomicAdd(&out[n * out_stride + threadIdx.x * 4 + j], v);
    }
  }
}
// whole-tensor column sums without atomics: block b sums its rows into partials[b][C] (grid-stride over row groups), the
// finish kernel adds the rows into out
__global__ __launch_bounds__(256) void colsum4_partial_kernel(const float* __restrict__ t, int cs, int co, int C, long long npix,
                                                              float* __restrict__ partials) {
  __shared__ float red[256][4];
  const int qn = C >> 2;
  const int rows = 256 / qn;
  const int q = threadIdx.x % qn, row = threadIdx.x / qn;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (row < rows) {
    const float* base = t + co + q * 4;
    const long long stride = (long long)gridDim.x * rows;
    long long p = (long long)blockIdx.x * rows + row;
    for (; p + 3 * stride < npix; p += 4 * stride) {
      const float4 a = *reinterpret_cast<const float4*>(base + p * cs);
      const float4 b = *reinterpret_cast<const float4*>(base + (p + stride) * cs);
      const float4 c2 = *reinterpret_cast<const float4*>(base + (p + 2 * stride) * cs);
      const float4 d = *reinterpret_cast<const float4*>(base + (p + 3 * stride) * cs);
      s[0] += (a.x + b.x) + (c2.x + d.x); s[1] += (a.y + b.y) + (c2.y + d.y);
      s[2] += (a.z + b.z) + (c2.z + d.z); s[3] += (a.w + b.w) + (c2.w + d.w);
    }
    for (; p < npix; p += stride) {
      const float4 a = *reinterpret_cast<const float4*>(base + p * cs);
      s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[threadIdx.x][j] = s[j];
  __syncthreads();
  if (threadIdx.x < qn) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = 0.f;
      for (int r = 0; r < rows; ++r) v += red[r * qn + threadIdx.x][j];
      partials[(size_t)blockIdx.x * C + threadIdx.x * 4 + j] = v;
    }
  }
}
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ partials, int nrows, int C,
                                                            float* __restrict__ out) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float v = 0.f;
  if (c < C)
#pragma unroll 8
    for (int r = rg; r < nrows; r += 16) v += partials[(size_t)r * C + c];
  red[rg][cl] = v;
  __syncthreads();
  if (rg == 0 && c < C) out[c] += sum16(red, cl);
}
int drs_launch_colsum(const float* t, int cs, int co, int C, long long npix, long long pix_per_image, int per_image,
                      int out_stride, float* out, hipStream_t s, float* partials) {
  if (npix == 0) return DRS_OK;
  const bool vec4 = C % 4 == 0 && cs % 4 == 0 && co % 4 == 0 && C <= 1024 && 256 % (C >> 2) == 0;
  if (partials && !per_image && vec4) {
    // whole-tensor sums (bias gradients): every block would end in C float atomics onto the same C addresses - 2048 blocks x
    // ~90 ns per serialised atomic set the launch time (95 - 210 us), not the 134 MB it reads
    // (1024 blocks: with 512, 8 waves per CU x 4 loads in flight moved 1 TB/s - 154 us for the 268 MB slice of grad.cat2;
    //  the partials buffer holds DRS_RED_BLOCKS x 2 x 1024 doubles, i.e. 4096 rows of <= 1024 floats)
    const int rows = 256 / (C >> 2);
    const unsigned blocks = grid1d(npix, rows, 2 * DRS_RED_BLOCKS);
    DRS_LAUNCH(colsum4_partial_kernel, dim3(blocks), dim3(256), 0, s, t, cs, co, C, npix, partials);
    DRS_LAUNCH(colsum_finish_kernel, dim3((C + 63) / 64), dim3(1024), 0, s, partials, (int)blocks, C, out);
    DRS_CHECK_HIP(hipGetLastError());
    return DRS_OK;
  }
  long long rpb = per_image ? 512 : 2048;  // per-image sums: pix_per_image / rpb atomics per address (128 at 256 x 256)
  if (per_image) {
    // at least ~2048 blocks (the 17 - 67 MB tensors of the deep levels ran one block per CU at 1 TB/s), a block's rows a
    // multiple of 4 loads x the pixel rows it covers per pass; blocks must not straddle images
    const long long pass = vec4 ? 4LL * (256 / (C >> 2)) : 1;
    while (rpb > pass && npix / rpb < 2048) rpb >>= 1;
    while (pix_per_image % rpb) rpb >>= 1;
  }
  const long long blocks = (npix + rpb - 1) / rpb;
  if (vec4) {
    DRS_LAUNCH(colsum4_kernel, dim3((unsigned)blocks), dim3(256), 0, s, t, cs, co, C, npix, pix_per_image, per_image,
                       out_stride, out, (int)rpb);
    DRS_CHECK_HIP(hipGetLastError());
    return DRS_OK;
  }
  DRS_LAUNCH(colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, t, cs, co, C, npix, pix_per_image, per_image,
                     out_stride, out, (int)rpb);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Element-wise helpers on channels-last slices
// ---------------------------------------------------------------------------------------------------------------
// g[p][c] *= (y[p][c] > 0)
__global__ void relu_mask_kernel(float* g, int g_cs, int g_co, const float* y, int y_cs, int y_co, int C, long long npix) {
  const long long total = npix * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long p = i / C;
    if (!(y[p * y_cs + y_co + c] > 0.f)) g[p * g_cs + g_co + c] = 0.f;
  }
}
int drs_launch_relu_mask(float* g, int g_cs, int g_co, const float* y, int y_cs, int y_co, int C, long long npix,
                         hipStream_t s) {
  DRS_LAUNCH(relu_mask_kernel, dim3(grid1d(npix * C, 256, 8192)), dim3(256), 0, s, g, g_cs, g_co, y, y_cs, y_co, C,
                     npix);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
// ---------------------------------------------------------------------------------------------------------------
// BatchNorm backward (training statistics).  g = gradient w.r.t. the BatchNorm output (after the optional ReLU mask
// that sat directly on it: relu_pre), zhat = (z - mean) * rstd:
//   dbeta = sum g, dgamma = sum g*zhat, dz = gamma*rstd*(g - dbeta/M - zhat*dgamma/M).
// Pass 1 reduces (fp64 atomics), pass 2 writes dz IN PLACE over z.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 drs_mask4(const float4& g, const float4& y) {  // g * (y > 0)
  return make_float4(y.x > 0.f ? g.x : 0.f, y.y > 0.f ? g.y : 0.f, y.z > 0.f ? g.z : 0.f, y.w > 0.f ? g.w : 0.f);
}
// thread = (pixel row in block, group of 4 channels): 16-byte loads, two pixels in flight per thread
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g, int g_cs, int g_co,
                                                            const float* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int relu_pre, int C,
                                                            long long npix, double* __restrict__ partials,
                                                            const float* __restrict__ mask_y, int my_cs, int my_co) {
  // mask_y: optional output of a ReLU that sat on top of g's tensor (g is taken as g * (mask_y > 0)): the residual block's
  // final ReLU, whose masking pass over gR this replaces
  __shared__ float red[2][256][4];
  const int c4n = C >> 2;
  const int rows = 256 / c4n;
  const int cg = threadIdx.x % c4n, row = threadIdx.x / c4n;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (row < rows) {
    const int c = cg * 4;
    const float4 m = *reinterpret_cast<const float4*>(mean + c), r = *reinterpret_cast<const float4*>(rstd + c);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
    const float mm[4] = {m.x, m.y, m.z, m.w}, rr[4] = {r.x, r.y, r.z, r.w}, gg[4] = {ga.x, ga.y, ga.z, ga.w},
                bb[4] = {be.x, be.y, be.z, be.w};
    auto one = [&](const float4& zv, const float4& gv4) __attribute__((always_inline)) {
      const float zz[4] = {zv.x, zv.y, zv.z, zv.w};
      float gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float zh = (zz[j] - mm[j]) * rr[j];
        if (relu_pre && !(zh * gg[j] + bb[j] > 0.f)) gv[j] = 0.f;
        s1[j] += gv[j];
        s2[j] += gv[j] * zh;
      }
    };
    const long long stride = (long long)gridDim.x * rows;
    long long p = (long long)blockIdx.x * rows + row;
    // four pixels in flight per thread (8 - 12 sixteen-byte loads): with two, the 134 MB layers ran at 2.1 TB/s (150 - 187 us)
    for (; p + 3 * stride < npix; p += 4 * stride) {
      float4 zv[4], gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        zv[u] = *reinterpret_cast<const float4*>(z + (p + u * stride) * C + c);
        gv[u] = *reinterpret_cast<const float4*>(g + (p + u * stride) * g_cs + g_co + c);
      }
      if (mask_y) {
        float4 yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) yv[u] = *reinterpret_cast<const float4*>(mask_y + (p + u * stride) * my_cs + my_co + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) gv[u] = drs_mask4(gv[u], yv[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(zv[u], gv[u]);
    }
    for (; p < npix; p += stride) {
      float4 g0 = *reinterpret_cast<const float4*>(g + p * g_cs + g_co + c);
      if (mask_y) g0 = drs_mask4(g0, *reinterpret_cast<const float4*>(mask_y + p * my_cs + my_co + c));
      one(*reinterpret_cast<const float4*>(z + p * C + c), g0);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][threadIdx.x][j] = s1[j]; red[1][threadIdx.x][j] = s2[j]; }
  __syncthreads();
  if (threadIdx.x < c4n) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double d1 = 0, d2 = 0;
      for (int r = 0; r < rows; ++r) { d1 += (double)red[0][r * c4n + threadIdx.x][j]; d2 += (double)red[1][r * c4n + threadIdx.x][j]; }
      // this block's row of the partials buffer (no atomics: see bn_stats_kernel); bn_bwd_finish_kernel adds the rows up
      partials[(size_t)blockIdx.x * 2 * C + threadIdx.x * 4 + j] = d1;
      partials[(size_t)blockIdx.x * 2 * C + C + threadIdx.x * 4 + j] = d2;
    }
  }
}
// sums[c] = sum of the partial rows (dbeta | dgamma), also written out as the parameter gradients; block = 64 channels x 4 row groups
__global__ __launch_bounds__(1024) void bn_bwd_finish_kernel(const double* __restrict__ partials, int nrows, int C,
                                                            double* __restrict__ sums, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
  __shared__ double red[2][16][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s1 = 0, s2 = 0;
  if (c < C)
#pragma unroll 8
    for (int r = rg; r < nrows; r += 16) {
      s1 += partials[(size_t)r * 2 * C + c];
      s2 += partials[(size_t)r * 2 * C + C + c];
    }
  red[0][rg][cl] = s1;
  red[1][rg][cl] = s2;
  __syncthreads();
  if (rg != 0 || c >= C) return;
  s1 = sum16(red[0], cl);
  s2 = sum16(red[1], cl);
  sums[c] = s1;
  sums[C + c] = s2;
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g, int g_cs, int g_co,
                                                           float* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int relu_pre, int C,
                                                           long long npix, const double* __restrict__ sums,
                                                           char* __restrict__ z_sp, const float* __restrict__ mask_y,
                                                           int my_cs, int my_co) {
  // z_sp: optional second copy of dz in SP format (split bf16 hi | lo per 32-channel group, drs_common.h): the operand form
  // of the wave-specialised data-gradient convolution that reads it next (C % 32 == 0)
  const int c4n = C >> 2;
  const long long total = npix * c4n;
  const double inv = 1.0 / (double)npix;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const long long p = i / c4n;
    const float4 m = *reinterpret_cast<const float4*>(mean + c), r = *reinterpret_cast<const float4*>(rstd + c);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
    const float4 zv = *reinterpret_cast<const float4*>(z + p * C + c);
    float4 gv4 = *reinterpret_cast<const float4*>(g + p * g_cs + g_co + c);
    if (mask_y) gv4 = drs_mask4(gv4, *reinterpret_cast<const float4*>(mask_y + p * my_cs + my_co + c));
    const float mm[4] = {m.x, m.y, m.z, m.w}, rr[4] = {r.x, r.y, r.z, r.w}, gg[4] = {ga.x, ga.y, ga.z, ga.w},
                bb[4] = {be.x, be.y, be.z, be.w}, zz[4] = {zv.x, zv.y, zv.z, zv.w};
    float gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w}, o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float zh = (zz[j] - mm[j]) * rr[j];
      if (relu_pre && !(zh * gg[j] + bb[j] > 0.f)) gv[j] = 0.f;
      o[j] = gg[j] * rr[j] * (gv[j] - (float)(sums[c + j] * inv) - zh * (float)(sums[C + c + j] * inv));
    }
    *reinterpret_cast<float4*>(z + p * C + c) = make_float4(o[0], o[1], o[2], o[3]);
    if (z_sp) {
      unsigned hi[2], lo[2];
      drs_sp_split4(o, hi, lo);
      char* gp = z_sp + ((size_t)p * C + (c & ~31)) * 4 + (c & 31) * 2;  // the group's hi half; the lo half 64 bytes further
      *reinterpret_cast<uint2*>(gp) = make_uint2(hi[0], hi[1]);
      *reinterpret_cast<uint2*>(gp + 64) = make_uint2(lo[0], lo[1]);
    }
  }
}
int drs_launch_bn_bwd(const float* g, int g_cs, int g_co, float* z, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, int relu_pre, int C, long long npix, double* partials,
                      double* sums, float* dgamma, float* dbeta, hipStream_t s, float* z_sp, const float* mask_y, int my_cs,
                      int my_co) {
  DRS_REQUIRE(!z_sp || C % 32 == 0, DRS_ERR_SHAPE, "bn_bwd: an SP copy needs C %% 32 == 0 (C=%d)", C);
  DRS_REQUIRE(!mask_y || ((my_cs & 3) == 0 && (my_co & 3) == 0), DRS_ERR_SHAPE, "bn_bwd: unaligned mask slice");
  // partials: DRS_RED_BLOCKS rows of 2 x C doubles (rewritten by every call: calls must be stream-ordered); sums: this layer's 2 x C totals
  DRS_REQUIRE(C % 4 == 0 && C <= 1024 && 256 % (C >> 2) == 0 && (g_cs & 3) == 0 && (g_co & 3) == 0, DRS_ERR_SHAPE,
              "bn_bwd: C=%d g_cs=%d g_co=%d", C, g_cs, g_co);
  const int rows = 256 / (C >> 2);
  const unsigned blocks = grid1d(npix, rows, DRS_RED_BLOCKS);
  DRS_LAUNCH(bn_bwd_reduce_kernel, dim3(blocks), dim3(256), 0, s, g, g_cs, g_co, z, mean, rstd, gamma, beta, relu_pre, C, npix,
             partials, mask_y, my_cs, my_co);
  DRS_LAUNCH(bn_bwd_finish_kernel, dim3((C + 63) / 64), dim3(1024), 0, s, partials, (int)blocks, C, sums, dgamma, dbeta);
  DRS_LAUNCH(bn_bwd_apply_kernel, dim3(grid1d(npix * (C >> 2), 256, 8192)), dim3(256), 0, s, g, g_cs, g_co, z, mean, rstd,
                     gamma, beta, relu_pre, C, npix, sums, reinterpret_cast<char*>(z_sp), mask_y, my_cs, my_co);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Attention gate backward (reference AttentionBlock :104-107).  Forward: Zr = up2(psi) * (Wr x) + br.
// With E = Wr^T dZr (a 1x1 dgrad, computed by the caller):
//   dpsi_pre[n][y][x] = psi (1 - psi) * sum_{2x2} sum_c x[n][2y+a][2x+b][c] * E[...][c]
//   dx[n][Y][X][c] += psi[n][Y/2][X/2] * E[n][Y][X][c]
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ x, const float* __restrict__ E,
                                                       const float* __restrict__ psi, float* __restrict__ dx,
                                                       float* __restrict__ dpsi_pre, int N, int LH, int LW, int C) {
  // one wave per low-resolution position: lanes stride the 4*C products
  const int lane = threadIdx.x & 63;
  const long long pos = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long total = (long long)N * LH * LW;
  if (pos >= total) return;
  const int lx = (int)(pos % LW), ly = (int)((pos / LW) % LH), n = (int)(pos / ((long long)LW * LH));
  const float ps = psi[pos];
  float s = 0.f;
  for (int i = lane; i < 4 * C; i += 64) {
    const int sub = i / C, c = i - sub * C;
    const long long hp = (((long long)n * 2 * LH + 2 * ly + (sub >> 1)) * 2 * LW + 2 * lx + (sub & 1)) * C + c;
    const float e = E[hp];
    s += x[hp] * e;
    dx[hp] += ps * e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) dpsi_pre[pos] = s * ps * (1.f - ps);
}
int drs_launch_gate_bwd(const float* x, const float* E, const float* psi, float* dx, float* dpsi_pre, int N, int LH,
                        int LW, int C, hipStream_t s) {
  const long long total = (long long)N * LH * LW;
  DRS_LAUNCH(gate_bwd_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, x, E, psi, dx, dpsi_pre, N, LH, LW,
                     C);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// psi conv backward (1x1 conv to one channel, reference :78,:104):  psi_pre = sum_c wpsi[c] P[p][c] + b.
//   dP[p][c] = wpsi[c] * dpsi_pre[p] * (P[p][c] > 0)   (P is a ReLU output: its mask is applied here)
//   dw[c] += sum_p P[p][c] * dpsi_pre[p],  db += sum_p dpsi_pre[p]
// thread = (pixel row in block, group of 4 channels); the block's sums go to ITS row of `partials` ([C] dw | db), the finish
// kernel adds the rows into dw / db (no atomics onto C + 1 shared addresses: see bn_stats_kernel)
__global__ __launch_bounds__(256) void psi_bwd_kernel(const float* __restrict__ Pm, const float* __restrict__ wpsi,
                                                      const float* __restrict__ dpsi_pre, float* __restrict__ dP,
                                                      float* __restrict__ partials, int C, long long npix, int rows_per_block) {
  __shared__ float red[256][5];
  const int c4n = C >> 2, rows = 256 / c4n;
  const int q = threadIdx.x % c4n, row = threadIdx.x / c4n;
  const long long p_begin = (long long)blockIdx.x * rows_per_block, p_end = min(npix, p_begin + rows_per_block);
  float sw[4] = {0.f, 0.f, 0.f, 0.f}, sb = 0.f;
  if (row < rows) {
    const float4 w = *reinterpret_cast<const float4*>(wpsi + q * 4);
    for (long long p = p_begin + row; p < p_end; p += rows) {
      const float dp = dpsi_pre[p];
      const float4 pv = *reinterpret_cast<const float4*>(Pm + p * C + q * 4);
      *reinterpret_cast<float4*>(dP + p * C + q * 4) =
          make_float4(pv.x > 0.f ? w.x * dp : 0.f, pv.y > 0.f ? w.y * dp : 0.f, pv.z > 0.f ? w.z * dp : 0.f, pv.w > 0.f ? w.w * dp : 0.f);
      sw[0] += pv.x * dp; sw[1] += pv.y * dp; sw[2] += pv.z * dp; sw[3] += pv.w * dp;
      if (q == 0) sb += dp;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[threadIdx.x][j] = sw[j];
  red[threadIdx.x][4] = sb;
  __syncthreads();
  if (threadIdx.x < c4n) {
    float* prow = partials + (size_t)blockIdx.x * (C + 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = 0.f;
      for (int r = 0; r < rows; ++r) v += red[r * c4n + threadIdx.x][j];
      prow[threadIdx.x * 4 + j] = v;
    }
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int r = 0; r < rows; ++r) t += red[r * c4n][4];
      prow[C] = t;
    }
  }
}
__global__ __launch_bounds__(1024) void psi_bwd_finish_kernel(const float* __restrict__ partials, int nrows, int C,
                                                             float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;  // column C = the bias gradient
  float v = 0.f;
  if (c <= C)
#pragma unroll 8
    for (int r = rg; r < nrows; r += 16) v += partials[(size_t)r * (C + 1) + c];
  red[rg][cl] = v;
  __syncthreads();
  if (rg == 0 && c <= C) {
    const float t = sum16(red, cl);
    if (c < C) dw[c] += t;
    else *db += t;
  }
}
int drs_launch_psi_bwd(const float* Pm, const float* wpsi, const float* dpsi_pre, float* dP, float* dw, float* db, int C,
                       long long npix, float* partials, hipStream_t s) {
  // partials: at most 2048 rows of C + 1 floats of scratch (stream-ordered use)
  DRS_REQUIRE(C % 4 == 0 && C <= 256 && 256 % (C >> 2) == 0, DRS_ERR_SHAPE, "psi_bwd: C=%d", C);
  if (npix == 0) return DRS_OK;
  long long rpb = 256;
  while ((npix + rpb - 1) / rpb > 2048) rpb *= 2;
  const unsigned blocks = (unsigned)((npix + rpb - 1) / rpb);
  DRS_LAUNCH(psi_bwd_kernel, dim3(blocks), dim3(256), 0, s, Pm, wpsi, dpsi_pre, dP, partials, C, npix, (int)rpb);
  DRS_LAUNCH(psi_bwd_finish_kernel, dim3((C + 1 + 63) / 64), dim3(1024), 0, s, partials, (int)blocks, C, dw, db);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Time-embedding MLP backward (reference :143-151,:161): out = relu(W2 silu(W1 e + b1) + b2), e = posenc(t) (constant).
// dim/8 blocks per MLP, each owning 8 rows of every gradient; loops over the batch.  dtemb = gradient w.r.t. out
// (already summed over pixels by the caller).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_mlp_bwd_kernel(const long long* __restrict__ t,
                                                           const float* __restrict__ inv_freq, DrsMlpBwdTable tab,
                                                           int stride, int B, const float* __restrict__ label_emb,
                                                           const long long* __restrict__ labels, int label_batch,
                                                           int num_classes, float* __restrict__ dlabel) {
  // blockIdx.y = which of the network's time MLPs (all of them in ONE launch: they were seven 73 us launches, each a
  // handful of latency-bound blocks, at the end of the backward's main stream)
  const DrsMlpBwd& e_ = tab.m[blockIdx.y];
  const int dim = e_.dim;
  if ((int)blockIdx.x * 8 >= dim) return;
  const float* __restrict__ W1 = e_.W1;
  const float* __restrict__ b1 = e_.b1;
  const float* __restrict__ W2 = e_.W2;
  const float* __restrict__ temb = e_.temb;
  const float* __restrict__ dtemb = e_.dtemb;
  float* __restrict__ dW1 = e_.dW1;
  float* __restrict__ db1 = e_.db1;
  float* __restrict__ dW2 = e_.dW2;
  float* __restrict__ db2 = e_.db2;
  // Block `blockIdx.x` owns rows [r0, r0 + RB) of dW2 / db2 (index c) and of dW1 / db1 (index k): it re-derives the
  // cheap per-sample vectors (e, pre1, h1, d2: dim x 100 MACs) and keeps its rows' sums over the batch in registers.
  constexpr int RB = 8;
  __shared__ float e[100], pre1[256], h1[256], d2[256], dpre1[RB];
  const int r0 = blockIdx.x * RB;
  const int tid = threadIdx.x;
  float acc2[RB], acc1[RB], accb2 = 0.f, accb1 = 0.f;
#pragma unroll
  for (int r = 0; r < RB; ++r) acc2[r] = acc1[r] = 0.f;
  for (int b = 0; b < B; ++b) {
    const float tf = (float)t[b];
    long long lab = labels ? labels[label_batch == 1 ? 0 : b] : 0;
    if (lab >= num_classes) lab = -1;  // out-of-range class id (the forward poisoned that row with NaN): never index with it
    if (tid < 50) {
      const float arg = tf * inv_freq[tid];
      e[tid] = sinf(arg);
      e[50 + tid] = cosf(arg);
    }
    __syncthreads();
    if (labels) {  // e = posenc(t) + label_emb[y] (generation variant)
      if (tid < 100 && lab >= 0) e[tid] += label_emb[lab * 100 + tid];
      __syncthreads();
    }
    if (tid < dim) {
      float a = b1[tid];
      for (int k = 0; k < 100; ++k) a = fmaf(W1[(size_t)tid * 100 + k], e[k], a);
      pre1[tid] = a;
      h1[tid] = a / (1.f + expf(-a));
      d2[tid] = temb[(size_t)b * stride + tid] > 0.f ? dtemb[(size_t)b * stride + tid] : 0.f;  // ReLU mask
    }
    __syncthreads();
    {  // dpre1[k] = silu'(pre1[k]) * sum_c W2[c][k] d2[c] for this block's RB rows k: 32 lanes per row
      const int rr = tid >> 5, l = tid & 31, k = r0 + rr;
      float dh = 0.f;
      if (k < dim)
        for (int c = l; c < dim; c += 32) dh = fmaf(W2[(size_t)c * dim + k], d2[c], dh);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) dh += __shfl_xor(dh, o, 32);
      if (l == 0) {
        float v = 0.f;
        if (k < dim) {
          const float sg = 1.f / (1.f + expf(-pre1[k]));
          v = dh * (sg * (1.f + pre1[k] * (1.f - sg)));  // d silu
        }
        dpre1[rr] = v;
      }
    }
    __syncthreads();
    if (tid < dim) {
#pragma unroll
      for (int r = 0; r < RB; ++r) acc2[r] = fmaf(r0 + r < dim ? d2[r0 + r] : 0.f, h1[tid], acc2[r]);
    }
    if (tid < 100) {
#pragma unroll
      for (int r = 0; r < RB; ++r) acc1[r] = fmaf(dpre1[r], e[tid], acc1[r]);
      if (dlabel && lab >= 0) {  // d e = W1^T dpre1: this block's RB rows of the sum, onto the embedding row of the sample's class
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < RB; ++r)
          if (r0 + r < dim) a = fmaf(W1[(size_t)(r0 + r) * 100 + tid], dpre1[r], a);
        atomicAdd(&dlabel[lab * 100 + tid], a);
      }
    }
    if (tid >= 128 && tid < 128 + RB) {
      const int r = tid - 128;
      if (r0 + r < dim) { accb2 += d2[r0 + r]; accb1 += dpre1[r]; }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    if (r0 + r >= dim) break;
    if (tid < dim) dW2[(size_t)(r0 + r) * dim + tid] += acc2[r];
    if (tid < 100) dW1[(size_t)(r0 + r) * 100 + tid] += acc1[r];
  }
  if (tid >= 128 && tid < 128 + RB && r0 + tid - 128 < dim) {
    db2[r0 + tid - 128] += accb2;
    db1[r0 + tid - 128] += accb1;
  }
}
int drs_launch_time_mlp_bwd(const long long* t, const float* inv_freq, const DrsMlpBwdTable& tab, int stride, int B,
                            const float* label_emb, const long long* labels, int label_batch, int num_classes, float* dlabel,
                            hipStream_t s) {
  if (tab.n == 0) return DRS_OK;
  int max_dim = 0;
  for (int i = 0; i < tab.n; ++i) {
    DRS_REQUIRE(tab.m[i].dim <= 256, DRS_ERR_SHAPE, "time_mlp_bwd: dim=%d", tab.m[i].dim);
    max_dim = tab.m[i].dim > max_dim ? tab.m[i].dim : max_dim;
  }
  // (the label-embedding gradient is accumulated with atomics by every MLP's blocks, as before)
  DRS_LAUNCH(time_mlp_bwd_kernel, dim3((max_dim + 7) / 8, tab.n), dim3(256), 0, s, t, inv_freq, tab, stride, B, label_emb, labels,
             label_batch, num_classes, dlabel);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Adjoint of the bicubic up-sampling (channels-last, C channels; forward: F.interpolate(mode='bicubic'), A = -0.75,
// align_corners=False, source indices clamped at the borders, reference UNet_model_superres.py:349).  GATHER form: one
// thread per element of dx collects the output-gradient pixels whose (clamped) taps land on it - at most 5*scale per
// axis - with the forward's own weights.  No atomics, no zero-fill, deterministic; the scatter form this replaces issued
// 16 float atomics per output element onto 3-channel rows (611 us per configs[2] step).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_w(float t, float c[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
  c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  c[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}
// weight with which output position o (of a size * scale axis) reads source index j (clamped taps summed)
__device__ __forceinline__ float bicubic_adjoint_w(int o, int j, int size, float rs) {
  const float sv = rs * ((float)o + 0.5f) - 0.5f;
  const float fv = floorf(sv);
  float cw[4];
  cubic_w(sv - fv, cw);
  float w = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) w += (min(max((int)fv - 1 + a, 0), size - 1) == j) ? cw[a] : 0.f;
  return w;
}
__global__ __launch_bounds__(256) void bicubic_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int C,
                                                          int H, int W, int scale) {
  const int OH = H * scale, OW = W * scale;
  const long long total = (long long)N * H * W * C;
  const float rs = 1.f / (float)scale;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int xx = (int)((i / C) % W), yy = (int)((i / ((long long)C * W)) % H);
    const int n = (int)(i / ((long long)C * W * H));
    // an output row oy reads source rows floor(sy) - 1 .. floor(sy) + 2 (then clamped): row yy is reached from
    // oy in [(yy - 2) * scale, (yy + 3) * scale); the exact weight (zero for most of that range's ends) decides
    const int oy0 = max((yy - 2) * scale, 0), oy1 = min((yy + 3) * scale, OH);
    const int ox0 = max((xx - 2) * scale, 0), ox1 = min((xx + 3) * scale, OW);
    float acc = 0.f;
    for (int oy = oy0; oy < oy1; ++oy) {
      const float wy = bicubic_adjoint_w(oy, yy, H, rs);
      if (wy == 0.f) continue;
      const float* row = dy + (((long long)n * OH + oy) * OW) * C + c;
      float racc = 0.f;
      for (int ox = ox0; ox < ox1; ++ox) racc = fmaf(bicubic_adjoint_w(ox, xx, W, rs), row[(long long)ox * C], racc);
      acc = fmaf(wy, racc, acc);
    }
    dx[i] = acc;
  }
}
int drs_launch_bicubic_bwd(const float* dy, float* dx, int N, int C, int H, int W, int scale, hipStream_t s) {
  const long long total = (long long)N * H * W * C;
  if (total == 0) return DRS_OK;
  DRS_LAUNCH(bicubic_bwd_kernel, dim3(grid1d(total, 256, 16384)), dim3(256), 0, s, dy, dx, N, C, H, W, scale);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Whole backward of ONE few-channel 3x3 layer of the LR / SAR encoder (reference ResidualBlock / RRDB,
// UNet_model_superres.py:237-260: CC -> CC channels, stride 1, pad 1, CC <= 4; channels-last tensors, pixel stride CC):
//   dW[co][ci][ky][kx] += sum_p gout[p][co] * in[p + (ky-1, kx-1)][ci]          db[co] += sum_p gout[p][co]
//   gin[p][ci] (+)= sum_{co,ky,kx} W[co][ci][ky][kx] * gout[p - (ky-1, kx-1)][co], then * (mask_y[p][ci] > 0) if mask_y
// (zero outside the image).  One thread per pixel, the CC*CC*9 + CC sums in registers -> wave shuffles -> LDS -> this
// block's row of `partials`; stem_wgrad_finish_kernel adds the rows in a fixed order into dW / db.  (A first version let the
// last block to arrive add the rows itself: one block summing 256 rows was a 40 us tail on a 25 us kernel.)  Replaces, per layer, a weight-gradient launch of the MFMA kernel on the side stream (35 us + a 50 us reduce
// for 81 sums) that the main stream had to wait for, a weight re-pack, a direct tap convolution and a mask pass: the seven
// layers were 0.9 ms of a 16.5 ms training step spent almost idle.
// ---------------------------------------------------------------------------------------------------------------
template <int CC>
__global__ __launch_bounds__(256) void small_conv_bwd_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                             const float* __restrict__ w, float* __restrict__ gin,
                                                             int accumulate, const float* __restrict__ mask_y, int N, int H,
                                                             int W, float* __restrict__ partials, const float* dW,
                                                             const float* db) {
  constexpr int NW = CC * CC * 9, NACC = NW + CC;
  __shared__ float red[4][NACC];
  const long long npix = (long long)N * H * W;
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
  float wr[NW];  // launch-uniform: scalar loads
#pragma unroll
  for (int i = 0; i < NW; ++i) wr[i] = w[i];
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
    const int x = (int)(p % W), y = (int)((p / W) % H);
    float go[3][3][CC], xi[3][3][CC];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + ky - 1, xx = x + kx - 1;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const long long q = (p + (long long)(ky - 1) * W + (kx - 1)) * CC;
#pragma unroll
        for (int c = 0; c < CC; ++c) {
          go[ky][kx][c] = ok ? gout[q + c] : 0.f;
          xi[ky][kx][c] = ok ? in[q + c] : 0.f;
        }
      }
    float o[CC];
#pragma unroll
    for (int ci = 0; ci < CC; ++ci) o[ci] = accumulate ? gin[p * CC + ci] : 0.f;
#pragma unroll
    for (int co = 0; co < CC; ++co) {
      acc[NW + co] += go[1][1][co];
#pragma unroll
      for (int ci = 0; ci < CC; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int wi = ((co * CC + ci) * 3 + ky) * 3 + kx;
            acc[wi] += go[1][1][co] * xi[ky][kx][ci];
            o[ci] += wr[wi] * go[2 - ky][2 - kx][co];
          }
    }
#pragma unroll
    for (int ci = 0; ci < CC; ++ci) {
      if (mask_y && !(mask_y[p * CC + ci] > 0.f)) o[ci] = 0.f;
      gin[p * CC + ci] = o[ci];
    }
  }
  if (!dW && !db) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    float v = acc[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NACC)
    partials[(size_t)blockIdx.x * NACC + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(1024) void stem_wgrad_finish_kernel(const float* __restrict__ partials, int nrows, int ncol, int nW,
                                                                float* __restrict__ dW, float* __restrict__ db);  // (below: adds partial rows into dW | db)
int drs_launch_small_conv_bwd(const float* in, const float* gout, const float* w, float* gin, int accumulate,
                              const float* mask_y, int N, int CC, int H, int W, float* partials, float* dW, float* db,
                              hipStream_t s) {
  DRS_REQUIRE(CC >= 1 && CC <= 4, DRS_ERR_SHAPE, "small_conv_bwd: CC=%d (1..4)", CC);
  const long long npix = (long long)N * H * W;
  if (npix == 0) return DRS_OK;
  const unsigned blocks = grid1d(npix, 256, 1024);  // <= 1024 partial rows of <= 148 floats
#define DRS_SCB(K) DRS_LAUNCH(small_conv_bwd_kernel<K>, dim3(blocks), dim3(256), 0, s, in, gout, w, gin, accumulate, mask_y, N, H, \
                              W, partials, dW, db)
  switch (CC) {
    case 1: DRS_SCB(1); break;
    case 2: DRS_SCB(2); break;
    case 3: DRS_SCB(3); break;
    default: DRS_SCB(4); break;
  }
#undef DRS_SCB
  if (dW || db) {
    const int ncol = 9 * CC * CC + CC;
    DRS_LAUNCH(stem_wgrad_finish_kernel, dim3((ncol + 63) / 64), dim3(1024), 0, s, partials, (int)blocks, ncol, 9 * CC * CC, dW, db);
  }
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Weight + bias gradient of a stem convolution (3x3, pad 1, CI <= 4 image-like channels -> 16; reference conv0 /
// conv_upsampled_lr_img, UNet_model_superres.py:281,287):
//   dW[co][ci][ky][kx] += sum_p g[p][co] * x[p + (ky-1, kx-1)][ci]      db[co] += sum_p g[p][co]
// g: channels-last, pixel stride g_cs (16 used); x: channels-last, CI channels.  Wave w of a block owns output channels
// 4w .. 4w+3 for 64 pixels per pass: 36 CI + 4 sums per lane in registers over the block's pixels, one shuffle reduction at
// the end, lane 0 writes the wave's columns of the block's partial row (column = flat dW index, then the 16 bias sums);
// stem_wgrad_finish_kernel adds the rows in a fixed order.  The MFMA weight-gradient kernel spent 230 us + a 35 - 90 us slice
// reduction per layer on these 432 sums (scalar staging of a 3-channel operand into 16-wide tiles).
// ---------------------------------------------------------------------------------------------------------------
template <int CI>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ g, int g_cs, const float* __restrict__ x,
                                                         int N, int H, int W, float* __restrict__ partials) {
  constexpr int NW = 4 * CI * 9, NACC = NW + 4, NCOL = 16 * CI * 9 + 16;
  const int lane = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const long long npix = (long long)N * H * W;
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
  for (long long p = (long long)blockIdx.x * 64 + lane; p < npix; p += (long long)gridDim.x * 64) {
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    const float4 gv4 = *reinterpret_cast<const float4*>(g + p * g_cs + cg * 4);
    const float gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w};
    float xi[9][CI];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int y2 = yy + ky - 1, x2 = xx + kx - 1;
        const bool ok = y2 >= 0 && y2 < H && x2 >= 0 && x2 < W;
        const long long q = (p + (long long)(ky - 1) * W + (kx - 1)) * CI;
#pragma unroll
        for (int c = 0; c < CI; ++c) xi[ky * 3 + kx][c] = ok ? x[q + c] : 0.f;
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[NW + j] += gv[j];
#pragma unroll
      for (int c = 0; c < CI; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[(j * CI + c) * 9 + t] += gv[j] * xi[t][c];
    }
  }
  float* row = partials + (size_t)blockIdx.x * NCOL;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    float v = acc[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0) {
      if (i < NW) row[cg * NW + i] = v;              // ((4 cg + j) * CI + c) * 9 + t
      else row[16 * CI * 9 + cg * 4 + (i - NW)] = v;
    }
  }
}
// dW[c] += sum of the partial rows for c < nW, db[c - nW] += ... for the rest; block = 64 columns x 16 row groups
__global__ __launch_bounds__(1024) void stem_wgrad_finish_kernel(const float* __restrict__ partials, int nrows, int ncol, int nW,
                                                                float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float v = 0.f;
  if (c < ncol)
#pragma unroll 8
    for (int r = rg; r < nrows; r += 16) v += partials[(size_t)r * ncol + c];
  red[rg][cl] = v;
  __syncthreads();
  if (rg != 0 || c >= ncol) return;
  const float s = sum16(red, cl);
  if (c < nW) { if (dW) dW[c] += s; }
  else if (db) db[c - nW] += s;
}
int drs_launch_stem_wgrad(const float* g, int g_cs, const float* x, int N, int CI, int H, int W, float* partials,
                          size_t partial_bytes, float* dW, float* db, hipStream_t s) {
  DRS_REQUIRE(CI >= 1 && CI <= 4 && (g_cs & 3) == 0, DRS_ERR_SHAPE, "stem_wgrad: CI=%d g_cs=%d", CI, g_cs);
  const long long npix = (long long)N * H * W;
  if (npix == 0 || (!dW && !db)) return DRS_OK;
  const int ncol = 16 * CI * 9 + 16;
  const unsigned blocks = grid1d(npix, 64, 512);
  DRS_REQUIRE(partial_bytes >= (size_t)blocks * ncol * 4, DRS_ERR_WORKSPACE, "stem_wgrad: partial workspace too small");
#define DRS_SWG(K) DRS_LAUNCH(stem_wgrad_kernel<K>, dim3(blocks), dim3(256), 0, s, g, g_cs, x, N, H, W, partials)
  switch (CI) {
    case 1: DRS_SWG(1); break;
    case 2: DRS_SWG(2); break;
    case 3: DRS_SWG(3); break;
    default: DRS_SWG(4); break;
  }
#undef DRS_SWG
  DRS_LAUNCH(stem_wgrad_finish_kernel, dim3((ncol + 63) / 64), dim3(1024), 0, s, partials, (int)blocks, ncol, 16 * CI * 9, dW, db);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Data gradient of a stem convolution (3x3, pad 1, C <= 4 image-like channels -> 16; reference conv_upsampled_lr_img /
// conv0, UNet_model_superres.py:281,287):  dx[n][y][x][c] = sum_{ky,kx,o} g[n][y+1-ky][x+1-kx][o] * w[o][c][ky][kx], zero
// outside the image.  Block = 16 x 16 pixels; the 18 x 18 x 16 tile of g goes through LDS once (the direct tap kernel this
// replaces re-read every pixel's 16 channels for each of the 9 taps from L1 / L2: 610 us per configs[2] step for a
// 3-channel result).  g: channels-last with pixel stride g_cs (16 used); w: the layer's own (16, C, 3, 3) parameter.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_dgrad_kernel(const float* __restrict__ g, int g_cs, const float* __restrict__ w,
                                                         float* __restrict__ dx, int H, int W, int C) {
  __shared__ float sg[18 * 18][17];  // (+1: a lane's 16-float rows start in different banks)
  __shared__ float sw[16 * 4 * 9];   // [o][c (padded to 4)][tap]
  const int n = blockIdx.z, y0 = blockIdx.y * 16, x0 = blockIdx.x * 16;
  for (int i = threadIdx.x; i < 16 * 4 * 9; i += 256) {
    const int tap = i % 9, c = (i / 9) & 3, o = i / 36;
    sw[i] = c < C ? w[(o * C + c) * 9 + tap] : 0.f;
  }
  for (int i = threadIdx.x; i < 18 * 18 * 4; i += 256) {  // one float4 (4 of the 16 channels) per step
    const int q = i & 3, pix = i >> 2;
    const int py = pix / 18, px = pix - py * 18;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y >= 0 && y < H && x >= 0 && x < W)
      v = *reinterpret_cast<const float4*>(g + (((long long)n * H + y) * W + x) * g_cs + q * 4);
    sg[pix][q * 4] = v.x; sg[pix][q * 4 + 1] = v.y; sg[pix][q * 4 + 2] = v.z; sg[pix][q * 4 + 3] = v.w;
  }
  __syncthreads();
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const int y = y0 + ty, x = x0 + tx;
  if (y >= H || x >= W) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float* gp = sg[(ty + 2 - ky) * 18 + (tx + 2 - kx)];  // g at (y + 1 - ky, x + 1 - kx)
#pragma unroll
      for (int o = 0; o < 16; ++o) {
        const float gv = gp[o];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(gv, sw[(o * 4 + c) * 9 + ky * 3 + kx], acc[c]);
      }
    }
  float* out = dx + (((long long)n * H + y) * W + x) * C;
  for (int c = 0; c < C; ++c) out[c] = acc[c];
}
int drs_launch_stem_dgrad(const float* g, int g_cs, const float* w, float* dx, int N, int H, int W, int C, hipStream_t s) {
  DRS_REQUIRE(C >= 1 && C <= 4 && (g_cs & 3) == 0 && g_cs >= 16, DRS_ERR_SHAPE, "stem_dgrad: C=%d g_cs=%d", C, g_cs);
  if ((long long)N * H * W == 0) return DRS_OK;
  DRS_LAUNCH(stem_dgrad_kernel, dim3((W + 15) / 16, (H + 15) / 16, N), dim3(256), 0, s, g, g_cs, w, dx, H, W, C);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-tensor Adam (torch.optim.Adam defaults), one launch for all parameters: grid = (chunks of the largest tensor,
// tensors); blocks beyond a tensor's length exit.  Same operation order as torch's single-tensor implementation.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_multi_kernel(const drs_adam_tensor* __restrict__ table, double lr, double b1d,
                                                         double b2d, float eps) {
  const drs_adam_tensor t = table[blockIdx.y];
  if (!t.g) return;  // no gradient this step: torch skips the parameter (state untouched)
  // torch forms 1 - beta in Python doubles and rounds the result to fp32 inside the kernels
  const float beta2 = (float)b2d, w1 = (float)(1.0 - b1d), w2 = (float)(1.0 - b2d);
  const long long lo = (long long)blockIdx.x * 4096;
  if (lo >= t.n) return;
  const long long hi = min(t.n, lo + 4096);
  // bias corrections in double like torch's Python-side scalars (step_size = lr / bc1, bias_correction2_sqrt)
  const double bc1 = 1.0 - pow(b1d, (double)t.step), bc2 = 1.0 - pow(b2d, (double)t.step);
  const float lr_over_bc1 = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float g = t.g[i];
    float m = t.m[i], v = t.v[i];
    m = w1 < 0.5f ? fmaf(w1, __fsub_rn(g, m), m) : __fsub_rn(g, __fmul_rn(__fsub_rn(g, m), __fsub_rn(1.f, w1)));  // lerp_
    v = __fadd_rn(__fmul_rn(v, beta2), __fmul_rn(__fmul_rn(w2, g), g));                                          // mul_, addcmul_
    const float denom = __fadd_rn(__fdiv_rn(sqrtf(v), bc2_sqrt), eps);
    t.m[i] = m;
    t.v[i] = v;
    t.p[i] = __fadd_rn(t.p[i], __fmul_rn(-lr_over_bc1, __fdiv_rn(m, denom)));                                    // addcdiv_
  }
}
extern "C" int drs_adam_multi(const drs_adam_tensor* table, int ntensors, int64_t max_numel, double lr, double beta1,
                              double beta2, double eps, drs_stream_t stream) {
  DRS_REQUIRE(table && ntensors >= 0, DRS_ERR_ARG, "adam_multi: bad arguments");
  if (ntensors == 0 || max_numel <= 0) return DRS_OK;
  const unsigned gx = (unsigned)((max_numel + 4095) / 4096);
  DRS_LAUNCH(adam_multi_kernel, dim3(gx, ntensors), dim3(256), 0, (hipStream_t)stream, table, lr, beta1, beta2,
                     (float)eps);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-tensor EMA of the parameters (include/drs_hip.h: drs_ema_multi; reference EMA, UNet_model_superres.py:12-55):
// grid = (chunks of the largest tensor, tensors).  mode 0: old * beta + (1 - beta) * new with torch's rounding (scalar
// operands rounded to fp32, two products and one sum, no fused multiply-add); mode 1: word copy (warm-up).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ema_multi_kernel(const drs_ema_tensor* __restrict__ table, float beta, float w, int mode) {
  const drs_ema_tensor t = table[blockIdx.y];
  const long long lo = (long long)blockIdx.x * 4096;
  if (lo >= t.n) return;
  const long long hi = min((long long)t.n, lo + 4096);
  if (mode == 1) {
    unsigned* dst = (unsigned*)t.ema;
    const unsigned* src = (const unsigned*)t.cur;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) dst[i] = src[i];
  } else {
    float* e = (float*)t.ema;
    const float* c = (const float*)t.cur;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) e[i] = __fadd_rn(__fmul_rn(e[i], beta), __fmul_rn(w, c[i]));
  }
}
extern "C" int drs_ema_multi(const drs_ema_tensor* table, int ntensors, int64_t max_numel, double beta, int mode,
                             drs_stream_t stream) {
  DRS_REQUIRE(table && ntensors >= 0 && (mode == 0 || mode == 1), DRS_ERR_ARG, "ema_multi: bad arguments");
  if (ntensors == 0 || max_numel <= 0) return DRS_OK;
  const unsigned gx = (unsigned)((max_numel + 4095) / 4096);
  DRS_LAUNCH(ema_multi_kernel, dim3(gx, ntensors), dim3(256), 0, (hipStream_t)stream, table, (float)beta,
                     (float)(1.0 - beta), mode);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
