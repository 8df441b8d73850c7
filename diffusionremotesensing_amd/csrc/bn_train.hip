// Train-mode BatchNorm over channels-last activations (reference nn.BatchNorm2d in training: batch mean / biased
// variance over (N,H,W), running statistics updated with momentum 0.1 and the UNBIASED variance; eps 1e-5).
// In train mode the normalisation cannot be folded into the preceding convolution, so the plan runs
//   conv (raw weights) -> Z  ->  bn_stats(Z)  ->  bn_finalize  ->  bn_apply(Z) with the block's activation / adds.
#include "drs_common.h"

template <typename T>
__device__ __forceinline__ T sum16(T (*red)[64], int cl) {  // the 16 row-group partials of channel column cl
  T v = 0;
#pragma unroll
  for (int g = 0; g < 16; ++g) v += red[g][cl];
  return v;
}

// Per-channel sum and sum of squares.  Thread = (pixel row in block, group of 4 channels): float4 loads, fp32 partials
// over a few thousand elements per thread, LDS reduction over the block, then the block's 2 x C fp64 partial sums go to
// ITS row of `partials` (gridDim.x rows); bn_finalize_kernel adds the rows up.  (Atomics onto 2 x C shared addresses
// serialise: 512 blocks x ~90 ns = 46 us per launch at ANY tensor size.)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, long long npix, int C, int cs,
                                                       int co, double* __restrict__ partials) {
  __shared__ float red[2][256][4];
  const int c4n = C >> 2;             // channel groups
  const int rows = 256 / c4n;         // pixel rows handled concurrently by the block (C <= 1024)
  const int cg = threadIdx.x % c4n, row = threadIdx.x / c4n;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  if (row < rows) {
    // four independent 16-byte loads in flight per thread (with one, 512 blocks keep 2 MB in flight: 1.0 - 1.6 TB/s measured)
    const long long stride = (long long)gridDim.x * rows;
    const float* base = z + co + cg * 4;
    long long p = (long long)blockIdx.x * rows + row;
    for (; p + 3 * stride < npix; p += 4 * stride) {
      const float4 v0 = *reinterpret_cast<const float4*>(base + p * cs);
      const float4 v1 = *reinterpret_cast<const float4*>(base + (p + stride) * cs);
      const float4 v2 = *reinterpret_cast<const float4*>(base + (p + 2 * stride) * cs);
      const float4 v3 = *reinterpret_cast<const float4*>(base + (p + 3 * stride) * cs);
      s[0] += (v0.x + v1.x) + (v2.x + v3.x); s[1] += (v0.y + v1.y) + (v2.y + v3.y);
      s[2] += (v0.z + v1.z) + (v2.z + v3.z); s[3] += (v0.w + v1.w) + (v2.w + v3.w);
      q[0] += (v0.x * v0.x + v1.x * v1.x) + (v2.x * v2.x + v3.x * v3.x);
      q[1] += (v0.y * v0.y + v1.y * v1.y) + (v2.y * v2.y + v3.y * v3.y);
      q[2] += (v0.z * v0.z + v1.z * v1.z) + (v2.z * v2.z + v3.z * v3.z);
      q[3] += (v0.w * v0.w + v1.w * v1.w) + (v2.w * v2.w + v3.w * v3.w);
    }
    for (; p < npix; p += stride) {
      const float4 v = *reinterpret_cast<const float4*>(base + p * cs);
      s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
      q[0] += v.x * v.x; q[1] += v.y * v.y; q[2] += v.z * v.z; q[3] += v.w * v.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][threadIdx.x][j] = s[j]; red[1][threadIdx.x][j] = q[j]; }
  __syncthreads();
  if (threadIdx.x < c4n) {
    double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
    for (int r = 0; r < rows; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ds[j] += (double)red[0][r * c4n + threadIdx.x][j];
        dq[j] += (double)red[1][r * c4n + threadIdx.x][j];
      }
    double* row = partials + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      row[threadIdx.x * 4 + j] = ds[j];
      row[C + threadIdx.x * 4 + j] = dq[j];
    }
  }
}

// Adds up the `nrows` partial rows and finalises: block = 64 channels x 4 row groups.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const double* __restrict__ partials, int nrows, long long npix, int C,
                                                          float eps, float momentum, float* __restrict__ mean,
                                                          float* __restrict__ rstd, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var) {
  __shared__ double red[2][16][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s = 0, q = 0;
  if (c < C)
#pragma unroll 8
    for (int r = rg; r < nrows; r += 16) {
      s += partials[(size_t)r * 2 * C + c];
      q += partials[(size_t)r * 2 * C + C + c];
    }
  red[0][rg][cl] = s;
  red[1][rg][cl] = q;
  __syncthreads();
  if (rg != 0 || c >= C) return;
  s = sum16(red[0], cl);
  q = sum16(red[1], cl);
  const double m = s / (double)npix;
  double var = q / (double)npix - m * m;  // biased
  if (var < 0) var = 0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = npix > 1 ? var * (double)npix / (double)(npix - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// y = relu_post( relu_pre( (z - mean) * rstd * gamma + beta ) + post_add[n][c] + res ), channels-last slices.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, int z_cs, int z_co,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ post_add, int post_cs,
                                                       const float* __restrict__ res, int res_cs, int res_co,
                                                       float* __restrict__ out, int out_cs, int out_co, long long npix,
                                                       long long pix_per_image, int C, int relu_pre, int relu_post) {
  const int c4n = C >> 2;
  const long long total = npix * c4n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c4n);
    const long long p = i / c4n;
    const int c = cg * 4;
    const float4 v = *reinterpret_cast<const float4*>(z + p * z_cs + z_co + c);
    const float4 m = *reinterpret_cast<const float4*>(mean + c), r = *reinterpret_cast<const float4*>(rstd + c);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
    float y[4] = {(v.x - m.x) * r.x * g.x + b.x, (v.y - m.y) * r.y * g.y + b.y, (v.z - m.z) * r.z * g.z + b.z,
                  (v.w - m.w) * r.w * g.w + b.w};
    if (relu_pre)
      for (int j = 0; j < 4; ++j) y[j] = drs_maxf(y[j], 0.f);
    if (post_add) {
      const float4 a = *reinterpret_cast<const float4*>(post_add + (p / pix_per_image) * post_cs + c);
      y[0] += a.x; y[1] += a.y; y[2] += a.z; y[3] += a.w;
    }
    if (res) {
      const float4 a = *reinterpret_cast<const float4*>(res + p * res_cs + res_co + c);
      y[0] += a.x; y[1] += a.y; y[2] += a.z; y[3] += a.w;
    }
    if (relu_post)
      for (int j = 0; j < 4; ++j) y[j] = drs_maxf(y[j], 0.f);
    *reinterpret_cast<float4*>(out + p * out_cs + out_co + c) = make_float4(y[0], y[1], y[2], y[3]);
  }
}

int drs_launch_bn_train(const float* z, int z_cs, int z_co, long long npix, long long pix_per_image, int C,
                        const float* gamma, const float* beta, float* running_mean, float* running_var, float eps,
                        float momentum, double* sums_scratch, float* mean, float* rstd, const float* post_add,
                        int post_cs, const float* res, int res_cs, int res_co, float* out, int out_cs, int out_co,
                        int relu_pre, int relu_post, hipStream_t s) {
  DRS_REQUIRE(C % 4 == 0 && C <= 1024, DRS_ERR_SHAPE, "bn_train: C=%d", C);
  DRS_REQUIRE((z_cs & 3) == 0 && (z_co & 3) == 0 && (out_cs & 3) == 0 && (out_co & 3) == 0, DRS_ERR_SHAPE,
              "bn_train: unaligned channel slices");
  // (sums_scratch: DRS_RED_BLOCKS rows of 2 x C fp64 partial sums, rewritten by every call: calls must be stream-ordered)
  const int rows = 256 / (C >> 2);
  long long blocks = (npix + rows - 1) / rows;
  if (blocks > DRS_RED_BLOCKS) blocks = DRS_RED_BLOCKS;
  if (blocks < 1) blocks = 1;
  DRS_LAUNCH(bn_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, s, z, npix, C, z_cs, z_co, sums_scratch);
  DRS_LAUNCH(bn_finalize_kernel, dim3((C + 63) / 64), dim3(1024), 0, s, sums_scratch, (int)blocks, npix, C, eps, momentum, mean,
                     rstd, running_mean, running_var);
  long long total = npix * (C >> 2);
  long long ab = (total + 255) / 256;
  if (ab > 8192) ab = 8192;
  DRS_LAUNCH(bn_apply_kernel, dim3((unsigned)ab), dim3(256), 0, s, z, z_cs, z_co, mean, rstd, gamma, beta, post_add,
                     post_cs, res, res_cs, res_co, out, out_cs, out_co, npix, pix_per_image, C, relu_pre, relu_post);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
