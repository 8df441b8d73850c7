// Train-mode BatchNorm over channels-last activations (reference nn.BatchNorm2d in training: batch mean / biased
// variance over (N,H,W), running statistics updated with momentum 0.1 and the UNBIASED variance; eps 1e-5).
// In train mode the normalisation cannot be folded into the preceding convolution, so the plan runs
//   conv (raw weights) -> Z  ->  bn_stats(Z)  ->  bn_finalize  ->  bn_apply(Z) with the block's activation / adds.
#include "drs_common.h"

// Per-channel sum and sum of squares.  Thread = (pixel row in block, group of 4 channels): float4 loads, fp32 partials
// over a few thousand elements per thread, LDS reduction over the block, then ONE fp64 atomic per channel per block.
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, long long npix, int C, int cs,
                                                       int co, double* __restrict__ sums) {
  __shared__ float red[2][256][4];
  const int c4n = C >> 2;             // channel groups
  const int rows = 256 / c4n;         // pixel rows handled concurrently by the block (C <= 1024)
  const int cg = threadIdx.x % c4n, row = threadIdx.x / c4n;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  if (row < rows) {
    for (long long p = (long long)blockIdx.x * rows + row; p < npix; p += (long long)gridDim.x * rows) {
      const float4 v = *reinterpret_cast<const float4*>(z + p * cs + co + cg * 4);
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
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      atomicAdd(&sums[threadIdx.x * 4 + j], ds[j]);
      atomicAdd(&sums[C + threadIdx.x * 4 + j], dq[j]);
    }
  }
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, long long npix, int C, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                                   float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = sums[c] / (double)npix;
  double var = sums[C + c] / (double)npix - m * m;  // biased
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
      for (int j = 0; j < 4; ++j) y[j] = fmaxf(y[j], 0.f);
    if (post_add) {
      const float4 a = *reinterpret_cast<const float4*>(post_add + (p / pix_per_image) * post_cs + c);
      y[0] += a.x; y[1] += a.y; y[2] += a.z; y[3] += a.w;
    }
    if (res) {
      const float4 a = *reinterpret_cast<const float4*>(res + p * res_cs + res_co + c);
      y[0] += a.x; y[1] += a.y; y[2] += a.z; y[3] += a.w;
    }
    if (relu_post)
      for (int j = 0; j < 4; ++j) y[j] = fmaxf(y[j], 0.f);
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
  DRS_CHECK_HIP(hipMemsetAsync(sums_scratch, 0, 2 * (size_t)C * sizeof(double), s));
  const int rows = 256 / (C >> 2);
  long long blocks = (npix + rows - 1) / rows;
  // every block ends with 2*C fp64 atomics on the same 2*C addresses: 2048 blocks made those the whole cost (182 us
  // at any size); 512 blocks still cover the chip twice
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  DRS_LAUNCH(bn_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, s, z, npix, C, z_cs, z_co, sums_scratch);
  DRS_LAUNCH(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums_scratch, npix, C, eps, momentum, mean,
                     rstd, running_mean, running_var);
  long long total = npix * (C >> 2);
  long long ab = (total + 255) / 256;
  if (ab > 8192) ab = 8192;
  DRS_LAUNCH(bn_apply_kernel, dim3((unsigned)ab), dim3(256), 0, s, z, z_cs, z_co, mean, rstd, gamma, beta, post_add,
                     post_cs, res, res_cs, res_co, out, out_cs, out_co, npix, pix_per_image, C, relu_pre, relu_post);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
