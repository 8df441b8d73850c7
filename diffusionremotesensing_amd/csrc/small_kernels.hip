// HBM-bound and tiny kernels of the denoising path: weight packing (BatchNorm fold), layout
// changes at the boundary, the 3-channel planar convolutions of the LR encoder, the stem,
// bicubic up-sampling, the fused time-embedding MLP and the diffusion element-wise updates.
#include "drs_common.h"
#include <algorithm>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// Weight packing.  Folds an eval-mode BatchNorm (reference nn.BatchNorm2d, eps 1e-5:
// y = (x - rm) / sqrt(rv + eps) * gamma + beta) that follows a convolution into the convolution:
//   scale[co] = gamma / sqrt(rv + eps);  W' = W * scale;  b' = (b - rm) * scale + beta
// ---------------------------------------------------------------------------------------------
__global__ void pack_conv_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                 float* __restrict__ dst_w, float* __restrict__ dst_b, int Cout, int Cin, int taps,
                                 int transposed, int mfma_layout) {
  const int64_t total = (int64_t)Cout * Cin * taps;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    // i enumerates the destination
    int tap, ci, co;
    if (mfma_layout) {  // [tap][Cout][Cin]
      ci = (int)(i % Cin);
      co = (int)((i / Cin) % Cout);
      tap = (int)(i / ((int64_t)Cin * Cout));
    } else {  // [tap][Cin][Cout]
      co = (int)(i % Cout);
      ci = (int)((i / Cout) % Cin);
      tap = (int)(i / ((int64_t)Cin * Cout));
    }
    const int64_t src = transposed ? (((int64_t)ci * Cout + co) * taps + tap) : (((int64_t)co * Cin + ci) * taps + tap);
    float v = w[src];
    if (gamma) v *= gamma[co] / sqrtf(rvar[co] + eps);
    dst_w[i] = v;
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

int drs_launch_pack_conv(const float* w, const float* b, const float* gamma, const float* beta, const float* rmean,
                         const float* rvar, float eps, float* dst_w, float* dst_b, int Cout, int Cin, int taps,
                         int transposed, int mfma_layout, hipStream_t s) {
  const int64_t total = (int64_t)Cout * Cin * taps;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  DRS_LAUNCH(pack_conv_kernel, dim3(blocks), dim3(256), 0, s, w, b, gamma, beta, rmean, rvar, eps, dst_w,
                     dst_b, Cout, Cin, taps, transposed, mfma_layout);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------
// NCHW <-> NHWC (boundary / debug only; the hot path never converts wide tensors)
// ---------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W,
                                    int dst_cs, int dst_co) {
  const int64_t total = (int64_t)N * C * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;  // n*H*W + y*W + x
    const int64_t hw = (int64_t)H * W;
    const int n = (int)(pix / hw);
    const int64_t r = pix % hw;
    dst[pix * dst_cs + dst_co + c] = src[((int64_t)n * C + c) * hw + r];
  }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W,
                                    int src_cs, int src_co) {
  const int64_t total = (int64_t)N * C * H * W;
  const int64_t hw = (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i % hw;
    const int c = (int)((i / hw) % C);
    const int n = (int)(i / (hw * C));
    dst[i] = src[((int64_t)n * hw + r) * src_cs + src_co + c];
  }
}
// SP-format tensor (drs_common.h) -> NCHW fp32 (parity taps): x = hi + lo
__global__ void sp_to_nchw_kernel(const char* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W,
                                  int src_cs, int src_co) {
  const int64_t total = (int64_t)N * C * H * W;
  const int64_t hw = (int64_t)H * W;
  const int gw = src_cs >= 32 ? 32 : src_cs;  // channels per group
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i % hw;
    const int c = (int)((i / hw) % C) + src_co;
    const int n = (int)(i / (hw * C));
    const char* g = src + (((int64_t)n * hw + r) * src_cs + (c / gw) * gw) * 4;
    const __bf16 h = *reinterpret_cast<const __bf16*>(g + (c % gw) * 2);
    const __bf16 l = *reinterpret_cast<const __bf16*>(g + gw * 2 + (c % gw) * 2);
    dst[i] = (float)h + (float)l;
  }
}
// dst = src + vec[n][c] on SP-format tensors of C (multiple of 32) channels: x + relu(time_mlp(t)) of an UpConvBlock
// (reference :199) when the producing convolution could not write it as its second output (small shapes).
__global__ void sp_add_rowvec_kernel(const char* __restrict__ src, char* __restrict__ dst, const float* __restrict__ vec,
                                     int vec_stride, int64_t slots, int64_t slots_per_image, int C) {
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  const int spp = C / 8;  // 16-byte (hi) slots per pixel
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i / spp;
    const int sl = (int)(i % spp), grp = sl >> 2, kg = sl & 3;
    const int n = (int)(i / slots_per_image);
    const int64_t off = (pix * C + grp * 32) * 4 + kg * 16;
    const bf16x8_t h = *reinterpret_cast<const bf16x8_t*>(src + off), l = *reinterpret_cast<const bf16x8_t*>(src + off + 64);
    const float* v = vec + (int64_t)n * vec_stride + grp * 32 + kg * 8;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (float)h[j] + (float)l[j] + v[j];
    u32x4 oh, ol;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const uint2 s_ = drs_split2(x[2 * q], x[2 * q + 1]); oh[q] = s_.x; ol[q] = s_.y; }
    *reinterpret_cast<u32x4*>(dst + off) = oh;
    *reinterpret_cast<u32x4*>(dst + off + 64) = ol;
  }
}
static inline int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}
int drs_launch_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int dst_cs, int dst_co,
                            hipStream_t s) {
  DRS_LAUNCH(nchw_to_nhwc_kernel, dim3(ew_blocks((int64_t)N * C * H * W)), dim3(256), 0, s, src, dst, N, C, H,
                     W, dst_cs, dst_co);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
int drs_launch_sp_to_nchw(const float* src, float* dst, int N, int C, int H, int W, int src_cs, int src_co,
                          hipStream_t s) {
  DRS_LAUNCH(sp_to_nchw_kernel, dim3(ew_blocks((int64_t)N * C * H * W)), dim3(256), 0, s,
                     reinterpret_cast<const char*>(src), dst, N, C, H, W, src_cs, src_co);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
int drs_launch_sp_add_rowvec(const float* src, float* dst, const float* vec, int vec_stride, int N, long long pix_per_image,
                             int C, hipStream_t s) {
  DRS_REQUIRE(C % 32 == 0, DRS_ERR_SHAPE, "sp_add_rowvec: C=%d", C);
  const int64_t per_image = (int64_t)pix_per_image * (C / 8), slots = per_image * N;
  if (slots == 0) return DRS_OK;
  DRS_LAUNCH(sp_add_rowvec_kernel, dim3(ew_blocks(slots)), dim3(256), 0, s, reinterpret_cast<const char*>(src),
                     reinterpret_cast<char*>(dst), vec, vec_stride, slots, per_image, C);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
int drs_launch_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, int src_cs, int src_co,
                            hipStream_t s) {
  DRS_LAUNCH(nhwc_to_nchw_kernel, dim3(ew_blocks((int64_t)N * C * H * W)), dim3(256), 0, s, src, dst, N, C, H,
                     W, src_cs, src_co);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------
// Planar 3x3 s1 p1 convolution for the few-channel LR encoder (RRDB, reference :230-260):
// out = [relu](conv(in) + b) [+ res].  Cin, Cout <= 4.  Weights are torch layout (Cout,Cin,3,3).
// HBM-bound: one lane per pixel, coalesced along x; neighbours come from L1.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3x3_planar_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                             const float* __restrict__ b,
                                                             const float* __restrict__ res, float* __restrict__ out,
                                                             int N, int Cin, int Cout, int H, int W, int relu) {
  __shared__ float sw[4 * 4 * 9 + 4];
  for (int i = threadIdx.x; i < Cout * Cin * 9; i += blockDim.x) sw[i] = w[i];
  if (threadIdx.x < Cout) sw[4 * 4 * 9 + threadIdx.x] = b ? b[threadIdx.x] : 0.f;
  __syncthreads();
  const int64_t hw = (int64_t)H * W;
  const int64_t total = (int64_t)N * hw;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(p % W);
    const int y = (int)((p / W) % H);
    const int n = (int)(p / hw);
    float acc[4];
#pragma unroll
    for (int co = 0; co < 4; ++co) acc[co] = co < Cout ? sw[4 * 4 * 9 + co] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
      const float* ip = in + ((int64_t)n * Cin + ci) * hw;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = y + ky - 1;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = x + kx - 1;
          if (ix < 0 || ix >= W) continue;
          const float a = ip[(int64_t)iy * W + ix];
#pragma unroll
          for (int co = 0; co < 4; ++co)
            if (co < Cout) acc[co] = fmaf(a, sw[(co * Cin + ci) * 9 + ky * 3 + kx], acc[co]);
        }
      }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      if (co >= Cout) break;
      float v = acc[co];
      if (relu) v = drs_maxf(v, 0.f);
      const int64_t o = ((int64_t)n * Cout + co) * hw + (int64_t)y * W + x;
      if (res) v += res[o];
      out[o] = v;
    }
  }
}
int drs_launch_conv3x3_planar(const float* in, const float* w, const float* b, const float* res, float* out, int N,
                              int Cin, int Cout, int H, int W, int relu, hipStream_t s) {
  DRS_REQUIRE(Cin >= 1 && Cin <= 4 && Cout >= 1 && Cout <= 4, DRS_ERR_SHAPE, "planar conv: Cin=%d Cout=%d (max 4)", Cin,
              Cout);
  DRS_LAUNCH(conv3x3_planar_kernel, dim3(ew_blocks((int64_t)N * H * W)), dim3(256), 0, s, in, w, b, res, out,
                     N, Cin, Cout, H, W, relu);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------
// Stem: 3x3 s1 p1 convolution from a planar few-channel image to channels-last Cout (=16) with an
// optional channels-last residual (the cached LR-conditioning term, broadcast over the batch when it
// has batch 1).  conv0 and conv_upsampled_lr_img of the reference (:342,:353-355).
// ---------------------------------------------------------------------------------------------
// A wave owns 64 consecutive pixels of kStemRows consecutive rows (a lane = one column, one accumulator set per row): the
// 27 x 16 weights are read from LDS once per column (broadcast reads were the kernel's largest cost at one pixel per lane)
// and the (rows + 2) x 3 input values of a channel serve all rows.  A pixel's 64 output bytes (16 fp32, or 16 bf16 hi | 16 bf16
// lo) and its 64 residual bytes cross a 4 KB per-wave LDS image, so every global load / store instruction of the
// channels-last tensors covers 1 KB of CONSECUTIVE bytes (16 bytes per lane at a 64-byte stride touches each line four
// times).
#ifndef DRS_STEM_ROWS
#define DRS_STEM_ROWS 4
#define DRS_STEM_BPC 2
#endif
constexpr int kStemRows = DRS_STEM_ROWS;  // rows per lane
__device__ __attribute__((aligned(16))) float stem_zero_line[4];  // (zero-initialised, never written)
template <int COUT>
__global__ __launch_bounds__(256, DRS_STEM_BPC) void stem_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ b, const float* __restrict__ res,
                                                   int res_batch, float* __restrict__ out, int N, int Cin, int H,
                                                   int W, int out_sp) {
  static_assert(COUT == 16, "a pixel is 64 bytes in both output formats");
  constexpr int R = kStemRows;
  __shared__ __attribute__((aligned(16))) float sw[COUT * 4 * 9 + COUT];  // [tap][ci][co] + bias
  __shared__ __attribute__((aligned(16))) char sT[4][64 * 64];            // per wave: 64 pixels x 64 bytes
  for (int i = threadIdx.x; i < COUT * Cin * 9; i += blockDim.x) {
    const int co = i / (Cin * 9), r = i % (Cin * 9), ci = r / 9, tap = r % 9;
    sw[(tap * Cin + ci) * COUT + co] = w[i];
  }
  if (threadIdx.x < COUT) sw[COUT * 4 * 9 + threadIdx.x] = b[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* tb = sT[wave];
  // The per-wave image is touched two ways: per pixel (lane p, 16-byte chunk q of its 64 bytes: a 64-byte lane stride, 4-way
  // bank conflicts in a plain layout - SQ_LDS_BANK_CONFLICT 0.07 of the kernel's LDS cycles in round 3) and linearly (lane L
  // takes bytes [16 L, 16 L + 16) of a 1 KB piece).  Chunk q of pixel p therefore sits at position q ^ ((p >> 2) & 3): the 16
  // lanes of a pass then hit 16 different 16-byte bank groups either way.  The linear side undoes the swizzle in its GLOBAL
  // address (same 1 KB per instruction, permuted inside each pixel's 64 bytes).
  const int sw_px = (lane >> 2) & 3;                       // per-pixel access: swizzle of this lane's pixel (p = lane)
  const int lin_q = (lane & 3) ^ ((lane >> 4) & 3);        // linear access: logical chunk behind physical position lane & 3 of pixel j * 16 + lane / 4
  const int lin_off = (lane >> 2) * 64 + lin_q * 16;       // ... and its byte offset inside the 1 KB piece of the tensor
  const int64_t hw = (int64_t)H * W;
  const int xbs = (W + 63) / 64, ygs = (H + R - 1) / R;
  const int units = N * ygs * xbs;  // (the launcher checks that this fits 31 bits)
  for (int u = blockIdx.x * 4 + wave; u < units; u += gridDim.x * 4) {
    const int xb = u % xbs, yg = (u / xbs) % ygs, n = u / (xbs * ygs);
    const int x0 = xb * 64, x = x0 + lane, y0 = yg * R;
    const int npx = min(64, W - x0);  // valid pixels of the segment (wave-uniform)
    float acc[R][COUT];
#pragma unroll
    for (int o = 0; o < R; ++o)
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[o][co] = sw[COUT * 4 * 9 + co];
    // residual rows: 4 x 1 KB of consecutive bytes per row, lanes past the segment read the zero word
    u32x4 rnext[4] = {};
    auto load_res = [&](int o) __attribute__((always_inline)) {
      const int y = min(y0 + o, H - 1);
      const char* rp = reinterpret_cast<const char*>(res + (res_batch == 1 ? ((int64_t)y * W + x0) : ((int64_t)n * H + y) * W + x0) * COUT);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        rnext[j] = *reinterpret_cast<const u32x4*>((j * 16 + (lane >> 2) < npx) ? rp + j * 1024 + lin_off
                                                                                 : reinterpret_cast<const char*>(stem_zero_line));
    };
    if (res) load_res(0);
#pragma unroll 1
    for (int ci = 0; ci < Cin; ++ci) {
      const float* ip = in + ((int64_t)n * Cin + ci) * hw;
      float a[R + 2][3];
      // (zero padding = the address of a zero word: a select on the loaded value makes the compiler load under a branch)
#pragma unroll
      for (int r = 0; r < R + 2; ++r) {
        const int iy = y0 - 1 + r;
        const bool yok = iy >= 0 && iy < H;
        const float* rowp = ip + (int64_t)iy * W;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = x - 1 + kx;
          a[r][kx] = *((yok && ix >= 0 && ix < W) ? rowp + ix : stem_zero_line);
        }
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float* wr = &sw[((ky * 3 + kx) * Cin + ci) * COUT];
          float wv[COUT];
#pragma unroll
          for (int co = 0; co < COUT; co += 4) {
            const float4 q = *reinterpret_cast<const float4*>(wr + co);
            wv[co] = q.x; wv[co + 1] = q.y; wv[co + 2] = q.z; wv[co + 3] = q.w;
          }
#pragma unroll
          for (int o = 0; o < R; ++o)
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[o][co] = fmaf(a[o + ky][kx], wv[co], acc[o][co]);
        }
    }
#pragma unroll
    for (int o = 0; o < R; ++o) {
      const int y = y0 + o;
      if (y >= H) break;  // (wave-uniform)
      const int64_t p0 = ((int64_t)n * H + y) * W + x0;  // first pixel of the segment
      if (res) {
        const u32x4 rv[4] = {rnext[0], rnext[1], rnext[2], rnext[3]};
        if (o + 1 < R) load_res(o + 1);  // (a row ahead: its latency hides behind this row's conversion and stores)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(tb + j * 1024 + lane * 16) = rv[j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 r4 = *reinterpret_cast<const float4*>(tb + lane * 64 + ((q ^ sw_px) << 4));
          acc[o][q * 4] += r4.x; acc[o][q * 4 + 1] += r4.y; acc[o][q * 4 + 2] += r4.z; acc[o][q * 4 + 3] += r4.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the image is rewritten below
      }
      if (out_sp) {  // SP format (drs_common.h): one group of [COUT x bf16 hi | COUT x bf16 lo] per pixel
#pragma unroll
        for (int co = 0; co < COUT; co += 8) {
          u32x4 h, l;
#pragma unroll
          for (int q = 0; q < 4; ++q) { const uint2 s_ = drs_split2(acc[o][co + 2 * q], acc[o][co + 2 * q + 1]); h[q] = s_.x; l[q] = s_.y; }
          *reinterpret_cast<u32x4*>(tb + lane * 64 + (((co >> 3) ^ sw_px) << 4)) = h;        // chunks 0, 1: hi halves
          *reinterpret_cast<u32x4*>(tb + lane * 64 + (((2 + (co >> 3)) ^ sw_px) << 4)) = l;  // chunks 2, 3: lo halves
        }
      } else {
#pragma unroll
        for (int co = 0; co < COUT; co += 4)
          *reinterpret_cast<float4*>(tb + lane * 64 + (((co >> 2) ^ sw_px) << 4)) = make_float4(acc[o][co], acc[o][co + 1], acc[o][co + 2], acc[o][co + 3]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      char* op = reinterpret_cast<char*>(out + p0 * COUT);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(tb + j * 1024 + lane * 16);
        if (j * 16 + (lane >> 2) < npx) *reinterpret_cast<u32x4*>(op + j * 1024 + lin_off) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the next row rewrites the image)
    }
  }
}
int drs_launch_stem(const float* in_nchw, const float* w, const float* b, const float* res_nhwc, int res_batch,
                    float* out_nhwc, int N, int Cin, int Cout, int H, int W, hipStream_t s, int out_sp) {
  DRS_REQUIRE(Cout == 16 && Cin >= 1 && Cin <= 4, DRS_ERR_SHAPE, "stem: Cin=%d Cout=%d unsupported", Cin, Cout);
  const int64_t units = (int64_t)N * ((H + kStemRows - 1) / kStemRows) * ((W + 63) / 64);  // a wave per kStemRows rows x 64 columns
  if (units == 0) return DRS_OK;
  DRS_REQUIRE(units < (1LL << 31), DRS_ERR_SHAPE, "stem: %lld row segments", (long long)units);
  const int64_t blocks = std::min<int64_t>((units + 3) / 4, 16384);
  DRS_LAUNCH(stem_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, in_nchw, w, b, res_nhwc,
                     res_batch, out_nhwc, N, Cin, H, W, out_sp);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------
// Bicubic up-sampling by an integer factor, F.interpolate(mode='bicubic', align_corners=False)
// semantics of PyTorch (reference :349): src = (dst + 0.5)/scale - 0.5 (not clamped), Keys kernel with
// A = -0.75, tap indices clamped to the border.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
  c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  c[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}
__global__ __launch_bounds__(256) void bicubic_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int H,
                                                      int W, int scale) {
  const int OH = H * scale, OW = W * scale;
  const int64_t total = (int64_t)NC * OH * OW;
  const float rs = 1.f / (float)scale;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int oy = (int)((i / OW) % OH);
    const int64_t nc = i / ((int64_t)OW * OH);
    const float sy = rs * ((float)oy + 0.5f) - 0.5f;
    const float sx = rs * ((float)ox + 0.5f) - 0.5f;
    const float fy = floorf(sy), fx = floorf(sx);
    const int iy = (int)fy, ix = (int)fx;
    float cy[4], cx[4];
    cubic_coeffs(sy - fy, cy);
    cubic_coeffs(sx - fx, cx);
    const float* ip = x + nc * (int64_t)H * W;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = min(max(iy - 1 + a, 0), H - 1);
      float row = 0.f;
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) {
        const int xx = min(max(ix - 1 + bq, 0), W - 1);
        row = fmaf(ip[(int64_t)yy * W + xx], cx[bq], row);
      }
      acc = fmaf(row, cy[a], acc);
    }
    y[i] = acc;
  }
}
int drs_launch_bicubic(const float* x, float* y, int N, int C, int H, int W, int scale, hipStream_t s) {
  DRS_REQUIRE(scale >= 1, DRS_ERR_SHAPE, "bicubic: scale=%d", scale);
  DRS_LAUNCH(bicubic_kernel, dim3(ew_blocks((int64_t)N * C * H * W * scale * scale)), dim3(256), 0, s, x, y,
                     N * C, H, W, scale);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------
// Fused time embedding: out[b] = relu(W2 silu(W1 e(t_b) + b1) + b2), e = [sin(t f_j) | cos(t f_j)].
// One block per (batch element); dim_out <= 256 threads-worth of rows handled by a strided loop.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_mlp_kernel(const int64_t* __restrict__ t, const float* __restrict__ inv_freq,
                                                       const float* __restrict__ W1, const float* __restrict__ b1,
                                                       const float* __restrict__ W2, const float* __restrict__ b2,
                                                       float* __restrict__ out, int out_stride, int dim_in,
                                                       int dim_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* e = smem;            // dim_in
  float* h = smem + dim_in;   // dim_out
  const int b = blockIdx.x;
  const float tf = (float)t[b];
  const int half = dim_in / 2;
  for (int j = threadIdx.x; j < half; j += blockDim.x) {
    const float arg = tf * inv_freq[j];
    e[j] = sinf(arg);
    e[half + j] = cosf(arg);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < dim_out; c += blockDim.x) {
    const float* wr = W1 + (int64_t)c * dim_in;
    float acc = 0.f;
    for (int k = 0; k < dim_in; ++k) acc = fmaf(wr[k], e[k], acc);
    acc += b1[c];
    h[c] = acc / (1.f + expf(-acc));  // SiLU
  }
  __syncthreads();
  for (int c = threadIdx.x; c < dim_out; c += blockDim.x) {
    const float* wr = W2 + (int64_t)c * dim_out;
    float acc = 0.f;
    for (int k = 0; k < dim_out; ++k) acc = fmaf(wr[k], h[k], acc);
    acc += b2[c];
    out[(int64_t)b * out_stride + c] = drs_maxf(acc, 0.f);
  }
}
int drs_launch_time_mlp(const int64_t* t, const float* inv_freq, const float* W1, const float* b1, const float* W2,
                        const float* b2, float* out, int out_stride, int B, int dim_in, int dim_out, hipStream_t s) {
  DRS_REQUIRE(dim_in % 2 == 0 && dim_in > 0 && dim_out > 0, DRS_ERR_SHAPE, "time_mlp: dims %d %d", dim_in, dim_out);
  if (B == 0) return DRS_OK;
  const size_t shmem = (size_t)(dim_in + dim_out) * sizeof(float);
  DRS_LAUNCH(time_mlp_kernel, dim3(B), dim3(256), shmem, s, t, inv_freq, W1, b1, W2, b2, out, out_stride,
                     dim_in, dim_out);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// All time-embedding MLPs of one forward in ONE launch: grid (B, number of MLPs).  `table` (device memory, written at
// pack time) holds per MLP: {byte offset of W1, b1, W2, b2 inside `packed`, dim, output column offset}.
__global__ __launch_bounds__(256) void time_mlp_multi_kernel(const int64_t* __restrict__ t,
                                                             const float* __restrict__ inv_freq,
                                                             const char* __restrict__ packed,
                                                             const long long* __restrict__ table,
                                                             float* __restrict__ out, int out_stride, int dim_in,
                                                             const float* __restrict__ label_emb,
                                                             const long long* __restrict__ labels, int label_batch,
                                                             int num_classes) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* e = smem;
  float* h = smem + dim_in;
  const int b = blockIdx.x;
  const long long* row = table + (size_t)blockIdx.y * 6;
  const float* W1 = reinterpret_cast<const float*>(packed + row[0]);
  const float* b1 = reinterpret_cast<const float*>(packed + row[1]);
  const float* W2 = reinterpret_cast<const float*>(packed + row[2]);
  const float* b2 = reinterpret_cast<const float*>(packed + row[3]);
  const int dim_out = (int)row[4];
  float* o = out + (size_t)b * out_stride + row[5];
  const float tf = (float)t[b];
  const int half = dim_in / 2;
  for (int j = threadIdx.x; j < half; j += blockDim.x) {
    const float arg = tf * inv_freq[j];
    e[j] = sinf(arg);
    e[half + j] = cosf(arg);
  }
  __syncthreads();
  if (labels) {  // class-conditional generation: t += label_emb(y) (reference UNet_model_generation.py:300-301)
    const long long lab = labels[label_batch == 1 ? 0 : b];  // < 0: this row runs unconditionally (no embedding)
    // a class id >= num_classes never indexes the table (nn.Embedding raises a device assert in the reference): the
    // row's encoding is poisoned with NaN instead, so the mistake shows in the output without a host synchronisation
    if (lab >= 0)
      for (int j = threadIdx.x; j < dim_in; j += blockDim.x)
        e[j] = lab < num_classes ? e[j] + label_emb[(size_t)lab * dim_in + j] : __builtin_nanf("");
    __syncthreads();
  }
  for (int c = threadIdx.x; c < dim_out; c += blockDim.x) {
    const float* wr = W1 + (size_t)c * dim_in;
    float acc = 0.f;
    for (int k = 0; k < dim_in; ++k) acc = fmaf(wr[k], e[k], acc);
    acc += b1[c];
    h[c] = acc / (1.f + expf(-acc));
  }
  __syncthreads();
  for (int c = threadIdx.x; c < dim_out; c += blockDim.x) {
    const float* wr = W2 + (size_t)c * dim_out;
    float acc = 0.f;
    for (int k = 0; k < dim_out; ++k) acc = fmaf(wr[k], h[k], acc);
    acc += b2[c];
    o[c] = drs_maxf(acc, 0.f);
  }
}
int drs_launch_time_mlp_multi(const int64_t* t, const float* inv_freq, const char* packed, const long long* table,
                              int nmlp, int max_dim, float* out, int out_stride, int B, int dim_in,
                              const float* label_emb, const long long* labels, int label_batch, int num_classes,
                              hipStream_t s) {
  if (B == 0 || nmlp == 0) return DRS_OK;
  const size_t shmem = (size_t)(dim_in + max_dim) * sizeof(float);
  DRS_LAUNCH(time_mlp_multi_kernel, dim3(B, nmlp), dim3(256), shmem, s, t, inv_freq, packed, table, out,
                     out_stride, dim_in, label_emb, labels, label_batch, num_classes);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// ---------------------------------------------------------------------------------------------
// Diffusion element-wise updates (float4 streaming, HBM-bound)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void noise_images_kernel(const float* __restrict__ x0, const float* __restrict__ eps,
                                                           const int64_t* __restrict__ t,
                                                           const float* __restrict__ alpha_hat,
                                                           float* __restrict__ xt, int64_t chw) {
  const int n = blockIdx.y;
  const float ah = alpha_hat[t[n]];
  const float a = sqrtf(ah), b = sqrtf(1.f - ah);
  const float* xp = x0 + (int64_t)n * chw;
  const float* ep = eps + (int64_t)n * chw;
  float* op = xt + (int64_t)n * chw;
  // reference: sqrt_alpha_hat * x + sqrt_one_minus_alpha_hat * epsilon (two products, one add)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < chw; i += (int64_t)gridDim.x * blockDim.x)
    op[i] = __fadd_rn(__fmul_rn(a, xp[i]), __fmul_rn(b, ep[i]));
}

extern "C" int drs_noise_images(const float* x0, const float* eps, const int64_t* t, const float* alpha_hat,
                                int noise_steps, float* x_t, int n, int64_t chw, drs_stream_t stream) {
  DRS_REQUIRE(x0 && eps && t && alpha_hat && x_t, DRS_ERR_ARG, "noise_images: null pointer");
  DRS_REQUIRE(n >= 0 && chw >= 0 && noise_steps > 0, DRS_ERR_SHAPE, "noise_images: n=%d chw=%lld", n, (long long)chw);
  if (n == 0 || chw == 0) return DRS_OK;
  int bx = (int)((chw + 255) / 256);
  if (bx > 2048) bx = 2048;
  DRS_LAUNCH(noise_images_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, x0, eps, t, alpha_hat, x_t,
                     chw);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// The three schedule coefficients of one step are read back once per plan of T steps by the host
// wrapper (they are T-long tables living on the device); here they arrive as device tables and a scalar t,
// and a 1-thread prologue would cost a launch, so the kernel below reads them itself.
__global__ __launch_bounds__(256) void sampler_step_tab_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                               const float* __restrict__ noise, int t,
                                                               const float* __restrict__ alpha,
                                                               const float* __restrict__ alpha_hat,
                                                               const float* __restrict__ beta, int64_t numel) {
  const float a = alpha[t], ah = alpha_hat[t], b = beta[t];
  const float c_inv = __fdiv_rn(1.f, sqrtf(a));
  const float c_eps = __fdiv_rn(__fsub_rn(1.f, a), sqrtf(__fsub_rn(1.f, ah)));
  const float c_sig = sqrtf(b);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
    float v = __fmul_rn(c_inv, __fsub_rn(x[i], __fmul_rn(c_eps, eps[i])));
    if (noise) v = __fadd_rn(v, __fmul_rn(c_sig, noise[i]));
    x[i] = v;
  }
}

extern "C" int drs_sampler_step(float* x, const float* eps_pred, const float* noise, int t, const float* alpha,
                                const float* alpha_hat, const float* beta, int noise_steps, int64_t numel,
                                drs_stream_t stream) {
  DRS_REQUIRE(x && eps_pred && alpha && alpha_hat && beta, DRS_ERR_ARG, "sampler_step: null pointer");
  DRS_REQUIRE(t >= 0 && t < noise_steps, DRS_ERR_ARG, "sampler_step: t=%d outside [0,%d)", t, noise_steps);
  if (numel <= 0) return DRS_OK;
  DRS_LAUNCH(sampler_step_tab_kernel, dim3(ew_blocks(numel)), dim3(256), 0, (hipStream_t)stream, x, eps_pred,
                     noise, t, alpha, alpha_hat, beta, numel);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

__global__ void sampler_step_cfg_kernel(float* __restrict__ x, const float* __restrict__ ec,
                                        const float* __restrict__ eu, float w, const float* __restrict__ noise, int t,
                                        const float* __restrict__ alpha, const float* __restrict__ alpha_hat,
                                        const float* __restrict__ beta, int64_t numel) {
  const float a = alpha[t], ah = alpha_hat[t], b = beta[t];
  // same operations, in the same order and without fused multiply-adds, as the reference expressions
  // (train_diffusion_generation.py:239 torch.lerp, :249 the update)
  const float c_inv = __fdiv_rn(1.f, sqrtf(a));
  const float c_eps = __fdiv_rn(__fsub_rn(1.f, a), sqrtf(__fsub_rn(1.f, ah)));
  const float c_sig = sqrtf(b);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
    const float s = eu[i], e = ec[i], d = __fsub_rn(e, s);
    // torch.lerp(start = uncond, end = cond, w): |w| < 0.5 ? fma(w, diff, start) : end - diff * (1 - w)
    const float eps = fabsf(w) < 0.5f ? fmaf(w, d, s) : __fsub_rn(e, __fmul_rn(d, __fsub_rn(1.f, w)));
    float v = __fmul_rn(c_inv, __fsub_rn(x[i], __fmul_rn(c_eps, eps)));
    if (noise) v = __fadd_rn(v, __fmul_rn(c_sig, noise[i]));
    x[i] = v;
  }
}
extern "C" int drs_sampler_step_cfg(float* x, const float* eps_cond, const float* eps_uncond, float cfg_scale,
                                    const float* noise, int t, const float* alpha, const float* alpha_hat,
                                    const float* beta, int noise_steps, int64_t numel, drs_stream_t stream) {
  DRS_REQUIRE(x && eps_cond && eps_uncond && alpha && alpha_hat && beta, DRS_ERR_ARG, "sampler_step_cfg: null pointer");
  DRS_REQUIRE(t >= 0 && t < noise_steps, DRS_ERR_ARG, "sampler_step_cfg: t=%d outside [0,%d)", t, noise_steps);
  if (numel <= 0) return DRS_OK;
  DRS_LAUNCH(sampler_step_cfg_kernel, dim3(ew_blocks(numel)), dim3(256), 0, (hipStream_t)stream, x, eps_cond,
                     eps_uncond, cfg_scale, noise, t, alpha, alpha_hat, beta, numel);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// Gaussian-weighted tile blend (Aggregation_Sampling.py:90-116): one thread per output pixel gathers, in tile order,
// every tile that covers it: the same sequential fp32 sums as the reference's `im_res[...] += patch * weight` loop,
// then the division and the clamp, in one pass and without the two full-size accumulators.
__global__ void aggregate_tiles_kernel(const float* __restrict__ tiles, const int* __restrict__ origins,
                                       const float* __restrict__ weight, float* __restrict__ out,
                                       int* __restrict__ uncovered, int n, int C, int S, int H, int W) {
  const int64_t hw = (int64_t)H * W;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(p / W), x = (int)(p % W);
    float cnt = 0.f;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < C; c0 += 4) {
      cnt = 0.f;
      acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
      for (int i = 0; i < n; ++i) {
        const int ly = y - origins[2 * i], lx = x - origins[2 * i + 1];
        if (ly < 0 || ly >= S || lx < 0 || lx >= S) continue;
        const float w = weight[ly * S + lx];
        cnt = __fadd_rn(cnt, w);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c0 + j < C) acc[j] = __fadd_rn(acc[j], __fmul_rn(tiles[(((int64_t)i * C + c0 + j) * S + ly) * S + lx], w));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + j < C) out[(int64_t)(c0 + j) * hw + p] = fminf(fmaxf(__fdiv_rn(acc[j], cnt), 0.f), 1.f);
    }
    if (cnt == 0.f && uncovered) atomicAdd(uncovered, 1);
  }
}
extern "C" int drs_aggregate_tiles(const float* tiles, const int32_t* origins, const float* weight, float* out,
                                   int32_t* uncovered, int n, int C, int S, int H, int W, drs_stream_t stream) {
  DRS_REQUIRE(tiles && origins && weight && out, DRS_ERR_ARG, "aggregate_tiles: null pointer");
  DRS_REQUIRE(n >= 1 && C >= 1 && S >= 1 && H >= S && W >= S, DRS_ERR_SHAPE, "aggregate_tiles: n=%d C=%d S=%d H=%d W=%d", n,
              C, S, H, W);
  if (uncovered) DRS_CHECK_HIP(hipMemsetAsync(uncovered, 0, sizeof(int32_t), (hipStream_t)stream));
  DRS_LAUNCH(aggregate_tiles_kernel, dim3(ew_blocks((int64_t)H * W)), dim3(256), 0, (hipStream_t)stream, tiles,
                     origins, weight, out, uncovered, n, C, S, H, W);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

extern "C" int drs_bicubic_upsample_nchw(const float* x, float* y, int N, int C, int H, int W, int scale,
                                         drs_stream_t stream) {
  DRS_REQUIRE(x && y, DRS_ERR_ARG, "bicubic: null pointer");
  if ((int64_t)N * C * H * W == 0) return DRS_OK;
  return drs_launch_bicubic(x, y, N, C, H, W, scale, (hipStream_t)stream);
}

extern "C" int drs_time_mlp(const int64_t* t, const float* inv_freq, const float* W1, const float* b1, const float* W2,
                            const float* b2, float* out, int B, int dim_in, int dim_out, drs_stream_t stream) {
  DRS_REQUIRE(t && inv_freq && W1 && b1 && W2 && b2 && out, DRS_ERR_ARG, "time_mlp: null pointer");
  return drs_launch_time_mlp(t, inv_freq, W1, b1, W2, b2, out, dim_out, B, dim_in, dim_out, (hipStream_t)stream);
}


// ------------------------------------------------------------------------------------------------------------------
// Gather copy: up to kCopyBatch small device-to-device copies (fp32 words) in ONE launch: the parameters a plan keeps
// verbatim in its packed image (RRDB / stem weights, the time MLPs, the output projection: 44 hipMemcpyAsync calls per
// re-pack before round 4, i.e. per training step).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_copy_kernel(DrsCopyBatch b) {
  const DrsCopyJob e = b.job[blockIdx.y];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < e.words; i += (long long)gridDim.x * 256) e.dst[i] = e.src[i];
}
int drs_launch_gather_copy(const DrsCopyJob* jobs, int n, hipStream_t s) {
  for (int i0 = 0; i0 < n; i0 += DRS_COPY_BATCH) {
    DrsCopyBatch b = {};
    const int m = n - i0 < DRS_COPY_BATCH ? n - i0 : DRS_COPY_BATCH;
    long long most = 1;
    for (int i = 0; i < m; ++i) { b.job[i] = jobs[i0 + i]; most = jobs[i0 + i].words > most ? jobs[i0 + i].words : most; }
    long long bx = (most + 255) / 256;
    if (bx > 64) bx = 64;
    DRS_LAUNCH(gather_copy_kernel, dim3((unsigned)bx, m), dim3(256), 0, s, b);
  }
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
