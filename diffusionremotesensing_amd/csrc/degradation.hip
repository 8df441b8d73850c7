// On-device "DownBlur" degradation of the super-resolution data feed: HR uint8 image -> LR image, exactly as the
// reference's dataset item does it on the host with Pillow (utils.py:140-158, get_data_superres.__getitem__):
//     x = y.resize((w/m, h/m), BICUBIC)  ->  x.filter(GaussianBlur(radius))  ->  ToTensor (uint8 / 255)
// The arithmetic is Pillow's (third-party; reference pins pillow==10.2.0), restated from its published algorithm and
// pinned bit-for-bit by fixtures produced with Pillow in the build container (tests/golden/degradation_golden.npz):
//   * resize (libImaging/Resample.c, 8-bit path): separable, horizontal pass first; per output pixel a bicubic
//     (a = -0.5) kernel stretched by the scale factor, support 2*scale, weights normalised in double precision and
//     quantised to 22-bit fixed point (round half away from zero); accumulator starts at 2^21, result
//     clip8(acc >> 22); the intermediate image is uint8.
//   * GaussianBlur (libImaging/BoxBlur.c): 3 horizontal then 3 vertical passes of a box filter whose fractional radius
//     comes from the Gaussian radius (float arithmetic), window weights ww = 2^24 / (2r+1) (float division, truncated)
//     and fw for the two far pixels, edge pixels replicated, every pass rounds to uint8:
//     (ww * sum + fw * (left + right) + 2^23) >> 24.
// All integer / byte work: results are bit-exact.  One thread per output byte per pass; the images are tiny next to
// the UNet step, the point is that the feed no longer runs PIL per item on the host's main thread.
#include <math.h>

#include <utility>

#include "drs_common.h"

namespace {

__device__ inline double bicubic_w(double x) {
#pragma clang fp contract(off)
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// one separable resize pass along the last axis of a (rows, in_size) byte image -> (rows, out_size); `stride_in` /
// `stride_out` are the element strides along that axis (1 for the horizontal pass, the row pitch for the vertical one)
__global__ void resize_pass_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, long long rows,
                                   int in_size, int out_size, long long line_in, long long line_out, long long stride_in,
                                   long long stride_out, long long outer, long long outer_in, long long outer_out) {
#pragma clang fp contract(off)
  // logical layout: element (o, r, x) at o*outer_* + r*line_* + x*stride_*  (o: plane, r: line, x: position on the axis)
  const long long total = outer * rows * out_size;
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const double ss = 1.0 / filterscale;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % out_size);
    const long long r = (i / out_size) % rows;
    const long long o = i / ((long long)out_size * rows);
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bicubic_w((x + xmin - center + 0.5) * ss);
    const unsigned char* src = in + o * outer_in + r * line_in + (long long)xmin * stride_in;
    int acc = 1 << 21;
    for (int x = 0; x < xmax; ++x) {
      double k = bicubic_w((x + xmin - center + 0.5) * ss);
      if (ww != 0.0) k /= ww;
      const int kk = k < 0 ? (int)(-0.5 + k * (double)(1 << 22)) : (int)(0.5 + k * (double)(1 << 22));
      acc += (int)src[(long long)x * stride_in] * kk;
    }
    int v = acc >> 22;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    out[o * outer_out + r * line_out + (long long)xx * stride_out] = (unsigned char)v;
  }
}

// one box-blur pass along an axis (same addressing scheme); in != out
__global__ void box_pass_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, long long rows,
                                int size, long long line, long long stride, long long outer, long long outer_stride,
                                int radius, unsigned ww, unsigned fw) {
  const long long total = outer * rows * size;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % size);
    const long long r = (i / size) % rows;
    const long long o = i / ((long long)size * rows);
    const unsigned char* src = in + o * outer_stride + r * line;
    unsigned acc = 0;
    for (int d = -radius; d <= radius; ++d) {
      const int p = min(max(x + d, 0), size - 1);
      acc += src[(long long)p * stride];
    }
    const int pl = min(max(x - radius - 1, 0), size - 1), pr = min(max(x + radius + 1, 0), size - 1);
    const unsigned far = (unsigned)src[(long long)pl * stride] + (unsigned)src[(long long)pr * stride];
    const unsigned bulk = acc * ww + far * fw;
    out[o * outer_stride + r * line + (long long)x * stride] = (unsigned char)((bulk + (1u << 23)) >> 24);
  }
}

__global__ void u8_to_unit_float_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = __fdiv_rn((float)in[i], 255.f);  // ToTensor: byte -> float32, true division by 255
}

inline unsigned blocks_for(long long n) {
  long long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// BoxBlur.c _gaussian_blur_radius (float variables, double intermediate where the C expression promotes)
float gaussian_box_radius(float radius, int passes) {
  float sigma2 = radius * radius / passes;
  float L = (float)sqrt(12.0 * sigma2 + 1.0);
  float l = (float)floor((L - 1.0) / 2.0);
  float a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
  a /= 6 * (sigma2 - (l + 1) * (l + 1));
  return l + a;
}

// x (N, C, H, W) += noise (N, H, W, C), clipped to [0, 1]: the device half of the `Gauss_noise=True` step (reference
// utils.py:15-38: `img += noise.astype(float32)` on the (H, W, C) view, `np.clip(img, 0, 1)`; the noise itself is drawn on the
// host from the reference's generators).  One float add and the clip per element: bit-exact with numpy's.
__global__ __launch_bounds__(256) void add_noise_clip_kernel(float* __restrict__ x, const float* __restrict__ noise, int C,
                                                             long long hw, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i % hw, nc = i / hw;
    const int c = (int)(nc % C);
    const long long n = nc / C;
    const float v = x[i] + noise[(n * hw + p) * C + c];
    x[i] = fminf(fmaxf(v, 0.f), 1.f);
  }
}
}  // namespace

extern "C" size_t drs_downblur_scratch_bytes(int N, int C, int H, int W, int out_h, int out_w) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return 0;
  // horizontally resized image (H x out_w) + two ping-pong LR byte images
  return (size_t)N * C * ((size_t)H * out_w + 2 * (size_t)out_h * out_w) + 768;
}

extern "C" int drs_downblur_u8(const uint8_t* hr, int N, int C, int H, int W, int out_h, int out_w, float blur_radius,
                               float* x_lr, float* y_hr, void* scratch, size_t scratch_bytes, drs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  DRS_REQUIRE(hr && x_lr && scratch, DRS_ERR_ARG, "downblur: null pointer");
  DRS_REQUIRE(N >= 1 && C >= 1 && H >= 1 && W >= 1 && out_h >= 1 && out_w >= 1, DRS_ERR_SHAPE,
              "downblur: N=%d C=%d H=%d W=%d -> %dx%d", N, C, H, W, out_h, out_w);
  DRS_REQUIRE(blur_radius >= 0.f && blur_radius < 64.f, DRS_ERR_ARG, "downblur: blur_radius=%f", blur_radius);
  DRS_REQUIRE(scratch_bytes >= drs_downblur_scratch_bytes(N, C, H, W, out_h, out_w), DRS_ERR_WORKSPACE,
              "downblur: scratch %zu too small", scratch_bytes);
  const long long planes = (long long)N * C;
  unsigned char* base = (unsigned char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  unsigned char* tmp = base;                                              // (planes, H, out_w)
  unsigned char* a = tmp + (((size_t)planes * H * out_w + 255) & ~(size_t)255);  // (planes, out_h, out_w)
  unsigned char* b = a + (((size_t)planes * out_h * out_w + 255) & ~(size_t)255);
  // resize: horizontal pass (skipped by Pillow when the width does not change), then vertical pass
  const unsigned char* hsrc = hr;
  int cur_w = W;
  if (out_w != W) {
    DRS_LAUNCH(resize_pass_kernel, dim3(blocks_for(planes * H * out_w)), dim3(256), 0, s, hr, tmp, (long long)H, W,
                       out_w, (long long)W, (long long)out_w, 1LL, 1LL, planes, (long long)H * W, (long long)H * out_w);
    hsrc = tmp;
    cur_w = out_w;
  }
  unsigned char* cur = a;
  if (out_h != H) {
    // lines = columns: "rows" = out_w columns, axis stride = row pitch
    DRS_LAUNCH(resize_pass_kernel, dim3(blocks_for(planes * out_w * out_h)), dim3(256), 0, s, hsrc, a, (long long)cur_w,
                       H, out_h, 1LL, 1LL, (long long)cur_w, (long long)out_w, planes, (long long)H * cur_w,
                       (long long)out_h * out_w);
  } else {
    DRS_CHECK_HIP(hipMemcpyAsync(a, hsrc, (size_t)planes * out_h * out_w, hipMemcpyDeviceToDevice, s));
  }
  if (blur_radius > 0.f) {
    const float fr = gaussian_box_radius(blur_radius, 3);
    if (fr != 0.f) {
      const int radius = (int)fr;
      const unsigned ww = (unsigned)((float)(1u << 24) / (fr * 2 + 1));
      const unsigned fw = ((1u << 24) - (unsigned)(radius * 2 + 1) * ww) / 2;
      unsigned char* other = b;
      for (int pass = 0; pass < 3; ++pass) {  // horizontal
        DRS_LAUNCH(box_pass_kernel, dim3(blocks_for(planes * out_h * out_w)), dim3(256), 0, s, cur, other,
                           (long long)out_h, out_w, (long long)out_w, 1LL, planes, (long long)out_h * out_w, radius, ww, fw);
        std::swap(cur, other);
      }
      for (int pass = 0; pass < 3; ++pass) {  // vertical
        DRS_LAUNCH(box_pass_kernel, dim3(blocks_for(planes * out_h * out_w)), dim3(256), 0, s, cur, other,
                           (long long)out_w, out_h, 1LL, (long long)out_w, planes, (long long)out_h * out_w, radius, ww, fw);
        std::swap(cur, other);
      }
    }
  }
  DRS_LAUNCH(u8_to_unit_float_kernel, dim3(blocks_for(planes * out_h * out_w)), dim3(256), 0, s, cur, x_lr,
                     planes * out_h * out_w);
  if (y_hr)
    DRS_LAUNCH(u8_to_unit_float_kernel, dim3(blocks_for(planes * H * W)), dim3(256), 0, s, hr, y_hr, planes * H * W);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

extern "C" int drs_add_noise_clip_f32(float* x, const float* noise_nhwc, int N, int C, int H, int W, drs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  DRS_REQUIRE(x && noise_nhwc, DRS_ERR_ARG, "add_noise_clip: null pointer");
  DRS_REQUIRE(N >= 0 && C >= 1 && H >= 1 && W >= 1, DRS_ERR_SHAPE, "add_noise_clip: N=%d C=%d H=%d W=%d", N, C, H, W);
  const long long hw = (long long)H * W, total = (long long)N * C * hw;
  if (total == 0) return DRS_OK;
  DRS_LAUNCH(add_noise_clip_kernel, dim3(blocks_for(total)), dim3(256), 0, s, x, noise_nhwc, C, hw, total);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
