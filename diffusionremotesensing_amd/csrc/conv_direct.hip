// fp32 VALU direct tap-convolution over channels-last activations: the on-device reference path
// (DRS_IMPL_DIRECT) and the fallback-free home of the odd shapes (Cout = 1 or 3).
// One lane = one logical output position; one block column = COT consecutive output channels, so the
// weight addresses are wave-uniform (scalar loads) and the activation vector of a pixel is read once
// per tap as float4s.
#include "drs_common.h"

// MASK: Cout is not a multiple of COT (2 or 3 output channels in a 4-wide tile): out-of-range columns are skipped.
template <int COT, bool MASK = (COT == 1)>
__global__ __launch_bounds__(256) void tapconv_direct_kernel(TapConv d) {
  const int64_t P = (int64_t)d.N * d.TH * d.TW;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int co0 = blockIdx.y * COT;
  if (p >= P) return;
  const int tx = (int)(p % d.TW);
  const int ty = (int)((p / d.TW) % d.TH);
  const int n = (int)(p / ((int64_t)d.TW * d.TH));

  float acc[COT];
#pragma unroll
  for (int j = 0; j < COT; ++j) acc[j] = 0.f;

  const float* addp = d.in_add ? d.in_add + (int64_t)n * d.in_add_cs : nullptr;
  for (int i = 0; i < d.ntaps; ++i) {
    const int iy = ty * d.in_stride + d.dy[i];
    const int ix = tx * d.in_stride + d.dx[i];
    if (iy < 0 || iy >= d.H || ix < 0 || ix >= d.W) continue;
    const float* ip = d.in + (((int64_t)n * d.H + iy) * d.W + ix) * d.in_cs + d.in_co;
    const float* wp = d.w + (int64_t)d.wtap[i] * d.Cin * d.Cout + co0;
    if ((d.Cin & 3) == 0 && (d.in_cs & 3) == 0 && (d.in_co & 3) == 0) {
      for (int ci = 0; ci < d.Cin; ci += 4) {
        float4 a = *reinterpret_cast<const float4*>(ip + ci);
        if (addp) {
          a.x += addp[ci];
          a.y += addp[ci + 1];
          a.z += addp[ci + 2];
          a.w += addp[ci + 3];
        }
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float* wr = wp + (int64_t)(ci + q) * d.Cout;
#pragma unroll
          for (int j = 0; j < COT; ++j)
            if (!MASK || co0 + j < d.Cout) acc[j] = fmaf(av[q], wr[j], acc[j]);
        }
      }
    } else {
      for (int ci = 0; ci < d.Cin; ++ci) {
        float a = ip[ci];
        if (addp) a += addp[ci];
        const float* wr = wp + (int64_t)ci * d.Cout;
#pragma unroll
        for (int j = 0; j < COT; ++j)
          if (!MASK || co0 + j < d.Cout) acc[j] = fmaf(a, wr[j], acc[j]);
      }
    }
  }

  const int oy = ty * d.out_scale + d.out_oy;
  const int ox = tx * d.out_scale + d.out_ox;
  float g = 1.f;
  if (d.gate) g = d.gate[((int64_t)n * (d.OH >> 1) + (oy >> 1)) * (d.OW >> 1) + (ox >> 1)];
  const int64_t opix = ((int64_t)n * d.OH + oy) * d.OW + ox;
  const int64_t rpix = d.res_bstride_zero ? ((int64_t)oy * d.OW + ox) : opix;
  float vals[COT];
  // channels-last output, aligned slice: the tile leaves as 16-byte stores (one 4-byte store per channel and lane cost the
  // 3 -> 32 channel data gradient of the `output` convolution 212 us per training step)
  const bool vec_store = !MASK && COT % 4 == 0 && !d.out_nchw && ((d.out_cs | d.out_co) & 3) == 0;
#pragma unroll
  for (int j = 0; j < COT; ++j) {
    const int co = co0 + j;
    if (co >= d.Cout) break;
    float v = acc[j];
    if (d.gate) v *= g;
    if (d.bias) v += d.bias[co];
    if (d.relu_pre) v = drs_maxf(v, 0.f);
    if (d.post_add) v += d.post_add[(int64_t)n * d.post_cs + co];
    if (d.res) v += d.res[rpix * d.res_cs + d.res_co + co];
    if (d.relu_post) v = drs_maxf(v, 0.f);
    if (d.sigmoid) v = 1.f / (1.f + expf(-v));
    vals[j] = v;
    if (vec_store) continue;
    if (d.out_nchw)
      d.out[(((int64_t)n * d.Cout + co) * d.OH + oy) * d.OW + ox] = v;
    else
      d.out[opix * d.out_cs + d.out_co + co] = v;
  }
  if (vec_store) {
    float* o = d.out + opix * d.out_cs + d.out_co + co0;
#pragma unroll
    for (int j = 0; j + 3 < COT; j += 4) *reinterpret_cast<float4*>(o + j) = make_float4(vals[j], vals[j + 1], vals[j + 2], vals[j + 3]);
  }
}

int drs_launch_tapconv_direct(const TapConv& d, hipStream_t s) {
  DRS_REQUIRE(d.in && d.w && d.out, DRS_ERR_ARG, "tapconv: null tensor");
  DRS_REQUIRE(d.ntaps >= 1 && d.ntaps <= DRS_MAX_TAPS, DRS_ERR_ARG, "tapconv: ntaps=%d", d.ntaps);
  const int64_t P = (int64_t)d.N * d.TH * d.TW;
  if (P == 0 || d.Cout == 0) return DRS_OK;
  dim3 block(256);
  if (d.Cout % 16 == 0) {
    dim3 grid((unsigned)((P + 255) / 256), d.Cout / 16);
    DRS_LAUNCH(tapconv_direct_kernel<16>, grid, block, 0, s, d);
  } else if (d.Cout % 4 == 0) {
    dim3 grid((unsigned)((P + 255) / 256), d.Cout / 4);
    DRS_LAUNCH(tapconv_direct_kernel<4>, grid, block, 0, s, d);
  } else if (d.Cout > 1 && d.Cout < 4) {  // image-channel outputs (2, 3): one pass over the input instead of Cout
    dim3 grid((unsigned)((P + 255) / 256), 1);
    DRS_LAUNCH((tapconv_direct_kernel<4, true>), grid, block, 0, s, d);
  } else {
    dim3 grid((unsigned)((P + 255) / 256), d.Cout);
    DRS_LAUNCH(tapconv_direct_kernel<1>, grid, block, 0, s, d);
  }
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
