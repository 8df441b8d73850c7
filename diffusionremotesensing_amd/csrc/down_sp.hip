// 3x3 stride-2 convolution of the 32-channel level (`downs.0`; built for 64 channels as well, not routed there: reference UNet_model_superres.py:366)
// with the input window staged in LDS.  conv_s2_sp.hip loads every tap of every output pixel straight into operand registers:
// 13 KB per wave and K-chunk, half of every 128-byte line requested twice (hi and lo halves by separate instructions), which
// is the L1's limit rather than HBM's (downs.0: 4.3 TB/s, downs.1: 2.5 TB/s).  Here a block owns 8 x 16 output pixels of one
// 32-channel output group: its 17 x 33 input pixels of one K-chunk (72 KB) travel global -> registers in whole 128-byte
// lines WHILE the previous window is multiplied, then registers -> LDS; the weights of the group (37 KB per K-chunk) stay
// in LDS for the lifetime of the block.  The window is stored as two column-parity planes (even columns | odd columns), so
// the stride-2 taps of 16 neighbouring output pixels are 16 neighbouring LDS lines (slot rotation (s + p) & 7 as in the
// other window kernels: conflict-free).  A wave owns one output row of the patch (16 pixels x 32 channels).
#include <stdio.h>
#include <stdlib.h>

#include "conv_epilogue.h"
#include "mfma_policy.h"

namespace {

constexpr int PH = 8, PW = 16, WH = 2 * PH + 1, WW = 2 * PW + 1;  // output patch, input window (17 x 33)
constexpr int NE = PW + 1, NO = PW;                                // even / odd columns of a window row
constexpr int EVEN_PIX = WH * NE, WPIX = WH * WW;                  // 289 even-column pixels, 561 in all
constexpr int WINB = WPIX * 128;                                   // 71808 bytes
constexpr int NLD = (WPIX * 8 + 511) / 512;                        // 16-byte pieces per lane and window: 9

template <int NCK>  // K-chunks = Cin / 32 = Cout / 32 (1 or 2)
__global__ __launch_bounds__(512, 1) void down_sp_kernel(TapConv d, unsigned w_gimage) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  using Frag = typename P::Frag;
  constexpr int IMG = NCK * 36 * 32 * 16;  // one operand image of the group's weights: [chunk][tap][k-group][32] slots
  char* sW = smem;
  char* sWin = smem + 2 * IMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kg = lane >> 4;
  const int ngroups = d.Cout / 32;
  const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3, nb8 = gridDim.x >> 3;
  const int grp = j8 % ngroups, member = j8 / ngroups, members = nb8 / ngroups;  // (the launcher makes nb8 a multiple of ngroups)
  const int n0 = grp * 32;
  {  // this group's weights: [image][chunk * 36 + tap * 4 + k-group][32 channels]
    const char* wsrc = reinterpret_cast<const char*>(d.w);
    for (int i = tid; i < 2 * NCK * 36 * 32; i += 512) {
      const int im = i >= NCK * 36 * 32, idx = im ? i - NCK * 36 * 32 : i;
      const int r = idx >> 5, j = idx & 31;
      *reinterpret_cast<u32x4*>(sW + (size_t)i * 16) =
          *reinterpret_cast<const u32x4*>(wsrc + (size_t)im * w_gimage + ((size_t)r * d.Cout + n0 + j) * 16);
    }
  }
  // patches of this XCD (a contiguous eighth: vertical neighbours share their boundary row in that L2)
  const int gx = d.OW / PW, gy = d.OH / PH;
  const int total = d.N * gy * gx;
  const int t_lo = (int)((long long)total * xcd / 8), t_hi = (int)((long long)total * (xcd + 1) / 8);
  const int my = member < t_hi - t_lo ? (t_hi - t_lo - member + members - 1) / members : 0;  // patches of this block
  const int S = my * NCK;  // steps: one K-chunk of one patch each
  const char* zero = reinterpret_cast<const char*>(d.zero_line);
  const char* in = reinterpret_cast<const char*>(d.in) + (size_t)d.in_co * 4;
  const int pixb = d.in_cs * 4;
  // window pieces of this lane: piece e = 512 i + tid: slot e & 7, window pixel e >> 3 (row-major over 17 x 33)
  u32x4 xr[NLD];
  auto coords = [&](int s, int& n, int& oy0, int& ox0, int& c) __attribute__((always_inline)) {
    const int it = s / NCK;
    c = s - it * NCK;
    const int q = t_lo + member + it * members;
    ox0 = (q % gx) * PW; oy0 = ((q / gx) % gy) * PH; n = q / (gx * gy);
  };
  auto win_load = [&](int s) __attribute__((always_inline)) {
    int n, oy0, ox0, c;
    coords(s, n, oy0, ox0, c);
    const char* base = in + (size_t)n * d.H * d.W * pixb + c * 128;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = i * 512 + tid, wp = e >> 3;
      const int wy = (wp * 1986) >> 16, wx = wp - wy * WW;  // wp / 33 (exact for wp < 1024)
      const int iy = 2 * oy0 - 1 + wy, ix = 2 * ox0 - 1 + wx;
      const bool ok = wp < WPIX && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
      xr[i] = *reinterpret_cast<const u32x4*>(ok ? base + ((size_t)iy * d.W + ix) * pixb + (e & 7) * 16 : zero);
    }
  };
  auto win_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = i * 512 + tid, wp = e >> 3, sl = e & 7;
      const int wy = (wp * 1986) >> 16, wx = wp - wy * WW;
      const int p = (wx & 1) ? EVEN_PIX + wy * NO + (wx >> 1) : wy * NE + (wx >> 1);  // parity plane, then row-major
      if (wp < WPIX) *reinterpret_cast<u32x4*>(sWin + p * 128 + ((sl + p) & 7) * 16) = xr[i];
    }
  };
  // operand of output pixel (row `wave`, column lr), tap (ky, kx): window pixel (2 * wave + ky, 2 * lr + kx)
  auto operand = [&](int ky, int kx) __attribute__((always_inline)) {
    const int wy = 2 * wave + ky;
    const int p = kx == 1 ? EVEN_PIX + wy * NO + lr : wy * NE + lr + (kx >> 1);
    const char* line = sWin + p * 128;
    const int s0 = ((kg + p) & 7) * 16;
    return Frag{*reinterpret_cast<const bf16x8*>(line + s0), *reinterpret_cast<const bf16x8*>(line + (s0 ^ 64))};
  };
  if (S == 0) return;
  float bias8[8];
  {
    const float4 a = d.bias ? *reinterpret_cast<const float4*>(d.bias + n0 + kg * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b = d.bias ? *reinterpret_cast<const float4*>(d.bias + n0 + kg * 8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    bias8[0] = a.x; bias8[1] = a.y; bias8[2] = a.z; bias8[3] = a.w; bias8[4] = b.x; bias8[5] = b.y; bias8[6] = b.z; bias8[7] = b.w;
  }
  win_load(0);
  const char* wlane = sW + ((size_t)kg * 32 + lr) * 16;
  f32x4 acc[1][2];
  for (int s = 0; s < S; ++s) {
    int n, oy0, ox0, c;
    coords(s, n, oy0, ox0, c);
    __syncthreads();  // every wave has left the previous window (first step: the weights are in place)
    win_store();
    if (s + 1 < S) win_load(s + 1);  // in flight under this step's MFMAs (and the epilogue's stores: issued before them)
    __syncthreads();
    if (c == 0) { acc[0][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[0][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const char* wc = wlane + (size_t)c * 36 * 32 * 16;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const Frag a = operand(ky, kx);
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[0][t] = P::mma(P::load(wc, (size_t)IMG, (size_t)((ky * 3 + kx) * 4 * 32 + t * 16) * 16), a, acc[0][t]);
      }
    if (c == NCK - 1) {
      SpEpiConst kc;
#pragma unroll
      for (int j = 0; j < 8; ++j) { kc.bias[j] = bias8[j]; kc.post[j] = 0.f; kc.post2[j] = 0.f; }
      tile_epilogue_sp_pre<1, false>(d, acc, kc, n, n0, oy0, ox0, wave, lr, kg);
    }
  }
}

}  // namespace

// DRS_DOWNK=0 keeps conv_s2_sp_kernel on these layers.
bool drs_down_sp_supported(const TapConv& d, int impl) {
  static const int env = getenv("DRS_DOWNK") ? atoi(getenv("DRS_DOWNK")) : 1;
  if (!env || !drs_conv_s2_sp_supported(d, impl)) return false;
  if (d.Cin != d.Cout || d.Cin != 32 || d.relu_pre || d.relu_post) return false;  // (64 channels: built, measured 34 vs 33 us: not routed here)
  return d.OH % PH == 0 && d.OW % PW == 0 && d.OH >= 2 * PH && d.OW >= 2 * PW;
}

int drs_launch_down_sp(const TapConv& d, hipStream_t s) {
  const int nck = d.Cin / 32;
  auto kern = nck == 1 ? down_sp_kernel<1> : down_sp_kernel<2>;
  const size_t lds = (size_t)2 * nck * 36 * 32 * 16 + WINB;
  int num_cu = 0;
  {
    const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(kern), 160 * 1024, &num_cu);
    if (rc) return rc;
  }
  const unsigned w_gimage = (unsigned)((size_t)nck * 9 * 4 * d.Cout * 16);
  const int ngroups = d.Cout / 32;
  int per_xcd = num_cu / 8;  // one block per CU; blocks of an XCD split into the channel groups
  per_xcd = per_xcd / ngroups * ngroups;
  if (per_xcd < ngroups) per_xcd = ngroups;
  DRS_LAUNCH(kern, dim3((unsigned)(8 * per_xcd)), dim3(512), lds, s, d, w_gimage);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
