// The first residual block of the encoder as ONE launch (reference ResConvBlock.forward, UNet_model_superres.py:153-172,
// block 0 with its skip convolution :353-355):
//   h   = relu(BN1(conv1(x))) + relu(time_mlp(t)) + skip(x)        16 -> 32 channels, 3x3 + 3x3
//   out = relu(BN2(conv2(h)) + BNs(shortcut(x)))                    32 -> 32 channels, 3x3 + 1x1
// Run as two launches (conv3x3_direct_sp.hip: the pair flavour, then conv2 with the shortcut as one more tap) h crosses
// HBM twice at the highest resolution: 134 MB written and read back of the 600 MB the two launches move, and both run at a
// third of the MFMA rate.  Here a block owns a 16 x 16 output patch:
//   phase A  h on the patch's 18 x 18 window (the halo is recomputed: 27 % more conv1 work), 21 MFMA pixel blocks of 16
//            consecutive window pixels spread over the 8 waves.  Operands come straight from global memory (x is 64 bytes
//            per pixel; a lane gathers the pixel of ITS tap: two taps of 16 channels share the K = 32 of an MFMA, k-groups
//            0-1 tap 2j, k-groups 2-3 tap 2j + 1, as in the pair flavour of the direct kernel); the result is split into
//            bf16 hi | lo - the rounding the SP tensor between the two launches had - and written into an LDS window in
//            the rotated pixel-major layout of the wave-specialised kernel (slot s of pixel p at position (s + p) & 7).
//            Window pixels outside the image are conv2's zero padding: zeros, not conv1 of padded x.
//   phase B  conv2 from the LDS window (shifted taps are shifted LDS addresses: no lane shuffles), the 1x1 shortcut of x as
//            one more tap, ReLU, SP store.  A wave owns two rows of the patch.
// All weights (41 + 37 + 4 KB) stay in LDS for the lifetime of the persistent block.  Two barriers per patch.
#include <stdio.h>
#include <stdlib.h>

#include "conv_epilogue.h"
#include "mfma_policy.h"

namespace {

constexpr int TS = 16, IW = TS + 2, NPIX = IW * IW, NBLK_A = (NPIX + 15) / 16;  // 324 window pixels, 21 pixel blocks
constexpr int IMG1 = 5 * 4 * 64 * 16, IMG2 = 36 * 32 * 16, IMGS = 4 * 32 * 16, WIN = NPIX * 128;
constexpr int NPOST = 64;  // images whose time-embedding rows are staged in LDS (later images read them from memory)

__global__ __launch_bounds__(512, 1) void resblock0_kernel(ResBlock0Desc d, unsigned w1_gimage, unsigned w2_gimage,
                                                           unsigned ws_gimage) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  using Frag = typename P::Frag;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kg = lane >> 4;
  char* sW1 = smem;               // [image 2][tap pair 5][k-group 4][64 channels: conv1 | skip] slots of 16 bytes
  char* sW2 = sW1 + 2 * IMG1;     // [image 2][tap 9][k-group 4][32]
  char* sWs = sW2 + 2 * IMG2;     // [image 2][k-group 4][32]
  char* sWin = sWs + 2 * IMGS;    // h window: pixel p = py * 18 + px at p * 128, operand slot s at ((s + p) & 7) * 16
  float* sB1 = reinterpret_cast<float*>(sWin + WIN);  // [64] conv1 | skip biases
  float* sB2 = sB1 + 64;                              // [32] conv2 + shortcut bias
  float* sPost = sB2 + 32;                            // [min(N, NPOST)][32] relu(time_mlp(t)) rows
  const int npost = min(d.N, NPOST);
  {
    const char* w1 = reinterpret_cast<const char*>(d.w1);
    for (int i = tid; i < 2 * 20 * 64; i += 512) {  // tap-major packed source -> two taps per K = 32 (the ninth pairs with zeros)
      const int im = i >= 20 * 64, idx = im ? i - 20 * 64 : i;
      const int ch = idx % 64, kgp = (idx / 64) & 3, j = idx / (4 * 64), tap = 2 * j + (kgp >> 1);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (tap < 9) v = *reinterpret_cast<const u32x4*>(w1 + (size_t)im * w1_gimage + ((size_t)(tap * 4 + (kgp & 1)) * 64 + ch) * 16);
      *reinterpret_cast<u32x4*>(sW1 + (size_t)im * IMG1 + (size_t)idx * 16) = v;
    }
    auto copy = [&](char* dst, const char* src, int bytes) __attribute__((always_inline)) {
      for (int o = tid * 16; o < bytes; o += 512 * 16) *reinterpret_cast<u32x4*>(dst + o) = *reinterpret_cast<const u32x4*>(src + o);
    };
    const char* w2 = reinterpret_cast<const char*>(d.w2);
    const char* wsc = reinterpret_cast<const char*>(d.ws);
    copy(sW2, w2, IMG2);
    copy(sW2 + IMG2, w2 + w2_gimage, IMG2);
    copy(sWs, wsc, IMGS);
    copy(sWs + IMGS, wsc + ws_gimage, IMGS);
    for (int i = tid; i < 64; i += 512) sB1[i] = d.b1[i];
    for (int i = tid; i < 32; i += 512) sB2[i] = d.b2[i] + d.bs[i];
    for (int i = tid; i < npost * 32; i += 512) sPost[i] = d.temb[(size_t)(i >> 5) * d.temb_cs + (i & 31)];
  }
  __syncthreads();

  // patches: blocks of an XCD (blockIdx % 8) take a contiguous eighth (neighbouring patches share their halo in that L2)
  const int gx = d.W / TS, gy = d.H / TS;
  const int total = d.N * gy * gx;
  const int xcd = blockIdx.x & 7, member = blockIdx.x >> 3, members = gridDim.x >> 3;
  const int t_lo = (int)((long long)total * xcd / 8), t_hi = (int)((long long)total * (xcd + 1) / 8);
  const char* zero = reinterpret_cast<const char*>(d.zero_line);
  const char* xb = reinterpret_cast<const char*>(d.x);
  // this lane's tap of pair j: k-groups 0-1 multiply tap 2j, k-groups 2-3 tap 2j + 1 (pair 4: tap 8 and zero weights)
  const int kc16 = (kg & 1) * 16;  // byte offset of the lane's 8 channels inside the 32-byte hi (or lo) half of a pixel
  int tdy[5], tdx[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int tap = min(2 * j + (kg >> 1), 8);
    tdy[j] = tap / 3 - 1; tdx[j] = tap % 3 - 1;
  }
  constexpr int MAXB = (NBLK_A + 7) / 8;  // pixel blocks of a wave in phase A: blocks wave, wave + 8, wave + 16
  const int nblk = wave + 16 < NBLK_A ? 3 : 2;  // (wave-uniform)

  for (int q = t_lo + member; q < t_hi; q += members) {
    const int tx0 = (q % gx) * TS, ty0 = ((q / gx) % gy) * TS, n = q / (gx * gy);
    const char* ximg = xb + (size_t)n * d.H * d.W * 64;
    // ================= phase A: h on the 18 x 18 window =================
    int wy[MAXB], wx[MAXB], wp[MAXB];  // this lane's window pixel per block: image coordinates, linear window index
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
      const int p = min((wave + 8 * b) * 16 + lr, NPIX - 1);
      const int py = (p * 3641) >> 16, px = p - py * IW;  // p / 18 (exact for p < 1024)
      wp[b] = p; wy[b] = ty0 - 1 + py; wx[b] = tx0 - 1 + px;
    }
    auto gather = [&](int b, int j) __attribute__((always_inline)) {
      const int sy = wy[b] + tdy[j], sx = wx[b] + tdx[j];
      const bool ok = sy >= 0 && sy < d.H && sx >= 0 && sx < d.W;
      const char* p = ok ? ximg + ((size_t)sy * d.W + sx) * 64 + kc16 : zero;
      return Frag{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(ok ? p + 32 : zero)};
    };
    f32x4 acc[MAXB][4];
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frag a[2][MAXB];
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
      if (b < nblk) a[0][b] = gather(b, 0);
    const char* w1lane = sW1 + ((size_t)kg * 64 + lr) * 16;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j + 1 < 5) {
#pragma unroll
        for (int b = 0; b < MAXB; ++b)
          if (b < nblk) a[(j + 1) & 1][b] = gather(b, j + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const Frag wf = P::load(w1lane, (size_t)IMG1, (size_t)(j * 4 * 64 + t * 16) * 16);
#pragma unroll
        for (int b = 0; b < MAXB; ++b)
          if (b < nblk) acc[b][t] = P::mma(wf, a[j & 1][b], acc[b][t]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the shortcut operand of phase B (the centre pixels of this wave's two rows): in flight across the barriers
    Frag xs[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const char* p = kg < 2 ? ximg + ((size_t)(ty0 + 2 * wave + r) * d.W + tx0 + lr) * 64 + kc16 : zero;
      xs[r] = Frag{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(kg < 2 ? p + 32 : zero)};
    }
    // h = relu(conv1 + b1) + temb[n] + (skip + bs): tiles 0, 1 (conv1) and 2, 3 (skip) of the same lane are the same channels
    float bm[8], bs[8], te[8];
    {
      const float4 m0 = *reinterpret_cast<const float4*>(sB1 + kg * 8), m1 = *reinterpret_cast<const float4*>(sB1 + kg * 8 + 4);
      const float4 s0 = *reinterpret_cast<const float4*>(sB1 + 32 + kg * 8), s1 = *reinterpret_cast<const float4*>(sB1 + 32 + kg * 8 + 4);
      const float* tp = n < npost ? sPost + n * 32 + kg * 8 : nullptr;
      float4 e0, e1;
      if (tp) { e0 = *reinterpret_cast<const float4*>(tp); e1 = *reinterpret_cast<const float4*>(tp + 4); }
      else {
        const float* gp = d.temb + (size_t)n * d.temb_cs + kg * 8;
        e0 = *reinterpret_cast<const float4*>(gp); e1 = *reinterpret_cast<const float4*>(gp + 4);
      }
      bm[0] = m0.x; bm[1] = m0.y; bm[2] = m0.z; bm[3] = m0.w; bm[4] = m1.x; bm[5] = m1.y; bm[6] = m1.z; bm[7] = m1.w;
      bs[0] = s0.x; bs[1] = s0.y; bs[2] = s0.z; bs[3] = s0.w; bs[4] = s1.x; bs[5] = s1.y; bs[6] = s1.z; bs[7] = s1.w;
      te[0] = e0.x; te[1] = e0.y; te[2] = e0.z; te[3] = e0.w; te[4] = e1.x; te[5] = e1.y; te[6] = e1.z; te[7] = e1.w;
    }
    u32x4 hh[MAXB], hl[MAXB];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
      float v[8];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          v[t * 4 + i] = (fmaxf(acc[b][t][i] + bm[t * 4 + i], 0.f) + (acc[b][t + 2][i] + bs[t * 4 + i])) + te[t * 4 + i];  // (the two-launch path's order)
      drs_sp_split8(v, hh[b], hl[b]);
      const bool inside = wy[b] >= 0 && wy[b] < d.H && wx[b] >= 0 && wx[b] < d.W;
      if (!inside) { hh[b] = u32x4{0u, 0u, 0u, 0u}; hl[b] = u32x4{0u, 0u, 0u, 0u}; }
    }
    __syncthreads();  // every wave has finished reading the previous patch's window
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
      if (b < nblk && (wave + 8 * b) * 16 + lr < NPIX) {
        char* wpix = sWin + wp[b] * 128;
        *reinterpret_cast<u32x4*>(wpix + ((kg + wp[b]) & 7) * 16) = hh[b];
        *reinterpret_cast<u32x4*>(wpix + ((4 + kg + wp[b]) & 7) * 16) = hl[b];
      }
    __syncthreads();  // the window is complete
    // ================= phase B: out = relu(conv2(h) + shortcut(x) + b) on rows 2 * wave, 2 * wave + 1 =================
    f32x4 o[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int t = 0; t < 2; ++t) o[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* w2lane = sW2 + ((size_t)kg * 32 + lr) * 16;
    auto win = [&](int row, int kx) __attribute__((always_inline)) {
      const int p = (2 * wave + row) * IW + lr + kx;
      const char* wpix = sWin + p * 128;
      return Frag{*reinterpret_cast<const bf16x8*>(wpix + ((kg + p) & 7) * 16), *reinterpret_cast<const bf16x8*>(wpix + ((4 + kg + p) & 7) * 16)};
    };
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const Frag a0 = win(ky, kx), a1 = win(ky + 1, kx);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const Frag wf = P::load(w2lane, (size_t)IMG2, (size_t)((ky * 3 + kx) * 4 * 32 + t * 16) * 16);
          o[0][t] = P::mma(wf, a0, o[0][t]);
          o[1][t] = P::mma(wf, a1, o[1][t]);
        }
      }
    }
    {
      const char* wslane = sWs + ((size_t)kg * 32 + lr) * 16;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const Frag wf = P::load(wslane, (size_t)IMGS, (size_t)(t * 16) * 16);
        o[0][t] = P::mma(wf, xs[0], o[0][t]);
        o[1][t] = P::mma(wf, xs[1], o[1][t]);
      }
    }
    {
      SpEpiConst kc;
      const float4 b0 = *reinterpret_cast<const float4*>(sB2 + kg * 8), b1 = *reinterpret_cast<const float4*>(sB2 + kg * 8 + 4);
      kc.bias[0] = b0.x; kc.bias[1] = b0.y; kc.bias[2] = b0.z; kc.bias[3] = b0.w;
      kc.bias[4] = b1.x; kc.bias[5] = b1.y; kc.bias[6] = b1.z; kc.bias[7] = b1.w;
#pragma unroll
      for (int i = 0; i < 8; ++i) { kc.post[i] = 0.f; kc.post2[i] = 0.f; }
      TapConv de = {};
      de.out = d.out; de.out_cs = 32; de.out_co = 0;
      de.OH = d.H; de.OW = d.W; de.TH = d.H; de.TW = d.W;
      de.relu_post = 1;
      tile_epilogue_sp_pre<2, false>(de, o, kc, n, 0, ty0, tx0, wave, lr, kg);
    }
  }
}

size_t resblock0_lds(int N) { return (size_t)2 * (IMG1 + IMG2 + IMGS) + WIN + (size_t)(64 + 32 + 32 * (N < NPOST ? N : NPOST)) * 4; }

}  // namespace

// Shape gate: the 16 -> 32 -> 32 block on images that split into 16 x 16 patches.  DRS_RB0=0 keeps the two launches.
bool drs_resblock0_supported(int Cin, int Cout, int H, int W) {
  static const bool env = !(getenv("DRS_RB0") && atoi(getenv("DRS_RB0")) == 0);
  return env && Cin == 16 && Cout == 32 && H >= 32 && W >= 32 && H % TS == 0 && W % TS == 0;
}

int drs_launch_resblock0(const ResBlock0Desc& d, hipStream_t s) {
  DRS_REQUIRE(drs_resblock0_supported(16, 32, d.H, d.W), DRS_ERR_SHAPE, "resblock0: H=%d W=%d", d.H, d.W);
  DRS_REQUIRE(d.x && d.w1 && d.b1 && d.temb && d.w2 && d.b2 && d.ws && d.bs && d.out && d.zero_line, DRS_ERR_ARG, "resblock0: null pointer");
  if (d.N == 0) return DRS_OK;
  int num_cu = 0;
  {
    const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(resblock0_kernel), 160 * 1024, &num_cu);
    if (rc) return rc;
  }
  // global operand images (drs_launch_pack_conv_mfma): [chunk 1][tap][k-group 4][channels] slots, one image per bf16 half
  const unsigned w1_gimage = 9u * 4u * 64u * 16u, w2_gimage = 9u * 4u * 32u * 16u, ws_gimage = 4u * 32u * 16u;
  const int blocks = num_cu / 8 * 8;  // one block per CU
  hipLaunchKernelGGL(resblock0_kernel, dim3((unsigned)blocks), dim3(512), resblock0_lds(d.N), s, d, w1_gimage, w2_gimage, ws_gimage);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
