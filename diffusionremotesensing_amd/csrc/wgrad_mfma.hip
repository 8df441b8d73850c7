// Weight gradients on the matrix cores (exact fp32: v_mfma_f32_16x16x4_f32).
//
// dW[f][v][tap] = sum over (n, ty, tx) of F[n, ty, tx, f] * V[n, ty*sv + oy[tap], tx*sv + ox[tap], v]
//
// A GEMM whose reduction dimension is the pixel index (16 x 256 x 256 = 1M at the top level) and whose output is tiny
// (32 x 32 x 9 ... 256 x 256 x 9).  "F" is the side that is read at the iteration position itself (dY for a
// convolution, the layer input for the transposed convolution), "V" the side read through the tap offsets and the
// stride (layer input for a convolution, dY for the transposed convolution); drs_launch_wgrad maps WgradDesc's
// A / B tensors onto them.  Replaces autograd's conv weight-gradient kernels behind
// `train_loss.backward()` (reference train_diffusion_superres.py:392).
//
// Work split: grid.y = (f-tile, v-tile) pairs, grid.x = pixel splits.  A block walks its share of 4x16-position
// tiles: both sides of a tile are staged to LDS once (channels-last rows, zero outside the image = the convolution's
// padding), every wave keeps MW x NW x NT accumulators (16x16 blocks, all taps) in registers and issues one MFMA per
// (tap, block) and 4-pixel K-step; operands are single ds_read_b32 with a pixel stride of 16 (mod 32) floats, which is
// conflict-free for the 4 pixel rows x 16 channels a wavefront reads.  At the end every wave-group writes its partial
// dW slice ([tap][f][v], coalesced) to a workspace and wgrad_reduce_kernel folds the slices into dW: no 1000-way
// contended float atomics on a 9216-element output.
//
// HBM traffic per tile is (F + V window) x channels x 4 B against 2 * CHF * CHV * NT * 64 flops: 3x3 layers are
// MFMA-bound (157 TFLOP/s fp32 peak), 1x1 layers HBM-bound (8 flop/B at 32x32 channels).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "drs_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int RT = 4, CT = 16;  // iteration-domain tile (positions)

struct WgradMfmaArgs {
  const float* F; int f_cs, f_co, Cf;
  const float* V; int v_cs, v_co, Cv, VH, VW, sv;
  int N, TH, TW;
  int oy[DRS_MAX_TAPS], ox[DRS_MAX_TAPS];
  int ymin, xmin, WR, WC;        // V window of one tile: origin offset and size (pixels)
  unsigned wc_magic;             // ceil(2^32 / WC): p / WC for p < 2^16
  const float* v_add; int v_add_cs;
  const float* v_gate;
  int tiles_x, tiles_y, ntiles;
  int ctv;                       // number of v-tiles (grid.y = ctf * ctv)
  float* partial; long long slice_stride;  // floats per slice: NT*Cf*Cv (+ Cf when the bias gradient rides along)
  int f_scalar, v_scalar;        // channel counts / strides not float4-able (3-channel images): scalar staging loads
  int bias;                      // also produce sum over positions of F[., f] (bias gradient of a convolution)
};

__host__ __device__ constexpr int pix_stride(int ch) { return (ch % 32 == 16) ? ch : ch + 16; }

template <int NT, int MW, int NW, int WM, int WN, int WK>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradMfmaArgs g) {
  static_assert(WM * WN * WK == 4, "4 waves per block");
  constexpr int CHF = WM * MW * 16, CHV = WN * NW * 16;
  constexpr int PSF = pix_stride(CHF), PSV = pix_stride(CHV);
  constexpr int QF = CHF / 4, QV = CHV / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sF = lds;
  float* sV = lds + RT * CT * PSF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
  const int f0 = ((int)blockIdx.y / g.ctv) * CHF, v0 = ((int)blockIdx.y % g.ctv) * CHV;
  const int l16 = lane & 15, kq = lane >> 4;

  f32x4 acc[NT][MW][NW];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // bias gradient: one more MFMA per K-step against a vector of ones (only the v-tile-0, wn == 0 waves)
  const bool do_bias = g.bias && v0 == 0 && wn == 0;
  f32x4 accb[MW];
#pragma unroll
  for (int i = 0; i < MW; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  int tapoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) tapoff[t] = ((g.oy[t] - g.ymin) * g.WC + (g.ox[t] - g.xmin)) * PSV;

  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    const int tx0 = (tile % g.tiles_x) * CT;
    const int ty0 = ((tile / g.tiles_x) % g.tiles_y) * RT;
    const int n = tile / (g.tiles_x * g.tiles_y);
    __syncthreads();  // every wave is done reading the previous tile
    // ---- stage F: RT x CT positions, CHF channels (zero outside the domain) ----
    if (!g.f_scalar) {
      for (int i = tid; i < RT * CT * QF; i += 256) {
        const int q = i % QF, p = i / QF;
        const int y = ty0 + p / CT, x = tx0 + p % CT, c = f0 + q * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y < g.TH && x < g.TW && c < g.Cf)
          v = *reinterpret_cast<const f32x4*>(g.F + (((long long)n * g.TH + y) * g.TW + x) * g.f_cs + g.f_co + c);
        *reinterpret_cast<f32x4*>(sF + p * PSF + q * 4) = v;
      }
    } else {
      for (int i = tid; i < RT * CT * CHF; i += 256) {
        const int ch = i % CHF, p = i / CHF;
        const int y = ty0 + p / CT, x = tx0 + p % CT, c = f0 + ch;
        float v = 0.f;
        if (y < g.TH && x < g.TW && c < g.Cf) v = g.F[(((long long)n * g.TH + y) * g.TW + x) * g.f_cs + g.f_co + c];
        sF[p * PSF + ch] = v;
      }
    }
    // ---- stage the V window: WR x WC pixels, CHV channels (zero outside the image: the convolution's padding) ----
    const int ybase = ty0 * g.sv + g.ymin, xbase = tx0 * g.sv + g.xmin;
    if (!g.v_scalar) {
      const int wtotal = g.WR * g.WC * QV;
      for (int i = tid; i < wtotal; i += 256) {
        const int q = i % QV, p = i / QV;
        const int wy = (int)__umulhi((unsigned)p, g.wc_magic), wx = p - wy * g.WC;
        const int y = ybase + wy, x = xbase + wx, c = v0 + q * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < g.VH && x >= 0 && x < g.VW && c < g.Cv) {
          v = *reinterpret_cast<const f32x4*>(g.V + (((long long)n * g.VH + y) * g.VW + x) * g.v_cs + g.v_co + c);
          if (g.v_add) v += *reinterpret_cast<const f32x4*>(g.v_add + (long long)n * g.v_add_cs + c);
          if (g.v_gate) v *= g.v_gate[((long long)n * (g.VH >> 1) + (y >> 1)) * (g.VW >> 1) + (x >> 1)];
        }
        *reinterpret_cast<f32x4*>(sV + p * PSV + q * 4) = v;
      }
    } else {
      const int wtotal = g.WR * g.WC * CHV;
      for (int i = tid; i < wtotal; i += 256) {
        const int ch = i % CHV, p = i / CHV;
        const int wy = (int)__umulhi((unsigned)p, g.wc_magic), wx = p - wy * g.WC;
        const int y = ybase + wy, x = xbase + wx, c = v0 + ch;
        float v = 0.f;
        if (y >= 0 && y < g.VH && x >= 0 && x < g.VW && c < g.Cv) {
          v = g.V[(((long long)n * g.VH + y) * g.VW + x) * g.v_cs + g.v_co + c];
          if (g.v_add) v += g.v_add[(long long)n * g.v_add_cs + c];
          if (g.v_gate) v *= g.v_gate[((long long)n * (g.VH >> 1) + (y >> 1)) * (g.VW >> 1) + (x >> 1)];
        }
        sV[p * PSV + ch] = v;
      }
    }
    __syncthreads();
    // ---- 16 K-steps of 4 positions (one row segment each); this wave-group takes every WK-th ----
    for (int ks = wk; ks < RT * CT / 4; ks += WK) {
      const int r = ks >> 2, x = (ks & 3) * 4 + kq;
      float a[MW];
#pragma unroll
      for (int i = 0; i < MW; ++i) a[i] = sF[(r * CT + x) * PSF + (wm * MW + i) * 16 + l16];
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < MW; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], 1.0f, accb[i], 0, 0, 0);
      }
      const float* vb = sV + ((r * g.sv) * g.WC + x * g.sv) * PSV + wn * NW * 16 + l16;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float b[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) b[j] = vb[tapoff[t] + j * 16];
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
          for (int j = 0; j < NW; ++j)
            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[t][i][j], 0, 0, 0);
      }
    }
  }
  // ---- partial slice of this wave-group: P[tap][f][v]; lane holds D[f = 4*kq + i][v = l16] of each 16x16 block ----
  float* P = g.partial + (long long)((int)blockIdx.x * WK + wk) * g.slice_stride;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int v = v0 + (wn * NW + j) * 16 + l16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = f0 + (wm * MW + i) * 16 + 4 * kq + e;
          if (f < g.Cf && v < g.Cv) P[((long long)t * g.Cf + f) * g.Cv + v] = acc[t][i][j][e];
        }
      }
  if (do_bias && l16 == 0) {
    float* Pb = P + (long long)NT * g.Cf * g.Cv;
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = f0 + (wm * MW + i) * 16 + 4 * kq + e;
        if (f < g.Cf) Pb[f] = accb[i][e];
      }
  }
}

struct ReduceArgs {
  const float* partial; long long slice_stride; int nslices, Cf, Cv, NT;
  int wtap[DRS_MAX_TAPS]; int of, ov, T_total; float* dW;
  float* dbias;  // non-null: the slices carry Cf more floats (sum of F over positions)
};
// dW[(f*of + v*ov)*T + wtap[tap]] += sum over slices of P[slice][tap][f][v].  grid.y splits the slices; a handful of
// float atomics per output element remain (one per grid.y), none when grid.y == 1.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(ReduceArgs r, int slices_per_y) {
  const long long numel = (long long)r.NT * r.Cf * r.Cv;
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (j >= numel + (r.dbias ? r.Cf : 0)) return;
  const int s0 = blockIdx.y * slices_per_y, s1 = min(r.nslices, s0 + slices_per_y);
  float s = 0.f;
  const float* p = r.partial + (long long)s0 * r.slice_stride + j;
  int k = s0;
  for (; k + 4 <= s1; k += 4, p += 4 * r.slice_stride)
    s += (p[0] + p[r.slice_stride]) + (p[2 * r.slice_stride] + p[3 * r.slice_stride]);
  for (; k < s1; ++k, p += r.slice_stride) s += p[0];
  if (j >= numel) {
    float* dst = r.dbias + (j - numel);
    if (gridDim.y == 1) *dst += s; else atomicAdd(dst, s);
    return;
  }
  const int v = (int)(j % r.Cv), f = (int)((j / r.Cv) % r.Cf), tap = (int)(j / ((long long)r.Cv * r.Cf));
  int wt = 0;
#pragma unroll
  for (int t = 0; t < DRS_MAX_TAPS; ++t) wt = (t == tap) ? r.wtap[t] : wt;
  float* dst = r.dW + ((long long)f * r.of + (long long)v * r.ov) * r.T_total + wt;
  if (gridDim.y == 1) *dst += s; else atomicAdd(dst, s);
}

template <int NT, int MW, int NW, int WM, int WN, int WK>
int launch_cfg(const WgradMfmaArgs& a0, int target_blocks, size_t partial_bytes, int* nslices, hipStream_t s) {
  WgradMfmaArgs a = a0;
  constexpr int CHF = WM * MW * 16, CHV = WN * NW * 16;
  const int ctf = (a.Cf + CHF - 1) / CHF;
  a.ctv = (a.Cv + CHV - 1) / CHV;
  const int ct = ctf * a.ctv;
  long long ksplit = target_blocks / ct;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > a.ntiles) ksplit = a.ntiles;
  const long long max_slices = (long long)(partial_bytes / ((size_t)a.slice_stride * 4));
  if (max_slices < WK) return -1;  // workspace too small for even one slice per wave-group
  if (ksplit * WK > max_slices) ksplit = max_slices / WK;
  const size_t lds = (size_t)(RT * CT * pix_stride(CHF) + a.WR * a.WC * pix_stride(CHV)) * 4;
  if (lds > 160 * 1024) return -1;
  auto kern = wgrad_mfma_kernel<NT, MW, NW, WM, WN, WK>;
  if (lds > 64 * 1024) DRS_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  DRS_LAUNCH(kern, dim3((unsigned)ksplit, ct), dim3(256), lds, s, a);
  DRS_CHECK_HIP(hipGetLastError());
  *nslices = (int)ksplit * WK;
  return DRS_OK;
}

template <int NT>
int launch_nt(const WgradMfmaArgs& a, size_t partial_bytes, int* nslices, hipStream_t s) {
  // blocks in flight: 3x3 tiles are MFMA-heavy (4 blocks/CU is plenty); 1x1 tiles are load-bound (more, smaller blocks)
  const int target = NT == 1 ? 2048 : 1024;
  if (a.Cv <= 16 && a.Cf <= 16) return launch_cfg<NT, 1, 1, 1, 1, 4>(a, target, partial_bytes, nslices, s);  // 16 x 16
  if (a.Cv <= 16) return launch_cfg<NT, 2, 1, 1, 1, 4>(a, target, partial_bytes, nslices, s);  // 32 x 16, pixel-split waves
  if (a.Cf <= 16) return launch_cfg<NT, 1, 2, 1, 1, 4>(a, target, partial_bytes, nslices, s);  // 16 x 32
  if (a.Cf % 64 == 0 && a.Cv % 64 == 0) return launch_cfg<NT, 2, 2, 2, 2, 1>(a, target, partial_bytes, nslices, s);  // 64 x 64
  return launch_cfg<NT, 1, 1, 2, 2, 1>(a, target, partial_bytes, nslices, s);                  // 32 x 32
}

}  // namespace

// fold the partial slices of a weight-gradient launch into dW (+ the bias gradient): shared by the fp32 and the split-bf16 form
int drs_wgrad_reduce(const float* partial, long long slice_stride, int nslices, int Cf, int Cv, int ntaps, const int* wtap, int of,
                     int ov, int T_total, float* dW, float* dbias, hipStream_t s) {
  ReduceArgs r = {};
  r.partial = partial; r.slice_stride = slice_stride; r.nslices = nslices; r.Cf = Cf; r.Cv = Cv; r.NT = ntaps;
  for (int t = 0; t < ntaps; ++t) r.wtap[t] = wtap[t];
  r.of = of; r.ov = ov; r.T_total = T_total; r.dW = dW;
  r.dbias = dbias;
  const long long numel = slice_stride;
  const unsigned gx = (unsigned)((numel + 255) / 256);
  // enough blocks to pull the partials at HBM speed: aim at >= 1024 blocks in all
  int gy = (int)std::min<long long>(nslices, std::max<long long>(1, 1024 / gx));
  const int spy = (nslices + gy - 1) / gy;
  gy = (nslices + spy - 1) / spy;
  DRS_LAUNCH(wgrad_reduce_kernel, dim3(gx, gy), dim3(256), 0, s, r, spy);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

bool drs_wgrad_mfma_supported(const WgradDesc& d) {
  // channel counts: multiples of 16, or a few-channel image side (<= 4 channels, scalar staging into a 16-wide tile)
  if ((d.Ca % 16 && d.Ca > 4) || (d.Cb % 16 && d.Cb > 4)) return false;
  if (d.ntaps != 1 && d.ntaps != 4 && d.ntaps != 9) return false;
  bool a_fixed = d.sa == 1, b_fixed = d.sb == 1;
  for (int t = 0; t < d.ntaps; ++t) {
    if (d.ay[t] || d.ax[t]) a_fixed = false;
    if (d.by[t] || d.bx[t]) b_fixed = false;
  }
  if (b_fixed) return d.BH == d.TH && d.BW == d.TW;                                     // convolution
  if (a_fixed) return d.AH == d.TH && d.AW == d.TW && !d.a_add && !d.a_gate;           // transposed convolution
  return false;
}

int drs_launch_wgrad_mfma(const WgradDesc& d, float* partial, size_t partial_bytes, hipStream_t s) {
  DRS_REQUIRE(drs_wgrad_mfma_supported(d), DRS_ERR_SHAPE, "wgrad_mfma: unsupported descriptor");
  const long long P = (long long)d.N * d.TH * d.TW;
  if (P == 0) return DRS_OK;
  bool b_fixed = d.sb == 1;
  for (int t = 0; t < d.ntaps; ++t)
    if (d.by[t] || d.bx[t]) b_fixed = false;
  WgradMfmaArgs a = {};
  int of, ov;
  if (b_fixed) {  // F = dY (b), V = layer input (a)
    a.F = d.B; a.f_cs = d.b_cs; a.f_co = d.b_co; a.Cf = d.Cb;
    a.V = d.A; a.v_cs = d.a_cs; a.v_co = d.a_co; a.Cv = d.Ca; a.VH = d.AH; a.VW = d.AW; a.sv = d.sa;
    for (int t = 0; t < d.ntaps; ++t) { a.oy[t] = d.ay[t]; a.ox[t] = d.ax[t]; }
    a.v_add = d.a_add; a.v_add_cs = d.a_add_cs; a.v_gate = d.a_gate;
    of = d.out_transposed ? 1 : d.Ca;  ov = d.out_transposed ? d.Cb : 1;   // index (b*Ca + a) or (a*Cb + b), f = b, v = a
  } else {        // F = layer input (a), V = dY (b)
    a.F = d.A; a.f_cs = d.a_cs; a.f_co = d.a_co; a.Cf = d.Ca;
    a.V = d.B; a.v_cs = d.b_cs; a.v_co = d.b_co; a.Cv = d.Cb; a.VH = d.BH; a.VW = d.BW; a.sv = d.sb;
    for (int t = 0; t < d.ntaps; ++t) { a.oy[t] = d.by[t]; a.ox[t] = d.bx[t]; }
    of = d.out_transposed ? d.Cb : 1;  ov = d.out_transposed ? 1 : d.Ca;   // f = a, v = b
  }
  a.N = d.N; a.TH = d.TH; a.TW = d.TW;
  a.f_scalar = (a.Cf % 4 || a.f_cs % 4 || a.f_co % 4) ? 1 : 0;
  a.v_scalar = (a.Cv % 4 || a.v_cs % 4 || a.v_co % 4) ? 1 : 0;
  a.bias = (b_fixed && d.dbias) ? 1 : 0;
  int ymin = a.oy[0], ymax = a.oy[0], xmin = a.ox[0], xmax = a.ox[0];
  for (int t = 1; t < d.ntaps; ++t) {
    ymin = std::min(ymin, a.oy[t]); ymax = std::max(ymax, a.oy[t]);
    xmin = std::min(xmin, a.ox[t]); xmax = std::max(xmax, a.ox[t]);
  }
  a.ymin = ymin; a.xmin = xmin;
  a.WR = (RT - 1) * a.sv + (ymax - ymin) + 1;
  a.WC = (CT - 1) * a.sv + (xmax - xmin) + 1;
  a.wc_magic = (unsigned)((0x100000000ull + a.WC - 1) / a.WC);
  a.tiles_x = (d.TW + CT - 1) / CT;
  a.tiles_y = (d.TH + RT - 1) / RT;
  a.ntiles = d.N * a.tiles_x * a.tiles_y;
  a.partial = partial;
  a.slice_stride = (long long)d.ntaps * a.Cf * a.Cv + (a.bias ? a.Cf : 0);
  int nslices = 0, rc;
  switch (d.ntaps) {
    case 1: rc = launch_nt<1>(a, partial_bytes, &nslices, s); break;
    case 4: rc = launch_nt<4>(a, partial_bytes, &nslices, s); break;
    default: rc = launch_nt<9>(a, partial_bytes, &nslices, s); break;
  }
  if (rc == -1) { DrsErr::set("wgrad_mfma: workspace of %zu bytes / LDS too small for this layer", partial_bytes); return DRS_ERR_WORKSPACE; }
  if (rc) return rc;
  return drs_wgrad_reduce(partial, a.slice_stride, nslices, a.Cf, a.Cv, d.ntaps, d.wtap, of, ov, d.T_total, d.dW,
                          a.bias ? d.dbias : nullptr, s);
}
