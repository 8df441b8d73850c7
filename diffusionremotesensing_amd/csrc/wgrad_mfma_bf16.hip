// Weight gradients on the bf16 matrix pipe, operands split hi + lo (3 MFMAs per product, fp32 accumulate): the split-bf16
// sibling of wgrad_mfma_kernel (wgrad_mfma.hip, exact fp32 at 1/16 of the bf16 MFMA rate).
//
//   dW[f][v][tap] = sum over (n, ty, tx) of F[n, ty, tx, f] * V[n, ty*sv + oy[tap], tx*sv + ox[tap], v]
//
// The reduction runs over PIXELS, which are the slow index of the channels-last tensors: an MFMA operand register group
// needs 8 consecutive pixels of ONE channel.  Both sides are therefore staged to LDS as [pixel][16 channels] rows of bf16
// (converted on the way: hi = bf16(x), lo = bf16(x - hi)) and read back column-wise with ds_read_b64_tr_b16, the
// hardware's transposing LDS read: a 16-lane group fetches 4 pixel rows x 16 channels and every lane receives its channel's
// 4 pixels (two reads = the 8 k-elements of a lane).  Rows are 32 bytes, every 8 rows are followed by 128 bytes of padding:
// the two blocks a 32-lane half reads (8 pixels apart) then sit in opposite halves of the bank row (conflict-free for
// stride-1 taps).  K-step = 32 positions (two rows of the 4 x 16-position tile), 3 MFMAs of 16x16x32 per (tap, block).
// Same work split, partial slices and reduction kernel as the fp32 form; reference: loss.backward(),
// train_diffusion_superres.py:392.  The products are summed over up to 1 M pixels in fp32 accumulators: 16-bit operand
// mantissas keep the gradient norms inside the 2e-4 bar of the golden gradient test (the forward stays exact fp32).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <algorithm>
#include <string>

#include "drs_common.h"
#include "mfma_policy.h"

namespace {

constexpr int RT = 4, CT = 16;  // iteration-domain tile (positions)
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradBf16Args {
  const float* F; int f_cs, f_co, Cf;
  const float* V; int v_cs, v_co, Cv, VH, VW, sv;
  int N, TH, TW;
  int oy[DRS_MAX_TAPS], ox[DRS_MAX_TAPS];
  int ymin, xmin, WR, WC;
  unsigned wc_magic;
  const float* v_add; int v_add_cs;
  const float* v_gate;
  int tiles_x, tiles_y, ntiles;
  int ctv;
  float* partial; long long slice_stride;
  int bias;
  int vt_bytes;  // bytes of one 16-channel tile of the V window image
};

// byte offset of pixel row r inside a 16-channel tile image: 32-byte rows, 128 bytes of padding after every 8 rows
__host__ __device__ constexpr int row_off(int r) { return (r >> 3) * 384 + (r & 7) * 32; }
constexpr int F_TILE_BYTES = (RT * CT / 8) * 384 + 32;  // 64 positions; + 32: the tiles' rows start 8 banks apart (staging writes)

__device__ __forceinline__ bf16x8 tr_pair(const char* a0, const char* a1) {
  const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 z = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
  return __builtin_bit_cast(bf16x8, z);
}

template <int NT, int MW, int NW, int NPV, int MINB, bool VADD = false>  // block = 2 x 2 waves, wave (wm, wn) owns MW x NW blocks of 16 x 16; NPV: window prefetch registers (x 16 bytes); MINB: blocks per CU the register budget is set for
__global__ __launch_bounds__(256, MINB) void wgrad_bf16_kernel(WgradBf16Args g) {
  using P = PolicyBF16X3;
  constexpr int WM = 2, WN = 2;
  constexpr int CHF = WM * MW * 16, CHV = WN * NW * 16;
  constexpr int TF = CHF / 16, TV = CHV / 16, QF = CHF / 4, QV = CHV / 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* sFh = lds;                         // [hi | lo][tile TF][row_off(position)] rows of 16 x bf16
  char* sFl = sFh + TF * F_TILE_BYTES;
  char* sVh = sFl + TF * F_TILE_BYTES;
  char* sVl = sVh + TV * g.vt_bytes;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int f0 = ((int)blockIdx.y / g.ctv) * CHF, v0 = ((int)blockIdx.y % g.ctv) * CHV;
  const int l16 = lane & 15, kq = lane >> 4, rq = l16 >> 2, cp = l16 & 3;  // tr-read role: row rq, columns 4 cp .. + 3

  f32x4 acc[NT][MW][NW];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = g.bias && v0 == 0 && wn == 0;
  f32x4 accb[MW];
#pragma unroll
  for (int i = 0; i < MW; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

  // (drs_split2: the staging's vector work is of the order of the tile's matrix work, and nothing overlaps the two in this kernel)
  auto split4 = [&](const f32x4& v, s16x4& h, s16x4& l) __attribute__((always_inline)) {
    uint2 hu, lu;
    { const uint2 s_ = drs_split2(v[0], v[1]); hu.x = s_.x; lu.x = s_.y; }
    { const uint2 s_ = drs_split2(v[2], v[3]); hu.y = s_.x; lu.y = s_.y; }
    h = __builtin_bit_cast(s16x4, hu);
    l = __builtin_bit_cast(s16x4, lu);
  };

  // Staging, software-pipelined: the global loads of the NEXT tile are issued before this tile's MFMAs and converted /
  // written to LDS after them (registers carry them across; a tile's loads are a memory round trip that nothing else on
  // the CU would cover at 2 blocks per CU).  Layers with an input add / gate or a window beyond the prefetch registers
  // (stride-2 taps) stage in place instead.
  constexpr int NPF = (RT * CT * QF + 255) / 256;
  const int wtotal = g.WR * g.WC * QV;
  const bool pipe = !g.v_gate && (VADD || !g.v_add) && wtotal <= NPV * 256;
  f32x4 pf[NPF], pv[NPV];
  // input add (UpConvBlock: x + relu(time_mlp(t)) before its convolution) under the pipeline: a thread's pieces all carry the
  // same 4 channels (256 % QV == 0), so ONE vector of the image's row rides along with the prefetch and is added at store
  // time to the pieces that were inside the image (bit u of pv_in).  Adding it at load time - what the in-place staging does -
  // would wait for every load right after issuing it: those layers ran unpipelined before (167 us against ~100).
  static_assert(256 % QV == 0, "a thread's window pieces share their channel quad");
  f32x4 padd = {0.f, 0.f, 0.f, 0.f};
  unsigned pv_in = 0;
  auto tile_coords = [&](int tile, int& n, int& ty0, int& tx0) __attribute__((always_inline)) {
    tx0 = (tile % g.tiles_x) * CT;
    ty0 = ((tile / g.tiles_x) % g.tiles_y) * RT;
    n = tile / (g.tiles_x * g.tiles_y);
  };
  auto load_f = [&](int i, int n, int ty0, int tx0) __attribute__((always_inline)) {
    const int q = i % QF, p = i / QF;
    const int y = ty0 + p / CT, x = tx0 + p % CT, c = f0 + q * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < RT * CT * QF && y < g.TH && x < g.TW && c < g.Cf)
      v = *reinterpret_cast<const f32x4*>(g.F + (((long long)n * g.TH + y) * g.TW + x) * g.f_cs + g.f_co + c);
    return v;
  };
  auto load_v = [&](int i, int n, int ty0, int tx0) __attribute__((always_inline)) {
    const int q = i % QV, p = i / QV;
    const int wy = (int)__umulhi((unsigned)p, g.wc_magic), wx = p - wy * g.WC;
    const int y = ty0 * g.sv + g.ymin + wy, x = tx0 * g.sv + g.xmin + wx, c = v0 + q * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < wtotal && y >= 0 && y < g.VH && x >= 0 && x < g.VW && c < g.Cv) {
      v = *reinterpret_cast<const f32x4*>(g.V + (((long long)n * g.VH + y) * g.VW + x) * g.v_cs + g.v_co + c);
      if (g.v_add) v += *reinterpret_cast<const f32x4*>(g.v_add + (long long)n * g.v_add_cs + c);
      if (g.v_gate) v *= g.v_gate[((long long)n * (g.VH >> 1) + (y >> 1)) * (g.VW >> 1) + (x >> 1)];
    }
    return v;
  };
  auto store_f = [&](int i, const f32x4& v) __attribute__((always_inline)) {
    if (i >= RT * CT * QF) return;
    const int q = i % QF, p = i / QF;
    s16x4 h, l;
    split4(v, h, l);
    const int off = (q >> 2) * F_TILE_BYTES + row_off(p) + (q & 3) * 8;
    *reinterpret_cast<s16x4*>(sFh + off) = h;
    *reinterpret_cast<s16x4*>(sFl + off) = l;
  };
  auto store_v = [&](int i, const f32x4& v) __attribute__((always_inline)) {
    if (i >= wtotal) return;
    const int q = i % QV, p = i / QV;
    s16x4 h, l;
    split4(v, h, l);
    const int off = (q >> 2) * g.vt_bytes + row_off(p) + (q & 3) * 8;
    *reinterpret_cast<s16x4*>(sVh + off) = h;
    *reinterpret_cast<s16x4*>(sVl + off) = l;
  };
  auto prefetch = [&](int tile) __attribute__((always_inline)) {
    int n, ty0, tx0;
    tile_coords(tile, n, ty0, tx0);
#pragma unroll
    for (int u = 0; u < NPF; ++u) pf[u] = load_f(tid + u * 256, n, ty0, tx0);
    if constexpr (VADD) {
      const int c = v0 + (tid % QV) * 4;
      padd = c < g.Cv ? *reinterpret_cast<const f32x4*>(g.v_add + (long long)n * g.v_add_cs + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      pv_in = 0;
    }
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
      if constexpr (VADD) {  // raw load + the in-image bit (the add follows at store time)
        const int i = tid + u * 256, q = i % QV, p = i / QV;
        const int wy = (int)__umulhi((unsigned)p, g.wc_magic), wx = p - wy * g.WC;
        const int y = ty0 * g.sv + g.ymin + wy, x = tx0 * g.sv + g.xmin + wx, c = v0 + q * 4;
        const bool ok = i < wtotal && y >= 0 && y < g.VH && x >= 0 && x < g.VW && c < g.Cv;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(g.V + (((long long)n * g.VH + y) * g.VW + x) * g.v_cs + g.v_co + c);
        pv[u] = v;
        pv_in |= (ok ? 1u : 0u) << u;
      } else {
        pv[u] = load_v(tid + u * 256, n, ty0, tx0);
      }
    }
  };
  if (pipe && (int)blockIdx.x < g.ntiles) prefetch(blockIdx.x);

  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    int n, ty0, tx0;
    tile_coords(tile, n, ty0, tx0);
    __syncthreads();  // every wave is done reading the previous tile
    if (pipe) {
#pragma unroll
      for (int u = 0; u < NPF; ++u) store_f(tid + u * 256, pf[u]);
#pragma unroll
      for (int u = 0; u < NPV; ++u) store_v(tid + u * 256, (VADD && ((pv_in >> u) & 1u)) ? pv[u] + padd : pv[u]);
    } else {
      // ---- stage F: RT x CT positions, CHF channels (zero outside the domain), converted to bf16 hi | lo ----
      for (int i = tid; i < RT * CT * QF; i += 256) store_f(i, load_f(i, n, ty0, tx0));
      // ---- stage the V window: WR x WC pixels, CHV channels (zero outside the image: the convolution's padding) ----
      for (int i = tid; i < wtotal; i += 256) store_v(i, load_v(i, n, ty0, tx0));
    }
    __syncthreads();
    if (pipe && tile + (int)gridDim.x < g.ntiles) prefetch(tile + gridDim.x);
    // ---- two K-steps of 32 positions: k = 8 kq + e  <->  position (row 2 s + (kq >> 1), x = (kq & 1) * 8 + e) ----
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int r = 2 * s + (kq >> 1), xk = (kq & 1) * 8;
      // F fragments: positions (r, xk .. xk + 7) = rows r*16 + xk + e of the F image: one 8-row block, rows rq and rq + 4
      typename P::Frag a[MW];
      {
        const int base = row_off(r * CT + xk) + rq * 32 + cp * 8;
#pragma unroll
        for (int i = 0; i < MW; ++i) {
          const int o = (wm * MW + i) * F_TILE_BYTES + base;
          a[i] = typename P::Frag{tr_pair(sFh + o, sFh + o + 128), tr_pair(sFl + o, sFl + o + 128)};
        }
      }
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < MW; ++i) {
          accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].hi, ones, accb[i], 0, 0, 0);
          accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].lo, ones, accb[i], 0, 0, 0);
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        // window pixel of position (r, x) through tap t: ((r*sv + oy - ymin) * WC + x*sv + ox - xmin)
        const int w0 = (r * g.sv + g.oy[t] - g.ymin) * g.WC + (xk + rq) * g.sv + (g.ox[t] - g.xmin);
        const int w1 = w0 + 4 * g.sv;
        const int o0 = row_off(w0) + cp * 8, o1 = row_off(w1) + cp * 8;
        typename P::Frag b[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          const int tb = (wn * NW + j) * g.vt_bytes;
          b[j] = typename P::Frag{tr_pair(sVh + tb + o0, sVh + tb + o1), tr_pair(sVl + tb + o0, sVl + tb + o1)};
        }
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
          for (int j = 0; j < NW; ++j) acc[t][i][j] = P::mma(a[i], b[j], acc[t][i][j]);
      }
    }
  }
  // ---- partial slice of this block: P[tap][f][v]; lane holds D[f = 4*kq + e][v = l16] of each 16x16 block ----
  float* Pp = g.partial + (long long)blockIdx.x * g.slice_stride;
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
          if (f < g.Cf && v < g.Cv) Pp[((long long)t * g.Cf + f) * g.Cv + v] = acc[t][i][j][e];
        }
      }
  if (do_bias && l16 == 0) {
    float* Pb = Pp + (long long)NT * g.Cf * g.Cv;
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = f0 + (wm * MW + i) * 16 + 4 * kq + e;
        if (f < g.Cf) Pb[f] = accb[i][e];
      }
  }
}

template <int NT, int MW, int NW, int NPV, int MINB, bool VADD = false>
int launch_cfg(const WgradBf16Args& a0, int target_blocks, size_t partial_bytes, int* nslices, hipStream_t s) {
  WgradBf16Args a = a0;
  constexpr int CHF = 2 * MW * 16, CHV = 2 * NW * 16;
  const int ctf = (a.Cf + CHF - 1) / CHF;
  a.ctv = (a.Cv + CHV - 1) / CHV;
  const int ct = ctf * a.ctv;
  long long ksplit = target_blocks / ct;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > a.ntiles) ksplit = a.ntiles;
  const long long max_slices = (long long)(partial_bytes / ((size_t)a.slice_stride * 4));
  if (max_slices < 1) return -1;
  if (ksplit > max_slices) ksplit = max_slices;
  a.vt_bytes = ((a.WR * a.WC + 7) / 8) * 384 + 32;
  const size_t lds = 2 * ((size_t)(CHF / 16) * F_TILE_BYTES + (size_t)(CHV / 16) * a.vt_bytes);
  if (lds > 160 * 1024) return -1;
  auto kern = wgrad_bf16_kernel<NT, MW, NW, NPV, MINB, VADD>;
  if (lds > 64 * 1024) DRS_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  DRS_LAUNCH(kern, dim3((unsigned)ksplit, ct), dim3(256), lds, s, a);
  DRS_CHECK_HIP(hipGetLastError());
  *nslices = (int)ksplit;
  return DRS_OK;
}

// Tile configurations.  Register budget first: the 64 x 64 tile of a 9-tap layer holds 144 accumulators and, with its
// prefetch registers, needs ~480 registers - ONE 4-wave block per CU (512 registers per lane and SIMD are shared by its waves).
// The 64 x 32 tile (72 accumulators) fits 256: two blocks per CU, twice the waves to cover the barriers and the staging.
template <int NT>
int launch_nt(const WgradBf16Args& a, size_t partial_bytes, int* nslices, hipStream_t s) {
  // blocks per launch = K-slices x channel tiles: one resident wave of blocks (two per CU); twice as many measured 0.5 % slower
  // per training step (twice the partial slices to write and to reduce)
  const int target = NT == 1 ? 1024 : 512;
  // stride-2 tap sets (ConvTranspose / stride-2 convolution: the V window of a 4 x 16 tile is 9 x 33 pixels): a 64-channel V tile
  // needs 117 KB of LDS - one block per CU, window staged in place with nothing to hide the round trip (255 - 300 us for the
  // 19 GFLOP of a `transform` layer); a 32-channel V tile has the window prefetched (ten registers: 2376 pieces): 167 us.
  // (324 registers, one block per CU; squeezed into 256 for two blocks - 9 spills - it measured 240 - 280 us)
  if (a.sv == 2 && a.Cf % 64 == 0 && a.Cv % 32 == 0) return launch_cfg<NT, 2, 1, 10, 1>(a, target, partial_bytes, nslices, s);  // 64 x 32
  // a per-image vector added to the window (UpConvBlock: x + relu(time_mlp(t)) before ups.i.conv): its own instantiation, the
  // add under the pipeline (167 -> 85 us per layer; in-place staging before)
  if constexpr (NT == 9)
    if (a.v_add && !a.v_gate && a.sv == 1 && a.Cf % 64 == 0 && a.Cv % 32 == 0)
      return launch_cfg<NT, 2, 1, 4, 2, true>(a, target, partial_bytes, nslices, s);
  // 64 x 32, two blocks per CU (the 64 x 64 tile it replaced - one block per CU - per layer: 264 -> 198 us, 105 -> 78, 60 -> 46)
  if (a.sv == 1 && a.Cf % 64 == 0 && a.Cv % 32 == 0) return launch_cfg<NT, 2, 1, 4, 2>(a, target, partial_bytes, nslices, s);
  return launch_cfg<NT, 1, 1, 10, 1>(a, target, partial_bytes, nslices, s);                                 // 32 x 32
}

}  // namespace

int drs_wgrad_reduce(const float* partial, long long slice_stride, int nslices, int Cf, int Cv, int ntaps, const int* wtap, int of,
                     int ov, int T_total, float* dW, float* dbias, hipStream_t s);  // wgrad_mfma.hip

// DRS_TRAIN_WGRAD_IMPL: mfma_bf16x3 (default) | mfma_f32
static bool wgrad_bf16_enabled() {
  static const int on = [] {
    const char* e = getenv("DRS_TRAIN_WGRAD_IMPL");
    return (e && std::string(e) == "mfma_f32") ? 0 : 1;
  }();
  return on != 0;
}

bool drs_wgrad_mfma_bf16_supported(const WgradDesc& d) {
  if (!wgrad_bf16_enabled() || !drs_wgrad_mfma_supported(d)) return false;
  // 32-channel blocks on both sides, float4-able slices (the few-channel image layers have their own kernel: stem_wgrad_kernel)
  // (a 16-channel side - the first residual block's input - fills half of a 32-channel block with zeros: 57 us instead of the
  //  fp32 form's 103 + an 89 us reduction of its slices)
  if ((d.Ca % 32 && d.Ca != 16) || (d.Cb % 32 && d.Cb != 16) || (d.a_cs & 3) || (d.a_co & 3) || (d.b_cs & 3) || (d.b_co & 3)) return false;
  return true;
}

int drs_launch_wgrad_mfma_bf16(const WgradDesc& d, float* partial, size_t partial_bytes, hipStream_t s) {
  DRS_REQUIRE(drs_wgrad_mfma_bf16_supported(d), DRS_ERR_SHAPE, "wgrad_mfma_bf16: unsupported descriptor");
  const long long Pn = (long long)d.N * d.TH * d.TW;
  if (Pn == 0) return DRS_OK;
  bool b_fixed = d.sb == 1;
  for (int t = 0; t < d.ntaps; ++t)
    if (d.by[t] || d.bx[t]) b_fixed = false;
  WgradBf16Args a = {};
  int of, ov;
  if (b_fixed) {  // F = dY (b), V = layer input (a)
    a.F = d.B; a.f_cs = d.b_cs; a.f_co = d.b_co; a.Cf = d.Cb;
    a.V = d.A; a.v_cs = d.a_cs; a.v_co = d.a_co; a.Cv = d.Ca; a.VH = d.AH; a.VW = d.AW; a.sv = d.sa;
    for (int t = 0; t < d.ntaps; ++t) { a.oy[t] = d.ay[t]; a.ox[t] = d.ax[t]; }
    a.v_add = d.a_add; a.v_add_cs = d.a_add_cs; a.v_gate = d.a_gate;
    of = d.out_transposed ? 1 : d.Ca;  ov = d.out_transposed ? d.Cb : 1;
  } else {        // F = layer input (a), V = dY (b)
    a.F = d.A; a.f_cs = d.a_cs; a.f_co = d.a_co; a.Cf = d.Ca;
    a.V = d.B; a.v_cs = d.b_cs; a.v_co = d.b_co; a.Cv = d.Cb; a.VH = d.BH; a.VW = d.BW; a.sv = d.sb;
    for (int t = 0; t < d.ntaps; ++t) { a.oy[t] = d.by[t]; a.ox[t] = d.bx[t]; }
    of = d.out_transposed ? d.Cb : 1;  ov = d.out_transposed ? 1 : d.Ca;
  }
  a.N = d.N; a.TH = d.TH; a.TW = d.TW;
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
  if (rc == -1) { DrsErr::set("wgrad_mfma_bf16: workspace of %zu bytes / LDS too small for this layer", partial_bytes); return DRS_ERR_WORKSPACE; }
  if (rc) return rc;
  return drs_wgrad_reduce(partial, a.slice_stride, nslices, a.Cf, a.Cv, d.ntaps, d.wtap, of, ov, d.T_total, d.dW,
                          a.bias ? d.dbias : nullptr, s);
}
