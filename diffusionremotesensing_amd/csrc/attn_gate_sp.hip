// Fused additive attention gate of one decoder stage (reference AttentionBlock.forward + gating_signal.forward,
// UNet_model_superres.py:89-108, 222-225) over SP-format activations, ONE launch instead of five:
//   g   = relu(BN(conv1x1(x)))                          gating signal, Cc -> Ch channels at the low resolution
//   g1  = w_g(g)                                        1x1, Ch -> Ch
//   p   = relu(g1 + w_x(x_res))                         2x2 stride 2 over the skip tensor (Ch channels, double resolution)
//   psi = sigmoid(conv1x1(p) -> 1 channel)
//   att = BN(conv1x1(nearest2x(psi) * x_res)) = nearest2x(psi) * (W' x_res) + b'     -> channel slice of the concat buffer
// The five ops are HBM-bound with next to no arithmetic; run one by one they move x, g, g1, p, psi and x_res (twice)
// through HBM.  Here a wave owns 16 low-resolution pixels and chains the GEMMs IN REGISTERS: with the output-channel
// permutation of the SP format (drs_sp_cout_perm: a lane's accumulators of a tile pair are 8 consecutive channels of its
// pixel) the bias + ReLU'd accumulator of one GEMM, split into bf16 hi | lo, IS the B operand of the next one.  The four
// x_res pixels under a low-resolution pixel are loaded once and serve both w_x and the gated result convolution.
// HBM traffic per stage: x + x_res in, att out - half of the unfused sequence; no intermediate ever leaves the CU.
// Weights of all four matrices stay in LDS for the lifetime of the (persistent) block; activations go global ->
// registers in MFMA operand layout (an SP slot = 8 channels of a pixel = one operand register group), no LDS staging,
// no barrier after the weight load.
#include "conv_epilogue.h"
#include "mfma_policy.h"

namespace {

// PSI_ONLY: stop at psi (written to psi_out): the consumer of `att` multiplies by psi itself (the top decoder stage, whose
// att-half convolution has the `result` convolution and the output projection folded into its weights: conv3x3_proj_sp_kernel)
template <int NT, bool PSI_ONLY = false>  // Ch = 16 * NT output channels everywhere: NT = 2 (Ch = 32), 4 (Ch = 64)
__global__ __launch_bounds__(NT == 2 ? 1024 : 512, 1) void attn_gate_sp_kernel(AttnGateDesc d) {
  // (a 32-channel stage needs 92 registers: 16 waves per CU instead of 8 - the kernel is a chain of memory round trips per
  // item and wave, with nothing but the other waves of the CU to fill them.  The 64-channel stage fits 168 registers, i.e.
  // 12 waves per CU: measured in round 4, 48 -> 55 us - more waves re-reading the LDS weight images is not what it lacks)
  constexpr int THREADS = NT == 2 ? 1024 : 512, NW = THREADS / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  constexpr int Ch = 16 * NT, NC = NT / 2;  // NC: 32-channel chunks of a Ch-channel tensor
  const int ncx = d.Cc / 32;                // chunks of the stage input
  // LDS images: gate [ncx][4][Ch], wg [NC][4][Ch], wx [NC][4 taps][4][Ch], res [NC][4][Ch] slots of 16 bytes, hi then lo
  const int gate_img = ncx * 4 * Ch * 16, wg_img = NC * 4 * Ch * 16, wx_img = NC * 16 * Ch * 16;
  char* sGate = smem;
  char* sWg = sGate + 2 * gate_img;
  char* sWx = sWg + 2 * wg_img;
  char* sRes = sWx + 2 * wx_img;
  float* sB = reinterpret_cast<float*>(sRes + 2 * wg_img);  // b_gate | b_wg + b_wx | w_psi | b_res | b_psi
  {
    const int tid = threadIdx.x;
    auto copy = [&](char* dst, const void* src, int bytes) {  // 8 loads in flight per thread (not a round trip per 8 KB)
      for (int o0 = tid * 16; o0 < bytes; o0 += 8 * THREADS * 16) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int o = o0 + u * THREADS * 16;
          if (o < bytes) v[u] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src) + o);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int o = o0 + u * THREADS * 16;
          if (o < bytes) *reinterpret_cast<u32x4*>(dst + o) = v[u];
        }
      }
    };
    copy(sGate, d.w_gate, 2 * gate_img);
    copy(sWg, d.w_wg, 2 * wg_img);
    copy(sWx, d.w_wx, 2 * wx_img);
    if constexpr (!PSI_ONLY) copy(sRes, d.w_res, 2 * wg_img);
    // bias vectors: a global load inside an item is a memory round trip on the wave's critical path (five of them per item)
    for (int i = tid; i < Ch; i += THREADS) {
      sB[i] = d.b_gate[i];
      sB[Ch + i] = d.b_wg[i] + d.b_wx[i];
      sB[2 * Ch + i] = d.w_psi[i];
      sB[3 * Ch + i] = PSI_ONLY ? 0.f : d.b_res[i];
    }
    if (tid == 0) sB[4 * Ch] = d.b_psi[0];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  auto wfrag = [&](const char* base, int img, int slot) {  // slot = ((chunk * taps + tap) * 4 + kg) * Ch + tile * 16 + lr
    return P::load(base, (size_t)img, (size_t)slot * 16);
  };
  // channel of (tile tt, register j) of this lane in SP order: (tt >> 1) * 32 + kg * 8 + (tt & 1) * 4 + j
  const int bw = (d.LW + 15) / 16;
  const long long nitems = (long long)d.N * d.LH * bw;
  const int OW = 2 * d.LW, OH = 2 * d.LH;
  for (long long it = (long long)blockIdx.x * NW + wave; it < nitems; it += (long long)gridDim.x * NW) {
    const int xb = (int)(it % bw), y = (int)((it / bw) % d.LH), n = (int)(it / ((long long)bw * d.LH));
    const int px_raw = xb * 16 + lr;
    const bool valid = px_raw < d.LW;
    const int px = valid ? px_raw : d.LW - 1;
    // ---- skip-tensor fragments of the 4 pixels under this low-resolution pixel (issued first: the longest latency) ----
    typename P::Frag xr[4][NC];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const size_t pix = ((size_t)n * OH + 2 * y + (t >> 1)) * OW + 2 * px + (t & 1);
      const char* g = reinterpret_cast<const char*>(d.xres) + (pix * d.r_cs + d.r_co) * 4 + kg * 16;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc)
        xr[t][cc] = typename P::Frag{*reinterpret_cast<const bf16x8*>(g + cc * 128), *reinterpret_cast<const bf16x8*>(g + cc * 128 + 64)};
    }
    // ---- gating signal: g = relu(Wg x + bg), K = Cc from global memory ----
    f32x4 acc[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      const char* g = reinterpret_cast<const char*>(d.x) + ((((size_t)n * d.LH + y) * d.LW + px) * d.x_cs + d.x_co) * 4 + kg * 16;
      for (int c = 0; c < ncx; c += 2) {  // two chunks in flight (Cc is a multiple of 64 on these stages)
        const typename P::Frag a0{*reinterpret_cast<const bf16x8*>(g + c * 128), *reinterpret_cast<const bf16x8*>(g + c * 128 + 64)};
        const typename P::Frag a1{*reinterpret_cast<const bf16x8*>(g + c * 128 + 128), *reinterpret_cast<const bf16x8*>(g + c * 128 + 192)};
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) acc[tt] = P::mma(wfrag(sGate, gate_img, (c * 4 + kg) * Ch + tt * 16 + lr), a0, acc[tt]);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) acc[tt] = P::mma(wfrag(sGate, gate_img, ((c + 1) * 4 + kg) * Ch + tt * 16 + lr), a1, acc[tt]);
      }
    }
    // bias + ReLU, split: the accumulators of tile pair cc are this lane's operand slot of chunk cc of the next GEMM
    typename P::Frag gfr[NC];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      const float* bsrc = d.b_gate_img ? d.b_gate_img + (size_t)n * Ch + cc * 32 + kg * 8 : sB + cc * 32 + kg * 8;
      const float4 b0 = *reinterpret_cast<const float4*>(bsrc);
      const float4 b1 = *reinterpret_cast<const float4*>(bsrc + 4);
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = drs_maxf(acc[2 * cc][j] + bb[j], 0.f);
        v[4 + j] = drs_maxf(acc[2 * cc + 1][j] + bb[4 + j], 0.f);
      }
      u32x4 h, l;
      drs_sp_split8(v, h, l);
      gfr[cc] = typename P::Frag{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, l)};
    }
    // ---- p = relu(w_g(g) + w_x(x_res) + biases) ----
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < NC; ++cc)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) acc[tt] = P::mma(wfrag(sWg, wg_img, (cc * 4 + kg) * Ch + tt * 16 + lr), gfr[cc], acc[tt]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int cc = 0; cc < NC; ++cc)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
          acc[tt] = P::mma(wfrag(sWx, wx_img, ((cc * 4 + t) * 4 + kg) * Ch + tt * 16 + lr), xr[t][cc], acc[tt]);
      __builtin_amdgcn_sched_barrier(0);  // keep the weight-fragment reads of the next tap behind these MFMAs (registers)
    }
    // ---- psi = sigmoid(w_psi . p + b_psi): in-lane partial over this lane's channels, then the 4 k-group lanes of the pixel ----
    float dot = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const int ch = (tt >> 1) * 32 + kg * 8 + (tt & 1) * 4;
      const float4 bs = *reinterpret_cast<const float4*>(sB + Ch + ch);
      const float4 wp = *reinterpret_cast<const float4*>(sB + 2 * Ch + ch);
      dot += drs_maxf(acc[tt][0] + bs.x, 0.f) * wp.x + drs_maxf(acc[tt][1] + bs.y, 0.f) * wp.y +
             drs_maxf(acc[tt][2] + bs.z, 0.f) * wp.z + drs_maxf(acc[tt][3] + bs.w, 0.f) * wp.w;
    }
    dot += __shfl_xor(dot, 16);
    dot += __shfl_xor(dot, 32);
    const float psi = 1.f / (1.f + expf(-(dot + sB[4 * Ch])));
    if (d.psi_out && valid && kg == 0) d.psi_out[((size_t)n * d.LH + y) * d.LW + px] = psi;
    // ---- att = psi * (W' x_res) + b' for the 4 pixels, stored as SP halves into the concat slice ----
    if constexpr (!PSI_ONLY)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < NC; ++cc)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) acc[tt] = P::mma(wfrag(sRes, wg_img, (cc * 4 + kg) * Ch + tt * 16 + lr), xr[t][cc], acc[tt]);
      __builtin_amdgcn_sched_barrier(0);
      const size_t pix = ((size_t)n * OH + 2 * y + (t >> 1)) * OW + 2 * px + (t & 1);
      char* o = reinterpret_cast<char*>(d.out) + (pix * d.out_cs + d.out_co) * 4 + kg * 16;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        const float4 b0 = *reinterpret_cast<const float4*>(sB + 3 * Ch + cc * 32 + kg * 8);
        const float4 b1 = *reinterpret_cast<const float4*>(sB + 3 * Ch + cc * 32 + kg * 8 + 4);
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = psi * acc[2 * cc][j] + bb[j];
          v[4 + j] = psi * acc[2 * cc + 1][j] + bb[4 + j];
        }
        u32x4 h, l;
        drs_sp_split8(v, h, l);
        if (valid) {
          drs_store16(o + cc * 128, h);
          drs_store16(o + cc * 128 + 64, l);
        }
      }
    }
  }
}

// ---- wide variant (Ch = 128: the first decoder stage, whose weights, 524 KB as bf16 hi | lo, cannot live in LDS) ----------
// (The first structure - every wave loading its own operands, weight fragments straight from global memory: 57 -> 48 us -
// was removed in round 4; the structure below superseded it at 36 us and is covered by the same goldens.)

// No operand is fetched twice.  Letting every wave load its own operands costs the 0.5 MB of weights twice per block (two pixel halves), x four
// times and x_res four times (four channel groups): 2.2 MB of L2 -> CU traffic per block for 0.7 MB of distinct bytes, all
// 256 CUs at once - it runs at the L2's pace (48 us, the pipelined form; 57 us before), not HBM's (118 MB: 22 us).
// Here a wave owns ONE 16-channel tile (its weights: 64 KB, read by nobody else) for ALL 64 pixels of the block, and the
// activations cross LDS: x (64 KB, coalesced 1 KB per pixel) for the gating phase, then the four x_res pixels under each
// low-resolution pixel as four 32 KB "taps", double-buffered in the same region, moved global -> registers -> LDS by all 512
// lanes a tap ahead.  LDS images are pixel-major 128-byte lines with the slot rotation of the wave-specialised kernels
// (slot s of pixel p at position (s + p) & 7: conflict-free for the loader, whose 8 consecutive lanes write one line, and
// for the operand reads, whose 16 lanes read one slot of 16 pixels).  A lane's accumulators are 4 consecutive channels of
// its pixel (half an SP slot): g goes to LDS and att to memory as 8-byte halves.
template <int NCX>
__global__ __launch_bounds__(512, 1) void attn_gate_wide2_kernel(AttnGateDesc d) {
  using P = PolicyBF16X3;
  using Frag = typename P::Frag;
  constexpr int Ch = 128, NG = 4;
  constexpr int XB = 4 * NCX * 16 * 128;  // x image: [pixel block 4][chunk NCX][pixel 16] lines of 128 bytes
  constexpr int TB = 4 * NG * 16 * 128;   // one x_res tap / the gating signal: [pixel block 4][chunk 4][pixel 16] lines
  constexpr int RB = XB > 2 * TB ? XB : 2 * TB;  // the two tap buffers live in the x image's region
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;                 // x during the gating phase, then the two x_res tap buffers
  char* sG = smem + RB;            // g = relu(gate(x))
  float* sPsi = reinterpret_cast<float*>(sG + TB);  // [wave 8][pixel 64] partial psi sums
  float* sB = sPsi + 8 * 64;       // b_wg + b_wx | w_psi | b_res | b_psi
  for (int i = threadIdx.x; i < Ch; i += 512) {
    sB[i] = d.b_wg[i] + d.b_wx[i];
    sB[Ch + i] = d.w_psi[i];
    sB[2 * Ch + i] = d.b_res[i];
  }
  if (threadIdx.x == 0) sB[3 * Ch] = d.b_psi[0];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int lr = lane & 15, kg = lane >> 4;
  // this lane's 4 channels (SP row permutation: the tile pair (2G, 2G + 1) of channel group G holds 8 consecutive channels per lane)
  const int G = wave >> 1, th = wave & 1, ch0 = G * 32 + kg * 8 + th * 4;
  const size_t gate_img = (size_t)NCX * 4 * Ch * 16, wg_img = (size_t)NG * 4 * Ch * 16, wx_img = (size_t)NG * 16 * Ch * 16;
  const int bw = (d.LW + 15) / 16;
  const long long nblk16 = (long long)d.N * d.LH * bw;
  const long long nitems = (nblk16 + 3) / 4;
  const int OW = 2 * d.LW, OH = 2 * d.LH;
  typedef const __attribute__((address_space(1))) char* gptr;
  typedef const __attribute__((address_space(1))) bf16x8* gfrag;
  typedef const __attribute__((address_space(1))) u32x4* gquad;
  auto uni = [](const void* q) __attribute__((always_inline)) {
    gptr g = (gptr)q;
    asm volatile("" : "+s"(g));
    return g;
  };
  const gptr gw_gate = uni(d.w_gate), gw_wg = uni(d.w_wg), gw_wx = uni(d.w_wx), gw_res = uni(d.w_res);
  const gptr gx = uni(reinterpret_cast<const char*>(d.x) + (size_t)d.x_co * 4);
  const gptr gr = uni(reinterpret_cast<const char*>(d.xres) + (size_t)d.r_co * 4);
  const unsigned lane_w = (unsigned)((kg * Ch + wave * 16 + lr) * 16);  // this lane's weight slot inside a (chunk, tap) row block
  auto opaque = [](unsigned v) __attribute__((always_inline)) { asm volatile("" : "+v"(v)); return v; };
  auto wfrag = [&](gptr base, size_t img, int urow) __attribute__((always_inline)) {  // urow (uniform) = (chunk * taps + tap) * 4: first k-group row
    const unsigned o = opaque(lane_w) + (unsigned)urow * (unsigned)(Ch * 16);
    return Frag{*(gfrag)(base + o), *(gfrag)(base + (o + (unsigned)img))};
  };
  // operand fragment of pixel block pb, chunk c of an LDS image with NC chunks per pixel block
  auto afrag = [&](const char* img, int nc, int pb, int c) __attribute__((always_inline)) {
    const char* line = img + ((pb * nc + c) * 16 + lr) * 128;
    const int s0 = ((kg + lr) & 7) * 16;
    return Frag{*reinterpret_cast<const bf16x8*>(line + s0), *reinterpret_cast<const bf16x8*>(line + (s0 ^ 64))};
  };
  auto mfma = [](const bf16x8& w, const bf16x8& x, const f32x4& c) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
  };
  // one step: the four pixel blocks' chains round-robin (a dependent MFMA issues a full latency behind its predecessor)
  auto step4 = [&](const Frag& w, const Frag (&a)[4], f32x4 (&acc)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = mfma(w.lo, a[b].hi, acc[b]);
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = mfma(w.hi, a[b].lo, acc[b]);
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = mfma(w.hi, a[b].hi, acc[b]);
  };
  __syncthreads();
  for (long long it = blockIdx.x; it < nitems; it += gridDim.x) {
    // the item's four pixel blocks: byte offsets of their first pixel in x / x_res / out, validity
    unsigned xo[4], ro[4], oo[4];
    int nn[4], npx[4];  // image, valid pixels of the block (0: block beyond the tensor)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const long long q = it * 4 + b;
      const long long qc = q < nblk16 ? q : nblk16 - 1;
      const int xb = (int)(qc % bw), y = (int)((qc / bw) % d.LH);
      nn[b] = (int)(qc / ((long long)bw * d.LH));
      npx[b] = q < nblk16 ? min(16, d.LW - xb * 16) : 0;
      xo[b] = (unsigned)((((size_t)nn[b] * d.LH + y) * d.LW + xb * 16) * d.x_cs * 4);
      ro[b] = (unsigned)((((size_t)nn[b] * OH + 2 * y) * OW + 2 * xb * 16) * d.r_cs * 4);
      oo[b] = (unsigned)((((size_t)nn[b] * OH + 2 * y) * OW + 2 * xb * 16) * d.out_cs * 4);
    }
    auto pick = [&](const unsigned (&v)[4], int b) __attribute__((always_inline)) { return b == 0 ? v[0] : b == 1 ? v[1] : b == 2 ? v[2] : v[3]; };
    auto picki = [&](const int (&v)[4], int b) __attribute__((always_inline)) { return b == 0 ? v[0] : b == 1 ? v[1] : b == 2 ? v[2] : v[3]; };
    // ---- loaders: 16-byte pieces e = 512 i + tid; slot e & 7, then chunk, then pixel: consecutive pieces are consecutive bytes
    u32x4 xr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = i * 512 + tid, pix = e / (8 * NCX), pb = pix >> 4, pl = pix & 15;
      const int plc = min(pl, max(picki(npx, pb) - 1, 0));  // (pixels beyond the row / the tensor: a valid address, the value is never stored to memory)
      xr[i] = *(gquad)(gx + (pick(xo, pb) + (unsigned)(plc * d.x_cs * 4) + (unsigned)((e % (8 * NCX)) * 16)));
    }
    u32x4 tr[4];  // the next tap's pieces
    auto tap_load = [&](int t4) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = i * 512 + tid, pix = e >> 5, pb = pix >> 4, pl = pix & 15;
        const int plc = min(pl, max(picki(npx, pb) - 1, 0));
        tr[i] = *(gquad)(gr + (pick(ro, pb) + (unsigned)((((t4 >> 1) * OW + 2 * plc + (t4 & 1)) * d.r_cs) * 4) + (unsigned)((e & 31) * 16)));
      }
    };
    auto tap_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = i * 512 + tid, pix = e >> 5, c = (e >> 3) & 3, sl = e & 7, pb = pix >> 4, pl = pix & 15;
        *reinterpret_cast<u32x4*>(sX + buf * TB + ((pb * NG + c) * 16 + pl) * 128 + ((sl + pl) & 7) * 16) = tr[i];
      }
    };
    tap_load(0);
    // result weights of this wave's tile: the same for all four taps, held
    Frag wr[NG];
#pragma unroll
    for (int cc = 0; cc < NG; ++cc) wr[cc] = wfrag(gw_res, wg_img, cc * 4);
    // weight ring over the item's NCX + 4 + 16 steps (gating chunks, w_g chunks, (tap, chunk) of w_x), WD steps ahead: a
    // step is 12 - 24 MFMAs (200 - 400 cycles), a fragment comes from L2 (one step ahead, every step waited for it)
    constexpr int WD = 3, NST = NCX + NG + 4 * NG;
    Frag wq[WD + 1];
    auto wstep = [&](int st) __attribute__((always_inline)) {
      if (st < NCX) return wfrag(gw_gate, gate_img, st * 4);
      if (st < NCX + NG) return wfrag(gw_wg, wg_img, (st - NCX) * 4);
      const int t4 = (st - NCX - NG) / NG, cc = (st - NCX - NG) % NG;
      return wfrag(gw_wx, wx_img, (cc * 4 + t4) * 4);
    };
#pragma unroll
    for (int st = 0; st < WD; ++st) wq[st] = wstep(st);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = i * 512 + tid, pix = e / (8 * NCX), c = (e / 8) % NCX, sl = e & 7, pb = pix >> 4, pl = pix & 15;
      *reinterpret_cast<u32x4*>(sX + ((pb * NCX + c) * 16 + pl) * 128 + ((sl + pl) & 7) * 16) = xr[i];
    }
    __syncthreads();
    // ---- gating signal: g = relu(W_gate x + b), this wave's 16 channels of all 64 pixels ----
    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCX; ++c) {
      if (c + WD < NST) wq[(c + WD) % (WD + 1)] = wstep(c + WD);
      Frag a[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) a[b] = afrag(sX, NCX, b, c);
      step4(wq[c % (WD + 1)], a, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float* bsrc = (d.b_gate_img ? d.b_gate_img + (size_t)nn[b] * Ch : d.b_gate) + ch0;  // (both global: no flat load)
        const float4 bb = *reinterpret_cast<const float4*>(bsrc);
        const float v[4] = {drs_maxf(acc[b][0] + bb.x, 0.f), drs_maxf(acc[b][1] + bb.y, 0.f), drs_maxf(acc[b][2] + bb.z, 0.f), drs_maxf(acc[b][3] + bb.w, 0.f)};
        unsigned h2[2], l2[2];
        drs_sp_split4(v, h2, l2);
        char* line = sG + ((b * NG + G) * 16 + lr) * 128;
        const int s0 = ((kg + lr) & 7) * 16 + th * 8;
        *reinterpret_cast<uint2*>(line + s0) = make_uint2(h2[0], h2[1]);
        *reinterpret_cast<uint2*>(line + (s0 ^ 64)) = make_uint2(l2[0], l2[1]);
      }
    }
    __syncthreads();  // g complete; every wave has left the x image
    tap_store(0);
    tap_load(1);
    __syncthreads();
    // ---- p = relu(w_g(g) + w_x(x_res) + b) and the ungated result W' x_res ----
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < NG; ++cc) {
      if (NCX + cc + WD < NST) wq[(NCX + cc + WD) % (WD + 1)] = wstep(NCX + cc + WD);
      Frag a[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) a[b] = afrag(sG, NG, b, cc);
      step4(wq[(NCX + cc) % (WD + 1)], a, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 att[4][4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
      for (int b = 0; b < 4; ++b) att[t4][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      if (t4 + 1 < 4) tap_store((t4 + 1) & 1);  // (its buffer was left by every wave before the previous barrier)
      if (t4 + 2 < 4) tap_load(t4 + 2);
#pragma unroll
      for (int cc = 0; cc < NG; ++cc) {
        const int st = NCX + NG + t4 * NG + cc;  // running step index: slot of the weight ring
        if (st + WD < NST) wq[(st + WD) % (WD + 1)] = wstep(st + WD);
        Frag a[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) a[b] = afrag(sX + (t4 & 1) * TB, NG, b, cc);
        step4(wq[st % (WD + 1)], a, acc);
        step4(wr[cc], a, att[t4]);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
    // ---- psi = sigmoid(w_psi . p + b): this wave's 16 channels, then across the 8 waves through LDS ----
    {
      const float4 bsum = *reinterpret_cast<const float4*>(sB + ch0), wp = *reinterpret_cast<const float4*>(sB + Ch + ch0);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        float dot = drs_maxf(acc[b][0] + bsum.x, 0.f) * wp.x + drs_maxf(acc[b][1] + bsum.y, 0.f) * wp.y +
                    drs_maxf(acc[b][2] + bsum.z, 0.f) * wp.z + drs_maxf(acc[b][3] + bsum.w, 0.f) * wp.w;
        dot += __shfl_xor(dot, 16);
        dot += __shfl_xor(dot, 32);
        if (kg == 0) sPsi[wave * 64 + b * 16 + lr] = dot;
      }
    }
    __syncthreads();
    {
      const float4 br = *reinterpret_cast<const float4*>(sB + 2 * Ch + ch0);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = b * 16 + lr;
        float sum = sB[3 * Ch];
#pragma unroll
        for (int w = 0; w < 8; ++w) sum += sPsi[w * 64 + i];
        const float psi = 1.f / (1.f + expf(-sum));
        const bool ok = lr < npx[b];
        if (d.psi_out && ok && kg == 0 && wave == 0) {
          const long long q = it * 4 + b;
          const int xb = (int)(q % bw), y = (int)((q / bw) % d.LH);
          d.psi_out[((size_t)nn[b] * d.LH + y) * d.LW + xb * 16 + lr] = psi;
        }
        // att = psi * (W' x_res) + b' for the 4 pixels under the low-resolution pixel: 8-byte halves of the SP slots
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const float v[4] = {psi * att[t4][b][0] + br.x, psi * att[t4][b][1] + br.y, psi * att[t4][b][2] + br.z, psi * att[t4][b][3] + br.w};
          unsigned h2[2], l2[2];
          drs_sp_split4(v, h2, l2);
          if (ok) {
            char* o = reinterpret_cast<char*>(d.out) + (size_t)d.out_co * 4 + oo[b] +
                      (size_t)(((t4 >> 1) * OW + 2 * lr + (t4 & 1)) * d.out_cs + G * 32) * 4 + kg * 16 + th * 8;
            *reinterpret_cast<uint2*>(o) = make_uint2(h2[0], h2[1]);
            *reinterpret_cast<uint2*>(o + 64) = make_uint2(l2[0], l2[1]);
          }
        }
      }
    }
    __syncthreads();  // sX / sG / sPsi are rewritten by the next item
  }
}

template <int NT, bool PSI_ONLY = false>
int attn_launch(const AttnGateDesc& d, size_t lds, hipStream_t s) {
  auto kern = attn_gate_sp_kernel<NT, PSI_ONLY>;
  int num_cu = 0;
  {
    const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(kern), 160 * 1024, &num_cu);
    if (rc) return rc;
  }
  const long long nitems = (long long)d.N * d.LH * ((d.LW + 15) / 16);
  constexpr int nw = NT == 2 ? 16 : 8;
  long long blocks = num_cu;  // one block per CU, persistent over the pixel blocks
  if (blocks * nw > nitems) blocks = (nitems + nw - 1) / nw;
  DRS_LAUNCH(kern, dim3((unsigned)blocks), dim3(nw * 64), lds, s, d);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// per-image gating bias for a stage input stored as x + vec[n] (drs_common.h: AttnGateDesc::b_gate_img).  One block per
// (image, 64 output channels): 4 slices of the input channels x 64 channels, loads unrolled (independent), partial sums
// through LDS in a fixed order.
__global__ __launch_bounds__(256) void gate_bias_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                                        const float* __restrict__ vec, int vec_stride, float* __restrict__ out,
                                                        int N, int Cc, int Ch) {
  __shared__ float part[4][64];
  const int n = blockIdx.y, co = blockIdx.x * 64 + (threadIdx.x & 63), ks = threadIdx.x >> 6;
  const float* v = vec + (size_t)n * vec_stride;
  float a = 0.f;
  if (co < Ch) {
    const int per = (Cc + 3) / 4, c0 = ks * per, c1 = min(Cc, c0 + per);
    int ci = c0;
    for (; ci + 8 <= c1; ci += 8) {
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = w[(size_t)(ci + u) * Ch + co];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += wv[u] * v[ci + u];
    }
    for (; ci < c1; ++ci) a += w[(size_t)ci * Ch + co] * v[ci];
  }
  part[ks][threadIdx.x & 63] = a;
  __syncthreads();
  if (ks == 0 && co < Ch) out[(size_t)n * Ch + co] = b[co] - (((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x]);
}

}  // namespace

size_t drs_attn_gate_lds_bytes(int Cc, int Ch) {
  // gate Ch x Cc, w_g Ch x Ch, w_x 4 x Ch x Ch, result Ch x Ch, two bf16 images each
  return (size_t)4 * Ch * ((size_t)Cc + 6 * (size_t)Ch) + (size_t)(4 * Ch + 4) * 4;  // + the bias vectors
}

bool drs_attn_gate_supported(int Cc, int Ch) {
  static const bool env = !(getenv("DRS_FUSE_GATE") && atoi(getenv("DRS_FUSE_GATE")) == 0);
  if (!env) return false;
  if (Ch == 128) return Cc == 256 || Cc == 128;  // wide variant: weights streamed from L2, pipeline unrolled over the input chunks
  return (Ch == 32 || Ch == 64) && Cc % 64 == 0 && drs_attn_gate_lds_bytes(Cc, Ch) <= 160 * 1024;
}

int drs_launch_attn_gate(const AttnGateDesc& d, hipStream_t s) {
  DRS_REQUIRE(drs_attn_gate_supported(d.Cc, d.Ch), DRS_ERR_SHAPE, "attn_gate: Cc=%d Ch=%d", d.Cc, d.Ch);
  const bool psi_only = !d.out;  // stop at psi (psi_out): the 32-channel stage only
  DRS_REQUIRE(d.x && d.xres && (d.out || (d.psi_out && d.Ch == 32)) && d.w_gate && d.w_wg && d.w_wx && (psi_only || d.w_res) && d.w_psi,
              DRS_ERR_ARG, "attn_gate: null pointer");
  DRS_REQUIRE(!(d.x_cs & 31) && !(d.x_co & 31) && !(d.r_cs & 31) && !(d.r_co & 31) && !(d.out_cs & 31) && !(d.out_co & 31),
              DRS_ERR_SHAPE, "attn_gate: channel strides / offsets must be multiples of 32");
  if ((long long)d.N * d.LH * d.LW == 0) return DRS_OK;
  if (d.Ch == 128) {
    int num_cu = 0;
    const long long nitems = ((long long)d.N * d.LH * ((d.LW + 15) / 16) + 3) / 4;
    auto kern = d.Cc == 256 ? attn_gate_wide2_kernel<8> : attn_gate_wide2_kernel<4>;
    const int ncx = d.Cc / 32;
    const size_t xb = (size_t)4 * ncx * 16 * 128, tb = (size_t)4 * 4 * 16 * 128;
    const size_t lds = (xb > 2 * tb ? xb : 2 * tb) + tb + (size_t)(8 * 64 + 3 * 128 + 4) * 4;
    const int rc = drs_kernel_prepare(reinterpret_cast<const void*>(kern), 160 * 1024, &num_cu);
    if (rc) return rc;
    const long long blocks = nitems < 4LL * num_cu ? nitems : 4LL * num_cu;
    DRS_LAUNCH(kern, dim3((unsigned)blocks), dim3(512), lds, s, d);
    DRS_CHECK_HIP(hipGetLastError());
    return DRS_OK;
  }
  const size_t lds = drs_attn_gate_lds_bytes(d.Cc, d.Ch);
  if (psi_only) return attn_launch<2, true>(d, lds, s);
  return d.Ch == 32 ? attn_launch<2>(d, lds, s) : attn_launch<4>(d, lds, s);
}

int drs_launch_gate_bias(const float* w, const float* b, const float* vec, int vec_stride, float* out, int N, int Cc, int Ch,
                         hipStream_t s) {
  if (N * Ch == 0) return DRS_OK;
  DRS_LAUNCH(gate_bias_kernel, dim3((Ch + 63) / 64, N), dim3(256), 0, s, w, b, vec, vec_stride, out, N, Cc, Ch);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
