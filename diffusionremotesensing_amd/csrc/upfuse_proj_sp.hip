// The top decoder stage's composite (ups.2.transform o x-half of up_convs.2 o output: upfuse_sp.hip has the algebra; reference
// UNet_model_superres.py:206-207,376-377,379) as a STREAMING direct-operand kernel, kernel rows in the MFMA's M dimension.
//
// With the `output` projection folded in the stage has out_dim <= 3 outputs, and the composite's 25 (phase, tap) pairs per
// low-resolution cell - 5 along y ((py 0: ty 0,1,2), (py 1: ty 1,2)) x 5 along x - fit the rows of ONE 16-row tile per x group:
// row 3 j + o = y pair j of output o.  An input row of 16 cells is then multiplied once per x group and 32-channel chunk
// (5 x nck MFMA triples; the wave-specialised kernel's folded form issues 25 x nck), into two accumulators (x-phase 0: groups
// tx 0,1,2; x-phase 1: tx 1,2); an output cell row (two pixel rows) is the sum of the right rows of three consecutive input
// rows' accumulators (lane reads).  Input rows stream through a short register ring (conv3x3_proj_sp_kernel's scheme): no LDS
// window, no mover waves, no counters.  The epilogue is the folded form's: + bias + what the output tensor already holds (the
// att-half) + the edge vectors on the image border (32-channel vectors whose channel 8 o is output o: upfuse_sp.hip).
#include <stdio.h>
#include <stdlib.h>

#include "conv_epilogue.h"
#include "mfma_policy.h"

namespace {

constexpr int RBC = 8;  // output cell rows per strip
struct RowOpU { PolicyBF16X3::Frag c, e; };  // c: cell x0 + lr of the row; e: lanes 0 / 15 hold cells x0 - 1 / x0 + 16

__device__ __forceinline__ bf16x8 dpp_shift_u(const bf16x8& edge, const bf16x8& own, bool left) {
  const u32x4 e = __builtin_bit_cast(u32x4, edge), r = __builtin_bit_cast(u32x4, own);
  u32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j)  // row_shr:1 (cell - 1): lane lr takes lane lr - 1, lane 0 keeps `edge`; row_shl:1 (cell + 1): lane 15 keeps it
    o[j] = left ? (unsigned)__builtin_amdgcn_update_dpp((int)e[j], (int)r[j], 0x111, 0xf, 0xf, false)
                : (unsigned)__builtin_amdgcn_update_dpp((int)e[j], (int)r[j], 0x101, 0xf, 0xf, false);
  return __builtin_bit_cast(bf16x8, o);
}
// (phase, tap) pairs along one axis, in row order j = 0..4, and the x groups in streaming order (upfuse_sp.hip)
__host__ __device__ constexpr int up_tap_p(int j) { return j >= 3 ? 1 : 0; }
__host__ __device__ constexpr int up_tap_t(int j) { return j >= 3 ? j - 2 : j; }
__host__ __device__ constexpr bool up_pair(int p, int t, int kv, int kw) { return p + kv - kw == 2 * (t - 1); }

template <int NCK>
__global__ __launch_bounds__(256, 3) void upfuse_proj_sp_kernel(UpFuseDesc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using P = PolicyBF16X3;
  using Frag = typename P::Frag;
  constexpr int NR = RBC + 2, PF = 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kg = lane >> 4;
  constexpr int IMG = NCK * 5 * 4 * 16 * 16;  // one operand image: [chunk][x pair 5][k-group][16 rows] slots
  char* sW = smem;
  float* sB = reinterpret_cast<float*>(smem + 2 * IMG);
  {
    const char* w = reinterpret_cast<const char*>(d.w);
    for (int o = tid * 16; o < 2 * IMG; o += 256 * 16) *reinterpret_cast<u32x4*>(sW + o) = *reinterpret_cast<const u32x4*>(w + o);
    if (tid < 4) sB[tid] = d.bias[min(tid, 3) * 8];  // (logical channel 8 o of the folded 32-channel layer)
  }
  __syncthreads();
  const int OH = 2 * d.LH, OW = 2 * d.LW;
  const int gx = (d.LW + 15) >> 4, gy = (d.LH + RBC - 1) / RBC;
  const int total = d.N * gy * gx;
  const int xcd = blockIdx.x & 7, member = blockIdx.x >> 3, members = gridDim.x >> 3;
  const int t_lo = (int)((long long)total * xcd / 8), t_hi = (int)((long long)total * (xcd + 1) / 8);
  const int stride = members * 4, first = member * 4 + wave;
  const int pixb = d.in_cs * 4;
  const char* zero = reinterpret_cast<const char*>(d.zero_line) + kg * 16;
  const char* wlane = sW + ((size_t)kg * 16 + lr) * 16;
  const size_t plane = (size_t)OH * OW;
  const float bk = sB[kg];
  const int ko = min(kg, d.fuse_dim - 1);
  for (int q0 = t_lo + first; q0 < t_hi; q0 += stride) {
    int q = q0;
    const int x0 = (q % gx) * 16; q /= gx;
    const int m0 = (q % gy) * RBC;
    const int n = q / gy;
    const int cx = x0 + lr;                 // this lane's cell column
    const bool own_ok = cx < d.LW && kg < d.fuse_dim;
    RowOpU R[PF + 1][NCK];
    auto load_row = [&](int slot, int wr) __attribute__((always_inline)) {
      const int iy = m0 - 1 + wr;
      const bool ok = iy >= 0 && iy < d.LH && cx < d.LW;
      const char* base = reinterpret_cast<const char*>(d.in) +
                         ((((long long)n * d.LH + iy) * d.LW + cx) * d.in_cs + d.in_co) * 4 + kg * 16;
#pragma unroll
      for (int c = 0; c < NCK; ++c) {
        const char* p = ok ? base + c * 128 : zero;
        R[slot][c].c = Frag{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(ok ? p + 64 : zero)};
      }
      if (lr == 0 || lr == 15) {
        const int ex = lr == 0 ? x0 - 1 : x0 + 16;
        const bool eok = iy >= 0 && iy < d.LH && ex >= 0 && ex < d.LW;
        // (lane 15 of a ragged last strip: its own cell may lie outside while the edge cell does too - both read zeros)
        const char* be = reinterpret_cast<const char*>(d.in) +
                         ((((long long)n * d.LH + iy) * d.LW + ex) * d.in_cs + d.in_co) * 4 + kg * 16;
#pragma unroll
        for (int c = 0; c < NCK; ++c) {
          const char* pe = eok ? be + c * 128 : zero;
          R[slot][c].e = Frag{*reinterpret_cast<const bf16x8*>(pe), *reinterpret_cast<const bf16x8*>(eok ? pe + 64 : zero)};
        }
      }
      (void)pixb;
    };
    // what the output tensor already holds + the edge vectors, for this lane's cells: float2 = (x-phase 0, x-phase 1)
    float* obase = d.fuse_out + ((size_t)n * d.fuse_dim + ko) * plane + 2 * min(cx, d.LW - 1);
    f32x4 S[3][2];  // accumulators of three consecutive input rows: [row % 3][x-phase]
#pragma unroll
    for (int i = 0; i < PF; ++i) load_row(i, i);
#pragma unroll
    for (int wr = 0; wr < NR; ++wr) {
      if (wr + PF < NR) load_row((wr + PF) % (PF + 1), wr + PF);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCK; ++c) {
        const RowOpU& r = R[wr % (PF + 1)][c];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const Frag a = tx == 1 ? r.c : Frag{dpp_shift_u(r.e.hi, r.c.hi, tx == 0), dpp_shift_u(r.e.lo, r.c.lo, tx == 0)};
          // x groups in image order: (tx0,px0) (tx1,px0) (tx1,px1) (tx2,px0) (tx2,px1)
          const int g0 = tx == 0 ? 0 : (tx == 1 ? 1 : 3);
          s0 = P::mma(P::load(wlane, (size_t)IMG, (size_t)((c * 5 + g0) * 4 * 16) * 16), a, s0);
          if (tx >= 1) s1 = P::mma(P::load(wlane, (size_t)IMG, (size_t)((c * 5 + g0 + 1) * 4 * 16) * 16), a, s1);
        }
      }
      S[wr % 3][0] = s0;
      S[wr % 3][1] = s1;
      __builtin_amdgcn_sched_barrier(0);
      if (wr >= 2) {
        // output cell row m = m0 + wr - 2: input rows wr - 2 (ty 0 -> its row is m - 1), wr - 1 (ty 1), wr (ty 2)
        const int m = m0 + wr - 2;
        const f32x4(&A)[2] = S[(wr - 2) % 3];
        const f32x4(&B)[2] = S[(wr - 1) % 3];
        const f32x4(&C)[2] = S[wr % 3];
        float v[2][2][3];  // [py][px][o]
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int o = 0; o < 3; ++o) {
            // row 3 j + o of pair j: (py0,ty0) = 0 from A, (py0,ty1) = 1 from B, (py0,ty2) = 2 from C; (py1,ty1) = 3 from B, (py1,ty2) = 4 from C
            auto pick = [&](const f32x4& t, int row) __attribute__((always_inline)) { return __shfl(t[row & 3], lr + 16 * (row >> 2), 64); };
            v[0][px][o] = pick(A[px], 0 + o) + pick(B[px], 3 + o) + pick(C[px], 6 + o);
            v[1][px][o] = pick(B[px], 9 + o) + pick(C[px], 12 + o);
          }
        if (m < d.LH) {
#pragma unroll
          for (int py = 0; py < 2; ++py) {
            const int oy = 2 * m + py;
            float w0 = kg == 0 ? v[py][0][0] : (kg == 1 ? v[py][0][1] : v[py][0][2]);
            float w1 = kg == 0 ? v[py][1][0] : (kg == 1 ? v[py][1][1] : v[py][1][2]);
            float2* op = reinterpret_cast<float2*>(obase + (size_t)oy * OW);
            float2 prev = make_float2(0.f, 0.f);
            if (own_ok && d.fuse_acc) prev = *op;
            w0 += bk + prev.x;
            w1 += bk + prev.y;
            if (d.eh && own_ok) {
              const int ch = ko * 8;
              if (oy == 0 || oy == OH - 1) {
                const float* e = d.eh + (((size_t)n * 2 + (oy ? 1 : 0)) * OW + 2 * cx) * d.Ch + ch;
                w0 += e[0];
                w1 += e[d.Ch];
              } else {
                if (cx == 0) w0 += d.ev[(((size_t)n * 2 + 0) * OH + oy) * d.Ch + ch];
                if (2 * cx + 1 == OW - 1) w1 += d.ev[(((size_t)n * 2 + 1) * OH + oy) * d.Ch + ch];
              }
            }
            if (own_ok) *op = make_float2(w0, w1);
          }
        }
      }
    }
  }
}

// operand image of upfuse_proj_sp_kernel from the folded up_convs.2 (vp: [32][Cc + Ch][9], row 8 o = output o:
// drs_launch_upfuse_fold_proj) and ups.2.transform: dst [image 2][chunk][x pair 5][k-group 4][16 rows] slots, row 3 j + o
__global__ __launch_bounds__(128) void upfuse_proj_pack_kernel(const float* __restrict__ vp, const float* __restrict__ t_w, int Cc,
                                                               int Ch, int fuse_dim, char* __restrict__ dst) {
  const int nck = Cc >> 5;
  const int nslots = nck * 5 * 4 * 16;
  const size_t img = (size_t)nslots * 16;
  const int cinv = Cc + Ch;
  for (int s = blockIdx.x * 128 + threadIdx.x; s < nslots; s += gridDim.x * 128) {
    const int row = s & 15, q = (s >> 4) & 3, g = (s >> 6) % 5, ck = s / 320;
    const int j = row / 3, o = row - 3 * j;
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < 15 && o < fuse_dim) {
      const int tx = g == 0 ? 0 : (g <= 2 ? 1 : 2), px = (g == 2 || g == 4) ? 1 : 0, py = up_tap_p(j), ty = up_tap_t(j);
      const int ci0 = ck * 32 + q * 8;
      for (int kvy = 0; kvy < 3; ++kvy)
        for (int kwy = 0; kwy < 3; ++kwy) {
          if (!up_pair(py, ty, kvy, kwy)) continue;
          for (int kvx = 0; kvx < 3; ++kvx)
            for (int kwx = 0; kwx < 3; ++kwx) {
              if (!up_pair(px, tx, kvx, kwx)) continue;
              const float* vq = vp + (size_t)(8 * o) * cinv * 9 + kvy * 3 + kvx;
              const float* wq = t_w + (size_t)ci0 * Cc * 9 + kwy * 3 + kwx;
              for (int c = 0; c < Cc; ++c) {
                const float vv = vq[(size_t)c * 9];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += vv * wq[((size_t)e * Cc + c) * 9];
              }
            }
        }
    }
    PolicyBF16X3::cvt_store(dst, img, (size_t)s * 16, x);
  }
}

}  // namespace

size_t drs_upfuse_proj_weight_bytes(int Cc) { return (size_t)2 * (Cc / 32) * 5 * 4 * 16 * 16; }

bool drs_upfuse_proj_supported(int Cc, int Ch, int fuse_dim) {
  return Ch == 32 && (Cc == 32 || Cc == 64) && fuse_dim >= 1 && fuse_dim <= 3;
}

int drs_launch_upfuse_proj_pack(const float* vp, const float* t_w, int Cc, int Ch, int fuse_dim, void* dst, hipStream_t s) {
  DRS_REQUIRE(vp && t_w && dst && drs_upfuse_proj_supported(Cc, Ch, fuse_dim), DRS_ERR_SHAPE, "upfuse_proj_pack: Cc=%d Ch=%d fuse_dim=%d", Cc, Ch, fuse_dim);
  const int nslots = (Cc / 32) * 5 * 4 * 16;
  DRS_LAUNCH(upfuse_proj_pack_kernel, dim3((nslots + 127) / 128), dim3(128), 0, s, vp, t_w, Cc, Ch, fuse_dim, (char*)dst);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}

// d.w: the image of drs_launch_upfuse_proj_pack; d.bias / d.eh / d.ev: those of the folded 32-channel layer (upfuse_sp.hip)
int drs_launch_upfuse_proj(const UpFuseDesc& d, hipStream_t s) {
  DRS_REQUIRE(d.in && d.w && d.bias && d.zero_line && d.fuse_out && d.proj && !d.res && !d.out && !d.out2, DRS_ERR_ARG, "upfuse_proj: bad descriptor");
  DRS_REQUIRE(drs_upfuse_proj_supported(d.Cc, d.Ch, d.fuse_dim) && (d.in_cs & 31) == 0 && (d.in_co & 31) == 0, DRS_ERR_SHAPE, "upfuse_proj: channels");
  DRS_REQUIRE((d.eh == nullptr) == (d.ev == nullptr), DRS_ERR_ARG, "upfuse_proj: edge vectors");
  if ((size_t)d.N * d.LH * d.LW == 0) return DRS_OK;
  int num_cu = 0;
  const void* kern = d.Cc == 64 ? reinterpret_cast<const void*>(upfuse_proj_sp_kernel<2>) : reinterpret_cast<const void*>(upfuse_proj_sp_kernel<1>);
  {
    const int rc = drs_kernel_prepare(kern, 0, &num_cu);
    if (rc) return rc;
  }
  const size_t lds = drs_upfuse_proj_weight_bytes(d.Cc) + 64;
  const long long strips = (long long)d.N * ((d.LH + RBC - 1) / RBC) * ((d.LW + 15) >> 4);
  long long blocks = (long long)num_cu * 3;
  if (blocks * 4 > strips) blocks = (strips + 3) / 4;
  blocks = (blocks + 7) / 8 * 8;
  if (d.Cc == 64) DRS_LAUNCH(upfuse_proj_sp_kernel<2>, dim3((unsigned)blocks), dim3(256), lds, s, d);
  else DRS_LAUNCH(upfuse_proj_sp_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, s, d);
  DRS_CHECK_HIP(hipGetLastError());
  return DRS_OK;
}
