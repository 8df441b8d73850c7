// Epilogues shared by the MFMA tap-convolution kernels (conv_mfma.hip, conv_mfma_ws.hip): accumulators in MFMA C/D
// layout -> bias / time embedding / residual / ReLU -> channels-last stores in full 128-byte lines.
#pragma once
#include "mfma_policy.h"

// 16-byte activation store.  DRS_WT_STORES=1 makes it WRITE-THROUGH (`sc1`): the idea was that a kernel which leaves its
// output dirty in the XCDs' L2s pays for the write-back at the kernel boundary (MI355X_MICROARCH.md, row "boundary").
// Measured A/B on one box (tools/build_variant.sh, round 3): 681.5 steps/s written through against 686.3 with plain stores
// - the dropped L2 lines cost the next kernel more than the boundary saves - so plain stores are the default.  (An asm
// store is invisible to hipcc's vmcnt bookkeeping and ends with s_nop 1 so that its data registers are read before the
// next instruction may overwrite them: cdna_hip_programming.md 5.7.)
#ifndef DRS_WT_STORES
#define DRS_WT_STORES 0
#endif
// Experiment switches (tools/build_variant.sh NAME "-DDRS_X_...=1"; never defined in the shipped build): what an item
// epilogue costs without its stores / bias-ReLU-add arithmetic / hi | lo split / lane swap.  Compile-time on purpose: a
// run-time switch read from memory inside the epilogue costs more than what it switches off.
__device__ __forceinline__ void drs_store16(void* p, const u32x4& v) {
#ifdef DRS_X_NOSTORE
  asm volatile("" :: "v"(v), "v"(p));
  return;
#endif
#if DRS_WT_STORES == 2  // experiment: non-temporal
  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif DRS_WT_STORES == 3  // experiment: system-coherent write-through
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif DRS_WT_STORES
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
  *reinterpret_cast<u32x4*>(p) = v;
#endif
}

// ---- epilogue of RPW rows x NT channel tiles held in MFMA layout ---------------------------------------------------
// Lane (lr, kg) holds channels t*16 + kg*4 .. +3 of pixel lr for every n-tile t.  Pairs of n-tiles are exchanged
// between lanes lr and lr^8 (one DPP row rotate) so that each store instruction covers 8 pixels x 32 channels =
// full 128-byte lines: pass h=0 writes pixels 0..7 of the row, pass h=1 pixels 8..15; lanes lr < 8 carry the even
// n-tile of the pair, lanes lr >= 8 the odd one.  Residual / gate loads use the same mapping and are issued per
// row group before the arithmetic (addresses clamped, stores predicated).
// RGS = rows whose residual / gate loads are in flight together: 0 = default (2: bounded register footprint),
// RPW = every load of the item is issued before its first store (gfx9 has ONE counter for loads and stores, so a load
// waited for after a store also waits for that store to reach memory).
template <int RPW, int NT, bool FUSE, int RGS = 0>
__device__ __forceinline__ void tile_epilogue(const TapConv& d, f32x4 (&acc)[RPW][NT], int n, int n0, int ty0, int tx0,
                                              int wave, int lr, int kg, int out_oy, int out_ox) {
  constexpr int NP = NT / 2;
  constexpr int RG = RGS > 0 ? RGS : (RPW > 2 ? 2 : RPW);
  const bool lo = lr < 8;
  const int pl = lr & 7;
  const int csel = (lo ? 0 : 16) + kg * 4;
  float4 bias4[NP], post4[NP];
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) {
    const int co = n0 + pr * 32 + csel;
    bias4[pr] = d.bias ? *reinterpret_cast<const float4*>(d.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.bias2) {
      const float4 b2 = *reinterpret_cast<const float4*>(d.bias2 + co);
      bias4[pr].x += b2.x; bias4[pr].y += b2.y; bias4[pr].z += b2.z; bias4[pr].w += b2.w;
    }
    post4[pr] = d.post_add ? *reinterpret_cast<const float4*>(d.post_add + (size_t)n * d.post_cs + co)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 fw[4][NP];
  if constexpr (FUSE) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int pr = 0; pr < NP; ++pr)
        fw[j][pr] = *reinterpret_cast<const float4*>(d.fuse_w + (size_t)min(j, d.fuse_dim - 1) * d.Cout + n0 + pr * 32 + csel);
  }
#pragma unroll
  for (int rg = 0; rg < RPW; rg += RG) {
    bool valid[RG][2];
    unsigned opix[RG][2];
    int oyx[RG][2][2];
    float gv[RG][2];
    float4 res4[RG][2][NP];
#pragma unroll
    for (int rr = 0; rr < RG; ++rr)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ty = ty0 + wave * RPW + rg + rr, tx = tx0 + pl + 8 * h;
        valid[rr][h] = ty < d.TH && tx < d.TW;
        const int oy = min(ty, d.TH - 1) * d.out_scale + out_oy, ox = min(tx, d.TW - 1) * d.out_scale + out_ox;
        oyx[rr][h][0] = oy; oyx[rr][h][1] = ox;
        opix[rr][h] = ((unsigned)n * d.OH + oy) * d.OW + ox;
        if (d.gate) gv[rr][h] = d.gate[((size_t)n * (d.OH >> 1) + (oy >> 1)) * (d.OW >> 1) + (ox >> 1)];
        if (d.res) {
          const size_t rp = d.res_bstride_zero ? (size_t)((unsigned)oy * d.OW + ox) : (size_t)opix[rr][h];
#pragma unroll
          for (int pr = 0; pr < NP; ++pr)
            res4[rr][h][pr] = *reinterpret_cast<const float4*>(d.res + rp * d.res_cs + d.res_co + n0 + pr * 32 + csel);
        }
      }
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
      const int r = rg + rr;
      f32x4 val[2][NP];
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) {
        f32x4 mine, theirs;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float send = lo ? acc[r][2 * pr + 1][j] : acc[r][2 * pr][j];
          theirs[j] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128, 0xf, 0xf, false));
          mine[j] = lo ? acc[r][2 * pr][j] : acc[r][2 * pr + 1][j];
        }
        val[0][pr] = lo ? mine : theirs;   // pass 0: pixel pl     (lo: own even tile,      hi: partner's odd tile)
        val[1][pr] = lo ? theirs : mine;   // pass 1: pixel pl + 8 (lo: partner's even tile, hi: own odd tile)
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float fz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
          f32x4 v = val[h][pr];
          if (d.gate) v *= gv[rr][h];
          v[0] += bias4[pr].x; v[1] += bias4[pr].y; v[2] += bias4[pr].z; v[3] += bias4[pr].w;
          if (d.relu_pre) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = drs_maxf(v[j], 0.f);
          }
          v[0] += post4[pr].x; v[1] += post4[pr].y; v[2] += post4[pr].z; v[3] += post4[pr].w;
          if (d.res) {
            v[0] += res4[rr][h][pr].x; v[1] += res4[rr][h][pr].y; v[2] += res4[rr][h][pr].z; v[3] += res4[rr][h][pr].w;
          }
          if (d.relu_post) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = drs_maxf(v[j], 0.f);
          }
          if (d.out && valid[rr][h])
            *reinterpret_cast<float4*>(d.out + (size_t)opix[rr][h] * d.out_cs + d.out_co + n0 + pr * 32 + csel) =
                make_float4(v[0], v[1], v[2], v[3]);
          if constexpr (FUSE) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              fz[j] += v[0] * fw[j][pr].x + v[1] * fw[j][pr].y + v[2] * fw[j][pr].z + v[3] * fw[j][pr].w;
          }
        }
        if constexpr (FUSE) {  // y[j] = sum over the pixel's 32 channels = 8 lanes: lr^8 (tile of the pair) x 4 k-groups
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            fz[j] += __shfl_xor(fz[j], 8);
            fz[j] += __shfl_xor(fz[j], 16);
            fz[j] += __shfl_xor(fz[j], 32);
          }
          if (valid[rr][h] && lo && kg == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < d.fuse_dim)
                d.fuse_out[(((size_t)n * d.fuse_dim + j) * d.OH + oyx[rr][h][0]) * d.OW + oyx[rr][h][1]] = fz[j] + d.fuse_b[j];
          }
        }
      }
    }
  }
}

// ---- SP-format epilogue (TapConv::out_sp): the tile is stored as bf16 hi | lo operand halves -------------------------------
// The layer's weights were packed with the output-channel permutation of drs_sp_cout_perm(): lane (lr, kg) then holds
// EIGHT CONSECUTIVE logical channels of pixel lr for every pair of n-tiles: n0 + pr*32 + kg*8 + {0..3} (tile 2*pr) and
// + {4..7} (tile 2*pr + 1) = one 16-byte hi slot and one 16-byte lo slot of the pixel's 128-byte channel group.
// Stores: lanes lr and lr^8 swap one half (DPP row rotate, as above) so that every store instruction writes 8 pixels x
// [4 hi slots | 4 lo slots] = full 128-byte lines: pass 0 covers pixels 0..7 of the row, pass 1 pixels 8..15; lanes
// lr < 8 store hi slots, lanes lr >= 8 lo slots.
__device__ __forceinline__ int drs_sp_group_bytes(int cs) { return cs >= 32 ? 64 : cs * 2; }  // bytes of the hi half of a group

typedef __bf16 drs_bf16x8 __attribute__((ext_vector_type(8)));
// hi = bf16(v), lo = bf16(v - hi) of 8 values (drs_split2: two values per conversion instruction; the item epilogue of the
// wave-specialised kernels was ~830 vector instructions per wave with element-wise conversions)
__device__ __forceinline__ void drs_sp_split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) { const uint2 s_ = drs_split2(v[2 * p], v[2 * p + 1]); hi[p] = s_.x; lo[p] = s_.y; }
}
__device__ __forceinline__ void drs_sp_split4(const float (&v)[4], unsigned (&hi)[2], unsigned (&lo)[2]) {  // half a slot
  { const uint2 s_ = drs_split2(v[0], v[1]); hi[0] = s_.x; lo[0] = s_.y; }
  { const uint2 s_ = drs_split2(v[2], v[3]); hi[1] = s_.x; lo[1] = s_.y; }
}
__device__ __forceinline__ void drs_sp_join8(const u32x4& hi, const u32x4& lo, float (&v)[8]) {
  const drs_bf16x8 h = __builtin_bit_cast(drs_bf16x8, hi), l = __builtin_bit_cast(drs_bf16x8, lo);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (float)h[j] + (float)l[j];
}
__device__ __forceinline__ u32x4 drs_dpp_swap8(const u32x4& x) {  // value of lane lr ^ 8 (same k-group row)
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x[j], 0x128, 0xf, 0xf, false);
  return r;
}

// OUT2: also write TapConv::out2 (= value + post2).  Register-lean: everything is loaded right where it is used.
// LEAN: bias only (no gate / ReLU / per-image add / residual): the transposed convolution, whose four accumulator sets
// leave no registers for the general form.
template <int RPW, int NT, bool OUT2 = false, bool LEAN = false>
__device__ __forceinline__ void tile_epilogue_sp(const TapConv& d, f32x4 (&acc)[RPW][NT], int n, int n0, int ty0, int tx0,
                                                 int wave, int lr, int kg, int out_oy, int out_ox) {
  constexpr int NP = NT / 2;
  const bool lo = lr < 8;
  const int pl = lr & 7;
  auto load8 = [&](const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  };
  // one store pass per half row: lanes lr < 8 write hi slots, lanes lr >= 8 lo slots, of pixels pl (+8): full 128-byte lines
  auto store_sp = [&](float* base, int cs, int co, const float (&w8)[8], int ty, int oy) {
    u32x4 H, L;
    drs_sp_split8(w8, H, L);
    const u32x4 got = drs_dpp_swap8(lo ? L : H);  // lr < 8 receives the partner's hi, lr >= 8 the partner's lo
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int tx = tx0 + pl + 8 * h;
      if (ty < d.TH && tx < d.TW) {
        const size_t opix = ((size_t)n * d.OH + oy) * d.OW + (size_t)tx * d.out_scale + out_ox;
        char* g = reinterpret_cast<char*>(base) + (opix * cs + co) * 4 + (lo ? 0 : 64) + kg * 16;
        // pass 0: lanes lr < 8 own the pixel (own hi), lanes lr >= 8 store the received lo of pixel lr - 8;
        // pass 1: lanes lr < 8 store the received hi of pixel lr + 8, lanes lr >= 8 own the pixel (own lo)
        drs_store16(g, (h == 0) ? (lo ? H : got) : (lo ? got : L));
      }
    }
  };
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) {
    const int cg = n0 + pr * 32;       // first logical channel of the 32-channel group
    const int c8 = cg + kg * 8;        // this lane's 8 channels
    float bias8[8], post8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias8[j] = post8[j] = 0.f;
    if (d.bias) load8(d.bias + c8, bias8);
    if (!LEAN && d.bias2) {
      float b2[8];
      load8(d.bias2 + c8, b2);
#pragma unroll
      for (int j = 0; j < 8; ++j) bias8[j] += b2[j];
    }
    if (!LEAN && d.post_add) load8(d.post_add + (size_t)n * d.post_cs + c8, post8);
    // (loaded before the first store: gfx9 counts loads and stores in ONE counter, a load issued after a store is only
    // complete once that store has reached memory)
    float post2_8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) post2_8[j] = 0.f;
    if constexpr (OUT2) {
      if (d.out2) load8(d.post2 + (size_t)n * d.post2_cs + c8, post2_8);
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int ty = ty0 + wave * RPW + r;
      const int oy = min(ty, d.TH - 1) * d.out_scale + out_oy;
      const int ox_own = min(tx0 + lr, d.TW - 1) * d.out_scale + out_ox;
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = acc[r][2 * pr][j]; v[4 + j] = acc[r][2 * pr + 1][j]; }
      if constexpr (LEAN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bias8[j];
      } else {
      if (d.gate) {
        const float gv = d.gate[((size_t)n * (d.OH >> 1) + (oy >> 1)) * (d.OW >> 1) + (ox_own >> 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= gv;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] += bias8[j];
        if (d.relu_pre) v[j] = drs_maxf(v[j], 0.f);
        v[j] += post8[j];
      }
      }
      if (!LEAN && d.res) {
        const size_t rp = d.res_bstride_zero ? (size_t)oy * d.OW + ox_own : ((size_t)n * d.OH + oy) * d.OW + ox_own;
        float rv[8];
        if (d.res_sp) {
          const char* g = reinterpret_cast<const char*>(d.res) + (rp * d.res_cs + d.res_co + cg) * 4 + kg * 16;
          drs_sp_join8(*reinterpret_cast<const u32x4*>(g), *reinterpret_cast<const u32x4*>(g + 64), rv);
        } else {
          load8(d.res + rp * d.res_cs + d.res_co + c8, rv);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rv[j];
      }
      if (!LEAN && d.relu_post) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = drs_maxf(v[j], 0.f);
      }
      if (d.out) store_sp(d.out, d.out_cs, d.out_co + cg, v, ty, oy);
      if constexpr (OUT2) {
        if (d.out2) {
          float p2[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) p2[j] = post2_8[j] + v[j];
          store_sp(d.out2, d.out2_cs, d.out2_co + cg, p2, ty, oy);
        }
      }
    }
  }
}

// The SP wave-specialised kernel's form: the per-item constants (bias, per-image adds) arrive in registers (its mover
// waves stage them in LDS: a global load at this point would expose a full memory round trip per item on every SIMD),
// row addresses are advanced instead of recomputed, no gate / residual (that kernel does not take such layers).
struct SpEpiConst { float bias[8], post[8], post2[8]; };  // this lane's 8 channels: bias (+ bias2), post_add, post2 (zeros if absent)
template <int RPW, bool OUT2>
__device__ __forceinline__ void tile_epilogue_sp_pre(const TapConv& d, f32x4 (&acc)[RPW][2], const SpEpiConst& k, int n, int cg,
                                                     int ty0, int tx0, int wave, int lr, int kg) {
  const bool lo = lr < 8;
  const int pl = lr & 7;
  // Written for FEW vector instructions (this epilogue runs on all eight consumer waves at once with the matrix pipe idle):
  // the affine part two values at a time (v_pk_add_f32), the two optional ReLUs behind launch-uniform branches (as
  // maximum(v, flag ? 0 : -inf) they compiled to a maximum AND a select per value), the hi / lo exchange with the lane 8
  // further by bank-masked DPP moves that leave the other half of the row untouched (no selects).  drs_maxf propagates a NaN.
  const bool relu_pre = d.relu_pre != 0, relu_post = d.relu_post != 0;
  const int tyb = ty0 + wave * RPW;
  const size_t pix0 = ((size_t)n * d.OH + tyb) * d.OW + tx0 + pl;  // (out_scale 1, no phase offset: 3x3 stride 1)
  const int lane_b = (lo ? 0 : 64) + kg * 16;
  char* o1 = d.out ? reinterpret_cast<char*>(d.out) + (pix0 * d.out_cs + d.out_co + cg) * 4 + lane_b : nullptr;
  char* o2 = (OUT2 && d.out2) ? reinterpret_cast<char*>(d.out2) + (pix0 * d.out2_cs + d.out2_co + cg) * 4 + lane_b : nullptr;
  const size_t row1 = (size_t)d.OW * d.out_cs * 4, row2 = OUT2 ? (size_t)d.OW * d.out2_cs * 4 : 0;
  const int h1 = 8 * d.out_cs * 4, h2 = OUT2 ? 8 * d.out2_cs * 4 : 0;
  const bool ok0 = tx0 + pl < d.TW, ok1 = tx0 + pl + 8 < d.TW;
  drs_f32x2 kb[4], kp[4], kq[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    kb[p] = drs_f32x2{k.bias[2 * p], k.bias[2 * p + 1]};
    kp[p] = drs_f32x2{k.post[2 * p], k.post[2 * p + 1]};
    kq[p] = drs_f32x2{k.post2[2 * p], k.post2[2 * p + 1]};
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    if (tyb + r < d.TH) {
      drs_f32x2 v[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) v[p] = drs_f32x2{acc[r][p >> 1][2 * (p & 1)], acc[r][p >> 1][2 * (p & 1) + 1]} + kb[p];
      if (relu_pre) {
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p] = drs_f32x2{drs_maxf(v[p][0], 0.f), drs_maxf(v[p][1], 0.f)};
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) v[p] = v[p] + kp[p];
      if (relu_post) {
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p] = drs_f32x2{drs_maxf(v[p][0], 0.f), drs_maxf(v[p][1], 0.f)};
      }
#ifdef DRS_X_NOAFFINE
#pragma unroll
      for (int p = 0; p < 4; ++p) v[p] = drs_f32x2{acc[r][p >> 1][2 * (p & 1)], acc[r][p >> 1][2 * (p & 1) + 1]};
#endif
      auto put = [&](char* g, int hb, const drs_f32x2 (&w2)[4]) __attribute__((always_inline)) {
        const float w8[8] = {w2[0][0], w2[0][1], w2[1][0], w2[1][1], w2[2][0], w2[2][1], w2[3][0], w2[3][1]};
        u32x4 H, L;
#ifdef DRS_X_NOSPLIT
        H = u32x4{__float_as_uint(w8[0]), __float_as_uint(w8[1]), __float_as_uint(w8[2]), __float_as_uint(w8[3])};
        L = u32x4{__float_as_uint(w8[4]), __float_as_uint(w8[5]), __float_as_uint(w8[6]), __float_as_uint(w8[7])};
#else
        drs_sp_split8(w8, H, L);
#endif
#ifdef DRS_X_NOSWAP
        if (ok0) drs_store16(g, H);
        if (ok1) drs_store16(g + hb, L);
#else
        // first store: lanes lr < 8 their own hi, lanes lr >= 8 the lo of the lane 8 below; second store: lanes lr < 8 the hi of
        // the lane 8 above, lanes lr >= 8 their own lo.  row_ror:8 = the lane lr ^ 8 of the 16-lane row; bank mask 0xc writes
        // lanes 8 - 15 only, 0x3 lanes 0 - 7 only; the rest keeps `old`.
        u32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a[j] = (unsigned)__builtin_amdgcn_update_dpp((int)H[j], (int)L[j], 0x128, 0xf, 0xc, false);
          b[j] = (unsigned)__builtin_amdgcn_update_dpp((int)L[j], (int)H[j], 0x128, 0xf, 0x3, false);
        }
        if (ok0) drs_store16(g, a);
        if (ok1) drs_store16(g + hb, b);
#endif
      };
      if (o1) put(o1 + r * row1, h1, v);
      if constexpr (OUT2) {
        if (o2) {
          drs_f32x2 p2[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) p2[p] = v[p] + kq[p];
          put(o2 + r * row2, h2, p2);
        }
      }
    }
  }
}

// ---- fused 1x1 projection epilogue (the UNet's `output` conv riding on up_convs.2) ------------------------------
// Works in MFMA layout: lane (lr, kg) holds channels t*16 + kg*4 .. +3 of pixel lr.  y[j] = fuse_b[j] + sum over the 32
// channels of (acc + bias) * fuse_w[j][co]: 8 in-lane products, then 2 cross-lane steps over the 4 k-group lanes.
// The 32-channel tensor itself is written only if d.out is set (parity taps); production runs never store it.
template <int RPW, int NT>
__device__ __forceinline__ void fuse_epilogue(const TapConv& d, f32x4 (&acc)[RPW][NT], int n, int n0, int ty0, int tx0,
                                              int wave, int lr, int kg) {
  float4 bias4[NT], fw[4][NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int co = n0 + t * 16 + kg * 4;
    bias4[t] = d.bias ? *reinterpret_cast<const float4*>(d.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      fw[j][t] = *reinterpret_cast<const float4*>(d.fuse_w + (size_t)min(j, d.fuse_dim - 1) * d.Cout + co);
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int ty = ty0 + wave * RPW + r, tx = tx0 + lr;
    const bool valid = ty < d.TH && tx < d.TW;
    const int oy = min(ty, d.TH - 1) * d.out_scale + d.out_oy, ox = min(tx, d.TW - 1) * d.out_scale + d.out_ox;
    float fz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 v = acc[r][t];
      v[0] += bias4[t].x; v[1] += bias4[t].y; v[2] += bias4[t].z; v[3] += bias4[t].w;
      if (d.out && valid)
        *reinterpret_cast<float4*>(d.out + (((size_t)n * d.OH + oy) * d.OW + ox) * d.out_cs + d.out_co + n0 + t * 16 + kg * 4) =
            make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) fz[j] += v[0] * fw[j][t].x + v[1] * fw[j][t].y + v[2] * fw[j][t].z + v[3] * fw[j][t].w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fz[j] += __shfl_xor(fz[j], 16);
      fz[j] += __shfl_xor(fz[j], 32);
    }
    if (valid && kg == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < d.fuse_dim) d.fuse_out[(((size_t)n * d.fuse_dim + j) * d.OH + oy) * d.OW + ox] = fz[j] + d.fuse_b[j];
    }
  }
}


// The same projection on the matrix pipe (split-bf16 kernels): y = W (acc + bias) is one more 16x16x32 GEMM per row of 16
// pixels.  A lane's eight accumulators of the two channel tiles (channels kg*4 + j and 16 + kg*4 + j) are its eight
// K-elements of the B operand once split into bf16 hi | lo; the A operand holds fuse_w in the same K order (rows >=
// fuse_dim are zero), and the result rows 0..3 land in the lanes of k-group 0: no cross-lane reduction, no LDS round trips
// (the shuffle form costs 8 ds_bpermute per row and was 35 us of up_convs.2's 220).  Same 2^-16 operand rounding as every
// other product of the split-bf16 path.
struct FuseEpiConst { float4 b0, b1; float w8[8]; float fb[4]; };  // bias of this lane's 8 channels, fuse_w row lr (same 8), fuse_b
template <int RPW>
__device__ __forceinline__ void fuse_epilogue_mfma_pre(const TapConv& d, f32x4 (&acc)[RPW][2], const FuseEpiConst& k, int n, int n0,
                                                       int ty0, int tx0, int wave, int lr, int kg) {
  using P = PolicyBF16X3;
  typename P::Frag wfr;
  {
    float w8[8];
    // output channel j sits in MFMA row 4 * j (the caller hands row lr the weights of channel lr >> 2): its results land in
    // register 0 of the lanes of k-group j, so ONE store instruction writes all fuse_dim planes (48 lanes) instead of one
    // 16-lane instruction per plane - a store costs its wave ~250 cycles under load, whatever it carries
    const float keep = ((lr & 3) == 0 && (lr >> 2) < d.fuse_dim) ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) w8[j] = k.w8[j] * keep;
    u32x4 h, l;
    drs_sp_split8(w8, h, l);
    wfr = typename P::Frag{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, l)};
  }
  const size_t plane = (size_t)d.OH * d.OW;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int ty = ty0 + wave * RPW + r, tx = tx0 + lr;
    const bool valid = ty < d.TH && tx < d.TW;
    const int oy = min(ty, d.TH - 1) * d.out_scale + d.out_oy, ox = min(tx, d.TW - 1) * d.out_scale + d.out_ox;
    float v[8];
    v[0] = acc[r][0][0] + k.b0.x; v[1] = acc[r][0][1] + k.b0.y; v[2] = acc[r][0][2] + k.b0.z; v[3] = acc[r][0][3] + k.b0.w;
    v[4] = acc[r][1][0] + k.b1.x; v[5] = acc[r][1][1] + k.b1.y; v[6] = acc[r][1][2] + k.b1.z; v[7] = acc[r][1][3] + k.b1.w;
    if (d.out && valid) {  // parity taps only: production runs never store the 32-channel tensor
      float* o = d.out + (((size_t)n * d.OH + oy) * d.OW + ox) * d.out_cs + d.out_co + n0 + kg * 4;
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(o + 16) = make_float4(v[4], v[5], v[6], v[7]);
    }
    u32x4 h, l;
    drs_sp_split8(v, h, l);
    const typename P::Frag vf{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, l)};
    const f32x4 y = P::mma(wfr, vf, f32x4{0.f, 0.f, 0.f, 0.f});  // register 0 of the lanes of k-group j: output j of pixel lr
    if (valid && kg < d.fuse_dim) {
      const float fbk = kg == 0 ? k.fb[0] : kg == 1 ? k.fb[1] : kg == 2 ? k.fb[2] : k.fb[3];
      d.fuse_out[((size_t)n * d.fuse_dim + kg) * plane + (size_t)oy * d.OW + ox] = y[0] + fbk;
    }
  }
}
template <int RPW>
__device__ __forceinline__ void fuse_epilogue_mfma(const TapConv& d, f32x4 (&acc)[RPW][2], int n, int n0, int ty0, int tx0,
                                                   int wave, int lr, int kg) {
  FuseEpiConst k;
  k.b0 = k.b1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (d.bias) {
    k.b0 = *reinterpret_cast<const float4*>(d.bias + n0 + kg * 4);
    k.b1 = *reinterpret_cast<const float4*>(d.bias + n0 + 16 + kg * 4);
  }
  const int m = min(lr >> 2, d.fuse_dim - 1);  // (row lr of the projection operand = channel lr >> 2: fuse_epilogue_mfma_pre)
  const float4 w0 = *reinterpret_cast<const float4*>(d.fuse_w + (size_t)m * d.Cout + n0 + kg * 4);
  const float4 w1 = *reinterpret_cast<const float4*>(d.fuse_w + (size_t)m * d.Cout + n0 + 16 + kg * 4);
  k.w8[0] = w0.x; k.w8[1] = w0.y; k.w8[2] = w0.z; k.w8[3] = w0.w; k.w8[4] = w1.x; k.w8[5] = w1.y; k.w8[6] = w1.z; k.w8[7] = w1.w;
#pragma unroll
  for (int j = 0; j < 4; ++j) k.fb[j] = d.fuse_b[min(j, d.fuse_dim - 1)];
  fuse_epilogue_mfma_pre<RPW>(d, acc, k, n, n0, ty0, tx0, wave, lr, kg);
}
