// Operand policies and launch geometry shared by the MFMA tap-convolution kernels (conv_mfma.hip, conv_mfma_ws.hip).
#pragma once
#include "drs_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- operand policies -------------------------------------------------------------------------------------------
struct PolicyF32 {  // exact fp32: 4 x v_mfma_f32_16x16x4_f32 per slot pair
  static constexpr int SLOT_CH = 4, IMAGES = 1, IMPL = DRS_IMPL_MFMA_F32;
  struct Frag { f32x4 v; };
  __device__ static void cvt_store(char* base, size_t img_stride, size_t off, const float* x) {
    *reinterpret_cast<f32x4*>(base + off) = f32x4{x[0], x[1], x[2], x[3]};
  }
  __device__ static Frag load(const char* base, size_t img_stride, size_t off) {
    return Frag{*reinterpret_cast<const f32x4*>(base + off)};
  }
  __device__ static f32x4 mma(const Frag& w, const Frag& a, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[j], a.v[j], c, 0, 0, 0);
    return c;
  }
};
struct PolicyF16 {  // fp16 operands, fp32 accumulate: 1 x v_mfma_f32_16x16x32_f16
  static constexpr int SLOT_CH = 8, IMAGES = 1, IMPL = DRS_IMPL_MFMA_F16;
  struct Frag { half8 v; };
  __device__ static void cvt_store(char* base, size_t img_stride, size_t off, const float* x) {
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)x[j];
    *reinterpret_cast<half8*>(base + off) = h;
  }
  __device__ static Frag load(const char* base, size_t img_stride, size_t off) {
    return Frag{*reinterpret_cast<const half8*>(base + off)};
  }
  __device__ static f32x4 mma(const Frag& w, const Frag& a, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(w.v, a.v, c, 0, 0, 0);
  }
};
struct PolicyBF16X3 {  // x = hi + lo (both bf16, 16 mantissa bits together): w*a ~ wl*ah + wh*al + wh*ah
  static constexpr int SLOT_CH = 8, IMAGES = 2, IMPL = DRS_IMPL_MFMA_BF16X3;
  struct Frag { bf16x8 hi, lo; };
  __device__ static void cvt_store(char* base, size_t img_stride, size_t off, const float* x) {
    u32x4 h, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) { const uint2 s_ = drs_split2(x[2 * p], x[2 * p + 1]); h[p] = s_.x; l[p] = s_.y; }
    *reinterpret_cast<u32x4*>(base + off) = h;
    *reinterpret_cast<u32x4*>(base + img_stride + off) = l;
  }
  __device__ static Frag load(const char* base, size_t img_stride, size_t off) {
    return Frag{*reinterpret_cast<const bf16x8*>(base + off), *reinterpret_cast<const bf16x8*>(base + img_stride + off)};
  }
  __device__ static f32x4 mma(const Frag& w, const Frag& a, f32x4 c) {
#ifndef DRS_EXPERIMENT_2MFMA  // (experiment, never shipped: tools/experiment_2mfma.sh drops the w_lo * a_hi product - DESIGN.md section 7)
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.lo, a.hi, c, 0, 0, 0);
#endif
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi, a.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi, a.hi, c, 0, 0, 0);
  }
};

struct MfmaGeom {
  int IH, IW;            // staged input window (rows, cols)
  int dy_min, dx_min;    // smallest tap offsets
  int tiles_x, tiles_y;  // patches per image
  int nchunks;           // ceil(Cin / KC)
  int a_plane;           // bytes of one activation k-group plane (multiple of 256)
  int a_image;           // bytes of one activation image (4 planes)
  int w_image;           // bytes of one weight image in LDS (ntaps*4*BN*16)
  int w_gimage;          // bytes of one weight image in global memory
  int nchunks2;          // K-chunks of the second input (0 = none)
  int iw_magic;          // ceil(65536 / IW): p / IW == (p * iw_magic) >> 16 for p < 1024
  int w2_gimage;         // bytes of one weight image of the second input in global memory
  int debug;             // ablation switches (DRS_DEBUG_FLAGS): 1 skip MFMA phase, 2 skip LDS staging stores, 4 skip global loads, 8 skip epilogue
};



// wave-specialised 3x3 kernel (conv_mfma_ws.hip); `g` is the CONV3X3 / 512-thread geometry of conv_mfma.hip
bool drs_tapconv_ws_supported(const TapConv& d, int impl);
int drs_launch_tapconv_ws(const TapConv& d, const MfmaGeom& g, int impl, hipStream_t s);
// wave-specialised 3x3 kernel over SP-format activations (conv_mfma_sp.hip); eligibility: drs_tapconv_sp_supported (drs_common.h)
int drs_launch_tapconv_sp(const TapConv& d, const MfmaGeom& g, hipStream_t s);
bool drs_tapconv_sp_f32out_supported(const TapConv& d, int impl);  // the same kernel with the fp32 channels-last epilogue
// the same item / step protocol in the FL arithmetic (conv_mfma_fl.hip: fp16 main product + block-scaled fp6 cross terms of
// tap pairs); DRS_FL_DECLINED: the launch prefers the split-bf16 kernel's 32-channel groups (fewer 64-channel items than half the CUs)
#define DRS_FL_DECLINED (-1000)
bool drs_tapconv_fl_supported(const TapConv& d, int impl);
int drs_launch_tapconv_fl(const TapConv& d, const MfmaGeom& g, hipStream_t s);
// the same structure for 8 x 8 images, four images per item (conv_mfma_sp8.hip): the bottleneck level of 64 x 64 models
bool drs_tapconv_sp8_supported(const TapConv& d, int impl);
int drs_launch_tapconv_sp8(const TapConv& d, hipStream_t s);
// 3x3 stride-2 convolution over SP tensors, operands straight from global memory (conv_s2_sp.hip; DRS_S2K=0 disables)
bool drs_conv_s2_sp_supported(const TapConv& d, int impl);
int drs_launch_conv_s2_sp(const TapConv& d, hipStream_t s);
bool drs_down_sp_supported(const TapConv& d, int impl);  // down_sp.hip: the 32- / 64-channel levels with the window staged in LDS
int drs_launch_down_sp(const TapConv& d, hipStream_t s);
// 3x3 stride 1 for the shallow layers: weights resident in LDS, operands straight from global memory (conv3x3_direct_sp.hip)
bool drs_conv3x3_direct_sp_supported(const TapConv& d, int impl);
int drs_launch_conv3x3_direct_sp(const TapConv& d, hipStream_t s);
// fused ConvTranspose2d(3, 2, 1, 1), same structure (conv_s2_sp.hip)
bool drs_convt_sp_supported(const TapConv& d, int impl);
int drs_launch_convt_sp(const TapConv& d, hipStream_t s);
