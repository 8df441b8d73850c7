// Shared host/device declarations of the gfx950 kernels behind include/drs_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/drs_hip.h"

#define DRS_MAX_TAPS 9
#define DRS_TAPMODE_CONVT 2  // TapConv::mode: all 4 phases of ConvTranspose2d(k3,s2,p1,op1) in one launch (MFMA family)

// One fused "tap convolution" over channels-last activations.  Every convolution flavour of the
// UNet (3x3 s1, 3x3 s2, 1x1, 2x2 s2, and each of the 4 output phases of ConvTranspose 3x3 s2) is
// an instance:
//   logical output position (n, ty, tx), ty < TH, tx < TW
//   physical output pixel   (ty*out_scale + out_oy, tx*out_scale + out_ox)
//   tap i reads input pixel (ty*in_stride + dy[i], tx*in_stride + dx[i]), zero outside [0,H)x[0,W)
//   acc[co] = sum_i sum_ci (in[..ci] + in_add[n][ci]) * w[wtap[i]][ci][co]
// Epilogue (in this order, each step optional):
//   v = acc * gate[n][oy/2][ox/2]  ->  v += bias[co]  ->  relu_pre  ->  v += post_add[n][co]
//   -> v += res[n][oy][ox][co]  ->  relu_post  ->  sigmoid  ->  store (NHWC slice or NCHW)
struct TapConv {
  const float* in;
  int in_cs, in_co;  // channel stride of the input buffer, first channel of the slice read
  int N, H, W, Cin;
  const float* w;     // layout depends on the kernel family (see pack kernels)
  const void* w_aux;  // second weight image (e.g. low halves for split-bf16), or null
  const float* bias;  // [Cout] or null
  float* out;
  int out_cs, out_co;
  int OH, OW, Cout;
  int TH, TW;
  int in_stride, out_scale, out_oy, out_ox;
  int ntaps;
  int dy[DRS_MAX_TAPS], dx[DRS_MAX_TAPS], wtap[DRS_MAX_TAPS];
  int wtaps_total;       // number of taps stored in w (stride between taps is Cin*Cout)
  const float* in_add;   // [N][in_add_cs] (already offset to this layer's slice) or null
  int in_add_cs;
  const float* post_add; // [N][post_cs] (already offset to this layer's slice) or null
  int post_cs;
  const float* res;      // NHWC, same spatial size as out, or null
  int res_cs, res_co;
  int res_bstride_zero;  // 1: res has batch 1 and is broadcast over n
  const float* gate;     // [N][OH/2][OW/2] or null
  int relu_pre, relu_post, sigmoid;
  int out_nchw;          // 1: out is (N,Cout,OH,OW) planar
  int mode;              // 0 = plain tap list; DRS_TAPMODE_CONVT = fused transposed convolution (TH,TW = input size)
  // optional second input ("K-concat", MFMA family): a 1x1 stride-1 convolution of in2 (same logical grid as the
  // output positions: pixel (ty, tx), size H2 x W2 = TH x TW) accumulated into the same output before the epilogue:
  //   acc[co] += sum_ci in2[n][ty][tx][ci] * w2[co][ci];  bias2 is added with bias.
  // Used for shortcut_conv -> conv2 (reference ResConvBlock :167-171) and w_g -> w_x (AttentionBlock :101-103).
  const float* in2;
  int in2_cs, in2_co, Cin2, H2, W2;
  const float* w2;
  const float* bias2;
  // optional fused 1x1 projection of the epilogue result (the UNet's `output` conv): fuse_out[n][j][oy][ox] =
  // fuse_b[j] + sum_co v[co] * fuse_w[j][co], j < fuse_dim <= 4; planar NCHW.  MFMA family only, Cout == 32.
  // With fuse_out set, `out` may be null (the wide tensor is then never written).
  const float* fuse_w;
  const float* fuse_b;
  float* fuse_out;
  int fuse_dim;
  // proj = 1: the projection is FOLDED INTO THE WEIGHTS (both maps are linear and nothing sits between them): `w` is a
  // 16-row x 3-column operand image (drs_launch_fold_proj: row 4 ky + o), Cout == 16, fuse_w unused:
  //   fuse_out[n][o][oy][ox] = the convolution (+ fuse_b[o] if set).  conv3x3_proj_sp_kernel (conv3x3_direct_sp.hip) only.
  //   With `gate` set the input is taken as in * nearest2x(gate) (the attention gate's psi: reference :105-106).
  int proj;
  // launch hint: this op runs next to another one on a second stream: one block per CU (80 KB of LDS each, so a block of
  // either kernel fits on every CU at the same time) and no 512-thread variant; 2 = two blocks per CU for the small-LDS
  // 1x1 flavours (37 KB each next to the partner's 80 KB)
  int shared_cu;
  // ---- split-bf16 activation storage ("SP" format; eval plans of DRS_IMPL_MFMA_BF16X3) --------------------------------
  // A tensor in SP format keeps the byte geometry of its fp32 channels-last form (pixel stride = cs * 4 bytes, a group of
  // 32 channels = 128 bytes), but every 32-channel group holds [32 x bf16 hi | 32 x bf16 lo] with x ~ hi + lo (16
  // mantissa bits: exactly the operands the split-bf16 MFMA path multiplies) instead of 32 floats.  A 16-channel tensor
  // (the stem output) is one group of [16 x hi | 16 x lo] = 64 bytes per pixel.  Producers write the operand halves
  // once, consumers stage them with no conversion (register staging, or LDS-DMA straight into the operand planes).
  int in_sp, in2_sp, out_sp, res_sp;  // 1: that tensor is in SP format (the pointers stay `float*`: same byte offsets)
  // optional second output of the epilogue (SP format, channels-last, same spatial size and channel count as `out`):
  //   out2[n][oy][ox][c] = v[c] + post2[n][c]   (v = the value stored to `out`)
  // Used for x + relu(time_mlp(t)) of the next UpConvBlock (reference :199): its 3x3 convolution then needs no input add.
  float* out2;
  int out2_cs, out2_co;
  const float* post2;  // [N][post2_cs] (already offset to the layer's slice)
  int post2_cs;
  const void* zero_line;  // >= 256 bytes of zeros in device memory (source of out-of-image pixels for LDS-DMA staging)
  // fused pair of 3x3 convolutions of the SAME input (ResConvBlock conv1 + its skip convolution, reference :153-166): `w`
  // is one operand image of 2 * Cout channels [main | skip], `bias` holds 2 * Cout values, and the epilogue computes
  //   out[c] = relu(acc[c] + bias[c]) + post_add[c] + acc[Cout + c] + bias[Cout + c].
  // Wave-specialised kernel only (both halves of a pixel end up in one lane: no exchange, no round trip of the skip tensor).
  int dual;
  // device word set to 1 by a wave of the wave-specialised SP kernels whose bounded poll of an LDS counter ran out (a
  // protocol bug: sp_sync.h); null = no report.  Read back by drs_unet_check_faults.
  unsigned* fault;
  // "FL" operand images of this layer's weights (conv_mfma_fl.hip: fp16 main image + block-scaled fp6 cross-term image,
  // derived from the packed split-bf16 images by drs_launch_fl_repack), or null: the layer then runs on the split-bf16
  // kernel.  w2_fl: the same for the second input's 1x1 weights.
  const void* w_fl;
  const void* w2_fl;
};

struct DrsErr {
  static void set(const char* fmt, ...);
};

#define DRS_CHECK_HIP(expr)                                                             \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      DrsErr::set("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return DRS_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

#define DRS_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) {                   \
      DrsErr::set(__VA_ARGS__);      \
      return (code);                 \
    }                                \
  } while (0)

static inline int drs_cdiv(int a, int b) { return (a + b - 1) / b; }
// ReLU and the other activation maxima: IEEE 754-2019 `maximum` (v_maximum3_f32), which PROPAGATES a NaN like torch.relu does.
// fmaxf (v_max_f32) returns the other operand: a NaN accumulator would leave a ReLU as 0 and the divergence it signals
// would be gone from the output (tests/test_gpu_parity.py: test_nan_reaches_the_output).
__device__ __forceinline__ float drs_maxf(float a, float b) { return __builtin_elementwise_maximum(a, b); }
#define DRS_RED_BLOCKS 512  // most blocks a partial-sum reduction of the training kernels uses (size of the partials buffer)

// Every kernel of the library is launched through DRS_LAUNCH (same arguments as hipLaunchKernelGGL).  While a plan runs a
// PROFILED forward (drs_unet_profile_enable) on this host thread, the launch is also appended to that plan's launch log
// (kernel name as the runtime reports it + the op of the schedule that issued it: drs_unet_profile_launch), which is what
// ties the per-dispatch rows of a rocprofv3 counter pass to the plan's ops (tools/collect_pmc.py) without a hand-kept list
// of kernel names.  Outside a profiled forward the hook is one thread-local load.
void drs_note_launch(const void* kernel_fn, const char* expr);
#define DRS_LAUNCH(kern, grid, block, lds, stream, ...)                       \
  do {                                                                        \
    drs_note_launch(reinterpret_cast<const void*>(kern), #kern);              \
    hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);          \
  } while (0)

// Per-DEVICE launch facts: sets the kernel's dynamic-LDS limit once per (current device, kernel) and returns that device's
// CU count.  Keyed by device, mutex-protected: correct with several devices in one process and from several host threads.
int drs_kernel_prepare(const void* kernel, int max_dynamic_lds, int* num_cu);

// ---- split bf16 of TWO fp32 values: hi = bf16(x) (round to nearest even), lo = bf16(x - hi), packed [first | second << 16] ----
// One v_cvt_pk_bf16_f32 per pair and part and one v_pk_add_f32 for the pair's remainders; element-wise `(__bf16)x`
// conversions compile to one conversion per VALUE plus a merge per dword (twice the vector instructions).
#ifdef __HIPCC__
typedef float drs_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 drs_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 drs_split2(float a, float b) {  // .x = hi pair, .y = lo pair
  const drs_f32x2 x = {a, b};
  const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, drs_bf16x2));
  const drs_f32x2 hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
  return uint2{hi, __builtin_bit_cast(unsigned, __builtin_convertvector(x - hf, drs_bf16x2))};
}
#endif

// ---- kernel launchers (each returns a DRS_* status) ------------------------------------------
int drs_launch_tapconv_direct(const TapConv& d, hipStream_t s);
int drs_launch_tapconv_mfma(const TapConv& d, int impl, hipStream_t s);
bool drs_tapconv_mfma_supported(const TapConv& d, int impl);
bool drs_tapconv_ws_supported(const TapConv& d, int impl);  // wave-specialised 3x3 kernel (conv_mfma_ws.hip) takes this op

// weight packing: src is torch layout (Cout,Cin,KH,KW) or, transposed, (Cin,Cout,KH,KW).
// bn = {gamma,beta,running_mean,running_var} or all null.  dst_w layout:
//   DIRECT:  [tap][Cin][Cout]      MFMA: [tap][Cout][Cin]
int drs_launch_pack_conv(const float* w, const float* b, const float* gamma, const float* beta, const float* rmean,
                         const float* rvar, float eps, float* dst_w, float* dst_b, int Cout, int Cin, int taps,
                         int transposed, int mfma_layout, hipStream_t s);

// MFMA-family packing: dst image(s) [chunk][tap][kgroup][Cout][slot]; see conv_mfma.hip
size_t drs_pack_conv_mfma_bytes(int Cout, int Cin, int taps, int impl);
// Packing queue of the calling host thread: between begin and flush, drs_launch_pack_conv_mfma only records its job; the flush
// runs the recorded jobs as a few batched launches (a plan re-packs ~40 layers after every optimizer step, and as many
// data-gradient images per backward: one 4 - 8 us launch each before round 4).
// several small fp32 device-to-device copies in one launch (small_kernels.hip)
#define DRS_COPY_BATCH 64
struct DrsCopyJob { const float* src; float* dst; long long words; };
struct DrsCopyBatch { DrsCopyJob job[DRS_COPY_BATCH]; };
int drs_launch_gather_copy(const DrsCopyJob* jobs, int n, hipStream_t s);
void drs_pack_queue_begin();
int drs_pack_queue_flush(hipStream_t s);
void drs_pack_queue_abandon();  // drop whatever is recorded and close the queue (error paths)
struct DrsPackQueueScope {  // opens the queue; an early return abandons it, the regular path calls flush()
  DrsPackQueueScope() { drs_pack_queue_begin(); }
  ~DrsPackQueueScope() { drs_pack_queue_abandon(); }
  int flush(hipStream_t s) { return drs_pack_queue_flush(s); }
};
int drs_launch_pack_conv_mfma(const float* w, const float* b, const float* gamma, const float* beta, const float* rmean,
                              const float* rvar, float eps, void* dst_w, float* dst_b, int Cout, int Cin, int taps,
                              int transposed, int impl, hipStream_t s, int cout_src = 0, int flip_taps = 0, int co_off = 0,
                              int partial = 0, int perm = 0, int cin_total = 0, int cin_off = 0);

// FL operand images (conv_mfma_fl.hip) of a layer from its packed split-bf16 images (hi image, then lo image); `flag`: device
// word, bit 0 is set when a folded weight lies outside what fp16 holds (|w| > 60000, or a layer whose largest weight is under 2^-10)
size_t drs_fl_image_bytes(int Cout, int Cin, int taps);
int drs_launch_fl_repack(const void* sp_images, void* dst, int Cout, int Cin, int taps, unsigned* flag, hipStream_t s);

int drs_launch_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int dst_cs, int dst_co,
                            hipStream_t s);
int drs_launch_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, int src_cs, int src_co,
                            hipStream_t s);

int drs_launch_sp_to_nchw(const float* src, float* dst, int N, int C, int H, int W, int src_cs, int src_co, hipStream_t s);

// fused attention gate of a decoder stage over SP-format activations (attn_gate_sp.hip)
struct AttnGateDesc {
  const float* x;    int x_cs, x_co;      // stage input (SP), Cc channels, N x LH x LW
  const float* xres; int r_cs, r_co;      // skip tensor (SP), Ch channels, N x 2LH x 2LW
  float* out;        int out_cs, out_co;  // concat buffer (SP): channels [out_co, out_co + Ch) at N x 2LH x 2LW
  float* psi_out;                         // optional (N, LH, LW) fp32 copy of psi, or null
  int N, LH, LW, Cc, Ch;
  // packed operand images (conv_mfma.hip: [chunk][tap][k-group][Ch][8 x bf16], hi image then lo image, output rows in SP
  // permutation), biases in logical channel order
  const void* w_gate; const float* b_gate;
  // optional per-image gating bias (N x Ch): the stage input given in `x` is x + relu(time_mlp(t)) (the only copy its
  // producer stores), and the row vector the gating convolution does not want is taken out through its bias:
  // b_gate_img[n][co] = b_gate[co] - sum_ci Wg[co][ci] * temb[n][ci]   (drs_launch_gate_bias)
  const float* b_gate_img;
  const void* w_wg;   const float* b_wg;
  const void* w_wx;   const float* b_wx;   // 4 taps (ky*2 + kx)
  const float* w_psi; const float* b_psi;  // Ch floats + 1
  const void* w_res;  const float* b_res;
};
bool drs_attn_gate_supported(int Cc, int Ch);
// out[n][co] = b[co] - sum_ci w[ci][co] * vec[n * vec_stride + ci]   (w: fp32 [Cc][Ch], BatchNorm folded)
int drs_launch_gate_bias(const float* w, const float* b, const float* vec, int vec_stride, float* out, int N, int Cc, int Ch,
                         hipStream_t s);
int drs_launch_attn_gate(const AttnGateDesc& d, hipStream_t s);

// ---- ups.i.transform composed with the x-half of up_convs.i (upfuse_sp.hip) -------------------------------------------------
// y = up_conv(cat[ConvTranspose(h), att]) has no non-linearity between the transposed convolution and the 3x3 convolution
// (reference UpConvBlock.forward returns self.transform(x), UNet_model_superres.py:206-207; up_convs[i] is a bare Conv2d,
// :320-322,376-377), so the x-path is ONE stride-2 transposed convolution of h with per-axis 3 input taps at even outputs
// and 2 at odd ones: 6.25 taps x Cc per output pixel instead of 2.25 x Cc (ConvT, Cc outputs) + 9 x Cc (x-half of up_conv).
//   out[n][2my+py][2mx+px][co] = sum_{ty in T(py)} sum_{tx in T(px)} sum_ci U[py][px][ty][tx][co][ci] * h[n][my+ty-1][mx+tx-1][ci]
//                                + bias[co] + res[n][oy][ox][co] + edge terms,      T(0) = {0,1,2}, T(1) = {1,2}
// `res` is the att-half of up_convs.i (a plain 3x3 convolution of the attention output, computed by the existing kernels),
// the edge terms (eh / ev, fp32) undo the composite's paths through the transposed convolution's cropped row / column -1
// and carry the position-dependent part of the folded ConvTranspose bias (upfuse_edges_kernel).
struct UpFuseDesc {
  const float* in; int in_cs, in_co;   // h = ups.i.conv output (SP), N x LH x LW x Cc
  int N, LH, LW, Cc, Ch;               // Cc input channels, Ch = output channels of up_convs.i
  const void* w;                       // composite operand image: [Ch/32][Cc/32][group 5][image 2][tap 5][k-group 4][32][8 x bf16]
  const float* bias;                   // [Ch]: up_convs bias + the ConvTranspose bias through all nine taps
  const float* res; int res_cs, res_co;  // att-half partial sums (SP), N x 2LH x 2LW x Ch, or null
  const float* eh;                     // [N][2: top, bottom][2LW][Ch] fp32, or null (then ev is null too)
  const float* ev;                     // [N][2: left, right][2LH][Ch] fp32 (zero in rows 0 and 2LH-1)
  float* out; int out_cs, out_co;      // SP result, N x 2LH x 2LW x Ch, or null (fused projection)
  float* out2; int out2_cs, out2_co;   // optional second SP output: out + post2[n][c]
  const float* post2; int post2_cs;
  const float* fuse_w; const float* fuse_b; float* fuse_out; int fuse_dim;  // Ch == 32: fp32 NCHW projection (the UNet's `output`)
  // fuse_acc: fuse_out already holds the PROJECTED att-half (the projection is linear: fuse_w (comp + att_half + b) =
  // fuse_w comp + fuse_w att_half + ...), written by the att-half convolution's own fused-projection epilogue: this kernel
  // adds its part.  12.6 MB instead of the 134 MB of 32-channel partial sums per forward at 256 x 256; `res` is null then.
  int fuse_acc;
  // proj = 1 (drs_launch_upfuse_proj, upfuse_proj_sp.hip): the projection is folded into the composite weights.  bias / eh / ev
  // are those of a 32-channel layer whose logical channel 8 o is output o, everything else zero (drs_launch_upfuse_fold_proj
  // feeds the unchanged pack / edge kernels), `w` is drs_launch_upfuse_proj_pack's image; fuse_w / fuse_b are unused (the output
  // bias is part of `bias`).
  int proj;
  const void* zero_line;
  unsigned* fault;
};
// (v_w', v_b') of that layer: v_w'[8 j][c][kv] = sum_co fw[j][co] v_w[co][c][kv] for c < Cc (the x-half of up_convs.i),
// v_b'[8 j] = fb[j] + sum_co fw[j][co] v_b[co]; dst_w: 32 x (Cc + Ch) x 9 floats, dst_b: 32 floats
// the folded top stage as a streaming direct-operand kernel, kernel rows in the MFMA's M dimension (upfuse_proj_sp.hip)
bool drs_upfuse_proj_supported(int Cc, int Ch, int fuse_dim);
size_t drs_upfuse_proj_weight_bytes(int Cc);
int drs_launch_upfuse_proj_pack(const float* vp, const float* t_w, int Cc, int Ch, int fuse_dim, void* dst, hipStream_t s);
int drs_launch_upfuse_proj(const UpFuseDesc& d, hipStream_t s);
int drs_launch_upfuse_fold_proj(const float* v_w, const float* v_b, const float* fw, const float* fb, int fuse_dim, int Cc, int Ch,
                                float* dst_w, float* dst_b, hipStream_t s);

// The first encoder block (16 -> 32 -> 32 channels) as one launch (resblock0_sp.hip): h = relu(conv1(x)) + temb + skip(x)
// lives in LDS only, out = relu(conv2(h) + shortcut(x)).  All convolutions BatchNorm-folded, operands in the packed
// layouts of drs_launch_pack_conv_mfma (w1: the 64-channel conv1 | skip pair image of the direct kernel's pair flavour).
struct ResBlock0Desc {
  const float* x;                  // block input (SP, 16 channels: 64 bytes per pixel), N x H x W
  const void* w1; const float* b1; // conv1 | skip: [image 2][tap 9][k-group 4][64] slots, biases [64]
  const float* temb; int temb_cs;  // relu(time_mlp(t)) rows [N][32], row stride
  const void* w2; const float* b2; // conv2: [image 2][tap 9][k-group 4][32], bias [32]
  const void* ws; const float* bs; // 1x1 shortcut: [image 2][k-group 4][32], bias [32]
  float* out;                      // block output (SP, 32 channels), N x H x W
  int N, H, W;
  const void* zero_line;           // >= 64 readable zero bytes
  unsigned* fault;                 // protocol watchdog word (TapConv::fault), or null
};
bool drs_resblock0_supported(int Cin, int Cout, int H, int W);
int drs_launch_resblock0(const ResBlock0Desc& d, hipStream_t s);
// edge-term / bias preparation weights (fp32): rt [5][Cc][Ch], rl [6][Cc][Ch], bt [9][Ch]
struct UpFuseEdgeDesc {
  const float* in; int in_cs, in_co; int N, LH, LW, Cc, Ch;
  const float* rt; const float* rl; const float* bt;
  float* eh; float* ev;
  const void* wimg;       // MFMA operand image of rt / rl (drs_upfuse_edge_image_bytes)
  const void* zero_line;
};
bool drs_upfuse_supported(int Cc, int Ch, int LH, int LW);
size_t drs_upfuse_weight_bytes(int Cc, int Ch);       // composite operand image
size_t drs_upfuse_aux_floats(int Cc, int Ch);         // rt | rl | bt | bias, in this order
// v_w: up_convs.i.weight (Ch, Cc + Ch, 3, 3); v_b: its bias; t_w: ups.i.transform.weight (Cc, Cc, 3, 3); t_b: its bias
size_t drs_upfuse_edge_image_bytes(int Cc, int Ch);  // rt | rl as MFMA operands (edge kernel)
int drs_launch_upfuse_pack(const float* v_w, const float* v_b, const float* t_w, const float* t_b, int Cc, int Ch, void* dst_w,
                           float* dst_aux, void* dst_edge, hipStream_t s);
int drs_launch_upfuse_edges(const UpFuseEdgeDesc& d, hipStream_t s);
int drs_launch_upfuse(const UpFuseDesc& d, hipStream_t s);
int drs_launch_nchw_to_sp(const float* src, float* dst, int N, int C, int H, int W, hipStream_t s);

int drs_launch_sp_add_rowvec(const float* src, float* dst, const float* vec, int vec_stride, int N, long long pix_per_image,
                             int C, hipStream_t s);
bool drs_tapconv_sp8_supported(const TapConv& d, int impl);  // its 8 x 8-image instance (conv_mfma_sp8.hip)
bool drs_tapconv_sp_supported(const TapConv& d, int impl);
// 3x3 stride 1 for the shallow layers (conv3x3_direct_sp.hip); with TapConv::proj the folded-projection flavour
int drs_launch_conv3x3_direct_sp(const TapConv& d, hipStream_t s);
// weights of a 3x3 convolution (Cmid outputs; input channels [cin_off, cin_off + Cin) of cin_total) followed by a 1x1
// projection fw[fuse_dim][Cmid], as ONE 3x3 convolution: dst[16][Cin][3] fp32, row 4 ky + o = kernel row ky of output o, the
// three kernel columns as the "taps" of a 16-channel layer (conv3x3_proj_sp_kernel)
int drs_launch_fold_proj(const float* w, int cin_total, int cin_off, int Cmid, int Cin, const float* fw, int fuse_dim, float* dst,
                         hipStream_t s);
bool drs_conv3x3_direct_sp_proj_supported(const TapConv& d, int impl);
// ... composed with the attention block's `result` convolution in front of it (1x1, C -> C, BatchNorm folded): dst2 in dst1's
// layout, tab[3][3][4] = what the constant part of `result` gives an output pixel of each (row, column) border class
int drs_launch_fold_result(const float* dst1, int C, const float* wr, const float* br, const float* gamma, const float* beta,
                           const float* rmean, const float* rvar, float eps, float* dst2, float* tab, hipStream_t s);
  // wave-specialised SP-format 3x3 kernel (conv_mfma_sp.hip) takes this op

// planar (NCHW) small-channel kernels
int drs_launch_conv3x3_planar(const float* in, const float* w, const float* b, const float* res, float* out, int N,
                              int Cin, int Cout, int H, int W, int relu, hipStream_t s);
int drs_launch_stem(const float* in_nchw, const float* w, const float* b, const float* res_nhwc, int res_batch,
                    float* out_nhwc, int N, int Cin, int Cout, int H, int W, hipStream_t s, int out_sp = 0);
int drs_launch_bicubic(const float* x, float* y, int N, int C, int H, int W, int scale, hipStream_t s);
int drs_launch_time_mlp(const int64_t* t, const float* inv_freq, const float* W1, const float* b1, const float* W2,
                        const float* b2, float* out, int out_stride, int B, int dim_in, int dim_out, hipStream_t s);
int drs_launch_time_mlp_multi(const int64_t* t, const float* inv_freq, const char* packed, const long long* table,
                              int nmlp, int max_dim, float* out, int out_stride, int B, int dim_in,
                              const float* label_emb, const long long* labels, int label_batch, int num_classes,
                              hipStream_t s);

// train-mode BatchNorm (bn_train.hip): statistics of Z, running-stat update, normalise + the block's epilogue
int drs_launch_bn_train(const float* z, int z_cs, int z_co, long long npix, long long pix_per_image, int C,
                        const float* gamma, const float* beta, float* running_mean, float* running_var, float eps,
                        float momentum, double* sums_scratch, float* mean, float* rstd, const float* post_add,
                        int post_cs, const float* res, int res_cs, int res_co, float* out, int out_cs, int out_co,
                        int relu_pre, int relu_post, hipStream_t s);

// ---- training backward (train_kernels.hip) ------------------------------------------------------------------------
struct WgradDesc {
  const float* A; int a_cs, a_co, Ca, AH, AW, sa;
  const float* B; int b_cs, b_co, Cb, BH, BW, sb;
  int N, TH, TW, ntaps;
  int ay[DRS_MAX_TAPS], ax[DRS_MAX_TAPS], by[DRS_MAX_TAPS], bx[DRS_MAX_TAPS];
  const float* a_add; int a_add_cs;
  const float* a_gate;
  float* dW; int T_total; int wtap[DRS_MAX_TAPS]; int out_transposed;
  float* partial; size_t partial_bytes;  // workspace of the MFMA path (null: direct fp32 kernel)
  float* dbias;  // convolution form only: also accumulate sum over positions of B[., b] (the bias gradient)
};
int drs_launch_wgrad(const WgradDesc& d, hipStream_t s);
bool drs_wgrad_mfma_supported(const WgradDesc& d);
int drs_launch_wgrad_mfma(const WgradDesc& d, float* partial, size_t partial_bytes, hipStream_t s);
// the same on the bf16 matrix pipe, operands split hi + lo (wgrad_mfma_bf16.hip); DRS_TRAIN_WGRAD_IMPL=mfma_f32 disables it
bool drs_wgrad_mfma_bf16_supported(const WgradDesc& d);
int drs_launch_wgrad_mfma_bf16(const WgradDesc& d, float* partial, size_t partial_bytes, hipStream_t s);
// `partials`: optional DRS_RED_BLOCKS x C floats of scratch (stream-ordered use): whole-tensor sums then run without atomics
int drs_launch_colsum(const float* t, int cs, int co, int C, long long npix, long long pix_per_image, int per_image,
                      int out_stride, float* out, hipStream_t s, float* partials = nullptr);
int drs_launch_relu_mask(float* g, int g_cs, int g_co, const float* y, int y_cs, int y_co, int C, long long npix,
                         hipStream_t s);
int drs_launch_bn_bwd(const float* g, int g_cs, int g_co, float* z, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, int relu_pre, int C, long long npix, double* partials,
                      double* sums, float* dgamma, float* dbeta, hipStream_t s, float* z_sp = nullptr,
                      const float* mask_y = nullptr, int my_cs = 0, int my_co = 0);
int drs_launch_gate_bwd(const float* x, const float* E, const float* psi, float* dx, float* dpsi_pre, int N, int LH,
                        int LW, int C, hipStream_t s);
int drs_launch_psi_bwd(const float* Pm, const float* wpsi, const float* dpsi_pre, float* dP, float* dw, float* db, int C,
                       long long npix, float* partials, hipStream_t s);
// one entry per time MLP of the network (at most 8): parameters, the MLP's slice of the embedding / its gradient, gradient outputs
struct DrsMlpBwd { const float *W1, *b1, *W2, *temb, *dtemb; float *dW1, *db1, *dW2, *db2; int dim; };
struct DrsMlpBwdTable { DrsMlpBwd m[8]; int n; };
int drs_launch_time_mlp_bwd(const long long* t, const float* inv_freq, const DrsMlpBwdTable& tab, int stride, int B,
                            const float* label_emb, const long long* labels, int label_batch, int num_classes, float* dlabel,
                            hipStream_t s);
// weight + bias gradient of a 3x3 stem convolution (CI <= 4 -> 16 channels); partials: >= 512 x (144 CI + 16) floats, stream-ordered
int drs_launch_stem_wgrad(const float* g, int g_cs, const float* x, int N, int CI, int H, int W, float* partials,
                          size_t partial_bytes, float* dW, float* db, hipStream_t s);
int drs_launch_stem_dgrad(const float* g, int g_cs, const float* w, float* dx, int N, int H, int W, int C, hipStream_t s);
int drs_launch_bicubic_bwd(const float* dy, float* dx, int N, int C, int H, int W, int scale, hipStream_t s);
// whole backward of one few-channel 3x3 layer (CC -> CC, CC <= 4): dW / db accumulate, gin (+)= conv^T(gout) [* (mask_y > 0)];
// partials: >= 1024 x (9 CC^2 + CC) floats of stream-ordered scratch
int drs_launch_small_conv_bwd(const float* in, const float* gout, const float* w, float* gin, int accumulate,
                              const float* mask_y, int N, int CC, int H, int W, float* partials, float* dW, float* db,
                              hipStream_t s);
