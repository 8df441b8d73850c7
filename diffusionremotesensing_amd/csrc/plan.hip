// C-ABI of include/drs_hip.h: error state, the operator-level convolution entry and the
// whole-UNet plan (weight packing + the launch schedule of one eval forward).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cxxabi.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "drs_common.h"

// ------------------------------------------------------------------------------------------------
// error state
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void DrsErr::set(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* drs_last_error(void) { return g_err; }
extern "C" int drs_abi_version(void) { return 7; }

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

int drs_kernel_prepare(const void* kernel, int max_dynamic_lds, int* num_cu) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, bool> attr_set;  // (device, kernel) -> dynamic-LDS attribute applied
  static std::map<int, int> cus;                                  // device -> CU count
  int dev = 0;
  DRS_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto it = cus.find(dev);
  if (it == cus.end()) {
    int n = 0;
    DRS_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    it = cus.emplace(dev, n).first;
  }
  *num_cu = it->second;
#ifdef DRS_X_NUM_CU  // experiment (tools/two_stream_probe.py): persistent kernels size their grids for DRS_X_NUM_CU compute units
  {
    static const int lim = getenv("DRS_X_NUM_CU") ? atoi(getenv("DRS_X_NUM_CU")) : 0;
    if (lim > 0 && lim < *num_cu) *num_cu = lim;
  }
#endif
  bool& done = attr_set[std::make_pair(dev, kernel)];
  if (!done && max_dynamic_lds > 0) {
    DRS_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, max_dynamic_lds));
    done = true;
  }
  return DRS_OK;
}

// ------------------------------------------------------------------------------------------------
// TapConv builders
// ------------------------------------------------------------------------------------------------
static TapConv conv_desc(const float* in, int N, int H, int W, int Cin, int in_cs, int in_co, const float* w,
                         const float* bias, float* out, int Cout, int out_cs, int out_co, int KH, int KW, int stride,
                         int pad) {
  TapConv d = {};
  d.in = in; d.in_cs = in_cs; d.in_co = in_co;
  d.N = N; d.H = H; d.W = W; d.Cin = Cin;
  d.w = w; d.bias = bias;
  d.out = out; d.out_cs = out_cs; d.out_co = out_co;
  d.OH = (H + 2 * pad - KH) / stride + 1;
  d.OW = (W + 2 * pad - KW) / stride + 1;
  d.Cout = Cout;
  d.TH = d.OH; d.TW = d.OW;
  d.in_stride = stride; d.out_scale = 1; d.out_oy = 0; d.out_ox = 0;
  d.ntaps = KH * KW;
  d.wtaps_total = KH * KW;
  for (int ky = 0; ky < KH; ++ky)
    for (int kx = 0; kx < KW; ++kx) {
      const int i = ky * KW + kx;
      d.dy[i] = ky - pad; d.dx[i] = kx - pad; d.wtap[i] = i;
    }
  return d;
}

// Phase (py,px) of ConvTranspose2d(k=3, s=2, p=1, output_padding=1): out[2*iy - 1 + ky] += in[iy] * w[ky]
// (reference UpConvBlock.transform, UNet_model_superres.py:185).  Even output rows take ky=1 from iy=t; odd rows
// take ky=0 from iy=t+1 and ky=2 from iy=t.  Output is (2H, 2W).
static TapConv convT_phase_desc(const float* in, int N, int H, int W, int Cin, int in_cs, int in_co, const float* w,
                                const float* bias, float* out, int Cout, int out_cs, int out_co, int py, int px) {
  TapConv d = {};
  d.in = in; d.in_cs = in_cs; d.in_co = in_co;
  d.N = N; d.H = H; d.W = W; d.Cin = Cin;
  d.w = w; d.bias = bias;
  d.out = out; d.out_cs = out_cs; d.out_co = out_co;
  d.OH = 2 * H; d.OW = 2 * W; d.Cout = Cout;
  d.TH = H; d.TW = W;
  d.in_stride = 1; d.out_scale = 2; d.out_oy = py; d.out_ox = px;
  d.wtaps_total = 9;
  int ydy[2], yk[2], ny, xdx[2], xk[2], nx;
  if (py == 0) { ny = 1; ydy[0] = 0; yk[0] = 1; } else { ny = 2; ydy[0] = 1; yk[0] = 0; ydy[1] = 0; yk[1] = 2; }
  if (px == 0) { nx = 1; xdx[0] = 0; xk[0] = 1; } else { nx = 2; xdx[0] = 1; xk[0] = 0; xdx[1] = 0; xk[1] = 2; }
  int i = 0;
  for (int a = 0; a < ny; ++a)
    for (int b = 0; b < nx; ++b, ++i) {
      d.dy[i] = ydy[a]; d.dx[i] = xdx[b]; d.wtap[i] = yk[a] * 3 + xk[b];
    }
  d.ntaps = i;
  return d;
}

// All four phases in one MFMA launch: TH x TW = input size, out_scale 2, the 9 weight taps in storage order.
static TapConv convT_fused_desc(const float* in, int N, int H, int W, int Cin, int in_cs, int in_co, const float* w,
                                const float* bias, float* out, int Cout, int out_cs, int out_co) {
  TapConv d = convT_phase_desc(in, N, H, W, Cin, in_cs, in_co, w, bias, out, Cout, out_cs, out_co, 1, 1);
  d.mode = DRS_TAPMODE_CONVT;
  d.out_oy = 0; d.out_ox = 0;
  d.ntaps = 9;
  for (int i = 0; i < 9; ++i) { d.dy[i] = 0; d.dx[i] = 0; d.wtap[i] = i; }
  return d;
}

// impl is the family the weights of this layer were packed for: no silent switch at launch time
static int run_conv(const TapConv& d, int impl, hipStream_t s) {
  if (impl != DRS_IMPL_DIRECT) return drs_launch_tapconv_mfma(d, impl, s);
  return drs_launch_tapconv_direct(d, s);
}
// algorithmic work of one tap-convolution (SURVEY.md 8(d) model: 2*MACs; fp32 input + output + weights)
static double conv_flops(const TapConv& d) {
  return 2.0 * d.N * d.TH * d.TW * (double)d.Cout * ((double)d.Cin * d.ntaps + (d.in2 ? d.Cin2 : 0));
}
static double conv_bytes(const TapConv& d, bool count_out_once = true) {
  const double in = (double)d.N * d.H * d.W * d.Cin;
  const double out = (double)d.N * d.TH * d.TW * d.Cout * (d.mode == DRS_TAPMODE_CONVT ? 4 : 1);
  (void)count_out_once;
  const double in2 = d.in2 ? (double)d.N * d.H2 * d.W2 * d.Cin2 + (double)d.Cin2 * d.Cout : 0.0;
  return 4.0 * (in + in2 + out + (double)d.ntaps * d.Cin * d.Cout);
}

// ------------------------------------------------------------------------------------------------
// operator-level convolution (NCHW boundary)
// ------------------------------------------------------------------------------------------------
static bool conv_flavour_ok(int KH, int KW, int stride, int pad, int transposed, int out_pad) {
  if (transposed) return KH == 3 && KW == 3 && stride == 2 && pad == 1 && out_pad == 1;
  if (KH == 3 && KW == 3 && pad == 1 && (stride == 1 || stride == 2)) return true;
  if (KH == 1 && KW == 1 && pad == 0 && stride == 1) return true;
  if (KH == 2 && KW == 2 && pad == 0 && stride == 2) return true;
  return false;
}
static void conv_out_hw(int H, int W, int KH, int KW, int stride, int pad, int transposed, int out_pad, int* OH,
                        int* OW) {
  if (transposed) {
    *OH = (H - 1) * stride - 2 * pad + KH + out_pad;
    *OW = (W - 1) * stride - 2 * pad + KW + out_pad;
  } else {
    *OH = (H + 2 * pad - KH) / stride + 1;
    *OW = (W + 2 * pad - KW) / stride + 1;
  }
}

extern "C" size_t drs_conv2d_workspace_bytes(int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                                             int transposed, int out_pad) {
  int OH, OW;
  conv_out_hw(H, W, KH, KW, stride, pad, transposed, out_pad, &OH, &OW);
  size_t b = 0;
  b += align_up((size_t)N * H * W * Cin * 4);
  b += align_up((size_t)N * OH * OW * Cout * 4);
  b += align_up(drs_pack_conv_mfma_bytes(Cout, Cin, KH * KW, DRS_IMPL_MFMA_BF16X3) + (size_t)Cout * Cin * KH * KW * 4);
  b += align_up((size_t)Cout * 4);
  return b + 256;
}

extern "C" int drs_conv2d_nchw(const float* x, const float* w, const float* b, float* y, int N, int Cin, int H, int W,
                               int Cout, int KH, int KW, int stride, int pad, int transposed, int out_pad, int relu,
                               void* workspace, size_t workspace_bytes, int impl, drs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) return DRS_OK;  // empty batch: nothing to do (torch hands out null pointers for empty tensors)
  DRS_REQUIRE(x && w && y && workspace, DRS_ERR_ARG, "conv2d: null pointer");
  DRS_REQUIRE(N >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, DRS_ERR_SHAPE, "conv2d: bad dims");
  DRS_REQUIRE(conv_flavour_ok(KH, KW, stride, pad, transposed, out_pad), DRS_ERR_SHAPE,
              "conv2d: unsupported flavour k=%dx%d s=%d p=%d transposed=%d out_pad=%d", KH, KW, stride, pad, transposed,
              out_pad);
  DRS_REQUIRE(impl >= DRS_IMPL_DIRECT && impl <= DRS_IMPL_MFMA_F16, DRS_ERR_ARG, "conv2d: impl=%d", impl);
  DRS_REQUIRE(workspace_bytes >= drs_conv2d_workspace_bytes(N, Cin, H, W, Cout, KH, KW, stride, pad, transposed, out_pad),
              DRS_ERR_WORKSPACE, "conv2d: workspace too small");
  if (N == 0) return DRS_OK;
  int OH, OW;
  conv_out_hw(H, W, KH, KW, stride, pad, transposed, out_pad, &OH, &OW);
  DRS_REQUIRE(OH > 0 && OW > 0, DRS_ERR_SHAPE, "conv2d: empty output");
  char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  float* xin = (float*)base; base += align_up((size_t)N * H * W * Cin * 4);
  float* yout = (float*)base; base += align_up((size_t)N * OH * OW * Cout * 4);
  float* pw = (float*)base;
  base += align_up(drs_pack_conv_mfma_bytes(Cout, Cin, KH * KW, DRS_IMPL_MFMA_BF16X3) + (size_t)Cout * Cin * KH * KW * 4);
  float* pb = (float*)base;
  int rc;
  if ((rc = drs_launch_nchw_to_nhwc(x, xin, N, Cin, H, W, Cin, 0, s))) return rc;

  // decide the kernel family on a probe descriptor, then pack in that family's layout
  TapConv probe = transposed ? convT_fused_desc(xin, N, H, W, Cin, Cin, 0, pw, pb, yout, Cout, Cout, 0)
                             : conv_desc(xin, N, H, W, Cin, Cin, 0, pw, pb, yout, Cout, Cout, 0, KH, KW, stride, pad);
  const bool mfma = impl != DRS_IMPL_DIRECT && drs_tapconv_mfma_supported(probe, impl);
  if (mfma)
    rc = drs_launch_pack_conv_mfma(w, b, nullptr, nullptr, nullptr, nullptr, 0.f, pw, pb, Cout, Cin, KH * KW, transposed,
                                   impl, s);
  else
    rc = drs_launch_pack_conv(w, b, nullptr, nullptr, nullptr, nullptr, 0.f, pw, pb, Cout, Cin, KH * KW, transposed, 0, s);
  if (rc) return rc;
  const int use_impl = mfma ? impl : DRS_IMPL_DIRECT;
  if (!transposed) {
    TapConv d = probe;
    d.relu_pre = relu;
    if ((rc = run_conv(d, use_impl, s))) return rc;
  } else if (mfma) {
    TapConv d = convT_fused_desc(xin, N, H, W, Cin, Cin, 0, pw, pb, yout, Cout, Cout, 0);
    d.relu_pre = relu;
    if ((rc = run_conv(d, use_impl, s))) return rc;
  } else {
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        TapConv d = convT_phase_desc(xin, N, H, W, Cin, Cin, 0, pw, pb, yout, Cout, Cout, 0, py, px);
        d.relu_pre = relu;
        if ((rc = run_conv(d, use_impl, s))) return rc;
      }
  }
  return drs_launch_nhwc_to_nchw(yout, y, N, Cout, OH, OW, Cout, 0, s);
}

// ------------------------------------------------------------------------------------------------
// operator-level fused up-sampling stage (NCHW boundary): y = conv3x3(cat[conv_transpose(h), att])
// ------------------------------------------------------------------------------------------------
static size_t upfused_sizes(int N, int Cc, int Ch, int LH, int LW, size_t* o) {
  // o[0] h (SP), o[1] att (SP), o[2] att-half partial sums, o[3] result (SP), o[4] composite image, o[5] aux, o[6] att-half
  // weights, o[7] att-half bias, o[8] eh, o[9] ev, o[10] zero line + fault word, o[11] edge operand image, o[12..14] fold scratch
  const size_t hi = (size_t)N * 4 * LH * LW;
  size_t b = 0;
  o[0] = b; b += align_up((size_t)N * LH * LW * Cc * 4);
  o[1] = b; b += align_up(hi * Ch * 4);
  o[2] = b; b += align_up(hi * Ch * 4);
  o[3] = b; b += align_up(hi * Ch * 4);
  o[4] = b; b += align_up(drs_upfuse_weight_bytes(Cc, Ch));
  o[5] = b; b += align_up(drs_upfuse_aux_floats(Cc, Ch) * 4);
  o[6] = b; b += align_up(drs_pack_conv_mfma_bytes(Ch, Ch, 9, DRS_IMPL_MFMA_BF16X3));
  o[7] = b; b += align_up((size_t)Ch * 4);
  o[8] = b; b += align_up((size_t)N * 2 * 2 * LW * Ch * 4);
  o[9] = b; b += align_up((size_t)N * 2 * 2 * LH * Ch * 4);
  o[10] = b; b += 512;
  o[11] = b; b += align_up(drs_upfuse_edge_image_bytes(Cc, Ch));
  // folded output projection (fuse_w given, shapes the direct kernel takes): the two fp32 contractions the images are packed from
  o[12] = b; b += align_up((size_t)16 * Ch * 9 * 4);
  o[13] = b; b += align_up((size_t)32 * (Cc + Ch) * 9 * 4);
  o[14] = b; b += align_up((size_t)32 * 4);
  o[15] = b; b += align_up(drs_upfuse_proj_weight_bytes(Cc > 64 ? 64 : Cc));
  return b + 256;
}
extern "C" size_t drs_upconv_fused_workspace_bytes(int N, int Cc, int Ch, int LH, int LW) {
  size_t o[16];
  return upfused_sizes(N, Cc, Ch, LH, LW, o);
}
extern "C" int drs_upconv_fused_nchw(const float* h, const float* att, const float* t_w, const float* t_b, const float* v_w,
                                     const float* v_b, const float* post2, const float* fuse_w, const float* fuse_b,
                                     int fuse_dim, float* y, float* y2, int N, int Cc, int Ch, int LH, int LW, void* workspace,
                                     size_t workspace_bytes, drs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) return DRS_OK;
  DRS_REQUIRE(h && att && t_w && t_b && v_w && v_b && y && workspace, DRS_ERR_ARG, "upconv_fused: null pointer");
  DRS_REQUIRE(N > 0 && LH > 0 && LW > 0 && Cc >= 32 && Ch >= 32 && Cc % 32 == 0 && Ch % 32 == 0, DRS_ERR_SHAPE,
              "upconv_fused: N=%d Cc=%d Ch=%d LH=%d LW=%d (channel counts must be multiples of 32)", N, Cc, Ch, LH, LW);
  DRS_REQUIRE(!fuse_w || (Ch == 32 && fuse_dim >= 1 && fuse_dim <= 4 && fuse_b && !post2 && !y2), DRS_ERR_SHAPE,
              "upconv_fused: the fused projection needs Ch == 32, fuse_dim <= 4 and no second output");
  DRS_REQUIRE((post2 == nullptr) == (y2 == nullptr), DRS_ERR_ARG, "upconv_fused: post2 and y2 come together");
  size_t o[16];
  DRS_REQUIRE(workspace_bytes >= upfused_sizes(N, Cc, Ch, LH, LW, o), DRS_ERR_WORKSPACE, "upconv_fused: workspace too small");
  char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const int OH = 2 * LH, OW = 2 * LW;
  int rc;
  DRS_CHECK_HIP(hipMemsetAsync(base + o[10], 0, 512, s));
  if ((rc = drs_launch_nchw_to_sp(h, (float*)(base + o[0]), N, Cc, LH, LW, s))) return rc;
  if ((rc = drs_launch_nchw_to_sp(att, (float*)(base + o[1]), N, Ch, OH, OW, s))) return rc;
  // the projection folded into both launches' weights where the direct kernel takes the att-half (what the plan's stage 2
  // does: DecStage::ah_proj / uf_proj), else as the matrix-pipe epilogue of the 32-channel layer
  bool fold = false;
  TapConv ah = conv_desc((const float*)(base + o[1]), N, OH, OW, Ch, Ch, 0, (const float*)(base + o[6]), nullptr, nullptr, 16, 16, 0, 3, 3, 1, 1);
  if (fuse_w) {
    ah.in_sp = 1; ah.zero_line = base + o[10]; ah.proj = 1; ah.fuse_out = y; ah.fuse_dim = fuse_dim;
    fold = drs_conv3x3_direct_sp_proj_supported(ah, DRS_IMPL_MFMA_BF16X3) && drs_upfuse_proj_supported(Cc, Ch, fuse_dim);
  }
  const float *uv_w = v_w, *uv_b = v_b;
  if (fold) {
    if ((rc = drs_launch_upfuse_fold_proj(v_w, v_b, fuse_w, fuse_b, fuse_dim, Cc, Ch, (float*)(base + o[13]), (float*)(base + o[14]), s))) return rc;
    uv_w = (const float*)(base + o[13]); uv_b = (const float*)(base + o[14]);
    if ((rc = drs_launch_upfuse_proj_pack(uv_w, t_w, Cc, Ch, fuse_dim, base + o[15], s))) return rc;
    if ((rc = drs_launch_fold_proj(v_w, Cc + Ch, Cc, Ch, Ch, fuse_w, fuse_dim, (float*)(base + o[12]), s))) return rc;
    if ((rc = drs_launch_pack_conv_mfma((const float*)(base + o[12]), nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, base + o[6],
                                        (float*)(base + o[7]), 16, Ch, 3, 0, DRS_IMPL_MFMA_BF16X3, s, 0, 0, 0, 0, 0)))
      return rc;
  }
  if ((rc = drs_launch_upfuse_pack(uv_w, uv_b, t_w, t_b, Cc, Ch, base + o[4], (float*)(base + o[5]), base + o[11], s))) return rc;
  if (!fold &&
      (rc = drs_launch_pack_conv_mfma(v_w, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, base + o[6], (float*)(base + o[7]), Ch,
                                      Ch, 9, 0, DRS_IMPL_MFMA_BF16X3, s, 0, 0, 0, 0, fuse_w ? 0 : 1, Cc + Ch, Cc)))
    return rc;
  const float* aux = (const float*)(base + o[5]);
  const size_t mat = (size_t)Cc * Ch;
  unsigned* fault = (unsigned*)(base + o[10] + 256);
  {
    UpFuseEdgeDesc e = {};
    e.in = (const float*)(base + o[0]); e.in_cs = Cc; e.in_co = 0;
    e.N = N; e.LH = LH; e.LW = LW; e.Cc = Cc; e.Ch = Ch;
    e.rt = aux; e.rl = aux + 5 * mat; e.bt = aux + 11 * mat;
    e.eh = (float*)(base + o[8]); e.ev = (float*)(base + o[9]);
    e.wimg = base + o[11]; e.zero_line = base + o[10];
    if ((rc = drs_launch_upfuse_edges(e, s))) return rc;
  }
  {
    TapConv d = conv_desc((const float*)(base + o[1]), N, OH, OW, Ch, Ch, 0, (const float*)(base + o[6]), (const float*)(base + o[7]),
                          (float*)(base + o[2]), Ch, Ch, 0, 3, 3, 1, 1);
    d.in_sp = d.out_sp = 1; d.zero_line = base + o[10]; d.fault = fault;
    if (fold) {
      if ((rc = drs_launch_conv3x3_direct_sp(ah, s))) return rc;
    } else {
    if (fuse_w) {  // projected att-half straight into y (the plan's stage 2)
      d.out = nullptr; d.out_sp = 0;
      d.fuse_w = fuse_w; d.fuse_b = (const float*)(base + o[7]); d.fuse_out = y; d.fuse_dim = fuse_dim;
    }
    if ((rc = drs_launch_tapconv_mfma(d, DRS_IMPL_MFMA_BF16X3, s))) return rc;
    }
  }
  {
    UpFuseDesc u = {};
    u.in = (const float*)(base + o[0]); u.in_cs = Cc; u.in_co = 0;
    u.N = N; u.LH = LH; u.LW = LW; u.Cc = Cc; u.Ch = Ch;
    u.w = base + o[4];
    u.bias = aux + 11 * mat + 9 * Ch;
    u.res = (const float*)(base + o[2]); u.res_cs = Ch; u.res_co = 0;
    u.eh = (const float*)(base + o[8]); u.ev = (const float*)(base + o[9]);
    u.zero_line = base + o[10]; u.fault = fault;
    if (fuse_w) {
      u.res = nullptr; u.fuse_acc = 1;
      if (fold) u.proj = 1;
      else { u.fuse_w = fuse_w; u.fuse_b = fuse_b; }
      u.fuse_out = y; u.fuse_dim = fuse_dim;
    } else {
      u.out = (float*)(base + o[3]); u.out_cs = Ch; u.out_co = 0;
      if (y2) { u.out2 = (float*)(base + o[1]); u.out2_cs = Ch; u.out2_co = 0; u.post2 = post2; u.post2_cs = Ch; }  // (att is consumed by now)
    }
    if (fold) { u.w = base + o[15]; rc = drs_launch_upfuse_proj(u, s); }
    else rc = drs_launch_upfuse(u, s);
    if (rc) return rc;
  }
  if (!fuse_w) {
    if ((rc = drs_launch_sp_to_nchw((const float*)(base + o[3]), y, N, Ch, OH, OW, Ch, 0, s))) return rc;
    if (y2 && (rc = drs_launch_sp_to_nchw((const float*)(base + o[1]), y2, N, Ch, OH, OW, Ch, 0, s))) return rc;
  }
  unsigned word = 0;
  DRS_CHECK_HIP(hipMemcpyAsync(&word, fault, 4, hipMemcpyDeviceToHost, s));
  DRS_CHECK_HIP(hipStreamSynchronize(s));
  DRS_REQUIRE(word == 0, DRS_ERR_HIP, "upconv_fused: a wave-specialised kernel timed out on an LDS counter (protocol fault)");
  return DRS_OK;
}

// ------------------------------------------------------------------------------------------------
// UNet plan
// ------------------------------------------------------------------------------------------------
namespace {

struct Param { std::string name; int64_t numel; };

struct ConvLayer {
  int w = -1, b = -1, bn = -1;  // param indices; bn = index of gamma (beta, mean, var follow)
  int Cout = 0, Cin = 0, taps = 0;
  bool transposed = false, mfma = false;
  bool out_sp = false;  // this layer stores its output in SP format: weights packed with the output-channel permutation
  size_t w_off = 0, b_off = 0;
  // eval split-bf16 plans: "FL" operand images of a wide 3x3 / 1x1 layer (conv_mfma_fl.hip: fp16 main + block-scaled fp6 cross
  // terms, derived from the packed split-bf16 images); fl_ok: the folded weights passed the pack-time fp16 range check
  size_t fl_off = 0;
  int fl_slot = -1;
  bool fl_ok = false;
  int t_Z = -1;         // train plans: pre-BatchNorm tensor
  int t_Zsp = -1;       // train plans, 3x3 stride-1 layers: SP-format copy of dZ for the wave-specialised data-gradient convolution
  size_t stats_off = 0;  // train plans: saved batch mean / rstd (2 x Cout floats) in the workspace
  size_t sums_off = 0;   // train plans: this layer's fp64 reduction slots (2 x Cout) inside the forward / backward sums regions
};
struct PlanarConv { int w = -1, b = -1; int Cout = 0, Cin = 0; size_t w_off = 0, b_off = 0; };
struct Mlp { int w1, b1, w2, b2, dim; size_t o_w1, o_b1, o_w2, o_b2; int temb_off; };

struct WsTensor {
  std::string name;
  size_t off;  // bytes into workspace
  int n, c, h, w;
  int cs, co;   // channel stride / offset (NHWC); planar tensors have cs = 0
  bool planar;
  bool sp = false;  // SP format (split bf16 hi | lo per 32-channel group, drs_common.h)
};

struct ResBlock {
  ConvLayer conv1, conv2, shortcut, skip; bool has_skip; Mlp mlp;
  // eval split-bf16 plans: conv1 (+BN1) and the skip convolution packed as ONE 2*Cout-channel operand image (TapConv::dual)
  bool dual = false;
  size_t dual_w_off = 0, dual_b_off = 0;
};
struct DecStage {
  ConvLayer gate, wg, wx, psi, result, conv, transform, upconv; Mlp mlp;
  // fused attention gate (attn_gate_sp.hip): w_g and w_x once more with the SP output-row permutation
  bool fused_gate = false;
  size_t fz_wg_off = 0, fz_wx_off = 0;
  // the stage input is stored ONLY as x + relu(time_mlp(t)) (what ups.i.conv reads) when the fused gate can take the row
  // vector out through a per-image bias: fp32 BatchNorm-folded gating weights [Cc][Ch] + bias, per-forward bias table
  size_t gf_w_off = 0, gf_b_off = 0, o_gbias = 0;
  // ups.i.transform composed with the x-half of up_convs.i (upfuse_sp.hip): composite operand image, edge / bias weights,
  // and the att-half of up_convs.i packed as its own Ch -> Ch 3x3 convolution (no bias: it is in the composite's)
  bool upfuse = false;
  size_t uf_w_off = 0, uf_aux_off = 0, uf_edge_off = 0, ah_w_off = 0, ah_b_off = 0;
  size_t ah_fl_off = 0;  // FL images of the att-half (stages 0 / 1)
  int ah_fl_slot = -1;
  bool ah_fl_ok = false;
  // stage 2: the `output` projection folded into the att-half's weights (conv3x3_direct_sp.hip, TapConv::proj): a 16-row image,
  // ah_tmp = the fp32 contraction it is packed from
  bool ah_proj = false;
  size_t ah_tmp_off = 0;
  // ... and into the composite's (UpFuseDesc::proj): the folded up_convs.2 x-half the composite is packed from
  bool uf_proj = false;
  size_t uf_tmpw_off = 0, uf_tmpb_off = 0;
  // ... and the attention block's `result` convolution folded in as well: the gate stops at psi (attn_gate_sp.hip, PSI_ONLY), the
  // att-half reads the skip tensor and multiplies by psi behind its MFMAs: `att` of the top stage never exists
  bool gate_psi = false;
  size_t ah_tmp2_off = 0, ah_tab_off = 0;
  // the folded composite on the streaming kernel (upfuse_proj_sp.hip): its own operand image
  bool uf_stream = false;
  size_t ufp_w_off = 0;
  int t_PA = -1;                  // att-half partial sums (SP), B x Ch x 2lh x 2lw
  size_t o_eh = 0, o_ev = 0;      // workspace: edge vectors of this forward
};

}  // namespace

struct drs_plan {
  drs_unet_config cfg;
  std::vector<Param> params;
  std::vector<ConvLayer*> convs;
  std::vector<PlanarConv*> planars;
  std::vector<Mlp*> mlps;
  std::vector<WsTensor> tensors;

  PlanarConv rrdb[7];
  PlanarConv stem0, stemc;  // conv0, conv_upsampled_lr_img (raw torch layout)
  ResBlock enc[4];          // conv_blocks.0..2, bottle_neck
  ConvLayer downs[3];
  DecStage dec[3];
  ConvLayer output;

  size_t packed_bytes = 0, ws_bytes = 0;
  size_t o_inv_freq = 0, o_mlp_table = 0, o_out_w = 0, o_out_b = 0, o_label = 0, o_zero = 0, o_fault = 0;
  // eval plans of the split-bf16 implementation keep every MFMA-consumed activation in SP format (drs_common.h)
  bool sp = false;
  int t_XT[3] = {-1, -1, -1};  // x + relu(time_mlp(t)) of UpConvBlock i (reference :199), second output of its producer
  int label_emb = -1;  // param index of label_emb.weight (generation variant)
  int temb_total = 0;
  std::vector<long long> mlp_table_host;
  std::vector<const void*> param_ptrs;  // as given to the last drs_unet_pack_weights
  // train plans: fp64 totals of the BatchNorm reductions, one (2 x Cout) slot per layer, a region for the forward statistics
  // followed by one for the backward sums; o_red = per-block partial sums of whichever reduction is running on the main
  // stream (BatchNorm statistics, BatchNorm backward, bias-gradient column sums: kRedBlocks x 2 x 1024 doubles).  Partials +
  // a small finishing kernel replaced per-block atomics onto the same 2 x Cout addresses: 512 blocks x 90 ns per serialised
  // atomic = a 46 us floor under every one of those launches, whatever the tensor size (round 3: 63 of them per step).
  size_t o_bn_sums = 0, bn_sums_bytes = 0, o_red = 0;
  // FL arithmetic (conv_mfma_fl.hip) for the layers the wave-specialised SP kernel takes at 64 channels per item; per-layer range
  // flags (device words, one per FL image: bit 0 = a folded weight outside what fp16 holds) are read back at pack time
  bool fl = false;
  bool fl_off = false;  // an activation left fp16's range (drs_unet_check_faults): the plan stays on the split-bf16 kernels
  int fl_slots = 0;
  size_t o_fl_flags = 0;
  bool packed_ok = false;
  const void* packed_ptr = nullptr;
  unsigned* fault_ptr = nullptr;  // device word of the current forward's packed buffer (TapConv::fault)

  // optional per-op timing (drs_unet_profile_*): events recorded on the forward's stream
  struct OpRec { std::string name; double flops, bytes; hipEvent_t e0, e1; };
  bool profiling = false;
  std::vector<OpRec> ops;
  // launch log of the last profiled forward (drs_note_launch): one entry per kernel launch, in host launch order
  struct LaunchRec { std::string op, kernel; };
  std::vector<LaunchRec> launches;
  std::string cur_op;  // op of the schedule whose prof_begin / prof_end bracket is open ("" between ops)

  // workspace offsets
  size_t o_lr[3], o_up, o_temb;
  int t_cond, t_x0, t_S[4], t_K0, t_H[4], t_R[4], t_D[3];
  int t_G[3], t_Q[3], t_P[3], t_PSI[3], t_U[3], t_CAT[3], t_X[3];
  int t_lrenc, t_up;
  // train plans: gradients of activations and channels-last copies of the 3-channel tensors (backward only)
  int g_out = -1, g_X[3], g_CAT[3], g_U[3], g_G[3], g_P[3], g_PSI[3], g_E[3], g_R[4], g_D[3], g_H[4], g_x0 = -1;
  int t_xn = -1, t_upn = -1, g_upn = -1, g_lr[4], t_rn[4], t_an[3];
  size_t o_dtemb = 0, o_scratch = 0, o_wgrad = 0;
  // second stream of the eval forward: the attention branch of a decoder stage runs next to the up-sampling branch
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_gbias = nullptr;
  hipEvent_t ev_edge_in[3] = {nullptr, nullptr, nullptr}, ev_edge_out[3] = {nullptr, nullptr, nullptr};  // edge vectors of a composite stage: side stream

  int P(const std::string& name, int64_t numel) {
    params.push_back({name, numel});
    return (int)params.size() - 1;
  }
  ConvLayer mk_conv(const std::string& pfx, int Cout, int Cin, int taps, const std::string& bn_pfx = "",
                    bool transposed = false) {
    ConvLayer L;
    L.Cout = Cout; L.Cin = Cin; L.taps = taps; L.transposed = transposed;
    L.w = P(pfx + ".weight", (int64_t)Cout * Cin * taps);
    L.b = P(pfx + ".bias", Cout);
    if (!bn_pfx.empty()) {
      L.bn = P(bn_pfx + ".weight", Cout);
      P(bn_pfx + ".bias", Cout);
      P(bn_pfx + ".running_mean", Cout);
      P(bn_pfx + ".running_var", Cout);
    }
    return L;
  }
  PlanarConv mk_planar(const std::string& pfx, int Cout, int Cin) {
    PlanarConv L;
    L.Cout = Cout; L.Cin = Cin;
    L.w = P(pfx + ".weight", (int64_t)Cout * Cin * 9);
    L.b = P(pfx + ".bias", Cout);
    return L;
  }
  Mlp mk_mlp(const std::string& pfx, int dim) {
    Mlp m;
    m.dim = dim;
    m.w1 = P(pfx + ".0.weight", (int64_t)dim * 100);
    m.b1 = P(pfx + ".0.bias", dim);
    m.w2 = P(pfx + ".2.weight", (int64_t)dim * dim);
    m.b2 = P(pfx + ".2.bias", dim);
    m.temb_off = temb_total;
    temb_total += dim;
    return m;
  }
  int T(const std::string& name, size_t& cursor, int n, int c, int h, int w, bool planar = false) {
    WsTensor t{name, cursor, n, c, h, w, planar ? 0 : c, 0, planar, false};
    cursor += align_up((size_t)n * c * h * w * 4);
    tensors.push_back(t);
    return (int)tensors.size() - 1;
  }
  int Tview(const std::string& name, int base, int c, int co) {
    WsTensor t = tensors[base];
    t.name = name; t.c = c; t.co = co;
    tensors.push_back(t);
    return (int)tensors.size() - 1;
  }
  float* tp(void* ws, int i) const { return (float*)((char*)ws + tensors[i].off); }
};

static const size_t kWgradPartialBytes = 64ull << 20;
static const int kRedBlocks = DRS_RED_BLOCKS;  // blocks of a partial-sum reduction (drs_common.h)
static const int kDown[5] = {16, 32, 64, 128, 256};
static const int kUp[5] = {256, 128, 64, 32, 16};

extern "C" int drs_unet_plan_create(drs_plan** out, const drs_unet_config* cfg) {
  DRS_REQUIRE(out && cfg, DRS_ERR_ARG, "plan_create: null pointer");
  DRS_REQUIRE(cfg->batch >= 1 && (cfg->lr_batch == cfg->batch || cfg->lr_batch == 1), DRS_ERR_SHAPE,
              "plan_create: batch=%d lr_batch=%d (lr batch must equal batch or be 1)", cfg->batch, cfg->lr_batch);
  DRS_REQUIRE(cfg->image_channels >= 1 && cfg->image_channels <= 4 && cfg->out_dim >= 1, DRS_ERR_SHAPE,
              "plan_create: image_channels=%d out_dim=%d", cfg->image_channels, cfg->out_dim);
  DRS_REQUIRE(cfg->magnification >= 1 && cfg->height > 0 && cfg->width > 0 && cfg->height % 8 == 0 &&
                  cfg->width % 8 == 0 && cfg->height % cfg->magnification == 0 && cfg->width % cfg->magnification == 0,
              DRS_ERR_SHAPE, "plan_create: H=%d W=%d must be divisible by 8 and by magnification=%d", cfg->height,
              cfg->width, cfg->magnification);
  DRS_REQUIRE(cfg->impl >= DRS_IMPL_DIRECT && cfg->impl <= DRS_IMPL_MFMA_F16, DRS_ERR_ARG, "plan_create: impl=%d",
              cfg->impl);
  DRS_REQUIRE(cfg->variant >= DRS_VARIANT_SUPERRES && cfg->variant <= DRS_VARIANT_GENERATION, DRS_ERR_ARG,
              "plan_create: variant=%d", cfg->variant);
  DRS_REQUIRE(cfg->num_classes >= 0, DRS_ERR_ARG, "plan_create: num_classes=%d", cfg->num_classes);
  drs_plan* p = new drs_plan();
  p->cfg = *cfg;
  if (p->cfg.bn_eps <= 0.f) p->cfg.bn_eps = 1e-5f;
  const int C = cfg->image_channels;
  if (p->cfg.variant == DRS_VARIANT_SUPERRES && p->cfg.cond_channels == 0) p->cfg.cond_channels = C;
  if (p->cfg.variant == DRS_VARIANT_GENERATION) p->cfg.cond_channels = 0;
  const int CC = p->cfg.cond_channels;  // conditioning image channels
  if (p->cfg.variant != DRS_VARIANT_GENERATION && (CC < 1 || CC > 4)) {
    DrsErr::set("plan_create: cond_channels=%d", CC);
    delete p;
    return DRS_ERR_SHAPE;
  }
  if (p->cfg.variant == DRS_VARIANT_SAR_TO_NDVI && cfg->magnification != 1) {
    DrsErr::set("plan_create: the SAR_TO_NDVI variant has no up-sampling (magnification must be 1)");
    delete p;
    return DRS_ERR_SHAPE;
  }
  {
    static const bool sp_env = !(getenv("DRS_SP") && atoi(getenv("DRS_SP")) == 0);
    p->sp = sp_env && !(cfg->flags & DRS_PLAN_TRAIN) && cfg->impl == DRS_IMPL_MFMA_BF16X3;
  }
  const bool has_cond = p->cfg.variant != DRS_VARIANT_GENERATION;
  const std::string enc_name = p->cfg.variant == DRS_VARIANT_SAR_TO_NDVI ? "SAR_encoder" : "LR_encoder";
  const std::string cond_name = p->cfg.variant == DRS_VARIANT_SAR_TO_NDVI ? "conv_SAR_img" : "conv_upsampled_lr_img";
  const std::string skip_name = p->cfg.variant == DRS_VARIANT_GENERATION ? "conv_skip" : cond_name;

  // ---- parameters, in a fixed canonical order (names = reference state_dict keys) ----
  p->stem0 = p->mk_planar("conv0", kDown[0], C);
  if (has_cond) {
    for (int i = 0; i < 3; ++i) {
      p->rrdb[2 * i] = p->mk_planar(enc_name + ".blocks." + std::to_string(i) + ".conv1", CC, CC);
      p->rrdb[2 * i + 1] = p->mk_planar(enc_name + ".blocks." + std::to_string(i) + ".conv2", CC, CC);
    }
    p->rrdb[6] = p->mk_planar(enc_name + ".conv_out", CC, CC);
    p->stemc = p->mk_planar(cond_name, kDown[0], CC);
  }
  if (p->cfg.variant == DRS_VARIANT_GENERATION && cfg->num_classes > 0)
    p->label_emb = p->P("label_emb.weight", (int64_t)cfg->num_classes * 100);
  for (int i = 0; i < 4; ++i) {
    const std::string pfx = i < 3 ? "conv_blocks." + std::to_string(i) : std::string("bottle_neck");
    const int ci = kDown[i], co = kDown[i + 1];
    ResBlock& rb = p->enc[i];
    rb.mlp = p->mk_mlp(pfx + ".time_mlp", co);
    rb.conv1 = p->mk_conv(pfx + ".conv1.0", co, ci, 9, pfx + ".batch_norm1");
    rb.conv2 = p->mk_conv(pfx + ".conv2.0", co, co, 9, pfx + ".batch_norm2");
    rb.shortcut = p->mk_conv(pfx + ".shortcut_conv.0", co, ci, 1, pfx + ".shortcut_batch_norm");
    rb.has_skip = (i == 0);
    if (rb.has_skip) rb.skip = p->mk_conv(pfx + "." + skip_name, co, ci, 9);
    if (i < 3) p->downs[i] = p->mk_conv("downs." + std::to_string(i), co, co, 9);
  }
  for (int i = 0; i < 3; ++i) {
    const std::string si = std::to_string(i);
    const int Cc = kUp[i], Ch = kUp[i + 1];
    DecStage& d = p->dec[i];
    d.gate = p->mk_conv("gating_signals." + si + ".conv", Ch, Cc, 1, "gating_signals." + si + ".batch_norm");
    d.wg = p->mk_conv("attention_blocks." + si + ".w_g.0", Ch, Ch, 1);
    d.wx = p->mk_conv("attention_blocks." + si + ".w_x.0", Ch, Ch, 4);
    d.psi = p->mk_conv("attention_blocks." + si + ".psi.0", 1, Ch, 1);
    d.result = p->mk_conv("attention_blocks." + si + ".result.0", Ch, Ch, 1, "attention_blocks." + si + ".result.1");
    d.mlp = p->mk_mlp("ups." + si + ".time_mlp", Cc);
    d.conv = p->mk_conv("ups." + si + ".conv", Cc, Cc, 9, "ups." + si + ".batch_norm");
    d.transform = p->mk_conv("ups." + si + ".transform", Cc, Cc, 9, "", true);
    d.upconv = p->mk_conv("up_convs." + si, Ch, Cc + Ch, 9);
  }
  p->output = p->mk_conv("output", cfg->out_dim, kUp[3], 1);
  if (p->sp) {
    for (int i = 0; i < 4; ++i) {
      p->enc[i].conv1.out_sp = p->enc[i].conv2.out_sp = true;
      p->enc[i].shortcut.out_sp = true;  // rides inside conv2 as extra K-chunks: same accumulator rows, same permutation
      if (i < 3) p->downs[i].out_sp = true;
    }
    for (int i = 0; i < 3; ++i) {
      DecStage& d = p->dec[i];
      d.gate.out_sp = d.result.out_sp = d.conv.out_sp = d.transform.out_sp = true;
      d.upconv.out_sp = i < 2;  // up_convs.2 feeds the fused / direct output projection in fp32
    }
  }

  // ---- registries ----
  for (int i = 0; i < 4; ++i) {
    p->convs.push_back(&p->enc[i].conv1);
    p->convs.push_back(&p->enc[i].conv2);
    p->convs.push_back(&p->enc[i].shortcut);
    if (p->enc[i].has_skip) p->convs.push_back(&p->enc[i].skip);
    p->mlps.push_back(&p->enc[i].mlp);
    if (i < 3) p->convs.push_back(&p->downs[i]);
  }
  for (int i = 0; i < 3; ++i) {
    DecStage& d = p->dec[i];
    ConvLayer* ls[] = {&d.gate, &d.wg, &d.wx, &d.psi, &d.result, &d.conv, &d.transform, &d.upconv};
    for (ConvLayer* l : ls) p->convs.push_back(l);
    p->mlps.push_back(&d.mlp);
  }
  p->convs.push_back(&p->output);
  p->planars.push_back(&p->stem0);
  if (has_cond) {
    for (int i = 0; i < 7; ++i) p->planars.push_back(&p->rrdb[i]);
    p->planars.push_back(&p->stemc);
  }

  // ---- packed buffer layout ----
  size_t cur = 0;
  p->o_inv_freq = cur; cur += align_up(50 * 4);
  for (ConvLayer* L : p->convs) {
    L->w_off = cur;  // room for whichever kernel family's image is largest
    {
      size_t need = (size_t)L->Cout * L->Cin * L->taps * 4;
      for (int im = DRS_IMPL_MFMA_F32; im <= DRS_IMPL_MFMA_F16; ++im) {
        const size_t m = drs_pack_conv_mfma_bytes(L->Cout, L->Cin, L->taps, im);
        need = m > need ? m : need;
      }
      cur += align_up(need);
    }
    L->b_off = cur; cur += align_up((size_t)L->Cout * 4);
  }
  for (int i = 0; i < 4; ++i) {
    ResBlock& rb = p->enc[i];
    rb.dual = rb.has_skip && !(cfg->flags & DRS_PLAN_TRAIN) && cfg->impl == DRS_IMPL_MFMA_BF16X3 && rb.conv1.Cout == 32 &&
              rb.skip.Cout == 32 && rb.skip.Cin == rb.conv1.Cin;
    if (rb.dual) {
      rb.dual_w_off = cur; cur += align_up(drs_pack_conv_mfma_bytes(64, rb.conv1.Cin, 9, DRS_IMPL_MFMA_BF16X3));
      rb.dual_b_off = cur; cur += align_up((size_t)64 * 4);
    }
  }
  for (int i = 0; i < 3; ++i) {
    DecStage& d = p->dec[i];
    d.fused_gate = p->sp && drs_attn_gate_supported(kUp[i], kUp[i + 1]);
    if (d.fused_gate) {
      d.fz_wg_off = cur; cur += align_up(drs_pack_conv_mfma_bytes(d.wg.Cout, d.wg.Cin, 1, DRS_IMPL_MFMA_BF16X3));
      d.fz_wx_off = cur; cur += align_up(drs_pack_conv_mfma_bytes(d.wx.Cout, d.wx.Cin, 4, DRS_IMPL_MFMA_BF16X3));
      d.gf_w_off = cur; cur += align_up((size_t)d.gate.Cout * d.gate.Cin * 4);
      d.gf_b_off = cur; cur += align_up((size_t)d.gate.Cout * 4);
    }
  }
  for (int i = 0; i < 3; ++i) {
    DecStage& d = p->dec[i];
    const int Cc = kUp[i], Ch = kUp[i + 1];
    // (KEEP_ALL plans stay unfused: ups.i is a parity tap; stage 2 needs the fused output projection: its result is fp32)
    d.upfuse = p->sp && !(cfg->flags & DRS_PLAN_KEEP_ALL) && (i < 2 || (Ch == 32 && cfg->out_dim <= 4)) &&
               drs_upfuse_supported(Cc, Ch, cfg->height >> (3 - i), cfg->width >> (3 - i));
    if (d.upfuse) {
      d.uf_w_off = cur; cur += align_up(drs_upfuse_weight_bytes(Cc, Ch));
      d.uf_aux_off = cur; cur += align_up(drs_upfuse_aux_floats(Cc, Ch) * 4);
      d.uf_edge_off = cur; cur += align_up(drs_upfuse_edge_image_bytes(Cc, Ch));
      d.ah_w_off = cur; cur += align_up(drs_pack_conv_mfma_bytes(Ch, Ch, 9, DRS_IMPL_MFMA_BF16X3));
      d.ah_b_off = cur; cur += align_up((size_t)Ch * 4);
      if (i == 2) {
        TapConv probe = conv_desc((const float*)256, cfg->batch, cfg->height, cfg->width, Ch, Cc + Ch, Cc, (const float*)256, nullptr, nullptr,
                                  16, 16, 0, 3, 3, 1, 1);
        probe.in_sp = 1; probe.zero_line = (const void*)256; probe.proj = 1; probe.fuse_out = (float*)256; probe.fuse_dim = cfg->out_dim;
        d.ah_proj = drs_conv3x3_direct_sp_proj_supported(probe, cfg->impl);
        if (d.ah_proj) { d.ah_tmp_off = cur; cur += align_up((size_t)16 * Ch * 9 * 4); }
        // the composite's folded form needs the att-half in the output tensor first (fuse_acc): both or neither
        d.uf_proj = d.ah_proj && drs_upfuse_proj_supported(Cc, Ch, cfg->out_dim);
        static const bool gp_env = !(getenv("DRS_GATE_PSI") && atoi(getenv("DRS_GATE_PSI")) == 0);
        d.gate_psi = gp_env && d.ah_proj && d.fused_gate && Ch == 32 && !(cfg->flags & DRS_PLAN_KEEP_ALL) && !(cfg->height & 1) && !(cfg->width & 1);
        if (d.gate_psi) {
          d.ah_tmp2_off = cur; cur += align_up((size_t)16 * Ch * 3 * 4);
          d.ah_tab_off = cur; cur += align_up((size_t)36 * 4);
        }
        if (d.uf_proj) {
          d.uf_tmpw_off = cur; cur += align_up((size_t)32 * (Cc + Ch) * 9 * 4);
          d.uf_tmpb_off = cur; cur += align_up((size_t)32 * 4);
          d.uf_stream = true;
          d.ufp_w_off = cur; cur += align_up(drs_upfuse_proj_weight_bytes(Cc));
        }
      }
    }
  }
  {
    static const bool fl_env = !(getenv("DRS_FL") && atoi(getenv("DRS_FL")) == 0);
    p->fl = fl_env && p->sp;
    if (p->fl) {
      for (ConvLayer* L : p->convs)
        if ((L->taps == 9 || L->taps == 1) && !L->transposed && L->Cout % 64 == 0 && L->Cin % 32 == 0) {
          L->fl_off = cur; cur += align_up(drs_fl_image_bytes(L->Cout, L->Cin, L->taps));
          L->fl_slot = p->fl_slots++;
        }
      for (int i = 0; i < 2; ++i) {
        DecStage& d = p->dec[i];
        if (d.upfuse && kUp[i + 1] % 64 == 0) {
          d.ah_fl_off = cur; cur += align_up(drs_fl_image_bytes(kUp[i + 1], kUp[i + 1], 9));
          d.ah_fl_slot = p->fl_slots++;
        }
      }
      p->o_fl_flags = cur; cur += align_up((size_t)p->fl_slots * 8);  // [range flag, largest |weight|] per image
    }
  }
  for (PlanarConv* L : p->planars) {
    L->w_off = cur; cur += align_up((size_t)L->Cout * L->Cin * 9 * 4);
    L->b_off = cur; cur += align_up((size_t)L->Cout * 4);
  }
  for (Mlp* m : p->mlps) {
    m->o_w1 = cur; cur += align_up((size_t)m->dim * 100 * 4);
    m->o_b1 = cur; cur += align_up((size_t)m->dim * 4);
    m->o_w2 = cur; cur += align_up((size_t)m->dim * m->dim * 4);
    m->o_b2 = cur; cur += align_up((size_t)m->dim * 4);
  }
  p->o_mlp_table = cur; cur += align_up(p->mlps.size() * 6 * sizeof(long long));
  p->o_label = cur; cur += align_up((size_t)(cfg->num_classes > 0 ? cfg->num_classes : 0) * 100 * 4);
  p->o_out_w = cur; cur += align_up((size_t)cfg->out_dim * kUp[3] * 4);
  p->o_out_b = cur; cur += align_up((size_t)cfg->out_dim * 4);
  p->o_zero = cur; cur += 256;  // a line of zeros: source of out-of-image pixels for LDS-DMA staging
  p->o_fault = cur; cur += 256;  // TapConv::fault word of the wave-specialised kernels (drs_unet_check_faults)
  p->packed_bytes = cur;

  // ---- workspace layout ----
  const int B = cfg->batch, Bl = cfg->lr_batch, H = cfg->height, W = cfg->width, mag = cfg->magnification;
  const int h = H / mag, w = W / mag;
  size_t ws = 0;
  const int CCw = CC > 0 ? CC : 1;
  for (int i = 0; i < 3; ++i) { p->o_lr[i] = ws; ws += align_up((size_t)Bl * CCw * h * w * 4); }
  p->o_temb = ws; ws += align_up((size_t)B * p->temb_total * 4);
  p->t_lrenc = p->T(enc_name, ws, Bl, CCw, h, w, true);
  p->t_up = p->T("upsampled_lr_img", ws, Bl, CCw, H, W, true);
  p->t_cond = p->T("cond", ws, Bl, kDown[0], H, W);
  p->t_x0 = p->T("x0", ws, B, kDown[0], H, W);
  for (int i = 0; i < 4; ++i) {
    const int co = kDown[i + 1], hh = H >> i, ww = W >> i;
    const std::string nm = i < 3 ? "conv_blocks." + std::to_string(i) : std::string("bottle_neck");
    p->t_S[i] = p->T(nm + ".shortcut", ws, B, co, hh, ww);
    if (i == 0) p->t_K0 = p->T(nm + ".skip", ws, B, co, hh, ww);
    p->t_H[i] = p->T(nm + ".h", ws, B, co, hh, ww);
    p->t_R[i] = p->T(nm, ws, B, co, hh, ww);
    if (i < 3) p->t_D[i] = p->T("downs." + std::to_string(i), ws, B, co, hh / 2, ww / 2);
  }
  for (int i = 0; i < 3; ++i) {
    const std::string si = std::to_string(i);
    const int Cc = kUp[i], Ch = kUp[i + 1];
    const int lh = H >> (3 - i), lw = W >> (3 - i);
    p->t_G[i] = p->T("gating_signals." + si, ws, B, Ch, lh, lw);
    p->t_Q[i] = p->T("attention_blocks." + si + ".g1", ws, B, Ch, lh, lw);
    p->t_P[i] = p->T("attention_blocks." + si + ".relu", ws, B, Ch, lh, lw);
    p->t_PSI[i] = p->T("attention_blocks." + si + ".psi", ws, B, 1, lh, lw);
    p->t_U[i] = p->T("ups." + si + ".conv", ws, B, Cc, lh, lw);
    p->t_CAT[i] = p->T("cat." + si, ws, B, Cc + Ch, 2 * lh, 2 * lw);
    p->Tview("ups." + si, p->t_CAT[i], Cc, 0);
    p->Tview("attention_blocks." + si, p->t_CAT[i], Ch, Cc);
    p->t_X[i] = p->T("up_convs." + si, ws, B, Ch, 2 * lh, 2 * lw);
  }
  if (p->sp) {
    for (int i = 0; i < 3; ++i) {
      const int lh = H >> (3 - i), lw = W >> (3 - i);
      p->t_XT[i] = p->T("ups." + std::to_string(i) + ".in", ws, B, kUp[i], lh, lw);
    }
    auto mark = [&](int t) { p->tensors[t].sp = true; };
    for (int i = 0; i < 3; ++i) {
      DecStage& d = p->dec[i];
      if (d.fused_gate) { d.o_gbias = ws; ws += align_up((size_t)B * kUp[i + 1] * 4); }
      if (!d.upfuse) continue;
      const int lh = H >> (3 - i), lw = W >> (3 - i), Ch = kUp[i + 1];
      if (i < 2) {  // (stage 2 hands its att-half over projected, through the output tensor)
        d.t_PA = p->T("up_convs." + std::to_string(i) + ".att_half", ws, B, Ch, 2 * lh, 2 * lw);
        mark(d.t_PA);
      }
      d.o_eh = ws; ws += align_up((size_t)B * 2 * (2 * lw) * Ch * 4);
      d.o_ev = ws; ws += align_up((size_t)B * 2 * (2 * lh) * Ch * 4);
    }
    mark(p->t_x0);
    for (int i = 0; i < 4; ++i) { mark(p->t_H[i]); mark(p->t_R[i]); if (i < 3) mark(p->t_D[i]); }
    for (int i = 0; i < 3; ++i) {
      mark(p->t_G[i]); mark(p->t_U[i]); mark(p->t_CAT[i]); mark(p->t_XT[i]);
      if (i < 2) mark(p->t_X[i]);
    }
    for (WsTensor& t : p->tensors)  // channel-slice views of the concat buffers
      for (int i = 0; i < 3; ++i)
        if (t.off == p->tensors[p->t_CAT[i]].off) t.sp = true;
  }
  if (cfg->flags & DRS_PLAN_TRAIN) {
    size_t sums_cur = 0;
    auto addz = [&](ConvLayer& L, const std::string& nm, int hh, int ww) {
      L.stats_off = ws; ws += align_up(2 * (size_t)L.Cout * 4);
      L.sums_off = sums_cur; sums_cur += 2 * (size_t)L.Cout * sizeof(double);
      L.t_Z = p->T(nm + ".pre_bn", ws, B, L.Cout, hh, ww);
      if (L.taps == 9 && L.Cout % 32 == 0 && hh > 8) L.t_Zsp = p->T(nm + ".dz_sp", ws, B, L.Cout, hh, ww);
    };
    for (int i = 0; i < 4; ++i) {
      const std::string nm = i < 3 ? "conv_blocks." + std::to_string(i) : std::string("bottle_neck");
      addz(p->enc[i].conv1, nm + ".conv1", H >> i, W >> i);
      addz(p->enc[i].conv2, nm + ".conv2", H >> i, W >> i);
      addz(p->enc[i].shortcut, nm + ".shortcut_conv", H >> i, W >> i);
    }
    for (int i = 0; i < 3; ++i) {
      const std::string si = std::to_string(i);
      const int lh = H >> (3 - i), lw = W >> (3 - i);
      addz(p->dec[i].gate, "gating_signals." + si, lh, lw);
      addz(p->dec[i].result, "attention_blocks." + si + ".result", 2 * lh, 2 * lw);
      addz(p->dec[i].conv, "ups." + si + ".conv_bn", lh, lw);
    }
    p->bn_sums_bytes = align_up(sums_cur);
    p->o_bn_sums = ws; ws += 2 * p->bn_sums_bytes;  // [forward | backward]
    p->o_red = ws; ws += align_up((size_t)kRedBlocks * 2 * 1024 * sizeof(double));
  }
  if (cfg->flags & DRS_PLAN_TRAIN) {
    p->o_dtemb = ws; ws += align_up((size_t)B * p->temb_total * 4);
    p->o_scratch = ws; ws += align_up(64 * 1024);
    p->o_wgrad = ws; ws += align_up(kWgradPartialBytes);  // partial dW slices of the MFMA weight-gradient kernel
    p->g_out = p->T("grad.out", ws, B, cfg->out_dim, H, W);
    p->g_x0 = p->T("grad.x0", ws, B, 32, H, W);  // 16 channels at a 32-float pixel stride (train_bwd.inc: kGx0Stride)
    p->t_xn = p->T("x.nhwc", ws, B, C, H, W);
    p->t_upn = p->T("upsampled_lr_img.nhwc", ws, B, CCw, H, W);
    p->g_upn = p->T("grad.upsampled_lr_img", ws, B, CCw, H, W);
    for (int i = 0; i < 4; ++i) p->g_lr[i] = p->T("grad.lr." + std::to_string(i), ws, B, CCw, h, w);
    for (int i = 0; i < 4; ++i) p->t_rn[i] = p->T("LR_encoder.r" + std::to_string(i) + ".nhwc", ws, B, CCw, h, w);
    for (int i = 0; i < 3; ++i) p->t_an[i] = p->T("LR_encoder.a" + std::to_string(i) + ".nhwc", ws, B, CCw, h, w);
    for (int i = 0; i < 4; ++i) {
      const int co = kDown[i + 1], hh = H >> i, ww = W >> i;
      p->g_R[i] = p->T("grad.R" + std::to_string(i), ws, B, co, hh, ww);
      p->g_H[i] = p->T("grad.H" + std::to_string(i), ws, B, co, hh, ww);
      if (i < 3) p->g_D[i] = p->T("grad.D" + std::to_string(i), ws, B, co, hh / 2, ww / 2);
    }
    for (int i = 0; i < 3; ++i) {
      const int Cc = kUp[i], Ch = kUp[i + 1];
      const int lh = H >> (3 - i), lw = W >> (3 - i);
      p->g_G[i] = p->T("grad.G" + std::to_string(i), ws, B, Ch, lh, lw);
      p->g_P[i] = p->T("grad.P" + std::to_string(i), ws, B, Ch, lh, lw);
      p->g_PSI[i] = p->T("grad.psi" + std::to_string(i), ws, B, 1, lh, lw);
      p->g_U[i] = p->T("grad.U" + std::to_string(i), ws, B, Cc, lh, lw);
      p->g_E[i] = p->T("grad.E" + std::to_string(i), ws, B, Ch, 2 * lh, 2 * lw);
      p->g_CAT[i] = p->T("grad.cat" + std::to_string(i), ws, B, Cc + Ch, 2 * lh, 2 * lw);
      p->g_X[i] = p->T("grad.X" + std::to_string(i), ws, B, Ch, 2 * lh, 2 * lw);
    }
  }
  p->ws_bytes = ws + 256;
  *out = p;
  return DRS_OK;
}

extern "C" void drs_unet_plan_destroy(drs_plan* plan) {
  if (!plan) return;
  if (plan->side) {
    (void)hipStreamSynchronize(plan->side);
    // (a failing destroy would leave a sticky HIP error for the caller's next runtime call to trip over)
    if (plan->ev_fork) (void)hipEventDestroy(plan->ev_fork);
    if (plan->ev_join) (void)hipEventDestroy(plan->ev_join);
    if (plan->ev_gbias) (void)hipEventDestroy(plan->ev_gbias);
    for (int i = 0; i < 3; ++i) {
      if (plan->ev_edge_in[i]) (void)hipEventDestroy(plan->ev_edge_in[i]);
      if (plan->ev_edge_out[i]) (void)hipEventDestroy(plan->ev_edge_out[i]);
    }
    (void)hipStreamDestroy(plan->side);
  }
  delete plan;
}
extern "C" int drs_unet_num_params(const drs_plan* plan) { return plan ? (int)plan->params.size() : 0; }
extern "C" const char* drs_unet_param_name(const drs_plan* plan, int i) {
  return (plan && i >= 0 && i < (int)plan->params.size()) ? plan->params[i].name.c_str() : nullptr;
}
extern "C" int64_t drs_unet_param_numel(const drs_plan* plan, int i) {
  return (plan && i >= 0 && i < (int)plan->params.size()) ? plan->params[i].numel : -1;
}
extern "C" size_t drs_unet_packed_bytes(const drs_plan* plan) { return plan ? plan->packed_bytes + 256 : 0; }
extern "C" size_t drs_unet_workspace_bytes(const drs_plan* plan) { return plan ? plan->ws_bytes : 0; }

static inline char* aligned_base(const void* p) { return (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255); }

extern "C" int drs_unet_pack_weights(drs_plan* plan, const void* const* params, const float* inv_freq_host,
                                     void* packed, size_t packed_bytes, drs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  DRS_REQUIRE(plan && params && inv_freq_host && packed, DRS_ERR_ARG, "pack_weights: null pointer");
  DRS_REQUIRE(packed_bytes >= drs_unet_packed_bytes(plan), DRS_ERR_WORKSPACE, "pack_weights: packed buffer too small");
  for (size_t i = 0; i < plan->params.size(); ++i)
    DRS_REQUIRE(params[i], DRS_ERR_ARG, "pack_weights: param %s is null", plan->params[i].name.c_str());
  char* base = aligned_base(packed);
  auto F = [&](int i) { return (const float*)params[i]; };
  int rc;
  DRS_CHECK_HIP(hipMemcpyAsync(base + plan->o_inv_freq, inv_freq_host, 50 * 4, hipMemcpyHostToDevice, s));
  const int impl = plan->cfg.impl;
  DrsPackQueueScope pack_queue;  // the MFMA operand images of all layers: a few batched launches at the end (drs_common.h)
  for (ConvLayer* L : plan->convs) {
    // kernel family per layer: decided on shape alone
    TapConv probe = {};
    probe.Cin = L->Cin; probe.Cout = L->Cout; probe.ntaps = L->taps;
    L->mfma = impl != DRS_IMPL_DIRECT && drs_tapconv_mfma_supported(probe, impl);
    const float *g = nullptr, *be = nullptr, *rm = nullptr, *rv = nullptr;
    if (L->bn >= 0 && !(plan->cfg.flags & DRS_PLAN_TRAIN)) { g = F(L->bn); be = F(L->bn + 1); rm = F(L->bn + 2); rv = F(L->bn + 3); }
    if (L->mfma)
      rc = drs_launch_pack_conv_mfma(F(L->w), F(L->b), g, be, rm, rv, plan->cfg.bn_eps, base + L->w_off,
                                     (float*)(base + L->b_off), L->Cout, L->Cin, L->taps, L->transposed ? 1 : 0, impl, s, 0, 0, 0,
                                     0, L->out_sp ? 1 : 0);
    else
      rc = drs_launch_pack_conv(F(L->w), F(L->b), g, be, rm, rv, plan->cfg.bn_eps, (float*)(base + L->w_off),
                                (float*)(base + L->b_off), L->Cout, L->Cin, L->taps, L->transposed ? 1 : 0, 0, s);
    if (rc) return rc;
  }
  for (int i = 0; i < 4; ++i) {
    const ResBlock& rb = plan->enc[i];
    if (!rb.dual) continue;
    const ConvLayer& a = rb.conv1;  // channels [0, 32): conv1 with BatchNorm1 folded; [32, 64): the skip convolution
    const ConvLayer& b = rb.skip;
    if ((rc = drs_launch_pack_conv_mfma(F(a.w), F(a.b), F(a.bn), F(a.bn + 1), F(a.bn + 2), F(a.bn + 3), plan->cfg.bn_eps,
                                        base + rb.dual_w_off, (float*)(base + rb.dual_b_off), 64, a.Cin, 9, 0, impl, s, 32, 0, 0, 1,
                                        plan->sp ? 1 : 0)))  // (partial, like the skip half: the two jobs share a launch and an image)
      return rc;
    if ((rc = drs_launch_pack_conv_mfma(F(b.w), F(b.b), nullptr, nullptr, nullptr, nullptr, 0.f, base + rb.dual_w_off,
                                        (float*)(base + rb.dual_b_off), 64, b.Cin, 9, 0, impl, s, 32, 0, 32, 1,
                                        plan->sp ? 1 : 0)))
      return rc;
  }
  for (int i = 0; i < 3; ++i) {
    const DecStage& d = plan->dec[i];
    if (!d.fused_gate) continue;
    // (no bias destination: the layers' own jobs - same batched launch - write d.wg.b_off / d.wx.b_off; two jobs of one launch
    //  storing to one slot was benign only while both computed bit-identical values)
    if ((rc = drs_launch_pack_conv_mfma(F(d.wg.w), F(d.wg.b), nullptr, nullptr, nullptr, nullptr, 0.f, base + d.fz_wg_off,
                                        nullptr, d.wg.Cout, d.wg.Cin, 1, 0, impl, s, 0, 0, 0, 0, 1)))
      return rc;
    if ((rc = drs_launch_pack_conv_mfma(F(d.wx.w), F(d.wx.b), nullptr, nullptr, nullptr, nullptr, 0.f, base + d.fz_wx_off,
                                        nullptr, d.wx.Cout, d.wx.Cin, 4, 0, impl, s, 0, 0, 0, 0, 1)))
      return rc;
    // fp32 [Cc][Ch] gating weights, BatchNorm folded (per-image bias of a stage input stored as x + temb)
    if ((rc = drs_launch_pack_conv(F(d.gate.w), F(d.gate.b), F(d.gate.bn), F(d.gate.bn + 1), F(d.gate.bn + 2), F(d.gate.bn + 3),
                                   plan->cfg.bn_eps, (float*)(base + d.gf_w_off), (float*)(base + d.gf_b_off), d.gate.Cout,
                                   d.gate.Cin, 1, 0, 0, s)))
      return rc;
  }
  for (int i = 0; i < 3; ++i) {
    const DecStage& d = plan->dec[i];
    if (!d.upfuse) continue;
    const int Cc = kUp[i], Ch = kUp[i + 1];
    const float *uv_w = F(d.upconv.w), *uv_b = F(d.upconv.b);
    if (d.uf_proj) {  // `output` folded into up_convs.2: the composite, its edge weights and its bias are built from the folded layer
      float* tw = (float*)(base + d.uf_tmpw_off);
      float* tb = (float*)(base + d.uf_tmpb_off);
      if ((rc = drs_launch_upfuse_fold_proj(uv_w, uv_b, F(plan->output.w), F(plan->output.b), plan->cfg.out_dim, Cc, Ch, tw, tb, s))) return rc;
      uv_w = tw; uv_b = tb;
      if (d.uf_stream && (rc = drs_launch_upfuse_proj_pack(tw, F(d.transform.w), Cc, Ch, plan->cfg.out_dim, base + d.ufp_w_off, s))) return rc;
    }
    if ((rc = drs_launch_upfuse_pack(uv_w, uv_b, F(d.transform.w), F(d.transform.b), Cc, Ch, base + d.uf_w_off,
                                     (float*)(base + d.uf_aux_off), base + d.uf_edge_off, s)))
      return rc;
    // att-half: input channels [Cc, Cc + Ch) of up_convs.i, zero bias; SP output rows in stages 0 / 1, plain MFMA rows in
    // stage 2, whose att-half goes through the fused output projection instead of being stored - or, where the direct kernel
    // takes the layer, has the projection folded into its weights: output o up_convs.2[att half] is ONE 3x3 convolution
    // Ch -> out_dim (reference :377,:379: no activation or normalisation between the two), half the MFMAs of the 32-channel form
    if (d.ah_proj) {
      float* tmp = (float*)(base + d.ah_tmp_off);
      if ((rc = drs_launch_fold_proj(F(d.upconv.w), Cc + Ch, Cc, Ch, Ch, F(plan->output.w), plan->cfg.out_dim, tmp, s))) return rc;
      if (d.gate_psi) {  // ... o attention_blocks.2.result (1x1 + BatchNorm, linear): the convolution then reads psi * x_res
        float* tmp2 = (float*)(base + d.ah_tmp2_off);
        const ConvLayer& R = d.result;
        if ((rc = drs_launch_fold_result(tmp, Ch, F(R.w), F(R.b), F(R.bn), F(R.bn + 1), F(R.bn + 2), F(R.bn + 3), plan->cfg.bn_eps, tmp2,
                                         (float*)(base + d.ah_tab_off), s)))
          return rc;
        tmp = tmp2;
      }
      if ((rc = drs_launch_pack_conv_mfma(tmp, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, base + d.ah_w_off,
                                          (float*)(base + d.ah_b_off), 16, Ch, 3, 0, impl, s, 0, 0, 0, 0, 0)))
        return rc;
      continue;
    }
    if ((rc = drs_launch_pack_conv_mfma(F(d.upconv.w), nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, base + d.ah_w_off,
                                        (float*)(base + d.ah_b_off), Ch, Ch, 9, 0, impl, s, 0, 0, 0, 0, i < 2 ? 1 : 0, Cc + Ch, Cc)))
      return rc;
  }
  // parameters kept verbatim in the packed image: one gather-copy launch (44 hipMemcpyAsync calls before round 4)
  std::vector<DrsCopyJob> copies;
  auto keep = [&](size_t off, int param, long long words) { copies.push_back({F(param), (float*)(base + off), words}); };
  for (PlanarConv* L : plan->planars) {
    keep(L->w_off, L->w, (long long)L->Cout * L->Cin * 9);
    keep(L->b_off, L->b, L->Cout);
  }
  for (Mlp* m : plan->mlps) {
    keep(m->o_w1, m->w1, (long long)m->dim * 100);
    keep(m->o_b1, m->b1, m->dim);
    keep(m->o_w2, m->w2, (long long)m->dim * m->dim);
    keep(m->o_b2, m->b2, m->dim);
  }
  {
    std::vector<long long> table;
    for (Mlp* m : plan->mlps) {
      const long long row[6] = {(long long)m->o_w1, (long long)m->o_b1, (long long)m->o_w2, (long long)m->o_b2, m->dim,
                                m->temb_off};
      table.insert(table.end(), row, row + 6);
    }
    plan->mlp_table_host = table;  // must outlive the async copy
    DRS_CHECK_HIP(hipMemcpyAsync(base + plan->o_mlp_table, plan->mlp_table_host.data(), table.size() * sizeof(long long),
                                 hipMemcpyHostToDevice, s));
    if (plan->label_emb >= 0) keep(plan->o_label, plan->label_emb, (long long)plan->cfg.num_classes * 100);
    keep(plan->o_out_w, plan->output.w, (long long)plan->cfg.out_dim * kUp[3]);
    keep(plan->o_out_b, plan->output.b, plan->cfg.out_dim);
  }
  if ((rc = drs_launch_gather_copy(copies.data(), (int)copies.size(), s))) return rc;
  if ((rc = pack_queue.flush(s))) return rc;
  if (plan->fl) {
    // FL images from the split-bf16 images just packed + the range check of the folded weights: a layer whose weights fp16 cannot
    // hold keeps the split-bf16 kernel (the flags cross to the host here: one stream synchronisation per pack of an eval plan)
    unsigned* flags = (unsigned*)(base + plan->o_fl_flags);
    DRS_CHECK_HIP(hipMemsetAsync(flags, 0, (size_t)plan->fl_slots * 8, s));
    for (ConvLayer* L : plan->convs)
      if (L->fl_off && L->mfma && (rc = drs_launch_fl_repack(base + L->w_off, base + L->fl_off, L->Cout, L->Cin, L->taps, flags + 2 * L->fl_slot, s)))
        return rc;
    for (int i = 0; i < 2; ++i) {
      const DecStage& d = plan->dec[i];
      if (d.ah_fl_off && (rc = drs_launch_fl_repack(base + d.ah_w_off, base + d.ah_fl_off, kUp[i + 1], kUp[i + 1], 9, flags + 2 * d.ah_fl_slot, s)))
        return rc;
    }
    std::vector<unsigned> host((size_t)plan->fl_slots * 2 + 2, 0u);
    DRS_CHECK_HIP(hipMemcpyAsync(host.data(), flags, (size_t)plan->fl_slots * 8, hipMemcpyDeviceToHost, s));
    DRS_CHECK_HIP(hipStreamSynchronize(s));
    for (ConvLayer* L : plan->convs) L->fl_ok = L->fl_off && L->mfma && host[2 * L->fl_slot] == 0u;
    for (int i = 0; i < 2; ++i) plan->dec[i].ah_fl_ok = plan->dec[i].ah_fl_off && host[2 * plan->dec[i].ah_fl_slot] == 0u;
  }
  DRS_CHECK_HIP(hipMemsetAsync(base + plan->o_zero, 0, 512, s));  // zero line + fault word
  plan->param_ptrs.assign(params, params + plan->params.size());
  plan->packed_ok = true;
  plan->packed_ptr = packed;
  return DRS_OK;
}

static void prof_begin(drs_plan* plan, const std::string& name, double flops, double bytes, hipStream_t s) {
  if (!plan->profiling) return;
  drs_plan::OpRec r{name, flops, bytes, nullptr, nullptr};
  (void)hipEventCreate(&r.e0);
  (void)hipEventCreate(&r.e1);
  (void)hipEventRecord(r.e0, s);
  plan->ops.push_back(r);
  plan->cur_op = name;
}
static void prof_end(drs_plan* plan, hipStream_t s) {
  if (!plan->profiling) return;
  (void)hipEventRecord(plan->ops.back().e1, s);
  plan->cur_op.clear();
}

// Launch log (DRS_LAUNCH, drs_common.h): the plan whose profiled forward is running on this host thread, if any.
static thread_local drs_plan* tls_logged_plan = nullptr;
void drs_note_launch(const void* kernel_fn, const char* expr) {
  drs_plan* plan = tls_logged_plan;
  if (!plan) return;
  const char* nm = hipKernelNameRefByPtr(kernel_fn, nullptr);  // mangled name of the device function
  std::string kname = nm ? nm : expr;
  int status = 0;
  if (char* dm = abi::__cxa_demangle(kname.c_str(), nullptr, nullptr, &status)) {
    if (status == 0) kname = dm;
    free(dm);
  }
  plan->launches.push_back({plan->cur_op, kname});
}
struct LaunchLogScope {  // active for the duration of one drs_unet_forward of a profiling plan
  explicit LaunchLogScope(drs_plan* p) { if (p && p->profiling) { p->launches.clear(); p->cur_op.clear(); tls_logged_plan = p; } }
  ~LaunchLogScope() { tls_logged_plan = nullptr; }
};
static int plan_conv(drs_plan* plan, const ConvLayer& L, const TapConv& d_in, hipStream_t s) {
  // second output (TapConv::out2): written by the wave-specialised SP kernel's epilogue; shapes that kernel does not
  // take get it from a separate pass over the first output
  TapConv d = d_in;
  d.fault = plan->fault_ptr;
  if (L.fl_ok && !plan->fl_off && !d.w_fl) d.w_fl = aligned_base(plan->packed_ptr) + L.fl_off;
  const bool split_out2 = d.out2 && !drs_tapconv_sp_supported(d, plan->cfg.impl) && !drs_tapconv_sp8_supported(d, plan->cfg.impl);
  if (split_out2) d.out2 = nullptr;
  std::string name = plan->params[L.w].name;
  name = name.substr(0, name.size() - 7);  // strip ".weight"
  if (d.out_scale == 2 && d.mode != DRS_TAPMODE_CONVT) name += ".phase" + std::to_string(d.out_oy * 2 + d.out_ox);
  prof_begin(plan, name, conv_flops(d), conv_bytes(d), s);
  int rc = run_conv(d, L.mfma ? plan->cfg.impl : DRS_IMPL_DIRECT, s);
  if (!rc && split_out2)
    rc = drs_launch_sp_add_rowvec(d_in.out, d_in.out2, d_in.post2, d_in.post2_cs, d_in.N, (long long)d_in.OH * d_in.OW,
                                  d_in.Cout, s);
  prof_end(plan, s);
  return rc;
}

// ------------------------------------------------------------------------------------------------
// forward schedule
// ------------------------------------------------------------------------------------------------
extern "C" int drs_unet_forward(drs_plan* plan, const void* packed, const float* x, const int64_t* t,
                                const float* lr_img, float* out, void* workspace, size_t workspace_bytes, int flags,
                                drs_stream_t stream) {
  return drs_unet_forward_labels(plan, packed, x, t, lr_img, nullptr, 0, out, workspace, workspace_bytes, flags, stream);
}

extern "C" int drs_unet_forward_labels(drs_plan* plan, const void* packed, const float* x, const int64_t* t,
                                       const float* lr_img, const int64_t* labels, int label_batch, float* out,
                                       void* workspace, size_t workspace_bytes, int flags, drs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  DRS_REQUIRE(plan && packed && x && t && out && workspace, DRS_ERR_ARG, "forward: null pointer");
  DRS_REQUIRE(plan->packed_ok && plan->packed_ptr == packed, DRS_ERR_STATE,
              "forward: weights not packed into this buffer (call drs_unet_pack_weights first)");
  DRS_REQUIRE(workspace_bytes >= plan->ws_bytes, DRS_ERR_WORKSPACE, "forward: workspace %zu < %zu", workspace_bytes,
              plan->ws_bytes);
  const bool reuse_cond = (flags & DRS_FWD_REUSE_COND) != 0;
  const bool has_cond = plan->cfg.variant != DRS_VARIANT_GENERATION;
  DRS_REQUIRE(!has_cond || reuse_cond || lr_img, DRS_ERR_ARG, "forward: conditioning image is null");
  DRS_REQUIRE(!labels || (plan->label_emb >= 0 && (label_batch == plan->cfg.batch || label_batch == 1)), DRS_ERR_ARG,
              "forward: labels need the generation variant with num_classes > 0 and label_batch == batch or 1");
  const drs_unet_config& c = plan->cfg;
  const int B = c.batch, Bl = c.lr_batch, C = c.image_channels, H = c.height, W = c.width, mag = c.magnification;
  const int CC = c.cond_channels;
  const int h = H / mag, w = W / mag;
  char* pk = aligned_base(packed);
  void* ws = aligned_base(workspace);
  auto PW = [&](const ConvLayer& L) { return (const float*)(pk + L.w_off); };
  auto PB = [&](const ConvLayer& L) { return (const float*)(pk + L.b_off); };
  auto TP = [&](int i) { return plan->tp(ws, i); };
  const bool train = (c.flags & DRS_PLAN_TRAIN) != 0;
  const int sp = plan->sp ? 1 : 0;  // SP-format activations (eval, split-bf16)
  const void* zero_line = pk + plan->o_zero;
  plan->fault_ptr = (unsigned*)(pk + plan->o_fault);
  // A convolution followed by BatchNorm.  Eval: BatchNorm is folded into the weights, one launch.  Train: the raw
  // convolution writes Z (input add and gate act before the norm and stay in the conv), then batch statistics,
  // running-stat update and the normalisation carry the rest of the block's epilogue.
  auto conv_bn = [&](const ConvLayer& L, const TapConv& d) -> int {
    if (!train) return plan_conv(plan, L, d, s);
    TapConv zc = d;
    zc.out = plan->tp(ws, L.t_Z); zc.out_cs = L.Cout; zc.out_co = 0;
    zc.relu_pre = zc.relu_post = 0; zc.post_add = nullptr; zc.res = nullptr;
    zc.in2 = nullptr; zc.w2 = nullptr; zc.bias2 = nullptr;
    int r = plan_conv(plan, L, zc, s);
    if (r) return r;
    float* stats = (float*)((char*)ws + L.stats_off);
    const long long ppi = (long long)d.OH * d.OW;
    return drs_launch_bn_train(zc.out, L.Cout, 0, (long long)d.N * ppi, ppi, L.Cout,
                               (const float*)plan->param_ptrs[L.bn], (const float*)plan->param_ptrs[L.bn + 1],
                               (float*)plan->param_ptrs[L.bn + 2], (float*)plan->param_ptrs[L.bn + 3], c.bn_eps, 0.1f,
                               (double*)((char*)ws + plan->o_red), stats, stats + L.Cout, d.post_add, d.post_cs, d.res,
                               d.res_cs, d.res_co, d.out, d.out_cs, d.out_co, d.relu_pre, d.relu_post, s);
  };
  int rc;
  if (plan->profiling) {
    for (auto& r : plan->ops) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    plan->ops.clear();
  }
  LaunchLogScope launch_log(plan);
#define RUN(expr) do { if ((rc = (expr))) return rc; } while (0)

  // --- second stream (eval plans): see the decoder section ---
  // DRS_CONCURRENT: 1 / 0 force the two-stream decoder stages on / off.  Default: on for the fp32-activation plans (their
  // kernels run two blocks per CU and leave room for a partner); off for SP plans, whose wave-specialised kernels own a
  // whole CU (154 KB of LDS, 12 waves) and whose attention gate is one fused launch: measured 498 vs 469 steps/s.
  static const int concurrent_env = getenv("DRS_CONCURRENT") ? atoi(getenv("DRS_CONCURRENT")) : -1;
  const bool concurrent = (concurrent_env < 0 ? !plan->sp : concurrent_env != 0) && !train && !plan->profiling &&
                          c.impl != DRS_IMPL_DIRECT;
  // the time MLPs (one small latency-bound launch) run next to conv0 in every eval plan unless DRS_CONCURRENT=0
  const bool mlp_side = concurrent_env != 0 && !train && !plan->profiling && c.impl != DRS_IMPL_DIRECT;
  if ((concurrent || mlp_side) && !plan->side) {
    DRS_CHECK_HIP(hipStreamCreateWithFlags(&plan->side, hipStreamNonBlocking));
    DRS_CHECK_HIP(hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming));
    DRS_CHECK_HIP(hipEventCreateWithFlags(&plan->ev_join, hipEventDisableTiming));
    DRS_CHECK_HIP(hipEventCreateWithFlags(&plan->ev_gbias, hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) {
      DRS_CHECK_HIP(hipEventCreateWithFlags(&plan->ev_edge_in[i], hipEventDisableTiming));
      DRS_CHECK_HIP(hipEventCreateWithFlags(&plan->ev_edge_out[i], hipEventDisableTiming));
    }
  }

  // --- time embeddings for the 7 blocks (reference :338-339 + every time_mlp) ---
  float* temb = (float*)((char*)ws + plan->o_temb);
  const float* inv_freq = (const float*)(pk + plan->o_inv_freq);
  // (eval: on the side stream, next to the conditioning branch / conv0; the first consumer is block 0's conv1)
  hipStream_t st_mlp = s;
  if (mlp_side) {
    st_mlp = plan->side;
    DRS_CHECK_HIP(hipEventRecord(plan->ev_fork, s));  // t / labels were produced on the caller's stream
    DRS_CHECK_HIP(hipStreamWaitEvent(st_mlp, plan->ev_fork, 0));
  }
  prof_begin(plan, "time_mlp", 0, 0, s);
  RUN(drs_launch_time_mlp_multi(t, inv_freq, pk, (const long long*)(pk + plan->o_mlp_table), (int)plan->mlps.size(), 256,
                                temb, plan->temb_total, B, 100, labels ? (const float*)(pk + plan->o_label) : nullptr,
                                (const long long*)labels, label_batch, plan->cfg.num_classes, st_mlp));
  prof_end(plan, s);
  // Stage inputs that are stored only as x + relu(time_mlp(t)) (below): the fused gate takes the row vector out again
  // through a per-image bias, b'[n] = b - Wg temb[n] (16 x Ch dot products per stage, next to the time MLPs).
  if (mlp_side) DRS_CHECK_HIP(hipEventRecord(plan->ev_join, st_mlp));  // (the encoder only needs the embeddings; the bias tables are for the decoder's gates)
  bool xt_only[3] = {false, false, false};
  if (plan->sp && !train && !(c.flags & DRS_PLAN_KEEP_ALL)) {
    static const int xt_env = getenv("DRS_XT_ONLY") ? atoi(getenv("DRS_XT_ONLY")) : 1;
    for (int i = 0; i < 3 && xt_env; ++i) {
      const DecStage& st = plan->dec[i];
      // producer: the bottleneck's conv2 on the wave-specialised SP kernel (16-row patches), or the composite kernel of stage i - 1
      const bool producer = i == 0 ? (H >> 3) > 8 && (W >> 3) > 8 : plan->dec[i - 1].upfuse;
      xt_only[i] = st.fused_gate && producer;
      // (stage 0: the bottleneck's conv2 may still decline the second output at its own probe below and clear xt_only[0]; the
      //  bias table launched here is then one unused 5 us side-stream launch, not an error)
      if (xt_only[i])
        RUN(drs_launch_gate_bias((const float*)(pk + st.gf_w_off), (const float*)(pk + st.gf_b_off), temb + st.mlp.temb_off,
                                 plan->temb_total, (float*)((char*)ws + st.o_gbias), B, kUp[i], kUp[i + 1], st_mlp));
    }
  }
  if (mlp_side) DRS_CHECK_HIP(hipEventRecord(plan->ev_gbias, st_mlp));

  // --- LR conditioning branch: RRDB -> bicubic -> conv (reference :345-353), constant per sampling chain ---
  if (has_cond && !reuse_cond) {
    prof_begin(plan, "lr_branch", 2.0 * Bl * (7.0 * h * w * CC * CC * 9 + (double)H * W * CC * kDown[0] * 9),
               4.0 * Bl * (15.0 * h * w * CC + (double)H * W * (2 * CC + kDown[0])), s);
    float* a = (float*)((char*)ws + plan->o_lr[0]);
    float* b = (float*)((char*)ws + plan->o_lr[1]);
    float* r = (float*)((char*)ws + plan->o_lr[2]);
    const float* cur = lr_img;
    if (train) RUN(drs_launch_nchw_to_nhwc(lr_img, TP(plan->t_rn[0]), Bl, CC, h, w, CC, 0, s));
    for (int i = 0; i < 3; ++i) {
      const PlanarConv& c1 = plan->rrdb[2 * i];
      const PlanarConv& c2 = plan->rrdb[2 * i + 1];
      RUN(drs_launch_conv3x3_planar(cur, (const float*)(pk + c1.w_off), (const float*)(pk + c1.b_off), nullptr, a, Bl,
                                    CC, CC, h, w, 1, s));
      float* dst = (cur == b) ? r : b;  // ping-pong so the residual source stays intact
      RUN(drs_launch_conv3x3_planar(a, (const float*)(pk + c2.w_off), (const float*)(pk + c2.b_off), cur, dst, Bl, CC,
                                    CC, h, w, 0, s));
      cur = dst;
      if (train) {  // the backward pass reads a_i (ReLU output) and r_{i+1} channels-last
        RUN(drs_launch_nchw_to_nhwc(a, TP(plan->t_an[i]), Bl, CC, h, w, CC, 0, s));
        RUN(drs_launch_nchw_to_nhwc(dst, TP(plan->t_rn[i + 1]), Bl, CC, h, w, CC, 0, s));
      }
    }
    const PlanarConv& co = plan->rrdb[6];
    RUN(drs_launch_conv3x3_planar(cur, (const float*)(pk + co.w_off), (const float*)(pk + co.b_off), lr_img,
                                  TP(plan->t_lrenc), Bl, CC, CC, h, w, 0, s));
    const float* upsrc = TP(plan->t_lrenc);  // SAR variant: the encoded image is used at its own resolution
    if (mag > 1 || train) {
      RUN(drs_launch_bicubic(TP(plan->t_lrenc), TP(plan->t_up), Bl, CC, h, w, mag, s));
      upsrc = TP(plan->t_up);
    }
    RUN(drs_launch_stem(upsrc, (const float*)(pk + plan->stemc.w_off), (const float*)(pk + plan->stemc.b_off),
                        nullptr, 0, TP(plan->t_cond), Bl, CC, kDown[0], H, W, s));
    prof_end(plan, s);
  }
  // --- x = conv0(x) + cond (reference :342,:355) ---
  prof_begin(plan, "conv0", 2.0 * B * H * W * C * kDown[0] * 9, 4.0 * B * H * W * (C + 2.0 * kDown[0]), s);
  RUN(drs_launch_stem(x, (const float*)(pk + plan->stem0.w_off), (const float*)(pk + plan->stem0.b_off),
                      has_cond ? TP(plan->t_cond) : nullptr, Bl, TP(plan->t_x0), B, C, kDown[0], H, W, s, plan->sp ? 1 : 0));
  prof_end(plan, s);

  if (mlp_side) DRS_CHECK_HIP(hipStreamWaitEvent(s, plan->ev_join, 0));  // time embeddings are ready

  // --- encoder + bottleneck: ResConvBlock (reference :153-172), downs (:366) ---
  const float* xin = TP(plan->t_x0);
  for (int i = 0; i < 4; ++i) {
    const ResBlock& rb = plan->enc[i];
    const int ci = kDown[i], co = kDown[i + 1], hh = H >> i, ww = W >> i;
    // shortcut_conv + BN (1x1) rides inside conv2 as extra K-chunks when both run on the MFMA family ("K-concat")
    const bool fuse_shortcut = !train && rb.conv2.mfma && rb.shortcut.mfma;
    if (!fuse_shortcut) {  // shortcut = BNs(conv1x1(x))
      TapConv d = conv_desc(xin, B, hh, ww, ci, ci, 0, PW(rb.shortcut), PB(rb.shortcut), TP(plan->t_S[i]), co, co, 0, 1,
                            1, 1, 0);
      d.in_sp = sp; d.out_sp = rb.shortcut.out_sp ? 1 : 0;
      RUN(conv_bn(rb.shortcut, d));
    }
    bool dual = false;
    // Block 0 (16 -> 32 -> 32 channels at full resolution) as ONE launch: h stays in LDS (resblock0_sp.hip)
    if (i == 0 && rb.dual && fuse_shortcut && sp && !(c.flags & DRS_PLAN_KEEP_ALL) && c.impl == DRS_IMPL_MFMA_BF16X3 &&
        drs_resblock0_supported(ci, co, hh, ww)) {
      ResBlock0Desc r0 = {};
      r0.x = xin;
      r0.w1 = pk + rb.dual_w_off; r0.b1 = (const float*)(pk + rb.dual_b_off);
      r0.temb = temb + rb.mlp.temb_off; r0.temb_cs = plan->temb_total;
      r0.w2 = PW(rb.conv2); r0.b2 = PB(rb.conv2);
      r0.ws = PW(rb.shortcut); r0.bs = PB(rb.shortcut);
      r0.out = TP(plan->t_R[0]);
      r0.N = B; r0.H = hh; r0.W = ww;
      r0.zero_line = zero_line; r0.fault = plan->fault_ptr;
      const double px0 = (double)B * hh * ww;
      prof_begin(plan, "conv_blocks.0.fused", 2.0 * px0 * (2.0 * 9 * ci * co + 9.0 * co * co + (double)ci * co),
                 4.0 * px0 * (ci + co) + 4.0 * (2.0 * 9 * ci * co + 9.0 * co * co + (double)ci * co), s);
      rc = drs_launch_resblock0(r0, s);
      prof_end(plan, s);
      if (rc) return rc;
      TapConv d = conv_desc(TP(plan->t_R[0]), B, hh, ww, co, co, 0, PW(plan->downs[0]), PB(plan->downs[0]), TP(plan->t_D[0]), co,
                            co, 0, 3, 3, 2, 1);
      d.in_sp = d.out_sp = sp; d.zero_line = zero_line;
      RUN(plan_conv(plan, plan->downs[0], d, s));
      xin = TP(plan->t_D[0]);
      continue;
    }
    if (rb.dual && !train && !(c.flags & DRS_PLAN_KEEP_ALL)) {
      // h = relu(BN1(conv1(x))) + skip(x) + relu(time_mlp(t)) in ONE launch: the skip tensor never exists in HBM
      TapConv d = conv_desc(xin, B, hh, ww, ci, ci, 0, (const float*)(pk + rb.dual_w_off), (const float*)(pk + rb.dual_b_off),
                            TP(plan->t_H[i]), co, co, 0, 3, 3, 1, 1);
      d.dual = 1;
      d.relu_pre = 1;
      d.in_sp = d.out_sp = sp; d.zero_line = zero_line; d.fault = plan->fault_ptr;
      d.post_add = temb + rb.mlp.temb_off; d.post_cs = plan->temb_total;
      if (drs_tapconv_ws_supported(d, c.impl) || drs_tapconv_sp_supported(d, c.impl)) {
        const std::string& wn = plan->params[rb.conv1.w].name;
        prof_begin(plan, wn.substr(0, wn.size() - 7) + "+skip", 2.0 * conv_flops(d), conv_bytes(d), s);
        rc = drs_launch_tapconv_mfma(d, c.impl, s);
        prof_end(plan, s);
        if (rc) return rc;
        dual = true;
      }
    }
    if (rb.has_skip && !dual) {  // conv_upsampled_lr_img(x_skip), x_skip == block input
      TapConv d = conv_desc(xin, B, hh, ww, ci, ci, 0, PW(rb.skip), PB(rb.skip), TP(plan->t_K0), co, co, 0, 3, 3, 1, 1);
      d.in_sp = sp; d.zero_line = zero_line;
      RUN(plan_conv(plan, rb.skip, d, s));
    }
    if (!dual) {  // h = relu(BN1(conv1(x))) [+ skip] + relu(time_mlp(t))
      TapConv d = conv_desc(xin, B, hh, ww, ci, ci, 0, PW(rb.conv1), PB(rb.conv1), TP(plan->t_H[i]), co, co, 0, 3, 3, 1,
                            1);
      d.relu_pre = 1;
      d.in_sp = d.out_sp = sp; d.zero_line = zero_line;
      d.post_add = temb + rb.mlp.temb_off; d.post_cs = plan->temb_total;
      if (rb.has_skip) { d.res = TP(plan->t_K0); d.res_cs = co; d.res_co = 0; }
      RUN(conv_bn(rb.conv1, d));
    }
    {  // out = relu(shortcut + BN2(conv2(h)))
      TapConv d = conv_desc(TP(plan->t_H[i]), B, hh, ww, co, co, 0, PW(rb.conv2), PB(rb.conv2), TP(plan->t_R[i]), co,
                            co, 0, 3, 3, 1, 1);
      if (fuse_shortcut) {
        d.in2 = xin; d.in2_cs = ci; d.in2_co = 0; d.Cin2 = ci; d.H2 = hh; d.W2 = ww;
        d.w2 = PW(rb.shortcut); d.bias2 = PB(rb.shortcut);
        d.in2_sp = sp;
        if (rb.shortcut.fl_ok && !plan->fl_off) d.w2_fl = pk + rb.shortcut.fl_off;
      } else {
        d.res = TP(plan->t_S[i]); d.res_cs = co; d.res_co = 0; d.res_sp = rb.shortcut.out_sp ? 1 : 0;
      }
      d.relu_post = 1;
      d.in_sp = d.out_sp = sp; d.zero_line = zero_line;
      if (sp && i == 3) {  // second output: x + relu(time_mlp(t)) of the first UpConvBlock (its conv then needs no input add)
        d.out2 = TP(plan->t_XT[0]); d.out2_cs = co; d.out2_co = 0;
        d.post2 = temb + plan->dec[0].mlp.temb_off; d.post2_cs = plan->temb_total;
        if (xt_only[0]) {  // both readers of the bottleneck output take x + temb: the plain copy is not written
          TapConv probe = d;
          probe.out = nullptr;
          if (drs_tapconv_sp_supported(probe, c.impl)) d.out = nullptr;
          else xt_only[0] = false;
        }
      }
      RUN(conv_bn(rb.conv2, d));
    }
    if (i < 3) {
      TapConv d = conv_desc(TP(plan->t_R[i]), B, hh, ww, co, co, 0, PW(plan->downs[i]), PB(plan->downs[i]),
                            TP(plan->t_D[i]), co, co, 0, 3, 3, 2, 1);
      d.in_sp = d.out_sp = sp; d.zero_line = zero_line;
      RUN(plan_conv(plan, plan->downs[i], d, s));
      xin = TP(plan->t_D[i]);
    }
  }

  if (mlp_side) DRS_CHECK_HIP(hipStreamWaitEvent(s, plan->ev_gbias, 0));  // per-image gating biases (side stream) are ready
  // --- decoder (reference :372-377) ---
  // Eval plans run the attention branch of a stage (gating, w_g, w_x, psi, result: HBM-bound 1x1 / 2x2 kernels) on a
  // second stream NEXT TO the up-sampling branch (3x3 conv + ConvTranspose: MFMA / LDS-bound): both only read the stage
  // input and the skip tensor and write disjoint channel slices of cat.i.  Every kernel of the pair is launched with
  // one block per CU, so a block of each fits on every CU at once (2 x 80 KB of LDS) and the two use complementary
  // resources.  Train plans and profiled runs keep the serial order.
  const float* xcur = TP(plan->t_R[3]);
  bool fused_output = false;
  for (int i = 0; i < 3; ++i) {
    const DecStage& st = plan->dec[i];
    const int Cc = kUp[i], Ch = kUp[i + 1];
    const int lh = H >> (3 - i), lw = W >> (3 - i);
    const float* xres = TP(plan->t_R[2 - i]);  // residual_inputs[-(i+1)]: (B, Ch, 2lh, 2lw)
    float* cat = TP(plan->t_CAT[i]);
    hipStream_t sa = s;  // stream of the attention branch
    if (concurrent) {
      sa = plan->side;
      DRS_CHECK_HIP(hipEventRecord(plan->ev_fork, s));
      DRS_CHECK_HIP(hipStreamWaitEvent(sa, plan->ev_fork, 0));
    }
    auto att_conv = [&](const ConvLayer& L, TapConv d) -> int {  // attention-branch op (eval: BN folded)
      if (!concurrent) return conv_bn(L, d);
      d.shared_cu = 1;
      return plan_conv(plan, L, d, sa);
    };
    auto att_plain = [&](const ConvLayer& L, TapConv d) -> int {
      if (concurrent) d.shared_cu = 1;
      return plan_conv(plan, L, d, sa);
    };
    auto attention_branch = [&]() -> int {
      int rc = DRS_OK;
    const bool fuse_gate = st.fused_gate && !(c.flags & DRS_PLAN_KEEP_ALL);
    if (fuse_gate) {
      // gating signal + attention gate in ONE launch (attn_gate_sp.hip): g, g1, p and psi never reach HBM
      AttnGateDesc a = {};
      a.x = xt_only[i] ? TP(plan->t_XT[i]) : xcur; a.x_cs = Cc; a.x_co = 0;
      a.b_gate_img = xt_only[i] ? (const float*)((char*)ws + st.o_gbias) : nullptr;
      a.xres = xres; a.r_cs = Ch; a.r_co = 0;
      a.out = cat; a.out_cs = Cc + Ch; a.out_co = Cc;
      a.psi_out = nullptr;
      if (st.gate_psi) { a.out = nullptr; a.psi_out = TP(plan->t_PSI[i]); }  // (the att-half multiplies by psi itself: below)
      a.N = B; a.LH = lh; a.LW = lw; a.Cc = Cc; a.Ch = Ch;
      a.w_gate = PW(st.gate); a.b_gate = PB(st.gate);
      a.w_wg = pk + st.fz_wg_off; a.b_wg = PB(st.wg);
      a.w_wx = pk + st.fz_wx_off; a.b_wx = PB(st.wx);
      a.w_psi = PW(st.psi); a.b_psi = PB(st.psi);
      a.w_res = PW(st.result); a.b_res = PB(st.result);
      const double px = (double)B * lh * lw;
      prof_begin(plan, "attention_gate." + std::to_string(i), 2.0 * px * Ch * (Cc + 10.0 * Ch),
                 4.0 * px * (Cc + 8.0 * Ch), sa);
      rc = drs_launch_attn_gate(a, sa);
      prof_end(plan, sa);
      if (rc) return rc;
      if (concurrent) DRS_CHECK_HIP(hipEventRecord(plan->ev_join, sa));
    } else {
    {  // gating = relu(BN(conv1x1(x)))   (:222-225)
      TapConv d = conv_desc(xcur, B, lh, lw, Cc, Cc, 0, PW(st.gate), PB(st.gate), TP(plan->t_G[i]), Ch, Ch, 0, 1, 1, 1,
                            0);
      d.relu_pre = 1;
      d.in_sp = d.out_sp = sp;
      RUN(att_conv(st.gate, d));
    }
    // (fusing w_g into the stride-2 w_x kernel was measured slower: its 16x32 window staging is 4x too large for g;
    //  running w_x early on the side stream, next to the encoder, slowed the encoder kernels more than it saved)
    {
      {  // g1 = w_g(g)   (:101)
        TapConv d = conv_desc(TP(plan->t_G[i]), B, lh, lw, Ch, Ch, 0, PW(st.wg), PB(st.wg), TP(plan->t_Q[i]), Ch, Ch, 0, 1,
                              1, 1, 0);
        d.in_sp = sp;
        RUN(att_plain(st.wg, d));
      }
      {  // relu(g1 + w_x(x))   (:102-103)
        TapConv d = conv_desc(xres, B, 2 * lh, 2 * lw, Ch, Ch, 0, PW(st.wx), PB(st.wx), TP(plan->t_P[i]), Ch, Ch, 0, 2, 2,
                              2, 0);
        d.res = TP(plan->t_Q[i]); d.res_cs = Ch; d.res_co = 0;
        d.relu_post = 1;
        d.in_sp = sp;
        RUN(att_plain(st.wx, d));
      }
    }
    {  // psi = sigmoid(conv1x1 -> 1 channel)   (:104)
      TapConv d = conv_desc(TP(plan->t_P[i]), B, lh, lw, Ch, Ch, 0, PW(st.psi), PB(st.psi), TP(plan->t_PSI[i]), 1, 1, 0,
                            1, 1, 1, 0);
      d.sigmoid = 1;
      RUN(att_plain(st.psi, d));
    }
    {  // attention = BN(conv1x1(nearest2x(psi) * x))  == nearest2x(psi) * (W' x) + b'   (:105-107), into cat[:, Cc:]
      TapConv d = conv_desc(xres, B, 2 * lh, 2 * lw, Ch, Ch, 0, PW(st.result), PB(st.result), cat, Ch, Cc + Ch, Cc, 1, 1,
                            1, 0);
      d.gate = TP(plan->t_PSI[i]);
      d.in_sp = d.out_sp = sp;
      RUN(att_conv(st.result, d));
      if (concurrent) DRS_CHECK_HIP(hipEventRecord(plan->ev_join, sa));
    }
    }  // !fuse_gate
      return rc;
    };
    auto upconv_block = [&]() -> int {
      int rc = DRS_OK;
    {  // UpConvBlock: relu(BN(conv(x + relu(time_mlp(t)))))   (:199-205)
      TapConv d = conv_desc(xcur, B, lh, lw, Cc, Cc, 0, PW(st.conv), PB(st.conv), TP(plan->t_U[i]), Cc, Cc, 0, 3, 3, 1,
                            1);
      d.relu_pre = 1;
      if (sp) {  // the producer of the stage input also wrote x + relu(time_mlp(t)) (TapConv::out2)
        d.in = TP(plan->t_XT[i]);
        d.in_sp = d.out_sp = 1; d.zero_line = zero_line;
      } else {
        d.in_add = temb + st.mlp.temb_off; d.in_add_cs = plan->temb_total;
      }
      d.shared_cu = concurrent ? 1 : 0;
      RUN(conv_bn(st.conv, d));
    }
      return rc;
    };
    // Order inside a stage.  A composite stage whose plan owns a side stream computes its edge vectors THERE, next to
    // the attention gate: they only need ups.i.conv's output, are three tiny launches' worth of latency (35 us per
    // forward on the main stream) and occupy a fraction of the CUs.  So: UpConvBlock conv, [edges || gate], att-half,
    // composite.  Everything else keeps the reference's order (gate first).
    const bool edges_aside = st.upfuse && mlp_side && !concurrent;
    if (edges_aside) {
      RUN(upconv_block());
      DRS_CHECK_HIP(hipEventRecord(plan->ev_edge_in[i], s));
      DRS_CHECK_HIP(hipStreamWaitEvent(plan->side, plan->ev_edge_in[i], 0));
      RUN(attention_branch());
    } else {
      RUN(attention_branch());
      RUN(upconv_block());
    }
    if (st.upfuse) {
      // ups.i.transform and the x-half of up_convs.i as ONE stride-2 transposed convolution of ups.i.conv's output
      // (upfuse_sp.hip; reference :206-207 returns transform(x) with no activation, :377 is a bare convolution): the
      // Cc-channel high-resolution tensor is never written.  Three launches: the edge vectors (first row / column of h),
      // the att-half of up_convs.i as a plain 3x3 convolution of the attention output, and the composite with the att-half
      // as its residual (+ the fused `output` projection in stage 2).
      const float* aux = (const float*)(pk + st.uf_aux_off);
      const size_t mat = (size_t)Cc * Ch;
      float* eh = (float*)((char*)ws + st.o_eh);
      float* ev = (float*)((char*)ws + st.o_ev);
      {
        UpFuseEdgeDesc e = {};
        e.in = TP(plan->t_U[i]); e.in_cs = Cc; e.in_co = 0;
        e.N = B; e.LH = lh; e.LW = lw; e.Cc = Cc; e.Ch = Ch;
        e.rt = aux; e.rl = aux + 5 * mat; e.bt = aux + 11 * mat;
        e.eh = eh; e.ev = ev;
        e.wimg = pk + st.uf_edge_off; e.zero_line = zero_line;
        const double epix = (double)B * 2.0 * (lh + lw);
        hipStream_t se = edges_aside ? plan->side : s;  // (profiled forwards have no side stream: edges_aside is false there)
        prof_begin(plan, "up_convs." + std::to_string(i) + ".edges", 2.0 * epix * 2.5 * Cc * Ch, 4.0 * epix * (Cc + 4.0 * Ch), se);
        rc = drs_launch_upfuse_edges(e, se);
        prof_end(plan, se);
        if (rc) return rc;
        if (edges_aside) DRS_CHECK_HIP(hipEventRecord(plan->ev_edge_out[i], plan->side));
      }
      if (concurrent) DRS_CHECK_HIP(hipStreamWaitEvent(s, plan->ev_join, 0));  // the attention half of cat.i is complete
      {
        TapConv d = conv_desc(cat, B, 2 * lh, 2 * lw, Ch, Cc + Ch, Cc, (const float*)(pk + st.ah_w_off),
                              (const float*)(pk + st.ah_b_off), i < 2 ? TP(st.t_PA) : nullptr, Ch, Ch, 0, 3, 3, 1, 1);
        d.in_sp = 1; d.out_sp = 1; d.zero_line = zero_line; d.fault = plan->fault_ptr;
        if (i < 2 && st.ah_fl_ok && !plan->fl_off) d.w_fl = pk + st.ah_fl_off;
        const double ah_flops = conv_flops(d), ah_bytes = conv_bytes(d);  // (the reference's op, whatever form runs)
        if (i == 2 && st.ah_proj) {
          // projection folded into the weights (pack time): a Ch -> out_dim 3x3 convolution straight into the caller's tensor
          d.out = nullptr; d.out_sp = 0; d.bias = nullptr;
          d.Cout = 16; d.out_cs = 16;
          d.proj = 1;
          if (st.gate_psi) {  // `result` folded in: the input is the skip tensor, gated by psi inside the kernel
            d.in = xres; d.in_cs = Ch; d.in_co = 0;
            d.gate = TP(plan->t_PSI[i]);
            d.bias = (const float*)(pk + st.ah_tab_off);
          }
          d.fuse_out = out; d.fuse_dim = c.out_dim; d.fuse_b = nullptr;
          prof_begin(plan, "up_convs.2.att", ah_flops, ah_bytes, s);
          rc = drs_launch_conv3x3_direct_sp(d, s);
          prof_end(plan, s);
          if (rc) return rc;
        } else {
        if (i == 2) {
          // the `output` projection is linear: the att-half is projected HERE (its own fused-projection epilogue, zero bias)
          // into the caller's output tensor and the composite kernel adds its part: 12.6 MB written and read back instead
          // of the 134 MB of 32-channel partial sums
          d.out = nullptr; d.out_sp = 0;
          d.fuse_w = (const float*)(pk + plan->o_out_w);
          d.fuse_b = (const float*)(pk + st.ah_b_off);  // zeros
          d.fuse_out = out;
          d.fuse_dim = c.out_dim;
        }
        prof_begin(plan, "up_convs." + std::to_string(i) + ".att", ah_flops, ah_bytes, s);
        rc = drs_launch_tapconv_mfma(d, c.impl, s);
        prof_end(plan, s);
        if (rc) return rc;
        }
      }
      {
        UpFuseDesc u = {};
        u.in = TP(plan->t_U[i]); u.in_cs = Cc; u.in_co = 0;
        u.N = B; u.LH = lh; u.LW = lw; u.Cc = Cc; u.Ch = Ch;
        u.w = pk + st.uf_w_off;
        u.bias = aux + 11 * mat + 9 * Ch;
        if (i < 2) { u.res = TP(st.t_PA); u.res_cs = Ch; u.res_co = 0; }
        u.eh = eh; u.ev = ev;
        u.zero_line = zero_line; u.fault = plan->fault_ptr;
        if (i == 2) {  // output 1x1 conv (:379) rides in the epilogue; the 32-channel tensor is never written
          u.res = nullptr; u.fuse_acc = 1;
          if (st.uf_proj) {
            u.proj = 1;  // the projection (and its bias) is inside the composite weights / bias / edge vectors
          } else {
            u.fuse_w = (const float*)(pk + plan->o_out_w);
            u.fuse_b = (const float*)(pk + plan->o_out_b);
          }
          u.fuse_out = out;
          u.fuse_dim = c.out_dim;
          fused_output = true;
        } else {
          if (!xt_only[i + 1]) { u.out = TP(plan->t_X[i]); u.out_cs = Ch; u.out_co = 0; }
          u.out2 = TP(plan->t_XT[i + 1]); u.out2_cs = Ch; u.out2_co = 0;  // x + temb of the next stage's UpConvBlock
          u.post2 = temb + plan->dec[i + 1].mlp.temb_off; u.post2_cs = plan->temb_total;
        }
        if (edges_aside) DRS_CHECK_HIP(hipStreamWaitEvent(s, plan->ev_edge_out[i], 0));
        const double opix = (double)B * 4.0 * lh * lw;
        // executed work: 6.25 composite taps per output pixel; bytes: h + att-half partial sums + result (+ weights)
        prof_begin(plan, "up_convs." + std::to_string(i) + ".fused", 2.0 * opix * 6.25 * Cc * Ch,
                   4.0 * (opix / 4.0 * Cc + 2.0 * opix * Ch + 25.0 * Cc * Ch), s);
        if (i == 2 && st.uf_stream) {
          u.w = pk + st.ufp_w_off;
          rc = drs_launch_upfuse_proj(u, s);
        } else {
          rc = drs_launch_upfuse(u, s);
        }
        prof_end(plan, s);
        if (rc) return rc;
      }
    } else {
    if (st.transform.mfma) {  // transform: ConvTranspose2d, into cat[:, :Cc]   (:206, :376), 4 phases in one launch
      TapConv d = convT_fused_desc(TP(plan->t_U[i]), B, lh, lw, Cc, Cc, 0, PW(st.transform), PB(st.transform), cat, Cc,
                                   Cc + Ch, 0);
      d.shared_cu = concurrent ? 1 : 0;
      d.in_sp = d.out_sp = sp; d.zero_line = zero_line;
      RUN(plan_conv(plan, st.transform, d, s));
    } else
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        TapConv d = convT_phase_desc(TP(plan->t_U[i]), B, lh, lw, Cc, Cc, 0, PW(st.transform), PB(st.transform), cat, Cc,
                                     Cc + Ch, 0, py, px);
        RUN(plan_conv(plan, st.transform, d, s));
      }
    if (concurrent) DRS_CHECK_HIP(hipStreamWaitEvent(s, plan->ev_join, 0));  // both halves of cat.i are complete
    {  // up_conv over the concatenation (:377), no norm / activation
      TapConv d = conv_desc(cat, B, 2 * lh, 2 * lw, Cc + Ch, Cc + Ch, 0, PW(st.upconv), PB(st.upconv), TP(plan->t_X[i]),
                            Ch, Ch, 0, 3, 3, 1, 1);
      if (i == 2 && st.upconv.mfma && c.out_dim <= 4) {  // output 1x1 conv (:379) rides in the epilogue
        d.fuse_w = (const float*)(pk + plan->o_out_w);
        d.fuse_b = (const float*)(pk + plan->o_out_b);
        d.fuse_out = out;
        d.fuse_dim = c.out_dim;
        if (!(c.flags & (DRS_PLAN_KEEP_ALL | DRS_PLAN_TRAIN))) d.out = nullptr;  // the wide tensor is only a parity tap
        fused_output = true;
      }
      d.in_sp = sp; d.zero_line = zero_line;
      d.out_sp = st.upconv.out_sp ? 1 : 0;
      if (sp && i < 2) {  // second output for the next stage's UpConvBlock
        d.out2 = TP(plan->t_XT[i + 1]); d.out2_cs = Ch; d.out2_co = 0;
        d.post2 = temb + plan->dec[i + 1].mlp.temb_off; d.post2_cs = plan->temb_total;
      }
      RUN(plan_conv(plan, st.upconv, d, s));
    }
    }  // !upfuse
    xcur = TP(plan->t_X[i]);
  }
  if (!fused_output) {  // output 1x1 conv (:379), straight to the caller's NCHW tensor
    TapConv d = conv_desc(xcur, B, H, W, kUp[3], kUp[3], 0, PW(plan->output), PB(plan->output), out, c.out_dim, c.out_dim,
                          0, 1, 1, 1, 0);
    d.out_nchw = 1;
    RUN(plan_conv(plan, plan->output, d, s));
  }
#undef RUN
  return DRS_OK;
}

// Synchronises `stream` and reports whether a wave of the wave-specialised kernels ran into its bounded poll since the
// weights were last packed into `packed` (a protocol bug: the forward's output is then incomplete).
extern "C" int drs_unet_check_faults(drs_plan* plan, const void* packed, drs_stream_t stream) {
  DRS_REQUIRE(plan && packed, DRS_ERR_ARG, "check_faults: null pointer");
  DRS_REQUIRE(plan->packed_ok && plan->packed_ptr == packed, DRS_ERR_STATE, "check_faults: weights not packed into this buffer");
  unsigned words[5] = {0, 0, 0, 0, 0};
  DRS_CHECK_HIP(hipMemcpyAsync(words, aligned_base(packed) + plan->o_fault, sizeof(words), hipMemcpyDeviceToHost, (hipStream_t)stream));
  DRS_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  const unsigned word = words[0];
  DRS_REQUIRE((word & 1u) == 0, DRS_ERR_HIP, "a wave-specialised kernel timed out on an LDS counter (protocol fault); results are incomplete");
  if (word & 2u) {  // the FL kernel's movers met an activation block whose maximum fp16 cannot hold
    plan->fl_off = true;
    DRS_CHECK_HIP(hipMemsetAsync(aligned_base(packed) + plan->o_fault, 0, 32, (hipStream_t)stream));
    DRS_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    DrsErr::set("an activation left fp16's range in the FL arithmetic (last report: layer Cin=%u Cout=%u, %u rows, %s input, block %u "
                "round %u lane %u, block scale exponent %u): the forward(s) since the last check are invalid; this plan now runs the "
                "split-bf16 kernels - run the forward / chain again", words[1] >> 16, words[1] & 0xffffu, words[2] >> 16,
                (words[2] & 1u) ? "second (1x1)" : "3x3", words[3] >> 16, (words[3] >> 8) & 0xffu, words[3] & 0xffu, words[4]);
    return DRS_ERR_RANGE;
  }
  return DRS_OK;
}

// ------------------------------------------------------------------------------------------------
// introspection
// ------------------------------------------------------------------------------------------------
extern "C" int drs_unet_num_tensors(const drs_plan* plan) { return plan ? (int)plan->tensors.size() : 0; }
extern "C" const char* drs_unet_tensor_name(const drs_plan* plan, int i) {
  return (plan && i >= 0 && i < (int)plan->tensors.size()) ? plan->tensors[i].name.c_str() : nullptr;
}
extern "C" int drs_unet_tensor_shape(const drs_plan* plan, int i, int* n, int* c, int* h, int* w) {
  DRS_REQUIRE(plan && i >= 0 && i < (int)plan->tensors.size() && n && c && h && w, DRS_ERR_ARG, "tensor_shape: bad index");
  const WsTensor& t = plan->tensors[i];
  *n = t.n; *c = t.c; *h = t.h; *w = t.w;
  return DRS_OK;
}
extern "C" int drs_unet_read_tensor(const drs_plan* plan, int i, const void* workspace, float* dst, drs_stream_t stream) {
  DRS_REQUIRE(plan && workspace && dst && i >= 0 && i < (int)plan->tensors.size(), DRS_ERR_ARG, "read_tensor: bad args");
  const WsTensor& t = plan->tensors[i];
  const float* src = (const float*)(aligned_base(workspace) + t.off);
  if (t.planar) {
    DRS_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)t.n * t.c * t.h * t.w * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DRS_OK;
  }
  if (t.sp) return drs_launch_sp_to_nchw(src, dst, t.n, t.c, t.h, t.w, t.cs, t.co, (hipStream_t)stream);
  return drs_launch_nhwc_to_nchw(src, dst, t.n, t.c, t.h, t.w, t.cs, t.co, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// per-op timing
// ------------------------------------------------------------------------------------------------
extern "C" int drs_unet_profile_enable(drs_plan* plan, int on) {
  DRS_REQUIRE(plan, DRS_ERR_ARG, "profile_enable: null plan");
  plan->profiling = on != 0;
  return DRS_OK;
}
extern "C" int drs_unet_profile_num_ops(const drs_plan* plan) { return plan ? (int)plan->ops.size() : 0; }
extern "C" int drs_unet_profile_read(drs_plan* plan, int i, char* name, int name_len, float* ms, double* flops,
                                     double* bytes) {
  DRS_REQUIRE(plan && i >= 0 && i < (int)plan->ops.size() && name && ms && flops && bytes, DRS_ERR_ARG,
              "profile_read: bad args");
  drs_plan::OpRec& r = plan->ops[i];
  DRS_CHECK_HIP(hipEventSynchronize(r.e1));
  DRS_CHECK_HIP(hipEventElapsedTime(ms, r.e0, r.e1));
  snprintf(name, name_len, "%s", r.name.c_str());
  *flops = r.flops;
  *bytes = r.bytes;
  return DRS_OK;
}

extern "C" int drs_unet_profile_num_launches(const drs_plan* plan) { return plan ? (int)plan->launches.size() : 0; }
extern "C" int drs_unet_profile_launch(const drs_plan* plan, int i, char* op, int op_len, char* kernel, int kernel_len) {
  DRS_REQUIRE(plan && i >= 0 && i < (int)plan->launches.size() && op && kernel && op_len > 0 && kernel_len > 0, DRS_ERR_ARG,
              "profile_launch: bad args");
  snprintf(op, op_len, "%s", plan->launches[i].op.c_str());
  snprintf(kernel, kernel_len, "%s", plan->launches[i].kernel.c_str());
  return DRS_OK;
}

#include "train_bwd.inc"
