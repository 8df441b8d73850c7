/*
 * drs_hip.h — C-ABI of the MI355X (gfx950) DDPM denoising hot path.
 *
 * This is the drop-in boundary: every entry point takes raw DEVICE pointers, explicit
 * shapes and a HIP stream, and returns an int status (0 = ok).  No torch types, no
 * exceptions, no ownership transfer: the caller (PyTorch's allocator in the shipped
 * host code) owns every buffer including the workspaces passed in.  The only opaque
 * state is `drs_plan`, created and destroyed explicitly.
 *
 * Each declaration cites the reference code it replaces.  Paths are relative to the
 * reference checkout of AdrianoEttari/DiffusionRemoteSensing (2024-10-22).
 *
 * Conventions: all floating tensors are fp32, row-major contiguous; images at the
 * boundary are NCHW exactly like the reference; timesteps are int64.  Internally the
 * plan keeps activations channels-last (NHWC) in the caller's workspace.
 */
#ifndef DRS_HIP_H
#define DRS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: these entry points are its whole dynamic symbol table
 * (tests/test_abi_symbols.py checks exported == declared). */
#define DRS_API __attribute__((visibility("default")))

typedef void* drs_stream_t; /* hipStream_t; NULL = default stream */
typedef struct drs_plan drs_plan;

enum {
  DRS_OK = 0,
  DRS_ERR_ARG = 1,       /* null pointer / bad enum */
  DRS_ERR_SHAPE = 2,     /* shape the kernels do not support (e.g. H,W not divisible by 8) */
  DRS_ERR_HIP = 3,       /* a HIP runtime call failed; see drs_last_error() */
  DRS_ERR_WORKSPACE = 4, /* workspace / packed buffer too small */
  DRS_ERR_STATE = 5,     /* plan used before its weights were packed */
  DRS_ERR_RANGE = 6      /* drs_unet_check_faults: an activation left the range of the fp16 main operand of the FL arithmetic
                          * (csrc/conv_mfma_fl.hip) during a forward since the last check: that forward's result is invalid; the
                          * plan has switched itself to the split-bf16 kernels - run the forward / the chain again */
};

/* Convolution implementations (same arithmetic, different kernels). */
enum {
  DRS_IMPL_DIRECT = 0, /* fp32 VALU direct convolution: the on-device reference path */
  DRS_IMPL_MFMA_F32 = 1, /* LDS-tiled implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32) */
  DRS_IMPL_MFMA_BF16X3 = 2, /* implicit GEMM, operands split hi+lo bf16, 3 MFMAs per product */
  DRS_IMPL_MFMA_F16 = 3  /* implicit GEMM, fp16 operands, fp32 accumulate */
};

/* Threading: a drs_plan (and the buffers bound to it) is used by ONE host thread at a time - its launches, events and side
 * streams are not locked.  Different plans may be driven from different threads and on different devices; the plan-less
 * entry points (noise_images, sampler steps, adam / ema, downblur, aggregate) are re-entrant.  The library's only
 * process-wide state is a mutex-guarded per-device cache of kernel attributes, plus kernel-family switches read ONCE per
 * process from the environment (A/B experiments and the variant tests; unset = the shipped defaults): DRS_SP,
 * DRS_FL, DRS_WS, DRS_D3K, DRS_S2K, DRS_NWG, DRS_BLOCKS_PER_CU, DRS_FUSE_GATE, DRS_UPFUSE, DRS_XT_ONLY, DRS_RB0, DRS_DOWNK, DRS_SP8,
 * DRS_FOLD_PROJ, DRS_GATE_PSI, DRS_CONCURRENT, DRS_DEBUG_FLAGS, DRS_TRAIN_BWD_IMPL, DRS_TRAIN_WGRAD_IMPL, DRS_WGRAD, DRS_WGRAD_STREAM.
 * Human-readable message for the last non-zero status returned on this thread. */
DRS_API const char* drs_last_error(void);
/* ABI version of this header; bumped on any signature change. */
DRS_API int drs_abi_version(void);  /* 7 */

/* ------------------------------------------------------------------------------------------
 * Diffusion arithmetic
 * ------------------------------------------------------------------------------------------ */

/* Forward process q(x_t | x_0): x_t[i] = sqrt(alpha_hat[t[i]]) * x0[i] + sqrt(1 - alpha_hat[t[i]]) * eps[i]
 * for n images of `chw` elements each.  `eps` is supplied by the caller (torch.randn_like).
 * Replaces Diffusion.noise_images, train_diffusion_superres.py:171-190. */
DRS_API int drs_noise_images(const float* x0, const float* eps, const int64_t* t, const float* alpha_hat,
                     int noise_steps, float* x_t, int n, int64_t chw, drs_stream_t stream);

/* One ancestral sampling update, in place on x:
 *   x = 1/sqrt(alpha[t]) * (x - (1 - alpha[t]) / sqrt(1 - alpha_hat[t]) * eps_pred) + sqrt(beta[t]) * noise
 * `noise` may be NULL (last step: reference uses zeros).  `t` is one scalar timestep shared by the
 * batch (the reference builds ones(n)*i).  Replaces the loop body at train_diffusion_superres.py:240-249. */
DRS_API int drs_sampler_step(float* x, const float* eps_pred, const float* noise, int t, const float* alpha,
                     const float* alpha_hat, const float* beta, int noise_steps, int64_t numel,
                     drs_stream_t stream);

/* Same update with classifier-free guidance folded in: eps = lerp(eps_uncond, eps_cond, cfg_scale) with torch.lerp's
 * formula (weight >= 0.5: end - (end - start) * (1 - weight)), then the ancestral update above.
 * Replaces generate_new_imgs/train_diffusion_generation.py:236-249. */
DRS_API int drs_sampler_step_cfg(float* x, const float* eps_cond, const float* eps_uncond, float cfg_scale, const float* noise,
                         int t, const float* alpha, const float* alpha_hat, const float* beta, int noise_steps,
                         int64_t numel, drs_stream_t stream);

/* One Adam step over many tensors in ONE launch (torch.optim.Adam defaults: no weight decay, no amsgrad):
 *   m = lerp(m, g, 1-beta1); v = v*beta2 + (1-beta2)*g*g; p -= lr/(1-beta1^step) * m / (sqrt(v)/sqrt(1-beta2^step) + eps)
 * `table` (device): ntensors x drs_adam_tensor {p, g, m, v, n, step}; entries with g == NULL are skipped (parameters
 * that got no gradient, like torch).  `step` is that tensor's 1-based step count after the increment (torch keeps one
 * per parameter: a parameter that skipped steps has its own bias correction).
 * Replaces `optimizer.step()` of torch.optim.Adam in the training loop body, train_diffusion_superres.py:337,393. */
typedef struct drs_adam_tensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  int64_t step;
} drs_adam_tensor;
DRS_API int drs_adam_multi(const drs_adam_tensor* table, int ntensors, int64_t max_numel, double lr, double beta1, double beta2,
                   double eps, drs_stream_t stream);

/* Multi-tensor exponential moving average of the parameters, ONE launch for all tensors:
 *   mode 0:  ema[i] = ema[i] * beta + (1 - beta) * cur[i]   (fp32; `1 - beta` is formed in double and rounded to fp32, the
 *            two products and the sum are rounded separately - reference EMA.update_average, UNet_model_superres.py:25-30,
 *            applied per parameter by EMA.update_model_average :18-23);
 *   mode 1:  ema[i] = cur[i] as 32-bit words (the warm-up copy of EMA.reset_parameters :52-55, which load_state_dict()s
 *            parameters AND buffers: int64 counters are two words).
 * `table` (device): ntensors x drs_ema_tensor {ema, cur, n}; n counts 32-bit elements.  Replaces the 176 (mode 0) /
 * 299 (mode 1) small kernels of `ema.step_ema(ema_model, model)` in the training loop body, train_diffusion_superres.py:396. */
typedef struct drs_ema_tensor {
  void* ema;
  const void* cur;
  int64_t n;
} drs_ema_tensor;
DRS_API int drs_ema_multi(const drs_ema_tensor* table, int ntensors, int64_t max_numel, double beta, int mode, drs_stream_t stream);

/* Gaussian-weighted blend of n overlapping super-resolved tiles into one image, normalised and clamped to [0,1]:
 *   out[c][y][x] = clamp( sum_i w[y-y0_i][x-x0_i] * tiles[i][c][y-y0_i][x-x0_i] / sum_i w[y-y0_i][x-x0_i], 0, 1 )
 * over the tiles i (in index order, like the reference's sequential `+=`) whose window [y0_i, y0_i+S) x [x0_i, x0_i+S)
 * contains (y, x).  tiles: (n,C,S,S); origins: n x 2 int32 (y0, x0) on the device; weight: (S,S); out: (C,H,W).
 * Returns DRS_ERR_SHAPE through `uncovered` (device int, may be NULL) != 0 semantics: the count of output pixels no
 * tile covers is written there (the reference asserts pixel_count != 0).
 * Replaces the loop + normalisation of split_aggregation_sampling.aggregation_sampling, Aggregation_Sampling.py:90-116. */
DRS_API int drs_aggregate_tiles(const float* tiles, const int32_t* origins, const float* weight, float* out, int32_t* uncovered,
                        int n, int C, int S, int H, int W, drs_stream_t stream);

/* "DownBlur" degradation of the super-resolution data feed on the device, bit-exact with the Pillow calls of the
 * reference's dataset item: x = ToTensor(GaussianBlur(radius)(resize(y, (out_w, out_h), BICUBIC))), y = ToTensor(hr).
 *   hr: (N,C,H,W) uint8;  x_lr: (N,C,out_h,out_w) float32 in [0,1];  y_hr: (N,C,H,W) float32 or NULL;
 *   blur_radius: Pillow's GaussianBlur radius (0 = no blur);  scratch: drs_downblur_scratch_bytes(...) bytes.
 * Replaces get_data_superres.__getitem__, utils.py:140-158 (Gauss_noise=False). */
DRS_API size_t drs_downblur_scratch_bytes(int N, int C, int H, int W, int out_h, int out_w);
DRS_API int drs_downblur_u8(const uint8_t* hr, int N, int C, int H, int W, int out_h, int out_w, float blur_radius, float* x_lr,
                    float* y_hr, void* scratch, size_t scratch_bytes, drs_stream_t stream);

/* The device half of the dataset item's `Gauss_noise=True` step: x (N,C,H,W) float32 += noise (N,H,W,C) float32, clipped to
 * [0, 1], in place.  The noise is drawn on the host from the generators the reference uses (Python `random`, numpy's global
 * generator), in its order, by diffusionremotesensing_amd.degradation.reference_noise.
 * Replaces the add and the clip of add_Gaussian_noise, utils.py:27-36 (called at utils.py:163-164). */
DRS_API int drs_add_noise_clip_f32(float* x, const float* noise_nhwc, int N, int C, int H, int W, drs_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Operator-level entry points (used by the parity tests for every convolution flavour
 * the UNet contains, at arbitrary/ragged shapes)
 * ------------------------------------------------------------------------------------------ */

/* y = conv2d(x, w, b) or conv_transpose2d(x, w, b), NCHW in/out like torch.
 *   x: (N,Cin,H,W)   w: (Cout,Cin,KH,KW) [transposed: (Cin,Cout,KH,KW)]   b: (Cout) or NULL
 *   y: (N,Cout,OH,OW) with OH = (H + 2*pad - KH)/stride + 1, transposed: (H-1)*stride - 2*pad + KH + out_pad
 * Supported flavours (all the reference uses): 3x3 s1 p1; 3x3 s2 p1; 1x1; 2x2 s2 p0;
 * transposed 3x3 s2 p1 out_pad 1.  `relu` != 0 applies max(.,0).
 * workspace: device scratch of at least drs_conv2d_workspace_bytes() bytes.
 * Replaces nn.Conv2d / nn.ConvTranspose2d calls at UNet_model_superres.py:70-85,123-141,184-185,217,298,321,325. */
DRS_API size_t drs_conv2d_workspace_bytes(int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride,
                                  int pad, int transposed, int out_pad);
DRS_API int drs_conv2d_nchw(const float* x, const float* w, const float* b, float* y, int N, int Cin, int H, int W,
                    int Cout, int KH, int KW, int stride, int pad, int transposed, int out_pad, int relu,
                    void* workspace, size_t workspace_bytes, int impl, drs_stream_t stream);

/* One up-sampling stage of the decoder as the eval plan runs it (split-bf16 kernels, activations in the plan's SP format):
 *   y = conv2d(cat([conv_transpose2d(h, t_w, t_b, stride 2, padding 1, output_padding 1), att], 1), v_w, v_b, padding 1)
 * computed WITHOUT the transposed convolution's output: ups.i.transform and the first Cc input channels of up_convs.i are
 * one linear map (no activation between them), composed at pack time into a stride-2 transposed convolution with 3 x 3 /
 * 3 x 2 / 2 x 3 / 2 x 2 taps per output phase (csrc/upfuse_sp.hip).
 *   h: (N,Cc,LH,LW)  att: (N,Ch,2LH,2LW)  t_w: (Cc,Cc,3,3)  t_b: (Cc)  v_w: (Ch,Cc+Ch,3,3)  v_b: (Ch)  y: (N,Ch,2LH,2LW)
 *   post2 / y2 (both or neither): y2 = y + post2[n][c] (the next UpConvBlock's x + relu(time_mlp(t)), reference :199)
 *   fuse_w (fuse_dim,Ch) / fuse_b (fuse_dim): with Ch == 32, y is (N,fuse_dim,2LH,2LW) = conv1x1(y32) (the UNet's `output`)
 * Cc, Ch multiples of 32.  Replaces UpConvBlock.transform + torch.cat + up_convs[i] (+ output),
 * UNet_model_superres.py:206-207,376-377,379. */
DRS_API size_t drs_upconv_fused_workspace_bytes(int N, int Cc, int Ch, int LH, int LW);
DRS_API int drs_upconv_fused_nchw(const float* h, const float* att, const float* t_w, const float* t_b, const float* v_w,
                          const float* v_b, const float* post2, const float* fuse_w, const float* fuse_b, int fuse_dim,
                          float* y, float* y2, int N, int Cc, int Ch, int LH, int LW, void* workspace, size_t workspace_bytes,
                          drs_stream_t stream);

/* y = F.interpolate(x, scale_factor=scale, mode='bicubic') (align_corners=False, A=-0.75, border clamp),
 * integer scale.  x: (N,C,H,W) -> y: (N,C,H*scale,W*scale).  Replaces UNet_model_superres.py:349. */
DRS_API int drs_bicubic_upsample_nchw(const float* x, float* y, int N, int C, int H, int W, int scale,
                              drs_stream_t stream);

/* out[b, :] = relu(W2 @ silu(W1 @ posenc(t[b]) + b1) + b2), posenc = [sin(t*f_j) | cos(t*f_j)], j < dim_in/2,
 * inv_freq[j] = f_j supplied by the caller (dim_in/2 floats).  W1: (dim_out, dim_in), W2: (dim_out, dim_out).
 * Replaces pos_encoding + time_mlp + ReLU, UNet_model_superres.py:328-335,143-151,161 (and :187-199). */
DRS_API int drs_time_mlp(const int64_t* t, const float* inv_freq, const float* W1, const float* b1, const float* W2,
                 const float* b2, float* out, int B, int dim_in, int dim_out, drs_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Whole-network plan: Residual_Attention_UNet_superres.forward (UNet_model_superres.py:337-379)
 * in eval mode (BatchNorm running statistics folded into the convolutions at pack time).
 * ------------------------------------------------------------------------------------------ */

typedef struct drs_unet_config {
  int batch;          /* n images in x */
  int lr_batch;       /* batch of lr_img: == batch, or 1 (broadcast, Diffusion.sample :224) */
  int image_channels; /* reference ctor arg (3) */
  int out_dim;        /* reference ctor arg (3) */
  int height, width;  /* of x; divisible by 8 and by magnification */
  int magnification;  /* lr_img is (height/mag, width/mag) */
  int impl;           /* DRS_IMPL_* used for the wide convolutions */
  float bn_eps;       /* 1e-5 */
  int flags;          /* DRS_PLAN_* */
  int variant;        /* DRS_VARIANT_* : which of the reference's three near-identical UNets */
  int cond_channels;  /* channels of the conditioning image (superres: = image_channels; SAR: 2; generation: 0) */
  int num_classes;    /* generation: rows of label_emb (0 = no label embedding) */
} drs_unet_config;
/* Variants (same kernels, different wiring and state_dict key names):
 *  SUPERRES    Residual_Attention_UNet_superres   UNet_model_superres.py:266-379   cond = lr_img, bicubic x mag
 *  SAR_TO_NDVI Residual_Attention_UNet_SAR_TO_NDVI UNet_model_SAR_TO_NDVI.py:263-370 cond = SAR image at full size
 *              (magnification must be 1), image_channels = out_dim = NDVI channels
 *  GENERATION  Residual_Attention_UNet_generation generate_new_imgs/UNet_model_generation.py:226-329  no conditioning
 *              image; optional class label added to the time encoding (t += label_emb(y), :300-301) */
#define DRS_VARIANT_SUPERRES 0
#define DRS_VARIANT_SAR_TO_NDVI 1
#define DRS_VARIANT_GENERATION 2
/* Keep every intermediate activation readable through drs_unet_read_tensor (parity tests).  Without it the
 * 32-channel output of up_convs.2 is never written: the final 1x1 `output` conv is fused into its epilogue. */
#define DRS_PLAN_KEEP_ALL 1
/* Train-mode plan: BatchNorm uses batch statistics (and updates running_mean / running_var in place through the
 * parameter pointers given to drs_unet_pack_weights), nothing is folded or fused across a BatchNorm, every
 * pre-normalisation tensor is kept.  Reference: model.train() + nn.BatchNorm2d defaults (eps 1e-5, momentum 0.1). */
#define DRS_PLAN_TRAIN 2

DRS_API int drs_unet_plan_create(drs_plan** plan, const drs_unet_config* cfg);
DRS_API void drs_unet_plan_destroy(drs_plan* plan);

/* The state_dict entries the plan consumes, in the order drs_unet_pack_weights expects. */
DRS_API int drs_unet_num_params(const drs_plan* plan);
DRS_API const char* drs_unet_param_name(const drs_plan* plan, int i);
DRS_API int64_t drs_unet_param_numel(const drs_plan* plan, int i);

DRS_API size_t drs_unet_packed_bytes(const drs_plan* plan);
DRS_API size_t drs_unet_workspace_bytes(const drs_plan* plan);

/* Fold BatchNorm (eval) into conv weights/biases and re-lay every weight for the kernels.
 * params[i] = device pointer of state_dict[drs_unet_param_name(i)] (fp32).  inv_freq = 50 floats (host
 * pointer) computed like reference :329-331.  Must be re-run whenever the parameters change. */
DRS_API int drs_unet_pack_weights(drs_plan* plan, const void* const* params, const float* inv_freq_host, void* packed,
                          size_t packed_bytes, drs_stream_t stream);

/* eps_pred = model(x, t, lr_img, magnification).  x: (batch,C,H,W)  t: (batch) int64
 * lr_img: (lr_batch,C,H/mag,W/mag)  out: (batch,out_dim,H,W).
 * flags bit 0 (DRS_FWD_REUSE_COND): skip the LR-conditioning branch (RRDB -> bicubic -> conv, reference
 * :345-353) and reuse the one left in the workspace by the previous call — valid while lr_img and the weights
 * are unchanged, i.e. inside one Diffusion.sample chain (the reference recomputes it every step). */
#define DRS_FWD_REUSE_COND 1
DRS_API int drs_unet_forward(drs_plan* plan, const void* packed, const float* x, const int64_t* t, const float* lr_img,
                     float* out, void* workspace, size_t workspace_bytes, int flags, drs_stream_t stream);
/* Same, with class labels for the GENERATION variant: labels = int64[label_batch] (label_batch == batch or 1,
 * broadcast) or NULL for the unconditional forward (reference forward(x, timestep, y=None)). */
DRS_API int drs_unet_forward_labels(drs_plan* plan, const void* packed, const float* x, const int64_t* t, const float* cond,
                            const int64_t* labels, int label_batch, float* out, void* workspace,
                            size_t workspace_bytes, int flags, drs_stream_t stream);

/* Introspection for block-level parity tests: intermediate activations left in the workspace by the last
 * forward, converted to NCHW into `dst`.  Names follow the reference module tree
 * ("conv_blocks.0", "downs.1", "attention_blocks.2", ...). */
DRS_API int drs_unet_num_tensors(const drs_plan* plan);
DRS_API const char* drs_unet_tensor_name(const drs_plan* plan, int i);
DRS_API int drs_unet_tensor_shape(const drs_plan* plan, int i, int* n, int* c, int* h, int* w);
DRS_API int drs_unet_read_tensor(const drs_plan* plan, int i, const void* workspace, float* dst_nchw, drs_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training step, backward half (reference loop body train_diffusion_superres.py:388-393: loss.backward()).
 * Given d(loss)/d(eps_pred) (`dout`, NCHW like the output) it produces d(loss)/d(parameter) for every parameter of
 * drs_unet_param_name(): grads[i] = device pointer of drs_unet_param_numel(i) floats, or NULL to skip (BatchNorm
 * running statistics, which receive no gradient, must be NULL).  Gradients are OVERWRITTEN (zeroed first).
 * Requires a DRS_PLAN_TRAIN plan with lr_batch == batch and the workspace exactly as the last drs_unet_forward
 * (train mode) on this plan left it; `packed_bwd` is scratch for the re-packed (transposed) weights.
 * ------------------------------------------------------------------------------------------ */
DRS_API size_t drs_unet_packed_bwd_bytes(const drs_plan* plan);
DRS_API int drs_unet_backward(drs_plan* plan, const void* packed, void* packed_bwd, size_t packed_bwd_bytes, const float* x,
                      const int64_t* t, const float* dout, float* const* grads, void* workspace, size_t workspace_bytes,
                      drs_stream_t stream);
/* Same for a forward that was given class labels (GENERATION variant): also produces d(label_emb.weight). */
DRS_API int drs_unet_backward_labels(drs_plan* plan, const void* packed, void* packed_bwd, size_t packed_bwd_bytes,
                             const float* x, const int64_t* t, const int64_t* labels, int label_batch,
                             const float* dout, float* const* grads, void* workspace, size_t workspace_bytes,
                             drs_stream_t stream);

/* Per-op timing of the forward schedule: with profiling on, drs_unet_forward brackets every op with HIP events on
 * the stream it launches on; afterwards read (name, milliseconds, algorithmic FLOPs, algorithmic bytes) per op.
 * Used by bench.py for the roofline of the dominant kernel.  Not for use inside graph capture. */
/* Synchronises `stream` and returns DRS_ERR_HIP if a wave of the wave-specialised kernels gave up waiting on an LDS
 * counter since the weights were last packed into `packed` (a protocol bug; such a wave records it and ends instead of
 * hanging or faulting the device: csrc/sp_sync.h).  Debug / test aid; a healthy run never sets it. */
DRS_API int drs_unet_check_faults(drs_plan* plan, const void* packed, drs_stream_t stream);

DRS_API int drs_unet_profile_enable(drs_plan* plan, int on);
DRS_API int drs_unet_profile_num_ops(const drs_plan* plan);
DRS_API int drs_unet_profile_read(drs_plan* plan, int i, char* name, int name_len, float* ms, double* flops, double* bytes);
/* Launch log of the last PROFILED forward (ABI 6): one entry per kernel the forward launched, in host launch order, with
 * the kernel's name as the runtime reports it (demangled) and the op of the schedule that issued it ("" for launches
 * outside any op bracket, e.g. the per-image gate-bias tables).  A rocprofv3 counter pass over the same process sees
 * exactly these dispatches, in this order, as the process's last launches: tools/collect_pmc.py joins the two and
 * refuses to attribute bytes if a kernel name differs. */
DRS_API int drs_unet_profile_num_launches(const drs_plan* plan);
DRS_API int drs_unet_profile_launch(const drs_plan* plan, int i, char* op, int op_len, char* kernel, int kernel_len);

#ifdef __cplusplus
}
#endif
#endif /* DRS_HIP_H */
