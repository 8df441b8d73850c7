/*
 * TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C restatement (naive loops, fp32, NCHW) of the eval-mode forward of the reference's
 * Residual_Attention_UNet_superres (UNet_model_superres.py:337-379 and the blocks at :57-260), independent of
 * PyTorch.  It exists so that the checker does not share code (ATen) with the thing it pins: tests compare it with
 * the golden vectors that tools/make_golden.py produced by running the reference itself (tests/golden), and with the
 * torch-functional oracle.  Only tests/ may load this library.
 *
 * Parameters are passed as an array of float pointers in the order given by drs_ref_param_name(); names are the
 * reference's state_dict keys.  Sizes are small-case sizes: everything is O(N*C*H*W*K) scalar loops.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXP 256
static char g_names[MAXP][96];
static int g_nparams = 0;
static const int DOWN[5] = {16, 32, 64, 128, 256};
static const int UP[5] = {256, 128, 64, 32, 16};

static int P(const char* fmt, const char* a, int i, const char* b) {
  snprintf(g_names[g_nparams], sizeof(g_names[0]), fmt, a, i, b);
  return g_nparams++;
}
static int Pn(const char* name) {
  snprintf(g_names[g_nparams], sizeof(g_names[0]), "%s", name);
  return g_nparams++;
}

/* index tables, filled once */
typedef struct { int w, b; } WB;
typedef struct { int g, be, rm, rv; } BN;
typedef struct { WB l0, l2; } MLP;
typedef struct { MLP mlp; WB conv1, conv2, shortcut, skip; BN bn1, bn2, bns; } RES;
static struct {
  WB conv0, cond, rrdb[7], downs[3], output;
  RES enc[4];
  struct { WB gate, wg, wx, psi, result, conv, transform, upconv; BN gate_bn, result_bn, bn; MLP mlp; } dec[3];
} T;

static WB wb(const char* pfx) {
  char buf[96];
  WB r;
  snprintf(buf, sizeof buf, "%s.weight", pfx); r.w = Pn(buf);
  snprintf(buf, sizeof buf, "%s.bias", pfx); r.b = Pn(buf);
  return r;
}
static BN bn(const char* pfx) {
  char buf[96];
  BN r;
  snprintf(buf, sizeof buf, "%s.weight", pfx); r.g = Pn(buf);
  snprintf(buf, sizeof buf, "%s.bias", pfx); r.be = Pn(buf);
  snprintf(buf, sizeof buf, "%s.running_mean", pfx); r.rm = Pn(buf);
  snprintf(buf, sizeof buf, "%s.running_var", pfx); r.rv = Pn(buf);
  return r;
}
static void init_names(void) {
  char a[96], b[96];
  if (g_nparams) return;
  (void)P;
  T.conv0 = wb("conv0");
  for (int i = 0; i < 3; ++i) {
    snprintf(a, sizeof a, "LR_encoder.blocks.%d.conv1", i); T.rrdb[2 * i] = wb(a);
    snprintf(a, sizeof a, "LR_encoder.blocks.%d.conv2", i); T.rrdb[2 * i + 1] = wb(a);
  }
  T.rrdb[6] = wb("LR_encoder.conv_out");
  T.cond = wb("conv_upsampled_lr_img");
  for (int i = 0; i < 4; ++i) {
    if (i < 3) snprintf(a, sizeof a, "conv_blocks.%d", i); else snprintf(a, sizeof a, "bottle_neck");
    RES* r = &T.enc[i];
    snprintf(b, sizeof b, "%s.time_mlp.0", a); r->mlp.l0 = wb(b);
    snprintf(b, sizeof b, "%s.time_mlp.2", a); r->mlp.l2 = wb(b);
    snprintf(b, sizeof b, "%s.conv1.0", a); r->conv1 = wb(b);
    snprintf(b, sizeof b, "%s.batch_norm1", a); r->bn1 = bn(b);
    snprintf(b, sizeof b, "%s.conv2.0", a); r->conv2 = wb(b);
    snprintf(b, sizeof b, "%s.batch_norm2", a); r->bn2 = bn(b);
    snprintf(b, sizeof b, "%s.shortcut_conv.0", a); r->shortcut = wb(b);
    snprintf(b, sizeof b, "%s.shortcut_batch_norm", a); r->bns = bn(b);
    if (i == 0) { snprintf(b, sizeof b, "%s.conv_upsampled_lr_img", a); r->skip = wb(b); }
    if (i < 3) { snprintf(b, sizeof b, "downs.%d", i); T.downs[i] = wb(b); }
  }
  for (int i = 0; i < 3; ++i) {
    snprintf(a, sizeof a, "gating_signals.%d.conv", i); T.dec[i].gate = wb(a);
    snprintf(a, sizeof a, "gating_signals.%d.batch_norm", i); T.dec[i].gate_bn = bn(a);
    snprintf(a, sizeof a, "attention_blocks.%d.w_g.0", i); T.dec[i].wg = wb(a);
    snprintf(a, sizeof a, "attention_blocks.%d.w_x.0", i); T.dec[i].wx = wb(a);
    snprintf(a, sizeof a, "attention_blocks.%d.psi.0", i); T.dec[i].psi = wb(a);
    snprintf(a, sizeof a, "attention_blocks.%d.result.0", i); T.dec[i].result = wb(a);
    snprintf(a, sizeof a, "attention_blocks.%d.result.1", i); T.dec[i].result_bn = bn(a);
    snprintf(a, sizeof a, "ups.%d.time_mlp.0", i); T.dec[i].mlp.l0 = wb(a);
    snprintf(a, sizeof a, "ups.%d.time_mlp.2", i); T.dec[i].mlp.l2 = wb(a);
    snprintf(a, sizeof a, "ups.%d.conv", i); T.dec[i].conv = wb(a);
    snprintf(a, sizeof a, "ups.%d.batch_norm", i); T.dec[i].bn = bn(a);
    snprintf(a, sizeof a, "ups.%d.transform", i); T.dec[i].transform = wb(a);
    snprintf(a, sizeof a, "up_convs.%d", i); T.dec[i].upconv = wb(a);
  }
  T.output = wb("output");
}
int drs_ref_num_params(void) { init_names(); return g_nparams; }
const char* drs_ref_param_name(int i) { init_names(); return (i >= 0 && i < g_nparams) ? g_names[i] : 0; }

/* ---- primitive ops (NCHW) ------------------------------------------------------------------------------- */
/* nn.Conv2d: out[n,co,oy,ox] = b[co] + sum in[n,ci,oy*s-p+ky,ox*s-p+kx] * w[co,ci,ky,kx] */
static float* conv2d(const float* in, int N, int Ci, int H, int W, const float* w, const float* b, int Co, int K, int s,
                     int p, int* OH, int* OW) {
  const int oh = (H + 2 * p - K) / s + 1, ow = (W + 2 * p - K) / s + 1;
  float* out = (float*)malloc(sizeof(float) * (size_t)N * Co * oh * ow);
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Co; ++co)
      for (int oy = 0; oy < oh; ++oy)
        for (int ox = 0; ox < ow; ++ox) {
          float acc = b ? b[co] : 0.f;
          for (int ci = 0; ci < Ci; ++ci)
            for (int ky = 0; ky < K; ++ky) {
              const int iy = oy * s - p + ky;
              if (iy < 0 || iy >= H) continue;
              for (int kx = 0; kx < K; ++kx) {
                const int ix = ox * s - p + kx;
                if (ix < 0 || ix >= W) continue;
                acc += in[(((size_t)n * Ci + ci) * H + iy) * W + ix] * w[(((size_t)co * Ci + ci) * K + ky) * K + kx];
              }
            }
          out[(((size_t)n * Co + co) * oh + oy) * ow + ox] = acc;
        }
  *OH = oh; *OW = ow;
  return out;
}
/* nn.ConvTranspose2d(k=3, s=2, p=1, output_padding=1): out[2*iy-1+ky, 2*ix-1+kx] += in[iy,ix] * w[ci,co,ky,kx] */
static float* convT(const float* in, int N, int Ci, int H, int W, const float* w, const float* b, int Co) {
  const int oh = 2 * H, ow = 2 * W;
  float* out = (float*)malloc(sizeof(float) * (size_t)N * Co * oh * ow);
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Co; ++co)
      for (int i = 0; i < oh * ow; ++i) out[((size_t)n * Co + co) * oh * ow + i] = b[co];
  for (int n = 0; n < N; ++n)
    for (int ci = 0; ci < Ci; ++ci)
      for (int iy = 0; iy < H; ++iy)
        for (int ix = 0; ix < W; ++ix) {
          const float v = in[(((size_t)n * Ci + ci) * H + iy) * W + ix];
          for (int co = 0; co < Co; ++co)
            for (int ky = 0; ky < 3; ++ky) {
              const int oy = 2 * iy - 1 + ky;
              if (oy < 0 || oy >= oh) continue;
              for (int kx = 0; kx < 3; ++kx) {
                const int ox = 2 * ix - 1 + kx;
                if (ox < 0 || ox >= ow) continue;
                out[(((size_t)n * Co + co) * oh + oy) * ow + ox] += v * w[(((size_t)ci * Co + co) * 3 + ky) * 3 + kx];
              }
            }
        }
  return out;
}
/* nn.BatchNorm2d eval: (x - rm) / sqrt(rv + 1e-5) * g + b */
static void bn_eval(float* x, int N, int C, int HW, const float* const* Pp, BN t) {
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      const float sc = Pp[t.g][c] / sqrtf(Pp[t.rv][c] + 1e-5f), sh = Pp[t.be][c] - Pp[t.rm][c] * sc;
      float* p = x + ((size_t)n * C + c) * HW;
      for (int i = 0; i < HW; ++i) p[i] = p[i] * sc + sh;
    }
}
static void relu(float* x, size_t n) { for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : 0.f; }
static void add(float* x, const float* y, size_t n) { for (size_t i = 0; i < n; ++i) x[i] += y[i]; }

/* relu(Linear(silu(Linear(e)))) with e = [sin(t f_j) | cos(t f_j)], f_j = 1/10000^(2j/100)  (:328-335, :143-151, :161) */
static float* time_mlp(const long long* t, int B, const float* const* Pp, MLP m, int dim) {
  float* out = (float*)malloc(sizeof(float) * (size_t)B * dim);
  float e[100], h[256];
  for (int b = 0; b < B; ++b) {
    for (int j = 0; j < 50; ++j) {
      const float f = 1.0f / powf(10000.0f, (float)(2 * j) / 100.0f);
      e[j] = sinf((float)t[b] * f);
      e[50 + j] = cosf((float)t[b] * f);
    }
    for (int c = 0; c < dim; ++c) {
      float a = Pp[m.l0.b][c];
      for (int k = 0; k < 100; ++k) a += Pp[m.l0.w][(size_t)c * 100 + k] * e[k];
      h[c] = a / (1.f + expf(-a));
    }
    for (int c = 0; c < dim; ++c) {
      float a = Pp[m.l2.b][c];
      for (int k = 0; k < dim; ++k) a += Pp[m.l2.w][(size_t)c * dim + k] * h[k];
      out[(size_t)b * dim + c] = a > 0.f ? a : 0.f;
    }
  }
  return out;
}
static void add_nc(float* x, const float* v, int N, int C, int HW) {
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c)
      for (int i = 0; i < HW; ++i) x[((size_t)n * C + c) * HW + i] += v[(size_t)n * C + c];
}
/* F.interpolate(mode='bicubic', align_corners=False): Keys kernel A = -0.75, src = (dst + .5)/scale - .5, clamp taps */
static float cc1(float x) { return ((-0.75f + 2.f) * x - (-0.75f + 3.f)) * x * x + 1.f; }
static float cc2(float x) { return ((-0.75f * x + 3.75f) * x - 6.f) * x + 3.f; }
static float* bicubic(const float* in, int NC, int H, int W, int s) {
  const int oh = H * s, ow = W * s;
  float* out = (float*)malloc(sizeof(float) * (size_t)NC * oh * ow);
  for (int c = 0; c < NC; ++c)
    for (int oy = 0; oy < oh; ++oy)
      for (int ox = 0; ox < ow; ++ox) {
        const float sy = (oy + 0.5f) / s - 0.5f, sx = (ox + 0.5f) / s - 0.5f;
        const float fy = floorf(sy), fx = floorf(sx), ty = sy - fy, tx = sx - fx;
        const float wy[4] = {cc2(ty + 1.f), cc1(ty), cc1(1.f - ty), cc2(2.f - ty)};
        const float wx[4] = {cc2(tx + 1.f), cc1(tx), cc1(1.f - tx), cc2(2.f - tx)};
        float acc = 0.f;
        for (int a = 0; a < 4; ++a) {
          int yy = (int)fy - 1 + a; yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
          float row = 0.f;
          for (int b = 0; b < 4; ++b) {
            int xx = (int)fx - 1 + b; xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
            row += in[((size_t)c * H + yy) * W + xx] * wx[b];
          }
          acc += row * wy[a];
        }
        out[((size_t)c * oh + oy) * ow + ox] = acc;
      }
  return out;
}

/* ResConvBlock.forward (:153-172) */
static float* res_block(const float* x, const float* x_skip, const long long* t, int B, int Ci, int Co, int H, int W,
                        const float* const* Pp, const RES* r) {
  int oh, ow;
  const size_t n = (size_t)B * Co * H * W;
  float* h = conv2d(x, B, Ci, H, W, Pp[r->conv1.w], Pp[r->conv1.b], Co, 3, 1, 1, &oh, &ow);
  bn_eval(h, B, Co, H * W, Pp, r->bn1);
  relu(h, n);
  if (x_skip) {
    float* k = conv2d(x_skip, B, Ci, H, W, Pp[r->skip.w], Pp[r->skip.b], Co, 3, 1, 1, &oh, &ow);
    add(h, k, n);
    free(k);
  }
  float* te = time_mlp(t, B, Pp, r->mlp, Co);
  add_nc(h, te, B, Co, H * W);
  free(te);
  float* h2 = conv2d(h, B, Co, H, W, Pp[r->conv2.w], Pp[r->conv2.b], Co, 3, 1, 1, &oh, &ow);
  bn_eval(h2, B, Co, H * W, Pp, r->bn2);
  free(h);
  float* s = conv2d(x, B, Ci, H, W, Pp[r->shortcut.w], Pp[r->shortcut.b], Co, 1, 1, 0, &oh, &ow);
  bn_eval(s, B, Co, H * W, Pp, r->bns);
  add(h2, s, n);
  free(s);
  relu(h2, n);
  return h2;
}

/* Residual_Attention_UNet_superres.forward (:337-379), eval mode.  Returns 0 on success.
 * tap_name / tap_out: optional copy of one named intermediate ("conv_blocks.1", "attention_blocks.0", ...). */
int drs_ref_unet_forward(const float* const* Pp, const float* x, const long long* t, const float* lr, float* out, int B,
                         int Bl, int C, int H, int W, int mag, const char* tap_name, float* tap_out) {
  init_names();
  int oh, ow;
  const int h = H / mag, w = W / mag;
#define TAP(name, ptr, count) do { if (tap_name && tap_out && !strcmp(tap_name, name)) memcpy(tap_out, ptr, sizeof(float) * (count)); } while (0)
  /* LR encoder: RRDB (:237-260) */
  float* cur = (float*)malloc(sizeof(float) * (size_t)Bl * C * h * w);
  memcpy(cur, lr, sizeof(float) * (size_t)Bl * C * h * w);
  for (int i = 0; i < 3; ++i) {
    float* a = conv2d(cur, Bl, C, h, w, Pp[T.rrdb[2 * i].w], Pp[T.rrdb[2 * i].b], C, 3, 1, 1, &oh, &ow);
    relu(a, (size_t)Bl * C * h * w);
    float* b = conv2d(a, Bl, C, h, w, Pp[T.rrdb[2 * i + 1].w], Pp[T.rrdb[2 * i + 1].b], C, 3, 1, 1, &oh, &ow);
    add(b, cur, (size_t)Bl * C * h * w);
    free(a); free(cur);
    cur = b;
  }
  float* enc = conv2d(cur, Bl, C, h, w, Pp[T.rrdb[6].w], Pp[T.rrdb[6].b], C, 3, 1, 1, &oh, &ow);
  add(enc, lr, (size_t)Bl * C * h * w);
  free(cur);
  TAP("LR_encoder", enc, (size_t)Bl * C * h * w);
  float* up = bicubic(enc, Bl * C, h, w, mag);
  free(enc);
  float* cond = conv2d(up, Bl, C, H, W, Pp[T.cond.w], Pp[T.cond.b], DOWN[0], 3, 1, 1, &oh, &ow);
  free(up);
  float* x0 = conv2d(x, B, C, H, W, Pp[T.conv0.w], Pp[T.conv0.b], DOWN[0], 3, 1, 1, &oh, &ow);
  for (int n = 0; n < B; ++n)  /* x + upsampled (broadcast when lr batch is 1, :355) */
    add(x0 + (size_t)n * DOWN[0] * H * W, cond + (size_t)(Bl == 1 ? 0 : n) * DOWN[0] * H * W, (size_t)DOWN[0] * H * W);
  free(cond);

  float* res[3];
  float* xin = x0;
  char nm[64];
  for (int i = 0; i < 3; ++i) {
    const int hh = H >> i, ww = W >> i;
    res[i] = res_block(xin, i == 0 ? xin : 0, t, B, DOWN[i], DOWN[i + 1], hh, ww, Pp, &T.enc[i]);
    snprintf(nm, sizeof nm, "conv_blocks.%d", i);
    TAP(nm, res[i], (size_t)B * DOWN[i + 1] * hh * ww);
    float* d = conv2d(res[i], B, DOWN[i + 1], hh, ww, Pp[T.downs[i].w], Pp[T.downs[i].b], DOWN[i + 1], 3, 2, 1, &oh, &ow);
    snprintf(nm, sizeof nm, "downs.%d", i);
    TAP(nm, d, (size_t)B * DOWN[i + 1] * oh * ow);
    free(xin);
    xin = d;
  }
  float* xc = res_block(xin, 0, t, B, DOWN[3], DOWN[4], H >> 3, W >> 3, Pp, &T.enc[3]);
  free(xin);
  TAP("bottle_neck", xc, (size_t)B * DOWN[4] * (H >> 3) * (W >> 3));

  for (int i = 0; i < 3; ++i) {
    const int Cc = UP[i], Ch = UP[i + 1], lh = H >> (3 - i), lw = W >> (3 - i), hh = 2 * lh, hw = 2 * lw;
    const float* xr = res[2 - i];
    /* gating_signal (:222-225) */
    float* g = conv2d(xc, B, Cc, lh, lw, Pp[T.dec[i].gate.w], Pp[T.dec[i].gate.b], Ch, 1, 1, 0, &oh, &ow);
    bn_eval(g, B, Ch, lh * lw, Pp, T.dec[i].gate_bn);
    relu(g, (size_t)B * Ch * lh * lw);
    /* AttentionBlock (:89-108) */
    float* g1 = conv2d(g, B, Ch, lh, lw, Pp[T.dec[i].wg.w], Pp[T.dec[i].wg.b], Ch, 1, 1, 0, &oh, &ow);
    float* x1 = conv2d(xr, B, Ch, hh, hw, Pp[T.dec[i].wx.w], Pp[T.dec[i].wx.b], Ch, 2, 2, 0, &oh, &ow);
    add(x1, g1, (size_t)B * Ch * lh * lw);
    relu(x1, (size_t)B * Ch * lh * lw);
    float* psi = conv2d(x1, B, Ch, lh, lw, Pp[T.dec[i].psi.w], Pp[T.dec[i].psi.b], 1, 1, 1, 0, &oh, &ow);
    for (size_t k = 0; k < (size_t)B * lh * lw; ++k) psi[k] = 1.f / (1.f + expf(-psi[k]));
    float* gated = (float*)malloc(sizeof(float) * (size_t)B * Ch * hh * hw);
    for (int n = 0; n < B; ++n)
      for (int c = 0; c < Ch; ++c)
        for (int y = 0; y < hh; ++y)
          for (int xx = 0; xx < hw; ++xx)  /* nearest x2 of psi, repeated over channels, times x (:105-107) */
            gated[(((size_t)n * Ch + c) * hh + y) * hw + xx] =
                psi[((size_t)n * lh + y / 2) * lw + xx / 2] * xr[(((size_t)n * Ch + c) * hh + y) * hw + xx];
    float* att = conv2d(gated, B, Ch, hh, hw, Pp[T.dec[i].result.w], Pp[T.dec[i].result.b], Ch, 1, 1, 0, &oh, &ow);
    bn_eval(att, B, Ch, hh * hw, Pp, T.dec[i].result_bn);
    free(g); free(g1); free(x1); free(psi); free(gated);
    snprintf(nm, sizeof nm, "attention_blocks.%d", i);
    TAP(nm, att, (size_t)B * Ch * hh * hw);
    /* UpConvBlock (:197-207) */
    float* te = time_mlp(t, B, Pp, T.dec[i].mlp, Cc);
    add_nc(xc, te, B, Cc, lh * lw);
    free(te);
    float* u = conv2d(xc, B, Cc, lh, lw, Pp[T.dec[i].conv.w], Pp[T.dec[i].conv.b], Cc, 3, 1, 1, &oh, &ow);
    bn_eval(u, B, Cc, lh * lw, Pp, T.dec[i].bn);
    relu(u, (size_t)B * Cc * lh * lw);
    float* ut = convT(u, B, Cc, lh, lw, Pp[T.dec[i].transform.w], Pp[T.dec[i].transform.b], Cc);
    free(u); free(xc);
    snprintf(nm, sizeof nm, "ups.%d", i);
    TAP(nm, ut, (size_t)B * Cc * hh * hw);
    /* cat + up_conv (:376-377) */
    float* cat = (float*)malloc(sizeof(float) * (size_t)B * (Cc + Ch) * hh * hw);
    for (int n = 0; n < B; ++n) {
      memcpy(cat + (size_t)n * (Cc + Ch) * hh * hw, ut + (size_t)n * Cc * hh * hw, sizeof(float) * (size_t)Cc * hh * hw);
      memcpy(cat + ((size_t)n * (Cc + Ch) + Cc) * hh * hw, att + (size_t)n * Ch * hh * hw, sizeof(float) * (size_t)Ch * hh * hw);
    }
    free(ut); free(att);
    xc = conv2d(cat, B, Cc + Ch, hh, hw, Pp[T.dec[i].upconv.w], Pp[T.dec[i].upconv.b], Ch, 3, 1, 1, &oh, &ow);
    free(cat);
    snprintf(nm, sizeof nm, "up_convs.%d", i);
    TAP(nm, xc, (size_t)B * Ch * hh * hw);
  }
  float* y = conv2d(xc, B, UP[3], H, W, Pp[T.output.w], Pp[T.output.b], C, 1, 1, 0, &oh, &ow);
  memcpy(out, y, sizeof(float) * (size_t)B * C * H * W);
  free(y); free(xc);
  for (int i = 0; i < 3; ++i) free(res[i]);
#undef TAP
  return 0;
}
