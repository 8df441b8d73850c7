"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement of the pack-time algebra behind csrc/upfuse_sp.hip: ConvTranspose2d(k3, s2, p1, op1) followed by the first
Cc input channels of a 3x3 convolution, composed into a stride-2 transposed convolution with 3 x 3 / 3 x 2 / 2 x 3 / 2 x 2
taps per output phase plus per-edge correction vectors.  Reference semantics: UNet_model_superres.py:206-207 (UpConvBlock
returns self.transform(x), no activation), :376-377 (torch.cat -> up_convs[i], a bare Conv2d).  `composite_forward`
follows the formulas of the HIP pack / edge kernels index for index; tests compare it with
F.conv2d(cat([F.conv_transpose2d(h), att])) so that an error in the algebra is caught on the CPU, before any GPU run.
"""
import torch
import torch.nn.functional as F


def pair(p, t, kv, kw):
    """(kv, kw) = (3x3 tap, transposed-convolution tap) contributes to tap t of output phase p (upfuse_sp.hip: uf_pair)."""
    return p + kv - kw == 2 * (t - 1)


def composite_weights(v_w, t_w, Cc):
    """U[py][px][ty][tx] (Ch, Cc) matrices (None where the phase has no such tap)."""
    Ch = v_w.shape[0]
    U = {}
    for py in range(2):
        for px in range(2):
            for ty in range(py, 3):
                for tx in range(px, 3):
                    m = torch.zeros(Ch, Cc, dtype=v_w.dtype)
                    for kvy in range(3):
                        for kwy in range(3):
                            if not pair(py, ty, kvy, kwy):
                                continue
                            for kvx in range(3):
                                for kwx in range(3):
                                    if pair(px, tx, kvx, kwx):
                                        # sum_c V[co][c][kvy][kvx] * W[ci][c][kwy][kwx]
                                        m += v_w[:, :Cc, kvy, kvx] @ t_w[:, :, kwy, kwx].t()
                    U[(py, px, ty, tx)] = m
    return U


def edge_weights(v_w, t_w, Cc):
    """rt[(p, t)] (first-row paths), rl[(p, t)] (first-column paths), rl0 (output row 0 variant of (0, 1))."""
    rt, rl = {}, {}
    for p in range(2):
        for t in range(p, 3):
            a = torch.zeros(v_w.shape[0], Cc, dtype=v_w.dtype)
            b = torch.zeros_like(a)
            for kv in range(3):
                for kw in range(3):
                    if pair(p, t, kv, kw):
                        a += v_w[:, :Cc, 0, kv] @ t_w[:, :, 0, kw].t()
                        b += v_w[:, :Cc, kv, 0] @ t_w[:, :, kw, 0].t()
            rt[(p, t)] = a
            rl[(p, t)] = b
    rl0 = torch.zeros_like(rl[(0, 1)])
    for kv, kw in ((1, 1), (2, 2)):
        rl0 += v_w[:, :Cc, kv, 0] @ t_w[:, :, kw, 0].t()
    return rt, rl, rl0


def composite_forward(h, att, t_w, t_b, v_w, v_b):
    """The fused stage exactly as the kernels compute it: composite taps + bias + att-half + edge vectors."""
    N, Cc, LH, LW = h.shape
    Ch = v_w.shape[0]
    OH, OW = 2 * LH, 2 * LW
    U = composite_weights(v_w, t_w, Cc)
    hp = F.pad(h, (1, 1, 1, 1))
    y = torch.zeros(N, Ch, OH, OW, dtype=h.dtype)
    for (py, px, ty, tx), m in U.items():
        src = hp[:, :, ty:ty + LH, tx:tx + LW]          # h[my + ty - 1][mx + tx - 1]
        y[:, :, py::2, px::2] += torch.einsum("oc,nchw->nohw", m, src)
    bt = torch.einsum("ocyx,c->oyx", v_w[:, :Cc], t_b)   # bt[co][kvy][kvx]
    y += (v_b + bt.sum((1, 2)))[None, :, None, None]
    y += F.conv2d(att, v_w[:, Cc:], None, padding=1)     # att-half partial sums
    # edge vectors (upfuse_edges_kernel)
    rt, rl, rl0 = edge_weights(v_w, t_w, Cc)
    for oy in range(OH):
        for ox in range(OW):
            if not (oy in (0, OH - 1) or ox in (0, OW - 1)):
                continue
            val = torch.zeros(N, Ch, dtype=h.dtype)
            for kvy in range(3):
                for kvx in range(3):
                    if (oy == 0 and kvy == 0) or (oy == OH - 1 and kvy == 2) or (ox == 0 and kvx == 0) or \
                            (ox == OW - 1 and kvx == 2):
                        val -= bt[:, kvy, kvx][None]
            if oy == 0:
                m, p = ox >> 1, ox & 1
                for t in range(p, 3):
                    x = m + t - 1
                    if 0 <= x < LW:
                        val -= h[:, :, 0, x] @ rt[(p, t)].t()
            if ox == 0:
                m, p = oy >> 1, oy & 1
                for t in range(p, 3):
                    yy = m + t - 1
                    if 0 <= yy < LH:
                        w = rl0 if (oy == 0 and t == 1) else rl[(p, t)]
                        val -= h[:, :, yy, 0] @ w.t()
            y[:, :, oy, ox] += val
    return y


def reference_forward(h, att, t_w, t_b, v_w, v_b):
    u = F.conv_transpose2d(h, t_w, t_b, stride=2, padding=1, output_padding=1)
    return F.conv2d(torch.cat([u, att], 1), v_w, v_b, padding=1)
