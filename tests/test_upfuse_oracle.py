"""The algebra behind csrc/upfuse_sp.hip, on the CPU: ConvTranspose2d -> cat -> Conv2d == composite transposed convolution
+ att-half convolution + edge vectors (oracle/upfuse_oracle.py follows the pack / edge kernels formula for formula)."""
import pytest
import torch


@pytest.mark.parametrize("case", [(2, 8, 4, 3, 5), (1, 6, 6, 1, 1), (1, 4, 2, 2, 7), (1, 5, 3, 4, 1)])
def test_composite_equals_convtranspose_cat_conv(case):
    from oracle import upfuse_oracle as O
    N, Cc, Ch, LH, LW = case
    g = torch.Generator().manual_seed(sum(case))
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    h, att = rnd(N, Cc, LH, LW), rnd(N, Ch, 2 * LH, 2 * LW)
    t_w, t_b, v_w, v_b = rnd(Cc, Cc, 3, 3), rnd(Cc), rnd(Ch, Cc + Ch, 3, 3), rnd(Ch)
    a = O.composite_forward(h, att, t_w, t_b, v_w, v_b)
    b = O.reference_forward(h, att, t_w, t_b, v_w, v_b)
    assert float((a - b).abs().max()) <= 1e-11 * float(b.abs().max())


def test_pair_table_matches_the_closed_forms():
    """y[2m] = v0w2 x[m-1] + (v0w0 + v1w1 + v2w2) x[m] + v2w0 x[m+1];  y[2m+1] = (v0w1 + v1w2) x[m] + (v1w0 + v2w1) x[m+1]."""
    from oracle.upfuse_oracle import pair
    table = {(p, t): sorted((kv, kw) for kv in range(3) for kw in range(3) if pair(p, t, kv, kw))
             for p in range(2) for t in range(3)}
    assert table == {(0, 0): [(0, 2)], (0, 1): [(0, 0), (1, 1), (2, 2)], (0, 2): [(2, 0)],
                     (1, 0): [], (1, 1): [(0, 1), (1, 2)], (1, 2): [(1, 0), (2, 1)]}
