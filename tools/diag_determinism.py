#!/usr/bin/env python3
"""Alternates the configs[1] eval forward (cached conditioning, as a sampling chain runs it) between two inputs A / B and
compares every output bit for bit with the first A / B run (the eval path has no atomics: every run must be identical;
alternating inputs makes a read of the PREVIOUS forward's data visible).  On a mismatch every intermediate tensor of the plan is
compared with its copy from the reference run of that input: the first one that differs names the kernel.
Usage: diag_determinism.py [forwards]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffusionremotesensing_amd import synthetic  # noqa: E402
from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda:0")
m = Residual_Attention_UNet_superres(3, 3, dev)
m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
eng = m.hip_engine()
lr = synthetic.tensor_uniform("det.lr", (1, 3, 128, 128)).to(dev)
xs = [synthetic.tensor_normal(f"det.x{i}", (16, 3, 256, 256)).to(dev) for i in range(2)]
ts = [torch.full((16,), 700 + i, dtype=torch.int64, device=dev) for i in range(2)]
with torch.no_grad():
    eng.forward(xs[0], ts[0], lr, 2, reuse_cond=False)
    refs, inter = [], []
    for i in range(2):
        y = eng.forward(xs[i], ts[i], lr, 2, reuse_cond=True, check_weights=False).clone()
        refs.append(y)
        inter.append({n: eng.read_tensor(n).clone() for n in eng.tensor_names()})
    bad = 0
    for r in range(runs):
        i = r & 1
        y = eng.forward(xs[i], ts[i], lr, 2, reuse_cond=True, check_weights=False)
        if r % 64 == 63 or r == runs - 1:  # (compare in batches: the comparison itself synchronises)
            pass
        if not torch.equal(y, refs[i]):
            bad += 1
            per_image = (y != refs[i]).flatten(1).sum(1).tolist()
            print(f"forward {r} (input {i}): output differs; per image {per_image}")
            first = True
            for n in eng.tensor_names():
                cur = eng.read_tensor(n)
                if cur.shape == inter[i][n].shape and not torch.equal(cur, inter[i][n]):
                    d = cur != inter[i][n]
                    pi = d.flatten(1).sum(1).tolist() if d.dim() > 1 else int(d.sum())
                    print(f"    {n}: {int(d.sum())} elements differ; per image {pi}")
                    if first and d.dim() == 4:
                        first = False
                        stale = (cur == inter[1 - i][n]) & d
                        print(f"      of which {int(stale.sum())} equal the OTHER input's value (data of the previous forward)")
                        idx = d.nonzero()
                        for img in sorted(set(idx[:, 0].tolist())):
                            sub = idx[idx[:, 0] == img]
                            print(f"      image {img}: channels {int(sub[:,1].min())}..{int(sub[:,1].max())} rows {int(sub[:,2].min())}..{int(sub[:,2].max())} "
                                  f"cols {int(sub[:,3].min())}..{int(sub[:,3].max())} ({len(sub)} elements)")
            if bad >= 6:
                break
torch.cuda.synchronize()
print(f"{bad} of {r + 1} forwards differ")
