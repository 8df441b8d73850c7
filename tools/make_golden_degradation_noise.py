#!/usr/bin/env python3
"""Fixtures for the `Gauss_noise=True` branch of the DownBlur feed: outputs of the reference's own `add_Gaussian_noise`
(utils.py:15-38) on seeded inputs, one case per branch.  Runs ONLY in the build container (needs /root/reference).
`utils.py` imports torchvision, cv2 and imageio at module level (dataset / video helpers the function never touches);
they are absent here and replaced by empty modules for the import.
    python tools/make_golden_degradation_noise.py   ->  tests/golden/degradation_noise_golden.npz
"""
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
for name in ("torchvision", "torchvision.transforms", "torchvision.models", "torchvision.datasets", "cv2", "imageio"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
import utils as ref_utils  # noqa: E402

g = {}
wanted = {"color": None, "gray": None, "cov": None}
seed = 0
while any(v is None for v in wanted.values()):
    random.seed(seed)
    np.random.seed(seed)
    random.randint(2, 10)
    r = np.random.rand()
    kind = "color" if r > 0.6 else ("gray" if r < 0.4 else "cov")
    if wanted[kind] is None:
        wanted[kind] = seed
    seed += 1
rng = np.random.default_rng(7)
for kind, sd in wanted.items():
    x = rng.random((3, 24, 20), dtype=np.float32)
    random.seed(sd)
    np.random.seed(sd)
    y = ref_utils.add_Gaussian_noise(torch.from_numpy(x.copy()), noise_level1=2, noise_level2=10).numpy()
    g[f"{kind}_seed"] = np.array(sd)
    g[f"{kind}_in"] = x
    g[f"{kind}_out"] = y
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "degradation_noise_golden.npz"), **g)
print("wrote", {k: int(v) for k, v in wanted.items()})
