#!/usr/bin/env python
"""Golden vectors for SURVEY 8a R9: the reference's optimizer on the feature parameter.

The reference builds it as `torch.optim.Adam(l, lr=0.0, eps=1e-15)` with one group
{'params': [_semantic_feature], 'lr': semantic_feature_lr, 'name': 'semantic_feature'}
(/root/reference/scene/gaussian_model.py:192-208) and steps it once per iteration
(/root/reference/train.py:221-223).  `training_setup` itself hard-codes device="cuda" and cannot run in this
container, so this script issues the identical constructor call on CPU tensors and records three steps.
Run here (CPU): python tests/golden/make_golden_adam.py  ->  tests/golden/adam_vectors.npz
"""
import os

import numpy as np
import torch

torch.manual_seed(7)
n, d, lr = 257, 19, 1e-3  # numel = 4883: not a multiple of 4
p = torch.nn.Parameter((torch.randn(n, d) * 0.1).contiguous())
opt = torch.optim.Adam([{"params": [p], "lr": lr, "name": "semantic_feature"}], lr=0.0, eps=1e-15)
out = {"p0": p.detach().numpy().copy(), "lr": np.float64(lr)}
for t in range(1, 4):
    g = torch.randn(n, d)
    g[torch.rand(n) < 0.3] = 0.0  # Gaussians that were not visible in this view: zero gradient rows
    p.grad = g.clone()
    opt.step()
    st = opt.state[p]
    out[f"g{t}"] = g.numpy().copy()
    out[f"p{t}"] = p.detach().numpy().copy()
    out[f"m{t}"] = st["exp_avg"].numpy().copy()
    out[f"v{t}"] = st["exp_avg_sq"].numpy().copy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "adam_vectors.npz"), **out)
print("wrote adam_vectors.npz", {k: getattr(v, "shape", ()) for k, v in out.items()})
