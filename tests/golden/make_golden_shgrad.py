#!/usr/bin/env python
"""Generate tests/golden/shgrad_vectors.npz by differentiating the reference's OWN `eval_sh` (/root/reference/utils/
sh_utils.py:57-112, plain torch ops) with autograd in this container: the gradient of the SH colours with respect to the
Gaussian positions through the view direction, as gsplat forms the colours (dirs = means - campos, normalised;
colour = clamp_min(eval_sh + 0.5, 0)).  Only data is stored.

    python tests/golden/make_golden_shgrad.py
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shgrad_vectors.npz")


def main():
    sys.path.insert(0, REF)
    from utils.sh_utils import eval_sh
    g = torch.Generator().manual_seed(4242)
    n = 96
    out = {}
    sh = torch.randn(n, 3, 16, generator=g, dtype=torch.float64) * 0.6          # reference layout [N, 3, 16]
    means = (torch.randn(n, 3, generator=g, dtype=torch.float64) * 2.0 + torch.tensor([0.0, 0.0, 6.0], dtype=torch.float64))
    campos = torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64)
    v_out = torch.randn(n, 3, generator=g, dtype=torch.float64)
    out.update(shg_coeffs=sh.float().numpy(), shg_means=means.float().numpy(), shg_campos=campos.float().numpy(),
               shg_v_out=v_out.float().numpy())
    for deg in range(4):
        # float64 evaluation on the float32-rounded inputs the tests feed the kernels
        m = means.float().double().requires_grad_(True)
        c = sh.float().double().requires_grad_(True)
        dirs = m - campos.float().double()
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        col = torch.clamp_min(eval_sh(deg, c, dirs) + 0.5, 0.0)
        (col * v_out.float().double()).sum().backward()
        out[f"shg_col_deg{deg}"] = col.detach().numpy()
        out[f"shg_vmeans_deg{deg}"] = np.zeros((n, 3)) if m.grad is None else m.grad.numpy()  # degree 0 ignores the direction
        out[f"shg_vcoeffs_deg{deg}"] = c.grad.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
