#!/usr/bin/env python
"""Run on the GPU box: which in-kernel formulas equal torch's exp / normalize / sigmoid bit for bit (the getters of
scene/gaussian_model.py:116-139 as torch evaluates them on this stack)?  Decides the prologue of gags_project_fwd_raw."""
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "bin", "libactprobe.so"))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
n = 4_000_000
p = lambda t: ctypes.c_void_p(t.data_ptr())
q = torch.randn(n, 4, device=dev, generator=g)
want = torch.nn.functional.normalize(q)
for v in range(6):
    out = torch.empty_like(q)
    lib.probe_norm(v, n, p(q), p(out), None)
    torch.cuda.synchronize()
    print(f"normalize variant {v}: {int((out != want).any(1).sum())} of {n} rows differ")
x = 1.5 * torch.randn(n, device=dev, generator=g) - 4.0
for m in (1.0, 0.7):
    out = torch.empty_like(x)
    lib.probe_exp(n, p(x), ctypes.c_float(m), p(out), None)
    torch.cuda.synchronize()
    print(f"exp * {m}: {int((out != torch.exp(x) * m).sum())} of {n} differ")
x = 1.5 * torch.randn(n, device=dev, generator=g)
for v in range(2):
    out = torch.empty_like(x)
    lib.probe_sigmoid(v, n, p(x), p(out), None)
    torch.cuda.synchronize()
    print(f"sigmoid variant {v}: {int((out != torch.sigmoid(x)).sum())} of {n} differ")
