#!/usr/bin/env python
"""The two fused CNN_decoder kernels alone at 1080p (f16 tier): ms of the forward chain and of the input-gradient chain, for
A/B runs of library builds (timing-only ablations: make ABL=n in tools/tmp).  usage: chain_bench.py [path/to/libgags_hip.so]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gags_amd._lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from gags_amd import decoders as D

dev = torch.device("cuda", 0)
H, W = 1080, 1920
dec = D.CNN_decoder(16, 512, "f16").to(dev)
params = [t for cv in dec.convs() for t in (cv.weight, cv.bias)]
x = torch.randn(H, W, 16, device=dev).permute(2, 0, 1)
mode = D._F16


def timed(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


logits, acts, wb, h, w, c_in = D._chain_forward(x, "decoder", params, mode)
fwd = timed(lambda: D._chain_forward(x, "decoder", params, mode))
dz = (torch.randn(H * W, 512, device=dev) * 1e-3).to(torch.float16)
shapes = [tuple(t.shape) for t in params[0::2]]
bwd = timed(lambda: D._chain_backward(dz, acts, wb, "decoder", h, w, c_in, shapes, True, [False] * 9, mode=mode))
print(f"{os.path.basename(L.LIB_PATH)}: chain forward {fwd:.3f} ms, input-gradient chain {bwd:.3f} ms")
