#!/usr/bin/env python
"""Timing of the decoder GEMM kernels alone at 1080p (P = 2 073 600 pixels): one 256 -> 256 hidden layer of
CNN_decoder (models/networks.py:109-218), its 256 -> 512 fp32 output layer, the input-gradient form (ReLU mask and
skip-connection gradient in the epilogue) and the weight gradient.  Prints one JSON line with ms, TFLOP/s and the
algorithmic GB/s (operands read once, result written once)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gags_amd import decoders as D

dev = torch.device("cuda", 0)
P = 1920 * 1080
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16


def rnd(*s):
    return (torch.randn(*s, device=dev, generator=g) * 0.1).to(bf)


a, a2, m = rnd(P, 256), rnd(P, 256), rnd(P, 256)
w256, w512 = rnd(256, 256), rnd(512, 256)
b256, b512 = torch.zeros(256, device=dev), torch.zeros(512, device=dev)
w32 = rnd(32, 256)
z512 = rnd(P, 512)
b32 = torch.zeros(32, device=dev)


def timed(fn, n=10):
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


cases = {
    "layer_256x256": (lambda: D._layer(P, w256, b256, a), 2 * P * 256 * 256, P * 256 * 4),
    "layer_256x256_two_sources": (lambda: D._layer(P, w256, b256, a, a2), 2 * P * 256 * 256, P * 256 * 6),
    "layer_512x256_f32": (lambda: D._layer(P, w512, b512, a, relu=False, f32=True), 2 * P * 512 * 256, P * (512 + 2048)),
    "dgrad_256x256_mask_residual": (lambda: D._layer(P, w256, None, a, relu=False, mask_src=m, residual=a2), 2 * P * 256 * 256, P * 256 * 8),
    "dgrad_32x256": (lambda: D._layer(P, w32, None, a, relu=False), 2 * P * 32 * 256, P * (512 + 64)),
    "wgrad_256x256": (lambda: D._wgrad(P, a, a2, None, 256, 256), 2 * P * 256 * 256, P * 256 * 4),
    "wgrad_512x256": (lambda: D._wgrad(P, z512, a2, None, 512, 256), 2 * P * 512 * 256, P * (1024 + 512)),
    "wgrad_256x256_two_sources": (lambda: D._wgrad(P, a, a2, m, 256, 256), 2 * P * 256 * 256, P * 256 * 6),
}
# the fp32-tensor tiers (csrc/decoder_exact.hip): operands split into 3 ("exact") or 2 ("bf16x2") bf16 terms
af, af2, mf = a.float(), a2.float(), m.float()
wf256, wf512 = w256.float(), w512.float()
for terms, tag in ((3, "exact"), (2, "bf16x2")):
    cases.update({
        f"{tag}_layer_256x256": (lambda t=terms: D._xlayer(P, wf256, b256, af, terms=t), 2 * P * 256 * 256, P * 256 * 8),
        f"{tag}_layer_256x256_two_sources": (lambda t=terms: D._xlayer(P, wf256, b256, af, af2, terms=t), 2 * P * 256 * 256, P * 256 * 12),
        f"{tag}_layer_512x256": (lambda t=terms: D._xlayer(P, wf512, b512, af, relu=False, terms=t), 2 * P * 512 * 256, P * (1024 + 2048)),
        f"{tag}_dgrad_256x256_mask_residual": (lambda t=terms: D._xlayer(P, wf256, None, af, relu=False, mask_src=mf, residual=af2, terms=t),
                                               2 * P * 256 * 256, P * 256 * 16),
        f"{tag}_wgrad_256x256": (lambda t=terms: D._xwgrad(P, af, af2, None, 256, 256, terms=t), 2 * P * 256 * 256, P * 256 * 8),
        f"{tag}_wgrad_256x256_two_sources": (lambda t=terms: D._xwgrad(P, af, af2, mf, 256, 256, terms=t), 2 * P * 256 * 256, P * 256 * 12),
    })
only = [k for k in cases if any(f in k for f in sys.argv[1:])] if sys.argv[1:] else list(cases)
out = {}
for k in only:
    fn, fl, by = cases[k]
    ms = timed(fn)
    out[k] = {"ms": round(ms, 4), "TFLOPs": round(fl / ms * 1e-9, 1), "GBs": round(by / ms * 1e-6, 1)}
print(json.dumps(out))
