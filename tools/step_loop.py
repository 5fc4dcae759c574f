#!/usr/bin/env python
"""The headline step (bench.py's step_ at N = 1: render -> <render, G> -> backward) in a bare loop, for
`rocprofv3 --kernel-trace -- python tools/step_loop.py [CONFIG] [STEPS]` + tools/iter_timeline.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import _CotangentLoss
from gags_amd import synthetic as syn
from gags_amd.gaussian_renderer import render

cfg = syn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
n, w, h, d = cfg["n"], cfg["width"], cfg["height"], cfg["d"]
pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev)
pc.training_setup()
pc.cache_activations(False)
cam = syn.make_camera(w, h, device=dev)
G = syn.make_cotangent(d, h, w, seed=1, device=dev)
bg = torch.zeros(3, device=dev)


def step():
    pc._semantic_feature.grad = None
    pkg = render(cam, pc, None, bg, feature_mode=True)
    loss = _CotangentLoss.apply(pkg["render"].permute(1, 2, 0), G.permute(1, 2, 0))
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"{1e3 * (time.perf_counter() - t0) / steps:.3f} ms per step")
