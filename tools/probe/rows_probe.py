#!/usr/bin/env python
"""Where does a wave of the colours-only backward's rows kernel (raster_bwd_rows_f16) spend its cycles?  Runs the C3 step on
the PROBE build (tools/probe/Makefile: -DGAGS_PROBE adds per-wave phase clocks; the shipped library contains none of this)
and prints the phase shares summed over all waves."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gags_amd import _lib

_lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "libgags_hip_probe.so")
_lib.load()
probe = ctypes.CDLL(_lib.LIB_PATH)
probe.gags_probe_set_bwd.argtypes = [ctypes.c_void_p]
from gags_amd import synthetic as syn
from gags_amd.gaussian_renderer import render

dev = torch.device("cuda", 0)
c = syn.CONFIGS["C3"]
n, w, h, d = c["n"], c["width"], c["height"], c["d"]
pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev)
pc.training_setup()
cam = syn.make_camera(w, h, device=dev)
bg = torch.zeros(3, device=dev)
G = syn.make_cotangent(d, h, w, seed=1, device=dev)


def step():
    pc._semantic_feature.grad = None
    (render(cam, pc, None, bg, feature_mode=True)["render"] * G).sum().backward()


for _ in range(2):
    step()
n_wg = ((w + 15) // 16) * ((h + 15) // 16) * (d // 128)
buf = torch.zeros(n_wg * 4 * 8, dtype=torch.int64, device=dev)
assert probe.gags_probe_set_bwd(ctypes.c_void_p(buf.data_ptr())) == 0
step()
torch.cuda.synchronize()
probe.gags_probe_set_bwd(None)
t = buf.view(n_wg * 4, 8).cpu().numpy().astype(np.float64)
t = t[t.sum(1) > 0]
names = ["prologue: metadata, slab loaded, scaled, split", "chunk bookkeeping + first barrier", "wait for the weight tile (vmcnt(0): + row stores in flight)",
         "row scale, split, 80 MFMAs", "keys, next loads issued", "unscale + park in LDS", "second barrier", "merge: LDS reads, adds, row stores"]
tot = t.sum()
print(f"{len(t)} waves with work; mean {t.sum(1).mean():.0f} clock ticks per wave (s_memtime)")
for i, nm in enumerate(names):
    print(f"  {nm:62s} {100 * t[:, i].sum() / tot:5.1f} %   mean {t[:, i].mean():9.0f}")
