#!/usr/bin/env python
"""Where does a wave of the feature pass (raster_fwd_feat<4>) spend its life?  Runs the C3 view on the PROBE build of the
library (tools/probe/Makefile: -DGAGS_PROBE adds wall-clock stamps at the phase boundaries of every wave; the shipped
library contains none of this) and prints the phase shares, the per-SIMD concurrency and the spread over the launch."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gags_amd import _lib

_lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "libgags_hip_probe.so")
lib = _lib.load()
probe = ctypes.CDLL(_lib.LIB_PATH)
probe.gags_probe_set.argtypes = [ctypes.c_void_p]
from gags_amd import synthetic as syn
from gags_amd.gaussian_renderer import render

dev = torch.device("cuda", 0)
c = syn.CONFIGS["C3"]
n, w, h, d = c["n"], c["width"], c["height"], c["d"]
pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev)
pc.training_setup()
cam = syn.make_camera(w, h, device=dev)
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    for _ in range(2):
        render(cam, pc, None, bg, feature_mode=True)
    n_waves = ((w + 15) // 16) * ((h + 15) // 16) * 4 * (d // 128)
    buf = torch.zeros(n_waves * 8, dtype=torch.int64, device=dev)
    assert probe.gags_probe_set(ctypes.c_void_p(buf.data_ptr())) == 0
    render(cam, pc, None, bg, feature_mode=True)
    torch.cuda.synchronize()
    probe.gags_probe_set(None)
t = buf.view(n_waves, 8).cpu().numpy().astype(np.int64)
ok = (t[:, 0] > 0) & (t[:, 7] > 4)   # waves that ran a K loop of more than one group (empty / tiny blocks write no loop stamps)
t = t[ok]
tick = 10.0  # wall_clock64: 100 MHz -> ns per tick
st = (t[:, :6] - t[:, :1]) * tick / 1e3  # us from the wave's start
life = st[:, 5]
print(f"{len(t)} waves; launch span {(t[:, 5].max() - t[:, 0].min()) * tick / 1e6:.3f} ms")
names = ["start -> pipeline primed (ids, 4 row groups, weights requested)", "-> first group multiplied (operands landed)",
         "-> K loop done", "-> upper half stored", "-> lower half stored"]
prev = np.zeros(len(t))
for i, nm in enumerate(names, start=1):
    seg = st[:, i] - prev
    print(f"  {nm:62s} mean {seg.mean():7.2f} us  median {np.median(seg):7.2f}  p90 {np.percentile(seg, 90):7.2f}   {100 * seg.sum() / life.sum():5.1f} % of wave time")
    prev = st[:, i]
steps = t[:, 7]
kloop = st[:, 3] - st[:, 2]
per_step = kloop / np.maximum(steps - 4, 1)
print(f"  wave life mean {life.mean():.2f} us; K-steps per wave mean {steps.mean():.1f}; K loop {np.mean(per_step) * 1e3:.0f} ns per K-step = "
      f"{np.mean(per_step) * 1e3 / 8:.0f} ns per MFMA issued by the wave (64 cycles = {64 / 2.4:.0f} ns at 2.4 GHz when it owns the pipe)")
# concurrency: how many waves are alive on the same SIMD while a wave is in its K loop?
hw = t[:, 6]
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("  HW_ID fields seen: simd", np.unique(simd).size, "cu", np.unique(cu).size, "sh", np.unique(sh).size, "se", np.unique(se).size)
order = np.argsort(t[:, 0])
print(f"  first wave starts at 0, last wave starts at {(t[:, 0].max() - t[:, 0].min()) * tick / 1e6:.3f} ms; "
      f"waves with zero K-steps: {(steps == 0).sum()}")
