#!/usr/bin/env python
"""Where does a wave of the DEFAULT feature pass (raster_fwd_feat_x16) spend its life?  Runs the C3 view on the PROBE build of
the library (tools/probe/Makefile: -DGAGS_PROBE adds wall-clock stamps at the phase boundaries of every wave; the shipped
library contains none of this) and prints the phase shares and the per-step time.
    make -C tools/probe && gpurun -- 'python tools/probe/feat16_probe.py'"""
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
c = syn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
n, w, h, d = c["n"], c["width"], c["height"], c["d"]
pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev)
pc.training_setup()
cam = syn.make_camera(w, h, device=dev)
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    for _ in range(2):
        render(cam, pc, None, bg, feature_mode=True)
    spw = int(os.environ.get("GAGS_FWD_SPW", "2"))  # 128-channel slices per wave (fwd_slices_per_wave, csrc/raster_fwd_mfma.hip)
    n_waves = ((w + 15) // 16) * ((h + 15) // 16) * 4 * (d // 128 // spw)
    buf = torch.zeros(n_waves * 8, dtype=torch.int64, device=dev)
    assert probe.gags_probe_set(ctypes.c_void_p(buf.data_ptr())) == 0
    render(cam, pc, None, bg, feature_mode=True)
    torch.cuda.synchronize()
    probe.gags_probe_set(None)
t = buf.view(n_waves, 8).cpu().numpy().astype(np.int64)
tick = 10.0  # wall_clock64: 100 MHz -> ns per tick
span = (t[:, 5].max() - t[:, 0].min()) * tick / 1e6
print(f"{n_waves} waves; launch span {span:.3f} ms; waves without a step: {(t[:, 7] == 0).sum()}")
ok = t[:, 7] >= 2
t = t[ok]
st = (t[:, :6] - t[:, :1]) * tick / 1e3  # us from the wave's start
life = st[:, 5]
names = ["start -> metadata arrived (offsets, block count)", "-> ids of step 0 arrived, rows + weights requested",
         "-> step 0 multiplied (its operands had landed)", "-> K loop of the FIRST slice done", "-> every slice's stores issued (the later slices' K loops included)"]
prev = np.zeros(len(t))
for i, nm in enumerate(names, start=1):
    seg = st[:, i] - prev
    print(f"  {nm:55s} mean {seg.mean():7.2f} us  median {np.median(seg):7.2f}  p90 {np.percentile(seg, 90):7.2f}   {100 * seg.sum() / life.sum():5.1f} % of wave time")
    prev = st[:, i]
steps = t[:, 7]
kloop = st[:, 4] - st[:, 3]
per_step = kloop / np.maximum(steps - 1, 1)
print(f"  wave life (to the last store ISSUED) mean {life.mean():.2f} us; steps per wave mean {steps.mean():.2f}; K loop after step 0: "
      f"{np.mean(per_step) * 1e3:.0f} ns per step (median {np.median(per_step) * 1e3:.0f}); 48 MFMAs own the pipe for 1536 cycles = 768 ns at 2.0 GHz")
# slot occupancy: the sum of wave lives against the launch span x 2048 slots
print(f"  sum of wave lives {life.sum() / 1e3:.1f} ms = {life.sum() / 1e3 / (span * 2048) * 100:.1f} % of span x 2048 wave slots "
      f"(the rest: stores draining after the last stamp, dispatch gaps)")
for lo, hi in ((2, 4), (4, 8), (8, 12), (12, 20), (20, 1000)):
    m = (steps >= lo) & (steps < hi)
    if m.any():
        print(f"  waves with {lo:3d}..{hi - 1:3d} steps: {m.sum():6d}, life {life[m].mean():6.2f} us, per step {np.mean(per_step[m]) * 1e3:6.0f} ns, "
              f"prologue {st[m, 3].mean():5.2f} us, epilogue {(st[m, 5] - st[m, 4]).mean():5.2f} us")
