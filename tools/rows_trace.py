#!/usr/bin/env python
"""Per-wave timeline of the staged backward's rows kernel (traced build: raster_flags = 16).

Prints the resident-wave average per CU, the empty time of the wave slots and how a wave's time splits into
load wait / MFMA burst / row stores.  This is the tool that showed the two waves of a SIMD running in lock step."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the trace buffer lives only in the diagnostics build (make -C gags_amd/csrc trace)
os.environ.setdefault("GAGS_HIP_LIBRARY", os.path.join(ROOT, "gags_amd", "csrc", "libgags_hip_trace.so"))
import numpy as np
import torch
from gags_amd import _lib, synthetic as syn
from gags_amd.gaussian_renderer import render

cfg = syn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
n, d, w, h = cfg["n"], cfg["d"], cfg["width"], cfg["height"]
dev = torch.device("cuda", 0)
pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev)
pc.training_setup()
cam = syn.make_camera(w, h, device=dev)
G = syn.make_cotangent(d, h, w, seed=1, device=dev)
for it in range(3):
    pc._semantic_feature.grad = None
    pkg = render(cam, pc, None, torch.zeros(3, device=dev), feature_mode=True, raster_flags=_lib.GAGS_BWD_TRACE)
    (pkg["render"] * G).sum().backward()
torch.cuda.synchronize()
nwg = ((w + 15) // 16) * ((h + 15) // 16) * 4 * (d // 128)
buf = np.zeros((nwg, 8), np.int64)
lib = _lib.load()
assert lib.gags_debug_rows_trace(buf.ctypes.data, nwg) == 0
b = buf[buf[:, 1] > 0]
t0, t2 = b[:, 0], b[:, 1]
xcc = (b[:, 2] >> 32) & 15
hw = b[:, 2] & 0xFFFFFFFF
cu = (hw >> 8) & 15; se = (hw >> 13) & 7; simd = (hw >> 4) & 3; wid = hw & 15
cnt = b[:, 3] >> 32
real_ns = (b[:, 3] & 0xFFFFFFFF).astype(np.float64) * 10.0
tiles = (cnt + 31) // 32
tick_per_ns = (t2 - t0).sum() / real_ns.sum()
print("waves %d  mean wave %.1f us  (counter: %.3f ticks/ns)" % (len(b), real_ns.mean() / 1e3, tick_per_ns))
print("sum(wave time) / 2048 slots = %.3f ms" % (real_ns.sum() / 1e6 / 2048))
key = xcc * 4096 + se * 64 + cu
avg, spans = [], []
for u in np.unique(key):
    m = key == u
    sp = t2[m].max() - t0[m].min()
    avg.append((t2[m] - t0[m]).sum() / sp); spans.append(sp / tick_per_ns / 1e6)
print("per CU: resident waves mean %.2f of 8 (min %.2f max %.2f); kernel span %.2f ms" % (
    np.mean(avg), np.min(avg), np.max(avg), np.mean(spans)))
print("kernel entry -> block parameters known: mean %.2f us  median %.2f us" % (b[:, 7].mean() / tick_per_ns / 1e3, np.median(b[:, 7]) / tick_per_ns / 1e3))
tot = (t2 - t0).sum()
print("share of wave time: load wait %.3f  MFMA burst %.3f  row stores %.3f  (rest: slab load, setup)" % (
    b[:, 4].sum() / tot, b[:, 5].sum() / tot, b[:, 6].sum() / tot))
print("per 32-slot tile: load wait %.0f  MFMA burst %.0f  row stores %.0f ticks (128 MFMAs alone = 8192 cycles)" % (
    b[:, 4].sum() / tiles.sum(), b[:, 5].sum() / tiles.sum(), b[:, 6].sum() / tiles.sum()))
u = np.unique(key)[0]
for s in range(4):
    for ws in range(2):
        mm = (key == u) & (simd == s) & (wid == ws)
        o = np.argsort(t0[mm]); a0 = t0[mm][o]; a2 = t2[mm][o]
        gaps = a0[1:] - a2[:-1]
        print("  CU0 simd %d slot %d: waves %d busy %.2f, slot empty between waves: median %.1f us" % (
            s, ws, mm.sum(), (a2 - a0).sum() / (a2.max() - a0.min()), np.median(gaps) / tick_per_ns / 1e3))
if len(sys.argv) > 2 and sys.argv[2] == "timeline":
    mm = (key == u) & (simd == 0)
    o = np.argsort(t0[mm])
    base = t0[mm].min()
    print("CU0 simd 0 timeline (us): slot start end tiles")
    for i in o[:40]:
        print("  slot %d  %8.1f -> %8.1f  tiles %d" % (wid[mm][i], (t0[mm][i] - base) / tick_per_ns / 1e3,
                                                      (t2[mm][i] - base) / tick_per_ns / 1e3, tiles[mm][i]))
