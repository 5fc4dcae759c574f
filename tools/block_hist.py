#!/usr/bin/env python
"""Distribution of the per-block slot counts (blk_rows) of the C3 view: the work items of the MFMA kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gags_amd import _lib, synthetic as syn
from gags_amd import rasterization as R
from gags_amd.gaussian_renderer import render

cfg = syn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
n, d, w, h = cfg["n"], cfg["d"], cfg["width"], cfg["height"]
dev = torch.device("cuda", 0)
pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev)
cam = syn.make_camera(w, h, device=dev)
captured = {}
orig = R._backward_staged
def spy(lib, rctx, offsets, n_isects, blk_rows, *a, **k):
    captured["blk_rows"] = blk_rows.clone(); captured["offsets"] = offsets[:-1].clone(); captured["n_isects"] = n_isects
    return orig(lib, rctx, offsets, n_isects, blk_rows, *a, **k)
R._backward_staged = spy
pc.training_setup()
pkg = render(cam, pc, None, torch.zeros(3, device=dev), feature_mode=True)
pkg["render"].sum().backward()
b = captured["blk_rows"].cpu().long()
off = captured["offsets"].cpu().long().reshape(-1)
L = torch.diff(torch.cat([off, torch.tensor([captured["n_isects"]])]))
print("blocks", b.numel(), "rows", int(b.sum()), "empty", int((b == 0).sum()))
qs = torch.tensor([0.1, 0.5, 0.9, 0.99, 0.999, 1.0])
print("blk_rows quantiles", [int(x) for x in torch.quantile(b.float(), qs)])
print("tile list length quantiles", [int(x) for x in torch.quantile(L.float(), qs)], "mean", float(L.float().mean()))
tiles32 = (b + 31) // 32
print("32-slot tiles", int(tiles32.sum()), "ideal", int(b.sum()) / 32)
tw = (w + 15) // 16; th = (h + 15) // 16
bt = b.view(th, tw, 4).sum(-1)
print("rows per tile-row (first/last 5):", bt.sum(1)[:5].tolist(), bt.sum(1)[-5:].tolist())
bands = [int(bt[i * 9:(i + 1) * 9].sum()) for i in range(8)]
print("rows per 9-tile band:", bands)
