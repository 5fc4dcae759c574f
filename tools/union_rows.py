#!/usr/bin/env python
"""How many rows of the feature gradient a by-view step has to exchange ("gradients are sparse in rows"):
per view the Gaussians that blended into a pixel (gags_blended_mask), and the union over the first 2 / 4 / 8 of C4's
yawed views -- what OverlappedGradReducer(rows="union") puts on the wire instead of all N rows."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gags_amd import rasterization, synthetic as syn
from gags_amd.gaussian_renderer import render

cfg = syn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
dev = torch.device("cuda", 0)
n, w, h, d = cfg["n"], cfg["width"], cfg["height"], 256   # the mask does not depend on D; 256 keeps the run short
pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev)
pc.training_setup()
bg = torch.zeros(3, device=dev)
masks = []
rasterization.default_context().grad_range_hook = lambda g, c0, c1: None
rasterization.default_context().grad_rows_hook = lambda m: masks.append(m.clone())
for v in range(8):
    pc._semantic_feature.grad = None
    out = render(syn.make_camera(w, h, view=v, device=dev), pc, None, bg, feature_mode=True)
    out["render"].sum().backward()
m = torch.stack(masks).bool()
res = {"n": n, "per_view": [int(x) for x in m.sum(1).tolist()]}
for k in (2, 4, 8):
    res[f"union_{k}_views"] = int(m[:k].any(0).sum())
    res[f"fraction_{k}_views"] = round(res[f"union_{k}_views"] / n, 4)
print(json.dumps(res))
