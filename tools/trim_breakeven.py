#!/usr/bin/env python
"""Where list trimming (gags_amd.rasterization._trim_lists) starts to pay: the C3 geometry at splat scales between the headline
workload's (0.0009 z_mean, 7.4 M intersections) and SURVEY 8d's literal one (0.004: 63 M), step time with the lists trimmed
and untrimmed.  Prints one JSON object."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import _CotangentLoss
from gags_amd import synthetic as syn
from gags_amd.gaussian_renderer import render
from gags_amd.rasterization import RasterContext

cfg = syn.CONFIGS["C3"]
n, d, w, h = cfg["n"], cfg["d"], cfg["width"], cfg["height"]
dev = torch.device("cuda", 0)
cam = syn.make_camera(w, h, device=dev)
bg = torch.zeros(3, device=dev)
G = syn.make_cotangent(d, h, w, seed=1, device=dev)
out = {}
for mult in [float(a) for a in sys.argv[1:]] or [1.0, 1.5, 2.0, 3.0, 4.44]:
    pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev, scale0=syn.SCALE0 * mult)
    pc.training_setup()
    row = {}
    for trim in (False, True):
        ctx = RasterContext()
        ctx.trim_lists = trim

        def step():
            pc._semantic_feature.grad = None
            pkg = render(cam, pc, None, bg, feature_mode=True, context=ctx)
            _CotangentLoss.apply(pkg["render"].permute(1, 2, 0), G.permute(1, 2, 0)).backward()
            return pkg

        for _ in range(2):
            pkg = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        row["trimmed" if trim else "full"] = round(1e3 * (time.perf_counter() - t0) / 6, 3)
        row["n_isects"] = pkg["info"]["n_isects"]
        if trim:
            row["n_isects_trimmed"] = pkg["info"]["n_isects_trimmed"]
        del pkg
        torch.cuda.empty_cache()
    out[f"scale x{mult}"] = row
    del pc
print(json.dumps(out, indent=1))
