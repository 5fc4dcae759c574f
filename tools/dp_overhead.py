#!/usr/bin/env python
"""The by-view step's single-GPU overhead (bench.py's `view_dp_overhead_ms`) as a function of the width of the channel ranges
the backward is asked for (RasterContext.grad_range_channels): loop-back exchange, rank 0 of 8, C3.

    python tools/dp_overhead.py [128 256 512 256,128,128 ...]      (a comma list = a tuple of range widths, in order)
    GAGS_DP_ROWS_GROUP=128|256|512: channels per launch of the rows kernel (RasterContext.grad_rows_group)
    GAGS_DP_NO_WIRE=1: the round-5 exchange shape (pack kernel instead of the reduce stage writing the exchanged rows)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import _CotangentLoss
from gags_amd import synthetic as syn
from gags_amd.dist import OverlappedGradReducer
from gags_amd.gaussian_renderer import render
from gags_amd.rasterization import default_context

cfg = syn.CONFIGS["C3"]
n, d, w, h = cfg["n"], cfg["d"], cfg["width"], cfg["height"]
dev = torch.device("cuda", 0)
pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev)
pc.training_setup()
cam = syn.make_camera(w, h, device=dev)
bg = torch.zeros(3, device=dev)
G = syn.make_cotangent(d, h, w, seed=1, device=dev)
union_rows = int(round(0.2977 * n))


def plain():
    pc._semantic_feature.grad = None
    pkg = render(cam, pc, None, bg, feature_mode=True)
    _CotangentLoss.apply(pkg["render"].permute(1, 2, 0), G.permute(1, 2, 0)).backward()


def timed(fn, k=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k


out = {"plain_ms": timed(plain)}
specs = [tuple(int(x) for x in a.split(",")) if "," in a else int(a) for a in sys.argv[1:]] or [128, 256, 512, (256, 128, 128)]
for rng in specs:
    default_context().grad_range_channels = rng
    if os.environ.get("GAGS_DP_ROWS_GROUP"):
        default_context().grad_rows_group = int(os.environ["GAGS_DP_ROWS_GROUP"])
    red = OverlappedGradReducer(mode="rs_ag", rows="union", param=pc._semantic_feature, loopback=(8, union_rows))
    if os.environ.get("GAGS_DP_NO_WIRE") == "1":
        red.wire_for_range = lambda c0, c1: None

    def dp():
        pc._semantic_feature.grad = None
        pkg = render(cam, pc, None, bg, feature_mode=True)
        loss = _CotangentLoss.apply(pkg["render"].permute(1, 2, 0), G.permute(1, 2, 0))
        with red:
            loss.backward()
        red.finish(pc._semantic_feature.grad)

    ms = timed(dp)
    out[f"range_{rng}"] = {"step_ms": ms, "overhead_ms": ms - out["plain_ms"], "exposed_ms": red.exposed_ms(), "range_ms": red.range_ms}
    del red
print(json.dumps(out, indent=1))
