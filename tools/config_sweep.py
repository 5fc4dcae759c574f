#!/usr/bin/env python
"""Step time (render + <render, G> + backward, the harness step of bench.py) of the other BASELINE.json configs and
variants: C1, C2, C5 with an fp32 / fp16 feature table, 512 + 1 channels, the C3 geometry at D = 16 ... 256.
Prints one JSON object.  (bench.py's `value` is C3; this table feeds BASELINE.md section 4.)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gags_amd import synthetic as syn
from gags_amd.gaussian_renderer import render
from bench import _CotangentLoss  # loss = <render, G> whose backward hands G itself to the rasterizer (no elementwise pass)

dev = torch.device("cuda", 0)


def run(name, n, w, h, d, half=False, steps=8, flags=0, full_grad=False, scale0=syn.SCALE0):
    pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev, scale0=scale0)
    pc.training_setup()
    # (render() hands the stored parameters to the projection kernel: the getters run inside it, every view)
    geo = [pc._xyz, pc._scaling, pc._rotation, pc._opacity]
    if full_grad:  # joint training: every geometry parameter gets its gradient too (SURVEY A9 in full)
        for q in geo:
            q.requires_grad_(True)
    if half:
        master = pc._semantic_feature
        pc.rewrite_semantic_feature(master.detach().half().requires_grad_(True))
    cam = syn.make_camera(w, h, device=dev)
    bg = torch.zeros(3, device=dev)
    G = syn.make_cotangent(d, h, w, seed=1, device=dev)

    def step():
        pc._semantic_feature.grad = None
        for q in geo:
            q.grad = None
        pkg = render(cam, pc, None, bg, feature_mode=True, raster_flags=flags)
        _CotangentLoss.apply(pkg["render"].permute(1, 2, 0), G.permute(1, 2, 0)).backward()
        return pkg

    for _ in range(2):
        pkg = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    peak = torch.cuda.max_memory_allocated() / 2**30
    res = {"ms_per_step": round(1e3 * dt, 3), "views_per_s": round(1 / dt, 1), "n_isects": pkg["info"]["n_isects"], "peak_GiB": round(peak, 1)}
    if pkg["info"].get("n_isects_trimmed") is not None:
        res["n_isects_trimmed"] = pkg["info"]["n_isects_trimmed"]  # (heavy views: the lists cut to what their tiles read)
    del pc, G, pkg
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    return name, res


C3 = (1_500_000, 1920, 1080)
C5 = (4_000_000, 1920, 1080)
CASES = [
    ("C1 10k/256x256/D=3", (10_000, 256, 256, 3), {}),
    ("C2 500k/1280x720/D=128", (500_000, 1280, 720, 128), {}),
    ("C3 geometry D=16", C3 + (16,), {}),
    ("C3 geometry D=64", C3 + (64,), {}),
    ("C3 geometry D=256", C3 + (256,), {}),
    ("C5 4M/1080p/D=512 fp32 table", C5 + (512,), {}),
    ("C5 4M/1080p/D=512 fp16 table", C5 + (512,), dict(half=True)),
    ("C5 4M/1080p/D=512 fp16 table, exact forward (GAGS_FWD_EXACT: halves widened, fp32 matrix instructions: round 5's default)", C5 + (512,), dict(half=True, flags=2048)),
    ("C5 4M/1080p/D=512 fp16 table, 16-bit matrix cores forward too (opt-in)", C5 + (512,), dict(half=True, flags=128)),
    ("C3 1.5M/1080p/D=512 fp16 table", C3 + (512,), dict(half=True)),
    ("C3 1.5M/1080p/D=512 fp16 table, 16-bit matrix cores forward too (opt-in)", C3 + (512,), dict(half=True, flags=128)),
    ("C3 1.5M/1080p/D=512 fp32 table (the bench line's workload)", C3 + (512,), {}),
    ("C3 1.5M/1080p/D=512 fp32 table, exact forward (GAGS_FWD_EXACT: fp32 matrix instructions, bit-identical to the oracle)", C3 + (512,), dict(flags=2048)),
    ("C3 1.5M/1080p/D=512 fp32 table, backward on the fp32 matrix instructions (GAGS_BWD_F32MFMA)", C3 + (512,), dict(flags=64)),
    ("C3 1.5M/1080p/D=512 fp32 table, round 4's rows kernel (GAGS_BWD_BLOCKWAVES: a wave per pixel block, rows merged in LDS)", C3 + (512,), dict(flags=4096)),
    ("C3H 1.5M/1080p/D=512, SURVEY-literal splats (63 M intersections)", C3 + (512,), dict(scale0=syn.SCALE0_SURVEY, steps=4)),
    ("C3H ..., round 4's rows kernel (GAGS_BWD_BLOCKWAVES)", C3 + (512,), dict(scale0=syn.SCALE0_SURVEY, steps=4, flags=4096)),
    ("C5H 4M/1080p/D=512 fp16 table, SURVEY-literal splats (169 M intersections)", C5 + (512,), dict(half=True, scale0=syn.SCALE0_SURVEY, steps=3)),
    ("C5 4M/1080p/D=513 (512+1) fp32", C5 + (513,), dict(steps=4)),
    ("C5 4M/1080p/D=513 (512+1) fp16 table -- BASELINE.json configs[4] as stated", C5 + (513,), dict(half=True, steps=4)),
    ("C5 4M/1080p/D=513 (512+1) fp16 table, exact forward (GAGS_FWD_EXACT: round 5's default)", C5 + (513,), dict(half=True, flags=2048, steps=4)),
    ("C5 4M/1080p/D=513 (512+1) fp16 table, 16-bit matrix cores forward too (opt-in)", C5 + (513,), dict(half=True, flags=128, steps=4)),
    ("C3 1.5M/1080p/D=512, ALL gradients (features + means, quats, scales, opacities)", C3 + (512,), dict(full_grad=True)),
    ("C3 ALL gradients, fp32 matrix instructions (rounds 1-2's kernels)", C3 + (512,), dict(full_grad=True, flags=64)),
    ("C3 ALL gradients, VALU + atomics backward (what round 1 ran)", C3 + (512,), dict(full_grad=True, flags=4, steps=3)),
    ("C2 500k/1280x720/D=128, ALL gradients", (500_000, 1280, 720, 128), dict(full_grad=True)),
    ("C3 geometry D=16, ALL gradients", C3 + (16,), dict(full_grad=True)),
]
sel = sys.argv[1] if len(sys.argv) > 1 else ""   # substring filter; a trailing "$" asks for the exact name
match = (lambda name: name == sel[:-1]) if sel.endswith("$") else (lambda name: sel in name)
out = dict(run(name, *shape, **kw) for name, shape, kw in CASES if match(name))
print(json.dumps(out))
