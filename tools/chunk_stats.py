#!/usr/bin/env python
"""Chunk statistics of the staged backward's rows kernel (tile-merged rows): emulates on the host the chunking the
kernel does on the fly and prints rows per chunk, run lengths (MFMA tile fill) and bursts per tile."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gags_amd import rasterization as R, synthetic as syn
from gags_amd.gaussian_renderer import render

cfg = syn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
n, d, w, h = cfg["n"], cfg["d"], cfg["width"], cfg["height"]
dev = torch.device("cuda", 0)
pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev, scale0=cfg.get("scale0", syn.SCALE0))
pc.training_setup()
cam = syn.make_camera(w, h, device=dev)
G = syn.make_cotangent(d, h, w, seed=1, device=dev)
cap = {}
orig = R._backward_staged


def spy(lib, rctx, offsets, n_isects, blk_rows, fwd_scratch, v_out, n_, d_, width, height, *extra, **kw):
    # (*extra: whatever _backward_staged grew since -- flags, flatten_ids; passed through untouched)
    ne = lib.gags_bwd_rowmap_elems(n_isects, width, height)
    rm = torch.empty(ne, dtype=torch.int32, device=v_out.device)
    tot = torch.empty(1, dtype=torch.int32, device=v_out.device)
    sb = lib.gags_bwd_rowmap_scratch_bytes(n_isects)
    tmp = torch.empty(max(sb, 4), dtype=torch.uint8, device=v_out.device)
    R.check(lib.gags_bwd_rowmap(n_isects, width, height, R.ptr(offsets), R.ptr(blk_rows), R.ptr(fwd_scratch), fwd_scratch.numel(),
                                R.ptr(rm), ne, R.ptr(tot), R.ptr(tmp), sb, None), "rowmap")
    torch.cuda.synchronize()
    cap.update(offsets=offsets.cpu().numpy().reshape(-1)[:-1], blk=blk_rows.cpu().numpy(), rm=rm.cpu().numpy(), I=n_isects)
    return orig(lib, rctx, offsets, n_isects, blk_rows, fwd_scratch, v_out, n_, d_, width, height, *extra, **kw)


R._backward_staged = spy
pkg = render(cam, pc, None, torch.zeros(3, device=dev), feature_mode=True)
(pkg["render"] * G).sum().backward()
torch.cuda.synchronize()
cap["flat"] = pkg["info"]["flatten_ids"].cpu().numpy()
off, blk, rm, I = cap["offsets"], cap["blk"], cap["rm"], cap["I"]
nt = off.size
slot_off = ((I + 1) * 4 + 255) // 256 * 256 // 4
trow, trs = rm[:I + 1], rm[slot_off:]
hit = np.diff(trow.astype(np.int64)) > 0
per_g = np.bincount(cap["flat"][hit], minlength=n)
print(f"tile rows {int(hit.sum())}, Gaussians with rows {int((per_g > 0).sum())}: exactly one row {np.mean(per_g[per_g > 0] == 1):.1%} of them "
      f"(= {np.sum(per_g == 1) / hit.sum():.1%} of the rows), two {np.mean(per_g[per_g > 0] == 2):.1%}, mean {per_g[per_g > 0].mean():.2f}")
CMAX = int(os.environ.get("CMAX", 64))
rows_per_chunk, runs, bursts, chunks_tile, old_tiles = [], [], 0, [], 0
rng = np.random.default_rng(0)
for t in rng.choice(nt, size=min(nt, 1500), replace=False):
    s, e = off[t], (I if t == nt - 1 else off[t + 1])
    lp = (e - s + 15) & ~15  # gags_slot_base (csrc/common.h)
    lists = []
    for b in range(4):
        c = blk[4 * t + b]
        sb = 4 * s + 64 * t + b * lp
        lists.append(trs[sb:sb + c])
        old_tiles += (c + 31) // 32
    r0, R1 = trow[s], trow[e]
    pb = [0, 0, 0, 0]
    nc = 0
    while r0 < R1:
        cand = [int(l[p + 32]) if p + 32 < len(l) else 2**31 - 1 for l, p in zip(lists, pb)]
        r1 = min(min(cand), r0 + CMAX, R1)
        for b in range(4):
            l = lists[b][pb[b]:pb[b] + 32]
            run = int((l < r1).sum())
            if run:
                bursts += 1
                runs.append(run)
            pb[b] += run
        rows_per_chunk.append(r1 - r0)
        r0 = r1
        nc += 1
    chunks_tile.append(nc)
runs = np.array(runs)
print(f"tiles sampled {len(chunks_tile)}, chunks/tile {np.mean(chunks_tile):.2f}, rows/chunk {np.mean(rows_per_chunk):.1f}, "
      f"bursts/tile {bursts / len(chunks_tile):.2f} (per-block rows kernel: {old_tiles / len(chunks_tile):.2f}), mean run {runs.mean():.1f} / 32, "
      f"runs <= 16: {(runs <= 16).mean():.2%}, runs <= 8: {(runs <= 8).mean():.2%}")
