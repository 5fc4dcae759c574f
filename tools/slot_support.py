#!/usr/bin/env python
"""How sparse are the weight rows of the C3 view?  For every used slot (Gaussian x 8x8 block): which 8x4 halves /
4x4 quadrants / pixel rows carry a nonzero weight.  (Potential of skipping K-steps in the rows kernel.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gags_amd import synthetic as syn
from gags_amd import rasterization as R
from gags_amd.gaussian_renderer import render

cfg = syn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
n, d, w, h = cfg["n"], 128, cfg["width"], cfg["height"]
dev = torch.device("cuda", 0)
pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev)
pc.training_setup()
cam = syn.make_camera(w, h, device=dev)
cap = {}
orig = R._backward_staged
def spy(lib, offsets, n_isects, blk_rows, fwd_scratch, *a, **k):
    cap.update(offsets=offsets.clone().reshape(-1).long(), n_isects=n_isects, blk_rows=blk_rows.clone().long(),
               scratch=fwd_scratch)
    return orig(lib, offsets, n_isects, blk_rows, fwd_scratch, *a, **k)
R._backward_staged = spy
pkg = render(cam, pc, None, torch.zeros(3, device=dev), feature_mode=True)
pkg["render"].sum().backward()
off, I, br = cap["offsets"], cap["n_isects"], cap["blk_rows"]
nt = off.numel()
L = torch.diff(torch.cat([off, torch.tensor([I], device=dev)]))
evenL = (L + 1) // 2 * 2
tile = torch.arange(nt, device=dev)
base = (4 * (off + tile))[:, None] + torch.arange(4, device=dev)[None, :] * evenL[:, None]   # [tiles, 4]
base = base.reshape(-1); cnt = br.reshape(-1)
slots_total = 4 * (I + nt) + 64
wt = cap["scratch"][: slots_total * 256].view(torch.float32).view(slots_total, 64)
mark = torch.zeros(slots_total + 1, dtype=torch.int32, device=dev)
mark.index_add_(0, base, torch.ones_like(base, dtype=torch.int32))
mark.index_add_(0, base + cnt, -torch.ones_like(base, dtype=torch.int32))
valid = torch.cumsum(mark[:-1], 0) > 0
W = wt[valid]                      # [rows, 64], element 2p + h: pixel p (8 wide x 4 high) of half h
nz = W != 0
print("used slots", W.shape[0], " nonzero weights per slot: mean %.1f of 64" % nz.sum(1).float().mean().item())
up, lo = nz[:, 0::2].any(1), nz[:, 1::2].any(1)
print("support by 8x4 half: upper only %.3f  lower only %.3f  both %.3f  none %.3f" % (
    (up & ~lo).float().mean(), (lo & ~up).float().mean(), (up & lo).float().mean(), (~up & ~lo).float().mean()))
p = torch.arange(32, device=dev)
left = (p & 7) < 4
q = torch.stack([nz[:, 0::2][:, left].any(1), nz[:, 0::2][:, ~left].any(1), nz[:, 1::2][:, left].any(1), nz[:, 1::2][:, ~left].any(1)], 1)
k = q.sum(1)
print("4x4 quadrants touched: " + "  ".join("%d: %.3f" % (i, (k == i).float().mean().item()) for i in range(5)),
      " mean %.2f of 4" % k.float().mean().item())
# pixel rows (8 rows of 8 px): row r = half h, p >> 3
rows = torch.stack([nz[:, hh::2][:, (p >> 3) == rr].any(1) for hh in range(2) for rr in range(4)], 1)
print("pixel rows touched (of 8): mean %.2f" % rows.sum(1).float().mean().item())
# MFMA K-steps needed if a 32-slot tile could skip K-steps that are zero for ALL its slots (depth order kept)
