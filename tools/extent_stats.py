#!/usr/bin/env python
"""How selective is the weights pass's conservative extent test?  Per (tile-list entry, 8x8 block): passes the
extent test (gets its alpha evaluated on all 64 pixels) vs. ends up as a slot (blends at least one pixel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gags_amd import synthetic as syn
from gags_amd import rasterization as R
from gags_amd.gaussian_renderer import render

cfg = syn.CONFIGS["C3"]
n, d, w, h = cfg["n"], 128, cfg["width"], cfg["height"]
dev = torch.device("cuda", 0)
pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev)
pc.training_setup()
cam = syn.make_camera(w, h, device=dev)
cap = {}
orig = R.tile_binning
def spy(*a, **k):
    out = orig(*a, **k)
    cap["ids"], cap["packed"] = out[0], out[4][out[1].long()]  # per-Gaussian records -> per sorted intersection
    return out
R.tile_binning = spy
orig_b = R._backward_staged
def spy_b(lib, rctx, offsets, n_isects, blk_rows, *a, **k):
    cap["blk_rows"] = blk_rows.clone()
    return orig_b(lib, rctx, offsets, n_isects, blk_rows, *a, **k)
R._backward_staged = spy_b
pkg = render(cam, pc, None, torch.zeros(3, device=dev), feature_mode=True)
pkg["render"].sum().backward()
ids, pk = cap["ids"], cap["packed"]
tile = (ids >> 32).long()
tw = (w + 15) // 16
tx, ty = (tile % tw).float() * 16, (tile // tw).float() * 16
x, y, ex, ey = pk[:, 0], pk[:, 1], pk[:, 6], pk[:, 7]
total_hits = 0
for blk in range(4):
    bx0 = tx + (blk & 1) * 8; by0 = ty + (blk >> 1) * 8
    hit = (x + ex >= bx0 + 0.5) & (x - ex <= bx0 + 7.5) & (y + ey >= by0 + 0.5) & (y - ey <= by0 + 7.5)
    total_hits += int(hit.sum())
I = ids.numel()
slots = int(cap["blk_rows"].sum())
print("tile-list entries %d; (entry, block) pairs %d; pass the extent test %d (%.1f %%); become slots %d (%.1f %% of the hits)" % (
    I, 4 * I, total_hits, 100.0 * total_hits / (4 * I), slots, 100.0 * slots / total_hits))

# of the hits that pass the box test: how many reach alpha >= 1/255 on at least one pixel centre of the block
# (what an exact geometric test could keep at most); the rest of the non-blending hits is occlusion (T <= 1e-4)
a, b, c, o = pk[:, 2], pk[:, 3], pk[:, 4], pk[:, 5]
geo = 0
pxs = torch.arange(8, device=dev).float() + 0.5
for blk in range(4):
    bx0 = tx + (blk & 1) * 8; by0 = ty + (blk >> 1) * 8
    hit = (x + ex >= bx0 + 0.5) & (x - ex <= bx0 + 7.5) & (y + ey >= by0 + 0.5) & (y - ey <= by0 + 7.5)
    idx = hit.nonzero().squeeze(1)
    for ch in torch.split(idx, 1 << 20):
        dx = x[ch, None] - (bx0[ch, None] + pxs[None, :])          # [m, 8]
        dy = y[ch, None] - (by0[ch, None] + pxs[None, :])          # [m, 8]
        sig = 0.5 * (a[ch, None, None] * dx[:, None, :] ** 2 + c[ch, None, None] * dy[:, :, None] ** 2) \
            + b[ch, None, None] * dx[:, None, :] * dy[:, :, None]  # [m, 8(y), 8(x)]
        al = torch.clamp(o[ch, None, None] * torch.exp(-sig), max=0.999)
        ok = ((sig >= 0) & (al >= 1.0 / 255.0)).flatten(1).any(1)
        geo += int(ok.sum())
print("hits with alpha >= 1/255 on some pixel of the block: %d (%.1f %% of the box hits); slots %d => occluded %.1f %%" % (
    geo, 100.0 * geo / total_hits, slots, 100.0 * (geo - slots) / max(geo, 1)))
