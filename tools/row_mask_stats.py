#!/usr/bin/env python
"""Which 8x8 blocks of its tile does a (tile, Gaussian) gradient row touch?  The rows kernel (csrc/raster_bwd_rows_cw.h) takes 32
consecutive rows of a tile as one chunk and multiplies all four blocks for it; this prices what ordering a tile's rows by the
blocks they touch would save: (block, chunk) products issued, today against rows grouped into TOP / BOTTOM / LEFT / RIGHT / FULL
classes (chunks of a half class skip two blocks).

    python tools/row_mask_stats.py [C3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gags_amd import synthetic as syn
from gags_amd import rasterization as R
from gags_amd.gaussian_renderer import render

cfg = syn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
n, d, w, h = cfg["n"], 128, cfg["width"], cfg["height"]
dev = torch.device("cuda", 0)
pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev, scale0=cfg.get("scale0", syn.SCALE0))
pc.training_setup()
cam = syn.make_camera(w, h, device=dev)
cap = {}
orig = R._backward_staged


def spy(lib, rctx, offsets, n_isects, blk_rows, fwd_scratch, v_out, n_, d_, width, height, *extra, **kw):
    ne = lib.gags_bwd_rowmap_elems(n_isects, width, height)
    rm = torch.empty(ne, dtype=torch.int32, device=v_out.device)
    tot = torch.empty(1, dtype=torch.int32, device=v_out.device)
    sb = lib.gags_bwd_rowmap_scratch_bytes(n_isects)
    tmp = torch.empty(max(sb, 4), dtype=torch.uint8, device=v_out.device)
    R.check(lib.gags_bwd_rowmap(n_isects, width, height, R.ptr(offsets), R.ptr(blk_rows), R.ptr(fwd_scratch), fwd_scratch.numel(),
                                R.ptr(rm), ne, R.ptr(tot), R.ptr(tmp), sb, None), "rowmap")
    torch.cuda.synchronize()
    cap.update(offsets=offsets.reshape(-1)[:-1].long().clone(), blk=blk_rows.long().clone(), rm=rm.long().clone(), I=n_isects,
               rows=int(tot.item()))
    return orig(lib, rctx, offsets, n_isects, blk_rows, fwd_scratch, v_out, n_, d_, width, height, *extra, **kw)


R._backward_staged = spy
pkg = render(cam, pc, None, torch.zeros(3, device=dev), feature_mode=True)
pkg["render"].sum().backward()
off, blk, rm, I, rows = cap["offsets"], cap["blk"], cap["rm"], cap["I"], cap["rows"]
nt = off.numel()
slot_off = ((I + 1) * 4 + 255) // 256 * 256 // 4
trow, trs = rm[:I + 1], rm[slot_off:]
L = torch.diff(torch.cat([off, torch.tensor([I], device=dev)]))
padL = (L + 15) // 16 * 16
tile = torch.arange(nt, device=dev)
base = (4 * off + 64 * tile)[:, None] + torch.arange(4, device=dev)[None, :] * padL[:, None]  # [tiles, 4] (gags_slot_base)
mask = torch.zeros(rows, dtype=torch.int64, device=dev)
for b in range(4):
    cnt = blk[b::4]
    idx = torch.repeat_interleave(base[:, b], cnt) + (torch.arange(int(cnt.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt))
    r = trs[idx]
    r = r[r < 0x7fffffff]
    mask[r] |= 1 << b
R0 = trow[off]
R1 = trow[torch.cat([off[1:], torch.tensor([I], device=dev)])]
rows_tile = R1 - R0
tile_of_row = torch.repeat_interleave(tile, rows_tile)
print("rows", rows, "tiles", nt, "rows per tile %.1f" % rows_tile.float().mean().item())
hist = torch.bincount(mask, minlength=16).float() / rows
print("block masks (bit b = block b; blocks 0 1 / 2 3 are the tile's upper / lower row):")
print("  " + "  ".join("%s:%.3f" % (format(m, "04b"), hist[m].item()) for m in range(1, 16)))
pop = torch.tensor([bin(m).count("1") for m in range(16)], device=dev)
print("blocks per row: mean %.2f" % pop[mask].float().mean().item())
TOP, BOT, LEFT, RIGHT = 0b0011, 0b1100, 0b0101, 0b1010
cls = torch.full_like(mask, 4)
cls[(mask & ~RIGHT) == 0] = 3
cls[(mask & ~LEFT) == 0] = 2
cls[(mask & ~BOT) == 0] = 1
cls[(mask & ~TOP) == 0] = 0
print("classes TOP / BOTTOM / LEFT / RIGHT / FULL: " + " ".join("%.3f" % (cls == c).float().mean().item() for c in range(5)))
# cost: (block, chunk) products.  today: 4 per chunk of 32 consecutive rows
today = (4 * ((rows_tile + 31) // 32)).sum().item()
# grouped: rows of a tile ordered by class; a chunk multiplies the union of its rows' classes' blocks
key = tile_of_row * 8 + cls
order = torch.argsort(key, stable=True)
cls_s, tile_s = cls[order], tile_of_row[order]
pos = torch.arange(rows, device=dev) - torch.repeat_interleave(R0, rows_tile)  # (rows are numbered tile by tile: position inside the tile)
chunk = tile_s * 4096 + pos // 32
cmask = torch.tensor([TOP, BOT, LEFT, RIGHT, 15], device=dev)[cls_s]
uniq, inv = torch.unique(chunk, return_inverse=True)
union = torch.zeros(uniq.numel(), dtype=torch.int64, device=dev)
for b in range(4):
    union |= (torch.zeros(uniq.numel(), dtype=torch.int64, device=dev).index_add_(0, inv, (cmask >> b) & 1) > 0).long() << b
grouped = pop[union].sum().item()
# exact masks instead of classes (rows ordered by class, a chunk multiplies the union of its rows' own masks)
mask_s = mask[order]
union2 = torch.zeros(uniq.numel(), dtype=torch.int64, device=dev)
for b in range(4):
    union2 |= (torch.zeros(uniq.numel(), dtype=torch.int64, device=dev).index_add_(0, inv, (mask_s >> b) & 1) > 0).long() << b
grouped2 = pop[union2].sum().item()
print("(block, chunk) products: today %d; rows grouped by class %d (%.3f); ... and blocks no row of the chunk touches skipped %d (%.3f)"
      % (today, grouped, grouped / today, grouped2, grouped2 / today))
