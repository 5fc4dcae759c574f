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
def spy(lib, rctx, offsets, n_isects, blk_rows, fwd_scratch, *a, **k):
    cap.update(offsets=offsets[:-1].clone().reshape(-1).long(), n_isects=n_isects, blk_rows=blk_rows.clone().long(),
               scratch=fwd_scratch)
    return orig(lib, rctx, offsets, n_isects, blk_rows, fwd_scratch, *a, **k)
R._backward_staged = spy
pkg = render(cam, pc, None, torch.zeros(3, device=dev), feature_mode=True)
pkg["render"].sum().backward()
off, I, br = cap["offsets"], cap["n_isects"], cap["blk_rows"]
nt = off.numel()
L = torch.diff(torch.cat([off, torch.tensor([I], device=dev)]))
padL = (L + 15) // 16 * 16  # gags_slot_base (csrc/common.h): regions padded to 16 slots
tile = torch.arange(nt, device=dev)
base = (4 * off + 64 * tile)[:, None] + torch.arange(4, device=dev)[None, :] * padL[:, None]   # [tiles, 4]
base = base.reshape(-1); cnt = br.reshape(-1)
slots_total = 4 * I + 64 * nt + 64  # gags_slot_count (csrc/common.h)
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
# ---- "the backward may reorder slots inside a run: measure what grouping by 8x4-half occupancy saves
# when single-half groups issue 16 K-steps instead of 32, at zero extra slots".  A run = 32 consecutive slots of a block
# (the rows kernel's chunks end earlier on average -- 26 of 32 -- so this is the optimistic fill).  Unit of cost: one
# K-step of a 32-row MFMA tile (v_mfma_f32_32x32x2_f32: two pixels x 32 slot rows), per 32 channels.
blk_id = torch.repeat_interleave(torch.arange(base.numel(), device=dev), cnt)
first = torch.cumsum(cnt, 0) - cnt
j = torch.arange(W.shape[0], device=dev) - torch.repeat_interleave(first, cnt)
run = blk_id * 4096 + j // 32
_, inv_run = torch.unique(run, return_inverse=True)
n_runs = int(inv_run.max().item()) + 1
cls_up, cls_lo, cls_both = (up & ~lo), (lo & ~up), (up & lo)
cnt_run = torch.bincount(inv_run, minlength=n_runs).float()
n_up = torch.bincount(inv_run, weights=cls_up.float(), minlength=n_runs)
n_lo = torch.bincount(inv_run, weights=cls_lo.float(), minlength=n_runs)
n_bo = torch.bincount(inv_run, weights=cls_both.float(), minlength=n_runs)
base_cost = 32.0 * n_runs                                   # today: every run is one 32-row tile over all 64 pixels
# (A) M = 32 tiles, a run split by class into up to three 32-row tiles (16 / 16 / 32 K-steps), or left alone
split_cost = 16.0 * (n_up > 0) + 16.0 * (n_lo > 0) + 32.0 * (n_bo > 0)
costA = torch.minimum(split_cost, torch.full_like(split_cost, 32.0)).sum().item()
# (B) M = 16 tiles (v_mfma_f32_16x16x4_f32, same flops per cycle): half a 32-row tile per 16 rows; single-half tiles
#     cost half of that again; leftovers of the two single-half classes may share one full-K tile
ceil16 = lambda t: torch.ceil(t / 16.0)
costB_plain = (16.0 * ceil16(cnt_run)).sum().item()
t_bo, t_up, t_lo = ceil16(n_bo), ceil16(n_up), ceil16(n_lo)
costB = (16.0 * t_bo + 8.0 * t_up + 8.0 * t_lo).sum().item()
# ... and with the remainders of all three classes packed into full-K tiles when that is cheaper
rem = (n_bo % 16) + (n_up % 16) + (n_lo % 16)
full = torch.floor(n_bo / 16) * 16.0 + torch.floor(n_up / 16) * 8.0 + torch.floor(n_lo / 16) * 8.0
costB_pack = torch.minimum(16.0 * t_bo + 8.0 * t_up + 8.0 * t_lo, full + 16.0 * ceil16(rem)).sum().item()
print("runs of <= 32 slots: %d, mean length %.1f; classes per run: upper-only %.2f lower-only %.2f both %.2f" % (
    n_runs, cnt_run.mean().item(), n_up.mean().item(), n_lo.mean().item(), n_bo.mean().item()))
print("issued K-steps relative to today's kernel (1.000 = one 32-row x 64-pixel tile per run):")
print("  (A) 32-row tiles, runs split by half occupancy where that is cheaper : %.3f" % (costA / base_cost))
print("  (B) 16-row tiles (16x16x4), no grouping                              : %.3f" % (costB_plain / base_cost))
print("  (B) 16-row tiles, grouped by half occupancy (single-half: 16 K-steps)  : %.3f" % (costB / base_cost))
print("  (B) ... remainders of the three classes packed into full-K tiles      : %.3f" % (costB_pack / base_cost))
