#!/usr/bin/env python
"""Timing of the SURVEY 8f rows around the rasterizer at the reference's real width (D = 16, train.py:68):
render(16-d) -> CNN_scale_decoder / CNN_decoder -> distillation losses -> backward through decoder and rasterizer,
i.e. one iteration of train.py:142-174 on synthetic inputs (1.5 M Gaussians, 1080p).  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gags_amd import losses as L, synthetic as syn
from gags_amd.decoders import CNN_decoder, CNN_scale_decoder
from gags_amd.distill import distillation_loss
from gags_amd.gaussian_renderer import render

dev = torch.device("cuda", 0)
cfg = syn.CONFIGS["C3"]
n, w, h, d = cfg["n"], cfg["width"], cfg["height"], 16
pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev)
pc.training_setup()
pc.cache_activations("--cache-activations" in sys.argv)  # default: getters evaluated on every render, as the reference does
cam = syn.make_camera(w, h, device=dev)
bg = torch.zeros(3, device=dev)
# default: fp32-equivalent ("exact"); --bf16x2: two-term split (16 significand bits; the reference's convs run in TF32);
# --bf16: the fast opt-in mode
PRECISION = "bf16" if "--bf16" in sys.argv else ("bf16x2" if "--bf16x2" in sys.argv else ("f16" if "--f16" in sys.argv else "exact"))
dec, sdec = CNN_decoder(16, 512, PRECISION).to(dev), CNN_scale_decoder(16, 3, PRECISION).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
n_emb = 300
img_embed = torch.nn.functional.normalize(torch.randn(n_emb, 512, device=dev, generator=g), dim=-1)
seg = torch.randint(-1, n_emb, (4, h // 8, w // 8), device=dev, generator=g).float().repeat_interleave(8, 1).repeat_interleave(8, 2)
seg = seg[:, :h, :w].contiguous()


FUSED = "--two-step" not in sys.argv   # default: the fused head + loss; --two-step: decoder, then distill_l1_map


def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e


def iteration(times=None):
    from gags_amd import decoders as _D
    _D.invalidate_packed()  # as after an optimizer step: the decoders' weights are repacked every iteration
    marks = [ev()]
    pkg = render(cam, pc, None, bg, feature_mode=True)
    fmap = pkg["render"]; marks.append(ev())
    # train.py:149-172 after iteration 15001 (all three loss terms): gags_amd/distill.py, the composition that
    # tests/test_iteration_gpu.py checks against the reference's own functions chained on one input
    loss, _ = distillation_loss(fmap, seg, img_embed, dec, sdec, iteration=20000, fused_head=FUSED); marks.append(ev())
    for m in (dec, sdec):
        m.zero_grad(set_to_none=True)
    pc._semantic_feature.grad = None
    loss.backward(); marks.append(ev())
    if times is not None:  # (read after the loop's one synchronize: a sync per iteration would drain the queue every time)
        times.append(marks)
    return loss


for _ in range(2):
    iteration()
torch.cuda.synchronize()
all_marks = []
t0 = time.perf_counter()
K = 8
for _ in range(K):
    iteration(all_marks)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
times = {}
for marks in all_marks:
    for k, (a, b) in zip(("render16", "decoders_and_losses_fwd", "backward"), zip(marks, marks[1:])):
        times.setdefault(k, []).append(a.elapsed_time(b))
print(json.dumps({"fused_head_loss": FUSED, "decoder_precision": PRECISION,
                  "workload": "train.py:142-174 iteration, 1.5M Gaussians, 1920x1080, D=16 -> CNN decoders -> losses",
                  "ms_per_iteration": 1e3 * dt, "iterations_per_s": 1 / dt,
                  "stages_ms": {k: sum(v) / len(v) for k, v in times.items()}}))
