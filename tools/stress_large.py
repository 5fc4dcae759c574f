#!/usr/bin/env python
"""Large-splat stress run (C3 geometry, D=128, Gaussian scale x2.5 / x4: 25 M / 53 M intersections): step time, peak
memory (the sparse slot space is 4 * n_isects * 256 B), the transpose identity, the exact matrix-core forward (GAGS_FWD_EXACT)
== the VALU kernels bit for bit, and the default forward (split bf16 operands) within its tolerance of them."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gags_amd import synthetic as syn, _lib
from gags_amd.gaussian_renderer import render
dev = torch.device('cuda', 0)
n, d, w, h = 1_500_000, 128, 1920, 1080
for mult in (2.5, 4.0):
    pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev, scale0=syn.SCALE0 * mult)
    pc.training_setup()
    cam = syn.make_camera(w, h, device=dev)
    G = syn.make_cotangent(d, h, w, seed=1, device=dev)
    for it in range(3):
        pc._semantic_feature.grad = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pkg = render(cam, pc, None, torch.zeros(3, device=dev), feature_mode=True)
        (pkg["render"] * G).sum().backward()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    g = pc._semantic_feature.grad
    # transpose identity as a correctness check at this size
    lhs = torch.dot(pkg["render"].detach().permute(1, 2, 0).reshape(-1).double(), G.permute(1, 2, 0).reshape(-1).double())
    rhs = torch.dot(pc._semantic_feature.detach().reshape(-1).double(), g.reshape(-1).double())
    # VALU kernel agreement on the forward: bitwise for the exact matrix-core kernel, rel-L2 for the default
    with torch.no_grad():
        out2 = render(cam, pc, None, torch.zeros(3, device=dev), feature_mode=True, raster_flags=_lib.GAGS_FWD_NO_MFMA)["render"]
        out3 = render(cam, pc, None, torch.zeros(3, device=dev), feature_mode=True, raster_flags=_lib.GAGS_FWD_EXACT)["render"]
    rel = ((pkg["render"].detach().double() - out2.double()).norm() / out2.double().norm()).item()
    print("scale x%.1f: n_isects %d  step %.1f ms  max mem %.1f GB  <Rf,G>-<f,RtG> rel %.2e  exact mfma==valu %s  default vs valu rel-L2 %.2e" % (
        mult, pkg["info"]["n_isects"], dt * 1e3, torch.cuda.max_memory_allocated() / 1e9,
        abs(lhs.item() - rhs.item()) / max(abs(lhs.item()), 1e-30), torch.equal(out2, out3), rel))
    del pc, pkg, G, g, out2, out3
    torch.cuda.empty_cache()
