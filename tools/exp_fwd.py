"""Experiment driver (GPU box): the forward feature pass on the 16-bit matrix cores against the exact kernel.

    python tools/exp_fwd.py [--config C3] [--iters 10]

Prints, for the default (split bf16 operands) and the GAGS_FWD_EXACT forward: per-kernel HIP-event times of the weights and
feature passes, the rel-L2 / max-abs distance between the two renders, and on a small scene both kernels' rel-L2 against
the float64 dense statement (oracle/dense_ref.py)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gags_amd import _lib, profiler, synthetic as syn  # noqa: E402
from gags_amd.gaussian_renderer import render  # noqa: E402


def time_forward(cam, pc, bg, flags, iters):
    for _ in range(2):
        render(cam, pc, None, bg, feature_mode=True, raster_flags=flags)
    torch.cuda.synchronize()
    profiler.enable(True)
    for _ in range(iters):
        with torch.no_grad():
            render(cam, pc, None, bg, feature_mode=True, raster_flags=flags)
    s = profiler.summary()
    profiler.enable(False)
    return {k: round(v[0], 4) for k, v in s.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--bg", type=float, default=0.0)
    ap.add_argument("--truth", type=int, default=0, help="compare against the float64-accumulating oracle on every n-th tile")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = syn.CONFIGS[args.config]
    w, h, n, d = cfg["width"], cfg["height"], cfg["n"], cfg["d"]
    cam = syn.make_camera(w, h, view=None, device=dev)
    pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device="cuda", scale0=cfg.get("scale0", syn.SCALE0))
    bg = torch.full((3,), args.bg, device=dev)
    out = {"config": args.config}
    with torch.no_grad():
        a = render(cam, pc, None, bg, feature_mode=True)["render"]
        b = render(cam, pc, None, bg, feature_mode=True, raster_flags=_lib.GAGS_FWD_EXACT)["render"]
        diff = (a.double() - b.double())
        out["x16_vs_exact_rel_l2"] = float(diff.norm() / b.double().norm())
        out["x16_vs_exact_max_abs"] = float(diff.abs().max())
        out["exact_max_abs"] = float(b.abs().max())
        del a, b, diff
    if args.truth:
        # both kernels against the float64 sum of the same fp32 products (oracle, every `truth`-th tile)
        from oracle import oracle as orc
        with torch.no_grad():
            pk = render(cam, pc, None, bg, feature_mode=True)
            info = pk["info"]
            step = args.truth
            ref = orc.raster_fwd_acc64(info["means2d"][0].cpu().numpy(), info["conics"][0].cpu().numpy(),
                                       info["opacities"][0].cpu().numpy(), pc.get_semantic_feature.detach().cpu().numpy(),
                                       np.full(d, args.bg, np.float32), w, h, info["isect_offsets"][0].cpu().numpy(),
                                       info["flatten_ids"].cpu().numpy(), tile_begin=0, tile_step=step)
            tw = (w + 15) // 16
            sel = np.zeros((h, w), bool)
            for t in range(0, tw * ((h + 15) // 16), step):
                ty, tx = divmod(t, tw)
                sel[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16] = True
            r = ref[sel]
            for name, fl in (("x16", 0), ("exact", _lib.GAGS_FWD_EXACT)):
                got = render(cam, pc, None, bg, feature_mode=True, raster_flags=fl)["render"].permute(1, 2, 0).cpu().numpy()[sel].astype(np.float64)
                out[f"{name}_vs_acc64_rel_l2"] = float(np.linalg.norm(got - r) / np.linalg.norm(r))
                out[f"{name}_vs_acc64_max_abs"] = float(np.abs(got - r).max())
            out["acc64_pixels"] = int(sel.sum())
            del ref, r, got, pk, info
    out["default_ms"] = time_forward(cam, pc, bg, 0, args.iters)
    out["exact_ms"] = time_forward(cam, pc, bg, _lib.GAGS_FWD_EXACT, args.iters)

    # The contraction alone against float64: the weights of a small scene are read out exactly by rendering the identity
    # (channel g of a pixel = alpha T of Gaussian g: one term per sum, exact in either kernel), then a random table is
    # rendered by both kernels and compared with W.double() @ F.double().
    ws, hs, ns, ds = 96, 64, 1024, 256
    cams = syn.make_camera(ws, hs, view=2, device=dev)
    pcs = syn.make_model(ns, ds, ws, hs, seed=3, device=dev, scale0=syn.SCALE0 * 6)
    z3 = torch.zeros(3, device=dev)
    with torch.no_grad():
        feat = pcs._semantic_feature.data.clone()
        pcs.rewrite_semantic_feature(torch.eye(ns, device=dev))
        Wx = render(cams, pcs, None, z3, feature_mode=True)["render"].permute(1, 2, 0).double()
        We = render(cams, pcs, None, z3, feature_mode=True, raster_flags=_lib.GAGS_FWD_EXACT)["render"].permute(1, 2, 0).double()
        out["weights_identical_through_both_kernels"] = bool(torch.equal(Wx, We))
        for scale_name, fs in (("unit", 1.0), ("tiny", 2.0 ** -60), ("huge", 2.0 ** 60)):
            ff = feat * fs
            # rows of very different magnitude inside one table
            ff[::3] *= 2.0 ** -20
            pcs.rewrite_semantic_feature(ff.contiguous())
            ref = We.reshape(-1, ns) @ ff.double()
            for name, fl in (("x16", 0), ("exact", _lib.GAGS_FWD_EXACT)):
                got = render(cams, pcs, None, z3, feature_mode=True, raster_flags=fl)["render"].permute(1, 2, 0).double().reshape(-1, ds)
                out[f"{name}_vs_float64_rel_l2_{scale_name}"] = float((got - ref).norm() / ref.norm())
                rown = (got - ref).norm(dim=1) / ref.norm(dim=1).clamp(min=1e-300)
                out[f"{name}_vs_float64_worst_pixel_{scale_name}"] = float(rown[ref.norm(dim=1) > 0].max())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
