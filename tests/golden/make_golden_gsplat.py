#!/usr/bin/env python
"""Pin the oracle to gsplat itself -- the day gsplat is reachable.

The reference's rasterizer is the pip package `gsplat`, imported at /root/reference/gaussian_renderer/__init__.py:17 and
called at :56-70; it is absent from the reference tree, unpinned (environment.yml:26) and not installable in the build
container (no network), so oracle/gags_oracle.c restates the published gsplat-1.4 algorithm and says "parity unpinned".
This script closes that gap wherever `import gsplat` works (a machine with gsplat and a GPU it supports):

    python tests/golden/make_golden_gsplat.py            # writes tests/golden/gsplat_vectors.npz
    python -m pytest tests/test_gsplat_fixture_cpu.py    # holds oracle/gags_oracle.c to it (CPU only)

It calls `gsplat.rasterization` with exactly the keyword arguments the reference's render(...) passes
(gaussian_renderer/__init__.py:56-70: means, quats, scales, opacities, colors, viewmats, Ks, backgrounds, width, height,
packed=False, sh_degree, render_mode) on small seeded scenes -- the generator of tests/helpers.py::scene_arrays, <= 2 000
Gaussians, 64 x 48 .. 97 x 61 pixels, D in {3, 4, 16, 33}, seeds 0-3, one with SH colours, one RGB+ED -- and stores the
INPUTS together with what gsplat returned: render_colors, render_alphas, info[radii, tiles_per_gauss, isect_ids,
flatten_ids, isect_offsets, means2d, depths, conics], the gradient of <render, G> w.r.t. colors (and means / quats / scales /
opacities for the first scene), and -- when gsplat's internal forward is reachable -- last_ids.  Data only; no gsplat
source travels.  The file records gsplat.__version__: the oracle targets the 1.x semantics of integer radii.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(HERE, "gsplat_vectors.npz")

# (name, n, width, height, D, seed, view, scale_mult, background value or None, sh_degree, render_mode, all_grads)
SCENES = [
    ("d3", 2000, 64, 48, 3, 0, None, 8.0, 0.0, None, "RGB", True),
    ("d4", 2000, 64, 48, 4, 1, 2, 8.0, 1.0, None, "RGB", False),
    ("d16", 2000, 64, 48, 16, 2, 5, 8.0, 0.0, None, "RGB", False),     # the width the reference rasterizes (train.py:68)
    ("d33", 1500, 97, 61, 33, 3, 3, 6.0, 0.3, None, "RGB", False),     # ragged image, two 32-channel chunks in gsplat
    ("sh", 1500, 64, 48, 0, 0, 1, 8.0, 0.5, 3, "RGB", False),          # feature_mode=False: SH colours, degree 3
    ("ed", 1500, 64, 48, 3, 1, None, 8.0, 0.0, None, "RGB+ED", False),  # render.py:118,127-133
]


def main():
    try:
        import gsplat
        from gsplat import rasterization
    except Exception as e:  # noqa: BLE001
        print(f"make_golden_gsplat: `import gsplat` failed ({e!r}); nothing written.  Run this where gsplat is installed.")
        return 2
    if not torch.cuda.is_available():
        print("make_golden_gsplat: gsplat needs a GPU; nothing written.")
        return 2
    from helpers import scene_arrays
    dev = torch.device("cuda", 0)
    out = {"gsplat_version": np.array(getattr(gsplat, "__version__", "unknown")), "scenes": np.array([s[0] for s in SCENES])}
    for name, n, w, h, d, seed, view, mult, bgv, sh_degree, mode, all_grads in SCENES:
        s = scene_arrays(n, max(d, 1), w, h, seed=seed, view=view, scale_mult=mult, sh=True)
        colors = s["sh"] if sh_degree is not None else s["colors"][:, :d]
        d_out = 3 if sh_degree is not None else d
        bg = None if bgv is None else np.full(d_out, bgv, np.float32)
        t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in
             dict(means=s["means"], quats=s["quats"], scales=s["scales"], opacities=s["opacities"], colors=colors).items()}
        leaves = ["colors"] + (["means", "quats", "scales", "opacities"] if all_grads else [])
        for k in leaves:
            t[k].requires_grad_(True)
        viewmat = torch.from_numpy(s["viewmat"]).to(dev)
        K = torch.from_numpy(s["K"]).to(dev)
        # the reference's call, keyword for keyword (gaussian_renderer/__init__.py:56-70)
        render_colors, render_alphas, info = rasterization(
            means=t["means"], quats=t["quats"], scales=t["scales"], opacities=t["opacities"], colors=t["colors"],
            viewmats=viewmat[None], Ks=K[None], backgrounds=None if bg is None else torch.from_numpy(bg).to(dev)[None],
            width=w, height=h, packed=False, sh_degree=sh_degree, render_mode=mode)
        dd = render_colors.shape[-1]
        G = torch.from_numpy(np.random.default_rng(seed + 100).standard_normal((h, w, dd)).astype(np.float32)).to(dev)
        (render_colors[0] * G).sum().backward()
        rec = dict(means=s["means"], quats=s["quats"], scales=s["scales"], opacities=s["opacities"], colors=colors,
                   viewmat=s["viewmat"], K=s["K"], width=np.int32(w), height=np.int32(h), cotangent=G.cpu().numpy(),
                   sh_degree=np.int32(-1 if sh_degree is None else sh_degree), render_mode=np.array(mode),
                   render_colors=render_colors[0].detach().cpu().numpy(), render_alphas=render_alphas[0, ..., 0].detach().cpu().numpy())
        if bg is not None:
            rec["backgrounds"] = bg
        for key in ("radii", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets", "means2d", "depths", "conics"):
            if key in info and torch.is_tensor(info[key]):
                rec["info_" + key] = info[key].detach().cpu().numpy()
        for k in leaves:
            rec["v_" + k] = t[k].grad.detach().cpu().numpy()
        try:  # last_ids: not part of gsplat's public return; its forward kernel hands it back
            from gsplat.cuda._wrapper import _make_lazy_cuda_func
            if sh_degree is None and mode == "RGB":
                res = _make_lazy_cuda_func("rasterize_to_pixels_fwd")(
                    info["means2d"].contiguous(), info["conics"].contiguous(), t["colors"].detach()[None].contiguous(),
                    t["opacities"].detach()[None].contiguous(), None if bg is None else torch.from_numpy(bg).to(dev)[None],
                    None, w, h, info["tile_size"], info["isect_offsets"].contiguous(), info["flatten_ids"].contiguous())
                rec["last_ids"] = res[2][0].cpu().numpy()
        except Exception as e:  # noqa: BLE001
            print(f"  ({name}: last_ids not captured: {e!r})")
        for k, v in rec.items():
            out[f"{name}/{k}"] = v
        print(f"{name}: N={n} {w}x{h} D={dd} isects={int(info['flatten_ids'].numel())}")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, f"({os.path.getsize(OUT) / 1e6:.1f} MB), gsplat", out["gsplat_version"])
    return 0


if __name__ == "__main__":
    sys.exit(main())
