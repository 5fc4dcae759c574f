#!/usr/bin/env python
"""Generate tests/golden/reference_vectors.npz by IMPORTING the reference's own Python
(/root/reference) in this container.  Only inputs and expected outputs are stored; no
reference source travels.  Run here (the reference tree does not exist on the GPU box):

    python tests/golden/make_golden.py

What is pinned (the parts of the hot path that live in the reference tree):
  sh_*     utils/sh_utils.py:57-112        eval_sh, degrees 0..3 (basis + sign convention)
  w2v_*    utils/graphics_utils.py:38-49   getWorld2View2, and :73-77 fov<->focal
  rot_*    utils/general_utils.py:78-98    build_rotation (wxyz quaternion convention)
  act_*    scene/gaussian_model.py:116-139 activation getters of GaussianModel
  call_*   gaussian_renderer/__init__.py:19-85  the EXACT arguments render(...) hands to
           gsplat.rasterization in each colour branch / render mode, captured with a
           recording stand-in for the (absent) gsplat module, and the dict it returns.
The reference hard-codes device="cuda"; this script strips the device argument so the same
code runs on CPU tensors.  gsplat / simple_knn / plyfile / cv2 are not installed: empty
stand-in modules satisfy the imports (none of their functionality is executed, except the
recorder standing in for gsplat.rasterization).
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.npz")


def _stub_modules(recorder):
    for name in ("plyfile", "cv2", "simple_knn", "simple_knn._C", "gsplat"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = object
    sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = lambda *a, **k: None
    sys.modules["simple_knn"]._C = sys.modules["simple_knn._C"]
    sys.modules["gsplat"].rasterization = recorder


class _NoCuda:
    """Drop device= from tensor factories and make .cuda() a no-op while the reference runs."""

    def __enter__(self):
        self.saved = {n: getattr(torch, n) for n in ("tensor", "zeros", "ones")}
        for n, f in self.saved.items():
            setattr(torch, n, (lambda f: lambda *a, **k: f(*a, **{kk: v for kk, v in k.items() if kk != "device"}))(f))
        self.cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        return self

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(torch, n, f)
        torch.Tensor.cuda = self.cuda


def main():
    calls = []

    def recorder(**kw):
        calls.append(kw)
        n, (w, h) = kw["means"].shape[0], (kw["width"], kw["height"])
        d = kw["colors"].shape[-1] if kw["sh_degree"] is None else 3
        if kw["render_mode"] == "RGB+ED":
            d += 1
        colors = torch.arange(h * w * d, dtype=torch.float32).reshape(1, h, w, d)
        info = {"radii": torch.arange(n, dtype=torch.int32)[None] % 3, "means2d": torch.zeros(1, n, 2)}
        return colors, torch.zeros(1, h, w, 1), info

    _stub_modules(recorder)
    sys.path.insert(0, REF)
    from utils.sh_utils import eval_sh
    from utils.graphics_utils import getWorld2View2, focal2fov, fov2focal
    from utils.general_utils import build_rotation
    from scene.gaussian_model import GaussianModel
    import gaussian_renderer as ref_renderer

    out = {}
    g = torch.Generator().manual_seed(1234)

    # --- SH ---
    sh = torch.randn(64, 3, 16, generator=g)
    dirs = torch.nn.functional.normalize(torch.randn(64, 3, generator=g))
    out["sh_coeffs"], out["sh_dirs"] = sh.numpy(), dirs.numpy()
    for deg in range(4):
        out[f"sh_out_deg{deg}"] = eval_sh(deg, sh, dirs).numpy()

    # --- camera matrices ---
    Rs, Ts, W2V = [], [], []
    for k in range(4):
        a = 0.3 * k
        R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]]) @ \
            np.array([[1, 0, 0], [0, math.cos(0.1 * k), -math.sin(0.1 * k)], [0, math.sin(0.1 * k), math.cos(0.1 * k)]])
        T = np.array([0.1 * k, -0.2 * k, 0.5 + k])
        Rs.append(R); Ts.append(T); W2V.append(getWorld2View2(R, T))
    out["w2v_R"], out["w2v_T"], out["w2v_out"] = np.array(Rs), np.array(Ts), np.array(W2V)
    out["w2v_translated"] = getWorld2View2(Rs[1], Ts[1], np.array([0.5, -1.0, 2.0]), 1.5)
    out["fov_in"] = np.array([[1728.0, 1920.0], [1152.0, 1280.0], [500.0, 640.0]])
    out["fov_out"] = np.array([focal2fov(f, p) for f, p in out["fov_in"]])
    out["focal_back"] = np.array([fov2focal(fv, p) for fv, (f, p) in zip(out["fov_out"], out["fov_in"])])

    # --- quaternion convention ---
    q = torch.randn(16, 4, generator=g)
    with _NoCuda():
        Rq = build_rotation(q)
    out["rot_q"], out["rot_R"] = q.numpy(), Rq.numpy()

    # --- GaussianModel getters + render() argument capture ---
    n, d = 37, 16
    with _NoCuda():
        pc = GaussianModel(3)
        pc._xyz = torch.randn(n, 3, generator=g)
        pc._features_dc = torch.randn(n, 1, 3, generator=g)
        pc._features_rest = torch.randn(n, 15, 3, generator=g)
        pc._scaling = torch.randn(n, 3, generator=g) - 3.0
        pc._rotation = torch.randn(n, 4, generator=g)
        pc._opacity = torch.randn(n, 1, generator=g)
        pc._semantic_feature = torch.randn(n, d, generator=g)
        pc.active_sh_degree = 2
        for name in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity", "semantic_feature"):
            out[f"act_raw_{name}"] = getattr(pc, "_" + name).numpy()
        out["act_scaling"] = pc.get_scaling.numpy()
        out["act_rotation"] = pc.get_rotation.numpy()
        out["act_opacity"] = pc.get_opacity.numpy()
        out["act_features"] = pc.get_features.numpy()

        cam = types.SimpleNamespace(FoVx=focal2fov(1728.0, 1920), FoVy=focal2fov(1700.0, 1080), image_width=1920,
                                    image_height=1080,
                                    world_view_transform=torch.tensor(getWorld2View2(Rs[2], Ts[2])).transpose(0, 1))
        small = types.SimpleNamespace(FoVx=0.9, FoVy=0.7, image_width=40, image_height=24,
                                      world_view_transform=torch.tensor(getWorld2View2(Rs[1], Ts[1])).transpose(0, 1))
        out["call_cam"] = np.array([cam.FoVx, cam.FoVy, cam.image_width, cam.image_height])
        out["call_cam_wvt"] = cam.world_view_transform.numpy()
        out["call_small"] = np.array([small.FoVx, small.FoVy, small.image_width, small.image_height])
        out["call_small_wvt"] = small.world_view_transform.numpy()
        bg = torch.tensor([1.0, 0.5, 0.25])
        override = torch.rand(n, 3, generator=g)
        out["call_bg"], out["call_override"] = bg.numpy(), override.numpy()
        cases = {
            "feature": dict(viewpoint_camera=cam, feature_mode=True),
            "feature_scaled": dict(viewpoint_camera=small, feature_mode=True, scaling_modifier=0.5),
            "override": dict(viewpoint_camera=small, feature_mode=False, override_color=override),
            "sh": dict(viewpoint_camera=small, feature_mode=False),
            "sh_ed": dict(viewpoint_camera=small, feature_mode=False, render_mode="RGB+ED"),
        }
        for name, kw in cases.items():
            calls.clear()
            res = ref_renderer.render(pc=pc, pipe=None, bg_color=bg, **kw)
            (c,) = calls
            for k in ("means", "quats", "scales", "opacities", "colors", "viewmats", "Ks", "backgrounds"):
                out[f"call_{name}_{k}"] = c[k].detach().numpy()
            out[f"call_{name}_scalars"] = np.array([c["width"], c["height"], int(c["packed"]),
                                                   -1 if c["sh_degree"] is None else c["sh_degree"]])
            out[f"call_{name}_render_mode"] = np.array(c["render_mode"])
            out[f"call_{name}_keys"] = np.array(sorted(res.keys()))
            out[f"call_{name}_render_shape"] = np.array(res["render"].shape)
            out[f"call_{name}_render_first"] = res["render"].reshape(res["render"].shape[0], -1)[:, :5].numpy()
            out[f"call_{name}_visibility"] = res["visibility_filter"].numpy()
            out[f"call_{name}_radii"] = res["radii"].numpy()
            out[f"call_{name}_vsp_shape"] = np.array(res["viewspace_points"].shape)

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
