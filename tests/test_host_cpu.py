"""Host-side logic that needs no GPU: synthetic workload generator, signature of the drop-in
boundary, configuration table, bench-line helpers."""
import inspect
import math

import numpy as np
import torch


def test_render_signature_is_the_reference_one():
    from gags_amd.gaussian_renderer import render
    p = list(inspect.signature(render).parameters.items())
    names = [k for k, _ in p]
    # gaussian_renderer/__init__.py:19
    assert names[:8] == ["viewpoint_camera", "pc", "pipe", "bg_color", "feature_mode", "scaling_modifier",
                         "override_color", "render_mode"]
    d = dict(p)
    assert d["feature_mode"].default is True and d["scaling_modifier"].default == 1.0
    assert d["override_color"].default is None and d["render_mode"].default == "RGB"


def test_rasterization_signature_accepts_the_reference_call():
    from gags_amd.rasterization import rasterization
    ps = inspect.signature(rasterization).parameters
    for k in ("means", "quats", "scales", "opacities", "colors", "viewmats", "Ks", "backgrounds", "width", "height",
              "packed", "sh_degree", "render_mode"):   # gaussian_renderer/__init__.py:56-70
        assert k in ps
    assert ps["eps2d"].default == 0.3 and ps["near_plane"].default == 0.01 and ps["far_plane"].default == 1e10
    assert ps["tile_size"].default == 16 and ps["radius_clip"].default == 0.0


def test_synthetic_workloads_are_deterministic_and_named():
    from gags_amd import synthetic as syn
    assert syn.CONFIGS["C3"] == dict(n=1_500_000, width=1920, height=1080, d=512)
    assert syn.CONFIGS["C2"] == dict(n=500_000, width=1280, height=720, d=128)
    a = syn.make_gaussians(500, 8, 320, 200, seed=3)
    b = syn.make_gaussians(500, 8, 320, 200, seed=3)
    for k in a:
        assert torch.equal(a[k], b[k])
    assert a["semantic_feature"].shape == (500, 8) and a["rotation"].shape == (500, 4)
    assert a["features_rest"].shape == (500, 15, 3) and a["opacity_logit"].shape == (500, 1)
    assert float(a["xyz"][:, 2].min()) >= syn.Z_NEAR and float(a["xyz"][:, 2].max()) <= syn.Z_FAR
    cam = syn.make_camera(1920, 1080, view=None, device="cpu")
    vm, K = syn.camera_matrices(cam)
    np.testing.assert_allclose(K, [[1728, 0, 960], [0, 1728, 540], [0, 0, 1]], rtol=1e-6)
    np.testing.assert_array_equal(vm.numpy(), np.eye(4, dtype=np.float32))
    yaw = [syn.make_camera(64, 64, view=v, device="cpu") for v in range(8)]
    ang = [math.degrees(math.atan2(c.world_view_transform[2, 0].item(), c.world_view_transform[0, 0].item())) for c in yaw]
    np.testing.assert_allclose(np.abs(np.diff(ang)), 5.0, atol=1e-4)
    G = syn.make_cotangent(4, 6, 5, seed=1)
    assert G.shape == (4, 6, 5) and G.permute(1, 2, 0).is_contiguous()


def test_workload_meets_the_declared_intent(oracle):
    """SURVEY 8d intent: few-pixel splats, a handful of tile intersections per visible Gaussian."""
    from helpers import scene_arrays
    w, h, n = 1920, 1080, 20000
    s = scene_arrays(n, 1, w, h, seed=0)
    radii, m2, z, con = oracle.project_fwd(s["means"], s["quats"], s["scales"], s["viewmat"], s["K"], w, h)
    b = oracle.tile_bin(m2, radii, z, w, h)
    vis = radii > 0
    assert 0.85 < vis.mean() < 0.99             # ~5 % of the means are off-screen
    assert 5 <= np.median(radii[vis]) <= 9      # median 3-sigma radius ~7 px
    assert 3.0 <= b["n_isects"] / vis.sum() <= 6.5


def test_frozen_activations_are_cached_and_invalidated():
    """GaussianModel getters of frozen geometry (scene/gaussian_model.py:116-139 recomputes exp / normalize / sigmoid on
    every render): cached per parameter version; an in-place update, a new tensor, or requires_grad bring back a fresh
    evaluation.  The cache is OPT-IN (`.data` writes bypass the version counter): by default every getter
    call evaluates afresh, exactly like the reference."""
    import torch
    from gags_amd.scene import GaussianModel
    m = GaussianModel(3)
    n = 50
    m._scaling = torch.nn.Parameter(torch.zeros(n, 3), requires_grad=False)
    first = m.get_scaling
    m._scaling.data.add_(1.0)                     # invisible to _version
    assert m.get_scaling is not first and torch.equal(m.get_scaling, torch.exp(m._scaling))
    m.cache_activations(True)
    g = torch.Generator().manual_seed(0)
    m._scaling = torch.nn.Parameter(torch.randn(n, 3, generator=g), requires_grad=False)
    m._rotation = torch.nn.Parameter(torch.randn(n, 4, generator=g), requires_grad=False)
    m._opacity = torch.nn.Parameter(torch.randn(n, 1, generator=g), requires_grad=False)
    a = m.get_scaling
    assert m.get_scaling is a and torch.equal(a, torch.exp(m._scaling))
    with torch.no_grad():
        m._scaling.add_(1.0)                      # in-place: version bump
    b = m.get_scaling
    assert b is not a and torch.equal(b, torch.exp(m._scaling))
    m._scaling = torch.nn.Parameter(torch.zeros(n, 3), requires_grad=False)   # replaced
    assert torch.equal(m.get_scaling, torch.ones(n, 3))
    stale = m.get_scaling
    m._scaling.data.add_(1.0)                     # the documented blind spot of the opt-in cache ...
    assert m.get_scaling is stale
    m.invalidate_activations()                    # ... and its remedy
    assert torch.equal(m.get_scaling, torch.exp(m._scaling))
    m._opacity.requires_grad_(True)               # trainable again: always evaluated, with a graph
    o = m.get_opacity
    assert o.requires_grad and m.get_opacity is not o
    assert torch.allclose(m.get_rotation.norm(dim=1), torch.ones(n))
