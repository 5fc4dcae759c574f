"""Golden vectors produced by the reference's own Python (tests/golden/make_golden.py, run in
the build container where /root/reference exists).  They pin every part of the hot path that
lives in the reference tree: SH basis, camera matrices, quaternion convention, GaussianModel
activations, and -- through a recording stand-in for gsplat.rasterization -- the exact
arguments and return dict of gaussian_renderer.render(...)."""
import os
import types

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.npz"))


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_basis_matches_reference_eval_sh(oracle, deg):
    want = np.maximum(G[f"sh_out_deg{deg}"] + 0.5, 0.0)  # gsplat adds 0.5 and clamps
    coeffs = np.ascontiguousarray(np.transpose(G["sh_coeffs"], (0, 2, 1)))  # reference [N,3,16] -> ours [N,16,3]
    got = oracle.sh_fwd(deg, G["sh_dirs"], np.zeros(3, np.float32), coeffs)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)


def test_camera_matrices_match_reference():
    from gags_amd.scene import getWorld2View2, focal2fov, fov2focal
    for R, T, want in zip(G["w2v_R"], G["w2v_T"], G["w2v_out"]):
        np.testing.assert_array_equal(getWorld2View2(R, T), want)
    np.testing.assert_array_equal(getWorld2View2(G["w2v_R"][1], G["w2v_T"][1], np.array([0.5, -1.0, 2.0]), 1.5),
                                  G["w2v_translated"])
    for (f, p), fov, back in zip(G["fov_in"], G["fov_out"], G["focal_back"]):
        assert focal2fov(f, p) == fov
        assert fov2focal(fov, p) == back


def test_quaternion_convention_matches_reference(oracle):
    from oracle import dense_ref as dr
    R = dr.quat_to_rotmat(torch.tensor(G["rot_q"], dtype=torch.float64)).numpy()
    np.testing.assert_allclose(R, G["rot_R"], rtol=0, atol=2e-6)
    # and the fp32 oracle uses the same convention: an anisotropic Gaussian rotated by q must
    # project to the conic of  J R S S^T R^T J^T  built from the reference's R.
    q = G["rot_q"][:8]
    n = len(q)
    means = np.tile(np.array([[0.0, 0.0, 4.0]], np.float32), (n, 1))
    scales = np.tile(np.array([[0.3, 0.1, 0.05]], np.float32), (n, 1))
    w = h = 65
    fx = 60.0
    K = np.array([[fx, 0, w / 2], [0, fx, h / 2], [0, 0, 1]], np.float32)
    radii, m2, z, con = oracle.project_fwd(means, q, scales, np.eye(4, dtype=np.float32), K, w, h)
    for i in range(n):
        Rr = G["rot_R"][i].astype(np.float64)
        cov = Rr @ np.diag(scales[i].astype(np.float64) ** 2) @ Rr.T
        J = np.array([[fx / 4.0, 0, 0], [0, fx / 4.0, 0]])
        c2 = J @ cov @ J.T + 0.3 * np.eye(2)
        inv = np.linalg.inv(c2)
        np.testing.assert_allclose(con[i], [inv[0, 0], inv[0, 1], inv[1, 1]], rtol=2e-5, atol=1e-6)


def _model_from_golden(device="cpu"):
    from gags_amd.scene import GaussianModel
    t = lambda k: torch.tensor(G[f"act_raw_{k}"], device=device)
    pc = GaussianModel.from_tensors(t("xyz"), t("scaling"), t("rotation"), t("opacity"), t("features_dc"),
                                    t("features_rest"), t("semantic_feature"), sh_degree=3, active_sh_degree=2)
    return pc


def test_gaussian_model_activations_match_reference():
    pc = _model_from_golden()
    np.testing.assert_array_equal(pc.get_scaling.detach().numpy(), G["act_scaling"])
    np.testing.assert_array_equal(pc.get_rotation.detach().numpy(), G["act_rotation"])
    np.testing.assert_array_equal(pc.get_opacity.detach().numpy(), G["act_opacity"])
    np.testing.assert_array_equal(pc.get_features.detach().numpy(), G["act_features"])
    np.testing.assert_array_equal(pc.get_xyz.detach().numpy(), G["act_raw_xyz"])
    np.testing.assert_array_equal(pc.get_semantic_feature.detach().numpy(), G["act_raw_semantic_feature"])
    # feature-only optimisation (scene/gaussian_model.py:192-208)
    opt = pc.training_setup(semantic_feature_lr=1e-3)
    assert [g["name"] for g in opt.param_groups] == ["semantic_feature"] and opt.defaults["eps"] == 1e-15
    assert pc._semantic_feature.requires_grad and not pc._xyz.requires_grad and not pc._opacity.requires_grad


CASES = {
    "feature": dict(cam="cam", feature_mode=True),
    "feature_scaled": dict(cam="small", feature_mode=True, scaling_modifier=0.5),
    "override": dict(cam="small", feature_mode=False, override=True),
    "sh": dict(cam="small", feature_mode=False),
    "sh_ed": dict(cam="small", feature_mode=False, render_mode="RGB+ED"),
}


@pytest.mark.parametrize("name", list(CASES))
def test_render_hands_the_rasterizer_what_the_reference_does(monkeypatch, name):
    """gags_amd.gaussian_renderer.render(...) vs the reference's render(...): identical keyword
    arguments reach `rasterization`, identical dict comes back (host logic only, CPU tensors,
    the rasterizer replaced by a recorder exactly as in make_golden.py)."""
    import gags_amd.gaussian_renderer as gr
    case = dict(CASES[name])
    camk = case.pop("cam")
    fovx, fovy, wd, ht = G[f"call_{camk}"]
    cam = types.SimpleNamespace(FoVx=float(fovx), FoVy=float(fovy), image_width=int(wd), image_height=int(ht),
                                world_view_transform=torch.tensor(G[f"call_{camk}_wvt"]))
    pc = _model_from_golden()
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

    monkeypatch.setattr(gr, "rasterization", recorder)
    kw = dict(feature_mode=case.get("feature_mode", True), scaling_modifier=case.get("scaling_modifier", 1.0),
              render_mode=case.get("render_mode", "RGB"))
    if case.get("override"):
        kw["override_color"] = torch.tensor(G["call_override"])
    res = gr.render(cam, pc, None, torch.tensor(G["call_bg"]), **kw)
    (c,) = calls
    for k in ("means", "quats", "scales", "opacities", "colors", "viewmats", "Ks", "backgrounds"):
        np.testing.assert_array_equal(c[k].detach().numpy(), G[f"call_{name}_{k}"], err_msg=k)
    w_, h_, packed, shd = G[f"call_{name}_scalars"]
    assert (c["width"], c["height"], int(c["packed"])) == (w_, h_, packed)
    assert (-1 if c["sh_degree"] is None else c["sh_degree"]) == shd
    assert c["render_mode"] == str(G[f"call_{name}_render_mode"])
    assert set(G[f"call_{name}_keys"].tolist()) <= set(res.keys())
    assert tuple(res["render"].shape) == tuple(G[f"call_{name}_render_shape"])
    np.testing.assert_array_equal(res["render"].reshape(res["render"].shape[0], -1)[:, :5].numpy(),
                                  G[f"call_{name}_render_first"])
    np.testing.assert_array_equal(res["visibility_filter"].numpy(), G[f"call_{name}_visibility"])
    np.testing.assert_array_equal(res["radii"].numpy(), G[f"call_{name}_radii"])
    assert tuple(res["viewspace_points"].shape) == tuple(G[f"call_{name}_vsp_shape"])


def test_oracle_adam_matches_the_reference_optimizer(oracle):
    """R9: the C restatement of Adam reproduces three steps of torch.optim.Adam configured exactly as
    scene/gaussian_model.py:192-208 does (fixture: tests/golden/make_golden_adam.py)."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "adam_vectors.npz"))
    p = z["p0"].copy()
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for t in range(1, 4):
        oracle.adam_step(p, np.ascontiguousarray(z[f"g{t}"]), m, v, float(z["lr"]), eps=1e-15, step=t)
        # fp32 tolerance: one rounding of the operands' magnitude (torch's lerp / addcmul may fuse or reorder
        # one operation); |g| ~ 1, so 2e-8 absolute where the result cancels
        np.testing.assert_allclose(m, z[f"m{t}"], rtol=2e-6, atol=2e-8)
        np.testing.assert_allclose(v, z[f"v{t}"], rtol=2e-6, atol=2e-8)
        np.testing.assert_allclose(p, z[f"p{t}"], rtol=2.5e-7, atol=1e-9)  # one ulp of p


def test_activate_oracle_against_the_reference_functions():
    """oracle/activate_ref.py against tests/golden/activate_vectors.npz = outputs of the reference's own activate_stream /
    lerf_localization / smooth (make_golden_activate.py): heat map, final mask, IoU, localisation."""
    from oracle import activate_ref as A
    Zp = np.load(os.path.join(os.path.dirname(__file__), "golden", "activate_vectors.npz"))
    valid, gt, th = Zp["act_valid_map"], Zp["act_gt_mask"], float(Zp["act_thresh"])
    acc = 0
    for k in range(valid.shape[0]):
        r = A.activate(valid[k], thresh=th)
        np.testing.assert_allclose(r["heatmap"], Zp["act_heatmap"][k], rtol=0, atol=1e-7)
        np.testing.assert_array_equal(r["mask"], Zp["act_mask"][k])
        np.testing.assert_array_equal(A.smooth_fast(r["mask_pred"]), r["mask"])
        iou = np.logical_and(gt[k], r["mask"]).sum() / np.logical_or(gt[k], r["mask"]).sum()
        assert abs(iou - Zp["act_iou"][k]) < 1e-12
        score, coords, hit = A.localize(valid[k], Zp[f"act_boxes{k}"])
        np.testing.assert_array_equal(coords, Zp[f"act_loc_coords{k}"])
        acc += int(hit)
    assert acc == int(Zp["act_loc_acc"])
    np.testing.assert_array_equal(A.smooth(Zp["act_smooth_in"]), Zp["act_smooth_out"])
    np.testing.assert_array_equal(A.smooth_fast(Zp["act_smooth_in"]), Zp["act_smooth_out"])


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_view_direction_gradient_oracle_matches_autograd_of_the_reference_eval_sh(deg):
    """oracle/dense_ref.sh_colors (float64 autograd) against tests/golden/shgrad_vectors.npz = autograd through the
    reference's own eval_sh: colours, d / d means (through the normalised view direction) and d / d coefficients."""
    from oracle import dense_ref as DR
    Zs = np.load(os.path.join(os.path.dirname(__file__), "golden", "shgrad_vectors.npz"))
    co = torch.from_numpy(Zs["shg_coeffs"]).double().permute(0, 2, 1).contiguous().requires_grad_(True)  # [N,3,16] -> [N,16,3]
    m = torch.from_numpy(Zs["shg_means"]).double().requires_grad_(True)
    col = DR.sh_colors(deg, co, m, torch.from_numpy(Zs["shg_campos"]).double())
    (col * torch.from_numpy(Zs["shg_v_out"]).double()).sum().backward()
    np.testing.assert_allclose(col.detach().numpy(), Zs[f"shg_col_deg{deg}"], rtol=0, atol=1e-12)
    vm = np.zeros_like(Zs[f"shg_vmeans_deg{deg}"]) if m.grad is None else m.grad.numpy()
    np.testing.assert_allclose(vm, Zs[f"shg_vmeans_deg{deg}"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(co.grad.permute(0, 2, 1).numpy(), Zs[f"shg_vcoeffs_deg{deg}"], rtol=0, atol=1e-12)
