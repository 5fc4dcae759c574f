"""CPU tests of the oracle itself: analytic known answers, agreement with
the independent dense float64 autograd restatement (values and every gradient), properties.
The reference has no tests and no runnable rasterizer (parity unpinned); these
are what pin the declared gsplat-1.4-style semantics."""
import math

import numpy as np
import pytest
import torch

from helpers import rel_l2, scene_arrays

IDENT = np.eye(4, dtype=np.float32)


def _K(fx, fy, w, h):
    return np.array([[fx, 0, w / 2], [0, fy, h / 2], [0, 0, 1]], np.float32)


def _one(oracle, mean, scale, opac, color, w=32, h=32, fx=40.0, bg=None, quat=(1, 0, 0, 0)):
    means = np.array([mean], np.float32)
    quats = np.array([quat], np.float32)
    scales = np.array([[scale] * 3], np.float32)
    return oracle.rasterization(means, quats, scales, np.array([opac], np.float32), np.array([color], np.float32),
                                IDENT, _K(fx, fx, w, h), bg, w, h)


def test_exp_polynomial_accuracy(oracle):
    s = np.concatenate([np.linspace(0, 20, 4001), np.array([0.0, 1e-8, 5.54, 87.0, 100.0])]).astype(np.float32)
    got = oracle.exp_neg(s).astype(np.float64)
    ref = np.exp(-(s.astype(np.float64)))
    err = np.abs(got - ref) / ref
    # polynomial: 1.4 ulp; plus the fp32 rounding of t = sigma*log2(e), which grows with sigma
    # (the same term CUDA's __expf = ex2.approx(x*log2e) carries).  Blended pairs have sigma <= 5.55.
    assert err[s <= 6.0].max() < 6e-7
    assert err[s <= 20.0].max() < 2e-6
    assert oracle.exp_neg([0.0])[0] == 1.0


def test_single_isotropic_gaussian_closed_form(oracle):
    """(a) on the optical axis, centred on a pixel centre: alpha = o, Sigma2 = (f s / z)^2 + 0.3,
    radius = ceil(3 sqrt(lambda)).  Odd image size puts the principal point on a pixel centre."""
    w = h = 33
    fx, s, z, o = 40.0, 0.05, 2.0, 0.6
    out, alpha, info = _one(oracle, [0.0, 0.0, z], s, o, [1.0, 2.0, 3.0], w, h, fx)
    var = (fx * s / z) ** 2 + 0.3
    assert info["radii"][0] == math.ceil(3 * math.sqrt(var + math.sqrt(0.01)))
    np.testing.assert_allclose(info["conics"][0], [1 / var, 0, 1 / var], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(info["means2d"][0], [16.5, 16.5], atol=1e-6)
    assert info["depths"][0] == z
    np.testing.assert_allclose(alpha[16, 16], o, rtol=1e-6)
    np.testing.assert_allclose(out[16, 16], np.array([1, 2, 3]) * o, rtol=1e-6)
    # one pixel to the right: sigma = 0.5 * conic
    np.testing.assert_allclose(alpha[16, 17], o * math.exp(-0.5 / var), rtol=1e-6)
    # two right, one down: sigma = 0.5 * (4 + 1) / var
    np.testing.assert_allclose(alpha[17, 18], o * math.exp(-2.5 / var), rtol=1e-6)


def test_two_coincident_gaussians_depth_order_and_transmittance(oracle):
    """(b) front one first; T product; (h) background blend out + T*bg."""
    w = h = 16
    fx = 20.0
    means = np.array([[0.0125, 0.0125, 3.0], [0.00833333, 0.00833333, 2.0]], np.float32)  # both project to ~(8.08, 8.08)
    quats = np.array([[1, 0, 0, 0]] * 2, np.float32)
    scales = np.full((2, 3), 0.3, np.float32)
    opac = np.array([0.5, 0.25], np.float32)
    cols = np.array([[1.0], [10.0]], np.float32)
    bg = np.array([100.0], np.float32)
    out, alpha, info = oracle.rasterization(means, quats, scales, opac, cols, IDENT, _K(fx, fx, w, h), bg, w, h)
    # sorted order in tile 0 is by depth: Gaussian 1 (z=2) then Gaussian 0 (z=3)
    np.testing.assert_array_equal(info["flatten_ids"], [1, 0])
    m2, con = info["means2d"], info["conics"]
    px, py = 8.5, 8.5
    al = []
    for g in (1, 0):
        dx, dy = m2[g, 0] - px, m2[g, 1] - py
        sig = 0.5 * (con[g, 0] * dx * dx + con[g, 2] * dy * dy) + con[g, 1] * dx * dy
        al.append(min(0.999, opac[g] * math.exp(-sig)))
    T1 = 1 - al[0]
    T2 = T1 * (1 - al[1])
    expect = 10.0 * al[0] + 1.0 * al[1] * T1 + T2 * 100.0
    np.testing.assert_allclose(out[8, 8, 0], expect, rtol=1e-5)
    np.testing.assert_allclose(alpha[8, 8], 1 - T2, rtol=1e-5)
    assert info["last_ids"][8, 8] == 1


def test_tile_aabb_membership_on_tile_corner(oracle):
    """(c) a Gaussian centred exactly on a tile corner touches the four tiles around it; (i) ragged H."""
    w, h, fx = 64, 40, 50.0  # 4 x 3 tiles, last tile row half covered (40 = 2.5 * 16)
    z = 2.0
    mean = [(32 - 32) * z / fx, (16 - 20) * z / fx, z]  # projects to (32, 16)
    out, alpha, info = _one(oracle, mean, 0.02, 0.9, [1.0], w, h, fx)
    r = info["radii"][0]
    assert r < 16
    assert info["tiles_per_gauss"][0] == 4
    tiles = sorted((info["isect_ids"] >> 32).tolist())
    assert tiles == [1, 2, 5, 6]
    off = info["isect_offsets"].reshape(-1)
    assert off.tolist() == [0, 0, 1, 2, 2, 2, 3, 4, 4, 4, 4, 4]
    assert info["isect_offsets"].shape == (3, 4)


def test_alpha_clamp_threshold_and_saturation(oracle):
    """(d) opacity 1 -> alpha clamped to 0.999; (e) 1/255 threshold; (f) stop at T <= 1e-4."""
    w = h = 16
    fx, z = 20.0, 2.0
    mean = [(8.5 - 8) * z / fx, (8.5 - 8) * z / fx, z]
    out, alpha, info = _one(oracle, mean, 0.5, 1.0, [1.0], w, h, fx)
    np.testing.assert_allclose(alpha[8, 8], 0.999, rtol=1e-6)
    # just below / above the 1/255 threshold at the centre pixel
    lo, _, _ = _one(oracle, mean, 0.5, 0.0039, [1.0], w, h, fx)
    hi, _, _ = _one(oracle, mean, 0.5, 0.0040, [1.0], w, h, fx)
    assert lo[8, 8, 0] == 0.0 and hi[8, 8, 0] > 0.0
    # saturation: alpha 0.999 each: T = 1e-3 after the first, the second would give 1e-6 <= 1e-4 -> stop, not blended
    n = 5
    means = np.array([[mean[0], mean[1], z + 0.1 * k] for k in range(n)], np.float32)
    out, alpha, info = oracle.rasterization(means, np.array([[1, 0, 0, 0]] * n, np.float32),
                                            np.full((n, 3), 0.5, np.float32), np.ones(n, np.float32),
                                            np.arange(1, n + 1, dtype=np.float32)[:, None], IDENT, _K(fx, fx, w, h),
                                            None, w, h)
    np.testing.assert_allclose(out[8, 8, 0], 1.0 * 0.999, rtol=1e-6)
    np.testing.assert_allclose(alpha[8, 8], 0.999, rtol=1e-6)
    assert info["last_ids"][8, 8] == info["isect_offsets"][0, 0]  # only the first of the tile's list


def test_near_plane_and_offscreen_culling(oracle):
    """(g) z < 0.01 culled; off-screen by more than the radius culled; culled rows are zero."""
    w = h = 32
    means = np.array([[0, 0, 0.005], [0, 0, -1.0], [50.0, 0, 2.0], [0, 0, 2.0]], np.float32)
    n = len(means)
    out, alpha, info = oracle.rasterization(means, np.array([[1, 0, 0, 0]] * n, np.float32),
                                            np.full((n, 3), 0.05, np.float32), np.full(n, 0.5, np.float32),
                                            np.ones((n, 1), np.float32), IDENT, _K(40, 40, w, h), None, w, h)
    assert info["radii"].tolist()[:3] == [0, 0, 0] and info["radii"][3] > 0
    assert np.all(info["means2d"][:3] == 0) and np.all(info["conics"][:3] == 0) and np.all(info["depths"][:3] == 0)
    assert info["tiles_per_gauss"][:3].tolist() == [0, 0, 0]


@pytest.mark.parametrize("d,view,seed", [(5, 2, 0), (16, None, 1), (1, 6, 2)])
def test_oracle_matches_dense_float64(oracle, d, view, seed):
    from oracle import dense_ref as dr
    w, h, n = 64, 48, 1200
    s = scene_arrays(n, d, w, h, seed=seed, view=view, scale_mult=8.0)
    bg = np.full(d, 0.3, np.float32)
    out, alphas, info = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"],
                                             s["viewmat"], s["K"], bg, w, h)
    rng = np.random.default_rng(seed)
    v_out = rng.standard_normal((h, w, d)).astype(np.float32)
    v_a = rng.standard_normal((h, w)).astype(np.float32)
    vc, vo, vm2, vcon = oracle.raster_bwd(info["means2d"], info["conics"], s["opacities"], s["colors"], bg, w, h,
                                          info["isect_offsets"], info["flatten_ids"], alphas, info["last_ids"], v_out, v_a)
    vM, vQ, vS = oracle.project_bwd(s["means"], s["quats"], s["scales"], s["viewmat"], s["K"], w, h, info["radii"], vm2,
                                    None, vcon)

    def tm(a, rg=False):
        return torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=rg)

    M, Q, S, O, C = tm(s["means"], True), tm(s["quats"], True), tm(s["scales"], True), tm(s["opacities"], True), tm(s["colors"], True)
    m2, z, con = dr.project(M, Q, S, tm(s["viewmat"]), tm(s["K"]), w, h)
    m2.retain_grad(); con.retain_grad()
    order = np.lexsort((np.arange(n), info["depths"]))
    o2, a2, last2, ninc = dr.composite(m2, con, O, C, tm(bg), w, h, info["radii"], order)
    vis = info["radii"] > 0
    assert np.abs(m2.detach().numpy()[vis] - info["means2d"][vis]).max() < 1e-4
    assert rel_l2(info["conics"][vis], con.detach().numpy()[vis]) < 1e-5
    assert ninc == info["n_blend"]
    assert rel_l2(out, o2.detach().numpy()) < 1e-5      # the BASELINE.json feature-render tolerance
    assert np.abs(alphas - a2.detach().numpy()).max() < 1e-5
    ((o2 * tm(v_out)).sum() + (a2 * tm(v_a)).sum()).backward()
    assert rel_l2(vc, C.grad.numpy()) < 1e-5
    assert rel_l2(vo, O.grad.numpy()) < 1e-5
    assert rel_l2(vm2, m2.grad.numpy()) < 1e-5
    assert rel_l2(vcon, con.grad.numpy()) < 1e-5
    assert rel_l2(vM, M.grad.numpy()) < 1e-5
    assert rel_l2(vQ, Q.grad.numpy()) < 1e-5
    assert rel_l2(vS, S.grad.numpy()) < 1e-5


def test_properties_linearity_concat_permutation(oracle):
    w, h, n = 80, 48, 900
    s = scene_arrays(n, 6, w, h, seed=9, view=4, scale_mult=6.0)
    args = (s["means"], s["quats"], s["scales"], s["opacities"])
    tail = (s["viewmat"], s["K"], None, w, h)
    c = s["colors"]
    full, a_full, i_full = oracle.rasterization(*args, c, *tail)
    left, a_l, _ = oracle.rasterization(*args, c[:, :2].copy(), *tail)
    right, a_r, _ = oracle.rasterization(*args, c[:, 2:].copy(), *tail)
    # render(D = a || b) == concat(render(a), render(b)), alpha identical across D -- bit-exact
    np.testing.assert_array_equal(full, np.concatenate([left, right], axis=-1))
    np.testing.assert_array_equal(a_full, a_l)
    np.testing.assert_array_equal(a_full, a_r)
    # linearity in colours
    two, _, _ = oracle.rasterization(*args, (2.0 * c).astype(np.float32), *tail)
    np.testing.assert_allclose(two, 2.0 * full, rtol=1e-6, atol=1e-7)
    # permutation of the input Gaussians does not change the image (distinct depths)
    perm = np.random.default_rng(0).permutation(n)
    p_out, p_a, _ = oracle.rasterization(*(a[perm] for a in args), c[perm], *tail)
    np.testing.assert_array_equal(p_a, a_full)
    np.testing.assert_array_equal(p_out, full)
    # culled Gaussians get exactly zero gradient
    v_out = np.ones((h, w, 6), np.float32)
    vc, vo, vm2, vcon = oracle.raster_bwd(i_full["means2d"], i_full["conics"], s["opacities"], c, None, w, h,
                                          i_full["isect_offsets"], i_full["flatten_ids"], a_full, i_full["last_ids"], v_out)
    culled = i_full["radii"] == 0
    assert culled.any()
    assert np.all(vc[culled] == 0) and np.all(vo[culled] == 0) and np.all(vm2[culled] == 0)


def test_sort_is_stable_on_depth_ties(oracle):
    """A7: equal (tile, depth) keys keep ascending Gaussian index."""
    n, w, h = 6, 16, 16
    means = np.tile(np.array([[0.0, 0.0, 2.0]], np.float32), (n, 1))
    out, alpha, info = oracle.rasterization(means, np.array([[1, 0, 0, 0]] * n, np.float32),
                                            np.full((n, 3), 0.05, np.float32), np.full(n, 0.1, np.float32),
                                            np.arange(n, dtype=np.float32)[:, None], IDENT, _K(20, 20, w, h), None, w, h)
    assert info["flatten_ids"].tolist() == list(range(n))


def test_rgb_ed_channel_semantics(oracle):
    """A10 / render.py:127-133: 4th channel = sum(w z) / max(alpha, 1e-10), background 0."""
    w = h = 16
    fx, z = 20.0, 2.5
    mean = [(8.5 - 8) * z / fx, (8.5 - 8) * z / fx, z]
    out, alpha, info = oracle.rasterization(np.array([mean], np.float32), np.array([[1, 0, 0, 0]], np.float32),
                                            np.full((1, 3), 0.3, np.float32), np.array([0.5], np.float32),
                                            np.array([[0.2, 0.4, 0.6]], np.float32), IDENT, _K(fx, fx, w, h),
                                            np.array([1.0, 1.0, 1.0], np.float32), w, h, render_mode="RGB+ED")
    assert out.shape == (h, w, 4)
    np.testing.assert_allclose(out[8, 8, 3], z, rtol=1e-6)
    np.testing.assert_allclose(out[8, 8, 0], 0.2 * 0.5 + 0.5 * 1.0, rtol=1e-6)
