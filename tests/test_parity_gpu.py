"""HIP path vs CPU oracle on identical seeded inputs, through the C ABI (via the Python
mirror of gsplat.rasterization).  Index tensors (radii, tiles_per_gauss, isect_ids,
flatten_ids, isect_offsets, last_ids) must be BIT-EXACT; forward renders are bit-exact too on
the fp32 paths (same op order, explicit exp polynomial, fmaf accumulation in sorted order);
gradients are summed in a different (fixed, or for the atomic kernels arbitrary) order than the
oracle's double-accumulated sums, so they are compared with rel-L2 <= 2e-5."""
import numpy as np
import pytest
import torch

from helpers import FWD_SPLIT_TOL, check_forward, rel_l2, scene_arrays, to_dev

pytestmark = pytest.mark.gpu

GRAD_TOL = 2e-5
# vs the gsplat-order oracle (T rebuilt back to front from 1 - alpha): measured <= 5e-7 on unsaturated scenes and
# 1.4e-4 on the large-splat scene, whose saturated pixels lose digits in 1 - alpha (an error of THAT order of summation,
# float64 check in tests/test_fullsize_gpu.py); bounds = measured x ~3
GSPLAT_ORDER_TOL = 2e-6
GSPLAT_ORDER_TOL_SATURATED = 4e-4


def _run_gpu(s, width, height, colors, bg, render_mode="RGB", sh_degree=None, need_geom=False, flags=0,
             v_out=None, v_alpha=None, context=None):
    from gags_amd.rasterization import rasterization
    means, quats, scales = to_dev(s["means"]), to_dev(s["quats"]), to_dev(s["scales"])
    opac, cols = to_dev(s["opacities"]), to_dev(colors)
    leaves = [cols]
    if need_geom:
        leaves += [means, quats, scales, opac]
    for t in leaves:
        t.requires_grad_(True)
    out, alphas, info = rasterization(means, quats, scales, opac, cols, to_dev(s["viewmat"])[None],
                                      to_dev(s["K"])[None], width, height,
                                      backgrounds=None if bg is None else to_dev(bg)[None],
                                      sh_degree=sh_degree, render_mode=render_mode, raster_flags=flags, context=context)
    grads = None
    if v_out is not None:
        loss = (out[0] * to_dev(v_out)).sum()
        if v_alpha is not None:
            loss = loss + (alphas[0, ..., 0] * to_dev(v_alpha)).sum()
        if need_geom:
            info["means2d"].retain_grad()
        loss.backward()
        grads = dict(colors=cols.grad.cpu().numpy())
        if need_geom:
            grads.update(means=means.grad.cpu().numpy(), quats=quats.grad.cpu().numpy(),
                         scales=scales.grad.cpu().numpy(), opacities=opac.grad.cpu().numpy(),
                         means2d=info["means2d"].grad[0].cpu().numpy())
    torch.cuda.synchronize()
    return out[0].detach().cpu().numpy(), alphas[0, ..., 0].detach().cpu().numpy(), info, grads


def _exact_forward(s, width, height, colors, bg, **kw):
    """The same render through the kernel that is the oracle's fmaf chain (GAGS_FWD_EXACT), for widths whose default is the
    16-bit matrix-core forward; None below 128 channels (the default IS exact there)."""
    from gags_amd import _lib
    if colors.shape[-1] < 128:
        return None
    return _run_gpu(s, width, height, colors, bg, flags=_lib.GAGS_FWD_EXACT, **kw)[0]


def _check_indices(info, oinfo):
    np.testing.assert_array_equal(info["radii"][0].cpu().numpy(), oinfo["radii"])
    np.testing.assert_array_equal(info["tiles_per_gauss"][0].cpu().numpy(), oinfo["tiles_per_gauss"])
    assert info["n_isects"] == oinfo["n_isects"]
    np.testing.assert_array_equal(info["isect_ids"].cpu().numpy(), oinfo["isect_ids"])
    np.testing.assert_array_equal(info["flatten_ids"].cpu().numpy(), oinfo["flatten_ids"])
    np.testing.assert_array_equal(info["isect_offsets"][0].cpu().numpy(), oinfo["isect_offsets"])
    np.testing.assert_array_equal(info["means2d"][0].detach().cpu().numpy(), oinfo["means2d"])
    np.testing.assert_array_equal(info["conics"][0].detach().cpu().numpy(), oinfo["conics"])
    np.testing.assert_array_equal(info["depths"][0].detach().cpu().numpy(), oinfo["depths"])
    np.testing.assert_array_equal(info["last_ids"].cpu().numpy(), oinfo["last_ids"])


@pytest.mark.parametrize("n,w,h,d,seed,view,mult,bgv", [
    (3000, 200, 152, 16, 0, None, 4.0, 0.0),   # reference default width (train.py:68)
    (3000, 200, 152, 3, 1, 2, 4.0, 1.0),       # RGB-like, white background, yawed camera
    (2000, 97, 61, 33, 2, 5, 6.0, 0.3),        # ragged image (not a multiple of 16), D=33 -> 2 chunks
    (5000, 256, 256, 4, 3, None, 2.0, None),   # no background
    (1500, 64, 48, 1, 4, 0, 8.0, 0.0),
    (4000, 160, 120, 64, 5, None, 4.0, 0.5),   # MFMA path, NB=2
    (2000, 97, 61, 32, 6, 3, 6.0, 0.3),        # MFMA path, NB=1, ragged image
    (5000, 200, 152, 128, 7, None, 4.0, 0.0),  # MFMA path, NB=4 (C2 width)
    (3000, 176, 130, 256, 8, 6, 5.0, 1.0),     # MFMA path, two 128-channel slices
    (2500, 130, 100, 512, 9, 1, 5.0, None),    # MFMA path, four 128-channel slices (C3 width)
    (6000, 96, 64, 128, 10, None, 16.0, 0.2),  # large splats: long per-tile lists, early termination
    (3000, 150, 110, 96, 11, 4, 5.0, 0.1),     # staged backward with 32-channel slices (D % 64 != 0)
    (3000, 150, 110, 160, 12, 7, 5.0, None),   # 32-channel slices, five of them
    (3000, 150, 110, 192, 13, None, 5.0, 0.0), # 64-channel slices
    (3000, 150, 110, 48, 14, 2, 5.0, 0.4),     # ragged last slice (D % 32 = 16) on the matrix-core path
    (2000, 97, 61, 20, 15, 5, 6.0, 0.3),       # one ragged slice, ragged image
])
def test_forward_and_colour_grad(oracle, n, w, h, d, seed, view, mult, bgv):
    s = scene_arrays(n, d, w, h, seed=seed, view=view, scale_mult=mult)
    bg = None if bgv is None else np.full(d, bgv, np.float32)
    rng = np.random.default_rng(seed + 100)
    v_out = rng.standard_normal((h, w, d)).astype(np.float32)
    o_out, o_alpha, oinfo = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"],
                                                 s["viewmat"], s["K"], bg, w, h)
    out, alpha, info, grads = _run_gpu(s, w, h, s["colors"], bg, v_out=v_out)
    _check_indices(info, oinfo)
    np.testing.assert_array_equal(alpha, o_alpha)
    check_forward(out, o_out, _exact_forward(s, w, h, s["colors"], bg))  # bit-exact forward (D >= 128: through GAGS_FWD_EXACT; default within FWD_SPLIT_TOL)
    # gsplat-order oracle (T rebuilt back to front from 1 - render_alpha: cancellation-prone on saturated
    # pixels) and forward-order oracle (same sum, alpha*T recomputed front to back).  The VALU kernels follow
    # the former, the MFMA colours-only kernel the latter; either way the looser gsplat-order bound holds.
    o_vc, _, _, _ = oracle.raster_bwd(oinfo["means2d"], oinfo["conics"], s["opacities"], s["colors"], bg, w, h,
                                      oinfo["isect_offsets"], oinfo["flatten_ids"], o_alpha, oinfo["last_ids"],
                                      v_out, None, colors_only=True)
    o_vf = oracle.raster_bwd_colors_fwdorder(oinfo["means2d"], oinfo["conics"], s["opacities"], d, w, h,
                                             oinfo["isect_offsets"], oinfo["flatten_ids"], v_out, n)
    assert min(rel_l2(grads["colors"], o_vc), rel_l2(grads["colors"], o_vf)) <= GRAD_TOL
    e_gsplat = rel_l2(grads["colors"], o_vc)
    assert e_gsplat <= (GSPLAT_ORDER_TOL_SATURATED if mult >= 16.0 else GSPLAT_ORDER_TOL), e_gsplat


def test_mfma_and_valu_paths_agree_bitwise(oracle):
    """The kernel families implement the same fmaf chain: identical bits, D=128 (split, fused, VALU)."""
    from gags_amd import _lib
    n, w, h, d = 4000, 192, 144, 128
    s = scene_arrays(n, d, w, h, seed=12, view=2, scale_mult=5.0)
    bg = np.full(d, 0.7, np.float32)
    a, aa, ia, _ = _run_gpu(s, w, h, s["colors"], bg, flags=_lib.GAGS_FWD_EXACT)
    for flags in (_lib.GAGS_FWD_NO_MFMA, _lib.GAGS_FWD_FUSED):
        b, ab, ib, _ = _run_gpu(s, w, h, s["colors"], bg, flags=flags)
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(aa, ab)
        np.testing.assert_array_equal(ia["last_ids"].cpu().numpy(), ib["last_ids"].cpu().numpy())


def test_backward_flavours_agree_and_staged_is_deterministic(oracle):
    """Colours-only backward at D=256: the staged (default) and the float-atomic kernels compute the same sum,
    after the split and after the fused forward; the staged one uses no atomics, so two runs are bit-identical."""
    from gags_amd import _lib
    n, w, h, d = 6000, 208, 160, 256
    s = scene_arrays(n, d, w, h, seed=14, view=5, scale_mult=6.0)
    bg = np.full(d, 0.1, np.float32)
    v_out = np.random.default_rng(3).standard_normal((h, w, d)).astype(np.float32)
    o_out, o_alpha, oinfo = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"],
                                                 s["viewmat"], s["K"], bg, w, h)
    o_vf = oracle.raster_bwd_colors_fwdorder(oinfo["means2d"], oinfo["conics"], s["opacities"], d, w, h,
                                             oinfo["isect_offsets"], oinfo["flatten_ids"], v_out, n)
    g = {}
    for name, flags in (("block", 0), ("block2", 0), ("atomic", _lib.GAGS_BWD_ATOMIC),
                        ("fused_atomic", _lib.GAGS_FWD_FUSED)):
        _, _, _, gr = _run_gpu(s, w, h, s["colors"], bg, flags=flags, v_out=v_out)
        g[name] = gr["colors"]
        assert rel_l2(g[name], o_vf) <= GRAD_TOL, name
    np.testing.assert_array_equal(g["block"], g["block2"])  # no atomics anywhere: reproducible bits
    culled = oinfo["radii"] == 0
    assert np.all(g["block"][culled] == 0)  # v_colors written in full


def test_fp16_feature_table_is_exact_on_the_rounded_table(oracle):
    """BASELINE.json configs[4]: fp16 feature storage.  With GAGS_FWD_EXACT the feature pass widens the halves exactly and
    runs the oracle's fp32 chain: against the oracle on the fp16-ROUNDED table the render is bit-identical (tolerance 0).  The
    DEFAULT (round 6) contracts on the bf16 matrix cores -- the half as two exact bf16 terms, the weight as three, five
    product terms -- and is within FWD_SPLIT_TOL of that chain like the fp32 table's default; below 128 channels nothing
    changes.  The gradient (returned in the table's dtype) agrees to fp16 rounding."""
    n, w, h, d = 4000, 192, 144, 256
    s = scene_arrays(n, d, w, h, seed=51, view=2, scale_mult=5.0)
    table = torch.from_numpy(s["colors"]).half()
    rounded = table.float().numpy()
    bg = np.full(d, 0.2, np.float32)
    v_out = np.random.default_rng(4).standard_normal((h, w, d)).astype(np.float32)
    o_out, o_alpha, oinfo = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], rounded, s["viewmat"],
                                                 s["K"], bg, w, h)
    from gags_amd.rasterization import rasterization
    cols = table.cuda().requires_grad_(True)
    out, alphas, info = rasterization(to_dev(s["means"]), to_dev(s["quats"]), to_dev(s["scales"]), to_dev(s["opacities"]), cols,
                                      to_dev(s["viewmat"])[None], to_dev(s["K"])[None], w, h, backgrounds=to_dev(bg)[None])
    assert out.dtype == torch.float32
    from gags_amd import _lib
    with torch.no_grad():
        ex = rasterization(to_dev(s["means"]), to_dev(s["quats"]), to_dev(s["scales"]), to_dev(s["opacities"]), cols.detach(),
                           to_dev(s["viewmat"])[None], to_dev(s["K"])[None], w, h, backgrounds=to_dev(bg)[None],
                           raster_flags=_lib.GAGS_FWD_EXACT)[0]
    check_forward(out[0].detach().cpu().numpy(), o_out, ex[0].cpu().numpy())
    np.testing.assert_array_equal(alphas[0, ..., 0].detach().cpu().numpy(), o_alpha)
    (out[0] * to_dev(v_out)).sum().backward()
    assert cols.grad.dtype == torch.float16
    o_vf = oracle.raster_bwd_colors_fwdorder(oinfo["means2d"], oinfo["conics"], s["opacities"], d, w, h, oinfo["isect_offsets"],
                                             oinfo["flatten_ids"], v_out, n)
    assert rel_l2(cols.grad.float().cpu().numpy(), o_vf) <= 5e-4  # fp16 rounding of the returned gradient
    # an fp32 master behind a .half() cast receives the fp32-accumulated gradient through autograd's cast
    master = torch.from_numpy(s["colors"]).cuda().requires_grad_(True)
    out2, _, _ = rasterization(to_dev(s["means"]), to_dev(s["quats"]), to_dev(s["scales"]), to_dev(s["opacities"]), master.half(),
                               to_dev(s["viewmat"])[None], to_dev(s["K"])[None], w, h, backgrounds=to_dev(bg)[None])
    assert torch.equal(out2, out)
    (out2[0] * to_dev(v_out)).sum().backward()
    assert master.grad.dtype == torch.float32 and rel_l2(master.grad.cpu().numpy(), o_vf) <= 5e-4


def test_fp16_table_on_the_16bit_matrix_cores(oracle):
    """Opt-in GAGS_FWD_F16MFMA: BASELINE.json configs[4] "fp16 features on CDNA4" on
    v_mfma_f32_32x32x16_f16.  Features exact, weights as fp16 head + tail: stated ~2^-22 per term, tested <= 2e-6 rel-L2
    of the oracle's render on the rounded table (indices and alphas stay bit-exact: they never see the feature path)."""
    from gags_amd import _lib
    from gags_amd.rasterization import rasterization
    n, w, h, d = 5000, 208, 160, 256
    s = scene_arrays(n, d, w, h, seed=71, view=1, scale_mult=6.0)
    table = torch.from_numpy(s["colors"]).half()
    bg = np.full(d, 0.4, np.float32)
    v_out = np.random.default_rng(6).standard_normal((h, w, d)).astype(np.float32)
    o_out, o_alpha, oinfo = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], table.float().numpy(),
                                                 s["viewmat"], s["K"], bg, w, h)
    o_vf = oracle.raster_bwd_colors_fwdorder(oinfo["means2d"], oinfo["conics"], s["opacities"], d, w, h, oinfo["isect_offsets"],
                                             oinfo["flatten_ids"], v_out, n)
    res = []
    for _ in range(2):
        cols = table.cuda().requires_grad_(True)
        out, alphas, info = rasterization(to_dev(s["means"]), to_dev(s["quats"]), to_dev(s["scales"]), to_dev(s["opacities"]), cols,
                                          to_dev(s["viewmat"])[None], to_dev(s["K"])[None], w, h, backgrounds=to_dev(bg)[None],
                                          raster_flags=_lib.GAGS_FWD_F16MFMA)
        (out[0] * to_dev(v_out)).sum().backward()
        res.append((out[0].detach().cpu().numpy(), cols.grad.float().cpu().numpy()))
    np.testing.assert_array_equal(alphas[0, ..., 0].detach().cpu().numpy(), o_alpha)
    np.testing.assert_array_equal(info["last_ids"].cpu().numpy(), oinfo["last_ids"])
    assert rel_l2(res[0][0], o_out) <= 2e-6
    assert np.abs(res[0][0] - o_out).max() <= 2e-6 * np.abs(o_out).max()
    np.testing.assert_array_equal(res[0][0], res[1][0])   # deterministic
    assert rel_l2(res[0][1], o_vf) <= 5e-4                # gradient returned in fp16


@pytest.mark.parametrize("d", [17, 37, 72, 130, 197, 200, 513, 1000, 1024])
def test_width_with_extra_channels(oracle, d):
    """Any D >= 16 is ONE rasterization on the matrix cores (513 = 512 CLIP channels + 1, BASELINE.json configs[4]): 128-
    channel slices, then 64, then 32-channel slices with a ragged last one; rows of an odd width are only 4-byte
    aligned.  Bit-exact forward, every gradient column from the atomic-free staged backward (forward-order sums),
    bit-reproducible."""
    n, w, h = 2500, 144, 112
    s = scene_arrays(n, d, w, h, seed=52, view=6, scale_mult=5.0)
    bg = np.linspace(0.0, 1.0, d).astype(np.float32)
    v_out = np.random.default_rng(5).standard_normal((h, w, d)).astype(np.float32)
    o_out, o_alpha, oinfo = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], s["viewmat"],
                                                 s["K"], bg, w, h)
    out, alpha, info, grads = _run_gpu(s, w, h, s["colors"], bg, v_out=v_out)
    _check_indices(info, oinfo)
    check_forward(out, o_out, _exact_forward(s, w, h, s["colors"], bg))
    o_vf = oracle.raster_bwd_colors_fwdorder(oinfo["means2d"], oinfo["conics"], s["opacities"], d, w, h, oinfo["isect_offsets"],
                                             oinfo["flatten_ids"], v_out, n)
    assert rel_l2(grads["colors"], o_vf) <= GRAD_TOL
    for c in range(d - d % 32, d):  # the ragged slice, column by column
        assert rel_l2(grads["colors"][:, c], o_vf[:, c]) <= GRAD_TOL, c
    _, _, _, grads2 = _run_gpu(s, w, h, s["colors"], bg, v_out=v_out)
    np.testing.assert_array_equal(grads["colors"], grads2["colors"])


@pytest.mark.parametrize("flags", ["default", "exact", "f16mfma"])
def test_fp16_table_with_513_channels(oracle, flags):
    """BASELINE.json configs[4] as stated: fp16 feature table AND 512 + 1 channels together (rows 1026 bytes: 2-byte
    aligned).  GAGS_FWD_EXACT: bit-exact on the rounded table; default (bf16 matrix cores, five terms): within FWD_SPLIT_TOL, the
    513th channel (a narrow slice: the exact kernel) bit-exact; round 2's opt-in f16 cores with a fixed weight scale: <= 2e-6.
    Gradient in fp16."""
    from gags_amd import _lib
    from gags_amd.rasterization import rasterization
    n, w, h, d = 3000, 160, 128, 513
    s = scene_arrays(n, d, w, h, seed=53, view=3, scale_mult=5.0)
    table = torch.from_numpy(s["colors"]).half()
    bg = np.linspace(0.0, 1.0, d).astype(np.float32)
    v_out = np.random.default_rng(7).standard_normal((h, w, d)).astype(np.float32)
    o_out, o_alpha, oinfo = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], table.float().numpy(),
                                                 s["viewmat"], s["K"], bg, w, h)
    o_vf = oracle.raster_bwd_colors_fwdorder(oinfo["means2d"], oinfo["conics"], s["opacities"], d, w, h, oinfo["isect_offsets"],
                                             oinfo["flatten_ids"], v_out, n)
    cols = table.cuda().requires_grad_(True)
    rf = {"default": 0, "exact": _lib.GAGS_FWD_EXACT, "f16mfma": _lib.GAGS_FWD_F16MFMA}[flags]
    out, alphas, info = rasterization(to_dev(s["means"]), to_dev(s["quats"]), to_dev(s["scales"]), to_dev(s["opacities"]), cols,
                                      to_dev(s["viewmat"])[None], to_dev(s["K"])[None], w, h, backgrounds=to_dev(bg)[None],
                                      raster_flags=rf)
    assert out.shape == (1, h, w, d)
    np.testing.assert_array_equal(alphas[0, ..., 0].detach().cpu().numpy(), o_alpha)
    if flags == "exact":
        np.testing.assert_array_equal(out[0].detach().cpu().numpy(), o_out)
    else:
        assert rel_l2(out[0].detach().cpu().numpy(), o_out) <= (FWD_SPLIT_TOL if flags == "default" else 2e-6)
        np.testing.assert_array_equal(out[0, ..., 512:].detach().cpu().numpy(), o_out[..., 512:])  # the tail slice stays exact
    (out[0] * to_dev(v_out)).sum().backward()
    assert cols.grad.dtype == torch.float16 and cols.grad.shape == (n, d)
    g = cols.grad.float().cpu().numpy()
    assert rel_l2(g, o_vf) <= 5e-4
    assert rel_l2(g[:, 512], o_vf[:, 512]) <= 5e-4


def test_fp16_table_with_geometry_gradients(oracle):
    """An fp16 feature table in joint training (every parameter requires grad): geometry gradients are computed from the
    exactly widened table and agree with the oracle on the rounded table; the table's own gradient comes back in fp16."""
    from gags_amd.rasterization import rasterization
    n, w, h, d = 3000, 160, 128, 64
    s = scene_arrays(n, d, w, h, seed=54, view=2, scale_mult=5.0)
    table = torch.from_numpy(s["colors"]).half()
    rounded = table.float().numpy()
    bg = np.full(d, 0.3, np.float32)
    rng = np.random.default_rng(9)
    v_out = rng.standard_normal((h, w, d)).astype(np.float32)
    v_alpha = rng.standard_normal((h, w)).astype(np.float32)
    o_out, o_alpha, oinfo = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], rounded, s["viewmat"],
                                                 s["K"], bg, w, h)
    o_vc, o_vo, o_vm, o_vcon = oracle.raster_bwd(oinfo["means2d"], oinfo["conics"], s["opacities"], rounded, bg, w, h,
                                                 oinfo["isect_offsets"], oinfo["flatten_ids"], o_alpha, oinfo["last_ids"],
                                                 v_out, v_alpha)
    cols = table.cuda().requires_grad_(True)
    opac = to_dev(s["opacities"]).requires_grad_(True)
    means = to_dev(s["means"]).requires_grad_(True)
    out, alphas, info = rasterization(means, to_dev(s["quats"]), to_dev(s["scales"]), opac, cols, to_dev(s["viewmat"])[None],
                                      to_dev(s["K"])[None], w, h, backgrounds=to_dev(bg)[None])
    np.testing.assert_array_equal(out[0].detach().cpu().numpy(), o_out)
    info["means2d"].retain_grad()
    ((out[0] * to_dev(v_out)).sum() + (alphas[0, ..., 0] * to_dev(v_alpha)).sum()).backward()
    assert cols.grad.dtype == torch.float16
    assert rel_l2(cols.grad.float().cpu().numpy(), o_vc) <= 5e-4
    assert rel_l2(opac.grad.cpu().numpy(), o_vo) <= 2e-4
    assert rel_l2(info["means2d"].grad[0].cpu().numpy(), o_vm) <= 2e-4
    assert means.grad is not None and torch.isfinite(means.grad).all()


def test_backward_kernels_on_both_matrix_pipes_are_within_tolerance(oracle):
    """The staged backward's contraction: DEFAULT = v_mfma_f32_32x32x16_f16 with fp32-equivalent split operands (weights and
    cotangent as two fp16 terms each after a per-row / per-column power-of-two scale: one fp32-level rounding per operand;
    three product terms), GAGS_BWD_F32MFMA = v_mfma_f32_32x32x2_f32.  Both inside the same 2e-5 gradient tolerance PER CHANNEL with
    cotangents spanning 12 orders of magnitude across channels (exercises the per-column scaling) and weights spanning
    the whole alpha*T range (per-row scaling), both reproducible bit for bit, and the forward is untouched."""
    from gags_amd import _lib
    n, w, h, d = 5000, 192, 144, 256
    s = scene_arrays(n, d, w, h, seed=61, view=4, scale_mult=6.0)
    bg = np.full(d, 0.3, np.float32)
    rng = np.random.default_rng(8)
    v_out = (rng.standard_normal((h, w, d)) * np.exp(rng.uniform(-14, 14, size=d))).astype(np.float32)
    o_out, o_alpha, oinfo = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], s["viewmat"],
                                                 s["K"], bg, w, h)
    o_vf = oracle.raster_bwd_colors_fwdorder(oinfo["means2d"], oinfo["conics"], s["opacities"], d, w, h, oinfo["isect_offsets"],
                                             oinfo["flatten_ids"], v_out, n)
    out, _, _, g_def = _run_gpu(s, w, h, s["colors"], bg, v_out=v_out)
    _, _, _, g_def2 = _run_gpu(s, w, h, s["colors"], bg, v_out=v_out)
    out2, _, _, g_f32 = _run_gpu(s, w, h, s["colors"], bg, v_out=v_out, flags=_lib.GAGS_BWD_F32MFMA)
    _, _, _, g_f32b = _run_gpu(s, w, h, s["colors"], bg, v_out=v_out, flags=_lib.GAGS_BWD_F32MFMA)
    check_forward(out, o_out, _exact_forward(s, w, h, s["colors"], bg))
    np.testing.assert_array_equal(out2, out)
    np.testing.assert_array_equal(g_def["colors"], g_def2["colors"])
    np.testing.assert_array_equal(g_f32["colors"], g_f32b["colors"])
    # per channel (the scales differ by 12 orders of magnitude: a global rel-L2 would only see the largest columns)
    worst = {}
    for name, g in (("split / 16-bit cores (default)", g_def["colors"]), ("fp32 matrix instructions", g_f32["colors"])):
        num = np.linalg.norm((g.astype(np.float64) - o_vf), axis=0)
        den = np.maximum(np.linalg.norm(o_vf.astype(np.float64), axis=0), 1e-300)
        worst[name] = float((num / den).max())
        assert worst[name] <= GRAD_TOL, (name, worst[name])
    print("worst channel vs the oracle:", worst)
    # per Gaussian row as well: rows whose weights are all tiny must not lose digits (per-row scaling)
    nz = np.linalg.norm(o_vf, axis=1) > 0
    rown = np.linalg.norm(g_def["colors"][nz].astype(np.float64) - o_vf[nz], axis=1) / np.linalg.norm(o_vf[nz].astype(np.float64), axis=1)
    rowf = np.linalg.norm(g_f32["colors"][nz].astype(np.float64) - o_vf[nz], axis=1) / np.linalg.norm(o_vf[nz].astype(np.float64), axis=1)
    assert np.median(rown) <= 1.5 * np.median(rowf) + 1e-9 and np.quantile(rown, 0.999) <= 2.0 * np.quantile(rowf, 0.999) + 1e-9


def test_sparse_reduce_with_overlapped_zero_fill_is_bit_identical(oracle):
    """Stage bit 128 of gags_raster_bwd_colors_staged (the gradient arrives zero-filled -- here by the opt-in overlap of
    RasterContext.overlap_zero_fill -- and the reduce stage skips the Gaussians that blended nothing): same bits as the dense reduce,
    every culled / unblended row exactly zero, for fp32 and fp16 gradients and an odd width."""
    from gags_amd import rasterization as R
    for d, half in ((256, False), (513, False), (128, True)):
        n, w, h = 5000, 176, 130
        s = scene_arrays(n, d, w, h, seed=33, view=3, scale_mult=5.0)
        v_out = np.random.default_rng(11).standard_normal((h, w, d)).astype(np.float32)
        res = []
        for overlap in (True, False):
            rctx, old_min = R.RasterContext(overlap_zero_fill=overlap), R.ZERO_FILL_MIN_ELEMS
            R.ZERO_FILL_MIN_ELEMS = 0
            try:
                cols = torch.from_numpy(s["colors"]).cuda()
                cols = (cols.half() if half else cols).requires_grad_(True)
                out, _, info = R.rasterization(to_dev(s["means"]), to_dev(s["quats"]), to_dev(s["scales"]), to_dev(s["opacities"]),
                                               cols, to_dev(s["viewmat"])[None], to_dev(s["K"])[None], w, h, context=rctx)
                (out[0] * to_dev(v_out)).sum().backward()
                torch.cuda.synchronize()
                res.append(cols.grad.clone())
            finally:
                R.ZERO_FILL_MIN_ELEMS = old_min
        assert torch.equal(res[0], res[1]), (d, half)
        assert bool((res[0][info["radii"][0] == 0] == 0).all())


def test_capacity_mode_keeps_the_counts_on_the_device_and_changes_nothing(oracle):
    """Steady state (second view of a shape onwards): no gags_read_i32 at all -- buffers sized by remembered capacities,
    sentinel keys, isect_offsets' last entry, counts read from a second stream after everything is enqueued -- and every
    output identical to the exact path, including info["n_isects"] and the sliced id arrays.  A capacity that turns out
    too small re-runs the pass and still returns the right answer."""
    from gags_amd import _lib, rasterization as R
    n, w, h, d = 6000, 208, 160, 128
    s = scene_arrays(n, d, w, h, seed=77, view=4, scale_mult=6.0)
    bg = np.full(d, 0.2, np.float32)
    v_out = np.random.default_rng(13).standard_normal((h, w, d)).astype(np.float32)
    lib = _lib.load()
    calls = []
    real = lib.gags_read_i32

    def counting(*a):
        calls.append(1)
        return real(*a)

    def once(rctx):
        out, alpha, info, grads = _run_gpu(s, w, h, s["colors"], bg, v_out=v_out, context=rctx)
        return out, alpha, info["n_isects"], info["isect_ids"].cpu().numpy(), info["flatten_ids"].cpu().numpy(), \
            info["isect_offsets"][0].cpu().numpy(), info["last_ids"].cpu().numpy(), grads["colors"]

    ref = once(R.RasterContext(capacity_mode=False))
    cap = R.RasterContext(capacity_mode=True)   # (opt-in; GAGS_CAPACITY_MODE=1 makes it the default of new contexts)
    first = once(cap)                   # learns the capacities (exact path)
    lib.gags_read_i32 = counting
    try:
        steady = once(cap)
        assert calls == [], "the steady state must not read a count back synchronously"
        # capacities far too small: both passes notice afterwards and run again
        for k in list(cap.cap_isects):
            cap.cap_isects[k] = 10
        for k in list(cap.cap_rows):
            cap.cap_rows[k] = 10
        small = once(cap)
    finally:
        lib.gags_read_i32 = real
    for got in (first, steady, small):
        assert got[2] == ref[2]
        for a, b in zip(got, ref):
            if isinstance(a, np.ndarray):
                np.testing.assert_array_equal(a, b)
    # the two overflows fell back to the exact path: the row count's through gags_read_i32, the intersection count's through
    # the early pinned copy of tile_binning (round 6: summed and sent before the depth sort, rasterization.EARLY_COUNT)
    assert len(calls) >= 1


def test_backward_by_channel_ranges_is_bit_identical(oracle):
    """The by-view multi-GPU step asks the staged backward for the gradient one 128-channel range at a time
    (RasterContext.grad_range_hook, gags_amd/dist.py): same kernels on a sub-range, so the same bits as the one-shot
    backward, and the hook sees the ranges in order on an alias of the tensor autograd receives."""
    from gags_amd import rasterization as R
    n, w, h, d = 5000, 192, 144, 384
    s = scene_arrays(n, d, w, h, seed=41, view=3, scale_mult=5.0)
    v_out = np.random.default_rng(9).standard_normal((h, w, d)).astype(np.float32)
    _, _, _, g_full = _run_gpu(s, w, h, s["colors"], None, v_out=v_out)
    seen = []
    rctx = R.RasterContext()
    rctx.grad_range_hook = lambda grad, c0, c1: seen.append((grad.data_ptr(), c0, c1))
    _, _, _, g_rng = _run_gpu(s, w, h, s["colors"], None, v_out=v_out, context=rctx)
    # ... and a render through another context in between never sees the hook (no global state)
    _, _, _, g_other = _run_gpu(s, w, h, s["colors"], None, v_out=v_out)
    np.testing.assert_array_equal(g_other["colors"], g_full["colors"])
    assert len(seen) == 3
    assert [(c0, c1) for _, c0, c1 in seen] == [(0, 128), (128, 256), (256, 384)]
    assert len({p for p, _, _ in seen}) == 1
    np.testing.assert_array_equal(g_rng["colors"], g_full["colors"])


def test_backward_through_a_range_sized_scratch_is_bit_identical(oracle):
    """Stage bit 256 of the staged backward (heavy views: the partial rows of ONE 128-channel range at a time through a
    [rows, 128] scratch instead of [rows, D]; rasterization.PROW_MAX_BYTES decides): same kernels, same order of summation,
    so the same bits as the one-shot backward -- fp32 and fp16 gradients."""
    from gags_amd import rasterization as R
    n, w, h = 5000, 192, 144
    for d, half in ((384, False), (256, True)):
        s = scene_arrays(n, d, w, h, seed=43, view=3, scale_mult=5.0)
        v_out = np.random.default_rng(9).standard_normal((h, w, d)).astype(np.float32)
        res = []
        saved = R.PROW_MAX_BYTES
        for limit in (saved, 0):
            R.PROW_MAX_BYTES = limit
            try:
                cols = torch.from_numpy(s["colors"]).cuda()
                cols = (cols.half() if half else cols).requires_grad_(True)
                out, _, _ = R.rasterization(to_dev(s["means"]), to_dev(s["quats"]), to_dev(s["scales"]), to_dev(s["opacities"]), cols,
                                            to_dev(s["viewmat"])[None], to_dev(s["K"])[None], w, h)
                (out[0] * to_dev(v_out)).sum().backward()
                torch.cuda.synchronize()
                res.append(cols.grad.clone())
            finally:
                R.PROW_MAX_BYTES = saved
        assert torch.equal(res[0], res[1]), (d, half)


@pytest.mark.parametrize("n,w,h,d,seed,view", [
    (2500, 128, 96, 16, 7, 3), (2000, 100, 70, 3, 8, None),
    (2000, 112, 80, 48, 9, 5),    # two channel chunks, the second one ragged: per-chunk bg dot, v_alpha on chunk 0 only
    (2000, 112, 80, 64, 10, None),
    (1500, 96, 64, 128, 11, 2),   # C2 width with every geometry gradient
    (1500, 96, 64, 256, 12, None),
    (1200, 96, 64, 512, 13, 4),   # C3 width: D >= 32, D % 8 == 0 take gags_raster_bwd_geom (dot pass on the matrix cores)
    (2000, 112, 80, 40, 14, 1),   # ... including a width that is no multiple of 32
    (800, 64, 48, 640, 15, None), # two accumulating channel passes of the dot kernel (512 + 128)
    (2000, 97, 61, 64, 16, 3),    # ragged image (not a multiple of the tile) through the matrix-core geometry path
    (2500, 130, 100, 256, 17, 6), # ... and at the dot kernel's full pass width
])
def test_full_backward(oracle, n, w, h, d, seed, view):
    s = scene_arrays(n, d, w, h, seed=seed, view=view, scale_mult=5.0)
    bg = np.full(d, 0.25, np.float32)
    rng = np.random.default_rng(seed)
    v_out = rng.standard_normal((h, w, d)).astype(np.float32)
    v_alpha = rng.standard_normal((h, w)).astype(np.float32)
    o_out, o_alpha, oinfo = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"],
                                                 s["viewmat"], s["K"], bg, w, h)
    o_vc, o_vo, o_vm2, o_vcon = oracle.raster_bwd(oinfo["means2d"], oinfo["conics"], s["opacities"], s["colors"], bg,
                                                  w, h, oinfo["isect_offsets"], oinfo["flatten_ids"], o_alpha,
                                                  oinfo["last_ids"], v_out, v_alpha)
    o_vmeans, o_vq, o_vs = oracle.project_bwd(s["means"], s["quats"], s["scales"], s["viewmat"], s["K"], w, h,
                                              oinfo["radii"], o_vm2, None, o_vcon)
    out, alpha, info, g = _run_gpu(s, w, h, s["colors"], bg, need_geom=True, v_out=v_out, v_alpha=v_alpha)
    check_forward(out, o_out, _exact_forward(s, w, h, s["colors"], bg))
    assert rel_l2(g["colors"], o_vc) <= GRAD_TOL
    assert rel_l2(g["opacities"], o_vo) <= 1e-4
    assert rel_l2(g["means2d"], o_vm2) <= 1e-4
    assert rel_l2(g["means"], o_vmeans) <= 1e-4
    assert rel_l2(g["quats"], o_vq) <= 1e-4
    assert rel_l2(g["scales"], o_vs) <= 1e-4


def test_wide_geometry_backward_matches_the_valu_kernel_and_is_deterministic():
    """gags_raster_bwd_geom (dot products on the matrix cores, front-to-back weights, row-sorted sums) against the VALU
    backward it replaces at wide D (1 - alpha rebuilt back to front, float atomics), with and without background /
    alpha cotangent; and bit-for-bit reproducible, which the atomic kernel is not."""
    from gags_amd import _lib
    n, w, h, d = 2500, 128, 96, 128
    s = scene_arrays(n, d, w, h, seed=21, view=2, scale_mult=5.0)
    rng = np.random.default_rng(5)
    v_out = rng.standard_normal((h, w, d)).astype(np.float32)
    v_alpha = rng.standard_normal((h, w)).astype(np.float32)
    for bg, va in ((np.full(d, 0.3, np.float32), v_alpha), (None, None)):
        _, _, _, g_new = _run_gpu(s, w, h, s["colors"], bg, need_geom=True, v_out=v_out, v_alpha=va)
        _, _, _, g_rep = _run_gpu(s, w, h, s["colors"], bg, need_geom=True, v_out=v_out, v_alpha=va)
        _, _, _, g_old = _run_gpu(s, w, h, s["colors"], bg, need_geom=True, v_out=v_out, v_alpha=va,
                                  flags=_lib.GAGS_BWD_ATOMIC)
        for k in ("means", "quats", "scales", "opacities", "means2d", "colors"):
            np.testing.assert_array_equal(g_new[k], g_rep[k])
            assert rel_l2(g_new[k], g_old[k]) <= 2e-5, k


def test_wide_geometry_backward_on_both_matrix_pipes_and_under_rescaling():
    """The dot pass of gags_raster_bwd_geom contracts on the 16-bit matrix cores with split operands (feature rows and
    cotangent as two fp16 terms each, power-of-two scales per Gaussian row and per 8x8 block); GAGS_BWD_F32MFMA keeps
    rounds 1-2's fp32-MFMA kernel.  The two agree to fp32 rounding, and -- the scales being powers of two -- multiplying
    the cotangent by 2^-30 (a mean-reduced loss) or the features by 2^12 multiplies every geometry gradient by exactly
    that factor, bit for bit."""
    from gags_amd import _lib
    n, w, h, d = 2000, 112, 80, 328          # 256-channel pass + a ragged 72-channel pass (4.5 k-steps of 16)
    s = scene_arrays(n, d, w, h, seed=33, view=4, scale_mult=5.0)
    rng = np.random.default_rng(9)
    v_out = rng.standard_normal((h, w, d)).astype(np.float32)
    v_out[:, : w // 2] *= 1e-4               # blocks of very different magnitude
    bg = np.full(d, 0.3, np.float32)
    _, _, _, g16 = _run_gpu(s, w, h, s["colors"], bg, need_geom=True, v_out=v_out, v_alpha=None)
    _, _, _, g32 = _run_gpu(s, w, h, s["colors"], bg, need_geom=True, v_out=v_out, v_alpha=None,
                            flags=_lib.GAGS_BWD_F32MFMA)
    for k in ("means", "quats", "scales", "opacities", "means2d"):
        assert rel_l2(g16[k], g32[k]) <= 3e-6, k
    _, _, _, gs = _run_gpu(s, w, h, s["colors"], bg, need_geom=True, v_out=v_out * np.float32(2.0 ** -30), v_alpha=None)
    for k in ("opacities", "means2d"):
        np.testing.assert_array_equal(gs[k], g16[k] * np.float32(2.0 ** -30))
    _, _, _, gc = _run_gpu(s, w, h, s["colors"] * np.float32(4096.0), bg * np.float32(4096.0), need_geom=True, v_out=v_out,
                           v_alpha=None)
    for k in ("opacities", "means2d"):
        np.testing.assert_array_equal(gc[k], g16[k] * np.float32(4096.0))


def test_full_backward_with_more_than_1024_slots_in_one_block(oracle):
    """A block that blends more Gaussians than the dot pass stages in LDS (SF_STAGE = 1024 slot ids and scales per 8x8
    block): the remaining slots take their ids and scales from memory.  1800 faint Gaussians (alpha ~ 1/170: the
    transmittance never reaches the 1e-4 stop) piled onto a handful of pixels, D = 40 (one 32-channel step + a ragged one).
    Reference: float64 autograd through the dense restatement -- the gsplat-order fp32 backward rebuilds T by 1800
    successive divisions and is itself ~1e-4 off on such a pile (asserted: the HIP gradients are closer)."""
    from oracle import dense_ref as dr
    n, w, h, d = 1800, 64, 48, 40
    s = scene_arrays(n, d, w, h, seed=41, view=None, scale_mult=1.0)
    rng = np.random.default_rng(41)
    s["means"] = np.tile(np.array([[0.02, 0.03, 6.0]], np.float32), (n, 1)) + rng.normal(0, 0.004, (n, 3)).astype(np.float32)
    s["scales"] = np.full((n, 3), 0.012, np.float32) * rng.uniform(0.8, 1.25, (n, 3)).astype(np.float32)
    s["opacities"] = np.full(n, 0.0065, np.float32) * rng.uniform(0.9, 1.1, n).astype(np.float32)
    bg = np.full(d, 0.1, np.float32)
    v_out = rng.standard_normal((h, w, d)).astype(np.float32)
    v_alpha = rng.standard_normal((h, w)).astype(np.float32)
    o_out, o_alpha, oi = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"],
                                              s["viewmat"], s["K"], bg, w, h)
    o_vc, o_vo, o_vm2, _ = oracle.raster_bwd(oi["means2d"], oi["conics"], s["opacities"], s["colors"], bg, w, h,
                                             oi["isect_offsets"], oi["flatten_ids"], o_alpha, oi["last_ids"], v_out, v_alpha)
    # the pile really is deeper than the staged part
    counts = np.diff(np.asarray(oi["isect_offsets"]).reshape(-1))
    assert counts.max() > 1200 and float(o_alpha.max()) > 0.99

    def tm(a, rg=False):
        return torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=rg)

    C, M2, OP = tm(s["colors"], True), tm(oi["means2d"], True), tm(s["opacities"], True)
    order = np.lexsort((np.arange(n), oi["depths"]))
    o2, a2, _, ninc = dr.composite(M2, tm(oi["conics"]), OP, C, tm(bg), w, h, oi["radii"], order)
    assert ninc == oi["n_blend"]
    ((o2 * tm(v_out)).sum() + (a2 * tm(v_alpha)).sum()).backward()
    out, alpha, info, g = _run_gpu(s, w, h, s["colors"], bg, need_geom=True, v_out=v_out, v_alpha=v_alpha)
    np.testing.assert_array_equal(out, o_out)
    for name, got, want, orc in (("colors", g["colors"], C.grad.numpy(), o_vc), ("opacities", g["opacities"], OP.grad.numpy(), o_vo),
                                 ("means2d", g["means2d"], M2.grad.numpy(), o_vm2)):
        e, e_orc = rel_l2(got, want), rel_l2(orc, want)
        assert e <= 5e-6 and e <= e_orc, (name, e, e_orc)


def test_wide_geometry_backward_in_the_sparse_slot_numbering():
    """gags_raster_bwd_geom with row_base = NULL (rows and dot products live in the forward's sparse slot space: no prefix
    sum, no readback, ~6x the keys to sort) gives the same gradients, bit for bit, as the compact numbering the wrapper
    uses -- the sums run over the same rows in the same order."""
    from gags_amd import rasterization as R
    n, w, h, d = 2000, 112, 80, 328
    s = scene_arrays(n, d, w, h, seed=34, view=2, scale_mult=5.0)
    rng = np.random.default_rng(10)
    v_out = rng.standard_normal((h, w, d)).astype(np.float32)
    v_alpha = rng.standard_normal((h, w)).astype(np.float32)
    bg = np.full(d, 0.2, np.float32)
    _, _, _, g_c = _run_gpu(s, w, h, s["colors"], bg, need_geom=True, v_out=v_out, v_alpha=v_alpha)
    R.GEOM_COMPACT_ROWS = False
    try:
        _, _, _, g_s = _run_gpu(s, w, h, s["colors"], bg, need_geom=True, v_out=v_out, v_alpha=v_alpha)
    finally:
        R.GEOM_COMPACT_ROWS = True
    for k in ("means", "quats", "scales", "opacities", "means2d", "colors"):
        np.testing.assert_array_equal(g_c[k], g_s[k])


def test_wide_geometry_backward_with_nothing_to_render():
    """Degenerate inputs through the matrix-core geometry path: every Gaussian behind the camera (no intersections at
    all), and a view that only a handful of Gaussians reach -- gradients are exact zeros where nothing blended."""
    from gags_amd.rasterization import rasterization
    n, w, h, d = 500, 64, 48, 64
    s = scene_arrays(n, d, w, h, seed=3, view=None, scale_mult=4.0)
    for shift in (-100.0, 0.0):
        means = s["means"].copy()
        means[:, 2] += shift                      # shift = -100: everything behind the camera
        leaves = {k: to_dev(v).requires_grad_(True) for k, v in
                  dict(means=means, quats=s["quats"], scales=s["scales"], opac=s["opacities"], colors=s["colors"]).items()}
        out, alphas, info = rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opac"], leaves["colors"],
                                          to_dev(s["viewmat"])[None], to_dev(s["K"])[None], w, h)
        (out[0].sum() + alphas.sum()).backward()
        torch.cuda.synchronize()
        for k, t in leaves.items():
            assert t.grad is not None and bool(torch.isfinite(t.grad).all()), k
        if shift < 0:
            assert info["n_isects"] == 0
            assert all(float(t.grad.abs().max()) == 0.0 for t in leaves.values())
        else:
            culled = info["radii"][0] == 0
            assert float(leaves["opac"].grad[culled].abs().sum()) == 0.0   # (possibly no Gaussian is culled)


def test_render_modes_and_sh(oracle):
    n, w, h = 3000, 160, 120
    s = scene_arrays(n, 3, w, h, seed=11, view=1, scale_mult=4.0)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    # RGB+ED through explicit colours
    o_out, o_alpha, oinfo = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"],
                                                 s["viewmat"], s["K"], bg, w, h, render_mode="RGB+ED")
    out, alpha, info, _ = _run_gpu(s, w, h, s["colors"], bg, render_mode="RGB+ED")
    assert out.shape == (h, w, 4)
    np.testing.assert_array_equal(out[..., :3], o_out[..., :3])
    np.testing.assert_allclose(out[..., 3], o_out[..., 3], rtol=1e-6, atol=0)
    # SH colours, degree 3
    o_out, o_alpha, oinfo = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["sh"],
                                                 s["viewmat"], s["K"], bg, w, h, sh_degree=3)
    out, alpha, info, _ = _run_gpu(s, w, h, s["sh"], bg, sh_degree=3)
    np.testing.assert_array_equal(alpha, o_alpha)
    assert rel_l2(out, o_out) <= 1e-6


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_sh_low_degrees(oracle, deg):
    """active_sh_degree < 3 (the reference raises it every 1000 iterations: train.py oneupSHdegree): the first
    (deg+1)^2 coefficients are used, the rest ignored."""
    n, w, h = 2000, 128, 96
    s = scene_arrays(n, 3, w, h, seed=20 + deg, view=deg, scale_mult=4.0)
    bg = np.array([0.0, 0.5, 1.0], np.float32)
    o_out, o_alpha, _ = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["sh"], s["viewmat"],
                                             s["K"], bg, w, h, sh_degree=deg)
    out, alpha, _, _ = _run_gpu(s, w, h, s["sh"], bg, sh_degree=deg)
    np.testing.assert_array_equal(alpha, o_alpha)
    assert rel_l2(out, o_out) <= 1e-6


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_view_direction_gradient_matches_the_reference_eval_sh(deg):
    """gags_sh_bwd_dirs + gags_sh_bwd against autograd through the reference's OWN eval_sh (tests/golden/
    shgrad_vectors.npz, float64): d colour / d means through normalize(mean - campos), and d / d coefficients."""
    import os
    from gags_amd import _lib
    from gags_amd.rasterization import _SH
    Zs = np.load(os.path.join(os.path.dirname(__file__), "golden", "shgrad_vectors.npz"))
    co = torch.from_numpy(np.ascontiguousarray(Zs["shg_coeffs"].transpose(0, 2, 1))).cuda().requires_grad_(True)
    m = torch.from_numpy(Zs["shg_means"]).cuda().requires_grad_(True)
    n = m.shape[0]
    radii = torch.ones(n, dtype=torch.int32, device="cuda")
    col = _SH.apply(co, m, torch.from_numpy(Zs["shg_campos"]).cuda(), radii, deg)
    assert np.abs(col.detach().cpu().numpy() - Zs[f"shg_col_deg{deg}"]).max() <= 2e-6
    (col * torch.from_numpy(Zs["shg_v_out"]).cuda()).sum().backward()
    want_m, want_c = Zs[f"shg_vmeans_deg{deg}"], Zs[f"shg_vcoeffs_deg{deg}"].transpose(0, 2, 1)
    if deg == 0:
        assert float(m.grad.abs().max()) == 0.0
    else:
        assert rel_l2(m.grad.cpu().numpy(), want_m) <= 2e-6
    assert rel_l2(co.grad.cpu().numpy(), want_c) <= 1e-6


def test_sh_colours_with_trainable_means_through_rasterization(oracle):
    """rasterization(..., sh_degree=3) with positions requiring grad (train.py:142 without --feature_mode): the gradient
    of the means is the projection's (K2) plus the view-direction term of the SH colours.  Checked against finite
    differences of the HIP forward itself in float32 along random directions (the two terms cannot be separated from
    outside), and the direction term alone must be non-zero."""
    from gags_amd.rasterization import rasterization
    n, w, h = 300, 64, 48
    s = scene_arrays(n, 3, w, h, seed=21, scale_mult=8.0)
    args = [to_dev(s["quats"]), to_dev(s["scales"]), to_dev(s["opacities"]), to_dev(s["sh"])]
    vm, K = to_dev(s["viewmat"])[None], to_dev(s["K"])[None]
    G = torch.from_numpy(np.random.default_rng(3).standard_normal((h, w, 3)).astype(np.float32)).cuda()

    def loss_of(means):
        out, _, _ = rasterization(means, *args, vm, K, w, h, sh_degree=3)
        return (out[0] * G).sum()

    means = to_dev(s["means"]).requires_grad_(True)
    loss_of(means).backward()
    g_full = means.grad.clone()
    # the same scene with the colours frozen at their forward values: only the projection term
    from gags_amd.rasterization import _SH
    with torch.no_grad():
        info_m = means.detach()
        campos = torch.inverse(vm[0].double())[:3, 3].float()
    means2 = to_dev(s["means"]).requires_grad_(True)
    out, _, info = rasterization(means2, *args[:3], _SH.apply(args[3], info_m, campos, torch.ones(n, dtype=torch.int32, device="cuda"), 3).detach(),
                                 vm, K, w, h)
    (out[0] * G).sum().backward()
    g_proj = means2.grad.clone()
    g_dir = g_full - g_proj
    assert float(g_dir.norm()) > 1e-3 * float(g_full.norm()) > 0
    # direction term alone = _SH's own gradient for the cotangent the rasterizer hands it (exact identity)
    cols = _SH.apply(args[3], means3 := to_dev(s["means"]).requires_grad_(True), campos, info["radii"][0], 3)
    out3, _, _ = rasterization(to_dev(s["means"]), *args[:3], cols, vm, K, w, h)
    (out3[0] * G).sum().backward()
    assert rel_l2(g_dir.cpu().numpy(), means3.grad.cpu().numpy()) <= 1e-5


def test_expected_depth_with_gradient(oracle):
    """RGB+ED when the output requires grad (the normalisation then runs through autograd, not in place)."""
    n, w, h = 1500, 96, 64
    s = scene_arrays(n, 3, w, h, seed=17, view=3, scale_mult=5.0)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    o_out, _, _ = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], s["viewmat"],
                                       s["K"], bg, w, h, render_mode="RGB+ED")
    v_out = np.zeros((h, w, 4), np.float32)
    v_out[..., :3] = np.random.default_rng(2).standard_normal((h, w, 3))
    out, _, _, grads = _run_gpu(s, w, h, s["colors"], bg, render_mode="RGB+ED", v_out=v_out)
    np.testing.assert_array_equal(out[..., :3], o_out[..., :3])
    np.testing.assert_allclose(out[..., 3], o_out[..., 3], rtol=1e-6, atol=0)
    assert np.isfinite(grads["colors"]).all() and np.abs(grads["colors"]).max() > 0


def test_no_gaussians_at_all():
    """N == 0 with colours requiring grad: background render, empty gradient, no scratch sized from garbage."""
    from gags_amd.rasterization import rasterization
    dev = torch.device("cuda", 0)
    d, w, h = 32, 48, 32
    cols = torch.zeros(0, d, device=dev, requires_grad=True)
    bg = torch.full((1, d), 0.25, device=dev)
    out, alphas, info = rasterization(torch.zeros(0, 3, device=dev), torch.zeros(0, 4, device=dev),
                                      torch.zeros(0, 3, device=dev), torch.zeros(0, device=dev), cols,
                                      torch.eye(4, device=dev)[None], torch.eye(3, device=dev)[None], w, h, backgrounds=bg)
    assert info["n_isects"] == 0
    assert torch.equal(out[0], bg.expand(h, w, d)) and float(alphas.detach().abs().max()) == 0.0
    out.sum().backward()
    assert cols.grad.shape == (0, d)


def test_empty_and_culled(oracle):
    from gags_amd.rasterization import rasterization
    w, h, d = 64, 48, 8
    s = scene_arrays(50, d, w, h, seed=3)
    means = s["means"].copy()
    means[:, 2] = -5.0  # everything behind the camera -> all culled, zero intersections
    s2 = dict(s, means=means)
    bg = np.full(d, 0.5, np.float32)
    out, alpha, info, grads = _run_gpu(s2, w, h, s["colors"], bg, v_out=np.ones((h, w, d), np.float32))
    assert info["n_isects"] == 0
    assert (info["radii"].cpu().numpy() == 0).all()
    np.testing.assert_array_equal(out, np.broadcast_to(bg, (h, w, d)))
    np.testing.assert_array_equal(alpha, np.zeros((h, w), np.float32))
    assert np.abs(grads["colors"]).max() == 0.0
    with pytest.raises(RuntimeError):
        rasterization(torch.zeros(4, 3), torch.zeros(4, 4), torch.zeros(4, 3), torch.zeros(4), torch.zeros(4, 3),
                      torch.eye(4)[None], torch.eye(3)[None], 16, 16)  # CPU tensors: no fallback, must raise


def test_render_boundary_matches_reference_contract(oracle):
    """render(...) keeps the reference's signature, reads and return dict
    (gaussian_renderer/__init__.py:19-85) and equals the oracle on the same inputs."""
    from gags_amd import synthetic as syn
    from gags_amd.gaussian_renderer import render
    w, h, n, d = 176, 130, 4000, 16
    cam = syn.make_camera(w, h, view=6, device="cuda")
    pc = syn.make_model(n, d, w, h, seed=21, device="cuda", scale0=syn.SCALE0 * 4)
    pc.training_setup()
    bg = torch.tensor([1.0, 1.0, 1.0], device="cuda")
    pkg = render(cam, pc, None, bg, feature_mode=True)
    assert pkg["render"].shape == (d, h, w)
    assert pkg["viewspace_points"].shape == (1, n, 2)
    assert pkg["visibility_filter"].dtype == torch.bool and pkg["radii"].dtype == torch.int32
    G = syn.make_cotangent(d, h, w, seed=1).cuda()
    (pkg["render"] * G).sum().backward()
    vm, K = syn.camera_matrices(cam)
    o_out, o_alpha, oinfo = oracle.rasterization(
        pc.get_xyz.detach().cpu().numpy(), pc.get_rotation.detach().cpu().numpy(),
        pc.get_scaling.detach().cpu().numpy(), pc.get_opacity.detach().cpu().numpy().reshape(-1),
        pc.get_semantic_feature.detach().cpu().numpy(), vm.cpu().numpy(), K, np.ones(d, np.float32), w, h)
    np.testing.assert_array_equal(pkg["radii"].cpu().numpy(), oinfo["radii"])
    np.testing.assert_array_equal(pkg["render"].permute(1, 2, 0).detach().cpu().numpy(), o_out)
    o_vc, _, _, _ = oracle.raster_bwd(oinfo["means2d"], oinfo["conics"], oinfo["opacities"], oinfo["colors"],
                                      np.ones(d, np.float32), w, h, oinfo["isect_offsets"], oinfo["flatten_ids"],
                                      o_alpha, oinfo["last_ids"], G.permute(1, 2, 0).cpu().numpy(), None,
                                      colors_only=True)
    assert rel_l2(pc._semantic_feature.grad.cpu().numpy(), o_vc) <= GRAD_TOL
    # RGB branches: override colour and SH
    pkg2 = render(cam, pc, None, bg, feature_mode=False, override_color=torch.rand(n, 3, device="cuda"))
    assert pkg2["render"].shape == (3, h, w)
    with torch.no_grad():
        pkg3 = render(cam, pc, None, bg, feature_mode=False, render_mode="RGB+ED")
    assert pkg3["render"].shape == (4, h, w)


@pytest.mark.gpu
def test_adam_step_matches_oracle_and_fixture(oracle):
    """R9 (scene/gaussian_model.py:192-208, train.py:221-223): the single-pass HIP Adam follows the oracle's
    operation order exactly; three steps from the committed torch.optim.Adam fixture, numel % 4 != 0."""
    import os
    import torch
    from gags_amd.optim import FeatureAdam
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "adam_vectors.npz"))
    dev = torch.device("cuda", 0)
    p = torch.nn.Parameter(torch.from_numpy(z["p0"].copy()).to(dev))
    opt = FeatureAdam([{"params": [p], "lr": float(z["lr"]), "name": "semantic_feature"}], lr=0.0, eps=1e-15)
    po, mo, vo = z["p0"].copy(), np.zeros_like(z["p0"]), np.zeros_like(z["p0"])
    for t in range(1, 4):
        g = np.ascontiguousarray(z[f"g{t}"])
        p.grad = torch.from_numpy(g).to(dev)
        opt.step()
        oracle.adam_step(po, g, mo, vo, float(z["lr"]), eps=1e-15, step=t)
        st = opt.state[p]
        np.testing.assert_array_equal(st["exp_avg"].cpu().numpy(), mo)      # same fp32 operations, same order
        np.testing.assert_array_equal(st["exp_avg_sq"].cpu().numpy(), vo)
        np.testing.assert_array_equal(p.detach().cpu().numpy(), po)
        np.testing.assert_allclose(p.detach().cpu().numpy(), z[f"p{t}"], rtol=2.5e-7, atol=1e-9)  # vs torch itself: one ulp
    assert int(st["step"].item()) == 3


@pytest.mark.gpu
def test_training_setup_uses_the_hip_optimizer_and_matches_torch_adam():
    """GaussianModel.training_setup on the GPU: same constructor arguments as the reference, the step equals the
    stock torch.optim.Adam on the same device to fp32 rounding, at a size that exercises the grid-stride loop."""
    import torch
    from gags_amd import synthetic as syn
    from gags_amd.optim import FeatureAdam
    dev = torch.device("cuda", 0)
    pc = syn.make_model(20000, 64, 64, 48, seed=3, device=dev, gen_device=dev)
    opt = pc.training_setup()
    assert isinstance(opt, FeatureAdam) and opt.param_groups[0]["eps"] == 1e-15 and opt.param_groups[0]["name"] == "semantic_feature"
    ref = torch.nn.Parameter(pc._semantic_feature.detach().clone())
    ropt = torch.optim.Adam([{"params": [ref], "lr": 1e-3}], lr=0.0, eps=1e-15)
    gen = torch.Generator(device=dev).manual_seed(5)
    for _ in range(3):
        g = torch.randn(ref.shape, device=dev, generator=gen)
        pc._semantic_feature.grad = g.clone(); ref.grad = g.clone()
        opt.step(); ropt.step()
    torch.testing.assert_close(pc._semantic_feature.detach(), ref.detach(), rtol=2.5e-7, atol=1e-9)
    with pytest.raises(RuntimeError, match="no CPU path"):
        q = torch.nn.Parameter(torch.zeros(4, 4)); q.grad = torch.ones(4, 4)
        FeatureAdam([q], lr=1e-3).step()


def test_full_size_properties_c3():
    """BASELINE's full size (C3: 1.5 M Gaussians, 1080p, D = 512), where the oracle would take minutes per view:
    size-independent properties of the operator instead.  The render is linear in the features, the colours-only
    backward is its transpose, and both are bit-reproducible."""
    from gags_amd import _lib, synthetic as syn
    from gags_amd.gaussian_renderer import render
    cfg = syn.CONFIGS["C3"]
    n, d, w, h = cfg["n"], cfg["d"], cfg["width"], cfg["height"]
    dev = torch.device("cuda", 0)
    pc = syn.make_model(n, d, w, h, seed=0, device=dev, gen_device=dev)
    pc.training_setup()
    cam = syn.make_camera(w, h, device=dev)
    bg = torch.zeros(3, device=dev)
    G = syn.make_cotangent(d, h, w, seed=1, device=dev)  # [D,H,W] view of [H,W,D] memory

    def fwd(flags=0):
        pkg = render(cam, pc, None, bg, feature_mode=True, raster_flags=flags)
        return pkg

    def grad_of(pkg, cot):
        pc._semantic_feature.grad = None
        (pkg["render"] * cot).sum().backward()
        return pc._semantic_feature.grad

    pkg = fwd()
    out = pkg["render"].detach()
    radii = pkg["radii"]
    assert out.shape == (d, h, w) and torch.isfinite(out).all()
    # (1) the exact matrix-core forward (GAGS_FWD_EXACT) and the VALU kernel are the same fmaf chain: identical bits at full
    # size; the default forward (16-bit matrix cores, split operands) is the same sum within FWD_SPLIT_TOL
    pkv = fwd(_lib.GAGS_FWD_NO_MFMA)
    pke = fwd(_lib.GAGS_FWD_EXACT)
    assert torch.equal(pke["render"].detach(), pkv["render"].detach())
    assert torch.equal(pkg["alphas"], pkv["alphas"]) and torch.equal(pkg["info"]["last_ids"], pkv["info"]["last_ids"])
    e = ((out.double() - pkv["render"].detach().double()).norm() / pkv["render"].detach().double().norm()).item()
    assert e <= FWD_SPLIT_TOL, e
    del pkv, pke
    # (2) determinism and (3) homogeneity (scaling by 2 is exact in fp32)
    g1 = grad_of(pkg, G).clone()
    g1b = grad_of(fwd(), G).clone()
    assert torch.equal(g1, g1b)  # no atomics on the default path
    del g1b
    with torch.no_grad():
        pc._semantic_feature.mul_(2.0)
    pk2 = fwd()
    assert torch.equal(pk2["render"].detach(), 2.0 * out)
    assert torch.equal(pk2["alphas"], pkg["alphas"])  # geometry only
    g2 = grad_of(pk2, 2.0 * G)
    assert torch.equal(g2, 2.0 * g1)
    del pk2, g2
    with torch.no_grad():
        pc._semantic_feature.mul_(0.5)
    # (4) transpose identity  <R f, G> = <f, R^T G>  in float64
    lhs = torch.dot(out.permute(1, 2, 0).reshape(-1).double(), G.permute(1, 2, 0).reshape(-1).double())
    rhs = torch.dot(pc._semantic_feature.detach().reshape(-1).double(), g1.reshape(-1).double())
    assert abs(lhs.item() - rhs.item()) <= 1e-5 * max(abs(lhs.item()), out.double().norm().item() * G.double().norm().item() * 1e-3)
    # (5) culled Gaussians receive exactly zero, and the gradient was written in full (no stale memory)
    assert torch.all(g1[radii <= 0] == 0)
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    # (6) the float-atomic fallback computes the same sums (different order)
    ga = grad_of(fwd(_lib.GAGS_BWD_ATOMIC), G)
    assert ((ga - g1).double().norm() / g1.double().norm()).item() <= 1e-5


def test_harness_dot_matches_float64():
    """Row H: the harness loss <render, G> (gags_dot_f32) against a float64 dot; ragged length, reproducible."""
    from gags_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(11)
    n = 3 * 1000 * 1000 + 3
    x = torch.randn(n + 4, device=dev, generator=gen)[:n]  # 16-B aligned start, n % 4 != 0
    y = torch.randn(n + 4, device=dev, generator=gen)[:n]
    nb = lib.gags_dot_scratch_bytes()
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(2):
        out = torch.empty(1, device=dev)
        _lib.check(lib.gags_dot_f32(n, _lib.ptr(x), _lib.ptr(y), _lib.ptr(out), _lib.ptr(scratch), nb, None), "dot")
        outs.append(out.item())
    ref = torch.dot(x.double(), y.double()).item()
    assert outs[0] == outs[1]
    assert abs(outs[0] - ref) <= 1e-5 * (x.double().norm() * y.double().norm()).item()


@pytest.mark.parametrize("modifier", [1.0, 0.7])
def test_raw_parameter_projection_is_bit_identical_to_the_getters(oracle, modifier):
    """gags_project_fwd_raw (R2 folded into R4): fed the STORED parameters (_rotation un-normalised, _scaling in log space,
    _opacity logits, scene/gaussian_model.py:48-61) it must produce, bit for bit, what gags_project_fwd -- and the oracle --
    produce from torch's own getters (exp, F.normalize, sigmoid, render()'s `* scaling_modifier`), the activated opacity and
    the optional activated quats / scales included."""
    from gags_amd import _lib, synthetic as syn
    from gags_amd.rasterization import _Project, _ProjectRaw
    w, h, n = 300, 200, 50_000
    p = syn.make_gaussians(n, 0, w, h, seed=31, scale0=syn.SCALE0 * 4)
    p["rotation"][:4] *= 1e-20                    # |q| < 1e-12: F.normalize's eps clamp decides the activated value
    cam = syn.make_camera(w, h, view=2, device="cuda")
    vm, K = syn.camera_matrices(cam)
    Kd = torch.from_numpy(K).cuda()
    xyz, rot, slog, logit = (p[k].cuda() for k in ("xyz", "rotation", "scaling_log", "opacity_logit"))
    q_act, s_act, o_act = torch.nn.functional.normalize(rot), torch.exp(slog) * modifier, torch.sigmoid(logit)
    cfg = (w, h, 0.3, 0.01, 1e10, 0.0)
    want = _Project.apply(xyz, q_act, s_act, vm, Kd, *cfg)
    got = _ProjectRaw.apply(xyz, rot, slog, logit, vm, Kd, *cfg, modifier, True)
    for a, b, name in zip(got[:5], want, ("radii", "means2d", "depths", "conics", "tiles_per_gauss")):
        assert torch.equal(a, b), name
    assert torch.equal(got[5], o_act.reshape(-1))
    # the per-Gaussian record table written on the way == what gags_pack_isects builds from the arrays (visible Gaussians)
    rec = torch.zeros(n, 8, device="cuda")
    _lib.check(_lib.load().gags_pack_isects(n, 1, _lib.ptr(torch.zeros(1, dtype=torch.int32, device="cuda")), _lib.ptr(want[1]),
                                            _lib.ptr(want[3]), _lib.ptr(o_act.reshape(-1).contiguous()), _lib.ptr(want[0]),
                                            _lib.ptr(rec), None, None), "gags_pack_isects")
    vis = want[0] > 0
    assert torch.equal(got[6][vis], rec[vis])
    assert int((want[0] > 0).sum()) > n // 2
    o_radii, o_m2d, o_depths, o_conics = oracle.project_fwd(xyz.cpu().numpy(), q_act.cpu().numpy(), s_act.cpu().numpy(),
                                                            vm.cpu().numpy(), K, w, h)
    np.testing.assert_array_equal(got[0].cpu().numpy(), o_radii)
    np.testing.assert_array_equal(got[1].cpu().numpy(), o_m2d)
    np.testing.assert_array_equal(got[3].cpu().numpy(), o_conics)
    # the optional activated copies (for callers that keep them)
    lib = _lib.load()
    outs = [torch.empty(n, dtype=torch.int32, device="cuda"), torch.empty(n, 2, device="cuda"), torch.empty(n, device="cuda"),
            torch.empty(n, 3, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda"), torch.empty(n, device="cuda"),
            torch.empty(n, 4, device="cuda"), torch.empty(n, 3, device="cuda")]
    _lib.check(lib.gags_project_fwd_raw(n, _lib.ptr(xyz), _lib.ptr(rot), _lib.ptr(slog), _lib.ptr(logit.reshape(-1).contiguous()),
                                        modifier, _lib.ptr(vm.contiguous()), _lib.ptr(Kd), w, h, 0.3, 0.01, 1e10, 0.0,
                                        *[_lib.ptr(t) for t in outs], None, None), "gags_project_fwd_raw")
    assert torch.equal(outs[6], q_act) and torch.equal(outs[7], s_act)


def test_raw_parameter_projection_gradients_match_autograd_through_the_getters():
    """render() with every parameter trainable: the raw-parameter path (getters and their backward inside the projection
    kernels) against the getter path (torch's exp / normalize / sigmoid and their autograd): same render bit for bit, same
    feature gradient bit for bit, geometry gradients of the STORED parameters to fp32 rounding."""
    import gags_amd.gaussian_renderer as gr
    from gags_amd import synthetic as syn
    w, h, n, d = 176, 130, 6000, 32
    cam = syn.make_camera(w, h, view=5, device="cuda")
    bg = torch.tensor([0.5, 0.5, 0.5], device="cuda")
    G = syn.make_cotangent(d, h, w, seed=4).cuda()
    res = {}
    for raw in (True, False):
        pc = syn.make_model(n, d, w, h, seed=33, device="cuda", scale0=syn.SCALE0 * 5)
        pc.training_setup()
        geo = [pc._xyz, pc._rotation, pc._scaling, pc._opacity]
        for q in geo:
            q.requires_grad_(True)
        gr.RAW_PARAMS = raw
        try:
            pkg = gr.render(cam, pc, None, bg, feature_mode=True, scaling_modifier=0.9)
            (pkg["render"] * G).sum().backward()
        finally:
            gr.RAW_PARAMS = True
        res[raw] = (pkg["render"].detach().clone(), pc._semantic_feature.grad.clone(), [q.grad.clone() for q in geo],
                    pkg["viewspace_points"].grad.clone())
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert torch.equal(res[True][3], res[False][3])  # d loss / d means2d (densification statistics) is untouched
    for a, b, name in zip(res[True][2], res[False][2], ("xyz", "rotation", "scaling", "opacity")):
        assert a.shape == b.shape and float(b.abs().max()) > 0, name
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 1e-6, name


@pytest.mark.parametrize("w,h,n,d,scale", [(200, 138, 6000, 256, 5.0), (64, 48, 300, 128, 1.0), (333, 77, 20000, 384, 12.0),
                                           (160, 160, 3000, 640, 3.0), (48, 33, 40, 128, 20.0)])
def test_both_shapes_of_the_rows_kernel_agree_and_are_reproducible(w, h, n, d, scale):
    """The staged backward's rows kernel, default shape (a wave per 32 channels, block contributions folded in the accumulators:
    csrc/raster_bwd_rows_cw.h; three product terms of two-term operands, or with GAGS_BWD_EXACT_WEIGHTS five terms of exact
    three-term weights) against round 4's (a wave per pixel block, rows merged in LDS: GAGS_BWD_BLOCKWAVES; five terms) and the
    fp32 matrix instructions -- gradients within 2e-6 rel-L2 of each other, each bit-reproducible, each
    within the oracle bound; scenes with ragged image borders, empty tiles, one to many chunks of rows per tile, one to five
    128-channel slices (640 = 512 + 128) and splats that cover whole tiles."""
    import torch
    from gags_amd import _lib, synthetic as syn
    from gags_amd.gaussian_renderer import render
    dev = torch.device("cuda", 0)
    pc = syn.make_model(n, d, w, h, seed=4, device=dev, scale0=syn.SCALE0 * scale)
    pc.training_setup()
    cam = syn.make_camera(w, h, view=1, device=dev)
    G = syn.make_cotangent(d, h, w, seed=2, device=dev)
    out = {}
    for name, fl in (("cw", 0), ("cw5", _lib.GAGS_BWD_EXACT_WEIGHTS), ("blockwaves", _lib.GAGS_BWD_BLOCKWAVES),
                     ("f32", _lib.GAGS_BWD_F32MFMA)):
        runs = []
        for _ in range(2):
            pc._semantic_feature.grad = None
            pkg = render(cam, pc, None, torch.zeros(3, device=dev), feature_mode=True, raster_flags=fl)
            (pkg["render"] * G).sum().backward()
            runs.append(pc._semantic_feature.grad.clone())
        assert torch.equal(runs[0], runs[1]), name
        out[name] = runs[0].double()
    ref = out["f32"]
    assert float(ref.abs().max()) > 0
    for name in ("cw", "cw5", "blockwaves"):
        e = float((out[name] - ref).norm() / ref.norm())
        assert e <= 2e-6, (name, e)
    assert float((out["cw"] - out["blockwaves"]).norm() / ref.norm()) <= 2e-6
    # same operands, same five terms, same per-(row, block) scales as round 4's kernel, another summation order
    assert float((out["cw5"] - out["blockwaves"]).norm() / ref.norm()) <= 5e-7


@pytest.mark.parametrize("flags_name", ["default", "GAGS_BWD_EXACT_WEIGHTS", "GAGS_BWD_BLOCKWAVES"])
def test_colour_backward_is_exact_under_power_of_two_rescaling_and_linear(flags_name):
    """Size-independent properties of the colours-only backward (linearity).  Every scale inside the rows kernels is a
    power of two taken from the data (per weight row, per cotangent column), so multiplying the cotangent by 2^k -- a mean-reduced
    loss at 1080p x 512 channels is 2^-30 -- must multiply the gradient by exactly 2^k, bit for bit, for magnitudes an unscaled
    fp16 operand could not hold; per-channel factors likewise (column scales are per channel); and the gradient of G1 + G2 is the
    sum of the gradients within fp32 rounding."""
    import torch
    from gags_amd import _lib, synthetic as syn
    from gags_amd.gaussian_renderer import render
    flags = 0 if flags_name == "default" else getattr(_lib, flags_name)
    w, h, n, d = 176, 120, 5000, 256
    dev = torch.device("cuda", 0)
    pc = syn.make_model(n, d, w, h, seed=9, device=dev, scale0=syn.SCALE0 * 6)
    pc.training_setup()
    cam = syn.make_camera(w, h, view=3, device=dev)
    G1 = syn.make_cotangent(d, h, w, seed=5, device=dev)
    G2 = syn.make_cotangent(d, h, w, seed=6, device=dev)

    def grad(G):
        pc._semantic_feature.grad = None
        pkg = render(cam, pc, None, torch.zeros(3, device=dev), feature_mode=True, raster_flags=flags)
        pkg["render"].backward(G)
        return pc._semantic_feature.grad.clone()

    g1 = grad(G1)
    assert float(g1.abs().max()) > 0
    for k in (-30, -60, 40):
        assert torch.equal(grad(G1 * 2.0 ** k), g1 * 2.0 ** k), k
    per_ch = torch.exp2(torch.randint(-20, 21, (d, 1, 1), device=dev, generator=torch.Generator(device=dev).manual_seed(1)).float())
    assert torch.equal(grad(G1 * per_ch), g1 * per_ch.reshape(1, d))
    g2, g12 = grad(G2), grad(G1 + G2)
    e = float((g12.double() - (g1.double() + g2.double())).norm() / g12.double().norm())
    assert e <= 5e-7, e


def test_rows_kernel_never_multiplies_unwritten_scratch(oracle):
    """ADVICE r5 (medium).  In the default rows kernel a block that does not hold a tile row loads SOME slot and gives it the
    scale 0; that slot used to be slot 0 of the view when the block was empty -- written only if the view's top-left 8x8 block
    blends something.  The forward scratch is torch.empty: an unwritten slot is allocator garbage, and 0 * NaN = NaN in every
    row of the chunk.  Here nothing blends into the top-left tile (the Gaussians that touch it are made too faint to pass
    alpha >= 1/255) and the caching allocator's free memory is poisoned with NaN before the render: the gradient must be
    finite and equal to the oracle's."""
    n, w, h, d = 1500, 128, 80, 256
    s = scene_arrays(n, d, w, h, seed=21, view=None, scale_mult=5.0)
    r, m2d, _, _ = oracle.project_fwd(s["means"], s["quats"], s["scales"], s["viewmat"], s["K"], w, h)
    touches = (r > 0) & (m2d[:, 0] - r < 16) & (m2d[:, 1] - r < 16)
    assert 0 < touches.sum() < n // 2
    s["opacities"] = np.where(touches, np.float32(0.003), s["opacities"]).astype(np.float32)
    v_out = np.random.default_rng(3).standard_normal((h, w, d)).astype(np.float32)
    o_out, o_alpha, oi = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], s["viewmat"],
                                              s["K"], None, w, h)
    assert not o_alpha[:16, :16].any() and oi["isect_offsets"].reshape(-1)[1] > 0  # tile 0: intersections, nothing blended
    dev = torch.device("cuda", 0)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = torch.full((1 << 28,), float("nan"), device=dev)  # 1 GiB of NaN bit patterns left behind in the allocator's pool
    torch.cuda.synchronize()
    del junk
    from gags_amd import _lib
    o_vf = oracle.raster_bwd_colors_fwdorder(oi["means2d"], oi["conics"], s["opacities"], d, w, h, oi["isect_offsets"],
                                             oi["flatten_ids"], v_out, n)
    for fl in (0, _lib.GAGS_BWD_EXACT_WEIGHTS):
        _, alpha, info, grads = _run_gpu(s, w, h, s["colors"], None, v_out=v_out, flags=fl)
        np.testing.assert_array_equal(alpha, o_alpha)
        assert np.isfinite(grads["colors"]).all(), fl
        assert rel_l2(grads["colors"], o_vf) <= GRAD_TOL


def test_colour_backward_survives_cotangents_near_the_bottom_of_the_fp32_range():
    """ADVICE r5 (low).  The cotangent's column scale is a power of two that brings the column's largest magnitude to
    [2^14, 2^15); below ~2^-113 that power overflowed to inf (v * inf, 0 * inf = NaN).  It is clamped at 2^126 now: a cotangent
    scaled by 2^-122 gives a finite gradient, equal to the scaled gradient up to the digits such magnitudes keep."""
    from gags_amd import _lib, synthetic as syn
    from gags_amd.gaussian_renderer import render
    w, h, n, d = 176, 120, 5000, 256
    dev = torch.device("cuda", 0)
    pc = syn.make_model(n, d, w, h, seed=9, device=dev, scale0=syn.SCALE0 * 6)
    pc.training_setup()
    cam = syn.make_camera(w, h, view=3, device=dev)
    G = syn.make_cotangent(d, h, w, seed=5, device=dev)
    for fl in (0, _lib.GAGS_BWD_BLOCKWAVES):
        def grad(Gx):
            pc._semantic_feature.grad = None
            pkg = render(cam, pc, None, torch.zeros(3, device=dev), feature_mode=True, raster_flags=fl)
            pkg["render"].backward(Gx)
            return pc._semantic_feature.grad.clone()
        g1 = grad(G).double()
        gs = grad(G * 2.0 ** -122)
        assert bool(torch.isfinite(gs).all()), fl
        e = float((gs.double() * 2.0 ** 122 - g1).norm() / g1.norm())
        assert e <= 2e-2, (fl, e)


def test_persistent_gradient_buffer_is_bit_identical_and_gives_way_to_other_holders():
    """Round 6: the colours-only backward reduces into a buffer the RasterContext keeps between steps and writes only the
    rows that have partial rows now or had some in the previous step (gags_raster_bwd_colors_staged_keep) -- the rows of
    Gaussians that blend nothing (73 % at C3) are never written again.  Over alternating views (different rows every step) every
    gradient equals the plain entry's bit for bit; a gradient the caller still holds, or wrote to in place, is never
    overwritten: the step that finds the storage referenced or its version counter moved runs on a new buffer."""
    from gags_amd import synthetic as syn
    from gags_amd.gaussian_renderer import render
    from gags_amd.rasterization import RasterContext
    dev = torch.device("cuda", 0)
    n, d, w, h = 6000, 256, 200, 138
    pc = syn.make_model(n, d, w, h, seed=4, device=dev, scale0=syn.SCALE0 * 3)
    pc.training_setup()
    cams = [syn.make_camera(w, h, view=v, device=dev) for v in (0, 7, 3)]
    G = syn.make_cotangent(d, h, w, seed=2, device=dev)
    bg = torch.zeros(3, device=dev)
    plain, keep = RasterContext(), RasterContext()
    plain.keep_grad_buffer = False
    assert keep.keep_grad_buffer

    def grad(ctx, cam):
        (render(cam, pc, None, bg, feature_mode=True, context=ctx)["render"] * G).sum().backward()
        g = pc._semantic_feature.grad
        pc._semantic_feature.grad = None
        return g

    want = [grad(plain, c).clone() for c in cams]
    assert not torch.equal(want[0] != 0, want[1] != 0)  # the views touch different rows
    ptrs = set()
    for rnd in range(3):
        for i, c in enumerate(cams):
            g = grad(keep, c)
            ptrs.add(g.data_ptr())
            assert torch.equal(g, want[i]), (rnd, i)
            del g
    assert len(ptrs) == 1, "the buffer was not reused"
    # a gradient the caller keeps: the next backward must not touch it
    held = grad(keep, cams[0])
    snapshot = held.clone()
    g2 = grad(keep, cams[1])
    assert g2.data_ptr() != held.data_ptr() and torch.equal(held, snapshot) and torch.equal(g2, want[1])
    del held, g2
    # a gradient written in place through torch (weight decay, clipping ...) and then released: its rows are no longer
    # "zero except last step's": the buffer is given up, values stay right
    g3 = grad(keep, cams[2])
    g3.add_(1.0)
    del g3
    g4 = grad(keep, cams[0])
    assert torch.equal(g4, want[0])
    del g4
    # accumulation into a live .grad (zero_grad(set_to_none=False) style): the first step's buffer becomes the .grad (its
    # storage is given up), every later step adds a kept buffer's alias into it and releases it again
    acc_ctx = RasterContext()
    pc._semantic_feature.grad = None
    for k in range(5):
        (render(cams[k % 3], pc, None, bg, feature_mode=True, context=acc_ctx)["render"] * G).sum().backward()
    expect = want[0] * 2 + want[1] * 2 + want[2]
    assert float((pc._semantic_feature.grad - expect).abs().max()) <= 1e-5 * float(expect.abs().max())
    assert max(acc_ctx._kept_fails.values()) <= 1
    pc._semantic_feature.grad = None


def test_scratch_that_does_not_fit_falls_back_loudly(oracle, monkeypatch):
    """INTEGRATION.md, memory model: when the forward's slot space (1 KB per tile intersection) cannot be allocated the view
    runs on the scratch-free kernels (single-kernel forward, atomic backward) -- with a RuntimeWarning that says so (VERDICT r5:
    "the silent OOM fallback"), the same render bit for bit, the same gradient within the atomic kernel's tolerance; an fp16
    table is widened first (those kernels read fp32)."""
    from gags_amd import _lib
    from gags_amd.rasterization import rasterization
    n, w, h, d = 3000, 160, 112, 128
    s = scene_arrays(n, d, w, h, seed=14, view=2, scale_mult=5.0)
    bg = np.full(d, 0.3, np.float32)
    v_out = np.random.default_rng(8).standard_normal((h, w, d)).astype(np.float32)
    o_out, o_alpha, oi = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], s["viewmat"],
                                              s["K"], bg, w, h)
    o_vf = oracle.raster_bwd_colors_fwdorder(oi["means2d"], oi["conics"], s["opacities"], d, w, h, oi["isect_offsets"],
                                             oi["flatten_ids"], v_out, n)
    lib = _lib.load()
    monkeypatch.setattr(lib, "gags_raster_fwd_scratch_bytes", lambda *a: 1 << 52)  # "does not fit"
    args = [to_dev(s[k]) for k in ("means", "quats", "scales", "opacities")]
    for table in (to_dev(s["colors"]), to_dev(s["colors"]).half()):
        cols = table.clone().requires_grad_(True)
        with pytest.warns(RuntimeWarning, match="scratch-free kernels"):
            out, alphas, info = rasterization(*args, cols, to_dev(s["viewmat"])[None], to_dev(s["K"])[None], w, h,
                                              backgrounds=to_dev(bg)[None])
        np.testing.assert_array_equal(alphas[0, ..., 0].detach().cpu().numpy(), o_alpha)
        if table.dtype == torch.float32:
            np.testing.assert_array_equal(out[0].detach().cpu().numpy(), o_out)
        (out[0] * to_dev(v_out)).sum().backward()
        assert cols.grad.dtype == table.dtype
        assert rel_l2(cols.grad.float().cpu().numpy(), o_vf) <= (GRAD_TOL if table.dtype == torch.float32 else 5e-4)


@pytest.mark.parametrize("n,w,h,d,mult,geom", [(6000, 200, 138, 128, 14.0, False), (3000, 160, 112, 16, 20.0, False),
                                               (20000, 333, 77, 256, 12.0, False), (2500, 144, 112, 64, 16.0, True),
                                               (400, 64, 48, 128, 1.0, False)])
def test_trimmed_lists_change_nothing(oracle, n, w, h, d, mult, geom):
    """Round 6, list trimming (gags_raster_list_need / gags_trim_lists / gags_trim_last_ids; automatic for views whose forward
    scratch would exceed 48 GiB, forced here): every tile's sorted list is cut to the entries its pixels read before the tile
    is done, and the raster passes -- forward, staged backward, the wide geometry backward -- run on the cut lists.  With
    opaque, large splats most of a list lies behind saturation (asserted: the cut is real); render, alphas, last_ids (in the
    FULL lists' numbering) and every gradient are bit-identical to the untrimmed run, and the index tensors the caller sees are
    the full ones, equal to the oracle's."""
    from gags_amd.rasterization import RasterContext
    s = scene_arrays(n, d, w, h, seed=77, view=4, scale_mult=mult)
    s["opacities"] = np.clip(s["opacities"] * 0.4 + 0.6, 0.0, 0.99).astype(np.float32)  # opaque: tiles saturate early
    bg = np.full(d, 0.2, np.float32)
    rng = np.random.default_rng(5)
    v_out = rng.standard_normal((h, w, d)).astype(np.float32)
    v_alpha = rng.standard_normal((h, w)).astype(np.float32) if geom else None
    res = {}
    for name, trim in (("full", False), ("trimmed", True)):
        ctx = RasterContext()
        ctx.trim_lists = trim
        res[name] = _run_gpu(s, w, h, s["colors"], bg, need_geom=geom, v_out=v_out, v_alpha=v_alpha, context=ctx)
    (o0, a0, i0, g0), (o1, a1, i1, g1) = res["full"], res["trimmed"]
    assert i0["n_isects_trimmed"] is None and i1["n_isects_trimmed"] is not None
    assert i1["n_isects"] == i0["n_isects"]
    assert i1["n_isects_trimmed"] <= i1["n_isects"]
    if n >= 6000:  # (the dense scenes: most of a list lies behind saturation)
        assert i1["n_isects_trimmed"] < 0.8 * i1["n_isects"], (i1["n_isects_trimmed"], i1["n_isects"])
    np.testing.assert_array_equal(o1, o0)
    np.testing.assert_array_equal(a1, a0)
    for key in ("last_ids", "flatten_ids", "isect_ids", "isect_offsets"):
        assert torch.equal(i1[key], i0[key]), key
    for key in g0:
        np.testing.assert_array_equal(g1[key], g0[key], err_msg=key)
    o_out, o_alpha, oi = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], s["viewmat"],
                                              s["K"], bg, w, h)
    _check_indices(i1, oi)
    np.testing.assert_array_equal(a1, o_alpha)


@pytest.mark.gpu
@pytest.mark.parametrize("n,w,h,d,trim", [(6000, 200, 138, 128, False), (3000, 160, 112, 16, False), (5000, 144, 112, 513, False),
                                         (6000, 200, 138, 128, True), (40, 64, 48, 128, False)])
def test_row_map_enqueued_by_the_forward_changes_nothing(oracle, n, w, h, d, trim):
    """Round 6, RasterContext.early_rowmap (default ON): a forward that will be differentiated w.r.t. the colours enqueues the
    backward's row map (gags_bwd_rowmap) behind its own kernels and sends the row count to pinned memory, so that the backward
    needs neither the prefix sum nor a readback.  Render and gradient are bit-identical to the backward that does it all itself
    (also on trimmed lists, and for two views whose forwards both ran before either backward); the gradient agrees with the
    oracle's forward-order sum."""
    from gags_amd.rasterization import RasterContext, rasterization
    s = scene_arrays(n, d, w, h, seed=91, view=2, scale_mult=6.0)
    rng = np.random.default_rng(9)
    v_out = rng.standard_normal((h, w, d)).astype(np.float32)
    res = {}
    for name, on in (("backward", False), ("forward", True)):
        ctx = RasterContext()
        ctx.early_rowmap = on
        ctx.trim_lists = trim
        res[name] = _run_gpu(s, w, h, s["colors"], None, need_geom=False, v_out=v_out, v_alpha=None, context=ctx)
    (o0, a0, i0, g0), (o1, a1, i1, g1) = res["backward"], res["forward"]
    np.testing.assert_array_equal(o1, o0)
    np.testing.assert_array_equal(a1, a0)
    np.testing.assert_array_equal(g1["colors"], g0["colors"])
    o_out, o_alpha, oi = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], s["viewmat"],
                                              s["K"], np.zeros(d, np.float32), w, h)
    o_vf = oracle.raster_bwd_colors_fwdorder(oi["means2d"], oi["conics"], s["opacities"], d, w, h, oi["isect_offsets"],
                                             oi["flatten_ids"], v_out, n)
    assert rel_l2(g1["colors"], o_vf) <= GRAD_TOL
    # two forwards in flight before the first backward: each keeps its own pinned count
    ctx = RasterContext()
    means, quats, scales, opac = (to_dev(s[k]) for k in ("means", "quats", "scales", "opacities"))
    cols = to_dev(s["colors"]).requires_grad_(True)
    outs = [rasterization(means, quats, scales, opac, cols, to_dev(s["viewmat"])[None], to_dev(s["K"])[None], w, h, context=ctx)[0]
            for _ in range(2)]
    (outs[0][0] * to_dev(v_out)).sum().backward()
    g_first = cols.grad.clone()
    cols.grad = None
    (outs[1][0] * to_dev(v_out)).sum().backward()
    assert torch.equal(cols.grad, g_first)
    np.testing.assert_array_equal(g_first.cpu().numpy(), g0["colors"])


@pytest.mark.gpu
@pytest.mark.parametrize("with_bg", [False, True])
def test_sixteen_channel_render_without_a_backward_needs_no_scratch(oracle, with_bg):
    """Round 6: at D = 16 (the reference's own width, train.py:68) the feature pass rides along with the weights pass, and a
    render nothing will be differentiated through (torch.no_grad: render.py, the relevancy queries) runs that one kernel
    WITHOUT weight tiles -- no 1 KB per intersection of scratch.  Render, alphas and last_ids are bit-identical to the
    differentiable render's and to the oracle's; the peak allocation is a fraction."""
    n, w, h, d = 20000, 333, 210, 16
    s = scene_arrays(n, d, w, h, seed=5, view=1, scale_mult=5.0)
    bg = np.full(d, 0.3, np.float32) if with_bg else None
    o_out, o_alpha, oi = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], s["viewmat"],
                                              s["K"], bg if with_bg else np.zeros(d, np.float32), w, h)
    rng = np.random.default_rng(3)
    v_out = rng.standard_normal((h, w, d)).astype(np.float32)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    out_g, alpha_g, info_g, _ = _run_gpu(s, w, h, s["colors"], bg, v_out=v_out)
    peak_grad = torch.cuda.max_memory_allocated() - base
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        out_n, alpha_n, info_n, _ = _run_gpu(s, w, h, s["colors"], bg)
    peak_lean = torch.cuda.max_memory_allocated() - base
    np.testing.assert_array_equal(out_n, out_g)
    np.testing.assert_array_equal(alpha_n, alpha_g)
    assert torch.equal(info_n["last_ids"], info_g["last_ids"])
    np.testing.assert_array_equal(out_n, o_out)
    np.testing.assert_array_equal(alpha_n, o_alpha)
    _check_indices(info_n, oi)
    n_isects = int(info_n["n_isects"])
    assert peak_grad - peak_lean > 900 * n_isects, (peak_grad, peak_lean, n_isects)   # (the slot space: ~1 KB per intersection)


def test_intersection_count_sent_before_the_sorts_is_the_prefix_sums_total():
    """rasterization.EARLY_COUNT: the count the host sizes the id lists with is the plain sum of the per-Gaussian tile counts,
    read back before the depth sort and the prefix sum are enqueued; it must be the total the prefix sum arrives at (the last
    entry of the offsets buffer) and leave every index tensor as the late readback leaves it."""
    from gags_amd import rasterization as R
    n, w, h, d = 5000, 200, 152, 16
    s = scene_arrays(n, d, w, h, seed=5, view=2, scale_mult=3.0)
    got = {}
    old = R.EARLY_COUNT
    try:
        for flag in (True, False):
            R.EARLY_COUNT = flag
            _, _, info, _ = _run_gpu(s, w, h, s["colors"], None, context=R.RasterContext())
            got[flag] = (info["n_isects"], info["isect_ids"].cpu().numpy(), info["flatten_ids"].cpu().numpy(),
                         info["isect_offsets"][0].cpu().numpy())
    finally:
        R.EARLY_COUNT = old
    assert got[True][0] == got[False][0] > 0
    for a, b in zip(got[True][1:], got[False][1:]):
        np.testing.assert_array_equal(a, b)
