"""Full-size parity: the HIP path against the C oracle at BASELINE.json's own sizes.

C1 (10 k / 256x256 / D=3: SH, override colour, RGB+ED), C2 (500 k / 1280x720 / D=128), C3 (1.5 M / 1920x1080 /
D=512) forward + colours-only backward, and a C5-size (4 M Gaussians, 1080p) forward.  Same assertions as the
small cases of test_parity_gpu.py: every index tensor and the forward render bit-exact, colour gradients within
GRAD_TOL (rel-L2) of the oracle's double-accumulated sums.  The oracle is OpenMP-parallel over tiles: a full C3
fwd + bwd takes a few seconds on the GPU box's host cores.

Also here: the accuracy claim of DESIGN.md section 2 as a test -- the HIP colours gradient and the gsplat-order
oracle gradient against the float64 dense restatement on a saturated scene.
"""
import numpy as np
import pytest
import torch

from helpers import FWD_SPLIT_TOL, rel_l2, scene_arrays, to_dev

pytestmark = pytest.mark.gpu

GRAD_TOL = 2e-5          # HIP (forward-order alpha*T, fixed summation order) vs the forward-order oracle
GSPLAT_ORDER_TOL = 2e-4  # vs the gsplat-order oracle (T rebuilt back to front; cancellation on saturated pixels)


def _big_equal(gpu_t, host_np, chunk=1 << 27):
    """torch.equal of a device tensor and a host array without a second full-size host copy."""
    flat = gpu_t.reshape(-1)
    ref = torch.from_numpy(np.ascontiguousarray(host_np)).reshape(-1)
    assert flat.numel() == ref.numel(), (flat.numel(), ref.numel())
    for o in range(0, flat.numel(), chunk):
        if not torch.equal(flat[o:o + chunk], ref[o:o + chunk].to(flat.device)):
            return False
    return True


def _big_rel_l2(gpu_t, host_np, chunk=1 << 27):
    flat = gpu_t.reshape(-1)
    ref = torch.from_numpy(np.ascontiguousarray(host_np)).reshape(-1)
    num = den = 0.0
    for o in range(0, flat.numel(), chunk):
        r = ref[o:o + chunk].to(flat.device).double()
        num += float(((flat[o:o + chunk].double() - r) ** 2).sum())
        den += float((r ** 2).sum())
    return (num / max(den, 1e-300)) ** 0.5


def _check_big_forward(out_t, o_out, rerender_exact):
    """The default render (D >= 128: 16-bit matrix cores on split operands) within FWD_SPLIT_TOL of the oracle's fmaf chain,
    and the GAGS_FWD_EXACT render of the same inputs bit-identical to it."""
    if o_out.shape[-1] < 128:
        assert _big_equal(out_t, o_out), "forward render differs from the oracle"
        return
    e = _big_rel_l2(out_t, o_out)
    assert e <= FWD_SPLIT_TOL, e
    with torch.no_grad():
        ex = rerender_exact()
    assert _big_equal(ex, o_out), "exact forward render differs from the oracle"


def _activated(n, d, w, h, seed, scale0=None):
    """Activated parameters of a synthetic scene, generated on the GPU (what crosses the rasterization boundary)."""
    from gags_amd import synthetic as syn
    dev = torch.device("cuda", 0)
    pc = syn.make_model(n, d, w, h, seed=seed, device=dev, gen_device=dev, scale0=scale0 or syn.SCALE0)
    cam = syn.make_camera(w, h, device=dev)
    vm, K = syn.camera_matrices(cam)
    with torch.no_grad():
        t = dict(means=pc.get_xyz.contiguous(), quats=pc.get_rotation.contiguous(), scales=pc.get_scaling.contiguous(),
                 opacities=pc.get_opacity.reshape(-1).contiguous(), colors=pc.get_semantic_feature.detach().contiguous())
    return t, vm.contiguous(), torch.from_numpy(K).to(dev), pc, cam


def _full_size_case(oracle, n, w, h, d, seed, backward=True, scale0=None, bgv=0.0):
    from gags_amd.rasterization import rasterization
    dev = torch.device("cuda", 0)
    t, vm, K, _, _ = _activated(n, d, w, h, seed, scale0)
    cols = t["colors"].clone().requires_grad_(backward)
    bg = None if bgv is None else torch.full((d,), bgv, device=dev)
    out, alphas, info = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], cols, vm[None], K[None], w, h,
                                      backgrounds=None if bg is None else bg[None])
    hv = {k: v.cpu().numpy() for k, v in t.items()}
    bgh = None if bg is None else bg.cpu().numpy()
    o_out, o_alpha, oi = oracle.rasterization(hv["means"], hv["quats"], hv["scales"], hv["opacities"], hv["colors"],
                                              vm.cpu().numpy(), K.cpu().numpy(), bgh, w, h)
    # index tensors and projections: bit-exact
    np.testing.assert_array_equal(info["radii"][0].cpu().numpy(), oi["radii"])
    np.testing.assert_array_equal(info["tiles_per_gauss"][0].cpu().numpy(), oi["tiles_per_gauss"])
    assert info["n_isects"] == oi["n_isects"]
    np.testing.assert_array_equal(info["isect_ids"].cpu().numpy(), oi["isect_ids"])
    np.testing.assert_array_equal(info["flatten_ids"].cpu().numpy(), oi["flatten_ids"])
    np.testing.assert_array_equal(info["isect_offsets"][0].cpu().numpy(), oi["isect_offsets"])
    np.testing.assert_array_equal(info["means2d"][0].detach().cpu().numpy(), oi["means2d"])
    np.testing.assert_array_equal(info["conics"][0].detach().cpu().numpy(), oi["conics"])
    np.testing.assert_array_equal(info["depths"][0].detach().cpu().numpy(), oi["depths"])
    np.testing.assert_array_equal(info["last_ids"].cpu().numpy(), oi["last_ids"])
    np.testing.assert_array_equal(alphas[0, ..., 0].detach().cpu().numpy(), o_alpha)
    from gags_amd import _lib
    _check_big_forward(out[0].detach(), o_out, lambda: rasterization(
        t["means"], t["quats"], t["scales"], t["opacities"], cols.detach(), vm[None], K[None], w, h,
        backgrounds=None if bg is None else bg[None], raster_flags=_lib.GAGS_FWD_EXACT)[0][0])  # bit-exact forward
    stats = dict(n_isects=oi["n_isects"], visible=int((oi["radii"] > 0).sum()), n_blend=oi["n_blend"])
    if not backward:
        return stats
    del o_out
    gen = torch.Generator(device=dev).manual_seed(seed + 100)
    v_out = torch.randn(h, w, d, device=dev, generator=gen)
    (out[0] * v_out).sum().backward()
    del out
    v_host = v_out.cpu().numpy()
    o_vf = oracle.raster_bwd_colors_fwdorder(oi["means2d"], oi["conics"], hv["opacities"], d, w, h, oi["isect_offsets"],
                                             oi["flatten_ids"], v_host, n)
    e_f = _big_rel_l2(cols.grad, o_vf)
    culled = torch.from_numpy(oi["radii"] == 0).to(dev)
    assert bool((cols.grad[culled] == 0).all())
    del o_vf
    o_vc, _, _, _ = oracle.raster_bwd(oi["means2d"], oi["conics"], hv["opacities"], hv["colors"], bgh, w, h,
                                      oi["isect_offsets"], oi["flatten_ids"], o_alpha, oi["last_ids"], v_host, None,
                                      colors_only=True)
    e_g = _big_rel_l2(cols.grad, o_vc)
    assert e_f <= GRAD_TOL, e_f
    assert e_g <= GSPLAT_ORDER_TOL, e_g
    stats.update(err_fwdorder=e_f, err_gsplat_order=e_g)
    return stats


def test_c1_exact_rgb_sh_override_ed(oracle):
    """BASELINE.json configs[0]: 10 k Gaussians, 256x256, 3-channel RGB -- through render(...), every colour branch
    of gaussian_renderer/__init__.py:44-53 and the RGB+ED mode of render.py:118."""
    from gags_amd import synthetic as syn
    from gags_amd.gaussian_renderer import render
    cfg = syn.CONFIGS["C1"]
    n, w, h = cfg["n"], cfg["width"], cfg["height"]
    t, vm, K, pc, cam = _activated(n, 3, w, h, seed=0, scale0=syn.SCALE0 * 4)  # 256^2 at fx = 0.9 W: x4 = the 1080p footprint in pixels
    hv = {k: v.cpu().numpy() for k, v in t.items()}
    vmh, Kh = vm.cpu().numpy(), K.cpu().numpy()
    bg = torch.tensor([0.2, 0.5, 0.9], device="cuda")
    bgh = bg.cpu().numpy()
    # (a) SH colours, active_sh_degree = 3
    with torch.no_grad():
        pkg = render(cam, pc, None, bg, feature_mode=False)
    sh = pc.get_features.detach().cpu().numpy()
    o_out, o_alpha, oi = oracle.rasterization(hv["means"], hv["quats"], hv["scales"], hv["opacities"], sh, vmh, Kh, bgh, w, h,
                                              sh_degree=3)
    np.testing.assert_array_equal(pkg["radii"].cpu().numpy(), oi["radii"])
    np.testing.assert_array_equal(pkg["info"]["isect_ids"].cpu().numpy(), oi["isect_ids"])
    np.testing.assert_array_equal(pkg["info"]["flatten_ids"].cpu().numpy(), oi["flatten_ids"])
    np.testing.assert_array_equal(pkg["info"]["last_ids"].cpu().numpy(), oi["last_ids"])
    np.testing.assert_array_equal(pkg["alphas"].reshape(h, w).cpu().numpy(), o_alpha)
    assert rel_l2(pkg["render"].permute(1, 2, 0).cpu().numpy(), o_out) <= 1e-6
    # (b) override colour, with the gradient
    gen = torch.Generator(device="cuda").manual_seed(5)
    oc = torch.rand(n, 3, device="cuda", generator=gen).requires_grad_(True)
    pkg = render(cam, pc, None, bg, feature_mode=False, override_color=oc)
    o_out, o_alpha, oi = oracle.rasterization(hv["means"], hv["quats"], hv["scales"], hv["opacities"],
                                              oc.detach().cpu().numpy(), vmh, Kh, bgh, w, h)
    np.testing.assert_array_equal(pkg["render"].permute(1, 2, 0).detach().cpu().numpy(), o_out)  # bit-exact
    G = syn.make_cotangent(3, h, w, seed=1, device="cuda")
    (pkg["render"] * G).sum().backward()
    o_vc, _, _, _ = oracle.raster_bwd(oi["means2d"], oi["conics"], hv["opacities"], oc.detach().cpu().numpy(), bgh, w, h,
                                      oi["isect_offsets"], oi["flatten_ids"], o_alpha, oi["last_ids"],
                                      G.permute(1, 2, 0).contiguous().cpu().numpy(), None, colors_only=True)
    assert rel_l2(oc.grad.cpu().numpy(), o_vc) <= GRAD_TOL
    # (c) RGB+ED (render.py:118,127-133)
    with torch.no_grad():
        pkg = render(cam, pc, None, bg, feature_mode=False, override_color=oc.detach(), render_mode="RGB+ED")
    o_out, _, _ = oracle.rasterization(hv["means"], hv["quats"], hv["scales"], hv["opacities"], oc.detach().cpu().numpy(),
                                       vmh, Kh, bgh, w, h, render_mode="RGB+ED")
    got = pkg["render"].permute(1, 2, 0).cpu().numpy()
    np.testing.assert_array_equal(got[..., :3], o_out[..., :3])
    np.testing.assert_allclose(got[..., 3], o_out[..., 3], rtol=1e-6, atol=0)


def test_c2_exact_forward_and_gradient(oracle):
    """BASELINE.json configs[1]: 500 k Gaussians, 1280x720, D = 128, fwd + bwd, against the oracle in full."""
    from gags_amd import synthetic as syn
    c = syn.CONFIGS["C2"]
    st = _full_size_case(oracle, c["n"], c["width"], c["height"], c["d"], seed=0)
    assert st["n_isects"] > 2 * st["visible"] > 0


def test_c3_exact_forward_and_gradient(oracle):
    """BASELINE.json configs[2], the metric's configuration: 1.5 M Gaussians, 1920x1080 (120x68 tiles, last row
    half covered), D = 512 -- the very inputs bench.py times (seed 0), against the oracle in full."""
    from gags_amd import synthetic as syn
    c = syn.CONFIGS["C3"]
    st = _full_size_case(oracle, c["n"], c["width"], c["height"], c["d"], seed=0)
    assert st["n_isects"] > 5_000_000 and st["n_blend"] > 100_000_000


def test_c3_heavy_splats_forward_and_gradient(oracle):
    """The C3 geometry at SURVEY 8d's literal scale constant (0.004 z_mean: ~4.4x larger splats, I/V ~ 13), at
    D = 128 to bound the run time of the oracle: long per-tile lists, saturation, early termination at 1080p."""
    from gags_amd import synthetic as syn
    c = syn.CONFIGS["C3"]
    st = _full_size_case(oracle, c["n"], c["width"], c["height"], 128, seed=0, scale0=syn.SCALE0_SURVEY, bgv=1.0)
    assert st["n_isects"] > 10 * st["visible"] > 0


def _heavy_tile_sample(oracle, n, w, h, d, step, fp16_table=False, min_isects_per_visible=10, scale0=None):
    """A heavy-splat view whose compositing the oracle can only afford on every `step`-th tile: projection, binning and
    sorting are checked in full; on the sampled tiles the exact render is bit-identical, the default render within
    FWD_SPLIT_TOL, and -- with a cotangent that is zero outside them, so that the gradient is exactly the sampled tiles'
    contribution -- the colours gradient within GRAD_TOL (5e-4 when it is returned in fp16)."""
    from gags_amd import _lib, synthetic as syn
    from gags_amd.rasterization import rasterization
    dev = torch.device("cuda", 0)
    torch.cuda.reset_peak_memory_stats()
    t, vm, K, _, _ = _activated(n, d, w, h, 0, syn.SCALE0_SURVEY if scale0 is None else scale0)
    table = t["colors"].half() if fp16_table else t["colors"]
    cols = table.clone().requires_grad_(True)
    bg = torch.full((d,), 0.25, device=dev)
    out, alphas, info = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], cols, vm[None], K[None], w, h,
                                      backgrounds=bg[None])
    hv = {k: v.cpu().numpy() for k, v in t.items() if k != "colors"}
    host_cols = table.float().cpu().numpy()
    o_out, o_alpha, oi = oracle.rasterization(hv["means"], hv["quats"], hv["scales"], hv["opacities"], host_cols,
                                              vm.cpu().numpy(), K.cpu().numpy(), bg.cpu().numpy(), w, h, tile_begin=0, tile_step=step)
    del host_cols
    assert info["n_isects"] == oi["n_isects"] > min_isects_per_visible * int((oi["radii"] > 0).sum())
    np.testing.assert_array_equal(info["radii"][0].cpu().numpy(), oi["radii"])
    np.testing.assert_array_equal(info["isect_ids"].cpu().numpy(), oi["isect_ids"])
    np.testing.assert_array_equal(info["flatten_ids"].cpu().numpy(), oi["flatten_ids"])
    np.testing.assert_array_equal(info["isect_offsets"][0].cpu().numpy(), oi["isect_offsets"])
    tw, th = (w + 15) // 16, (h + 15) // 16
    sampled = (torch.arange(th * tw, device=dev) % step == 0).view(th, tw)
    pix = sampled.repeat_interleave(16, 0).repeat_interleave(16, 1)[:h, :w]   # [H,W] bool: pixels of the sampled tiles
    assert int(pix.sum()) > 50_000
    ph = pix.cpu().numpy()
    np.testing.assert_array_equal(alphas[0, ..., 0].detach().cpu().numpy()[ph], o_alpha[ph])
    np.testing.assert_array_equal(info["last_ids"].cpu().numpy()[ph], oi["last_ids"][ph])
    ref_px = torch.from_numpy(o_out[ph]).to(dev)
    # fp32 and fp16 tables alike: the default contracts on the bf16 matrix cores (exact operand terms), GAGS_FWD_EXACT is the chain
    e = ((out[0].detach()[pix].double() - ref_px.double()).norm() / ref_px.double().norm()).item()
    assert e <= FWD_SPLIT_TOL, e
    with torch.no_grad():
        ex = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], cols.detach(), vm[None], K[None], w, h,
                           backgrounds=bg[None], raster_flags=_lib.GAGS_FWD_EXACT)[0][0]
    assert torch.equal(ex[pix], ref_px), "exact render differs on the sampled tiles"
    del ex
    del o_out, ref_px
    gen = torch.Generator(device=dev).manual_seed(100)
    v_out = torch.randn(h, w, d, device=dev, generator=gen) * pix[..., None]
    (out[0] * v_out).sum().backward()
    del out
    o_vf = oracle.raster_bwd_colors_fwdorder(oi["means2d"], oi["conics"], hv["opacities"], d, w, h, oi["isect_offsets"],
                                             oi["flatten_ids"], v_out.cpu().numpy(), n, tile_begin=0, tile_step=step)
    e = _big_rel_l2(cols.grad, o_vf)
    assert e <= (5e-4 if fp16_table else GRAD_TOL), e
    return dict(n_isects=int(oi["n_isects"]), n_isects_trimmed=info.get("n_isects_trimmed"),
                peak_gib=torch.cuda.max_memory_allocated() / 2 ** 30)


def test_c3_heavy_splats_at_the_full_width_on_a_tile_sample(oracle):
    """C3H as bench.py's `heavy_workload` runs it: 1.5 M Gaussians, 1080p, D = 512, SURVEY 8d's literal splat scale (63 M
    intersections), compositing checked on every 16th tile (510 of 8160)."""
    from gags_amd import synthetic as syn
    c = syn.CONFIGS["C3"]
    st = _heavy_tile_sample(oracle, c["n"], c["width"], c["height"], 512, step=16)
    print("C3H:", st)


def test_c5_heavy_splats_on_a_tile_sample(oracle):
    """C5H: BASELINE.json configs[4]'s scene size -- 4 M Gaussians, 1080p, 512-d, fp16 feature table -- with SURVEY 8d's
    literal splat scale, i.e. a Mip-NeRF360-like ~42 tiles per Gaussian: ~170 M intersections, above the 2^27 the C ABI
    indexed until round 4 (GAGS_MAX_ISECTS is 2^28 now; the slot space is ~180 GB of the 288).  Index tensors in full,
    compositing on every 32nd tile; forward + colours-only backward."""
    from gags_amd import synthetic as syn
    c = syn.CONFIGS["C5"]
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 100 * 2 ** 30:
        pytest.skip(f"needs ~60 GiB of device memory (free: {free / 2 ** 30:.0f} GiB)")
    # (until round 6 this view needed ~200 GiB: 1 KB of forward scratch per list entry.  Its lists are now cut to what their
    # tiles read before saturation -- gags_amd.rasterization._trim_lists, automatic above 48 GiB of scratch: 169 M -> 1.5 M)
    st = _heavy_tile_sample(oracle, c["n"], c["width"], c["height"], 512, step=32, fp16_table=True, min_isects_per_visible=30)
    print("C5H:", st)
    assert st["n_isects"] > (1 << 27)
    assert st["n_isects_trimmed"] is not None and st["n_isects_trimmed"] < st["n_isects"] // 20
    assert st["peak_gib"] < 80, st


def test_c5_fp32_master_table_through_the_default_forward_on_a_tile_sample(oracle):
    """BASELINE.json configs[4]'s scene with the table a TRAINING loop holds: 4 M x 512 fp32 (scene/gaussian_model.py:188-190:
    `_semantic_feature` is an fp32 parameter) = 8 GiB, i.e. past the 32-bit byte offsets of the default feature pass --
    raster_fwd_feat_x16<true>, the instantiation that keeps row offsets in float units.  Index tensors in full; on every 16th
    tile the default render within FWD_SPLIT_TOL of the oracle's chain, the GAGS_FWD_EXACT render bit-identical to it, and
    the colours gradient of a cotangent confined to those tiles within GRAD_TOL."""
    from gags_amd import synthetic as syn
    c = syn.CONFIGS["C5"]
    assert c["n"] * 512 * 4 >= (1 << 32) > c["n"] * 512  # BIG offsets, below the 2^32-element fallback
    st = _heavy_tile_sample(oracle, c["n"], c["width"], c["height"], 512, step=16, min_isects_per_visible=2, scale0=syn.SCALE0)
    print("C5 fp32 table:", st)


def test_table_of_two_to_the_32_elements_takes_the_64_bit_offsets(oracle):
    """A feature table with N * D >= 2^32 elements (8.4 M x 512 fp32 = 16 GiB) is past what the default feature pass can
    address (raster_fwd_feat_x16 keeps row offsets in 32 bits: bytes, or floats in its BIG instantiation): such a table is
    routed to the fp32-matrix-instruction kernel with 64-bit row offsets instead of wrapping silently.  The visible
    Gaussians are the LAST 3000 rows of the table -- offsets past 2^32 elements -- all others sit behind the camera; the
    oracle renders those 3000 alone (same order, same depths): the default render must equal it bit for bit (it IS the
    oracle's chain), the gradient lands in the last rows and nowhere else."""
    from gags_amd.rasterization import rasterization
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2 ** 30:
        pytest.skip(f"needs ~45 GiB of device memory (free: {free / 2 ** 30:.0f} GiB)")
    k, d, w, h = 3000, 512, 160, 112
    n = (1 << 32) // d + 4096 + k
    assert (n - k) * d >= (1 << 32)
    s = scene_arrays(k, d, w, h, seed=9, scale_mult=4.0)
    means = torch.zeros(n, 3, device=dev); means[:, 2] = -5.0  # behind the camera: culled
    quats = torch.zeros(n, 4, device=dev); quats[:, 0] = 1.0
    scales = torch.full((n, 3), 0.01, device=dev)
    opac = torch.full((n,), 0.5, device=dev)
    table = torch.zeros(n, d, device=dev)
    means[n - k:], quats[n - k:], scales[n - k:] = to_dev(s["means"]), to_dev(s["quats"]), to_dev(s["scales"])
    opac[n - k:], table[n - k:] = to_dev(s["opacities"]), to_dev(s["colors"])
    cols = table.requires_grad_(True)
    vm, K = to_dev(s["viewmat"]), to_dev(s["K"])
    bg = np.full(d, 0.125, np.float32)
    out, alphas, info = rasterization(means, quats, scales, opac, cols, vm[None], K[None], w, h, backgrounds=to_dev(bg)[None])
    o_out, o_alpha, oi = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], s["viewmat"],
                                              s["K"], bg, w, h)
    assert oi["n_isects"] > 0 and info["n_isects"] == oi["n_isects"]
    np.testing.assert_array_equal(info["radii"][0, n - k:].cpu().numpy(), oi["radii"])
    assert int((info["radii"][0, :n - k] != 0).sum()) == 0
    np.testing.assert_array_equal(info["flatten_ids"].cpu().numpy(), oi["flatten_ids"] + (n - k))
    np.testing.assert_array_equal(alphas[0, ..., 0].detach().cpu().numpy(), o_alpha)
    np.testing.assert_array_equal(out[0].detach().cpu().numpy(), o_out)
    v_out = torch.randn(h, w, d, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    (out[0] * v_out).sum().backward()
    o_vf = oracle.raster_bwd_colors_fwdorder(oi["means2d"], oi["conics"], s["opacities"], d, w, h, oi["isect_offsets"],
                                             oi["flatten_ids"], v_out.cpu().numpy(), k)
    assert rel_l2(cols.grad[n - k:].cpu().numpy(), o_vf) <= GRAD_TOL
    assert not bool(cols.grad[:n - k].any())


@pytest.mark.parametrize("d", [512, 513])
def test_c5_as_stated_fp16_table(oracle, d):
    """BASELINE.json configs[4] as it is stated: 4 M Gaussians, 1080p, 512-d features (+ the granularity channel:
    D = 513), fp16 feature table.  Against the oracle on the fp16-rounded table: the default forward (bf16 matrix cores, the
    half as two exact bf16 terms) within FWD_SPLIT_TOL, the GAGS_FWD_EXACT forward bit-exact, colours gradient
    (returned in fp16) <= 5e-4 of the oracle's forward-order sums, the whole view."""
    from gags_amd import synthetic as syn
    from gags_amd.rasterization import rasterization
    c = syn.CONFIGS["C5"]
    n, w, h = c["n"], c["width"], c["height"]
    dev = torch.device("cuda", 0)
    t, vm, K, _, _ = _activated(n, d, w, h, seed=0)
    table = t.pop("colors").half()
    cols = table.clone().requires_grad_(True)
    out, alphas, info = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], cols, vm[None], K[None], w, h,
                                      backgrounds=torch.zeros(1, d, device=dev))
    assert out.shape == (1, h, w, d) and out.dtype == torch.float32
    hv = {k: v.cpu().numpy() for k, v in t.items()}
    rounded = table.float().cpu().numpy()
    o_out, o_alpha, oi = oracle.rasterization(hv["means"], hv["quats"], hv["scales"], hv["opacities"], rounded,
                                              vm.cpu().numpy(), K.cpu().numpy(), np.zeros(d, np.float32), w, h)
    del rounded
    assert info["n_isects"] == oi["n_isects"] and int((oi["radii"] > 0).sum()) > 3_000_000
    np.testing.assert_array_equal(info["flatten_ids"].cpu().numpy(), oi["flatten_ids"])
    np.testing.assert_array_equal(info["isect_offsets"][0].cpu().numpy(), oi["isect_offsets"])
    np.testing.assert_array_equal(info["last_ids"].cpu().numpy(), oi["last_ids"])
    np.testing.assert_array_equal(alphas[0, ..., 0].detach().cpu().numpy(), o_alpha)
    from gags_amd import _lib
    e = _big_rel_l2(out[0].detach(), o_out)
    assert e <= FWD_SPLIT_TOL, e
    if d == 513:
        assert torch.equal(out[0, ..., 512].detach().cpu(), torch.from_numpy(o_out[..., 512])), "the granularity channel runs the exact kernel"
    with torch.no_grad():
        ex = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], cols.detach(), vm[None], K[None], w, h,
                           backgrounds=torch.zeros(1, d, device=dev), raster_flags=_lib.GAGS_FWD_EXACT)[0][0]
    assert _big_equal(ex, o_out), "exact forward render differs from the oracle"
    del o_out, ex
    gen = torch.Generator(device=dev).manual_seed(100)
    v_out = torch.randn(h, w, d, device=dev, generator=gen)
    (out[0] * v_out).sum().backward()
    del out
    assert cols.grad.dtype == torch.float16
    o_vf = oracle.raster_bwd_colors_fwdorder(oi["means2d"], oi["conics"], hv["opacities"], d, w, h, oi["isect_offsets"],
                                             oi["flatten_ids"], v_out.cpu().numpy(), n)
    assert _big_rel_l2(cols.grad, o_vf) <= 5e-4
    if d == 513:  # the granularity channel on its own
        g = cols.grad[:, 512].float().cpu().numpy()
        assert rel_l2(g, o_vf[:, 512]) <= 5e-4


def test_default_forward_is_as_close_to_float64_as_the_exact_kernel(oracle):
    """What makes the 16-bit matrix-core forward the default: against the float64 sum of the SAME fp32 products (the oracle's
    alpha / transmittance chain, colour sums in double: orc_raster_fwd_acc64) on every 64th tile of the C3 view it is as
    close as the kernel that is the sequential fp32 fmaf chain -- measured 1.98e-7 against 1.99e-7 on bench.py's very inputs,
    1.94e-7 against 1.70e-7 here (rows of mixed magnitude, background 0.5): asserted within 25 % of it and below 3e-7.  And with the features rescaled by 2^60 / 2^-60 and every third row by another 2^-20 (rows of very different
    magnitude in one step) the relative error does not move: bf16 terms keep fp32's exponent range, nothing is scaled."""
    from gags_amd import _lib, synthetic as syn
    from gags_amd.rasterization import rasterization
    c = syn.CONFIGS["C3"]
    n, w, h, d, step = c["n"], c["width"], c["height"], c["d"], 64
    dev = torch.device("cuda", 0)
    t, vm, K, _, _ = _activated(n, d, w, h, 0)
    bg = torch.full((d,), 0.5, device=dev)
    tw, th = (w + 15) // 16, (h + 15) // 16
    pix = (torch.arange(th * tw, device=dev) % step == 0).view(th, tw).repeat_interleave(16, 0).repeat_interleave(16, 1)[:h, :w]
    ph = pix.cpu().numpy()
    hv = {k: v.cpu().numpy() for k, v in t.items()}
    errs = {}
    with torch.no_grad():
        for scale in (1.0, 2.0 ** 60, 2.0 ** -60, "half"):
            if scale == "half":  # an fp16 table (round 6's default: the half as TWO exact bf16 terms, five product terms)
                feat = t["colors"].clone()
                feat[::3] *= 2.0 ** -6
                feat = feat.half()
                hfeat = feat.float().cpu().numpy()
                _, _, oi = oracle.rasterization(hv["means"], hv["quats"], hv["scales"], hv["opacities"], hfeat[:, :4],
                                                vm.cpu().numpy(), K.cpu().numpy(), None, w, h, tile_begin=0, tile_step=1 << 20)
                ref = oracle.raster_fwd_acc64(oi["means2d"], oi["conics"], hv["opacities"], hfeat, bg.cpu().numpy(), w, h,
                                              oi["isect_offsets"], oi["flatten_ids"], tile_begin=0, tile_step=step)[ph]
                for name, fl in (("default", 0), ("exact", _lib.GAGS_FWD_EXACT)):
                    out = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], feat, vm[None], K[None], w, h,
                                        backgrounds=bg[None], raster_flags=fl)[0][0]
                    got = out[pix].double().cpu().numpy()
                    errs[(name, scale)] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
                    del out, got
                del ref, hfeat
                continue
            feat = t["colors"] * scale
            feat[::3] *= 2.0 ** -20
            _, _, oi = oracle.rasterization(hv["means"], hv["quats"], hv["scales"], hv["opacities"], feat[:, :4].cpu().numpy(),
                                            vm.cpu().numpy(), K.cpu().numpy(), None, w, h, tile_begin=0, tile_step=1 << 20)
            ref = oracle.raster_fwd_acc64(oi["means2d"], oi["conics"], hv["opacities"], feat.cpu().numpy(),
                                          (bg * scale).cpu().numpy(), w, h, oi["isect_offsets"], oi["flatten_ids"],
                                          tile_begin=0, tile_step=step)[ph]
            for name, fl in (("default", 0), ("exact", _lib.GAGS_FWD_EXACT)):
                out = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], feat, vm[None], K[None], w, h,
                                    backgrounds=(bg * scale)[None], raster_flags=fl)[0][0]
                got = out[pix].double().cpu().numpy()
                errs[(name, scale)] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
                del out, got
            del ref
    print("forward vs the float64 sum of the same products:", errs)
    for scale in (1.0, 2.0 ** 60, 2.0 ** -60, "half"):
        assert errs[("default", scale)] <= 1.25 * errs[("exact", scale)] + 1e-9, errs
        assert errs[("default", scale)] <= 3e-7, errs
    assert abs(errs[("default", 2.0 ** 60)] - errs[("default", 1.0)]) <= 1e-9 and \
        abs(errs[("default", 2.0 ** -60)] - errs[("default", 1.0)]) <= 1e-9, errs


def test_colour_gradient_accuracy_against_float64(oracle):
    """DESIGN.md section 2's deviation as a test.  gsplat rebuilds T in the backward from 1 - render_alpha, which
    cancels on nearly saturated pixels; the HIP colours-only backward uses the forward's own alpha*T.  Against the
    float64 dense restatement (oracle/dense_ref.py) on a saturated scene the HIP gradient must be at least as
    accurate as the gsplat-order one, and within GRAD_TOL."""
    from oracle import dense_ref as dr
    n, w, h, d = 2500, 80, 48, 128  # D % 128 == 0: the width the 16-bit-matrix-core backward serves
    s = scene_arrays(n, d, w, h, seed=31, view=None, scale_mult=14.0)
    opac = np.clip(s["opacities"] * 0.5 + 0.5, 0.0, 0.995).astype(np.float32)  # opaque: most pixels saturate
    bg = np.full(d, 0.3, np.float32)
    v_out = np.random.default_rng(7).standard_normal((h, w, d)).astype(np.float32)
    o_out, o_alpha, oi = oracle.rasterization(s["means"], s["quats"], s["scales"], opac, s["colors"], s["viewmat"], s["K"],
                                              bg, w, h)
    assert (o_alpha > 0.999).mean() > 0.5  # the scene is saturated where it matters
    o_vc, _, _, _ = oracle.raster_bwd(oi["means2d"], oi["conics"], opac, s["colors"], bg, w, h, oi["isect_offsets"],
                                      oi["flatten_ids"], o_alpha, oi["last_ids"], v_out, None, colors_only=True)

    def tm(a, rg=False):
        return torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=rg)

    C = tm(s["colors"], True)
    order = np.lexsort((np.arange(n), oi["depths"]))
    o2, _, _, ninc = dr.composite(tm(oi["means2d"]), tm(oi["conics"]), tm(opac), C, tm(bg), w, h, oi["radii"], order)
    assert ninc == oi["n_blend"]  # both statements blend the same (pixel, Gaussian) pairs
    (o2 * tm(v_out)).sum().backward()
    ref = C.grad.numpy()

    from gags_amd.rasterization import rasterization
    cols = to_dev(s["colors"]).requires_grad_(True)
    out, _, info = rasterization(to_dev(s["means"]), to_dev(s["quats"]), to_dev(s["scales"]), to_dev(opac), cols,
                                 to_dev(s["viewmat"])[None], to_dev(s["K"])[None], w, h, backgrounds=to_dev(bg)[None])
    assert rel_l2(out[0].detach().cpu().numpy(), o_out) <= FWD_SPLIT_TOL
    (out[0] * to_dev(v_out)).sum().backward()
    e_hip = rel_l2(cols.grad.cpu().numpy(), ref)
    e_gsplat = rel_l2(o_vc, ref)
    # the DEFAULT above contracts on the 16-bit matrix cores with split operands: weights and cotangent as two fp16 terms each (one
    # fp32-level rounding per operand), three product terms (csrc/raster_bwd_rows_cw.h); GAGS_BWD_EXACT_WEIGHTS: weights as three
    # terms (exact), five product terms; GAGS_BWD_F32MFMA is rounds 1-2's kernel on the fp32 matrix instructions
    from gags_amd import _lib

    def grad_with(flags):
        c_ = to_dev(s["colors"]).requires_grad_(True)
        o_, _, _ = rasterization(to_dev(s["means"]), to_dev(s["quats"]), to_dev(s["scales"]), to_dev(opac), c_,
                                 to_dev(s["viewmat"])[None], to_dev(s["K"])[None], w, h, backgrounds=to_dev(bg)[None], raster_flags=flags)
        (o_[0] * to_dev(v_out)).sum().backward()
        return c_

    cols2 = grad_with(_lib.GAGS_BWD_F32MFMA)
    e_f32 = rel_l2(cols2.grad.cpu().numpy(), ref)
    e_5 = rel_l2(grad_with(_lib.GAGS_BWD_EXACT_WEIGHTS).grad.cpu().numpy(), ref)
    assert e_5 <= e_hip * 1.02, (e_5, e_hip)   # exact weights are never worse than the three-term default
    # per channel too: the split must not be worse than fp32 matrix arithmetic in ANY column
    ref64 = ref.astype(np.float64)
    den = np.maximum(np.linalg.norm(ref64, axis=0), 1e-300)
    ch_def = np.linalg.norm(cols.grad.cpu().numpy().astype(np.float64) - ref64, axis=0) / den
    ch_f32 = np.linalg.norm(cols2.grad.cpu().numpy().astype(np.float64) - ref64, axis=0) / den
    print(f"colour gradient vs float64: default (two-term operands, three product terms) {e_hip:.2e} [worst channel {ch_def.max():.2e}], "
          f"exact weights / five terms {e_5:.2e}, "
          f"fp32 matrix instructions {e_f32:.2e} [worst channel {ch_f32.max():.2e}], gsplat-order fp32 {e_gsplat:.2e}")
    assert e_hip <= e_gsplat and e_f32 <= e_gsplat, (e_hip, e_f32, e_gsplat)
    assert e_hip <= 2e-6 and e_f32 <= 2e-6, (e_hip, e_f32)
    assert e_hip <= 1.05 * e_f32, (e_hip, e_f32)                    # what makes it the default
    assert ch_def.max() <= 1.25 * ch_f32.max(), (ch_def.max(), ch_f32.max())


def test_c2_all_gradients_against_the_oracle(oracle):
    """BASELINE.json configs[1] (500 k, 1280x720, D = 128) with EVERY gradient of SURVEY A9 / K2: features through the
    staged backward, opacity / means2d / conics through gags_raster_bwd_geom (dot products on the matrix cores), then
    means / quats / scales through the projection backward -- against the oracle's full backward on the same inputs,
    with a background and an alpha cotangent."""
    from gags_amd import synthetic as syn
    from gags_amd.rasterization import rasterization
    cfg = syn.CONFIGS["C2"]
    n, w, h, d = cfg["n"], cfg["width"], cfg["height"], cfg["d"]
    dev = torch.device("cuda", 0)
    t, vm, K, _, _ = _activated(n, d, w, h, seed=0)
    leaf = {k: v.clone().requires_grad_(True) for k, v in t.items()}
    bg = torch.full((d,), 0.25, device=dev)
    out, alphas, info = rasterization(leaf["means"], leaf["quats"], leaf["scales"], leaf["opacities"], leaf["colors"],
                                      vm[None], K[None], w, h, backgrounds=bg[None])
    gen = torch.Generator(device=dev).manual_seed(77)
    v_out = torch.randn(h, w, d, device=dev, generator=gen)
    v_alpha = torch.randn(h, w, device=dev, generator=gen)
    ((out[0] * v_out).sum() + (alphas[0, ..., 0] * v_alpha).sum()).backward()
    hv = {k: v.cpu().numpy() for k, v in t.items()}
    vmh, Kh, bgh = vm.cpu().numpy(), K.cpu().numpy(), bg.cpu().numpy()
    o_out, o_alpha, oi = oracle.rasterization(hv["means"], hv["quats"], hv["scales"], hv["opacities"], hv["colors"], vmh, Kh,
                                              bgh, w, h)
    assert _big_rel_l2(out[0].detach(), o_out) <= FWD_SPLIT_TOL
    o_vc, o_vo, o_vm2, o_vcon = oracle.raster_bwd(oi["means2d"], oi["conics"], hv["opacities"], hv["colors"], bgh, w, h,
                                                  oi["isect_offsets"], oi["flatten_ids"], o_alpha, oi["last_ids"],
                                                  v_out.cpu().numpy(), v_alpha.cpu().numpy())
    o_vmeans, o_vq, o_vs = oracle.project_bwd(hv["means"], hv["quats"], hv["scales"], vmh, Kh, w, h, oi["radii"], o_vm2,
                                              None, o_vcon)
    errs = dict(colors=_big_rel_l2(leaf["colors"].grad, o_vc), opacities=_big_rel_l2(leaf["opacities"].grad, o_vo),
                means=_big_rel_l2(leaf["means"].grad, o_vmeans), quats=_big_rel_l2(leaf["quats"].grad, o_vq),
                scales=_big_rel_l2(leaf["scales"].grad, o_vs))
    print("C2 all gradients, rel-L2 vs the oracle:", errs)
    assert errs["colors"] <= GSPLAT_ORDER_TOL
    assert max(errs[k] for k in ("opacities", "means", "quats", "scales")) <= 1e-4, errs


def test_geometry_gradient_accuracy_against_float64(oracle):
    """The same comparison for the geometry side (v_opacities, v_means2d of SURVEY A9) at wide D: gags_raster_bwd_geom
    uses the forward's own weights (T = f / alpha), the gsplat-order statement rebuilds T from 1 - render_alpha.
    Against float64 autograd through the dense restatement, on the saturated scene, the HIP gradients must be at
    least as accurate as the gsplat-order oracle's and within 2e-6 (measured 2.7e-7; the oracle: 1e-4)."""
    from oracle import dense_ref as dr
    from gags_amd.rasterization import rasterization
    n, w, h, d = 2500, 80, 48, 128
    s = scene_arrays(n, d, w, h, seed=31, view=None, scale_mult=14.0)
    opac = np.clip(s["opacities"] * 0.5 + 0.5, 0.0, 0.995).astype(np.float32)
    bg = np.full(d, 0.3, np.float32)
    rng = np.random.default_rng(7)
    v_out = rng.standard_normal((h, w, d)).astype(np.float32)
    v_alpha = rng.standard_normal((h, w)).astype(np.float32)
    o_out, o_alpha, oi = oracle.rasterization(s["means"], s["quats"], s["scales"], opac, s["colors"], s["viewmat"], s["K"],
                                              bg, w, h)
    _, o_vo, o_vm2, _ = oracle.raster_bwd(oi["means2d"], oi["conics"], opac, s["colors"], bg, w, h, oi["isect_offsets"],
                                          oi["flatten_ids"], o_alpha, oi["last_ids"], v_out, v_alpha)

    def tm(a, rg=False):
        return torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=rg)

    M2, OP = tm(oi["means2d"], True), tm(opac, True)
    order = np.lexsort((np.arange(n), oi["depths"]))
    o2, a2, _, ninc = dr.composite(M2, tm(oi["conics"]), OP, tm(s["colors"]), tm(bg), w, h, oi["radii"], order)
    assert ninc == oi["n_blend"]
    ((o2 * tm(v_out)).sum() + (a2 * tm(v_alpha)).sum()).backward()
    ref_o, ref_m = OP.grad.numpy(), M2.grad.numpy()

    leaves = {k: to_dev(v).requires_grad_(True) for k, v in
              dict(means=s["means"], quats=s["quats"], scales=s["scales"], opac=opac, colors=s["colors"]).items()}
    out, alphas, info = rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opac"], leaves["colors"],
                                      to_dev(s["viewmat"])[None], to_dev(s["K"])[None], w, h, backgrounds=to_dev(bg)[None])
    info["means2d"].retain_grad()
    ((out[0] * to_dev(v_out)).sum() + (alphas[0, ..., 0] * to_dev(v_alpha)).sum()).backward()
    e_o, e_m = rel_l2(leaves["opac"].grad.cpu().numpy(), ref_o), rel_l2(info["means2d"].grad[0].cpu().numpy(), ref_m)
    g_o, g_m = rel_l2(o_vo, ref_o), rel_l2(o_vm2, ref_m)
    print(f"geometry gradients vs float64: HIP v_opacities {e_o:.2e}, v_means2d {e_m:.2e}; "
          f"gsplat-order oracle {g_o:.2e}, {g_m:.2e}")
    assert e_o <= g_o and e_m <= g_m, (e_o, g_o, e_m, g_m)
    assert max(e_o, e_m) <= 2e-6
