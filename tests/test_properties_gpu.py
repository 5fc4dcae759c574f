"""Size-independent properties of the rasterizer ("Property" row), as `hypothesis` tests over random
small scenes, through the DEFAULT kernels of `rasterization(...)`:

* permutation invariance: the order in which the Gaussians are handed in changes neither the render nor (after undoing the
  permutation) any gradient row -- the depth sort, not the input order, decides the compositing order;
* channel independence: render(a || b) == concat(render(a), render(b)), gradients likewise;
* alphas / last_ids do not depend on the feature width D;
* linearity in the colours: render(2^k c) == 2^k render(c) bit for bit, render(c1 + c2) == render(c1) + render(c2) to fp32
  rounding; the gradient is linear in the cotangent;
* culled Gaussians receive exactly zero gradient.

Depths are made pairwise distinct by construction, so that the stable depth sort has no ties whose order would depend on the
Gaussian index: with distinct depths every statement above is an equality, not a tolerance, wherever the same
kernel serves both sides.
"""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from helpers import FWD_SPLIT_TOL

pytestmark = pytest.mark.gpu

SETTINGS = dict(max_examples=12, deadline=None, derandomize=True, database=None)


def _scene(n, d, w, h, seed, mult, view):
    """Activated parameters on the GPU with pairwise distinct depths."""
    from gags_amd import synthetic as syn
    dev = torch.device("cuda", 0)
    p = syn.make_gaussians(n, d, w, h, seed=seed, scale0=syn.SCALE0 * mult)
    g = torch.Generator().manual_seed(seed + 1)
    z = torch.linspace(syn.Z_NEAR, syn.Z_FAR, n)[torch.randperm(n, generator=g)]
    xyz = p["xyz"].clone()
    xyz[:, :2] *= (z / xyz[:, 2])[:, None]   # same screen position, new depth
    xyz[:, 2] = z
    cam = syn.make_camera(w, h, view=view, device=dev)
    vm, K = syn.camera_matrices(cam)
    t = dict(means=xyz, quats=torch.nn.functional.normalize(p["rotation"]), scales=p["scaling_log"].exp(),
             opacities=torch.sigmoid(p["opacity_logit"]).reshape(-1), colors=p["semantic_feature"])
    return {k: v.to(dev).contiguous() for k, v in t.items()}, vm.contiguous()[None], torch.from_numpy(K).to(dev)[None]


def _render(t, vm, K, w, h, colors=None, v_out=None, bg=None, flags=0):
    from gags_amd.rasterization import rasterization
    cols = (t["colors"] if colors is None else colors).clone().requires_grad_(v_out is not None)
    out, alphas, info = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], cols, vm, K, w, h,
                                      backgrounds=bg, raster_flags=flags)
    grad = None
    if v_out is not None:
        out.backward(v_out[None])
        grad = cols.grad
    return out[0].detach(), alphas[0, ..., 0].detach(), info, grad


scene_st = dict(n=st.integers(40, 1500), w=st.integers(17, 150), h=st.integers(17, 100), seed=st.integers(0, 10_000),
                mult=st.sampled_from([1.0, 3.0, 6.0, 12.0]), view=st.sampled_from([None, 0, 3, 7]))


@settings(**SETTINGS)
@given(d=st.sampled_from([16, 20, 64, 128, 256]), **scene_st)
def test_input_order_of_the_gaussians_does_not_matter(n, w, h, seed, mult, view, d):
    t, vm, K = _scene(n, d, w, h, seed, mult, view)
    dev = t["means"].device
    v_out = torch.randn(h, w, d, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
    out, alphas, info, grad = _render(t, vm, K, w, h, v_out=v_out)
    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(seed + 7))
    tp = {k: v[perm].contiguous() for k, v in t.items()}
    out_p, alphas_p, info_p, grad_p = _render(tp, vm, K, w, h, v_out=v_out)
    assert info_p["n_isects"] == info["n_isects"]
    assert torch.equal(alphas_p, alphas)
    assert torch.equal(out_p, out)                      # same sorted lists, same chain
    assert torch.equal(perm[info_p["flatten_ids"].long()], info["flatten_ids"].long())  # the same Gaussians in the same order
    assert torch.equal(info_p["radii"][0], info["radii"][0][perm])
    assert torch.equal(grad_p, grad[perm])              # fixed summation order per Gaussian: tile rows in sorted order


@settings(**SETTINGS)
@given(da=st.sampled_from([16, 32, 48, 100]), db=st.sampled_from([16, 20, 36, 64]), **scene_st)
def test_channels_are_independent(n, w, h, seed, mult, view, da, db):
    """render(a || b) == concat(render(a), render(b)), and the colour gradient splits the same way."""
    t, vm, K = _scene(n, da + db, w, h, seed, mult, view)
    dev = t["means"].device
    v_out = torch.randn(h, w, da + db, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
    bg = torch.linspace(0.0, 1.0, da + db, device=dev)[None]
    full, a_f, _, g_f = _render(t, vm, K, w, h, v_out=v_out, bg=bg)
    ca, cb = t["colors"][:, :da].contiguous(), t["colors"][:, da:].contiguous()
    ra, a_a, _, g_a = _render(t, vm, K, w, h, colors=ca, v_out=v_out[..., :da].contiguous(), bg=bg[:, :da].contiguous())
    rb, a_b, _, g_b = _render(t, vm, K, w, h, colors=cb, v_out=v_out[..., da:].contiguous(), bg=bg[:, da:].contiguous())
    assert torch.equal(a_a, a_f) and torch.equal(a_b, a_f)
    cat = torch.cat([ra, rb], dim=-1)
    if da + db < 128:
        assert torch.equal(full, cat)  # below 128 channels every width runs the kernel that IS the sequential chain
    else:  # the joint width's first 128 channels take the 16-bit matrix cores on split operands: fp32-equivalent, not identical
        e = float((full.double() - cat.double()).norm() / cat.double().norm().clamp_min(1e-30))
        assert e <= FWD_SPLIT_TOL, e
    gcat = torch.cat([g_a, g_b], dim=-1).double()
    e = float((g_f.double() - gcat).norm() / gcat.norm().clamp_min(1e-30))
    assert e <= 2e-6, e


@settings(**SETTINGS)
@given(**scene_st)
def test_alpha_and_indices_do_not_depend_on_the_feature_width(n, w, h, seed, mult, view):
    ref = None
    for d in (3, 16, 40, 130):
        t, vm, K = _scene(n, d, w, h, seed, mult, view)   # (same geometry for every d: the generator draws it first)
        out, alphas, info, _ = _render(t, vm, K, w, h)
        cur = (alphas, info["last_ids"], info["flatten_ids"], info["isect_offsets"])
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(cur, ref)), d


@settings(**SETTINGS)
@given(d=st.sampled_from([16, 48, 128, 384]), k=st.sampled_from([-40, -3, 1, 30]), **scene_st)
def test_render_and_gradient_are_linear(n, w, h, seed, mult, view, d, k):
    t, vm, K = _scene(n, d, w, h, seed, mult, view)
    dev = t["means"].device
    gen = torch.Generator(device=dev).manual_seed(seed)
    v1, v2 = torch.randn(h, w, d, device=dev, generator=gen), torch.randn(h, w, d, device=dev, generator=gen)
    c2 = torch.randn(n, d, device=dev, generator=gen)
    r1, _, _, g1 = _render(t, vm, K, w, h, v_out=v1)
    # powers of two are exact at every stage: bf16 / fp16 terms are split off after power-of-two scalings taken from the data
    rs, _, _, gs = _render(t, vm, K, w, h, colors=t["colors"] * 2.0 ** k, v_out=v1 * 2.0 ** k)
    assert torch.equal(rs, r1 * 2.0 ** k)
    assert torch.equal(gs, g1 * 2.0 ** k)
    r2, _, _, g2 = _render(t, vm, K, w, h, colors=c2, v_out=v2)
    r12, _, _, _ = _render(t, vm, K, w, h, colors=t["colors"] + c2)
    _, _, _, g12 = _render(t, vm, K, w, h, v_out=v1 + v2)
    den = r12.double().norm().clamp_min(1e-30)
    assert float((r12.double() - (r1.double() + r2.double())).norm() / den) <= 1e-6
    # (the colour gradient does not depend on the colours: g(v1 + v2) = g(v1) + g(v2))
    den = g12.double().norm().clamp_min(1e-30)
    assert float((g12.double() - (g1.double() + g2.double())).norm() / den) <= 1e-6


@settings(**SETTINGS)
@given(d=st.sampled_from([16, 128]), **scene_st)
def test_culled_gaussians_receive_no_gradient(n, w, h, seed, mult, view, d):
    t, vm, K = _scene(n, d, w, h, seed, mult, view)
    dev = t["means"].device
    t["means"][::5, 2] = -1.0          # behind the camera
    t["means"][1::5, 0] += 1e4         # far off screen
    v_out = torch.randn(h, w, d, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
    _, _, info, grad = _render(t, vm, K, w, h, v_out=v_out)
    culled = info["radii"][0] == 0
    assert bool(culled[::5].all()) and bool(culled[1::5].all())
    assert not bool(grad[culled].any())
    assert np.isfinite(grad.cpu().numpy()).all()
