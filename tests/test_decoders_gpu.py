"""N1: the decoders against the reference modules' own outputs (tests/golden/make_golden_next.py runs
models/networks.py:109-248 on CPU in fp32 with the seeded weights of `decoder_weights`)."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
Z = np.load(os.path.join(HERE, "golden", "next_vectors.npz"))
sys.path.insert(0, os.path.join(HERE, "golden"))

# precision="exact" (the default: fp32 tensors, operands as three bf16 terms, fp32 accumulation) against the reference
# modules' fp32 results: what is left is the order of fp32 summation
XFWD_TOL = 1e-5
XBWD_TOL = 1e-3  # every gradient, rel-L2 (measured ~1e-6: see the test)
# precision="bf16" (fast opt-in): bf16 operands (8-bit mantissa) through 9 / 6 layers
FWD_TOL = 6e-3
BWD_TOL = 3e-2   # gradients: bf16 activations AND bf16 gradients through every layer


def _load(model, weights):
    with torch.no_grad():
        for m, (W, b) in zip(model.convs(), weights):
            m.weight.copy_(W[:, :, None, None])
            m.bias.copy_(b)
    return model.cuda()


def test_state_dict_names_match_the_reference_modules():
    from gags_amd.decoders import CNN_decoder, CNN_scale_decoder
    d, s = CNN_decoder(16, 512), CNN_scale_decoder(16, 3)
    assert [k for k in d.state_dict()] == [f"decoder.{2 * i}.{n}" for i in range(9) for n in ("weight", "bias")]
    assert d.state_dict()["decoder.16.weight"].shape == (512, 256, 1, 1)
    assert [tuple(v.shape) for k, v in s.state_dict().items() if k.endswith("weight")] == \
        [(64, 16, 1, 1), (128, 64, 1, 1), (64, 128, 1, 1), (32, 64, 1, 1), (16, 32, 1, 1), (3, 16, 1, 1)]


@pytest.mark.parametrize("precision", ["exact", "bf16"])
def test_decoder_forward_matches_reference(precision):
    from gags_amd.decoders import CNN_decoder, CNN_scale_decoder
    from make_golden_next import decoder_weights
    wd, ws = decoder_weights(0)
    assert CNN_decoder(16, 512).precision == "exact"  # the reference's precision is the default
    dec, sdec = _load(CNN_decoder(16, 512, precision), wd), _load(CNN_scale_decoder(16, 3, precision), ws)
    FWD_TOL = XFWD_TOL if precision == "exact" else globals()["FWD_TOL"]
    with torch.no_grad():
        y = dec(torch.from_numpy(Z["dec_x"]).cuda())
        ys = sdec(torch.from_numpy(Z["sdec_x"]).cuda())
    assert y.shape == Z["dec_y"].shape and ys.shape == Z["sdec_y"].shape
    assert rel_l2(y.cpu().numpy(), Z["dec_y"]) <= FWD_TOL
    assert rel_l2(ys.cpu().numpy(), Z["sdec_y"]) <= FWD_TOL
    np.testing.assert_allclose(y.double().pow(2).sum(0).sqrt().cpu().numpy(), 1.0, rtol=1e-5)  # unit norm per pixel
    np.testing.assert_allclose(ys.double().sum(0).cpu().numpy(), 1.0, rtol=1e-5)
    # the rasterizer's output layout ([H,W,C] memory viewed as [C,H,W]) is consumed without a transposing copy
    xp = torch.from_numpy(Z["dec_x"]).cuda().permute(1, 2, 0).contiguous().permute(2, 0, 1)
    with torch.no_grad():
        assert torch.equal(dec(xp), y)


@pytest.mark.parametrize("precision", ["exact", "bf16"])
def test_decoder_at_render_resolution_against_fp32_torch(precision):
    """1080p: the kernels against the same network evaluated by torch in fp32 (conv2d, TF32 off) on the GPU."""
    from gags_amd.decoders import CNN_decoder
    from make_golden_next import decoder_weights
    wd, _ = decoder_weights(0)
    dec = _load(CNN_decoder(16, 512, precision), wd)
    FWD_TOL = 2e-5 if precision == "exact" else globals()["FWD_TOL"]  # (torch's fp32 GEMM rounds too)
    g = torch.Generator(device="cuda").manual_seed(0)
    H, W = 1080, 1920
    x = torch.randn(H, W, 16, device="cuda", generator=g).permute(2, 0, 1)
    with torch.no_grad():
        y = dec(x)
        torch.backends.cudnn.allow_tf32 = False
        torch.backends.cuda.matmul.allow_tf32 = False
        xs = x[:, 500:532, 900:1028].contiguous()  # a 32 x 128 window is enough for the fp32 statement
        m = dec.decoder

        def conv(i, t):
            return torch.nn.functional.conv2d(t[None], m[i].weight, m[i].bias)[0]

        x1 = torch.relu(conv(0, xs))
        x2 = torch.relu(conv(4, torch.relu(conv(2, x1))))
        x3 = torch.relu(conv(6, x1 + x2))
        x4 = torch.relu(conv(10, torch.relu(conv(8, x3))))
        x5 = torch.relu(conv(14, torch.relu(conv(12, x3 + x4))))
        ref = torch.nn.functional.normalize(conv(16, x5), dim=0)
    got = y[:, 500:532, 900:1028]
    assert ((got - ref).double().norm() / ref.double().norm()).item() <= FWD_TOL


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


@pytest.mark.parametrize("precision,XFWD_TOL,XBWD_TOL", [("exact", XFWD_TOL, XBWD_TOL), ("bf16x2", 1e-4, 1e-3)])
def test_decoder_backward_matches_reference_gradients_at_the_reference_precision(precision, XFWD_TOL, XBWD_TOL):
    """precision="exact" (default) and "bf16x2" (two bf16 terms per operand: 16 significand bits, three matrix terms per
    product -- still 32x tighter than the TF32 arithmetic torch runs the reference's convs in; bounds asked by VERDICT r3
    item 4: outputs <= 1e-4, gradients <= 1e-3): input gradient (the rasterizer's cotangent in the real flow, train.py:159,174) and
    EVERY weight / bias gradient of both decoders against autograd through the reference modules in fp32
    (tests/golden/next_vectors.npz): rel-L2 <= 1e-3 each, forward <= 1e-5; and bit-reproducible (no atomics)."""
    from gags_amd.decoders import CNN_decoder, CNN_scale_decoder
    from make_golden_next import decoder_weights
    wd, ws = decoder_weights(0)
    worst = 0.0
    for model, weights, pre in ((CNN_decoder(16, 512, precision), wd, "dec"), (CNN_scale_decoder(16, 3, precision), ws, "sdec")):
        m = _load(model, weights)
        runs = []
        for _ in range(2):
            m.zero_grad(set_to_none=True)
            x = torch.from_numpy(Z[f"{pre}_x"]).cuda().requires_grad_(True)
            y = m(x)
            (y * torch.from_numpy(Z[f"{pre}_G"]).cuda()).sum().backward()
            runs.append([x.grad.clone()] + [t.grad.clone() for cv in m.convs() for t in (cv.weight, cv.bias)])
        assert all(torch.equal(a, b) for a, b in zip(*runs)), pre  # deterministic
        assert rel_l2(y.detach().cpu().numpy(), Z[f"{pre}_y"]) <= XFWD_TOL
        e = rel_l2(x.grad.cpu().numpy(), Z[f"{pre}_vx"])
        assert e <= XBWD_TOL, (pre, "input", e)
        worst = max(worst, e)
        for i, cv in enumerate(m.convs()):
            gw, gb = cv.weight.grad[:, :, 0, 0].cpu().numpy(), cv.bias.grad.cpu().numpy()
            ref = Z[f"{pre}_vw{i}"]
            e = rel_l2(gw[:ref.shape[0]], ref)
            assert e <= XBWD_TOL, (pre, "weight", i, e)
            worst = max(worst, e)
            if f"{pre}_vw{i}_norm" in Z.files:  # layers stored as their first rows + the norm of the whole matrix
                assert abs(np.linalg.norm(gw.astype(np.float64)) / float(Z[f"{pre}_vw{i}_norm"]) - 1.0) <= XBWD_TOL
            e = rel_l2(gb, Z[f"{pre}_vb{i}"])
            assert e <= XBWD_TOL, (pre, "bias", i, e)
            worst = max(worst, e)
    print(precision, "worst gradient rel-L2 vs the reference modules:", worst)


def test_decoder_backward_against_reference_gradients_bf16():
    """precision="bf16" (fast opt-in).  A low-precision forward flips the ReLU of the few units whose pre-activation is
    within rounding of zero (a fraction f ~ 0.5 % per layer with 8-bit mantissas; TF32, what the reference itself runs,
    flips ~0.1 %), and a flipped unit changes its gradient contribution entirely: rel-L2 ~ sqrt(f) per layer whatever
    the precision of the backward.  Hence direction (cosine) + a loose norm bound here -- this is the stated bound of the
    opt-in mode, not parity -- and the exact check of its backward kernels in the next test."""
    from gags_amd.decoders import CNN_decoder, CNN_scale_decoder
    from make_golden_next import decoder_weights
    wd, ws = decoder_weights(0)
    for model, weights, pre in ((CNN_decoder(16, 512, "bf16"), wd, "dec"), (CNN_scale_decoder(16, 3, "bf16"), ws, "sdec")):
        m = _load(model, weights)
        x = torch.from_numpy(Z[f"{pre}_x"]).cuda().requires_grad_(True)
        y = m(x)
        (y * torch.from_numpy(Z[f"{pre}_G"]).cuda()).sum().backward()
        assert _cos(x.grad.cpu().numpy(), Z[f"{pre}_vx"]) >= 0.985 and rel_l2(x.grad.cpu().numpy(), Z[f"{pre}_vx"]) <= 0.2, pre
        for i, cv in enumerate(m.convs()):
            gw, gb = cv.weight.grad[:, :, 0, 0].cpu().numpy(), cv.bias.grad.cpu().numpy()
            ref = Z[f"{pre}_vw{i}"]
            assert _cos(gw[:ref.shape[0]], ref) >= 0.985 and rel_l2(gw[:ref.shape[0]], ref) <= 0.2, (pre, i)
            assert _cos(gb, Z[f"{pre}_vb{i}"]) >= 0.98, (pre, i)
    # the last layer sits above every ReLU: its gradients only carry the rounding of the operands
    last = m.convs()[-1]
    assert rel_l2(last.weight.grad[:, :, 0, 0].cpu().numpy(), Z["sdec_vw5"]) <= 1.5e-2


def test_frozen_parameters_and_detached_inputs_cost_no_gradient_work():
    """needs_input_grad is honoured: the scale decoder is fed feature_map.detach() (train.py:149) -- no input
    gradient -- and frozen layers get no weight gradient; what is computed equals the full backward."""
    from gags_amd.decoders import CNN_scale_decoder
    from make_golden_next import decoder_weights
    _, ws = decoder_weights(0)
    for precision in ("exact", "bf16"):
        m = _load(CNN_scale_decoder(16, 3, precision), ws)
        x = torch.from_numpy(Z["sdec_x"]).cuda()
        G = torch.from_numpy(Z["sdec_G"]).cuda()
        xr = x.clone().requires_grad_(True)
        (m(xr) * G).sum().backward()
        full = [cv.weight.grad.clone() for cv in m.convs()]
        m.zero_grad(set_to_none=True)
        for q in (m.convs()[1].weight, m.convs()[1].bias):
            q.requires_grad_(False)
        (m(x) * G).sum().backward()          # detached input
        assert m.convs()[1].weight.grad is None and m.convs()[1].bias.grad is None
        for i, cv in enumerate(m.convs()):
            if i != 1:
                assert torch.equal(cv.weight.grad, full[i]), (precision, i)


def test_decoder_backward_kernels_against_fp32_on_the_same_masks():
    """The backward kernels proper: fp32 torch autograd through the SAME network with the weights rounded to bf16 and
    the ReLU decisions of the kernels' own forward (a straight-through mask), at 256 x 320 pixels.  What is left is
    the bf16 rounding of the activations and of the gradients between layers: <= 1.5e-2 rel-L2."""
    from gags_amd import decoders as D
    from make_golden_next import decoder_weights
    wd, _ = decoder_weights(0)
    dec = _load(D.CNN_decoder(16, 512, "bf16"), wd)
    g = torch.Generator(device="cuda").manual_seed(3)
    H, W = 256, 320
    x = torch.randn(H, W, 16, device="cuda", generator=g).permute(2, 0, 1).requires_grad_(True)
    G = torch.randn(512, H, W, device="cuda", generator=g)
    y = dec(x)
    (y * G).sum().backward()
    got = {"x": x.grad.clone(), **{f"w{i}": c.weight.grad[:, :, 0, 0].clone() for i, c in enumerate(dec.convs())},
           **{f"b{i}": c.bias.grad.clone() for i, c in enumerate(dec.convs())}}
    # fp32 statement with the kernels' masks
    wb = D._pack_weights([c.weight for c in dec.convs()], [c.bias for c in dec.convs()])
    p = H * W
    xp = x.detach().permute(1, 2, 0).reshape(p, 16)
    a0 = torch.zeros(p, 32, device="cuda"); a0[:, :16] = xp.to(torch.bfloat16).float()
    Wf = [w.float().requires_grad_(True) for w, _ in wb]
    Bf = [b.clone().requires_grad_(True) for _, b in wb]
    a0.requires_grad_(True)

    # the kernels' own chain of activations: its ReLU decisions are the masks of the fp32 statement
    with torch.no_grad():
        k0 = a0.detach().to(torch.bfloat16)
        k1 = D._layer(p, *wb[0], k0); kt1 = D._layer(p, *wb[1], k1); k2 = D._layer(p, *wb[2], kt1)
        k3 = D._layer(p, *wb[3], k1, k2); kt4 = D._layer(p, *wb[4], k3); k4 = D._layer(p, *wb[5], kt4)
        kt6 = D._layer(p, *wb[6], k3, k4); kt7 = D._layer(p, *wb[7], kt6)

    def lay(i, kern, a, a2=None):
        inp = a if a2 is None else a + a2
        return (inp @ Wf[i].t() + Bf[i]) * (kern.float() > 0)

    x1 = lay(0, k1, a0); t1 = lay(1, kt1, x1); x2 = lay(2, k2, t1); x3 = lay(3, k3, x1, x2); t4 = lay(4, kt4, x3)
    x4 = lay(5, k4, t4); t6 = lay(6, kt6, x3, x4); t7 = lay(7, kt7, t6)
    logits = t7 @ Wf[8].t() + Bf[8]
    ref_y = torch.nn.functional.normalize(logits, dim=1)
    (ref_y * G.reshape(512, p).t()).sum().backward()
    assert ((y.reshape(512, p).t() - ref_y).double().norm() / ref_y.double().norm()).item() <= 1.5e-2
    errs = {"x": ((got["x"].permute(1, 2, 0).reshape(p, 16) - a0.grad[:, :16]).double().norm() / a0.grad[:, :16].double().norm()).item()}
    for i, c in enumerate(dec.convs()):
        co, ci = c.weight.shape[:2]
        errs[f"w{i}"] = ((got[f"w{i}"] - Wf[i].grad[:co, :ci]).double().norm() / Wf[i].grad[:co, :ci].double().norm()).item()
        errs[f"b{i}"] = ((got[f"b{i}"] - Bf[i].grad[:co]).double().norm() / Bf[i].grad[:co].double().norm()).item()
    assert max(errs.values()) <= 1.5e-2, errs


@pytest.mark.parametrize("c,ld,layout", [(512, 512, 0), (3, 32, 0), (100, 128, 0), (37, 40, 0), (520, 544, 0),
                                         (512, 512, 1), (100, 128, 1), (16, 32, 1), (4, 8, 0), (2, 16, 0)])
@pytest.mark.parametrize("mode", [0, 1])
def test_head_kernels_against_torch(c, ld, layout, mode):
    """gags_decoder_head / _head_bwd alone (models/networks.py:192 normalize, :242 softmax) in fp32 against torch, on
    the kernels behind the entry: the 32-pixel register-resident one (ld <= 512, ld % 32 == 0), the general one, and
    the pixel-major ones (layout 1: output and cotangent as [P, C] rows, nothing transposed)."""
    from gags_amd import _lib
    from gags_amd.decoders import _st
    from gags_amd._lib import check, ptr
    lib = _lib.load()
    p = 1000 + 13
    g = torch.Generator(device="cuda").manual_seed(c + mode)
    x = torch.zeros(p, ld, device="cuda")
    x[:, :c] = torch.randn(p, c, device="cuda", generator=g) * 2
    G = torch.randn(c, p, device="cuda", generator=g)
    out = torch.empty((p, c) if layout else (c, p), device="cuda")
    Gk = G.t().contiguous() if layout else G
    dz = torch.full((p, ld), 7.0, device="cuda", dtype=torch.bfloat16)
    check(lib.gags_decoder_head(p, c, ld, mode, ptr(x), ptr(out), layout, _st()), "head")
    check(lib.gags_decoder_head_bwd(p, c, ld, mode, ptr(x), ptr(Gk), ptr(dz), layout, _st()), "head_bwd")
    if layout:
        out = out.t()
    xr = x[:, :c].clone().double().requires_grad_(True)
    ref = torch.nn.functional.normalize(xr, dim=1) if mode == 0 else torch.softmax(xr, dim=1)
    (ref * G.t().double()).sum().backward()
    assert ((out.t().double() - ref).norm() / ref.norm()).item() <= 1e-6
    assert ((dz[:, :c].double() - xr.grad).norm() / xr.grad.norm()).item() <= 4e-3   # bf16 output
    assert torch.count_nonzero(dz[:, c:]).item() == 0


@pytest.mark.parametrize("n,k,two", [(256, 256, False), (256, 256, True), (512, 256, False), (128, 256, True),
                                     (256, 32, False), (64, 128, True), (32, 64, False), (128, 64, False), (64, 32, True),
                                     (32, 32, False), (48, 32, False), (512, 64, False)])
def test_weight_gradient_kernels_against_torch(n, k, two):
    """gags_decoder_wgrad alone: dW = dz^T (a1 + a2), db = sum_p dz in fp32 on the bf16 operands, on both kernels
    behind the entry (the transposing-LDS-read one for 256 inputs, the general one), with a pixel count that is not a
    multiple of anything (the last chunk is ragged) and with random, asymmetric operands."""
    from gags_amd import decoders as D
    p = 70000 + 77
    g = torch.Generator(device="cuda").manual_seed(n + k)
    dz = torch.randn(p, n, device="cuda", generator=g).to(torch.bfloat16)
    a1 = torch.randn(p, k, device="cuda", generator=g).to(torch.bfloat16)
    a2 = torch.randn(p, k, device="cuda", generator=g).to(torch.bfloat16) if two else None
    dw, db = D._wgrad(p, dz, a1, a2, n, k)
    a = a1.float() if a2 is None else (a1.float() + a2.float()).to(torch.bfloat16).float()  # the sum is rounded once
    ref_w = dz.double().t() @ a.double()
    ref_b = dz.double().sum(0)
    assert ((dw.double() - ref_w).norm() / ref_w.norm()).item() <= 2e-6
    assert ((db.double() - ref_b).norm() / ref_b.norm()).item() <= 2e-6


@pytest.mark.parametrize("precision,tol", [("exact", 1e-5), ("bf16", 2e-2)])
def test_decoder_output_is_pixel_major_and_its_gradient_flows_without_transposes(precision, tol):
    """CNN_decoder returns [C,H,W] as a permuted view of [H,W,C] memory (like render()); the fused distillation loss
    consumes that memory directly and hands its gradient back in the same layout.  Values and gradients equal those of
    the channel-major (contiguous) route."""
    from gags_amd import losses as L
    from gags_amd.decoders import CNN_decoder
    from make_golden_next import decoder_weights
    wd, _ = decoder_weights(0)
    dec = _load(CNN_decoder(16, 512, precision), wd)
    g = torch.Generator(device="cuda").manual_seed(9)
    H, W, n_emb = 96, 130, 40
    x = torch.randn(H, W, 16, device="cuda", generator=g).permute(2, 0, 1)
    emb = torch.nn.functional.normalize(torch.randn(n_emb, 512, device="cuda", generator=g), dim=-1)
    seg = torch.randint(-1, n_emb, (4, H, W), device="cuda", generator=g).float()
    scale = torch.softmax(torch.randn(3, H, W, device="cuda", generator=g), 0)
    res = []
    for contiguous in (False, True):
        xi = x.clone().requires_grad_(True)
        dec.zero_grad(set_to_none=True)
        y = dec(xi)
        assert y.shape == (512, H, W) and not y.is_contiguous() and y.permute(1, 2, 0).is_contiguous()
        yy = y.contiguous() if contiguous else y
        l1, mask = L.distill_l1_map(yy, emb, seg, scale)
        (l1 * torch.linspace(0.5, 1.5, H * W, device="cuda").reshape(H, W)).sum().backward()
        res.append((l1.detach().clone(), xi.grad.clone(), dec.convs()[-1].weight.grad.clone()))
    (l1_a, gx_a, gw_a), (l1_b, gx_b, gw_b) = res
    assert ((l1_a - l1_b).double().norm() / l1_b.double().norm()).item() <= 1e-6
    assert ((gx_a - gx_b).double().norm() / gx_b.double().norm()).item() <= tol   # (bf16: rounded layers behind a different summation order)
    assert ((gw_a - gw_b).double().norm() / gw_b.double().norm()).item() <= tol


@pytest.mark.parametrize("precision,gtol", [("bf16", 2e-2), ("exact", 5e-3), ("bf16x2", 5e-3), ("f16", 5e-3)])
@pytest.mark.parametrize("H,W,h,w", [(96, 130, 96, 130), (60, 77, 30, 40)])
def test_fused_head_and_distillation_loss_equals_the_two_step_route(H, W, h, w, precision, gtol):
    """CNN_decoder.distill_l1 (head fused into the loss: gags_decoder_head_distill_fwd / _bwd) against
    distill_l1_map(decoder(x), ...): the loss map, the mask, and every gradient (input, all decoder parameters, scale
    map), with an identity-size and a resized (bilinear, four taps) segmentation map."""
    from gags_amd import losses as L
    from gags_amd.decoders import CNN_decoder
    from make_golden_next import decoder_weights
    wd, _ = decoder_weights(0)
    # every tier has the fused head + loss (the fp32-tensor tiers keep the logits' gradient in fp32).  Gradient bounds: the
    # bf16 mode rounds the logits' gradient to bf16; in the fp32 tiers the two routes differ by an ulp in y = x / |x|, and the
    # L1's sign(y m - gt m) flips for the one or two of the ~6 M differences that are that close to zero (each flip: 8e-4)
    dec = _load(CNN_decoder(16, 512, precision), wd)
    g = torch.Generator(device="cuda").manual_seed(12)
    n_emb = 40
    x = torch.randn(H, W, 16, device="cuda", generator=g).permute(2, 0, 1)
    emb = torch.nn.functional.normalize(torch.randn(n_emb, 512, device="cuda", generator=g), dim=-1)
    seg = torch.randint(-1, n_emb, (4, h, w), device="cuda", generator=g).float()
    scale0 = torch.softmax(torch.randn(3, H, W, device="cuda", generator=g), 0)
    wgt = torch.linspace(0.5, 1.5, H * W, device="cuda").reshape(H, W)
    res = []
    for fused in (False, True):
        xi = x.clone().requires_grad_(True)
        sc = scale0.clone().requires_grad_(True)
        dec.zero_grad(set_to_none=True)
        if fused:
            l1, mask = dec.distill_l1(xi, emb, seg, sc)
        else:
            l1, mask = L.distill_l1_map(dec(xi), emb, seg, sc)
        (l1 * wgt).sum().backward()
        res.append((l1.detach().clone(), mask.clone(), xi.grad.clone(), sc.grad.clone(),
                    [c.weight.grad.clone() for c in dec.convs()], [c.bias.grad.clone() for c in dec.convs()]))
    a, b = res

    def rel(u, v):
        return ((u - v).double().norm() / v.double().norm().clamp_min(1e-300)).item()

    assert torch.equal(a[1], b[1]) and a[1].shape == (1, H, W)
    assert rel(b[0], a[0]) <= 1e-6
    assert rel(b[3], a[3]) <= 1e-5                       # scale-map gradient: fp32 on both routes
    assert rel(b[2], a[2]) <= gtol
    for u, v in zip(b[4] + b[5], a[4] + a[5]):
        assert rel(u, v) <= gtol


@pytest.mark.parametrize("n,k,two,p", [(256, 256, True, 20011), (512, 256, False, 9001), (3, 16, False, 5003), (16, 3, False, 5003),
                                       (64, 16, False, 777), (130, 70, True, 4099)])
def test_exact_layer_and_weight_gradient_kernels_against_float64(n, k, two, p):
    """gags_decoder_layer_exact / gags_decoder_wgrad_exact alone against float64 torch, every epilogue option, ragged
    sizes (the scale decoder's 3-wide head, pixel counts that are multiples of nothing): <= 5e-7 rel-L2 -- fp32 GEMM
    accuracy from bf16 matrix instructions; the weight gradient twice, bit-identical (no atomics)."""
    from gags_amd import decoders as D
    g = torch.Generator(device="cuda").manual_seed(n * 7 + k)
    a1 = torch.randn(p, k, device="cuda", generator=g) * torch.exp(torch.randn(p, 1, device="cuda", generator=g))
    a2 = torch.randn(p, k, device="cuda", generator=g) if two else None
    w = torch.randn(n, k, device="cuda", generator=g) / k ** 0.5
    b = torch.randn(n, device="cuda", generator=g)
    mask = torch.randn(p, n, device="cuda", generator=g)
    res = torch.randn(p, n, device="cuda", generator=g)
    y, ypre = D._xlayer(p, w, b, a1, a2, relu=True, mask_src=mask, residual=res, premask=True)
    a = a1.double() if a2 is None else (a1 + a2).double()   # the kernel adds the two sources in fp32, as torch does
    ref_pre = torch.relu(a @ w.double().t() + b.double()) + res.double()
    ref = ref_pre * (mask > 0)
    assert ((ypre.double() - ref_pre).norm() / ref_pre.norm()).item() <= 5e-7
    assert ((y.double() - ref).norm() / ref.norm()).item() <= 5e-7
    y2 = D._xlayer(p, w, None, a1, None, relu=False, ldy=n + 5)
    assert ((y2[:, :n].double() - a1.double() @ w.double().t()).norm() / (a1.double() @ w.double().t()).norm()).item() <= 5e-7
    assert torch.count_nonzero(y2[:, n:]).item() == 0
    dz = torch.randn(p, n, device="cuda", generator=g)
    dw, db = D._xwgrad(p, dz, a1, a2, n, k)
    dw2, db2 = D._xwgrad(p, dz, a1, a2, n, k)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    ref_w, ref_b = dz.double().t() @ a, dz.double().sum(0)
    assert ((dw.double() - ref_w).norm() / ref_w.norm()).item() <= 5e-7
    assert ((db.double() - ref_b).norm() / ref_b.norm()).item() <= 1e-6


@pytest.mark.parametrize("c,ldx,layout,mode", [(512, 512, 1, 0), (512, 512, 0, 0), (3, 8, 0, 1), (100, 104, 0, 1), (37, 40, 1, 0)])
def test_exact_head_backward_against_float64(c, ldx, layout, mode):
    from gags_amd import _lib
    from gags_amd.decoders import _st
    from gags_amd._lib import check, ptr
    p = 1000 + 13
    g = torch.Generator(device="cuda").manual_seed(c + mode)
    x = torch.zeros(p, ldx, device="cuda")
    x[:, :c] = torch.randn(p, c, device="cuda", generator=g) * 2
    G = torch.randn(c, p, device="cuda", generator=g)
    Gk = G.t().contiguous() if layout else G
    dz = torch.full((p, c), 7.0, device="cuda")
    check(_lib.load().gags_decoder_head_bwd_exact(p, c, ldx, mode, ptr(x), ptr(Gk), layout, ptr(dz), c, _st()), "head_bwd_exact")
    xr = x[:, :c].clone().double().requires_grad_(True)
    ref = torch.nn.functional.normalize(xr, dim=1) if mode == 0 else torch.softmax(xr, dim=1)
    (ref * G.t().double()).sum().backward()
    assert ((dz.double() - xr.grad).norm() / xr.grad.norm()).item() <= 2e-6


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_fused_decoder_kernels_are_bit_identical_to_the_layer_by_layer_chain(precision):
    """bf16 / f16 modes: CNN_decoder's forward as ONE kernel (activations resident in LDS, csrc/decoder_fused.hip) and its input-
    gradient chain as one kernel against the same chain run layer by layer through gags_decoder_layer: the arithmetic is
    the same (bf16 operands, fp32 accumulation in ascending k, one rounding per activation), so every output, every kept
    activation and every gradient must be IDENTICAL; ragged pixel count (not a multiple of the 64-pixel tile)."""
    from gags_amd import decoders as D
    from make_golden_next import decoder_weights
    wd, _ = decoder_weights(0)
    dec = _load(D.CNN_decoder(16, 512, precision), wd)
    mode = D._F16 if precision == "f16" else D._BF16
    g = torch.Generator(device="cuda").manual_seed(21)
    H, W = 67, 93
    x = torch.randn(H, W, 16, device="cuda", generator=g).permute(2, 0, 1)
    G = torch.randn(512, H, W, device="cuda", generator=g)
    res = []
    for fused in (True, False):
        D.FUSED = fused
        try:
            xi = x.clone().requires_grad_(True)
            dec.zero_grad(set_to_none=True)
            params = [t for m in dec.convs() for t in (m.weight, m.bias)]
            logits, acts, wb, h, w, c_in = D._chain_forward(xi.detach(), "decoder", params, mode)
            y = dec(xi)
            (y * G).sum().backward()
            res.append((logits.clone(), [a.clone() for a in acts], y.detach().clone(), xi.grad.clone(),
                        [c.weight.grad.clone() for c in dec.convs()], [c.bias.grad.clone() for c in dec.convs()]))
        finally:
            D.FUSED = True
    a, b = res
    assert torch.equal(a[0], b[0]), "logits"
    for i, (u, v) in enumerate(zip(a[1], b[1])):
        if i in (3, 6):  # the fused forward keeps the residual sums x1 + x2 / x3 + x4 (what layers 3 / 6 read) in place of x2 / x4
            v = (b[1][i - 2].float() + v.float()).to(mode.dtype)
        assert torch.equal(u, v), f"activation {i}"
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    for u, v in zip(a[4] + a[5], b[4] + b[5]):
        assert torch.equal(u, v)


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_fused_scale_decoder_kernels_are_bit_identical_to_the_layer_by_layer_chain(precision):
    """bf16 / f16 modes: CNN_scale_decoder's six layers as ONE kernel, a wave per 32-pixel tile (csrc/decoder_scale.hip), against
    the same chain run layer by layer: logits, every kept activation, the softmax output, the input gradient and every
    weight / bias gradient IDENTICAL; ragged pixel count (not a multiple of the tile), and the ReLU bit masks the fused
    forward leaves behind equal [activation > 0]."""
    from gags_amd import decoders as D
    from make_golden_next import decoder_weights
    _, ws = decoder_weights(0)
    sdec = _load(D.CNN_scale_decoder(16, 3, precision), ws)
    mode = D._F16 if precision == "f16" else D._BF16
    g = torch.Generator(device="cuda").manual_seed(22)
    H, W = 61, 97
    x = torch.randn(H, W, 16, device="cuda", generator=g).permute(2, 0, 1)
    G = torch.randn(3, H, W, device="cuda", generator=g)
    res = []
    for fused in (True, False):
        D.FUSED = fused
        try:
            xi = x.clone().requires_grad_(True)
            sdec.zero_grad(set_to_none=True)
            params = [t for m in sdec.convs() for t in (m.weight, m.bias)]
            logits, acts, wb, h, w, c_in = D._chain_forward(xi.detach(), "scale", params, mode)
            y = sdec(xi)
            (y * G).sum().backward()
            res.append((logits.clone(), [a.clone() for a in acts], y.detach().clone(), xi.grad.clone(),
                        [c.weight.grad.clone() for c in sdec.convs()], [c.bias.grad.clone() for c in sdec.convs()]))
        finally:
            D.FUSED = True
    a, b = res
    assert len(a[1]) == 7 and len(b[1]) == 6                       # the fused forward also keeps the masks
    assert torch.equal(a[0][:, :3], b[0][:, :3]), "logits"
    for i, (u, v) in enumerate(zip(a[1][:6], b[1])):
        assert torch.equal(u, v), f"activation {i}"
    masks = a[1][6]
    off = 0
    for act in b[1][1:]:                                           # words of a1 .. a5: bit n % 32 of word n / 32 = [a > 0]
        n = act.shape[1]
        bits = (act.float() > 0).view(-1, n // 32, 32).long()
        want = (bits << torch.arange(32, device="cuda")).sum(-1)
        got = masks[:, off:off + n // 32].long() & 0xffffffff
        assert torch.equal(got, want), f"mask words at {off}"
        off += n // 32
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    for u, v in zip(a[4] + a[5], b[4] + b[5]):
        assert torch.equal(u, v)


def test_packed_weights_follow_in_place_updates():
    """The packed (padded bf16 / transposed / fragment-order) weights are cached on the parameters' version counters: an
    in-place update under no_grad (what every optimizer does) repacks them, an unchanged module does not launch the pack
    kernel again, and the packed forms equal torch's own cast / transpose / permute."""
    from gags_amd import decoders as D
    from make_golden_next import decoder_weights
    _, ws = decoder_weights(0)
    sdec = _load(D.CNN_scale_decoder(16, 3, "bf16"), ws)
    x = torch.randn(16, 40, 50, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    params = [t for m in sdec.convs() for t in (m.weight, m.bias)]
    wb1 = D._pack_weights(params[0::2], params[1::2])
    assert D._pack_weights(params[0::2], params[1::2]) is wb1                       # cached
    for (w, b), conv in zip(wb1, sdec.convs()):
        co, ci = conv.weight.shape[:2]
        ref = torch.zeros_like(w)
        ref[:co, :ci] = conv.weight.detach()[:, :, 0, 0].to(torch.bfloat16)
        assert torch.equal(w, ref) and torch.equal(D._transposed(w), ref.t().contiguous())
        n, k = ref.shape
        assert torch.equal(D._frag_layout(w), ref.view(n // 32, 32, k // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous())
        rt = ref.t().contiguous()
        assert torch.equal(D._frag_layout(D._transposed(w)), rt.view(k // 32, 32, n // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous())
        assert torch.equal(b[:co], conv.bias.detach()) and float(b[co:].abs().sum()) == 0.0
    y1 = sdec(x).clone()
    with torch.no_grad():
        sdec.convs()[2].weight.mul_(1.5)
    wb2 = D._pack_weights(params[0::2], params[1::2])
    assert wb2 is not wb1
    y2 = sdec(x)
    assert not torch.equal(y1, y2)
    D.invalidate_packed()
    assert torch.equal(sdec(x), y2)


# ---------------------------------------------------------------- precision="f16": IEEE-half operands (TF32's significand)
def _tf32(t):
    """Round an fp32 tensor to TF32's 10 explicit significand bits (nearest even): what the reference's convs do to both
    operands of every product on the GPU its README names (torch.backends.cudnn.allow_tf32 defaults to True)."""
    i = t.contiguous().view(torch.int32)
    i = (i + 0x0FFF + ((i >> 13) & 1)) & ~0x1FFF
    return i.view(torch.float32)


class _RoundedLinear(torch.autograd.Function):
    """a @ W^T with `rnd` applied to both operands of every product, forward AND backward (cuDNN's TF32 convolutions round
    the operands of the data- and weight-gradient convolutions as well)."""

    @staticmethod
    def forward(ctx, a, W, rnd):
        ctx.rnd = rnd
        ctx.save_for_backward(a, W)
        return rnd(a) @ rnd(W).t()

    @staticmethod
    def backward(ctx, g):
        a, W = ctx.saved_tensors
        r = ctx.rnd
        return r(g) @ r(W), r(g).t() @ r(a), None


def _torch_chain(x, weights, kind, rnd):
    """The reference modules' arithmetic (models/networks.py:189-218, 236-248) as torch matmuls with `rnd` applied to both
    operands of every product (identity: fp32; _tf32: the reference's own convs on its GPU).  x [P, C] -> output [P, C_out]."""
    def lin(a, i):
        W, b = weights[i]
        return _RoundedLinear.apply(a, W, rnd) + b
    if kind == "decoder":
        x1 = torch.relu(lin(x, 0)); x2 = torch.relu(lin(torch.relu(lin(x1, 1)), 2))
        x3 = torch.relu(lin(x1 + x2, 3)); x4 = torch.relu(lin(torch.relu(lin(x3, 4)), 5))
        t = torch.relu(lin(x3 + x4, 6)); t = torch.relu(lin(t, 7))
        return torch.nn.functional.normalize(lin(t, 8), dim=1)
    a = x
    for i in range(6):
        a = lin(a, i)
        if i < 5:
            a = torch.relu(a)
    return torch.softmax(a, dim=1)


def test_f16_tier_is_as_close_to_fp32_as_the_tf32_arithmetic_the_reference_runs():
    """precision="f16": operands in IEEE half (11-bit significand = TF32's), fp32 accumulation.  Stated bound: outputs and
    every gradient are as close to the reference modules' fp32 results (tests/golden/next_vectors.npz) as the SAME chain run
    with TF32-rounded operands -- the arithmetic the reference's nn.Conv2d layers use on its own GPU -- within a factor 2
    (the tiers round at different places: half also rounds the stored activations, TF32 rounds them on use; same bits)."""
    from gags_amd.decoders import CNN_decoder, CNN_scale_decoder
    from make_golden_next import decoder_weights
    wd, ws = decoder_weights(0)
    gen = torch.Generator(device="cuda").manual_seed(31)
    for model, weights, pre, kind, c_out in ((CNN_decoder(16, 512, "f16"), wd, "dec", "decoder", 512),
                                             (CNN_scale_decoder(16, 3, "f16"), ws, "sdec", "scale", 3)):
        m = _load(model, weights)
        with torch.no_grad():  # the reference fixture first (120 pixels)
            assert rel_l2(m(torch.from_numpy(Z[f"{pre}_x"]).cuda()).cpu().numpy(), Z[f"{pre}_y"]) <= 1e-3
        # a larger seeded input of the fixture's magnitude, so that the comparison of two roundings is not a 120-pixel accident
        h, w = 48, 64
        x0 = torch.randn(h, w, 16, device="cuda", generator=gen).permute(2, 0, 1) * float(np.abs(Z[f"{pre}_x"]).std())
        G = torch.randn(c_out, h, w, device="cuda", generator=gen)
        x = x0.clone().requires_grad_(True)
        y = m(x)
        (y * G).sum().backward()
        got = [y.detach(), x.grad] + [t.grad[:, :, 0, 0] if t.dim() == 4 else t.grad for cv in m.convs() for t in (cv.weight, cv.bias)]
        res = {}
        for name, rnd in (("fp32", lambda t: t), ("tf32", _tf32)):
            wt = [(W.clone().cuda().requires_grad_(True), b.clone().cuda().requires_grad_(True)) for W, b in weights]
            xr = x0.clone().requires_grad_(True)
            yr = _torch_chain(xr.reshape(16, -1).t(), wt, kind, rnd).t().reshape(-1, h, w)
            (yr * G).sum().backward()
            res[name] = [yr.detach(), xr.grad] + [t.grad for pair in wt for t in pair]
        worst = (0.0, 0.0)
        for i, (a, t32, f32) in enumerate(zip(got, res["tf32"], res["fp32"])):
            ref = f32.double()
            e_f16 = ((a.double() - ref).norm() / ref.norm()).item()
            e_tf32 = ((t32.double() - ref).norm() / ref.norm()).item()
            assert e_f16 <= 2.0 * e_tf32 + 1e-6, (pre, i, e_f16, e_tf32)
            worst = max(worst, (e_f16, e_tf32))
        print(pre, "f16 tier vs fp32: worst rel-L2 %.3e (TF32 emulation at the same tensor: %.3e)" % worst)


def test_f16_gradient_scaling_is_exact_in_powers_of_two():
    """The f16 tier multiplies the cotangent by a power of two chosen on the device from its magnitude and divides the results
    by it: a cotangent scaled by 2^k must give every gradient scaled by exactly 2^k (bit-identical significands), from
    magnitudes half could not hold unscaled (2^-40: flushed; 2^24: overflow) -- and nothing non-finite ever."""
    from gags_amd.decoders import CNN_decoder
    from make_golden_next import decoder_weights
    wd, _ = decoder_weights(0)
    dec = _load(CNN_decoder(16, 512, "f16"), wd)
    x0 = torch.from_numpy(Z["dec_x"]).cuda()
    G = torch.from_numpy(Z["dec_G"]).cuda()
    out = []
    for k in (0, -40, 24):
        dec.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        (dec(x) * (G * 2.0 ** k)).sum().backward()
        out.append([x.grad.clone()] + [t.grad.clone() for cv in dec.convs() for t in (cv.weight, cv.bias)])
    for k, res in zip((-40, 24), out[1:]):
        for a, b in zip(out[0], res):
            assert torch.isfinite(b).all()
            assert torch.equal(a * 2.0 ** k, b), k
    assert all(float(t.abs().max()) > 0 for t in out[0])
    # zero cotangent: scale 1, zero gradients, no NaN
    dec.zero_grad(set_to_none=True)
    x = x0.clone().requires_grad_(True)
    (dec(x) * 0.0).sum().backward()
    assert float(x.grad.abs().max()) == 0.0 and all(float(cv.weight.grad.abs().max()) == 0.0 for cv in dec.convs())


def test_f16_conversions_saturate_instead_of_overflowing():
    """Inputs far outside half's range (|x| up to 1e6) saturate at +-65504 in the packing kernel: the output stays finite
    and unit-norm, the gradients finite."""
    from gags_amd.decoders import CNN_decoder
    from make_golden_next import decoder_weights
    wd, _ = decoder_weights(0)
    dec = _load(CNN_decoder(16, 512, "f16"), wd)
    x = (torch.from_numpy(Z["dec_x"]).cuda() * 1e6).requires_grad_(True)
    y = dec(x)
    (y * torch.from_numpy(Z["dec_G"]).cuda()).sum().backward()
    assert torch.isfinite(y).all() and torch.isfinite(x.grad).all()
    np.testing.assert_allclose(y.detach().double().pow(2).sum(0).sqrt().cpu().numpy(), 1.0, rtol=1e-4)


def test_head_and_distillation_backward_treats_exact_ties_as_torch_sign_does():
    """torch.sign(0) = 0: an element whose normalised prediction EQUALS the ground truth contributes nothing.  The fused head +
    loss backward takes copysign(1, diff) on its fast path (round 6) and must notice exact ties (about one fp32 element in 10^7
    on real data) and redo the pixel in the {-1, 0, +1} form: pixels built to tie in all 512 channels (logits = an embedding of
    norm exactly 1, scale map (1, 0, 0)), in all but one, and ordinary pixels, against fp32 autograd through torch.sign."""
    import ctypes
    from gags_amd import _lib, losses as L
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(21)
    H, W, n_emb, c = 16, 24, 7, 512
    emb = torch.nn.functional.normalize(torch.randn(n_emb, c, device="cuda", generator=g), dim=-1)
    emb[0] = 0.0
    emb[0, :4] = 0.5                                   # |e0| = 1 exactly
    seg = torch.randint(0, n_emb, (4, H, W), device="cuda", generator=g).float()
    sc = torch.softmax(torch.randn(3, H, W, device="cuda", generator=g), 0)
    x = torch.randn(H * W, c, device="cuda", generator=g)
    tie_all, tie_most = [5, 40, 41, 200], [9, 77, 300]
    for p in tie_all + tie_most:
        seg[1].view(-1)[p] = 0
        sc.view(3, -1)[:, p] = torch.tensor([1.0, 0.0, 0.0], device="cuda")
        x[p] = emb[0] * (2.0 if p % 2 else 1.0)        # y = x / |x| = e0 exactly
    for p in tie_most:
        x[p, 3] = -x[p, 3]                             # one channel off, the 511 others tie
    v = torch.rand(H, W, device="cuda", generator=g) + 0.5
    # fp32 autograd through torch.sign
    xr, scr = x.clone().requires_grad_(True), sc.clone().requires_grad_(True)
    y = xr / xr.norm(dim=1, keepdim=True).clamp_min(1e-12)
    feat, mask = L.read_sam_clip_feature(emb, seg, scr)
    m = mask.float()
    l1 = (y.t().reshape(c, H, W) * m - feat * m).abs().mean(0)
    (l1 * v).sum().backward()
    assert float(xr.grad[tie_all].abs().max()) == 0.0  # (the reference: nothing flows through a pixel that ties everywhere)
    dz = torch.empty(H * W, c, device="cuda")
    vs = torch.empty(3, H, W, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    assert lib.gags_decoder_head_distill_bwd_f32(c, c, H, W, H, W, n_emb, P(x), P(emb), P(seg), P(sc), P(v), P(dz), P(vs), st) == 0
    torch.cuda.synchronize()
    assert float(dz[tie_all].abs().max()) == 0.0 and float(vs.view(3, -1)[:, tie_all].abs().max()) == 0.0
    # (elsewhere the two routes differ by an ulp in y = x / |x|: a sign may flip where |diff| is below that, 1 / 512 of a pixel)
    assert rel_l2(dz.cpu().numpy(), xr.grad.cpu().numpy()) <= 2e-3
    assert rel_l2(vs.cpu().numpy(), scr.grad.cpu().numpy()) <= 2e-3
    one = tie_most[0]  # 511 ties and one channel that differs: only that channel carries a sign
    np.testing.assert_allclose(dz[one].cpu().numpy(), xr.grad[one].cpu().numpy(), rtol=1e-4, atol=1e-9)
