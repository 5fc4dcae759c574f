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

# bf16 operands (8-bit mantissa) through 9 / 6 layers against the fp32 reference
FWD_TOL = 6e-3


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


def test_decoder_forward_matches_reference():
    from gags_amd.decoders import CNN_decoder, CNN_scale_decoder
    from make_golden_next import decoder_weights
    wd, ws = decoder_weights(0)
    dec, sdec = _load(CNN_decoder(16, 512), wd), _load(CNN_scale_decoder(16, 3), ws)
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


def test_decoder_at_render_resolution_against_fp32_torch():
    """1080p: the bf16 kernels against the same network evaluated by torch in fp32 (conv2d, TF32 off) on the GPU."""
    from gags_amd.decoders import CNN_decoder
    from make_golden_next import decoder_weights
    wd, _ = decoder_weights(0)
    dec = _load(CNN_decoder(16, 512), wd)
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
