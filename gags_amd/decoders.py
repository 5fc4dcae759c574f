"""The per-pixel decoders of the reference, same class names, constructor arguments, parameter
names (`decoder.<i>.weight` / `.bias`: a reference state_dict loads unchanged) and outputs as
models/networks.py:109-218 `CNN_decoder` and :220-248 `CNN_scale_decoder`, computed by hand-written bf16 matrix-core
GEMM kernels (include/gags_next.h N1, csrc/decoder.hip) instead of cuDNN 1x1 convolutions.

    CNN_decoder(16, 512):       x [16,H,W] -> 9 x (1x1 conv, 256 hidden, ReLU), x3 = x1 + x2, x5 = x3 + x4 -> F.normalize(dim=0)
    CNN_scale_decoder(16, 3):   x [16,H,W] -> 16-64-128-64-32-16-3 (ReLU between) -> softmax(dim=0)

Input: [C,H,W]; when it is the rasterizer's output (a permuted view of [H,W,C] memory, gaussian_renderer.py) the
pixel-major layout the kernels want is already there and nothing is transposed.  Output: [C_out,H,W] fp32, contiguous.

Precision: bf16 operands, fp32 accumulation, activations kept in bf16 between layers.  The reference's cuDNN path uses
TF32 by PyTorch default; against its fp32 CPU results the outputs agree to ~5e-3 relative (tests/test_decoders_gpu.py).
"""
import ctypes

import torch
from torch import nn

from . import _lib
from ._lib import check, ptr


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pad32(n):
    return (n + 31) // 32 * 32


def _pixel_major(x):
    """[C,H,W] -> ([P,C] fp32 contiguous, H, W) without a copy when x is a permuted view of [H,W,C] memory."""
    c, h, w = x.shape
    xp = x.permute(1, 2, 0)
    if not (xp.is_contiguous() and xp.dtype == torch.float32):
        xp = xp.contiguous().float()
    return xp.reshape(h * w, c), h, w


class _Stack(nn.Module):
    """A stack of 1x1 convolutions held as nn.Conv2d (+ nn.ReLU entries: the reference's ModuleList indices, so the
    parameter names match) whose forward is run by the GEMM kernels."""

    def __init__(self, dims_in, dims_out):
        super().__init__()
        layers = []
        for i, (ci, co) in enumerate(zip(dims_in, dims_out)):
            if i > 0:
                layers.append(nn.ReLU())
            layers.append(nn.Conv2d(ci, co, kernel_size=1))
        self.decoder = nn.ModuleList(layers)

    def convs(self):
        return [m for m in self.decoder if isinstance(m, nn.Conv2d)]

    def _packed(self):
        """bf16 [N_pad, K_pad] weights (zero padded so that every K is a multiple of 32 and every N but the last one
        too) and fp32 [N_pad] biases of every layer."""
        out = []
        convs = self.convs()
        for i, m in enumerate(convs):
            co, ci = m.weight.shape[:2]
            kp = _pad32(ci)
            np_ = _pad32(co) if i + 1 < len(convs) else (co + 3) // 4 * 4
            w = torch.zeros(np_, kp, device=m.weight.device, dtype=torch.bfloat16)
            w[:co, :ci] = m.weight.detach()[:, :, 0, 0].to(torch.bfloat16)
            b = torch.zeros(np_, device=m.weight.device)
            b[:co] = m.bias.detach()
            out.append((w, b))
        return out

    @staticmethod
    def _layer(n_pix, w, b, a1, a2=None, relu=True, f32=False):
        n, k = w.shape
        dev = w.device
        y = None if f32 else torch.empty(n_pix, n, dtype=torch.bfloat16, device=dev)
        yf = torch.empty(n_pix, n, device=dev) if f32 else None
        check(_lib.load().gags_decoder_layer(n_pix, n, k, ptr(a1), ptr(a2), ptr(w), ptr(b), int(relu), None, None, ptr(y),
                                             ptr(yf), _st()), "gags_decoder_layer")
        return yf if f32 else y

    def _input(self, x, k_pad):
        if not x.is_cuda:
            raise RuntimeError("gags_amd.decoders: tensors must live on the GPU (there is no CPU path)")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("gags_amd.decoders: forward only so far (run under torch.no_grad(); the reference "
                                      "uses the decoders without grad in render.py / evaluate_iou_loc.py)")
        xp, h, w = _pixel_major(x)
        a = torch.empty(h * w, k_pad, dtype=torch.bfloat16, device=x.device)
        check(_lib.load().gags_decoder_pack_input(h * w, xp.shape[1], k_pad, ptr(xp), ptr(a), _st()), "gags_decoder_pack_input")
        return a, h, w

    @staticmethod
    def _head(logits, c, mode, h, w):
        out = torch.empty(c, h, w, device=logits.device)
        check(_lib.load().gags_decoder_head(h * w, c, logits.shape[1], mode, ptr(logits), ptr(out), _st()), "gags_decoder_head")
        return out


class CNN_decoder(_Stack):
    def __init__(self, input_dim, output_dim):
        super().__init__([input_dim] + [256] * 8, [256] * 8 + [output_dim])
        self.output_dim = output_dim

    def forward(self, x):
        wb = self._packed()
        a, h, w = self._input(x, wb[0][0].shape[1])
        p = h * w
        x1 = self._layer(p, *wb[0], a)
        x2 = self._layer(p, *wb[2], self._layer(p, *wb[1], x1))
        x3 = self._layer(p, *wb[3], x1, x2)                      # conv(x1 + x2)
        x4 = self._layer(p, *wb[5], self._layer(p, *wb[4], x3))
        x5 = self._layer(p, *wb[6], x3, x4)                      # conv(x3 + x4)
        x5 = self._layer(p, *wb[7], x5)
        logits = self._layer(p, *wb[8], x5, relu=False, f32=True)
        return self._head(logits, self.output_dim, 0, h, w)       # F.normalize(dim=0)


class CNN_scale_decoder(_Stack):
    def __init__(self, input_dim, output_dim):
        dims = [64, 128, 64, 32, 16, output_dim]
        super().__init__([input_dim] + dims[:-1], dims)
        self.output_dim = output_dim

    def forward(self, x):
        wb = self._packed()
        a, h, w = self._input(x, wb[0][0].shape[1])
        p = h * w
        for i, (wt, b) in enumerate(wb):
            last = i + 1 == len(wb)
            a = self._layer(p, wt, b, a, relu=not last, f32=last)
        return self._head(a, self.output_dim, 1, h, w)            # softmax(dim=0)
