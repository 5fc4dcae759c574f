"""The per-pixel decoders of the reference, same class names, constructor arguments, parameter
names (`decoder.<i>.weight` / `.bias`: a reference state_dict loads unchanged) and outputs as
models/networks.py:109-218 `CNN_decoder` and :220-248 `CNN_scale_decoder`, computed by hand-written bf16 matrix-core
GEMM kernels (include/gags_next.h N1, csrc/decoder.hip) instead of cuDNN 1x1 convolutions.

    CNN_decoder(16, 512):       x [16,H,W] -> 9 x (1x1 conv, 256 hidden, ReLU), x3 = x1 + x2, x5 = x3 + x4 -> F.normalize(dim=0)
    CNN_scale_decoder(16, 3):   x [16,H,W] -> 16-64-128-64-32-16-3 (ReLU between) -> softmax(dim=0)

Input: [C,H,W]; when it is the rasterizer's output (a permuted view of [H,W,C] memory, gaussian_renderer.py) the
pixel-major layout the kernels want is already there and nothing is transposed.  Output: [C_out,H,W] fp32, contiguous.

Precision.  The reference's modules are fp32 Conv2d stacks (models/networks.py:139-218, 220-248; train.py:149,159), and
`precision="exact"` -- the DEFAULT -- computes them with fp32 tensors and fp32-equivalent arithmetic on the 16-bit matrix
cores: every operand as three bfloat16 terms, every product as its six terms of order <= 2, fp32 accumulation
(csrc/decoder_exact.hip); weight gradients are summed over pixel chunks in a fixed order, without atomics
(bit-reproducible).  Against the reference modules' own fp32 results (tests/golden/next_vectors.npz): outputs <= 1e-5,
every gradient <= 1e-3 rel-L2 (tests/test_decoders_gpu.py).
`precision="bf16x2"` (round 4) is the tier matched to what the reference actually computes: torch leaves
`torch.backends.cudnn.allow_tf32` at its default (True) and the reference never touches it, so on the GPU its README names
(RTX 4090, PyTorch 2.1: README.md:26-31) the nn.Conv2d stacks of models/networks.py:145-149,229-233 run in TF32 -- operands
rounded to 10-bit significands.  gfx950 has no TF32; this tier splits every operand into TWO bfloat16 terms (16 significand
bits) and multiplies three matrix terms per product (h h' + h m' + m h'), fp32 tensors and accumulation, the same
deterministic reductions: relative error <= ~2^-16 per product, 32x tighter than TF32, half the matrix work of "exact".
Against the reference modules' own fp32 results: outputs <= 1e-4, every gradient <= 1e-3 (tests/test_decoders_gpu.py).
`precision="f16"` (round 5) is the tier of EQUAL width to the reference's arithmetic: IEEE-half operands carry TF32's own
11-bit significand, products accumulate in fp32 (v_mfma_f32_32x32x16_f16, the rate of the bf16 instruction), activations
and their gradients are stored as half between layers.  It runs the bf16 mode's kernels compiled a second time for the
other operand type (csrc/half16.h: entry points ..._h16), including the fused chains.  What half lacks is exponent range:
conversions saturate at +-65504 instead of producing inf, and the backward multiplies the cotangent by a power of two
chosen on the device from its magnitude (no host sync; `_pow2_scale`) and divides the results by it -- exact, so a
cotangent scaled by 2^k gives gradients scaled by exactly 2^k.  Measured against the reference modules' fp32 autograd:
4.5e-2 worst gradient rel-L2, the same chain with TF32-rounded operands (what the reference's cuDNN convolutions do): 4.6e-2
(tests/test_decoders_gpu.py::test_f16_tier_is_as_close_to_fp32_as_the_tf32_arithmetic_the_reference_runs); one 1080p
iteration 20.4 ms against bf16x2's 62 ms (profiles/r05_d16_iterations.txt).
`precision="bf16"` is the fast opt-in (`dec.precision = "bf16"` or the constructor argument): bf16 operands, fp32
accumulation, activations and their gradients kept in bf16 between layers -- outputs ~5e-3, gradients ~2e-2 of the fp32
results (a low-precision forward flips the ReLU of units within rounding of zero).  Its weight gradients are
deterministic too (partials + ordered sum).  Backward in both modes: input-gradient GEMMs are the forward kernel with
W^T (ReLU mask and skip-connection gradient fused into the epilogue), weight gradients contract over the pixels; the
input gradient comes back in the rasterizer's [H,W,C] layout; gradients nobody asked for (a detached input, frozen
parameters) are not computed.
"""
import ctypes
import weakref

import torch
from torch import nn

from . import _lib
from ._lib import check, ptr


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _H16Mode:
    """The 16-bit operand type of the fast kernels: which instantiation of the library's entry points (suffix) and which
    torch dtype their 16-bit tensors have.  bf16: the "bf16" mode; f16: IEEE half, the "f16" tier (csrc/half16.h)."""

    def __init__(self, name, sfx, dtype):
        self.name, self.sfx, self.dtype = name, sfx, dtype

    def fn(self, base):
        return getattr(_lib.load(), base + self.sfx)


_BF16 = _H16Mode("bf16", "", torch.bfloat16)
_F16 = _H16Mode("f16", "_h16", torch.float16)
# f16 tier: the gradient entering a backward chain is multiplied by a power of two S such that S * max|cotangent proxy| is
# 2^F16_TARGET_LOG2 (half has 5 exponent bits: unscaled, the 1e-9 .. 1e-6 gradients of a mean-reduced loss at 1080p would be
# flushed; conversions saturate at 65504, so a proxy that underestimates costs accuracy on the largest elements, never an inf)
F16_TARGET_LOG2 = 8.0


def _pow2_scale(amax, target_log2=F16_TARGET_LOG2, div=1.0):
    """Device scalars (S, 1 / S), S = 2^floor(target - log2(amax / div)) (1 when amax is 0 or not finite); no host sync, one
    launch (gags_pow2_scale)."""
    a = amax.detach().reshape(1)
    a = a if a.dtype == torch.float32 else a.float()
    out = torch.empty(2, device=a.device)
    check(_lib.load().gags_pow2_scale(ptr(a), float(div), float(target_log2), ptr(out), _st()), "gags_pow2_scale")
    return out[0:1], out[1:2]


def _pad32(n):
    return (n + 31) // 32 * 32


def _pixel_major(x):
    """[C,H,W] -> ([P,C] fp32 contiguous, H, W) without a copy when x is a permuted view of [H,W,C] memory."""
    c, h, w = x.shape
    xp = x.permute(1, 2, 0)
    if not (xp.is_contiguous() and xp.dtype == torch.float32):
        xp = xp.contiguous().float()
    return xp.reshape(h * w, c), h, w


def _layer(n_pix, w, b, a1, a2=None, relu=True, f32=False, mask_src=None, residual=None, premask=False, mode=_BF16):
    """One GEMM layer (gags_decoder_layer).  Returns y (16-bit, or fp32 with f32=True) and, with premask=True, also the
    16-bit value before the mask."""
    n, k = w.shape
    dev = w.device
    y = None if f32 else torch.empty(n_pix, n, dtype=mode.dtype, device=dev)
    yf = torch.empty(n_pix, n, device=dev) if f32 else None
    ypre = torch.empty(n_pix, n, dtype=mode.dtype, device=dev) if premask else None
    check(mode.fn("gags_decoder_layer")(n_pix, n, k, ptr(a1), ptr(a2), ptr(w), ptr(b), int(relu), ptr(mask_src),
                                        ptr(residual), ptr(y), ptr(ypre), ptr(yf), _st()), "gags_decoder_layer")
    out = yf if f32 else y
    return (out, ypre) if premask else out


def _wgrad(n_pix, dz, a1, a2, n, k, mode=_BF16, shape=None, scale=None):
    """Weight and bias gradient of one layer from the padded operands.  shape = the parameter's own (co, ci, ...): the gradient
    is written in that shape (gags_decoder_wgrad_out: no slice copy afterwards); scale: a device scalar every sum is multiplied
    by (the f16 tier's power of two)."""
    co, ci = (n, k) if shape is None else (shape[0], shape[1])
    dw = torch.empty((co, ci) if shape is None else shape, device=dz.device)
    db = torch.empty(co, device=dz.device)
    nb = mode.fn("gags_decoder_wgrad_scratch_bytes")(n_pix, n, k)
    scratch = torch.empty(max(nb, 4), dtype=torch.uint8, device=dz.device)  # partial matrices per pixel chunk
    check(mode.fn("gags_decoder_wgrad_out")(n_pix, n, k, ptr(dz), ptr(a1), ptr(a2), ptr(dw), ptr(db), co, ci, ptr(scale),
                                            ptr(scratch), nb, _st()), "gags_decoder_wgrad_out")
    return dw, db


_PACK_CACHE = {}  # (id of the first weight) -> (versions, packed): decoders whose parameters did not change are not repacked


def _pack_weights(weights, biases, mode=_BF16):
    """bf16 [N_pad, K_pad] weights and fp32 [N_pad] biases, every dimension zero-padded to a multiple of 32; each packed
    weight also carries its transpose (for the input-gradient GEMMs) and both in MFMA-fragment order (for the fused
    kernels) as attributes -- one gags_decoder_pack_layers launch for the whole decoder, and none while the parameters' version
    counters stand still (frozen decoders, the second use within one iteration's backward)."""
    key = (id(weights[0]), mode.name)
    vers = tuple((t.data_ptr(), t._version) for t in list(weights) + list(biases))
    hit = _PACK_CACHE.get(key)
    # (the weak references tell a parameter from a later one that got the same id, address and version counter)
    if hit is not None and hit[0] == vers and all(r() is t for r, t in zip(hit[2], list(weights) + list(biases))):
        return hit[1]
    out, srcs, bsrcs, dims = [], [], [], []
    for wt, bs in zip(weights, biases):
        co, ci = wt.shape[:2]
        n, k = _pad32(co), _pad32(ci)
        dev = wt.device
        w = torch.empty(n, k, device=dev, dtype=mode.dtype)
        w_t = torch.empty(k, n, device=dev, dtype=mode.dtype)
        wf = torch.empty(n // 32, k // 16, 2, 32, 8, device=dev, dtype=mode.dtype)
        wtf = torch.empty(k // 32, n // 16, 2, 32, 8, device=dev, dtype=mode.dtype)
        b = torch.empty(n, device=dev)
        src = wt.detach().reshape(co, ci)
        srcs.append(src if (src.is_contiguous() and src.dtype == torch.float32) else src.contiguous().float())
        bsrc = bs.detach()
        bsrcs.append(bsrc if (bsrc.is_contiguous() and bsrc.dtype == torch.float32) else bsrc.contiguous().float())
        dims.append((co, ci))
        w._gags_frag, w._gags_t = wf, w_t
        w_t._gags_frag = wtf
        out.append((w, b))
    for i0 in range(0, len(out), 12):  # one launch per twelve layers (gags_decoder_pack_layers)
        sl = slice(i0, i0 + 12)
        m = len(out[sl])
        ints, ptrs = ctypes.c_int * m, ctypes.c_void_p * m
        check(mode.fn("gags_decoder_pack_layers")(
            m, ints(*[d[0] for d in dims[sl]]), ints(*[d[1] for d in dims[sl]]), ptrs(*[t.data_ptr() for t in srcs[sl]]),
            ptrs(*[t.data_ptr() for t in bsrcs[sl]]), ptrs(*[w.data_ptr() for w, _ in out[sl]]),
            ptrs(*[w._gags_t.data_ptr() for w, _ in out[sl]]), ptrs(*[w._gags_frag.data_ptr() for w, _ in out[sl]]),
            ptrs(*[w._gags_t._gags_frag.data_ptr() for w, _ in out[sl]]), ptrs(*[b.data_ptr() for _, b in out[sl]]), _st()),
            "gags_decoder_pack_layers")
    if len(_PACK_CACHE) > 16:
        _PACK_CACHE.clear()
    _PACK_CACHE[key] = (vers, out, [weakref.ref(t) for t in list(weights) + list(biases)])
    return out


def invalidate_packed():
    """Forget every packed weight set.  Needed only after a parameter was written through `.data` (which does not move its
    version counter); optimizers, load_state_dict and in-place ops under no_grad all do."""
    _PACK_CACHE.clear()


def _transposed(w):
    """[K_pad, N_pad] of a packed weight (contracts over N in the input-gradient GEMMs)."""
    t = getattr(w, "_gags_t", None)
    return t if t is not None else w.t().contiguous()


def _xlayer(n_pix, w, b, a1, a2=None, relu=True, mask_src=None, residual=None, premask=False, ldy=None, terms=3):
    """One layer on fp32 tensors with every operand split into `terms` bfloat16 terms (gags_decoder_layer_split; 3: exact,
    2: the bf16x2 tier); w [n_out, k_in]; a1 / a2 [n_pix, >= k_in].  ldy > n_out: the extra columns are zero (the head
    kernels want padded logits rows)."""
    n, k = w.shape
    dev = w.device
    ldy = ldy or n
    y = (torch.zeros if ldy > n else torch.empty)(n_pix, ldy, device=dev)
    ypre = torch.empty(n_pix, ldy, device=dev) if premask else None
    check(_lib.load().gags_decoder_layer_split(n_pix, n, k, ptr(a1), ptr(a2), a1.shape[1], ptr(w), ptr(b), int(relu),
                                               ptr(mask_src), ptr(residual), ptr(y), ptr(ypre), ldy, terms, _st()),
          "gags_decoder_layer_split")
    return (y, ypre) if premask else y


def _xwgrad(n_pix, dz, a1, a2, n, k, want_bias=True, terms=3):
    lib = _lib.load()
    dev = dz.device
    dw = torch.empty(n, k, device=dev)
    db = torch.empty(n, device=dev) if want_bias else None
    nb = lib.gags_decoder_wgrad_exact_scratch_bytes(n_pix, n, k)
    scratch = torch.empty(max(nb, 4), dtype=torch.uint8, device=dev)
    check(lib.gags_decoder_wgrad_split(n_pix, n, k, ptr(dz), dz.shape[1], ptr(a1), ptr(a2), a1.shape[1], ptr(dw), ptr(db),
                                       ptr(scratch), nb, terms, _st()), "gags_decoder_wgrad_split")
    return dw, db


def _logits_ld(c_out):
    """Row length of the logits the head kernels read: c_out itself when it is a multiple of 32, else padded to 8."""
    return c_out if c_out % 32 == 0 else (c_out + 7) // 8 * 8


def _chain_forward_exact(x, kind, params, terms=3):
    """The fp32 chain of a decoder up to its logits (precision="exact": terms = 3; "bf16x2": terms = 2).  Same return tuple
    as _chain_forward; `wb` holds the fp32 [out, in] weight matrices and biases."""
    import functools
    _xl = functools.partial(_xlayer, terms=terms)
    weights, biases = params[0::2], params[1::2]
    # private copies (2.4 MB for CNN_decoder), with the transposes the input-gradient GEMMs contract made NOW: for fp32
    # [co, ci, 1, 1] weights `[:, :, 0, 0].contiguous().float()` is a VIEW of the live parameter, and a backward that runs
    # after an in-place optimizer step would silently differentiate the new weights
    wb = []
    for wt, bs in zip(weights, biases):
        wm = wt.detach()[:, :, 0, 0].float().clone()
        wm._gags_t = wm.t().contiguous()
        wb.append((wm, bs.detach().float().clone()))
    xp, h, w = _pixel_major(x)
    p = h * w
    a0 = xp
    ld = _logits_ld(wb[-1][0].shape[0])
    if kind == "decoder":
        x1 = _xl(p, *wb[0], a0)
        t1 = _xl(p, *wb[1], x1)
        x2 = _xl(p, *wb[2], t1)
        x3 = _xl(p, *wb[3], x1, x2)   # conv(x1 + x2)
        t4 = _xl(p, *wb[4], x3)
        x4 = _xl(p, *wb[5], t4)
        t6 = _xl(p, *wb[6], x3, x4)   # conv(x3 + x4)
        t7 = _xl(p, *wb[7], t6)
        logits = _xl(p, *wb[8], t7, relu=False, ldy=ld)
        acts = [a0, x1, t1, x2, x3, t4, x4, t6, t7]
    else:
        acts = [a0]
        a = a0
        for i, (wt, b) in enumerate(wb):
            last = i + 1 == len(wb)
            a = _xl(p, wt, b, a, relu=not last, ldy=ld if last else None)
            if not last:
                acts.append(a)
        logits = a
    return logits, acts, wb, h, w, xp.shape[1]


def _chain_backward_exact(dz, acts, wb, kind, h, w, c_in, shapes, need_x=True, need_w=None, terms=3):
    """fp32 backward of the chain from dz [P, c_out] (precision="exact" / "bf16x2")."""
    p = h * w
    need_w = need_w or [True] * len(wb)
    wt = [_transposed(wgt) for wgt, _ in wb]  # [k_in, n_out] (made in the forward): the input-gradient GEMM contracts over n_out
    dws = [None] * len(wb)

    def wg(i, dz_i, a1, a2=None):
        if need_w[i]:
            dws[i] = _xwgrad(p, dz_i, a1, a2, *wb[i][0].shape, terms=terms)

    def dx(i, dz_i, mask_src=None, residual=None, premask=False):
        return _xlayer(p, wt[i], None, dz_i, relu=False, mask_src=mask_src, residual=residual, premask=premask, terms=terms)

    gin = None
    if kind == "decoder":
        a0, x1, t1, x2, x3, t4, x4, t6, t7 = acts
        wg(8, dz, t7)
        dz7 = dx(8, dz, mask_src=t7)
        wg(7, dz7, t6)
        dz6 = dx(7, dz7, mask_src=t6)
        wg(6, dz6, x3, x4)
        dz5, g36 = dx(6, dz6, mask_src=x4, premask=True)
        wg(5, dz5, t4)
        dz4 = dx(5, dz5, mask_src=t4)
        wg(4, dz4, x3)
        dz3 = dx(4, dz4, mask_src=x3, residual=g36)
        wg(3, dz3, x1, x2)
        dz2, g13 = dx(3, dz3, mask_src=x2, premask=True)
        wg(2, dz2, t1)
        dz1 = dx(2, dz2, mask_src=t1)
        wg(1, dz1, x1)
        dz0 = dx(1, dz1, mask_src=x1, residual=g13)
        wg(0, dz0, a0)
        if need_x:
            gin = dx(0, dz0)
    else:
        cur = dz
        for i in range(len(wb) - 1, -1, -1):
            wg(i, cur, acts[i])
            if i > 0 or need_x:
                cur = dx(i, cur, mask_src=acts[i] if i > 0 else None)
        gin = cur if need_x else None
    gx = None if gin is None else gin.view(h, w, c_in).permute(2, 0, 1)
    grads = []
    for pair, shp in zip(dws, shapes):
        if pair is None:
            grads += [None, None]
        else:
            grads += [pair[0].reshape(shp), pair[1]]
    return gx, grads


FUSED = True  # bf16 mode: CNN_decoder's forward (and input-gradient) chain as one kernel each; False: layer by layer


def _frag_layout(w):
    """bf16 [N, K] (N % 32 == 0, K % 16 == 0) -> [N / 32, K / 16, 64, 8]: the A operand of one v_mfma_f32_32x32x16_bf16 (lane
    32 kh + n holds k = 16 s + 8 kh .. + 7 of row 32 t + n) as one contiguous kilobyte (csrc/decoder_fused.hip: layer_mma)."""
    f = getattr(w, "_gags_frag", None)  # packed by gags_decoder_pack_layer
    if f is not None:
        return f
    n, k = w.shape
    return w.view(n // 32, 32, k // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()


def _fusable(wb, c_in):
    """The reference's CNN_decoder shape: 9 layers, 256 hidden, c_in <= 32, an output width that is a multiple of 256."""
    shapes = [tuple(w.shape) for w, _ in wb]
    return (len(wb) == 9 and c_in <= 32 and shapes[0] == (256, 32) and all(sh == (256, 256) for sh in shapes[1:8])
            and shapes[8][1] == 256 and shapes[8][0] % 256 == 0)


_SCALE_SHAPES = [(64, 32), (128, 64), (64, 128), (32, 64), (32, 32), (32, 32)]  # CNN_scale_decoder, padded to 32s


def _scale_fusable(wb, c_in):
    """The reference's CNN_scale_decoder shape (16 -> 64 -> 128 -> 64 -> 32 -> 16 -> 3): csrc/decoder_scale.hip."""
    return c_in <= 32 and [tuple(w.shape) for w, _ in wb] == _SCALE_SHAPES


def _chain_forward(x, kind, params, mode=_BF16, head_out=None):
    """The GEMM chain of a decoder up to its fp32 logits.  Returns (logits [P, ld], activations, packed weights, h, w,
    c_in).  head_out (a [3, H, W] fp32 tensor): the fused CNN_scale_decoder kernel writes its softmax head there itself and
    `logits` comes back None (nothing reads them: the head's backward works from its output)."""
    weights, biases = params[0::2], params[1::2]
    wb = _pack_weights(weights, biases, mode)
    xp, h, w = _pixel_major(x)
    p = h * w
    a0 = torch.empty(p, wb[0][0].shape[1], dtype=mode.dtype, device=x.device)
    _layer_m = lambda *a, **k: _layer(*a, mode=mode, **k)  # noqa: E731
    if kind == "scale" and FUSED and _scale_fusable(wb, xp.shape[1]):
        # the six layers in one kernel, a wave per 32-pixel tile (csrc/decoder_scale.hip): bit-identical to the chain below
        dev = x.device
        acts = [a0] + [torch.empty(p, wgt.shape[0], dtype=mode.dtype, device=dev) for wgt, _ in wb[:5]]
        logits = torch.empty(p, 32, device=dev) if head_out is None else None
        arr = ctypes.c_void_p * 6
        wf = [_frag_layout(wgt) for wgt, _ in wb]
        masks = torch.empty(p, 11, dtype=torch.int32, device=dev)  # the ReLU decisions as bits, for the fused backward
        check(mode.fn("gags_scale_decoder_fwd_fused_head")(p, xp.shape[1], ptr(xp), arr(*[t.data_ptr() for t in wf]),
                                                            arr(*[b.data_ptr() for _, b in wb]), arr(*[t.data_ptr() for t in acts]),
                                                            ptr(masks), ptr(logits), ptr(head_out), _st()),
              "gags_scale_decoder_fwd_fused_head")
        return logits, acts + [masks], wb, h, w, xp.shape[1]
    fused = kind == "decoder" and FUSED and _fusable(wb, xp.shape[1])
    if not fused:  # (the fused kernel converts its input tile itself and keeps it as a0)
        check(mode.fn("gags_decoder_pack_input")(p, xp.shape[1], a0.shape[1], ptr(xp), ptr(a0), _st()), "gags_decoder_pack_input")
    acts = [a0]
    if fused:
        # the nine layers in one kernel, activations resident in LDS (csrc/decoder_fused.hip): bit-identical to the chain below
        dev = x.device
        acts = [a0] + [torch.empty(p, 256, dtype=mode.dtype, device=dev) for _ in range(8)]
        logits = torch.empty(p, wb[8][0].shape[0], device=dev)
        arr = ctypes.c_void_p * 9
        wf = [_frag_layout(wgt) for wgt, _ in wb]
        masks = torch.empty(8, (p + 63) // 64 * 64, 8, dtype=torch.int32, device=dev)  # the ReLU decisions as bits, for the fused
        # backward: 8 layers x whole 64-pixel groups x 8 words (an opaque buffer: the two kernels' own layout)
        check(mode.fn("gags_decoder_fwd_fused")(p, xp.shape[1], logits.shape[1], ptr(xp), arr(*[t.data_ptr() for t in wf]),
                                                 arr(*[b.data_ptr() for _, b in wb]), arr(*[t.data_ptr() for t in acts]),
                                                 ptr(masks), ptr(logits), _st()), "gags_decoder_fwd_fused")
        return logits, acts + [masks], wb, h, w, xp.shape[1]
    if kind == "decoder":
        x1 = _layer_m(p, *wb[0], a0)
        t1 = _layer_m(p, *wb[1], x1)
        x2 = _layer_m(p, *wb[2], t1)
        x3 = _layer_m(p, *wb[3], x1, x2)   # conv(x1 + x2)
        t4 = _layer_m(p, *wb[4], x3)
        x4 = _layer_m(p, *wb[5], t4)
        t6 = _layer_m(p, *wb[6], x3, x4)   # conv(x3 + x4)
        t7 = _layer_m(p, *wb[7], t6)
        logits = _layer_m(p, *wb[8], t7, relu=False, f32=True)
        acts += [x1, t1, x2, x3, t4, x4, t6, t7]
    else:
        a = a0
        for i, (wt, b) in enumerate(wb):
            last = i + 1 == len(wb)
            a = _layer_m(p, wt, b, a, relu=not last, f32=last)
            if not last:
                acts.append(a)
        logits = a
    return logits, acts, wb, h, w, xp.shape[1]


def _chain_backward(dz, acts, wb, kind, h, w, c_in, shapes, need_x=True, need_w=None, mode=_BF16, scale=None):
    """From the bf16 gradient of the logits dz [P, ld] back through the chain: (input gradient as a [C_in,H,W] view of
    [H,W,C_in] memory, weight / bias gradients in parameter order, whether the input gradient already carries `scale`).
    need_x / need_w: what autograd asked for; scale: a device scalar the parameter gradients (and, in the fused kernel, the
    input gradient) are multiplied by on their way out."""
    p = h * w
    need_w = need_w or [True] * len(wb)
    wt = [_transposed(wgt) for wgt, _ in wb]  # [K_pad, N_pad]: the input-gradient GEMM contracts over N
    dws = [None] * len(wb)

    def wg(i, dz_i, a1, a2=None):
        if need_w[i]:  # (written in the parameter's shape, times `scale`: nothing left to do per parameter afterwards)
            dws[i] = _wgrad(p, dz_i, a1, a2, *wb[i][0].shape, mode=mode, shape=shapes[i], scale=scale)

    def dx(i, dz_i, mask_src=None, residual=None, premask=False):
        return _layer(p, wt[i], None, dz_i, relu=False, mask_src=mask_src, residual=residual, premask=premask, mode=mode)

    if kind == "decoder" and len(acts) == 10:  # the fused forward ran (its bit masks ride along as the tenth entry)
        # the nine input-gradient GEMMs in one kernel (csrc/decoder_fused.hip), then the weight gradients
        a0, x1, t1, s12, x3, t4, s34, t6, t7, masks = acts  # (the fused forward keeps x1 + x2 and x3 + x4 in place of x2 / x4)
        dev = dz.device
        dzs = [torch.empty(p, 256, dtype=mode.dtype, device=dev) for _ in range(8)]
        gx = torch.empty(h, w, c_in, device=dev) if need_x else None
        arr = ctypes.c_void_p * 9
        arr8 = ctypes.c_void_p * 8
        wtf = [_frag_layout(t) for t in wt]
        check(mode.fn("gags_decoder_bwd_fused_scaled")(p, c_in, dz.shape[1], ptr(dz), arr(*[t.data_ptr() for t in wtf]), ptr(masks),
                                                        arr8(*[t.data_ptr() for t in dzs]), ptr(gx), ptr(scale), _st()),
              "gags_decoder_bwd_fused_scaled")
        scaled_gx = True
        wg(8, dz, t7); wg(7, dzs[7], t6); wg(6, dzs[6], s34); wg(5, dzs[5], t4); wg(4, dzs[4], x3)
        wg(3, dzs[3], s12); wg(2, dzs[2], t1); wg(1, dzs[1], x1); wg(0, dzs[0], a0)
        grads = []
        for pair in dws:
            grads += [None, None] if pair is None else [pair[0], pair[1]]
        return (None if gx is None else gx.permute(2, 0, 1)), grads, scaled_gx
    if kind == "decoder":
        a0, x1, t1, x2, x3, t4, x4, t6, t7 = acts
        wg(8, dz, t7)
        dz7 = dx(8, dz, mask_src=t7)
        wg(7, dz7, t6)
        dz6 = dx(7, dz7, mask_src=t6)
        wg(6, dz6, x3, x4)
        dz5, g36 = dx(6, dz6, mask_src=x4, premask=True)     # d(x3 + x4): masked for x4's ReLU, raw for the skip to x3
        wg(5, dz5, t4)
        dz4 = dx(5, dz5, mask_src=t4)
        wg(4, dz4, x3)
        dz3 = dx(4, dz4, mask_src=x3, residual=g36)           # both paths into x3, then its ReLU
        wg(3, dz3, x1, x2)
        dz2, g13 = dx(3, dz3, mask_src=x2, premask=True)
        wg(2, dz2, t1)
        dz1 = dx(2, dz2, mask_src=t1)
        wg(1, dz1, x1)
        dz0 = dx(1, dz1, mask_src=x1, residual=g13)
        wg(0, dz0, a0)
        gin = dx(0, dz0) if need_x else None
    else:
        smask = acts[len(wb)] if len(acts) == len(wb) + 1 else None  # the fused forward's ReLU bit masks (csrc/decoder_scale.hip)
        if smask is not None and FUSED and dz.shape[1] == 32:
            # the five input-gradient GEMMs in one kernel (bit-identical to the loop below), then the weight gradients
            dzs = [torch.empty(p, wgt.shape[0], dtype=mode.dtype, device=dz.device) for wgt, _ in wb[:5]]
            arr6, arr5 = ctypes.c_void_p * 6, ctypes.c_void_p * 5
            wtf = [None] + [_frag_layout(t) for t in wt[1:]]
            check(mode.fn("gags_scale_decoder_bwd_fused")(p, ptr(dz), arr6(*[None if t is None else t.data_ptr() for t in wtf]),
                                                           ptr(smask), arr5(*[t.data_ptr() for t in dzs]), _st()),
                  "gags_scale_decoder_bwd_fused")
            for i in range(len(wb) - 1, -1, -1):
                wg(i, dz if i == 5 else dzs[i], acts[i])
            gin = dx(0, dzs[0]) if need_x else None
        else:
            cur = dz
            for i in range(len(wb) - 1, -1, -1):
                wg(i, cur, acts[i])
                if i > 0 or need_x:
                    cur = dx(i, cur, mask_src=acts[i] if i > 0 else None)
            gin = cur if need_x else None
    gx = None
    if gin is not None:
        gx = torch.empty(h, w, c_in, device=dz.device)
        check(mode.fn("gags_decoder_unpack_grad")(p, c_in, gin.shape[1], ptr(gin), ptr(gx), _st()), "gags_decoder_unpack_grad")
        gx = gx.permute(2, 0, 1)
    grads = []
    for pair in dws:
        grads += [None, None] if pair is None else [pair[0], pair[1]]
    return gx, grads, False


def _needs(ctx, first_param):
    """(input gradient wanted, per-layer: weight or bias gradient wanted) from autograd's needs_input_grad."""
    ng = ctx.needs_input_grad
    return bool(ng[0]), [bool(ng[i] or ng[i + 1]) for i in range(first_param, len(ng), 2)]


class _DecoderFn(torch.autograd.Function):
    """Forward and backward of a decoder as GEMM launches.  `kind` = "decoder" (CNN_decoder: two residual sums,
    normalize head) or "scale" (CNN_scale_decoder: plain chain, softmax head).  params = w0, b0, w1, b1, ..."""

    @staticmethod
    def forward(ctx, x, kind, c_out, precision, *params):
        exact = precision in ("exact", "bf16x2")  # fp32 tensors, split operands: three terms / two terms
        terms = 2 if precision == "bf16x2" else 3
        h16 = _F16 if precision == "f16" else _BF16
        head_out = None
        if not exact and kind == "scale" and c_out == 3 and x.dim() == 3:
            head_out = torch.empty(3, x.shape[1], x.shape[2], device=x.device)  # (used when the fused kernel serves the shape)
        logits, acts, wb, h, w, c_in = (_chain_forward_exact(x, kind, params, terms) if exact
                                        else _chain_forward(x, kind, params, h16, head_out=head_out))
        p = h * w
        ctx.kind, ctx.c_out, ctx.hw, ctx.c_in, ctx.exact, ctx.terms = kind, c_out, (h, w), c_in, exact, terms
        ctx.wb, ctx.h16 = wb, h16
        ctx.shapes = [tuple(t.shape) for t in params[0::2]]
        ctx.from_output = logits is None
        if logits is None:  # the fused scale decoder wrote its softmax head itself; the backward starts from this output
            ctx.save_for_backward(head_out, *acts)
            return head_out
        # CNN_decoder's [C,H,W] output is a permuted view of PIXEL-major memory (like render()'s own output): the head
        # writes rows, the losses read rows, the reference's next step (.permute(1,2,0)) is free.  The 3-channel scale
        # map stays channel-major (its consumers index it by plane).
        pm = kind == "decoder" and c_out % 4 == 0 and logits.shape[1] <= 512 and logits.shape[1] % 32 == 0
        out = torch.empty((h, w, c_out) if pm else (c_out, h, w), device=x.device)
        check(_lib.load().gags_decoder_head(p, c_out, logits.shape[1], 0 if kind == "decoder" else 1, ptr(logits), ptr(out),
                                            1 if pm else 0, _st()), "gags_decoder_head")
        if pm:
            out = out.permute(2, 0, 1)
        ctx.kind, ctx.c_out, ctx.hw, ctx.c_in, ctx.exact, ctx.terms = kind, c_out, (h, w), c_in, exact, terms
        ctx.wb, ctx.h16 = wb, h16
        ctx.shapes = [tuple(t.shape) for t in params[0::2]]
        ctx.save_for_backward(logits, *acts)
        return out

    @staticmethod
    def backward(ctx, g):
        logits, *acts = ctx.saved_tensors
        wb, (h, w), kind = ctx.wb, ctx.hw, ctx.kind
        p = h * w
        lib = _lib.load()
        if ctx.from_output:  # `logits` is the softmax output y [3, H, W]: dz = y (g - <y, g>) (gags_softmax_head_bwd_y)
            y, h16, inv = logits, ctx.h16, None
            g = g if (g.is_contiguous() and g.dtype == torch.float32) else g.contiguous().float()
            if h16 is _F16:
                s, inv = _pow2_scale(torch.linalg.vector_norm(g.reshape(-1), float("inf")))
                g = g * s
            dz = torch.empty(p, 32, dtype=h16.dtype, device=g.device)
            check(h16.fn("gags_softmax_head_bwd_y")(p, ctx.c_out, 32, ptr(y), ptr(g), ptr(dz), _st()), "gags_softmax_head_bwd_y")
            need_x, need_w = _needs(ctx, 4)
            gx, grads, done = _chain_backward(dz, acts, wb, kind, h, w, ctx.c_in, ctx.shapes, need_x, need_w, mode=h16, scale=inv)
            if inv is not None and not done:
                gx = None if gx is None else gx * inv
            return (gx, None, None, None, *grads)
        gp = g.permute(1, 2, 0)
        pm = (g.dtype == torch.float32 and gp.is_contiguous() and not g.is_contiguous() and ctx.c_out % 4 == 0
              and logits.shape[1] <= 512)  # the cotangent came back in the output's own pixel-major layout
        g = gp if pm else (g if (g.is_contiguous() and g.dtype == torch.float32) else g.contiguous().float())
        need_x, need_w = _needs(ctx, 4)
        mode = 0 if kind == "decoder" else 1
        if ctx.exact:
            dz = torch.empty(p, ctx.c_out, device=g.device)
            check(lib.gags_decoder_head_bwd_exact(p, ctx.c_out, logits.shape[1], mode, ptr(logits), ptr(g), 1 if pm else 0,
                                                  ptr(dz), ctx.c_out, _st()), "gags_decoder_head_bwd_exact")
            gx, grads = _chain_backward_exact(dz, acts, wb, kind, h, w, ctx.c_in, ctx.shapes, need_x, need_w, ctx.terms)
        else:
            h16, inv = ctx.h16, None
            if h16 is _F16:  # the cotangent scaled into half's range by a power of two (exact), the results scaled back
                s, inv = _pow2_scale(torch.linalg.vector_norm(g.reshape(-1), float("inf")))  # (max |g| without an |g| temporary)
                g = g * s
            dz = torch.empty(p, logits.shape[1], dtype=h16.dtype, device=g.device)
            check(h16.fn("gags_decoder_head_bwd")(p, ctx.c_out, logits.shape[1], mode, ptr(logits), ptr(g), ptr(dz),
                                                  1 if pm else 0, _st()), "gags_decoder_head_bwd")
            gx, grads, done = _chain_backward(dz, acts, wb, kind, h, w, ctx.c_in, ctx.shapes, need_x, need_w, mode=h16, scale=inv)
            if inv is not None and not done:  # (the parameter gradients left their sums already multiplied)
                gx = None if gx is None else gx * inv
        return (gx, None, None, None, *grads)


class _DecoderDistillFn(torch.autograd.Function):
    """CNN_decoder + the distillation L1 of train.py:159-166 with the output head fused into the loss: the GEMM chain up
    to the fp32 logits, then ONE kernel to l1_map / mask (gags_decoder_head_distill_fwd) and one back to the logits'
    gradient -- the normalised [512,H,W] map and its gradient are never written or read (21 GB per iteration at 1080p)."""

    @staticmethod
    def forward(ctx, x, img_embed, seg_map, scale_map, c_out, precision, *params):
        from .losses import _f
        ctx.terms = {"exact": 3, "bf16x2": 2}.get(precision, 0)  # 0: the 16-bit tensor modes
        ctx.h16 = _F16 if precision == "f16" else _BF16
        logits, acts, wb, h, w, c_in = (_chain_forward_exact(x, "decoder", params, ctx.terms) if ctx.terms
                                        else _chain_forward(x, "decoder", params, ctx.h16))
        e, seg, sc = _f(img_embed), _f(seg_map), _f(scale_map)
        if e.shape[1] != c_out or tuple(sc.shape) != (3, h, w) or seg.dim() != 3 or seg.shape[0] != 4:
            raise ValueError(f"embeddings {tuple(e.shape)}, seg_map {tuple(seg.shape)}, scale_map {tuple(sc.shape)} "
                             f"vs a {c_out}-channel {h}x{w} decoder output")
        l1 = torch.empty(h, w, device=x.device)
        mask = torch.empty(h, w, device=x.device)
        check(_lib.load().gags_decoder_head_distill_fwd(c_out, logits.shape[1], h, w, seg.shape[1], seg.shape[2], e.shape[0],
                                                        ptr(logits), ptr(e), ptr(seg), ptr(sc), ptr(l1), ptr(mask), _st()),
              "gags_decoder_head_distill_fwd")
        ctx.c_out, ctx.hw, ctx.c_in, ctx.wb = c_out, (h, w), c_in, wb
        ctx.shapes = [tuple(t.shape) for t in params[0::2]]
        ctx.save_for_backward(logits, e, seg, sc, *acts)
        ctx.mark_non_differentiable(mask)
        return l1, mask

    @staticmethod
    def backward(ctx, v_map, _v_mask):
        from .losses import _f
        logits, e, seg, sc, *acts = ctx.saved_tensors
        (h, w), c = ctx.hw, ctx.c_out
        vs = torch.empty(3, h, w, device=logits.device)
        need_x, need_w = _needs(ctx, 6)
        if ctx.terms:  # fp32-tensor tiers: the logits' gradient stays fp32
            dz = torch.empty(h * w, logits.shape[1], device=logits.device)
            check(_lib.load().gags_decoder_head_distill_bwd_f32(c, logits.shape[1], h, w, seg.shape[1], seg.shape[2], e.shape[0],
                                                                ptr(logits), ptr(e), ptr(seg), ptr(sc), ptr(_f(v_map)), ptr(dz),
                                                                ptr(vs), _st()), "gags_decoder_head_distill_bwd_f32")
            gx, grads = _chain_backward_exact(dz, acts, ctx.wb, "decoder", h, w, ctx.c_in, ctx.shapes, need_x, need_w, ctx.terms)
        elif ctx.h16 is _F16:
            # the logits' gradient is about v_map / (c |logits|): scaled by a power of two chosen on the device from
            # max |v_map| / c (half has 5 exponent bits), the chain's results scaled back; the scale map's gradient is fp32
            vm = _f(v_map)
            s, inv = _pow2_scale(torch.linalg.vector_norm(vm.reshape(-1), float("inf")), div=c)
            dz = torch.empty(h * w, logits.shape[1], dtype=torch.float16, device=logits.device)
            check(_lib.load().gags_decoder_head_distill_bwd_h16(c, logits.shape[1], h, w, seg.shape[1], seg.shape[2], e.shape[0],
                                                                ptr(logits), ptr(e), ptr(seg), ptr(sc), ptr(vm), ptr(dz), ptr(s),
                                                                ptr(vs), _st()), "gags_decoder_head_distill_bwd_h16")
            gx, grads, done = _chain_backward(dz, acts, ctx.wb, "decoder", h, w, ctx.c_in, ctx.shapes, need_x, need_w, mode=_F16,
                                              scale=inv)
            gx = gx if (gx is None or done) else gx * inv
        else:
            dz = torch.empty(h * w, logits.shape[1], dtype=torch.bfloat16, device=logits.device)
            check(_lib.load().gags_decoder_head_distill_bwd(c, logits.shape[1], h, w, seg.shape[1], seg.shape[2], e.shape[0],
                                                            ptr(logits), ptr(e), ptr(seg), ptr(sc), ptr(_f(v_map)), ptr(dz), ptr(vs),
                                                            _st()), "gags_decoder_head_distill_bwd")
            gx, grads, _ = _chain_backward(dz, acts, ctx.wb, "decoder", h, w, ctx.c_in, ctx.shapes, need_x, need_w)
        return (gx, None, None, vs, None, None, *grads)


class _Stack(nn.Module):
    """A stack of 1x1 convolutions held as nn.Conv2d (+ nn.ReLU entries: the reference's ModuleList indices, so the
    parameter names match) whose forward and backward are run by the GEMM kernels."""
    kind = "scale"

    def __init__(self, dims_in, dims_out, precision="exact"):
        super().__init__()
        if precision not in ("exact", "bf16x2", "f16", "bf16"):
            raise ValueError("precision must be 'exact' (fp32-equivalent, the default), 'bf16x2' (two-term split: 16 significand "
                             "bits, tighter than the TF32 the reference's convs run in), 'f16' (IEEE half operands: TF32's own "
                             "11-bit significand, fp32 accumulation, power-of-two gradient scaling) or 'bf16' (8 bits, fast opt-in)")
        self.precision = precision
        layers = []
        for i, (ci, co) in enumerate(zip(dims_in, dims_out)):
            if i > 0:
                layers.append(nn.ReLU())
            layers.append(nn.Conv2d(ci, co, kernel_size=1))
        self.decoder = nn.ModuleList(layers)

    def convs(self):
        return [m for m in self.decoder if isinstance(m, nn.Conv2d)]

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("gags_amd.decoders: tensors must live on the GPU (there is no CPU path)")
        params = [t for m in self.convs() for t in (m.weight, m.bias)]
        return _DecoderFn.apply(x, self.kind, self.output_dim, self.precision, *params)


class CNN_decoder(_Stack):
    kind = "decoder"

    def __init__(self, input_dim, output_dim, precision="exact"):
        super().__init__([input_dim] + [256] * 8, [256] * 8 + [output_dim], precision)
        self.output_dim = output_dim

    def distill_l1(self, x, img_embed, seg_map, scale_map):
        """train.py:159-166 in one call -- equal to
            feature_map = self(x); gt, mask = read_sam_clip_feature(img_embed, seg_map, scale_map)
            l1_loss_map(feature_map * mask, gt * mask), mask
        with the decoder's normalising head fused into the loss (the [512,H,W] map is never materialised).  Returns
        (l1_map [H,W], mask [1,H,W] bool); gradients reach x, the decoder's parameters and scale_map.  Not part of the
        reference's module: an optional fast path (every precision tier: the fp32-tensor tiers keep the logits' gradient
        in fp32, the bf16 mode rounds it to bf16)."""
        if not x.is_cuda:
            raise RuntimeError("gags_amd.decoders: tensors must live on the GPU (there is no CPU path)")
        if self.output_dim != 512:
            # other widths: the decoder, then the fused ground-truth assembly + L1 map
            l1, mask = __import__("gags_amd.losses", fromlist=["distill_l1_map"]).distill_l1_map(self(x), img_embed, seg_map, scale_map)
            return l1, mask
        params = [t for m in self.convs() for t in (m.weight, m.bias)]
        l1, mask = _DecoderDistillFn.apply(x, img_embed, seg_map, scale_map, self.output_dim, self.precision, *params)
        return l1, (mask != 0)[None]


class CNN_scale_decoder(_Stack):
    def __init__(self, input_dim, output_dim, precision="exact"):
        dims = [64, 128, 64, 32, 16, output_dim]
        super().__init__([input_dim] + dims[:-1], dims, precision)
        self.output_dim = output_dim
