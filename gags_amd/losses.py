"""The image-space losses and ground-truth assembly of the distillation step, same function
names, arguments and results as the reference's, on HIP kernels (include/gags_next.h, csrc/losses.hip):

    utils/loss_utils.py:20-24     l1_loss, l1_loss_map
    utils/loss_utils.py:32-57     Scale_balance_loss(loss_map, seg_map, mask, mix_seg=True)
    utils/loss_utils.py:59-66     scale_regulation_loss(scale_map)
    utils/loss_utils.py:103-136   scale_region_regulation_loss(scale_map, seg_map, mix_seg=True)
    utils/loss_utils.py:138-154   get_trained_seg(seg_map, scale_map)
    scene/dataset_readers.py:54-121  read_sam_clip_feature(img_embed, seg_map, scale_map)   (default mode)
and the fusion the reference's train.py:165-166 spells as three full-size elementwise passes plus a [512,H,W]
ground-truth tensor:
    distill_l1_map(pred, img_embed, seg_map, scale_map) == l1_loss_map(pred * mask, gt * mask), mask

The reference walks the segment ids in a Python loop (one boolean-mask pass over the image per id); here one pass
over the pixels accumulates every segment's moments (gags_segment_stats) and the tiny per-segment arithmetic is done
on [n_seg]-sized tensors.  Only the variants train.py reaches are implemented (mix_seg=True; default gather mode);
anything else raises.  GPU tensors only: there is no CPU path.
"""
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import check, ptr


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f(t):
    if not t.is_cuda:
        raise RuntimeError("gags_amd.losses: tensors must live on the GPU (there is no CPU path)")
    return t if (t.is_contiguous() and t.dtype == torch.float32) else t.contiguous().float()


_NSEG_CACHE = {}  # id(tensor) -> (weak reference, data_ptr, version counter, n_seg)


def _n_seg(seg_map):
    """Size of the per-segment tables: any bound above the largest id gives the same sums (rows nobody hits stay empty, ids
    outside [0, n_seg) are skipped by the kernels).  ids are small non-negative integers stored as floats; the bound costs ONE
    readback per segmentation map, not one per iteration (a host sync in the middle of an iteration drains the queue: the
    dozens of small launches behind it then run at the host's pace -- ~0.5 ms each time at 1080p): a view's SAM map does not
    change between iterations, so the answer is kept per tensor (identity, address and version counter), and
    get_trained_seg hands the bound of its source map on to the map it makes (`_gags_n_seg`)."""
    hint = getattr(seg_map, "_gags_n_seg", None)
    if hint is not None:
        return hint
    key = id(seg_map)
    hit = _NSEG_CACHE.get(key)
    if hit is not None and hit[0]() is seg_map and hit[1] == seg_map.data_ptr() and hit[2] == seg_map._version:
        return hit[3]
    n = max(int(seg_map.max().item()) + 1, 1)
    if len(_NSEG_CACHE) > 4096:
        _NSEG_CACHE.clear()
    _NSEG_CACHE[key] = (weakref.ref(seg_map), seg_map.data_ptr(), seg_map._version, n)
    return n


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


def l1_loss_map(network_output, gt):
    return torch.abs(network_output - gt).mean(dim=0)


class _Entropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale_map):
        s = _f(scale_map)
        acc = torch.zeros(1, dtype=torch.float64, device=s.device)
        check(_lib.load().gags_entropy_fwd(s.numel(), ptr(s), ptr(acc), _st()), "gags_entropy_fwd")
        ctx.save_for_backward(s)
        return (acc[0] / s.numel()).float()

    @staticmethod
    def backward(ctx, v):
        (s,) = ctx.saved_tensors
        vs = torch.empty_like(s)
        vd = v.detach().reshape(1).float().contiguous()  # (read on the device: no readback inside the backward pass)
        check(_lib.load().gags_entropy_bwd_dev(s.numel(), ptr(s), ptr(vd), ptr(vs), _st()), "gags_entropy_bwd_dev")
        return vs


def scale_regulation_loss(scale_map):
    """mean(-s * log(s + 1e-6)) over the [3,H,W] scale map (utils/loss_utils.py:59-66)."""
    return _Entropy.apply(scale_map)


RUNS = os.environ.get("GAGS_SEGMENT_RUNS", "1") != "0"  # 0: the wave-sum + double-atomic kernel for every shape
_STAT_COPIES = 16  # private accumulator sets of gags_segment_stats_multi (the double atomics serialize per address)


def _segment_copies(x, seg_map, n_seg, pixel_major=False):
    """Per-segment moments of x ([c, n_pix], or [n_pix, c] with pixel_major) as private copies: s1, s2 [k, n_seg, c] doubles
    and counts [k, n_seg]; the caller sums over k."""
    n_pix = seg_map.numel()
    c = x.shape[1] if pixel_major else x.shape[0]
    lib = _lib.load()
    k = lib.gags_segment_stats_runs_copies(n_pix, c, n_seg, 1 if pixel_major else 0) if RUNS else 0
    if k > 0:  # sums by runs of equal ids into one private table per workgroup: no global atomics (csrc/losses.hip)
        s1 = torch.empty(k, n_seg, c, dtype=torch.float64, device=x.device)
        s2 = torch.empty_like(s1)
        cnt = torch.empty(k, n_seg, dtype=torch.int32, device=x.device)
        check(lib.gags_segment_stats_runs(n_pix, c, ptr(x), ptr(seg_map), n_seg, k, ptr(s1), ptr(s2), ptr(cnt),
                                          1 if pixel_major else 0, _st()), "gags_segment_stats_runs")
        return s1, s2, cnt
    k = _STAT_COPIES if n_seg * c * _STAT_COPIES <= (1 << 22) else 1
    s1 = torch.zeros(k, n_seg, c, dtype=torch.float64, device=x.device)
    s2 = torch.zeros_like(s1)
    cnt = torch.zeros(k, n_seg, dtype=torch.int32, device=x.device)
    check(lib.gags_segment_stats_multi(n_pix, c, ptr(x), ptr(seg_map), n_seg, k, ptr(s1), ptr(s2), ptr(cnt),
                                       1 if pixel_major else 0, _st()), "gags_segment_stats_multi")
    return s1, s2, cnt


def _segment_stats(x, seg_map, n_seg, pixel_major=False):
    """The moments summed over the copies: s1, s2 [n_seg, c], counts [n_seg]."""
    s1, s2, cnt = _segment_copies(x, seg_map, n_seg, pixel_major)
    return s1.sum(0), s2.sum(0), cnt.sum(0, dtype=torch.int32)


def _segment_loss(mode, x, seg_map, n_seg, pixel_major=False):
    """(loss [1] float, coef [n_seg] float, mean [n_seg, c] float or None) of gags_segment_loss: mode 0 = Scale_balance_loss,
    1 = scale_region_regulation_loss -- the copies' sum and all of the double-precision arithmetic on [n_seg] vectors in two
    launches."""
    s1c, s2c, cntc = _segment_copies(x, seg_map, n_seg, pixel_major)
    k, _, c = s1c.shape
    dev = x.device
    s1 = torch.empty(n_seg, c, dtype=torch.float64, device=dev)
    s2 = torch.empty_like(s1)
    cnt = torch.empty(n_seg, dtype=torch.int32, device=dev)
    loss = torch.empty(1, device=dev)
    coef = torch.empty(n_seg, device=dev)
    mean = torch.empty(n_seg, c, device=dev) if mode == 1 else None
    check(_lib.load().gags_segment_loss(mode, n_seg, c, k, seg_map.numel(), ptr(s1c), ptr(s2c), ptr(cntc), ptr(s1), ptr(s2),
                                        ptr(cnt), ptr(loss), ptr(coef), ptr(mean), _st()), "gags_segment_loss")
    return loss, coef, mean


class _ScaleBalance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loss_map, seg_map):
        lm, seg = _f(loss_map), _f(seg_map)
        n_seg = _n_seg(seg_map)
        loss, coef, _ = _segment_loss(0, lm.reshape(1, -1), seg.reshape(-1), n_seg)
        ctx.save_for_backward(seg, coef)
        ctx.n_seg = n_seg
        return loss[0]

    @staticmethod
    def backward(ctx, v):
        seg, coef = ctx.saved_tensors
        out = torch.empty_like(seg)
        check(_lib.load().gags_gather_seg_coef(seg.numel(), ptr(seg), ctx.n_seg, ptr((coef * v).contiguous()), ptr(out), _st()),
              "gags_gather_seg_coef")
        return out, None


def Scale_balance_loss(loss_map, seg_map, mask, scale_select_idx=1, mix_seg=False):
    """Mean over the segments present in seg_map [H,W] of the segment's mean of loss_map [H,W]
    (utils/loss_utils.py:32-57 with mix_seg=True, the only form train.py:167 uses; `mask` is unused there too)."""
    if not mix_seg:
        raise NotImplementedError("Scale_balance_loss: only mix_seg=True (train.py:167) is implemented")
    return _ScaleBalance.apply(loss_map, seg_map)


class _RegionVar(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seg_map):
        seg = _f(seg_map)
        c, hh, ww = x.shape
        # the rasterizer hands its [H,W,C] memory over as a [C,H,W] view: read it as it lies (pixel-major) instead of
        # copying 132 MB into channel-major order every iteration
        pm = x.is_cuda and x.dtype == torch.float32 and not x.is_contiguous() and x.permute(1, 2, 0).is_contiguous()
        x = x.permute(1, 2, 0) if pm else _f(x)
        n_seg = _n_seg(seg_map)
        # unbiased variances (torch.var) from moments accumulated in double about a member of each group (csrc/losses.hip): the
        # subtraction is a double-precision one on accurately summed terms; segments of 0 or 1 pixels are skipped
        # (loss_utils.py:124-125); copies' sum, variances, loss and the backward's tables in two launches (gags_segment_loss)
        loss, coef, mean = _segment_loss(1, x.reshape(-1, c) if pm else x.reshape(c, -1), seg.reshape(-1), n_seg, pixel_major=pm)
        ctx.save_for_backward(x, seg, mean, coef)
        ctx.n_seg, ctx.pm, ctx.c = n_seg, pm, c
        return loss[0]

    @staticmethod
    def backward(ctx, v):
        x, seg, mean, coef = ctx.saved_tensors
        vx = torch.empty_like(x)  # (pixel-major input: [H,W,C] memory, returned as the [C,H,W] view of it)
        check(_lib.load().gags_region_var_bwd_layout(seg.numel(), ctx.c, ptr(x), ptr(seg), ctx.n_seg, ptr(mean),
                                                     ptr((coef * v).contiguous()), ptr(vx), 1 if ctx.pm else 0, _st()),
              "gags_region_var_bwd_layout")
        return (vx.permute(2, 0, 1) if ctx.pm else vx), None


class _RegionVarTee(torch.autograd.Function):
    """_RegionVar that also hands its input on: (loss, x') with x' an alias of x for the NEXT consumer of the map.  The
    backward then receives that consumer's gradient together with the loss's cotangent and adds the two in the kernel that
    forms the loss's own gradient (gags_region_var_bwd_add) -- autograd's separate sum of two 132 MB maps per 1080p iteration
    is gone.  Same values: one fp32 addition per element either way."""

    @staticmethod
    def forward(ctx, x, seg_map):
        loss = _RegionVar.forward(ctx, x, seg_map)
        return loss, x.view_as(x)

    @staticmethod
    def backward(ctx, v, g_next):
        if g_next is None:
            return _RegionVar.backward(ctx, v)
        x, seg, mean, coef = ctx.saved_tensors
        gp = g_next.permute(1, 2, 0) if ctx.pm else g_next
        if not (ctx.pm and ctx.c % 4 == 0 and gp.is_contiguous() and gp.dtype == torch.float32):
            vx, _ = _RegionVar.backward(ctx, v)
            return vx + g_next, None
        vx = torch.empty_like(x)
        check(_lib.load().gags_region_var_bwd_add(seg.numel(), ctx.c, ptr(x), ptr(seg), ctx.n_seg, ptr(mean),
                                                  ptr((coef * v).contiguous()), ptr(gp), ptr(vx), _st()), "gags_region_var_bwd_add")
        return vx.permute(2, 0, 1), None


def scale_region_regulation_loss_tee(scale_map, seg_map):
    """(scale_region_regulation_loss(scale_map, seg_map, mix_seg=True), scale_map'): use scale_map' wherever the map is
    consumed afterwards (gags_amd/distill.py feeds it to CNN_decoder) and the two gradients are summed in one kernel."""
    return _RegionVarTee.apply(scale_map, seg_map)


def scale_region_regulation_loss(scale_map, seg_map, scale_bal_idx=1, mix_seg=False):
    """sum over segments of n_seg * mean_c(var_c) / (H*W) of the map [C,H,W] under seg_map [H,W]
    (utils/loss_utils.py:103-136 with mix_seg=True; train.py:153 feeds it the rasterized feature map)."""
    if not mix_seg:
        raise NotImplementedError("scale_region_regulation_loss: only mix_seg=True (train.py:153) is implemented")
    return _RegionVar.apply(scale_map, seg_map)


def get_trained_seg(seg_map, scale_map):
    """seg_map [4,H,W], scale_map [3,H,W] -> [H,W]: the segment id of the level the 5x5-smoothed scale map prefers."""
    seg, sc = _f(seg_map), _f(scale_map.detach())
    _, h, w = seg.shape
    out = torch.empty(h, w, device=seg.device)
    check(_lib.load().gags_trained_seg(h, w, ptr(seg), ptr(sc), ptr(out), _st()), "gags_trained_seg")
    out._gags_n_seg = _n_seg(seg_map)  # (every id of the result is an id of seg_map)
    return out


class _SamFeature(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img_embed, seg_map, scale_map):
        e, seg, sc = _f(img_embed), _f(seg_map), _f(scale_map)
        c, (_, h, w), (_, H, W) = e.shape[1], seg.shape, sc.shape
        feat = torch.empty(c, H, W, device=e.device)
        mask = torch.empty(H, W, device=e.device)
        check(_lib.load().gags_sam_clip_feature(c, H, W, h, w, e.shape[0], ptr(e), ptr(seg), ptr(sc), ptr(feat), ptr(mask),
                                                _st()), "gags_sam_clip_feature")
        ctx.save_for_backward(e, seg)
        ctx.dims = (c, H, W, h, w)
        ctx.mark_non_differentiable(mask)
        return feat, mask

    @staticmethod
    def backward(ctx, v_feat, _v_mask):
        e, seg = ctx.saved_tensors
        c, H, W, h, w = ctx.dims
        vs = torch.empty(3, H, W, device=e.device)
        check(_lib.load().gags_sam_clip_feature_bwd_scale(c, H, W, h, w, e.shape[0], ptr(e), ptr(seg), ptr(_f(v_feat)), ptr(vs),
                                                          _st()), "gags_sam_clip_feature_bwd_scale")
        return None, None, vs


def read_sam_clip_feature(img_embed, seg_map, scale_map, max_mode=False, median_mode=False, show_scale_map=False):
    """(feature_map [C,H,W], mask [1,H,W] bool) as scene/dataset_readers.py:54-121 in its default mode."""
    if max_mode or median_mode or show_scale_map:
        raise NotImplementedError("read_sam_clip_feature: only the default mode (train.py:162,165) is implemented")
    feat, mask = _SamFeature.apply(img_embed, seg_map, scale_map)
    return feat, (mask != 0)[None]


def _pixel_major(t):
    """True when the [C,H,W] tensor is a permuted view of contiguous [H,W,C] fp32 memory (the decoder's and the
    rasterizer's output layout)."""
    return (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and not t.is_contiguous()
            and t.permute(1, 2, 0).is_contiguous())


class _DistillL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, img_embed, seg_map, scale_map):
        e, seg, sc = _f(img_embed), _f(seg_map), _f(scale_map)
        c, (_, h, w), (_, H, W) = e.shape[1], seg.shape, sc.shape
        if tuple(pred.shape) != (c, H, W):
            raise ValueError(f"pred {tuple(pred.shape)} vs embeddings of width {c} and a {H}x{W} scale map")
        pm = _pixel_major(pred) and c % 4 == 0
        p = pred if pm else _f(pred)  # pixel-major memory is consumed as it is (layout 1): rows in, rows out
        l1 = torch.empty(H, W, device=p.device)
        mask = torch.empty(H, W, device=p.device)
        check(_lib.load().gags_distill_l1_map_fwd(c, H, W, h, w, e.shape[0], ptr(p), ptr(e), ptr(seg), ptr(sc), ptr(l1),
                                                  ptr(mask), 1 if pm else 0, _st()), "gags_distill_l1_map_fwd")
        ctx.save_for_backward(p, e, seg, sc)
        ctx.dims = (c, H, W, h, w)
        ctx.pm = pm
        ctx.mark_non_differentiable(mask)
        return l1, mask

    @staticmethod
    def backward(ctx, v_map, _v_mask):
        p, e, seg, sc = ctx.saved_tensors
        c, H, W, h, w = ctx.dims
        # the gradient in the prediction's own layout: [H,W,c] memory behind a [c,H,W] view when it is pixel-major
        vp = torch.empty(H, W, c, device=p.device).permute(2, 0, 1) if ctx.pm else torch.empty_like(p)
        vs = torch.empty(3, H, W, device=p.device)
        check(_lib.load().gags_distill_l1_map_bwd(c, H, W, h, w, e.shape[0], ptr(p), ptr(e), ptr(seg), ptr(sc), ptr(_f(v_map)),
                                                  ptr(vp), ptr(vs), 1 if ctx.pm else 0, _st()), "gags_distill_l1_map_bwd")
        return vp, None, None, vs


def distill_l1_map(pred, img_embed, seg_map, scale_map):
    """train.py:165-166 in one pass: (l1_loss_map(pred * mask, gt * mask) [H,W], mask [1,H,W] bool) with
    gt, mask = read_sam_clip_feature(img_embed, seg_map, scale_map); gradients reach `pred` and `scale_map`."""
    l1, mask = _DistillL1.apply(pred, img_embed, seg_map, scale_map)
    return l1, (mask != 0)[None]
