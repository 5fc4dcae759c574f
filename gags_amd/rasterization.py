"""`rasterization(...)`: the operator the reference imports from gsplat
(/root/reference/gaussian_renderer/__init__.py:17,56-70), re-implemented on hand-written
gfx950 kernels behind the C ABI of include/gags_raster.h.

Same names, argument meaning, return triple and `info` keys as the call the reference makes:
    render_colors [1,H,W,D'], render_alphas [1,H,W,1], info = rasterization(
        means=[N,3], quats=[N,4] wxyz, scales=[N,3], opacities=[N], colors=[N,D] | [N,K,3],
        viewmats=[1,4,4], Ks=[1,3,3], backgrounds=[1,D], width=, height=, packed=False,
        sh_degree=None|int, render_mode="RGB"|"D"|"ED"|"RGB+D"|"RGB+ED")
Defaults the reference relies on by not passing them are gsplat's: near_plane=0.01,
far_plane=1e10, radius_clip=0, eps2d=0.3, tile_size=16, rasterize_mode="classic".

Autograd is split in the same places gsplat splits it, so that `info["means2d"]` is a
non-leaf tensor callers can `retain_grad()` (gaussian_renderer/__init__.py:75-78):
    _Project (K1/K2)  ->  [_SH (K3)]  ->  _Rasterize (K4-K10)
When only `colors` requires grad -- the GAD flow, scene/gaussian_model.py:192-208 -- the
backward launches the colours-only kernel and nothing else.
PyTorch supplies device memory, streams and autograd plumbing; all arithmetic is in HIP.
"""
import ctypes
import os
import threading

import torch

from . import _lib, profiler
from ._lib import check, ptr

TILE = 16
MAX_ISECTS = 1 << 28  # GAGS_MAX_ISECTS of the C ABI (include/gags_raster.h: 32-bit offsets into the slot tables)
ZERO_FILL_MIN_ELEMS = 1 << 24  # (below that the second stream's hand-over costs more than the zeros)
GEOM_COMPACT_ROWS = True   # gags_raster_bwd_geom: per-slot rows numbered compactly (one prefix sum + a 4-byte readback)
CAP_MARGIN = 1.05
# Channel ranges of the range-staged backward of a by-view multi-GPU step (RasterContext.grad_range_channels): an int (uniform
# ranges) or a tuple of widths (multiples of 128) applied in order, the last one repeated / cut to cover D.
GRAD_RANGE_CHANNELS = 128
# Channels per launch of the rows kernel under the range-staged backward (whole ranges; see _backward_staged).  256 costs 0.15-
# 0.3 ms less on ONE GPU (the weight tiles travel from HBM twice instead of four times) but delivers the first range 0.6 ms
# later; with ~1 ms of xGMI time per 128-channel range the exchange, not the GPU, paces the step at N = 8, and it is kept busy
# earliest by one launch per range (DESIGN.md section 6 has the arithmetic; bench.py reports both).
GRAD_ROWS_GROUP = 128
PROW_MAX_BYTES = 24 << 30  # staged backward: partial rows beyond this are produced per 128-channel range (see _backward_staged)
# List trimming (RasterContext.trim_lists; _trim_lists): None = automatic, for views whose forward scratch would exceed
# TRIM_AUTO_BYTES; True / False force it on / off (GAGS_TRIM_LISTS=1 / 0)
TRIM_LISTS = {"1": True, "0": False}.get(os.environ.get("GAGS_TRIM_LISTS", ""), None)
TRIM_AUTO_BYTES = 48 << 30
# Persistent gradient buffer of the colours-only backward (_KeptGrad): default ON
KEEP_GRAD = os.environ.get("GAGS_KEEP_GRAD", "1") != "0"
KEEP_GRAD_MIN_ELEMS = 0   # (every shape: the small test scenes exercise the same path as C3)
KEEP_GRAD_SHAPES = 2      # buffers a context keeps alive at once (least recently used shape is dropped)
# Row map of the colours-only backward at FORWARD time (RasterContext.early_rowmap; _early_rowmap): default ON
EARLY_COUNT = os.environ.get("GAGS_EARLY_COUNT", "1") != "0"  # 0: the intersection count is read back after the prefix sum
EARLY_ROWMAP = os.environ.get("GAGS_EARLY_ROWMAP", "1") != "0"
# Default of RasterContext.capacity_mode (GAGS_CAPACITY_MODE=1; OFF otherwise): see RasterContext.
CAPACITY_MODE = os.environ.get("GAGS_CAPACITY_MODE", "0") == "1"


class RasterContext:
    """Everything `rasterization(...)` remembers or is told between calls -- SURVEY 8b asks for a boundary that is
    "re-entrant per stream, no global state": the C library has none, and the Python layer keeps its own in an object the
    caller may own.  `rasterization(..., context=ctx)` / `render(..., context=ctx)` use `ctx`; without one, the calling
    THREAD's default context (default_context()) is used, so two renderers in one process -- a training view inside a
    gradient-exchange block and an evaluation view, two threads -- never see each other's hooks or capacities.

    grad_range_hook / grad_rows_hook / grad_range_channels
        Multi-GPU by-view step (gags_amd/dist.py: OverlappedGradReducer): when grad_range_hook is set, the staged
        colours-only backward produces the feature gradient one channel range at a time and calls
            grad_range_hook(v_colors_alias [N,D], ch_begin, ch_end)
        right after the kernels of each range were enqueued, so that the exchange of one range overlaps the computation of
        the next (the alias shares storage with the tensor handed to autograd).  With grad_rows_hook also set, the backward
        first calls grad_rows_hook(mask uint8 [N]): mask[g] = 1 for every Gaussian that blended into a pixel of this view,
        i.e. the only rows of the gradient that can be non-zero ("gradients are sparse in rows").
        With grad_wire_hook set, the backward asks it before the reduce stage of every range,
            grad_wire_hook(ch_begin, ch_end) -> (pos int32 [N], wire fp32 [rows, ch_end - ch_begin]) or None,
        and the reduce kernel writes row pos[g] of `wire` for every Gaussian with pos[g] >= 0 next to the gradient itself
        (gags_raster_bwd_colors_staged_wire); grad_range_hook then gets `wire` as a fourth argument.  grad_range_channels: an
        int or a tuple of range widths (see GRAD_RANGE_CHANNELS).
    capacity_mode, cap_isects, cap_rows
        Capacity mode (OFF by default): the two counts a view produces on the device -- tile intersections, partial
        gradient rows -- are NOT waited for before the kernels that need them are launched.  Buffers are sized by a capacity
        remembered from earlier views of the same (N, width, height), the kernels take their ranges from device memory
        (isect_offsets' last entry, sentinel keys), and the counts are read from a second stream once everything is
        enqueued: the host still learns them (info["n_isects"] is exact, capacities are checked) but the queue never
        drains.  A count above its capacity -- nothing is written out of bounds -- re-runs that pass with exact sizes; the
        first view of a shape runs the exact path.  Measured A/B on one box (C3, D = 512 / D = 16): 10.76-10.94 ms
        against 10.62-10.68 ms, 2.59 against 2.51-2.54 ms with a 25 % margin -- the two readbacks cost the GPU ~40 us of
        idle queue per step (the host is far ahead of the device), the sentinel keys cost more in the two sorts.  Kept as a
        switch for callers whose host is the bottleneck.
    overlap_zero_fill
        Experiment, OFF: zero-fill the colour gradient on a second stream during the forward's binning and let the reduce
        stage write only the rows that exist (stage bit 128; 73 % of the Gaussians blend nothing at C3).  Measured at C3:
        reduce 1.27 -> 0.95 ms, but the fill kernel takes the CUs from whatever it runs beside -- under the rows kernel that
        kernel slowed by 0.38 ms, under the binning kernels the step grew by 1.7 ms.  The C-ABI flag stays for callers
        that own a zeroed buffer anyway.
    early_rowmap
        ON by default (GAGS_EARLY_ROWMAP=0): the staged backward begins by numbering the view's partial gradient rows (a prefix
        sum over the forward's hit flags, gags_bwd_rowmap) and has to know their COUNT on the host before it can size its
        scratch -- a 4-byte readback in the middle of the backward, with the launch queue drained around it (60-90 us).  None of
        that depends on the cotangent: a forward that will be differentiated w.r.t. the colours alone enqueues the row map right
        behind its own kernels and sends the count to pinned memory; by the time the loss has been computed it has long arrived
        and the backward opens with the rows kernel.  A forward that is never differentiated has run 0.12 ms of small kernels for nothing.
    Also here: the pinned 4-byte buffers of the deferred count readbacks, the side streams, and render()'s cache of the
    intrinsics matrix."""

    def __init__(self, capacity_mode=None, overlap_zero_fill=False):
        self.grad_range_hook = None
        self.grad_rows_hook = None
        self.grad_wire_hook = None
        self.grad_range_channels = GRAD_RANGE_CHANNELS
        self.grad_rows_group = GRAD_ROWS_GROUP
        self.capacity_mode = CAPACITY_MODE if capacity_mode is None else bool(capacity_mode)
        self.cap_isects = {}
        self.cap_rows = {}
        self.overlap_zero_fill = bool(overlap_zero_fill)
        # the backward's row map enqueued behind the forward, its count on the way to the host meanwhile (_early_rowmap)
        self.early_rowmap = EARLY_ROWMAP
        self._pinned_pool = {}
        self.k_cache = {}
        self._pinned = {}
        self._side = {}
        # list trimming for heavy views (_trim_lists): None = automatic above TRIM_AUTO_BYTES of forward scratch
        self.trim_lists = TRIM_LISTS
        # persistent gradient buffer of the colours-only backward (_KeptGrad): on unless GAGS_KEEP_GRAD=0
        self.keep_grad_buffer = KEEP_GRAD
        self._kept = {}        # (n, d, dtype, device) -> _KeptGrad, at most KEEP_GRAD_SHAPES shapes (least recently used out)
        self._kept_fails = {}  # consecutive steps that found the buffer still referenced / written

    def forget_kept(self, n, d, dtype, dev):
        self._kept.pop((n, d, dtype, dev.index), None)

    def forget_all_kept(self):
        self._kept.clear()

    def kept_grad(self, n, d, dtype, dev):
        """(alias, flags_prev, flags_cur) of this shape's persistent gradient buffer, or None when it is not to be used."""
        if not self.keep_grad_buffer or not _CAN_COUNT_REFS or n * d < KEEP_GRAD_MIN_ELEMS or n == 0:
            return None
        key = (n, d, dtype, dev.index)
        if self._kept_fails.get(key, 0) >= 3:
            return None
        ent = self._kept.pop(key, None)
        if ent is not None and not ent.untouched():
            self._kept_fails[key] = self._kept_fails.get(key, 0) + 1  # the holder keeps that storage; a new one below
            ent = None
            if self._kept_fails[key] >= 3:
                return None
        elif ent is not None:
            self._kept_fails[key] = 0
        if ent is None:
            ent = _KeptGrad(n, d, dtype, dev)
        self._kept[key] = ent  # (re-inserted last: most recently used)
        while len(self._kept) > KEEP_GRAD_SHAPES:
            self._kept.pop(next(iter(self._kept)))
        return ent.hand_out()

    def side_stream(self, dev):
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=dev)
        return self._side[key]

    def take_pinned(self, dev):
        """A pinned int32 of the caller's own (returned with give_pinned): counts of several forwards may be in flight."""
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        pool = self._pinned_pool.setdefault(key, [])
        return pool.pop() if pool else torch.empty(1, dtype=torch.int32).pin_memory()

    def give_pinned(self, dev, t):
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        pool = self._pinned_pool.setdefault(key, [])
        if len(pool) < 8:
            pool.append(t)

    def pinned_i64(self, dev):
        key = ("i64", dev.index if dev.index is not None else torch.cuda.current_device())
        if key not in self._pinned:
            self._pinned[key] = torch.empty(1, dtype=torch.int64).pin_memory()
        return self._pinned[key]

    def pinned_i32(self, dev):
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        if key not in self._pinned:
            self._pinned[key] = torch.empty(1, dtype=torch.int32).pin_memory()
        return self._pinned[key]


_TLS = threading.local()


def default_context():
    """The calling thread's own RasterContext (created on first use)."""
    ctx = getattr(_TLS, "ctx", None)
    if ctx is None:
        ctx = _TLS.ctx = RasterContext()
    return ctx


def _storage_refs(t):
    """References to `t`'s storage (the temporary made here included: only compared with a baseline taken the same way)."""
    return torch._C._storage_Use_Count(t.untyped_storage()._cdata)


# (a private torch binding: without it nobody can tell whether a buffer is still held elsewhere, and the mechanism stays off)
_CAN_COUNT_REFS = hasattr(torch._C, "_storage_Use_Count")


class _KeptGrad:
    """The persistent gradient buffer of the colours-only backward (RasterContext.keep_grad_buffer).  d loss / d colors is a
    dense [N, D] tensor of which 73 % of the rows are zero at C3 (Gaussians that blend nothing); written afresh every step
    those zeros are 2.2 GB of HBM writes.  Here the context keeps ONE zero-initialised buffer per (N, D, dtype), every
    backward hands autograd a fresh alias of it (own TensorImpl, same storage: autograd adopts it without a copy) and the
    reduce stage only writes the rows that have partial rows now and re-zeroes the rows that had some in the previous step
    (two flag arrays, gags_raster_bwd_colors_staged_keep).  The buffer is only reused when NOBODY else still refers to its
    storage (reference count back at its baseline: the previous step's .grad was released, e.g. zero_grad(set_to_none=True)
    or `.grad = None`) and nobody wrote to it in place through torch (version counter unchanged; the kernels write through raw
    pointers and never bump it).  Otherwise the holder keeps the old storage and this step runs on a new buffer; after three
    such steps in a row the mechanism switches itself off for the shape (a loop that accumulates into a live .grad)."""

    def __init__(self, n, d, dtype, dev):
        self.buf = torch.zeros(n, d, dtype=dtype, device=dev)
        self.flags = torch.zeros(2, max(n, 1), dtype=torch.uint8, device=dev)
        self.cur = 0
        self.base = _storage_refs(self.buf)
        self.version = self.buf._version

    def untouched(self):
        return _storage_refs(self.buf) <= self.base and self.buf._version == self.version

    def hand_out(self):
        prev, cur = self.flags[self.cur], self.flags[1 - self.cur]
        self.cur ^= 1
        return self.buf.detach(), prev, cur


class _DeferredCount:
    """A device-side int32 read back without draining the launch stream: the copy runs on a second stream behind an
    event recorded right after the kernel that produced the value."""

    def __init__(self, scalar, rctx):
        self.t = scalar
        self.rctx = rctx
        self.ev = torch.cuda.Event()
        self.ev.record()

    def get(self):
        dev = self.t.device
        side = self.rctx.side_stream(dev)
        host = self.rctx.pinned_i32(dev)
        done = torch.cuda.Event()
        with torch.cuda.stream(side):
            side.wait_event(self.ev)
            host.copy_(self.t, non_blocking=True)
            done.record()
        self.t.record_stream(side)
        done.synchronize()
        return int(host[0])


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("gags_amd.rasterization: all tensors must live on the GPU "
                               "(there is no CPU path; see oracle/ for the test-only CPU restatement)")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:  # kernels are launched on the current device's stream
            raise RuntimeError(f"gags_amd.rasterization: tensor on cuda:{t.device.index} but the current device is "
                               f"cuda:{cur}; one process per GPU, torch.cuda.set_device(local_rank) first")


def _c(t):
    return t if (t.is_contiguous() and t.dtype == torch.float32) else t.contiguous().float()


class _Project(torch.autograd.Function):
    """K1 + K4 forward, K2 backward."""

    @staticmethod
    def forward(ctx, means, quats, scales, viewmat, K, width, height, eps2d, near, far, radius_clip):
        lib = _lib.load()
        means, quats, scales, viewmat, K = _c(means), _c(quats), _c(scales), _c(viewmat), _c(K)
        n = means.shape[0]
        dev = means.device
        radii = torch.empty(n, dtype=torch.int32, device=dev)
        means2d = torch.empty(n, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(n, dtype=torch.float32, device=dev)
        conics = torch.empty(n, 3, dtype=torch.float32, device=dev)
        tiles = torch.empty(n, dtype=torch.int32, device=dev)
        with profiler.stage("project_fwd"):
            check(lib.gags_project_fwd(n, ptr(means), ptr(quats), ptr(scales), ptr(viewmat), ptr(K), width, height,
                                       eps2d, near, far, radius_clip, ptr(radii), ptr(means2d), ptr(depths),
                                       ptr(conics), ptr(tiles), _stream()), "gags_project_fwd")
        ctx.save_for_backward(means, quats, scales, viewmat, K, radii, conics)
        ctx.cfg = (width, height, eps2d)
        ctx.mark_non_differentiable(radii, tiles)
        return radii, means2d, depths, conics, tiles

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics, _v_tiles):
        lib = _lib.load()
        means, quats, scales, viewmat, K, radii, conics = ctx.saved_tensors
        width, height, eps2d = ctx.cfg
        n = means.shape[0]
        dev = means.device
        v_means2d = torch.zeros(n, 2, device=dev) if v_means2d is None else _c(v_means2d)
        v_conics = torch.zeros(n, 3, device=dev) if v_conics is None else _c(v_conics)
        v_depths = None if v_depths is None else _c(v_depths)
        v_means = torch.empty(n, 3, device=dev)
        v_quats = torch.empty(n, 4, device=dev)
        v_scales = torch.empty(n, 3, device=dev)
        check(lib.gags_project_bwd(n, ptr(means), ptr(quats), ptr(scales), ptr(viewmat), ptr(K), width, height,
                                   eps2d, ptr(radii), ptr(conics), ptr(v_means2d), ptr(v_depths), ptr(v_conics),
                                   ptr(v_means), ptr(v_quats), ptr(v_scales), _stream()), "gags_project_bwd")
        return v_means, v_quats, v_scales, None, None, None, None, None, None, None, None


class _ProjectRaw(torch.autograd.Function):
    """K1 + K4 on the STORED parameters (gags_project_fwd_raw / gags_project_bwd_raw): the getters of
    scene/gaussian_model.py:116-139 -- exp, F.normalize, sigmoid -- and render()'s `* scaling_modifier` run inside the
    projection kernel, bit for bit as torch evaluates them; the backward returns gradients of the stored parameters."""

    @staticmethod
    def forward(ctx, means, rotation, scaling_log, opacity_logit, viewmat, K, width, height, eps2d, near, far, radius_clip,
                scaling_modifier, want_records):
        lib = _lib.load()
        means, rotation, scaling_log, viewmat, K = _c(means), _c(rotation), _c(scaling_log), _c(viewmat), _c(K)
        logit = _c(opacity_logit.reshape(-1))
        n = means.shape[0]
        dev = means.device
        radii = torch.empty(n, dtype=torch.int32, device=dev)
        means2d = torch.empty(n, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(n, dtype=torch.float32, device=dev)
        conics = torch.empty(n, 3, dtype=torch.float32, device=dev)
        tiles = torch.empty(n, dtype=torch.int32, device=dev)
        opac = torch.empty(n, dtype=torch.float32, device=dev)
        # the per-Gaussian records of the matrix-core raster kernels (K8b), written on the way: no record kernel in the binning
        grec = torch.empty(max(n, 1), 8, dtype=torch.float32, device=dev) if want_records else torch.empty(0, device=dev)
        with profiler.stage("project_fwd"):
            check(lib.gags_project_fwd_raw(n, ptr(means), ptr(rotation), ptr(scaling_log), ptr(logit), scaling_modifier,
                                           ptr(viewmat), ptr(K), width, height, eps2d, near, far, radius_clip, ptr(radii),
                                           ptr(means2d), ptr(depths), ptr(conics), ptr(tiles), ptr(opac), None, None,
                                           ptr(grec) if want_records else None, _stream()), "gags_project_fwd_raw")
        ctx.save_for_backward(means, rotation, scaling_log, logit, viewmat, K, radii)
        ctx.cfg = (width, height, eps2d, scaling_modifier, tuple(opacity_logit.shape))
        ctx.mark_non_differentiable(radii, tiles, grec)
        return radii, means2d, depths, conics, tiles, opac, grec

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics, _v_tiles, v_opac, _v_grec):
        lib = _lib.load()
        means, rotation, scaling_log, logit, viewmat, K, radii = ctx.saved_tensors
        width, height, eps2d, modifier, oshape = ctx.cfg
        n = means.shape[0]
        dev = means.device
        v_means2d = torch.zeros(n, 2, device=dev) if v_means2d is None else _c(v_means2d)
        v_conics = torch.zeros(n, 3, device=dev) if v_conics is None else _c(v_conics)
        v_depths = None if v_depths is None else _c(v_depths)
        v_opac = None if v_opac is None else _c(v_opac)
        v_means = torch.empty(n, 3, device=dev)
        v_rot = torch.empty(n, 4, device=dev)
        v_scal = torch.empty(n, 3, device=dev)
        v_logit = torch.empty(n, device=dev) if ctx.needs_input_grad[3] else None
        check(lib.gags_project_bwd_raw(n, ptr(means), ptr(rotation), ptr(scaling_log), ptr(logit), modifier, ptr(viewmat),
                                       ptr(K), width, height, eps2d, ptr(radii), ptr(v_means2d), ptr(v_depths),
                                       ptr(v_conics), ptr(v_opac), ptr(v_means), ptr(v_rot), ptr(v_scal), ptr(v_logit),
                                       _stream()), "gags_project_bwd_raw")
        return (v_means, v_rot, v_scal, None if v_logit is None else v_logit.reshape(oshape), None, None, None, None, None,
                None, None, None, None, None)


class _SH(torch.autograd.Function):
    """K3: SH colour and its backward: coefficients, and -- when positions are trainable -- the view direction
    (d colour / d means through normalize(mean - campos), as gsplat propagates it)."""

    @staticmethod
    def forward(ctx, coeffs, means, campos, radii, degree):
        lib = _lib.load()
        coeffs, means, campos = _c(coeffs), _c(means), _c(campos)
        n, kc = coeffs.shape[0], coeffs.shape[1]
        out = torch.empty(n, 3, device=coeffs.device)
        check(lib.gags_sh_fwd(n, kc, degree, ptr(means), ptr(campos), ptr(coeffs), ptr(radii), ptr(out), _stream()),
              "gags_sh_fwd")
        ctx.save_for_backward(means, campos, radii, out, coeffs if ctx.needs_input_grad[1] else None)
        ctx.cfg = (kc, degree)
        return out

    @staticmethod
    def backward(ctx, v_out):
        lib = _lib.load()
        means, campos, radii, out, coeffs = ctx.saved_tensors
        kc, degree = ctx.cfg
        n = means.shape[0]
        v_out = _c(v_out)
        v_coeffs = v_means = None
        if ctx.needs_input_grad[0]:
            v_coeffs = torch.empty(n, kc, 3, device=means.device)
            check(lib.gags_sh_bwd(n, kc, degree, ptr(means), ptr(campos), ptr(radii), ptr(out), ptr(v_out),
                                  ptr(v_coeffs), _stream()), "gags_sh_bwd")
        if ctx.needs_input_grad[1]:
            v_means = torch.empty(n, 3, device=means.device)
            check(lib.gags_sh_bwd_dirs(n, kc, degree, ptr(means), ptr(campos), ptr(coeffs), ptr(radii), ptr(out),
                                       ptr(v_out), ptr(v_means), _stream()), "gags_sh_bwd_dirs")
        return v_coeffs, v_means, None, None, None


def tile_binning(means2d, radii, depths, tiles_per_gauss, width, height, conics=None, opacities=None, cap=None, records=None,
                 context=None):
    """K5-K8 (+K8b) on device.  Returns (isect_ids sorted int64, flatten_ids sorted int32, isect_offsets [th,tw] int32,
    n_isects, packed [N,8] per-Gaussian records or None, offsets_full = the buffer behind isect_offsets: th * tw + 1 entries
    whose last one is the count).
    cap None: one host readback (n_isects), as in gsplat; the id arrays have exactly n_isects entries.
    cap = capacity: nothing is read back here; the id arrays have `cap` entries (sentinels past the count) and n_isects is a
    _DeferredCount the caller resolves after enqueuing what follows."""
    lib = _lib.load()
    st = _stream()
    dev = means2d.device
    n = radii.shape[0]
    tile_w, tile_h = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    n_tiles = tile_w * tile_h
    tile_bits = max(1, n_tiles.bit_length())  # (room for the sentinel tile id n_tiles of the capacity mode)
    cum = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    early = None
    if cap is None and EARLY_COUNT and n > 0:
        # the count does not depend on the order: summed and sent to the host BEFORE the depth sort and the prefix sum are
        # enqueued, so that the readback's round trip and the host work behind it (allocations, the next launches) run under
        # those ~0.2 ms of kernels instead of after them with the queue empty (round 6: a 60-200 us bubble per view)
        host64 = (context or default_context()).pinned_i64(dev)
        host64.copy_(tiles_per_gauss.sum().reshape(1), non_blocking=True)
        early = torch.cuda.Event()
        early.record()
    # Gaussians in depth order first (N keys), intersections emitted in that order: the per-intersection sort
    # (n_isects ~ 5 N keys) then only groups by tile -- 2 radix passes instead of 6, same sorted result
    order = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    dsb = lib.gags_depth_order_scratch_bytes(n)
    dscratch = torch.empty(dsb, dtype=torch.uint8, device=dev)
    check(lib.gags_depth_order(n, ptr(depths), None, ptr(order), None, ptr(dscratch), dsb, st), "gags_depth_order")
    sb = lib.gags_scan_scratch_bytes(n)
    scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
    # prefix sum of the tile counts IN DEPTH ORDER, read through `order` (no permuted copy)
    check(lib.gags_cumsum_gather_i32(n, ptr(tiles_per_gauss), ptr(order), ptr(cum), ptr(total), ptr(scratch), sb, st),
          "gags_cumsum_gather_i32")
    if cap is None:
        if early is not None:
            early.synchronize()
            n_isects = int(host64[0])
        else:
            host = ctypes.c_int32(0)
            check(lib.gags_read_i32(ptr(total), ctypes.byref(host), st), "gags_read_i32")
            n_isects = int(host.value)
        _check_isects(n_isects, n_tiles)
        size, count = n_isects, n_isects
    else:
        size, count = int(cap), _DeferredCount(total, context or default_context())
    offsets = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
    ids = torch.empty(max(size, 1), dtype=torch.int64, device=dev)
    flat = torch.empty(max(size, 1), dtype=torch.int32, device=dev)
    ids_s = torch.empty_like(ids)
    flat_s = torch.empty_like(flat)
    if size > 0:
        check(lib.gags_tile_emit_cap(n, ptr(means2d), ptr(radii), ptr(depths), ptr(cum), ptr(order), tile_w, tile_h,
                                     ptr(ids), ptr(flat), size, ptr(total) if cap is not None else None, st),
              "gags_tile_emit_cap")
        ssb = lib.gags_sort_scratch_bytes(size)
        sscratch = torch.empty(ssb, dtype=torch.uint8, device=dev)
        check(lib.gags_sort_pairs(size, tile_bits, 1, ptr(ids), ptr(flat), ptr(ids_s), ptr(flat_s), ptr(sscratch),
                                  ssb, st), "gags_sort_pairs")
    check(lib.gags_tile_offsets(size, ptr(ids_s), n_tiles, ptr(offsets), st), "gags_tile_offsets")
    packed = records  # (gags_project_fwd_raw already wrote the per-Gaussian table)
    if conics is not None and packed is None:
        # one 32-byte record per GAUSSIAN; the raster kernels gather it through flatten_ids themselves
        # (GAGS_RECS_BY_GAUSSIAN): no per-intersection copy of the records, no gather kernel
        packed = torch.empty(max(n, 1), 8, dtype=torch.float32, device=dev)
        check(lib.gags_pack_isects(n, size, ptr(flat_s), ptr(means2d), ptr(conics), ptr(opacities), ptr(radii),
                                   ptr(packed), None, st), "gags_pack_isects")
    # `offsets` has n_tiles + 1 entries, the last one = the intersection count (gags_tile_offsets): the raster kernels read
    # isect_offsets[tile + 1] as a tile's end.  Callers get gsplat's [tile_h, tile_w] view AND the buffer itself.
    off_view = offsets[:n_tiles].view(tile_h, tile_w)
    return ids_s[:size], flat_s[:size], off_view, count, packed, offsets


def _trim_lists(lib, n, width, height, offsets_full, flatten_ids, n_isects, packed):
    """Round 6, heavy views: every tile's sorted list cut to the entries its pixels actually read before the tile is done
    (gags_raster_list_need: the weights pass's own walk, outputs dropped), so that the split forward's scratch -- ~1 KB per
    list entry -- and every pass over the lists shrink by the same factor (C5H: 169 M entries, a few hundred of a tile's 20 k
    are read).  Returns (offsets [n_tiles + 1] with the count, flatten_ids, count) of the trimmed lists: on them every raster
    entry computes bit for bit what it computes on the full lists; only last_ids index the trimmed list
    (gags_trim_last_ids translates a copy back).  One more 4-byte readback (the trimmed count sizes the id list)."""
    st = _stream()
    dev = flatten_ids.device
    n_tiles = ((width + TILE - 1) // TILE) * ((height + TILE - 1) // TILE)
    need = torch.empty(n_tiles, dtype=torch.int32, device=dev)
    with profiler.stage("list_need"):
        check(lib.gags_raster_list_need(n, width, height, ptr(offsets_full), ptr(flatten_ids), n_isects, ptr(packed),
                                        _lib.GAGS_RECS_BY_GAUSSIAN, ptr(need), st), "gags_raster_list_need")
    cum = torch.empty(n_tiles, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    sb = lib.gags_scan_scratch_bytes(n_tiles)
    scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
    check(lib.gags_cumsum_i32(n_tiles, ptr(need), ptr(cum), ptr(total), ptr(scratch), sb, st), "gags_cumsum_i32")
    host = ctypes.c_int32(0)
    check(lib.gags_read_i32(ptr(total), ctypes.byref(host), st), "gags_read_i32")
    t = int(host.value)
    offs_t = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
    flat_t = torch.empty(t, dtype=torch.int32, device=dev)
    check(lib.gags_trim_lists(width, height, ptr(offsets_full), ptr(cum), ptr(flatten_ids), ptr(offs_t),
                              ptr(flat_t) if t > 0 else None, st), "gags_trim_lists")
    return offs_t, flat_t, t


def _check_isects(n_isects, n_tiles=0):
    # the int32 prefix sum wrapped, or the slot space (4 I + 64 tiles + 64 slots, 32-bit byte offsets of its id table) would
    if n_isects < 0 or n_isects >= MAX_ISECTS or 4 * n_isects + 64 * n_tiles + 64 >= (1 << 30):
        raise RuntimeError(f"gags_amd.rasterization: {n_isects if n_isects >= 0 else '> 2^31'} tile intersections in one "
                           f"view; the kernels index at most 2^28 = {MAX_ISECTS} less 16 per tile (INTEGRATION.md, memory model)")


def _mfma_width(d):
    """Feature widths served by the split matrix-core path (gags_mfma_width in csrc/common.h): every D >= 16 -- 16 is
    what the reference rasterizes (train.py:68), 513 = 512 + 1 is BASELINE.json configs[4]."""
    return d >= 16


class _Rasterize(torch.autograd.Function):
    """K9 forward / K10 backward over pre-binned intersections.

    Matrix-core widths (D >= 16, D % 4 == 0) run the split forward: one weights pass (alpha, transmittance,
    stop rule: once per view) that leaves weight tiles in a scratch buffer, then the feature stream.
    The same scratch feeds the staged, atomic-free colours-only backward."""

    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, packed, width, height,
                flags, prezero=None, rctx=None, n_isects=None, grad_on=True):
        """offsets: the buffer of n_tiles + 1 int32 entries tile_binning returns as `offsets_full` (last entry = the
        intersection count), or a gsplat-style [tile_h, tile_w] tensor together with `n_isects` (the count is then attached
        here).  rctx: the RasterContext of the call."""
        lib = _lib.load()
        # (ctx.needs_input_grad says what requires grad, not whether a graph is being recorded: under torch.no_grad() it is
        # still True -- `grad_on` is torch.is_grad_enabled() at the call, taken by rasterization() outside this forward)
        needs = tuple(bool(g) and bool(grad_on) for g in ctx.needs_input_grad)
        means2d, conics, opacities = _c(means2d), _c(conics), _c(opacities)
        # an fp16 feature table (BASELINE.json configs[4]) is read as it is by the matrix-core feature pass: widened
        # exactly, same fp32 arithmetic; every other kernel gets fp32
        half = (colors.dtype == torch.float16 and packed is not None and colors.shape[0] > 0
                and not (flags & (_lib.GAGS_FWD_NO_MFMA | _lib.GAGS_FWD_FUSED)))
        colors = (colors if colors.is_contiguous() else colors.contiguous()) if half else _c(colors)
        backgrounds = None if backgrounds is None else _c(backgrounds)
        n, d = colors.shape
        dev = colors.device
        n_tiles = ((width + TILE - 1) // TILE) * ((height + TILE - 1) // TILE)
        offsets = _offsets_with_count(offsets, n_tiles, flatten_ids.shape[0] if n_isects is None else n_isects)
        n_isects = flatten_ids.shape[0]  # (capacity mode: the buffers' size; the kernels take the count from `offsets`)
        out = torch.empty(height, width, d, device=dev)
        alphas = torch.empty(height, width, device=dev)
        last_ids = torch.empty(height, width, dtype=torch.int32, device=dev)
        split = n > 0 and packed is not None and not (flags & (_lib.GAGS_FWD_NO_MFMA | _lib.GAGS_FWD_FUSED))
        scratch = blk_rows = None
        nbytes = 0
        # a 16-channel fp32 render that nothing will be differentiated through (evaluation: render.py, the relevancy queries):
        # the fused weights + feature pass alone (csrc/raster_weights.hip) -- no weight tiles, no 1 KB per intersection of scratch
        lean16 = (split and d == 16 and not half and n_isects > 0 and not any(needs)
                  and not (flags & _lib.GAGS_FWD_EXACT) and not profiler.ENABLED)
        if split and not lean16:
            nbytes = lib.gags_raster_fwd_scratch_bytes(n_isects, width, height)
            try:
                scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            except torch.OutOfMemoryError:
                # slot space (1 KB per intersection) does not fit even after the caching allocator gave its blocks
                # back: scratch-free kernels, which read an fp32 table only.  Said out loud: they are several times slower
                # (single-kernel forward, atomic backward) and the caller should know why
                import warnings
                warnings.warn(f"gags_amd.rasterization: {nbytes / 2 ** 30:.1f} GiB of forward scratch for {n_isects} tile "
                              "intersections could not be allocated; this view runs on the scratch-free kernels "
                              "(INTEGRATION.md, memory model)", RuntimeWarning, stacklevel=3)
                split, nbytes, scratch = False, 0, None
                if half:
                    half, colors = False, colors.float()
        if lean16:
            split = False
        if split:
            blk_rows = torch.empty(n_tiles * 4, dtype=torch.int32, device=dev)  # per 8x8 pixel block
        cflags = ((flags & 3) | (flags & _lib.GAGS_FWD_EXACT) | _lib.GAGS_RECS_BY_GAUSSIAN | (_lib.GAGS_FEAT_F16 if half else 0)
                  | (64 if (half and d >= 128 and (flags & _lib.GAGS_FWD_F16MFMA)) else 0))

        def launch(extra=0):
            check(lib.gags_raster_fwd(d, n, width, height, ptr(means2d), ptr(conics), ptr(opacities), ptr(colors),
                                      ptr(backgrounds), ptr(offsets), ptr(flatten_ids), n_isects, ptr(packed),
                                      ptr(out), ptr(alphas), ptr(last_ids), ptr(scratch), nbytes, ptr(blk_rows),
                                      cflags | extra, _stream()), "gags_raster_fwd")

        with profiler.stage("raster_fwd"):
            if profiler.ENABLED and split and n_isects > 0:  # one event pair per kernel, for bench.py's roofline line
                with profiler.stage("raster_weights"):
                    launch(_lib.GAGS_FWD_ONLY_WEIGHTS)
                with profiler.stage("raster_fwd_feat"):
                    launch(_lib.GAGS_FWD_ONLY_FEATURES)
            else:
                launch()
        if split:
            profiler.note("fwd_blk_rows", blk_rows)
        need_geom = needs[0] or needs[1] or needs[3]
        rctx_ = rctx if rctx is not None else default_context()
        early = None
        if (split and n_isects > 0 and rctx_.early_rowmap and needs[2] and not need_geom and _mfma_width(d)
                and d <= 1024 and not (flags & _lib.GAGS_BWD_ATOMIC) and not rctx_.capacity_mode):
            with profiler.stage("bwd_rowcount"):
                early = _early_rowmap(lib, rctx_, offsets, blk_rows, scratch, n_isects, width, height, dev)
        # wide-D geometry gradients on the matrix cores (gags_raster_bwd_geom) also consume the forward's scratch
        geom_mfma = split and need_geom and _geom_mfma_width(d) and not (flags & _lib.GAGS_BWD_ATOMIC)
        staged = (split and _mfma_width(d) and d <= 1024 and (needs[2] or geom_mfma)
                  and not (flags & _lib.GAGS_BWD_ATOMIC))
        ctx.geom_mfma = bool(geom_mfma and staged)
        ctx.save_for_backward(means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, packed, alphas,
                              last_ids, scratch if staged else None, blk_rows if staged else None)
        ctx.cfg = (width, height, flags)
        ctx.half = half
        ctx.rctx = rctx if rctx is not None else default_context()
        ctx.prezero = prezero if staged else None
        ctx.early = early if staged else None
        ctx.mark_non_differentiable(last_ids)
        return out, alphas, last_ids

    @staticmethod
    def backward(ctx, v_out, v_alphas, _v_last):
        lib = _lib.load()
        (means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, packed, alphas,
         last_ids, fwd_scratch, blk_rows) = ctx.saved_tensors
        width, height, flags = ctx.cfg
        n, d = colors.shape
        dev = colors.device
        n_isects = flatten_ids.shape[0]
        need_geom = ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[3]
        v_out = torch.zeros(height, width, d, device=dev) if v_out is None else _c(v_out)
        v_alphas = None if v_alphas is None else _c(v_alphas)
        v_bg = None
        if backgrounds is not None and ctx.needs_input_grad[4]:
            v_bg = ((1.0 - alphas)[..., None] * v_out).sum(dim=(0, 1))
        if not need_geom and blk_rows is not None:
            # an fp16 table gets its gradient in fp16 straight from the reduce kernel (fp32 sums, rounded once): no fp32
            # tensor + cast pass (autograd wants the table's dtype; an fp32 master sits behind a .half() cast)
            v_colors = _backward_staged(lib, ctx.rctx, offsets, n_isects, blk_rows, fwd_scratch, v_out, n, d, width, height,
                                        (32 if (flags & _lib.GAGS_BWD_F32MFMA) else 0) | (64 if ctx.half else 0) | (512 if (flags & _lib.GAGS_BWD_BLOCKWAVES) else 0) | (1024 if (flags & _lib.GAGS_BWD_EXACT_WEIGHTS) else 0),
                                        flatten_ids, ctx.prezero, early=ctx.early)
            return None, None, v_colors, None, v_bg, None, None, None, None, None, None, None, None, None, None
        if need_geom and blk_rows is not None and ctx.geom_mfma:
            # wide D: colours through the staged backward, geometry through the matrix-core dot pass + scalar pass
            v_colors = None
            if ctx.needs_input_grad[2]:
                v_colors = _backward_staged(lib, ctx.rctx, offsets, n_isects, blk_rows, fwd_scratch, v_out, n, d, width, height,
                                            (32 if (flags & _lib.GAGS_BWD_F32MFMA) else 0) | (64 if ctx.half else 0) | (512 if (flags & _lib.GAGS_BWD_BLOCKWAVES) else 0) | (1024 if (flags & _lib.GAGS_BWD_EXACT_WEIGHTS) else 0),
                                            flatten_ids)
            if ctx.half:  # the geometry kernels read an fp32 table: widen the halves (exact) for this backward
                colors = colors.float()
            # compact numbering of the per-slot rows: one small prefix sum and a 4-byte readback instead of sorting the
            # whole sparse slot space (6x the keys)
            if GEOM_COMPACT_ROWS:
                incl = torch.cumsum(blk_rows, 0, dtype=torch.int32)
                row_base = (incl - blk_rows).contiguous()
                n_rows = int(incl[-1].item()) if incl.numel() else 0
            else:  # the C ABI's other numbering: rows (and the dot products) in the sparse slot space, no count needed
                row_base, n_rows = None, -1
            nb = lib.gags_raster_bwd_geom_scratch_bytes(n_isects, width, height, n, d, n_rows)
            gscratch = torch.empty(nb, dtype=torch.uint8, device=dev)
            v_geo = torch.empty(n, 8, device=dev)
            with profiler.stage("raster_bwd_geom"):
                check(lib.gags_raster_bwd_geom(d, n, width, height, ptr(colors), ptr(backgrounds), ptr(offsets), n_isects,
                                               ptr(packed), ptr(v_out), ptr(v_alphas), ptr(blk_rows), ptr(fwd_scratch),
                                               fwd_scratch.numel(), ptr(gscratch), nb, ptr(v_geo), ptr(flatten_ids),
                                               ptr(row_base), n_rows,
                                               _lib.GAGS_RECS_BY_GAUSSIAN | (32 if (flags & _lib.GAGS_BWD_F32MFMA) else 0),
                                               _stream()),
                      "gags_raster_bwd_geom")
            v_con, v_m2d, v_opac = v_geo[:, 0:3].contiguous(), v_geo[:, 3:5].contiguous(), v_geo[:, 5].contiguous()
            return v_m2d, v_con, v_colors, v_opac, v_bg, None, None, None, None, None, None, None, None, None, None
        if ctx.half:  # VALU / atomic kernels read an fp32 table: widen the halves (exact); gradient returned in the table's dtype
            colors = colors.float()
        v_colors = torch.zeros(n, d, device=dev)
        if need_geom:
            v_opac = torch.zeros(n, device=dev)
            v_m2d = torch.zeros(n, 2, device=dev)
            v_con = torch.zeros(n, 3, device=dev)
            bflags = (flags & 3) | _lib.GAGS_RECS_BY_GAUSSIAN
        else:
            v_opac = v_m2d = v_con = None
            bflags = (flags & 3) | _lib.GAGS_BWD_COLORS_ONLY | _lib.GAGS_RECS_BY_GAUSSIAN
        with profiler.stage("raster_bwd"):
            check(lib.gags_raster_bwd(d, width, height, ptr(means2d), ptr(conics), ptr(opacities), ptr(colors),
                                      ptr(backgrounds), ptr(offsets), ptr(flatten_ids), n_isects, ptr(packed),
                                      ptr(alphas), ptr(last_ids), ptr(v_out), ptr(v_alphas), ptr(v_colors),
                                      ptr(v_opac), ptr(v_m2d), ptr(v_con), bflags, _stream()), "gags_raster_bwd")
        if ctx.half:
            v_colors = v_colors.half()
        return v_m2d, v_con, v_colors, v_opac, v_bg, None, None, None, None, None, None, None, None, None, None


def _offsets_with_count(offsets, n_tiles, n_isects):
    """ABI v2: every raster kernel reads `isect_offsets[tile + 1]` as a tile's end, so the buffer must hold n_tiles + 1
    entries, the last one = the intersection count (include/gags_raster.h).  tile_binning's `offsets_full` is such a buffer
    and passes through; a gsplat-style [tile_h, tile_w] tensor (no entry behind its last tile) is copied into one with the
    TRUE count `n_isects` the caller states -- never a buffer size."""
    if offsets.dim() == 1 and offsets.numel() == n_tiles + 1 and offsets.is_contiguous() and offsets.dtype == torch.int32:
        return offsets
    if offsets.numel() != n_tiles:
        raise ValueError(f"isect_offsets must have {n_tiles} (= tile_h * tile_w) entries, or {n_tiles + 1} with the count")
    full = torch.empty(n_tiles + 1, dtype=torch.int32, device=offsets.device)
    full[:n_tiles] = offsets.reshape(-1)
    full[n_tiles] = int(n_isects)
    return full


def _channel_ranges(d, spec):
    """[(c0, c1), ...] covering [0, d) for the range-staged backward, or None when D is served in one piece.  spec: an int
    (uniform ranges; D must be a multiple) or a tuple of widths, multiples of 128, applied in order -- the last one
    repeats until D is covered, a range that would overshoot is cut at D (which must then be a multiple of 128)."""
    if isinstance(spec, int):
        if spec <= 0 or d % spec != 0 or d <= spec or spec % 32 != 0:
            return None
        return [(c, c + spec) for c in range(0, d, spec)]
    widths = [int(w) for w in spec]
    if not widths or any(w <= 0 or w % 128 != 0 for w in widths) or d % 128 != 0 or d <= widths[0]:
        return None
    out, c, i = [], 0, 0
    while c < d:
        w = min(widths[min(i, len(widths) - 1)], d - c)
        out.append((c, c + w))
        c, i = c + w, i + 1
    return out


def _geom_mfma_width(d):
    """Widths whose geometry gradients run through gags_raster_bwd_geom (below that the VALU kernel is faster)."""
    return d >= 16 and d % 8 == 0 and d <= 1024


class _EarlyRowmap:
    """The backward's row map, enqueued by the forward (RasterContext.early_rowmap): the map, and its total on the way to a
    pinned buffer behind an event."""

    def __init__(self, trow, total, stmp, host, ev, rctx, dev):
        self.trow, self.total, self.stmp, self.host, self.ev, self.rctx, self.dev = trow, total, stmp, host, ev, rctx, dev
        self._rows = None

    def rows(self):
        if self._rows is None:
            self.ev.synchronize()  # (long passed by the time a backward asks)
            self._rows = int(self.host[0])
            self.rctx.give_pinned(self.dev, self.host)
            self.host = None
        return self._rows

    def __del__(self):
        # a forward that was never differentiated: its pinned word goes back to the pool (a later user's copy is enqueued behind
        # this one's on the same stream, and is read behind its own event)
        try:
            if self.host is not None:
                self.rctx.give_pinned(self.dev, self.host)
                self.host = None
        except Exception:
            pass


def _early_rowmap(lib, rctx, offsets, blk_rows, fwd_scratch, n_isects, width, height, dev):
    ne = lib.gags_bwd_rowmap_elems(n_isects, width, height)
    trow = torch.empty(ne, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    sb = lib.gags_bwd_rowmap_scratch_bytes(n_isects)
    stmp = torch.empty(max(sb, 4), dtype=torch.uint8, device=dev)
    check(lib.gags_bwd_rowmap(n_isects, width, height, ptr(offsets), ptr(blk_rows), ptr(fwd_scratch), fwd_scratch.numel(),
                              ptr(trow), ne, ptr(total), ptr(stmp), sb, _stream()), "gags_bwd_rowmap")
    host = rctx.take_pinned(dev)
    host.copy_(total, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return _EarlyRowmap(trow, total, stmp, host, ev, rctx, dev)


def _backward_staged(lib, rctx, offsets, n_isects, blk_rows, fwd_scratch, v_out, n, d, width, height, xflag=0,
                     flatten_ids=None, prezero=None, exact_rows=False, early=None):
    """Colours-only backward without atomics: hit flags of the forward -> prefix sum (one row per (tile, Gaussian)
    pair that blended anything) -> one 4-byte readback (total rows) -> merged partial rows -> sort by Gaussian ->
    segmented sum."""
    dev = v_out.device
    st = _stream()
    if rctx.grad_rows_hook is not None and rctx.grad_range_hook is not None and flatten_ids is not None:
        mask = torch.empty(n, dtype=torch.uint8, device=dev)
        check(lib.gags_blended_mask(n_isects, width, height, n, ptr(flatten_ids), ptr(fwd_scratch), fwd_scratch.numel(),
                                    ptr(mask), st), "gags_blended_mask")
        rctx.grad_rows_hook(mask)  # before the readback below: the ranks agree on the union while the backward starts
    hook = rctx.grad_range_hook
    ranges = _channel_ranges(d, rctx.grad_range_channels) if hook is not None else None
    cap_key = (n, width, height, dev.index)
    pending = None
    if early is not None:
        # the forward enqueued the row map behind its own kernels and its count has reached the host meanwhile: no prefix sum,
        # no readback, no drained queue here
        trow, total, rows = early.trow, early.total, early.rows()
    else:
        ne = lib.gags_bwd_rowmap_elems(n_isects, width, height)
        trow = torch.empty(ne, dtype=torch.int32, device=dev)
        total = torch.empty(1, dtype=torch.int32, device=dev)
        sb = lib.gags_bwd_rowmap_scratch_bytes(n_isects)
        stmp = torch.empty(max(sb, 4), dtype=torch.uint8, device=dev)
    if early is None:
        with profiler.stage("bwd_rowcount"):
            check(lib.gags_bwd_rowmap(n_isects, width, height, ptr(offsets), ptr(blk_rows), ptr(fwd_scratch),
                                      fwd_scratch.numel(), ptr(trow), ne, ptr(total), ptr(stmp), sb, st), "gags_bwd_rowmap")
            if rctx.capacity_mode and hook is None and cap_key in rctx.cap_rows and not exact_rows:
                # capacity mode (see RasterContext): the row count stays on the device until the backward is enqueued
                rows = min(max(n_isects, 1), int(rctx.cap_rows[cap_key] * CAP_MARGIN) + 1024)
                pending = _DeferredCount(total, rctx)
            else:
                host = ctypes.c_int32(0)
                check(lib.gags_read_i32(ptr(total), ctypes.byref(host), st), "gags_read_i32")
                rows = int(host.value)
    # a heavy view's partial rows ([rows, D] fp32) can outgrow the device (C5H: 80 M rows x 2 KB): beyond PROW_MAX_BYTES the
    # gradient is produced one 128-channel range at a time through a [rows, 128] scratch (stage bit 256)
    narrow = (hook is None and pending is None and d % 128 == 0 and d > 128 and rows * d * 4 > PROW_MAX_BYTES)
    nbytes = lib.gags_bwd_staged_scratch_bytes(rows, n, 128 if narrow else d)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    v_dtype = torch.float16 if (xflag & 64) else torch.float32
    # the persistent buffer (rows of Gaussians that blend nothing are never written again): not under the exchange hooks (the
    # ranks' sum lands in rows this view did not write), the capacity mode or the zero-fill experiment
    # ... under the exchange hooks only when the reduce stage writes the exchanged block itself: its rows -- where finish()
    # writes the ranks' sum -- then count as written (gags_raster_bwd_colors_staged_wire)
    wire_hook = rctx.grad_wire_hook if (hook is not None and ranges is not None and not (xflag & 64)) else None
    kept = None
    if pending is None and prezero is None and (hook is None or (wire_hook is not None and rctx.grad_rows_hook is not None)):
        kept = rctx.kept_grad(n, d, v_dtype, dev)
    v_colors = kept[0] if kept is not None else torch.empty(n, d, device=dev, dtype=v_dtype)
    rows_dev = ptr(total) if pending is not None else None

    def keep_or_range(stage, c0, cw):
        """One staged call for channels [c0, c0 + cw): through the persistent buffer's entry when the call holds the reduce stage."""
        if kept is not None and (stage & 15) in (0, 3):
            check(lib.gags_raster_bwd_colors_staged_keep(
                d, n, width, height, ptr(offsets), n_isects, ptr(v_out), ptr(blk_rows), ptr(trow), rows, ptr(fwd_scratch),
                fwd_scratch.numel(), ptr(scratch), nbytes, ptr(v_colors), stage, c0, cw, ptr(kept[1]), ptr(kept[2]), st),
                "gags_raster_bwd_colors_staged_keep")
        else:
            check(lib.gags_raster_bwd_colors_staged_cap(
                d, n, width, height, ptr(offsets), n_isects, ptr(v_out), ptr(blk_rows), ptr(trow), rows, ptr(fwd_scratch),
                fwd_scratch.numel(), ptr(scratch), nbytes, ptr(v_colors), stage, c0, cw, rows_dev, st),
                "gags_raster_bwd_colors_staged_cap")

    def run(stage):
        keep_or_range(stage | xflag, 0, d)

    if prezero is not None and hook is None:
        # the forward started a zero-fill of this tensor on a second stream while the binning kernels (small, latency-bound:
        # the memory system idles) ran; the reduce stage then writes only the rows that exist -- 73 % of the Gaussians
        # blend nothing at C3: 3.07 GB -> 0.83 GB written here.  (Filled under the rows kernel instead, the fill slowed
        # that kernel by as much as the reduce stage gained.)
        buf, ev = prezero
        if buf.shape == v_colors.shape and buf.dtype == v_colors.dtype:
            torch.cuda.current_stream().wait_event(ev)
            v_colors = buf
            xflag |= 128
    if narrow:
        for c0 in range(0, d, 128):
            for stage in ((1, 2, 3) if c0 == 0 else (1, 3)):
                with profiler.stage(("bwd_rows", "bwd_sort", "bwd_reduce")[stage - 1]):
                    keep_or_range(stage | xflag | 256, c0, 128)
    elif hook is not None and ranges is not None:
        alias = v_colors.detach()  # own TensorImpl, same storage: autograd may still adopt v_colors without a copy
        # the partial rows are produced for `grad_rows_group` channels per launch: every rows launch streams the view's weight
        # tiles from HBM once for all of its 128-channel slices (they share them through L2), so wider groups re-read them less
        # often -- and deliver their first range later; the reduce stage and the exchange keep the narrower range either way
        group = max(int(rctx.grad_rows_group), 1)
        rows_done = 0
        for c0, c1 in ranges:
            if c1 > rows_done:
                g1 = rows_done
                while g1 < c1 or (g1 - rows_done < group and g1 < d):
                    g1 = next(b for a, b in ranges if a == g1)
                for stage in ((1, 2) if rows_done == 0 else (1,)):
                    with profiler.stage(("bwd_rows", "bwd_sort")[stage - 1]):
                        check(lib.gags_raster_bwd_colors_staged_range(
                            d, n, width, height, ptr(offsets), n_isects, ptr(v_out), ptr(blk_rows), ptr(trow), rows,
                            ptr(fwd_scratch), fwd_scratch.numel(), ptr(scratch), nbytes, ptr(v_colors), stage | xflag, rows_done,
                            g1 - rows_done, st), "gags_raster_bwd_colors_staged_range")
                rows_done = g1
            # the rows the ranks exchange leave from the reduce kernel itself (no pack pass over the range afterwards)
            w = wire_hook(c0, c1) if wire_hook is not None else None
            if w is None and kept is not None:
                # the exchange packs for itself after all (bf16 wire, all rows): its sum may land in rows the flags do not
                # cover.  This step's reduce writes every row; the buffer is not kept
                rctx.forget_kept(n, d, v_dtype, dev)
                kept = None
            with profiler.stage("bwd_reduce"):
                check(lib.gags_raster_bwd_colors_staged_wire(
                    d, n, width, height, ptr(offsets), n_isects, ptr(v_out), ptr(blk_rows), ptr(trow), rows,
                    ptr(fwd_scratch), fwd_scratch.numel(), ptr(scratch), nbytes, ptr(v_colors), 3 | xflag, c0, c1 - c0,
                    ptr(w[0]) if w else None, ptr(w[1]) if w else None, ptr(kept[1]) if kept else None,
                    ptr(kept[2]) if kept else None, st), "gags_raster_bwd_colors_staged_wire")
            if w is not None:
                hook(alias, c0, c1, w[1])
            else:
                hook(alias, c0, c1)
    elif profiler.ENABLED:  # one event pair per kernel (group), for the roofline line of bench.py
        for stage, name in enumerate(("bwd_rows", "bwd_sort", "bwd_reduce"), start=1):
            with profiler.stage(name):
                run(stage)
    else:
        run(0)
        if hook is not None:
            hook(v_colors.detach(), 0, d)
    if pending is not None:
        true_rows = pending.get()
        if true_rows > rows:  # more rows than the remembered capacity (none was stored out of bounds): again, exact
            rctx.cap_rows[cap_key] = true_rows
            return _backward_staged(lib, rctx, offsets, n_isects, blk_rows, fwd_scratch, v_out, n, d, width, height, xflag & ~128,
                                    flatten_ids, None, exact_rows=True)
        rows = true_rows
    if hook is None and rctx.capacity_mode:
        rctx.cap_rows[cap_key] = max(rows, int(0.97 * rctx.cap_rows.get(cap_key, 0)))
    profiler.note("bwd_rows", rows)
    return v_colors


def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height,
                  near_plane=0.01, far_plane=1e10, radius_clip=0.0, eps2d=0.3, sh_degree=None, packed=False,
                  tile_size=16, backgrounds=None, render_mode="RGB", sparse_grad=False, absgrad=False,
                  rasterize_mode="classic", channel_chunk=32, distributed=False, camera_model="pinhole",
                  covars=None, raster_flags=0, raw_params=False, scaling_modifier=1.0, context=None):
    """See module docstring.  `channel_chunk` is accepted and ignored: any D is composited in a
    single pass over the sorted lists (SURVEY A12 shows this is identical per channel).
    raw_params=True (not part of gsplat's signature): `quats`, `scales`, `opacities` are the STORED parameters of
    scene/gaussian_model.py:48-61 -- `_rotation` [N,4] un-normalised, `_scaling` [N,3] log-space, `_opacity` [N] or [N,1]
    logits -- and the getters of :116-139 together with `* scaling_modifier` run inside the projection kernel
    (gags_project_fwd_raw; bit-identical to torch's exp / F.normalize / sigmoid): no elementwise launches, no extra passes
    over N, and the backward returns the gradients of the stored parameters.
    context (not part of gsplat's signature): the RasterContext that carries this caller's hooks and capacities; None = the
    calling thread's default_context()."""
    rctx = context if context is not None else default_context()
    if tile_size != TILE:
        raise NotImplementedError("tile_size must be 16 (the gsplat default the reference relies on)")
    if rasterize_mode != "classic" or camera_model != "pinhole" or covars is not None or distributed or absgrad:
        raise NotImplementedError("only the options the reference uses are implemented "
                                  "(classic mode, pinhole camera, quats+scales)")
    if render_mode not in ("RGB", "D", "ED", "RGB+D", "RGB+ED"):
        raise ValueError(f"unknown render_mode {render_mode}")
    if viewmats.dim() != 3 or viewmats.shape[0] != 1 or Ks.shape[0] != 1:
        raise NotImplementedError("one camera per call (the reference renders one view per iteration, train.py:134-142)")
    n = means.shape[0]
    if means.shape != (n, 3) or quats.shape != (n, 4) or scales.shape != (n, 3):
        raise ValueError("means [N,3], quats [N,4], scales [N,3] expected")
    if opacities.shape != (n,) and not (raw_params and opacities.shape == (n, 1)):  # (the stored logits are [N,1])
        raise ValueError("opacities [N] expected" + (" (or the stored [N,1] logits with raw_params)" if raw_params else ""))
    _need_cuda(means, quats, scales, opacities, colors, viewmats, Ks, backgrounds)
    width, height = int(width), int(height)
    viewmat, K = viewmats[0], Ks[0]

    records = None
    if raw_params:
        # (the record table is only read by the matrix-core path: D >= 16 after the depth channel of RGB+D / RGB+ED is appended)
        dfinal = (3 if sh_degree is not None else colors.shape[-1]) + (1 if render_mode in ("RGB+D", "RGB+ED") else 0)
        if render_mode in ("D", "ED"):
            dfinal = 1
        radii, means2d, depths, conics, tiles, opacities, records = _ProjectRaw.apply(
            means, quats, scales, opacities, viewmat, K, width, height, float(eps2d), float(near_plane), float(far_plane),
            float(radius_clip), float(scaling_modifier), bool(_mfma_width(dfinal) and n > 0))
        if records.numel() == 0:
            records = None
    else:
        radii, means2d, depths, conics, tiles = _Project.apply(means, quats, scales, viewmat, K, width, height,
                                                               float(eps2d), float(near_plane), float(far_plane),
                                                               float(radius_clip))
    # [1,N,2] node callers may retain_grad() on (gaussian_renderer/__init__.py:75-78); the
    # rasterizer consumes a view of it so its .grad receives d loss / d means2d.
    means2d_c = means2d[None]
    means2d = means2d_c[0]
    if sh_degree is not None:
        if colors.dim() != 3 or colors.shape[2] != 3:
            raise ValueError("SH colours must be [N,K,3]")
        campos = torch.inverse(viewmat.double())[:3, 3].float()
        cols = _SH.apply(colors, means, campos, radii, int(sh_degree))
    else:
        if colors.dim() != 2 or colors.shape[0] != n:
            raise ValueError("colors must be [N,D] when sh_degree is None")
        cols = colors
    bg = None if backgrounds is None else backgrounds.reshape(-1)
    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat([cols, depths[:, None]], dim=-1)
        if bg is not None:
            bg = torch.cat([bg, torch.zeros(1, device=bg.device)])
    elif render_mode in ("D", "ED"):
        cols = depths[:, None]
        bg = None if bg is None else torch.zeros(1, device=bg.device)

    prezero = None
    dz = cols.shape[-1]
    if (rctx.overlap_zero_fill and rctx.grad_range_hook is None and torch.is_grad_enabled() and cols.requires_grad and _mfma_width(dz)
            and dz <= 1024 and n * dz >= ZERO_FILL_MIN_ELEMS and not (raster_flags & (_lib.GAGS_BWD_ATOMIC | _lib.GAGS_FWD_NO_MFMA
                                                                                      | _lib.GAGS_FWD_FUSED))
            and not (means.requires_grad or quats.requires_grad or scales.requires_grad or opacities.requires_grad)):
        # colours-only (GAD) backward ahead: its gradient tensor is mostly rows of zeros.  Fill it now, on a second stream,
        # under the binning kernels; the backward's reduce stage then writes only the rows that exist (_backward_staged)
        vbuf = torch.empty(n, dz, device=cols.device, dtype=torch.float16 if cols.dtype == torch.float16 else torch.float32)
        side = rctx.side_stream(cols.device)
        ev0, ev1 = torch.cuda.Event(), torch.cuda.Event()
        ev0.record()
        with torch.cuda.stream(side):
            side.wait_event(ev0)
            vbuf.zero_()
            ev1.record()
        vbuf.record_stream(side)
        prezero = (vbuf, ev1)

    dcols = cols.shape[-1]
    wide = _mfma_width(dcols)  # the matrix-core path wants packed records
    cap_key = (n, width, height, means.device.index)

    def run(cap):
        with torch.no_grad(), profiler.stage("binning"):
            b = tile_binning(means2d, radii, depths, tiles, width, height, conics if wide else None,
                             _c(opacities) if wide else None, cap, records=records if wide else None, context=rctx)
        offs, flat = b[5], b[1]
        # heavy views: the lists cut to what their tiles read (_trim_lists) -- the raster passes, their scratch and the backward
        # work on the cut lists; callers still get the full ones in `info`
        trimmed = None
        n_full = flat.shape[0]
        if (cap is None and wide and b[4] is not None and n_full > 0 and rctx.trim_lists is not False
                and not (raster_flags & (_lib.GAGS_FWD_NO_MFMA | _lib.GAGS_FWD_FUSED))):
            lib_ = _lib.load()
            if rctx.trim_lists or lib_.gags_raster_fwd_scratch_bytes(n_full, width, height) > TRIM_AUTO_BYTES:
                with torch.no_grad():
                    trimmed = _trim_lists(lib_, n, width, height, offs, flat, n_full, b[4])
                offs, flat = trimmed[0], trimmed[1]
        # any width in ONE rasterization: 513 = 512 CLIP channels + 1 (BASELINE.json configs[4] "512-d feat + granularity")
        # is four 128-channel slices and one lane of a narrow slice on the same matrix-core kernels, into one output tensor
        r = _Rasterize.apply(means2d, conics, cols, opacities, bg, offs, flat, b[4], width, height, int(raster_flags), prezero,
                             rctx, None, torch.is_grad_enabled())
        if trimmed is not None:
            # last_ids are sorted indices: a COPY goes back to the full lists' numbering for the caller (the autograd node keeps
            # its own, which matches the lists it saved)
            with torch.no_grad():
                last_user = r[2].clone()
                check(_lib.load().gags_trim_last_ids(width, height, ptr(b[5]), ptr(offs), ptr(r[1]), ptr(last_user), _stream()),
                      "gags_trim_last_ids")
            profiler.note("isects_trimmed", trimmed[2])
            r = (r[0], r[1], last_user, trimmed[2])
        else:
            r = (r[0], r[1], r[2], None)
        return b[:5], r

    cap = None
    if rctx.capacity_mode and cap_key in rctx.cap_isects:
        cap = min(MAX_ISECTS - 1, int(rctx.cap_isects[cap_key] * CAP_MARGIN) + 4096)
    (isect_ids, flatten_ids, isect_offsets, n_isects, packed), (out, alphas, last_ids, n_trimmed) = run(cap)
    if cap is not None:
        n_true = n_isects.get()  # (the scan that produced it finished long ago: everything above is already enqueued)
        _check_isects(n_true, ((width + TILE - 1) // TILE) * ((height + TILE - 1) // TILE))
        if n_true > cap:  # more intersections than the remembered capacity: nothing was written out of bounds; run again, exact
            (isect_ids, flatten_ids, isect_offsets, n_isects, packed), (out, alphas, last_ids, n_trimmed) = run(None)
        else:
            n_isects = n_true
            isect_ids, flatten_ids = isect_ids[:n_true], flatten_ids[:n_true]
    if rctx.capacity_mode:
        rctx.cap_isects[cap_key] = max(n_isects, int(0.97 * rctx.cap_isects.get(cap_key, 0)))
    if render_mode in ("ED", "RGB+ED"):
        if out.requires_grad:
            out = torch.cat([out[..., :-1], out[..., -1:] / alphas[..., None].clamp(min=1e-10)], dim=-1)
        else:  # K12 in place
            check(_lib.load().gags_ed_normalize(height * width, out.shape[-1], ptr(out), ptr(alphas), _stream()),
                  "gags_ed_normalize")

    tile_w, tile_h = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    info = {
        "camera_ids": None, "gaussian_ids": None,
        "radii": radii[None], "means2d": means2d_c, "depths": depths[None], "conics": conics[None],
        "opacities": opacities[None], "tile_width": tile_w, "tile_height": tile_h,
        "tiles_per_gauss": tiles[None], "isect_ids": isect_ids, "flatten_ids": flatten_ids,
        "isect_offsets": isect_offsets[None], "last_ids": last_ids, "width": width, "height": height,
        "tile_size": TILE, "n_cameras": 1, "n_isects": n_isects,
        "n_isects_trimmed": n_trimmed,  # (not gsplat's: list entries the raster passes worked on when the lists were trimmed, else None)
    }
    return out[None], alphas[None, ..., None], info
