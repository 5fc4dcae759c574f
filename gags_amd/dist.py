"""Multi-GPU step: one process per GPU, geometry replicated.  Two decompositions of a step of
`world` views:

* by VIEW (north_star / config C4): features replicated, GPU r renders view r, and the ONLY exchange on the
  path is the sum over ranks of d loss / d _semantic_feature ([N,D] fp32) at step end -- `torch.distributed`
  backend "nccl" is RCCL over xGMI on ROCm; the same code runs on "gloo" for the CPU tests.
  xGMI is a point-to-point full mesh (7 links per GPU), so a ring all-reduce of the 3 GB C3 gradient is bound
  by ONE link.  The default here is therefore reduce-scatter + all-gather over row buckets (each rank
  exchanges a distinct 1/world shard with every peer concurrently); `mode="allreduce"` keeps the plain
  collective for comparison.
* by CHANNEL: the rasterization is independent per feature channel, so GPU r can own channels
  [c0, c1) of the feature table -- parameter, gradient and optimizer state -- and render EVERY view of the step
  for them.  Results are bit-identical to the single-GPU ones for those channels and there is NO data-path
  collective at all; the price is that the per-view work that does not depend on D (binning, weights pass, row
  sort: ~1.9 ms at C3) is repeated on every GPU.  With only 1 (N=2) or 3 (N=4) xGMI links between the ranks the
  3 GB gradient exchange of the by-view step costs more than the whole compute, so this is the faster
  decomposition there; `bench.py --parallel auto` measures both and keeps the faster one.
"""
import torch
import torch.distributed as dist

BUCKET_BYTES = 256 << 20


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_views(n_views, rank=None, world_size=None):
    """Views rendered by `rank`: round-robin over the camera list (independent cameras per GPU)."""
    ws = world() if world_size is None else world_size
    r = (dist.get_rank() if ws > 1 and rank is None else (rank or 0))
    return list(range(r, n_views, ws))


def channel_shard(d, rank=None, world_size=None, multiple=16):
    """Channel range [c0, c1) of a D-wide feature table owned by `rank` in the by-channel decomposition.
    Ranges are contiguous, cover [0, D) and are multiples of `multiple` wide (16: every shard then runs the
    matrix-core kernels) except possibly the last; ranks beyond D / multiple own nothing (c0 == c1)."""
    ws = world() if world_size is None else world_size
    r = (dist.get_rank() if ws > 1 and rank is None else (rank or 0))
    units = (d + multiple - 1) // multiple
    lo = (units * r) // ws * multiple
    hi = (units * (r + 1)) // ws * multiple
    return min(lo, d), min(hi, d)


_AG_IN_PLACE = True
_RS_IN_PLACE = True


def _reduce_scatter_in_place(shard, full):
    """reduce-scatter whose output is the rank's own slot of the input (NCCL's in-place form).  A backend that refuses
    aliased buffers gets a staging shard from then on."""
    global _RS_IN_PLACE
    if _RS_IN_PLACE:
        try:
            dist.reduce_scatter_tensor(shard, full)
            return
        except (RuntimeError, ValueError):
            _RS_IN_PLACE = False
    tmp = torch.empty_like(shard)
    dist.reduce_scatter_tensor(tmp, full)
    shard.copy_(tmp)



def _all_gather_in_place(out, shard):
    """all-gather whose input is the rank's own slot of the output (NCCL's in-place form: no staging copy).  A backend
    that refuses aliased buffers costs one clone per bucket from then on."""
    global _AG_IN_PLACE
    if _AG_IN_PLACE:
        try:
            dist.all_gather_into_tensor(out, shard)
            return
        except (RuntimeError, ValueError):
            _AG_IN_PLACE = False
    dist.all_gather_into_tensor(out, shard.clone())


def reduce_feature_grad(grad, mode="rs_ag", average=False, bucket_bytes=BUCKET_BYTES):
    """In-place sum (or mean) over ranks of a [N,D] gradient.  No-op at world size 1."""
    ws = world()
    if ws == 1:
        return grad
    assert grad.is_contiguous()
    flat = grad.view(-1)
    numel = flat.numel()
    if mode == "allreduce":
        step = max(1, bucket_bytes // flat.element_size())
        works = [dist.all_reduce(flat[o:o + step], async_op=True) for o in range(0, numel, step)]
        for w in works:
            w.wait()
    elif mode == "rs_ag":
        # bucket so that every bucket splits evenly into `ws` shards; the ragged tail is all-reduced
        per = max(ws, (bucket_bytes // flat.element_size()) // ws * ws)
        main = numel // per * per
        for o in range(0, main, per):
            b = flat[o:o + per]
            shard = b.view(ws, per // ws)[dist.get_rank()]
            _reduce_scatter_in_place(shard, b)
            _all_gather_in_place(b, shard)
        if main < numel:
            dist.all_reduce(flat[main:])
    else:
        raise ValueError(mode)
    if average:
        flat.div_(ws)
    return grad


class OverlappedGradReducer:
    """By-view step with the gradient exchange overlapped with the backward (SURVEY 8e "overlapped with the tail of
    bwd").  Used as a context manager around `loss.backward()`: the staged backward then produces the feature
    gradient one 128-channel range at a time (gags_amd.rasterization.GRAD_RANGE_HOOK) and every finished range is
    packed, summed over the ranks and unpacked on a second stream while the next range is still being computed:

        red = OverlappedGradReducer(mode="rs_ag")
        with red:
            loss.backward()
        red.finish(pc._semantic_feature.grad)     # compute stream waits for the exchange; exact fp32 sum

    wire="bf16" (opt-in) halves the bytes on xGMI: the range is rounded to bfloat16, summed in bfloat16 by the
    collective and widened again; the result differs from the fp32 sum by ~1e-2 relative (tests/test_dist_cpu.py
    states and checks the bound), so it is never the default.
    rows="union" (the default): a view's gradient is non-zero only in the rows of the Gaussians that blended into one
    of its pixels (27 % of N at C3).  The backward hands over that mask first (GRAD_ROWS_HOOK); the ranks take its
    union (a max-all-reduce of N bytes) and every range is exchanged as the [|union|, 128] block of those rows -- the
    same collectives on fewer bytes, the same exact fp32 sum (rows outside the union are zero on every rank).
    rows="all" exchanges all N rows.
    If autograd did not adopt the tensor the hook saw (another consumer of the gradient forced a copy), finish() falls
    back to the plain reduction of the final gradient: always correct, overlap lost.
    `exposed_ms()` = time the compute stream had to wait for the exchange after the backward had finished."""

    def __init__(self, mode="rs_ag", wire=None, bucket_bytes=BUCKET_BYTES, rows="union"):
        if rows not in ("union", "all"):
            raise ValueError(rows)
        self.mode, self.wire, self.bucket_bytes, self.rows = mode, wire, bucket_bytes, rows
        self.comm = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.rows_exchanged = None  # |union| of the last step (None: all rows)
        self._reset()

    def _reset(self):
        self._ptr, self._covered, self._ev = None, 0, None
        self._mask, self._idx = None, None

    def __enter__(self):
        from . import rasterization
        self._reset()
        self._prev = (rasterization.GRAD_RANGE_HOOK, rasterization.GRAD_ROWS_HOOK)
        rasterization.GRAD_RANGE_HOOK = self.on_range
        rasterization.GRAD_ROWS_HOOK = self.on_rows if (self.rows == "union" and world() > 1) else None
        return self

    def __exit__(self, *exc):
        from . import rasterization
        rasterization.GRAD_RANGE_HOOK, rasterization.GRAD_ROWS_HOOK = self._prev
        return False

    def on_rows(self, mask):
        """mask uint8 [N] of this rank's view; the union over the ranks is formed on the exchange stream right away
        (N bytes), its index list when the first range arrives (by then it has long finished: no stall)."""
        if self._mask is not None:  # a second view in the same step: all rows from here on
            self._mask, self._idx = False, None
            return
        if mask.is_cuda and self.comm is not None:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                dist.all_reduce(mask, op=dist.ReduceOp.MAX)
            mask.record_stream(self.comm)
        else:
            dist.all_reduce(mask, op=dist.ReduceOp.MAX)
        self._mask = mask

    def _union_rows(self):
        if self._idx is None and self._mask is not None and self._mask is not False:
            self._idx = torch.nonzero(self._mask).squeeze(1)  # one host sync, on the exchange stream's past work only
            self.rows_exchanged = int(self._idx.numel())
        return self._idx

    def _exchange(self, grad, c0, c1):
        part = grad[:, c0:c1]
        idx = self._union_rows()
        if idx is not None:
            buf = part.index_select(0, idx)  # pack: the rows of the union only
            if self.wire == "bf16":
                w = buf.to(torch.bfloat16)
                reduce_feature_grad(w, mode=self.mode, bucket_bytes=self.bucket_bytes)
                buf.copy_(w)
            elif self.wire in (None, "fp32"):
                reduce_feature_grad(buf, mode=self.mode, bucket_bytes=self.bucket_bytes)
            else:
                raise ValueError(self.wire)
            part.index_copy_(0, idx, buf)  # unpack; every other row is zero on every rank
            return
        self.rows_exchanged = None
        buf = part.contiguous()  # pack (a copy unless the range is the whole row)
        if self.wire == "bf16":
            w = buf.to(torch.bfloat16)
            reduce_feature_grad(w, mode=self.mode, bucket_bytes=self.bucket_bytes)
            buf.copy_(w)
        elif self.wire in (None, "fp32"):
            reduce_feature_grad(buf, mode=self.mode, bucket_bytes=self.bucket_bytes)
        else:
            raise ValueError(self.wire)
        if buf.data_ptr() != part.data_ptr():
            part.copy_(buf)  # unpack

    def on_range(self, grad, c0, c1):
        if world() > 1:
            if grad.is_cuda and self.comm is not None:
                ev = torch.cuda.Event()
                ev.record()  # the range's kernels, on the compute stream
                with torch.cuda.stream(self.comm):
                    # the union's index list BEFORE the exchange stream is made to wait for this range: its host sync
                    # then covers the mask all-reduce only, while the GPU still has this whole range queued
                    self._union_rows()
                    self.comm.wait_event(ev)
                    self._exchange(grad, c0, c1)
                grad.record_stream(self.comm)
            else:
                self._exchange(grad, c0, c1)
        self._ptr = grad.data_ptr()
        self._covered += c1 - c0

    def finish(self, param_grad):
        """Make the reduced gradient visible to the compute stream; returns True if the overlapped exchange was used."""
        cuda = param_grad.is_cuda and self.comm is not None
        if cuda:
            self._bwd_done = torch.cuda.Event(enable_timing=True)
            self._bwd_done.record()
            torch.cuda.current_stream().wait_stream(self.comm)
            self._all_done = torch.cuda.Event(enable_timing=True)
            self._all_done.record()
        used = self._ptr == param_grad.data_ptr() and self._covered == param_grad.shape[1]
        if not used and world() > 1:
            reduce_feature_grad(param_grad, mode=self.mode, bucket_bytes=self.bucket_bytes)
        return used

    def exposed_ms(self):
        if getattr(self, "_all_done", None) is None:
            return 0.0
        self._all_done.synchronize()
        return float(self._bwd_done.elapsed_time(self._all_done))


def distributed_step(render_fn, cams, pc, bg, cotangents, mode="rs_ag"):
    """One step over `cams`.  mode "rs_ag" / "allreduce": render this rank's share of the views, backprop
    <render, G_v> for each, then reduce the feature gradient over ranks.  mode "channel": `pc` holds this
    rank's channel shard of the features ([N, c1-c0], see channel_shard) and `cotangents[v]` the matching
    channels of G_v; every view is rendered, the gradient of the shard accumulates locally, nothing is
    exchanged.  Returns the local sum of losses."""
    if mode == "channel":
        pc._semantic_feature.grad = None
        total = None
        for v in range(len(cams)):
            pkg = render_fn(cams[v], pc, None, bg, feature_mode=True)
            loss = (pkg["render"] * cotangents[v]).sum()
            loss.backward()
            total = loss.detach() if total is None else total + loss.detach()
        return total
    mine = shard_views(len(cams))
    pc._semantic_feature.grad = None
    total = None
    for v in mine:
        pkg = render_fn(cams[v], pc, None, bg, feature_mode=True)
        loss = (pkg["render"] * cotangents[v]).sum()
        loss.backward()
        total = loss.detach() if total is None else total + loss.detach()
    if pc._semantic_feature.grad is None:
        pc._semantic_feature.grad = torch.zeros_like(pc._semantic_feature)
    reduce_feature_grad(pc._semantic_feature.grad, mode=mode)
    return total
