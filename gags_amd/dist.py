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
import os

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


# NCCL's in-place forms (reduce-scatter into the rank's own slot of the input, all-gather from it) save one staging
# buffer of bucket / world bytes each.  They are OFF unless GAGS_DIST_INPLACE=1: the decision is made once, here, not by
# catching an exception in the middle of a step (a collective that raised on one rank leaves the others waiting), and
# the staging copies cost 2 x 32 MB per 256 MB bucket at 8 ranks -- nothing next to the exchange itself.
_IN_PLACE = os.environ.get("GAGS_DIST_INPLACE", "0") == "1"


def _reduce_scatter(shard, full):
    if _IN_PLACE:
        dist.reduce_scatter_tensor(shard, full)
        return
    tmp = torch.empty_like(shard)
    dist.reduce_scatter_tensor(tmp, full)
    shard.copy_(tmp)


def _all_gather(out, shard):
    dist.all_gather_into_tensor(out, shard if _IN_PLACE else shard.clone())


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
        # (a tensor smaller than one bucket is ONE bucket: with per > numel everything used to fall through to the
        # tail's plain all-reduce -- exactly the 228 MB union-row blocks of the C4 step)
        per = max(ws, min(bucket_bytes // flat.element_size(), numel) // ws * ws)
        main = numel // per * per
        for o in range(0, main, per):
            b = flat[o:o + per]
            shard = b.view(ws, per // ws)[dist.get_rank()]
            _reduce_scatter(shard, b)
            _all_gather(b, shard)
        if main < numel:
            dist.all_reduce(flat[main:])
    else:
        raise ValueError(mode)
    if average:
        flat.div_(ws)
    return grad


def reduce_feature_grad_oop(src, mode="rs_ag", bucket_bytes=BUCKET_BYTES):
    """Sum over ranks of `src` into a NEW tensor; `src` is left as it is.  The bucketed reduce-scatter + all-gather is out of
    place by nature -- every bucket is reduce-scattered into a shard-sized buffer and all-gathered into the result -- so
    keeping the local values costs no copy (the overlapped exchange needs them whenever autograd did not adopt the tensor its
    hook saw: OverlappedGradReducer.finish).  mode "allreduce" copies first and reduces the copy."""
    ws = world()
    flat = src.reshape(-1)
    if ws == 1:
        return src.clone()
    out = torch.empty_like(flat)
    numel = flat.numel()
    if mode == "allreduce":
        out.copy_(flat)
        step = max(1, bucket_bytes // flat.element_size())
        works = [dist.all_reduce(out[o:o + step], async_op=True) for o in range(0, numel, step)]
        for w in works:
            w.wait()
    elif mode == "rs_ag":
        per = max(ws, min(bucket_bytes // flat.element_size(), numel) // ws * ws)
        main = numel // per * per
        for o in range(0, main, per):
            shard = torch.empty(per // ws, dtype=flat.dtype, device=flat.device)
            dist.reduce_scatter_tensor(shard, flat[o:o + per])
            dist.all_gather_into_tensor(out[o:o + per], shard)
        if main < numel:
            out[main:].copy_(flat[main:])
            dist.all_reduce(out[main:])
    else:
        raise ValueError(mode)
    return out.view(src.shape)


GEOMETRY_PARAMS = ("_xyz", "_rotation", "_scaling", "_opacity")  # [N,3] [N,4] [N,3] [N,1]: SURVEY 8e's [N, 3+4+3+1]


def reduce_geometry_grads(pc, mode="rs_ag", average=False, names=GEOMETRY_PARAMS):
    """By-view step with trainable geometry (north_star: "all-reduce of feature / GEOMETRY gradients"): sum over the ranks
    of d loss / d (_xyz, _rotation, _scaling, _opacity) as ONE packed [N, 11] fp32 block -- one bucket through the same
    reduce-scatter + all-gather as the feature gradient (66 MB at C3, against 3 GB of features) -- written back into each
    parameter's .grad.  Parameters that do not require grad are skipped (the reference's GAD stage freezes them all,
    scene/gaussian_model.py:183-208: then this is a no-op); a trainable parameter whose .grad is None on this rank (its
    view blended nothing) contributes zeros and receives the sum.  Every rank must call it with the same set of trainable
    parameters.  Returns the list of reduced parameter names."""
    params = [(k, getattr(pc, k)) for k in names if getattr(pc, k, None) is not None and getattr(pc, k).requires_grad]
    if not params or world() == 1:
        return [k for k, _ in params]
    n = params[0][1].shape[0]
    cols = [p.numel() // max(n, 1) for _, p in params]
    block = torch.empty(n, sum(cols), dtype=torch.float32, device=params[0][1].device)
    c = 0
    for (_, p), w in zip(params, cols):
        if p.grad is None:
            block[:, c:c + w].zero_()
        else:
            block[:, c:c + w].copy_(p.grad.reshape(n, w))
        c += w
    reduce_feature_grad(block, mode=mode, average=average)
    c = 0
    for (_, p), w in zip(params, cols):
        part = block[:, c:c + w].reshape(p.shape)
        if p.grad is None:
            p.grad = part.to(p.dtype).contiguous()
        else:
            p.grad.copy_(part)
        c += w
    return [k for k, _ in params]


_TYPE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _pack_rows(grad, idx, c0, c1, wire_dtype):
    """wire[r, :] = grad[idx[r], c0:c1] (idx None: every row) as a fresh contiguous [rows, c1 - c0] tensor of
    `wire_dtype`.  Device tensors: gags_pack_rows (HIP); host tensors (the gloo tests of the host logic): torch."""
    rows = grad.shape[0] if idx is None else idx.numel()
    if not grad.is_cuda:
        part = grad[:, c0:c1]
        return (part if idx is None else part.index_select(0, idx)).to(wire_dtype, copy=True).contiguous()
    from . import _lib
    wire = torch.empty(rows, c1 - c0, dtype=wire_dtype, device=grad.device)
    _lib.check(_lib.load().gags_pack_rows(rows, _lib.ptr(idx), _lib.ptr(grad), _TYPE[grad.dtype], grad.shape[1], c0,
                                          c1 - c0, _lib.ptr(wire), _TYPE[wire_dtype],
                                          torch.cuda.current_stream().cuda_stream), "gags_pack_rows")
    return wire


def _unpack_rows(grad, idx, c0, c1, wire, local=None):
    """grad[idx[r], c0:c1] = wire[r] (local None) or += wire[r] - local[r]."""
    if not grad.is_cuda:
        part = grad[:, c0:c1]
        val = wire.to(grad.dtype) if local is None else (wire.float() - local.float()).to(grad.dtype)
        if idx is None:
            part.copy_(val) if local is None else part.add_(val)
        elif local is None:
            part.index_copy_(0, idx, val)
        else:
            part.index_add_(0, idx, val)
        return
    from . import _lib
    rows = grad.shape[0] if idx is None else idx.numel()
    _lib.check(_lib.load().gags_unpack_rows(rows, _lib.ptr(idx), _lib.ptr(wire), _TYPE[wire.dtype], _lib.ptr(local),
                                            _lib.ptr(grad), _TYPE[grad.dtype], grad.shape[1], c0, c1 - c0,
                                            torch.cuda.current_stream().cuda_stream), "gags_unpack_rows")


class OverlappedGradReducer:
    """By-view step with the gradient exchange overlapped with the backward (SURVEY 8e "overlapped with the tail of
    bwd").  Used as a context manager around ONE `loss.backward()`: the staged backward then produces the feature
    gradient one 128-channel range at a time (RasterContext.grad_range_hook) and every finished range is
    packed into a private buffer, summed over the ranks on a second stream while the next range is still being computed,
    and written into the parameter's gradient by finish():

        red = OverlappedGradReducer(mode="rs_ag", param=pc._semantic_feature)
        with red:
            loss.backward()
        red.finish(pc._semantic_feature.grad)     # compute stream waits for the exchange; the fp32 sum (exact when adopted)

    The tensor autograd consumes is never written by the exchange stream (the hook only READS it); the reduced ranges
    live in the reducer until finish().  finish() ASSIGNS them only when the gradient tensor is the very tensor the
    hook saw (autograd adopted it: same storage) AND nobody wrote to it in place since (its version counter is the one
    the hook saw).  In every other case -- the parameter has a second consumer in the graph (a regulariser on
    `_semantic_feature`: autograd sums the terms into a fresh tensor, or adds in place), accumulation over several views,
    a clone or a cast with unknown history -- the gradient holds terms that are not this backward's local rows, so
    finish() adds `sum over ranks - local` from the packed local rows it always keeps (one more fp32 rounding per element
    than the assigned sum: grad + (sum - local) is not bit-identical to the sum).  Keeping them costs no copy since round 6:
    the collective is out of place (reduce_feature_grad_oop: a bucket is reduce-scattered into a shard-sized buffer and
    all-gathered into the RESULT block), so the packed block stays this rank's local rows -- whether autograd adopts the
    hook's tensor, and whether anybody writes to it in place afterwards, is only known in finish(); terms from other graph
    paths stay rank-local and are the caller's to reduce
    (reduce_feature_grad).  A gradient that has been reduced once is never
    reduced again.

    wire="bf16" (opt-in) halves the bytes on xGMI: the range is rounded to bfloat16, summed in bfloat16 by the
    collective and widened again; the result differs from the fp32 sum by ~1e-2 relative (tests/test_dist_cpu.py
    states and checks the bound), so it is never the default.
    rows="union" (the default): a view's gradient is non-zero only in the rows of the Gaussians that blended into one
    of its pixels (27 % of N at C3).  The backward hands over that mask first (RasterContext.grad_rows_hook); the ranks take its
    union (a max-all-reduce of N bytes) and every range is exchanged as the [|union|, 128] block of those rows -- the
    same collectives on fewer bytes, the same exact fp32 sum (rows outside the union are zero on every rank).
    rows="all" exchanges all N rows.
    If the backward never called the hook (narrow D, atomic kernels), finish() runs the plain reduction of the
    untouched gradient.  `exposed_ms()` = time the compute stream spent on the exchange after the backward had finished."""

    def __init__(self, mode="rs_ag", wire=None, bucket_bytes=BUCKET_BYTES, rows="union", param=None, sync_free=False,
                 cap_margin=1.1, cap_slack=1024, context=None, loopback=None, early_unpack=True):
        if rows not in ("union", "all"):
            raise ValueError(rows)
        if wire not in (None, "fp32", "bf16"):
            raise ValueError(wire)
        self.mode, self.wire, self.bucket_bytes, self.rows, self.param = mode, wire, bucket_bytes, rows, param
        self.sync_free, self.cap_margin, self.cap_slack = bool(sync_free), float(cap_margin), int(cap_slack)
        self.context = context  # the RasterContext whose backward feeds this reducer (None: the entering thread's default)
        # loopback = (world_size, union_rows): ONE process runs the step exactly as rank 0 of `world_size` would -- the
        # range-staged backward, gags_blended_mask, gags_compact_mask, pack and unpack of a block of `union_rows` rows (this
        # view's own rows, topped up to that count) -- with each collective replaced by two device copies of the block (the
        # bytes a reduce-scatter + all-gather move through this GPU's memory).  bench.py prices the N > 1 code path on one
        # GPU with it (`view_dp_overhead_ms`); the "sum" it returns is this rank's own gradient.
        self.loopback = loopback
        # early_unpack (device tensors, not with sync_free): the sums of a range that left the wire two ranges ago are written
        # into the gradient DURING the backward, on the compute stream, between the kernels of later ranges -- finish() is left
        # with the last two ranges.  The tensor autograd will consume then already holds the ranks' sum in those columns, which
        # is what every consumer is meant to see (an adopted gradient: as assigned by finish(); a copied / accumulated / cast one:
        # the sum instead of the local rows + finish()'s `sum - local`); written in stream order, never from the exchange stream.
        self.early_unpack = bool(early_unpack)
        self._cap_hint, self._pinned = {}, {}
        self.comm = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.rows_exchanged = None  # |union| of the last step (None: all rows)
        self.range_ms = None        # per range of the last step: ms of pack + collective on the exchange stream
        self._reset()

    def _reset(self):
        self._alias, self._covered, self._entries = None, 0, []
        self._mask, self._idx = None, None
        self._pos, self._union_ev = None, None
        self._count, self._padded = None, None
        self._alias_version = None

    def __enter__(self):
        from . import rasterization
        self._reset()
        # the hooks go into ONE RasterContext (the one given, else this thread's default): renders through any other context
        # -- an evaluation view, another thread -- are not touched
        self._ctx = self.context if self.context is not None else rasterization.default_context()
        self._prev = (self._ctx.grad_range_hook, self._ctx.grad_rows_hook)
        self._prev = self._prev + (self._ctx.grad_wire_hook,)
        self._ctx.grad_range_hook = self.on_range
        self._ctx.grad_rows_hook = self.on_rows if (self.rows == "union" and self._world() > 1) else None
        self._ctx.grad_wire_hook = self.wire_for_range
        return self

    def __exit__(self, *exc):
        self._ctx.grad_range_hook, self._ctx.grad_rows_hook, self._ctx.grad_wire_hook = self._prev
        return False

    def _world(self):
        return self.loopback[0] if self.loopback else world()

    def _collective(self, wire):
        """Sum over the ranks of the block `wire`, OUT OF PLACE: returns a new block, `wire` keeps this rank's rows."""
        if self.loopback:  # two passes over the block, like the shard exchange of a reduce-scatter + all-gather
            tmp = torch.empty_like(wire)
            tmp.copy_(wire)
            out = torch.empty_like(wire)
            out.copy_(tmp)
            return out
        return reduce_feature_grad_oop(wire, mode=self.mode, bucket_bytes=self.bucket_bytes)

    def _union_mask(self, mask):
        if not self.loopback:
            dist.all_reduce(mask, op=dist.ReduceOp.MAX)
            return
        # loop-back: rows of other ranks' views are emulated by marking further rows until the union has `union_rows`
        # (all on the device: a host readback here would drain the launch queue right at the start of the backward -- a stall
        # the real step, whose all-reduce is just another kernel on the exchange stream, does not have)
        want = int(self.loopback[1])
        if want:
            free = mask == 0
            short = want - mask.sum(dtype=torch.int64)  # rows still missing (device scalar; <= 0: nothing to add)
            mask |= (free & (torch.cumsum(free, 0, dtype=torch.int32) <= short)).to(mask.dtype)

    def on_rows(self, mask):
        """mask uint8 [N] of this rank's view; the union over the ranks is formed on the exchange stream right away
        (N bytes), its index list when the first range arrives (by then it has long finished: no stall)."""
        if self._mask is not None or self._entries:
            raise RuntimeError("OverlappedGradReducer: one backward per `with` block (call finish() between views)")
        if mask.is_cuda and self.comm is not None:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                self._union_mask(mask)
            mask.record_stream(self.comm)
        else:
            self._union_mask(mask)
        self._mask = mask

    def _union_rows(self):
        """Row list of the union (ascending; identical on every rank).  Device tensors: gags_compact_mask -- prefix sum +
        scatter on the exchange stream, no torch.nonzero.  What the host needs is the COUNT (it sizes the collective):
        * default: a 4-byte copy to pinned memory behind an event, waited for right here -- the host waits for the mask's
          all-reduce + three tiny kernels only (the compute stream keeps the whole first range queued meanwhile; the
          backward's own row-count readback has already passed this point);
        * sync_free=True: the block is sized by a capacity remembered from earlier steps (identical on every rank, because
          the union is), padded with rows of zeros (idx = -1), and the true count is read in finish(); a union larger than
          the capacity -- nothing was written out of bounds -- is exchanged again in finish() with all N rows."""
        if self._idx is not None or self._mask is None:
            return self._idx
        mask = self._mask
        if not mask.is_cuda:
            self._idx = torch.nonzero(mask).squeeze(1)
            self.rows_exchanged = int(self._idx.numel())
            return self._idx
        from . import _lib
        lib = _lib.load()
        n = mask.numel()
        dev = mask.device
        cap = n
        if self.sync_free and self._cap_hint.get(n):
            cap = min(n, int(self._cap_hint[n] * self.cap_margin) + self.cap_slack)
        idx = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
        pos = torch.empty(max(n, 1), dtype=torch.int32, device=dev)  # the inverse: position of row g in idx, or -1
        count = torch.empty(1, dtype=torch.int32, device=dev)
        sb = lib.gags_compact_mask_scratch_bytes(n)
        scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
        _lib.check(lib.gags_compact_mask_pos(n, _lib.ptr(mask), cap, _lib.ptr(idx), _lib.ptr(pos), _lib.ptr(count),
                                             _lib.ptr(scratch), sb, torch.cuda.current_stream().cuda_stream),
                   "gags_compact_mask_pos")
        self._pos = pos
        self._union_ev = torch.cuda.Event()
        self._union_ev.record()  # (on the exchange stream: the backward's reduce stage waits for it before it reads `pos`)
        if dev.index not in self._pinned:
            self._pinned[dev.index] = torch.empty(1, dtype=torch.int32).pin_memory()
        host = self._pinned[dev.index]
        host.copy_(count, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._count = (host, ev, count)
        if cap == n:  # exact: wait for the count now (see above) and cut the list to it
            ev.synchronize()
            c = int(host[0])
            self._cap_hint[n] = max(c, int(0.97 * self._cap_hint.get(n, 0)))
            self._idx, self.rows_exchanged, self._count = idx[:c], c, None
        else:
            self._idx, self.rows_exchanged, self._padded = idx[:cap], cap, cap
        return self._idx

    def _resolve_count(self):
        """sync_free: the union's true size, known by now; True when it exceeded the capacity the block was sized with."""
        if self._count is None:
            return False
        host, ev, _ = self._count
        ev.synchronize()
        c = int(host[0])
        n = self._mask.numel() if self._mask is not None else 0
        self._cap_hint[n] = max(c, int(0.97 * self._cap_hint.get(n, 0)))
        self._count = None
        over = c > self._padded
        self.rows_exchanged = c
        return over

    def wire_for_range(self, c0, c1):
        """RasterContext.grad_wire_hook: called by the staged backward BEFORE the reduce stage of range [c0, c1).  Returns
        (pos int32 [N], wire fp32 [|union|, c1 - c0]) -- the reduce kernel then writes the rows the ranks exchange straight
        into `wire` (gags_raster_bwd_colors_staged_wire), next to the gradient itself: no pack kernel re-reads the range --
        or None when the exchange packs for itself (all rows, the bf16 wire, host tensors).  The calling (compute) stream is
        made to wait for the union's index list, which the exchange stream builds from the all-reduced mask; the host waits
        for its COUNT here (it sizes the block) while the range's rows kernel is already queued."""
        if (self._world() == 1 or self.wire == "bf16" or self.rows != "union" or self._mask is None
                or not self._mask.is_cuda or self.comm is None):
            return None
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(self.comm):
            idx = self._union_rows()
        if idx is None or self._pos is None or idx.numel() == 0:  # (an empty union: nothing to write, the exchange packs 0 rows)
            return None
        cur.wait_event(self._union_ev)
        self._pos.record_stream(cur)
        wire = torch.empty(idx.numel(), c1 - c0, dtype=torch.float32, device=self._mask.device)
        if self._padded is not None:
            wire.zero_()  # sync_free: the rows between the union's count and the capacity the block was sized with
        wire.record_stream(self.comm)
        return self._pos, wire

    def _exchange(self, grad, c0, c1, wire=None):
        idx = self._union_rows()
        if idx is None:
            self.rows_exchanged = None
        if wire is None:
            wire = _pack_rows(grad, idx, c0, c1, torch.bfloat16 if self.wire == "bf16" else torch.float32)
        # out of place: `wire` stays this rank's rows (finish() needs them whenever autograd did not adopt the hook's tensor)
        return dict(c0=c0, c1=c1, wire=self._collective(wire), local=wire, idx=idx)

    def on_range(self, grad, c0, c1, wire=None):
        if any(e["c0"] < c1 and c0 < e["c1"] for e in self._entries) or (
                self._alias is not None and self._alias.data_ptr() != grad.data_ptr()):
            raise RuntimeError("OverlappedGradReducer: one backward per `with` block (call finish() between views)")
        if self._alias is None:
            self._alias_version = grad._version  # kernels write through raw pointers: only torch in-place ops bump it
        self._alias = grad
        self._covered += c1 - c0
        if self._world() == 1:
            return
        if grad.is_cuda and self.comm is not None:
            ev = torch.cuda.Event()
            ev.record()  # the range's kernels, on the compute stream
            with torch.cuda.stream(self.comm):
                # the union's index list BEFORE the exchange stream is made to wait for this range: its host sync
                # then covers the mask all-reduce only, while the GPU still has this whole range queued
                self._union_rows()
                self.comm.wait_event(ev)
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                e = self._exchange(grad, c0, c1, wire)  # reads `grad` (or the block the reduce stage wrote), writes private buffers only
                t1.record()
                e["ev"] = (t0, t1)
            grad.record_stream(self.comm)
        else:
            e = self._exchange(grad, c0, c1, wire)
        self._entries.append(e)
        if grad.is_cuda and self.comm is not None and self.early_unpack and not self.sync_free:
            cur = torch.cuda.current_stream()
            for old in self._entries[:-2]:  # (two ranges old: its exchange has had two ranges' worth of kernels to finish)
                if old.get("unpacked") or "ev" not in old:
                    continue
                cur.wait_event(old["ev"][1])
                _unpack_rows(self._alias, old["idx"], old["c0"], old["c1"], old["wire"], None)
                old["wire"].record_stream(cur)
                old["unpacked"] = True

    def finish(self, param_grad):
        """Bring the sum over the ranks into `param_grad` on the compute stream; returns True if the overlapped exchange
        was used (False: plain reduction of the whole gradient)."""
        cuda = param_grad.is_cuda and self.comm is not None
        if cuda:
            self._bwd_done = torch.cuda.Event(enable_timing=True)
            self._bwd_done.record()
        ws = self._world()
        used = ws > 1 and bool(self._entries) and self._covered == param_grad.shape[1]
        if self._entries and not used:
            raise RuntimeError(f"OverlappedGradReducer: the backward delivered {self._covered} of "
                               f"{param_grad.shape[1]} channels")
        if used and self._resolve_count():
            # sync_free: this step's union outgrew the remembered capacity; the surplus rows were never packed.  Same
            # decision on every rank (the union is identical): exchange every range again with all N rows, here
            # (the sum now lands in rows the backward's persistent-buffer flags do not cover: that buffer is not kept)
            getattr(self._ctx, "forget_all_kept", lambda: None)()
            if cuda:
                torch.cuda.current_stream().wait_stream(self.comm)
            for e in self._entries:
                e.pop("ev", None)
                wire = _pack_rows(self._alias, None, e["c0"], e["c1"], e["wire"].dtype)
                e.update(idx=None, local=wire, wire=self._collective(wire))
            self.rows_exchanged = None  # (all N rows went over the wire in the end)
        if used:
            # adopted: the parameter's gradient IS the tensor the hook saw (same storage, same shape) and no in-place op
            # touched it since (the alias shares its version counter) => it holds exactly this backward's local rows
            adopted = (self._alias is not None and self._alias.data_ptr() == param_grad.data_ptr()
                       and self._alias.shape == param_grad.shape and param_grad.dtype == self._alias.dtype
                       and param_grad._version == self._alias_version == self._alias._version)
            self.assigned = bool(adopted)
            for e in self._entries:
                if e.get("unpacked"):  # (written during the backward: early_unpack)
                    continue
                assign = adopted
                # range by range: the compute stream waits for THIS range's exchange only, so the sums of the early ranges are
                # written while the later ranges are still on the wire -- what stays exposed after the last exchange is one
                # range's unpack, not four
                if cuda and "ev" in e:
                    torch.cuda.current_stream().wait_event(e["ev"][1])
                elif cuda:
                    torch.cuda.current_stream().wait_stream(self.comm)
                _unpack_rows(param_grad, e["idx"], e["c0"], e["c1"], e["wire"], None if assign else e["local"])
                if cuda:
                    e["wire"].record_stream(torch.cuda.current_stream())
                    if e["local"] is not None:
                        e["local"].record_stream(torch.cuda.current_stream())
        elif ws > 1 and not self.loopback:
            if cuda:
                torch.cuda.current_stream().wait_stream(self.comm)
            reduce_feature_grad(param_grad, mode=self.mode, bucket_bytes=self.bucket_bytes)
        if cuda:
            torch.cuda.current_stream().wait_stream(self.comm)
            self._all_done = torch.cuda.Event(enable_timing=True)
            self._all_done.record()
        self._timed = [e["ev"] for e in self._entries if "ev" in e]
        self._entries, self._alias, self._mask, self._covered = [], None, None, 0
        return used

    def exposed_ms(self):
        if getattr(self, "_all_done", None) is None:
            return 0.0
        self._all_done.synchronize()
        self.range_ms = [float(a.elapsed_time(b)) for a, b in getattr(self, "_timed", [])]
        return float(self._bwd_done.elapsed_time(self._all_done))


def distributed_step(render_fn, cams, pc, bg, cotangents, mode="rs_ag"):
    """One step over `cams`.  mode "rs_ag" / "allreduce": render this rank's share of the views, backprop
    <render, G_v> for each, then reduce the feature gradient over ranks.  mode "channel": `pc` holds this
    rank's channel shard of the features ([N, c1-c0], see channel_shard) and `cotangents[v]` the matching
    channels of G_v; every view is rendered, the gradient of the shard accumulates locally, nothing is
    exchanged.  Trainable geometry (any of GEOMETRY_PARAMS requiring grad) is reduced too, as one packed [N,11] block.
    Returns the local sum of losses."""
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
    reduce_geometry_grads(pc, mode=mode)
    return total
