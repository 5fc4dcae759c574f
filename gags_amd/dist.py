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
            dist.reduce_scatter_tensor(shard, b)
            dist.all_gather_into_tensor(b, shard.clone())
        if main < numel:
            dist.all_reduce(flat[main:])
    else:
        raise ValueError(mode)
    if average:
        flat.div_(ws)
    return grad


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
