"""Two-rank steps with the real HIP kernels (both ranks on cuda:0, gloo transport: RCCL needs one GPU per rank
and the 8-GPU runs are the driver's).  SURVEY.md 4 "Distributed": the reduced feature gradient of the by-view
step equals the single-process sum over the same views; the by-channel step reproduces the single-process
gradient columns bit for bit without any exchange."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, D, W, H, VIEWS = 4000, 64, 160, 112, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup():
    from gags_amd import synthetic as syn
    dev = torch.device("cuda", 0)
    pc = syn.make_model(N, D, W, H, seed=3, device=dev, gen_device=dev, scale0=syn.SCALE0 * 6.0)
    pc.training_setup()
    cams = [syn.make_camera(W, H, view=v + 2, device=dev) for v in range(VIEWS)]
    G = [syn.make_cotangent(D, H, W, seed=10 + v, device=dev) for v in range(VIEWS)]
    return dev, pc, cams, G


def _single_process_grad():
    from gags_amd.gaussian_renderer import render
    dev, pc, cams, G = _setup()
    bg = torch.zeros(3, device=dev)
    pc._semantic_feature.grad = None
    for v in range(VIEWS):
        (render(cams[v], pc, None, bg, feature_mode=True)["render"] * G[v]).sum().backward()
    return pc._semantic_feature.grad.detach().cpu().numpy()


def _worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gags_amd.dist import channel_shard, distributed_step
    from gags_amd.gaussian_renderer import render
    dev, pc, cams, G = _setup()
    bg = torch.zeros(3, device=dev)
    if mode == "channel":
        c0, c1 = channel_shard(D)
        pc._semantic_feature = torch.nn.Parameter(pc._semantic_feature.detach()[:, c0:c1].contiguous())
        cots = [g[c0:c1].permute(1, 2, 0).contiguous().permute(2, 0, 1) for g in G]
        distributed_step(render, cams, pc, bg, cots, mode="channel")
    else:
        distributed_step(render, cams, pc, bg, G, mode="allreduce")  # gloo has no reduce-scatter for GPU tensors
    np.save(os.path.join(out_dir, f"grad_{mode}_{rank}.npy"), pc._semantic_feature.grad.detach().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run(mode, tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return [np.load(tmp_path / f"grad_{mode}_{r}.npy") for r in range(world)]


def test_view_sharded_step_equals_single_process_sum(tmp_path):
    ref = _single_process_grad()
    g0, g1 = _run("view", tmp_path)
    np.testing.assert_array_equal(g0, g1)  # every rank ends with the same reduced gradient
    # each view's gradient is bit-reproducible; only the order of the final sum over views may differ
    assert np.linalg.norm(g0.astype(np.float64) - ref) <= 1e-6 * np.linalg.norm(ref)
    assert np.abs(ref).max() > 0


def test_channel_sharded_step_is_bit_identical_and_exchange_free(tmp_path):
    ref = _single_process_grad()
    g0, g1 = _run("channel", tmp_path)
    assert g0.shape == (N, D // 2) and g1.shape == (N, D // 2)
    np.testing.assert_array_equal(np.concatenate([g0, g1], axis=1), ref)


def _overlap_step_worker(rank, world, port, out_dir):
    """By-view step with the REAL kernels and the overlapped, row-compacted exchange: each rank renders its view at
    D = 256 (two 128-channel ranges), the backward hands the reducer the mask of blended Gaussians and then the
    ranges; gloo moves the bytes (all-reduce: it has no reduce-scatter for device tensors)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gags_amd import synthetic as syn
    from gags_amd.dist import OverlappedGradReducer
    from gags_amd.gaussian_renderer import render
    dev = torch.device("cuda", 0)
    d = 256
    pc = syn.make_model(N, d, W, H, seed=3, device=dev, gen_device=dev, scale0=syn.SCALE0 * 6.0)
    pc.training_setup()
    cam = syn.make_camera(W, H, view=rank + 2, device=dev)
    G = syn.make_cotangent(d, H, W, seed=10 + rank, device=dev)
    bg = torch.zeros(3, device=dev)
    red = OverlappedGradReducer(mode="allreduce", rows="union", sync_free=True, cap_margin=1.0, cap_slack=16)

    def one_step():
        pc._semantic_feature.grad = None
        loss = (render(cam, pc, None, bg, feature_mode=True)["render"] * G).sum()
        with red:
            loss.backward()
        used = red.finish(pc._semantic_feature.grad)
        torch.cuda.synchronize()
        return used, pc._semantic_feature.grad.detach().cpu().numpy()

    used, g_first = one_step()          # first step of a shape: the exact count is waited for (no capacity yet)
    rows_first = red.rows_exchanged
    used2, g_cap = one_step()           # second step: capacity-sized block, padding rows, count read in finish()
    padded = red._padded
    red._cap_hint = {k: 8 for k in red._cap_hint}   # a union far above the remembered capacity: all-rows fallback
    used3, g_over = one_step()
    np.save(os.path.join(out_dir, f"ogr_{rank}.npy"), g_first)
    np.save(os.path.join(out_dir, f"ogr_cap_{rank}.npy"), g_cap)
    np.save(os.path.join(out_dir, f"ogr_over_{rank}.npy"), g_over)
    np.save(os.path.join(out_dir, f"ogr_meta_{rank}.npy"),
            np.array([int(used and used2 and used3), rows_first or -1, padded or -1, red.rows_exchanged or -1]))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_union_row_exchange_with_the_real_backward(tmp_path):
    from gags_amd import synthetic as syn
    from gags_amd.gaussian_renderer import render
    dev = torch.device("cuda", 0)
    d = 256
    pc = syn.make_model(N, d, W, H, seed=3, device=dev, gen_device=dev, scale0=syn.SCALE0 * 6.0)
    pc.training_setup()
    bg = torch.zeros(3, device=dev)
    per_view = []
    for v in range(2):
        pc._semantic_feature.grad = None
        cam = syn.make_camera(W, H, view=v + 2, device=dev)
        G = syn.make_cotangent(d, H, W, seed=10 + v, device=dev)
        (render(cam, pc, None, bg, feature_mode=True)["render"] * G).sum().backward()
        per_view.append(pc._semantic_feature.grad.detach().clone())
    ref = (per_view[0] + per_view[1]).cpu().numpy()
    union = int(((per_view[0] != 0).any(1) | (per_view[1] != 0).any(1)).sum())
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_overlap_step_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for r in range(world):
        used, rows, padded, rows_last = np.load(tmp_path / f"ogr_meta_{r}.npy")
        for tag in ("ogr", "ogr_cap", "ogr_over"):   # exact count / capacity-sized padded block / over-capacity fallback
            np.testing.assert_array_equal(np.load(tmp_path / f"{tag}_{r}.npy"), ref)   # two addends per element: exact
        assert used == 1 and rows == union and 0 < union < N
        assert union < padded < N and rows_last == -1   # (the over-capacity step re-sent ALL rows: rows_exchanged is None)


def _early_unpack_worker(rank, world, port, out_dir):
    """D = 512: four 128-channel ranges, so that the sums of the first two are written into the gradient DURING the backward
    (OverlappedGradReducer(early_unpack=True), the default) and finish() is left with the last two.  Four steps per rank:
    {early, late} x {the gradient adopted by autograd, a second consumer of the parameter in the graph}."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gags_amd import synthetic as syn
    from gags_amd.dist import OverlappedGradReducer
    from gags_amd.gaussian_renderer import render
    dev = torch.device("cuda", 0)
    d = 512
    pc = syn.make_model(N, d, W, H, seed=3, device=dev, gen_device=dev, scale0=syn.SCALE0 * 6.0)
    pc.training_setup()
    cam = syn.make_camera(W, H, view=rank + 2, device=dev)
    G = syn.make_cotangent(d, H, W, seed=10 + rank, device=dev)
    R = syn.make_cotangent(d, 1, N, seed=77, device=dev)[:, 0, :].t().contiguous()  # [N, d]: the second consumer's weights
    bg = torch.zeros(3, device=dev)
    for early in (True, False):
        red = OverlappedGradReducer(mode="allreduce", rows="union", early_unpack=early)
        for second in (False, True):
            pc._semantic_feature.grad = None
            loss = (render(cam, pc, None, bg, feature_mode=True)["render"] * G).sum()
            if second:
                loss = loss + (pc._semantic_feature * R).sum()
            with red:
                loss.backward()
            used = red.finish(pc._semantic_feature.grad)
            torch.cuda.synchronize()
            assert used and (red.assigned is (not second))
            np.save(os.path.join(out_dir, f"eu_{int(early)}_{int(second)}_{rank}.npy"), pc._semantic_feature.grad.detach().cpu().numpy())
    np.save(os.path.join(out_dir, f"eu_R_{rank}.npy"), R.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sums_written_during_the_backward_equal_the_sums_written_by_finish(tmp_path):
    from gags_amd import synthetic as syn
    from gags_amd.gaussian_renderer import render
    dev = torch.device("cuda", 0)
    d = 512
    pc = syn.make_model(N, d, W, H, seed=3, device=dev, gen_device=dev, scale0=syn.SCALE0 * 6.0)
    pc.training_setup()
    bg = torch.zeros(3, device=dev)
    per_view = []
    for v in range(2):
        pc._semantic_feature.grad = None
        cam = syn.make_camera(W, H, view=v + 2, device=dev)
        G = syn.make_cotangent(d, H, W, seed=10 + v, device=dev)
        (render(cam, pc, None, bg, feature_mode=True)["render"] * G).sum().backward()
        per_view.append(pc._semantic_feature.grad.detach().clone())
    ref = (per_view[0] + per_view[1]).cpu().numpy()
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_early_unpack_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for r in range(world):
        R = np.load(tmp_path / f"eu_R_{r}.npy")
        for early in (0, 1):
            np.testing.assert_array_equal(np.load(tmp_path / f"eu_{early}_0_{r}.npy"), ref)   # adopted: assigned sums, exact
            # a second consumer: autograd sums the two terms itself; the regulariser's term stays rank-local.  The early ranges
            # hold R + sum (one addition), the late ones (R + local) + (sum - local): equal up to an fp32 rounding or two
            got = np.load(tmp_path / f"eu_{early}_1_{r}.npy")
            np.testing.assert_allclose(got, ref + R, rtol=0, atol=4e-6 * max(1.0, float(np.abs(ref).max())))
        # ... and where both write by assignment the two orders agree bit for bit
        np.testing.assert_array_equal(np.load(tmp_path / f"eu_1_0_{r}.npy"), np.load(tmp_path / f"eu_0_0_{r}.npy"))


@pytest.mark.parametrize("n,p", [(1, 1.0), (7, 0.5), (2048, 0.3), (2049, 0.01), (100_003, 0.27), (1_500_000, 0.3), (5000, 0.0)])
def test_compact_mask_is_the_ascending_nonzero_list(n, p):
    """gags_compact_mask against torch.nonzero: ascending row numbers, -1 padding behind the count, a capacity below the
    count drops the surplus without writing out of bounds."""
    from gags_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(n)
    mask = (torch.rand(n, generator=g) < p).to(torch.uint8).to(dev) * 3   # any non-zero byte counts
    want = torch.nonzero(mask).squeeze(1)
    sb = lib.gags_compact_mask_scratch_bytes(n)
    scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
    for cap in (n, max(int(want.numel()) // 2, 0), int(want.numel()) + 5):
        idx = torch.full((cap + 4,), 12345, dtype=torch.int64, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.gags_compact_mask(n, _lib.ptr(mask), cap, _lib.ptr(idx), _lib.ptr(count), _lib.ptr(scratch), sb, None),
                   "gags_compact_mask")
        torch.cuda.synchronize()
        c = int(count.item())
        assert c == want.numel()
        k = min(c, cap)
        assert torch.equal(idx[:k], want[:k])
        assert bool((idx[k:cap] == -1).all()) and bool((idx[cap:] == 12345).all())
        # ... and with the inverse map: pos[r] = position of row r in idx, -1 for unset rows and for rows past the capacity
        idx2 = torch.full((cap + 4,), 12345, dtype=torch.int64, device=dev)
        pos = torch.full((n + 3,), 777, dtype=torch.int32, device=dev)
        _lib.check(lib.gags_compact_mask_pos(n, _lib.ptr(mask), cap, _lib.ptr(idx2), _lib.ptr(pos), _lib.ptr(count),
                                             _lib.ptr(scratch), sb, None), "gags_compact_mask_pos")
        torch.cuda.synchronize()
        assert torch.equal(idx2, idx) and int(count.item()) == c
        expect = torch.full((n,), -1, dtype=torch.int32, device=dev)
        expect[want[:k]] = torch.arange(k, dtype=torch.int32, device=dev)
        assert torch.equal(pos[:n], expect) and bool((pos[n:] == 777).all())


def test_reduce_stage_writes_the_exchanged_rows_itself():
    """gags_raster_bwd_colors_staged_wire (round 6): under the by-view step the reduce kernel of a channel range also writes
    the block the ranks exchange -- row pos[g] of a dense [union rows, range] fp32 block for every Gaussian of the union --
    so no pack kernel re-reads the gradient.  Driven through the hooks of a RasterContext as gags_amd/dist.py drives them:
    every block equals the gradient's rows (zeros for union rows this view did not touch), with 128- and 256-channel ranges
    and with the rows kernel launched per range or per group of ranges; the gradient itself is bit-identical to the plain
    backward's."""
    from gags_amd import _lib, synthetic as syn
    from gags_amd.gaussian_renderer import render
    from gags_amd.rasterization import RasterContext
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n, d, w, h = 6000, 512, 200, 138
    pc = syn.make_model(n, d, w, h, seed=4, device=dev, scale0=syn.SCALE0 * 5)
    pc.training_setup()
    cam = syn.make_camera(w, h, view=1, device=dev)
    G = syn.make_cotangent(d, h, w, seed=2, device=dev)
    bg = torch.zeros(3, device=dev)
    (render(cam, pc, None, bg, feature_mode=True)["render"] * G).sum().backward()
    plain = pc._semantic_feature.grad.clone()
    for spec, group in ((128, 256), (128, 128), (256, 256), ((256, 128, 128), 512)):
        ctx = RasterContext()
        ctx.grad_range_channels, ctx.grad_rows_group = spec, group
        state = {"blocks": []}

        def on_rows(mask):
            extra = torch.zeros_like(mask)
            extra[::7] = 1   # rows of "other ranks": in the union, untouched by this view
            m = mask | extra
            idx = torch.empty(n, dtype=torch.int64, device=dev)
            pos = torch.empty(n, dtype=torch.int32, device=dev)
            count = torch.zeros(1, dtype=torch.int32, device=dev)
            sb = lib.gags_compact_mask_scratch_bytes(n)
            scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
            _lib.check(lib.gags_compact_mask_pos(n, _lib.ptr(m), n, _lib.ptr(idx), _lib.ptr(pos), _lib.ptr(count), _lib.ptr(scratch),
                                                 sb, torch.cuda.current_stream().cuda_stream), "gags_compact_mask_pos")
            c = int(count.item())
            state.update(idx=idx[:c], pos=pos)

        def wire_hook(c0, c1):
            return state["pos"], torch.full((state["idx"].numel(), c1 - c0), float("nan"), device=dev)

        def on_range(grad, c0, c1, wire=None):
            assert wire is not None
            state["blocks"].append((c0, c1, wire))

        ctx.grad_rows_hook, ctx.grad_wire_hook, ctx.grad_range_hook = on_rows, wire_hook, on_range
        pc._semantic_feature.grad = None
        (render(cam, pc, None, bg, feature_mode=True, context=ctx)["render"] * G).sum().backward()
        torch.cuda.synchronize()
        g = pc._semantic_feature.grad
        assert torch.equal(g, plain), (spec, group)
        assert sum(c1 - c0 for c0, c1, _ in state["blocks"]) == d and 0 < state["idx"].numel() < n
        for c0, c1, wire in state["blocks"]:
            assert torch.equal(wire, g[state["idx"], c0:c1]), (spec, group, c0)


def _nccl_worker(rank, world, port, q):
    """The default exchange of the by-view step on DEVICE tensors over RCCL: bucketed reduce-scatter + in-place
    all-gather (gags_amd/dist.py:reduce_feature_grad, mode rs_ag), one GPU per rank."""
    import os
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from gags_amd.dist import OverlappedGradReducer, reduce_feature_grad
    n, d = 10007, 512
    grads = [torch.randn(n, d, generator=torch.Generator().manual_seed(3 + r)) for r in range(world)]
    expect = sum(grads).to(dev)
    g = grads[rank].to(dev)
    reduce_feature_grad(g, mode="rs_ag", bucket_bytes=1 << 20)
    ok1 = torch.allclose(g, expect, rtol=1e-6, atol=1e-6)
    g2 = grads[rank].to(dev)
    red = OverlappedGradReducer(mode="rs_ag")
    with red:  # the hook the staged backward calls per channel range lives on the RasterContext (no module globals)
        from gags_amd import rasterization
        alias = g2.detach()
        for c0 in range(0, d, 128):
            rasterization.default_context().grad_range_hook(alias, c0, c0 + 128)
    used = red.finish(g2)
    torch.cuda.synchronize()
    ok2 = used and torch.allclose(g2, expect, rtol=1e-6, atol=1e-6)
    q.put((rank, bool(ok1), bool(ok2)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rs_ag_on_device_tensors_over_rccl():
    """Needs two GPUs (skipped on the one-GPU test box; the driver's multi-GPU runs exercise the same code through
    bench.py --gpus N)."""
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(a and b for _, a, b in res)


def test_bench_two_ranks_under_torch_distributed_run():
    """The whole N > 1 control flow of bench.py exactly as the driver launches it (python -m torch.distributed.run
    --nproc-per-node 2 ... bench.py --gpus 2), both ranks on the one GPU of this box with gloo as the transport (RCCL
    needs one GPU per rank): overlapped by-view step, union-of-rows exchange, the JSON line's multi-GPU fields."""
    import json
    import subprocess
    env = dict(os.environ, GAGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--config", "C2", "--n-gaussians", "20000", "--feature-dim", "256",  # (torchrun's argparse claims a bare --n)
            "--no-cpu-baseline", "--no-heavy"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    ge = line["config"]["grad_exchange"]
    assert ge["overlapped_with_backward"] is True and ge["rows"] == "union"
    assert ge["exposed_ms_last_step"] is not None and ge["exposed_ms_last_step"] >= 0.0
    assert 0 < ge["rows_exchanged_last_step"] <= 20000
    assert len(ge["range_exchange_ms_last_step"]) == 2  # two 128-channel ranges at D = 256
    assert "view-dp2" in line["config"]["parallelism"]
    assert ge["collective_ms_per_step"]["allreduce"] > 0 and "version" in ge["rccl"]
    assert "rs_ag" in ge["collective_ms_per_step"]  # the other collective was attempted (gloo refuses it on device tensors: null)


@pytest.mark.parametrize("d,c0,c1", [(256, 128, 256), (37, 5, 30), (130, 2, 130), (64, 0, 64)])
@pytest.mark.parametrize("gdt", [torch.float32, torch.float16])
@pytest.mark.parametrize("wdt", [torch.float32, torch.float16, torch.bfloat16])
def test_pack_and_unpack_rows_against_torch_indexing(d, c0, c1, gdt, wdt):
    """gags_pack_rows / gags_unpack_rows (the row-compacted exchange's gather and scatter) against torch indexing:
    vector path (D, c0, width multiples of 4) and element-wise path, every gradient / wire type, assign and delta."""
    from gags_amd.dist import _pack_rows, _unpack_rows
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(d * 7 + c0)
    n = 1000
    grad = torch.randn(n, d, generator=g).to(gdt).to(dev)
    idx = torch.randperm(n, generator=g)[:317].sort().values.to(dev)
    # padding rows (idx = -1, gags_compact_mask's capacity-sized lists): zeros on the wire, skipped on the way back
    padded = torch.cat([idx, torch.full((9,), -1, dtype=idx.dtype, device=dev)])
    wire_p = _pack_rows(grad, padded, c0, c1, wdt)
    assert torch.equal(wire_p[:317], grad[idx, c0:c1].to(wdt)) and bool((wire_p[317:] == 0).all())
    before = grad.clone()
    _unpack_rows(grad, padded, c0, c1, wire_p)
    assert torch.equal(grad[:, c0:c1].float(), before[:, c0:c1].to(wdt).to(gdt).float().where(
        torch.zeros(n, 1, dtype=torch.bool, device=dev).index_fill_(0, idx, True), before[:, c0:c1].float()))
    grad.copy_(before)
    for ix in (idx, None):
        wire = _pack_rows(grad, ix, c0, c1, wdt)
        want = (grad[:, c0:c1] if ix is None else grad[ix, c0:c1]).to(wdt)
        assert torch.equal(wire, want)
        # assign
        new = torch.randn(wire.shape, generator=g).to(wdt).to(dev)
        tgt = grad.clone()
        _unpack_rows(tgt, ix, c0, c1, new)
        ref = grad.clone()
        if ix is None:
            ref[:, c0:c1] = new.to(gdt)
        else:
            ref[ix, c0:c1] = new.to(gdt)
        assert torch.equal(tgt, ref)
        # delta: grad += wire - local, formed in fp32, rounded once
        tgt = grad.clone()
        _unpack_rows(tgt, ix, c0, c1, new, local=wire)
        ref = grad.clone().float()
        delta = new.float() - wire.float()
        if ix is None:
            ref[:, c0:c1] += delta
        else:
            ref[ix, c0:c1] += delta
        assert torch.equal(tgt, ref.to(gdt))


def test_two_renderers_in_one_process_do_not_share_hooks():
    """"re-entrant per stream, no global state".  One model trains inside an OverlappedGradReducer block (its
    backward is asked for the gradient range by range), a second one is rendered and differentiated in between through its
    own RasterContext: the reducer sees only the first model's ranges, the second model's gradient is the plain one-shot
    gradient, and both equal what each gives alone."""
    import torch
    from gags_amd import synthetic as syn
    from gags_amd.dist import OverlappedGradReducer
    from gags_amd.gaussian_renderer import render
    from gags_amd.rasterization import RasterContext
    dev = torch.device("cuda", 0)
    w, h, d = 160, 112, 256
    cam = syn.make_camera(w, h, view=3, device=dev)
    bg = torch.zeros(3, device=dev)
    G = syn.make_cotangent(d, h, w, seed=1).to(dev)
    pcs = [syn.make_model(3000, d, w, h, seed=sd, device=dev, scale0=syn.SCALE0 * 4) for sd in (5, 6)]
    for pc in pcs:
        pc.training_setup()

    def alone(pc):
        pc._semantic_feature.grad = None
        (render(cam, pc, None, bg, feature_mode=True, context=RasterContext())["render"] * G).sum().backward()
        return pc._semantic_feature.grad.clone()

    ref = [alone(pc) for pc in pcs]
    for pc in pcs:
        pc._semantic_feature.grad = None
    train_ctx, eval_ctx = RasterContext(), RasterContext()
    red = OverlappedGradReducer(mode="allreduce", rows="all", context=train_ctx)
    seen = []
    with red:
        inner = train_ctx.grad_range_hook
        train_ctx.grad_range_hook = lambda g, c0, c1: (seen.append((g.shape[0], c0, c1)), inner(g, c0, c1))[1]
        pk_train = render(cam, pcs[0], None, bg, feature_mode=True, context=train_ctx)
        pk_eval = render(cam, pcs[1], None, bg, feature_mode=True, context=eval_ctx)   # the other renderer, in between
        (pk_eval["render"] * G).sum().backward()
        (pk_train["render"] * G).sum().backward()
        assert eval_ctx.grad_range_hook is None
    red.finish(pcs[0]._semantic_feature.grad)
    torch.cuda.synchronize()
    assert [(c0, c1) for _, c0, c1 in seen] == [(0, 128), (128, 256)]            # only the training model's backward
    assert train_ctx.grad_range_hook is None                                       # restored on exit
    assert torch.equal(pcs[1]._semantic_feature.grad, ref[1])
    assert torch.equal(pcs[0]._semantic_feature.grad, ref[0])                      # world size 1: the sum is the local gradient
