#!/usr/bin/env python
"""bench.py -- feature-raster fwd+bwd views/s (BASELINE.json metric) on N GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = the harness counterpart of train.py:134-174 on ONE view per
GPU: render(cam, gaussians, pipe, bg, feature_mode=True) -> loss = <render, G> with a fixed
random cotangent G -> loss.backward() -> (N>1) sum of d loss/d _semantic_feature over ranks
(RCCL).  Inputs are synthetic (gags_amd/synthetic.py), resident in HBM before the timed
region.  Workload at N=1 = BASELINE.json configs[2] "C3": 1.5 M Gaussians, 1920x1080, D=512 --
the configuration the metric is quoted on; it fits one MI355X.

Prints ONE JSON line (rank 0) with the driver's contract keys plus `roofline` (dominant
kernel, measured with HIP events inside the timed region) and `cpu_baseline` (the CPU oracle
timed on a bounded tile sample of the same workload, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 MFMA == fp32 vector peak (dense)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak


class _CotangentLoss(torch.autograd.Function):
    """loss = <x, G>; terminal loss of the harness: backward hands G itself to the rasterizer
    (d loss / d x = G exactly), so no extra elementwise kernels sit inside the timed region."""

    @staticmethod
    def forward(ctx, x, G):
        from gags_amd import _lib
        ctx.save_for_backward(G)
        x, G = x.contiguous(), G.contiguous()  # both already are: [H,W,D] memory
        lib = _lib.load()
        out = torch.empty(1, device=x.device)
        nb = lib.gags_dot_scratch_bytes()
        scratch = torch.empty(nb, dtype=torch.uint8, device=x.device)
        _lib.check(lib.gags_dot_f32(x.numel(), _lib.ptr(x), _lib.ptr(G), _lib.ptr(out), _lib.ptr(scratch), nb,
                                    torch.cuda.current_stream().cuda_stream), "gags_dot_f32")
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (G,) = ctx.saved_tensors
        return G, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C5", "C3H"])
    ap.add_argument("--no-heavy", action="store_true",
                    help="skip the second measurement (C3 geometry at SURVEY 8d's literal splat scale, 'C3H')")
    ap.add_argument("--n", "--n-gaussians", dest="n", type=int, default=None)
    ap.add_argument("--d", "--feature-dim", dest="d", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=30.0,
                    help="target CPU time of the oracle sample (the whole view when it fits: ~7 s on the GPU box's 128 cores)")
    ap.add_argument("--no-allreduce", action="store_true", help="(debug) skip the gradient reduction at N>1")
    ap.add_argument("--grad-reduce", default="rs_ag", choices=["rs_ag", "allreduce"],
                    help="by-view step: bucketed reduce-scatter + all-gather (default) or plain all-reduce")
    ap.add_argument("--no-overlap", action="store_true",
                    help="by-view step: exchange the whole gradient after the backward instead of range by range during it")
    ap.add_argument("--rows", default="union", choices=["union", "all"],
                    help="by-view step: exchange only the rows of Gaussians that blended in some rank's view (exact; default) or all N rows")
    ap.add_argument("--wire", default="fp32", choices=["fp32", "bf16"],
                    help="by-view step: dtype of the gradient on the wire (bf16: opt-in, ~1e-2 relative error, halves the bytes)")
    ap.add_argument("--raster-flags", type=int, default=0,
                    help="gags_amd._lib flags; 64 = GAGS_BWD_F32MFMA (backward contraction on the fp32 matrix instructions "
                         "instead of the default fp32-equivalent split operands on the 16-bit matrix cores)")
    ap.add_argument("--parallel", default="view", choices=["auto", "view", "channel"],
                    help="N>1: one view per GPU + gradient exchange (view: north_star's decomposition, the default), "
                         "every GPU renders all N views for its channel shard with no exchange (channel), or whichever "
                         "two probe steps show faster (auto)")
    return ap.parse_args()


def cpu_baseline(pc, cam, d, width, height, target_s):
    """Time the CPU oracle (oracle/gags_oracle.c, kind 'port': the reference has no CPU path and
    no compilable native source) on the same C3 inputs: projection + binning in
    full, raster fwd + colours-only bwd on every `tile_step`-th tile, scaled to the whole view."""
    import numpy as np
    from oracle import oracle as orc
    from gags_amd import synthetic as syn
    orc.build()
    vm, K = syn.camera_matrices(cam)
    means = pc.get_xyz.detach().cpu().numpy()
    quats = pc.get_rotation.detach().cpu().numpy()
    scales = pc.get_scaling.detach().cpu().numpy()
    opac = pc.get_opacity.detach().cpu().numpy().reshape(-1)
    feats = pc.get_semantic_feature.detach().cpu().numpy()
    bg = np.zeros(d, np.float32)
    t0 = time.perf_counter()
    radii, means2d, depths, conics = orc.project_fwd(means, quats, scales, vm.cpu().numpy(), K, width, height)
    b = orc.tile_bin(means2d, radii, depths, width, height)
    t_bin = time.perf_counter() - t0
    n_tiles = b["tile_width"] * b["tile_height"]
    rng = np.random.default_rng(1)

    def sample(step):
        v_out = np.zeros((height, width, d), np.float32)
        # cotangent only where the sampled tiles are (others are never read)
        t1 = time.perf_counter()
        out, alphas, last, st = orc.raster_fwd(means2d, conics, opac, feats, bg, width, height, b["isect_offsets"],
                                               b["flatten_ids"], 0, step)
        t_f = time.perf_counter() - t1
        v_out[:] = 1.0
        t2 = time.perf_counter()
        orc.raster_bwd(means2d, conics, opac, feats, bg, width, height, b["isect_offsets"], b["flatten_ids"],
                       alphas, last, v_out, None, colors_only=True, tile_begin=0, tile_step=step)
        t_b = time.perf_counter() - t2
        return t_f + t_b

    step = max(1, n_tiles // 512)   # pilot: enough tiles to load every host thread a few times (64 tiles on 128
    t_s = sample(step)              # threads over-estimated the per-tile time 2x); also warms page cache / allocator
    n_s = len(range(0, n_tiles, step))
    per_tile = t_s / n_s
    want = max(1, min(n_tiles, int(target_s / max(per_tile, 1e-9))))
    step2 = max(1, n_tiles // want)
    if step2 < step:
        t_s = sample(step2)
        step = step2
        n_s = len(range(0, n_tiles, step))
    t_view = t_bin + t_s * (n_tiles / n_s)
    return {
        "value": 1.0 / t_view, "unit": "views/s", "cores": orc.max_threads(), "host": host_cpu(), "kind": "port",
        "sample": (f"oracle/gags_oracle.c (OpenMP, fp32): projection+binning of the full view ({t_bin:.2f} s) + raster "
                   f"fwd + colours-only bwd on every {step}-th tile ({n_s} of {n_tiles} tiles, {t_s:.2f} s), "
                   f"raster time scaled by {n_tiles / n_s:.1f}x"),
    }


def iteration_d16(dev, n, width, height, steps):
    """One train.py:142-174 iteration at D = 16 (tools/decoder_bench.py's loop, inside bench.py): ms per iteration with the
    decoders in the tier matched to the reference's arithmetic (bf16x2: operands as two bf16 terms, 16 significand bits --
    the reference's convolutions run in TF32, 11 bits, by PyTorch's default) and, labelled as NARROWER than the reference,
    in plain bf16."""
    from gags_amd import decoders as D, synthetic as syn
    from gags_amd.decoders import CNN_decoder, CNN_scale_decoder
    from gags_amd.distill import distillation_loss
    from gags_amd.gaussian_renderer import render
    d = 16
    pc = syn.make_model(n, d, width, height, seed=0, device=dev, gen_device=dev)
    pc.training_setup()
    cam = syn.make_camera(width, height, device=dev)
    bg = torch.zeros(3, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    n_emb = 300
    img_embed = torch.nn.functional.normalize(torch.randn(n_emb, 512, device=dev, generator=g), dim=-1)
    seg = torch.randint(-1, n_emb, (4, height // 8 + 1, width // 8 + 1), device=dev, generator=g).float()
    seg = seg.repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :height, :width].contiguous()
    out = {"workload": f"train.py:142-174 iteration: {n} Gaussians, {width}x{height}, D=16 -> CNN_scale_decoder + CNN_decoder "
                       "(16 -> 512) -> distillation losses (all three terms) -> backward through decoders and rasterizer"}
    k = max(3, min(steps, 8))
    for precision in ("bf16x2", "f16", "bf16"):
        dec, sdec = CNN_decoder(16, 512, precision).to(dev), CNN_scale_decoder(16, 3, precision).to(dev)

        def iteration(marks=None):
            D.invalidate_packed()  # as after an optimizer step: the decoders' weights are repacked every iteration
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            fmap = render(cam, pc, None, bg, feature_mode=True)["render"]
            ev[1].record()
            loss, _ = distillation_loss(fmap, seg, img_embed, dec, sdec, iteration=20000, fused_head=True)
            ev[2].record()
            for m in (dec, sdec):
                m.zero_grad(set_to_none=True)
            pc._semantic_feature.grad = None
            loss.backward()
            ev[3].record()
            if marks is not None:
                marks.append(ev)

        for _ in range(2):
            iteration()
        torch.cuda.synchronize()
        marks = []
        t0 = time.perf_counter()
        for _ in range(k):
            iteration(marks)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / k
        st = {nm: sum(e[i].elapsed_time(e[i + 1]) for e in marks) / k
              for i, nm in enumerate(("render16", "decoders_and_losses_fwd", "backward"))}
        out[precision] = {"ms_per_iteration": ms, "iterations_per_s": 1e3 / ms, "steps": k, "stages_ms": st,
                          "precision_note": {
                              "bf16x2": "operands as two bf16 terms (16 significand bits), three matrix terms per product, fp32 "
                                        "accumulation: at or above the TF32 (11-bit) convolutions the reference runs",
                              "f16": "IEEE-half operands: TF32's own 11-bit significand, fp32 accumulation, activations stored as "
                                     "half, gradients scaled by a device-chosen power of two (half has 5 exponent bits); within 2x of "
                                     "an emulated TF32 chain's distance to fp32 at every tensor (tests/test_decoders_gpu.py)",
                              "bf16": "plain bf16 operands (8 significand bits): NARROWER than the reference's TF32 -- reported "
                                      "for comparison, not a creditable number"}[precision]}
        del dec, sdec
        torch.cuda.empty_cache()
    return out


def rccl_info():
    """What decides the collective's algorithm on this stack: library version and the NCCL_* / RCCL_* environment."""
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        ver = None
    env = {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_")) or k == "HSA_ENABLE_IPC_MODE_LEGACY"}
    return {"version": ver, "NCCL_ALGO": os.environ.get("NCCL_ALGO"), "NCCL_PROTO": os.environ.get("NCCL_PROTO"), "env": env}


def host_cpu():
    """Sockets x cores x threads and the model name of the host (cpu_baseline.cores = OpenMP threads used)."""
    try:
        phys, cores, model, logical = set(), set(), "", 0
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                logical += 1
            elif k == "model name":
                model = v
            elif k == "physical id":
                pid = v; phys.add(v)
            elif k == "core id":
                cid = v; cores.add((pid, cid))
        return {"model": model, "sockets": len(phys) or 1, "physical_cores": len(cores) or logical, "hw_threads": logical}
    except OSError:
        return None


# kernels behind each bracketed stage, by their short rocprofv3 names (tools/pmc_summary.py)
# (matched as prefixes: template arguments differ between rounds, e.g. "raster_fwd_feat<4, false>")
STAGE_KERNELS = {
    "raster_weights": ("raster_weights_kernel",),
    "raster_fwd_feat": ("raster_fwd_feat_x16",),   # (GAGS_FWD_EXACT: "raster_fwd_feat<4")
    "bwd_rows": ("raster_bwd_rows",),
    "bwd_reduce": ("reduce_rows_kernel",),
}
# flops one matrix instruction of each kernel issues (SQ_INSTS_MFMA x this = issued matrix work per launch)
MFMA_FLOP = {"raster_fwd_feat": 2 * 32 * 32 * 16,   # v_mfma_f32_32x32x16_bf16 (six terms per product)
             "bwd_rows": 2 * 32 * 32 * 16}           # v_mfma_f32_32x32x16_f16
F16_MATRIX_PEAK_TFLOPS = 2516.6   # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak


def _traffic_of(traffic, prefix, key="hbm_bytes"):
    """Per-launch counter (default: HBM bytes) of the kernel whose short rocprofv3 name starts with `prefix` (None when
    absent or ambiguous)."""
    hits = [v.get(key) for k, v in traffic.items() if k.startswith(prefix)]
    return hits[0] if len(hits) == 1 else None


def pmc_traffic(args):
    """Per-kernel HBM bytes per launch measured with rocprofv3 --pmc on the default workload; None elsewhere."""
    import glob
    if args.config != "C3" or args.n or args.d or args.raster_flags:
        return {}, None
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_c3_pmc_traffic.json")))
    if not files:
        return {}, None
    return json.load(open(files[-1])), os.path.join("profiles", os.path.basename(files[-1]))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    local_rank %= torch.cuda.device_count()  # (debugging on fewer GPUs than ranks; the driver runs one rank per GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("GAGS_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; gloo only for debugging
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
            args.grad_reduce = "allreduce"  # gloo has no reduce-scatter for device tensors

    from gags_amd import _lib, profiler, synthetic as syn
    from gags_amd.gaussian_renderer import render
    from gags_amd.dist import OverlappedGradReducer, reduce_feature_grad, reduce_geometry_grads
    _lib.load()
    exposed = []  # per step: ms the compute stream waited for the gradient exchange after the backward (by-view, N>1)

    cfg = dict(syn.CONFIGS[args.config])
    if args.n:
        cfg["n"] = args.n
    if args.d:
        cfg["d"] = args.d
    n, d, width, height = cfg["n"], cfg["d"], cfg["width"], cfg["height"]
    scale0 = cfg.get("scale0", syn.SCALE0)

    from gags_amd.dist import channel_shard

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def build(mode, grad_reduce=None, scale0=scale0, cache=False, share=None):
        """(step function, model, camera of the last view, local feature width) of one decomposition (gags_amd/dist.py).
        view   : Gaussians + features replicated (same seed on every rank), one yawed view per rank (C4's cameras),
                 reduce-scatter + all-gather (or all-reduce) of the feature gradient at step end.
        channel: every rank renders all `world` views for its channel shard; no exchange."""
        grad_reduce = grad_reduce or args.grad_reduce
        first = max(0, (8 - world) // 2)  # the `world` middle cameras of C4's eight (yaw (v - 3.5) * 5 degrees)
        if mode == "channel":
            c0, c1 = channel_shard(d)
            dl = c1 - c0
            views = [first + v for v in range(world)]
        else:
            dl = d
            views = [first + rank] if world > 1 else [None]
        if share is None:
            pc_ = syn.make_model(n, dl, width, height, seed=0, device=dev, gen_device=dev, scale0=scale0)
        else:  # (probe of another decomposition: the same geometry tensors, a feature table of this mode's width)
            from gags_amd.scene import GaussianModel
            pc_ = GaussianModel.from_tensors(share._xyz.data, share._scaling.data, share._rotation.data, share._opacity.data,
                                             share._features_dc.data, share._features_rest.data,
                                             share._semantic_feature.data[:, :dl].contiguous())
        pc_.training_setup()
        # the getters (exp / normalize / sigmoid of the frozen geometry) are evaluated on EVERY render, inside the timed
        # region, as the reference does (scene/gaussian_model.py:116-139) -- fused into the projection kernel
        pc_.cache_activations(cache)
        cams = [syn.make_camera(width, height, view=(v % 8) if v is not None else None, device=dev) for v in views]
        G_ = syn.make_cotangent(dl, height, width, seed=1, device=dev)  # [D,H,W] view of [H,W,D] memory

        exchange = mode == "view" and world > 1 and not args.no_allreduce
        overlap = exchange and not args.no_overlap
        ex = {"mode": grad_reduce, "red": None}

        def set_collective(name):
            """Switch the by-view step's collective (same model, same cameras): rs_ag | allreduce."""
            ex["mode"] = name
            ex["red"] = (OverlappedGradReducer(mode=name, wire=args.wire, rows=args.rows, param=pc_._semantic_feature)
                         if overlap else None)
        set_collective(grad_reduce)

        def step_():
            red = ex["red"]
            pc_._semantic_feature.grad = None
            for cam_ in cams:
                pkg_ = render(cam_, pc_, None, bg, feature_mode=True, raster_flags=args.raster_flags)
                loss = _CotangentLoss.apply(pkg_["render"].permute(1, 2, 0), G_.permute(1, 2, 0))
                if red is not None:
                    with red:  # the feature gradient is exchanged range by range while later ranges are computed
                        loss.backward()
                    red.finish(pc_._semantic_feature.grad)
                    exposed[:] = [red]
                else:
                    loss.backward()
            if exchange and red is None:
                if args.wire == "bf16":
                    g16 = pc_._semantic_feature.grad.to(torch.bfloat16)
                    reduce_feature_grad(g16, mode=ex["mode"])
                    pc_._semantic_feature.grad.copy_(g16)
                else:
                    reduce_feature_grad(pc_._semantic_feature.grad, mode=ex["mode"])
            if exchange:  # trainable geometry (not the GAD stage): its gradients travel as one packed [N,11] block
                reduce_geometry_grads(pc_, mode=ex["mode"])
            return pkg_
        step_.set_collective = set_collective
        return step_, pc_, cams[-1], dl

    bg = torch.zeros(3, device=dev)
    mode, probe = "view", None
    if world > 1 and args.parallel != "view":
        lo, hi = channel_shard(d)
        widths = torch.tensor([hi - lo], device=dev)
        torch.distributed.all_reduce(widths, op=torch.distributed.ReduceOp.MIN)
        channel_ok = int(widths.item()) >= 16  # every rank gets a matrix-core-wide shard
        if args.parallel == "channel":
            if not channel_ok:
                raise SystemExit(f"--parallel channel needs D >= 16 * {world}")
            mode = "channel"
        elif channel_ok:  # auto: two probe steps of each candidate, same decision on every rank (MAX over ranks)
            probe = {}
            cands = {"view": ("view", "rs_ag"), "view/allreduce": ("view", "allreduce"), "channel": ("channel", None)}
            base_fn, base_pc, _, _ = build("view", "rs_ag")  # ONE model: the collectives switch on it, the channel probe shares its geometry
            for name, (m, gr) in cands.items():
                if m == "view":
                    fn, pc_ = base_fn, base_pc
                    fn.set_collective(gr)
                else:
                    fn, pc_, _, _ = build(m, gr, share=base_pc)
                try:  # a collective flavour the backend refuses costs its candidate, not the run
                    fn()
                    torch.cuda.synchronize(); barrier()
                    t0 = time.perf_counter()
                    fn(); fn()
                    torch.cuda.synchronize(); barrier()
                    elapsed = time.perf_counter() - t0
                except RuntimeError as e:
                    print(f"[bench] rank {rank}: probe of '{name}' failed: {e}", file=sys.stderr, flush=True)
                    elapsed = float("inf")
                tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                probe[name] = 1e3 * float(tt.item()) / 2 if tt.item() != float("inf") else None
                del fn, pc_
            del base_fn, base_pc
            torch.cuda.empty_cache()
            best = min((k_ for k_ in probe if probe[k_] is not None), key=probe.get, default="channel")
            mode, chosen_reduce = cands[best]
            if chosen_reduce:
                args.grad_reduce = chosen_reduce
    step, pc, cam, d_local = build(mode)
    if world > 1 and mode == "view" and not args.no_overlap:
        # the overlapped exchange drives RCCL from a second stream: if this stack refuses it, say so and measure the plain
        # exchange instead of dying (same decision on every rank)
        ok = 1
        try:
            step()
            torch.cuda.synchronize()
        except RuntimeError as e:
            print(f"[bench] rank {rank}: overlapped gradient exchange failed ({e}); using the plain one", file=sys.stderr, flush=True)
            ok = 0
        flag = torch.tensor([ok], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        if int(flag.item()) == 0:
            args.no_overlap = True
            del step, pc
            torch.cuda.empty_cache()
            step, pc, cam, d_local = build(mode)

    for _ in range(args.warmup):
        pkg = step()
    torch.cuda.synchronize()
    info = pkg["info"]
    n_isects = info["n_isects"]
    n_visible = int((pkg["radii"] > 0).sum().item())

    # workload statistics for the roofline model (outside the timed region)
    counts = torch.zeros(2, dtype=torch.int64, device=dev)
    _lib.check(_lib.load().gags_raster_stats(width, height, _lib.ptr(info["means2d"][0]), _lib.ptr(info["conics"][0]),
                                             _lib.ptr(info["opacities"][0].contiguous()),
                                             _lib.ptr(info["isect_offsets"]), _lib.ptr(info["flatten_ids"]),
                                             n_isects, _lib.ptr(counts), None), "gags_raster_stats")
    torch.cuda.synchronize()
    q_eval, q_blend = (int(v) for v in counts.tolist())
    del pkg, info

    profiler.enable(True)
    barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        step()
    marks[args.steps].record()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    stages = profiler.summary()
    profiler.enable(False)
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))

    # R9 (outside the metric, which is fwd+bwd): the optimizer step on the gradient the last step left
    adam = None
    if rank == 0 and pc._semantic_feature.grad is not None:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        pc.optimizer.step()
        for i in range(5):
            ev[i].record()
            pc.optimizer.step()
        ev[5].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[5]) / 5
        nbytes = 28.0 * pc._semantic_feature.numel()  # p, g, m, v read; p, m, v written
        adam = {"kernel": "adam_step_kernel", "avg_launch_ms": ms, "bound": "hbm", "achieved": nbytes / ms / 1e6,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / ms / 1e6 / HBM_PEAK_GBS}

    exposed_ms = exposed[-1].exposed_ms() if exposed else None
    range_ms = exposed[-1].range_ms if exposed else None
    # "verify RCCL picks the direct/mesh algorithm": the same step with the OTHER collective, a few steps, so
    # that one N-GPU run yields the rs_ag / allreduce comparison (MAX over ranks; outside `value`)
    collective_ms = None
    if world > 1 and mode == "view" and not args.no_allreduce:
        collective_ms = {args.grad_reduce: 1e3 * dt / args.steps}
        other = "allreduce" if args.grad_reduce == "rs_ag" else "rs_ag"
        # (under the gloo debug transport rs_ag on device tensors is slow or refused: recorded as measured / null, same code path)
        ok = 1
        try:
            step.set_collective(other)
            step()
            torch.cuda.synchronize(); barrier()
            osteps = max(2, min(args.steps, 5))
            t1 = time.perf_counter()
            for _ in range(osteps):
                step()
            torch.cuda.synchronize(); barrier()
            odt = time.perf_counter() - t1
        except RuntimeError as e:
            print(f"[bench] rank {rank}: collective '{other}' failed: {e}", file=sys.stderr, flush=True)
            ok, odt, osteps = 0, 0.0, 1
        tt = torch.tensor([odt if ok else float("inf")], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        collective_ms[other] = 1e3 * float(tt.item()) / osteps if tt.item() != float("inf") else None
        step.set_collective(args.grad_reduce)
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    ms_per_step = 1e3 * dt / args.steps
    value = world * args.steps / dt  # every rank renders one view per step (weak scaling)

    if rank == 0:
        # Per-kernel roofline.  Each entry is ONE kernel bracketed by HIP events on the
        # launch stream; algorithmic work per launch:
        #   raster_fwd  : 14*Q_eval + 2*D*Q_blend flops              (weights + feature passes; fp32 MFMA bound)
        #   bwd_rows    : 2*D*Q_blend flops                           (weight tile, MFMA, row stores)
        #   bwd_reduce  : 2*B*D*4 bytes, B = Gaussians that blended anything: each one's gradient row formed from at least one
        #                 partial row and written once (HBM bound; the kernel actually reads one row per touched (tile,
        #                 Gaussian) -- 4.7 per blended Gaussian at C3 -- and, since round 6's persistent gradient buffer, no
        #                 longer writes the zero rows of the other N - B Gaussians)
        #   raster_bwd  : 14*Q_eval + 2*D*Q_blend flops              (single-kernel atomic backward, if used)
        rows = profiler.notes().get("bwd_rows", 0)
        g_last = pc._semantic_feature.grad
        n_blended = int((g_last != 0).any(dim=1).sum().item()) if g_last is not None else 0  # rows of the gradient that are not zero
        blk = profiler.notes().get("fwd_blk_rows")
        slots = int(blk.sum().item()) if blk is not None else 0  # (block, Gaussian) pairs that blended: one weight row each
        dl = d_local  # feature width of one launch on this rank (D, or the rank's channel shard)
        pix = width * height
        #   raster_weights : HBM bytes = 40 I (ids, records, hit flags) + 264 slots (weight row, ids) + 12 P (alpha, last id, T)
        #                    (reported against HBM; the kernel is VALU-bound -- 14 flops per evaluated pair -- and says so)
        #   raster_fwd_feat: 2*D*Q_blend flops  (priced against the fp32 matrix peak: the arithmetic it replaces; issued as six
        #                    v_mfma_f32_32x32x16_bf16 terms per product on operands split into three bf16 terms)
        work = {
            "raster_weights": ("hbm", 40.0 * n_isects + 264.0 * slots + 12.0 * pix),
            "raster_fwd_feat": ("mfma", 2.0 * dl * q_blend),
            "bwd_rows": ("mfma", 2.0 * dl * q_blend),
            "bwd_reduce": ("hbm", 8.0 * dl * n_blended),
            "raster_bwd": ("mfma", 14.0 * q_eval + 2.0 * dl * q_blend),
        }
        kernels = {}
        for name, (bound, amount) in work.items():
            if name in stages and stages[name][0] > 0:
                ms = stages[name][0]
                if bound == "mfma":
                    ach, peak, unit = amount / (ms * 1e-3) / 1e12, FP32_MATRIX_PEAK_TFLOPS, "TFLOP/s"
                else:
                    ach, peak, unit = amount / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
                kernels[name] = {"bound": bound, "avg_launch_ms": ms, "achieved": ach, "peak": peak, "unit": unit,
                                 "frac": ach / peak}
        # HBM traffic per launch from the committed rocprofv3 PMC passes of this very command
        # (tools/pmc_passes.sh -> profiles/*_pmc_traffic.json: 2*FETCH_SIZE + WRITE_SIZE, separate passes)
        traffic, traffic_src = pmc_traffic(args) if world == 1 else ({}, None)
        for name, members in STAGE_KERNELS.items():
            if name in kernels:
                tb = [_traffic_of(traffic, m) for m in members]
                kernels[name]["traffic"] = sum(tb) if all(t is not None for t in tb) else None
                # matrix work the kernel ISSUES per launch (SQ_INSTS_MFMA x flops per instruction, same PMC passes) next
                # to the algorithmic flops: the ratio is the padding (zero halves, ragged 32-row tiles, split terms)
                mi = _traffic_of(traffic, members[0], "mfma_insts")
                if name in MFMA_FLOP and mi:
                    ms = kernels[name]["avg_launch_ms"]
                    issued = mi * MFMA_FLOP[name]
                    pipe_peak = (FP32_MATRIX_PEAK_TFLOPS if (name == "bwd_rows" and (args.raster_flags & _lib.GAGS_BWD_F32MFMA))
                                 or (name == "raster_fwd_feat" and (args.raster_flags & _lib.GAGS_FWD_EXACT)) else F16_MATRIX_PEAK_TFLOPS)
                    kernels[name]["issued_flops"] = issued
                    kernels[name]["mfma_issue_frac"] = issued / (ms * 1e-3) / 1e12 / pipe_peak
                    kernels[name]["mfma_busy_cycles"] = _traffic_of(traffic, members[0], "mfma_busy_cycles")
                    kernels[name]["busy_cu_cycles"] = _traffic_of(traffic, members[0], "busy_cu_cycles")
        if "raster_weights" in kernels:
            kernels["raster_weights"]["note"] = ("VALU-bound (14 flops per evaluated pair: 14*Q_eval = %.1f GFLOP per launch); priced "
                                                 "against HBM by the bytes it has to move" % (14.0 * q_eval / 1e9))
        if "raster_fwd_feat" in kernels and not (args.raster_flags & _lib.GAGS_FWD_EXACT):
            kernels["raster_fwd_feat"]["note"] = ("algorithmic fp32 flops (2 D Q_blend) against the fp32 matrix peak; issued as 6 "
                                                  "v_mfma_f32_32x32x16_bf16 terms per product, both operands split into three bf16 terms "
                                                  "in registers (exact operands; fp32-equivalent, DESIGN.md 4)")
        if "bwd_rows" in kernels and not (args.raster_flags & _lib.GAGS_BWD_F32MFMA):
            kernels["bwd_rows"]["note"] = ("algorithmic fp32 flops (2 D Q_blend) against the fp32 matrix peak; the kernel issues them as "
                                           "3 v_mfma_f32_32x32x16_f16 terms per product on two-term split operands (one fp32-level rounding per operand: as close "
                                           "to float64 as fp32 matrix arithmetic, DESIGN.md 4); 32-row MFMA tiles "
                                           "are chunks of 32 consecutive tile rows x one 8x8 block (58 % of their rows are non-zero at C3)")
        dom = max(kernels, key=lambda k_: kernels[k_]["avg_launch_ms"])
        prefix = STAGE_KERNELS.get(dom, (dom,))[0]
        names = [k_ for k_ in traffic if k_.startswith(prefix)]  # the kernel's full short name in the committed profile
        roof = dict(kernels[dom], kernel=dom, rocprof_name=names[0] if len(names) == 1 else prefix, traffic_source=traffic_src)
        line = {
            "metric": "feature-raster fwd+bwd views/s",
            "value": value, "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 tensors and fp32(-equivalent) arithmetic: forward contraction on v_mfma_f32_32x32x16_bf16 with both operands "
                          "split into three bf16 terms (exact operands, six terms per product: as close to float64 as the fp32 chain; "
                          "GAGS_FWD_EXACT = v_mfma_f32_32x32x2_f32, bit-identical to the oracle); backward contraction on "
                          "v_mfma_f32_32x32x16_f16 with both operands as two fp16 terms after exact power-of-two scalings (the cotangent's per "
                          "(tile, channel), the weights' one constant 2^15: a weight lies in [2^-21.3, 1)) and three product terms (<= 3 * 2^-24 "
                          "|w v| + 2^-39 |v| per product; 1.7e-7 of float64 where fp32 matrix arithmetic gives 1.90e-7; exact three-term "
                          "weights with five product terms: GAGS_BWD_EXACT_WEIGHTS); fp32 accumulation in both",
            "config": {"workload": f"{args.config}: {n} Gaussians, {width}x{height}, D={d}, {world} view(s)/step"
                                   + (f", every GPU renders all {world} views for its {d_local} channels"
                                      if mode == "channel" else ", 1 view/GPU/step"),
                       "n_gaussians": n, "width": width, "height": height, "feature_dim": d,
                       "splat_scale0": scale0, "splat_scale_note": ("SURVEY 8d literal 0.004 z_mean" if scale0 == syn.SCALE0_SURVEY
                                                                     else "0.0009 z_mean: meets SURVEY 8d's stated median radius / I~5N; "
                                                                          "the literal 0.004 constant is reported as heavy_workload"),
                       "isects_per_visible": n_isects / max(n_visible, 1),
                       "visible": n_visible, "n_isects": n_isects, "pairs_evaluated": q_eval,
                       "pairs_blended": q_blend, "bwd_rows": rows, "gaussians_blended": n_blended,
                       "parallelism": (f"channel-shard{world} (no data-path collective)" if mode == "channel"
                                       else f"view-dp{world}" + ((" + RCCL all-reduce of the feature gradient" if args.grad_reduce == "allreduce"
                                                                  else " + RCCL reduce-scatter/all-gather of the feature gradient")
                                                                 if world > 1 else "")),
                       "parallel_probe_ms_per_step": probe,
                       "grad_exchange": (None if world == 1 or mode != "view" else
                                         {"overlapped_with_backward": not args.no_overlap, "wire": args.wire,
                                          "collective": args.grad_reduce, "exposed_ms_last_step": exposed_ms,
                                          "range_exchange_ms_last_step": range_ms,
                                          "rows": args.rows if not args.no_overlap else "all",
                                          "rows_exchanged_last_step": (exposed[-1].rows_exchanged if exposed else None),
                                          "collective_ms_per_step": collective_ms,
                                          "rccl": rccl_info()})},
            "roofline": roof,
            "kernels": kernels,
            "activation_getters": "fused: exp / normalize / sigmoid of scene/gaussian_model.py:116-139 run inside the projection kernel "
                                  "(gags_project_fwd_raw, bit-identical to torch's) on every render, inside the timed region; no getter "
                                  "kernels are launched, so the activation cache of gags_amd/scene.py has nothing left to save here",
            "stages_ms": {k: round(v[0], 4) for k, v in sorted(stages.items())},
            "step_ms": {"median": per_step[len(per_step) // 2], "p10": per_step[int(0.1 * (len(per_step) - 1))],
                        "p90": per_step[int(round(0.9 * (len(per_step) - 1)))]},
            "optimizer_step": adam,
        }
        # whole-step view of the metric's "HBM-roofline %": compulsory traffic of one view, every tensor once
        # (SURVEY.md 8d):  72 N + 36 I + 2 V D s_f + 2 P (D s_o + 8) + 4 V D   bytes, s_f = s_o = 4
        b_view = 72.0 * n + 36.0 * n_isects + 2.0 * n_visible * dl * 4 + 2.0 * pix * (dl * 4 + 8) + 4.0 * n_visible * dl
        views_per_gpu_step = world if mode == "channel" else 1
        gbs = b_view * views_per_gpu_step / (ms_per_step * 1e-3) / 1e9
        line["hbm_roofline_step"] = {"algorithmic_bytes_per_view": b_view, "achieved": gbs, "peak": HBM_PEAK_GBS,
                                     "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pc, cam, d, width, height, args.cpu_seconds)
        if world == 1 and d % 128 == 0 and not (args.no_heavy or args.raster_flags):
            # the same workload with the backward's contraction on the fp32 matrix instructions (rounds 1-2's default,
            # gags_amd._lib.GAGS_BWD_F32MFMA), reported next to `value` for comparison; DESIGN.md section 4
            args.raster_flags = _lib.GAGS_BWD_F32MFMA
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            fsteps = max(3, min(args.steps, 10))
            t0 = time.perf_counter()
            for _ in range(fsteps):
                step()
            torch.cuda.synchronize()
            fdt = time.perf_counter() - t0
            args.raster_flags = 0
            line["backward_f32mfma"] = {
                "note": "same workload, forward unchanged; the backward contracts with v_mfma_f32_32x32x2_f32 (rounds 1-2's kernel) "
                        "instead of the default: v_mfma_f32_32x32x16_f16 on split operands (weights and cotangent as two fp16 terms each: "
                        "one fp32-level rounding per operand; three MFMA terms per product; vs float64 1.68e-7 against this kernel's 1.90e-7)",
                "value": fsteps / fdt, "unit": "views/s", "ms_per_step": 1e3 * fdt / fsteps, "steps": fsteps}
            # ... and with the weights of the default rows kernel kept exact (three fp16 terms, five product terms: rounds
            # 3-4's arithmetic in round 5's kernel shape, gags_amd._lib.GAGS_BWD_EXACT_WEIGHTS)
            args.raster_flags = _lib.GAGS_BWD_EXACT_WEIGHTS
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(fsteps):
                step()
            torch.cuda.synchronize()
            fdt = time.perf_counter() - t0
            args.raster_flags = 0
            line["backward_exact_weights"] = {
                "note": "same workload, same kernels, the backward's weights as three fp16 terms (exact) and five MFMA terms per product "
                        "instead of two terms / three products: 1.60e-7 of float64 instead of 1.68e-7 (fp32 matrix arithmetic: 1.90e-7)",
                "value": fsteps / fdt, "unit": "views/s", "ms_per_step": 1e3 * fdt / fsteps, "steps": fsteps}
            # ... and the strict-fp32 step as ONE number: forward on the fp32 matrix instructions (the oracle's fmaf chain, bit for
            # bit) AND backward on the fp32 matrix instructions -- no split operands anywhere
            args.raster_flags = _lib.GAGS_FWD_EXACT | _lib.GAGS_BWD_F32MFMA
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(fsteps):
                step()
            torch.cuda.synchronize()
            fdt = time.perf_counter() - t0
            args.raster_flags = 0
            line["all_exact"] = {
                "note": "same workload with GAGS_FWD_EXACT | GAGS_BWD_F32MFMA: both contractions on v_mfma_f32_32x32x2_f32 (forward "
                        "bit-identical to the oracle's sequential chain, backward plain fp32 matrix arithmetic): the strict-fp32 step",
                "value": fsteps / fdt, "unit": "views/s", "ms_per_step": 1e3 * fdt / fsteps, "steps": fsteps}
        if world == 1 and not (args.no_heavy or args.raster_flags):
            # north_star's "feature / geometry gradients": the same workload with EVERY parameter requiring grad (joint
            # training; the reference's GAD stage freezes the geometry).  Reported next to `value`, never as `value`.
            geo = [pc._xyz, pc._scaling, pc._rotation, pc._opacity]
            for q in geo:
                q.requires_grad_(True)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            gsteps = max(3, min(args.steps, 10))
            t0 = time.perf_counter()
            for _ in range(gsteps):
                for q in geo:
                    q.grad = None
                step()
            torch.cuda.synchronize()
            gdt = time.perf_counter() - t0
            for q in geo:
                q.requires_grad_(False)
                q.grad = None
            line["all_gradients"] = {
                "note": "same workload, gradients of features AND means, quats, scales, opacities (SURVEY A9 + K2): geometry "
                        "dot products on the 16-bit matrix cores with fp32-equivalent split operands (gags_raster_bwd_geom), no atomics",
                "value": gsteps / gdt, "unit": "views/s", "ms_per_step": 1e3 * gdt / gsteps, "steps": gsteps}
        if world == 1 and d % 128 == 0 and d > 128 and not (args.no_heavy or args.raster_flags):
            # The N > 1 code path priced on ONE GPU (no multi-GPU box is reachable from the build): the by-view step exactly as
            # rank 0 of 8 runs it, with every collective replaced by two device copies of the block it would exchange
            # (OverlappedGradReducer(loopback=...)): range-staged backward, gags_blended_mask, gags_compact_mask, pack and unpack
            # of C4's union block (446 525 rows = 29.8 % of N at C3: tools/union_rows.py), on the exchange stream under the
            # backward.  What it does NOT contain is the wire time of RCCL over xGMI (DESIGN.md section 6 prices that).
            union_rows = int(round(0.2977 * n))
            red = OverlappedGradReducer(mode="rs_ag", rows="union", param=pc._semantic_feature, loopback=(8, union_rows))

            def dp_step():
                pc._semantic_feature.grad = None
                pkg_ = render(cam, pc, None, bg, feature_mode=True)
                loss = _CotangentLoss.apply(pkg_["render"].permute(1, 2, 0), G_dp.permute(1, 2, 0))
                with red:
                    loss.backward()
                red.finish(pc._semantic_feature.grad)

            G_dp = syn.make_cotangent(d, height, width, seed=1, device=dev)
            for _ in range(2):
                dp_step()
            torch.cuda.synchronize()
            vsteps = max(3, min(args.steps, 10))
            t0 = time.perf_counter()
            for _ in range(vsteps):
                dp_step()
            torch.cuda.synchronize()
            vdt = 1e3 * (time.perf_counter() - t0) / vsteps
            t0 = time.perf_counter()
            for _ in range(vsteps):
                step()
            torch.cuda.synchronize()
            pdt = 1e3 * (time.perf_counter() - t0) / vsteps
            # the same with the rows kernel launched per 256 channels (fewer re-reads of the weight tiles: cheaper on one GPU,
            # but the first range -- and with it the first exchange -- is ready 0.6 ms later: DESIGN.md section 6)
            from gags_amd.rasterization import default_context
            g0 = default_context().grad_rows_group
            default_context().grad_rows_group = 256
            for _ in range(2):
                dp_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(vsteps):
                dp_step()
            torch.cuda.synchronize()
            vdt256 = 1e3 * (time.perf_counter() - t0) / vsteps
            exposed256 = red.exposed_ms()
            default_context().grad_rows_group = g0
            dp_step()
            torch.cuda.synchronize()
            line["view_dp_overhead_ms"] = {
                "note": "by-view step as rank 0 of 8 runs it, loop-back exchange (two device copies per 128-channel block instead of "
                        "the collective, out of place): range-staged backward (rows, reduce + exchange per 128 channels) + "
                        "gags_blended_mask + gags_compact_mask_pos + the reduce stage writing the union block itself + unpack, "
                        "overlapped with the backward on a second stream; xGMI wire time not included",
                "plain_step_ms": pdt, "view_dp_step_ms": vdt, "overhead_ms": vdt - pdt, "union_rows": red.rows_exchanged,
                "rows_per_256_channels": {"view_dp_step_ms": vdt256, "overhead_ms": vdt256 - pdt, "exposed_ms_last_step": exposed256},
                "exposed_ms_last_step": red.exposed_ms(), "range_exchange_ms_last_step": red.range_ms, "steps": vsteps}
            del red, G_dp
        if world == 1 and args.config == "C3" and not (args.no_heavy or args.n or args.d):
            # second reading of SURVEY 8d (gags_amd/synthetic.py): same N / resolution / D, ~4.4x larger splats
            del step, pc
            torch.cuda.empty_cache()
            hstep, hpc, _, _ = build("view", scale0=syn.SCALE0_SURVEY)
            for _ in range(2):
                hp = hstep()
            torch.cuda.synchronize()
            hi = hp["info"]["n_isects"]
            hv = int((hp["radii"] > 0).sum().item())
            del hp
            hsteps = max(3, min(args.steps, 10))
            t0 = time.perf_counter()
            for _ in range(hsteps):
                hstep()
            torch.cuda.synchronize()
            hdt = time.perf_counter() - t0
            line["heavy_workload"] = {
                "workload": f"C3H: {n} Gaussians, {width}x{height}, D={d}, splat scale 0.004 z_mean (SURVEY 8d literal)",
                "value": hsteps / hdt, "unit": "views/s", "ms_per_step": 1e3 * hdt / hsteps, "steps": hsteps,
                "n_isects": hi, "visible": hv, "isects_per_visible": hi / max(hv, 1)}
        if world == 1 and args.config == "C3" and not (args.no_heavy or args.n or args.d or args.raster_flags):
            # BASELINE.json configs[2] says "full GAD.sh distillation loop": one train.py:142-174 iteration at the width the
            # reference really rasterizes (D = 16, train.py:68) -- render -> CNN_scale_decoder / CNN_decoder -> losses ->
            # backward through decoders and rasterizer -- through gags_amd.distill.distillation_loss, the composition
            # tests/test_iteration_gpu.py pins against the reference's own functions.  Not `value` (the metric is the
            # feature rasterizer); on the driver's clock so that it is not a builder-only number.
            torch.cuda.empty_cache()
            line["iteration_d16"] = iteration_d16(dev, n, width, height, args.steps)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
