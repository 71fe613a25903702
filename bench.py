#!/usr/bin/env python3
"""bench.py — denoising-steps/sec of the SynTalker hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--mode sample|train|guided]

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks (one process per GPU,
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`); under an external launcher
(the driver's torchrun line) it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment as before.

One "step" = one DDPM denoising step (denoiser evaluation + posterior update) over a batch of B synthetic
128-frame clips per GPU (latent (B,1536,1,32)), issued the way SpacedDiffusion.p_sample_loop issues it:
device-resident timestep schedule, hipGraph replays of 10 captured steps (single-step graph for the
remainder); metric = clip-steps/s over ALL GPUs (BASELINE.json: "denoising-steps/sec (128-frame clips)").  Workload = configs[1]
(diffusion_rvqvae_128 sampling, p_sample_loop, bf16 operands / fp32 accumulate): random-init weights
of the reference architecture, synthetic audio/word/seed conditioning, computed ONCE per clip before
the timed region exactly as the fused p_sample_loop does (its cost is reported separately).
Clips shard across GPUs with no collective (weak scaling: B per GPU is fixed).

--mode train times BASELINE.json configs[2] instead (training step: 32 clips per GPU, uniform timestep sampler,
training_losses forward + backward, clip 0.99, Adam 5e-5 / (0.5, 0.999), DDP all-reduce of the 29.6 M gradients over
RCCL; reference seam train.py:87-94): metric = training samples/s over all GPUs + a forward / backward / optimiser split.

--mode guided times BASELINE.json configs[3] and configs[4]: classifier-free guidance as ONE fused batch (cfg_sampler.py:10-167) over
the h3d-style denoiser at the _hf.yaml shapes - `ClassifierFreeSampleModel` (scale 2.5, 2 reference evaluations per step, V = 2 variants,
DDPM-1000 steps on the headline line and DDIM-50 steps beside it) and `TwoClassifierFreeSampleModel_Bodypart` (upper + lower prompts:
9 reference evaluations, 4 unique variants, DDIM-50; h3d_diffusion_new_trainer.py:560-572): metric = guided clip-steps/s, plus
reference-evaluation-equivalents/s and the roofline of the step kernel (algorithmic FLOPs = the V UNIQUE evaluations).

The timed region is exactly what the contract says: W untimed steps, then K timed ones (`--prime N` adds N untimed 10-step replays in
front of the warm-up, default 0; `steady_state` on the JSON line is a separate, later measurement of 200 more steps - a 1000-step
loop spends 95 % of its time at clocks the device only reaches ~50 steps after a graph capture).

Extra objects on the JSON line:
  roofline      dominant kernel: algorithmic FLOPs per launch / average launch duration (hipEvents on the replay
                stream around every graph replay of the timed region, divided by the steps it holds; the step is one
                kernel) vs the dense bf16 MFMA peak; one eager launch afterwards only names the kernel.
  cpu_baseline  the as-written CPU restatement of the reference forward (oracle/) timed on the host cores.
  train_step / train_step_ddp_1rank / guided   (default sample-mode line, one GPU; --no-extras skips them) BASELINE configs[2] - [4] through the same
                code paths --mode train | guided time: the captured training step (20 timed steps), the same step under the DDP wrapper over a 1-rank
                RCCL group (child process, 200 steps), CFG V = 2 with DDPM / DDIM-50 steps and the body-part wrapper V = 4 (20 timed steps each).
  small_batch   step time at B = 1 .. 256 (the reference's own sampling scripts run B = 1).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

F_STEP = 1_192_755_200            # algorithmic FLOPs per clip-step
PEAK_BF16 = 2.5e15                # dense bf16 MFMA peak, MI355X_MICROARCH.md
STAGES = ["in_gemm", "qkv_gemm", "attention", "proj_gemm", "fc1_gemm", "fc2_gemm", "cfg_combine", "out_gemm"]
# kernel symbol + algorithmic FLOPs per clip per launch for each stage class
_ATT = 2 * 2 * 4 * 32 * 32 * 128
STAGE_INFO_LEGACY = {
    "in_gemm": ("k_gemm<MT,EPI_IN>", 2 * 32 * 1536 * 512),
    "qkv_gemm": ("k_gemm<MT,EPI_QKV>", 2 * 32 * 512 * 1536),
    "attention": ("k_attn", _ATT),
    "proj_gemm": ("k_gemm<MT,EPI_RESID>", 2 * 32 * 512 * 512),
    "fc1_gemm": ("k_gemm<MT,EPI_GELU>", 2 * 32 * 512 * 1024),
    "fc2_gemm": ("k_gemm<MT,EPI_RESID>", 2 * 32 * 1024 * 512),
    "out_gemm": ("k_gemm<MT,EPI_OUT>", 2 * 32 * 512 * 1536),
}
_BLOCK = 2 * 32 * 512 * 1536 + _ATT + 2 * 32 * 512 * 512 + 2 * 2 * 32 * 512 * 1024
# production: input GEMM + 8 blocks + output GEMM + posterior update are ONE kernel
STAGE_INFO_STACK = {
    "fc2_gemm": ("k_stack<MT>", 2 * 32 * 1536 * 512 + 8 * _BLOCK + 2 * 32 * 512 * 1536),
}

def cpu_baseline(budget_s: float):
    """Reference forward as written (no hoisting / folding), fp32, torch CPU, B=1 - SURVEY.md §8d protocol.
    torch's intra-op threading does not scale to every core on a 32-token problem, so a probe (>= 10 forwards per setting) ranks
    {all cores, 32, 16, 8} threads; the reported value is the MEDIAN of three equal windows at the probe's winner, with their spread.
    A box whose sustained rate falls more than 25 % short of its own probe (round 4's driver box: probe 77, sustained 39.5 - the thread pool
    was re-sized four times in front of the sample) also times the runner-up and keeps the better one, and the line says so."""
    from oracle import denoiser_ref as dr
    from syntalker_amd import synth
    from syntalker_amd.denoiser import MDM
    # the oracle runs on the reference's state_dict: the product module carries the same entries, filled by the same name-keyed initialiser
    sd = {k: v.detach().clone() for k, v in synth.synth_fill_(MDM(synth.default_args()).eval(), 0).state_dict().items()}
    y, x = synth.synth_clip_inputs(1, seed=1), synth.synth_latent(1, seed=1)
    all_cores = torch.get_num_threads()

    def window(seconds, min_n, fn):
        n, t0 = 0, time.perf_counter()
        while True:
            fn(n)
            n += 1
            dt = time.perf_counter() - t0
            if dt > seconds and n >= min_n:
                return n, dt

    as_written = lambda i: dr.mdm_forward(sd, x, torch.tensor([996 - i % 900]), y)

    def sustained(nt):
        """three windows at nt threads -> (median rate, spread = (max - min) / median, forwards, seconds)"""
        torch.set_num_threads(nt)
        for i in range(3):
            as_written(i)                                   # the pool at its new size, warm
        wins = [window(budget_s / 3, 7, as_written) for _ in range(3)]
        rates = sorted(n / dt for n, dt in wins)
        return rates[1], (rates[2] - rates[0]) / rates[1], sum(n for n, _ in wins), sum(dt for _, dt in wins)

    probe, note = {}, ""
    with torch.no_grad():
        for nt in sorted({all_cores, 32, 16, 8}, reverse=True):
            if nt > all_cores:
                continue
            torch.set_num_threads(nt)
            as_written(0); as_written(1)
            n, dt = window(0.25, 10, as_written)
            probe[nt] = n / dt
        ranked = sorted(probe, key=probe.get, reverse=True)
        best = ranked[0]
        value, spread, n, dt = sustained(best)
        if value < 0.75 * probe[best]:
            note = f"; sustained rate at {best} threads ({value:.1f}) was more than 25 % below its probe ({probe[best]:.1f})"
            if len(ranked) > 1:
                v2, s2, n2, dt2 = sustained(ranked[1])
                note += f", runner-up {ranked[1]} threads sustained {v2:.1f}"
                if v2 > value:
                    best, value, spread, n, dt = ranked[1], v2, s2, n2, dt2
            note += f": reported = the better one ({best} threads)"
        torch.set_num_threads(best)
        # the same CPU path with the build's algebra (conditioning hoisted, input stage folded), so that the GPU
        # speed-up can be split into its algorithmic and its hardware part
        fw = dr.fold_weights(sd)
        cond, te = dr.clip_conditioning(sd, y, fw), dr.time_table(sd, fw)
        nh, dth = window(budget_s / 4, 20, lambda i: dr.mdm_forward_folded(sd, fw, cond, te, x, torch.tensor([996 - i % 900])))
        # SURVEY 8d also asks for B = 40 (the reference's own test batch): a bounded sample of whole-batch forwards
        y40, x40 = synth.synth_clip_inputs(40, seed=2), synth.synth_latent(40, seed=2)
        dr.mdm_forward(sd, x40, torch.full((40,), 999), y40)
        n40, dt40 = window(budget_s / 3, 2, lambda i: dr.mdm_forward(sd, x40, torch.full((40,), 996 - i), y40))
    torch.set_num_threads(all_cores)
    return {"value": round(value, 2), "unit": "clip-steps/s", "cores": best, "kind": "port", "spread": round(spread, 3),
            "probe": {str(k): round(v, 1) for k, v in probe.items()}, "sustained_over_probe": round(value / probe[best], 3),
            "hoisted_value": round(nh / dth, 2), "value_b40": round(40 * n40 / dt40, 2),
            "sample": f"median of three {budget_s / 3:.1f}-s windows ({n} as-written MDM forwards at B=1 in all, {dt:.1f} s; spread (max - min) / median = "
                      f"{spread:.1%}); fp32, torch {torch.__version__} CPU, {best} of {all_cores} threads = the winner of a probe of >= 10 forwards per "
                      f"setting {({k: round(v, 1) for k, v in probe.items()})}{note}; conditioning recomputed every step like the reference; "
                      f"hoisted_value = the same with the conditioning computed once and the input stage folded ({nh} forwards, {dth:.1f} s); "
                      f"value_b40 = as-written forwards at B=40 ({n40} forwards, {dt40:.1f} s, same thread count)"}


def practical_peak(dev, seconds: float = 0.25):
    """The matrix pipe alone on THIS box: `syn_test_mfma_rate` (one wave per SIMD like k_seq, 12 independent v_mfma_f32_32x32x16_bf16
    accumulators, register operands, nothing else in the stream), hipEvent-timed over a few launches of ~40 ms after a warm-up long enough for
    the clock to settle under the load.  2.5 PFLOP/s assumes 2.4 GHz; what the chip holds on this instruction under its power cap depends on the box
    and on the operands' switching activity (pseudo-random operand images here).  Reported BESIDE the nominal peak (roofline.peak), never instead."""
    import ctypes
    from syntalker_amd import _lib
    lib = _lib.load()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    out = torch.empty(cus * 256, device=dev, dtype=torch.float32)
    flops = ctypes.c_int64(0)
    st = _lib.current_stream(dev)
    iters = 20000                                   # 20000 x 192 MFMAs x ~34 cycles ~ 65 ms at 2 GHz
    for _ in range(3):
        _lib.check(lib.syn_test_mfma_rate(iters, out.data_ptr(), ctypes.byref(flops), st), "syn_test_mfma_rate")
    torch.cuda.synchronize()
    reps = max(2, int(seconds / 0.065))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(lib.syn_test_mfma_rate(iters, out.data_ptr(), ctypes.byref(flops), st), "syn_test_mfma_rate")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    rate = flops.value / (ms * 1e-3)
    mfmas_per_wave = iters * 192
    return {"tflops": round(rate / 1e12, 1), "frac_of_nominal_peak": round(rate / PEAK_BF16, 4), "ms_per_launch": round(ms, 3),
            "effective_ghz_at_32_cycles_per_mfma": round(mfmas_per_wave * 32 / (ms * 1e-3) / 1e9, 3),
            "kernel": "syn_test_mfma_rate: bare v_mfma_f32_32x32x16_bf16 loop, one wave per SIMD, 12 accumulators, register operands",
            "note": "the matrix pipe's own rate on this box at the clock it holds under that load (power cap); k_seq's instruction stream adds LDS fragment reads, "
                    "tape DMA, a chunk barrier per 16 MFMAs, LayerNorm / softmax / GELU and the posterior update on top of this loop"}


def small_batch_probe(pm, coef, dev, sizes=(1, 8, 16, 32, 64, 256), reps=300):
    """Step time at the batch sizes of the reference's own sampling scripts (test.py / demo.py denoise one window at
    a time): the library picks the persistent feature-split kernel there.  Same graph-replayed step as the headline
    number (posterior update + in-epilogue Philox noise), random conditioning, hipEvent-timed on the replay stream;
    the whole-step kernel is timed beside it at B = 1."""
    from syntalker_amd import engine

    def timed(B, mode):
        sb = engine.StepBuffers(B, 1, dev, layer_mode=mode)
        sb.cond.normal_()
        sb.load_x(torch.randn(B, 1536, 1, 32, device=dev))
        sb.set_rng(7, 0)
        sb.t_model.fill_(500); sb.t_coef.fill_(500)
        g = engine.StepGraph(pm, sb, coef, True, fused_rng=True)
        n = reps if B <= 32 else max(30, reps // 5)
        for _ in range(10):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        sb.check_sync()
        return e0.elapsed_time(e1) * 1e3 / n

    us = {B: timed(B, 0) for B in sizes}
    return {"kernel": "library's choice: k_lat up to 8 clips (output features split over the 32 CUs of an XCD, L2-local group barriers), k_stack with every "
                      "32-row tile split over 4 / 2 CUs of an XCD at 9-64 / 65-128 sequences, one 32-row tile per CU at 256 (SURVEY 8d config 2: B in {1, 16, 64, 256})",
            "us_per_step": {str(B): round(t, 1) for B, t in us.items()},
            "clip_steps_per_s": {str(B): round(B / t * 1e6, 0) for B, t in us.items()},
            "frac_of_bf16_mfma_peak": {str(B): round(B / t * 1e6 * F_STEP / PEAK_BF16, 4) for B, t in us.items()},
            "whole_step_kernel_us_per_step_B1": round(timed(1, 4), 1)}

def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one process per GPU) through
    torch.distributed.run on a free local port, stream rank 0's JSON line through, return the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on these hosts (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def timed_region(step, K, W, world, dist, sync):
    """The contract's timing: W untimed steps, then exactly K steps between barrier + device synchronisation on both
    sides; returns the MAX over ranks of the elapsed seconds."""
    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()
    for _ in range(W):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            tt = tt.cuda()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def dry_run(args, rank, world, dist):
    """Launch-path check on a GPU-less machine: the ranks rendezvous over gloo, run the barrier-bracketed timing loop
    around a trivial CPU step and rank 0 reports.  Nothing here is a measurement of the hot path.
    train mode additionally builds what the real run builds before its first step - the product MDM (29.6 M parameters) wrapped by
    `training.make_ddp(capturable=...)` over the process group - and reports the wrapper's configuration; guided mode builds the
    h3d-style model and plans the two guidance wrappers (variants and weights) exactly as `run_guided` does."""
    acc = torch.zeros(1)
    def step():
        acc.add_(1.0)
    K, W = args.steps, args.warmup
    dt = timed_region(step, K, W, world, dist, lambda: None)
    assert float(acc) == K + W
    B = args.batch or {"train": 32, "guided": 512}.get(args.mode, 1024)
    extra = {}
    if args.mode == "train":
        from syntalker_amd import synth, training
        from syntalker_amd.denoiser import MDM
        graph = args.train_graph if args.train_graph is not None else (args.backend != "gloo" or world == 1 and not args.force_ddp)
        model = synth.synth_fill_(MDM(synth.default_args()).train(), 0)
        n_params = sum(p.numel() for p in model.parameters())
        if world > 1 or args.force_ddp:
            net = training.make_ddp(model, None, capturable=graph)
            frozen = sorted(n for n, p in net.module.named_parameters() if not p.requires_grad)
            extra["ddp"] = {"find_unused_parameters": bool(net.find_unused_parameters), "frozen": frozen, "bucket_cap_mb": training.DDP_BUCKET_MB,
                            "broadcast_buffers": bool(net.broadcast_buffers)}
        extra.update(parameters=n_params, graph_replayed=bool(graph))
    if args.mode == "guided":
        from syntalker_amd import guidance, synth
        from syntalker_amd.denoiser_h3d import MDM
        model = synth.synth_fill_(MDM(synth.default_args()).eval(), 0)
        y = synth.synth_clip_inputs(2, seed=7, style_dim=256, style_zero=False)
        y["scale"] = torch.ones(1) * 2.5
        _, plan_fn = guidance.resolve(guidance.ClassifierFreeSampleModel(model))
        p3 = plan_fn(dict(y))
        parts = {"upper_mask": torch.randn(1, 256), "hands_mask": None, "lower_mask": torch.randn(1, 256)}
        _, plan_fn = guidance.resolve(guidance.TwoClassifierFreeSampleModel_Bodypart(model))
        p4 = plan_fn(dict(y, style_feature=parts))
        extra["plans"] = {"cfg": {"variants": len(p3.variants), "weights": p3.weights}, "bodypart_twocfg": {"variants": len(p4.variants), "weights": p4.weights}}
    return dict({"metric": "dry run (launch path only, no GPU work)", "value": None, "unit": None, "n_gpus": world, "steps": K, "warmup": W,
                 "ms_per_step": round(dt / K * 1e3, 4), "dry_run": True, "mode": args.mode, "clips_per_gpu": B}, **extra) if rank == 0 else None


def capture_or_fallback(build, world, dist, sync, dev, force=False):
    """The capture of a step with RCCL collectives inside has run on hardware with ONE rank only (this pool has no multi-GPU node): if `build()`
    raises on some rank of a larger job, EVERY rank drops to the eager step (same arithmetic, host-bound) and the line says so (`graph_fallback`)
    instead of the whole bench aborting; the ranks agree through a MIN all-reduce of their capture status.  With one rank the exception propagates
    (unless `force`, the test hook).  -> (captured step | None, reason | None)"""
    g, reason = None, None
    try:
        g = build()
    except Exception as e:                                       # noqa: BLE001 - anything the capture throws is reported, not swallowed
        if world == 1 and not force:
            raise
        reason = f"{type(e).__name__}: {e}"[:300]
    if world > 1:
        sync()
        ok = torch.tensor([0 if reason else 1], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok) == 0:
            if g is not None:
                g.close()
            g, reason = None, reason or "the capture failed on another rank"
    return g, reason


F_TRAIN = 3 * 5.90e9               # algorithmic FLOPs per training sample: fwd + bwd through the as-written model


def run_train(args, rank, local, world, dev, dist):
    """BASELINE configs[2]: one optimiser step over 32 clips per GPU (global batch 32 x N), data-parallel.  The step is the body
    of the reference's loop (diffusion_rvqvae_trainer.py:339-356, 555-560): sample t, training_losses forward, backward (DDP: bucketed
    RCCL all-reduce of the gradients overlapped with it), clip_grad_norm 0.99, Adam.  Nothing is skipped inside the timed region."""
    from syntalker_amd import synth, training
    from syntalker_amd.denoiser import MDM
    from syntalker_amd.process import create_gaussian_diffusion
    from syntalker_amd.resample import create_named_schedule_sampler

    B, K, W = args.batch or 32, args.steps, args.warmup
    diff = create_gaussian_diffusion()
    sampler = create_named_schedule_sampler("uniform", diff)
    y = synth.to_device(synth.synth_clip_inputs(B, seed=1 + rank, mask_batch=B), dev)
    y["audio"] = torch.randn(B, 68266, 2, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + rank))   # training clip length
    x0 = synth.synth_latent(B, seed=1 + rank, name="x0").to(dev)
    model = synth.synth_fill_(MDM(synth.default_args()).train(), 0).to(dev)
    # the captured step is the default; gloo collectives cannot be captured in a hipGraph, so a gloo run (functional checks) steps eagerly
    graph = args.train_graph if args.train_graph is not None else (args.backend != "gloo" or world == 1 and not args.force_ddp)
    ddp = world > 1 or args.force_ddp          # --force-ddp: the wrapper (and RCCL, one rank) on a 1-GPU box - what the N-GPU run will use
    net = model
    side = torch.cuda.Stream(device=dev) if graph else torch.cuda.current_stream(dev)
    if ddp:
        with torch.cuda.stream(side):                           # DDP is built on the stream its iterations run on
            net = training.make_ddp(model, local, capturable=graph)
        torch.cuda.current_stream(dev).wait_stream(side)
    # clip_grad_norm_(0.99) + Adam as the reference trainer runs them, on training.ClipAdam's three kernels (--torch-adam: PyTorch's foreach norm /
    # multiply + fused multi-tensor Adam, the same update)
    if args.torch_adam:
        opt = torch.optim.Adam(net.parameters(), lr=5e-5, betas=(0.5, 0.999), capturable=graph, fused=True)
    else:
        opt = training.ClipAdam(net.parameters(), lr=5e-5, betas=(0.5, 0.999), max_norm=0.99)
    fallback, g = None, None
    if graph:
        def build():
            if args.inject_capture_failure:
                raise RuntimeError("injected by --inject-capture-failure")
            return training.GraphedTrainStep(net, diff, opt, x0, {"y": y}, warmup=11 if ddp else 3, stream=side)
        g, fallback = capture_or_fallback(build, world, dist, lambda: torch.cuda.synchronize(dev), dev, force=args.inject_capture_failure)
        graph = g is not None
    if graph:
        last = {}
        def step():
            last["loss"] = g(x0, sampler.sample(B, x0.device)[0], {"y": y})
    else:
        last = {}
        def step():
            last["loss"] = training.train_step(net, diff, sampler, opt, x0, {"y": y})
    dt = timed_region(step, K, W, world, dist, torch.cuda.synchronize)
    loss = float(last["loss"])
    assert loss == loss, "non-finite training loss"
    direct = training.direct_grad_report(model) if ddp else None   # gradients the backward wrote straight into the reducer's buckets
    # forward / backward / clip+Adam split of one EAGER step (hipEvents on the current stream; rank 0 only reports it).  After a captured
    # run the same three eager steps follow the timed region: issued from Python the step is host-bound, so these are upper bounds of the
    # device time of each part (the rocprofv3 kernel-time split of the replayed step is profiles/r04_train_step_split.txt)
    split = None
    if graph:
        g.close()
    if True:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        acc = [0.0, 0.0, 0.0]
        for _ in range(3):
            t, _w = sampler.sample(B, x0.device)
            opt.zero_grad(set_to_none=True)
            ev[0].record()
            l = diff.training_losses(net, x0, t, model_kwargs={"y": y})["loss"].mean()
            ev[1].record()
            l.backward()
            ev[2].record()
            if args.torch_adam:
                torch.nn.utils.clip_grad_norm_(net.parameters(), 0.99)
            opt.step()
            ev[3].record()
            torch.cuda.synchronize()
            for i in range(3):
                acc[i] += ev[i].elapsed_time(ev[i + 1]) / 3
        split = {"forward_ms": round(acc[0], 3), "backward_ms" + ("_incl_allreduce" if world > 1 else ""): round(acc[1], 3),
                 "clip_adam_ms": round(acc[2], 3), "issued": "eager, from Python (host-bound upper bounds)"}
    if rank != 0:
        return None
    value = world * B * K / dt
    return {"metric": "training samples/sec (128-frame clips)", "value": round(value, 1), "unit": "samples/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16 GEMM operands / fp32 accumulate, fp32 convolutions and master weights", "data": "synthetic",
            "config": {"workload": f"diffusion_rvqvae_128.yaml training step: {B} clips/GPU (global {world * B}), training_losses fwd+bwd, "
                                   "clip 0.99, Adam 5e-5 (0.5, 0.999), MDM denoiser 8x512 + trained WavEncoder, random-init",
                       "clips_per_gpu": B, "global_batch": world * B,
                       "parallelism": f"dp{world}: one process per GPU, DDP bucketed all-reduce of 29.6 M gradients over RCCL" if ddp else "single GPU",
                       "graph_replayed": graph, "ddp_wrapper": ddp, **({"graph_fallback": fallback} if fallback else {}),
                       **({"gradients_written_into_buckets": {"direct": direct[0], "bound": direct[1], "copied_by_the_reducer": direct[2][:24]}} if direct else {})},
            "loss": round(loss, 5), "step_split": split,
            "roofline": {"bound": "mfma", "achieved": round(value / world * F_TRAIN / 1e12, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                         "frac": round(value / world * F_TRAIN / PEAK_BF16, 4), "traffic": None,
                         "note": "whole step (~170 launches, the largest 10 % of it: profiles/r06_train_step_split.txt), F_train = 17.7 GFLOP per sample; no single dominant kernel"}}


def run_guided(args, rank, local, world, dev, dist):
    """BASELINE configs[3] / [4]: classifier-free guidance as ONE fused batch (cfg_sampler.py:10-167).  The wrapper only plans the
    variants (guidance._Plan); all V variants of all clips share x_t and run as one step-kernel launch, combined in the output stage."""
    from syntalker_amd import engine, guidance, synth
    from syntalker_amd.denoiser_h3d import MDM
    from syntalker_amd.process import create_gaussian_diffusion

    K, W = args.steps, args.warmup
    model = synth.synth_fill_(MDM(synth.default_args()).eval(), seed=0).to(dev)
    model.m_tile, model.layer_mode = args.m_tile, args.layer_mode
    pm = model.packed()
    ddpm, ddim = create_gaussian_diffusion(), create_gaussian_diffusion(use_ddim=True)
    coef_ddpm, coef_ddim = engine.posterior_coefs(ddpm.tables(), dev), engine.ddim_coefs(ddim.tables(), 0.0, dev)
    g = torch.Generator().manual_seed(77)
    parts = {"upper_mask": torch.randn(1, 256, generator=g).to(dev), "hands_mask": None, "lower_mask": torch.randn(1, 256, generator=g).to(dev)}

    def one(tag, wrapper, B, y_extra, coef, noisy, evals, want_steady):
        """one guided workload -> dict of its numbers (rank 0) or None"""
        only = os.environ.get("SYN_BENCH_ONLY")
        if only and only not in tag:
            return {"workload": tag, "guided_clip_steps_per_s": 0.0, "ms_per_step": 0.0, "variants": 0, "reference_evaluation_equivalents_per_s": 0.0, "roofline": None}
        mdm, plan_fn = guidance.resolve(wrapper)
        if os.environ.get("SYN_BENCH_TRACE"):
            print(f"[guided] {tag[:40]} B={B}", file=sys.stderr, flush=True)
        chunk = 128
        plan = sb = None
        cond_ms = 0.0
        for b0 in range(0, B, chunk):                               # conditioning of all variants, once per clip, in chunks
            n = min(chunk, B - b0)
            y = synth.to_device(synth.synth_clip_inputs(n, seed=1000 * rank + b0 + 7, style_dim=256, style_zero=False), dev)
            y.update(y_extra)
            plan = plan_fn(y)
            V = len(plan.variants)
            if sb is None:
                sb = mdm.step_buffers(B, V)
                sb.cfg_w.copy_(plan.tensor(dev))
                mdm.variant_conds(y, plan.variants)                 # (first call of the process: lazy initialisation, not timed)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            cv = mdm.variant_conds(y, plan.variants)
            e1.record()
            sb.cond.view(V, B, 32, 512)[:, b0:b0 + n].copy_(cv.reshape(V, n, 32, 512))
            torch.cuda.synchronize()
            cond_ms += e0.elapsed_time(e1)
        V = len(plan.variants)
        sb.load_x(torch.randn(B, 1536, 1, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(rank)))
        sb.set_rng(4321, first_clip=rank * B)
        if os.environ.get("SYN_BENCH_TRACE"):
            torch.cuda.synchronize(); print("[guided] conditioning done", file=sys.stderr, flush=True)
        dt, replay_ms, launch_ms, steady_ms = timed_loop(pm, sb, coef, noisy, K, W, args.prime, world, dist, dev, want_steady and rank == 0)
        sb.check_sync()
        if rank != 0:
            return None
        value = world * B * K / dt
        kern = "k_seq" if getattr(sb, "fragment", False) else "library's choice (k_stack / k_lat)"
        per_step_s = (launch_ms / LOOP_CH if K >= LOOP_CH else replay_ms) * 1e-3
        out = {"workload": tag, "clips_per_gpu": B, "variants": V, "reference_evaluations_per_step": evals,
               "guided_clip_steps_per_s": round(value, 1), "ms_per_step": round(dt / K * 1e3, 4),
               "reference_evaluation_equivalents_per_s": round(value * evals, 1),
               "roofline": {"bound": "mfma", "kernel": kern, "achieved": round(B * V * F_STEP / per_step_s / 1e12, 2), "peak": PEAK_BF16 / 1e12,
                            "unit": "TFLOP/s", "frac": round(B * V * F_STEP / per_step_s / PEAK_BF16, 4),
                            "whole_step_frac": round(value / world * V * F_STEP / PEAK_BF16, 4), "traffic": None,
                            "note": "algorithmic FLOPs = the V unique evaluations of a guided step (the reference runs "
                                    f"{evals}); duration = hipEvents around the replays of the timed region"}}
        if steady_ms is not None:
            out["steady_state_ms_per_step"] = round(steady_ms, 4)
        # what the per-clip conditioning (outside the timed region: once per clip, all V variants) adds to a whole loop of this sampler
        L = 1000 if noisy else 50
        loop_ms = L * dt / K * 1e3
        out["per_clip_conditioning"] = {"ms_per_clip": round(cond_ms / B, 5), "loop_steps": L, "share_of_a_whole_loop": round(cond_ms / (cond_ms + loop_ms), 4),
                                        "clip_steps_per_s_including_it": round(world * B * L / ((cond_ms + loop_ms) * 1e-3), 1),
                                        "note": f"audio encoder + word / seed / style paths for the {V} variants of {B} clips, hipEvent-timed, NOT in guided_clip_steps_per_s: "
                                                f"it is paid once per clip, i.e. spread over the {L} steps of this sampler's loop"}
        del sb
        return out

    cfg = guidance.ClassifierFreeSampleModel(model)
    B3 = args.batch or 512
    main = one("configs[3] diffusion_rvqvae_128_hf.yaml shapes: ClassifierFreeSampleModel(scale 2.5) over the h3d-style denoiser, DDPM-1000 steps "
               "(noise drawn in the epilogue), cond + uncond as one fused batch", cfg, B3, {"scale": torch.ones(1, device=dev) * 2.5}, coef_ddpm, True, 2, True)
    d50 = one("configs[3], DDIM-50 steps (eta 0)", cfg, B3, {"scale": torch.ones(1, device=dev) * 2.5}, coef_ddim, False, 2, False)
    B4 = max(1, B3 // 2)
    body = one("configs[4] diffusion_h3d.yaml: TwoClassifierFreeSampleModel_Bodypart (upper + lower prompts, audio scale 1, prompt scale 4), DDIM-50 steps",
               guidance.TwoClassifierFreeSampleModel_Bodypart(model), B4, {"style_feature": parts}, coef_ddim, False, 9, False)
    if rank != 0:
        return None
    return {"metric": "guided denoising-steps/sec (128-frame clips, classifier-free guidance as one fused batch)", "value": main["guided_clip_steps_per_s"],
            "unit": "guided clip-steps/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": main["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": main["workload"], "clips_per_gpu": B3, "variants": main["variants"], "global_clips": world * B3,
                       "parallelism": f"clip-sharded x{world}, no collective", "primed_steps": args.prime * LOOP_CH},
            "reference_evaluation_equivalents_per_s": main["reference_evaluation_equivalents_per_s"],
            "roofline": main["roofline"], "steady_state_ms_per_step": main.get("steady_state_ms_per_step"), "per_clip_conditioning": main.get("per_clip_conditioning"),
            "ddim50": d50, "bodypart_twocfg": body}


def extras(args, local, dev) -> dict:
    """BASELINE configs[2] - [4] on the default 1-GPU line: `train_step` (the captured training step of --mode train, 20 timed steps; plus
    the SAME step under the DDP wrapper over a 1-rank RCCL group, run as a child process so that a collective that aborts cannot take
    the line with it) and `guided` (--mode guided's three workloads, 20 timed steps each).  Same code paths as the modes themselves."""
    import subprocess
    out = {}
    sub = argparse.Namespace(**vars(args))
    sub.batch, sub.steps, sub.warmup, sub.prime, sub.force_ddp = None, 20, 10, 0, False
    torch.cuda.empty_cache()
    try:
        tr = run_train(sub, 0, local, 1, dev, None)
        out["train_step"] = {"workload": tr["config"]["workload"], "ms_per_step": tr["ms_per_step"], "samples_per_s": tr["value"], "steps": tr["steps"],
                             "warmup": tr["warmup"], "graph_replayed": tr["config"]["graph_replayed"], "loss": tr["loss"],
                             "frac": tr["roofline"]["frac"], "achieved_tflops": tr["roofline"]["achieved"], "roofline_note": tr["roofline"]["note"],
                             "eager_step_split_ms": tr["step_split"]}
    except Exception as e:                                   # the headline line survives a failure of an extra
        out["train_step"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    try:
        cmd = [sys.executable, os.path.abspath(__file__), "--mode", "train", "--force-ddp", "--steps", "200", "--warmup", "10"]
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env)
        line = next((l for l in r.stdout.splitlines() if l.startswith("{")), None)
        if r.returncode == 0 and line:
            d = json.loads(line)
            out["train_step_ddp_1rank"] = {"ms_per_step": d["ms_per_step"], "samples_per_s": d["value"], "steps": d["steps"], "warmup": d["warmup"],
                                           "graph_replayed": d["config"]["graph_replayed"], "ddp_wrapper": d["config"]["ddp_wrapper"],
                                           "rccl_ranks": d.get("rccl_ranks"), "loss": d["loss"],
                                           "note": "the captured step under training.make_ddp(capturable=True) over a 1-rank RCCL group "
                                                   "(bucketed all-reduces inside the graph): the wrapper the N-GPU run uses; child process"}
        else:
            out["train_step_ddp_1rank"] = {"error": f"rc {r.returncode}", "stderr_tail": r.stderr[-600:]}
    except Exception as e:
        out["train_step_ddp_1rank"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        gd = run_guided(sub, 0, local, 1, dev, None)
        keep = ("workload", "clips_per_gpu", "variants", "reference_evaluations_per_step", "guided_clip_steps_per_s", "ms_per_step",
                "reference_evaluation_equivalents_per_s", "steady_state_ms_per_step", "per_clip_conditioning")
        def brief(d):
            b = {k: d[k] for k in keep if k in d}
            rf = d.get("roofline") or {}
            b.update(frac=rf.get("frac"), whole_step_frac=rf.get("whole_step_frac"), kernel=rf.get("kernel"))
            return b
        main_ = {"workload": gd["config"]["workload"], "clips_per_gpu": gd["config"]["clips_per_gpu"], "variants": gd["config"]["variants"],
                 "reference_evaluations_per_step": 2, "guided_clip_steps_per_s": gd["value"], "ms_per_step": gd["ms_per_step"],
                 "reference_evaluation_equivalents_per_s": gd["reference_evaluation_equivalents_per_s"],
                 "steady_state_ms_per_step": gd.get("steady_state_ms_per_step"), "roofline": gd["roofline"], "per_clip_conditioning": gd.get("per_clip_conditioning")}
        out["guided"] = {"steps": gd["steps"], "warmup": gd["warmup"], "cfg_ddpm": brief(main_), "cfg_ddim50": brief(gd["ddim50"]),
                         "bodypart_twocfg_ddim50": brief(gd["bodypart_twocfg"])}
    except Exception as e:
        out["guided"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default 1024 in sample mode, 32 in train mode)")
    ap.add_argument("--m-tile", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the B = 1 / 8 / 32 probe (clean per-kernel profiles)")
    ap.add_argument("--mode", choices=("sample", "train", "guided"), default="sample",
                    help="sample: BASELINE configs[1] (DDPM p_sample_loop steps); train: configs[2] (DDP training step, 32 clips/GPU); "
                         "guided: configs[3] / [4] (classifier-free guidance as one fused batch, h3d-style denoiser)")
    ap.add_argument("--prime", type=int, default=0, help="untimed 10-step replays in front of the --warmup steps (default 0: W is the only warm-up)")
    ap.add_argument("--train-graph", action=argparse.BooleanOptionalAction, default=None,
                    help="train mode: replay the whole step (incl. DDP's bucketed all-reduces) from one hipGraph; default on at every world size "
                         "(the step is ~170 launches: issued from Python it is host-bound, 7-9 ms against 4.2 ms replayed); --no-train-graph = the eager step")
    ap.add_argument("--force-ddp", action="store_true",
                    help="train mode with one GPU: still wrap the model in DDP over a 1-rank RCCL process group, i.e. time the wrapper the multi-GPU run uses")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="process-group backend of a multi-rank run (nccl = RCCL; gloo: functional checks)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="map the ranks onto the GPUs that exist (rank % count) - with --backend gloo a functional check of the N > 1 path on a 1-GPU box; "
                         "its throughput is NOT a scaling measurement")
    ap.add_argument("--no-extras", action="store_true",
                    help="sample mode: skip the train_step / guided sub-objects (BASELINE configs[2]-[4]) the default 1-GPU line carries")
    ap.add_argument("--torch-adam", action="store_true", help="train mode: torch.nn.utils.clip_grad_norm_ + torch.optim.Adam(fused) instead of training.ClipAdam (A/B)")
    ap.add_argument("--inject-capture-failure", action="store_true", help="train mode, test hook: the step's hipGraph capture raises, the run falls back to the eager step")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: exercises rank start-up, rendezvous (gloo), the barrier / max-over-ranks timing and the JSON line on CPU")
    ap.add_argument("--layer-mode", type=int, default=0, help="0 library's choice (whole-step kernel at the bench batch), 4 / 3 pin the whole-step / small-batch kernel, "
                         "5 the wave-per-sequence kernel, 1 five kernels per block (the restatement the whole-step kernel is checked against)", choices=[0, 1, 3, 4, 5])
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dist = None
    if args.dry_run:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            sys.exit("bench.py: no GPU visible (the hot path has no CPU fallback; --dry-run exercises the launch path only)")
        if args.share_gpu:                                      # functional check on a box with fewer GPUs than ranks: ranks share devices
            local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1 or (args.force_ddp and args.mode == "train"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_free_port()) if world == 1 else "29533")
        if args.dry_run:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            if args.mode == "train" and args.train_graph is not False:
                os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")   # whole-step capture: no watchdog thread on the stream
            if args.backend == "gloo":                          # (--share-gpu runs: RCCL refuses two ranks on one device)
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # backend "nccl" = RCCL on ROCm
    ranks_seen = dist.get_world_size() if dist is not None else 1
    if args.dry_run:
        out = dry_run(args, rank, world, dist)
    elif args.mode == "train":
        out = run_train(args, rank, local, world, dev, dist)
    elif args.mode == "guided":
        out = run_guided(args, rank, local, world, dev, dist)
    else:
        out = run_sample(args, rank, local, world, dev, dist)
    if rank == 0 and world == 1 and args.mode == "sample" and not args.dry_run and not args.no_extras:
        out.update(extras(args, local, dev))
    if rank == 0:
        out["rccl_ranks" if not (args.dry_run or args.backend == "gloo") else "gloo_ranks"] = ranks_seen
        if args.share_gpu and not args.dry_run:
            out["shared_gpu"] = f"{world} ranks on {torch.cuda.device_count()} device(s): a functional run of the multi-rank path, not a scaling number"
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()                 # ranks leave together (rank 0 may still have been timing the CPU baseline)
        dist.destroy_process_group()


LOOP_CH = 10          # steps per hipGraph replay of the hook-free stretches of a sampling loop (process.py `_fused`)


def timed_loop(pm, sb, coef, noisy, K, W, prime, world, dist, dev, want_steady):
    """The loop exactly as SpacedDiffusion.p_sample_loop / ddim_sample_loop run it (process.py `_fused`): the timestep schedule lives
    on the device, hook-free stretches replay a hipGraph of LOOP_CH = 10 steps (one schedule kernel + syn_denoise_steps: 10 step
    kernels, or ONE persistent 10-step launch of k_seq when the latent is in fragment order), the remainder a single-step graph.
    Noise (DDPM) ~ Philox(seed, step = t, global element index), drawn in the epilogue.
    W untimed steps, then exactly K steps between barrier + synchronize on both sides.
    -> (seconds of the K steps (max over ranks), per-step ms of the replays inside the timed region by hipEvents, ms of a
        LOOP_CH-step replay, per-step ms of 200 further steps - the steady state of a long loop - or None)"""
    from syntalker_amd import engine
    CH, MAXS = LOOP_CH, engine.StepGraph.MAX_STEPS
    n_rows = int(coef.shape[0])                             # 1000 (DDPM) or 50 (DDIM-50): rows of the coefficient table = steps of the loop
    tc = [n_rows - 1 - (i % n_rows) for i in range(MAXS)]   # step index K-1, K-2, ... (wraps: any step is a valid one to time)
    tm = [t * (1000 // n_rows) for t in tc]                 # its original timestep (respace.py: range(0, 1000, 20) for ddim50)
    g10 = engine.StepGraph(pm, sb, coef, noisy, fused_rng=noisy, scheduled=True, steps=CH)
    g10.set_schedule(tc, tm)
    g1 = None                                               # captured only when K or W is not a multiple of CH
    if K % CH or W % CH:
        g1 = engine.StepGraph(pm, sb, coef, noisy, fused_rng=noisy, scheduled=True)
        g1.set_schedule(tc, tm)
    state = {"pos": 0, "last": None}

    def run_steps(n, events=None):
        """n consecutive steps; events: list that receives (start, end, steps) hipEvent brackets on the replay stream."""
        done = 0
        while done < n:
            g, k = (g10, CH) if n - done >= CH else (g1, 1)
            if state["pos"] + k > MAXS:
                state["pos"], state["last"] = 0, None
            if state["last"] is not g:                       # hand the schedule position over (one tiny fill per switch)
                g.counter.fill_(state["pos"]); state["last"] = g
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record()
                events.append((e0, e1, k))
            else:
                g.replay()
            state["pos"] += k; done += k

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(prime):                                  # (--prime, default 0: not part of the contract's warm-up)
        run_steps(CH)
    run_steps(W)
    barrier()
    ev = []
    t0 = time.perf_counter()
    run_steps(K, ev)
    barrier()
    dt = time.perf_counter() - t0
    assert sum(k for _, _, k in ev) == K
    replay_ms = sum(a.elapsed_time(b) for a, b, _ in ev) / K   # average per-step duration of the replays inside the timed region
    big = [(a, b) for a, b, k in ev if k == CH]
    launch_ms = sum(a.elapsed_time(b) for a, b in big) / len(big) if big else replay_ms   # average duration of a CH-step replay
    timed_loop.last_replays_ms = [round(a.elapsed_time(b) / k, 4) for a, b, k in ev][:64]    # per-step ms of every replay of the timed region, in order
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(sb.x).all(), "non-finite latent after the timed steps"
    steady_ms = None
    if want_steady:
        # A 1000-step loop runs ~50 steps after a graph capture at the clocks it then holds (same box: 1.167 ms per step after 5
        # warm-up steps, 1.113 after 50).  Reported beside the contract's number, never instead of it.
        run_steps(3 * CH)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(20 * CH)
        torch.cuda.synchronize()
        steady_ms = (time.perf_counter() - t1) / (20 * CH) * 1e3
    return dt, replay_ms, launch_ms, steady_ms


def run_sample(args, rank, local, world, dev, dist):
    from syntalker_amd import _lib, engine, synth
    from syntalker_amd.denoiser import MDM
    from syntalker_amd.process import create_gaussian_diffusion

    B, K, W = args.batch or 1024, args.steps, args.warmup
    model = synth.synth_fill_(MDM(synth.default_args()).eval(), seed=0).to(dev)
    model.m_tile = args.m_tile
    model.layer_mode = args.layer_mode
    STAGE_INFO = {0: STAGE_INFO_STACK, 4: STAGE_INFO_STACK, 3: STAGE_INFO_STACK, 5: STAGE_INFO_STACK, 1: STAGE_INFO_LEGACY}[args.layer_mode]
    diff = create_gaussian_diffusion()
    pm = model.packed()

    # per-clip conditioning, once, in chunks (the audio encoder's activations are the only large temporaries).
    # Timed with hipEvents around the encoder calls only (synthetic-input generation and H2D copies excluded),
    # first chunk discarded as warm-up (MIOpen solver selection).
    sb = model.step_buffers(B, 1)
    chunk, cond_ms, cond_clips = 256, 0.0, 0
    for b0 in range(0, B, chunk):
        n = min(chunk, B - b0)
        y = synth.to_device(synth.synth_clip_inputs(n, seed=1000 * rank + b0), dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        c = pm.conditioner.cond(y)
        e1.record()
        sb.cond.view(B, 32, 512)[b0:b0 + n].copy_(c)
        torch.cuda.synchronize()
        if b0 > 0 or B <= chunk:
            cond_ms += e0.elapsed_time(e1); cond_clips += n
    cond_ms_per_clip = cond_ms / max(cond_clips, 1)

    x_T = torch.randn(B, 1536, 1, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
    sb.load_x(x_T)
    coef = engine.posterior_coefs(diff.tables(), dev)
    sb.set_rng(1234, first_clip=rank * B)
    dt, replay_ms, launch_ms, steady_ms = timed_loop(pm, sb, coef, True, K, W, args.prime, world, dist, dev, rank == 0)
    CH = LOOP_CH

    if rank == 0:
        value = world * B * K / dt
        # ---- per-kernel durations: hipEvents around every launch of 5 eager steps ----------------
        ms = (C.c_float * 8)()
        cnt = (C.c_int32 * 8)()
        tot, launches = [0.0] * 8, [0] * 8
        reps = 5 if args.layer_mode == 1 else 1     # single-kernel steps are timed by the replay brackets; one eager launch
                                                         # only names the kernel (keeps the rocprofv3 average = the replay average)
        frag = bool(getattr(sb, "fragment", False))   # the step runs on the wave-per-sequence kernel (latent in fragment order):
        if frag:                                      # every k_seq launch of this command is one CH-step replay (no eager launch
            reps = 0                                  # here: the rocprofv3 average of the kernel stays the replay average)
        for r in range(reps):
            sb.t_coef.fill_(500); sb.t_model.fill_(500)
            sb.c.coef = coef.data_ptr(); sb.c.noise = None; sb.c.rng = sb.rng.data_ptr()
            _lib.check(_lib.load().syn_denoise_step_profile(C.byref(pm.c), C.byref(sb.c), _lib.current_stream(), ms, cnt),
                       "syn_denoise_step_profile")
            for c in range(8):
                tot[c] += ms[c]; launches[c] += cnt[c]
        rename = {} if args.layer_mode == 1 else {"fc2_gemm": "step_kernel"}
        stage_ms = {STAGES[c]: tot[c] / max(reps, 1) for c in range(8) if launches[c]}
        # group by kernel symbol (proj and fc2 share one)
        by_kernel = {}
        for name, t in stage_ms.items():
            if name not in STAGE_INFO:
                continue
            sym, fl = STAGE_INFO[name]
            n_l = launches[STAGES.index(name)] // reps
            e = by_kernel.setdefault(sym, {"ms": 0.0, "launches": 0, "flops": 0.0})
            e["ms"] += t; e["launches"] += n_l; e["flops"] += fl * B * n_l
        spl = 1                                     # steps per launch of the dominant kernel
        if frag:
            # one persistent launch per CH-step replay (syn_denoise_steps): flops and duration per LAUNCH cover CH steps
            # (more workgroups than CUs: a CH-step replay is several launches, one CU-filling slice of the batch each, see step_impl)
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            slices = -(-((B + 3) // 4) // cus)
            dom, spl = "k_seq", (CH if K >= CH else 1)
            if spl == 1:
                slices = 1                                    # (a single step is one launch whatever the batch)
            d = {"ms": 0.0, "launches": slices, "flops": F_STEP * B * spl}
            avg_s = (launch_ms if spl == CH else replay_ms) * 1e-3 / slices
        else:
            dom = max(by_kernel, key=lambda k: by_kernel[k]["ms"])
            d = by_kernel[dom]
            avg_s = d["ms"] * 1e-3 / d["launches"]
            if args.layer_mode in (0, 3, 4, 5):      # the step IS one kernel: use the hipEvent brackets of the K timed replays
                avg_s = replay_ms * 1e-3
        achieved = d["flops"] / d["launches"] / avg_s
        # HBM/fabric bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this
        # same command, summarised and committed under profiles/; null when no summary matches this batch size.
        traffic = None
        tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                pj = json.load(open(tpath))
                if pj.get("batch") == B and pj.get("layer_mode", 0) == args.layer_mode and pj.get("steps_per_launch", 1) == spl:
                    traffic = pj.get("hbm_bytes_per_launch", {}).get(dom, pj.get("hbm_bytes_per_launch", {}).get(dom + "<MT>"))
            except Exception:
                traffic = None
        if traffic is not None and traffic / avg_s > 8.0e12:       # more bytes than HBM can move in one launch: a parsing artefact
            print(f"bench.py: roofline.traffic {traffic} B / {avg_s * 1e6:.0f} us exceeds 8 TB/s - dropped", file=sys.stderr)
            traffic = None
        roofline = {"bound": "mfma", "kernel": dom.replace("MT", str(args.m_tile or "auto")),
                    "achieved": round(achieved / 1e12, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16, 4), "traffic": traffic,
                    "avg_launch_us": round(avg_s * 1e6, 2), "launches_per_step": d["launches"] / spl, "steps_per_launch": spl,
                    "whole_step_frac": round(value / world * F_STEP / PEAK_BF16, 4),
                    "timed_region_replay_ms": round(replay_ms, 4),
                    "timed_region_replays_ms_per_step": getattr(timed_loop, "last_replays_ms", None),
                    "eager_stage_ms_per_step": {rename.get(k, k): round(v, 4) for k, v in stage_ms.items()}}
        if world == 1 and not args.no_extras:
            try:
                pp = practical_peak(dev)
                roofline["practical_peak"] = pp
                roofline["frac_of_practical_peak"] = round(achieved / 1e12 / pp["tflops"], 4)
            except Exception as e:                               # noqa: BLE001 - a diagnostic must not take the line with it
                roofline["practical_peak"] = {"error": f"{type(e).__name__}: {e}"}
        out = {
            "metric": "denoising-steps/sec (128-frame clips)", "value": round(value, 1), "unit": "clip-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "diffusion_rvqvae_128.yaml sampling: DDPM p_sample_loop steps (1000-step schedule), "
                                   f"{B} clips/GPU x (1536,1,32) latents, MDM denoiser 8x512, random-init",
                       "clips_per_gpu": B, "global_clips": world * B, "parallelism": f"clip-sharded x{world}, no collective",
                       "primed_steps": args.prime * CH,
                       "m_tile": args.m_tile or "auto"},
            "latency_note": "one step advances every clip of the batch; per-clip conditioning (audio encoder: HIP implicit-GEMM convs; word / seed / pooling: two fp32 HIP launches; "
                            f"once per clip, outside the timed region): {cond_ms_per_clip:.3f} ms/clip = "
                            f"{cond_ms_per_clip / (dt / K * 1e3 / B):.0f} denoising steps' worth",
            "roofline": roofline,
            "steady_state": {"ms_per_step": round(steady_ms, 4), "clip_steps_per_s": round(B / steady_ms * 1e3, 1),
                             "whole_step_frac": round(B / steady_ms * 1e3 * F_STEP / PEAK_BF16, 4),
                             "note": "200 further steps after the timed region (wall clock, this rank): the clocks a 1000-step loop runs at"},
        }
        if args.layer_mode == 0 and not args.no_small_batch and world == 1:
            out["small_batch"] = small_batch_probe(pm, coef, dev)
        if not args.no_cpu and world == 1:      # the CPU baseline is a 1-GPU-run item: the other ranks would only wait for it
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        return out
    return None


if __name__ == "__main__":
    main()
