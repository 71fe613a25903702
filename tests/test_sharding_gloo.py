"""world_size-2 gloo tests (CPU) of the multi-GPU paths: clip-sharded sampling (slices, first_clip bookkeeping, ragged gather)
and data-parallel training (DDP wiring of `training.make_ddp`).  First with toy modules (the sharding / DDP code alone), then
with the PRODUCT's `MDM` - 29.6 M parameters, its packed() / step_buffers() / variant_conds() plumbing, `process._fused` - over the CPU
stand-ins of the device engine in tests/cpu_engine.py (the HIP kernels need a GPU; the stand-ins run the oracle's arithmetic)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from syntalker_amd import process
from syntalker_amd.sharding import gather_clips, sample_sharded, shard_range


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor(0.6))

    def forward(self, x, t, y=None):
        return self.w * x + y["seed"].mean(dim=(1, 2)).view(-1, 1, 1, 1) + 0.001 * t.view(-1, 1, 1, 1).float()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case(n=5):
    g = torch.Generator().manual_seed(0)
    shape = (n, 6, 1, 4)
    y = {"seed": torch.randn(n, 4, 6, generator=g), "mask": torch.ones(n, 1, 1, 4, dtype=torch.bool), "scalar": 3}
    return shape, y, torch.randn(*shape, generator=g), torch.randn(8, *shape, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shape, y, xT, sn = _case()
        d = process.create_gaussian_diffusion()
        full = sample_sharded(d, Toy(), shape, {"y": y}, noise=xT, clip_denoised=False, skip_timesteps=992, step_noise=sn)
        lo, hi = shard_range(shape[0], rank, world)
        ragged = gather_clips(torch.full((hi - lo, 2), float(rank)), shape[0])
        # no x_T given: every rank draws its clips' x_T from (seed, GLOBAL clip index); with seed=None rank 0's draw is broadcast
        drawn = sample_sharded(d, Toy(), shape, {"y": y}, noise=None, seed=11, clip_denoised=False, skip_timesteps=992, step_noise=sn)
        torch.manual_seed(100 + rank)                      # ranks whose own RNGs disagree must still agree on the broadcast seed
        auto = sample_sharded(d, Toy(), shape, {"y": y}, noise=None, clip_denoised=False, skip_timesteps=992, step_noise=sn, gather=False)
        auto_all = gather_clips(auto, shape[0])
        if rank == 0:
            torch.save({"full": full, "ragged": ragged, "drawn": drawn, "auto": auto_all}, out)
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    for n in (1, 5, 8, 1023):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_two_rank_sharded_sampling_equals_single_process(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    shape, y, xT, sn = _case()
    d = process.create_gaussian_diffusion()
    want = d.p_sample_loop(Toy(), shape, noise=xT, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=992, step_noise=sn)
    assert torch.equal(got["full"], want)
    assert got["ragged"].shape == (5, 2) and got["ragged"][:3].eq(0).all() and got["ragged"][3:].eq(1).all()
    # x_T drawn inside: identical to one process drawing all five clips from the same seed (rank-count independence) ...
    assert torch.equal(got["drawn"], sample_sharded(d, Toy(), shape, {"y": y}, noise=None, seed=11, clip_denoised=False,
                                                    skip_timesteps=992, step_noise=sn))
    assert not torch.equal(got["drawn"], want)
    # ... and distinct clips got distinct x_T although both ranks' torch RNGs were seeded alike
    assert (got["drawn"][0] - got["drawn"][3]).abs().max() > 1e-3
    assert torch.isfinite(got["auto"]).all() and got["auto"].shape == want.shape


# ---- data-parallel training wiring: torch DDP as configured by training.make_ddp, gloo on CPU ----
def _ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from syntalker_amd.training import make_ddp
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        net.add_module("unused", torch.nn.Linear(2, 2))          # like embed_style: never reached in forward
        fwd = lambda m, x: m[2](m[1](m[0](x)))
        g = torch.Generator().manual_seed(7)
        data = torch.randn(8, 6, generator=g)
        lo, hi = shard_range(8, rank, world)

        class Wrap(torch.nn.Module):
            def __init__(self, m):
                super().__init__()
                self.m = m

            def forward(self, x):
                return fwd(self.m, x)
        w = make_ddp(Wrap(net))
        w.zero_grad()
        ((w(data[lo:hi])) ** 2).mean().backward()
        if rank == 0:
            torch.save({k: p.grad.clone() for k, p in w.module.m.named_parameters() if p.grad is not None}, out)
    finally:
        dist.destroy_process_group()


def test_two_rank_ddp_gradients_equal_full_batch(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_ddp_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    data = torch.randn(8, 6, generator=torch.Generator().manual_seed(7))
    ((net(data)) ** 2).mean().backward()                          # equal shards: mean of shard means == full mean
    for k, p in net.named_parameters():
        assert torch.allclose(got[k], p.grad, atol=1e-6), k
    assert not any(k.startswith("unused") for k in got)


def _ddp_capturable_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from syntalker_amd.training import make_ddp

        class Net(torch.nn.Module):
            def __init__(self):
                super().__init__()
                torch.manual_seed(0)
                self.a, self.b = torch.nn.Linear(6, 5), torch.nn.Linear(5, 3)
                self.embed_style = torch.nn.Linear(6, 4)           # defined, never used (models/denoiser.py)

            def forward(self, x):
                return self.b(torch.tanh(self.a(x)))
        data = torch.randn(8, 6, generator=torch.Generator().manual_seed(7))
        lo, hi = shard_range(8, rank, world)
        w = make_ddp(Net(), capturable=True)                        # no unused-parameter search: the unused ones are frozen
        grads = []
        for _ in range(3):                                          # a second / third iteration would raise if a trainable
            w.zero_grad()                                           # parameter had been left without a gradient
            (w(data[lo:hi]) ** 2).mean().backward()
            grads.append({k: p.grad.clone() for k, p in w.module.named_parameters() if p.grad is not None})
        if rank == 0:
            torch.save({"grads": grads[-1], "frozen": [k for k, p in w.module.named_parameters() if not p.requires_grad],
                        "find_unused": w.find_unused_parameters}, out)
    finally:
        dist.destroy_process_group()


def test_two_rank_ddp_prepared_for_graph_capture(tmp_path):
    out = str(tmp_path / "gc.pt")
    mp.spawn(_ddp_capturable_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["find_unused"] is False and sorted(got["frozen"]) == ["embed_style.bias", "embed_style.weight"]
    torch.manual_seed(0)
    a, b = torch.nn.Linear(6, 5), torch.nn.Linear(5, 3)
    data = torch.randn(8, 6, generator=torch.Generator().manual_seed(7))
    # rank-mean of the two half-batch means = full-batch mean
    (b(torch.tanh(a(data))) ** 2).mean().backward()
    for k, ref in (("a.weight", a.weight.grad), ("b.weight", b.weight.grad), ("b.bias", b.bias.grad)):
        assert torch.allclose(got["grads"][k], ref, atol=1e-6), k


def _ddp_direct_grad_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from syntalker_amd import training

        class LinFn(torch.autograd.Function):                       # a backward that asks `training._grad_out` where to write, like the HIP ones
            @staticmethod
            def forward(ctx, x, w):
                ctx.save_for_backward(x)
                ctx.owner = w
                return x @ w.detach().t()

            @staticmethod
            def backward(ctx, dy):
                (x,) = ctx.saved_tensors
                dw = training._grad_out(ctx.owner, ctx.owner.shape)
                torch.mm(dy.t(), x, out=dw)
                return dy @ ctx.owner.detach(), dw

        class Net(torch.nn.Module):
            def __init__(self):
                super().__init__()
                torch.manual_seed(0)
                self.w1, self.w2 = torch.nn.Parameter(torch.randn(5, 6) * 0.3), torch.nn.Parameter(torch.randn(3, 5) * 0.3)
                self.embed_style = torch.nn.Linear(6, 4)

            def forward(self, x):
                return LinFn.apply(torch.tanh(LinFn.apply(x, self.w1)), self.w2)
        data = torch.randn(8, 6, generator=torch.Generator().manual_seed(7))
        lo, hi = shard_range(8, rank, world)
        w = training.make_ddp(Net(), capturable=True)      # averaging comm hook registered (gloo: torch's divide + all-reduce)
        log = []
        for it in range(5):
            w.zero_grad(set_to_none=True)                                 # what GraphedTrainStep does in front of every backward
            (w(data[lo:hi]) ** 2).mean().backward()
            if it == 2:
                assert training.bind_grad_buffers(w) == 2                 # buckets rebuilt: .grad is a bucket view from here on
            bufs = {k: getattr(p, "_syn_grad_buf", None) for k, p in w.module.named_parameters()}
            log.append({"grads": {k: p.grad.clone() for k, p in w.module.named_parameters() if p.grad is not None},
                        "aliased": all(bufs[k] is not None and p.grad.data_ptr() == bufs[k].data_ptr()
                                       for k, p in w.module.named_parameters() if p.grad is not None)})
        # a parameter that still HOLDS a gradient is never handed its bound buffer (accumulation must not overwrite)
        (w(data[lo:hi]) ** 2).mean().backward()
        acc = {k: p.grad.clone() for k, p in w.module.named_parameters() if p.grad is not None}
        training.unbind_grad_buffers(w)
        if rank == 0:
            torch.save({"log": log, "acc": acc, "left": [k for k, p in w.module.named_parameters() if hasattr(p, "_syn_grad_buf")]}, out)
    finally:
        dist.destroy_process_group()


def test_two_rank_ddp_with_gradients_written_into_the_buckets(tmp_path):
    """`training.bind_grad_buffers` + `_grad_out` + the averaging comm hook (the captured DDP step's gradient path, on gloo): the
    rank-averaged gradients are the full-batch ones before and after binding, after binding every .grad IS its bucket view, and a
    backward on top of existing gradients accumulates instead of overwriting."""
    out = str(tmp_path / "dg.pt")
    mp.spawn(_ddp_direct_grad_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    w1, w2 = torch.nn.Parameter(torch.randn(5, 6) * 0.3), torch.nn.Parameter(torch.randn(3, 5) * 0.3)
    data = torch.randn(8, 6, generator=torch.Generator().manual_seed(7))
    ((torch.tanh(data @ w1.t()) @ w2.t()) ** 2).mean().backward()
    for it, rec in enumerate(got["log"]):
        assert sorted(rec["grads"]) == ["w1", "w2"]
        for k, ref in (("w1", w1.grad), ("w2", w2.grad)):
            assert torch.allclose(rec["grads"][k], ref, atol=1e-6), (it, k)
        assert rec["aliased"] == (it >= 2), it
    for k, ref in (("w1", w1.grad), ("w2", w2.grad)):
        assert torch.allclose(got["acc"][k], 2 * ref, atol=1e-6), k
    assert got["left"] == []


# ---- the product MDM through process._fused, clip-sharded over 2 ranks (engine = tests/cpu_engine.py) -------------------------
def _mdm_case(variant):
    from syntalker_amd import synth
    n = 5
    if variant == "h3d":
        y = synth.synth_clip_inputs(n, seed=21, style_dim=256, style_zero=False)
        y["scale"] = torch.ones(1) * 2.5
    else:
        y = synth.synth_clip_inputs(n, seed=21)
    return (n, 1536, 1, 32), y


def _mdm_model(variant):
    from syntalker_amd import guidance, synth
    if variant == "h3d":
        from syntalker_amd.denoiser_h3d import MDM
        return guidance.ClassifierFreeSampleModel(synth.synth_fill_(MDM(synth.default_args()).eval(), seed=0))
    from syntalker_amd.denoiser import MDM
    return synth.synth_fill_(MDM(synth.default_args()).eval(), seed=0)


def _mdm_sample(variant, sharded):
    """24 DDPM steps with the noise drawn per (seed, step, global clip): two 10-step replays + four single steps."""
    from tests import cpu_engine
    cpu_engine.install_plain()
    d = process.create_gaussian_diffusion()
    shape, y = _mdm_case(variant)
    model = _mdm_model(variant)
    kw = dict(noise=None, seed=11, clip_denoised=False, skip_timesteps=976)
    if sharded:
        return sample_sharded(d, model, shape, {"y": dict(y)}, **kw)
    from syntalker_amd.sharding import draw_x_T
    return d.p_sample_loop(model, shape, noise=draw_x_T(shape[0], shape[1:], 11, 0, "cpu"), clip_denoised=False,
                           model_kwargs={"y": dict(y)}, skip_timesteps=976, seed=11)


def _mdm_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = {v: _mdm_sample(v, True) for v in ("beatx", "h3d")}
        if rank == 0:
            torch.save(res, out)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_sampling_of_the_product_mdm(tmp_path):
    """sample_sharded -> p_sample_loop -> process._fused -> MDM.packed / buffers / variant_conds -> StepGraph schedule hand-over,
    with the plain BEAT-X model and with ClassifierFreeSampleModel over the h3d one (V = 2 fused variants, cfg weights): the 2-rank
    result equals the single-process one - x_T and the step noise follow the GLOBAL clip index (first_clip), not the local one."""
    out = str(tmp_path / "mdm.pt")
    mp.spawn(_mdm_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    for v in ("beatx", "h3d"):
        want = _mdm_sample(v, False)
        assert got[v].shape == want.shape == (5, 1536, 1, 32) and torch.isfinite(want).all()
        err = float((got[v] - want).norm() / want.norm())
        assert err < 1e-5, (v, err)                    # (CPU GEMMs of 2 / 3 clips against 5: fp32 re-association only)
        # the wrong bookkeeping (local clip index on rank 1) would give clip 3 the noise of clip 0
        assert float((got[v][3] - got[v][0]).abs().max()) > 1e-3


def test_batch_between_two_pass_sizes_runs_as_slices(monkeypatch):
    """engine.plan_slices: q P + r clips with 0 < r <= P / 2 (P = clips of one full wave-per-sequence pass) run as the first q P clips
    and the rest, and `process._fused` over the slices gives the unsliced batch's result - x_T is the single draw, the step noise
    follows the global clip index (first_clip + lo), guided y entries are sliced with the clips."""
    from syntalker_amd import engine
    from tests import cpu_engine
    monkeypatch.setattr(engine, "seq_pass_clips", lambda V, dev: 0 if not 1 <= V <= 4 else 256 * (4 // V))
    ps = lambda n, V=1: engine.plan_slices(n, V, "cpu")
    assert ps(1024) == [(0, 1024)] and ps(700) == [(0, 700)] and ps(1537) == [(0, 1537)] and ps(2048) == [(0, 2048)]
    assert ps(1025) == [(0, 1024), (1024, 1025)] and ps(1536) == [(0, 1024), (1024, 1536)] and ps(2560) == [(0, 2048), (2048, 2560)]
    assert ps(768, 2) == [(0, 512), (512, 768)] and ps(300, 4) == [(0, 256), (256, 300)] and ps(400, 4) == [(0, 400)]
    assert ps(700, 3) == [(0, 700)] and ps(5000, 5) == [(0, 5000)]
    monkeypatch.undo()
    assert engine.plan_slices(1536, 1, "cpu") == [(0, 1536)]            # (no device, no kernel: nothing to plan)

    cpu_engine.install(monkeypatch)
    d = process.create_gaussian_diffusion()
    for variant in ("beatx", "h3d"):
        shape, y = _mdm_case(variant)
        model = _mdm_model(variant)
        x_T = torch.randn(*shape, generator=torch.Generator().manual_seed(3))
        run = lambda: d.p_sample_loop(model, shape, noise=x_T, clip_denoised=False, model_kwargs={"y": dict(y)}, skip_timesteps=988, seed=11)
        whole = run()
        calls = []
        monkeypatch.setattr(engine, "plan_slices", lambda n, V, dev: calls.append((n, V)) or [(0, 3), (3, n)])
        sliced = run()
        monkeypatch.setattr(engine, "plan_slices", lambda n, V, dev: [(0, n)])
        assert calls == [(5, 2 if variant == "h3d" else 1)]
        assert sliced.shape == whole.shape and float((sliced - whole).norm() / whole.norm()) < 1e-5, variant
        assert float((sliced[3] - sliced[0]).abs().max()) > 1e-3       # (local clip index in the second slice would repeat clip 0's noise)


# ---- DDP over the product's real parameter set (training.make_ddp, capturable: frozen unused parameters, no search) ---------
def _oracle_train_forward(m, x, timesteps, y, drop_path=0.0):
    """CPU stand-in for training.train_forward: the oracle's functional forward over the MODULE's own parameters (train-mode
    BatchNorm), so autograd reaches every parameter the product's forward reaches."""
    from oracle import denoiser_ref as dr
    sd = dict(m.state_dict(keep_vars=True))          # (the live Parameters; shared buffers under every name the reference uses)
    return dr.mdm_forward(sd, x, timesteps, y, variant=m.variant, train_bn=m.training)


def _mdm_ddp_step(model, lo, hi):
    from syntalker_amd import synth
    d = process.create_gaussian_diffusion()
    y = synth.synth_clip_inputs(4, seed=5)
    x0, eps = synth.synth_latent(4, seed=5, name="x0"), synth.synth_latent(4, seed=6, name="eps")
    t4 = torch.tensor([0, 17, 500, 999])
    from syntalker_amd.sharding import shard_kwargs
    loss = d.training_losses(model, x0[lo:hi], t4[lo:hi], model_kwargs={"y": shard_kwargs(y, lo, hi, 4)}, noise=eps[lo:hi])["loss"].mean()
    loss.backward()
    return float(loss)


def _mdm_ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from syntalker_amd import synth, training
        from syntalker_amd.denoiser import MDM
        training.train_forward = _oracle_train_forward
        model = synth.synth_fill_(MDM(synth.default_args()).eval(), seed=0)     # eval(): running-statistics BatchNorm, so that the
        model.differentiable_eval = True                                         # rank-mean of shard gradients IS the full-batch gradient
        w = training.make_ddp(model, capturable=True)
        losses = []
        for _ in range(3):                      # (a second iteration raises if a trainable parameter was left without a gradient; the
                                                # bucket plan is rebuilt after the first and reported from the third on)
            w.zero_grad()
            lo, hi = shard_range(4, rank, world)
            losses.append(_mdm_ddp_step(w, lo, hi))
        if rank == 0:
            names = ["mytimmblocks.0.attn.qkv.weight", "mytimmblocks.7.mlp.fc2.weight", "WavEncoder.feat_extractor.0.conv1.weight",
                     "text_pre_encoder_body.weight", "output_process.poseFinal.bias"]
            params = dict(w.module.named_parameters())
            torch.save({"grads": {n: params[n].grad.clone() for n in names},
                        "frozen": sorted(k for k, p in params.items() if not p.requires_grad),
                        "n_params": sum(p.numel() for p in params.values()),
                        "buckets": training.ddp_bucket_sizes(w), "find_unused": w.find_unused_parameters}, out)
    finally:
        dist.destroy_process_group()


def test_two_rank_ddp_over_the_real_parameter_set(tmp_path):
    """make_ddp(capturable=True) around the product MDM (29.6 M parameters): the parameters the forward never reaches are frozen,
    no unused-parameter search, the bucket plan covers the 118 MB of gradients in several all-reduces, and the rank-averaged
    gradients of two half batches equal the single-process gradient of the whole batch (reference seam: train.py:87-94)."""
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_mdm_ddp_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["find_unused"] is False and got["n_params"] == 29_607_012
    assert got["frozen"] == ["embed_style.bias", "embed_style.weight"]
    mb = [b / 2 ** 20 for b in got["buckets"]]
    assert len(mb) >= 4 and max(mb) <= 40 and abs(sum(mb) - 29_607_012 * 4 / 2 ** 20 + (6 * 64 + 64) * 4 / 2 ** 20) < 1.0, mb
    from syntalker_amd import synth, training
    from syntalker_amd.denoiser import MDM
    orig = training.train_forward
    training.train_forward = _oracle_train_forward
    try:
        model = synth.synth_fill_(MDM(synth.default_args()).eval(), seed=0)
        model.differentiable_eval = True
        _mdm_ddp_step(model, 0, 4)
    finally:
        training.train_forward = orig
    params = dict(model.named_parameters())
    for n, g in got["grads"].items():
        err = float((g - params[n].grad).norm() / params[n].grad.norm())
        assert err < 1e-4, (n, err)


# ---- per-rank data sharding of the training driver (reference train.py:54-61, :277: DistributedSampler + set_epoch) --------------------
def test_epoch_indices_equal_torch_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    from syntalker_amd.sharding import epoch_indices
    for n in (1, 7, 40, 101):
        for world in (1, 2, 3, 8):
            for epoch in (0, 1, 5):
                for drop_last in (False, True):
                    for r in range(world):
                        ref = DistributedSampler(range(n), num_replicas=world, rank=r, shuffle=True, seed=3, drop_last=drop_last)
                        ref.set_epoch(epoch)
                        assert epoch_indices(n, r, world, epoch, seed=3, drop_last=drop_last).tolist() == list(ref), (n, world, epoch, r, drop_last)
    assert epoch_indices(10, 1, 4, 0, shuffle=False).tolist() == [1, 5, 9]      # padded by wrapping: 0..9, 0, 1 dealt round-robin


def _sampler_worker(rank, world, port, out, npz):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("train_from_config", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                         "scripts", "train_from_config.py"))
        drv = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(drv)
        seen = []
        for epoch in (0, 1):
            # the driver's own batch iterator: the latent's first value carries the item's index
            ids = [x0[:, 0, 0, 0].long() for x0, _ in drv.batches_from(npz, None, 3, "cpu", rank, world, epoch, seed=0)]
            seen.append(torch.cat(ids) if ids else torch.zeros(0, dtype=torch.long))
        mine = torch.stack(seen)                                              # (epochs, items of this rank)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        if rank == 0:
            torch.save(torch.stack(parts), out)                              # (rank, epoch, item)
    finally:
        dist.destroy_process_group()


def test_two_rank_training_driver_batches_are_disjoint_and_cover_the_epoch(tmp_path):
    import numpy as np
    n, B = 14, 3
    npz = str(tmp_path / "data.npz")
    lat = np.zeros((n, 1536, 1, 32), np.float32)
    lat[:, 0, 0, 0] = np.arange(n)
    np.savez(npz, latent=lat, audio=np.zeros((n, 8, 2), np.float32), word=np.zeros((n, 128), np.int64), seed=np.zeros((n, 4, 1536), np.float32))
    out = str(tmp_path / "seen.pt")
    mp.spawn(_sampler_worker, args=(2, _free_port(), out, npz), nprocs=2, join=True)
    seen = torch.load(out)                                                    # (2 ranks, 2 epochs, 6 items): 7 per rank, batches of 3, last dropped
    assert seen.shape == (2, 2, 6)
    for e in range(2):
        a, b = set(seen[0, e].tolist()), set(seen[1, e].tolist())
        assert len(a) == 6 and len(b) == 6 and not (a & b), (a, b)          # disjoint index sets: a global batch of 2 x B per step
        assert (a | b) <= set(range(n)) and len(a | b) == 12                 # 12 of the 14 items (each rank's 7th item falls to drop_last)
    assert seen[:, 0].tolist() != seen[:, 1].tolist()                        # set_epoch: a new permutation per epoch
    # over the ranks' full index lists (before batching) the epoch is covered exactly
    from syntalker_amd.sharding import epoch_indices
    both = torch.cat([epoch_indices(n, r, 2, 0) for r in range(2)])
    assert sorted(both.tolist()) == list(range(n))
