"""world_size-2 gloo test (CPU) of the clip-sharded sampling path: slices, first_clip bookkeeping, ragged
gather.  The denoiser is a stand-in module (the HIP MDM needs a GPU); the sharding code is the product's."""
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
