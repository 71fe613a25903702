"""Clip-sharded sampling across the GPUs of one node.

Inference shards with NO data-path collective: clips are independent, every rank holds the full
(38 MB) weight set and runs its own hipGraph loop over a contiguous slice of the clip batch; the only
communication is the optional gather of the finished latents.  Noise is keyed by (seed, step, GLOBAL
clip index) (syn_randn / the in-epilogue Philox), so the result does not depend on the number of ranks.
One process per GPU, ``torch.distributed`` (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).

Windows of ONE long sequence are sequential (each window's seed is the tail of the previous one,
diffusion_rvqvae_trainer.py:428-431): shard across sequences, never across windows.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_clips: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of the clip batch owned by `rank` (first ranks get the remainder)."""
    per, extra = divmod(n_clips, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def epoch_indices(n_items: int, rank: int, world: int, epoch: int, seed: int = 0, shuffle: bool = True, drop_last: bool = False) -> torch.Tensor:
    """The indices rank `rank` of `world` walks in epoch `epoch` - torch.utils.data.distributed.DistributedSampler's rule, which is what the
    reference's DDP branch hands its DataLoader (train.py:60, `sampler.set_epoch(epoch)` at train.py:277): one permutation of the whole
    set per epoch from a generator seeded with seed + epoch (identical on every rank), padded by wrapping around to a multiple of `world`
    (or cut to one with drop_last), then dealt round-robin: rank r takes positions r, r + world, ...  The ranks' index sets are therefore
    disjoint (up to the wrap-around padding) and together cover the set; every rank gets the same count, so DDP's collectives line up."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside [0, {world})")
    if shuffle:
        order = torch.randperm(n_items, generator=torch.Generator().manual_seed(seed + epoch))
    else:
        order = torch.arange(n_items)
    if drop_last and n_items % world:
        total = n_items // world * world
        order = order[:total]
    else:
        total = -(-n_items // world) * world
        pad = total - n_items
        if pad:
            reps = -(-pad // max(n_items, 1))
            order = torch.cat([order, order.repeat(reps)[:pad]])
    return order[rank:total:world]


def epoch_batches(n_items: int, batch_size: int, rank: int, world: int, epoch: int, seed: int = 0, shuffle: bool = True):
    """This rank's batches of item indices for one epoch: `epoch_indices` cut into batches of `batch_size`, last partial batch dropped
    (the reference's DataLoader(..., batch_size, drop_last=True, sampler=DistributedSampler), train.py:54-61).  With one rank this is the
    reference's non-DDP loader: a fresh shuffle of the whole set per epoch."""
    idx = epoch_indices(n_items, rank, world, epoch, seed, shuffle)
    for lo in range(0, idx.numel() - batch_size + 1, batch_size):
        yield idx[lo:lo + batch_size]


def shard_kwargs(y: dict, lo: int, hi: int, n_clips: int) -> dict:
    """Slice every per-clip tensor of model_kwargs['y'] (leading dim == n_clips); leave the rest alone."""
    out = {}
    for k, v in y.items():
        if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n_clips:
            out[k] = v[lo:hi]
        elif isinstance(v, dict):
            out[k] = shard_kwargs(v, lo, hi, n_clips)
        else:
            out[k] = v
    return out


def gather_clips(local: torch.Tensor, n_clips: int, group=None) -> torch.Tensor:
    """all_gather of ragged per-rank slices -> the full (n_clips, ...) tensor on every rank."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(n_clips, r, world) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


X_T_STREAM = 1 << 20        # stream id of the x_T draw (the per-step noise uses stream ids 0 .. n_steps - 1)


def draw_x_T(n_local: int, shape_tail, seed: int, first_clip: int, device) -> torch.Tensor:
    """x_T ~ N(0, I) for the clips [first_clip, first_clip + n_local) of a global batch, keyed by (seed, GLOBAL element
    index): the same values whatever the number of ranks.  Counter-based generator of the step kernels (syn_randn) on the
    GPU; on CPU (host-logic tests) one seeded torch generator per clip."""
    per = int(torch.tensor(shape_tail).prod())
    if torch.device(device).type == "cuda":
        from . import _lib
        out = torch.empty((n_local,) + tuple(shape_tail), dtype=torch.float32, device=device)
        _lib.check(_lib.load().syn_randn(out.data_ptr(), out.numel(), seed, X_T_STREAM, first_clip * per, _lib.current_stream(out.device)),
                   "syn_randn")
        return out
    rows = [torch.randn(tuple(shape_tail), generator=torch.Generator().manual_seed((seed * 1000003 + first_clip + i) % (2 ** 63)))
            for i in range(n_local)]
    return torch.stack(rows).to(device) if rows else torch.empty((0,) + tuple(shape_tail), device=device)


def sample_sharded(diffusion, model, shape, model_kwargs, *, ddim=False, noise=None, seed=None, gather=True, **loop_kw):
    """p_sample_loop / ddim_sample_loop over this rank's slice of a global batch of shape[0] clips.

    `noise` (x_T) and per-clip entries of model_kwargs['y'] are GLOBAL tensors; each rank slices its part.  Without
    `noise`, x_T is drawn per clip from (seed, global clip index) (`draw_x_T`), and without `seed` rank 0 draws one and
    broadcasts it: in every case the result is independent of the number of ranks.
    Returns the full batch on every rank when `gather`, else the local slice."""
    n = shape[0]
    on = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if on else 0
    world = dist.get_world_size() if on else 1
    lo, hi = shard_range(n, rank, world)
    if seed is None:
        dev0 = next(model.parameters()).device
        t = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
        if on and world > 1:
            t = t.to(dev0) if dist.get_backend() == "nccl" else t
            dist.broadcast(t, src=0)
        seed = int(t.item())
    if noise is None:
        noise_local = draw_x_T(hi - lo, shape[1:], seed, lo, next(model.parameters()).device)
    else:
        noise_local = noise[lo:hi]
    kw = dict(model_kwargs)
    kw["y"] = shard_kwargs(model_kwargs["y"], lo, hi, n)
    loop = diffusion.ddim_sample_loop if ddim else diffusion.p_sample_loop
    if "step_noise" in loop_kw and loop_kw["step_noise"] is not None:
        loop_kw = dict(loop_kw, step_noise=loop_kw["step_noise"][:, lo:hi])
    local = loop(model, (hi - lo,) + tuple(shape[1:]), noise=noise_local, model_kwargs=kw, seed=seed, first_clip=lo, **loop_kw)
    return gather_clips(local, n) if gather else local
