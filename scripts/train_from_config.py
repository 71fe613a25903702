#!/usr/bin/env python3
"""The reference's `train.py -c <yaml>` loop on this build (train.py:255-285, diffusion_rvqvae_trainer.py:543-575):
YAML -> args namespace -> denoiser named by `model:` / `g_name:` in train() mode + diffusion + uniform schedule sampler + Adam
(lr_base, opt_betas) -> for every epoch: the training step over the epoch's batches (sample t, training_losses forward, backward,
clip_grad_norm(grad_norm), Adam), then the StepLR scheduler's epoch step (`opt_s.step(epoch)`), and every `test_period` epochs
`save_checkpoints(<out>/last_<epoch>.bin)` in the reference's format ({'model_state': state_dict}).

    python scripts/train_from_config.py configs.yaml [--epochs E] [--steps-per-epoch S] [--batch-size B] [--out DIR]
                                        [--resume last_N.bin] [--data batches.npz] [--graph] [--seed 0] [--random-init]

The reference's data loaders (lmdb caches of BEAT-X / HumanML3D, dataloaders/*.py) are out of scope and their data is not in this
image: without --data the loop trains on synthetic batches of the shapes `_load_data` hands to `_g_training` (x0 (B,1536,1,32) =
the RVQ-VAE latents / vqvae_latent_scale, audio (B, 68266, 2), word ids (B, 128), seed latents (B, 4, 1536)).
--data      npz with arrays latent (N,1536,1,32) [already divided by vqvae_latent_scale], audio (N,68266,2), word (N,128), seed (N,4,1536)
            [, style_feature (N,256) for the h3d configuration]; an epoch walks it in batches of batch_size.
--resume    a checkpoint of the reference's format; keys with nn.DataParallel's `module.` prefix are accepted.
--graph     replay the whole step from one hipGraph (training.GraphedTrainStep; static batch shape).
With more than one rank (torchrun) the model is wrapped by training.make_ddp and every rank trains on its own batches: with --data the
epoch's permutation is dealt over the ranks as the reference's DistributedSampler does (train.py:60, set_epoch at :277; `sharding.epoch_indices`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syntalker_amd import checkpoint, config, synth, training          # noqa: E402


def synthetic_batch(args, B, seed, dev):
    h3d = config.is_h3d(args)
    y = synth.synth_clip_inputs(B, seed=seed, mask_batch=B, **({"style_dim": 256, "style_zero": False} if h3d else {}))
    y["audio"] = torch.randn(B, 68266, 2, generator=torch.Generator().manual_seed(seed + 17))          # the training clip length
    return synth.synth_latent(B, seed=seed, name="x0").to(dev), synth.to_device(y, dev)


def batches_from(npz, args, B, dev, rank=0, world=1, epoch=0, seed=0):
    """One epoch of THIS rank's batches (train.py:54-61: DataLoader(batch_size, drop_last=True) over DistributedSampler(train_data) under DDP,
    shuffle=True otherwise; the sampler is re-seeded per epoch, train.py:277): `sharding.epoch_batches` deals the epoch's permutation
    round-robin over the ranks, so an N-rank run trains on N disjoint batches per step - a global batch of N x B."""
    from syntalker_amd.sharding import epoch_batches
    z = dict(np.load(npz))                           # every array read once (an NpzFile re-reads the member on every access)
    n = z["latent"].shape[0]
    for idx in epoch_batches(n, B, rank, world, epoch, seed):
        ix = idx.numpy()
        y = {"audio": torch.from_numpy(z["audio"][ix]).float(), "word": torch.from_numpy(z["word"][ix]).long(),
             "seed": torch.from_numpy(z["seed"][ix]).float(), "mask": torch.ones(B, 1, 1, 32, dtype=torch.bool),
             "style_feature": torch.from_numpy(z["style_feature"][ix]).float() if "style_feature" in z else torch.zeros(B, 512)}
        yield torch.from_numpy(z["latent"][ix]).float().to(dev), synth.to_device(y, dev)


def main(argv=None) -> dict:
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--epochs", type=int)
    ap.add_argument("--steps-per-epoch", type=int, default=4, help="synthetic batches per epoch (ignored with --data)")
    ap.add_argument("--batch-size", type=int)
    ap.add_argument("--out", default="outputs/train_from_config")
    ap.add_argument("--resume")
    ap.add_argument("--data")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--force-ddp", action="store_true", help="wrap the model in DDP (RCCL process group) even with one rank: what a multi-GPU run uses")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--random-init", action="store_true", help="name-keyed synthetic weights instead of the modules' own initialisation")
    a = ap.parse_args(argv)
    args = config.load_args(a.config)
    epochs = a.epochs if a.epochs is not None else int(getattr(args, "epochs", 2000))
    B = a.batch_size or int(getattr(args, "batch_size", 40))
    test_period = int(getattr(args, "test_period", 20))

    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ddp = world > 1 or a.force_ddp
    if ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if a.graph:                                     # whole-step capture with the all-reduces inside: no watchdog thread on the stream
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")    # (GraphedTrainStep's protocol, training.py)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    torch.manual_seed(a.seed + rank); np.random.seed(a.seed + rank)          # the schedule sampler draws from numpy's global RNG
    t = config.build_trainer(args, dev)
    if a.random_init:
        synth.synth_fill_(t.model, 0)
    if a.resume:
        checkpoint.load_checkpoints(t.model, a.resume)
    net = t.model
    side = torch.cuda.Stream(device=dev) if (a.graph and ddp) else None
    if ddp:
        if side is not None:                            # DDP is built on the stream its captured iterations run on
            with torch.cuda.stream(side):
                net = training.make_ddp(t.model, local, capturable=True)
            torch.cuda.current_stream(dev).wait_stream(side)
        else:
            net = training.make_ddp(t.model, local)
    os.makedirs(a.out, exist_ok=True)
    watch = {n: p.detach().clone() for n, p in t.model.named_parameters() if n.split(".")[0] in ("uncon_text_embeddings", "uncon_audio_embeddings", "embed_style")}
    step_fn, log, saved = None, [], []
    for epoch in range(epochs + 1):                                           # train.py:270: range(args.epochs + 1), the last one only saves
        if epoch != epochs:
            net.train()
            it = batches_from(a.data, args, B, dev, rank, world, epoch, a.seed) if a.data else (synthetic_batch(args, B, 1000 * epoch + s + 100000 * rank, dev)
                                                                     for s in range(a.steps_per_epoch))
            t0, losses = time.time(), []
            for x0, y in it:
                if a.graph:
                    if step_fn is None:                 # (config.build_trainer's ClipAdam keeps its rate and step count on the device: capturable as it is)
                        step_fn = training.GraphedTrainStep(net, t.diffusion, t.opt, x0, {"y": y}, grad_norm=t.grad_norm, warmup=11 if ddp else 3,
                                                            stream=side)
                    losses.append(step_fn(x0, t.schedule_sampler.sample(B, dev)[0], {"y": y}))
                else:
                    losses.append(training.train_step(net, t.diffusion, t.schedule_sampler, t.opt, x0, {"y": y}, grad_norm=t.grad_norm))
            torch.cuda.synchronize()
            lr = config.step_lr(args, epoch)                                  # diffusion_rvqvae_trainer.py:571: self.opt_s.step(epoch)
            for g in t.opt.param_groups:
                if torch.is_tensor(g["lr"]):
                    g["lr"].fill_(lr)                                         # (capturable Adam keeps lr on the device: the graph reads it)
                else:
                    g["lr"] = lr
            log.append({"epoch": epoch, "loss": float(torch.stack([l.float() for l in losses]).mean()), "lr": lr, "steps": len(losses),
                        "seconds": round(time.time() - t0, 3)})
            if rank == 0:
                print(json.dumps(log[-1]), flush=True)
        if epoch % test_period == 0 and epoch != 0 and rank == 0:             # train.py:283-286
            path = os.path.join(a.out, f"last_{epoch}.bin")
            checkpoint.save_checkpoints(path, net)
            saved.append(path)
    if step_fn is not None:
        step_fn.close()
    if ddp:
        torch.distributed.destroy_process_group()
    moved = {n: bool((p.detach() != watch[n]).any()) for n, p in t.model.named_parameters() if n in watch}      # which of the rarely-read parameters trained
    return {"log": log, "saved": saved, "model": type(t.model).__module__ + "." + type(t.model).__name__, "moved": moved,
            "frozen": sorted(n for n, p in t.model.named_parameters() if not p.requires_grad), "ddp": ddp, "graph": bool(a.graph)}


if __name__ == "__main__":
    rep = main()
    if int(os.environ.get("RANK", 0)) == 0:
        print("REPORT " + json.dumps({k: rep[k] for k in ("model", "moved", "frozen", "ddp", "graph", "saved")} | {"epochs": len(rep["log"])}), flush=True)
