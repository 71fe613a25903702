"""Two data-parallel ranks on ONE GPU (the pool's boxes have one): a real multi-rank run of the training step's gradient exchange on the HIP kernels.
Both processes use cuda:0 and meet over a gloo process group (RCCL refuses two ranks on one device; gloo stages device tensors through the host), so
the collectives themselves are not RCCL's - what IS exercised on hardware is everything around them: the model under SyncBatchNorm (train.py:90: the
batch statistics of the audio encoder's 16 BatchNorms reduced over the ranks through `SyncBnActFn`), the DDP wrapper of `training.make_ddp`, bucketed
gradient averaging, per-rank batches.  Check: the rank-averaged gradients of two half-batches = the gradients of the full batch in one process
(BatchNorm over 8 clips = SyncBatchNorm over 4 + 4; mean of two half-batch means = the full-batch mean).
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P scripts/check_ddp_two_ranks_one_gpu.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syntalker_amd import synth, training                    # noqa: E402
from syntalker_amd.denoiser import MDM                       # noqa: E402
from syntalker_amd.process import create_gaussian_diffusion  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert world == 2
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
B = 8
y = synth.synth_clip_inputs(B, seed=91, mask_batch=B)
y["audio"] = torch.randn(B, 68266, 2, generator=torch.Generator().manual_seed(92))
x0, eps = synth.synth_latent(B, seed=91, name="x0"), synth.synth_latent(B, seed=93, name="eps")
t = torch.tensor([3, 120, 250, 400, 555, 700, 850, 999])
d = create_gaussian_diffusion()


def build():
    m = synth.synth_fill_(MDM(synth.default_args()).train(), 0).to(dev)
    m.drop_path = 0.0                                        # DropPath draws per process: off, so that the two runs are comparable
    return m


def loss_of(model, lo, hi):
    yy = {k: (v[lo:hi].to(dev) if torch.is_tensor(v) else v) for k, v in y.items()}
    return d.training_losses(model, x0[lo:hi].to(dev), t[lo:hi].to(dev), model_kwargs={"y": yy}, noise=eps[lo:hi].to(dev))["loss"].mean()


lo, hi = rank * B // 2, (rank + 1) * B // 2
ddp = training.make_ddp(build(), 0, sync_bn=True)            # SyncBatchNorm + DDP (eager: gloo collectives cannot be captured)
assert sum(isinstance(mod, torch.nn.SyncBatchNorm) for mod in ddp.modules()) == 16
loss = loss_of(ddp, lo, hi)
loss.backward()
torch.cuda.synchronize()
got = {n: p.grad.detach().cpu() for n, p in ddp.module.named_parameters() if p.grad is not None}
bn = ddp.module.WavEncoder.feat_extractor[3].bn1
stats = (bn.running_mean.detach().cpu(), bn.running_var.detach().cpu())
losses = [torch.zeros(1), torch.zeros(1)]
dist.all_gather(losses, loss.detach().cpu().reshape(1))
if rank == 0:
    full = build()
    lf = loss_of(full, 0, B)
    lf.backward()
    torch.cuda.synchronize()
    worst, worst_n, n_cmp = 0.0, "", 0
    for n, p in full.named_parameters():
        if p.grad is None:
            continue
        g, w = got[n].double(), p.grad.detach().cpu().double()
        if float(w.norm()) < 1e-7:                           # conv biases in front of a batch-statistics BatchNorm: exactly zero on both sides
            assert float(g.norm()) < 1e-5, n
            continue
        e = float((g - w).norm() / w.norm())
        n_cmp += 1
        if e > worst:
            worst, worst_n = e, n
    fb = full.WavEncoder.feat_extractor[3].bn1
    e_mean = float((stats[0] - fb.running_mean.cpu()).abs().max())
    e_var = float((stats[1] - fb.running_var.cpu()).abs().max() / fb.running_var.abs().max())
    print(f"two ranks x 4 clips on one GPU (gloo): loss {float(losses[0]):.6f} / {float(losses[1]):.6f}, mean {float((losses[0] + losses[1]) / 2):.6f}; "
          f"full batch of 8: {float(lf):.6f}", flush=True)
    print(f"rank-averaged gradients vs the full-batch gradients: {n_cmp} tensors, worst rel-L2 {worst:.3e} ({worst_n}); "
          f"SyncBatchNorm running statistics vs BatchNorm over the full batch: mean {e_mean:.2e}, var {e_var:.2e}", flush=True)
    assert abs(float((losses[0] + losses[1]) / 2) - float(lf)) < 2e-3 * abs(float(lf))
    assert worst < 2e-2 and e_mean < 1e-4 and e_var < 1e-3
    print("TWO_RANK_CHECK_OK", flush=True)
dist.barrier()
dist.destroy_process_group()
