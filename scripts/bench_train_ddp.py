"""DDP training step, eager vs graph-replayed (one process per GPU over RCCL; also runs with a single rank).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29531 scripts/bench_train_ddp.py [B=32] [steps=20]"""
import os, sys, time, torch
os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")       # whole-step capture: no watchdog thread touching the stream
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syntalker_amd import synth, training
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
from syntalker_amd.resample import create_named_schedule_sampler
rank, local, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = create_gaussian_diffusion(); s = create_named_schedule_sampler("uniform", d)
y = synth.to_device(synth.synth_clip_inputs(B, seed=1 + rank, mask_batch=B), 'cuda')
y["audio"] = torch.randn(B, 68266, 2, device='cuda')
x0 = synth.synth_latent(B, seed=1 + rank, name="x0").cuda()


import numpy as np


def run(capturable):
    np.random.seed(1234 + rank); torch.manual_seed(1234 + rank)            # the schedule sampler draws from numpy's global RNG, DropPath from torch's
    m = synth.synth_fill_(MDM(synth.default_args()).train(), 0).cuda()
    side = torch.cuda.Stream() if capturable else torch.cuda.current_stream()
    with torch.cuda.stream(side):                       # DDP built on the stream its iterations run on
        ddp = training.make_ddp(m, local, capturable=capturable)
    torch.cuda.current_stream().wait_stream(side)
    opt = torch.optim.Adam(ddp.parameters(), lr=5e-5, betas=(0.5, 0.999), capturable=capturable)
    if not capturable:
        step = lambda: training.train_step(ddp, d, s, opt, x0, {"y": y})
    else:
        g = training.GraphedTrainStep(ddp, d, opt, x0, {"y": y}, warmup=11, stream=side)
        step = lambda: g(x0, s.sample(B, x0.device)[0], {"y": y})
    for _ in range(3): loss = step()
    torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    for _ in range(n): loss = step()
    torch.cuda.synchronize(); dist.barrier(); dt = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    check = float(sum(p.detach().double().abs().sum() for p in m.parameters()))
    params = {k: v.detach().clone() for k, v in m.named_parameters()}
    note = ""
    if capturable:
        # gradients written straight into DDP's buckets (training.bind_grad_buffers): every bound parameter's .grad IS its bucket view
        alias = sum(1 for p in m.parameters() if getattr(p, "_syn_grad_buf", None) is not None and p.grad is not None
                    and p.grad.data_ptr() == p._syn_grad_buf.data_ptr())
        note = f", {g.bound} gradient buffers bound ({alias} aliased after the last step)"
        assert g.bound > 100 and alias == g.bound, (g.bound, alias)
    if rank == 0:
        print(f"{'graph-replayed' if capturable else 'eager'} DDP step, {world} rank(s) x {B} clips: {dt*1e3:.2f} ms "
              f"({world*B/dt:.0f} samples/s){note}, parameter checksum {check:.6f}, loss {float(loss):.4f}", flush=True)
    if capturable: g.close()
    return check, params


run(False)
c0, p0 = run(True)
c1, p1 = run(True)
assert c0 == c1, (c0, c1, [(k, float((p0[k] - p1[k]).abs().max())) for k in p0 if not torch.equal(p0[k], p1[k])][:12])            # two captured runs from the same seeds: the same parameters bit for bit (every reduction in a fixed order)
dist.barrier(); dist.destroy_process_group()
