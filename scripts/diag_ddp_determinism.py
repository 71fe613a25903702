"""Two captured DDP runs (1 rank over RCCL) from the same seeds: which parameters differ, and by how much.  h3d with argv[1] == h3d."""
import os, sys, torch
os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
import torch.distributed as dist
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syntalker_amd import synth, training
from syntalker_amd.process import create_gaussian_diffusion
from syntalker_amd.resample import create_named_schedule_sampler
variant = sys.argv[1] if len(sys.argv) > 1 else "beatx"
ddp_on = (sys.argv[2] if len(sys.argv) > 2 else "ddp") == "ddp"
torch.cuda.set_device(0)
if ddp_on:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
B = int(sys.argv[3]) if len(sys.argv) > 3 else (4 if variant == "h3d" else 32)
OPT = sys.argv[4] if len(sys.argv) > 4 else "clip"
d = create_gaussian_diffusion(); s = create_named_schedule_sampler("uniform", d)
if variant == "h3d":
    from syntalker_amd.denoiser_h3d import MDM
    y = synth.to_device(synth.synth_clip_inputs(B, seed=1, mask_batch=B), 'cuda')
    y["style_feature"] = torch.randn(B, 256, device='cuda')
else:
    from syntalker_amd.denoiser import MDM
    y = synth.to_device(synth.synth_clip_inputs(B, seed=1, mask_batch=B), 'cuda')
y["audio"] = torch.randn(B, 68266, 2, device='cuda')
x0 = synth.synth_latent(B, seed=1, name="x0").cuda()


def run():
    np.random.seed(1234); torch.manual_seed(1234)
    m = synth.synth_fill_(MDM(synth.default_args()).train(), 0).cuda()
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    side = torch.cuda.Stream()
    net = m
    if ddp_on:
        with torch.cuda.stream(side):
            net = training.make_ddp(m, 0, capturable=True)
        torch.cuda.current_stream().wait_stream(side)
    opt = (training.ClipAdam(net.parameters(), lr=5e-5, betas=(0.5, 0.999), max_norm=0.99) if OPT == "clip" else
           torch.optim.Adam(net.parameters(), lr=5e-5, betas=(0.5, 0.999), capturable=True))
    g = training.GraphedTrainStep(net, d, opt, x0, {"y": y}, warmup=11 if ddp_on else 3, stream=side if ddp_on else None)
    for _ in range(5):
        loss = g(x0, s.sample(B, x0.device)[0], {"y": y})
    torch.cuda.synchronize()
    out = {n: p.detach().double().clone() for n, p in m.named_parameters()}
    moved = {n: bool((out[n].float() != before[n]).any()) for n in out}
    grads = {n: (None if p.grad is None else float(p.grad.double().abs().sum())) for n, p in m.named_parameters()}
    g.close()
    return out, moved, grads, float(loss)


a, mv, gr, la = run()
b, _, _, lb = run()
c, _, _, lc = run()
print("variant", variant, "ddp", ddp_on, "B", B, OPT, "loss", la, lb, lc)
bad3 = [(n, float((a[n] - c[n]).abs().max())) for n in a if not torch.equal(a[n], c[n])]
print("run 1 vs run 3 differ:", len(bad3), bad3[:10])
bad = [(n, float((a[n] - b[n]).abs().max())) for n in a if not torch.equal(a[n], b[n])]
print("parameters that differ between two identical runs:", len(bad))
for n, e in bad[:40]:
    print("  ", n, e)
print("not moved (other than conv biases):", [n for n, v in mv.items() if not v and "conv" not in n and "downsample" not in n])
if ddp_on:
    dist.destroy_process_group()
