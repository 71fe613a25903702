"""Steady-state kernel time split of one training step (after MIOpen's solver search), via torch.profiler."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from syntalker_amd import synth, training
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
from syntalker_amd.resample import create_named_schedule_sampler
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = synth.synth_fill_(MDM(synth.default_args()).train(), 0).cuda()
d = create_gaussian_diffusion(); s = create_named_schedule_sampler("uniform", d)
opt = training.ClipAdam(m.parameters(), lr=5e-5, betas=(0.5, 0.999), max_norm=0.99)          # (what bench.py --mode train runs)
y = synth.to_device(synth.synth_clip_inputs(B, seed=1, mask_batch=B), 'cuda')
y["audio"] = torch.randn(B, 68266, 2, device='cuda')
x0 = synth.synth_latent(B, seed=1, name="x0").cuda()
for _ in range(4): training.train_step(m, d, s, opt, x0, {"y": y})
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5): training.train_step(m, d, s, opt, x0, {"y": y})
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:int(sys.argv[2]) if len(sys.argv) > 2 else 22]
tot = sum(e.device_time_total for e in prof.key_averages())
print(f"total device time per step: {tot / 5 / 1e3:.2f} ms")
for e in rows:
    print(f"{e.device_time_total / 5 / 1e3:8.3f} ms  {e.count // 5:5d} calls  {e.key[:110]}")
