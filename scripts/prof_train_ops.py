"""Which PyTorch ops of one training step launch the glue kernels: self device time and launch count per aten op and input shape
(torch.profiler, CPU + device activities, 3 eager steps after warm-up).  Usage: python scripts/prof_train_ops.py [clips] [rows]"""
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
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(N): training.train_step(m, d, s, opt, x0, {"y": y})
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
print(f"{'self device us/step':>20s} {'calls/step':>10s}  op  shapes")
for e in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 70]:
    print(f"{e.self_device_time_total / N:20.1f} {e.count / N:10.1f}  {e.key[:40]:40s} {str(e.input_shapes)[:110]}")
