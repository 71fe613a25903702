"""Training-step timing (not the contract bench): B clips per GPU, fwd+bwd+Adam, 1 GPU."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import synth, training
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
from syntalker_amd.resample import create_named_schedule_sampler
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
d = create_gaussian_diffusion(); s = create_named_schedule_sampler("uniform", d)
y = synth.to_device(synth.synth_clip_inputs(B, seed=1, mask_batch=B), 'cuda')
y["audio"] = torch.randn(B, 68266, 2, device='cuda')          # training clip length (beat_sep_lower.py:678)
x0 = synth.synth_latent(B, seed=1, name="x0").cuda()
m = synth.synth_fill_(MDM(synth.default_args()).train(), 0).cuda()
opt = torch.optim.Adam(m.parameters(), lr=5e-5, betas=(0.5, 0.999), fused=True)
for _ in range(3): training.train_step(m, d, s, opt, x0, {"y": y})
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for _ in range(n): training.train_step(m, d, s, opt, x0, {"y": y})
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"train step B={B}: {dt*1e3:.1f} ms  ({B/dt:.0f} samples/s on 1 GPU)", flush=True)

# the same step captured in a hipGraph (single process)
m2 = synth.synth_fill_(MDM(synth.default_args()).train(), 0).cuda()
opt2 = torch.optim.Adam(m2.parameters(), lr=5e-5, betas=(0.5, 0.999), capturable=True, fused=True)
step = training.GraphedTrainStep(m2, d, opt2, x0, {"y": y})
for _ in range(3): step(x0, s.sample(B, x0.device)[0], {"y": y})
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): loss = step(x0, s.sample(B, x0.device)[0], {"y": y})
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"graph-replayed train step B={B}: {dt*1e3:.1f} ms  ({B/dt:.0f} samples/s on 1 GPU), loss {float(loss):.4f}", flush=True)
step.close()
