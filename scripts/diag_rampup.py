"""Per-replay duration of the first 10-step replays after the step graph's capture (B = 1024): where does the start-up cost of a
loop sit?  Usage: python scripts/diag_rampup.py [idle_ms before the first replay]"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import engine, synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
B = 1024
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda()
pm = m.packed()
coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), 'cuda')
sb = engine.StepBuffers(B, 1, 'cuda')
sb.cond.normal_(); sb.load_x(torch.randn(B, 1536, 1, 32, device='cuda')); sb.set_rng(7, 0)
g = engine.StepGraph(pm, sb, coef, True, fused_rng=True, scheduled=True, steps=10)
ts = [999 - (i % 1000) for i in range(g.MAX_STEPS)]
g.set_schedule(ts, ts)
torch.cuda.synchronize()
time.sleep(idle / 1e3)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record(); g.replay(); b.record()
torch.cuda.synchronize()
print(f"idle {idle:.0f} ms before the first replay; ms per step of replays 1..30:", " ".join(f"{a.elapsed_time(b) / 10:.3f}" for a, b in ev), flush=True)
