"""BASELINE configs[1] literally: one 1000-step p_sample_loop call through the reference's API (conditioning, graph
capture and every host-side step included), B clips on one GPU.  Usage: python scripts/bench_full_loop.py [B=1024] [ddim=0]"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ddim = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda()
d = create_gaussian_diffusion(use_ddim=ddim)
loop = d.ddim_sample_loop if ddim else d.p_sample_loop
chunks = [synth.to_device(synth.synth_clip_inputs(min(256, B - b0), seed=b0), 'cuda') for b0 in range(0, B, 256)]
y = {k: (torch.cat([c[k] for c in chunks]) if torch.is_tensor(chunks[0][k]) else chunks[0][k]) for k in chunks[0]}
steps = 50 if ddim else 1000
for rep in range(2):                      # first call: library load, graph capture, MIOpen solver selection
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = loop(m, (B, 1536, 1, 32), clip_denoised=False, model_kwargs={"y": y}, seed=rep)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    print(f"{'ddim_sample_loop' if ddim else 'p_sample_loop'} call {rep}: B = {B}, {steps} steps: {dt:.3f} s wall = "
          f"{B * steps / dt / 1e3:.0f} k clip-steps/s end to end, {dt / steps * 1e3:.3f} ms per step", flush=True)
