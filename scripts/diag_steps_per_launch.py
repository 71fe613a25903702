"""Steps per persistent k_seq launch (hook-free stretches of p_sample_loop): step time at B clips for several chunk lengths.
Usage: python scripts/diag_steps_per_launch.py [B] [chunk lengths...]"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import engine, synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunks = [int(a) for a in sys.argv[2:]] or [1, 10, 20, 50, 100]
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda()
pm = m.packed()
coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), 'cuda')
F_STEP = 1_192_755_200
for CH in chunks:
    sb = engine.StepBuffers(B, 1, 'cuda')
    sb.cond.normal_(); sb.load_x(torch.randn(B, 1536, 1, 32, device='cuda')); sb.set_rng(7, 0)
    g = engine.StepGraph(pm, sb, coef, True, fused_rng=True, scheduled=True, steps=CH)
    ts = [999 - (i % 1000) for i in range(g.MAX_STEPS)]
    g.set_schedule(ts, ts)
    total = 1000 // CH * CH
    for _ in range(max(1, 100 // CH)): g.replay()
    g.counter.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(total // CH): g.replay()
    e1.record(); torch.cuda.synchronize(); sb.check_sync()
    us = e0.elapsed_time(e1) * 1e3 / total
    print(f"B={B} steps per launch {CH:4d}: {us:8.1f} us per step  {B / us * 1e3:8.1f} k clip-steps/s  frac {B * F_STEP / (us * 1e-6) / 2.5e15:.4f}", flush=True)
