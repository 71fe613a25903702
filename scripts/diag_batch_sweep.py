"""Steady-state step time over batch sizes with the library's own kernel choice: a DDPM step with in-epilogue noise, replayed as
10-step graphs (what p_sample_loop does between hooks).  Usage: python scripts/diag_batch_sweep.py [sizes...]"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import engine, synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
sizes = [int(a) for a in sys.argv[1:]] or [1, 8, 16, 32, 64, 128, 192, 256, 384, 512, 640, 768, 1024, 1100, 1280, 1536, 1792, 2048, 2560, 4096]
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda()
pm = m.packed()
coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), 'cuda')
F_STEP = 1_192_755_200
print("B: us per step, k clip-steps/s, fraction of the 2.5 PFLOP/s dense bf16 peak, kernel")
def one(B):
    sb = engine.StepBuffers(B, 1, 'cuda')
    sb.cond.normal_(); sb.load_x(torch.randn(B, 1536, 1, 32, device='cuda')); sb.set_rng(7, 0)
    g = engine.StepGraph(pm, sb, coef, True, fused_rng=True, scheduled=True, steps=10)
    ts = [999 - (i % 1000) for i in range(g.MAX_STEPS)]
    g.set_schedule(ts, ts)
    kern = "k_seq" if sb.fragment else ("k_lat" if B <= 8 else ("k_stack split x4" if B <= 64 else ("k_stack split x2" if B <= 128 else ("k_stack<32>" if B <= 256 else "k_stack<64>"))))
    return sb, g, kern


for B in sizes:
    parts = [one(hi - lo) for lo, hi in engine.plan_slices(B, 1, 'cuda')]       # (process._fused runs the slices one after the other)
    reps = max(3, min(60, int(4000 / max(B, 16))))
    for _ in range(2):
        for _, g, _ in parts: g.replay()
    for _, g, _ in parts: g.counter.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _, g, _ in parts:
        for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    for sb, _, _ in parts: sb.check_sync()
    us = e0.elapsed_time(e1) * 1e3 / (reps * 10)
    kern = " + ".join(f"{k} ({sb.B})" if len(parts) > 1 else k for sb, _, k in parts)
    print(f"B={B:5d}: {us:8.1f} us  {B / us * 1e3:8.1f} k  {B * F_STEP / (us * 1e-6) / 2.5e15:.3f}  {kern}", flush=True)
    del parts
