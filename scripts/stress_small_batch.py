"""Stress test of the small-batch kernel's inter-workgroup synchronisation: long graph-replayed loops, repeated,
must be bitwise reproducible (a missed release / stale L1 line / lost barrier arrival shows up as a mismatch or as
the sticky error flag).  Usage: python scripts/stress_small_batch.py [steps=400] [repeats=4]"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import synth, engine
from syntalker_amd.denoiser_h3d import MDM
from syntalker_amd.process import create_gaussian_diffusion
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda()
pm = m.packed()
coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), 'cuda')
bad = 0
for B, V in ((1, 1), (1, 2), (1, 4), (2, 1), (3, 1), (7, 1), (8, 1), (8, 2), (13, 1), (16, 2), (24, 1), (32, 1), (5, 3)):
    g = torch.Generator().manual_seed(B * 10 + V)
    cond = torch.randn(V * B * 32, 512, generator=g).cuda() * 0.5
    x0 = torch.randn(B, 1536, 1, 32, generator=g).cuda()
    ref = None
    for r in range(reps):
        sb = engine.StepBuffers(B, V, 'cuda', layer_mode=3)
        sb.cond.copy_(cond); sb.load_x(x0); sb.set_rng(99, 0)
        if V > 1: sb.cfg_w.copy_(torch.tensor([[1.5, -0.5, 0.25, -0.25][:V]] * 3) / sum([1.5, -0.5, 0.25, -0.25][:V]))
        graph = engine.StepGraph(pm, sb, coef, True, True)
        for i in range(steps):
            t = 999 - (i % 1000)
            sb.t_coef.fill_(t); sb.t_model.fill_(t)
            graph.replay()
        torch.cuda.synchronize()
        sb.check_sync()
        out = sb.x.clone()
        assert torch.isfinite(out).all(), (B, V, r)
        if ref is None: ref = out
        elif not torch.equal(out, ref):
            bad += 1
            print(f"MISMATCH B={B} V={V} repeat {r}: max abs diff {(out - ref).abs().max().item():.3e}")
    print(f"B={B} V={V}: {reps} x {steps} steps reproducible" if bad == 0 else f"B={B} V={V}: failures so far {bad}")
print("OK" if bad == 0 else f"FAILED: {bad} mismatches")
sys.exit(1 if bad else 0)
