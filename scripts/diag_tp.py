"""Tensor-parallel mode of the whole-step kernel (33..128 sequences: a 32-row tile split over 2 / 4 workgroups of one XCD)
vs the plain mode (layer_mode 8 = never split): agreement and time per step.  Usage: python scripts/diag_tp.py [B ...]"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import engine, synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda()
pm = m.packed()
coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), 'cuda')
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for B in [int(a) for a in sys.argv[1:]] or [33, 40, 64, 65, 100, 128]:
    x = torch.randn(B, 1536, 1, 32, device='cuda', generator=torch.Generator(device='cuda').manual_seed(B))
    cond = torch.randn(B * 32, 512, device='cuda', generator=torch.Generator(device='cuda').manual_seed(B + 1))
    outs, times = {}, {}
    for mode in (8, 0):
        sb = engine.StepBuffers(B, 1, 'cuda', layer_mode=mode)
        sb.cond.copy_(cond); sb.load_x(x); sb.t_model.fill_(500); sb.t_coef.fill_(500); sb.set_rng(7, 0)
        for _ in range(3):
            engine.run_step(pm, sb, coef, True, fused_rng=True)
        sb.check_sync()
        outs[mode] = sb.read(sb.x).clone()
        g = engine.StepGraph(pm, sb, coef, True, fused_rng=True)
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); times[mode] = (time.perf_counter() - t0) / 50
        sb.check_sync()
    again = engine.StepBuffers(B, 1, 'cuda', layer_mode=0)
    again.cond.copy_(cond); again.load_x(x); again.t_model.fill_(500); again.t_coef.fill_(500); again.set_rng(7, 0)
    for _ in range(3):
        engine.run_step(pm, again, coef, True, fused_rng=True)
    print(f"B = {B:4d}: plain {times[8]*1e6:7.1f} us, split {times[0]*1e6:7.1f} us per step ({times[8]/times[0]:.2f}x); "
          f"3 steps split vs plain rel-L2 {rel(outs[0], outs[8]):.2e}; split run twice bitwise equal: {torch.equal(again.read(again.x), outs[0])}; "
          f"finite {bool(torch.isfinite(outs[0]).all())}", flush=True)
