"""Guided sampling (BASELINE configs 4 and 5): step time of the fused variant batch.  One guided step of
ClassifierFreeSampleModel = 2 reference evaluations (V = 2 here), of TwoClassifierFreeSampleModel_Bodypart = 9
(V = 4 unique variants here).  Usage: python scripts/diag_guidance.py [reps]"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import synth, engine
from syntalker_amd.denoiser_h3d import MDM
from syntalker_amd.process import create_gaussian_diffusion
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda()
pm = m.packed()
coef = engine.ddim_coefs(create_gaussian_diffusion(use_ddim=True).tables(), 0.0, 'cuda')
print("clips B, variants V: us per guided step, guided clip-steps/s, reference-evaluation-equivalents/s")
F_STEP = 1_192_755_200
for B, V, evals, mode in ((1, 2, 2, 0), (1, 4, 9, 0), (8, 2, 2, 0), (8, 4, 9, 0), (512, 2, 2, 4), (512, 2, 2, 0), (256, 4, 9, 4), (256, 4, 9, 0),
                          (256, 3, 3, 4), (256, 3, 3, 0), (1024, 2, 2, 4), (1024, 2, 2, 0)):
    sb = engine.StepBuffers(B, V, 'cuda', layer_mode=mode)      # 0: the library's choice, 4: the token-resident kernel pinned
    sb.cond.normal_(); sb.cfg_w.copy_(torch.tensor([[2.5, -1.5, 0, 0][:V]] * 3)); sb.load_x(torch.randn(B, 1536, 1, 32, device='cuda'))
    sb.t_model.fill_(25); sb.t_coef.fill_(25); sb.set_rng(3, 0)
    g = engine.StepGraph(pm, sb, coef, True, True)
    for _ in range(5): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = reps if B * V <= 64 else max(10, reps // 5)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize(); sb.check_sync()
    us = e0.elapsed_time(e1) * 1e3 / n
    kern = "k_seq" if sb.fragment else ("pinned k_stack" if mode == 4 else "library")
    print(f"  B={B:5d} V={V} {kern:14s}: {us:9.1f} us  {B / us * 1e6:10.0f} guided clip-steps/s  {B * evals / us * 1e6:11.0f} ref-eval/s  "
          f"frac {B * V * F_STEP / (us * 1e-6) / 2.5e15:.3f}")
