"""k_seq (wave-per-sequence step kernel) timing: graph-replayed DDPM step with in-epilogue noise at a few batch sizes,
beside the token-resident kernel; per-workgroup cycle stamps (input stage, each block, output stage)."""
import sys, ctypes as C, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import _lib, engine, synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
sizes = [int(a) for a in sys.argv[1:]] or [1024]
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda()
pm = m.packed()
coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), 'cuda')
lib = _lib.load()
lib.syn_debug_timing.argtypes = [C.c_void_p, C.c_void_p]
F_STEP = 1_192_755_200
for B in sizes:
    for mode, name in ((5, "k_seq"), (4, "k_stack")):
        sb = engine.StepBuffers(B, 1, 'cuda', layer_mode=mode)
        sb.cond.normal_(); sb.load_x(torch.randn(B, 1536, 1, 32, device='cuda')); sb.set_rng(7, 0)
        sb.t_model.fill_(500); sb.t_coef.fill_(500)
        g = engine.StepGraph(pm, sb, coef, True, fused_rng=True)
        for _ in range(5): g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n): g.replay()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        ok = bool(torch.isfinite(sb.x).all())
        print(f"B={B:5d} {name:8s} {ms*1e3:9.1f} us/step  {B/ms:9.1f} k clip-steps/s  frac {B*F_STEP/(ms*1e-3)/2.5e15:.3f}  finite={ok}", flush=True)
        if mode == 5:
            nwg = (B + 3) // 4
            dm = torch.zeros(nwg * 32, dtype=torch.int64, device='cuda')
            lib.syn_debug_timing(None, dm.data_ptr())
            engine.run_step(pm, sb, coef, True, fused_rng=True); torch.cuda.synchronize()
            lib.syn_debug_timing(None, None)
            t = dm.view(-1, 32).cpu().numpy().astype(np.int64)
            med = lambda a: int(np.median(a))
            blocks = [med(t[:, 2 + l] - t[:, 1 + l]) for l in range(8)]
            print(f"        cycles/workgroup: input {med(t[:, 1] - t[:, 0])}  blocks {blocks}  output {med(t[:, 10] - t[:, 9])}  total {med(t[:, 10] - t[:, 0])}", flush=True)
