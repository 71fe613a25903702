"""One line per library variant: k_seq step time (10-step persistent launches, graph-replayed, in-epilogue noise) and the per-stage
cycle stamps of one workgroup-median.  Usage: SYN_HIP_LIB=<variant.so> python scripts/diag_seq_quick.py [B] [tag]"""
import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syntalker_amd import _lib, engine, synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tag = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.environ.get("SYN_HIP_LIB", "default"))
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda()
pm = m.packed()
coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), 'cuda')
lib = _lib.load()
lib.syn_debug_timing.argtypes = [C.c_void_p, C.c_void_p]
sb = engine.StepBuffers(B, 1, 'cuda', layer_mode=5)
sb.cond.normal_(); sb.load_x(torch.randn(B, 1536, 1, 32, device='cuda')); sb.set_rng(7, 0)
gm = engine.StepGraph(pm, sb, coef, True, fused_rng=True, scheduled=True, steps=10)
ts = [999 - (i % 1000) for i in range(gm.MAX_STEPS)]
gm.set_schedule(ts, ts)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = []
for rep in range(3):
    for _ in range(3): gm.replay()
    gm.counter.zero_()
    e0.record()
    for _ in range(8): gm.replay()
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 80)
ok = bool(torch.isfinite(sb.x).all())
del gm
sb.t_model.fill_(500); sb.t_coef.fill_(500)
nwg = (B + 3) // 4
dm = torch.zeros(nwg * 32, dtype=torch.int64, device='cuda')
lib.syn_debug_timing(None, dm.data_ptr())
engine.run_step(pm, sb, coef, True, fused_rng=True); torch.cuda.synchronize()
lib.syn_debug_timing(None, None)
t = dm.view(-1, 32).cpu().numpy().astype(np.int64)
med = lambda a: int(np.median(a))
blocks = [med(t[:, 2 + l] - t[:, 1 + l]) for l in range(8)]
us = min(best) * 1e3
print(f"{tag:28s} {us:8.1f} us/step (runs {[round(b * 1e3, 1) for b in best]})  frac {B * 1192755200 / (us * 1e-6) / 2.5e15:.3f}  "
      f"cycles: input {med(t[:, 1] - t[:, 0])} block {int(np.median(blocks[1:]))} output {med(t[:, 10] - t[:, 9])} total {med(t[:, 10] - t[:, 0])}  finite={ok}", flush=True)
