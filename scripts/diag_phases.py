import sys, ctypes as C, torch, numpy as np
sys.path.insert(0, '.')
from syntalker_amd import _lib, engine, synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mt = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda(); m.m_tile = mt
pm = m.packed(); sb = m.buffers(B, 1)
sb.cond.normal_(); sb.load_x(torch.randn(B, 1536, 1, 32, device='cuda'))
coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), 'cuda')
lib = _lib.load()
nm = (B * 32 + min(mt, 64) - 1) // min(mt, 64)
dm = torch.zeros(nm * 32, dtype=torch.int64, device='cuda')
for i in range(3):
    engine.run_step(pm, sb, coef, True)
lib.syn_debug_timing.argtypes = [C.c_void_p, C.c_void_p]
lib.syn_debug_timing(None, dm.data_ptr())
engine.run_step(pm, sb, coef, True); torch.cuda.synchronize()
lib.syn_debug_timing(None, None)
t = dm.view(-1, 32).cpu().numpy().astype(np.int64)
names = ["entry->LN1(l3) [h load + layers 0-2]", "QKV head0", "attention head0", "proj-partial head0", "heads 1-3", "LN2", "MLP", "layers 4-7 + store"]
dur = np.diff(t[:, :9], axis=1)
for n, d in zip(names, np.median(dur, axis=0)):
    print(f"  {n:40s} {int(d):8d} cycles")
print("  total per WG (median):", int(np.median(t[:, 8] - t[:, 0])))
