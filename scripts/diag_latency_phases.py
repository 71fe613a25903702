"""Cycle stamps of the small-batch kernel (block 3's four phases: work vs barrier wait).  Usage: B [V]"""
import sys, ctypes as C, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import _lib, engine, synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
V = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda(); m.layer_mode = 3
pm = m.packed(); sb = m.step_buffers(B, V)
sb.cond.normal_(); sb.load_x(torch.randn(B, 1536, 1, 32, device='cuda'))
if V > 1: sb.cfg_w.fill_(1.0 / V)
coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), 'cuda')
lib = _lib.load()
dm = torch.zeros(256 * 64, dtype=torch.int64, device='cuda')
sb.set_rng(7, 0)
for i in range(3):
    engine.run_step(pm, sb, coef, True, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
lib.syn_debug_timing.argtypes = [C.c_void_p, C.c_void_p]
lib.syn_debug_timing(None, dm.data_ptr())
e0.record(); engine.run_step(pm, sb, coef, True, True); e1.record(); torch.cuda.synchronize()
lib.syn_debug_timing(None, None)
t = dm.view(-1, 64).cpu().numpy().astype(np.int64)
t = t[t[:, 0] != 0]
t = t[t[:, 6] != 0]
med = lambda a: int(np.median(a)); mx = lambda a: int(np.max(a))
tot = med(t[:, 6] - t[:, 0])
print(f"B={B} V={V}: active workgroups {t.shape[0]}, kernel {e0.elapsed_time(e1)*1e3:.1f} us, total stamps {tot} ticks -> {e0.elapsed_time(e1)*1e3/tot*1e3:.2f} ns/tick, err flag {int(sb.sync[256])}")
def ph(name, a, b, c):
    print(f"  {name:34s} work med {med(t[:, b] - t[:, a]):7d} max {mx(t[:, b] - t[:, a]):7d}   barrier wait med {med(t[:, c] - t[:, b]):6d} min {int(np.min(t[:, c] - t[:, b])):6d}")
ph("in  (x.A^T + cond + rotary)", 0, 20, 1)
print(f"  blocks 0-2                         {med(t[:, 7] - t[:, 1])}")
ph("block 3: LN1 + qkv", 7, 12, 2)
ph("block 3: attention + proj", 2, 13, 3)
ph("block 3: LN2 + fc1 + gelu", 3, 14, 4)
ph("block 3: fc2", 4, 15, 5)
print(f"  blocks 4-7                         {med(t[:, 8] - t[:, 5])}")
print(f"  out (combine + GEMM + posterior)   {med(t[:, 6] - t[:, 8])}")
print(f"  block 3 qkv detail: LN stage {med(t[:, 30] - t[:, 7])}, sync {med(t[:, 31] - t[:, 30])}, gemm+epilogue {med(t[:, 12] - t[:, 31])}")
print(f"  block 3 attn detail: attention {med(t[:, 32] - t[:, 2])}, sync {med(t[:, 33] - t[:, 32])}, proj {med(t[:, 13] - t[:, 33])}")
if t[:, 40].any(): print(f"  LN1 probe: loads landed after {med(t[:, 40] - t[:, 7])}, reduce+normalise+LDS {med(t[:, 30] - t[:, 40])}")
