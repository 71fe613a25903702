import sys, ctypes as C, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import _lib, engine, synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mt = int(sys.argv[2]) if len(sys.argv) > 2 else 64
NOISE = (sys.argv[3] if len(sys.argv) > 3 else 'rng')      # rng | buf | none
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda(); m.m_tile = mt
pm = m.packed(); sb = m.step_buffers(B, 1)
sb.cond.normal_(); sb.load_x(torch.randn(B, 1536, 1, 32, device='cuda'))
coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), 'cuda')
lib = _lib.load()
nm = 256 if mt == 0 else (B * 32 + min(mt, 64) - 1) // min(mt, 64)      # mt = 0: the library's choice (tile split: <= 256 workgroups)
dm = torch.zeros(nm * 32, dtype=torch.int64, device='cuda')
sb.set_rng(7, 0)
kw = dict(use_noise=NOISE != 'none', fused_rng=NOISE == 'rng')
for i in range(3):
    engine.run_step(pm, sb, coef, **kw)
lib.syn_debug_timing.argtypes = [C.c_void_p, C.c_void_p]
lib.syn_debug_timing(None, dm.data_ptr())
engine.run_step(pm, sb, coef, **kw); torch.cuda.synchronize()
lib.syn_debug_timing(None, None)
t = dm.view(-1, 32).cpu().numpy().astype(np.int64)
t = t[t[:, 8] > 0]                                  # workgroups that ran to the end (padding groups of a split launch leave early)
med = lambda a: int(np.median(a))
print(f"B={B} workgroups {t.shape[0]}")
print("  input stage (x.A^T + cond + rotary)      ", med(t[:, 9] - t[:, 0]))
print("  blocks 0-2 (+LN1 of block 3)             ", med(t[:, 1] - t[:, 9]))
print("  block 3: QKV head0                       ", med(t[:, 2] - t[:, 1]))
print("  block 3: attention head0                 ", med(t[:, 3] - t[:, 2]))
print("  block 3: proj-partial head0              ", med(t[:, 4] - t[:, 3]))
print("  block 3: heads 1-3                       ", med(t[:, 5] - t[:, 4]))
print("  block 3: LN2                             ", med(t[:, 6] - t[:, 5]))
print("  block 3: MLP                             ", med(t[:, 7] - t[:, 6]))
print("  blocks 4-7                               ", med(t[:, 10] - t[:, 7]))
print("  output stage (3 x 512 cols + posterior)  ", med(t[:, 8] - t[:, 10]))
print("  total per workgroup                      ", med(t[:, 8] - t[:, 0]))
print("  output stage detail: to-LDS+gemm0, epi0, gemm1, epi1, gemm2, epi2:", [med(t[:, 11] - t[:, 10]), med(t[:, 12] - t[:, 11]), med(t[:, 13] - t[:, 12]), med(t[:, 14] - t[:, 13]), med(t[:, 15] - t[:, 14]), med(t[:, 16] - t[:, 15])])
print("  input stage detail: cond/te + first x_t piece in LDS, gemm0, gemm1, gemm2, rotary:", [med(t[:, 17] - t[:, 0]), med(t[:, 18] - t[:, 17]), med(t[:, 19] - t[:, 18]), med(t[:, 20] - t[:, 19]), med(t[:, 9] - t[:, 20])])
if t[:, 24].max() > 0:
    print("  tile-split exchange after the attention of block 3: stores + drain, barrier, reads + sums:",
          [med(t[:, 22] - t[:, 21]), med(t[:, 23] - t[:, 22]), med(t[:, 24] - t[:, 23])], " barrier wait min / max:",
          int((t[:, 23] - t[:, 22]).min()), int((t[:, 23] - t[:, 22]).max()))
