"""WavEncoder forward: HIP implicit-GEMM convs vs the PyTorch-ROCm/MIOpen convs of the same folded weights.
Usage: python scripts/bench_wavenc.py [B=64]"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import conditioning, synth
from tests import refmodel
from syntalker_amd.denoiser import MDM
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0)
sd = {k: v.detach() for k, v in m.state_dict().items()}
blocks = conditioning.fold_wav_encoder(sd)
cu = [{k: (tuple(t.cuda() for t in v) if isinstance(v, tuple) else v) for k, v in b.items()} for b in blocks]
wav = torch.randn(B, 68224, 2, device='cuda')
enc = conditioning.HipWavEncoder(blocks, torch.device('cuda'))
def timed(fn, n):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_hip = timed(lambda: enc(wav), 10)
t_ref = timed(lambda: refmodel.wav_features(cu, wav), 3)
gf = 2 * 2.31e9 * B
print(f"B={B}: HIP {t_hip:.3f} ms ({t_hip / B * 1e3:.1f} us/clip, {gf / t_hip / 1e9:.0f} TFLOP/s)   "
      f"PyTorch-ROCm/MIOpen {t_ref:.3f} ms ({t_ref / B * 1e3:.1f} us/clip)   speed-up {t_ref / t_hip:.1f}x")
