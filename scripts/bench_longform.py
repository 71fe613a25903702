"""Long take, window by window (SURVEY §8 f3 = the reference's test.py path): seconds of audio in, latents out.
Usage: python scripts/bench_longform.py [seconds=60] [takes=1]"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import longform, synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(secs * 30)
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda()
g = torch.Generator().manual_seed(0)
audio = torch.randn(B, n * 533, 2, generator=g).cuda()
word = torch.randint(0, synth.VOCAB, (B, n), generator=g).cuda()
seed = torch.randn(B, n // 4, 1536, generator=g).cuda()
round_l, rounds, remain = longform.window_plan(n)
print(f"{secs:.0f} s of speech = {n} pose frames -> {rounds} windows of 128 (overlap 16), {B} take(s) in parallel")
for name, ddim in (("DDPM-1000", False), ("DDIM-50", True)):
    d = create_gaussian_diffusion(use_ddim=ddim)
    longform.sample_long(d, m, audio[:, :240 * 533], word[:, :240], seed[:, :60], 240, use_ddim=ddim, seed=1)   # warm-up (graph capture, MIOpen)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = longform.sample_long(d, m, audio, word, seed, n, use_ddim=ddim, seed=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = rounds * (50 if ddim else 1000)
    print(f"  {name}: {dt:.3f} s for {out.shape[1] * 4} pose frames x {B} take(s) = {secs * B / dt:.1f}x real time; "
          f"{dt / rounds * 1e3:.1f} ms per window, {dt / steps * 1e6:.0f} us per denoising step incl. conditioning and host")
