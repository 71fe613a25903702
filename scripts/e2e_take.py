"""End to end on the device, the reference's test.py scenario with random-init weights: seconds of speech in (audio
samples, word ids, seed latents) -> window-by-window sampling -> RVQ-VAE decoding of the whole take -> body-part poses and
root translation.  Prints the time of each stage.  Usage: python scripts/e2e_take.py [seconds=60] [takes=1]"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import config, longform, rvqvae, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(secs * 30)
s = config.build_sampler(config.load_args())                      # random-init: no checkpoint paths in the default args
synth.synth_fill_(s.model, 0)
for part, m in s.vq.items():
    m.load_state_dict(synth.synth_vq_state_dict(m.input_width, seed=11))
g = torch.Generator().manual_seed(0)
audio = torch.randn(B, n * 533, 2, generator=g).cuda()
word = torch.randint(0, synth.VOCAB, (B, n), generator=g).cuda()
seed = torch.randn(B, n // 4, 1536, generator=g).cuda()
_, rounds, _ = longform.window_plan(n)
print(f"{secs:.0f} s of speech = {n} pose frames, {rounds} windows, {B} take(s)")
from syntalker_amd.process import create_gaussian_diffusion
for name, ddim in (("DDIM-50", True), ("DDPM-1000", False)):
    d = create_gaussian_diffusion(use_ddim=ddim)
    lat = longform.sample_long(d, s.model, audio[:, :240 * 533], word[:, :240], seed[:, :60], 240, use_ddim=ddim, seed=1)
    longform.decode_take(lat, s.vq["upper"], s.vq["hands"], s.vq["lower"], s.latent_scale)          # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lat = longform.sample_long(d, s.model, audio, word, seed, n, use_ddim=ddim, seed=1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = longform.decode_take(lat, s.vq["upper"], s.vq["hands"], s.vq["lower"], s.latent_scale)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    assert all(torch.isfinite(v).all() for v in out.values())
    print(f"  {name}: sampling {t1 - t0:.3f} s + decoding {(t2 - t1) * 1e3:.2f} ms -> upper {tuple(out['upper'].shape)}, hands "
          f"{tuple(out['hands'].shape)}, lower {tuple(out['lower'].shape)}, trans {tuple(out['trans'].shape)}; "
          f"{secs * B / (t2 - t0):.1f}x real time")
