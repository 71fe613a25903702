"""Run-to-run determinism of whole sampling loops at the bench batch: the same seed must give the same bits (Philox noise
keyed by (seed, step, element), no atomics on the data path), different seeds different samples.
Usage: python scripts/soak_determinism.py [B=1024] [repeats=3]"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from syntalker_amd import synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = synth.synth_fill_(MDM(synth.default_args()).eval(), 0).cuda()
chunks = [synth.to_device(synth.synth_clip_inputs(min(256, B - b0), seed=b0), 'cuda') for b0 in range(0, B, 256)]
y = {k: (torch.cat([c[k] for c in chunks]) if torch.is_tensor(chunks[0][k]) else chunks[0][k]) for k in chunks[0]}
for name, ddim in (("p_sample_loop (1000 steps)", False), ("ddim_sample_loop (50 steps)", True)):
    d = create_gaussian_diffusion(use_ddim=ddim)
    loop = d.ddim_sample_loop if ddim else d.p_sample_loop
    xT = torch.randn(B, 1536, 1, 32, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3))
    outs = []
    t0 = time.perf_counter()
    for r in range(R):
        outs.append(loop(m, (B, 1536, 1, 32), noise=xT.clone(), clip_denoised=False, model_kwargs={"y": y}, seed=11))
    other = loop(m, (B, 1536, 1, 32), noise=xT.clone(), clip_denoised=False, model_kwargs={"y": y}, seed=12)
    torch.cuda.synchronize()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"{name}, B = {B}: {R} runs with one seed bitwise equal: {same}; another seed differs: {not torch.equal(outs[0], other)}; "
          f"finite: {bool(torch.isfinite(outs[0]).all())}; {time.perf_counter() - t0:.1f} s", flush=True)
    assert same and torch.isfinite(outs[0]).all()
    assert ddim or not torch.equal(outs[0], other)      # DDIM (eta = 0) draws no noise: the seed does not enter
