"""Repeat the same training forward + backward (fixed t, noise, DropPath off) N times: parameters whose gradient is not bit-identical every time."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syntalker_amd import synth
from syntalker_amd.denoiser import MDM
from syntalker_amd.process import create_gaussian_diffusion
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
d = create_gaussian_diffusion()
for B in (4, 32):
    m = synth.synth_fill_(MDM(synth.default_args()).train(), 0).cuda()
    m.drop_path = 0.0
    y = synth.to_device(synth.synth_clip_inputs(B, seed=2, mask_batch=B), 'cuda')
    x0, eps = synth.synth_latent(B, seed=1, name="x0").cuda(), synth.synth_latent(B, seed=3, name="eps").cuda()
    t = (torch.arange(B, device='cuda') * 31) % 1000
    ref, bad = None, {}
    for it in range(N):
        m.zero_grad(set_to_none=True)
        d.training_losses(m, x0, t, model_kwargs={"y": y}, noise=eps)["loss"].mean().backward()
        g = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        if ref is None:
            ref = g
        else:
            for n in g:
                if not torch.equal(g[n], ref[n]):
                    bad[n] = bad.get(n, 0) + 1
    torch.cuda.synchronize()
    print(f"B = {B}: {N} repeats, gradients that were not bit-identical every time: {bad}")
