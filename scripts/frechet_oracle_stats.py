#!/usr/bin/env python3
"""The oracle side of scripts/frechet_check.py, computed ONCE on host cores (the oracle is CPU arithmetic; 26 min of a GPU box's time
in rounds 1-3) and committed as Gaussian statistics: two independent oracle sample sets (noise seeds 8000 / 9000), N clips, DDIM-50,
time-averaged latents in the seeded random projection -> mu / sigma of each set + their mutual Frechet distance (the noise floor).
    python scripts/frechet_oracle_stats.py [N=2048] [dim=240]   -> tests/golden/frechet_oracle_n<N>.npz
"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import denoiser_ref as dr                      # noqa: E402
from oracle.frechet_ref import embed_latents, frechet_distance   # noqa: E402
from oracle.process_ref import RefProcess                  # noqa: E402
from syntalker_amd import synth                            # noqa: E402
from tests.refmodel import synth_state_dict               # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2048
DIM = int(sys.argv[2]) if len(sys.argv) > 2 else 240
CH = 64
sd = synth_state_dict("beatx")
fw = dr.fold_weights(sd)
te = dr.time_table(sd, fw)
ref = RefProcess(True)


def oracle_samples(noise_seed):
    out = []
    for b0 in range(0, N, CH):
        y = synth.synth_clip_inputs(CH, seed=100 + b0)
        cond = dr.clip_conditioning(sd, y, fw)
        fn = lambda x, t, yy: dr.mdm_forward_folded(sd, fw, cond, te, x, t)
        xT = torch.randn(CH, 1536, 1, 32, generator=torch.Generator().manual_seed(noise_seed + b0))
        with torch.no_grad():
            out.append(ref.ddim_sample_loop(fn, (CH, 1536, 1, 32), y, noise=xT))
        print(f"seed {noise_seed}: {b0 + CH}/{N} clips, {time.time() - t0:.0f} s", flush=True)
    return torch.cat(out)[:N].numpy()


def main():
    global t0
    t0 = time.time()
    ea, eb = embed_latents(oracle_samples(8_000), DIM), embed_latents(oracle_samples(9_000), DIM)
    out = {"n": np.int64(N), "dim": np.int64(DIM), "floor": np.float64(frechet_distance(ea, eb)),
           "mu_a": ea.mean(0), "sigma_a": np.cov(ea, rowvar=False), "mu_b": eb.mean(0), "sigma_b": np.cov(eb, rowvar=False),
           "seconds": np.float64(time.time() - t0)}
    path = os.path.join(REPO, "tests", "golden", f"frechet_oracle_n{N}.npz")
    np.savez_compressed(path, **{k: (v.astype(np.float32) if getattr(v, "ndim", 0) else v) for k, v in out.items()})
    print("floor", float(out["floor"]), "->", path)



def driver_case_stats():
    """The oracle side of tests/test_longform.py::test_sample_from_config_driver (48 one-window takes, DDIM-50, 16-d embedding), computed once on host
    cores: mu / sigma of one oracle set and its Frechet distance to a second one (the noise floor) -> tests/golden/sample_driver_oracle_stats.npz.
        python scripts/frechet_oracle_stats.py driver"""
    from syntalker_amd import longform, metrics
    B, n, dim = 48, 128, 16
    g = torch.Generator().manual_seed(5)
    audio, word = torch.randn(B, n * longform.AUDIO_PER_POSE, 2, generator=g), torch.randint(0, synth.VOCAB, (B, n), generator=g)
    seed_lat = torch.randn(B, n // 4, 1536, generator=g)
    y = longform.window_inputs(0, audio, word, seed_lat, None, 112)
    with torch.no_grad():
        cond = dr.clip_conditioning(sd, y, fw)
        fn = lambda a, b, c: dr.mdm_forward_folded(sd, fw, cond, te, a, b)
        refs = []
        for s in (11, 12):
            gg = torch.Generator().manual_seed(s)
            x = RefProcess(True).ddim_sample_loop(fn, (B, 1536, 1, 32), y, noise=torch.randn(B, 1536, 1, 32, generator=gg), step_noise=torch.zeros(50, B, 1536, 1, 32))
            refs.append(metrics.latent_embedding(x[:, :, 0, :].permute(0, 2, 1).numpy(), dim=dim))
    mu, sigma = metrics.gaussian_stats(refs[0])
    path = os.path.join(REPO, "tests", "golden", "sample_driver_oracle_stats.npz")
    np.savez_compressed(path, mu=mu, sigma=sigma, floor=np.float64(metrics.frechet_distance(refs[0], refs[1])), takes=np.int64(B), dim=np.int64(dim))
    print("floor", metrics.frechet_distance(refs[0], refs[1]), "->", path)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "driver":
        driver_case_stats()
    else:
        main()
