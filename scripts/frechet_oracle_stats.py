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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
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


t0 = time.time()
ea, eb = embed_latents(oracle_samples(8_000), DIM), embed_latents(oracle_samples(9_000), DIM)
out = {"n": np.int64(N), "dim": np.int64(DIM), "floor": np.float64(frechet_distance(ea, eb)),
       "mu_a": ea.mean(0), "sigma_a": np.cov(ea, rowvar=False), "mu_b": eb.mean(0), "sigma_b": np.cov(eb, rowvar=False),
       "seconds": np.float64(time.time() - t0)}
path = os.path.join(REPO, "tests", "golden", f"frechet_oracle_n{N}.npz")
np.savez_compressed(path, **{k: (v.astype(np.float32) if getattr(v, "ndim", 0) else v) for k, v in out.items()})
print("floor", float(out["floor"]), "->", path)
