#!/usr/bin/env python3
"""'FGD vs ref', synthetic form: Fréchet distance between N samples of the HIP
path and N samples of the CPU oracle (= the reference's arithmetic), same weights and conditioning, independent
noise, next to the oracle-vs-oracle noise floor.  DDIM-50, time-averaged latents in a seeded random projection.

    python scripts/frechet_check.py [N=256] [dim=64]        (GPU box; writes gpurun_out/frechet.json)
"""
import json
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
from syntalker_amd.denoiser import MDM                     # noqa: E402
from syntalker_amd.process import create_gaussian_diffusion  # noqa: E402
from tests.refmodel import synth_state_dict               # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
DIM = int(sys.argv[2]) if len(sys.argv) > 2 else 64
CH = 64
sd = synth_state_dict("beatx")
fw = dr.fold_weights(sd)
te = dr.time_table(sd, fw)
ref = RefProcess(True)
dev = torch.device("cuda")
model = MDM(synth.default_args()).eval()
model.load_state_dict(sd, strict=False)
model = model.to(dev)
diff = create_gaussian_diffusion(use_ddim=True)


def oracle_samples(noise_seed):
    out = []
    for b0 in range(0, N, CH):
        y = synth.synth_clip_inputs(CH, seed=100 + b0)
        cond = dr.clip_conditioning(sd, y, fw)
        fn = lambda x, t, yy: dr.mdm_forward_folded(sd, fw, cond, te, x, t)
        xT = torch.randn(CH, 1536, 1, 32, generator=torch.Generator().manual_seed(noise_seed + b0))
        with torch.no_grad():
            out.append(ref.ddim_sample_loop(fn, (CH, 1536, 1, 32), y, noise=xT))
    return torch.cat(out)[:N].numpy()


def hip_samples(noise_seed):
    out = []
    for b0 in range(0, N, CH):
        y = synth.to_device(synth.synth_clip_inputs(CH, seed=100 + b0), dev)
        xT = torch.randn(CH, 1536, 1, 32, generator=torch.Generator().manual_seed(noise_seed + b0)).to(dev)
        out.append(diff.ddim_sample_loop(model, (CH, 1536, 1, 32), noise=xT, clip_denoised=False, model_kwargs={"y": y}).cpu())
    return torch.cat(out)[:N].numpy()


t0 = time.time()
hip = hip_samples(7_000)
t_hip = time.time() - t0
t0 = time.time()
ora, orb = oracle_samples(8_000), oracle_samples(9_000)
t_cpu = time.time() - t0
e = lambda s: embed_latents(s, DIM)
res = {
    "n_samples": N, "projection_dim": DIM, "sampler": "DDIM-50 (eta=0), x_T ~ N(0,1), independent noise per set",
    "frechet_hip_vs_oracle": frechet_distance(e(hip), e(ora)),
    "frechet_hip_vs_oracle_b": frechet_distance(e(hip), e(orb)),
    "frechet_oracle_vs_oracle (noise floor)": frechet_distance(e(ora), e(orb)),
    "embedding_scale (trace of oracle covariance)": float(np.trace(np.cov(e(ora), rowvar=False))),
    "seconds_hip": round(t_hip, 1), "seconds_cpu_oracle_two_sets": round(t_cpu, 1),
}
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(REPO, "gpurun_out", "frechet.json"), "w"), indent=1)
