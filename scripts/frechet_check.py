#!/usr/bin/env python3
"""'FGD vs ref', synthetic form: Fréchet distance between N samples of the HIP
path and N samples of the CPU oracle (= the reference's arithmetic), same weights and conditioning, independent
noise, next to the oracle-vs-oracle noise floor.  DDIM-50, time-averaged latents in a seeded random projection.

    python scripts/frechet_check.py [N=256] [dim=64] [hip_batch=64]       (GPU box; writes gpurun_out/frechet.json)
When tests/golden/frechet_oracle_n<N>.npz exists (scripts/frechet_oracle_stats.py: the two oracle sets' Gaussian statistics, computed
once on host cores) and its projection dimension matches, the oracle side comes from it and only the HIP samples are drawn here.
hip_batch = clips per sampling call: 64 runs the split-tile kernel, 1024 the wave-per-sequence kernel `bench.py` times (conditioning and
x_T of clip i are the same whatever the batch: they are keyed per 64-clip group).
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
HIP_BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 64
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
    for g0 in range(0, N, HIP_BATCH):
        ys, xs = [], []
        for b0 in range(g0, min(g0 + HIP_BATCH, N), CH):                # inputs and x_T keyed per 64-clip group, as on the oracle side
            ys.append(synth.synth_clip_inputs(CH, seed=100 + b0))
            xs.append(torch.randn(CH, 1536, 1, 32, generator=torch.Generator().manual_seed(noise_seed + b0)))
        y = synth.to_device({k: torch.cat([yy[k] for yy in ys]) for k in ys[0]}, dev)
        xT = torch.cat(xs).to(dev)
        out.append(diff.ddim_sample_loop(model, tuple(xT.shape), noise=xT, clip_denoised=False, model_kwargs={"y": y}).cpu())
    return torch.cat(out)[:N].numpy()


def frechet_to_stats(a, mu, sigma):
    """frechet_distance(a, b) with b given by its Gaussian statistics (same formula, oracle/frechet_ref.py)."""
    from scipy import linalg
    mu1, c1 = a.mean(0), np.cov(a, rowvar=False)
    covmean = linalg.sqrtm(c1.dot(sigma))
    covmean = covmean.real if np.iscomplexobj(covmean) else covmean
    d = mu1 - mu
    return float(d.dot(d) + np.trace(c1) + np.trace(sigma) - 2 * np.trace(covmean))


t0 = time.time()
hip = hip_samples(7_000)
torch.cuda.synchronize()
t_hip = time.time() - t0
e = lambda s: embed_latents(s, DIM)
stats_path = os.path.join(REPO, "tests", "golden", f"frechet_oracle_n{N}.npz")
stats = np.load(stats_path) if os.path.exists(stats_path) else None
if stats is not None and int(stats["dim"]) == DIM:
    sg = {k: stats[k].astype(np.float64) for k in ("mu_a", "sigma_a", "mu_b", "sigma_b")}
    res = {
        "n_samples": N, "projection_dim": DIM, "sampler": "DDIM-50 (eta=0), x_T ~ N(0,1), independent noise per set",
        "hip_clips_per_call": HIP_BATCH,
        "frechet_hip_vs_oracle": frechet_to_stats(e(hip), sg["mu_a"], sg["sigma_a"]),
        "frechet_hip_vs_oracle_b": frechet_to_stats(e(hip), sg["mu_b"], sg["sigma_b"]),
        "frechet_oracle_vs_oracle (noise floor)": float(stats["floor"]),
        "embedding_scale (trace of oracle covariance)": float(np.trace(sg["sigma_a"])),
        "seconds_hip": round(t_hip, 1), "oracle_side": os.path.relpath(stats_path, REPO) + f" ({float(stats['seconds']):.0f} s of host cores, once)",
    }
else:
    t0 = time.time()
    ora, orb = oracle_samples(8_000), oracle_samples(9_000)
    t_cpu = time.time() - t0
    res = {
        "n_samples": N, "projection_dim": DIM, "sampler": "DDIM-50 (eta=0), x_T ~ N(0,1), independent noise per set",
        "hip_clips_per_call": HIP_BATCH,
        "frechet_hip_vs_oracle": frechet_distance(e(hip), e(ora)),
        "frechet_hip_vs_oracle_b": frechet_distance(e(hip), e(orb)),
        "frechet_oracle_vs_oracle (noise floor)": frechet_distance(e(ora), e(orb)),
        "embedding_scale (trace of oracle covariance)": float(np.trace(np.cov(e(ora), rowvar=False))),
        "seconds_hip": round(t_hip, 1), "seconds_cpu_oracle_two_sets": round(t_cpu, 1),
    }
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(REPO, "gpurun_out", "frechet.json"), "w"), indent=1)
