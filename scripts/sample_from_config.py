#!/usr/bin/env python3
"""The reference's `test.py -c <yaml>` scenario on this build (diffusion_rvqvae_trainer.py:577-732):
YAML -> args namespace -> denoiser + diffusion + the three body-part RVQ-VAEs (+ translation statistics) ->
window-by-window sampling of whole takes -> RVQ-VAE decoding -> poses / root translation, all on the device -> optional
Fréchet statistic of the sampled latents against reference statistics.

    python scripts/sample_from_config.py configs.yaml [--seconds 8] [--takes 4] [--ddim] [--seed 1]
                                         [--random-init] [--inputs in.npz] [--ref-stats ref.npz] [--out out.npz]

--random-init   ignore the checkpoint / statistics paths of the YAML and use the name-keyed synthetic weights (there is no
                network for the reference's checkpoints here); without it every configured path must exist.
--inputs        npz with audio (B, n*533, 2), word (B, n), seed (B, n/4, 1536); default: synthetic inputs.
--ref-stats     npz with mu / sigma of `metrics.latent_embedding` (dim given by its shape) of reference samples.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syntalker_amd import config, longform, metrics, synth          # noqa: E402
from syntalker_amd.process import create_gaussian_diffusion         # noqa: E402

PATH_KEYS = ("test_ckpt", "vqvae_upper_path", "vqvae_hands_path", "vqvae_lower_path", "vqvae_lower_trans_path",
             "mean_trans_path", "std_trans_path")


def main(argv=None) -> dict:
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--takes", type=int, default=1)
    ap.add_argument("--ddim", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--random-init", action="store_true")
    ap.add_argument("--inputs")
    ap.add_argument("--ref-stats")
    ap.add_argument("--out")
    a = ap.parse_args(argv)
    over = {k: None for k in PATH_KEYS} if a.random_init else {}
    args = config.load_args(a.config, **over)
    s = config.build_sampler(args)
    if a.random_init:
        synth.synth_fill_(s.model, 0)
        for m in s.vq.values():
            m.load_state_dict(synth.synth_vq_state_dict(m.input_width, seed=11))
    pose_length, pre, squeeze = int(getattr(args, "pose_length", 128)), int(getattr(args, "pre_frames", 4)), int(getattr(args, "vqvae_squeeze_scale", 4))
    dev = next(s.model.parameters()).device
    if a.inputs:
        z = np.load(a.inputs)
        audio, word, seed_lat = (torch.from_numpy(z[k]).to(dev) for k in ("audio", "word", "seed"))
    else:
        n, B = int(a.seconds * int(getattr(args, "pose_fps", 30))), a.takes
        g = torch.Generator().manual_seed(a.seed)
        audio = torch.randn(B, n * longform.AUDIO_PER_POSE, 2, generator=g).to(dev)
        word = torch.randint(0, synth.VOCAB, (B, n), generator=g).to(dev)
        seed_lat = torch.randn(B, n // squeeze, 1536, generator=g).to(dev)
    n = word.shape[1]
    _, rounds, _ = longform.window_plan(n, pose_length, pre, squeeze)
    if rounds < 1:
        raise SystemExit(f"{n} pose frames are less than one window of {pose_length}")
    diffusion = create_gaussian_diffusion(use_ddim=a.ddim)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lat = longform.sample_long(diffusion, s.model, audio, word, seed_lat, n, pose_length=pose_length, pre_frames=pre, squeeze=squeeze,
                               use_ddim=a.ddim, seed=a.seed)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = longform.decode_take(lat, s.vq["upper"], s.vq["hands"], s.vq["lower"], s.latent_scale, use_trans=s.use_trans,
                               trans_mean=s.trans_mean, trans_std=s.trans_std)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    rep = {"config": a.config, "pose_frames": int(n), "windows": int(rounds), "takes": int(word.shape[0]),
           "sampler": "ddim50" if a.ddim else "ddpm1000", "sampling_s": round(t1 - t0, 4), "decoding_s": round(t2 - t1, 4),
           "latents": list(lat.shape), "poses": {k: (None if v is None else list(v.shape)) for k, v in out.items()},
           "finite": bool(all(v is None or bool(torch.isfinite(v).all()) for v in out.values()))}
    if a.ref_stats:
        z = np.load(a.ref_stats)
        emb = metrics.latent_embedding(lat.float().cpu().numpy(), dim=int(z["mu"].shape[0]))
        rep["frechet_vs_ref"] = metrics.frechet_from_stats(*metrics.gaussian_stats(emb), z["mu"], z["sigma"])
    if a.out:
        np.savez(a.out, latents=lat.float().cpu().numpy(), **{k: v.float().cpu().numpy() for k, v in out.items() if v is not None})
    print(json.dumps(rep))
    rep["_latents"] = lat
    return rep


if __name__ == "__main__":
    main()
