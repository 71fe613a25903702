"""Fréchet distance between Gaussian fits of two embedding sets.

The reference's FGD (dataloaders/data_tools.py:1615-1685, `FIDCalculator.frechet_distance` / `calculate_frechet_distance`)
is the pytorch-fid formula  d^2 = |mu1 - mu2|^2 + Tr(C1 + C2 - 2 sqrt(C1 C2))  over embeddings of its `VAESKConv` motion
encoder (weights/AESKConv_240_100.bin - not distributed with the repository).  This module is the formula, host side
(numpy / scipy like the reference's), over whatever embedding the caller has: the evaluator's when its checkpoint is
available, or `latent_embedding` - time-averaged sampler latents in a fixed seeded projection - as the stand-in.
"""
from __future__ import annotations

import numpy as np


def gaussian_stats(emb) -> tuple[np.ndarray, np.ndarray]:
    e = np.asarray(emb, dtype=np.float64)
    return e.mean(0), np.cov(e, rowvar=False)


def frechet_from_stats(mu1, c1, mu2, c2, eps: float = 1e-6) -> float:
    from scipy import linalg
    mu1, mu2, c1, c2 = (np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (mu1, mu2, c1, c2))
    c1, c2 = np.atleast_2d(c1), np.atleast_2d(c2)
    covmean, _ = linalg.sqrtm(c1.dot(c2), disp=False)
    if not np.isfinite(covmean).all():                     # singular product: the reference's epsilon retry (:1668-1672)
        off = np.eye(c1.shape[0]) * eps
        covmean = linalg.sqrtm((c1 + off).dot(c2 + off))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):       # :1675-1678 raises ValueError, which `frechet_distance` (:1615-1626) turns into 1e10
            return 1e10
        covmean = covmean.real
    d = mu1 - mu2
    return float(d.dot(d) + np.trace(c1) + np.trace(c2) - 2.0 * np.trace(covmean))


def frechet_distance(emb_a, emb_b) -> float:
    return frechet_from_stats(*gaussian_stats(emb_a), *gaussian_stats(emb_b))


def latent_embedding(latents, dim: int = 240, seed: int = 2021) -> np.ndarray:
    """(N, T, 1536) sampler latents -> (N, dim): mean over time, fixed Gaussian projection (stand-in for the motion encoder)."""
    x = np.asarray(latents, dtype=np.float64)
    x = x.reshape(x.shape[0], -1, x.shape[-1]).mean(1)
    proj = np.random.RandomState(seed).randn(x.shape[-1], dim) / np.sqrt(x.shape[-1])
    return x @ proj
