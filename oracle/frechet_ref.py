"""Fréchet distance between two sample sets.  Oracle: test infrastructure only.

Restates reference dataloaders/data_tools.py:1615-1685 (`FIDCalculator.frechet_distance` /
`calculate_frechet_distance`, itself the pytorch-fid formula):
    d^2 = ||mu1 - mu2||^2 + Tr(C1 + C2 - 2 sqrt(C1 C2)).
The reference applies it to embeddings of the `VAESKConv` motion encoder (weights/AESKConv_240_100.bin, absent
here, as is the BEAT-X data); Pinned by tests/golden/frechet_reference.npz (the reference's two static methods executed by
tests/golden/make_frechet_golden.py).  `embed_latents` is the synthetic substitute of SURVEY.md §8(d): time-averaged
latents in a fixed, seeded random projection.
"""
import numpy as np
from scipy import linalg


def frechet_distance(a: np.ndarray, b: np.ndarray, eps: float = 1e-6) -> float:
    mu1, mu2 = np.atleast_1d(a.mean(0)), np.atleast_1d(b.mean(0))
    c1, c2 = np.atleast_2d(np.cov(a, rowvar=False)), np.atleast_2d(np.cov(b, rowvar=False))      # (:1649-1653; one-dimensional embeddings)
    covmean, _ = linalg.sqrtm(c1.dot(c2), disp=False)
    if not np.isfinite(covmean).all():
        off = np.eye(c1.shape[0]) * eps
        covmean = linalg.sqrtm((c1 + off).dot(c2 + off))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            return 1e10
        covmean = covmean.real
    d = mu1 - mu2
    return float(d.dot(d) + np.trace(c1) + np.trace(c2) - 2 * np.trace(covmean))


def embed_latents(samples, dim: int = 240, seed: int = 2021) -> np.ndarray:
    """(N, 1536, 1, 32) latents -> (N, dim): mean over the 32 frames, then a fixed Gaussian projection."""
    x = np.asarray(samples, dtype=np.float64).reshape(len(samples), 1536, -1).mean(-1)
    proj = np.random.RandomState(seed).randn(1536, dim) / np.sqrt(1536)
    return x @ proj
