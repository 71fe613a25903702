"""Noise schedule + posterior tables (fp64 numpy).  Oracle: test infrastructure only.

Follows reference diffusion/gaussian_diffusion.py:20-64 (cosine betas), :161-197 (tables) and
diffusion/respace.py:8-87 (timestep respacing).
"""
import math

import numpy as np


def cosine_betas(n: int, max_beta: float = 0.999) -> np.ndarray:
    # gaussian_diffusion.py:38-42,47-64
    def abar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return np.array([min(1 - abar((i + 1) / n) / abar(i / n), max_beta) for i in range(n)])


def tables(betas: np.ndarray) -> dict:
    # gaussian_diffusion.py:161-197
    betas = np.asarray(betas, dtype=np.float64)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas)
    ac_prev = np.append(1.0, ac[:-1])
    ac_next = np.append(ac[1:], 0.0)
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "alphas_cumprod_next": ac_next,
        "sqrt_alphas_cumprod": np.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": np.log(np.append(post_var[1], post_var[1:])),
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    }


def kept_steps(n: int, spec) -> list:
    # respace.py:8-61 ("ddimN" -> fixed stride; list of counts -> per-section fractional stride)
    if isinstance(spec, str):
        if spec.startswith("ddim"):
            want = int(spec[4:])
            for stride in range(1, n):
                if len(range(0, n, stride)) == want:
                    return sorted(range(0, n, stride))
            raise ValueError("no integer stride gives %d steps" % want)
        spec = [int(s) for s in spec.split(",")]
    per, extra = divmod(n, len(spec))
    out, start = [], 0
    for i, cnt in enumerate(spec):
        size = per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError("section too small")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            out.append(start + round(cur))
            cur += stride
        start += size
    return sorted(set(out))


def respaced(n: int = 1000, spec=None):
    """-> (tables over the kept steps, timestep_map).  respace.py:72-87; spec None == [n]."""
    base = tables(cosine_betas(n))
    keep = set(kept_steps(n, spec if spec is not None else [n]))
    last, new_betas, tmap = 1.0, [], []
    for i, a in enumerate(base["alphas_cumprod"]):
        if i in keep:
            new_betas.append(1 - a / last)
            last = a
            tmap.append(i)
    return tables(np.array(new_betas)), tmap
