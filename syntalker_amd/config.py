"""Config / construction compatibility for the hot path (first part).

`load_args(path)` reads one of the reference's YAML files (configs/diffusion_rvqvae_128.yaml, ..._hf.yaml, diffusion_h3d.yaml)
into the attribute namespace its modules are constructed from (the reference goes through configargparse,
utils/config.py; only the keys are needed here).  `build_sampler(args)` is the part of the trainer's constructor that
concerns this path (diffusion_rvqvae_trainer.py:60-66 model, :87-161 RVQ-VAEs, :185-187 diffusion + schedule sampler,
:191-196 translation statistics): it returns the objects `test.py` / `demo.py` drive, loaded from the checkpoints the
config names when they exist.
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import numpy as np
import torch
import yaml

from . import checkpoint, rvqvae, synth
from .process import create_gaussian_diffusion
from .resample import create_named_schedule_sampler

BODY_DIMS = {"upper": 78, "hands": 180, "lower": 54}          # diffusion_rvqvae_trainer.py:105,121,136


def load_args(path: str | None = None, **overrides) -> SimpleNamespace:
    """YAML keys -> namespace; keys the hot path reads but a file omits get the defaults of synth.default_args()."""
    a = vars(synth.default_args()).copy()
    if path is not None:
        with open(path) as f:
            a.update(yaml.safe_load(f) or {})
    a.update(overrides)
    return SimpleNamespace(**a)


def _require(path: str, key: str):
    if not os.path.exists(path):
        raise FileNotFoundError(f"{key} = {path!r} is configured but does not exist (unset the key to run with random initialisation)")


def build_vq_models(args, device="cuda"):
    """The three body-part RVQ-VAEs as diffusion_rvqvae_trainer.py:87-161 builds and loads them."""
    if getattr(args, "vqvae_type", "rvqvae") != "rvqvae":
        raise NotImplementedError("only vqvae_type == 'rvqvae' (the configuration of diffusion_rvqvae_128.yaml) is built")
    use_trans = bool(getattr(args, "use_trans", True))
    out = {}
    for part, dim in BODY_DIMS.items():
        if part == "lower" and use_trans:
            dim += 3                                               # root velocity rides on the lower-body model (:137-139)
        m = rvqvae.build(dim)
        key = "vqvae_lower_trans_path" if (part == "lower" and use_trans) else f"vqvae_{part}_path"
        path = getattr(args, key, None)
        if path:                                                   # configured: it must exist (the reference raises too);
            _require(path, key)                                    # only an absent / empty key leaves the random initialisation
            m.load_state_dict(torch.load(path, map_location="cpu")["net"])       # :153-155
        out[part] = m.to(device)
    return out


def build_sampler(args, device="cuda", model_cls=None):
    from .denoiser import MDM
    model = (model_cls or MDM)(args)
    ckpt = getattr(args, "test_ckpt", None)
    if ckpt:
        _require(ckpt, "test_ckpt")
        checkpoint.load_checkpoints(model, ckpt)
    diffusion = create_gaussian_diffusion()
    s = SimpleNamespace(model=model.to(device).eval(), diffusion=diffusion,
                        schedule_sampler=create_named_schedule_sampler("uniform", diffusion),
                        vq=build_vq_models(args, device), latent_scale=float(getattr(args, "vqvae_latent_scale", 5)),
                        use_trans=bool(getattr(args, "use_trans", True)), trans_mean=None, trans_std=None)
    for name in ("mean_trans_path", "std_trans_path"):
        p = getattr(args, name, None)
        if s.use_trans and p:
            _require(p, name)
            setattr(s, "trans_mean" if name.startswith("mean") else "trans_std", torch.from_numpy(np.load(p)).float().to(device))
    return s
