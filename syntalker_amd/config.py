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

BODY_DIMS = {"upper": 78, "hands": 180, "lower": 54}          # diffusion_rvqvae_trainer.py:105,121,136 (BEAT-X, SMPL-X rot6d)
BODY_DIMS_H3D = {"upper": 156, "hands": 360, "lower": 107}    # h3d_diffusion_new_trainer.py:104,119,134 (HumanML3D-style 623-d poses)
MODEL_MODULES = ("denoiser", "denoiser_h3d")                   # what `model:` may name on this path (configs/diffusion_*.yaml:101-106)


def is_h3d(args) -> bool:
    """The text-prompt configuration (configs/diffusion_h3d.yaml: model denoiser_h3d, trainer h3d_diffusion_new, pose_rep h3d623)."""
    return getattr(args, "model", "denoiser") == "denoiser_h3d" or getattr(args, "trainer", "") == "h3d_diffusion_new"


def model_class(args):
    """train.py:85-94 / test.py:78-87: `getattr(__import__(f"models.{args.model}"), args.g_name)` - the same lookup in this package.
    Configurations of other trainers (the RVQ-VAE pre-training ones name motion_representation / VQVAEConvZero) are not this path."""
    import importlib
    name, cls = getattr(args, "model", "denoiser"), getattr(args, "g_name", "MDM")
    if name not in MODEL_MODULES:
        raise NotImplementedError(f"model: {name!r} is not a denoiser of the diffusion hot path (supported: {', '.join(MODEL_MODULES)}); "
                                  "the reference's other trainers are out of scope (SURVEY.md 2)")
    mod = importlib.import_module(f"{__package__}.{name}")
    if not hasattr(mod, cls):
        raise AttributeError(f"g_name: {cls!r} is not defined by {mod.__name__} (the reference's modules define MDM)")
    return getattr(mod, cls)


def body_dims(args) -> dict:
    return dict(BODY_DIMS_H3D if is_h3d(args) else BODY_DIMS)


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
    """The three body-part RVQ-VAEs as diffusion_rvqvae_trainer.py:87-161 (BEAT-X: 78 / 180 / 54 (+3 root velocity) channels) or
    h3d_diffusion_new_trainer.py:104-157 (156 / 360 / 107, no separate translation model) builds and loads them."""
    if getattr(args, "vqvae_type", "rvqvae") != "rvqvae":
        raise NotImplementedError("only vqvae_type == 'rvqvae' (the configuration of diffusion_rvqvae_128.yaml) is built")
    use_trans = bool(getattr(args, "use_trans", True)) and not is_h3d(args)
    out = {}
    for part, dim in body_dims(args).items():
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
    """The denoiser class comes from `args.model` / `args.g_name` (diffusion_h3d.yaml -> denoiser_h3d.MDM), the body-part widths from
    the configuration's trainer."""
    model = (model_cls or model_class(args))(args)
    ckpt = getattr(args, "test_ckpt", None)
    if ckpt:
        _require(ckpt, "test_ckpt")
        checkpoint.load_checkpoints(model, ckpt)
    diffusion = create_gaussian_diffusion()
    s = SimpleNamespace(model=model.to(device).eval(), diffusion=diffusion,
                        schedule_sampler=create_named_schedule_sampler("uniform", diffusion),
                        vq=build_vq_models(args, device), latent_scale=float(getattr(args, "vqvae_latent_scale", 5)),
                        use_trans=bool(getattr(args, "use_trans", True)) and not is_h3d(args), trans_mean=None, trans_std=None)
    for name in ("mean_trans_path", "std_trans_path"):
        p = getattr(args, name, None)
        if s.use_trans and p:
            _require(p, name)
            setattr(s, "trans_mean" if name.startswith("mean") else "trans_std", torch.from_numpy(np.load(p)).float().to(device))
    return s


# what the reference's parser falls back to when the YAML omits the key (utils/config.py:204 `--grad_norm` default 0 = no clipping,
# :208 `--lr_base` default 2.5e-4); every shipped diffusion_*.yaml sets both (0.99, 5e-5)
GRAD_NORM_DEFAULT, LR_BASE_DEFAULT = 0.0, 2.5e-4


def step_lr(args, epoch: int) -> float:
    """optimizers/timm/step_lr.py:46-51 as `create_scheduler` configures it for lr_policy 'step' (scheduler_factory.py:58-69; defaults
    of utils/config.py:207-217): lr_base * decay_rate ** (epoch // decay_epochs), no warm-up unless warmup_epochs > 0."""
    if getattr(args, "lr_policy", "step") != "step":
        raise NotImplementedError("only lr_policy == 'step' (the default every diffusion_*.yaml uses) is implemented")
    base, t = float(getattr(args, "lr_base", LR_BASE_DEFAULT)), int(getattr(args, "decay_epochs", 9999))
    w, w0 = int(getattr(args, "warmup_epochs", 0)), float(getattr(args, "warmup_lr", 5e-4))
    if epoch < w:
        return w0 + epoch * (base - w0) / w
    return base * float(getattr(args, "decay_rate", 0.1)) ** (epoch // t)


def build_trainer(args, device="cuda", model_cls=None):
    """What the trainer's constructor builds for the training loop (train.py:85-94,145-146; diffusion_rvqvae_trainer.py:60-66,185-187):
    the denoiser in train() mode, the diffusion process + uniform schedule sampler, Adam(lr_base, opt_betas) (optim_factory.py:122)."""
    model = (model_cls or model_class(args))(args).to(device).train()
    if getattr(args, "opt", "adam") != "adam":
        raise NotImplementedError("only opt == 'adam' (utils/config.py:207, every diffusion_*.yaml) is implemented")
    diffusion = create_gaussian_diffusion()
    betas = tuple(getattr(args, "opt_betas", (0.5, 0.999)))
    grad_norm, wd = float(getattr(args, "grad_norm", GRAD_NORM_DEFAULT)), float(getattr(args, "weight_decay", 0.0))
    if torch.device(device).type == "cuda":
        # clip_grad_norm_(grad_norm) + Adam as one optimizer step on the hand-written kernels; the rate is a device tensor, so the same
        # optimizer serves the eager loop and the captured step (the per-epoch scheduler writes it in place)
        from .training import ClipAdam
        opt = ClipAdam(model.parameters(), lr=torch.tensor(step_lr(args, 0), dtype=torch.float32, device=device), betas=betas, weight_decay=wd,
                       max_norm=grad_norm)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=step_lr(args, 0), betas=betas, weight_decay=wd)
    return SimpleNamespace(model=model, diffusion=diffusion, schedule_sampler=create_named_schedule_sampler("uniform", diffusion), opt=opt,
                           grad_norm=grad_norm, latent_scale=float(getattr(args, "vqvae_latent_scale", 5)))
