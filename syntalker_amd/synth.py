"""Deterministic synthetic weights and inputs for the denoising hot path.

There is no network for checkpoints or datasets, so benchmarks, parity tests and the
golden-vector generator all draw weights and inputs from the same name-keyed, seed-keyed
CPU generators.  Everything here is plain CPU torch so the same bytes are produced in the
build container (where the reference is importable) and on the GPU box (where it is not).

Shapes follow the reference call sites:
  x_T / noise   (B, 1536, 1, 32)   diffusion_rvqvae_trainer.py:444
  audio         (B, 68224, 2)      diffusion_rvqvae_trainer.py:422  (16000//30 * 128 samples)
  word          (B, 128) int64     diffusion_rvqvae_trainer.py:433-442
  seed          (B, 4, 1536)       diffusion_rvqvae_trainer.py:428-431
  style_feature (B, 512) zeros for denoiser.MDM (:442) / (B, 256) for denoiser_h3d.MDM
  mask          (B, 1, 1, 32) bool diffusion_rvqvae_trainer.py:347
"""
from __future__ import annotations

import zlib
from types import SimpleNamespace

import torch

LATENT_C = 1536      # 3 body parts x 512 RVQ latent channels
LATENT_T = 32        # 128 pose frames / vqvae_squeeze_scale 4
AUDIO_LEN = 16000 // 30 * 128   # 68224, test-path length
VOCAB = 11195
WORD_DIM = 300


def default_args(**over):
    """The argparse keys MDM.__init__ reads (models/denoiser.py:36-71), yaml defaults of
    configs/diffusion_rvqvae_128.yaml."""
    a = dict(vqvae_type="rvqvae", use_motionclip=False, audio_rep="onset+amplitude",
             audio_f=256, word_f=256, data_path="", t_fix_pre=False, vqvae_squeeze_scale=4)
    a.update(over)
    return SimpleNamespace(**a)


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(name: str, shape, seed: int = 0) -> torch.Tensor | None:
    """Value for one state_dict entry, or None to leave the module's own value (deterministic
    buffers: positional table, rotary inv_freq, BatchNorm counters)."""
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf in ("pe", "inv_freq", "num_batches_tracked"):
        return None
    g = _gen(name, seed)
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "codebook":              # residual VQ: layer q quantises what layers < q left over (models/vq/residual_vq.py)
        q = int(name.split(".")[-2])
        return 0.06 * (0.6 ** q) * torch.randn(shape, generator=g)
    if name.startswith("uncon_"):
        return 0.5 * torch.randn(shape, generator=g)
    if name == "text_pre_encoder_body.weight":
        return 0.3 * torch.randn(shape, generator=g)
    if len(shape) == 1:
        is_norm_scale = leaf == "weight"
        if is_norm_scale:               # LayerNorm / BatchNorm gamma
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)   # every bias / beta
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) * (fan_in ** -0.5)


@torch.no_grad()
def synth_fill_(module_or_sd, seed: int = 0):
    """Overwrite every entry of a state_dict (or a module's) in place with synth_tensor()."""
    sd = module_or_sd if isinstance(module_or_sd, dict) else module_or_sd.state_dict()
    for k, v in sd.items():
        t = synth_tensor(k, v.shape, seed)
        if t is not None:
            v.copy_(t.to(v.dtype))
    return module_or_sd


def synth_clip_inputs(batch: int, seed: int = 0, style_dim: int = 512, style_zero: bool = True,
                      mask_batch: int | None = None):
    """model_kwargs['y'] exactly as the trainers build it, on CPU."""
    g = _gen("inputs", seed)
    y = {
        "audio": torch.randn(batch, AUDIO_LEN, 2, generator=g),
        "word": torch.randint(0, VOCAB, (batch, 128), generator=g),
        "id": torch.zeros(batch, 1, dtype=torch.long),
        "seed": torch.randn(batch, 4, LATENT_C, generator=g),
        "mask": torch.ones(mask_batch or batch, 1, 1, LATENT_T, dtype=torch.bool),
    }
    if style_zero:
        y["style_feature"] = torch.zeros(batch, style_dim)
    else:
        y["style_feature"] = torch.randn(batch, style_dim, generator=g)
    return y


def synth_latent(batch: int, seed: int = 0, name: str = "x_T") -> torch.Tensor:
    return torch.randn(batch, LATENT_C, 1, LATENT_T, generator=_gen(name, seed))


def synth_step_noise(steps: int, batch: int, seed: int = 0) -> torch.Tensor:
    """Pre-drawn per-step noise (K, B, 1536, 1, 32); row k is used by the k-th executed step."""
    return torch.randn(steps, batch, LATENT_C, 1, LATENT_T, generator=_gen("step_noise", seed))


def to_device(y: dict, device):
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in y.items()}


# ---- RVQ-VAE (the models either side of the sampling loop): synthetic weights / poses / latents ----------------------
def synth_vq_state_dict(dim: int, seed: int = 11) -> dict:
    from . import rvqvae
    sd = {k: torch.zeros_like(v) for k, v in rvqvae.build(dim).state_dict().items()}
    return synth_fill_(sd, seed)


def synth_vq_pose(part: str, dim: int, n: int = 2, t: int = 64, seed: int = 3) -> torch.Tensor:
    return synth_tensor(f"vq.pose.{part}", (n, t, dim), seed=seed) * dim ** 0.5           # N(0,1) entries


def synth_vq_rec_latent(sd: dict, part: str, n: int = 2, t: int = 16) -> torch.Tensor:
    """A latent the residual quantiser has something to say about: a sum of one code per layer (indices from a seeded
    generator) plus noise at 30 % of the last layer's scale -- what the sampler's output looks like after training."""
    from .rvqvae import NB_CODE, NUM_Q
    g = _gen(f"vq.rec.idx.{part}", 4)
    rec = 0.3 * 0.06 * 0.6 ** 5 * synth_tensor(f"vq.rec.{part}", (n, t, 512), seed=4) * (n * t * 512) ** 0.5
    for q in range(NUM_Q):
        idx = torch.randint(0, NB_CODE, (n, t), generator=g)
        rec = rec + sd[f"quantizer.layers.{q}.codebook"].cpu()[idx]
    return rec


# ---- a long take for the chunked driver (diffusion_rvqvae_trainer.py:359-541): what `_load_data` hands `_g_test` ---------------
def synth_long_take(n_pose: int, seed: int = 21) -> dict:
    """One take (batch 1, as the reference's test loader delivers it): axis-angle pose (1, n, 165), audio (1, n*533, 2),
    word ids (1, n), ground-truth latents (1, n/4, 1536) (only their first `pre_frames` rows are ever read)."""
    g = _gen("long_take", seed)
    return {"pose": 0.3 * torch.randn(1, n_pose, 165, generator=g),
            "audio": torch.randn(1, n_pose * (16000 // 30), 2, generator=g),
            "word": torch.randint(0, VOCAB, (1, n_pose), generator=g),
            "latent": torch.randn(1, n_pose // 4, LATENT_C, generator=g)}


def synth_long_noise(window: int, steps: int, seed: int = 22):
    """(x_T (1,1536,1,32), per-step noise (steps,1,1536,1,32)) of window `window` of a long take."""
    g = _gen(f"long_noise.{window}", seed)
    return torch.randn(1, LATENT_C, 1, LATENT_T, generator=g), torch.randn(steps, 1, LATENT_C, 1, LATENT_T, generator=g)


def synth_joint_masks() -> dict:
    """0/1 masks over the 165 axis-angle channels with the reference's counts per body part (13 / 30 / 9 joints,
    diffusion_rvqvae_trainer.py:52-60); which joints they are does not matter to the path under test."""
    import numpy as np
    m = {k: np.zeros(165) for k in ("upper", "hands", "lower")}
    for j in (3, 6, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21):
        m["upper"][3 * j:3 * j + 3] = 1
    m["hands"][75:165] = 1
    for j in (0, 1, 2, 4, 5, 7, 8, 10, 11):
        m["lower"][3 * j:3 * j + 3] = 1
    return m


def synth_pose_stats(seed: int = 23) -> dict:
    """(mean, std) per body part and for the root velocity: seeded stand-ins for mean_std/*.npy (trainer :187-224)."""
    g = _gen("pose_stats", seed)
    out = {}
    for name, dim in (("upper", 78), ("hands", 180), ("lower", 54), ("trans", 3)):
        out[name] = (0.2 * torch.randn(dim, generator=g), 0.5 + torch.rand(dim, generator=g))
    return out


# ---- the text-prompt (h3d) trainer's test scenario (h3d_diffusion_new_trainer.py:465-615) ---------------------------------------
H3D_PROMPTS = {"upper": "a person waves the right hand", "lower": "a person walks forward"}


def synth_prompt_vector(prompt: str) -> torch.Tensor:
    """(1, 256) stand-in for `textencoder(prompt).loc` (the TMR text encoder's mean, :489-511), seeded by the prompt's text."""
    return torch.randn(1, 256, generator=_gen("prompt:" + prompt, 25))


def synth_h3d_part_index() -> dict:
    """Disjoint index sets over the 623 HumanML3D-style pose channels with the reference's counts per body part (156 / 360 / 107)."""
    perm = torch.randperm(623, generator=_gen("h3d_parts", 26))
    return {"upper": perm[:156].sort().values, "hands": perm[156:516].sort().values, "lower": perm[516:].sort().values}


def synth_pose_clip(batch: int, n_pose: int, seed: int = 27) -> dict:
    """What the dataloader hands `_load_data` (diffusion_rvqvae_trainer.py:245-249): axis-angle poses (B, n, 165) - rotations up to ~1.5 rad,
    a few exactly zero (the small-angle branch of the conversion) - and the root velocity (B, n, 3)."""
    g = _gen("pose_clip", seed)
    pose = 0.5 * torch.randn(batch, n_pose, 165, generator=g)
    pose[:, ::7, 9:12] = 0.0
    return {"pose": pose, "trans_v": 0.05 * torch.randn(batch, n_pose, 3, generator=g)}
