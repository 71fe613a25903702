"""Chunked long-sequence sampling: the window loop of the reference's `_g_test`
(diffusion_rvqvae_trainer.py:413-531, same arithmetic in h3d_diffusion_new_trainer.py:514-640).

A long take of n pose frames is generated window by window: windows are `pose_length` = 128 pose frames
(32 latent frames), consecutive windows overlap by `pre_frames` latent frames (4; `pre_frames * vqvae_squeeze_scale`
= 16 pose frames), and window i > 0 is *seeded* with the last `pre_frames` latent frames the previous window
produced - a serial dependency, which is why the reference samples with batch size 1 and why the small-batch step
kernel exists.  Several independent takes can be advanced together (batch dimension).

`sample_long` is the diffusion part (latents in, latents out); `decode_take` is what the reference does with those
latents next (trainer :472-500): split per body part, scale, `latent2origin` of the three RVQ-VAEs (rvqvae.py, SURVEY
§8 f2), un-normalise, integrate the root velocity.  The SMPL-X / rotation post-processing after that stays outside.
"""
from __future__ import annotations

import torch

AUDIO_PER_POSE = 16000 // 30          # audio samples per pose frame (diffusion_rvqvae_trainer.py:422)


def window_plan(n_pose: int, pose_length: int = 128, pre_frames: int = 4, squeeze: int = 4):
    """(round_l, rounds, remain) exactly as diffusion_rvqvae_trainer.py:414-416."""
    overlap = pre_frames * squeeze
    round_l = pose_length - overlap
    rounds = (n_pose - overlap) // round_l
    remain = (n_pose - overlap) % round_l
    return round_l, rounds, remain


def window_inputs(i: int, audio, word, seed_latent, last_sample, round_l: int, pre_frames: int = 4, squeeze: int = 4,
                  style_dim: int = 512, y_extra: dict | None = None):
    """model_kwargs['y'] of window i (trainer lines 419-442).  audio (B, n*533[, 2]), word (B, n) int64,
    seed_latent (B, n/squeeze, 1536) ground-truth latents (only its first pre_frames rows are ever used),
    last_sample (B, 32, 1536) previous window's output or None.  y_extra: entries that replace / extend the window's y - the
    text-prompt trainer's `style_feature` (a tensor or the per-body-part dict) and guidance scales (h3d_diffusion_new_trainer.py:553-558)."""
    bs = word.shape[0]
    lo, hi = i * round_l, (i + 1) * round_l + pre_frames * squeeze
    y = {
        "audio": audio[:, lo * AUDIO_PER_POSE:hi * AUDIO_PER_POSE],
        "word": word[:, lo:hi],
        "seed": seed_latent[:, :pre_frames] if i == 0 else last_sample[:, -pre_frames:],
        "mask": torch.ones(bs, 1, 1, hi - lo, dtype=torch.bool, device=word.device),
        "style_feature": torch.zeros(bs, style_dim, device=word.device),
    }
    y.update(y_extra or {})
    return y


def sample_long(diffusion, model, audio, word, seed_latent, n_pose: int | None = None, *, pose_length: int = 128,
                pre_frames: int = 4, squeeze: int = 4, use_ddim: bool = False, style_dim: int = 512, noise_fn=None,
                step_noise_fn=None, seed: int | None = None, skip_timesteps: int = 0, progress: bool = False, y_extra: dict | None = None):
    """Returns latents (B, rounds*round_l/squeeze + pre_frames, 1536): window 0 whole, later windows without their
    first pre_frames rows (trainer lines 468-476).
    noise_fn(i) -> x_T of window i or None (library draws it); step_noise_fn(i) -> injected per-step noise or None;
    seed: base key of the library's counter-based generator (window i uses seed + i)."""
    n_pose = word.shape[1] if n_pose is None else n_pose
    round_l, rounds, _ = window_plan(n_pose, pose_length, pre_frames, squeeze)
    bs = word.shape[0]
    loop = diffusion.ddim_sample_loop if use_ddim else diffusion.p_sample_loop
    pieces, last = [], None
    for i in range(rounds):
        y = window_inputs(i, audio, word, seed_latent, last, round_l, pre_frames, squeeze, style_dim, y_extra)
        kw = {}
        if step_noise_fn is not None:
            kw["step_noise"] = step_noise_fn(i)
        if seed is not None:
            kw["seed"] = seed + i
        sample = loop(model, (bs, 1536, 1, pose_length // squeeze), noise=None if noise_fn is None else noise_fn(i),
                      clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=skip_timesteps, progress=progress, **kw)
        last = sample[:, :, 0, :].permute(0, 2, 1).contiguous()          # (B, 32, 1536): trainer's squeeze/permute, batched
        pieces.append(last if i == 0 else last[:, pre_frames:])
    return torch.cat(pieces, dim=1)


def decode_take(latents, vq_upper, vq_hands, vq_lower, latent_scale: float = 5.0, *, use_trans: bool = True,
                trans_mean=None, trans_std=None, pose_stats: dict | None = None):
    """latents (B, T', 1536) from `sample_long` -> dict(upper (B,4T',78), hands (B,4T',180), lower (B,4T',54), trans
    (B,4T',3) or None).  diffusion_rvqvae_trainer.py:458-500: channel thirds are upper / hands / lower latents, scaled
    by vqvae_latent_scale before decoding; with use_trans the last 3 lower channels are the root velocity
    (de-normalised, x and z integrated over time, y kept absolute); pose_norm statistics are applied if given
    (pose_stats = {"upper": (mean, std), ...})."""
    parts = {}
    for k, (name, vq) in enumerate((("upper", vq_upper), ("hands", vq_hands), ("lower", vq_lower))):
        parts[name] = vq.latent2origin(latents[..., 512 * k:512 * (k + 1)].contiguous() * latent_scale)[0]
    trans = None
    if use_trans:
        v = parts["lower"][..., -3:]
        if trans_std is not None:
            v = v * trans_std + trans_mean
        trans = torch.cumsum(v, dim=-2)
        trans[..., 1] = v[..., 1]
        parts["lower"] = parts["lower"][..., :-3]
    if pose_stats is not None:
        for name, (mean, std) in pose_stats.items():
            parts[name] = parts[name] * std + mean
    parts["trans"] = trans
    return parts
