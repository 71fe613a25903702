"""Pose formats either side of the RVQ-VAEs: what the reference's trainer does to a batch before the diffusion
sees it - `CustomTrainer._load_data`, diffusion_rvqvae_trainer.py:244-295 - and to the sampler's output after the decoders -
the tail of `_g_test`, :503-531.

SMPL-X joint rotations arrive as axis-angle vectors (B, n, 165); the RVQ-VAEs, the diffusion's x_0 and the metrics work on the 6D
representation (first two rows of the rotation matrix).  The reference composes four helpers of utils/rotation_conversions.py per
direction; here each direction is one HIP launch (`syn_axis_angle_to_rot6d`, `syn_rot6d_to_axis_angle`, csrc/syn_pose.inc) with the same
arithmetic.  Device tensors only, like the rest of the package: a CPU tensor or a missing library raises."""
from __future__ import annotations

import torch

from . import _lib, engine


def axis_angle_to_rotation_6d(aa: torch.Tensor) -> torch.Tensor:
    """(..., 3) axis-angle -> (..., 6): rc.matrix_to_rotation_6d(rc.axis_angle_to_matrix(aa)) (utils/rotation_conversions.py:416-430,535-550)."""
    engine._require_cuda(aa, "axis-angle rotations")
    if aa.shape[-1] != 3:
        raise ValueError(f"axis_angle_to_rotation_6d: the last dimension must be 3, got {tuple(aa.shape)}")
    x = aa.detach().float().contiguous()
    out = torch.empty(*x.shape[:-1], 6, dtype=torch.float32, device=x.device)
    if x.numel() == 0:
        return out
    _lib.check(_lib.load().syn_axis_angle_to_rot6d(x.data_ptr(), x.numel() // 3, out.data_ptr(), _lib.current_stream(x.device)),
               "syn_axis_angle_to_rot6d")
    return out


def rotation_6d_to_axis_angle(d6: torch.Tensor) -> torch.Tensor:
    """(..., 6) -> (..., 3) axis-angle: rc.matrix_to_axis_angle(rc.rotation_6d_to_matrix(d6)) (utils/rotation_conversions.py:511-533,432-446)."""
    engine._require_cuda(d6, "6D rotations")
    if d6.shape[-1] != 6:
        raise ValueError(f"rotation_6d_to_axis_angle: the last dimension must be 6, got {tuple(d6.shape)}")
    x = d6.detach().float().contiguous()
    out = torch.empty(*x.shape[:-1], 3, dtype=torch.float32, device=x.device)
    if x.numel() == 0:
        return out
    _lib.check(_lib.load().syn_rot6d_to_axis_angle(x.data_ptr(), x.numel() // 6, out.data_ptr(), _lib.current_stream(x.device)),
               "syn_rot6d_to_axis_angle")
    return out


def _index(mask, device):
    """A body part's 0/1 mask over the 165 axis-angle channels (trainer :52-60) -> its channel indices on the device."""
    m = torch.as_tensor(mask)
    return (torch.where(m != 0)[0] if m.dtype is not torch.long or m.numel() == 165 else m).to(device)


def encode_take(pose165, trans_v, vq_upper, vq_hands, vq_lower, masks: dict, pose_stats: dict | None = None, trans_stats=None,
                latent_scale: float = 5.0) -> dict:
    """`_load_data` (diffusion_rvqvae_trainer.py:255-294) from the axis-angle poses to `latent_in`.
    pose165 (B, n, 165) axis-angle, n % 4 == 0; trans_v (B, n, 3) root velocity or None (use_trans False); masks {"upper", "lower"}:
    0/1 arrays over the 165 channels (the hands are channels 75:165, :262); pose_stats {"upper": (mean, std), ...} (pose_norm) or None;
    trans_stats (mean, std) or None.  -> tar_pose_upper / _hands / _lower (what the RVQ-VAEs encode), latent_in (B, n/4, 1536) = x_0 of the
    diffusion / its seed rows, tar_pose_6d (B, n, 330)."""
    engine._require_cuda(pose165, "pose")
    bs, n, c = pose165.shape
    if c != 165:
        raise ValueError(f"encode_take: poses are (B, n, 165) axis-angle channels, got {tuple(pose165.shape)}")
    d6 = axis_angle_to_rotation_6d(pose165.reshape(bs, n, 55, 3))                       # every joint once: (B, n, 55, 6)
    joints = lambda m: torch.div(_index(m, pose165.device)[::3], 3, rounding_mode="floor")
    parts = {"upper": d6[:, :, joints(masks["upper"])].reshape(bs, n, -1), "hands": d6[:, :, 25:55].reshape(bs, n, 180),
             "lower": d6[:, :, joints(masks["lower"])].reshape(bs, n, -1)}
    if pose_stats is not None:
        for k in parts:
            mean, std = pose_stats[k]
            parts[k] = (parts[k] - mean.to(d6.device)) / std.to(d6.device)
    if trans_v is not None:
        tv = trans_v if trans_stats is None else (trans_v - trans_stats[0].to(d6.device)) / trans_stats[1].to(d6.device)
        parts["lower"] = torch.cat([parts["lower"], tv], dim=-1)
    lat = [vq.map2latent(parts[k].contiguous()) for k, vq in (("upper", vq_upper), ("hands", vq_hands), ("lower", vq_lower))]
    return {"tar_pose_upper": parts["upper"], "tar_pose_hands": parts["hands"], "tar_pose_lower": parts["lower"],
            "latent_in": torch.cat(lat, dim=2) / latent_scale, "tar_pose_6d": d6.reshape(bs, n, 330)}


def assemble_pose(rec_upper, rec_hands, rec_lower, tar_pose165, masks: dict) -> torch.Tensor:
    """The tail of `_g_test` (diffusion_rvqvae_trainer.py:503-531): the decoders' de-normalised 6D outputs (B, n, 78 / 180 / >= 54) ->
    axis-angle per part -> scattered into the 165 channels (`inverse_selection_tensor`, :236-242) -> jaw (channels 66:69) from the
    target -> every joint back to 6D: rec_pose (B, n, 330).  `longform.decode_take` produces the inputs."""
    engine._require_cuda(rec_upper, "decoded pose")
    bs, n, _ = rec_upper.shape
    dev = rec_upper.device
    rec = torch.zeros(bs * n, 165, device=dev)
    for x, j, m in ((rec_upper, 13, masks["upper"]), (rec_lower[..., :54], 9, masks["lower"]), (rec_hands, 30, masks["hands"])):
        rec[:, _index(m, dev)] = rotation_6d_to_axis_angle(x.reshape(bs, n, j, 6)).reshape(bs * n, j * 3)
    rec[:, 66:69] = tar_pose165.reshape(bs * n, 165)[:, 66:69].to(dev)
    return axis_angle_to_rotation_6d(rec.reshape(bs * n, 55, 3)).reshape(bs, n, 330)
