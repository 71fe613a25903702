"""Oracle (test infrastructure only) for the chunked long-sequence loop: CPU restatement of
diffusion_rvqvae_trainer.py:413-476 on top of oracle.process_ref / oracle.denoiser_ref.
Pinned to a run of the reference's own `_g_test`: tests/golden/make_longform_golden.py executes the reference method (lifted
from its file, on a stand-in `self`) over a 3-window take and `tests/test_longform.py::test_longform_oracle_vs_reference_g_test`
holds this restatement - window slices, seed carry, stitching, decode - to what it wrote (tests/golden/longform_outputs.npz)."""
import torch

from .process_ref import RefProcess


def sample_long_ref(model_fn, audio, word, seed_latent, n_pose, x_T, step_noise, use_ddim=False, skip_timesteps=0,
                    pose_length=128, pre_frames=4, squeeze=4, style_dim=512, ancestral=False, y_extra=None):
    """x_T: list of (B,1536,1,32) per window; step_noise: list of per-step noise tensors per window.
    ancestral: `p_sample_loop` even on the respaced (use_ddim) process - what `_g_test` calls, whatever `self.diffusion` is (:361).
    y_extra: entries that replace / extend every window's y (the text-prompt trainer's style_feature and scales,
    h3d_diffusion_new_trainer.py:553-558)."""
    overlap = pre_frames * squeeze
    round_l = pose_length - overlap                                     # :416
    roundt = (n_pose - overlap) // round_l                              # :414
    proc = RefProcess(use_ddim)
    loop = proc.ddim_sample_loop if (use_ddim and not ancestral) else proc.p_sample_loop
    bs = word.shape[0]
    out, last = [], None
    for i in range(roundt):                                              # :419
        w = word[:, i * round_l:(i + 1) * round_l + overlap]            # :420
        a = audio[:, i * (16000 // 30 * round_l):(i + 1) * (16000 // 30 * round_l) + 16000 // 30 * overlap]   # :422
        s = seed_latent[:, i * round_l // squeeze:(i + 1) * round_l // squeeze + pre_frames]                    # :424
        s = s[:, :pre_frames] if i == 0 else last[:, -pre_frames:]       # :428-431
        y = {"audio": a, "word": w, "seed": s, "mask": torch.ones(bs, 1, 1, pose_length, dtype=torch.bool),
             "style_feature": torch.zeros(bs, style_dim)}                # :433-442
        y.update(y_extra or {})
        sample = loop(model_fn, (bs, 1536, 1, pose_length // squeeze), y, noise=x_T[i], step_noise=step_noise[i],
                      skip_timesteps=skip_timesteps)
        last = sample[:, :, 0, :].permute(0, 2, 1)                       # :458 (batched form of squeeze().permute(1,0))
        out.append(last if i == 0 else last[:, pre_frames:])             # :468-476
    return torch.cat(out, dim=1)


def decode_take_ref(vq_sds, latents, latent_scale=5.0, use_trans=True, trans_mean=None, trans_std=None):
    """diffusion_rvqvae_trainer.py:458-500 on the RVQ-VAE restatement (oracle/rvq_ref.py): vq_sds = {"upper": sd, ...}."""
    from oracle import rvq_ref as rr
    parts = {}
    for k, name in enumerate(("upper", "hands", "lower")):
        parts[name] = rr.latent2origin(vq_sds[name], latents[..., 512 * k:512 * (k + 1)] * latent_scale)[0]
    trans = None
    if use_trans:
        v = parts["lower"][..., -3:]
        if trans_std is not None:
            v = v * trans_std + trans_mean
        trans = torch.cumsum(v, dim=-2)
        trans[..., 1] = v[..., 1]
        parts["lower"] = parts["lower"][..., :-3]
    parts["trans"] = trans
    return parts
