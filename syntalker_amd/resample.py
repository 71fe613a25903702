"""Timestep samplers for training — drop-in for reference diffusion/resample.py:8-67 (uniform only;
the loss-second-moment resampler is never selected by the reference's trainers)."""
import numpy as np
import torch


class UniformSampler:
    """t ~ U{0..T-1} through numpy's GLOBAL RNG (so ``np.random.seed`` reproduces the reference's draws,
    resample.py:52-58); importance weights are all one."""

    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights

    def sample(self, batch_size, device):
        w = self.weights()
        p = w / np.sum(w)
        idx = np.random.choice(len(p), size=(batch_size,), p=p)
        return (torch.from_numpy(idx).long().to(device),
                torch.from_numpy(1 / (len(p) * p[idx])).float().to(device))


class LossSecondMomentResampler:
    """resample.py:124-154 (importance sampling from a history of per-timestep losses, fed by `update_with_local_losses`, resample.py:70-121, with
    an all_gather per step).  The reference's trainers hard-code 'uniform' (diffusion_rvqvae_trainer.py:186): not built."""

    def __init__(self, *a, **k):
        raise NotImplementedError("LossSecondMomentResampler (reference diffusion/resample.py:124) is never selected by the reference's trainers "
                                  "(schedule_sampler_type = 'uniform', diffusion_rvqvae_trainer.py:186); not built")


def create_named_schedule_sampler(name, diffusion):
    """resample.py:8-22."""
    if name == "uniform":
        return UniformSampler(diffusion)
    if name == "loss-second-moment":
        return LossSecondMomentResampler(diffusion)
    raise NotImplementedError(f"unknown schedule sampler: {name}")
