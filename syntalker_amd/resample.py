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


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    raise NotImplementedError(f"unknown or unsupported schedule sampler: {name}")
