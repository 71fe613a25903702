"""DDPM / DDIM process restated on CPU torch.  Oracle: test infrastructure only.

Configuration is the one the reference's factory fixes (diffusion/model_util.py:8-50):
cosine schedule, 1000 steps, model predicts x_0 (START_X), FIXED_SMALL variance, MSE loss type,
no timestep rescaling; ``use_ddim`` -> "ddim50" respacing.

``model_fn(x, t_original, y)`` receives ORIGINAL timestep numbers, i.e. after the
``timestep_map`` lookup of respace.py:124-129.
Per-step noise can be injected (``step_noise[k]`` is consumed by the k-th executed step) because
GPU and CPU RNG streams differ; ``None`` draws with torch.randn_like exactly where the reference does
(gaussian_diffusion.py:541, :782).
"""
import numpy as np
import torch

from .schedule_ref import respaced


def _take(arr, t, like):
    # gaussian_diffusion.py:1606-1619: fp64 table -> gather -> .float() -> broadcast
    r = torch.from_numpy(np.asarray(arr))[t].float()
    return r.view(-1, *([1] * (like.dim() - 1)))


class RefProcess:
    def __init__(self, use_ddim: bool = False):
        self.tab, self.tmap = respaced(1000, "ddim50" if use_ddim else None)
        self.num_timesteps = len(self.tmap)
        self._tmap_t = torch.tensor(self.tmap, dtype=torch.long)

    # --- forward process -------------------------------------------------------------------
    def q_sample(self, x0, t, noise):
        # gaussian_diffusion.py:235-253
        return _take(self.tab["sqrt_alphas_cumprod"], t, x0) * x0 + \
            _take(self.tab["sqrt_one_minus_alphas_cumprod"], t, x0) * noise

    # --- reverse process -------------------------------------------------------------------
    def predict(self, model_fn, x, t, y, clip_denoised=False):
        out = model_fn(x, self._tmap_t[t], y)          # respace.py:124-129
        if "inpainting_mask" in y and "inpainted_motion" in y:   # gaussian_diffusion.py:316-320
            m = y["inpainting_mask"]
            out = out * ~m + y["inpainted_motion"] * m
        return out.clamp(-1, 1) if clip_denoised else out

    def p_sample(self, model_fn, x, t, y, noise, clip_denoised=False, const_noise=False):
        # gaussian_diffusion.py:279-397 (START_X, FIXED_SMALL) + :505-557
        x0 = self.predict(model_fn, x, t, y, clip_denoised)
        if const_noise:                                   # :543-544: the first sample's draw for every sample
            noise = noise[[0]].repeat(x.shape[0], 1, 1, 1)
        mean = _take(self.tab["posterior_mean_coef1"], t, x) * x0 + \
            _take(self.tab["posterior_mean_coef2"], t, x) * x
        logvar = _take(self.tab["posterior_log_variance_clipped"], t, x)
        nz = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
        return mean + nz * torch.exp(0.5 * logvar) * noise, x0

    def ddim_sample(self, model_fn, x, t, y, noise, eta=0.0, clip_denoised=False):
        # gaussian_diffusion.py:741-791
        x0 = self.predict(model_fn, x, t, y, clip_denoised)
        eps = (_take(self.tab["sqrt_recip_alphas_cumprod"], t, x) * x - x0) / \
            _take(self.tab["sqrt_recipm1_alphas_cumprod"], t, x)
        ab = _take(self.tab["alphas_cumprod"], t, x)
        abp = _take(self.tab["alphas_cumprod_prev"], t, x)
        sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
        mean = x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps
        nz = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
        return mean + nz * sigma * noise, x0

    def _loop(self, step, model_fn, shape, y, noise, step_noise, skip_timesteps, init_image, trace, dump_steps=None):
        # gaussian_diffusion.py:672-739 / :937-1002
        img = noise if noise is not None else torch.randn(*shape)
        if skip_timesteps and init_image is None:
            init_image = torch.zeros_like(img)
        idx = list(range(self.num_timesteps - skip_timesteps))[::-1]
        if init_image is not None:
            img = self.q_sample(init_image, torch.full((shape[0],), idx[0], dtype=torch.long), img)
        dump = []
        for k, i in enumerate(idx):
            t = torch.full((shape[0],), i, dtype=torch.long)
            eps = step_noise[k] if step_noise is not None else torch.randn_like(img)
            img, x0 = step(model_fn, img, t, y, eps)
            if trace is not None:
                trace.append((img.clone(), x0.clone()))
            if dump_steps is not None and k in dump_steps:         # :660-661: the enumeration index of the executed step, not its timestep
                dump.append(img.clone())
        return dump if dump_steps is not None else img

    @torch.no_grad()
    def p_sample_loop(self, model_fn, shape, y, noise=None, step_noise=None, skip_timesteps=0,
                      init_image=None, clip_denoised=False, trace=None, const_noise=False, dump_steps=None):
        step = lambda m, x, t, yy, e: self.p_sample(m, x, t, yy, e, clip_denoised, const_noise)
        return self._loop(step, model_fn, shape, y, noise, step_noise, skip_timesteps, init_image, trace, dump_steps)

    @torch.no_grad()
    def ddim_sample_loop(self, model_fn, shape, y, noise=None, step_noise=None, skip_timesteps=0,
                         init_image=None, eta=0.0, clip_denoised=False, trace=None):
        step = lambda m, x, t, yy, e: self.ddim_sample(m, x, t, yy, e, eta, clip_denoised)
        return self._loop(step, model_fn, shape, y, noise, step_noise, skip_timesteps, init_image, trace)

    # --- training objective -------------------------------------------------------------------
    def training_losses(self, model_fn, x0, t, y, noise):
        # gaussian_diffusion.py:1236-1363, MSE branch, all lambda_* = 0; masked_l2 (:202-215) is a
        # masked SmoothL1(beta=1) summed over (C,1,T) and divided by sum(mask) * C * 1.
        x_t = self.q_sample(x0, t, noise)
        out = model_fn(x_t, self._tmap_t[t], y)
        mask = y["mask"]
        l = torch.nn.functional.smooth_l1_loss(x0, out, reduction="none") * mask.float()
        l = l.reshape(l.shape[0], -1).sum(1)
        denom = mask.reshape(mask.shape[0], -1).sum(1) * (x0.shape[1] * x0.shape[2])
        rot = l / denom
        return {"rot_mse": rot, "loss": rot}
