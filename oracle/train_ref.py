"""The reference's training loop body restated on CPU torch.  Oracle: test infrastructure only.

`CustomTrainer.train` (diffusion_rvqvae_trainer.py:549-559) per batch:
    opt.zero_grad(); loss = _g_training(...) (:339-356 -> training_losses(...)["loss"].mean()); loss.backward();
    clip_grad_norm_(model.parameters(), args.grad_norm) (:555-556); opt.step()
with `opt = create_optimizer(args, model)` = torch.optim.Adam(model.parameters(), lr=lr_base, weight_decay=0, betas=opt_betas)
(optimizers/optim_factory.py:61-70, :122-123; configs/diffusion_rvqvae_128.yaml: lr_base 5e-5, grad_norm 0.99; utils/config.py:209 betas (0.5, 0.999)).
Pinned by tests/golden/train_trajectory.npz, which tests/golden/make_train_golden.py produces by executing the reference's own
`_g_training`, optimizer factory and schedule sampler.
"""
import torch

from . import denoiser_ref as dr
from .process_ref import RefProcess

BUFFERS = ("running_mean", "running_var", "num_batches_tracked", ".pe", "inv_freq")


def train_trajectory(sd, y, x0, t_steps, eps_steps, lr=5e-5, betas=(0.5, 0.999), grad_norm=0.99, variant="beatx"):
    """Run len(t_steps) training steps on the state dict `sd` (modified in place: parameters by Adam, BatchNorm buffers by their
    momentum updates) over ONE batch (x0, y) with the given timesteps / noise per step.
    Returns (losses, total gradient norms before clipping)."""
    for k, v in sd.items():
        v.requires_grad_(v.is_floating_point() and not k.endswith(BUFFERS))
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=lr, weight_decay=0.0, betas=tuple(betas))
    proc = RefProcess(False)
    losses, norms = [], []
    for t, eps in zip(t_steps, eps_steps):
        opt.zero_grad()
        nb = {}
        fn = lambda a, b, c: dr.mdm_forward(sd, a, b, c, variant=variant, train_bn=True, new_buffers=nb)
        loss = proc.training_losses(fn, x0, t, y, eps)["loss"].mean()
        loss.backward()
        live = [p for p in params if p.grad is not None]                       # (embed_style etc. never receive a gradient)
        norms.append(float(torch.nn.utils.clip_grad_norm_(live, grad_norm)) if grad_norm else float("nan"))
        opt.step()
        with torch.no_grad():
            for k, v in nb.items():                                             # the BatchNorm buffers as train() leaves them
                sd[k] = v.detach().clone() if k not in sd or sd[k].shape != v.shape else sd[k].copy_(v)
        losses.append(float(loss.detach()))
    return losses, norms
