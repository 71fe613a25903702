#!/usr/bin/env python3
"""Golden vectors for a K-step TRAINING TRAJECTORY, produced by EXECUTING the reference's own training code in the
build container: `CustomTrainer._g_training` (diffusion_rvqvae_trainer.py:339-356) lifted out of the trainer file with `ast` (as
make_longform_golden.py lifts `_g_test`: compiled as it stands from /root/reference at run time, nothing of it is stored here), the reference's
`create_optimizer` (optimizers/optim_factory.py:61-70 -> :122-123 `optim.Adam(parameters, lr=lr_base, weight_decay=0, betas=opt_betas)`), its
`UniformSampler` (diffusion/resample.py:42-58, numpy's global RNG) and the loop body of `CustomTrainer.train`
(diffusion_rvqvae_trainer.py:549-559: zero_grad, `_g_training`, backward, clip_grad_norm_(grad_norm), step) - those five statements are
re-issued below because `train()` itself is wound around the data loader, the tracker and wandb.

Fixed for reproducibility: the reference `MDM` in train() mode with name-keyed synthetic weights (syntalker_amd.synth) and DropPath's
probability 0 (its only random element in this configuration); ONE data batch of 4 clips for all steps; per step k, numpy's global RNG seeded
with 100 + k in front of the schedule sampler's draw, and `th.randn_like(x_start)` inside `training_losses` (gaussian_diffusion.py:1260)
replaced by a seeded draw the tests regenerate.  Stored: OUTPUTS only - the K losses, the sampled timesteps, the total gradient norm
`clip_grad_norm_` returned at every step, and for a few named parameters the norm of (parameter after K steps - initial) and that
difference's first 4096 elements.
    python tests/golden/make_train_golden.py            (5 steps -> train_trajectory.npz)
    python tests/golden/make_train_golden.py 50         (50 steps -> train_trajectory_k50.npz: does the arithmetic drift? VERDICT r5 item 3)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
from make_golden import InjectNoise, import_reference  # noqa: E402  (also puts /root/reference on sys.path)
from make_longform_golden import lift_methods  # noqa: E402
from syntalker_amd import synth  # noqa: E402

K_STEPS, BATCH = (int(sys.argv[1]) if __name__ == "__main__" and len(sys.argv) > 1 else 5), 4
WATCH = ["mytimmblocks.0.attn.qkv.weight", "mytimmblocks.7.mlp.fc2.weight", "mytimmblocks.3.norm1.weight", "input_process2.weight",
         "output_process.poseFinal.bias", "embed_timestep.time_embed.0.weight", "WavEncoder.feat_extractor.0.conv2.weight",
         "WavEncoder.feat_extractor.0.bn1.weight", "WavEncoder.feat_extractor.5.conv1.weight", "text_encoder_body.weight"]


def trajectory_inputs(steps=None):
    """The data batch and the per-step noise, from seeds alone (the tests regenerate them)."""
    y = synth.synth_clip_inputs(BATCH, seed=41)
    x0 = synth.synth_latent(BATCH, seed=41, name="x0")                          # (B, 1536, 1, 32)
    eps = [synth.synth_latent(BATCH, seed=60 + k, name="eps") for k in range(K_STEPS if steps is None else steps)]
    return y, x0, eps


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    RefMDM, _, make_diff, _, data_path = import_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self
    from diffusion.resample import create_named_schedule_sampler
    from optimizers.optim_factory import create_optimizer
    (g_training,) = lift_methods("_g_training")

    args = synth.default_args(data_path=data_path, opt="adam", lr_base=5e-5, weight_decay=0., momentum=0.8, opt_betas=[0.5, 0.999],
                              grad_norm=0.99, batch_size=BATCH, pre_frames=4, pose_length=128)      # configs/diffusion_rvqvae_128.yaml, utils/config.py:204-219
    me = types.SimpleNamespace(args=args, joints=55)
    me.model = synth.synth_fill_(RefMDM(args).train(), seed=0)
    for mod in me.model.modules():
        if type(mod).__name__ == "DropPath":
            mod.drop_prob = 0.0
    me.diffusion = make_diff(use_ddim=False)
    me.schedule_sampler = create_named_schedule_sampler("uniform", me.diffusion)
    me.tracker = types.SimpleNamespace(update_meter=lambda *a, **k: None)
    me.opt = create_optimizer(args, me.model)
    assert type(me.opt) is torch.optim.Adam and tuple(me.opt.defaults["betas"]) == (0.5, 0.999) and me.opt.defaults["lr"] == 5e-5

    y, x0, eps = trajectory_inputs()
    latent_in = x0.squeeze(2).permute(0, 2, 1).contiguous()                     # `_g_training` permutes it back: x0 = latent_in.permute(0, 2, 1).unsqueeze(2)
    latent_in[:, :args.pre_frames] = y["seed"]                                  # ... and takes the seed from its first pre_frames rows (:346)
    x0 = latent_in.permute(0, 2, 1).unsqueeze(2).contiguous()
    data = {"tar_pose": torch.zeros(BATCH, 128, 330), "in_audio": y["audio"], "in_word": y["word"], "tar_id": y["id"], "latent_in": latent_in,
            "style_feature": y["style_feature"]}
    init = {n: p.detach().clone() for n, p in me.model.named_parameters()}
    seen_t = []
    sample = me.schedule_sampler.sample
    me.schedule_sampler.sample = lambda *a, **k: (lambda r: (seen_t.append(r[0].clone()), r)[1])(sample(*a, **k))
    losses, norms = [], []
    for k in range(K_STEPS):                                                    # diffusion_rvqvae_trainer.py:549-559
        np.random.seed(100 + k)
        me.opt.zero_grad()
        with InjectNoise([eps[k]]) as inj:
            loss = 0 + g_training(me, data, False, "train", 0)
        assert inj.k == 1
        loss.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(me.model.parameters(), args.grad_norm)))
        me.opt.step()
        losses.append(float(loss))
        print(f"step {k}: t = {seen_t[-1].tolist()}  loss = {losses[-1]:.6f}  grad norm = {norms[-1]:.4f}", flush=True)
    out = {"steps": np.int64(K_STEPS), "batch": np.int64(BATCH), "loss": np.array(losses, np.float64), "grad_norm": np.array(norms, np.float64),
           "t": torch.stack(seen_t).numpy().astype(np.int64), "x0_with_seed": x0.numpy(), "watch": np.array(WATCH)}
    params = dict(me.model.named_parameters())
    for n in WATCH:
        d = (params[n].detach() - init[n]).reshape(-1)
        out[f"delta_norm.{n}"] = np.float64(d.double().norm())
        out[f"delta_head.{n}"] = d[:4096].numpy()
        print(n, tuple(params[n].shape), "delta norm", float(out[f"delta_norm.{n}"]))
    # BatchNorm running statistics after K training forwards (momentum 0.1 each step)
    sd = me.model.state_dict()
    for n in ("WavEncoder.feat_extractor.0.bn1.running_mean", "WavEncoder.feat_extractor.5.bn2.running_var", "WavEncoder.feat_extractor.0.bn1.num_batches_tracked"):
        out[f"buffer.{n}"] = sd[n].double().numpy()
    del out["x0_with_seed"]                                                     # (regenerated by the tests: trajectory_inputs + the seed rows)
    for n, v in sd.items():                                                     # every BatchNorm running statistic of the encoder after K steps
        if "running_" in n:
            out[f"buffer.{n}"] = v.double().numpy()
    name = "train_trajectory.npz" if K_STEPS == 5 else f"train_trajectory_k{K_STEPS}.npz"
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, sum(np.asarray(v).nbytes for v in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
