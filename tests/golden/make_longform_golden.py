#!/usr/bin/env python3
"""Golden vectors for the chunked long-sequence driver, produced by EXECUTING the reference's own
`CustomTrainer._g_test` (diffusion_rvqvae_trainer.py:359-541) in the build container.

The trainer module cannot be imported here (pynvml, smplx, wandb, librosa, clip ... are absent), so the script lifts the
two methods it needs - `_g_test` and `inverse_selection_tensor` - out of the reference file with `ast`, compiles them as
they stand (no edit, no copy into this repository: the code object is built from /root/reference at run time) and calls
`_g_test` on a stand-in `self` that carries what `CustomTrainer.__init__` would have put there:

  * `model`      the reference `models.denoiser.MDM` with the name-keyed synthetic weights of syntalker_amd.synth
  * `diffusion`  the reference `create_gaussian_diffusion(use_ddim=True)`: `_g_test` calls its `p_sample_loop`, i.e. the
                 ancestral sampler over the 50 kept timesteps (150 denoiser evaluations for 3 windows instead of 3000)
  * `vq_model_*` the reference `models.vq.model.RVQVAE` (synthetic weights, `Tensor.cuda` = identity)
  * joint masks, normalisation statistics, `args`: synthetic, seeded (syntalker_amd.synth / this file)

x_T of every window (`th.randn`, gaussian_diffusion.py:703) and every step's noise (`th.randn_like`, :541) are replaced
by seeded draws the tests regenerate.  Stored: the OUTPUTS only - per-window samples, the stitched latents handed to
`latent2origin` (trainer :476-478), its three outputs after de-normalisation (:494-496) and `rec_trans` (:485-491).
    python tests/golden/make_longform_golden.py
"""
import ast
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
from make_golden import REF, import_reference  # noqa: E402  (also puts /root/reference on sys.path)
from make_vq_golden import PARTS, vq_args  # noqa: E402
from syntalker_amd import synth  # noqa: E402

WINDOWS = 3
N_POSE = 128 + (WINDOWS - 1) * 112 + 5            # 357: `remain = n % 8` (trainer :378-388) trims the take to 352 frames


def lift_methods(*names, trainer="diffusion_rvqvae_trainer.py", extra_globals=None):
    """The reference's own method bodies, compiled from its file (no import of the trainer module)."""
    path = os.path.join(REF, trainer)
    tree = ast.parse(open(path).read(), filename=path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CustomTrainer")
    fns = [f for f in cls.body if isinstance(f, ast.FunctionDef) and f.name in names]
    assert sorted(f.name for f in fns) == sorted(names)
    from utils import rotation_conversions as rc
    ns = {"torch": torch, "np": np, "rc": rc, **(extra_globals or {})}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


class SeededDraws:
    """`th.randn(*shape, device=...)` and `th.randn_like(x)` inside the reference's loop pop the next pre-drawn tensor."""

    def __init__(self, draws):
        self.draws, self.k = draws, 0

    def _pop(self, shape):
        r = self.draws[self.k]
        self.k += 1
        assert tuple(r.shape) == tuple(shape), (r.shape, shape)
        return r.clone()

    def __enter__(self):
        self._randn, self._like = torch.randn, torch.randn_like
        torch.randn = lambda *shape, **kw: self._pop(shape[0] if len(shape) == 1 and not isinstance(shape[0], int) else shape)
        torch.randn_like = lambda x, *a, **kw: self._pop(x.shape)
        return self

    def __exit__(self, *a):
        torch.randn, torch.randn_like = self._randn, self._like


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    torch.set_grad_enabled(False)
    RefMDM, _, make_diff, _, data_path = import_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self
    from models.vq.model import RVQVAE
    g_test, inverse_selection_tensor = lift_methods("_g_test", "inverse_selection_tensor")

    me = types.SimpleNamespace()
    me.args = types.SimpleNamespace(vqvae_squeeze_scale=4, pre_frames=4, pose_length=128, pose_dims=330, batch_size=1, pose_norm=True)
    me.joints = 55
    masks = synth.synth_joint_masks()
    me.joint_mask_upper, me.joint_mask_hands, me.joint_mask_lower = masks["upper"], masks["hands"], masks["lower"]
    me.inverse_selection_tensor = lambda *a: inverse_selection_tensor(me, *a)
    me.model = synth.synth_fill_(RefMDM(synth.default_args(data_path=data_path)).eval(), seed=0)
    me.diffusion = make_diff(use_ddim=True)
    me.vqvae_latent_scale = 5.0
    me.use_trans = True
    stats = synth.synth_pose_stats()
    me.trans_mean, me.trans_std = stats["trans"]
    (me.mean_upper, me.std_upper), (me.mean_hands, me.std_hands), (me.mean_lower, me.std_lower) = stats["upper"], stats["hands"], stats["lower"]
    rec = {}
    for part, dim in PARTS:
        vq = synth.synth_fill_(RVQVAE(vq_args(), dim, 512, 512, 512, 2, 2, 512, 3, 3, "relu", None).eval(), seed=11)
        orig = vq.latent2origin

        def tapped(x, part=part, orig=orig):
            rec[f"{part}.latent_in"] = x.clone()
            out = orig(x)
            rec[f"{part}.latent2origin"] = out[0].clone()
            return out
        vq.latent2origin = tapped
        setattr(me, f"vq_model_{part}", vq)

    take = synth.synth_long_take(N_POSE, seed=21)
    data = {"tar_pose": take["pose"], "tar_beta": torch.zeros(1, N_POSE, 300), "tar_exps": torch.zeros(1, N_POSE, 100),
            "tar_contact": torch.zeros(1, N_POSE, 4), "tar_trans": torch.zeros(1, N_POSE, 3), "in_word": take["word"],
            "in_audio": take["audio"], "latent_in": take["latent"], "tar_id": torch.zeros(1, N_POSE, 1, dtype=torch.long)}
    K = me.diffusion.num_timesteps
    draws, samples = [], []
    for w in range(WINDOWS):
        xT, sn = synth.synth_long_noise(w, K, seed=22)
        draws += [xT] + list(sn)
    loop = me.diffusion.p_sample_loop
    me.diffusion.p_sample_loop = lambda *a, **kw: (samples.append(loop(*a, **kw)), samples[-1])[1]
    with SeededDraws(draws) as sd:
        res = g_test(me, data)
    assert sd.k == len(draws) == WINDOWS * (K + 1) and len(samples) == WINDOWS
    out = {"n_pose": np.int64(N_POSE), "steps": np.int64(K), "windows": np.int64(WINDOWS),
           "samples": torch.stack(samples).numpy(),                                               # (W, 1, 1536, 1, 32)
           "rec_trans": res["rec_trans"].numpy()}
    for part, _ in PARTS:
        out[f"{part}.latent_in"] = rec[f"{part}.latent_in"].numpy()                               # (1, 88, 512), already x latent scale
        out[f"{part}.latent2origin"] = rec[f"{part}.latent2origin"].numpy()                       # (1, 352, dim) before de-normalisation
    out["rec_pose"] = res["rec_pose"].numpy()                                                     # (1, 352, 330): the post-processing (out of scope) ran
    for k, v in out.items():
        print(k, getattr(v, "shape", v), float(np.abs(v).mean()) if getattr(v, "ndim", 0) else "")
    np.savez_compressed(os.path.join(HERE, "longform_outputs.npz"), **out)
    print("wrote longform_outputs.npz", sum(np.asarray(v).nbytes for v in out.values()) // 1024, "KiB")


def main_h3d():
    """The text-prompt trainer's `_g_test` (h3d_diffusion_new_trainer.py:465-615): DDIM-50 (hard-coded there, :468-471) through
    `TwoClassifierFreeSampleModel_Bodypart` (test(), :834) with an upper-body and a lower-body prompt - 9 denoiser evaluations per step -
    over the same 3-window take, on a stand-in `self`: reference `denoiser_h3d.MDM` + reference wrapper, a `textencoder` that returns the
    seeded (1, 256) vector of synth.synth_prompt_vector(prompt) as `.loc`, the 156 / 360 / 107-channel RVQ-VAEs, index masks over the 623 pose
    channels.  Stored: per-window samples, the stitched latents handed to `latent2origin` and `rec_pose`."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    torch.set_grad_enabled(False)
    _, RefMDMH3D, make_diff, cfgmod, data_path = import_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self
    from models.vq.model import RVQVAE
    (g_test,) = lift_methods("_g_test", trainer="h3d_diffusion_new_trainer.py", extra_globals={
        "create_gaussian_diffusion": make_diff, "ClassifierFreeSampleModel": cfgmod.ClassifierFreeSampleModel,
        "TwoClassifierFreeSampleModel": cfgmod.TwoClassifierFreeSampleModel,
        "ClassifierFreeSampleModel_Bodypart": cfgmod.ClassifierFreeSampleModel_Bodypart,
        "TwoClassifierFreeSampleModel_Bodypart": cfgmod.TwoClassifierFreeSampleModel_Bodypart})
    me = types.SimpleNamespace()
    me.args = types.SimpleNamespace(pre_frames=4, pose_length=128, pose_dims=330, batch_size=1, prompt_scale=1.0, audio_scale=1.0,
                                    upper_prompt=synth.H3D_PROMPTS["upper"], hands_prompt=None, lower_prompt=synth.H3D_PROMPTS["lower"])
    me.joints = 55
    me.model = cfgmod.TwoClassifierFreeSampleModel_Bodypart(synth.synth_fill_(RefMDMH3D(synth.default_args(data_path=data_path)).eval(), seed=0))
    me.diffusion = make_diff()                                                       # (replaced by the DDIM-50 one inside _g_test)
    vectors = {p: synth.synth_prompt_vector(p) for p in synth.H3D_PROMPTS.values()}       # (drawn before torch.randn is replaced)
    me.textencoder = lambda prompt: types.SimpleNamespace(loc=vectors[prompt].clone())
    me.vqvae_latent_scale = 10.0
    idx = synth.synth_h3d_part_index()
    me.joint_mask_upper, me.joint_mask_hands, me.joint_mask_lower = idx["upper"], idx["hands"], idx["lower"]
    rec = {}
    for part, dim in (("upper", 156), ("hands", 360), ("lower", 107)):
        vq = synth.synth_fill_(RVQVAE(vq_args(), dim, 512, 512, 512, 2, 2, 512, 3, 3, "relu", None).eval(), seed=11)
        orig = vq.latent2origin

        def tapped(x, part=part, orig=orig):
            rec[f"{part}.latent_in"] = x.clone()
            out = orig(x)
            rec[f"{part}.latent2origin"] = out[0].clone()
            return out
        vq.latent2origin = tapped
        setattr(me, f"vq_model_{part}", vq)
    take = synth.synth_long_take(N_POSE, seed=21)
    data = {"tar_pose": torch.zeros(1, N_POSE, 623), "in_word": take["word"], "in_audio": take["audio"], "latent_in": take["latent"],
            "tar_id": torch.zeros(1, N_POSE, 1, dtype=torch.long)}
    K = 50
    draws, samples = [], []
    for w in range(WINDOWS):
        xT, sn = synth.synth_long_noise(w, K, seed=24)
        draws += [xT] + list(sn)                                   # (ddim_sample still DRAWS its noise at eta = 0, gaussian_diffusion.py:782)
    real_make = make_diff

    def tapped_make(**kw):
        d = real_make(**kw)
        loop = d.ddim_sample_loop
        d.ddim_sample_loop = lambda *a, **k: (samples.append(loop(*a, **k)), samples[-1])[1]
        return d
    g_test.__globals__["create_gaussian_diffusion"] = tapped_make
    with SeededDraws(draws) as sd:
        res = g_test(me, data)
    assert sd.k == len(draws) == WINDOWS * (K + 1) and len(samples) == WINDOWS, (sd.k, len(draws), len(samples))
    out = {"n_pose": np.int64(N_POSE), "steps": np.int64(K), "windows": np.int64(WINDOWS), "samples": torch.stack(samples).numpy(),
           "rec_pose": res["rec_pose"].numpy()}
    for part in ("upper", "hands", "lower"):
        out[f"{part}.latent_in"] = rec[f"{part}.latent_in"].numpy()
        out[f"{part}.latent2origin"] = rec[f"{part}.latent2origin"].numpy()
    for k, v in out.items():
        print(k, getattr(v, "shape", v), float(np.abs(v).mean()) if getattr(v, "ndim", 0) else "")
    np.savez_compressed(os.path.join(HERE, "longform_h3d_outputs.npz"), **out)
    print("wrote longform_h3d_outputs.npz", sum(np.asarray(v).nbytes for v in out.values()) // 1024, "KiB")


def main_loaddata():
    """`CustomTrainer._load_data` (diffusion_rvqvae_trainer.py:244-295), lifted and run the same way: axis-angle poses -> 6D per body part ->
    normalised -> `RVQVAE.map2latent` x 3 -> `latent_in`, the x_0 the diffusion trains on and the seed rows the sampler starts from."""
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    import_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self
    from models.vq.model import RVQVAE
    (load_data,) = lift_methods("_load_data")
    me = types.SimpleNamespace(rank="cpu", joints=55, use_trans=True)
    me.args = types.SimpleNamespace(pose_norm=True, vqvae_latent_scale=5.0, use_motionclip=False)
    masks = synth.synth_joint_masks()
    me.joint_mask_upper, me.joint_mask_hands, me.joint_mask_lower = masks["upper"], masks["hands"], masks["lower"]
    stats = synth.synth_pose_stats()
    me.trans_mean, me.trans_std = stats["trans"]
    (me.mean_upper, me.std_upper), (me.mean_hands, me.std_hands), (me.mean_lower, me.std_lower) = stats["upper"], stats["hands"], stats["lower"]
    me.vq_model_face = types.SimpleNamespace(map2latent=lambda x: torch.zeros(x.shape[0], x.shape[1] // 4, 256))     # (not part of latent_in)
    for part, dim in PARTS:
        setattr(me, f"vq_model_{part}", synth.synth_fill_(RVQVAE(vq_args(), dim, 512, 512, 512, 2, 2, 512, 3, 3, "relu", None).eval(), seed=11))
    n = 64
    clip = synth.synth_pose_clip(2, n, seed=27)
    res = load_data(me, {"pose": torch.cat([clip["pose"], torch.zeros(2, n, 4)], dim=-1), "trans": torch.zeros(2, n, 3), "trans_v": clip["trans_v"],
                         "facial": torch.zeros(2, n, 100), "audio": torch.zeros(2, n * 533, 2), "word": torch.zeros(2, n, dtype=torch.long),
                         "beta": torch.zeros(2, n, 300), "id": torch.zeros(2, n, 1)})
    out = {k: res[k].numpy() for k in ("tar_pose_upper", "tar_pose_hands", "tar_pose_lower", "latent_in", "tar_pose_6d")}
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).mean()))
    np.savez_compressed(os.path.join(HERE, "loaddata_outputs.npz"), **out)
    print("wrote loaddata_outputs.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    if "loaddata" in sys.argv[1:] or len(sys.argv) == 1:
        main_loaddata()
    if "h3d" in sys.argv[1:] or len(sys.argv) == 1:
        main_h3d()
    if "beatx" in sys.argv[1:] or len(sys.argv) == 1:
        main()
