"""The drop-in claim, executed: the REFERENCE's own driver code - `CustomTrainer._g_test` of diffusion_rvqvae_trainer.py, lifted from its file and
compiled unchanged - runs against THIS build's `MDM` and `create_gaussian_diffusion()` (through the reference's import names, syntalker_amd/dropin)
and produces what `longform.sample_long` produces.  Build container only (the reference tree is not on the GPU box); the device engine is replaced
by its CPU stand-ins (tests/cpu_engine.py), everything above it - `MDM`, `SpacedDiffusion.p_sample_loop`, `process._fused`, the caches - is the product."""
import importlib.util
import os
import types

import pytest
import torch

from syntalker_amd import longform, synth

REF_TREE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir(REF_TREE), reason="the reference tree exists in the build container only")
def test_reference_g_test_drives_this_build(monkeypatch):
    import sys
    from tests import cpu_engine
    cpu_engine.install(monkeypatch)
    monkeypatch.setattr(sys, "path", list(sys.path))                         # (the generator puts the reference tree on sys.path: undone afterwards)
    for name in ("utils", "utils.rotation_conversions", "make_golden", "make_vq_golden"):
        monkeypatch.delitem(sys.modules, name, raising=False)                # ... and whatever it imports from there leaves sys.modules again
        monkeypatch.setitem(sys.modules, name, None)
        monkeypatch.delitem(sys.modules, name)
    spec = importlib.util.spec_from_file_location("make_longform_golden", os.path.join(HERE, "golden", "make_longform_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    monkeypatch.syspath_prepend(os.path.join(HERE, "golden"))
    spec.loader.exec_module(gen)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self, raising=False)
    g_test, inverse_selection_tensor = gen.lift_methods("_g_test", "inverse_selection_tensor")
    # the reference's import statements, resolved to this build (train.py:85-94, diffusion_rvqvae_trainer.py:26-27)
    from syntalker_amd.dropin.diffusion.model_util import create_gaussian_diffusion
    from syntalker_amd.dropin.models.denoiser import MDM
    me = types.SimpleNamespace()
    me.args = types.SimpleNamespace(vqvae_squeeze_scale=4, pre_frames=4, pose_length=128, pose_dims=330, batch_size=1, pose_norm=True)
    me.joints = 55
    masks = synth.synth_joint_masks()
    me.joint_mask_upper, me.joint_mask_hands, me.joint_mask_lower = masks["upper"], masks["hands"], masks["lower"]
    me.inverse_selection_tensor = lambda *a: inverse_selection_tensor(me, *a)
    me.model = synth.synth_fill_(MDM(synth.default_args()).eval(), seed=0)
    me.diffusion = create_gaussian_diffusion(use_ddim=True)                   # `_g_test` calls its p_sample_loop: 50 ancestral steps per window
    me.vqvae_latent_scale, me.use_trans = 5.0, True
    stats = synth.synth_pose_stats()
    me.trans_mean, me.trans_std = stats["trans"]
    (me.mean_upper, me.std_upper), (me.mean_hands, me.std_hands), (me.mean_lower, me.std_lower) = stats["upper"], stats["hands"], stats["lower"]
    seen = {}
    for part, dim in (("upper", 78), ("hands", 180), ("lower", 57)):          # the decoders are not the subject here: record what they are handed
        def l2o(x, part=part, dim=dim):
            seen[part] = x.clone()
            return torch.zeros(x.shape[0], 4 * x.shape[1], dim), 0.0, 0.0
        setattr(me, f"vq_model_{part}", types.SimpleNamespace(latent2origin=l2o))
    n = gen.N_POSE
    take = synth.synth_long_take(n, seed=21)
    data = {"tar_pose": take["pose"], "tar_beta": torch.zeros(1, n, 300), "tar_exps": torch.zeros(1, n, 100), "tar_contact": torch.zeros(1, n, 4),
            "tar_trans": torch.zeros(1, n, 3), "in_word": take["word"], "in_audio": take["audio"], "latent_in": take["latent"],
            "tar_id": torch.zeros(1, n, 1, dtype=torch.long)}
    torch.manual_seed(5)
    with torch.no_grad():
        res = g_test(me, data)                                                # the reference's lines, this build's model and loop
    assert res["rec_pose"].shape == (1, 352, 330) and res["rec_trans"].shape == (1, 352, 3)
    theirs = torch.cat([seen[p] for p in ("upper", "hands", "lower")], dim=-1) / 5.0
    torch.manual_seed(5)                                                      # the same draws (x_T, then the loop's seed) in the same order
    ours = longform.sample_long(me.diffusion, me.model, take["audio"], take["word"], take["latent"], n)
    assert theirs.shape == ours.shape == (1, 88, 1536) and torch.isfinite(ours).all()
    assert float((theirs - ours).abs().max()) < 1e-5                          # (progress bar on: single-step replays instead of ten-step ones)
