"""Pose formats either side of the RVQ-VAEs (`_load_data`, the tail of `_g_test`): the oracle against the reference's own runs (CPU), the
HIP kernels against the oracle and the same fixtures (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import rotation_ref as R
from syntalker_amd import synth
from tests.conftest import GOLDEN, rel_l2

PARTS = (("upper", 78), ("hands", 180), ("lower", 57))


def _loaddata_case():
    fx = np.load(os.path.join(GOLDEN, "loaddata_outputs.npz"))
    clip = synth.synth_pose_clip(2, 64, seed=27)
    return fx, clip, synth.synth_joint_masks(), synth.synth_pose_stats()


def test_load_data_oracle_vs_reference():
    """oracle.rotation_ref.load_data_ref against what the reference's `_load_data` returned (tests/golden/make_longform_golden.py loaddata)."""
    fx, clip, masks, stats = _loaddata_case()
    vq_sds = {p: synth.synth_vq_state_dict(d) for p, d in PARTS}
    with torch.no_grad():
        got = R.load_data_ref(clip["pose"], clip["trans_v"], masks, stats, vq_sds)
    for k in ("tar_pose_upper", "tar_pose_hands", "tar_pose_lower", "tar_pose_6d"):
        assert float((got[k] - torch.from_numpy(fx[k])).abs().max()) < 2e-6, k
    assert rel_l2(got["latent_in"], fx["latent_in"]) < 5e-6


def test_assemble_pose_oracle_vs_reference_g_test():
    """oracle.rotation_ref.assemble_pose_ref on the reference's decoder outputs = the reference's `rec_pose`, bit for bit."""
    fx = np.load(os.path.join(GOLDEN, "longform_outputs.npz"))
    st, masks = synth.synth_pose_stats(), synth.synth_joint_masks()
    dn = lambda p, cut=None: torch.from_numpy(fx[f"{p}.latent2origin"])[..., :cut] * st[p][1] + st[p][0]
    tar = synth.synth_long_take(int(fx["n_pose"]), seed=21)["pose"][:, :352]
    got = R.assemble_pose_ref(dn("upper"), dn("hands"), dn("lower", -3), tar, masks)
    assert torch.equal(got, torch.from_numpy(fx["rec_pose"]))


def test_rotation_helpers_known_answers():
    aa = torch.tensor([[0.0, 0.0, 0.0], [np.pi / 2, 0.0, 0.0], [0.0, 0.0, 1e-7]])
    d6 = R.aa_to_6d(aa)
    assert torch.allclose(d6[0], torch.tensor([1.0, 0, 0, 0, 1, 0])) and torch.allclose(d6[1], torch.tensor([1.0, 0, 0, 0, 0, -1.0]), atol=1e-6)
    back = R.d6_to_aa(d6)
    assert torch.allclose(back[:2], aa[:2], atol=1e-6) and float(back[2].abs().max()) < 1e-6
    g = torch.Generator().manual_seed(0)
    v = torch.randn(500, 3, generator=g)
    v = v / v.norm(dim=-1, keepdim=True) * torch.rand(500, 1, generator=g) * 3.0          # angles below pi: the round trip is the identity
    assert float((R.d6_to_aa(R.aa_to_6d(v)) - v).abs().max()) < 2e-4


@pytest.mark.gpu
def test_rotation_kernels_vs_oracle():
    from syntalker_amd import poses
    g = torch.Generator().manual_seed(1)
    aa = torch.randn(4, 77, 55, 3, generator=g) * 0.8
    aa[0, :5] = 0.0                                              # the small-angle branch
    aa[1, 0, 0] = torch.tensor([1e-7, -2e-7, 0.0])
    d6 = poses.axis_angle_to_rotation_6d(aa.cuda()).cpu()
    assert d6.shape == (4, 77, 55, 6) and float((d6 - R.aa_to_6d(aa)).abs().max()) < 2e-6
    raw = torch.randn(3, 50, 30, 6, generator=g)                 # what a decoder emits: not orthonormal
    got = poses.rotation_6d_to_axis_angle(raw.cuda()).cpu()
    want = R.d6_to_aa(raw)
    # the axis flips sign where the rotation angle is pi (a quaternion's real part at 0): compare as rotations there
    assert rel_l2(R.aa_to_6d(got), R.aa_to_6d(want)) < 1e-5
    ok = want.norm(dim=-1) < 3.0
    assert float((got - want)[ok].abs().max()) < 5e-4
    assert poses.axis_angle_to_rotation_6d(torch.zeros(0, 3, device="cuda")).shape == (0, 6)       # empty input: no launch
    with pytest.raises(Exception):
        poses.axis_angle_to_rotation_6d(torch.zeros(2, 3))                                         # CPU tensors fail loudly


@pytest.mark.gpu
def test_encode_take_and_assemble_pose_vs_reference():
    """The product on the reference's `_load_data` / `_g_test` fixtures: poses -> x_0 latents (three RVQ-VAE encoders), decoder outputs -> rec_pose."""
    from syntalker_amd import poses, rvqvae
    fx, clip, masks, stats = _loaddata_case()
    dev = "cuda"
    vqs = []
    for p, dim in PARTS:
        vq = rvqvae.build(dim).eval()
        vq.load_state_dict(synth.synth_vq_state_dict(dim))
        vqs.append(vq.to(dev))
    got = poses.encode_take(clip["pose"].to(dev), clip["trans_v"].to(dev), *vqs, masks, stats, stats["trans"])
    for k in ("tar_pose_upper", "tar_pose_hands", "tar_pose_lower", "tar_pose_6d"):
        assert float((got[k].cpu() - torch.from_numpy(fx[k])).abs().max()) < 5e-6, k
    e = rel_l2(got["latent_in"].cpu(), fx["latent_in"])
    print(f"encode_take latent_in vs the reference's _load_data: rel-L2 {e:.3e}")
    assert got["latent_in"].shape == (2, 16, 1536) and e < 2e-2                       # bf16 operands through the 16 encoder convolutions
    lf = np.load(os.path.join(GOLDEN, "longform_outputs.npz"))
    dn = lambda p, cut=None: (torch.from_numpy(lf[f"{p}.latent2origin"])[..., :cut] * stats[p][1] + stats[p][0]).to(dev)
    tar = synth.synth_long_take(int(lf["n_pose"]), seed=21)["pose"][:, :352]
    rec = poses.assemble_pose(dn("upper"), dn("hands"), dn("lower", -3), tar.to(dev), masks).cpu()
    err = float((rec - torch.from_numpy(lf["rec_pose"])).abs().max())
    print(f"assemble_pose vs the reference's rec_pose: max abs {err:.3e}, rel-L2 {rel_l2(rec, lf['rec_pose']):.3e}")
    # (fp32 on both sides; a few ill-conditioned joints - decoder outputs whose two 3-vectors are nearly parallel - amplify the last-bit
    # differences of sin / atan2 between the device and the host library: 4e-4 at worst, 4e-6 overall)
    assert rec.shape == (1, 352, 330) and rel_l2(rec, lf["rec_pose"]) < 2e-5 and err < 2e-3
