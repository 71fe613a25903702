"""The RVQ-VAE restatement (oracle/rvq_ref.py) against outputs of the reference itself (tests/golden/vq_outputs.npz)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rvq_ref as rr            # noqa: E402
from syntalker_amd import rvqvae, synth      # noqa: E402

PARTS = (("upper", 78), ("hands", 180), ("lower", 57))


@pytest.fixture(scope="module")
def vq_golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "vq_outputs.npz"))


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("part,dim", PARTS)
def test_oracle_matches_reference_outputs(vq_golden, part, dim):
    sd = synth.synth_vq_state_dict(dim, seed=11)
    pose = synth.synth_vq_pose(part, dim)
    assert rel_l2(rr.map2latent(sd, pose), vq_golden[f"{part}.map2latent"]) < 1e-5
    idx = rr.encode(sd, pose)
    assert np.array_equal(idx.numpy(), vq_golden[f"{part}.encode.idx"])
    assert rel_l2(rr.forward_decoder(sd, idx), vq_golden[f"{part}.forward_decoder"]) < 1e-5
    rec = synth.synth_vq_rec_latent(sd, part)
    xq, qidx, commit, perp = rr.residual_vq(sd, rec.permute(0, 2, 1))
    assert np.array_equal(qidx.numpy(), vq_golden[f"{part}.quantizer.idx"])
    assert rel_l2(xq, vq_golden[f"{part}.quantizer.out"]) < 1e-6
    y, commit, perp = rr.latent2origin(sd, rec)
    assert rel_l2(y, vq_golden[f"{part}.latent2origin"]) < 1e-5
    assert abs(float(commit) - float(vq_golden[f"{part}.commit"])) < 1e-6 * max(1.0, abs(float(commit)))
    assert abs(float(perp) - float(vq_golden[f"{part}.perplexity"])) < 1e-4 * float(perp)


@pytest.mark.parametrize("part,dim", PARTS)
def test_state_dict_layout_is_the_references(vq_golden, part, dim):
    """A reference `net` checkpoint must load key for key (diffusion_rvqvae_trainer.py:153-155)."""
    mine = [f"{k}:{'x'.join(map(str, v.shape))}" for k, v in rvqvae.build(dim).state_dict().items()]
    assert mine == [str(k) for k in vq_golden[f"{part}.state_keys"]]


def test_product_module_refuses_cpu_and_training():
    m = rvqvae.build(78)
    with pytest.raises(NotImplementedError):
        m.train()
    with pytest.raises(Exception):
        m.map2latent(torch.zeros(1, 8, 78))          # CPU tensors: no fallback


def test_conv_fragment_packing_layout():
    w = torch.arange(128 * 32 * 3, dtype=torch.float32).view(128, 32, 3) % 251
    p = rvqvae.pack_conv(w, 32, 128).float()          # [taps][8][1][g][r][8]
    assert p.shape == (3, 8, 1, 4, 16, 8)
    for tap, f, g, r, e in ((0, 0, 0, 0, 0), (2, 7, 3, 15, 7), (1, 3, 2, 5, 4)):
        assert p[tap, f, 0, g, r, e] == w[16 * f + r, 8 * g + e, tap]
