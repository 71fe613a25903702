"""CPU pin of the wave-per-sequence kernel's design (syntalker_amd/tape.py + the lane-level model tests/kseq_model.py):
the weight tape, consumed fragment by fragment through v_mfma_f32_32x32x16_bf16 register layouts exactly as `k_seq`
does it, reproduces the oracle's folded forward (oracle/denoiser_ref.py, itself pinned to the reference's outputs)."""
import numpy as np
import torch

from oracle import denoiser_ref as dr
from syntalker_amd import synth, tape
from tests import kseq_model as km
from tests.refmodel import synth_state_dict


def test_fragment_order_round_trip():
    x = torch.randn(3, 32, 1536)
    f = tape.to_fragment_order(x)
    assert f.shape == (3, 48, 4, 64, 4)
    assert torch.equal(tape.from_fragment_order(f), x)
    # what a lane holds: channel = 32 nf + 8 q + 4 hi + r of token lane & 31
    nf, q, lane, r = 17, 2, 45, 3
    assert f[1, nf, q, lane, r] == x[1, lane & 31, 32 * nf + 8 * q + 4 * (lane >> 5) + r]
    b = tape.to_fragment_order_bf16(x)
    c, e = 1, 6
    assert b[2, nf, c, lane, e] == x[2, lane & 31, 32 * nf + 16 * c + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3)].to(torch.bfloat16)


def test_mfma_model_is_a_matrix_product():
    rng = np.random.default_rng(0)
    A, B = rng.standard_normal((32, 16)).astype(np.float32), rng.standard_normal((16, 32)).astype(np.float32)
    a, b = np.zeros((64, 8), np.float32), np.zeros((64, 8), np.float32)
    for l in range(64):
        a[l], b[l] = A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8], B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31]
    d = km.mfma32(a, b, np.zeros((64, 16), np.float32))
    want = A @ B
    for l in range(64):
        for v in range(16):
            assert abs(d[l, v] - want[(v & 3) + 8 * (v >> 2) + 4 * (l >> 5), l & 31]) < 1e-4


def test_tape_through_the_lane_level_model_reproduces_the_oracle():
    sd = synth_state_dict("beatx")
    fw = dr.fold_weights(sd)
    t_tape, bias = tape.build_tape(sd, fw["A"])
    assert t_tape.shape == (tape.TAPE_ALLOC_FRAGS, 64, 8) and bias.shape == (9, 2048)
    # the head once more behind the tail: the kernel's DMA look-ahead across a step boundary (syn_seq.inc)
    assert torch.equal(t_tape[tape.TAPE_FRAGS:], t_tape[:tape.LOOK_CHUNKS * tape.CHUNK_FRAGS])
    t_tape = t_tape[:tape.TAPE_FRAGS]
    assert tape.TAPE_FRAGS == 36096 and tape.TAPE_FRAGS % tape.CHUNK_FRAGS == 0 and tape.TAPE_FRAGS // tape.CHUNK_FRAGS == 2256
    y, x = synth.synth_clip_inputs(1, seed=5), synth.synth_latent(1, seed=5)
    cond, te = dr.clip_conditioning(sd, y, fw), dr.time_table(sd, fw)
    t = torch.tensor([417])
    with torch.no_grad():
        want = dr.mdm_forward_folded(sd, fw, cond, te, x, t)                       # (1, 1536, 1, 32)
    x_btc = x.reshape(1, 1536, 32).transpose(1, 2).contiguous()
    xb = tape.to_fragment_order_bf16(x_btc)[0].float().numpy()
    rc, rs = (lambda fr: (fr.cos().numpy(), fr.sin().numpy()))(torch.einsum("i,j->ij", torch.arange(32.), sd["rel_pos.inv_freq"].float()))
    gelu = lambda a: torch.nn.functional.gelu(torch.from_numpy(a)).numpy()
    w = km.Wave(t_tape.float().numpy(), bias.numpy())
    out = w.step(xb, cond[0].numpy(), te[417].numpy(), rc, rs, gelu)
    got = tape.from_fragment_order(torch.from_numpy(out)[None])[0].T.reshape(1, 1536, 1, 32)   # (32, 1536) -> (1, 1536, 1, 32)
    err = float((got - want).norm() / want.norm())
    assert err < 1.5e-2, err          # bf16 operands, fp32 accumulation: the kernel's rounding points
