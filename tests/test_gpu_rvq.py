"""RVQ-VAE on the HIP kernels (csrc/syn_rvq.inc, through the C ABI) vs the oracle and the reference's own outputs.

Tolerances: the conv stacks run bf16 operands with fp32 accumulation and an fp32 residual chain: rel-L2 <= 2e-2 (the
bar SURVEY §8(c) sets for bf16 kernels).  The residual quantiser is fp32 and a discrete decision: indices must EQUAL
the reference's on the golden latents, the quantised rows must agree to fp32 round-off.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rvq_ref as rr            # noqa: E402
from syntalker_amd import _lib, rvqvae, synth       # noqa: E402

pytestmark = pytest.mark.gpu
PARTS = (("upper", 78), ("hands", 180), ("lower", 57))
DEV = "cuda"


@pytest.fixture(scope="module")
def vq_golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "vq_outputs.npz"))


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm())


def _model(dim):
    m = rvqvae.build(dim)
    m.load_state_dict(synth.synth_vq_state_dict(dim, seed=11))
    return m.to(DEV)


@pytest.mark.parametrize("cfg", [
    dict(cin=96, cout=128, taps=3, stride=1, dil=1, pad=1, up=0, t=64, relu_in=0, relu_out=1),
    dict(cin=512, cout=512, taps=4, stride=2, dil=1, pad=1, up=0, t=64, relu_in=0, relu_out=0),
    dict(cin=512, cout=512, taps=3, stride=1, dil=9, pad=9, up=0, t=32, relu_in=1, relu_out=0),
    dict(cin=512, cout=512, taps=3, stride=1, dil=3, pad=3, up=0, t=128, relu_in=1, relu_out=0),
    dict(cin=512, cout=512, taps=1, stride=1, dil=1, pad=0, up=0, t=16, relu_in=1, relu_out=0, resid=1),
    dict(cin=512, cout=512, taps=3, stride=1, dil=1, pad=1, up=1, t=20, relu_in=0, relu_out=0),
    dict(cin=512, cout=256, taps=3, stride=1, dil=1, pad=1, up=0, t=100, relu_in=0, relu_out=0, valid=180),
])
def test_conv1d_kernel_vs_torch(cfg):
    """One launch of the generic Conv1d kernel against F.conv1d on the same bf16-rounded operands (fp32 accumulate both
    sides: agreement to accumulation-order round-off), for every geometry the encoder / decoder uses + ragged lengths."""
    g = torch.Generator().manual_seed(5)
    n, t, cin, cout = 3, cfg["t"], cfg["cin"], cfg["cout"]
    valid = cfg.get("valid", cout)
    x = torch.randn(n, t, cin, generator=g).to(torch.bfloat16)
    w = (torch.randn(valid, cin, cfg["taps"], generator=g) * (cin * cfg["taps"]) ** -0.5).to(torch.bfloat16).float()
    b = torch.randn(valid, generator=g) * 0.1
    t_up = t << cfg["up"]
    t_out = (t_up + 2 * cfg["pad"] - cfg["dil"] * (cfg["taps"] - 1) - 1) // cfg["stride"] + 1
    resid = torch.randn(n, t_out, cout, generator=g) if cfg.get("resid") else None
    xin = x.float().permute(0, 2, 1)
    if cfg["relu_in"]:
        xin = F.relu(xin)
    if cfg["up"]:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    want = F.conv1d(xin, w, b, stride=cfg["stride"], padding=cfg["pad"], dilation=cfg["dil"]).permute(0, 2, 1)
    if resid is not None:
        want = want + resid
    if cfg["relu_out"]:
        want = F.relu(want)
    wp = rvqvae.pack_conv(w.to(DEV), cin, cout)
    bp = torch.zeros(cout, device=DEV)
    bp[:valid] = b.to(DEV)
    cv = _lib.SynVqConv(wp.data_ptr(), bp.data_ptr(), cin, cout, valid, cfg["taps"], cfg["stride"], cfg["dil"], cfg["pad"], cfg["up"],
                        cfg["relu_in"], cfg["relu_out"])
    xd = x.to(DEV)
    yf = torch.full((n, t_out, valid), float("nan"), device=DEV)
    yb = torch.empty(n, t_out, cout, device=DEV, dtype=torch.bfloat16)
    rd = resid.to(DEV) if resid is not None else None
    _lib.check(_lib.load().syn_vq_conv1d(C.byref(cv), xd.data_ptr(), _lib.ptr(rd), yf.data_ptr(), valid, yb.data_ptr(), n, t, t_out,
                                         torch.cuda.current_stream().cuda_stream), "syn_vq_conv1d")
    torch.cuda.synchronize()
    assert torch.isfinite(yf).all()
    assert rel_l2(yf, want) < 2e-5
    assert rel_l2(yb[..., :valid].float(), want) < 4e-3


@pytest.mark.parametrize("part,dim", PARTS)
def test_rvqvae_vs_reference_outputs(vq_golden, part, dim):
    m = _model(dim)
    sd = synth.synth_vq_state_dict(dim, seed=11)
    pose = synth.synth_vq_pose(part, dim)
    lat = m.map2latent(pose.to(DEV))
    assert lat.shape == (2, 16, 512)
    e = rel_l2(lat, vq_golden[f"{part}.map2latent"])
    print(part, "map2latent rel-L2 vs reference", e)
    assert e < 2e-2
    # residual quantiser on the golden latent: the reference's indices, fp32 round-off on the rows
    rec = synth.synth_vq_rec_latent(sd, part)
    qf, idx, commit, perp = m._quantize(rec.to(DEV))
    assert np.array_equal(idx.cpu().numpy(), vq_golden[f"{part}.quantizer.idx"])
    assert rel_l2(qf.permute(0, 2, 1), vq_golden[f"{part}.quantizer.out"]) < 1e-6
    y, commit2, perp2 = m.latent2origin(rec.to(DEV))
    assert y.shape == (2, 64, dim)
    e = rel_l2(y, vq_golden[f"{part}.latent2origin"])
    print(part, "latent2origin rel-L2 vs reference", e)
    assert e < 2e-2
    assert abs(float(commit2) - float(vq_golden[f"{part}.commit"])) < 1e-4 * abs(float(vq_golden[f"{part}.commit"])) + 1e-9
    assert abs(float(perp2) - float(vq_golden[f"{part}.perplexity"])) < 1e-4 * float(vq_golden[f"{part}.perplexity"])
    # indices -> pose
    y2 = m.forward_decoder(torch.from_numpy(vq_golden[f"{part}.encode.idx"]).to(DEV))
    assert rel_l2(y2, vq_golden[f"{part}.forward_decoder"]) < 2e-2
    # encode(): the device quantiser on the device latent must pick what the oracle picks on that same latent
    idx_e, codes = m.encode(pose.to(DEV))
    want_idx = rr.residual_vq(sd, lat.cpu().permute(0, 2, 1))[1]
    assert (idx_e.cpu() == want_idx).float().mean() > 0.98
    assert codes.shape == (6, 2, 512, 16)
    assert rel_l2(codes.sum(0), rr.codes_from_indices(sd, idx_e.cpu())) < 1e-5


def test_rvqvae_ragged_batch_and_lengths():
    """3 clips of 20 / 36 latent rows (not multiples of the 16-row quantiser groups or of the 32/64-position conv tiles)."""
    dim = 57
    m = _model(dim)
    sd = synth.synth_vq_state_dict(dim, seed=11)
    for n, t in ((3, 5), (1, 9), (5, 32)):
        rec = synth.synth_vq_rec_latent(sd, "lower", n=n, t=t)
        y, commit, perp = m.latent2origin(rec.to(DEV))
        want, wc, wp = rr.latent2origin(sd, rec)
        assert y.shape == want.shape == (n, 4 * t, dim)
        assert rel_l2(y, want) < 2e-2
        assert abs(float(commit) - float(wc)) < 1e-4 * float(wc) + 1e-9 and abs(float(perp) - float(wp)) < 1e-3 * float(wp)
        pose = synth.synth_vq_pose("lower", dim, n=n, t=4 * t)
        assert rel_l2(m.map2latent(pose.to(DEV)), rr.map2latent(sd, pose)) < 2e-2


def test_rvqvae_batch_independence():
    """A clip's output does not depend on what else is in the batch (tiles never span clips)."""
    dim = 78
    m = _model(dim)
    sd = synth.synth_vq_state_dict(dim, seed=11)
    rec = synth.synth_vq_rec_latent(sd, "upper", n=6, t=32).to(DEV)
    all6 = m.latent2origin(rec)[0]
    one = m.latent2origin(rec[4:5])[0]
    assert torch.equal(all6[4:5], one)


def test_decode_take_vs_oracle():
    """The step after the sampler (trainer :458-500): 1536-channel latents of a whole take -> three body-part poses + root
    translation, against the restatement.  The latents are sums of codes at 1/5 scale, as a trained sampler emits them."""
    from oracle.longform_ref import decode_take_ref
    from syntalker_amd import longform
    dims = {"upper": 78, "hands": 180, "lower": 57}
    sds = {k: synth.synth_vq_state_dict(d, seed=11) for k, d in dims.items()}
    vqs = {k: _model(d) for k, d in dims.items()}
    lat = torch.cat([synth.synth_vq_rec_latent(sds[k], k, n=2, t=60) for k in ("upper", "hands", "lower")], dim=-1) / 5.0
    tm, ts = torch.tensor([0.01, 0.9, -0.02]), torch.tensor([0.5, 0.1, 0.4])
    got = longform.decode_take(lat.to(DEV), vqs["upper"], vqs["hands"], vqs["lower"], 5.0, trans_mean=tm.to(DEV), trans_std=ts.to(DEV))
    want = decode_take_ref(sds, lat, 5.0, trans_mean=tm, trans_std=ts)
    for k in ("upper", "hands", "lower", "trans"):
        assert got[k].shape == want[k].shape
        e = rel_l2(got[k], want[k])
        print(k, e)
        assert e < 2e-2
    assert got["lower"].shape[-1] == 54 and got["trans"].shape == (2, 240, 3)


def test_rvqvae_h3d_body_part_widths():
    """The text-prompt trainer builds the same model on 156 / 360 / 107 pose channels (h3d_diffusion_new_trainer.py:104-146)."""
    for dim in (156, 360, 107):
        m = _model(dim)
        sd = synth.synth_vq_state_dict(dim, seed=11)
        rec = synth.synth_vq_rec_latent(sd, "h3d", n=2, t=8)
        pose = synth.synth_vq_pose("h3d", dim, n=2, t=32)
        assert rel_l2(m.latent2origin(rec.to(DEV))[0], rr.latent2origin(sd, rec)[0]) < 2e-2
        assert rel_l2(m.map2latent(pose.to(DEV)), rr.map2latent(sd, pose)) < 2e-2


def test_rvqvae_round_trip_properties_at_batch_size():
    """256 clips x 128 frames (the batch the training loader hands over), no oracle needed:
    (1) forward() = latent2origin(map2latent(.)) and is finite;
    (2) decoding the indices encode() returns reproduces that reconstruction (the two differ only by the rounding of the
        straight-through sum vs the plain sum of codes, amplified to the bf16 noise floor by the decoder);
    (3) the histogram behind the perplexity counts every row once per layer; commit loss = mean squared residual >= 0."""
    dim = 78
    m = _model(dim)
    pose = synth.synth_vq_pose("upper", dim, n=256, t=128).to(DEV)
    out = m(pose)
    lat = m.map2latent(pose)
    y, commit, perp = m.latent2origin(lat)
    assert torch.equal(out["rec_pose"], y) and torch.isfinite(y).all() and y.shape == (256, 128, dim)
    idx, codes = m.encode(pose)
    assert idx.shape == (256, 32, 6) and int(idx.min()) >= 0 and int(idx.max()) < 512
    y2 = m.forward_decoder(idx)
    assert rel_l2(y2, y) < 1e-2          # fp32-round-off differences in the decoder input re-roll its bf16 roundings (3.7e-3)
    assert float(commit) >= 0 and 1.0 <= float(perp) <= 512.0
    qf, idx2, _, _ = m._quantize(lat)
    assert torch.equal(idx2, idx)
    assert rel_l2(codes.sum(0).permute(0, 2, 1), qf) < 1e-6


def test_quantiser_on_the_fp32_matrix_pipe_keeps_the_reference_indices():
    """More than 1024 rows run the 16-row quantiser, whose distances come from v_mfma_f32_16x16x4_f32 (fp32 in, the same
    fmaf chain over the 512 dims as the vector version); up to 1024 rows the 4-row vector version runs.  Indices are discrete
    decisions: both must be the oracle's, bit for bit, and so must the rows they share."""
    dim = 78
    m = _model(dim)
    sd = synth.synth_vq_state_dict(dim, seed=11)
    rec = synth.synth_vq_rec_latent(sd, "upper", n=40, t=32)             # 1280 rows
    rec = rec + 0.05 * torch.randn(rec.shape, generator=torch.Generator().manual_seed(3))      # off the codes: real decisions
    qf16, idx16, c16, p16 = m._quantize(rec.to(DEV))
    qf4, idx4, _, _ = m._quantize(rec[:32].to(DEV))                      # 1024 rows: vector version
    want_q, want_idx, wc, wp = rr.residual_vq(sd, rec.permute(0, 2, 1))
    assert torch.equal(idx16.cpu(), want_idx) and torch.equal(idx4.cpu(), want_idx[:32])
    assert torch.equal(qf16[:32], qf4)
    assert rel_l2(qf16.cpu(), want_q.permute(0, 2, 1)) < 1e-6
    assert abs(float(c16) - float(wc)) < 1e-4 * float(wc) + 1e-9 and abs(float(p16) - float(wp)) < 1e-3 * float(wp)
