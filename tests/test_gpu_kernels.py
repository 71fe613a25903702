"""Unit parity of the individual HIP stages, through the C ABI, vs plain fp32 torch on the same
bf16-rounded operands.  Asymmetric random data so a transposed MFMA fragment cannot pass."""
import pytest
import torch

from tests import block_ops, refmodel
from tests.conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from syntalker_amd import _lib
    return _lib


def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("mt", [32, 64, 128])
@pytest.mark.parametrize("m,n,k", [(32, 512, 512), (96, 1536, 512), (160, 512, 1024), (288, 1024, 1536)])
def test_gemm_matches_fp32_matmul(lib, mt, m, n, k):
    from syntalker_amd import engine
    g = torch.Generator().manual_seed(m * 7 + n + k)
    x = torch.randn(m, k, generator=g).cuda()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).cuda()
    b = torch.randn(n, generator=g).cuda()
    xb = _bf(x).contiguous()
    wp = engine.pack_weight(w)
    y = torch.empty(m, n, device="cuda")
    lib.check(lib.load().syn_test_gemm(xb.data_ptr(), wp.data_ptr(), b.data_ptr(), m, n, k, mt, y.data_ptr(),
                                       lib.current_stream()), "syn_test_gemm")
    torch.cuda.synchronize()
    want = xb.float() @ _bf(w).float().T + b
    # same bf16 operands, fp32 accumulate on both sides: only summation order differs
    assert rel_l2(y.cpu(), want.cpu()) < 2e-6
    assert torch.isfinite(y).all()


@pytest.mark.parametrize("m,n,k", [(32, 512, 6144), (20, 512, 2048), (64, 1024, 2176), (1024, 512, 512), (1000, 1536, 640), (256, 512, 2304)])
def test_training_linear_kernels_match_fp32_matmul(lib, m, n, k):
    """syn_linear's kernels for the training step's shapes - 128-column tiles with the activation block resident in the LDS (row tile by
    shape), the split-K kernel for a few rows against a long K (embed_text: 32 x 6144), the streaming loop beyond their limits - against
    an fp32 matmul of the same bf16 operands, ragged row counts included; and the resident kernels bitwise against the streaming loop
    (same products, same order) where K is not split."""
    from syntalker_amd import engine
    g = torch.Generator().manual_seed(m + n + k)
    xb = _bf(torch.randn(m, k, generator=g)).cuda().contiguous()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).cuda()
    b = torch.randn(n, generator=g).cuda()
    wp = engine.pack_weight(w)
    want = xb.float() @ _bf(w).float().T + b
    outs = {}
    try:
        for mode in (2, 0):
            lib.load().syn_debug_gemm_resident(mode)
            y = torch.full((m, n), float("nan"), device="cuda")
            lib.check(lib.load().syn_linear(xb.data_ptr(), wp.data_ptr(), b.data_ptr(), m, n, k, y.data_ptr(), lib.current_stream()), "syn_linear")
            torch.cuda.synchronize()
            assert torch.isfinite(y).all() and rel_l2(y.cpu(), want.cpu()) < 2e-6, (mode, rel_l2(y.cpu(), want.cpu()))
            outs[mode] = y
    finally:
        lib.load().syn_debug_gemm_resident(2)
    if not (m <= 64 and k >= 2048):
        assert torch.equal(outs[2], outs[0])


@pytest.mark.parametrize("m,n,k,rps", [(1024, 512, 1024, 32), (1000, 512, 640, 8), (96, 1024, 4224, 32), (2304, 512, 512, 64)])
def test_linear_epilogues_residual_droppath_and_gelu(lib, m, n, k, rps):
    """syn_linear_res (residual + per-sample factor * (x W^T + b), the tail of a pre-LN branch) and syn_linear_gelu (fp32 pre-activation + bf16
    GELU) against fp32 torch on the same bf16 operands, on every kernel the shapes select: 128-column resident tiles (row tile 64 / 16), the
    streaming loop (K too long for the LDS) and the large-row-tile path (m > 2048, the x^T pack as a launch of its own)."""
    import torch.nn.functional as F
    from syntalker_amd import engine
    g = torch.Generator().manual_seed(m + n + k)
    xb = _bf(torch.randn(m, k, generator=g)).cuda().contiguous()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).cuda()
    b = torch.randn(n, generator=g).cuda()
    res = torch.randn(m, n, generator=g).cuda()
    fac = (torch.rand((m + rps - 1) // rps, generator=g) < 0.8).float().div(0.8).cuda()
    wp = engine.pack_weight(w)
    lin = xb.float() @ _bf(w).float().T + b
    L, st = lib.load(), lib.current_stream()
    pack = m % 32 == 0 and k % 16 == 0
    xt = torch.empty(k * m * 2, dtype=torch.uint8, device="cuda") if pack else None
    y = torch.full((m, n), float("nan"), device="cuda")
    lib.check(L.syn_linear_res(xb.data_ptr(), wp.data_ptr(), b.data_ptr(), res.data_ptr(), fac.data_ptr(), rps, m, n, k, y.data_ptr(), lib.ptr(xt), st),
              "syn_linear_res")
    want = res + lin * fac.repeat_interleave(rps)[:m, None]
    assert rel_l2(y.cpu(), want.cpu()) < 2e-6
    y2 = torch.full((m, n), float("nan"), device="cuda")
    lib.check(L.syn_linear_res(xb.data_ptr(), wp.data_ptr(), None, res.data_ptr(), None, 0, m, n, k, y2.data_ptr(), None, st), "syn_linear_res")
    assert rel_l2(y2.cpu(), (res + lin - b).cpu()) < 2e-6                 # no bias, no factor: residual + x W^T
    pre, act = torch.full((m, n), float("nan"), device="cuda"), torch.zeros(m, n, dtype=torch.bfloat16, device="cuda")
    xt2 = torch.empty_like(xt) if pack else None
    lib.check(L.syn_linear_gelu(xb.data_ptr(), wp.data_ptr(), b.data_ptr(), m, n, k, pre.data_ptr(), act.data_ptr(), lib.ptr(xt2), st), "syn_linear_gelu")
    torch.cuda.synchronize()
    assert rel_l2(pre.cpu(), lin.cpu()) < 2e-6
    assert rel_l2(act.float().cpu(), _bf(F.gelu(pre)).float().cpu()) < 1e-4      # (bf16 of the exact-erf GELU; erff vs torch's erf: a last-bit flip here and there)
    if pack:
        from syntalker_amd import training
        assert torch.equal(xt, training._pack_t(xb, k, m)) and torch.equal(xt2, xt)     # the x^T fragments of the backward's weight-gradient GEMM


def test_linear_backward_operand_pass_and_pair_bias_sum(lib):
    """syn_linear_bwd_prep (bf16 copy, bf16 transpose, per-64-row column sums of factor * dy; dy read in place as a column slice of a wider
    tensor, or as the backward of an average pool over 4 rows) and the bias-gradient sum riding syn_linear_pair, against torch."""
    from syntalker_amd import engine, training
    g = torch.Generator().manual_seed(9)
    M, N, K = 256, 1024, 512
    wide = torch.randn(M, N + 256, generator=g).cuda()               # dy = columns 128 .. 128 + N of a wider gradient (one piece of a torch.cat's backward)
    pooled = torch.randn(M // 4, N, generator=g).cuda()              # dy = the backward of an average pool over 4 consecutive rows
    fac = (torch.rand(M // 32, generator=g) < 0.7).float().div(0.7).cuda()
    L, st = lib.load(), lib.current_stream()
    for case in ("factors", "slice", "pool", "plain"):
        scale, ld, row_div, cs = None, N, 1, 1.0
        if case == "slice":
            src, ld = wide[:, 128:128 + N], N + 256
            eff = src.clone()
        elif case == "pool":
            src, row_div, cs = pooled, 4, 0.25
            eff = pooled.repeat_interleave(4, 0) * 0.25
        else:
            src = wide[:, :N].contiguous()
            eff = src.clone()
            if case == "factors":
                scale = fac
                eff = eff * scale.repeat_interleave(32)[:, None]
        dyb, dybt = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.empty(N, M, dtype=torch.bfloat16, device="cuda")
        part = torch.empty(M // 64, N, device="cuda")
        lib.check(L.syn_linear_bwd_prep(src.data_ptr(), ld, row_div, cs, M, N, lib.ptr(scale), 32, dyb.data_ptr(), dybt.data_ptr(), part.data_ptr(), st),
                  "syn_linear_bwd_prep")
        torch.cuda.synchronize()
        assert rel_l2(dyb.float().cpu(), eff.cpu()) < 3e-3 and torch.equal(dybt, dyb.t().contiguous()), case
        assert rel_l2(part.sum(0).cpu(), eff.sum(0).cpu()) < 2e-5, case
        # the pair launch: dx = dy . W, dW = dy^T . x, db = the sum of the partials
        xb = _bf(torch.randn(M, K, generator=g)).cuda().contiguous()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
        wt, xt = training._pack_t(w, K, N), training._pack_t(xb, K, M)
        dx, dw, db = torch.empty(M, K, device="cuda"), torch.empty(N, K, device="cuda"), torch.full((N,), float("nan"), device="cuda")
        lib.check(L.syn_linear_pair(dyb.data_ptr(), wt.data_ptr(), M, K, N, dx.data_ptr(), dybt.data_ptr(), xt.data_ptr(), N, K, M, dw.data_ptr(),
                                    part.data_ptr(), M // 64, N, db.data_ptr(), st), "syn_linear_pair")
        torch.cuda.synchronize()
        assert rel_l2(db.cpu(), part.sum(0).cpu()) < 1e-6
        assert rel_l2(dx.cpu(), (dyb.float() @ _bf(w).float()).cpu()) < 2e-6 and rel_l2(dw.cpu(), (dyb.float().T @ xb.float()).cpu()) < 2e-6
        db2 = torch.empty(N, device="cuda")
        lib.check(L.syn_colsum_parts(part.data_ptr(), M // 64, N, db2.data_ptr(), st), "syn_colsum_parts")
        assert torch.equal(db2, db)


def test_linear_pair_on_128_column_shapes(lib):
    """syn_linear_pair where an output width is a multiple of 128 but not of 512 (input_process2: 1280 input features; mix_audio_text: 256 outputs):
    the pair runs on the 128-column tiles, and equals two syn_linear launches bitwise.  Beyond 4096 rows (text_encoder_body's weight gradient at more
    than 32 clips) the 128-column kernel's resident activation block no longer holds the contraction: slices of 4096, added up in order
    (4608 = 36 clips, 9216 = 72 clips: a partial and a full last slice)."""
    from syntalker_amd import training
    g = torch.Generator().manual_seed(19)
    L, st = lib.load(), lib.current_stream()
    for M, N, K in ((1024, 512, 1280), (1024, 256, 512), (4096, 256, 384), (4608, 256, 384), (9216, 256, 384), (8192, 128, 128)):
        dyb = _bf(torch.randn(M, N, generator=g)).cuda().contiguous()
        dybt = dyb.t().contiguous()
        xb = _bf(torch.randn(M, K, generator=g)).cuda().contiguous()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
        wt, xt = training._pack_t(w, K, N), training._pack_t(xb, K, M)
        part = torch.randn(M // 64, N, generator=g).cuda()
        dx, dw, db = torch.empty(M, K, device="cuda"), torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
        lib.check(L.syn_linear_pair(dyb.data_ptr(), wt.data_ptr(), M, K, N, dx.data_ptr(), dybt.data_ptr(), xt.data_ptr(), N, K, M, dw.data_ptr(),
                                    part.data_ptr(), M // 64, N, db.data_ptr(), st), "syn_linear_pair")
        dx1, dw1 = torch.empty_like(dx), torch.empty_like(dw)
        lib.check(L.syn_linear(dyb.data_ptr(), wt.data_ptr(), None, M, K, N, dx1.data_ptr(), st), "syn_linear")
        lib.check(L.syn_linear(dybt.data_ptr(), xt.data_ptr(), None, N, K, M, dw1.data_ptr(), st), "syn_linear")
        torch.cuda.synchronize()
        assert torch.equal(dx, dx1) and torch.equal(dw, dw1), (M, N, K)
        assert rel_l2(db.cpu(), part.sum(0).cpu()) < 1e-6
        assert rel_l2(dx.cpu(), (dyb.float() @ _bf(w).float()).cpu()) < 2e-6 and rel_l2(dw.cpu(), (dyb.float().T @ xb.float()).cpu()) < 3e-6


def test_gemm_tile_sizes_agree_bitwise(lib):
    """The per-element K-summation order does not depend on the M tile."""
    from syntalker_amd import engine
    g = torch.Generator().manual_seed(5)
    x, w = _bf(torch.randn(256, 512, generator=g)).cuda(), torch.randn(512, 512, generator=g).cuda()
    wp = engine.pack_weight(w)
    outs = []
    for mt in (32, 64, 128):
        y = torch.empty(256, 512, device="cuda")
        lib.check(lib.load().syn_test_gemm(x.data_ptr(), wp.data_ptr(), None, 256, 512, 512, mt, y.data_ptr(),
                                           lib.current_stream()), "syn_test_gemm")
        outs.append(y.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


@pytest.mark.parametrize("n_seq", [1, 5])
def test_attention_matches_sdpa(lib, n_seq):
    g = torch.Generator().manual_seed(n_seq)
    q, k, v = (torch.randn(n_seq, 32, 4, 128, generator=g) for _ in range(3))
    q = q * 2.0                                      # sharper softmax than unit-variance logits
    qb, kb, vb = _bf(q).cuda(), _bf(k).cuda(), _bf(v).cuda()
    vt = vb.permute(0, 2, 3, 1).contiguous()         # [seq][head][d][token]
    o = torch.empty(n_seq, 32, 512, dtype=torch.bfloat16, device="cuda")
    lib.check(lib.load().syn_test_attention(qb.reshape(n_seq, 32, 512).contiguous().data_ptr(),
                                            kb.reshape(n_seq, 32, 512).contiguous().data_ptr(), vt.data_ptr(), n_seq,
                                            o.data_ptr(), lib.current_stream()), "syn_test_attention")
    torch.cuda.synchronize()
    want = torch.nn.functional.scaled_dot_product_attention(
        qb.float().permute(0, 2, 1, 3).cpu(), kb.float().permute(0, 2, 1, 3).cpu(), vb.float().permute(0, 2, 1, 3).cpu())
    want = want.permute(0, 2, 1, 3).reshape(n_seq, 32, 512)
    assert rel_l2(o.float().cpu(), want) < 8e-3      # P and O are rounded to bf16


def test_layout_round_trip_and_bf16_shadow(lib):
    x = torch.randn(3, 1536, 1, 32).cuda()
    tm, tb = torch.empty(3 * 32, 1536, device="cuda"), torch.empty(3 * 32, 1536, dtype=torch.bfloat16, device="cuda")
    lib.check(lib.load().syn_to_token_major(x.data_ptr(), 3, tm.data_ptr(), tb.data_ptr(), lib.current_stream()), "to")
    assert torch.equal(tm.view(3, 32, 1536), x[:, :, 0, :].permute(0, 2, 1))
    assert torch.equal(tb, tm.to(torch.bfloat16))
    back = torch.empty_like(x)
    lib.check(lib.load().syn_from_token_major(tm.data_ptr(), 3, back.data_ptr(), lib.current_stream()), "from")
    assert torch.equal(back, x)


def test_randn_is_standard_normal_and_shard_invariant(lib):
    n = 1 << 20
    a = torch.empty(n, device="cuda")
    lib.check(lib.load().syn_randn(a.data_ptr(), n, 1234, 7, 0, lib.current_stream()), "randn")
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1) < 5e-3
    assert abs(float((a ** 4).mean()) - 3.0) < 0.05          # kurtosis of N(0,1)
    b = torch.empty(n // 2, device="cuda")                   # second half drawn on "another rank"
    lib.check(lib.load().syn_randn(b.data_ptr(), n // 2, 1234, 7, n // 2, lib.current_stream()), "randn")
    assert torch.equal(b, a[n // 2:])
    c = torch.empty(n, device="cuda")
    lib.check(lib.load().syn_randn(c.data_ptr(), n, 1234, 8, 0, lib.current_stream()), "randn")
    assert abs(float((a * c).mean())) < 5e-3                 # different step -> independent


def _philox4x32_10(idx4, stream_id, seed):
    """Philox4x32-10 (Salmon et al., SC'11) in numpy: counter = (idx4 lo, idx4 hi, stream lo, stream hi), key = seed."""
    import numpy as np
    c = [np.asarray(idx4 & 0xFFFFFFFF, np.uint64), np.asarray(idx4 >> 32, np.uint64),
         np.full(idx4.shape, stream_id & 0xFFFFFFFF, np.uint64), np.full(idx4.shape, stream_id >> 32, np.uint64)]
    k0, k1 = seed & 0xFFFFFFFF, seed >> 32
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c[0], np.uint64(0xCD9E8D57) * c[2]
        n0 = (p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)
        c = [n0, p1 & np.uint64(0xFFFFFFFF), n2, p0 & np.uint64(0xFFFFFFFF)]
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c


def test_randn_is_philox_box_muller(lib):
    """The generator's definition, pinned from outside: element 4q+j of (seed, stream) is Box-Muller of the Philox4x32-10
    block with counter (q, stream) - cos/sin of word pairs (0,1) and (2,3).  The device uses the hardware log2 / sqrt /
    sin / cos and fp32 uniforms, so the comparison is numeric (median error 6e-8), not bitwise."""
    import numpy as np
    n, seed, stream, first = 1 << 16, 0x1234567890ABCDEF, 917, 4096
    a = torch.empty(n, device="cuda")
    lib.check(lib.load().syn_randn(a.data_ptr(), n, seed, stream, first, lib.current_stream()), "randn")
    w = _philox4x32_10(np.arange(first // 4, (first + n) // 4, dtype=np.uint64), stream, seed)
    u = [(x.astype(np.float64) + 0.5) * 2.0 ** -32 for x in w]
    want = np.empty((n // 4, 4))
    for p in range(2):
        rad = np.sqrt(-2.0 * np.log(u[2 * p]))
        want[:, 2 * p], want[:, 2 * p + 1] = rad * np.cos(2 * np.pi * u[2 * p + 1]), rad * np.sin(2 * np.pi * u[2 * p + 1])
    got = a.cpu().double().numpy().reshape(-1, 4)
    d = np.abs(got - want)
    assert np.median(d) < 1e-6 and d.max() < 1e-4        # worst case: u1 within 1e-6 of 1, where fp32 u1 loses -ln(u1)'s digits


@pytest.mark.parametrize("B,L", [(3, 68224), (2, 68266), (1, 20000)])
def test_wav_encoder_vs_torch(B, L):
    """the HIP WavEncoder (channels-last bf16 implicit-GEMM convs, BN folded) against the fp32 PyTorch
    convolutions of the same folded weights on the CPU (models/denoiser.py:304-322).  bf16 activations through
    12 convolutions: rel-L2 <= 2e-2 (same budget as one denoiser evaluation)."""
    from syntalker_amd import conditioning, synth
    from tests.refmodel import synth_state_dict
    from tests.conftest import rel_l2
    sd = synth_state_dict("beatx")
    blocks = conditioning.fold_wav_encoder(sd)
    g = torch.Generator().manual_seed(L)
    wav = torch.randn(B, L, 2, generator=g)
    want = refmodel.wav_features(blocks, wav)
    enc = conditioning.HipWavEncoder(blocks, torch.device("cuda"))
    got = enc(wav.to("cuda")).cpu()
    assert got.shape == want.shape
    e = rel_l2(got, want)
    print(f"wav encoder B={B} L={L}: out {tuple(got.shape)} rel-L2 {e:.3e}")
    assert e < 2e-2
    assert torch.equal(enc(wav.to("cuda")).cpu(), got)          # deterministic, workspace halos intact on reuse
    # ... and against the ORACLE's encoder (un-folded convolutions + eval BatchNorm as the reference writes them,
    # oracle/denoiser_ref.py:wav_encoder, pinned to the imported reference by tests/test_oracle_golden.py)
    from oracle import denoiser_ref as dr
    with torch.no_grad():
        ref = dr.wav_encoder(sd, wav)
    assert rel_l2(want, ref) < 1e-5                              # the BatchNorm fold itself
    assert rel_l2(got, ref) < 2e-2


def test_training_block_ops_vs_torch_autograd():
    """fp32 forward / backward kernels of the training path (LayerNorm, GELU, 32-token attention) against PyTorch
    autograd on the same inputs (fp32 vs fp32: 1e-5)."""
    import torch.nn.functional as F
    from syntalker_amd import training
    dev = "cuda"
    g = torch.Generator().manual_seed(3)
    # LayerNorm
    x = (torch.randn(200, 512, generator=g) * 2 + 0.5).to(dev).requires_grad_()
    w, b = torch.randn(512, generator=g).to(dev).requires_grad_(), torch.randn(512, generator=g).to(dev).requires_grad_()
    up = torch.randn(200, 512, generator=g).to(dev)
    y = block_ops.HipLayerNormFn.apply(x, w, b)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), up)
    yr = F.layer_norm(x, (512,), w, b, 1e-5)
    rx, rw, rb = torch.autograd.grad(yr, (x, w, b), up)
    for a_, b_ in ((y, yr), (gx, rx), (gw, rw), (gb, rb)):
        assert rel_l2(a_.detach().cpu(), b_.detach().cpu()) < 1e-5
    # GELU
    x = (torch.randn(64, 1024, generator=g) * 3).to(dev).requires_grad_()
    up = torch.randn(64, 1024, generator=g).to(dev)
    y = block_ops.HipGeluFn.apply(x)
    gx, = torch.autograd.grad(y, x, up)
    yr = F.gelu(x)
    rx, = torch.autograd.grad(yr, x, up)
    assert rel_l2(y.detach().cpu(), yr.detach().cpu()) < 1e-6 and rel_l2(gx.cpu(), rx.cpu()) < 1e-5
    # attention on the packed qkv tensor
    qkv = torch.randn(5, 32, 1536, generator=g).to(dev).requires_grad_()
    up = torch.randn(5, 32, 512, generator=g).to(dev)
    o = block_ops.HipAttentionFn.apply(qkv)
    gq, = torch.autograd.grad(o, qkv, up)
    t = qkv.reshape(5, 32, 3, 4, 128).permute(2, 0, 3, 1, 4)
    orf = F.scaled_dot_product_attention(t[0], t[1], t[2]).transpose(1, 2).reshape(5, 32, 512)
    rq, = torch.autograd.grad(orf, qkv, up)
    assert rel_l2(o.detach().cpu(), orf.detach().cpu()) < 1e-5 and rel_l2(gq.cpu(), rq.cpu()) < 1e-5


def test_fused_residual_branches_equal_the_op_by_op_composition():
    """training.AttnBranchFn / MlpBranchFn (a pre-LN residual branch as one autograd node: bf16 rows straight from the LayerNorm /
    attention / GELU kernels, `x + DropPath factor * (.)` in the last GEMM's epilogue, the factor applied to dy by the backward's prep
    pass, bias gradients summed by the GEMM-pair launch) against the same branch built from the single ops with PyTorch glue: the
    same kernels on the same bf16 operands - outputs and all gradients agree to fp32 rounding of the epilogue (1e-6), with and without
    DropPath factors, at 128 rows (one row tile round) and 1024 (the bench's)."""
    import torch.nn as nn
    from syntalker_amd import training
    dev = "cuda"
    for B in (4, 32):
        torch.manual_seed(B)
        n1, n2 = nn.LayerNorm(512).to(dev), nn.LayerNorm(512).to(dev)
        qkv, proj = nn.Linear(512, 1536, bias=False).to(dev), nn.Linear(512, 512).to(dev)
        fc1, fc2 = nn.Linear(512, 1024).to(dev), nn.Linear(1024, 512).to(dev)
        with torch.no_grad():
            for ln in (n1, n2):
                ln.weight.uniform_(0.5, 1.5); ln.bias.normal_(0, 0.2)
        layers = (qkv, proj, fc1, fc2)
        params = [p for mod in (n1, n2) + layers for p in mod.parameters()]
        packs = training.WeightPacks([l.weight for l in layers])
        packs.refresh()
        training._packs = packs
        try:
            h0 = torch.randn(B, 32, 512, device=dev)
            up = torch.randn(B, 32, 512, device=dev)
            for factors in (None, torch.empty(2, B, 1, 1, device=dev).bernoulli_(0.7).div_(0.7)):
                f = (None, None) if factors is None else (factors[0], factors[1])
                res = {}
                for fused in (True, False):
                    h = h0.clone().requires_grad_()
                    if fused:
                        a = training.AttnBranchFn.apply(h, n1.weight, n1.bias, qkv.weight, qkv.bias, proj.weight, proj.bias, f[0])
                        out = training.MlpBranchFn.apply(a, n2.weight, n2.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias, f[1])
                    else:
                        z, r = block_ops.HipLnForkFn.apply(h, n1.weight, n1.bias)
                        br = training.lin(block_ops.HipAttentionFn.apply(training.lin(z, qkv)), proj)
                        a = r + br if f[0] is None else torch.addcmul(r, br, f[0])
                        z, r = block_ops.HipLnForkFn.apply(a, n2.weight, n2.bias)
                        br = training.lin(block_ops.HipGeluFn.apply(training.lin(z, fc1)), fc2)
                        out = r + br if f[1] is None else torch.addcmul(r, br, f[1])
                    res[fused] = [out.detach()] + list(torch.autograd.grad(out, [h] + params, up))
                names = ["out", "dh"] + [f"{n}.{k}" for n, mod in zip(("n1", "n2", "qkv", "proj", "fc1", "fc2"), (n1, n2) + layers)
                                         for k, _ in mod.named_parameters()]
                for name, a_, b_ in zip(names, res[True], res[False]):
                    assert torch.isfinite(a_).all() and rel_l2(a_.cpu(), b_.cpu()) < 1e-6, (B, factors is not None, name, rel_l2(a_.cpu(), b_.cpu()))
                if factors is not None:                    # a dropped sample's branch contributes nothing: out == h there
                    dropped = (factors[0].view(-1) == 0) & (factors[1].view(-1) == 0)
                    if dropped.any():
                        assert torch.equal(res[True][0][dropped], h0[dropped])
        finally:
            training._packs = None


def test_clip_adam_equals_clip_grad_norm_plus_torch_adam():
    """training.ClipAdam (syn_opt_sqnorm / syn_opt_scalars / syn_opt_adam) against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam on the
    same parameters and gradients over five steps: tensors of awkward sizes (1 element, not a multiple of 4, several 8192-element
    chunks, a misaligned view), 70 tensors (two argument lists), with the clip active and inactive, weight decay, a device-resident
    learning rate changed between steps; then torch's state_dict loaded into ClipAdam and the other way round."""
    from syntalker_amd import training
    dev = "cuda"
    g = torch.Generator().manual_seed(11)
    sizes = [(1,), (3,), (5, 7), (8192,), (8193,), (300, 129), (70001,)] + [(17 + i, 3) for i in range(63)]
    base = [torch.randn(*sz, generator=g) * 0.3 for sz in sizes]
    flat = torch.randn(1000 + 3, generator=g)
    for max_norm, wd in ((0.99, 0.0), (1e9, 0.01), (0.05, 0.01)):
        mine = [torch.nn.Parameter(b.clone().to(dev)) for b in base]
        ref = [torch.nn.Parameter(b.clone().to(dev)) for b in base]
        # a parameter whose storage starts 4 bytes off a 16-byte boundary
        buf_m, buf_r = flat.clone().to(dev), flat.clone().to(dev)
        mine.append(torch.nn.Parameter(buf_m[1:1001])); ref.append(torch.nn.Parameter(buf_r[1:1001]))
        lr_m, lr_r = torch.tensor(3e-3, device=dev), torch.tensor(3e-3, device=dev)
        om = training.ClipAdam(mine, lr=lr_m, betas=(0.5, 0.999), weight_decay=wd, max_norm=max_norm)
        orf = torch.optim.Adam(ref, lr=lr_r, betas=(0.5, 0.999), weight_decay=wd, capturable=True, foreach=True)
        for it in range(5):
            gg = torch.Generator().manual_seed(100 + it)
            for a, b in zip(mine, ref):
                gr = torch.randn(a.shape, generator=gg).to(dev) * (10.0 if it == 2 else 0.01)       # (step 2 is far above the clip threshold)
                a.grad, b.grad = gr.clone(), gr.clone()
            want_norm = torch.nn.utils.clip_grad_norm_(ref, max_norm)
            orf.step(); om.step()
            assert abs(float(om.last_norm()) / float(want_norm) - 1) < 1e-5
            if it == 2:
                lr_m.fill_(1e-3); lr_r.fill_(1e-3)
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(mine, ref)):
            assert torch.isfinite(a).all() and rel_l2(a.detach().cpu(), b.detach().cpu()) < 2e-6, (max_norm, wd, i, rel_l2(a.detach().cpu(), b.detach().cpu()))
            # (the second moment carries the clip factor squared; the factor itself differs in its last bits - the total norm is summed in
            #  double from 8192-element partial sums here, in fp32 from per-tensor norms there)
            assert rel_l2(om.state[a]["exp_avg_sq"].cpu(), orf.state[b]["exp_avg_sq"].cpu()) < 5e-5
        assert float(om.state[mine[0]]["step"]) == 5.0
    # checkpoints travel both ways (same state layout)
    import copy
    sd = orf.state_dict()
    om2 = training.ClipAdam(mine, lr=1e-3, betas=(0.5, 0.999), max_norm=0.99)
    om2.load_state_dict(copy.deepcopy(sd))             # (Optimizer.load_state_dict keeps same-dtype tensors by reference)
    for a in mine:
        a.grad = torch.ones_like(a) * 1e-3
    om2.step()
    assert float(om2.state[mine[0]]["step"]) == 6.0 and om2.state[mine[3]]["step"] is om2.state[mine[0]]["step"]
    # a second load replaces the moment tensors under an optimizer that has already stepped: the next step must write the NEW ones
    om2.load_state_dict(copy.deepcopy(sd))
    before = om2.state[mine[5]]["exp_avg"].clone()
    for a in mine:
        a.grad = torch.ones_like(a)
    om2.step()
    torch.cuda.synchronize()
    assert not torch.equal(om2.state[mine[5]]["exp_avg"], before) and float(om2.state[mine[0]]["step"]) == 6.0
    orf2 = torch.optim.Adam(ref, lr=1e-3, betas=(0.5, 0.999), capturable=True, foreach=True)
    orf2.load_state_dict(om.state_dict())
    assert float(orf2.state[ref[0]]["step"]) == 5.0


def test_wav_encoder_single_channel():
    """audio_rep variants with one waveform channel (models/denoiser.py:64-67): the first layer has cin = 1."""
    from syntalker_amd import conditioning
    from syntalker_amd.denoiser import _WavEncoder
    torch.manual_seed(5)
    enc = _WavEncoder(256, 1).eval()
    with torch.no_grad():
        for mod in enc.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.8, 1.2); mod.bias.normal_(0, 0.1)
    sd = {"WavEncoder." + k: v for k, v in enc.state_dict().items()}
    blocks = conditioning.fold_wav_encoder(sd)
    wav = torch.randn(2, 30000, generator=torch.Generator().manual_seed(6))
    want = refmodel.wav_features(blocks, wav)
    got = conditioning.HipWavEncoder(blocks, torch.device("cuda"))(wav.to("cuda")).cpu()
    assert got.shape == want.shape and rel_l2(got, want) < 2e-2


def test_q_sample_on_the_device_vs_oracle():
    """q_sample (gaussian_diffusion.py:235-253) as the training step issues it - one syn_axpby_rows launch with
    device-resident coefficient tables - against the oracle's fp64-table restatement."""
    from oracle.process_ref import RefProcess
    from syntalker_amd import synth
    from syntalker_amd.process import create_gaussian_diffusion
    from tests.conftest import rel_l2
    d = create_gaussian_diffusion()
    B = 7
    x0, eps = synth.synth_latent(B, seed=3, name="x0"), synth.synth_latent(B, seed=4, name="eps")
    t = torch.tensor([0, 1, 17, 250, 500, 998, 999])
    got = d.q_sample(x0.cuda(), t.cuda(), noise=eps.cuda())
    want = RefProcess(False).q_sample(x0, t, eps)
    assert got.is_cuda and rel_l2(got.cpu(), want) < 1e-6
    # the autograd-visible form (a tensor that requires grad) takes the torch expression and agrees with the kernel
    xg = x0.cuda().requires_grad_(True)
    assert rel_l2(d.q_sample(xg, t.cuda(), noise=eps.cuda()).detach().cpu(), got.cpu()) < 1e-6


@pytest.mark.parametrize("words,loaded", [(64, False), (1024, True), (4096, True)])
def test_group_barrier_acquire_under_adversarial_reuse(lib, words, loaded):
    """What the members of an XCD group (small-batch kernel k_lat, tile-split mode of k_stack) must do to read each other's
    data, in the most hostile form: a consumer CU that keeps the exchanged words L1-resident (the same <= 16 KB re-read every
    round), a new payload every round, every word checked, the rest of the XCD streaming from HBM.
      * plain loads are stale on this chip - and `buffer_inv sc0`, which round 1's barrier used, changes nothing (the check is
        sharp: it would catch a broken acquire);
      * the agent-scope `buffer_inv sc1` works but costs ~14 us per barrier in k_lat;
      * loads that bypass the L1 (non-temporal: `ld_xcd` in syn_latency.inc; or sc1 through a buffer descriptor) never read a
        stale word and need no cache maintenance at all: that is what the kernels do."""
    from syntalker_amd import _lib
    dev = "cuda"
    stream = torch.randn(64 << 20, device=dev) if loaded else None
    rounds = 20000

    def run(mode):
        sync = torch.zeros(320, dtype=torch.int32, device=dev)
        buf = torch.zeros(8, 4096, dtype=torch.int32, device=dev)
        stale = torch.zeros(9, dtype=torch.int32, device=dev)
        _lib.check(_lib.load().syn_test_handoff(sync.data_ptr(), buf.data_ptr(), _lib.ptr(stream), 0 if stream is None else stream.numel(),
                                                stale.data_ptr(), words, rounds, mode, _lib.current_stream()), "syn_test_handoff")
        torch.cuda.synchronize()
        assert int(sync.view(10, 32)[:8, 2].min()) == rounds, "a producer / consumer pair did not finish (spin limit)"
        return int(stale[:8].sum())

    res = {name: run(mode) for mode, name in enumerate(("plain loads", "buffer_inv sc0", "buffer_inv sc1", "non-temporal loads", "sc1 buffer loads"))}
    print(f"stale words over {rounds} rounds x 8 XCDs x {words} words: {res}")
    assert res["non-temporal loads"] == 0 and res["sc1 buffer loads"] == 0 and res["buffer_inv sc1"] == 0
    assert res["plain loads"] > 0, "no staleness with plain loads: this test would not catch a broken hand-off"


@pytest.mark.parametrize("variant", ["beatx", "h3d"])
def test_conditioning_kernels_vs_oracle(variant):
    """everything behind the audio encoder (word embedding gather, 300->256, concat, mix 512->256, avg-pool 4,
    . W2c^T, seed / style terms, all biases; models/denoiser.py:147-157,160-174) as two fp32 HIP launches (`syn_cond_encode`),
    (a) fed the oracle's fp32 audio features: <= 1e-5 vs oracle/denoiser_ref.clip_conditioning;
    (b) the whole per-clip conditioning (HIP bf16 audio encoder in front): <= 6e-3, and nothing but HIP launches."""
    import ctypes as C
    from oracle import denoiser_ref as dr
    from syntalker_amd import _lib, conditioning, synth
    from tests.refmodel import synth_state_dict
    sd = synth_state_dict(variant)
    style = variant == "h3d"
    B = 19                                                       # ragged: clip tiles of 16 in the seed GEMM
    y = synth.synth_clip_inputs(B, seed=8, style_dim=256 if style else 512, style_zero=not style)
    fo = dr.fold_weights(sd, variant)
    sd_dev = {k: v.cuda() for k, v in sd.items()}
    fw = conditioning.fold_input_stage(sd_dev, style)
    cc = conditioning.ClipConditioner(sd_dev, fw, variant, style)
    yd = synth.to_device(y, "cuda")
    for flags in ((False, False), (True, True)):
        yy = dict(y, uncond=flags[0], uncond_audio=flags[1])
        with torch.no_grad():
            want = dr.clip_conditioning(sd, yy, fo, variant)
            audio, word = cc.audio_word_of(y, flags[1])
            feat = dr.wav_encoder(sd, audio).contiguous().cuda()
        st = cc.style_of(yd, flags[0], B)
        st = None if st is None else st.float().contiguous()     # (operands stay referenced until the launch is enqueued)
        word_d, seed_d = word.cuda().contiguous(), yd["seed"].reshape(B, -1).contiguous()
        out, d = torch.empty(B, 32, 512, device="cuda"), torch.empty(8, B, 512, device="cuda")   # SYN_COND_SCRATCH_ROWS partial sums per clip
        _lib.check(_lib.load().syn_cond_encode(C.byref(cc.weights.c_struct()), feat.data_ptr(), word_d.data_ptr(), seed_d.data_ptr(),
                                               _lib.ptr(st), B, d.data_ptr(), out.data_ptr(), _lib.current_stream()), "syn_cond_encode")
        e = rel_l2(out.cpu(), want)
        print(f"{variant} {flags}: syn_cond_encode on fp32 features rel-L2 {e:.2e}")
        assert e < 1e-5
        full = cc.cond(yd, *flags).cpu()
        e = rel_l2(full, want)
        print(f"{variant} {flags}: whole per-clip conditioning rel-L2 {e:.2e}")
        assert e < 6e-3


@pytest.mark.parametrize("cin,stride,cout,pad,L", [(64, 1, 64, 7, 300), (128, 1, 128, 7, 133), (256, 1, 256, 7, 64), (64, 6, 64, 0, 1001),
                                                  (64, 6, 128, 0, 379), (128, 3, 256, 0, 200)])
def test_training_conv_forward_on_split_operands_vs_fp32(cin, stride, cout, pad, L):
    """syn_conv1d_train_fwd (the WavEncoder's Conv1d(k = 15) layers in training mode: hi + lo bf16 operands, three MFMAs per
    product) against PyTorch's fp32 convolution in float64: fp32-grade (a plain bf16 forward is at 3e-3), on ragged lengths;
    and its autograd wrapper returns fp32 gradients for both inputs."""
    from syntalker_amd import training
    g = torch.Generator().manual_seed(cin + stride)
    DEV = "cuda"
    x = torch.randn(3, cin, 1, L, generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 15, generator=g) / (cin * 15) ** 0.5).to(DEV).requires_grad_(True)
    y = training.ConvSplitFn.apply(x, w, stride, pad)
    want = torch.nn.functional.conv2d(x.detach().double(), w.detach().double(), None, stride=(1, stride), padding=(0, pad))
    assert y.shape == want.shape
    e = rel_l2(y.detach().double().cpu(), want.cpu())
    print(f"conv {cin}x{stride}->{cout}: rel-L2 vs float64 {e:.2e}")
    assert e < 2e-5
    gy = torch.randn(y.shape, generator=g).to(DEV)
    y.backward(gy)
    x2, w2 = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    torch.nn.functional.conv2d(x2, w2, None, stride=(1, stride), padding=(0, pad)).backward(gy)
    # data gradient: both cross products (fp32-grade).  Weight gradient, the library's default: x read rounded to bf16 (one cross product dropped,
    # round 5) - 2^-9 per element, averaged down by the sum over positions in the real layers; with all three products (syn_debug_conv_terms(3)) fp32-grade
    ew = rel_l2(w.grad.cpu(), w2.grad.cpu())
    print(f"conv {cin}x{stride}->{cout}: weight gradient rel-L2 {ew:.2e} (default: two of three products)")
    assert rel_l2(x.grad.cpu(), x2.grad.cpu()) < 1e-5 and ew < 4e-3
    from syntalker_amd import _lib
    lib = _lib.load()
    try:
        lib.syn_debug_conv_terms(3)
        x3, w3 = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
        training.ConvSplitFn.apply(x3, w3, stride, pad).backward(gy)
        assert rel_l2(w3.grad.cpu(), w2.grad.cpu()) < 1e-5 and rel_l2(x3.grad.cpu(), x2.grad.cpu()) < 1e-5
    finally:
        lib.syn_debug_conv_terms(-1)


def test_embedding_gradient_vs_torch(lib):
    """syn_embedding_wgrad (the word-embedding table's gradient: a memset node + one deterministic launch) against
    torch.nn.functional.embedding's autograd: repeated ids, untouched rows exactly zero, dy read as a column slice of a wider tensor (pitch 384:
    text_encoder_body's padded data gradient), bitwise run-to-run; and syn_embed_rows_bf16 = bf16(table[ids]) with zero padding."""
    DEV = "cuda"
    g = torch.Generator().manual_seed(11)
    V, D, LD = 11195, 300, 384
    w = torch.randn(V, D, generator=g).to(DEV)
    ids = torch.randint(0, V, (32, 128), generator=g)
    ids[:, :40] = torch.randint(0, 50, (32, 40), generator=g)         # heavy repeats
    ids[0, 0], ids[0, 1] = 0, V - 1
    ids = ids.to(DEV).reshape(-1).contiguous()
    n = ids.numel()
    dyw = torch.randn(n, LD, generator=g).to(DEV)
    L, st = lib.load(), lib.current_stream()
    outs = []
    for _ in range(2):
        dw = torch.full((V, D), float("nan"), device=DEV)
        lib.check(L.syn_embedding_wgrad(ids.data_ptr(), dyw.data_ptr(), LD, n, V, D, dw.data_ptr(), st), "syn_embedding_wgrad")
        outs.append(dw)
    assert torch.equal(*outs)
    w2 = w.clone().requires_grad_(True)
    torch.nn.functional.embedding(ids, w2).backward(dyw[:, :D].contiguous())
    assert rel_l2(outs[0].cpu(), w2.grad.cpu()) < 1e-6
    untouched = torch.ones(V, dtype=torch.bool); untouched[ids.cpu()] = False
    assert float(outs[0].cpu()[untouched].abs().max()) == 0.0
    E = torch.full((n, LD), 7.0, dtype=torch.bfloat16, device=DEV)
    lib.check(L.syn_embed_rows_bf16(ids.data_ptr(), w.data_ptr(), V, D, n, LD, E.data_ptr(), st), "syn_embed_rows_bf16")
    assert torch.equal(E[:, :D], w[ids].bfloat16()) and float(E[:, D:].float().abs().max()) == 0.0


def test_glue_kernels_vs_torch(lib):
    """csrc/syn_glue.inc against the PyTorch ops they stand for: the concatenated bf16 operand (per-clip vectors repeated over the frames, a sum of two,
    an average pool over 4 rows, zero padding), the (B, C, 1, T) -> rows transpose, group sums, the pool's backward."""
    DEV = "cuda"
    g = torch.Generator().manual_seed(23)
    L, st = lib.load(), lib.current_stream()
    B, T, D, A = 8, 32, 512, 256
    M = B * T
    es, et = torch.randn(B, D, generator=g).to(DEV), torch.randn(B, D, generator=g).to(DEV)
    x_ = torch.randn(M, D, generator=g).to(DEV)
    at4 = torch.randn(M * 4, A, generator=g).to(DEV)
    arr = (lib.SynConcatSrc * 3)()
    for i, (t, t2, width, ld, div, pool) in enumerate(((es, et, D, D, T, 1), (x_, None, D, D, 1, 1), (at4, None, A, A, 1, 4))):
        arr[i].p, arr[i].p2, arr[i].width, arr[i].ld, arr[i].row_div, arr[i].pool = t.data_ptr(), lib.ptr(t2), width, ld, div, pool
    out_ld = 1408                                                     # 1280 columns + padding
    out = torch.full((M, out_ld), 7.0, dtype=torch.bfloat16, device=DEV)
    lib.check(L.syn_rows_concat_bf16(arr, 3, M, out_ld, out.data_ptr(), st), "syn_rows_concat_bf16")
    want = torch.cat([(es + et).repeat_interleave(T, 0), x_, at4.view(M, 4, A).mean(1)], 1)
    assert rel_l2(out[:, :1280].float().cpu(), want.bfloat16().float().cpu()) < 1e-3 and float(out[:, 1280:].float().abs().max()) == 0.0
    assert torch.equal(out[:, :1024], want[:, :1024].bfloat16())       # (no arithmetic but one add: exact)
    x = torch.randn(B, 1536, 1, T, generator=g).to(DEV)
    xr = torch.empty(M, 1536, dtype=torch.bfloat16, device=DEV)
    lib.check(L.syn_bct_to_rows_bf16(x.data_ptr(), B, 1536, T, xr.data_ptr(), st), "syn_bct_to_rows_bf16")
    assert torch.equal(xr.view(B, T, 1536), x.reshape(B, 1536, T).transpose(1, 2).bfloat16())
    wide = torch.randn(M, 1280, generator=g).to(DEV)
    gs = torch.empty(B, D, device=DEV)
    lib.check(L.syn_rows_group_sum(wide.data_ptr() + 4 * 512, 1280, D, T, B, gs.data_ptr(), st), "syn_rows_group_sum")
    assert rel_l2(gs.cpu(), wide[:, 512:1024].reshape(B, T, D).sum(1).cpu()) < 1e-6
    ex = torch.empty(M * 4, A, device=DEV)
    lib.check(L.syn_rows_expand(wide.data_ptr() + 4 * 1024, 1280, A, 4, 0.25, M * 4, ex.data_ptr(), st), "syn_rows_expand")
    assert torch.equal(ex, wide[:, 1024:1280].repeat_interleave(4, 0) * 0.25)
    lib.check(L.syn_touch(wide.data_ptr(), wide.numel() * 4, st), "syn_touch")
    torch.cuda.synchronize()


def test_step_weight_packs_and_linear_node():
    """syn_pack_weights (every Linear weight and its transpose as bf16 fragments, one launch; an input width that is not a multiple of 128 zero-padded)
    against the per-use packers, bitwise; a weight updated in place falls out of the cache until the next refresh; `HipLinearFn` on those packs and on
    packs made on the spot: the same bits, run-to-run identical, against torch on the bf16-rounded operands - at the shapes of the step (1024 rows),
    of input_process2 (1280 inputs), of text_encoder_body (300 inputs, 4096 rows), and at a row count that is not a multiple of 128."""
    from syntalker_amd import training, engine
    import torch.nn.functional as F
    DEV = "cuda"
    g = torch.Generator().manual_seed(5)
    ws = [torch.randn(n, k, generator=g).to(DEV) for n, k in ((512, 512), (1536, 512), (512, 2048), (1024, 512), (256, 300), (512, 1280))]
    pk = training.WeightPacks(ws)
    pk.refresh()
    for w in ws:
        N, K = w.shape
        Kp = (K + 127) // 128 * 128
        wpad = F.pad(w, (0, Kp - K)).contiguous()
        fwd, tr = pk.lookup(w)
        assert torch.equal(fwd, engine.pack_weight(wpad).view(torch.uint8).reshape(-1)), (N, K)
        assert torch.equal(tr, training._pack_t(wpad, Kp, N)), (N, K)
    ws[0].add_(1.0)
    assert pk.lookup(ws[0]) == (None, None)                        # stale until the next refresh
    pk.refresh()
    assert torch.equal(pk.lookup(ws[0])[0], engine.pack_weight(ws[0]).view(torch.uint8).reshape(-1))
    for M, N, K in ((1024, 512, 512), (128, 1536, 512), (1024, 512, 2048), (1024, 512, 1280), (4096, 256, 300), (160, 512, 512)):
        w = next(t for t in ws if t.shape == (N, K)).requires_grad_(True)
        b = torch.randn(N, generator=g).to(DEV).requires_grad_(True)
        x = torch.randn(M, K, generator=g).to(DEV).requires_grad_(True)
        dy = torch.randn(M, N, generator=g).to(DEV)
        grads = []
        for cached in (True, True, False):
            training._packs = pk if cached else None
            try:
                w.grad = b.grad = x.grad = None
                y = training.HipLinearFn.apply(x, w, b)
                y.backward(dy)
                grads.append((y.detach().clone(), b.grad.clone(), x.grad.clone(), w.grad.clone()))
            finally:
                training._packs = None
        assert all(torch.equal(p, q) for p, q in zip(grads[0], grads[1])) and all(torch.equal(p, q) for p, q in zip(grads[0], grads[2])), (M, N, K)
        xb, wb, dyb = x.detach().bfloat16().float(), w.detach().bfloat16().float(), dy.bfloat16().float()
        assert rel_l2(grads[0][0].cpu(), (xb @ wb.t() + b.detach()).cpu()) < 1e-5
        assert rel_l2(grads[0][1].cpu(), dy.sum(0).cpu()) < 1e-5
        assert rel_l2(grads[0][2].cpu(), (dyb @ wb).cpu()) < 1e-5 and rel_l2(grads[0][3].cpu(), (dyb.t() @ xb).cpu()) < 1e-5, (M, N, K)
        w.requires_grad_(False)


@pytest.mark.parametrize("cin,stride,pad,L", [(1, 5, 1700, 6001), (2, 5, 1700, 12345), (2, 5, 0, 644), (1, 3, 4, 77)])
def test_training_first_layer_conv_vs_fp64(cin, stride, pad, L):
    """syn_conv1d_first_fwd / _wgrad (block 0's conv1 and shortcut convolution: Conv1d(1 | 2 -> 64, k 15, stride 5, padding 1700),
    models/denoiser.py:308, on the waveform layout (N, L, cin)) against PyTorch's convolution in float64: plain fp32 FMAs, so
    fp32 rounding only, on ragged lengths and several position chunks per clip; no gradient for the waveform."""
    from syntalker_amd import training
    g = torch.Generator().manual_seed(cin + L)
    DEV = "cuda"
    wav = torch.randn(3, L, cin, generator=g).to(DEV)
    w = (torch.randn(64, cin, 15, generator=g) / (cin * 15) ** 0.5).to(DEV).requires_grad_(True)
    y = training.ConvFirstFn.apply(wav, w, stride, pad)
    x4 = wav.permute(0, 2, 1).unsqueeze(2)                                      # (N, cin, 1, L)
    w2 = w.detach().double().clone().requires_grad_(True)
    want = torch.nn.functional.conv2d(x4.double(), w2.unsqueeze(2), None, stride=(1, stride), padding=(0, pad))
    assert y.shape == want.shape and y.is_contiguous(memory_format=torch.channels_last)
    e = rel_l2(y.detach().double().cpu(), want.detach().cpu())
    gy = torch.randn(y.shape, generator=g).to(DEV)
    y.backward(gy)
    want.backward(gy.double())
    eg = rel_l2(w.grad.double().cpu(), w2.grad.cpu())
    print(f"first layer cin {cin} stride {stride} L {L}: forward {e:.2e}, weight gradient {eg:.2e} vs float64")
    assert e < 1e-6 and eg < 1e-5


@pytest.mark.parametrize("C,L,short,act", [(64, 333, False, True), (128, 77, True, True), (256, 40, False, False), (64, 1500, True, True)])
def test_fused_batchnorm_shortcut_leakyrelu_vs_torch_autograd(C, L, short, act):
    """syn_bn_act_fwd / _bwd (training-mode BatchNorm1d on batch statistics [+ shortcut] [+ LeakyReLU(0.01)], the tail of every
    convolution of the audio encoder's BasicBlock) against PyTorch's batch_norm + add + leaky_relu and their autograd:
    output, running statistics (the convolution's bias enters only there), and the gradients of y, gamma, beta, shortcut."""
    from syntalker_amd import training
    DEV = "cuda"
    g = torch.Generator().manual_seed(C + L)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    y = (mk(3, C, 1, L) * 2 + 0.5).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sc = mk(3, C, 1, L).contiguous(memory_format=torch.channels_last).requires_grad_(True) if short else None
    gamma, beta, cb = (mk(C) * 0.5 + 1).requires_grad_(True), mk(C).requires_grad_(True), mk(C).requires_grad_(True)
    rm, rv = mk(C), mk(C).abs() + 0.5
    rm2, rv2 = rm.clone(), rv.clone()
    z = training.BnActFn.apply(y, gamma, beta, cb, sc, rm, rv, 0.1, 1e-5, act)
    dz = mk(*z.shape)
    z.backward(dz)
    y2, g2, b2, cb2 = (t.detach().clone().requires_grad_(True) for t in (y, gamma, beta, cb))
    sc2 = sc.detach().clone().requires_grad_(True) if short else None
    p = torch.nn.functional.batch_norm(y2 + cb2.view(1, -1, 1, 1), rm2, rv2, g2, b2, True, 0.1, 1e-5)
    if short:
        p = p + sc2
    want = torch.nn.functional.leaky_relu(p, 0.01) if act else p
    want.backward(dz)
    assert rel_l2(z.detach().cpu(), want.detach().cpu()) < 1e-5
    assert torch.allclose(rm, rm2, rtol=1e-5, atol=1e-6) and torch.allclose(rv, rv2, rtol=1e-4, atol=1e-6)
    assert rel_l2(y.grad.cpu(), y2.grad.cpu()) < 1e-4
    assert rel_l2(gamma.grad.cpu(), g2.grad.cpu()) < 1e-4 and rel_l2(beta.grad.cpu(), b2.grad.cpu()) < 1e-4
    assert float(cb.grad.abs().max()) == 0.0 and float(cb2.grad.abs().max()) < 1e-3 * float(dz.abs().sum() / C)
    if short:
        assert rel_l2(sc.grad.cpu(), sc2.grad.cpu()) < 1e-6


@pytest.mark.parametrize("C,L,short,act", [(64, 700, False, True), (128, 333, True, True), (256, 130, False, False)])
def test_sync_batchnorm_over_two_emulated_ranks_vs_torch_autograd(C, L, short, act, monkeypatch):
    """`SyncBnActFn` (nn.SyncBatchNorm on the fused kernels, reference train.py:90): a batch of 4 split over two "ranks" of 3 and 1 clips
    - ragged on purpose - whose all-reduce is played by the test (first the ranks' local sums are recorded, then every rank is re-run
    with the totals handed to it) against PyTorch's batch_norm over the WHOLE batch and its autograd: every rank's outputs and data
    gradients are its slice of the full-batch ones, the weight / bias gradients are each rank's own contribution (they add up to the
    full-batch gradients; DDP would average them), the running statistics move towards the global batch statistics."""
    from syntalker_amd import training
    DEV = "cuda"
    g = torch.Generator().manual_seed(C + L)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    y = (mk(4, C, 1, L) * 2 + 0.5).contiguous(memory_format=torch.channels_last)
    sc = mk(4, C, 1, L).contiguous(memory_format=torch.channels_last) if short else None
    gamma, beta, cb = mk(C) * 0.5 + 1, mk(C), mk(C)
    rm0, rv0 = mk(C), mk(C).abs() + 0.5
    dz = mk(4, C, 1, L).contiguous(memory_format=torch.channels_last)
    # the whole batch through PyTorch
    y2, g2, b2 = (t.detach().clone().requires_grad_(True) for t in (y, gamma, beta))
    sc2 = sc.detach().clone().requires_grad_(True) if short else None
    rm2, rv2 = rm0.clone(), rv0.clone()
    p = torch.nn.functional.batch_norm(y2 + cb.view(1, -1, 1, 1), rm2, rv2, g2, b2, True, 0.1, 1e-5)
    if short:
        p = p + sc2
    want = torch.nn.functional.leaky_relu(p, 0.01) if act else p
    want.backward(dz)
    ranks = [slice(0, 3), slice(3, 4)]
    rec, state = {}, {"mode": "record", "rank": 0}

    def fake_all_reduce(t, group, tag=None):
        if state["mode"] == "record" or tag[1] not in state["replay"]:
            rec.setdefault(tag, {})[state["rank"]] = t.clone()
            return t
        return rec[tag][0] + rec[tag][1]
    monkeypatch.setattr(training, "_all_reduce_sum", fake_all_reduce)

    def run(r):
        state["rank"] = r
        sl = ranks[r]
        yy, gg, bb, cc = (t.detach().clone().requires_grad_(True) for t in (y[sl].contiguous(memory_format=torch.channels_last), gamma, beta, cb))
        ss = sc[sl].detach().clone().contiguous(memory_format=torch.channels_last).requires_grad_(True) if short else None
        rm, rv = rm0.clone(), rv0.clone()
        z = training.SyncBnActFn.apply(yy, gg, bb, cc, ss, rm, rv, 0.1, 1e-5, act, None, "op")
        z.backward(dz[sl].contiguous(memory_format=torch.channels_last))
        return z.detach(), yy.grad, gg.grad, bb.grad, (ss.grad if short else None), rm, rv, cc.grad
    for mode, replay in (("record", ()), ("replay", ("fwd",)), ("replay", ("fwd", "bwd"))):    # forward sums, then backward sums, then the real run
        state["mode"], state["replay"] = mode, replay
        outs = [run(0), run(1)]
    dg, db = outs[0][2] + outs[1][2], outs[0][3] + outs[1][3]
    for r, sl in enumerate(ranks):
        z, dy, _, _, dsh, rm, rv, dcb = outs[r]
        assert rel_l2(z.cpu(), want[sl].detach().cpu()) < 1e-5
        assert rel_l2(dy.cpu(), y2.grad[sl].cpu()) < 1e-4
        if short:
            assert rel_l2(dsh.cpu(), sc2.grad[sl].cpu()) < 1e-6
        assert torch.allclose(rm, rm2, rtol=1e-5, atol=1e-6) and torch.allclose(rv, rv2, rtol=1e-4, atol=1e-6)      # global statistics on every rank
        assert float(dcb.abs().max()) == 0.0
    assert rel_l2(dg.cpu(), g2.grad.cpu()) < 1e-4 and rel_l2(db.cpu(), b2.grad.cpu()) < 1e-4


def test_fused_rotary_equals_the_op_by_op_chain():
    """`syn_rotary` against the reference's own chain of views / cat / mul / add on the hidden state (models/denoiser.py:178-186, 324-343), forward,
    and its inverse launch against that chain's autograd gradient."""
    from syntalker_amd import synth, training
    from syntalker_amd.denoiser import MDM
    m = MDM(synth.default_args()).cuda()
    g = torch.Generator().manual_seed(3)
    h = torch.randn(5, 32, 512, generator=g).cuda().requires_grad_(True)
    w = torch.randn(5, 32, 512, generator=g).cuda()
    B, T, _ = h.shape
    gq = h.view(B, T, 8, -1).permute(0, 2, 1, 3).reshape(B * 8, T, -1)
    pos = torch.arange(T, device=h.device).type_as(m.rel_pos.inv_freq)
    fr = torch.einsum("i,j->ij", pos, m.rel_pos.inv_freq)
    fr = torch.cat((fr, fr), dim=-1)
    half = gq.shape[-1] // 2
    gq = gq * fr.cos() + torch.cat((-gq[..., half:], gq[..., :half]), dim=-1) * fr.sin()
    want = gq.reshape(B, 8, T, -1).permute(0, 2, 1, 3).reshape(B, T, -1)
    (want * w).sum().backward()
    cs, sn = training._rotary_tables(m, T, h.device)
    y = training._rotary_launch(h.detach().contiguous(), cs, sn, False)
    dh = training._rotary_launch(w.contiguous(), cs, sn, True)
    assert float((y - want.detach()).abs().max()) < 2e-6 and float((dh - h.grad).abs().max()) < 2e-6
    assert float((y.norm(dim=-1) - h.detach().norm(dim=-1)).abs().max()) < 1e-3          # a rotation


def test_small_row_linear_gradients_vs_torch():
    """`syn_linear_wgrad_rows` through HipLinearFn (the timestep MLP / embed_text case: one row per clip) against fp32 autograd on the same
    bf16-rounded input."""
    from syntalker_amd import training
    g = torch.Generator().manual_seed(4)
    for M, K, N in ((32, 512, 512), (4, 6144, 512), (64, 512, 1536), (3, 512, 512)):
        x = torch.randn(M, K, generator=g).cuda().requires_grad_(True)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().requires_grad_(True)
        b = torch.randn(N, generator=g).cuda().requires_grad_(True)
        dy = torch.randn(M, N, generator=g).cuda()
        y = training.HipLinearFn.apply(x, w, b)
        y.backward(dy)
        xr, wr = _bf(x.detach()).float(), _bf(w.detach()).float()
        assert rel_l2(y.detach(), xr @ wr.t() + b.detach()) < 1e-5
        assert rel_l2(w.grad, dy.t() @ xr) < 1e-6 and rel_l2(b.grad, dy.sum(0)) < 1e-6            # fp32 FMAs on the operand the forward took
        assert rel_l2(x.grad, _bf(dy).float() @ wr) < 1e-5


def test_fused_masked_smooth_l1_vs_torch():
    """`syn_masked_smooth_l1` / `_grad` against the reference's masked_l2 composition (gaussian_diffusion.py:202-215): `out` as a (B, C, 1, T) tensor
    and as the permuted view of [B][T][C] rows that the training forward returns (read and differentiated in place, no permuted copy)."""
    from syntalker_amd import training
    from syntalker_amd.process import create_gaussian_diffusion
    d = create_gaussian_diffusion()
    g = torch.Generator().manual_seed(5)
    a = torch.randn(6, 1536, 1, 32, generator=g).cuda()
    b0 = a + 1.5 * torch.randn(6, 1536, 1, 32, generator=g).cuda()                                 # both branches of the SmoothL1
    mask = (torch.rand(6, 1, 1, 32, generator=g) < 0.8).cuda()
    wgt = torch.rand(6, generator=g).cuda()
    b = b0.clone().requires_grad_(True)
    ref = torch.nn.functional.smooth_l1_loss(a, b, reduction="none") * mask.float()
    ref = ref.sum(dim=[1, 2, 3]) / (mask.sum(dim=[1, 2, 3]) * 1536)
    (ref * wgt).sum().backward()
    want = (ref.detach().clone(), b.grad.clone())
    for rows in (False, True):
        if rows:
            base = b0.reshape(6, 1536, 32).transpose(1, 2).contiguous().requires_grad_(True)      # [B][T][C]
            out = base.permute(0, 2, 1).unsqueeze(2)
            assert training._is_rows_view(out)
        else:
            base = out = b0.clone().requires_grad_(True)
        loss = d.masked_l2(a, out, mask)
        (loss * wgt).sum().backward()
        grad = base.grad.permute(0, 2, 1).unsqueeze(2) if rows else base.grad
        assert loss.shape == (6,) and rel_l2(loss.detach(), want[0]) < 1e-6 and rel_l2(grad, want[1]) < 1e-6, rows
        if rows:
            assert base.grad.is_contiguous()


@pytest.mark.parametrize("B,drop", [(4, False), (12, True), (32, True)])
def test_persistent_block_stack_forward_backward_vs_torch_fp32(B, drop):
    """`syn_train_stack_fwd` / `_bwd` / `_wgrad` (training.StackFn: the eight pre-LN blocks of the training forward as one persistent launch, their
    data-gradient chain as another, the 32 weight-gradient GEMMs four per launch) against a plain PyTorch fp32 restatement of the same blocks
    (models/timm_transformer/transformer.py:83-104, 145-151, 195-198; DropPath :21-38 with the same factors) and its autograd: output, the gradient
    of the input and of every parameter.  Tolerance = the sampling kernels' (bf16 GEMM operands, q / k / v / softmax numerators in bf16): 1.5e-2 on the
    output after eight blocks, 3e-2 per gradient tensor."""
    import torch.nn as nn
    import torch.nn.functional as F
    from syntalker_amd import training
    dev = "cuda"
    torch.manual_seed(100 + B)

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            self.norm1, self.norm2 = nn.LayerNorm(512), nn.LayerNorm(512)
            self.qkv, self.proj = nn.Linear(512, 1536, bias=False), nn.Linear(512, 512)
            self.fc1, self.fc2 = nn.Linear(512, 1024), nn.Linear(1024, 512)
    blocks = nn.ModuleList([Blk() for _ in range(8)]).to(dev)
    with torch.no_grad():
        for p in blocks.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))             # LayerNorm gains / shifts and biases off their initial values
    h0 = torch.randn(B, 32, 512, device=dev).requires_grad_(True)
    dp = (torch.empty(16, B, 1, 1, device=dev).bernoulli_(0.8).div_(0.8)) if drop else None
    up = torch.randn(B, 32, 512, device=dev)

    def reference(h):
        for i, b in enumerate(blocks):
            z = F.layer_norm(h, (512,), b.norm1.weight, b.norm1.bias, 1e-5)
            q, k, v = b.qkv(z).reshape(B, 32, 3, 4, 128).permute(2, 0, 3, 1, 4)
            o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, 32, 512)
            br = b.proj(o)
            h = h + (br if dp is None else br * dp[2 * i])
            z = F.layer_norm(h, (512,), b.norm2.weight, b.norm2.bias, 1e-5)
            br = b.fc2(F.gelu(b.fc1(z)))
            h = h + (br if dp is None else br * dp[2 * i + 1])
        return h
    want = reference(h0)
    want.backward(up)
    ref = {n: p.grad.clone() for n, p in blocks.named_parameters()}
    ref_h = h0.grad.clone()
    blocks.zero_grad()
    h0.grad = None
    # the step's weight packs, as train_forward prepares them
    pk = training.WeightPacks([m.weight for m in blocks.modules() if isinstance(m, nn.Linear)])
    pk.refresh()
    keep = training._packs_blocks
    training._packs_blocks = pk
    try:
        ps = []
        for b in blocks:
            ps += [b.norm1.weight, b.norm1.bias, b.qkv.weight, b.proj.weight, b.proj.bias, b.norm2.weight, b.norm2.bias, b.fc1.weight, b.fc1.bias,
                   b.fc2.weight, b.fc2.bias]
        got = training.StackFn.apply(h0, dp, *ps)
        e_out = rel_l2(got.detach().cpu(), want.detach().cpu())
        got.backward(up)
    finally:
        training._packs_blocks = keep
    e_h = rel_l2(h0.grad.cpu(), ref_h.cpu())
    worst = max(((n, rel_l2(p.grad.cpu(), ref[n].cpu())) for n, p in blocks.named_parameters()), key=lambda v: v[1])
    print(f"persistent block stack, B = {B}, DropPath {drop}: output rel-L2 {e_out:.2e}, input gradient {e_h:.2e}, worst parameter gradient {worst[1]:.2e} ({worst[0]})")
    assert e_out < 1.5e-2 and e_h < 3e-2 and worst[1] < 3e-2


def test_training_convolutions_at_the_bench_shapes():
    """The training-mode convolutions at the bench shapes (32 clips: the 224-position tiles, the channel blocks on grid z, the one-launch strided data
    gradients), forward with its BatchNorm partial sums and data gradient against torch's fp32 convolution (`scripts/ubench_conv_variants.py`)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "ubench_conv_variants.py"), "32"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [l for l in out.stdout.splitlines() if " fwd " in l]
    assert len(rows) == 7, out.stdout
    for l in rows:
        f, s, q, d = (float(v) for v in re.search(r"\(rel ([\d.e+-]+), sums ([\d.e+-]+) / ([\d.e+-]+)\).*\(rel ([\d.e+-]+)\)", l).groups())
        assert f < 2e-5 and d < 2e-5 and s < 1e-3 and q < 1e-4, l


@pytest.mark.parametrize("n,L,fold", [(32, 68266, False), (32, 68266, True), (5, 20011, True)])
def test_first_layer_weight_gradient_at_the_bench_shape_vs_fp64(n, L, fold):
    """syn_conv1d_first_wgrad / _bn at the bench shape (32 clips of 68266 samples x 2 channels: 1024 persistent workgroups walking position pairs across
    the batch, one partial gradient each) against torch's convolution weight gradient in float64; with `fold` the BatchNorm + LeakyReLU backward of the
    convolution's output is formed from (dz, y, statistics, affine, dgamma / dbeta) as the values are loaded: dy = scale (dp - dbeta / M - xhat dgamma / M),
    dp = dz act'(scale y + shift).  The goldens of the training step run 4-8 clips: this is the many-workgroup path of the kernel."""
    from syntalker_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + L)
    cin, stride, pad = 2, 5, 1700
    l_out = (L + 2 * pad - 15) // stride + 1
    x = torch.randn(n, L, cin, generator=g).cuda()
    dz = torch.randn(n, l_out, 64, generator=g).cuda()
    st = _lib.current_stream(x.device)
    ws = torch.empty(lib.syn_conv1d_first_parts(n, l_out) * 64 * cin * 15, device="cuda")
    dw = torch.empty(64, cin, 15, device="cuda")
    if fold:
        y = torch.randn(n, l_out, 64, generator=g).cuda()
        mu, rs = torch.randn(64, generator=g).cuda() * 0.1, torch.rand(64, generator=g).cuda() + 0.5
        sc, sh = torch.rand(64, generator=g).cuda() + 0.5, torch.randn(64, generator=g).cuda() * 0.3
        dgb = torch.randn(3, 64, generator=g).cuda() * (n * l_out) ** 0.5
        stats, aff = torch.cat([mu, rs]).contiguous(), torch.cat([sc, sh]).contiguous()
        _lib.check(lib.syn_conv1d_first_wgrad_bn(x.data_ptr(), dz.data_ptr(), y.data_ptr(), stats.data_ptr(), aff.data_ptr(), dgb.data_ptr(), 1,
                                                 n, L, cin, stride, pad, ws.data_ptr(), dw.data_ptr(), st), "syn_conv1d_first_wgrad_bn")
        m = float(n * l_out)
        dp = torch.where(y.double() * sc.double() + sh.double() > 0, dz.double(), 0.01 * dz.double())
        dy = sc.double() * (dp - dgb[1].double() / m - (y.double() - mu.double()) * rs.double() * dgb[0].double() / m)
    else:
        _lib.check(lib.syn_conv1d_first_wgrad(x.data_ptr(), dz.data_ptr(), n, L, cin, stride, pad, ws.data_ptr(), dw.data_ptr(), st), "syn_conv1d_first_wgrad")
        dy = dz.double()
    x4 = x.permute(0, 2, 1).unsqueeze(2).double()                               # (N, cin, 1, L)
    want = torch.nn.grad.conv2d_weight(x4, (64, cin, 1, 15), dy.permute(0, 2, 1).unsqueeze(2).contiguous(), stride=(1, stride), padding=(0, pad)).squeeze(2)
    e = rel_l2(dw.double().cpu(), want.cpu())
    print(f"first-layer weight gradient, {n} clips x {L} samples, fold {fold}: rel-L2 vs float64 {e:.2e}")
    assert e < 1e-5


@pytest.mark.parametrize("n,L,cin", [(32, 68266, 2), (3, 20011, 2), (1, 977, 1)])
def test_block0_conv1_weight_gradient_and_bn1_sums_in_one_pass(n, L, cin):
    """syn_conv1d_first_wgrad_bn_lin (round 6): dW of block 0's conv1 = scale (S1 - dbeta / M S2 - dgamma / M S3) with the three sums and bn1's dbeta / dgamma
    accumulated in ONE pass over (dz, y) - against the two passes it replaces (syn_bn_bwd_stats, then syn_conv1d_first_wgrad_bn with the finished sums) and against
    float64.  The subtraction is where accuracy could go: the waveform here has a DC offset (S2 far from zero) and dz a per-channel mean (dbeta far from zero)."""
    from syntalker_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + L)
    stride, pad, C = 5, 1700, 64
    l_out = (L + 2 * pad - 15) // stride + 1
    rows = n * l_out
    x = (torch.randn(n, L, cin, generator=g) + 0.3).cuda()
    y = torch.randn(n, l_out, C, generator=g).cuda()
    dz = (torch.randn(n, l_out, C, generator=g) + 0.2 * torch.randn(C, generator=g)).cuda()
    mu, rs = torch.randn(C, generator=g).cuda() * 0.1, torch.rand(C, generator=g).cuda() + 0.5
    gam, bet = torch.rand(C, generator=g).cuda() + 0.5, torch.randn(C, generator=g).cuda() * 0.3
    stats, aff = torch.cat([mu, rs]).contiguous(), torch.cat([rs * gam, bet - mu * rs * gam]).contiguous()
    st = _lib.current_stream(x.device)
    parts = lib.syn_conv1d_first_parts(n, l_out)
    # two passes
    ws1, dgb_a = torch.empty(2 * lib.syn_bn_chunks(rows) * C, device="cuda"), torch.empty(3, C, device="cuda")
    _lib.check(lib.syn_bn_bwd_stats(dz.data_ptr(), None, y.data_ptr(), stats.data_ptr(), gam.data_ptr(), bet.data_ptr(), rows, C, 1, ws1.data_ptr(), dgb_a.data_ptr(), st), "syn_bn_bwd_stats")
    wsa, dwa = torch.empty(parts * C * cin * 15, device="cuda"), torch.empty(C, cin, 15, device="cuda")
    _lib.check(lib.syn_conv1d_first_wgrad_bn(x.data_ptr(), dz.data_ptr(), y.data_ptr(), stats.data_ptr(), aff.data_ptr(), dgb_a.data_ptr(), 1, n, L, cin, stride, pad,
                                             wsa.data_ptr(), dwa.data_ptr(), st), "syn_conv1d_first_wgrad_bn")
    # one pass
    wsb, dwb, dgb_b = torch.empty(parts * (2 * C * cin * 15 + 160), device="cuda"), torch.empty(C, cin, 15, device="cuda"), torch.empty(3, C, device="cuda")
    _lib.check(lib.syn_conv1d_first_wgrad_bn_lin(x.data_ptr(), dz.data_ptr(), y.data_ptr(), stats.data_ptr(), aff.data_ptr(), 1, n, L, cin, stride, pad,
                                                 wsb.data_ptr(), dwb.data_ptr(), dgb_b.data_ptr(), st), "syn_conv1d_first_wgrad_bn_lin")
    torch.cuda.synchronize()
    D = lambda t: t.double()
    m = float(rows)
    xh = (D(y) - D(mu)) * D(rs)
    dp = torch.where(D(y) * D(aff[:C]) + D(aff[C:]) > 0, D(dz), 0.01 * D(dz))
    dbeta, dgamma = dp.sum((0, 1)), (dp * xh).sum((0, 1))
    dy = D(aff[:C]) * (dp - dbeta / m - xh * dgamma / m)
    x4 = x.permute(0, 2, 1).unsqueeze(2).double()
    want = torch.nn.grad.conv2d_weight(x4, (C, cin, 1, 15), dy.permute(0, 2, 1).unsqueeze(2).contiguous(), stride=(1, stride), padding=(0, pad)).squeeze(2)
    ea, eb = rel_l2(dwa.double().cpu(), want.cpu()), rel_l2(dwb.double().cpu(), want.cpu())
    eg = max(rel_l2(dgb_b[0].double().cpu(), dgamma.cpu()), rel_l2(dgb_b[1].double().cpu(), dbeta.cpu()))
    print(f"block 0 conv1, {n} clips: dW vs float64 - two passes {ea:.2e}, one pass {eb:.2e}; dgamma / dbeta {eg:.2e}")
    assert eb < 1e-5 and eg < 1e-5 and float(dgb_b[2].abs().max()) == 0.0
    assert rel_l2(dgb_b[:2].cpu(), dgb_a[:2].cpu()) < 1e-5


@pytest.mark.parametrize("n,L", [(32, 13437), (5, 2000)])
def test_paired_strided_weight_gradients_equal_two_single_launches(n, L):
    """syn_conv1d_train_wgrad_pair (round 6): conv1's and the shortcut convolution's weight gradients of block 1 from one launch that stages their common input
    once (the tile's two halves are the two dy; a share of the partial sums holds both gradients).  Same shares, same chunk order, same products: bit-equal to
    the two single launches."""
    from syntalker_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + L)
    cin, stride, cout = 64, 6, 64
    l_out = (L - 15) // stride + 1
    x = torch.randn(n, L, cin, generator=g).cuda()
    dya, dyb = torch.randn(n, l_out, cout, generator=g).cuda(), torch.randn(n, l_out, cout, generator=g).cuda()
    st = _lib.current_stream(x.device)
    shares, per = lib.syn_conv1d_wgrad_shares(n, l_out, stride * cin, cout), cout * 18 * cin
    singles = []
    for dy in (dya, dyb):
        ws, dw = torch.empty(shares * per, device="cuda"), torch.empty(cout, cin, 15, device="cuda")
        _lib.check(lib.syn_conv1d_train_wgrad(x.data_ptr(), dy.data_ptr(), n, L, cin, stride, 0, cout, ws.data_ptr(), dw.data_ptr(), st), "syn_conv1d_train_wgrad")
        singles.append(dw)
    ws = torch.empty(shares * 2 * per, device="cuda")
    _lib.check(lib.syn_conv1d_train_wgrad_pair(x.data_ptr(), dya.data_ptr(), dyb.data_ptr(), n, L, cin, stride, 0, cout, ws.data_ptr(), st), "syn_conv1d_train_wgrad_pair")
    outs = [torch.empty(cout, cin, 15, device="cuda") for _ in range(2)]
    jobs = (_lib.SynWgradSumJob * 2)()
    for i in range(2):
        jobs[i].part, jobs[i].dw, jobs[i].n_clips, jobs[i].l_out, jobs[i].cin, jobs[i].stride, jobs[i].cout, jobs[i].first_layer = ws.data_ptr() + 4 * per * i, outs[i].data_ptr(), n, l_out, cin, stride, cout, 0
        jobs[i].share_pitch = 2 * per
    _lib.check(lib.syn_conv1d_wgrad_sums(jobs, 2, st), "syn_conv1d_wgrad_sums")
    torch.cuda.synchronize()
    assert torch.equal(outs[0], singles[0]) and torch.equal(outs[1], singles[1])
    assert lib.syn_conv1d_train_wgrad_pair(x.data_ptr(), dya.data_ptr(), dyb.data_ptr(), n, L, cin, 3, 0, cout, ws.data_ptr(), st) != 0      # (block 1's geometry only)


@pytest.mark.parametrize("n,L,cin", [(32, 68266, 2), (3, 20011, 2), (1, 977, 1)])
def test_block0_backward_tail_folded_into_the_shortcut_weight_gradient(n, L, cin):
    """syn_conv1d_first_wgrad_tail (round 6): block 0's tail backward (both BatchNorms' data gradients from dout, y2, y_sc) and the shortcut convolution's
    weight gradient as ONE kernel - dy2 written, the shortcut's dy fed straight into the matrix products.  Against the two-kernel path it replaces
    (syn_bn_block_bwd's apply pass, then syn_conv1d_first_wgrad on the stored dshortcut) and against float64."""
    from syntalker_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + L)
    stride, pad, C = 5, 1700, 64
    l_out = (L + 2 * pad - 15) // stride + 1
    rows = n * l_out
    x = torch.randn(n, L, cin, generator=g).cuda()
    dout, y2, ysc = (torch.randn(n, l_out, C, generator=g).cuda() for _ in range(3))
    mk = lambda: (torch.cat([torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5]).cuda(),
                  torch.cat([torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3]).cuda())
    (st2, af2), (sts, afs) = mk(), mk()
    st = _lib.current_stream(x.device)
    ws = torch.empty(3 * lib.syn_bn_chunks(rows) * C, device="cuda")
    dgb2, dgbs = torch.empty(3, C, device="cuda"), torch.empty(3, C, device="cuda")
    dy2_a, dsh_a = torch.empty_like(y2), torch.empty_like(y2)
    args = (dout.data_ptr(), y2.data_ptr(), ysc.data_ptr(), st2.data_ptr(), af2.data_ptr(), sts.data_ptr(), afs.data_ptr(), rows, C, 1, ws.data_ptr(), dgb2.data_ptr(), dgbs.data_ptr())
    _lib.check(lib.syn_bn_block_bwd(*args, dy2_a.data_ptr(), dsh_a.data_ptr(), st), "syn_bn_block_bwd")
    parts = lib.syn_conv1d_first_parts(n, l_out)
    wsa, dwa = torch.empty(parts * C * cin * 15, device="cuda"), torch.empty(C, cin, 15, device="cuda")
    _lib.check(lib.syn_conv1d_first_wgrad(x.data_ptr(), dsh_a.data_ptr(), n, L, cin, stride, pad, wsa.data_ptr(), dwa.data_ptr(), st), "syn_conv1d_first_wgrad")
    g2a, gsa = dgb2.clone(), dgbs.clone()
    # the fused path: statistics only, then the one kernel; the partial sums added by the block's reduction launch
    _lib.check(lib.syn_bn_block_bwd(*args, None, None, st), "syn_bn_block_bwd (statistics only)")
    assert torch.equal(dgb2, g2a) and torch.equal(dgbs, gsa)
    dy2_b = torch.empty_like(y2)
    wsb, dwb = torch.empty(parts * C * cin * 15, device="cuda"), torch.empty(C, cin, 15, device="cuda")
    _lib.check(lib.syn_conv1d_first_wgrad_tail(x.data_ptr(), dout.data_ptr(), y2.data_ptr(), ysc.data_ptr(), st2.data_ptr(), af2.data_ptr(), sts.data_ptr(), afs.data_ptr(),
                                               dgb2.data_ptr(), dgbs.data_ptr(), 1, n, L, cin, stride, pad, wsb.data_ptr(), dy2_b.data_ptr(), st), "syn_conv1d_first_wgrad_tail")
    job = (_lib.SynWgradSumJob * 1)()
    job[0].part, job[0].dw, job[0].n_clips, job[0].l_out, job[0].cin, job[0].stride, job[0].cout, job[0].first_layer = wsb.data_ptr(), dwb.data_ptr(), n, l_out, cin, stride, C, 1
    _lib.check(lib.syn_conv1d_wgrad_sums(job, 1, st), "syn_conv1d_wgrad_sums")
    torch.cuda.synchronize()
    assert rel_l2(dy2_b.cpu(), dy2_a.cpu()) < 1e-6 and rel_l2(dwb.cpu(), dwa.cpu()) < 1e-5
    # float64 restatement
    m = float(rows)
    D = lambda t: t.double()
    p = D(y2) * D(af2[:C]) + D(af2[C:]) + D(ysc) * D(afs[:C]) + D(afs[C:])
    d = torch.where(p > 0, D(dout), 0.01 * D(dout))
    dy2 = D(af2[:C]) * (d - D(dgb2[1]) / m - (D(y2) - D(st2[:C])) * D(st2[C:]) * D(dgb2[0]) / m)
    dys = D(afs[:C]) * (d - D(dgbs[1]) / m - (D(ysc) - D(sts[:C])) * D(sts[C:]) * D(dgbs[0]) / m)
    x4 = x.permute(0, 2, 1).unsqueeze(2).double()
    want = torch.nn.grad.conv2d_weight(x4, (C, cin, 1, 15), dys.permute(0, 2, 1).unsqueeze(2).contiguous(), stride=(1, stride), padding=(0, pad)).squeeze(2)
    e1, e2 = rel_l2(dy2_b.double().cpu(), dy2.cpu()), rel_l2(dwb.double().cpu(), want.cpu())
    print(f"block-0 tail folded into the shortcut's weight gradient, {n} clips: dy2 {e1:.2e}, dW {e2:.2e} vs float64")
    assert e1 < 1e-6 and e2 < 1e-5


@pytest.mark.parametrize("cin,stride,cout,pad,L", [(64, 1, 64, 7, 14331), (64, 6, 64, 0, 14331), (64, 1, 64, 7, 2387), (64, 6, 128, 0, 2387),
                                                  (128, 1, 128, 7, 396), (128, 3, 256, 0, 396), (256, 1, 256, 7, 128)])
def test_training_conv_weight_gradients_at_the_bench_shapes(cin, stride, cout, pad, L):
    """Weight gradients of the encoder's convolutions at the bench shapes (32 clips: 256 / 155-220 / 64 / 16 shares of position chunks, the partial sums
    added by k_conv_wgrad_sum; strided layers on k_conv_wgrad_s) against torch's fp32 weight gradient.  Default products (x read rounded to bf16:
    two of three): 1.7e-3 = bf16's rounding of x, which independent random operands do not average down (signal and error are both random sums over
    the positions; the training step's gradients against the oracle are the measure for real data)."""
    from syntalker_amd import training
    g = torch.Generator().manual_seed(cin + stride + L)
    x = torch.randn(32, cin, 1, L, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 15, generator=g) / (cin * 15) ** 0.5).cuda().requires_grad_(True)
    y = training.ConvSplitFn.apply(x, w, stride, pad)
    gy = torch.randn(y.shape, generator=g).cuda()
    y.backward(gy)
    want = torch.nn.grad.conv2d_weight(x, w.shape, gy, stride=(1, stride), padding=(0, pad))
    e = rel_l2(w.grad.cpu(), want.cpu())
    print(f"conv {cin}x{stride}->{cout} at {L} positions x 32 clips: weight gradient rel-L2 vs torch fp32 {e:.2e}")
    assert e < 4e-3
