"""The wave-per-sequence step kernel k_seq (csrc/syn_seq.inc; large batches, latent in fragment order) through the C ABI:
against the CPU oracle, against the token-resident kernel, and its size-independent properties.  (Its goldens - forward,
DDPM-10, DDIM-50, the h3d flag combinations - run in tests/test_gpu_parity.py through the `kernel` fixture.)"""
import pytest
import torch

from syntalker_amd import synth, tape
from tests.conftest import rel_l2
from tests.refmodel import synth_state_dict
from tests.test_gpu_parity import DEV, FWD_TOL, LOOP_TOL, _model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def beatx():
    return _model("beatx")


def test_fragment_order_kernels_match_the_host_definition():
    from syntalker_amd import _lib
    lib = _lib.load()
    B = 3
    x = torch.randn(B, 1536, 1, 32, device=DEV)
    f32 = torch.empty(B, 48, 4, 64, 4, device=DEV)
    b16 = torch.empty(B, 48, 2, 64, 8, dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.syn_x_to_fragment(x.data_ptr(), B, f32.data_ptr(), b16.data_ptr(), _lib.current_stream()), "to")
    x_btc = x.reshape(B, 1536, 32).transpose(1, 2).contiguous()
    assert torch.equal(f32, tape.to_fragment_order(x_btc))
    assert torch.equal(b16, tape.to_fragment_order_bf16(x_btc))
    back = torch.empty_like(x)
    _lib.check(lib.syn_x_from_fragment(f32.data_ptr(), B, back.data_ptr(), _lib.current_stream()), "from")
    assert torch.equal(back, x)


@pytest.mark.parametrize("B", [1, 3, 4, 5, 9])
def test_forward_ragged_workgroups_vs_oracle(beatx, B):
    """4 sequences per workgroup: fewer sequences than waves, exactly one workgroup, a ragged last workgroup."""
    from oracle import denoiser_ref as dr
    y, x = synth.synth_clip_inputs(B, seed=13), synth.synth_latent(B, seed=13)
    t = (torch.arange(B) * 53 + 1) % 1000
    with torch.no_grad():
        want = dr.mdm_forward(synth_state_dict("beatx"), x, t, y)
        beatx.layer_mode = 5
        try:
            got = beatx(x.to(DEV), t.to(DEV), synth.to_device(y, DEV)).cpu()
        finally:
            beatx.layer_mode = 0
    e = rel_l2(got, want)
    print(f"k_seq B={B} forward rel-L2 vs oracle {e:.3e}")
    assert e < FWD_TOL


def test_rows_are_independent_deterministic_and_close_to_the_token_resident_kernel(beatx):
    """Bitwise: repeat runs; clip b of a batch == the same clip alone == the same clip in another wave / workgroup.
    Against k_stack only the rounding points of the LayerNorm gain differ (folded into the bf16 weights here)."""
    from syntalker_amd import engine
    y, x = synth.to_device(synth.synth_clip_inputs(3, seed=12), DEV), synth.synth_latent(3, seed=12).to(DEV)
    t = torch.tensor([10, 400, 900], device=DEV)
    pm = beatx.packed()
    cond = beatx.variant_conds(y, [(False, False, None)])[0]
    ident = engine.identity_coefs(DEV)

    def run(B, xs, cs, ts, mode=5):
        sb = engine.StepBuffers(B, 1, DEV, layer_mode=mode)
        sb.cond.copy_(cs.reshape(-1, 512)); sb.load_x(xs); sb.t_model.copy_(ts.int()); sb.t_coef.zero_()
        engine.run_step(pm, sb, ident, False)
        return sb.read(sb.x).cpu()

    full = run(3, x, cond, t)
    assert torch.equal(full, run(3, x, cond, t))
    assert torch.equal(run(1, x[1:2], cond[1:2], t[1:2]), full[1:2])
    big = run(66, x.repeat(22, 1, 1, 1), cond.repeat(22, 1, 1), t.repeat(22))        # 17 workgroups, the last one half empty
    assert torch.equal(big, full.repeat(22, 1, 1, 1))
    assert rel_l2(full, run(3, x, cond, t, mode=4)) < 8e-3


def test_noisy_steps_fused_generator_and_injected_noise_agree(beatx):
    """One DDPM step at t = 500 three ways: noise drawn in the kernel's epilogue (Philox keyed by the token-major element
    index), the same values drawn by syn_randn and injected, and the fp64 recombination of the captured x0_hat."""
    from syntalker_amd import engine
    from syntalker_amd.process import create_gaussian_diffusion
    B = 6
    y = synth.to_device(synth.synth_clip_inputs(B, seed=31), DEV)
    xT = synth.synth_latent(B, seed=31).to(DEV)
    pm = beatx.packed()
    coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), DEV)
    outs = []
    for fused in (True, False):
        sb = engine.StepBuffers(B, 1, DEV, want_x0=True, layer_mode=5)
        sb.cond.copy_(beatx.variant_conds(y, [(False, False, None)]).reshape(-1, 512))
        sb.load_x(xT); sb.t_model.fill_(500); sb.t_coef.fill_(500)
        x_before = sb.x.clone()
        sb.draw_noise(9, 500, first_clip=3); sb.set_rng(9, 3)
        engine.run_step(pm, sb, coef, True, fused_rng=fused)
        c = coef[500].double()
        want = c[0] * sb.x0.double() + c[1] * x_before.double() + c[2] * sb.noise.double()
        assert rel_l2(sb.x.double().cpu(), want.cpu()) < 1e-6
        # the bf16 operands the kernel leaves for the NEXT step's input GEMM are exactly bf16(x_next), in the operand layout:
        # fp32 [clip][nf][q = 2c + eh][lane][r]  ->  bf16 [clip][nf][c][lane][e = 4 eh + r]
        shadow = sb.x.view(B, 48, 2, 2, 64, 4).permute(0, 1, 2, 4, 3, 5).reshape(B, 48, 2, 64, 8).to(torch.bfloat16)
        assert torch.equal(sb.xb.view(torch.bfloat16).reshape(B, 48, 2, 64, 8), shadow)
        outs.append(sb.read(sb.x).cpu())
    assert torch.equal(outs[0], outs[1])


def test_seeded_ddpm_steps_vs_oracle_with_regenerated_noise(beatx):
    """The path bench.py times - DDPM, noise drawn in the epilogue, 10-step graph replays - against the oracle fed the
    identical noise regenerated with syn_randn(seed, t, first_clip)."""
    from oracle import denoiser_ref as dr
    from oracle.process_ref import RefProcess
    from syntalker_amd import _lib
    from syntalker_amd.process import create_gaussian_diffusion
    B, K, seed = 4, 20, 77
    y, xT = synth.synth_clip_inputs(B, seed=41), synth.synth_latent(B, seed=41)
    d = create_gaussian_diffusion()
    beatx.layer_mode = 5
    try:
        got = d.p_sample_loop(beatx, (B, 1536, 1, 32), noise=xT.to(DEV), clip_denoised=False,
                              model_kwargs={"y": synth.to_device(y, DEV)}, skip_timesteps=1000 - K, seed=seed).cpu()
    finally:
        beatx.layer_mode = 0
    lib = _lib.load()
    sn = []
    for t in range(K - 1, -1, -1):                      # the loop visits t = K-1 .. 0; stream id = t
        buf = torch.empty(B, 32, 1536, device=DEV)
        _lib.check(lib.syn_randn(buf.data_ptr(), buf.numel(), seed, t, 0, _lib.current_stream()), "syn_randn")
        sn.append(buf.transpose(1, 2).reshape(B, 1536, 1, 32).cpu())
    sd = synth_state_dict("beatx")
    want = RefProcess(False).p_sample_loop(lambda a, b, c: dr.mdm_forward(sd, a, b, c), (B, 1536, 1, 32), y,
                                           noise=xT.clone(), step_noise=torch.stack(sn), skip_timesteps=1000 - K)
    e = rel_l2(got, want)
    print(f"k_seq seeded 20-step DDPM vs oracle rel-L2 {e:.3e}")
    assert e < LOOP_TOL


@pytest.mark.parametrize("noisy", [True, False])
def test_persistent_multi_step_launch_equals_single_steps_bitwise(beatx, noisy):
    """syn_denoise_steps on fragment-order latents is ONE launch in which every workgroup carries its sequences through all
    the steps (skewed starts, the x_next a wave stores is the x_t it loads one step later): same bits as step by step."""
    from syntalker_amd import engine
    from syntalker_amd.process import create_gaussian_diffusion
    B, K = 9, 7                                                        # three workgroups, the last one ragged
    y = synth.to_device(synth.synth_clip_inputs(B, seed=51), DEV)
    xT = synth.synth_latent(B, seed=51).to(DEV)
    pm = beatx.packed()
    coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), DEV)
    ts = [600 - 37 * i for i in range(K)]
    outs = []
    for steps in (1, K):
        sb = engine.StepBuffers(B, 1, DEV, layer_mode=5)
        sb.cond.copy_(beatx.variant_conds(y, [(False, False, None)]).reshape(-1, 512))
        sb.set_rng(5, 2)
        g = engine.StepGraph(pm, sb, coef, noisy, fused_rng=noisy, scheduled=True, steps=steps)
        sb.load_x(xT)                                                  # (capturing ran a warm-up step)
        g.set_schedule(ts, ts)
        for _ in range(K // steps):
            g.replay()
        assert int(g.counter.item()) == K
        outs.append((sb.read(sb.x).cpu(), sb.xb.clone().cpu()))
    assert torch.isfinite(outs[0][0]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("V,B", [(2, 5), (3, 3), (4, 2)])
def test_guided_batches_on_the_wave_per_sequence_kernel(V, B):
    """The V variants of a clip are the waves of one workgroup (two clips per workgroup at V = 2: B = 5 leaves a ragged one) and
    meet in the output stage through LDS.  One noisy guided step against the token-resident kernel (which combines the
    variants' residual streams BEFORE the output projection: same value up to the bf16 rounding of the combined stream),
    the fp64 recombination of per-variant evaluations, and a 6-step persistent launch against the same steps one by one."""
    from syntalker_amd import engine
    from syntalker_amd.process import create_gaussian_diffusion
    m = _model("h3d")
    pm = m.packed()
    coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), DEV)
    g = torch.Generator().manual_seed(5)
    cond = torch.randn(V * B * 32, 512, generator=g).to(DEV) * 0.5
    w = torch.tensor([[2.5, -1.5, 0.0, 0.0][:V], [1.0, 0.5, -0.5, 0.0][:V], [0.25, 0.25, 0.25, 0.25][:V]])
    w[2, 0] += 1.0 - w[2].sum(); w[1, 0] += 1.0 - w[1].sum()          # every row sums to 1, as every guidance formula does
    xT = synth.synth_latent(B, seed=61).to(DEV)
    tm = (torch.arange(V * B) % B * 97 + 300).int().to(DEV)             # a clip's variants share its timestep

    def run(mode, steps=1, replays=1, want_x0=False):
        sb = engine.StepBuffers(B, V, DEV, want_x0=want_x0, layer_mode=mode)
        sb.cond.copy_(cond); sb.cfg_w.copy_(w.to(DEV)); sb.set_rng(11, 4)
        if steps == 1 and replays == 1:
            sb.load_x(xT); sb.t_model.copy_(tm); sb.t_coef.copy_(tm[:B])
            engine.run_step(pm, sb, coef, True, fused_rng=True)
        else:
            gr = engine.StepGraph(pm, sb, coef, True, fused_rng=True, scheduled=True, steps=steps)
            sb.load_x(xT)
            ts = [500 - 41 * i for i in range(steps * replays)]
            gr.set_schedule(ts, ts)
            for _ in range(replays):
                gr.replay()
        return sb.read(sb.x).cpu(), (sb.read(sb.x0).cpu() if want_x0 else None), sb

    x5, x05, sb5 = run(5, want_x0=True)
    assert sb5.fragment
    x4, x04, _ = run(4, want_x0=True)
    e = rel_l2(x05, x04)
    print(f"guided V={V} B={B}: x0_hat k_seq vs k_stack rel-L2 {e:.3e}")
    assert e < 1.5e-2 and rel_l2(x5, x4) < 1.5e-2
    # per-variant evaluations (V = 1 batches of the same kernel), recombined in fp64
    acc = torch.zeros(B, 1536, 1, 32, dtype=torch.float64)
    for v in range(V):
        sb = engine.StepBuffers(B, 1, DEV, layer_mode=5)
        sb.cond.copy_(cond[v * B * 32:(v + 1) * B * 32]); sb.load_x(xT); sb.t_model.copy_(tm[:B]); sb.t_coef.zero_()
        engine.run_step(pm, sb, engine.identity_coefs(DEV), False)
        xv = sb.read(sb.x).cpu().double()
        for c in range(3):
            acc[:, 512 * c:512 * (c + 1)] += float(w[c, v]) * xv[:, 512 * c:512 * (c + 1)]
    assert rel_l2(x05.double(), acc) < 1e-5
    a, _, _ = run(5, steps=1, replays=6)
    b, _, _ = run(5, steps=6, replays=1)
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_persistent_launch_at_the_bench_size_equals_single_steps_bitwise(beatx):
    """1024 clips = one workgroup on every CU, ten steps per launch (what bench.py times): every clip must come out of the
    persistent launch with the bits ten single-step launches give it; 1280 clips (more workgroups than CUs) go out as
    two CU-filling slices, each carried through the ten steps by its own launch, and must agree as well."""
    from syntalker_amd import engine
    from syntalker_amd.process import create_gaussian_diffusion
    pm = beatx.packed()
    coef = engine.posterior_coefs(create_gaussian_diffusion().tables(), DEV)
    for B in (1024, 1280):
        g = torch.Generator(device=DEV).manual_seed(B)
        cond = torch.randn(B * 32, 512, device=DEV, generator=g) * 0.5
        xT = torch.randn(B, 1536, 1, 32, device=DEV, generator=g)
        ts = [999 - 7 * i for i in range(10)]
        outs = []
        for steps in (1, 10):
            sb = engine.StepBuffers(B, 1, DEV, layer_mode=5)
            sb.cond.copy_(cond); sb.set_rng(2024, 17)
            gr = engine.StepGraph(pm, sb, coef, True, fused_rng=True, scheduled=True, steps=steps)
            sb.load_x(xT)
            gr.set_schedule(ts, ts)
            for _ in range(10 // steps):
                gr.replay()
            outs.append(sb.x.clone())
        assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]), B
