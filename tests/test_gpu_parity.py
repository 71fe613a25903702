"""Parity of the HIP path (through MDM / the diffusion loops -> C ABI) against
 (a) the golden vectors produced by the imported reference, and (b) the CPU oracle on the same inputs.
Tolerances (SURVEY.md §8c, bf16 operands / fp32 accumulate): rel-L2 <= 2e-2 per model evaluation,
<= 3e-2 at the end of a sampling loop, always with identical injected noise."""
import numpy as np
import pytest
import torch

from syntalker_amd import synth
from tests.conftest import rel_l2
from tests.refmodel import synth_state_dict

pytestmark = pytest.mark.gpu
FWD_TOL, LOOP_TOL = 2e-2, 3e-2
DEV = "cuda"


def _model(variant="beatx"):
    if variant == "h3d":
        from syntalker_amd.denoiser_h3d import MDM
    else:
        from syntalker_amd.denoiser import MDM
    m = MDM(synth.default_args()).eval()
    missing, unexpected = m.load_state_dict(synth_state_dict(variant), strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    return m.to(DEV)


@pytest.fixture(scope="module")
def beatx():
    return _model("beatx")


@pytest.fixture(scope="module")
def h3d():
    return _model("h3d")


@pytest.fixture(params=[4, 3, 5], ids=["whole-step-kernel", "small-batch-kernel", "wave-per-sequence-kernel"])
def kernel(request, beatx, h3d):
    """Pin one of the step kernels (syn_step.reserved: 4 = token-resident whole-step kernel, 3 = persistent
    feature-split small-batch kernel, 5 = wave-per-sequence kernel with the latent in fragment order - guided batches: the variants of a clip are the waves of one workgroup);
    layer_mode 0 = the library's own choice by batch size."""
    beatx.layer_mode = h3d.layer_mode = request.param
    yield request.param
    beatx.layer_mode = h3d.layer_mode = 0


def test_forward_vs_golden(beatx, golden, kernel):
    y, x = synth.to_device(synth.synth_clip_inputs(2, seed=1), DEV), synth.synth_latent(2, seed=1).to(DEV)
    with torch.no_grad():
        o1 = beatx(x, torch.tensor([0, 3], device=DEV), y)
        o2 = beatx(x, torch.tensor([500, 999], device=DEV), y)
    e1, e2 = rel_l2(o1.cpu(), golden["beatx.fwd.t0_3"]), rel_l2(o2.cpu(), golden["beatx.fwd.t500_999"])
    print(f"forward rel-L2: {e1:.3e} {e2:.3e}")
    assert e1 < FWD_TOL and e2 < FWD_TOL


def test_forward_residual_stream_vs_golden(beatx, golden):
    """The per-operation path (layer_mode 1) leaves bf16(h8), the last residual stream, in the workspace (in production the
    whole step is one kernel and h never leaves the chip): layer-level bisecting tap against the reference."""
    y, x = synth.to_device(synth.synth_clip_inputs(2, seed=1), DEV), synth.synth_latent(2, seed=1).to(DEV)
    beatx.layer_mode = 1
    try:
        with torch.no_grad():
            beatx(x, torch.tensor([500, 999], device=DEV), y)
        h8 = beatx.step_buffers(2, 1).xn.float().view(2, 32, 512).cpu()
    finally:
        beatx.layer_mode = 0
    assert rel_l2(h8, golden["beatx.tap.h8"]) < FWD_TOL


@pytest.mark.parametrize("mt", [32, 64, 128])
def test_forward_every_tile_size_and_ragged_batch(beatx, mt):
    """B=5 (160 rows: a ragged last workgroup for the 64/128-row tiles) vs the CPU oracle."""
    from oracle import denoiser_ref as dr
    y, x = synth.synth_clip_inputs(5, seed=11), synth.synth_latent(5, seed=11)
    t = torch.tensor([1, 250, 500, 750, 999])
    with torch.no_grad():
        want = dr.mdm_forward(synth_state_dict("beatx"), x, t, y)
        beatx.m_tile, beatx.layer_mode = mt, 4
        try:
            got = beatx(x.to(DEV), t.to(DEV), synth.to_device(y, DEV)).cpu()
        finally:
            beatx.m_tile, beatx.layer_mode = 0, 0
    assert rel_l2(got, want) < FWD_TOL


@pytest.mark.parametrize("B", [1, 5, 9, 19])
def test_small_batch_kernel_ragged_groups(beatx, B):
    """The small-batch kernel splits the clips over 8 groups (XCDs): fewer clips than groups, uneven groups, more
    than one chunk of two sequences per group - all against the CPU oracle."""
    from oracle import denoiser_ref as dr
    y, x = synth.synth_clip_inputs(B, seed=13), synth.synth_latent(B, seed=13)
    t = (torch.arange(B) * 53 + 1) % 1000
    with torch.no_grad():
        want = dr.mdm_forward(synth_state_dict("beatx"), x, t, y)
        beatx.layer_mode = 3
        try:
            got = beatx(x.to(DEV), t.to(DEV), synth.to_device(y, DEV)).cpu()
        finally:
            beatx.layer_mode = 0
    assert rel_l2(got, want) < FWD_TOL


@pytest.mark.parametrize("mode", [4, 3], ids=["whole-step-kernel", "small-batch-kernel"])
def test_batch_rows_are_independent(beatx, mode):
    """Size-independent properties of the step kernels, bitwise: (1) repeat runs are deterministic,
    (2) clip b of a batch equals the same clip evaluated alone (for the small-batch kernel: whichever group /
    chunk it lands in), (3) every workgroup tile size of the whole-step kernel gives the same bits.  The per-clip conditioning tensor is computed once and shared, because the
    PyTorch/MIOpen conditioning ops are only fp32-reproducible (~3e-7) across batch sizes, and a
    perturbation of that size re-rolls bf16 roundings (measured: 3e-7 in -> 3e-3 out)."""
    from syntalker_amd import engine
    y, x = synth.to_device(synth.synth_clip_inputs(3, seed=12), DEV), synth.synth_latent(3, seed=12).to(DEV)
    t = torch.tensor([10, 400, 900], device=DEV)
    pm = beatx.packed()
    cond = beatx.variant_conds(y, [(False, False, None)])[0]
    ident = engine.identity_coefs(DEV)

    def run(B, xs, cs, ts, mt=0):
        sb = engine.StepBuffers(B, 1, DEV, m_tile=mt, layer_mode=mode)
        sb.cond.copy_(cs.reshape(-1, 512)); sb.load_x(xs); sb.t_model.copy_(ts.int()); sb.t_coef.zero_()
        engine.run_step(pm, sb, ident, False)
        return sb.read(sb.x).cpu()

    full = run(3, x, cond, t)
    assert torch.equal(full, run(3, x, cond, t))
    assert torch.equal(run(1, x[1:2], cond[1:2], t[1:2]), full[1:2])
    if mode == 4:
        for mt in (32, 64, 128):
            assert torch.equal(run(3, x, cond, t, mt), full)
    else:       # 20 copies of the 3 clips: groups of 7-8 clips, four chunks each; every copy must equal the original
        big = run(60, x.repeat(20, 1, 1, 1), cond.repeat(20, 1, 1), t.repeat(20))
        assert torch.equal(big, full.repeat(20, 1, 1, 1))


def test_fused_layer_kernels_equal_unfused_bitwise(beatx):
    """The whole-stack kernel (mode 4) and the five-kernels-per-block path (mode 1) round at exactly the same points (LayerNorm outputs, q/k/v, P, o, hidden in bf16; fp32
    accumulation in the same k order) -> identical bits, for every tile size."""
    y, x = synth.to_device(synth.synth_clip_inputs(5, seed=41), DEV), synth.synth_latent(5, seed=41).to(DEV)
    t = torch.tensor([3, 100, 450, 800, 999], device=DEV)
    outs = {}
    with torch.no_grad():
        for mode in (4, 1):
            for mt in (32, 64, 128):
                beatx.layer_mode, beatx.m_tile = mode, mt
                try:
                    outs[(mode, mt)] = beatx(x, t, y).cpu()
                finally:
                    beatx.layer_mode, beatx.m_tile = 0, 0
    ref = outs[(1, 32)]
    assert torch.isfinite(ref).all()
    for k, v in outs.items():
        assert torch.equal(v, ref), k


def test_ddpm10_and_ddim50_vs_golden(beatx, golden, kernel):
    from syntalker_amd.process import create_gaussian_diffusion
    y, xT = synth.to_device(synth.synth_clip_inputs(1, seed=2), DEV), synth.synth_latent(1, seed=2).to(DEV)
    s = create_gaussian_diffusion().p_sample_loop(beatx, (1, 1536, 1, 32), noise=xT.clone(), clip_denoised=False,
                                                  model_kwargs={"y": y}, skip_timesteps=990,
                                                  step_noise=synth.synth_step_noise(10, 1, seed=3))
    e = rel_l2(s.cpu(), golden["beatx.ddpm10.sample"])
    print(f"ddpm10 rel-L2 {e:.3e}")
    assert e < LOOP_TOL
    s = create_gaussian_diffusion(use_ddim=True).ddim_sample_loop(
        beatx, (1, 1536, 1, 32), noise=xT.clone(), clip_denoised=False, model_kwargs={"y": y},
        step_noise=synth.synth_step_noise(50, 1, seed=4))
    e = rel_l2(s.cpu(), golden["beatx.ddim50.sample"])
    print(f"ddim50 rel-L2 {e:.3e}")
    assert e < LOOP_TOL


def test_fused_loop_equals_generic_loop(beatx):
    """The hipGraph loop and the per-step generic path (MDM.forward + torch posterior) agree to the bf16
    re-rounding floor: the two posterior updates differ by fp32 rounding (~1e-7), which is enough to flip
    bf16 roundings downstream (see test_batch_rows_are_independent: 3e-7 in -> 3e-3 out)."""
    from syntalker_amd.process import create_gaussian_diffusion
    d = create_gaussian_diffusion()
    y, xT = synth.to_device(synth.synth_clip_inputs(2, seed=21), DEV), synth.synth_latent(2, seed=21).to(DEV)
    sn = synth.synth_step_noise(6, 2, seed=22)
    fused = d.p_sample_loop(beatx, (2, 1536, 1, 32), noise=xT.clone(), clip_denoised=False, model_kwargs={"y": y},
                            skip_timesteps=994, step_noise=sn)
    final = None
    for out in d.p_sample_loop_progressive(beatx, (2, 1536, 1, 32), noise=xT.clone(), clip_denoised=False,
                                           model_kwargs={"y": y}, skip_timesteps=994, step_noise=sn):
        final = out["sample"]
    assert rel_l2(fused.cpu(), final.cpu()) < 1e-2


def test_h3d_flags_vs_golden(h3d, golden, kernel):
    y = synth.to_device(synth.synth_clip_inputs(2, seed=7, style_dim=256, style_zero=False), DEV)
    x, t = synth.synth_latent(2, seed=7).to(DEV), torch.tensor([10, 700], device=DEV)
    with torch.no_grad():
        for tag, fl in (("cond", {}), ("uncond", {"uncond": True}), ("noaudio", {"uncond_audio": True}),
                        ("both", {"uncond": True, "uncond_audio": True})):
            e = rel_l2(h3d(x, t, dict(y, **fl)).cpu(), golden[f"h3d.fwd.{tag}"])
            assert e < FWD_TOL, (tag, e)


def test_h3d_guidance_vs_golden(h3d, golden, kernel):
    from syntalker_amd import guidance as G
    y = synth.to_device(synth.synth_clip_inputs(2, seed=7, style_dim=256, style_zero=False), DEV)
    x, t = synth.synth_latent(2, seed=7).to(DEV), torch.tensor([10, 700], device=DEV)
    with torch.no_grad():
        yc = dict(y, scale=torch.ones(1, device=DEV) * 2.5)
        e = rel_l2(G.ClassifierFreeSampleModel(h3d)(x, t, yc).cpu(), golden["h3d.cfg"])
        assert e < FWD_TOL * 2, e       # guidance extrapolates: (1-s)*u + s*c amplifies rounding by ~|1-s|+|s|
        assert yc["uncond_audio"] is True            # reference quirk: the caller's dict is mutated
        yc = dict(y, scale_audio=torch.ones(1, device=DEV), scale_prompt=torch.ones(1, device=DEV) * 4.0)
        e = rel_l2(G.TwoClassifierFreeSampleModel(h3d)(x, t, yc).cpu(), golden["h3d.twocfg"])
        assert e < FWD_TOL * 4, e


def test_per_clip_guidance_scales_vs_reference(h3d, kernel):
    """One guidance scale per clip (the reference's y['scale'].view(-1, 1, 1, 1), diffusion/cfg_sampler.py:28,54): three clips, three scales, on each of
    the step kernels - a per-clip weight table [B][3][V] in the output stage's combination - against the reference's own wrappers
    (tests/golden/per_sample_scales_outputs.npz), single evaluations and a guided DDIM-50 loop with injected noise."""
    import os
    from syntalker_amd import guidance as G
    from syntalker_amd.process import create_gaussian_diffusion
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "per_sample_scales_outputs.npz"))
    y = synth.to_device(synth.synth_clip_inputs(3, seed=51, style_dim=256, style_zero=False), DEV)
    x, t = synth.synth_latent(3, seed=51).to(DEV), torch.tensor([10, 700, 333], device=DEV)
    with torch.no_grad():
        yc = dict(y, scale=torch.tensor([1.5, 2.5, 0.0], device=DEV))
        e = rel_l2(G.ClassifierFreeSampleModel(h3d)(x, t, yc).cpu(), fx["cfg"])
        assert e < FWD_TOL * 2, e
        yc = dict(y, scale_audio=torch.tensor([0.5, 1.0, 1.0], device=DEV), scale_prompt=torch.tensor([4.0, 2.0, 0.0], device=DEV))
        e = rel_l2(G.TwoClassifierFreeSampleModel(h3d)(x, t, yc).cpu(), fx["twocfg"])
        assert e < FWD_TOL * 4, e
        sn = synth.synth_step_noise(50, 3, seed=52).to(DEV)
        s = create_gaussian_diffusion(use_ddim=True).ddim_sample_loop(
            G.ClassifierFreeSampleModel(h3d), (3, 1536, 1, 32), noise=x.clone(), clip_denoised=False,
            model_kwargs={"y": dict(y, scale=torch.tensor([1.5, 2.5, 4.0], device=DEV))}, step_noise=sn)
        e = rel_l2(s.cpu(), fx["cfg.ddim50.sample"])
        print(f"guided DDIM-50 with per-clip scales (kernel {kernel}): rel-L2 {e:.3e}")
        assert e < LOOP_TOL, e


def test_h3d_guidance_vs_golden_in_a_split_tile_batch(h3d, golden):
    """The guided golden case (2 clips) tiled 6 times: 12 clips x 2 / 3 variants = 24 / 36 sequences, the range in which the
    whole-step kernel splits every tile over 4 workgroups; every copy must reproduce the reference's guided outputs."""
    from syntalker_amd import guidance as G
    y2 = synth.synth_clip_inputs(2, seed=7, style_dim=256, style_zero=False)
    y = synth.to_device({k: (v.repeat(6, *([1] * (v.dim() - 1))) if torch.is_tensor(v) else v) for k, v in y2.items()}, DEV)
    x = synth.synth_latent(2, seed=7).repeat(6, 1, 1, 1).to(DEV)
    t = torch.tensor([10, 700], device=DEV).repeat(6)
    with torch.no_grad():
        yc = dict(y, scale=torch.ones(1, device=DEV) * 2.5)
        out = G.ClassifierFreeSampleModel(h3d)(x, t, yc).cpu()
        for k in range(6):
            assert rel_l2(out[2 * k:2 * k + 2], golden["h3d.cfg"]) < FWD_TOL * 2, k
        yc = dict(y, scale_audio=torch.ones(1, device=DEV), scale_prompt=torch.ones(1, device=DEV) * 4.0)
        out = G.TwoClassifierFreeSampleModel(h3d)(x, t, yc).cpu()
        for k in range(6):
            assert rel_l2(out[2 * k:2 * k + 2], golden["h3d.twocfg"]) < FWD_TOL * 4, k


def _bodypart_case():
    y = synth.synth_clip_inputs(1, seed=8, style_dim=256, style_zero=False)
    g = synth._gen("part_prompts", 8)
    parts = {"upper_mask": torch.randn(1, 256, generator=g).to(DEV), "hands_mask": None,
             "lower_mask": torch.randn(1, 256, generator=g).to(DEV)}
    return synth.to_device(y, DEV), synth.synth_latent(1, seed=8).to(DEV), parts


def test_h3d_bodypart_guidance_vs_golden(h3d, golden, kernel):
    from syntalker_amd import guidance as G
    from syntalker_amd.process import create_gaussian_diffusion
    y, x, parts = _bodypart_case()
    t = torch.tensor([321], device=DEV)
    with torch.no_grad():
        w = G.TwoClassifierFreeSampleModel_Bodypart(h3d)
        plan = w.plan(dict(y, style_feature=parts))
        assert len(plan.variants) == 4               # 9 reference evaluations: 5 are duplicates or carry zero weight
        e = rel_l2(w(x, t, dict(y, style_feature=parts)).cpu(), golden["h3d.twocfg_bodypart"])
        assert e < FWD_TOL * 4, e
        w2 = G.ClassifierFreeSampleModel_Bodypart(h3d)
        e = rel_l2(w2(x, t, dict(y, style_feature=parts, scale=torch.ones(1, device=DEV) * 2.5)).cpu(),
                   golden["h3d.cfg_bodypart"])
        assert e < FWD_TOL * 2, e
    s = create_gaussian_diffusion(use_ddim=True).ddim_sample_loop(
        w, (1, 1536, 1, 32), noise=x.clone(), clip_denoised=False, model_kwargs={"y": dict(y, style_feature=parts)},
        step_noise=synth.synth_step_noise(50, 1, seed=9))
    e = rel_l2(s.cpu(), golden["h3d.ddim50_bodypart.sample"])
    print(f"guided ddim50 rel-L2 {e:.3e}")
    assert e < 6e-2, e


def test_cpu_tensors_fail_loudly():
    from syntalker_amd._lib import SynHipError
    from syntalker_amd.denoiser import MDM
    m = MDM(synth.default_args()).eval()
    with pytest.raises(SynHipError):
        m(synth.synth_latent(1), torch.tensor([3]), synth.synth_clip_inputs(1))


def test_bench_size_batch_is_copies_of_its_clips(beatx):
    """The bench batch (1024 clips = 256 workgroups x 4 waves of the wave-per-sequence kernel k_seq, one sequence per wave) built as 128 copies
    of 8 distinct clips: every copy must carry the bits of the original wherever it sits in the batch, and the 8
    originals must match the CPU oracle.  Conditioning is computed once for the 8 clips and tiled (see
    test_batch_rows_are_independent for why)."""
    from oracle import denoiser_ref as dr
    from syntalker_amd import engine
    y8, x8 = synth.synth_clip_inputs(8, seed=21), synth.synth_latent(8, seed=21)
    t8 = torch.tensor([0, 1, 50, 333, 500, 777, 998, 999])
    cond8 = beatx.variant_conds(synth.to_device(y8, DEV), [(False, False, None)])[0]
    B, R = 1024, 128
    sb = engine.StepBuffers(B, 1, DEV)
    sb.cond.copy_(cond8.repeat(R, 1, 1).reshape(-1, 512)); sb.load_x(x8.to(DEV).repeat(R, 1, 1, 1))
    sb.t_model.copy_(t8.int().to(DEV).repeat(R)); sb.t_coef.zero_()
    engine.run_step(beatx.packed(), sb, engine.identity_coefs(DEV), False)
    out = sb.read(sb.x)
    assert torch.equal(out, out[:8].repeat(R, 1, 1, 1))
    with torch.no_grad():
        want = dr.mdm_forward(synth_state_dict("beatx"), x8, t8, y8)
    assert rel_l2(out[:8].cpu(), want) < FWD_TOL


def test_largest_and_empty_batches(beatx):
    """4096 clips (131 072 token rows, 805 MB of fp32 latents: past any 16-bit / 27-bit index shortcut) as 512 copies of 8
    clips - every copy bitwise equal to the first, through both tile sizes the library would pick; and the empty batch is
    refused with an error instead of a launch."""
    from syntalker_amd import engine
    from syntalker_amd._lib import SynHipError
    y8, x8 = synth.synth_clip_inputs(8, seed=22), synth.synth_latent(8, seed=22)
    t8 = torch.tensor([0, 2, 51, 334, 501, 778, 997, 999])
    cond8 = beatx.variant_conds(synth.to_device(y8, DEV), [(False, False, None)])[0]
    B, R = 4096, 512
    sb = engine.StepBuffers(B, 1, DEV)
    sb.cond.copy_(cond8.repeat(R, 1, 1).reshape(-1, 512)); sb.load_x(x8.to(DEV).repeat(R, 1, 1, 1))
    sb.t_model.copy_(t8.int().to(DEV).repeat(R)); sb.t_coef.zero_()
    engine.run_step(beatx.packed(), sb, engine.identity_coefs(DEV), False)
    out = sb.read(sb.x)
    assert torch.isfinite(out).all() and torch.equal(out, out[:8].repeat(R, 1, 1, 1))
    del sb, out
    torch.cuda.empty_cache()
    with pytest.raises((SynHipError, ValueError, RuntimeError)):
        beatx(torch.zeros(0, 1536, 1, 32, device=DEV), torch.zeros(0, dtype=torch.long, device=DEV),
              synth.to_device(synth.synth_clip_inputs(1, seed=1), DEV))


@pytest.mark.parametrize("B,V", [(9, 1), (17, 1), (40, 1), (64, 1), (65, 1), (100, 1), (128, 1), (10, 4), (5, 2)])
def test_tile_split_over_workgroups_equals_one_workgroup_per_tile(beatx, B, V):
    """9..128 sequences: the library splits every 32-row tile of the whole-step kernel over 4 (<= 64 sequences) or 2
    workgroups of one XCD (heads / MLP slices / output chunks dealt to the members, partial residual streams summed in
    member order).  Against the same kernel with one workgroup per tile (layer_mode 12) the only difference is the
    association of fp32 partial sums (which moves a bf16 evaluation by its noise floor, see test_batch_rows_are_independent);
    both must match the oracle, runs must be bitwise reproducible, ragged group
    counts (9, 17, 65, 100: padding groups on some XCDs) included."""
    from oracle import denoiser_ref as dr
    from syntalker_amd import engine
    n = min(B, 4)
    y, x = synth.synth_clip_inputs(n, seed=31), synth.synth_latent(n, seed=31)
    t = torch.tensor([3, 250, 600, 999][:n])
    rep = (B + n - 1) // n
    xs = x.repeat(rep, 1, 1, 1)[:B].to(DEV)
    ts = t.repeat(rep)[:B].to(DEV)
    flags = [(False, False, None), (True, False, None), (False, True, None), (True, True, None)][:V]
    conds = beatx.variant_conds(synth.to_device(y, DEV), flags)                     # V x (n, 32, 512)
    outs = {}
    for mode in (0, 12, 0):                        # 12 = whole-step kernel pinned, one workgroup per tile
        sb = engine.StepBuffers(B, V, DEV, layer_mode=mode)
        sb.cond.copy_(torch.cat([c.repeat(rep, 1, 1)[:B] for c in conds]).reshape(-1, 512))
        if V > 1:
            sb.cfg_w.copy_(torch.tensor([[-1.5, 1.0, 0.5, 1.0][:V] if V == 4 else [-1.5, 2.5]] * 3, device=DEV))
        sb.load_x(xs); sb.t_model.copy_(ts.int().repeat(V)); sb.t_coef.zero_()
        engine.run_step(beatx.packed(), sb, engine.identity_coefs(DEV), False)
        got = sb.read(sb.x).cpu()
        if mode in outs:
            assert torch.equal(got, outs[mode])                                    # reproducible
        outs[mode] = got
    assert torch.isfinite(outs[0]).all()
    assert rel_l2(outs[0], outs[12]) < 1e-2        # 3e-3 measured: fp32 re-association re-rolls the bf16 roundings downstream
    assert torch.equal(outs[0][:n], outs[0][n:2 * n]) or B < 2 * n                  # copies of a clip agree wherever they sit
    if V == 1:
        with torch.no_grad():
            want = dr.mdm_forward(synth_state_dict("beatx"), x, t, y)
        assert rel_l2(outs[0][:n], want) < FWD_TOL


@pytest.mark.parametrize("B", [256, 1024], ids=["256-token-resident-kernel", "1024-wave-per-sequence-kernel"])
def test_full_size_properties(beatx, B):
    """BASELINE-size batches (256 clips: k_stack; 1024 clips, the bench's batch: the library keeps the latent in fragment
    order and runs k_seq): properties that need no oracle.
       (1) t=0 step adds no noise: result independent of the injected noise;
       (2) the posterior update is the stated linear form of (x0_hat, x_t, eps): checked by running the
           same step with pred_x0 captured and recombining in fp64;
       (3) seeded in-library noise is reproducible and differs across seeds."""
    from syntalker_amd import engine
    from syntalker_amd.process import create_gaussian_diffusion
    d = create_gaussian_diffusion()
    assert beatx.step_buffers(B, 1).fragment == (B == 1024)
    y1 = synth.synth_clip_inputs(4, seed=31)
    y = {k: (v.repeat(B // 4, *([1] * (v.dim() - 1))) if torch.is_tensor(v) else v) for k, v in y1.items()}
    y = synth.to_device(y, DEV)
    xT = torch.randn(B, 1536, 1, 32, generator=torch.Generator().manual_seed(1)).to(DEV)
    a = d.p_sample_loop(beatx, (B, 1536, 1, 32), noise=xT.clone(), clip_denoised=False, model_kwargs={"y": y},
                        skip_timesteps=999, seed=1)
    b = d.p_sample_loop(beatx, (B, 1536, 1, 32), noise=xT.clone(), clip_denoised=False, model_kwargs={"y": y},
                        skip_timesteps=999, seed=2)
    assert torch.equal(a, b)                                        # (1) single step at t=0
    s1 = d.p_sample_loop(beatx, (B, 1536, 1, 32), noise=xT.clone(), clip_denoised=False, model_kwargs={"y": y},
                         skip_timesteps=997, seed=5)
    s2 = d.p_sample_loop(beatx, (B, 1536, 1, 32), noise=xT.clone(), clip_denoised=False, model_kwargs={"y": y},
                         skip_timesteps=997, seed=5)
    s3 = d.p_sample_loop(beatx, (B, 1536, 1, 32), noise=xT.clone(), clip_denoised=False, model_kwargs={"y": y},
                         skip_timesteps=997, seed=6)
    assert torch.equal(s1, s2) and not torch.equal(s1, s3)          # (3)
    assert torch.isfinite(s1).all()
    # (2): one explicit step at t=500 with captured x0_hat
    pm, sb = beatx.packed(), beatx.step_buffers(B, 1, want_x0=True)
    sb.cond.copy_(beatx.variant_conds(y, [(False, False, None)]).reshape(-1, 512))
    sb.load_x(xT)
    x_before = sb.x.clone()
    sb.t_model.fill_(500); sb.t_coef.fill_(500)
    sb.draw_noise(9, 500)                     # the separate generator kernel ...
    sb.set_rng(9, 0)                          # ... and the in-epilogue generator must agree bit for bit
    coef = engine.posterior_coefs(d.tables(), DEV)
    engine.run_step(pm, sb, coef, True, fused_rng=True)
    torch.cuda.synchronize()
    c = coef[500].double()
    want = c[0] * sb.x0.double() + c[1] * x_before.double() + c[2] * sb.noise.double()
    assert rel_l2(sb.x.double().cpu(), want.cpu()) < 1e-6
    # identical clips (the batch tiles 4 distinct clips) produce identical rows of x0_hat
    x0 = sb.read(sb.x0)
    assert rel_l2(x0[0:4].cpu(), x0[4:8].cpu()) > 1e-3              # different x_T rows -> different outputs


# ---------------------------------------------------------------------------------------------------------
# training path: loss value, gradient norms vs the reference goldens, full gradients vs the oracle
def test_ddim_loop_after_a_ddpm_loop_on_the_same_buffers(beatx):
    """A model's StepBuffers are cached per batch size: a DDIM-50 loop (50-row coefficient table) that follows a DDPM loop (1000
    rows, timestep vectors left at indices up to 999) must not launch with the stale indices (regression: the scheduled graph's
    warm-up launch read row 999 of the 50-row table - an out-of-bounds read that faulted in `bench.py --mode guided`)."""
    from syntalker_amd.process import create_gaussian_diffusion
    y = synth.to_device(synth.synth_clip_inputs(3, seed=61), DEV)
    xT = synth.synth_latent(3, seed=61).to(DEV)
    ddim = create_gaussian_diffusion(use_ddim=True)
    fresh = _model("beatx")
    want = ddim.ddim_sample_loop(fresh, (3, 1536, 1, 32), noise=xT, clip_denoised=False, model_kwargs={"y": y})
    create_gaussian_diffusion().p_sample_loop(beatx, (3, 1536, 1, 32), noise=xT, clip_denoised=False, model_kwargs={"y": y},
                                              skip_timesteps=0, seed=5)          # leaves t_coef = 0 after the last step ...
    sb = beatx.step_buffers(3, 1)
    sb.t_coef.fill_(999); sb.t_model.fill_(999)                                # ... so put the worst case there explicitly
    got = ddim.ddim_sample_loop(beatx, (3, 1536, 1, 32), noise=xT, clip_denoised=False, model_kwargs={"y": y})
    assert torch.isfinite(got).all() and torch.equal(got, want)


def test_training_loss_and_gradients(golden):
    from oracle import denoiser_ref as dr
    from oracle.process_ref import RefProcess
    from syntalker_amd.process import create_gaussian_diffusion
    m = _model("beatx")
    m.differentiable_eval = True                      # the goldens were taken with an eval()-mode model (deterministic)
    y = synth.synth_clip_inputs(4, seed=5)
    x0, eps = synth.synth_latent(4, seed=5, name="x0"), synth.synth_latent(4, seed=6, name="eps")
    t4 = torch.tensor([0, 17, 500, 999])
    d = create_gaussian_diffusion()
    terms = d.training_losses(m, x0.to(DEV), t4.to(DEV), model_kwargs={"y": synth.to_device(y, DEV)}, noise=eps.to(DEV))
    loss = terms["loss"]
    assert np.allclose(loss.detach().cpu().numpy(), golden["beatx.train.loss"], rtol=2e-2)
    loss.mean().backward()
    params = dict(m.named_parameters())
    names = [str(n) for n in golden["beatx.train.gradnorm_names"]]
    got = np.array([params[n].grad.norm().item() for n in names])
    print("grad norms got/want:", got / golden["beatx.train.gradnorm"])
    assert np.allclose(got, golden["beatx.train.gradnorm"], rtol=3e-2)
    # full gradients of every parameter against autograd through the CPU oracle
    buffers = ("running_mean", "running_var", ".pe", "inv_freq")
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(buffers)) for k, v in synth_state_dict("beatx").items()}
    ref = RefProcess(False).training_losses(lambda a, b, c: dr.mdm_forward(sd, a, b, c), x0, t4, y, eps)["loss"].mean()
    ref.backward()
    worst = 0.0
    for n, p in params.items():
        if p.grad is None or n not in sd or sd[n].grad is None or float(sd[n].grad.norm()) == 0.0:
            continue
        e = rel_l2(p.grad.cpu(), sd[n].grad)
        worst = max(worst, e)
        assert e < 3e-2, (n, e)                      # bf16 GEMM operands in forward, dgrad and wgrad
    print(f"worst per-tensor gradient rel-L2 vs oracle: {worst:.3e}")
    assert params["embed_style.weight"].grad is None     # never used in forward


def test_train_mode_loss_gradients_and_bn_buffers_vs_golden(golden):
    """The path `bench.py --mode train` and the reference's `_g_training` (diffusion_rvqvae_trainer.py:339-356) run: model.train(),
    BatchNorm on BATCH statistics through the fused `BnActFn` tail (conv bias dropped and re-added into the running mean, sums from
    the convolution's epilogue, first layer on `ConvFirstFn`), against the reference in train() mode with DropPath's probability 0
    (tests/golden/make_golden.py): loss, gradient norms, the BatchNorm buffers after the forward, and every parameter's
    gradient against autograd through the oracle's train-mode branch (gaussian_diffusion.py:1236-1363, models/utils/layer.py:144-184)."""
    from oracle import denoiser_ref as dr
    from oracle.process_ref import RefProcess
    from syntalker_amd.process import create_gaussian_diffusion
    m = _model("beatx").train()
    m.drop_path = 0.0                                  # (the golden's only departure from train(): DropPath is random)
    y = synth.synth_clip_inputs(4, seed=5)
    x0, eps = synth.synth_latent(4, seed=5, name="x0"), synth.synth_latent(4, seed=6, name="eps")
    t4 = torch.tensor([0, 17, 500, 999])
    d = create_gaussian_diffusion()
    terms = d.training_losses(m, x0.to(DEV), t4.to(DEV), model_kwargs={"y": synth.to_device(y, DEV)}, noise=eps.to(DEV))
    loss = terms["loss"]
    print("train-mode loss got / want:", loss.detach().cpu().numpy() / golden["beatx.trainmode.loss"])
    assert np.allclose(loss.detach().cpu().numpy(), golden["beatx.trainmode.loss"], rtol=2e-2)
    assert not np.allclose(golden["beatx.trainmode.loss"], golden["beatx.train.loss"], rtol=1e-3)     # (it is not the eval-mode number)
    loss.mean().backward()
    params = dict(m.named_parameters())
    names = [str(n) for n in golden["beatx.train.gradnorm_names"]]
    got = np.array([params[n].grad.norm().item() for n in names])
    print("train-mode grad norms got/want:", got / golden["beatx.trainmode.gradnorm"])
    assert np.allclose(got, golden["beatx.trainmode.gradnorm"], rtol=3e-2)
    # BatchNorm buffers after ONE training forward: running = 0.9 old + 0.1 batch (unbiased variance), counters incremented
    sd_after = m.state_dict()
    for key, name in [(f"beatx.trainmode.bn.{i}.bn1.{b}", f"WavEncoder.feat_extractor.{i}.bn1.{b}") for i in (0, 3, 5)
                      for b in ("running_mean", "running_var", "num_batches_tracked")] + \
                     [("beatx.trainmode.bn.0.downsample.running_mean", "WavEncoder.feat_extractor.0.downsample.1.running_mean"),
                      ("beatx.trainmode.bn.5.bn2.running_var", "WavEncoder.feat_extractor.5.bn2.running_var")]:
        have, want = sd_after[name].double().cpu().numpy(), golden[key]
        assert np.allclose(have, want, rtol=2e-3, atol=2e-4), (name, np.abs(have - want).max())
    # every parameter's gradient against autograd through the oracle in the same mode
    buffers = ("running_mean", "running_var", "num_batches_tracked", ".pe", "inv_freq")
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(buffers)) for k, v in synth_state_dict("beatx").items()}
    ref = RefProcess(False).training_losses(lambda a, b, c: dr.mdm_forward(sd, a, b, c, train_bn=True), x0, t4, y, eps)["loss"].mean()
    ref.backward()
    worst = 0.0
    for n, p in params.items():
        if p.grad is None or n not in sd or sd[n].grad is None or float(sd[n].grad.norm()) == 0.0:
            continue
        if float(sd[n].grad.norm()) < 1e-5:             # a convolution's bias in front of a batch-statistics BatchNorm: zero up to
            assert float(p.grad.norm()) < 1e-5, n       # rounding in the oracle (3e-8), exactly zero here
            continue
        e = rel_l2(p.grad.cpu(), sd[n].grad)
        worst = max(worst, e)
        assert e < 3e-2, (n, e)
    print(f"train mode: worst per-tensor gradient rel-L2 vs oracle: {worst:.3e}")
    # conv biases in front of a batch-statistics BatchNorm take exactly zero gradient in both
    assert float(params["WavEncoder.feat_extractor.1.conv1.bias"].grad.abs().max()) < 1e-6


@pytest.mark.parametrize("variant,B", [("beatx", 5), ("h3d", 3), ("beatx", 1)])
def test_training_step_at_batch_sizes_that_are_not_multiples_of_four(variant, B):
    """The block kernels and the GEMM pairs work on tiles of 4 clips; other batch sizes get empty clips appended behind the audio encoder (zero rows,
    DropPath factor 1) whose rows carry zero gradients and are cut from the output (`training.train_forward`).  5, 3 and 1 clips - the text-prompt variant
    with its style input and input_process3 among them - in train mode: loss and every parameter gradient against autograd through the oracle."""
    from oracle import denoiser_ref as dr
    from oracle.process_ref import RefProcess
    from syntalker_amd.process import create_gaussian_diffusion
    m = _model(variant).train()
    m.drop_path = 0.0
    m.cond_mask_prob = 0.0                             # (h3d: the Bernoulli style dropout is random)
    y = synth.synth_clip_inputs(B, seed=15, style_dim=256, style_zero=False) if variant == "h3d" else synth.synth_clip_inputs(B, seed=15)
    x0, eps = synth.synth_latent(B, seed=15, name="x0"), synth.synth_latent(B, seed=16, name="eps")
    t = (torch.arange(B) * 211 + 3) % 1000
    d = create_gaussian_diffusion()
    loss = d.training_losses(m, x0.to(DEV), t.to(DEV), model_kwargs={"y": synth.to_device(y, DEV)}, noise=eps.to(DEV))["loss"]
    assert loss.shape == (B,)
    loss.mean().backward()
    buffers = ("running_mean", "running_var", "num_batches_tracked", ".pe", "inv_freq")
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(buffers)) for k, v in synth_state_dict(variant).items()}
    fwd = (lambda a, b, c: dr.mdm_forward(sd, a, b, c, train_bn=True, variant="h3d")) if variant == "h3d" else (lambda a, b, c: dr.mdm_forward(sd, a, b, c, train_bn=True))
    ref = RefProcess(False).training_losses(fwd, x0, t, y, eps)["loss"]
    assert np.allclose(loss.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-2), (loss, ref)
    ref.mean().backward()
    worst = ("", 0.0)
    for n, p in m.named_parameters():
        if p.grad is None or n not in sd or sd[n].grad is None or float(sd[n].grad.norm()) < 1e-5:
            continue
        e = rel_l2(p.grad.cpu(), sd[n].grad)
        worst = max(worst, (n, e), key=lambda v: v[1])
        assert e < 3e-2, (n, e)
    print(f"{variant}, {B} clips: worst per-tensor gradient rel-L2 vs oracle {worst[1]:.3e} ({worst[0]})")


@pytest.mark.parametrize("variant", ["beatx", "h3d"])
def test_training_step_beyond_64_clips_equals_the_step_on_half_the_batch_twice(variant):
    """Above 64 clips the eight blocks leave the persistent kernels for the per-branch nodes (`training.AttnBranchFn` / `MlpBranchFn`), and the 9 216 word
    positions of 72 clips reach the embedding gradient in two calls (`training.InputStageFn.backward`).  A size-independent property instead of the oracle
    (its autograd at 72 clips takes minutes): a batch that holds 36 clips twice has the BatchNorm statistics, the mean loss and the mean gradient of
    the 36 clips - which run on the persistent kernels."""
    from syntalker_amd.process import create_gaussian_diffusion
    half, d = 36, create_gaussian_diffusion()
    y = synth.synth_clip_inputs(half, seed=21, style_dim=256, style_zero=False) if variant == "h3d" else synth.synth_clip_inputs(half, seed=21)
    x0, eps = synth.synth_latent(half, seed=21, name="x0"), synth.synth_latent(half, seed=22, name="eps")
    t = (torch.arange(half) * 83 + 11) % 1000
    twice = lambda v: torch.cat([v, v]) if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == half else v
    runs = []
    for rep in (False, True):
        m = _model(variant).train()
        m.drop_path = 0.0
        m.cond_mask_prob = 0.0
        yy = {k: (twice(v) if rep else v) for k, v in y.items()}
        a, b, c = (twice(v) if rep else v for v in (x0, t, eps))
        loss = d.training_losses(m, a.to(DEV), b.to(DEV), model_kwargs={"y": synth.to_device(yy, DEV)}, noise=c.to(DEV))["loss"]
        loss.mean().backward()
        torch.cuda.synchronize()
        runs.append((loss.detach().cpu(), {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None},
                     {n: b_.detach().cpu().clone() for n, b_ in m.named_buffers() if "running" in n}))
    (l1, g1, b1), (l2, g2, b2) = runs
    assert l2.shape == (2 * half,)
    assert torch.allclose(l2[:half], l2[half:], rtol=1e-5, atol=1e-7)          # the two copies of a clip
    assert torch.allclose(l2[:half], l1, rtol=5e-3), float((l2[:half] - l1).abs().max())
    assert g1.keys() == g2.keys()
    worst = max(((n, rel_l2(g2[n], g1[n])) for n in g1 if float(g1[n].norm()) > 1e-5), key=lambda v: v[1])
    print(f"{variant}: 72 clips on the per-branch nodes vs 36 on the persistent kernels, worst per-tensor gradient rel-L2 {worst[1]:.3e} ({worst[0]})")
    assert worst[1] < 2e-2, worst
    for n in b1:
        if n.endswith("running_mean"):
            assert rel_l2(b2[n], b1[n]) < 1e-4, n
        # (running_var takes the UNBIASED batch variance, rows / (rows - 1): not the same number for twice the rows - off by ~1e-6 here)
        else:
            assert rel_l2(b2[n], b1[n]) < 1e-3, n


@pytest.mark.parametrize("B", [40, 64, 66, 128, 130, 200])
def test_training_step_runs_at_every_batch_size_class(B):
    """Shape classes of the training step that no golden covers: above 32 clips (a weight gradient's contraction beyond the 128-column GEMM's resident
    block), the persistent kernels' last size (64), the per-branch nodes behind it (66: padded to 68), 128 / 130 / 200 clips (16 384+ word positions: the
    embedding gradient in several calls; GEMMs beyond 2048 rows on the larger row tiles).  A batch made of ONE clip B times: every clip's loss is the
    single clip's batch-of-4 loss, and the gradients are finite - the arithmetic itself is pinned by the tests around this one."""
    from syntalker_amd.process import create_gaussian_diffusion
    d = create_gaussian_diffusion()
    y1 = synth.synth_clip_inputs(1, seed=31)
    x1, e1 = synth.synth_latent(1, seed=31, name="x0"), synth.synth_latent(1, seed=32, name="eps")
    losses = []
    for n in (4, B):
        m = _model().train()
        m.drop_path = 0.0
        rep = lambda v: v.expand(n, *v.shape[1:]).contiguous() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 1 else v
        yy = {k: rep(v) for k, v in y1.items()}
        t = torch.full((n,), 417)
        loss = d.training_losses(m, rep(x1).to(DEV), t.to(DEV), model_kwargs={"y": synth.to_device(yy, DEV)}, noise=rep(e1).to(DEV))["loss"]
        loss.mean().backward()
        torch.cuda.synchronize()
        assert loss.shape == (n,) and bool(torch.isfinite(loss).all())
        for name, p in m.named_parameters():
            assert p.grad is None or bool(torch.isfinite(p.grad).all()), name
        losses.append(loss.detach().cpu())
    assert torch.allclose(losses[1], losses[1][:1].expand(B), rtol=1e-5)
    assert torch.allclose(losses[1][:4], losses[0], rtol=5e-3), (losses[0], losses[1][:4])


def test_fused_wav_block_equals_the_per_convolution_nodes():
    """training.WavBlockFn (round 5: a BasicBlock of the audio encoder as one autograd node - bn1 + LeakyReLU applied by conv2 as it stages its
    tile, the shortcut's BatchNorm inside the block's one elementwise pass, one statistics + one apply pass for both BatchNorms in the backward,
    no z1 / shortcut / output saved) against the per-convolution nodes (`_conv_bn_act`: the path SyncBatchNorm takes, and the one the train-mode
    golden was first pinned on): same loss, every parameter gradient, the BatchNorm buffers.  Differences are fp32 reassociation (a(v) = v * scale + shift
    against (v - mean) * rstd * gamma + beta) on top of the split-operand convolutions."""
    from syntalker_amd import training
    from syntalker_amd.process import create_gaussian_diffusion
    y = synth.to_device(synth.synth_clip_inputs(4, seed=5), DEV)
    x0, eps = synth.synth_latent(4, seed=5, name="x0").to(DEV), synth.synth_latent(4, seed=6, name="eps").to(DEV)
    t4 = torch.tensor([0, 17, 500, 999], device=DEV)
    d = create_gaussian_diffusion()
    res = {}
    for fused in (False, True):
        m = _model("beatx").train()
        if not fused:                                   # SyncBatchNorm (train.py:90; one rank here) takes the per-convolution nodes
            m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m)
            assert not training._wav_block_fused_ok(m.WavEncoder.feat_extractor[1], torch.zeros(1, device=DEV), False)
        m.drop_path = 0.0
        loss = d.training_losses(m, x0, t4, model_kwargs={"y": y}, noise=eps)["loss"]
        loss.mean().backward()
        res[fused] = (loss.detach().cpu(), {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None},
                      {k: v.detach().cpu() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k})
    (l0, g0, b0), (l1, g1, b1) = res[False], res[True]
    assert torch.allclose(l0, l1, rtol=5e-4), (l0, l1)      # (bf16 operands downstream: an fp32-rounding difference in the conditioning flips bf16 roundings)
    assert g0.keys() == g1.keys()
    worst = ("", 0.0)
    for n in g0:
        if float(g0[n].norm()) < 1e-6:                      # conv biases in front of a batch-statistics BatchNorm: exactly zero in both
            assert float(g1[n].norm()) < 1e-6, n
            continue
        e = rel_l2(g1[n], g0[n])
        worst = max(worst, (n, e), key=lambda v: v[1])
        # (the worst tensor is always block 0's conv1 weight: its gradient is what is left after BatchNorm's backward has projected
        # out the mean and the x-hat component, so fp32 reassociation - other tile sizes, other partial-sum orders - shows up amplified:
        # 9.4e-3 with the round-5 tiles, 1.2e-2 with round 5b's.  Both paths are within 2.2e-2 of the oracle's
        # autograd, which is the parity gate: test_train_mode_loss_gradients_and_bn_buffers_vs_golden)
        assert e < 2e-2, (n, e)
    print(f"fused block vs per-convolution nodes: loss ratio {float((l1 / l0).mean()):.7f}, worst gradient rel-L2 {worst[1]:.2e} ({worst[0]})")
    for k in b0:
        assert torch.allclose(b0[k].double(), b1[k].double(), rtol=1e-5, atol=1e-6), k


def test_persistent_stack_forward_equals_the_per_branch_nodes(monkeypatch):
    """training.StackFn (round 5: the eight blocks' training forward and backward as persistent launches, `syn_train_stack_fwd` = the sampling path's
    whole-step kernel in its tile-split mode + DropPath + the tensors the backward takes) against the per-branch nodes larger batches run (`AttnBranchFn` /
    `MlpBranchFn`): loss and every parameter gradient, with DropPath factors drawn from the same generator state.  The two forwards differ where
    the sampling kernels do: q / k / v and the softmax numerators are rounded to bf16 for the attention's MFMAs (fp32 in `syn_attn_fwd`)."""
    from syntalker_amd import training
    from syntalker_amd.process import create_gaussian_diffusion
    y = synth.to_device(synth.synth_clip_inputs(8, seed=5, mask_batch=8), DEV)
    x0, eps = synth.synth_latent(8, seed=5, name="x0").to(DEV), synth.synth_latent(8, seed=6, name="eps").to(DEV)
    t5 = torch.tensor([0, 17, 500, 999, 250, 3, 750, 100], device=DEV)      # (8 clips: the fused paths take row counts that are multiples of 128)
    d = create_gaussian_diffusion()
    res = {}
    for fused in (False, True):
        if not fused:
            monkeypatch.setattr(training, "_stack_ok", lambda *a: False)       # the path batches of more than 64 clips take
        else:
            monkeypatch.undo()
        m = _model("beatx").train()
        m.drop_path = 0.25
        torch.manual_seed(1234)
        loss = d.training_losses(m, x0, t5, model_kwargs={"y": y}, noise=eps)["loss"]
        loss.mean().backward()
        res[fused] = (loss.detach().cpu(), {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None})
    (l0, g0), (l1, g1) = res[False], res[True]
    print("persistent stack vs per-branch nodes, loss ratio:", (l1 / l0).numpy())
    assert torch.allclose(l0, l1, rtol=1e-2), (l0, l1)
    assert g0.keys() == g1.keys()
    worst = ("", 0.0)
    for n in g0:
        if float(g0[n].norm()) < 1e-6:
            assert float(g1[n].norm()) < 1e-6, n
            continue
        e = rel_l2(g1[n], g0[n])
        worst = max(worst, (n, e), key=lambda v: v[1])
        assert e < 3e-2, (n, e)
    print(f"persistent stack vs per-branch nodes: worst gradient rel-L2 {worst[1]:.2e} ({worst[0]})")


def test_sync_batchnorm_model_on_the_native_path(golden):
    """train.py:90 converts every BatchNorm of the model to nn.SyncBatchNorm before DDP: the converted model keeps the audio encoder on
    the hand-written kernels (`SyncBnActFn`: fp64 sums -> all-reduce -> finalise) and, with a one-rank process group, reproduces the
    train-mode golden of the reference like the unconverted one (the two-rank arithmetic is the kernel-level test's)."""
    import torch.distributed as dist
    from syntalker_amd.process import create_gaussian_diffusion
    created = False
    if not dist.is_initialized():
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        created = True
    try:
        m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(_model("beatx")).train()
        assert sum(isinstance(x, torch.nn.SyncBatchNorm) for x in m.modules()) == 16 and m.variant == "beatx"
        m.drop_path = 0.0
        y = synth.synth_clip_inputs(4, seed=5)
        x0, eps = synth.synth_latent(4, seed=5, name="x0"), synth.synth_latent(4, seed=6, name="eps")
        t4 = torch.tensor([0, 17, 500, 999])
        loss = create_gaussian_diffusion().training_losses(m, x0.to(DEV), t4.to(DEV), model_kwargs={"y": synth.to_device(y, DEV)}, noise=eps.to(DEV))["loss"]
        assert np.allclose(loss.detach().cpu().numpy(), golden["beatx.trainmode.loss"], rtol=2e-2)
        loss.mean().backward()
        params = dict(m.named_parameters())
        names = [str(n) for n in golden["beatx.train.gradnorm_names"]]
        got = np.array([params[n].grad.norm().item() for n in names])
        assert np.allclose(got, golden["beatx.trainmode.gradnorm"], rtol=3e-2), got / golden["beatx.trainmode.gradnorm"]
        sd_after = m.state_dict()
        assert np.allclose(sd_after["WavEncoder.feat_extractor.3.bn1.running_var"].double().cpu().numpy(), golden["beatx.trainmode.bn.3.bn1.running_var"],
                           rtol=2e-3, atol=2e-4)
        assert int(sd_after["WavEncoder.feat_extractor.3.bn1.num_batches_tracked"]) == 1
    finally:
        if created:
            dist.destroy_process_group()


def test_train_mode_step_runs_and_updates(beatx):
    """train(): BatchNorm batch statistics + DropPath; one Adam step (lr, betas of optimizers/optim_factory.py:122)."""
    from syntalker_amd import training
    from syntalker_amd.process import create_gaussian_diffusion
    from syntalker_amd.resample import create_named_schedule_sampler
    m = _model("h3d").train()
    d = create_gaussian_diffusion()
    opt = torch.optim.Adam(m.parameters(), lr=5e-5, betas=(0.5, 0.999))
    y = synth.to_device(synth.synth_clip_inputs(4, seed=51, style_dim=256, style_zero=False), DEV)
    x0 = synth.synth_latent(4, seed=51, name="x0").to(DEV)
    before = m.mytimmblocks[0].attn.qkv.weight.detach().clone()
    rm = m.WavEncoder.feat_extractor[0].bn1.running_mean.clone()
    np.random.seed(0); torch.manual_seed(0)
    l1 = training.train_step(m, d, create_named_schedule_sampler("uniform", d), opt, x0, {"y": y})
    l2 = training.train_step(m, d, create_named_schedule_sampler("uniform", d), opt, x0, {"y": y})
    assert torch.isfinite(l1) and torch.isfinite(l2)
    assert not torch.equal(before, m.mytimmblocks[0].attn.qkv.weight)            # parameters moved
    assert not torch.equal(rm, m.WavEncoder.feat_extractor[0].bn1.running_mean)  # BatchNorm used batch statistics
    # the h3d forward reads the learned null prompt (denoiser_h3d.py:119-122): it trains; the null audio embedding is never read
    assert m.uncon_text_embeddings.grad is not None and m.uncon_audio_embeddings.grad is None and m.embed_style.weight.grad is None
    assert training.unused_in_forward(m) == ("embed_style", "uncon_audio_embeddings")
    assert training.unused_in_forward(_model("beatx")) == ("embed_style",)
    m.eval()
    with torch.no_grad():                                                         # packed weights follow the update
        out = m(x0, torch.tensor([5, 6, 7, 8], device=DEV), y)
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("variant", ["beatx", "h3d"])
def test_used_model_deep_copies_and_pickles_like_an_nn_module(variant):
    """copy.deepcopy(model) (an EMA copy next to the trained model) and torch.save(model) / torch.load of a model that has already sampled and trained:
    its caches are ctypes structs of raw pointers into its own packed tensors and must not travel (`MDM.__getstate__`).  The copy samples bit-equal,
    trains, and is independent of the original."""
    import copy
    import io
    from syntalker_amd.process import create_gaussian_diffusion
    d = create_gaussian_diffusion()
    m = _model(variant)
    y = synth.synth_clip_inputs(4, seed=91, style_dim=256, style_zero=False) if variant == "h3d" else synth.synth_clip_inputs(4, seed=91)
    y = synth.to_device(y, DEV)
    x0, eps = synth.synth_latent(4, seed=91, name="x0").to(DEV), synth.synth_latent(4, seed=92, name="eps").to(DEV)
    tt = torch.tensor([10, 200, 600, 990], device=DEV)

    def train_once(mod):
        mod.train()
        mod.drop_path, mod.cond_mask_prob = 0.0, 0.0
        mod.zero_grad(set_to_none=True)
        d.training_losses(mod, x0, tt, model_kwargs={"y": y}, noise=eps)["loss"].mean().backward()
        mod.eval()

    with torch.no_grad():
        m(x0, tt, y)
        d.p_sample_loop(m, tuple(x0.shape), clip_denoised=False, model_kwargs={"y": y}, progress=False, skip_timesteps=997)      # (captured loops cached on the model)
    train_once(m)                                      # (moves the BatchNorm running statistics)
    with torch.no_grad():
        o1 = m(x0, tt, y)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    for m2 in (copy.deepcopy(m), torch.load(buf, weights_only=False)):
        assert m2._packed is None and not any(k.startswith("_syn_") or k == "_graphs" for k in m2.__dict__)
        with torch.no_grad():
            assert torch.equal(m2(x0, tt, y), o1)
        train_once(m2)
        g, g2 = m.output_process.poseFinal.weight.grad, m2.output_process.poseFinal.weight.grad
        assert g2 is not None and g2.data_ptr() != g.data_ptr() and torch.equal(g, g2)
        with torch.no_grad():
            m2.output_process.poseFinal.weight.mul_(0.5)                       # the copy's weights are its own
            assert not torch.equal(m2(x0, tt, y), o1) and torch.equal(m(x0, tt, y), o1)


@pytest.mark.parametrize("variant", ["beatx", "h3d"])
def test_frozen_parameter_groups_weighted_losses_and_accumulation(variant):
    """What a fine-tuning script does to the reference model and autograd takes in its stride; here every group of operations is ONE hand-written
    autograd node, so each case is the node's own bookkeeping: parameter groups with requires_grad False (the audio encoder; everything but the
    blocks; the blocks; the word embedding), the schedule sampler's per-sample weights on the loss (diffusion_rvqvae_trainer.py:345-349), two
    backward passes into the same .grad, and a train()-mode forward under no_grad (a validation loss without eval(): batch statistics, running
    statistics move).  Gradients of what stays trainable must equal the all-trainable run's."""
    from syntalker_amd.process import create_gaussian_diffusion
    d = create_gaussian_diffusion()
    B = 8
    y = synth.synth_clip_inputs(B, seed=81, style_dim=256, style_zero=False) if variant == "h3d" else synth.synth_clip_inputs(B, seed=81)
    y = synth.to_device(y, DEV)
    x0, eps = synth.synth_latent(B, seed=81, name="x0").to(DEV), synth.synth_latent(B, seed=82, name="eps").to(DEV)
    t = ((torch.arange(B) * 131 + 7) % 1000).to(DEV)
    wts = torch.linspace(0.5, 2.0, B, device=DEV)

    hit = lambda n, freeze: any(n == f.rstrip(".") or n.startswith(f) for f in freeze)

    def run(freeze=(), weights=None, passes=1):
        m = _model(variant).train()
        m.drop_path = 0.0
        m.cond_mask_prob = 0.0
        for n, p in m.named_parameters():
            if hit(n, freeze):
                p.requires_grad_(False)
        for _ in range(passes):
            loss = d.training_losses(m, x0, t, model_kwargs={"y": y}, noise=eps)["loss"]
            ((loss * weights).mean() if weights is not None else loss.mean()).backward()
        torch.cuda.synchronize()
        return loss.detach(), {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()}, m

    l0, g0, m0 = run()
    for freeze in (("WavEncoder.",), tuple(n.split(".")[0] + "." for n, _ in m0.named_parameters() if not n.startswith("mytimmblocks")), ("mytimmblocks.",),
                   ("text_pre_encoder_body.",), ("WavEncoder.feat_extractor.0.", "mytimmblocks.3.", "output_process."),
                   tuple(n for n, _ in m0.named_parameters() if n.endswith(".bias"))):
        l1, g1, _ = run(freeze=freeze)
        assert torch.equal(l1, l0), freeze
        for n in g0:
            if hit(n, freeze):
                assert g1[n] is None, (freeze[:2], n)
            elif g0[n] is None:
                assert g1[n] is None, n
            else:
                assert g1[n] is not None and rel_l2(g1[n].cpu(), g0[n].cpu()) < 1e-5, (freeze[:2], n, rel_l2(g1[n].cpu(), g0[n].cpu()))
    # per-sample weights: the gradient is linear in them - weights w and 2 w
    _, gw, _ = run(weights=wts)
    _, g2w, _ = run(weights=2 * wts)
    _, gacc, _ = run(weights=wts, passes=2)
    for n in gw:
        if gw[n] is None or float(gw[n].norm()) < 1e-6:
            continue
        assert rel_l2(g2w[n].cpu(), (2 * gw[n]).cpu()) < 1e-5, n
        # (the second pass starts from updated BatchNorm running statistics only - batch statistics are what the forward uses)
        assert rel_l2(gacc[n].cpu(), (2 * gw[n]).cpu()) < 1e-5, n
    assert rel_l2(gw["output_process.poseFinal.weight"].cpu(), g0["output_process.poseFinal.weight"].cpu()) > 1e-2      # (the weights do something)
    # train() under no_grad
    m = _model(variant).train()
    m.drop_path = 0.0
    m.cond_mask_prob = 0.0
    rm = m.WavEncoder.feat_extractor[2].bn2.running_mean.clone()
    with torch.no_grad():
        lv = d.training_losses(m, x0, t, model_kwargs={"y": y}, noise=eps)["loss"]
    assert torch.allclose(lv, l0, rtol=2e-3) and not lv.requires_grad       # (other kernels where nothing is saved for a backward: bf16-level differences)
    assert not torch.equal(rm, m.WavEncoder.feat_extractor[2].bn2.running_mean)


def test_two_forwards_then_one_backward_and_two_models_alive():
    """The fused residual branches carry the W^T fragments their forward saw on the autograd node - views of the model's per-step
    pack buffers, which the NEXT forward rewrites in place (training.WeightPacks.refresh).  Gradient accumulation (two forwards, then
    one backward through both) and a second model's forward between a forward and its backward must give the gradients of the
    plain sequence forward-backward, forward-backward."""
    from syntalker_amd.process import create_gaussian_diffusion
    d = create_gaussian_diffusion()
    t4 = torch.tensor([3, 170, 500, 998], device=DEV)

    def case(seed):
        y = synth.to_device(synth.synth_clip_inputs(4, seed=seed), DEV)
        return synth.synth_latent(4, seed=seed, name="x0").to(DEV), synth.synth_latent(4, seed=seed + 1, name="eps").to(DEV), y

    def loss(m, c):
        return d.training_losses(m, c[0], t4, model_kwargs={"y": c[2]}, noise=c[1])["loss"].mean()

    def grads(m):
        g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        m.zero_grad(set_to_none=True)
        return g
    m = _model("beatx")
    m.differentiable_eval = True
    a, b = case(71), case(73)
    loss(m, a).backward(); ga = grads(m)
    loss(m, b).backward(); gb = grads(m)
    la, lb = loss(m, a), loss(m, b)                   # two forwards (the second refreshes the pack buffers the first one's nodes hold)
    (la + lb).backward()
    both = grads(m)
    assert set(both) == set(ga) == set(gb) and "mytimmblocks.0.attn.qkv.weight" in both
    for n in both:
        assert rel_l2(both[n], ga[n] + gb[n]) < 1e-5, n
    from syntalker_amd.denoiser import MDM
    other = MDM(synth.default_args()).eval()
    other.load_state_dict({k: v * 1.5 if v.is_floating_point() and v.dim() == 2 else v for k, v in synth_state_dict("beatx").items()}, strict=False)
    other = other.to(DEV)
    other.differentiable_eval = True
    la = loss(m, a)
    lo = loss(other, b)                               # another model's forward (its own packs become the module-level current ones)
    la.backward()
    for n, g in grads(m).items():
        assert rel_l2(g, ga[n]) < 1e-6, n
    lo.backward()
    assert all(torch.isfinite(p.grad).all() for p in other.parameters() if p.grad is not None)


def test_guided_small_batch_both_group_layouts(h3d):
    """Small-batch kernel with guidance variants: (a) sequences dealt to the XCDs + k_guided_update (ws_x0v given),
    (b) whole clips per XCD with the combination inside the kernel (ws_x0v = NULL), (c) the whole-step kernel path.
    Same step, same noise: all three agree to the bf16 re-rounding floor, (a) and (b) are each deterministic."""
    from syntalker_amd import engine
    from syntalker_amd.process import create_gaussian_diffusion
    B, V = 3, 3
    pm = h3d.packed()
    g = torch.Generator().manual_seed(77)
    cond = (torch.randn(V * B * 32, 512, generator=g) * 0.5).to(DEV)
    x = torch.randn(B, 1536, 1, 32, generator=g).to(DEV)
    w = torch.tensor([[2.5, -1.0, -0.5], [1.0, 0.0, 0.0], [0.2, 0.3, 0.5]], device=DEV)
    coef = engine.ddim_coefs(create_gaussian_diffusion(use_ddim=True).tables(), 0.0, DEV)

    def run(mode, by_seq):
        sb = engine.StepBuffers(B, V, DEV, want_x0=True, layer_mode=mode)
        if not by_seq:
            sb.c.ws_x0v = None
        sb.cond.copy_(cond); sb.cfg_w.copy_(w); sb.load_x(x); sb.set_rng(5, 0)
        sb.t_model.copy_(torch.tensor([40] * (V * B), dtype=torch.int32)); sb.t_coef.fill_(40)
        engine.run_step(pm, sb, coef, True, True)
        return sb.read(sb.x).cpu(), sb.read(sb.x0).cpu()

    a1, a0 = run(3, True)
    b1, b0 = run(3, False)
    c1, c0 = run(4, True)
    assert torch.equal(run(3, True)[0], a1) and torch.equal(run(3, False)[0], b1)
    for got in (a0, b0):
        assert rel_l2(got, c0) < 1.5e-2
    assert rel_l2(a1, c1) < 1.5e-2 and rel_l2(b1, c1) < 1.5e-2


@pytest.mark.parametrize("how", ["captured-step", "eager-clipadam-eval-bn"])
def test_sampling_after_training_sees_the_trained_weights(how):
    """The reference's trainer samples between epochs (diffusion_rvqvae_trainer.py: `val` / `test` from `train`).  The sampling kernels read folded, packed
    copies of the weights (`MDM.packed()`), cached against the tensors' in-place version counters - which neither a hipGraph replay nor `ClipAdam`'s
    raw-pointer update moves.  After training either way, the model's samples must equal those of a FRESH model loaded from its state_dict."""
    from syntalker_amd import training
    from syntalker_amd.process import create_gaussian_diffusion
    d = create_gaussian_diffusion()
    m = _model("beatx")
    y = synth.to_device(synth.synth_clip_inputs(4, seed=71), DEV)
    x0 = synth.synth_latent(4, seed=71, name="x0").to(DEV)
    t = torch.tensor([100, 300, 500, 700], device=DEV)
    xT = synth.synth_latent(4, seed=72, name="xT").to(DEV)

    def sample(mod):
        mod.eval()
        with torch.no_grad():
            return d.p_sample_loop(mod, tuple(x0.shape), noise=xT, clip_denoised=False, model_kwargs={"y": y}, progress=False, skip_timesteps=995, seed=9).clone()

    before = sample(m)                                     # the packed copies exist from here on
    opt = training.ClipAdam(m.parameters(), lr=1e-3, max_norm=0.99)
    if how == "captured-step":
        m.train()
        step = training.GraphedTrainStep(m, d, opt, x0, {"y": y})
        before = sample(m)                                 # (constructing the step restored the weights with tensor ops: pack again, THEN replay)
        m.train()
        for _ in range(3):
            step(x0, t, {"y": y})
        torch.cuda.synchronize()
        step.close()
    else:
        m.eval()
        m.differentiable_eval = True                       # fine-tuning with frozen BatchNorm statistics: nothing in the step touches a version counter
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            d.training_losses(m, x0, t, model_kwargs={"y": y})["loss"].mean().backward()
            opt.step()
        m.differentiable_eval = False
    after = sample(m)
    fresh = _model("beatx")
    fresh.load_state_dict(m.state_dict())
    want = sample(fresh)
    assert rel_l2(after.cpu(), before.cpu()) > 1e-3        # three steps at lr 1e-3 move the samples
    assert torch.equal(after, want), rel_l2(after.cpu(), want.cpu())


def test_graph_replayed_train_step(beatx):
    """training.GraphedTrainStep: the captured step trains (parameters move, loss finite and falling on a fixed batch)."""
    from syntalker_amd import training
    from syntalker_amd.process import create_gaussian_diffusion
    m = _model("beatx").train()
    d = create_gaussian_diffusion()
    opt = torch.optim.Adam(m.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True)
    y = synth.to_device(synth.synth_clip_inputs(4, seed=61), DEV)
    x0 = synth.synth_latent(4, seed=61, name="x0").to(DEV)
    t = torch.tensor([100, 300, 500, 700], device=DEV)
    w0 = m.mytimmblocks[0].attn.qkv.weight.detach().clone()
    bn = m.WavEncoder.feat_extractor[0].bn1
    rm0, nb0 = bn.running_mean.clone(), int(bn.num_batches_tracked)
    step = training.GraphedTrainStep(m, d, opt, x0, {"y": y})
    # the warm-up iterations in front of the capture are real optimizer steps (construction batch, t = 0): they are undone, so the
    # first replay is update number 1 of the run (Adam's bias correction included) and BatchNorm's statistics start where they were
    assert torch.equal(w0, m.mytimmblocks[0].attn.qkv.weight) and torch.equal(rm0, bn.running_mean) and int(bn.num_batches_tracked) == nb0
    st = opt.state[m.mytimmblocks[0].attn.qkv.weight]
    assert float(st["step"]) == 0.0 and float(st["exp_avg"].abs().max()) == 0.0 and float(st["exp_avg_sq"].abs().max()) == 0.0
    before = m.mytimmblocks[0].attn.qkv.weight.detach().clone()
    losses = [float(step(x0, t, {"y": y})) for _ in range(25)]
    assert float(st["step"]) == 25.0 and int(bn.num_batches_tracked) == nb0 + 25
    # what the capture made a constant cannot change silently between replays: each of these raises and leaves the step usable
    for change, undo in ((lambda: opt.param_groups[0].__setitem__("lr", 1e-5), lambda: opt.param_groups[0].__setitem__("lr", 2e-4)),
                         (lambda: m.eval(), lambda: m.train()),
                         (lambda: setattr(m, "drop_path", 0.3), lambda: setattr(m, "drop_path", 0.1)),
                         (lambda: m.embed_text.weight.requires_grad_(False), lambda: m.embed_text.weight.requires_grad_(True))):
        dp0 = m.drop_path
        change()
        with pytest.raises(RuntimeError, match="captured step is fixed"):
            step(x0, t, {"y": y})
        undo()
        m.drop_path = dp0
    with pytest.raises(RuntimeError, match="captured step is fixed"):
        step(x0, t, {"y": dict(y, uncond=True)})
    with pytest.raises(RuntimeError, match="static shapes"):
        step(x0[:2], t[:2], {"y": y})
    losses.append(float(step(x0, t, {"y": y})))
    step.close()
    assert all(np.isfinite(losses)) and not torch.equal(before, m.mytimmblocks[0].attn.qkv.weight)
    assert np.mean(losses[-5:]) < np.mean(losses[:5])


def test_five_step_training_trajectory_vs_the_reference_loop():
    """The reference's own training loop - `_g_training` lifted from diffusion_rvqvae_trainer.py:339-356, its Adam (optimizers/optim_factory.py:
    122-123), clip_grad_norm_(0.99), five updates on one 4-clip batch with fixed timesteps / noise (tests/golden/make_train_golden.py ->
    train_trajectory.npz) - against this build's `GraphedTrainStep` + `ClipAdam` (the captured step bench.py times): the loss of every step
    (2.28 -> 1.49: each one carries the four updates before it), the gradient norm the clip saw, ten tensors' parameter changes, BatchNorm
    buffers.  Tolerances: loss 2e-2 (bf16 operands); parameter changes are compared by cosine: Adam's first updates are sign-like
    (|m / sqrt(v)| ~ 1 whatever the gradient's size), so an element whose gradient is below the bf16 error of its tensor moves by +-lr on
    either side - rel-L2 of the change measures the share of such elements, not an arithmetic error."""
    import os
    from syntalker_amd import training
    from syntalker_amd.process import create_gaussian_diffusion
    from tests.test_oracle_golden import trajectory_case
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_trajectory.npz"))
    y, x0, eps = trajectory_case()
    m = _model("beatx").train()
    m.drop_path = 0.0
    d = create_gaussian_diffusion()
    opt = training.ClipAdam(m.parameters(), lr=5e-5, betas=(0.5, 0.999), max_norm=0.99)
    params = dict(m.named_parameters())
    init = {str(n): params[str(n)].detach().clone() for n in g["watch"]}
    yd = synth.to_device(y, DEV)
    step = training.GraphedTrainStep(m, d, opt, x0.to(DEV), {"y": yd}, grad_norm=0.99, noise=eps[0].to(DEV))
    losses, norms = [], []
    for k in range(5):
        losses.append(float(step(x0.to(DEV), torch.from_numpy(g["t"][k]).to(DEV), {"y": yd}, noise=eps[k].to(DEV))))
        norms.append(float(opt.last_norm()))
    step.close()
    print("trajectory loss got / want:", np.array(losses) / g["loss"], " grad norm got / want:", np.array(norms) / g["grad_norm"])
    assert np.allclose(losses, g["loss"], rtol=2e-2), (losses, g["loss"])
    assert np.allclose(norms, g["grad_norm"], rtol=3e-2), (norms, g["grad_norm"])
    worst_cos, worst_norm = 1.0, 0.0
    for n in init:
        dlt = (params[n].detach() - init[n]).reshape(-1).cpu()
        want = torch.from_numpy(g[f"delta_head.{n}"])
        cos = float(torch.nn.functional.cosine_similarity(dlt[:4096], want, dim=0))
        nr = float(dlt.double().norm()) / float(g[f"delta_norm.{n}"])
        print(f"  {n}: delta norm got / want {nr:.4f}, cosine {cos:.4f}, rel-L2 {rel_l2(dlt[:4096], want):.3e}")
        worst_cos, worst_norm = min(worst_cos, cos), max(worst_norm, abs(nr - 1))
        assert cos > 0.97 and abs(nr - 1) < 5e-2, (n, cos, nr)
    print(f"trajectory: worst delta cosine {worst_cos:.4f}, worst |norm ratio - 1| {worst_norm:.3e}")
    sd = m.state_dict()
    for n in ("WavEncoder.feat_extractor.0.bn1.running_mean", "WavEncoder.feat_extractor.5.bn2.running_var", "WavEncoder.feat_extractor.0.bn1.num_batches_tracked"):
        assert np.allclose(sd[n].double().cpu().numpy(), g[f"buffer.{n}"], rtol=5e-3, atol=5e-4), n


def test_fifty_step_training_trajectory_does_not_drift_from_the_reference_loop():
    """Fifty updates of the reference's own training loop (`_g_training` + its Adam + clip_grad_norm_(0.99) on one 4-clip batch, fresh timesteps and noise
    every step: tests/golden/make_train_golden.py 50 -> train_trajectory_k50.npz) against `GraphedTrainStep` + `ClipAdam`.  Every step's loss depends on all
    the updates before it, so an arithmetic difference that compounded - bf16 GEMM operands, the two-product convolution weight gradients, bf16 attention
    operands in the block kernels - would show as a loss curve walking away from the reference's.  Asserted: every loss within 2e-2, every clipped gradient
    norm within 4e-2, the mean loss of the last ten steps within 1 %, and the audio encoder's BatchNorm running statistics after fifty training forwards."""
    import os
    from syntalker_amd import training
    from syntalker_amd.process import create_gaussian_diffusion
    from tests.test_oracle_golden import trajectory_case
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_trajectory_k50.npz"))
    K = int(g["steps"])
    y, x0, eps = trajectory_case(K)
    m = _model("beatx").train()
    m.drop_path = 0.0
    d = create_gaussian_diffusion()
    opt = training.ClipAdam(m.parameters(), lr=5e-5, betas=(0.5, 0.999), max_norm=0.99)
    yd = synth.to_device(y, DEV)
    step = training.GraphedTrainStep(m, d, opt, x0.to(DEV), {"y": yd}, grad_norm=0.99, noise=eps[0].to(DEV))
    losses, norms = [], []
    for k in range(K):
        losses.append(step(x0.to(DEV), torch.from_numpy(g["t"][k]).to(DEV), {"y": yd}, noise=eps[k].to(DEV)))
        norms.append(opt.last_norm().clone())
    training.check_stack_sync(DEV)
    step.close()
    losses, norms = np.array([float(v) for v in losses]), np.array([float(v) for v in norms])
    rl, rn = losses / g["loss"], norms / g["grad_norm"]
    print(f"50-step trajectory: loss ratio min {rl.min():.4f} max {rl.max():.4f}, first five {np.round(rl[:5], 4)}, last five {np.round(rl[-5:], 4)}; "
          f"gradient-norm ratio min {rn.min():.4f} max {rn.max():.4f}; reference loss {g['loss'][0]:.3f} -> {g['loss'][-1]:.3f}")
    assert np.allclose(losses, g["loss"], rtol=2e-2), (rl.min(), rl.max())
    assert np.allclose(norms, g["grad_norm"], rtol=4e-2), (rn.min(), rn.max())
    assert abs(losses[-10:].mean() / g["loss"][-10:].mean() - 1) < 1e-2
    sd = m.state_dict()
    worst = ("", 0.0)
    for key in g.files:
        if key.startswith("buffer.") and "running_" in key:
            # (statistics of activations behind fifty sign-like Adam updates of the layers in front: compared per tensor, not per element)
            e = rel_l2(torch.from_numpy(sd[key[7:]].double().cpu().numpy()), torch.from_numpy(g[key]))
            worst = max(worst, (key[7:], e), key=lambda v: v[1])
            assert e < 2e-2, (key, e)
    print(f"50-step trajectory: worst BatchNorm running statistic rel-L2 {worst[1]:.2e} ({worst[0]})")


def test_captured_train_step_at_the_bench_size_replays_back_to_back(beatx):
    """The whole training step of BASELINE config 3 (32 clips, 68 266 audio samples x 2 channels) captured in one hipGraph and
    replayed 40 times with NOTHING waiting between the replays.  Round 1 had to synchronise after every replay (HSA
    memory-aperture violation otherwise, and with the bias-gradient partial sums even then); the op behind it was PyTorch-ROCm's
    embedding_dense_backward, replaced by training.EmbeddingFn.  Loss finite and falling on a fixed batch."""
    from syntalker_amd import training
    from syntalker_amd.process import create_gaussian_diffusion
    B = 32
    m = _model("beatx").train()
    d = create_gaussian_diffusion()
    opt = torch.optim.Adam(m.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True, fused=True)
    y = synth.to_device(synth.synth_clip_inputs(B, seed=62, mask_batch=B), DEV)
    y["audio"] = torch.randn(B, 68266, 2, generator=torch.Generator().manual_seed(62)).to(DEV)
    x0 = synth.synth_latent(B, seed=62, name="x0").to(DEV)
    t = (torch.arange(B, device=DEV) * 31) % 1000
    step = training.GraphedTrainStep(m, d, opt, x0, {"y": y})
    losses = [step(x0, t, {"y": y}).clone() for _ in range(40)]          # (clones are stream-ordered copies of the static loss)
    torch.cuda.synchronize()
    step.close()
    losses = [float(v) for v in losses]
    assert all(np.isfinite(losses)) and np.mean(losses[-5:]) < np.mean(losses[:5])


# ---------------------------------------------------------------------------------------------------------
# the path bench.py times: DDPM, noise drawn in the kernel's epilogue, 10-step graph replays - against the oracle fed the
# identical noise, regenerated with syn_randn(seed, stream = t, first_clip) (gaussian_diffusion.py:607-739)
class _RegeneratedStepNoise:
    """step_noise[k] for the oracle's loop = what the step kernel drew at its k-th step: syn_randn(seed, stream id = timestep,
    first element = first_clip * 32 * 1536), regenerated on demand (1000 steps x 8 clips would be 1.5 GB if materialised)."""

    def __init__(self, B, ts, seed, first_clip=0):
        self.B, self.ts, self.seed, self.first = B, list(ts), seed, first_clip

    def __getitem__(self, k):
        from syntalker_amd import _lib
        buf = torch.empty(self.B, 32, 1536, device=DEV)
        _lib.check(_lib.load().syn_randn(buf.data_ptr(), buf.numel(), self.seed, int(self.ts[k]), self.first * 32 * 1536,
                                         _lib.current_stream()), "syn_randn")
        return buf.transpose(1, 2).reshape(self.B, 1536, 1, 32).cpu()


def _regenerated_step_noise(B, ts, seed, first_clip=0):
    return _RegeneratedStepNoise(B, ts, seed, first_clip)


@pytest.mark.parametrize("mode,B", [(4, 4), (3, 4), (0, 24)], ids=["whole-step-kernel", "small-batch-kernel", "library-choice-split-tiles"])
def test_seeded_20_step_ddpm_vs_oracle(beatx, mode, B):
    from oracle import denoiser_ref as dr
    from oracle.process_ref import RefProcess
    from syntalker_amd.process import create_gaussian_diffusion
    K, seed = 20, 123
    y, xT = synth.synth_clip_inputs(B, seed=51), synth.synth_latent(B, seed=51)
    beatx.layer_mode = mode
    try:
        got = create_gaussian_diffusion().p_sample_loop(beatx, (B, 1536, 1, 32), noise=xT.to(DEV), clip_denoised=False,
                                                        model_kwargs={"y": synth.to_device(y, DEV)}, skip_timesteps=1000 - K,
                                                        seed=seed).cpu()
    finally:
        beatx.layer_mode = 0
    sd = synth_state_dict("beatx")
    fw = dr.fold_weights(sd)
    with torch.no_grad():
        cond, te = dr.clip_conditioning(sd, y, fw), dr.time_table(sd, fw)
        model_fn = lambda a, b, c: dr.mdm_forward_folded(sd, fw, cond, te, a, b)      # pinned to the as-written forward on the CPU
        want = RefProcess(False).p_sample_loop(model_fn, (B, 1536, 1, 32), y, noise=xT.clone(),
                                               step_noise=_regenerated_step_noise(B, range(K - 1, -1, -1), seed), skip_timesteps=1000 - K)
    e = rel_l2(got, want)
    print(f"seeded {K}-step DDPM (mode {mode}, B={B}) rel-L2 vs oracle {e:.3e}")
    assert e < LOOP_TOL


def test_batch_between_two_pass_sizes_runs_as_two_slices_vs_oracle(beatx, monkeypatch):
    """1030 clips = one full pass of the wave-per-sequence kernel (1024) + 6: `engine.plan_slices` runs them as two slices (`k_seq`, then
    the small-batch kernel) instead of three passes of `k_stack`.  12 seeded DDPM steps; the 8 clips that straddle the cut (1022..1029)
    against the oracle with the noise regenerated for their GLOBAL clip indices, and the whole batch against the unsliced run."""
    from oracle import denoiser_ref as dr
    from oracle.process_ref import RefProcess
    from syntalker_amd import engine
    from syntalker_amd.process import create_gaussian_diffusion
    B, K, seed, lo = 1030, 12, 77, 1022
    assert engine.plan_slices(B, 1, DEV) == [(0, 1024), (1024, 1030)] and engine.plan_slices(1024, 1, DEV) == [(0, 1024)]
    assert engine.plan_slices(1536, 1, DEV) == [(0, 1024), (1024, 1536)] and engine.plan_slices(1537, 1, DEV) == [(0, 1537)]
    y8, x8 = synth.synth_clip_inputs(8, seed=61), synth.synth_latent(8, seed=61)
    rep = lambda t: t.repeat((129,) + (1,) * (t.dim() - 1))[:B] if torch.is_tensor(t) and t.dim() and t.shape[0] == 8 else t
    y = {k: rep(v) for k, v in y8.items()}                        # clip i = copy of clip i % 8 (its step noise is its own)
    xT = rep(x8)
    d = create_gaussian_diffusion()
    run = lambda: d.p_sample_loop(beatx, (B, 1536, 1, 32), noise=xT.to(DEV), clip_denoised=False, model_kwargs={"y": synth.to_device(y, DEV)},
                                  skip_timesteps=1000 - K, seed=seed).cpu()
    got = run()
    monkeypatch.setattr(engine, "plan_slices", lambda n, V, dev: [(0, n)])
    whole = run()
    monkeypatch.undo()
    assert got.shape == whole.shape == (B, 1536, 1, 32) and torch.isfinite(got).all()
    e_whole = rel_l2(got, whole)
    ys = {k: (v[lo:B] if torch.is_tensor(v) and v.dim() and v.shape[0] == B else v) for k, v in y.items()}
    sd = synth_state_dict("beatx")
    fw = dr.fold_weights(sd)
    with torch.no_grad():
        cond, te = dr.clip_conditioning(sd, ys, fw), dr.time_table(sd, fw)
        model_fn = lambda a, b, c: dr.mdm_forward_folded(sd, fw, cond, te, a, b)
        want = RefProcess(False).p_sample_loop(model_fn, (B - lo, 1536, 1, 32), ys, noise=xT[lo:B].clone(),
                                               step_noise=_regenerated_step_noise(B - lo, range(K - 1, -1, -1), seed, first_clip=lo),
                                               skip_timesteps=1000 - K)
    e = rel_l2(got[lo:B], want)
    print(f"1030 clips as 1024 + 6: rel-L2 vs oracle over the cut {e:.3e}, vs the unsliced run (k_stack) {e_whole:.3e}")
    assert e < LOOP_TOL and e_whole < LOOP_TOL


def test_full_1000_step_p_sample_loop_vs_oracle(beatx):
    """One whole p_sample_loop as the reference's sampler runs it (1000 DDPM steps, noise drawn in the step kernel, 10-step
    graph replays) against the oracle over the same 1000 regenerated noise tensors - once with the library's own kernel choice at
    B = 8 (the small-batch kernel) and once pinned to the wave-per-sequence kernel `k_seq`, the one `bench.py` times (100 persistent
    10-step launches).  The oracle's loop runs once (1000 CPU forwards on 8 threads, tests/conftest.py)."""
    from oracle import denoiser_ref as dr
    from oracle.process_ref import RefProcess
    from syntalker_amd.process import create_gaussian_diffusion
    B, seed = 8, 2024
    y, xT = synth.synth_clip_inputs(B, seed=61), synth.synth_latent(B, seed=61)
    got = {}
    for mode, name in ((0, "library's choice"), (5, "k_seq")):
        beatx.layer_mode = mode
        try:
            got[name] = create_gaussian_diffusion().p_sample_loop(beatx, (B, 1536, 1, 32), noise=xT.to(DEV), clip_denoised=False,
                                                                  model_kwargs={"y": synth.to_device(y, DEV)}, seed=seed).cpu()
        finally:
            beatx.layer_mode = 0
    assert beatx.step_buffers(B, 1).fragment is False
    sd = synth_state_dict("beatx")
    fw = dr.fold_weights(sd)
    with torch.no_grad():
        cond, te = dr.clip_conditioning(sd, y, fw), dr.time_table(sd, fw)
        model_fn = lambda a, b, c: dr.mdm_forward_folded(sd, fw, cond, te, a, b)
        want = RefProcess(False).p_sample_loop(model_fn, (B, 1536, 1, 32), y, noise=xT.clone(),
                                               step_noise=_regenerated_step_noise(B, range(999, -1, -1), seed))
    for name, g in got.items():
        e = rel_l2(g, want)
        print(f"1000-step p_sample_loop rel-L2 vs oracle {e:.3e}  ({name})")
        assert torch.isfinite(g).all() and e < LOOP_TOL, name


@pytest.mark.parametrize("case", range(5), ids=["dump_steps", "const_noise", "init_image", "clip_denoised", "inpainting"])
def test_rarely_used_loop_arguments_vs_reference(beatx, case):
    """`p_sample_loop`'s full argument list (gaussian_diffusion.py:607-739) against the reference's own outputs (loop_kwargs_outputs.npz):
    dump_steps and init_image + skip_timesteps on the fused device loop, const_noise / clip_denoised / the in-painting blend (:316-320) on
    the per-step path they fall back to.  progress=True must not change a result."""
    import os
    from syntalker_amd.process import create_gaussian_diffusion
    from tests.conftest import GOLDEN
    from tests.test_oracle_golden import _loop_kwargs_cases
    key, steps, seed, kw, extra = _loop_kwargs_cases()[case]
    fx = np.load(os.path.join(GOLDEN, "loop_kwargs_outputs.npz"))
    y = synth.to_device(dict(synth.synth_clip_inputs(2, seed=31), **extra), DEV)
    sn = synth.synth_step_noise(steps, 2, seed=seed)
    kw = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
    kw.setdefault("clip_denoised", False)
    d = create_gaussian_diffusion()
    run = lambda **more: d.p_sample_loop(beatx, (2, 1536, 1, 32), noise=synth.synth_latent(2, seed=31).to(DEV), model_kwargs={"y": y},
                                         step_noise=sn, **kw, **more)
    got = run()
    got = torch.stack(got) if isinstance(got, list) else got
    e = rel_l2(got.cpu(), fx[key])
    print(f"{key}: rel-L2 vs the reference {e:.3e}")
    assert got.shape == fx[key].shape and e < LOOP_TOL, (key, e)
    again = run(progress=True)
    again = torch.stack(again) if isinstance(again, list) else again
    assert rel_l2(again.cpu(), got.cpu()) < 5e-3                     # (another kernel grouping of the same steps: bf16 re-rounding at most)


def test_wrapper_eval_branches_and_ddim_eta_vs_reference(beatx, h3d):
    """The guidance wrappers built with `eval=True` (cfg_sampler.py:25-26, 76-80, 141-146: the unconditional evaluation only) and the DDIM loop
    with eta = 0.5 (noise drawn AND used, gaussian_diffusion.py:741-791) against the reference's outputs."""
    import os
    from syntalker_amd import guidance as G
    from syntalker_amd.process import create_gaussian_diffusion
    from tests.conftest import GOLDEN
    fx = np.load(os.path.join(GOLDEN, "loop_kwargs_outputs.npz"))
    with torch.no_grad():
        y = synth.to_device(synth.synth_clip_inputs(2, seed=7, style_dim=256, style_zero=False), DEV)
        x, t = synth.synth_latent(2, seed=7).to(DEV), torch.tensor([10, 700], device=DEV)
        e = rel_l2(G.ClassifierFreeSampleModel(h3d, eval=True)(x, t, dict(y, scale=torch.ones(1, device=DEV) * 2.5)).cpu(), fx["h3d.cfg.eval"])
        assert e < FWD_TOL, e
        yb, xb, parts = _bodypart_case()
        tb = torch.tensor([321], device=DEV)
        e = rel_l2(G.TwoClassifierFreeSampleModel_Bodypart(h3d, eval=True)(xb, tb, dict(yb, style_feature=parts)).cpu(), fx["h3d.twocfg_bodypart.eval"])
        assert e < FWD_TOL, e
        e = rel_l2(G.ClassifierFreeSampleModel_Bodypart(h3d, eval=True)(xb, tb, dict(yb, style_feature=parts, scale=torch.ones(1, device=DEV) * 2.5)).cpu(),
                   fx["h3d.cfg_bodypart.eval"])
        assert e < FWD_TOL, e
    y1 = synth.to_device(synth.synth_clip_inputs(1, seed=39), DEV)
    got = create_gaussian_diffusion(use_ddim=True).ddim_sample_loop(beatx, (1, 1536, 1, 32), noise=synth.synth_latent(1, seed=39).to(DEV), clip_denoised=False,
                                                                   model_kwargs={"y": y1}, eta=0.5, step_noise=synth.synth_step_noise(50, 1, seed=40))
    e = rel_l2(got.cpu(), fx["ddim50_eta05"])
    print(f"DDIM-50 eta 0.5 vs the reference: {e:.3e}")
    assert e < LOOP_TOL


def test_motionclip_variant_vs_reference(kernel):
    """`MDM(args)` with use_motionclip=True (models/denoiser.py:103-104, 172-174) on each step kernel against the reference's forward."""
    import os
    from tests.conftest import GOLDEN
    from tests.test_oracle_golden import _motionclip_sd
    fx = np.load(os.path.join(GOLDEN, "loop_kwargs_outputs.npz"))
    m, _ = _motionclip_sd()
    m = m.to(DEV)
    m.layer_mode = kernel
    y = synth.to_device(synth.synth_clip_inputs(2, seed=41, style_dim=512, style_zero=False), DEV)
    x, t = synth.synth_latent(2, seed=41).to(DEV), torch.tensor([5, 900], device=DEV)
    with torch.no_grad():
        assert rel_l2(m(x, t, y).cpu(), fx["motionclip.fwd.cond"]) < FWD_TOL
        assert rel_l2(m(x, t, dict(y, uncond=True)).cpu(), fx["motionclip.fwd.uncond"]) < FWD_TOL


def test_data_parallel_wrapper_and_caller_owned_kwargs(beatx):
    """The reference's DEFAULT model wrap is `nn.DataParallel(model, args.gpus).cuda()` (train.py:94), and its drivers hand that object to
    `p_sample_loop` / `training_losses`: the wrapper is looked through (same fused loop, same bits), `training_losses` works on it, and a
    forward never mutates the caller's `y` (models/denoiser.py:139 copies it)."""
    from syntalker_amd.process import create_gaussian_diffusion
    d = create_gaussian_diffusion()
    wrapped = torch.nn.DataParallel(beatx, [0])
    y = synth.to_device(synth.synth_clip_inputs(2, seed=44), DEV)
    keys, ptrs = sorted(y), {k: v.data_ptr() for k, v in y.items() if torch.is_tensor(v)}
    x = synth.synth_latent(2, seed=44).to(DEV)
    sn = synth.synth_step_noise(12, 2, seed=45)
    a = d.p_sample_loop(beatx, (2, 1536, 1, 32), noise=x.clone(), clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=988, step_noise=sn)
    b = d.p_sample_loop(wrapped, (2, 1536, 1, 32), noise=x.clone(), clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=988, step_noise=sn)
    assert torch.equal(a, b)
    with torch.no_grad():
        o1 = beatx(x, torch.tensor([3, 700], device=DEV), y)
        o2 = wrapped(x, torch.tensor([3, 700], device=DEV), y=y)
    assert torch.equal(o1, o2)
    assert sorted(y) == keys and all(y[k].data_ptr() == p for k, p in ptrs.items())          # nothing added, removed or replaced
    m = _model("beatx")
    m.differentiable_eval = True
    terms = d.training_losses(torch.nn.DataParallel(m, [0]), synth.synth_latent(2, seed=46, name="x0").to(DEV), torch.tensor([10, 500], device=DEV),
                              model_kwargs={"y": y})
    assert terms["loss"].shape == (2,) and torch.isfinite(terms["loss"]).all() and torch.equal(terms["loss"], terms["rot_mse"])
    terms["loss"].mean().backward()
    assert m.mytimmblocks[0].attn.qkv.weight.grad is not None


def test_h3d_training_loss_and_gradient_norms_vs_reference():
    """The text-prompt denoiser's training step (h3d_diffusion_new_trainer.py:446-463: style_feature through input_process3) on the differentiable
    HIP path against the reference's loss and gradient norms (eval-mode modules, fixed t and noise)."""
    import os
    from syntalker_amd.process import create_gaussian_diffusion
    from tests.conftest import GOLDEN
    fx = np.load(os.path.join(GOLDEN, "loop_kwargs_outputs.npz"))
    m = _model("h3d")
    m.differentiable_eval = True
    y = synth.to_device(synth.synth_clip_inputs(4, seed=42, style_dim=256, style_zero=False), DEV)
    x0, eps = synth.synth_latent(4, seed=42, name="x0").to(DEV), synth.synth_latent(4, seed=43, name="eps").to(DEV)
    terms = create_gaussian_diffusion().training_losses(m, x0, torch.tensor([1, 250, 640, 998], device=DEV), model_kwargs={"y": y}, noise=eps)
    assert np.allclose(terms["loss"].detach().cpu().numpy(), fx["h3d.train.loss"], rtol=2e-2)
    terms["loss"].mean().backward()
    params = dict(m.named_parameters())
    got = np.array([params[str(n)].grad.norm().item() for n in fx["h3d.train.gradnorm_names"]])
    print("h3d grad norms got / want:", got / fx["h3d.train.gradnorm"])
    assert np.allclose(got, fx["h3d.train.gradnorm"], rtol=3e-2)
    assert m.uncon_text_embeddings.grad is None and m.uncon_audio_embeddings.grad is None      # eval(): the null prompt is not reached, as in the reference


def test_training_step_gradients_are_run_to_run_identical():
    """Every reduction of the training step adds in a fixed order (partial sums per workgroup / share / wave, no atomics on a result): two backward
    passes from identical state give bit-identical gradients for every parameter - the audio encoder's convolution, first-layer and BatchNorm
    gradients, the persistent block kernels, the embedding table."""
    from syntalker_amd.process import create_gaussian_diffusion
    y = synth.to_device(synth.synth_clip_inputs(8, seed=5, mask_batch=8), DEV)
    x0, eps = synth.synth_latent(8, seed=5, name="x0").to(DEV), synth.synth_latent(8, seed=6, name="eps").to(DEV)
    t = torch.tensor([0, 17, 500, 999, 250, 3, 750, 100], device=DEV)
    d = create_gaussian_diffusion()
    m = _model("beatx").train()
    m.drop_path = 0.0
    runs = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        loss = d.training_losses(m, x0, t, model_kwargs={"y": y}, noise=eps)["loss"]
        loss.mean().backward()
        torch.cuda.synchronize()
        runs.append((loss.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
    (l0, g0), (l1, g1) = runs
    assert torch.equal(l0, l1)
    assert g0.keys() == g1.keys() and len(g0) > 100
    diff = [n for n in g0 if not torch.equal(g0[n], g1[n])]
    assert not diff, diff[:8]
