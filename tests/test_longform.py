"""Chunked long-sequence driver: window arithmetic on CPU, end-to-end parity on the GPU."""
import os

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from syntalker_amd import longform, synth


def test_window_plan_matches_reference_arithmetic():
    # diffusion_rvqvae_trainer.py:414-416 with pose_length 128, pre_frames 4, vqvae_squeeze_scale 4
    assert longform.window_plan(128) == (112, 1, 0)
    assert longform.window_plan(128 + 112 * 3 + 5) == (112, 4, 5)
    assert longform.window_plan(127) == (112, 0, 111)
    round_l, rounds, _ = longform.window_plan(464)
    assert rounds * round_l + 16 == 464                        # windows tile the take with 16-pose-frame overlaps


def test_window_inputs_slices_and_seeding():
    n = 128 + 112
    audio = torch.arange(n * longform.AUDIO_PER_POSE).float().view(1, -1)
    word = torch.arange(n).view(1, -1)
    seed = torch.randn(1, n // 4, 1536)
    last = torch.randn(1, 32, 1536)
    y0 = longform.window_inputs(0, audio, word, seed, None, 112)
    y1 = longform.window_inputs(1, audio, word, seed, last, 112)
    assert y0["word"].shape == (1, 128) and y1["word"][0, 0] == 112 and y1["word"][0, -1] == n - 1
    assert y0["audio"].shape == (1, 128 * 533) and y1["audio"][0, 0] == 112 * 533
    assert torch.equal(y0["seed"], seed[:, :4]) and torch.equal(y1["seed"], last[:, -4:])
    assert y0["mask"].shape == (1, 1, 1, 128) and y0["style_feature"].shape == (1, 512)


class _Toy(torch.nn.Module):
    """x0-predictor whose output depends on the seed rows: the generic (non-HIP) path of the loops, on CPU."""
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor(0.5))

    def forward(self, x, t, y=None):
        return self.w * x + y["seed"].mean() + 1e-3 * y["word"].float().mean()


def test_sample_long_cpu_generic_path_equals_manual_loop():
    from syntalker_amd.process import create_gaussian_diffusion
    d = create_gaussian_diffusion()
    n = 128 + 2 * 112
    g = torch.Generator().manual_seed(0)
    audio, word = torch.randn(2, n * 533, generator=g), torch.randint(0, 100, (2, n), generator=g)
    seed = torch.randn(2, n // 4, 1536, generator=g)
    xs = [torch.randn(2, 1536, 1, 32, generator=g) for _ in range(3)]
    sn = [torch.randn(3, 2, 1536, 1, 32, generator=g) for _ in range(3)]
    m = _Toy()
    got = longform.sample_long(d, m, audio, word, seed, n, noise_fn=lambda i: xs[i].clone(), step_noise_fn=lambda i: sn[i],
                               skip_timesteps=997)
    assert got.shape == (2, 3 * 28 + 4, 1536)
    last, pieces = None, []
    for i in range(3):
        y = longform.window_inputs(i, audio, word, seed, last, 112)
        s = d.p_sample_loop(m, (2, 1536, 1, 32), noise=xs[i].clone(), clip_denoised=False, model_kwargs={"y": y},
                            skip_timesteps=997, step_noise=sn[i])
        last = s[:, :, 0].permute(0, 2, 1).contiguous()
        pieces.append(last if i == 0 else last[:, 4:])
    assert torch.equal(got, torch.cat(pieces, 1))


# ---------------------------------------------------------------------------------------------------------
# the reference's own `_g_test` (diffusion_rvqvae_trainer.py:359-541), executed by tests/golden/make_longform_golden.py:
# 3 windows of one take, its p_sample_loop over the 50 kept timesteps, x_T and step noise from seeded generators
def _g_test_fixture():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "longform_outputs.npz"))


def _g_test_inputs(fx):
    n, K, W = int(fx["n_pose"]), int(fx["steps"]), int(fx["windows"])
    take = synth.synth_long_take(n, seed=21)
    draws = [synth.synth_long_noise(w, K, seed=22) for w in range(W)]
    return n, K, W, take, [d[0] for d in draws], [d[1] for d in draws]


def test_longform_oracle_vs_reference_g_test():
    """oracle/longform_ref.py (window slices, seed carry, stitching, decode) against what the reference's `_g_test` produced."""
    from oracle import denoiser_ref as dr
    from oracle.longform_ref import decode_take_ref, sample_long_ref
    from tests.conftest import rel_l2
    from tests.refmodel import synth_state_dict
    fx = _g_test_fixture()
    n, K, W, take, xs, sn = _g_test_inputs(fx)
    assert (n, K, W) == (357, 50, 3)
    sd = synth_state_dict("beatx")
    with torch.no_grad():
        lat = sample_long_ref(lambda a, b, c: dr.mdm_forward(sd, a, b, c), take["audio"], take["word"], take["latent"], n, xs, sn,
                              use_ddim=True, ancestral=True)
    assert lat.shape == (1, 88, 1536)
    for k, part in enumerate(("upper", "hands", "lower")):
        assert rel_l2(lat[..., 512 * k:512 * (k + 1)] * 5.0, fx[f"{part}.latent_in"]) < 5e-6, part
    # the decode either side of it, fed the reference's own stitched latents (a code flip would be a 1e-1 event, not a 1e-6 one)
    stats = synth.synth_pose_stats()
    ref_lat = torch.cat([torch.from_numpy(fx[f"{p}.latent_in"]) for p in ("upper", "hands", "lower")], dim=-1) / 5.0
    vq_sds = {p: synth.synth_vq_state_dict(d) for p, d in (("upper", 78), ("hands", 180), ("lower", 57))}
    with torch.no_grad():
        parts = decode_take_ref(vq_sds, ref_lat, 5.0, True, *stats["trans"])
    assert rel_l2(parts["trans"], fx["rec_trans"]) < 5e-6
    for p in ("upper", "hands"):
        assert rel_l2(parts[p], fx[f"{p}.latent2origin"]) < 5e-6, p
    assert rel_l2(parts["lower"], fx["lower.latent2origin"][..., :-3]) < 5e-6


# the text-prompt trainer's `_g_test` (h3d_diffusion_new_trainer.py:465-615): DDIM-50 through TwoClassifierFreeSampleModel_Bodypart, upper + lower prompts
def _h3d_fixture():
    import numpy as np
    fx = np.load(os.path.join(REPO, "tests", "golden", "longform_h3d_outputs.npz"))
    n, K, W = int(fx["n_pose"]), int(fx["steps"]), int(fx["windows"])
    take = synth.synth_long_take(n, seed=21)
    draws = [synth.synth_long_noise(w, K, seed=24) for w in range(W)]
    parts = {"upper_mask": synth.synth_prompt_vector(synth.H3D_PROMPTS["upper"]), "hands_mask": None,
             "lower_mask": synth.synth_prompt_vector(synth.H3D_PROMPTS["lower"])}
    return fx, n, take, [d[0] for d in draws], [d[1] for d in draws], parts


def test_h3d_longform_oracle_vs_reference_g_test():
    """Guided long-form sampling of the text-prompt configuration: the oracle (guidance_ref + longform_ref over the folded h3d forward)
    against what the reference's own `_g_test` produced (9 evaluations per DDIM step, three windows, seed carried between them)."""
    from oracle import denoiser_ref as dr
    from oracle import guidance_ref as gr
    from oracle.longform_ref import sample_long_ref
    from tests.conftest import rel_l2
    from tests.refmodel import synth_state_dict
    fx, n, take, xs, sn, parts = _h3d_fixture()
    sd = synth_state_dict("h3d")
    fw = dr.fold_weights(sd, variant="h3d")
    te = dr.time_table(sd, fw)
    cache = {}

    def fn(x, t, y):                     # conditioning per (flags, style, window), hoisted like the build does: the loop's 1350 calls share 15
        st = y["style_feature"]
        key = (bool(y.get("uncond")), bool(y.get("uncond_audio")), None if y.get("uncond") else tuple(st.flatten()[:3].tolist()),
               float(y["seed"].flatten()[:16].sum()), float(y["audio"][0, :64].sum()))
        if key not in cache:
            cache[key] = dr.clip_conditioning(sd, y, fw, variant="h3d")
        return dr.mdm_forward_folded(sd, fw, cache[key], te, x, t)
    with torch.no_grad():
        lat = sample_long_ref(lambda x, t, y: gr.two_cfg_bodypart(fn, x, t, y), take["audio"], take["word"], take["latent"], n, xs, sn,
                              use_ddim=True, y_extra={"style_feature": parts, "scale": torch.ones(1)})
    assert lat.shape == (1, 88, 1536) and len(cache) == 15          # 5 distinct (flags, style) sets x 3 windows; the reference evaluates 9 per step
    for k, part in enumerate(("upper", "hands", "lower")):
        assert rel_l2(lat[..., 512 * k:512 * (k + 1)] * 10.0, fx[f"{part}.latent_in"]) < 2e-5, part


@pytest.mark.gpu
def test_h3d_guided_sample_long_vs_reference_g_test():
    """The product on the same scenario: `sample_long` with the body-part wrapper (4 unique variants as one fused batch per step),
    DDIM-50, the 156 / 360 / 107-channel RVQ-VAE decode - against the reference's `_g_test` fixture."""
    from syntalker_amd import guidance, rvqvae
    from syntalker_amd.denoiser_h3d import MDM
    from syntalker_amd.process import create_gaussian_diffusion
    from tests.conftest import rel_l2
    from tests.refmodel import synth_state_dict
    dev = "cuda"
    fx, n, take, xs, sn, parts = _h3d_fixture()
    m = MDM(synth.default_args()).eval()
    m.load_state_dict(synth_state_dict("h3d"), strict=False)
    m = m.to(dev)
    wrapped = guidance.TwoClassifierFreeSampleModel_Bodypart(m)
    d = create_gaussian_diffusion(use_ddim=True)
    dparts = {k: (None if v is None else v.to(dev)) for k, v in parts.items()}
    got = longform.sample_long(d, wrapped, take["audio"].to(dev), take["word"].to(dev), take["latent"].to(dev), n, use_ddim=True,
                               noise_fn=lambda i: xs[i].to(dev), step_noise_fn=lambda i: sn[i],
                               y_extra={"style_feature": dparts, "scale": torch.ones(1, device=dev)}).cpu()
    want = torch.cat([torch.from_numpy(fx[f"{p}.latent_in"]) for p in ("upper", "hands", "lower")], dim=-1) / 10.0
    e = rel_l2(got, want)
    print(f"guided 3-window take vs the reference's h3d _g_test: stitched latents rel-L2 {e:.3e}")
    assert got.shape == (1, 88, 1536) and e < 3e-2
    for k, (p, dim) in enumerate((("upper", 156), ("hands", 360), ("lower", 107))):
        vq = rvqvae.build(dim).eval()
        vq.load_state_dict(synth.synth_vq_state_dict(dim))
        pose = vq.to(dev).latent2origin(torch.from_numpy(fx[f"{p}.latent_in"]).to(dev))[0].cpu()
        assert rel_l2(pose, fx[f"{p}.latent2origin"]) < 3e-2, p


@pytest.mark.gpu
def test_sample_long_and_decode_vs_reference_g_test():
    """The product's `sample_long` + `decode_take` against the fixture the reference's `_g_test` wrote: same take, same x_T and
    step noise, the ancestral sampler over the 50 kept timesteps (`create_gaussian_diffusion(use_ddim=True).p_sample_loop`)."""
    from syntalker_amd import rvqvae
    from syntalker_amd.denoiser import MDM
    from syntalker_amd.process import create_gaussian_diffusion
    from tests.conftest import rel_l2
    from tests.refmodel import synth_state_dict
    dev = "cuda"
    fx = _g_test_fixture()
    n, K, W, take, xs, sn = _g_test_inputs(fx)
    m = MDM(synth.default_args()).eval()
    m.load_state_dict(synth_state_dict("beatx"), strict=False)
    m = m.to(dev)
    d = create_gaussian_diffusion(use_ddim=True)
    got = longform.sample_long(d, m, take["audio"].to(dev), take["word"].to(dev), take["latent"].to(dev), n,
                               noise_fn=lambda i: xs[i].to(dev), step_noise_fn=lambda i: sn[i]).cpu()
    assert got.shape == (1, 88, 1536)
    want = torch.cat([torch.from_numpy(fx[f"{p}.latent_in"]) for p in ("upper", "hands", "lower")], dim=-1) / 5.0
    e = rel_l2(got, want)
    print(f"3-window take vs the reference's _g_test: stitched latents rel-L2 {e:.3e}")
    assert e < 3e-2
    vqs = []
    for p, dim in (("upper", 78), ("hands", 180), ("lower", 57)):
        vq = rvqvae.build(dim).eval()
        vq.load_state_dict(synth.synth_vq_state_dict(dim))
        vqs.append(vq.to(dev))
    stats = synth.synth_pose_stats()
    parts = longform.decode_take(want.to(dev), *vqs, 5.0, use_trans=True, trans_mean=stats["trans"][0].to(dev), trans_std=stats["trans"][1].to(dev))
    for p in ("upper", "hands"):
        e = rel_l2(parts[p].cpu(), fx[f"{p}.latent2origin"])
        print(f"  decode {p}: {e:.3e}")
        assert e < 3e-2, p
    assert rel_l2(parts["lower"].cpu(), fx["lower.latent2origin"][..., :-3]) < 3e-2
    assert rel_l2(parts["trans"].cpu(), fx["rec_trans"]) < 3e-2


@pytest.mark.gpu
@pytest.mark.parametrize("use_ddim", [False, True])
def test_sample_long_vs_oracle(use_ddim):
    """3 windows, batch of 2 takes, a short schedule tail (DDPM: last 6 steps; DDIM-50: last 6 of 50), injected noise:
    HIP loop (small-batch kernel, seeded window to window) against the CPU oracle of the same loop."""
    from oracle import denoiser_ref as dr
    from oracle.longform_ref import sample_long_ref
    from syntalker_amd.denoiser import MDM
    from syntalker_amd.process import create_gaussian_diffusion
    from tests.conftest import rel_l2
    from tests.refmodel import synth_state_dict
    dev = "cuda"
    m = MDM(synth.default_args()).eval()
    m.load_state_dict(synth_state_dict("beatx"), strict=False)
    m = m.to(dev)
    sd = synth_state_dict("beatx")
    n, B, W = 128 + 2 * 112, 2, 3
    g = torch.Generator().manual_seed(5)
    audio = torch.randn(B, n * 533, 2, generator=g)
    word = torch.randint(0, synth.VOCAB, (B, n), generator=g)
    seed = torch.randn(B, n // 4, 1536, generator=g)
    steps, skip = (6, 44) if use_ddim else (6, 994)
    xs = [torch.randn(B, 1536, 1, 32, generator=g) for _ in range(W)]
    sn = [torch.randn(steps, B, 1536, 1, 32, generator=g) for _ in range(W)]
    d = create_gaussian_diffusion(use_ddim=use_ddim)
    got = longform.sample_long(d, m, audio.to(dev), word.to(dev), seed.to(dev), n, use_ddim=use_ddim,
                               noise_fn=lambda i: xs[i].to(dev), step_noise_fn=lambda i: sn[i], skip_timesteps=skip).cpu()
    want = sample_long_ref(lambda a, b, c: dr.mdm_forward(sd, a, b, c), audio, word, seed, n, xs, sn, use_ddim=use_ddim,
                           skip_timesteps=skip)
    assert got.shape == want.shape == (B, W * 28 + 4, 1536)
    e = rel_l2(got, want)
    print(f"long-form ({'ddim' if use_ddim else 'ddpm'}) rel-L2 vs oracle: {e:.3e}")
    assert e < 3e-2


@pytest.mark.gpu
def test_sample_from_config_driver(tmp_path):
    """the `test.py -c <yaml>` scenario end to end - reference-style YAML -> config.load_args -> build_sampler ->
    sample_long -> decode_take -> Fréchet statistic - through scripts/sample_from_config.py.  48 one-window takes, DDIM-50; the
    statistic against the oracle's samples (same weights and conditioning, its own noise) must sit at the oracle-vs-oracle
    noise floor."""
    import importlib.util
    import numpy as np
    from syntalker_amd import longform, synth
    spec = importlib.util.spec_from_file_location("sample_from_config", os.path.join(REPO, "scripts", "sample_from_config.py"))
    drv = importlib.util.module_from_spec(spec); spec.loader.exec_module(drv)
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text("vqvae_type: rvqvae\nvqvae_squeeze_scale: 4\nvqvae_latent_scale: 5\nuse_trans: True\naudio_f: 256\nword_f: 256\n"
                   "pose_length: 128\npre_frames: 4\npose_fps: 30\naudio_rep: onset+amplitude\n"
                   "test_ckpt: ./ckpt/beatx2_cospeech_diffusion/last_500.bin\nvqvae_upper_path: ./ckpt/beatx2_rvqvae/RVQVAE_upper/net_300000.pth\n")
    with pytest.raises(FileNotFoundError):                       # configured checkpoints must exist unless --random-init
        drv.main([str(cfg), "--takes", "1"])
    B, n, dim = 48, 128, 16
    g = torch.Generator().manual_seed(5)
    audio, word = torch.randn(B, n * longform.AUDIO_PER_POSE, 2, generator=g), torch.randint(0, synth.VOCAB, (B, n), generator=g)
    seed_lat = torch.randn(B, n // 4, 1536, generator=g)
    inp = tmp_path / "in.npz"
    np.savez(inp, audio=audio.numpy(), word=word.numpy(), seed=seed_lat.numpy())
    # reference side: the oracle's DDIM-50 samples for the same windows, two independent noise draws - deterministic CPU arithmetic, computed once by
    # scripts/frechet_oracle_stats.py driver (30 s on 8 idle cores, minutes on a busy GPU box's host) and committed as Gaussian statistics
    st = np.load(os.path.join(REPO, "tests", "golden", "sample_driver_oracle_stats.npz"))
    assert int(st["takes"]) == B and int(st["dim"]) == dim
    floor = float(st["floor"])
    np.savez(tmp_path / "ref.npz", mu=st["mu"], sigma=st["sigma"])
    rep = drv.main([str(cfg), "--random-init", "--ddim", "--inputs", str(inp), "--ref-stats", str(tmp_path / "ref.npz"),
                    "--out", str(tmp_path / "out.npz")])
    assert rep["finite"] and rep["windows"] == 1 and rep["latents"] == [B, 32, 1536]
    assert rep["poses"]["upper"] == [B, 128, 78] and rep["poses"]["hands"] == [B, 128, 180] and rep["poses"]["lower"] == [B, 128, 54]
    assert rep["poses"]["trans"] == [B, 128, 3]
    print(f"Frechet (dim {dim}, N {B}): HIP vs oracle {rep['frechet_vs_ref']:.4f}, oracle vs oracle {floor:.4f}")
    assert rep["frechet_vs_ref"] < 2.0 * floor + 1e-3
    z = np.load(tmp_path / "out.npz")
    assert z["upper"].shape == (B, 128, 78) and np.isfinite(z["trans"]).all()
