"""Pin the oracle (oracle/*.py) against outputs of the reference itself (tests/golden/*.npz,
produced by tests/golden/make_golden.py in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import denoiser_ref as dr
from oracle import guidance_ref as gr
from oracle import schedule_ref as sr
from oracle.process_ref import RefProcess
from syntalker_amd import synth
from tests.conftest import rel_l2
from tests.refmodel import synth_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FP32_TOL = 2e-6      # fp32 reference vs fp32 restatement: op order differs slightly, nothing else


@pytest.mark.parametrize("tag,spec", [("ddpm", None), ("ddim", "ddim50")])
def test_schedule_tables_bitexact(golden_tables, tag, spec):
    tab, tmap = sr.respaced(1000, spec)
    assert np.array_equal(np.asarray(tmap), golden_tables[f"{tag}.timestep_map"])
    for k, v in tab.items():
        assert np.array_equal(v, golden_tables[f"{tag}.{k}"]), k


def test_schedule_known_answers():
    tab, tmap = sr.respaced(1000, "ddim50")
    assert tmap[:4] == [0, 20, 40, 60] and tmap[-1] == 980 and len(tmap) == 50
    b = sr.cosine_betas(1000)
    assert b.shape == (1000,) and b.max() == 0.999 and 0 < b.min() < 1e-4


def _model_fn(sd, variant="beatx"):
    return lambda x, t, y: dr.mdm_forward(sd, x, t, y, variant=variant)


def test_forward_as_written_matches_reference(golden):
    sd = synth_state_dict("beatx")
    y, x = synth.synth_clip_inputs(2, seed=1), synth.synth_latent(2, seed=1)
    with torch.no_grad():
        o1 = dr.mdm_forward(sd, x, torch.tensor([0, 3]), y)
        taps = {}
        o2 = dr.mdm_forward(sd, x, torch.tensor([500, 999]), y, taps=taps)
    assert rel_l2(o1, golden["beatx.fwd.t0_3"]) < FP32_TOL
    assert rel_l2(o2, golden["beatx.fwd.t500_999"]) < FP32_TOL
    for k, v in taps.items():
        assert rel_l2(v, golden[f"beatx.tap.{k}"]) < FP32_TOL, k


def test_forward_folded_matches_reference(golden):
    """The hoisted + folded algebra (what the kernels compute), in fp64, vs the fp32 reference."""
    sd = dr.cast_sd(synth_state_dict("beatx"), torch.float64)
    y, x = synth.synth_clip_inputs(2, seed=1), synth.synth_latent(2, seed=1).double()
    y = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in y.items()}
    with torch.no_grad():
        fw = dr.fold_weights(sd)
        cond, te = dr.clip_conditioning(sd, y, fw), dr.time_table(sd, fw)
        o = dr.mdm_forward_folded(sd, fw, cond, te, x, torch.tensor([500, 999]))
    assert rel_l2(o, golden["beatx.fwd.t500_999"]) < 5e-6


def test_ddpm10_and_ddim50_loops(golden):
    sd = synth_state_dict("beatx")
    y, xT = synth.synth_clip_inputs(1, seed=2), synth.synth_latent(1, seed=2)
    s = RefProcess(False).p_sample_loop(_model_fn(sd), (1, 1536, 1, 32), y, noise=xT.clone(),
                                        step_noise=synth.synth_step_noise(10, 1, seed=3), skip_timesteps=990)
    assert rel_l2(s, golden["beatx.ddpm10.sample"]) < 5e-6
    s = RefProcess(True).ddim_sample_loop(_model_fn(sd), (1, 1536, 1, 32), y, noise=xT.clone(),
                                          step_noise=synth.synth_step_noise(50, 1, seed=4))
    assert rel_l2(s, golden["beatx.ddim50.sample"]) < 2e-5


def test_training_loss_value(golden):
    sd = synth_state_dict("beatx")
    y = synth.synth_clip_inputs(4, seed=5)
    x0, eps = synth.synth_latent(4, seed=5, name="x0"), synth.synth_latent(4, seed=6, name="eps")
    with torch.no_grad():
        terms = RefProcess(False).training_losses(_model_fn(sd), x0, torch.tensor([0, 17, 500, 999]), y, eps)
    assert np.allclose(terms["loss"].numpy(), golden["beatx.train.loss"], rtol=2e-6, atol=0)


def test_train_mode_loss_gradients_and_bn_buffers(golden):
    """The oracle's train() branch (BatchNorm on batch statistics) against the reference in train() mode with DropPath off:
    loss, the six gradient norms and the BatchNorm buffers after one forward (gaussian_diffusion.py:1236-1363 driving
    models/utils/layer.py:144-184; diffusion_rvqvae_trainer.py:339-356)."""
    buffers = ("running_mean", "running_var", "num_batches_tracked", ".pe", "inv_freq")
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(buffers)) for k, v in synth_state_dict("beatx").items()}
    y = synth.synth_clip_inputs(4, seed=5)
    x0, eps = synth.synth_latent(4, seed=5, name="x0"), synth.synth_latent(4, seed=6, name="eps")
    nb = {}
    fn = lambda a, b, c: dr.mdm_forward(sd, a, b, c, train_bn=True, new_buffers=nb)
    terms = RefProcess(False).training_losses(fn, x0, torch.tensor([0, 17, 500, 999]), y, eps)
    assert np.allclose(terms["loss"].detach().numpy(), golden["beatx.trainmode.loss"], rtol=5e-6, atol=0)
    assert not np.allclose(golden["beatx.trainmode.loss"], golden["beatx.train.loss"], rtol=1e-3)      # the two modes differ
    terms["loss"].mean().backward()
    names = [str(n) for n in golden["beatx.train.gradnorm_names"]]
    got = np.array([sd[n].grad.norm().item() for n in names])
    assert np.allclose(got, golden["beatx.trainmode.gradnorm"], rtol=2e-4), got / golden["beatx.trainmode.gradnorm"]
    for i in (0, 3, 5):
        for b in ("running_mean", "running_var", "num_batches_tracked"):
            want = golden[f"beatx.trainmode.bn.{i}.bn1.{b}"]
            have = nb[f"WavEncoder.feat_extractor.{i}.bn1.{b}"].double().numpy()
            assert np.allclose(have, want, rtol=1e-5, atol=1e-6), (i, b)
    assert np.allclose(nb["WavEncoder.feat_extractor.0.downsample.1.running_mean"].double().numpy(),
                       golden["beatx.trainmode.bn.0.downsample.running_mean"], rtol=1e-5, atol=1e-6)
    assert np.allclose(nb["WavEncoder.feat_extractor.5.bn2.running_var"].double().numpy(),
                       golden["beatx.trainmode.bn.5.bn2.running_var"], rtol=1e-5, atol=1e-6)


def test_h3d_flags_and_guidance(golden):
    sd = synth_state_dict("h3d")
    fn = _model_fn(sd, "h3d")
    y = synth.synth_clip_inputs(2, seed=7, style_dim=256, style_zero=False)
    x, t = synth.synth_latent(2, seed=7), torch.tensor([10, 700])
    with torch.no_grad():
        for tag, fl in (("cond", {}), ("uncond", {"uncond": True}), ("noaudio", {"uncond_audio": True}),
                        ("both", {"uncond": True, "uncond_audio": True})):
            assert rel_l2(fn(x, t, dict(y, **fl)), golden[f"h3d.fwd.{tag}"]) < FP32_TOL, tag
        o = gr.cfg(fn, x, t, dict(y, scale=torch.ones(1) * 2.5))
        assert rel_l2(o, golden["h3d.cfg"]) < 5e-6
        o = gr.two_cfg(fn, x, t, dict(y, scale_audio=torch.ones(1), scale_prompt=torch.ones(1) * 4.0))
        assert rel_l2(o, golden["h3d.twocfg"]) < 5e-6


def test_per_clip_guidance_scales_vs_reference():
    """tests/golden/per_sample_scales_outputs.npz (make_golden.py per_sample_scales: the reference's wrappers with one scale per clip): the oracle's
    restatement, and the product's wrappers on their generic path around the oracle's model function."""
    from syntalker_amd import guidance as G
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "per_sample_scales_outputs.npz"))
    sd = synth_state_dict("h3d")
    fn = _model_fn(sd, "h3d")
    y = synth.synth_clip_inputs(3, seed=51, style_dim=256, style_zero=False)
    x, t = synth.synth_latent(3, seed=51), torch.tensor([10, 700, 333])

    class Wrapped(torch.nn.Module):
        def forward(self, a, b, c):
            return fn(a, b, c)
    with torch.no_grad():
        yc = dict(y, scale=torch.tensor([1.5, 2.5, 0.0]))
        assert rel_l2(gr.cfg(fn, x, t, dict(yc)), fx["cfg"]) < 5e-6
        assert rel_l2(G.ClassifierFreeSampleModel(Wrapped())(x, t, dict(yc)), fx["cfg"]) < 5e-6
        yc = dict(y, scale_audio=torch.tensor([0.5, 1.0, 1.0]), scale_prompt=torch.tensor([4.0, 2.0, 0.0]))
        assert rel_l2(gr.two_cfg(fn, x, t, dict(yc)), fx["twocfg"]) < 5e-6
        assert rel_l2(G.TwoClassifierFreeSampleModel(Wrapped())(x, t, dict(yc)), fx["twocfg"]) < 5e-6


def _bodypart_case():
    y = synth.synth_clip_inputs(1, seed=8, style_dim=256, style_zero=False)
    g = synth._gen("part_prompts", 8)
    parts = {"upper_mask": torch.randn(1, 256, generator=g), "hands_mask": None,
             "lower_mask": torch.randn(1, 256, generator=g)}
    return y, synth.synth_latent(1, seed=8), parts


def test_h3d_bodypart_guidance(golden):
    sd = synth_state_dict("h3d")
    fn = _model_fn(sd, "h3d")
    y, x, parts = _bodypart_case()
    t = torch.tensor([321])
    with torch.no_grad():
        o = gr.two_cfg_bodypart(fn, x, t, dict(y, style_feature=parts))
        assert rel_l2(o, golden["h3d.twocfg_bodypart"]) < 5e-6
        o = gr.cfg_bodypart(fn, x, t, dict(y, style_feature=parts, scale=torch.ones(1) * 2.5))
        assert rel_l2(o, golden["h3d.cfg_bodypart"]) < 5e-6


@pytest.mark.slow
def test_h3d_ddim50_bodypart_loop(golden):
    sd = synth_state_dict("h3d")
    fn = _model_fn(sd, "h3d")
    y, x, parts = _bodypart_case()
    guided = lambda xx, tt, yy: gr.two_cfg_bodypart(fn, xx, tt, yy)
    s = RefProcess(True).ddim_sample_loop(guided, (1, 1536, 1, 32), dict(y, style_feature=parts), noise=x.clone(),
                                          step_noise=synth.synth_step_noise(50, 1, seed=9))
    assert rel_l2(s, golden["h3d.ddim50_bodypart.sample"]) < 5e-5


def _loop_kwargs_cases():
    """(key in loop_kwargs_outputs.npz, steps, noise seed, p_sample_loop keyword arguments, extra y entries)"""
    g = synth._gen("inpainting", 37)
    mask = torch.rand(2, 1536, 1, 32, generator=g) < 0.25
    motion = torch.randn(2, 1536, 1, 32, generator=g)
    return [("dump_steps_0_3_9", 10, 32, dict(skip_timesteps=990, dump_steps=[0, 3, 9]), {}),
            ("const_noise", 5, 33, dict(skip_timesteps=995, const_noise=True), {}),
            ("init_image_skip992", 8, 35, dict(skip_timesteps=992, init_image=synth.synth_latent(2, seed=34, name="init_image")), {}),
            ("clip_denoised", 5, 36, dict(skip_timesteps=995, clip_denoised=True), {}),
            ("inpainting", 5, 38, dict(skip_timesteps=995), {"inpainting_mask": mask, "inpainted_motion": motion})]


@pytest.mark.parametrize("case", range(5), ids=["dump_steps", "const_noise", "init_image", "clip_denoised", "inpainting"])
def test_rarely_used_loop_arguments_match_reference(case):
    """gaussian_diffusion.py:607-739 / :316-320 / :543-544: the oracle's loop under the arguments the trainers never pass but the API has."""
    key, steps, seed, kw, extra = _loop_kwargs_cases()[case]
    fx = np.load(__import__("os").path.join(__import__("tests.conftest", fromlist=["GOLDEN"]).GOLDEN, "loop_kwargs_outputs.npz"))
    sd = synth_state_dict("beatx")
    y = dict(synth.synth_clip_inputs(2, seed=31), **extra)
    sn = synth.synth_step_noise(steps, 2, seed=seed)
    with torch.no_grad():
        got = RefProcess(False).p_sample_loop(_model_fn(sd), (2, 1536, 1, 32), y, noise=synth.synth_latent(2, seed=31), step_noise=sn, **kw)
    got = torch.stack(got) if isinstance(got, list) else got
    assert rel_l2(got, fx[key]) < 1e-5, key


def _loop_fixture():
    import os
    from tests.conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "loop_kwargs_outputs.npz"))


def test_wrapper_eval_branches_and_ddim_eta_match_reference():
    """The `eval=True` branches of the guidance wrappers (cfg_sampler.py:25-26, 76-80, 141-146) and DDIM with eta = 0.5
    (gaussian_diffusion.py:741-791) on the oracle, against the reference's outputs."""
    fx = _loop_fixture()
    sd = synth_state_dict("h3d")
    fn = _model_fn(sd, "h3d")
    with torch.no_grad():
        y, x, t = synth.synth_clip_inputs(2, seed=7, style_dim=256, style_zero=False), synth.synth_latent(2, seed=7), torch.tensor([10, 700])
        assert rel_l2(gr.cfg(fn, x, t, dict(y, scale=torch.ones(1) * 2.5), eval_metric=True), fx["h3d.cfg.eval"]) < 5e-6
        yb, xb, parts = _bodypart_case()
        tb = torch.tensor([321])
        assert rel_l2(gr.two_cfg_bodypart(fn, xb, tb, dict(yb, style_feature=parts), eval_metric=True), fx["h3d.twocfg_bodypart.eval"]) < 5e-6
        assert rel_l2(gr.cfg_bodypart(fn, xb, tb, dict(yb, style_feature=parts, scale=torch.ones(1) * 2.5), eval_metric=True),
                      fx["h3d.cfg_bodypart.eval"]) < 5e-6
        sdb = synth_state_dict("beatx")
        y1, x1 = synth.synth_clip_inputs(1, seed=39), synth.synth_latent(1, seed=39)
        got = RefProcess(True).ddim_sample_loop(_model_fn(sdb), (1, 1536, 1, 32), y1, noise=x1, step_noise=synth.synth_step_noise(50, 1, seed=40), eta=0.5)
        assert rel_l2(got, fx["ddim50_eta05"]) < 1e-5


def _motionclip_sd():
    from syntalker_amd.denoiser import MDM
    m = MDM(synth.default_args(use_motionclip=True)).eval()
    synth.synth_fill_(m, seed=0)
    return m, {k: v.detach().clone() for k, v in m.state_dict().items()}


def test_motionclip_variant_matches_reference():
    """models/denoiser.py with use_motionclip=True (:103-104, 172-174): the 512-d style input through input_process3, zeroed by `uncond`."""
    fx = _loop_fixture()
    _, sd = _motionclip_sd()
    y, x, t = synth.synth_clip_inputs(2, seed=41, style_dim=512, style_zero=False), synth.synth_latent(2, seed=41), torch.tensor([5, 900])
    with torch.no_grad():
        assert rel_l2(dr.mdm_forward(sd, x, t, y, use_motionclip=True), fx["motionclip.fwd.cond"]) < FP32_TOL
        assert rel_l2(dr.mdm_forward(sd, x, t, dict(y, uncond=True), use_motionclip=True), fx["motionclip.fwd.uncond"]) < FP32_TOL
    assert rel_l2(fx["motionclip.fwd.cond"], fx["motionclip.fwd.uncond"]) > 1e-2          # the style input matters


def test_h3d_training_loss_and_gradient_norms_match_reference():
    """training_losses through the text-prompt denoiser (h3d_diffusion_new_trainer.py:446-463), eval-mode modules: loss per sample and the gradient
    norms of six tensors (input_process3, a block, embed_text, an encoder convolution, the word table) on the oracle vs the reference."""
    fx = _loop_fixture()
    buffers = ("running_mean", "running_var", "num_batches_tracked", ".pe", "inv_freq")
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(buffers)) for k, v in synth_state_dict("h3d").items()}
    y = synth.synth_clip_inputs(4, seed=42, style_dim=256, style_zero=False)
    x0, eps = synth.synth_latent(4, seed=42, name="x0"), synth.synth_latent(4, seed=43, name="eps")
    terms = RefProcess(False).training_losses(lambda a, b, c: dr.mdm_forward(sd, a, b, c, variant="h3d"), x0, torch.tensor([1, 250, 640, 998]), y, eps)
    assert np.allclose(terms["loss"].detach().numpy(), fx["h3d.train.loss"], rtol=5e-6, atol=0)
    terms["loss"].mean().backward()
    got = np.array([sd[str(n)].grad.norm().item() for n in fx["h3d.train.gradnorm_names"]])
    assert np.allclose(got, fx["h3d.train.gradnorm"], rtol=2e-4), got / fx["h3d.train.gradnorm"]


def trajectory_case(steps=5):
    """Inputs of tests/golden/make_train_golden.py, from seeds alone: one 4-clip batch (the seed rows written into x0 as
    `_g_training` reads them back, diffusion_rvqvae_trainer.py:346-349) and the per-step noise."""
    y = synth.synth_clip_inputs(4, seed=41)
    x0 = synth.synth_latent(4, seed=41, name="x0")
    lat = x0.squeeze(2).permute(0, 2, 1).contiguous()
    lat[:, :4] = y["seed"]
    x0 = lat.permute(0, 2, 1).unsqueeze(2).contiguous()
    y = dict(y, seed=lat[:, :4].clone())
    eps = [synth.synth_latent(4, seed=60 + k, name="eps") for k in range(steps)]
    return y, x0, eps


def test_five_step_training_trajectory_matches_the_reference_loop():
    """K = 5 steps of the reference's OWN training code - `_g_training` lifted from diffusion_rvqvae_trainer.py:339-356, its Adam from
    optimizers/optim_factory.py:122-123, clip_grad_norm_(0.99), its UniformSampler - against oracle/train_ref.py: the loss of every step
    (each depends on all the updates before it: 2.28 -> 1.49), the gradient norm clip_grad_norm_ saw, the parameter changes of ten named
    tensors after the five updates, and the BatchNorm buffers after five training forwards."""
    from oracle import train_ref
    g = np.load(os.path.join(GOLD, "train_trajectory.npz"))
    y, x0, eps = trajectory_case()
    sd = {k: v.clone() for k, v in synth_state_dict("beatx").items()}
    init = {str(n): sd[str(n)].clone() for n in g["watch"]}
    t_steps = [torch.from_numpy(row) for row in g["t"]]
    np.random.seed(100)                                    # the timesteps are what the reference's sampler drew from numpy's global RNG
    from syntalker_amd.resample import UniformSampler
    from syntalker_amd.process import create_gaussian_diffusion
    assert UniformSampler(create_gaussian_diffusion()).sample(4, "cpu")[0].tolist() == g["t"][0].tolist()
    losses, norms = train_ref.train_trajectory(sd, y, x0, t_steps, eps)
    print("loss got / want:", np.array(losses) / g["loss"], "norm got / want:", np.array(norms) / g["grad_norm"])
    assert np.allclose(losses, g["loss"], rtol=1e-5), (losses, g["loss"])
    assert np.allclose(norms, g["grad_norm"], rtol=2e-4), (norms, g["grad_norm"])
    worst = 0.0
    for n in init:
        d = (sd[n].detach() - init[n]).reshape(-1)
        assert abs(float(d.double().norm()) / float(g[f"delta_norm.{n}"]) - 1) < 1e-3, n
        e = float((d[:4096] - torch.from_numpy(g[f"delta_head.{n}"])).norm() / torch.from_numpy(g[f"delta_head.{n}"]).norm())
        worst = max(worst, e)
        assert e < 1e-5, (n, e)        # (measured 0: gradient differences of 1e-7 relative move an update of ~5e-5 by less than the parameter's fp32 spacing)
    print(f"worst parameter-delta rel-L2 vs the reference loop: {worst:.2e}")
    for n in ("WavEncoder.feat_extractor.0.bn1.running_mean", "WavEncoder.feat_extractor.5.bn2.running_var", "WavEncoder.feat_extractor.0.bn1.num_batches_tracked"):
        assert np.allclose(sd[n].double().numpy(), g[f"buffer.{n}"], rtol=1e-5, atol=1e-6), n
