#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself on CPU.

Runs only in the build container, where /root/reference exists (it never travels to the GPU
box).  Recipe from SURVEY.md §8(c): stub the three import-time-only modules, synthesise
weights/vocab.pkl, build the reference modules, load the name-keyed deterministic weights from
syntalker_amd.synth, feed seeded synthetic inputs, save outputs as fp32 .npz.

Only inputs' seeds and the reference's OUTPUTS are stored; no reference source is copied.
    python tests/golden/make_golden.py
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from syntalker_amd import synth  # noqa: E402


def import_reference():
    for m in ("lmdb", "fasttext", "loguru"):
        sys.modules.setdefault(m, types.ModuleType(m))

    class _Quiet:
        def __getattr__(self, k):
            return lambda *a, **kw: None
    sys.modules["loguru"].logger = _Quiet()
    tmp = tempfile.mkdtemp(prefix="syn_vocab_")
    os.makedirs(os.path.join(tmp, "weights"))
    with open(os.path.join(tmp, "weights", "vocab.pkl"), "wb") as f:
        pickle.dump(types.SimpleNamespace(
            word_embedding_weights=np.zeros((synth.VOCAB, synth.WORD_DIM), np.float32)), f)
    from models.denoiser import MDM as RefMDM
    from models.denoiser_h3d import MDM as RefMDMH3D
    from diffusion.model_util import create_gaussian_diffusion
    from diffusion import cfg_sampler
    return RefMDM, RefMDMH3D, create_gaussian_diffusion, cfg_sampler, tmp + "/"


class InjectNoise:
    """Make the reference's internal th.randn_like (gaussian_diffusion.py:541,782) pop pre-drawn rows."""

    def __init__(self, rows):
        self.rows, self.k = rows, 0

    def __enter__(self):
        self._orig = torch.randn_like

        def fake(x, *a, **kw):
            r = self.rows[self.k]
            self.k += 1
            assert r.shape == x.shape
            return r.clone()
        torch.randn_like = fake
        return self

    def __exit__(self, *a):
        torch.randn_like = self._orig


def f32(t):
    return t.detach().to(torch.float32).cpu().numpy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    RefMDM, RefMDMH3D, make_diff, cfgmod, data_path = import_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self      # *_Bodypart wrappers call .cuda()
    args = synth.default_args(data_path=data_path)
    out = {}

    # ---- a1-a3: schedule tables -------------------------------------------------------
    names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next",
             "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
             "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
             "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]
    tabs = {}
    for tag, ddim in (("ddpm", False), ("ddim", True)):
        d = make_diff(use_ddim=ddim)
        for n in names:
            tabs[f"{tag}.{n}"] = np.asarray(getattr(d, n), dtype=np.float64)
        tabs[f"{tag}.timestep_map"] = np.asarray(d.timestep_map, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **tabs)

    # ---- a12-a20: denoiser forward (beatx) --------------------------------------------
    model = synth.synth_fill_(RefMDM(args).eval(), seed=0)
    y2 = synth.synth_clip_inputs(2, seed=1)
    x2 = synth.synth_latent(2, seed=1)
    taps = {}
    hooks = [model.mytimmblocks[0].register_forward_pre_hook(lambda m, a: taps.__setitem__("h0", a[0].clone()))]
    for i, b in enumerate(model.mytimmblocks):
        hooks.append(b.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"h{i + 1}", o.clone())))
    with torch.no_grad():
        out["beatx.fwd.t0_3"] = f32(model(x2, torch.tensor([0, 3]), y2))
        out["beatx.fwd.t500_999"] = f32(model(x2, torch.tensor([500, 999]), y2))
    for k, v in taps.items():                       # taps of the (500, 999) call
        out[f"beatx.tap.{k}"] = f32(v)
    for h in hooks:
        h.remove()

    # ---- a6-a9: loops with injected noise ----------------------------------------------
    y1 = synth.synth_clip_inputs(1, seed=2)
    xT = synth.synth_latent(1, seed=2)
    ddpm = make_diff(use_ddim=False)
    sn = synth.synth_step_noise(10, 1, seed=3)
    with InjectNoise(list(sn)):
        s = ddpm.p_sample_loop(model, (1, 1536, 1, 32), noise=xT.clone(), clip_denoised=False,
                               model_kwargs={"y": y1}, skip_timesteps=990)
    out["beatx.ddpm10.sample"] = f32(s)
    ddim = make_diff(use_ddim=True)
    sn = synth.synth_step_noise(50, 1, seed=4)
    with InjectNoise(list(sn)):
        s = ddim.ddim_sample_loop(model, (1, 1536, 1, 32), noise=xT.clone(), clip_denoised=False,
                                  model_kwargs={"y": y1})
    out["beatx.ddim50.sample"] = f32(s)

    # ---- a10: training_losses (eval-mode model so BN/DropPath are deterministic) -------
    y4 = synth.synth_clip_inputs(4, seed=5)
    x0 = synth.synth_latent(4, seed=5, name="x0")
    eps = synth.synth_latent(4, seed=6, name="eps")
    t4 = torch.tensor([0, 17, 500, 999])
    model.zero_grad()
    terms = ddpm.training_losses(model, x0, t4, model_kwargs={"y": y4}, noise=eps)
    out["beatx.train.loss"] = f32(terms["loss"])
    terms["loss"].mean().backward()
    gn = {}
    for n in ["mytimmblocks.0.attn.qkv.weight", "mytimmblocks.7.mlp.fc2.weight",
              "input_process.poseEmbedding.weight", "output_process.poseFinal.bias",
              "embed_timestep.time_embed.0.weight", "WavEncoder.feat_extractor.0.conv1.weight"]:
        gn[n] = dict(model.named_parameters())[n].grad.norm().item()
    out["beatx.train.gradnorm"] = np.array(list(gn.values()), np.float64)
    out["beatx.train.gradnorm_names"] = np.array(list(gn.keys()))

    # ---- a10, train mode ("a DropPath-disabled train-mode variant"): BatchNorm on BATCH statistics, the path
    # _g_training (diffusion_rvqvae_trainer.py:339-356) really runs.  DropPath is the only random element of the beatx model in
    # train() mode (its style dropout sits behind use_motionclip), so its probability is set to 0 on the reference's modules.
    mt = synth.synth_fill_(RefMDM(args).train(), seed=0)
    for mod in mt.modules():
        if type(mod).__name__ == "DropPath":
            mod.drop_prob = 0.0
    mt.zero_grad()
    terms = ddpm.training_losses(mt, x0, t4, model_kwargs={"y": y4}, noise=eps)
    out["beatx.trainmode.loss"] = f32(terms["loss"])
    terms["loss"].mean().backward()
    out["beatx.trainmode.gradnorm"] = np.array([dict(mt.named_parameters())[n].grad.norm().item() for n in gn], np.float64)
    sdt = mt.state_dict()
    for i in (0, 3, 5):                              # the BatchNorm buffers after this ONE training forward (momentum 0.1)
        for b in ("running_mean", "running_var", "num_batches_tracked"):
            out[f"beatx.trainmode.bn.{i}.bn1.{b}"] = sdt[f"WavEncoder.feat_extractor.{i}.bn1.{b}"].double().numpy()
    out["beatx.trainmode.bn.0.downsample.running_mean"] = sdt["WavEncoder.feat_extractor.0.downsample.1.running_mean"].double().numpy()
    out["beatx.trainmode.bn.5.bn2.running_var"] = sdt["WavEncoder.feat_extractor.5.bn2.running_var"].double().numpy()

    # ---- h3d variant: flags + CFG wrappers ----------------------------------------------
    mh = synth.synth_fill_(RefMDMH3D(args).eval(), seed=0)
    yh = synth.synth_clip_inputs(2, seed=7, style_dim=256, style_zero=False)
    xh = synth.synth_latent(2, seed=7)
    th = torch.tensor([10, 700])
    with torch.no_grad():
        for tag, fl in (("cond", {}), ("uncond", {"uncond": True}), ("noaudio", {"uncond_audio": True}),
                        ("both", {"uncond": True, "uncond_audio": True})):
            out[f"h3d.fwd.{tag}"] = f32(mh(xh, th, dict(yh, **fl)))
        yc = dict(yh, scale=torch.ones(1) * 2.5)
        out["h3d.cfg"] = f32(cfgmod.ClassifierFreeSampleModel(mh)(xh, th, yc))
        yc = dict(yh, scale_audio=torch.ones(1) * 1.0, scale_prompt=torch.ones(1) * 4.0)
        out["h3d.twocfg"] = f32(cfgmod.TwoClassifierFreeSampleModel(mh)(xh, th, yc))
        # body-part wrappers are batch-1 in the reference (zeros([1,256]) style, cfg_sampler.py:84)
        yb = synth.synth_clip_inputs(1, seed=8, style_dim=256, style_zero=False)
        xb = synth.synth_latent(1, seed=8)
        g = synth._gen("part_prompts", 8)
        parts = {"upper_mask": torch.randn(1, 256, generator=g), "hands_mask": None,
                 "lower_mask": torch.randn(1, 256, generator=g)}
        tb = torch.tensor([321])
        out["h3d.twocfg_bodypart"] = f32(cfgmod.TwoClassifierFreeSampleModel_Bodypart(mh)(
            xb, tb, dict(yb, style_feature=parts)))
        out["h3d.cfg_bodypart"] = f32(cfgmod.ClassifierFreeSampleModel_Bodypart(mh)(
            xb, tb, dict(yb, style_feature=parts, scale=torch.ones(1) * 2.5)))
        # one DDIM-50 guided loop, B=1 (h3d_diffusion_new_trainer.py:468-471,560-572)
        sn = synth.synth_step_noise(50, 1, seed=9)
        wrapped = cfgmod.TwoClassifierFreeSampleModel_Bodypart(mh)
        with InjectNoise(list(sn)):
            s = make_diff(use_ddim=True).ddim_sample_loop(
                wrapped, (1, 1536, 1, 32), noise=xb.clone(), clip_denoised=False,
                model_kwargs={"y": dict(yb, style_feature=parts)})
        out["h3d.ddim50_bodypart.sample"] = f32(s)

    np.savez_compressed(os.path.join(HERE, "reference_outputs.npz"), **out)
    for k, v in out.items():
        print(f"{k:34s} {tuple(v.shape)}")
    print("checksum of synth weights:", float(sum(v.double().abs().sum() for v in model.state_dict().values())))


def main_loop_kwargs():
    """The rarely-used arguments of `p_sample_loop` (gaussian_diffusion.py:607-739) run on the reference: dump_steps, const_noise,
    init_image + skip_timesteps, clip_denoised, the in-painting blend of p_mean_variance (:316-320) -> loop_kwargs_outputs.npz."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    RefMDM, _, make_diff, _, data_path = import_reference()
    model = synth.synth_fill_(RefMDM(synth.default_args(data_path=data_path)).eval(), seed=0)
    ddpm = make_diff(use_ddim=False)
    y2, x2 = synth.synth_clip_inputs(2, seed=31), synth.synth_latent(2, seed=31)
    out = {}
    with torch.no_grad():
        sn = synth.synth_step_noise(10, 2, seed=32)
        with InjectNoise(list(sn)):
            dump = ddpm.p_sample_loop(model, (2, 1536, 1, 32), noise=x2.clone(), clip_denoised=False, model_kwargs={"y": y2},
                                      skip_timesteps=990, dump_steps=[0, 3, 9])
        assert isinstance(dump, list) and len(dump) == 3
        out["dump_steps_0_3_9"] = f32(torch.stack(dump))
        sn = synth.synth_step_noise(5, 2, seed=33)
        with InjectNoise(list(sn)):                                   # const_noise: row 0 of every draw for all samples (:543-544)
            out["const_noise"] = f32(ddpm.p_sample_loop(model, (2, 1536, 1, 32), noise=x2.clone(), clip_denoised=False, model_kwargs={"y": y2},
                                                        skip_timesteps=995, const_noise=True))
        init = synth.synth_latent(2, seed=34, name="init_image")
        sn = synth.synth_step_noise(8, 2, seed=35)
        with InjectNoise(list(sn)):                                   # x_T = q_sample(init_image, t = 7, noise) (:705-712)
            out["init_image_skip992"] = f32(ddpm.p_sample_loop(model, (2, 1536, 1, 32), noise=x2.clone(), clip_denoised=False, model_kwargs={"y": y2},
                                                               skip_timesteps=992, init_image=init))
        sn = synth.synth_step_noise(5, 2, seed=36)
        with InjectNoise(list(sn)):
            out["clip_denoised"] = f32(ddpm.p_sample_loop(model, (2, 1536, 1, 32), noise=x2.clone(), clip_denoised=True, model_kwargs={"y": y2},
                                                          skip_timesteps=995))
        g = synth._gen("inpainting", 37)
        mask = torch.rand(2, 1536, 1, 32, generator=g) < 0.25
        motion = torch.randn(2, 1536, 1, 32, generator=g)
        sn = synth.synth_step_noise(5, 2, seed=38)
        with InjectNoise(list(sn)):
            out["inpainting"] = f32(ddpm.p_sample_loop(model, (2, 1536, 1, 32), noise=x2.clone(), clip_denoised=False,
                                                       model_kwargs={"y": dict(y2, inpainting_mask=mask, inpainted_motion=motion)}, skip_timesteps=995))
        # DDIM with eta != 0 (gaussian_diffusion.py:741-791: sigma = eta sqrt((1 - a_prev) / (1 - a)) sqrt(1 - a / a_prev)), 50 steps, one clip
        y1, x1 = synth.synth_clip_inputs(1, seed=39), synth.synth_latent(1, seed=39)
        sn = synth.synth_step_noise(50, 1, seed=40)
        with InjectNoise(list(sn)):
            out["ddim50_eta05"] = f32(make_diff(use_ddim=True).ddim_sample_loop(model, (1, 1536, 1, 32), noise=x1.clone(), clip_denoised=False,
                                                                              model_kwargs={"y": y1}, eta=0.5))
    # the wrappers' `eval=True` branches (cfg_sampler.py:25-26, 76-80, 141-146: "accelerate the sample process for evaluating metrics")
    _, RefMDMH3D, _, cfgmod, _ = import_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self
    mh = synth.synth_fill_(RefMDMH3D(synth.default_args(data_path=data_path)).eval(), seed=0)
    with torch.no_grad():
        yh, xh, th = synth.synth_clip_inputs(2, seed=7, style_dim=256, style_zero=False), synth.synth_latent(2, seed=7), torch.tensor([10, 700])
        out["h3d.cfg.eval"] = f32(cfgmod.ClassifierFreeSampleModel(mh, eval=True)(xh, th, dict(yh, scale=torch.ones(1) * 2.5)))
        yb, xb, tb = synth.synth_clip_inputs(1, seed=8, style_dim=256, style_zero=False), synth.synth_latent(1, seed=8), torch.tensor([321])
        g = synth._gen("part_prompts", 8)
        parts = {"upper_mask": torch.randn(1, 256, generator=g), "hands_mask": None, "lower_mask": torch.randn(1, 256, generator=g)}
        out["h3d.twocfg_bodypart.eval"] = f32(cfgmod.TwoClassifierFreeSampleModel_Bodypart(mh, eval=True)(xb, tb, dict(yb, style_feature=parts)))
        out["h3d.cfg_bodypart.eval"] = f32(cfgmod.ClassifierFreeSampleModel_Bodypart(mh, eval=True)(
            xb, tb, dict(yb, style_feature=parts, scale=torch.ones(1) * 2.5)))
    # models/denoiser.py with use_motionclip (:103-104, 172-174): a 512-d style vector enters through input_process3; `uncond` zeroes it (:110-113)
    mm = synth.synth_fill_(RefMDM(synth.default_args(data_path=data_path, use_motionclip=True)).eval(), seed=0)
    with torch.no_grad():
        ym, xm, tm = synth.synth_clip_inputs(2, seed=41, style_dim=512, style_zero=False), synth.synth_latent(2, seed=41), torch.tensor([5, 900])
        out["motionclip.fwd.cond"] = f32(mm(xm, tm, ym))
        out["motionclip.fwd.uncond"] = f32(mm(xm, tm, dict(ym, uncond=True)))
    # training_losses through the text-prompt denoiser (h3d_diffusion_new_trainer.py:446-463 `_g_training`), eval-mode modules (deterministic)
    mh.zero_grad()
    y4 = synth.synth_clip_inputs(4, seed=42, style_dim=256, style_zero=False)
    x0h, epsh, t4 = synth.synth_latent(4, seed=42, name="x0"), synth.synth_latent(4, seed=43, name="eps"), torch.tensor([1, 250, 640, 998])
    terms = ddpm.training_losses(mh, x0h, t4, model_kwargs={"y": y4}, noise=epsh)
    out["h3d.train.loss"] = f32(terms["loss"])
    terms["loss"].mean().backward()
    names = ["input_process3.weight", "input_process3.bias", "mytimmblocks.3.mlp.fc1.weight", "embed_text.weight", "WavEncoder.feat_extractor.2.conv2.weight",
             "text_pre_encoder_body.weight"]
    out["h3d.train.gradnorm"] = np.array([dict(mh.named_parameters())[n].grad.norm().item() for n in names], np.float64)
    out["h3d.train.gradnorm_names"] = np.array(names)
    assert mh.uncon_text_embeddings.grad is None                   # eval(): no dropout towards the null prompt, the parameter is not reached
    # the checkpoint surface: every state_dict entry of the three model configurations, name:shape:dtype in the reference's order
    keys = lambda m: np.array([f"{k}:{'x'.join(map(str, v.shape))}:{str(v.dtype).replace('torch.', '')}" for k, v in m.state_dict().items()])
    out["state_keys.beatx"], out["state_keys.h3d"], out["state_keys.motionclip"] = keys(model), keys(mh), keys(mm)
    # respace.py:8-61 `space_timesteps` over specifications the factory does not use but the function accepts
    from diffusion.respace import space_timesteps as ref_space
    for tag, (n, spec) in {"ddim25": (1000, "ddim25"), "100": (1000, "100"), "10_10_10": (300, "10,10,10"), "list_250": (1000, [250]),
                           "7_3": (37, "7,3"), "ddim10_of_100": (100, "ddim10")}.items():
        out["space_timesteps." + tag] = np.array(sorted(ref_space(n, spec)), np.int64)
    # the call surface the drivers use: positional-or-keyword parameters (name=default) of the reference's public entry points
    import inspect
    from diffusion import resample as ref_resample
    from diffusion import model_util as ref_model_util
    d0 = make_diff()

    def sig(fn):
        ps = [p for p in inspect.signature(fn).parameters.values() if p.kind in (p.POSITIONAL_OR_KEYWORD, p.POSITIONAL_ONLY) and p.name != "self"]
        return ",".join(p.name + ("" if p.default is p.empty else "=" + repr(p.default)) for p in ps)
    base = type(d0).__mro__[1]                      # GaussianDiffusion: respace.py forwards training_losses / p_mean_variance as (model, *args, **kwargs)
    surf = {"SpacedDiffusion." + n: getattr(base if n in ("training_losses", "p_mean_variance") else type(d0), n) for n in ("p_sample_loop", "ddim_sample_loop", "training_losses", "q_sample", "p_sample", "ddim_sample",
                                                                     "p_mean_variance", "p_sample_loop_progressive", "ddim_sample_loop_progressive")}
    surf.update({"MDM.forward": RefMDM.forward, "MDM_h3d.forward": RefMDMH3D.forward, "create_gaussian_diffusion": ref_model_util.create_gaussian_diffusion,
                 "create_named_schedule_sampler": ref_resample.create_named_schedule_sampler, "UniformSampler.sample": ref_resample.UniformSampler.sample})
    for cls in ("ClassifierFreeSampleModel", "TwoClassifierFreeSampleModel", "TwoClassifierFreeSampleModel_Bodypart", "ClassifierFreeSampleModel_Bodypart"):
        surf[cls + ".__init__"] = getattr(cfgmod, cls).__init__
        surf[cls + ".forward"] = getattr(cfgmod, cls).forward
    out["signatures"] = np.array([f"{k}({sig(f)})" for k, f in surf.items()])
    np.savez_compressed(os.path.join(HERE, "loop_kwargs_outputs.npz"), **out)
    for k, v in out.items():
        print(f"{k:24s} {tuple(v.shape)}" + (f" {float(np.abs(v).mean()):.4f}" if v.dtype.kind == "f" else ""))


class ToyDenoiser(torch.nn.Module):
    """A denoiser-shaped function for the parts of the API that only need *a* model (the diffusion classes call model(x, t, **model_kwargs))."""

    def forward(self, x, t, y=None):
        return 0.8 * torch.tanh(x) + 0.0005 * t.view(-1, 1, 1, 1).float() + y["seed"].mean(dim=(1, 2)).view(-1, 1, 1, 1)


def surface_inputs():
    x = synth.synth_latent(3, seed=71, name="surface.x")[:, :48]
    other = synth.synth_latent(3, seed=72, name="surface.other")[:, :48]
    y = {"seed": synth.synth_clip_inputs(3, seed=73)["seed"], "mask": torch.ones(3, 1, 1, 32, dtype=torch.bool)}
    return x, other, y


def main_surface():
    """The rest of GaussianDiffusion's public surface: q_mean_variance (gaussian_diffusion.py:218), _predict_xstart_from_eps /
    _from_xprev / _predict_eps_from_xstart (:399-420), ddim_reverse_sample (:850) on the full and the ddim50-spaced process -> surface_outputs.npz."""
    torch.manual_seed(0)
    _, _, make_diff, _, _ = import_reference()
    x, other, y = surface_inputs()
    out = {}
    with torch.no_grad():
        for tag, ddim, t in (("ddpm", False, torch.tensor([0, 412, 999])), ("ddim", True, torch.tensor([0, 23, 49]))):
            d = make_diff(use_ddim=ddim)
            for i, v in enumerate(d.q_mean_variance(x, t)):
                out[f"{tag}.q_mean_variance.{i}"] = f32(v)
            out[f"{tag}.xstart_from_eps"] = f32(d._predict_xstart_from_eps(x, t, other))
            out[f"{tag}.xstart_from_xprev"] = f32(d._predict_xstart_from_xprev(x, t, other))
            out[f"{tag}.eps_from_xstart"] = f32(d._predict_eps_from_xstart(x, t, other))
            r = d.ddim_reverse_sample(ToyDenoiser(), x, t, clip_denoised=False, model_kwargs={"y": y})
            out[f"{tag}.ddim_reverse.sample"], out[f"{tag}.ddim_reverse.pred_xstart"] = f32(r["sample"]), f32(r["pred_xstart"])
            r = d.ddim_reverse_sample(ToyDenoiser(), x, t, clip_denoised=True, model_kwargs={"y": y})
            out[f"{tag}.ddim_reverse_clipped.sample"] = f32(r["sample"])
    np.savez_compressed(os.path.join(HERE, "surface_outputs.npz"), **out)
    for k, v in out.items():
        print(f"{k:36s} {tuple(v.shape)} {float(np.abs(v).mean()):.5f}")


def main_per_sample_scales():
    """Per-sample guidance scales: the reference combines with y['scale'].view(-1, 1, 1, 1) - one scale per clip (diffusion/cfg_sampler.py:28,54) - on
    the text-prompt denoiser, three clips with three different scales: ClassifierFreeSampleModel, TwoClassifierFreeSampleModel (single evaluations) and a
    guided DDIM-50 loop -> per_sample_scales_outputs.npz."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _, RefMDMH3D, make_diff, cfgmod, data_path = import_reference()
    mh = synth.synth_fill_(RefMDMH3D(synth.default_args(data_path=data_path)).eval(), seed=0)
    yh, xh, th = synth.synth_clip_inputs(3, seed=51, style_dim=256, style_zero=False), synth.synth_latent(3, seed=51), torch.tensor([10, 700, 333])
    out = {}
    with torch.no_grad():
        out["cfg"] = f32(cfgmod.ClassifierFreeSampleModel(mh)(xh, th, dict(yh, scale=torch.tensor([1.5, 2.5, 0.0]))))
        out["twocfg"] = f32(cfgmod.TwoClassifierFreeSampleModel(mh)(xh, th, dict(yh, scale_audio=torch.tensor([0.5, 1.0, 1.0]),
                                                                              scale_prompt=torch.tensor([4.0, 2.0, 0.0]))))
        sn = synth.synth_step_noise(50, 3, seed=52)
        with InjectNoise(list(sn)):
            out["cfg.ddim50.sample"] = f32(make_diff(use_ddim=True).ddim_sample_loop(
                cfgmod.ClassifierFreeSampleModel(mh), (3, 1536, 1, 32), noise=xh.clone(), clip_denoised=False,
                model_kwargs={"y": dict(yh, scale=torch.tensor([1.5, 2.5, 4.0]))}))
    np.savez_compressed(os.path.join(HERE, "per_sample_scales_outputs.npz"), **out)
    for k, v in out.items():
        print(f"{k:36s} {tuple(v.shape)} {float(np.abs(v).mean()):.5f}")


if __name__ == "__main__":
    if "per_sample_scales" in sys.argv[1:]:
        main_per_sample_scales()
    elif "loop_kwargs" in sys.argv[1:]:
        main_loop_kwargs()
    elif "surface" in sys.argv[1:]:
        main_surface()
    else:
        main()
