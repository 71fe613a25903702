"""CPU tests of the host-side logic (no GPU, no HIP compute): diffusion process arithmetic, step
coefficient tables, guidance plans, weight folding + per-clip conditioning, state_dict surface,
checkpoint format, schedule sampler, and the C-ABI export list."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import denoiser_ref as dr
from oracle import guidance_ref as gr
from oracle.process_ref import RefProcess
from syntalker_amd import checkpoint, conditioning, engine, guidance, process, resample, synth
from syntalker_amd._lib import SynHipError
from tests.conftest import REPO, rel_l2
from tests import refmodel
from tests.refmodel import state_spec, synth_state_dict


class ToyModel(torch.nn.Module):
    """A stand-in denoiser with the MDM calling convention whose output depends on every flag."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor(0.7))

    def forward(self, x, t, y=None):
        s = y["style_feature"].mean() if torch.is_tensor(y.get("style_feature")) else 0.0
        out = self.w * x + 0.01 * t.view(-1, 1, 1, 1).float() + s
        if y.get("uncond", False):
            out = out - 0.3 * x
        if y.get("uncond_audio", False):
            out = out + 0.2 * x.flip(1)
        return out


def test_tables_and_factory_match_reference(golden_tables):
    for tag, ddim in (("ddpm", False), ("ddim", True)):
        d = process.create_gaussian_diffusion(use_ddim=ddim)
        assert isinstance(d, process.SpacedDiffusion) and d.num_timesteps == (50 if ddim else 1000)
        assert list(d.timestep_map) == list(golden_tables[f"{tag}.timestep_map"])
        for k, v in d.tables().items():
            assert np.array_equal(v, golden_tables[f"{tag}.{k}"]), k
    with pytest.raises(NotImplementedError):
        process.GaussianDiffusion(betas=[0.1], model_mean_type=process.ModelMeanType.EPSILON,
                                  model_var_type=process.ModelVarType.FIXED_SMALL, loss_type=process.LossType.MSE)


@pytest.mark.parametrize("ddim", [False, True])
def test_generic_loops_equal_oracle(ddim):
    """The per-step generic path (any nn.Module) reproduces the oracle's loop on the same injected noise."""
    toy = ToyModel()
    d = process.create_gaussian_diffusion(use_ddim=ddim)
    ref = RefProcess(ddim)
    y = {"style_feature": torch.randn(3, 4), "mask": torch.ones(3, 1, 1, 8, dtype=torch.bool)}
    shape = (3, 6, 1, 8)
    xT = torch.randn(*shape)
    steps = 50 if ddim else 12
    sn = torch.randn(steps, *shape)
    fn = lambda a, b, c: toy(a, b, c)
    if ddim:
        got = d.ddim_sample_loop(toy, shape, noise=xT.clone(), clip_denoised=False, model_kwargs={"y": y}, step_noise=sn)
        want = ref.ddim_sample_loop(fn, shape, y, noise=xT.clone(), step_noise=sn)
    else:
        got = d.p_sample_loop(toy, shape, noise=xT.clone(), clip_denoised=False, model_kwargs={"y": y},
                              skip_timesteps=988, step_noise=sn)
        want = ref.p_sample_loop(fn, shape, y, noise=xT.clone(), step_noise=sn, skip_timesteps=988)
    assert rel_l2(got, want) < 1e-6
    x0, t, eps = torch.randn(*shape), torch.tensor([0, 25, 49] if ddim else [0, 500, 999]), torch.randn(*shape)
    a = d.training_losses(toy, x0, t, model_kwargs={"y": y}, noise=eps)
    b = ref.training_losses(fn, x0, t, y, eps)
    assert torch.allclose(a["loss"], b["loss"], rtol=1e-6) and torch.equal(a["loss"], a["rot_mse"])


def test_step_coefficients_are_the_reference_updates():
    """engine.posterior_coefs / ddim_coefs: x_next = c0*x0 + c1*x_t + sigma*eps must equal p_sample / ddim_sample.
    Checked (a) against the fp32 oracle step (the DDIM eps-recovery (sqrt(1/ab) x - x0)/sqrt(1/ab - 1) cancels
    badly in fp32 at small t, so the reference itself carries ~1e-4 there), and (b) against the same formulas
    evaluated in fp64, where the linear form must be exact."""
    x0, xt, eps = (torch.randn(1, 5, 1, 7, dtype=torch.float64) for _ in range(3))
    for ddim in (False, True):
        ref = RefProcess(ddim)
        tab = ref.tab
        coef64 = engine.ddim_coefs(tab, 0.0, "cpu") if ddim else engine.posterior_coefs(tab, "cpu")
        for i in (0, 1, ref.num_timesteps // 2, ref.num_timesteps - 1):
            t = torch.tensor([i])
            step = ref.ddim_sample if ddim else ref.p_sample
            want, _ = step(lambda a, b, c: x0.float(), xt.float(), t, {}, eps.float())
            c = coef64[i].double()
            got = c[0] * x0 + c[1] * xt + c[2] * eps
            assert rel_l2(got, want) < (2e-4 if ddim else 2e-6), (ddim, i)
            if ddim:    # fp64 evaluation of gaussian_diffusion.py:771-791
                e_hat = (tab["sqrt_recip_alphas_cumprod"][i] * xt - x0) / tab["sqrt_recipm1_alphas_cumprod"][i]
                exact = x0 * np.sqrt(tab["alphas_cumprod_prev"][i]) + np.sqrt(1 - tab["alphas_cumprod_prev"][i]) * e_hat
            else:       # :255-277, :546-556
                nz = 0.0 if i == 0 else 1.0
                exact = tab["posterior_mean_coef1"][i] * x0 + tab["posterior_mean_coef2"][i] * xt + \
                    nz * np.exp(0.5 * tab["posterior_log_variance_clipped"][i]) * eps
            assert rel_l2(got, exact) < 1e-6, (ddim, i)           # fp32 rounding of the coefficient table only
    assert float(engine.posterior_coefs(RefProcess(False).tab, "cpu")[0, 2]) == 0.0     # no noise at t == 0
    e = engine.ddim_coefs(RefProcess(True).tab, 0.7, "cpu")                              # eta > 0 draws noise
    assert float(e[10, 2]) > 0 and float(e[0, 2]) == 0.0


def test_guidance_plans_and_generic_path_equal_oracle():
    toy = ToyModel()
    fn = lambda a, b, c: toy(a, b, c)
    x, t = torch.randn(1, 1536, 1, 4), torch.tensor([17])
    y = {"style_feature": torch.randn(1, 256), "seed": torch.zeros(1, 4, 1536)}
    got = guidance.ClassifierFreeSampleModel(toy)(x, t, dict(y, scale=torch.ones(1) * 2.5))
    assert rel_l2(got, gr.cfg(fn, x, t, dict(y, scale=torch.ones(1) * 2.5))) < 1e-6
    yy = dict(y, scale_audio=torch.ones(1) * 1.0, scale_prompt=torch.ones(1) * 4.0)
    assert rel_l2(guidance.TwoClassifierFreeSampleModel(toy)(x, t, dict(yy)), gr.two_cfg(fn, x, t, dict(yy))) < 1e-6
    parts = {"upper_mask": torch.randn(1, 256), "hands_mask": None, "lower_mask": torch.randn(1, 256)}
    w = guidance.TwoClassifierFreeSampleModel_Bodypart(toy)
    assert rel_l2(w(x, t, dict(y, style_feature=parts)), gr.two_cfg_bodypart(fn, x, t, dict(y, style_feature=parts))) < 1e-6
    plan = w.plan(dict(y, style_feature=parts))
    assert len(plan.variants) == 4                               # 9 reference evaluations, de-duplicated (MDM semantics)
    assert np.allclose(np.array(plan.weights).sum(1), 1.0)       # per channel block the weights sum to one
    w2 = guidance.ClassifierFreeSampleModel_Bodypart(toy)
    yb = dict(y, style_feature=parts, scale=torch.ones(1) * 2.5)
    assert rel_l2(w2(x, t, dict(yb)), gr.cfg_bodypart(fn, x, t, dict(yb))) < 1e-6
    assert np.allclose(np.array(w2.plan(yb).weights).sum(1), 1.0)
    assert guidance.resolve(toy) == (None, None)


def test_per_clip_guidance_scales_plan_and_generic_path_equal_oracle():
    """The reference combines with y['scale'].view(-1, 1, 1, 1): one scale per clip (cfg_sampler.py:28,54).  The planner carries such scales as
    per-clip weight tables (B, 3, V); the generic path (any wrapped module) applies them per clip; both equal the oracle's restatement."""
    toy = ToyModel()
    fn = lambda a, b, c: toy(a, b, c)
    x, t = torch.randn(3, 1536, 1, 4), torch.tensor([17, 400, 3])
    y = {"style_feature": torch.randn(3, 256), "seed": torch.zeros(3, 4, 1536)}
    sc = torch.tensor([1.5, 2.5, 0.0])
    w = guidance.ClassifierFreeSampleModel(toy)
    assert rel_l2(w(x, t, dict(y, scale=sc)), gr.cfg(fn, x, t, dict(y, scale=sc))) < 1e-6
    W = w.plan(dict(y, scale=sc)).tensor("cpu")
    assert W.shape == (3, 3, 2) and torch.allclose(W.sum(2), torch.ones(3, 3)) and torch.allclose(W[:, 0, 0], sc)
    assert w.plan(dict(y, scale=torch.ones(3) * 2.5)).tensor("cpu").shape == (3, 2)          # equal per-clip scales stay one table
    yy = dict(y, scale_audio=torch.tensor([0.5, 1.0, 1.0]), scale_prompt=torch.tensor([4.0, 2.0, 0.0]))
    w2 = guidance.TwoClassifierFreeSampleModel(toy)
    assert rel_l2(w2(x, t, dict(yy)), gr.two_cfg(fn, x, t, dict(yy))) < 1e-6
    W2 = w2.plan(dict(yy)).tensor("cpu")
    assert W2.shape == (3, 3, 3) and torch.allclose(W2.sum(2), torch.ones(3, 3))
    with pytest.raises((ValueError, RuntimeError)):                  # scales of different lengths
        w2.plan(dict(y, scale_audio=torch.tensor([0.5, 1.0, 1.0]), scale_prompt=torch.tensor([4.0, 2.0]))).tensor("cpu")


@pytest.mark.parametrize("variant", ["beatx", "h3d"])
def test_folding_and_conditioning_equal_oracle(variant):
    sd = synth_state_dict(variant)
    style = variant == "h3d"
    fw = conditioning.fold_input_stage(sd, style)
    fo = dr.fold_weights(dr.cast_sd(sd, torch.float64), variant)
    for k in ("A", "cbias", "W2a", "W2c"):
        assert rel_l2(fw[k], fo[k]) < 1e-12, k
    y = synth.synth_clip_inputs(2, seed=3, style_dim=256 if style else 512, style_zero=not style)
    cc = conditioning.ClipConditioner(sd, fw, variant, style)
    for flags in ((False, False), (True, True)):
        yy = dict(y, uncond=flags[0], uncond_audio=flags[1])
        want = dr.clip_conditioning(sd, yy, {k: (v.float() if v is not None else None) for k, v in fo.items()}, variant)
        # the fold behind the conditioning kernels (word path as a table, everything affine collapsed), evaluated on the host:
        # BN-folded convolutions -> folded tables -> cond
        audio, word = cc.audio_word_of(y, flags[1])
        got = cc.weights.host_eval(refmodel.wav_features(cc.wav_blocks, audio), word, y["seed"], cc.style_of(y, flags[0], 2))
        assert rel_l2(got, want) < 5e-6, flags
        with pytest.raises(SynHipError):                # the product path itself has no CPU fallback
            cc.cond(y, *flags)
    te = conditioning.time_table(sd, fw["W2a"], 1000)
    assert rel_l2(te, dr.time_table(sd, {k: (v.float() if v is not None else None) for k, v in fo.items()})) < 5e-6
    rc, rs = conditioning.rotary_tables(sd["rel_pos.inv_freq"])
    assert rc.shape == (32, 32) and torch.allclose(rc ** 2 + rs ** 2, torch.ones(32, 32), atol=1e-6)


@pytest.mark.parametrize("variant", ["beatx", "h3d"])
def test_state_dict_surface_and_checkpoint_format(variant, tmp_path):
    from syntalker_amd.denoiser import MDM, MDM_RVQ
    from syntalker_amd.denoiser_h3d import MDM as MDMH
    m = (MDMH if variant == "h3d" else MDM)(synth.default_args())
    assert MDM_RVQ is MDM
    sd = m.state_dict()
    for k, shp in state_spec(variant).items():
        assert k in sd and tuple(sd[k].shape) == tuple(shp), k
    assert sum(p.numel() for p in m.parameters()) == (30001252 if variant == "h3d" else 29607012)
    synth.synth_fill_(m, seed=1)
    path = tmp_path / "last_1.bin"
    torch.save({"model_state": {"module." + k: v for k, v in m.state_dict().items()}}, path)       # DataParallel keys
    m2 = (MDMH if variant == "h3d" else MDM)(synth.default_args())
    checkpoint.load_checkpoints(m2, str(path))
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    checkpoint.save_checkpoints(str(path), m2)
    assert set(torch.load(str(path))) == {"model_state"}


def test_cpu_tensors_fail_loudly():
    from syntalker_amd._lib import SynHipError
    from syntalker_amd.denoiser import MDM
    m = MDM(synth.default_args()).eval()
    with pytest.raises(SynHipError):
        m(synth.synth_latent(1), torch.tensor([3]), synth.synth_clip_inputs(1))
    with pytest.raises(SynHipError):                     # the training path has no CPU fallback either
        m.train()(synth.synth_latent(1), torch.tensor([3]), synth.synth_clip_inputs(1))


def test_uniform_sampler_follows_numpy_global_rng():
    d = process.create_gaussian_diffusion()
    s = resample.create_named_schedule_sampler("uniform", d)
    np.random.seed(2021)
    t, w = s.sample(40, "cpu")
    np.random.seed(2021)
    want = np.random.choice(1000, size=(40,), p=np.ones(1000) / 1000)
    assert np.array_equal(t.numpy(), want) and torch.all(w == 1) and t.dtype == torch.int64
    with pytest.raises(NotImplementedError):
        resample.create_named_schedule_sampler("loss-second-moment", d)


def test_c_abi_library_exports_every_declared_symbol():
    """Every function declared in include/syn_hip.h is exported by the built shared object (no compute calls)."""
    from syntalker_amd import _lib
    header = open(os.path.join(REPO, "include", "syn_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int32_t|int64_t|const char\*)\s+(syn_[a-z_0-9]+)\s*\(", header, flags=re.M))
    assert declared and declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} missing: run __graft_entry__.build()")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    # ... and nothing is exported that the header does not declare (the diagnostics switches are `void`, declared in their own section)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines() if " T syn_" in l}
    diagnostics = set(re.findall(r"^\s*void\s+(syn_debug_[a-z_0-9]+)\s*\(", header, flags=re.M))
    assert diagnostics == set(_lib.DIAGNOSTICS) and exported == declared | diagnostics, exported ^ (declared | diagnostics)
    assert lib.syn_version() == 9 == _lib.ABI_VERSION
    # every entry point that takes arguments has its ctypes signature declared (ctypes' default passes a 64-bit pointer or count as a C int)
    loaded = _lib.load()
    no_args = {"syn_version", "syn_last_error"}
    missing = [n for n in _lib.EXPORTS if n not in no_args and getattr(loaded, n).argtypes is None]
    assert not missing, missing
    # struct sizes exactly as the C compiler lays out include/syn_hip.h (gcc, same ABI as hipcc's host side)
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "sz.c")
        open(src, "w").write('#include <stdio.h>\n#include "syn_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu", sizeof(syn_step), '
                             'sizeof(syn_layer), sizeof(syn_model), sizeof(syn_wavenc), sizeof(syn_vq_model));'
                             'printf(" %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(syn_train_block_save), sizeof(syn_train_stack), sizeof(syn_train_block_grad), '
                             'sizeof(syn_train_stack_grad), sizeof(syn_opt_list), sizeof(syn_concat_src), sizeof(syn_wgrad_sum_job), sizeof(syn_bn_finalize_job), '
                             'sizeof(syn_conv_pack_req), sizeof(syn_wav_conv), sizeof(syn_cond_weights), sizeof(syn_vq_conv));return 0;}\n')
        subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), src, "-o", os.path.join(td, "sz")], check=True)
        sizes = [int(v) for v in subprocess.run([os.path.join(td, "sz")], capture_output=True, text=True, check=True).stdout.split()]
    mirrors = (_lib.SynStep, _lib.SynLayer, _lib.SynModel, _lib.SynWavEnc, _lib.SynVqModel, _lib.SynTrainBlockSave, _lib.SynTrainStack, _lib.SynTrainBlockGrad,
               _lib.SynTrainStackGrad, _lib.SynOptList, _lib.SynConcatSrc, _lib.SynWgradSumJob, _lib.SynBnFinalizeJob, _lib.SynConvPackReq, _lib.SynWavConv,
               _lib.SynCondWeights, _lib.SynVqConv)                    # every struct the ctypes bindings mirror (r6: syn_wgrad_sum_job grew a field)
    assert sizes == [ctypes.sizeof(c) for c in mirrors], (sizes, [ctypes.sizeof(c) for c in mirrors])
    assert sizes[0] == 16 + 8 * 23 + 8 and sizes[2] == 8 * 5 + 88 * 8 + 16 + 24


def test_c_abi_entry_points_reject_null_and_zero_arguments_without_crashing():
    """A C host gets no exceptions: every entry point called with NULL pointers and zero / negative sizes must come back with an error code (the size
    queries with 0), not take the process down (round 6: three size queries divided by a zero stride - SIGFPE).  Each call in a forked child; no
    call gets as far as a launch, so this runs without a GPU."""
    from syntalker_amd import _lib
    lib = _lib.load()
    queries = {"syn_wav_workspace_bytes", "syn_wav_out_frames", "syn_bn_chunks", "syn_conv1d_first_parts", "syn_vq_quantize_groups", "syn_prefers_fragment_order",
               "syn_conv1d_first_tiles", "syn_conv1d_train_fwd_tiles", "syn_conv1d_pack_bytes", "syn_conv1d_wgrad_shares", "syn_vq_workspace_bytes", "syn_opt_blocks"}

    def args_of(f, fill):
        out = []
        for a in f.argtypes:
            if fill == "zeroed structs" and hasattr(a, "_type_") and isinstance(a._type_, type) and issubclass(a._type_, ctypes.Structure):
                out.append(ctypes.pointer(a._type_()))              # a struct of NULL pointers and zero counts
            elif a in (ctypes.c_void_p, ctypes.c_char_p) or (hasattr(a, "_type_") and isinstance(a._type_, type)):
                out.append(None)
            elif a in (ctypes.c_float, ctypes.c_double):
                out.append(0.0)
            else:
                out.append(0 if fill == "zeroed structs" else fill)
        return out

    crashed, accepted = [], []
    for name in _lib.EXPORTS:
        f = getattr(lib, name)
        if f.argtypes is None:
            continue
        for fill in (0, -1, "zeroed structs"):
            pid = os.fork()
            if pid == 0:
                try:
                    os._exit(0 if f(*args_of(f, fill)) == 0 else 1)
                except BaseException:
                    os._exit(2)
            _, st = os.waitpid(pid, 0)
            if os.WIFSIGNALED(st):
                crashed.append((name, fill, os.WTERMSIG(st)))
            elif os.WEXITSTATUS(st) == 0 and name not in queries:
                accepted.append((name, fill))
    assert not crashed, crashed
    assert not accepted, accepted


def test_dropin_aliases_resolve():
    from syntalker_amd import dropin
    dropin.install()
    from diffusion.model_util import create_gaussian_diffusion
    from diffusion.cfg_sampler import TwoClassifierFreeSampleModel_Bodypart
    from diffusion.resample import create_named_schedule_sampler
    import importlib
    MDM = getattr(importlib.import_module("models.denoiser"), "MDM")
    assert create_gaussian_diffusion is process.create_gaussian_diffusion
    assert TwoClassifierFreeSampleModel_Bodypart is guidance.TwoClassifierFreeSampleModel_Bodypart
    assert create_named_schedule_sampler is resample.create_named_schedule_sampler and MDM.__name__ == "MDM"
    dropin.install(rvqvae=True)
    from syntalker_amd import rvqvae
    assert getattr(importlib.import_module("models.vq.model"), "RVQVAE") is rvqvae.RVQVAE


def test_config_loader_reads_reference_style_yaml(tmp_path):
    """YAML of the shape of configs/diffusion_rvqvae_128.yaml -> the namespace MDM / the RVQ-VAE builders read."""
    from syntalker_amd import config
    from syntalker_amd.denoiser import MDM
    y = tmp_path / "cfg.yaml"
    y.write_text("vqvae_type: rvqvae\nvqvae_squeeze_scale: 4\nvqvae_latent_scale: 5\nuse_trans: True\naudio_f: 256\n"
                 "word_f: 256\npose_length: 128\npre_frames: 4\naudio_rep: onset+amplitude\ntraining_speakers: [2]\n")
    a = config.load_args(str(y), batch_size=3)
    assert a.vqvae_latent_scale == 5 and a.use_trans is True and a.batch_size == 3 and a.training_speakers == [2]
    assert a.t_fix_pre is False                     # default filled in for a key the file omits
    m = MDM(a)                                      # constructible from it
    assert m.state_dict()["mytimmblocks.0.attn.qkv.weight"].shape == (1536, 512)
    assert config.BODY_DIMS == {"upper": 78, "hands": 180, "lower": 54}


def test_checkpoint_loads_into_data_parallel_wrapper(tmp_path):
    """The reference's drivers hand `load_checkpoints` an nn.DataParallel-wrapped model and a checkpoint whose keys carry the
    "module." prefix (test.py:87,208; utils/other_tools.py:757-790)."""
    import torch
    from syntalker_amd import checkpoint
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    src = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    for prefixed in (True, False):
        sd = {("module." + k if prefixed else k): v for k, v in src.state_dict().items()}
        path = tmp_path / f"ck{int(prefixed)}.bin"
        torch.save({"model_state": sd}, path)
        for target in (torch.nn.DataParallel(net), net):
            with torch.no_grad():
                net[0].weight.zero_()
            assert checkpoint.load_checkpoints(target, str(path)) is target
            assert torch.equal(net[0].weight, src[0].weight)
    torch.save({"model_state": {"module.0.weight": src[0].weight}}, tmp_path / "bad.bin")
    with pytest.raises(KeyError):
        checkpoint.load_checkpoints(torch.nn.DataParallel(net), str(tmp_path / "bad.bin"))


def test_configured_but_missing_checkpoint_paths_raise(tmp_path):
    from syntalker_amd import config
    a = config.load_args(None, vqvae_upper_path=str(tmp_path / "nope.bin"))
    with pytest.raises(FileNotFoundError, match="vqvae_upper_path"):
        config.build_vq_models(a, device="cpu")


def test_frechet_oracle_and_metric_match_the_reference_function():
    """oracle/frechet_ref.py and syntalker_amd/metrics.py against the reference's own `FIDCalculator.frechet_distance` (dataloaders/data_tools.py:
    1615-1685), executed by tests/golden/make_frechet_golden.py (the two static methods lifted with `ast`) on seeded sample sets - including a
    one-dimensional case, identical sets and singular (rank-deficient) covariances."""
    import importlib.util
    from oracle.frechet_ref import frechet_distance
    from syntalker_amd import metrics
    spec = importlib.util.spec_from_file_location("make_frechet_golden", os.path.join(REPO, "tests", "golden", "make_frechet_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    want = np.load(os.path.join(REPO, "tests", "golden", "frechet_reference.npz"))
    for name, (a, b) in gen.cases().items():
        for fn in (frechet_distance, metrics.frechet_distance):
            got = fn(a, b)
            assert abs(got - float(want[name])) <= 1e-9 * max(1.0, abs(float(want[name]))), (name, fn.__module__, got, float(want[name]))


def test_frechet_metric_equals_oracle_restatement():
    """syntalker_amd/metrics.py (product, host side) vs oracle/frechet_ref.py (restatement of data_tools.py:1615-1685)."""
    from oracle.frechet_ref import embed_latents, frechet_distance
    from syntalker_amd import metrics
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((300, 24)), rng.standard_normal((300, 24)) * 1.2 + 0.3
    assert abs(metrics.frechet_distance(a, b) - frechet_distance(a, b)) < 1e-9
    assert metrics.frechet_distance(a, a) < 1e-6
    lat = rng.standard_normal((5, 32, 1536))
    want = embed_latents(np.transpose(lat, (0, 2, 1))[:, :, None, :], dim=12)        # the oracle takes (N, 1536, 1, 32)
    assert np.allclose(metrics.latent_embedding(lat, dim=12), want)


def test_loop_arguments_the_reference_refuses_are_refused():
    """gaussian_diffusion.py:912-915: the DDIM loop raises NotImplementedError for dump_steps / const_noise; cond_fn (classifier guidance),
    randomize_class and cond_fn_with_grad are never passed by the reference's callers and raise here instead of being ignored."""
    d = process.create_gaussian_diffusion(use_ddim=True)
    toy = torch.nn.Linear(1, 1)
    for kw in ({"dump_steps": [0]}, {"const_noise": True}):
        with pytest.raises(NotImplementedError):
            d.ddim_sample_loop(toy, (1, 4, 1, 2), **kw)
    ddpm = process.create_gaussian_diffusion()
    for kw in ({"randomize_class": True}, {"cond_fn_with_grad": True}):
        with pytest.raises(NotImplementedError):
            ddpm.p_sample_loop(toy, (1, 4, 1, 2), skip_timesteps=999, **kw)
    with pytest.raises(NotImplementedError):
        ddpm.p_sample_loop(lambda x, t, **k: x, (1, 4, 1, 2), skip_timesteps=999, cond_fn=lambda *a, **k: None, device="cpu")


def test_word_table_comes_from_the_dataset_vocabulary(tmp_path):
    """models/denoiser.py:68-72: `MDM.__init__` reads `<data_path>weights/vocab.pkl` and initialises the (trainable unless t_fix_pre) word table
    from its `word_embedding_weights`; an unreadable pickle warns and leaves zeros for the checkpoint to fill, a missing file likewise."""
    import pickle
    import types
    from syntalker_amd.denoiser import MDM
    os.makedirs(tmp_path / "weights")
    table = np.random.RandomState(0).randn(11195, 300).astype(np.float32)
    with open(tmp_path / "weights" / "vocab.pkl", "wb") as f:
        pickle.dump(types.SimpleNamespace(word_embedding_weights=table), f)
    m = MDM(synth.default_args(data_path=str(tmp_path) + "/"))
    assert np.array_equal(m.text_pre_encoder_body.weight.detach().numpy(), table) and m.text_pre_encoder_body.weight.requires_grad
    assert not MDM(synth.default_args(data_path=str(tmp_path) + "/", t_fix_pre=True)).text_pre_encoder_body.weight.requires_grad
    (tmp_path / "weights" / "vocab.pkl").write_bytes(b"not a pickle")
    with pytest.warns(UserWarning):
        z = MDM(synth.default_args(data_path=str(tmp_path) + "/"))
    assert float(z.text_pre_encoder_body.weight.detach().abs().max()) == 0.0 and z.text_pre_encoder_body.weight.shape == (11195, 300)


def test_save_checkpoints_keeps_the_reference_file_layout(tmp_path):
    """utils/other_tools.py:757-769: {'model_state'} alone, + 'epoch' (epoch + 1) and 'opt_state' with an optimizer, + 'lrs' with a scheduler."""
    from syntalker_amd import checkpoint
    net = torch.nn.Linear(3, 2)
    opt = torch.optim.Adam(net.parameters())
    sch = torch.optim.lr_scheduler.StepLR(opt, 10)
    p = str(tmp_path / "c.bin")
    checkpoint.save_checkpoints(p, net)
    assert sorted(torch.load(p)) == ["model_state"]
    checkpoint.save_checkpoints(p, net, opt=opt, epoch=4)
    st = torch.load(p)
    assert sorted(st) == ["epoch", "model_state", "opt_state"] and st["epoch"] == 5
    checkpoint.save_checkpoints(p, net, opt=opt, epoch=4, lrs=sch)
    assert sorted(torch.load(p)) == ["epoch", "lrs", "model_state", "opt_state"]


@pytest.mark.parametrize("which", ["beatx", "h3d", "motionclip"])
def test_state_dict_is_the_reference_state_dict_entry_for_entry(which):
    """Every `state_dict()` entry - parameters AND buffers (positional table, rotary frequencies, BatchNorm counters) - with the reference
    module's name, shape, dtype and order (tests/golden/make_golden.py loop_kwargs: taken from `models.denoiser[_h3d].MDM(args).state_dict()`)."""
    from syntalker_amd.denoiser import MDM
    from syntalker_amd.denoiser_h3d import MDM as MDMH
    fx = np.load(os.path.join(REPO, "tests", "golden", "loop_kwargs_outputs.npz"))
    m = MDMH(synth.default_args()) if which == "h3d" else MDM(synth.default_args(use_motionclip=which == "motionclip"))
    got = [f"{k}:{'x'.join(map(str, v.shape))}:{str(v.dtype).replace('torch.', '')}" for k, v in m.state_dict().items()]
    want = [str(s) for s in fx[f"state_keys.{which}"]]
    assert set(got) == set(want), (sorted(set(got) - set(want))[:5], sorted(set(want) - set(got))[:5])
    assert got == want                                             # and in the same order


def test_call_signatures_are_the_reference_signatures():
    """Parameter names, order and defaults of the entry points the reference's drivers call (tests/golden/make_golden.py loop_kwargs took them
    from the reference with `inspect`): the build's versions start with exactly those parameters; what they add is keyword-only."""
    import inspect
    from syntalker_amd import guidance, resample
    from syntalker_amd.denoiser import MDM
    from syntalker_amd.denoiser_h3d import MDM as MDMH
    fx = np.load(os.path.join(REPO, "tests", "golden", "loop_kwargs_outputs.npz"))
    d = process.create_gaussian_diffusion()
    here = {"MDM.forward": MDM.forward, "MDM_h3d.forward": MDMH.forward, "create_gaussian_diffusion": process.create_gaussian_diffusion,
            "create_named_schedule_sampler": resample.create_named_schedule_sampler, "UniformSampler.sample": resample.UniformSampler.sample}
    for entry in (str(s) for s in fx["signatures"]):
        name, want = entry.split("(", 1)
        want = want[:-1]
        if name.startswith("SpacedDiffusion."):
            fn = getattr(type(d), name.split(".", 1)[1])
        elif name in here:
            fn = here[name]
        else:
            cls, meth = name.split(".")
            fn = getattr(getattr(guidance, cls), meth)
        ps = [p for p in inspect.signature(fn).parameters.values() if p.name != "self"]
        pos = [p for p in ps if p.kind in (p.POSITIONAL_OR_KEYWORD, p.POSITIONAL_ONLY)]
        got = ",".join(p.name + ("" if p.default is p.empty else "=" + repr(p.default)) for p in pos)
        if name == "create_gaussian_diffusion":                  # (the default class lives under another module path)
            got, want = ",".join(x.split("=")[0] for x in got.split(",")), ",".join(x.split("=")[0] for x in want.split(","))
        assert got == want, (name, got, want)
        assert all(p.kind is p.KEYWORD_ONLY and p.default is not p.empty for p in ps if p not in pos), name


def test_space_timesteps_over_other_specifications_matches_reference():
    """respace.py:8-61 beyond the two specifications the factory uses ("ddimN" strides, comma-separated section counts, a list, uneven sections)."""
    fx = np.load(os.path.join(REPO, "tests", "golden", "loop_kwargs_outputs.npz"))
    cases = {"ddim25": (1000, "ddim25"), "100": (1000, "100"), "10_10_10": (300, "10,10,10"), "list_250": (1000, [250]), "7_3": (37, "7,3"),
             "ddim10_of_100": (100, "ddim10")}
    for tag, (n, spec) in cases.items():
        assert sorted(process.space_timesteps(n, spec)) == list(fx["space_timesteps." + tag]), tag
    with pytest.raises(ValueError):
        process.space_timesteps(10, "7,7")
    with pytest.raises(ValueError):
        process.space_timesteps(1000, "ddim999")


def test_rest_of_the_gaussian_diffusion_surface_matches_reference():
    """q_mean_variance, _predict_xstart_from_eps / _from_xprev, _predict_eps_from_xstart, ddim_reverse_sample (gaussian_diffusion.py:218, 399-420,
    850) on the generic torch path against the reference's outputs (tests/golden/make_golden.py surface), full and ddim50-spaced process; the
    methods nobody reaches (plms_*, calc_bpd_loop, *_with_grad, condition_*) and the loss-second-moment resampler raise NotImplementedError
    naming the reference line instead of AttributeError."""
    import importlib.util
    from syntalker_amd import process, resample
    spec = importlib.util.spec_from_file_location("make_golden_surface", os.path.join(REPO, "tests", "golden", "make_golden.py"))
    src = open(spec.origin).read()
    ns = {"torch": torch, "synth": __import__("syntalker_amd.synth", fromlist=["synth"])}
    # the generator's input recipe and toy model, without importing the module (its import pulls the reference tree onto sys.path)
    import ast
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in ("ToyDenoiser", "surface_inputs")]
    exec(compile(ast.Module(body=keep, type_ignores=[]), spec.origin, "exec"), ns)
    x, other, y = ns["surface_inputs"]()
    toy = ns["ToyDenoiser"]()
    g = np.load(os.path.join(REPO, "tests", "golden", "surface_outputs.npz"))
    def close(a, name):
        return (np.allclose(a.numpy(), g[name], rtol=2e-6, atol=1e-6 * float(np.abs(g[name]).max())), name)
    with torch.no_grad():
        for tag, ddim, t in (("ddpm", False, torch.tensor([0, 412, 999])), ("ddim", True, torch.tensor([0, 23, 49]))):
            d = process.create_gaussian_diffusion(use_ddim=ddim)
            for i, v in enumerate(d.q_mean_variance(x, t)):
                assert close(v, f"{tag}.q_mean_variance.{i}")[0], (tag, i)
            assert close(d._predict_xstart_from_eps(x, t, other), f"{tag}.xstart_from_eps")[0]
            assert close(d._predict_xstart_from_xprev(x, t, other), f"{tag}.xstart_from_xprev")[0]
            assert close(d._predict_eps_from_xstart(x, t, other), f"{tag}.eps_from_xstart")[0]
            r = d.ddim_reverse_sample(toy, x, t, clip_denoised=False, model_kwargs={"y": y})
            assert close(r["sample"], f"{tag}.ddim_reverse.sample")[0] and close(r["pred_xstart"], f"{tag}.ddim_reverse.pred_xstart")[0]
            r = d.ddim_reverse_sample(toy, x, t, clip_denoised=True, model_kwargs={"y": y})
            assert close(r["sample"], f"{tag}.ddim_reverse_clipped.sample")[0]
    d = process.create_gaussian_diffusion()
    for name in ("plms_sample", "plms_sample_loop", "plms_sample_loop_progressive", "calc_bpd_loop", "_vb_terms_bpd", "_prior_bpd", "p_sample_with_grad",
                 "ddim_sample_with_grad", "condition_mean", "condition_score"):
        with pytest.raises(NotImplementedError, match="gaussian_diffusion.py"):
            getattr(d, name)(None)
    with pytest.raises(NotImplementedError, match="resample.py:124"):
        resample.create_named_schedule_sampler("loss-second-moment", d)
    with pytest.raises(AssertionError):
        d.ddim_reverse_sample(toy, x, torch.tensor([0, 1, 2]), model_kwargs={"y": y}, eta=0.5)


def test_small_gradients_move_into_bound_bucket_buffers():
    """training._into_bound_buffers (the DDP-wrapped captured step): rows of the kernels' [3][C] outputs go into the parameters' bound gradient
    buffers (DDP's bucket views) and the bound tensors are returned in their place - only for parameters that carry a bound buffer, hold no
    gradient and have not been handed their buffer in this backward; everything else is left as it is.  `direct_grad_report` counts them."""
    from syntalker_amd import training
    bn = torch.nn.BatchNorm1d(8)
    conv = torch.nn.Conv1d(4, 8, 15)
    other = torch.nn.Parameter(torch.zeros(8))
    bucket = torch.zeros(24)
    bn.weight._syn_grad_buf, bn.bias._syn_grad_buf, conv.bias._syn_grad_buf = bucket[0:8], bucket[8:16], bucket[16:24]
    try:
        dgb = torch.arange(24, dtype=torch.float32).reshape(3, 8) + 1
        gw = torch.ones(8, 4, 15)
        grads = [gw, dgb[2], dgb[0], dgb[1], torch.full((8,), 7.0)]
        owners = [None, conv.bias, bn.weight, bn.bias, other]          # (the weight gradient is written in place by its kernel: no owner here)
        with torch.no_grad():
            training._into_bound_buffers(grads, owners)
        assert grads[0] is gw and torch.equal(grads[4], torch.full((8,), 7.0))                       # untouched
        assert grads[2].data_ptr() == bucket[0:8].data_ptr() and grads[3].data_ptr() == bucket[8:16].data_ptr()
        assert grads[1].data_ptr() == bucket[16:24].data_ptr()
        assert torch.equal(bucket, torch.cat([dgb[0], dgb[1], dgb[2]]))
        m = torch.nn.ModuleList([bn, conv])
        direct, bound, rest = training.direct_grad_report(m)
        assert (direct, bound, rest) == (3, 3, [])
        # a second request in the same backward gets a fresh tensor, and the bucket is not written again
        grads2 = [dgb[0] * 2]
        with torch.no_grad():
            training._into_bound_buffers(grads2, [bn.weight])
        assert grads2[0].data_ptr() != bucket[0:8].data_ptr() and torch.equal(bucket[0:8], dgb[0])
        training._reset_handed(m)
        # a parameter that still holds a gradient keeps the autograd default (accumulation), not its bound buffer
        bn.weight.grad = torch.zeros(8)
        grads3 = [dgb[0] * 3]
        with torch.no_grad():
            training._into_bound_buffers(grads3, [bn.weight])
        assert grads3[0].data_ptr() != bucket[0:8].data_ptr() and torch.equal(bucket[0:8], dgb[0])
    finally:
        training.unbind_grad_buffers(torch.nn.ModuleList([bn, conv]))
