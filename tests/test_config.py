"""configuration / checkpoint compatibility of the hot path.

CPU part: the reference's own YAML files (read in the build container, where the reference tree exists) select the denoiser class and
the body-part widths their trainers build; the StepLR policy; the checkpoint format.  GPU part: the `train.py -c <yaml>` loop of
scripts/train_from_config.py (epochs, per-epoch StepLR, save_checkpoints, resume from a `module.`-prefixed checkpoint) and the
sampler built from the h3d configuration."""
import glob
import importlib.util
import os

import pytest
import torch

from syntalker_amd import checkpoint, config

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CONFIGS = "/root/reference/configs"

H3D_YAML = ("model: denoiser_h3d\ng_name: MDM\ntrainer: h3d_diffusion_new\nvqvae_type: rvqvae\nvqvae_squeeze_scale: 4\nvqvae_latent_scale: 10.0\n"
            "audio_f: 256\nword_f: 256\npose_length: 128\npre_frames: 4\npose_fps: 30\naudio_rep: onset+amplitude\nbatch_size: 200\nlr_base: 5e-5\n"
            "decay_epochs: 200\ngrad_norm: 0.99\nepochs: 2000\ntest_period: 20\n")
BEATX_YAML = ("model: denoiser\ng_name: MDM\ntrainer: diffusion_rvqvae\nvqvae_type: rvqvae\nvqvae_squeeze_scale: 4\nvqvae_latent_scale: 5\nuse_trans: True\n"
              "audio_f: 256\nword_f: 256\npose_length: 128\npre_frames: 4\npose_fps: 30\naudio_rep: onset+amplitude\nbatch_size: 40\nlr_base: 5e-5\n"
              "grad_norm: 0.99\nepochs: 2000\ntest_period: 20\n")


def _driver(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "scripts", name + ".py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="the reference's configs exist only in the build container")
def test_the_references_own_yaml_files_select_class_and_widths():
    """train.py:85-94 `getattr(__import__(f"models.{args.model}"), args.g_name)`; diffusion_rvqvae_trainer.py:105-139 /
    h3d_diffusion_new_trainer.py:104-146 body-part widths."""
    seen = {}
    for path in sorted(glob.glob(os.path.join(REF_CONFIGS, "*.yaml"))):
        a = config.load_args(path)
        name = os.path.basename(path)
        if a.model in config.MODEL_MODULES:
            cls = config.model_class(a)
            seen[name] = (cls.__module__, cls.__name__, config.body_dims(a), config.is_h3d(a))
        else:                                                      # the RVQ-VAE pre-training configurations are another trainer's
            with pytest.raises(NotImplementedError):
                config.model_class(a)
    assert seen["diffusion_h3d.yaml"] == ("syntalker_amd.denoiser_h3d", "MDM", {"upper": 156, "hands": 360, "lower": 107}, True)
    for n in ("diffusion_rvqvae_128.yaml", "diffusion_rvqvae_128_hf.yaml", "diffusion_rvqvae_128_all.yaml"):
        assert seen[n] == ("syntalker_amd.denoiser", "MDM", {"upper": 78, "hands": 180, "lower": 54}, False)


def test_model_class_step_lr_and_checkpoint_format(tmp_path):
    (tmp_path / "h3d.yaml").write_text(H3D_YAML); (tmp_path / "beatx.yaml").write_text(BEATX_YAML)
    h, b = config.load_args(str(tmp_path / "h3d.yaml")), config.load_args(str(tmp_path / "beatx.yaml"))
    from syntalker_amd import denoiser, denoiser_h3d
    assert config.model_class(h) is denoiser_h3d.MDM and config.model_class(b) is denoiser.MDM
    assert config.body_dims(h)["hands"] == 360 and config.body_dims(b)["hands"] == 180
    with pytest.raises(AttributeError):
        config.model_class(config.load_args(None, model="denoiser", g_name="NoSuchClass"))
    # optimizers/timm/step_lr.py:46-51 through scheduler_factory.py:58-69: base * rate ** (epoch // decay_epochs), rate 0.1 by default
    assert [config.step_lr(h, e) for e in (0, 199, 200, 399, 400)] == pytest.approx([5e-5, 5e-5, 5e-6, 5e-6, 5e-7])
    assert config.step_lr(b, 1999) == pytest.approx(5e-5)                       # decay_epochs defaults to 9999 (utils/config.py:215)
    # utils/other_tools.py:757-790: {'model_state': state_dict}, keys possibly prefixed by nn.DataParallel
    net = torch.nn.Sequential(torch.nn.Linear(3, 2))
    path = str(tmp_path / "last_1.bin")
    checkpoint.save_checkpoints(path, torch.nn.DataParallel(net))
    sd = torch.load(path)["model_state"]
    assert sorted(sd) == ["module.0.bias", "module.0.weight"]
    other = torch.nn.Sequential(torch.nn.Linear(3, 2))
    checkpoint.load_checkpoints(other, path)
    assert torch.equal(other[0].weight.cpu(), net[0].weight.cpu())     # (DataParallel moves the module to cuda:0 when there is one)
    with pytest.raises(KeyError):
        checkpoint.load_checkpoints(torch.nn.Sequential(torch.nn.Linear(3, 2), torch.nn.Linear(2, 2)), path)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["beatx", "h3d"])
def test_train_from_config_driver(tmp_path, which):
    """`train.py -c <yaml>` on this build: epochs, the training step (BatchNorm batch statistics, DropPath, Adam, clip 0.99), StepLR
    after every epoch, last_<epoch>.bin every test_period epochs, and a second run resumed from a `module.`-prefixed checkpoint."""
    drv = _driver("train_from_config")
    text = (H3D_YAML if which == "h3d" else BEATX_YAML).replace("test_period: 20", "test_period: 1").replace("decay_epochs: 200", "decay_epochs: 1")
    if which == "beatx":
        text += "decay_epochs: 1\n"
    (tmp_path / "cfg.yaml").write_text(text)
    out = str(tmp_path / "run")
    rep = drv.main([str(tmp_path / "cfg.yaml"), "--epochs", "2", "--steps-per-epoch", "2", "--batch-size", "4", "--random-init", "--out", out])
    assert rep["model"] == ("syntalker_amd.denoiser_h3d.MDM" if which == "h3d" else "syntalker_amd.denoiser.MDM")
    assert [e["epoch"] for e in rep["log"]] == [0, 1] and all(e["steps"] == 2 and e["loss"] == e["loss"] for e in rep["log"])
    assert [e["lr"] for e in rep["log"]] == pytest.approx([5e-5, 5e-6])          # rate 0.1 per decay_epochs = 1 epoch
    assert [os.path.basename(p) for p in rep["saved"]] == ["last_1.bin", "last_2.bin"]
    sd = torch.load(rep["saved"][1], map_location="cpu")["model_state"]
    assert "mytimmblocks.0.attn.qkv.weight" in sd and ("input_process3.weight" in sd) == (which == "h3d")
    assert int(sd["WavEncoder.feat_extractor.0.bn1.num_batches_tracked"]) == 4    # 2 epochs x 2 training forwards
    # resume as the reference's drivers do, from a checkpoint written under nn.DataParallel (prefixed keys)
    pref = str(tmp_path / "prefixed.bin")
    torch.save({"model_state": {"module." + k: v for k, v in sd.items()}}, pref)
    rep2 = drv.main([str(tmp_path / "cfg.yaml"), "--epochs", "1", "--steps-per-epoch", "1", "--batch-size", "4", "--resume", pref, "--out", out + "2"])
    sd2 = torch.load(rep2["saved"][0], map_location="cpu")["model_state"]
    assert int(sd2["WavEncoder.feat_extractor.0.bn1.num_batches_tracked"]) == 5   # continued from the loaded state
    assert not torch.equal(sd2["mytimmblocks.0.attn.qkv.weight"], sd["mytimmblocks.0.attn.qkv.weight"])
    if which == "beatx":
        # the same loop with the whole step replayed from one hipGraph: config.build_trainer's ClipAdam is captured as it is (rate and step
        # count on the device), the per-epoch StepLR writes the rate the replays read
        rep3 = drv.main([str(tmp_path / "cfg.yaml"), "--epochs", "2", "--steps-per-epoch", "3", "--batch-size", "4", "--random-init", "--graph", "--out", out + "3"])
        assert [e["lr"] for e in rep3["log"]] == pytest.approx([5e-5, 5e-6]) and all(e["loss"] == e["loss"] and e["steps"] == 3 for e in rep3["log"])
        sd3 = torch.load(rep3["saved"][1], map_location="cpu")["model_state"]
        assert torch.isfinite(sd3["mytimmblocks.0.attn.qkv.weight"]).all()
        # (the captured step's eager warm-up iterations are undone - parameters, BatchNorm buffers, optimizer state - and the capture itself
        # executes nothing: exactly the 6 replays count, as in the eager loop and the reference)
        assert int(sd3["WavEncoder.feat_extractor.0.bn1.num_batches_tracked"]) == 6


@pytest.mark.gpu
def test_sampler_built_from_the_h3d_configuration(tmp_path):
    """config.build_sampler honours `model: denoiser_h3d`: the h3d denoiser (style input, learned null embedding), the
    156 / 360 / 107-channel RVQ-VAEs, no separate translation model; one guided DDIM window decodes to those widths."""
    from syntalker_amd import denoiser_h3d, guidance, synth
    (tmp_path / "h3d.yaml").write_text(H3D_YAML)
    args = config.load_args(str(tmp_path / "h3d.yaml"))
    s = config.build_sampler(args)
    assert isinstance(s.model, denoiser_h3d.MDM) and s.use_trans is False and s.latent_scale == 10.0
    assert {k: m.input_width for k, m in s.vq.items()} == {"upper": 156, "hands": 360, "lower": 107}
    synth.synth_fill_(s.model, 0)
    for m in s.vq.values():
        m.load_state_dict(synth.synth_vq_state_dict(m.input_width, seed=11))
    y = synth.to_device(synth.synth_clip_inputs(2, seed=3, style_dim=256, style_zero=False), "cuda")
    y["scale"] = torch.ones(1, device="cuda") * 2.5
    from syntalker_amd.process import create_gaussian_diffusion
    x = create_gaussian_diffusion(use_ddim=True).ddim_sample_loop(guidance.ClassifierFreeSampleModel(s.model), (2, 1536, 1, 32),
                                                                   clip_denoised=False, model_kwargs={"y": y})
    lat = x[:, :, 0, :].permute(0, 2, 1) * s.latent_scale                        # (B, 32, 1536): three 512-wide body-part latents
    for i, part in enumerate(("upper", "hands", "lower")):
        pose = s.vq[part].latent2origin(lat[..., 512 * i:512 * (i + 1)].contiguous())[0]
        assert pose.shape == (2, 128, config.BODY_DIMS_H3D[part]) and torch.isfinite(pose).all()
