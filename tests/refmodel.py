"""Shared helpers for tests: the synthetic state_dicts the golden vectors were produced with."""
import functools

import torch

from syntalker_amd import synth

# name -> shape for every floating-point entry MDM.state_dict() holds (reference models/denoiser.py).


def _wav_spec():
    chans = [(2, 64, True), (64, 64, True), (64, 64, False), (64, 128, True), (128, 128, False), (128, 256, True)]
    spec = {}
    for i, (ci, co, down) in enumerate(chans):
        p = f"WavEncoder.feat_extractor.{i}."
        spec[p + "conv1.weight"] = (co, ci, 15); spec[p + "conv1.bias"] = (co,)
        for bn in ("bn1", "bn2"):
            if bn == "bn2":
                spec[p + "conv2.weight"] = (co, co, 15); spec[p + "conv2.bias"] = (co,)
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                spec[p + f"{bn}.{leaf}"] = (co,)
        if down:
            spec[p + "downsample.0.weight"] = (co, ci, 15); spec[p + "downsample.0.bias"] = (co,)
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                spec[p + f"downsample.1.{leaf}"] = (co,)
    return spec


def state_spec(variant="beatx"):
    s = _wav_spec()
    s["text_encoder_body.weight"] = (256, 300); s["text_encoder_body.bias"] = (256,)
    s["text_pre_encoder_body.weight"] = (synth.VOCAB, 300)
    for i in range(8):
        p = f"mytimmblocks.{i}."
        s[p + "norm1.weight"] = (512,); s[p + "norm1.bias"] = (512,)
        s[p + "attn.qkv.weight"] = (1536, 512)
        s[p + "attn.proj.weight"] = (512, 512); s[p + "attn.proj.bias"] = (512,)
        s[p + "norm2.weight"] = (512,); s[p + "norm2.bias"] = (512,)
        s[p + "mlp.fc1.weight"] = (1024, 512); s[p + "mlp.fc1.bias"] = (1024,)
        s[p + "mlp.fc2.weight"] = (512, 1024); s[p + "mlp.fc2.bias"] = (512,)
    for j in (0, 2):
        s[f"embed_timestep.time_embed.{j}.weight"] = (512, 512); s[f"embed_timestep.time_embed.{j}.bias"] = (512,)
    s["embed_style.weight"] = (64, 6); s["embed_style.bias"] = (64,)
    s["embed_text.weight"] = (512, 6144); s["embed_text.bias"] = (512,)
    s["output_process.poseFinal.weight"] = (1536, 512); s["output_process.poseFinal.bias"] = (1536,)
    s["input_process.poseEmbedding.weight"] = (512, 1536); s["input_process.poseEmbedding.bias"] = (512,)
    s["input_process2.weight"] = (512, 1280); s["input_process2.bias"] = (512,)
    s["mix_audio_text.weight"] = (256, 512); s["mix_audio_text.bias"] = (256,)
    if variant == "h3d":
        s["uncon_text_embeddings"] = (1, 256); s["uncon_audio_embeddings"] = (1, 256)
        s["input_process3.weight"] = (512, 768); s["input_process3.bias"] = (512,)
    return s


def pos_table(d=512, n=5000):
    """reference models/denoiser.py:215-220 (fp32 arithmetic, same op order)."""
    import numpy as np
    pe = torch.zeros(n, d)
    pos = torch.arange(0, n, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2).float() * (-np.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(1)


@functools.lru_cache(maxsize=4)
def synth_state_dict(variant="beatx", seed=0):
    """The same tensors make_golden.py loaded into the reference modules (name-keyed draws)."""
    sd = {k: synth.synth_tensor(k, shp, seed) for k, shp in state_spec(variant).items()}
    sd["embed_timestep.sequence_pos_encoder.pe"] = pos_table()
    sd["sequence_pos_encoder.pe"] = sd["embed_timestep.sequence_pos_encoder.pe"]
    sd["rel_pos.inv_freq"] = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))   # denoiser.py:327
    return sd


def wav_features(blocks, wav):
    """The BN-folded WavEncoder (syntalker_amd.conditioning.fold_wav_encoder's blocks) as plain PyTorch convolutions: what the HIP encoder
    (syn_wav_encode) is checked against.  wav (B, L, 2) or (B, L) -> (B, 128, 256)  (models/denoiser.py:304-322, models/utils/layer.py:144-184)."""
    import torch.nn.functional as F
    x = wav.unsqueeze(1) if wav.dim() == 2 else wav.transpose(1, 2)
    for blk in blocks:
        z = F.leaky_relu(F.conv1d(x, *blk["c1"], stride=blk["stride"], padding=blk["pad"]), 0.01)
        z = F.conv1d(z, *blk["c2"], padding=7)
        if blk["sc"] is not None:
            x = F.conv1d(x, *blk["sc"], stride=blk["stride"], padding=blk["pad"])
        x = F.leaky_relu(z + x, 0.01)
    return x.transpose(1, 2)
