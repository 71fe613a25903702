"""``MDM`` — drop-in for the reference's denoiser module (models/denoiser.py:12, models/denoiser_h3d.py:12).

Same constructor (`MDM(args)`, reading the same argparse keys), same ``forward(x, timesteps, y=None,
uncond_info=False)`` contract and the same ``state_dict()`` keys, so reference checkpoints
(``{'model_state': sd}``, optionally ``module.``-prefixed; utils/other_tools.py:757-790) load unchanged.
The modules below only HOLD parameters; evaluation goes through the HIP step kernels
(engine.py -> libsyn_hip.so).  There is no CPU / eager fallback: CPU tensors raise.
"""
from __future__ import annotations

import os
import pickle
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import engine
from ._lib import SynHipError

VOCAB_ROWS, WORD_DIM = 11195, 300


# ---- parameter holders (names fixed by the reference's state_dict) -----------------------------
class _WavBlock(nn.Module):          # models/utils/layer.py:144-184
    def __init__(self, cin, cout, stride, pad, down):
        super().__init__()
        self.conv1 = nn.Conv1d(cin, cout, 15, stride=stride, padding=pad)
        self.bn1 = nn.BatchNorm1d(cout)
        self.conv2 = nn.Conv1d(cout, cout, 15, padding=7)
        self.bn2 = nn.BatchNorm1d(cout)
        self.downsample = nn.Sequential(nn.Conv1d(cin, cout, 15, stride=stride, padding=pad),
                                        nn.BatchNorm1d(cout)) if down else None


class _WavEncoder(nn.Module):        # models/denoiser.py:304-315
    def __init__(self, out_dim, audio_in):
        super().__init__()
        q, h = out_dim // 4, out_dim // 2
        spec = [(audio_in, q, 5, 1700, True), (q, q, 6, 0, True), (q, q, 1, 7, False),
                (q, h, 6, 0, True), (h, h, 1, 7, False), (h, out_dim, 3, 0, True)]
        self.feat_extractor = nn.Sequential(*[_WavBlock(*s) for s in spec])


class _Attn(nn.Module):              # timm_transformer/transformer.py:56-81
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim, bias=False)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):               # timm_transformer/transformer.py:117-143
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):             # timm_transformer/transformer.py:154-198
    def __init__(self, dim, hidden):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _Attn(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, hidden)


class _PosTable(nn.Module):          # models/denoiser.py:210-222
    def __init__(self, d_model, max_len=5000):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(0, 1))


class _TimeEmbed(nn.Module):         # models/denoiser.py:231-242
    def __init__(self, dim, pos):
        super().__init__()
        self.sequence_pos_encoder = pos
        self.time_embed = nn.Sequential(nn.Linear(dim, dim), nn.SiLU(), nn.Linear(dim, dim))


class _PoseIn(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.poseEmbedding = nn.Linear(cin, dim)


class _PoseOut(nn.Module):
    def __init__(self, dim, cout):
        super().__init__()
        self.poseFinal = nn.Linear(dim, cout)


class _Rotary(nn.Module):            # models/denoiser.py:324-328
    def __init__(self, dim):
        super().__init__()
        self.register_buffer("inv_freq", 1. / (10000 ** (torch.arange(0, dim, 2).float() / dim)))


def _pretrained_words(args):
    path = f"{getattr(args, 'data_path', '')}weights/vocab.pkl"      # models/denoiser.py:68-71
    if os.path.exists(path):
        try:
            with open(path, "rb") as f:
                return torch.FloatTensor(pickle.load(f).word_embedding_weights)
        except Exception as e:                                         # the pickle needs the dataset's vocab class
            warnings.warn(f"could not unpickle {path} ({e}); word table starts at zero until a checkpoint is loaded")
    return torch.zeros(int(getattr(args, "word_index_num", VOCAB_ROWS)), WORD_DIM)


class MDM(nn.Module):
    variant = "beatx"

    def __init__(self, args):
        super().__init__()
        vq = getattr(args, "vqvae_type", "rvqvae")
        self.njoints = {"rvqvae": 1536, "novqvae": 312}.get(vq, 768) if self.variant == "beatx" else 1536
        if self.njoints != engine.CH:
            raise NotImplementedError(f"HIP step kernels are built for 1536 latent channels (vqvae_type='rvqvae'), got {self.njoints}")
        self.args = args
        self.latent_dim, self.ff_size, self.num_layers, self.num_heads = 512, 1024, 8, 4
        self.cond_mask_prob = 0.3
        self.drop_path = 0.1               # models/denoiser.py:83: every block's DropPath probability (self.dropout); train() mode only
        self.use_motionclip = bool(getattr(args, "use_motionclip", False)) if self.variant == "beatx" else False
        audio_f, word_f = args.audio_f, args.word_f
        if getattr(args, "audio_rep", "onset+amplitude") != "onset+amplitude":
            raise NotImplementedError("only audio_rep='onset+amplitude' has a WavEncoder in the reference")
        self.WavEncoder = _WavEncoder(audio_f, audio_in=2)
        self.text_encoder_body = nn.Linear(WORD_DIM, audio_f)
        self.text_pre_encoder_body = nn.Embedding.from_pretrained(_pretrained_words(args),
                                                                  freeze=bool(getattr(args, "t_fix_pre", False)))
        self.sequence_pos_encoder = _PosTable(self.latent_dim)
        self.mytimmblocks = nn.ModuleList([_Block(self.latent_dim, self.ff_size) for _ in range(self.num_layers)])
        self.embed_timestep = _TimeEmbed(self.latent_dim, self.sequence_pos_encoder)
        self.embed_style = nn.Linear(6, 64)                              # defined, never used
        self.embed_text = nn.Linear(self.njoints * 4, self.latent_dim)
        self.output_process = _PoseOut(self.latent_dim, self.njoints)
        self.rel_pos = _Rotary(self.latent_dim // 8)
        self.input_process = _PoseIn(self.njoints, self.latent_dim)
        self.input_process2 = nn.Linear(self.latent_dim * 2 + audio_f, self.latent_dim)
        if self.variant == "h3d":
            self.uncon_text_embeddings = nn.Parameter(torch.zeros(1, 256))
            self.uncon_audio_embeddings = nn.Parameter(torch.zeros(1, audio_f))
            self.input_process3 = nn.Linear(self.latent_dim + 256, self.latent_dim)
        elif self.use_motionclip:
            self.input_process3 = nn.Linear(self.latent_dim + 512, self.latent_dim)
        self.mix_audio_text = nn.Linear(audio_f + word_f, 256)
        self._packed, self._packed_key = None, None
        self._bufs, self._cond_entry = {}, None
        self.m_tile = 0
        self.layer_mode = 0            # syn_step.reserved: 0 library's choice (small-batch kernel for few sequences, else the
                                       # whole-step kernel); 4 / 3 / 5 pin one of them; 1: five kernels per block (bitwise cross-check, h8 tap)
        self.differentiable_eval = False   # eval() + autograd on: take the differentiable path (gradient tests)

    def __getstate__(self):
        """copy.deepcopy(model) (an EMA copy) and torch.save(model) take the module as nn.Module defines it - parameters, buffers, attributes - and none
        of the device-side caches derived from them (ctypes structs of raw pointers into THIS module's packed tensors): the copy builds its own."""
        st = self.__dict__.copy()
        st["_packed"], st["_packed_key"], st["_bufs"], st["_cond_entry"] = None, None, {}, None
        for k in [k for k in st if k.startswith("_syn_") or k in ("_ident", "_graphs")]:          # (`_graphs`: the captured loops of process._fused)
            del st[k]
        return st

    # ---- engine plumbing ----------------------------------------------------------------------
    @property
    def uses_style(self):
        return self.variant == "h3d" or self.use_motionclip

    def _weights_key(self):
        return (engine.raw_write_epoch(),) + tuple((t.data_ptr(), t._version) for t in self.state_dict(keep_vars=True).values())

    def packed(self) -> engine.PackedModel:
        """Folded/packed weights, rebuilt when any parameter was modified in place or moved."""
        key = self._weights_key()
        if self._packed is None or key != self._packed_key:
            sd = {k: v.detach() for k, v in self.state_dict(keep_vars=True).items()}
            self._packed, self._packed_key = engine.PackedModel(sd, self.variant, self.uses_style), key
            self._cond_entry = None
        return self._packed

    def step_buffers(self, B, V=1, want_x0=False) -> engine.StepBuffers:
        """The device buffers of a loop over B clips x V variants (cached).  (Named so that `nn.Module.buffers()` stays what PyTorch's wrappers -
        nn.DataParallel.forward, the reference's default wrap, train.py:94 - expect it to be.)"""
        k = (B, V, want_x0, self.m_tile, self.layer_mode)
        if k not in self._bufs:
            if len(self._bufs) > 4:
                self._bufs.clear()
            self._bufs[k] = engine.StepBuffers(B, V, next(self.parameters()).device, want_x0, self.m_tile, self.layer_mode)
        return self._bufs[k]

    def variant_conds(self, y: dict, variants) -> torch.Tensor:
        """cond rows for a list of (uncond, uncond_audio, style_override) variants -> (V, B, 32, 512).
        Cached on the identity + version of the tensors in ``y`` so the 1000 calls of a sampling loop
        pay for the audio encoder once (the hoist of SURVEY.md §0.1)."""
        pm = self.packed()
        tens = [y.get(k) for k in ("audio", "word", "seed", "style_feature")]
        sig = tuple((id(t), t._version) if torch.is_tensor(t) else None for t in tens)
        vkey = tuple((bool(u), bool(ua), None if st is None else (id(st), st._version)) for u, ua, st in variants)
        ent = self._cond_entry
        if ent is not None and ent[0] == (sig, vkey, id(pm)):
            return ent[2]
        frame_cache, rows = {}, []
        for uncond, uncond_audio, style in variants:
            yy = y if style is None else dict(y, style_feature=style)
            rows.append(pm.conditioner.cond(yy, uncond, uncond_audio, frame_cache))
        cond = torch.stack(rows, 0).contiguous()
        self._cond_entry = ((sig, vkey, id(pm)), (tens, [v[2] for v in variants]), cond)   # keep refs: ids stay unique
        return cond

    def own_variant(self, y: dict):
        return (bool(y.get("uncond", False)), bool(y.get("uncond_audio", False)), None)

    # ---- reference-compatible call --------------------------------------------------------------
    def forward(self, x, timesteps, y=None, uncond_info=False):
        """x (B, 1536, 1, T=32), timesteps (B,) -> predicted x_0, same shape (models/denoiser.py:132-196)."""
        # nn.Module semantics follow self.training, not the autograd mode: train() under no_grad (a validation loss
        # computed without .eval()) still means batch statistics, DropPath and style dropout, as in the reference.
        if self.training or (torch.is_grad_enabled() and self.differentiable_eval):
            from . import training                      # differentiable path: HIP GEMMs fwd/dgrad/wgrad (training.py)
            return training.train_forward(self, x, timesteps, y, drop_path=self.drop_path)
        if torch.is_grad_enabled() and x.requires_grad:
            raise RuntimeError("MDM.eval() runs the fused inference kernels, which are not differentiable: the output would be "
                               "detached from x.  Set model.differentiable_eval = True (or call .train()) for gradients.")
        return self.forward_variants(x, timesteps, y, [self.own_variant(y)], None)

    def forward_variants(self, x, timesteps, y, variants, weights):
        """Evaluate V conditioning variants as one fused batch and return sum_v weights[:, v] * out_v
        (weights (3, V): one row per 512-channel body-part block; None for V == 1)."""
        engine._require_cuda(x, "x")
        if self.training and torch.is_grad_enabled():
            raise NotImplementedError("guidance / fused sampling is inference-only: call .eval() and torch.no_grad()")
        B, Cc, _, Tt = x.shape
        if Cc != engine.CH or Tt != engine.T:
            raise SynHipError(f"step kernels are specialised for (B,1536,1,32) latents, got {tuple(x.shape)}")
        V = len(variants)
        pm = self.packed()
        sb = self.step_buffers(B, V)
        with torch.no_grad():
            sb.cond.copy_(self.variant_conds(y, variants).reshape(-1, engine.D))
            sb.load_x(x)
            sb.t_model.copy_(timesteps.to(torch.int32).repeat(V))
            sb.t_coef.zero_()
            if V > 1:
                sb.cfg_w.copy_(weights.to(sb.cfg_w))
            engine.run_step(pm, sb, self._identity(), use_noise=False)
            return sb.read(sb.x)

    def _identity(self):
        if getattr(self, "_ident", None) is None or self._ident.device != next(self.parameters()).device:
            self._ident = engine.identity_coefs(next(self.parameters()).device)
        return self._ident


MDM_RVQ = MDM      # the name BASELINE.json's north_star uses for this class


def unwrap(model):
    """Strip nn.DataParallel / DDP / the respacing wrapper down to the module the drivers built."""
    seen = 0
    while seen < 8:
        if isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel)):
            model = model.module
        elif model.__class__.__name__ == "_WrappedModel":
            model = model.model
        else:
            break
        seen += 1
    return model
