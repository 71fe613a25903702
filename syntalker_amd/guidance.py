"""Classifier-free-guidance wrappers — drop-ins for reference diffusion/cfg_sampler.py:10-167.

The reference evaluates the wrapped model 2 / 3 / up to 9 times per step on deep-copied kwargs and
combines the outputs.  Every combination is linear with coefficients that sum to one per channel, so
here a wrapper only *plans* the evaluation: a de-duplicated list of conditioning variants plus a
(3, V) weight table (one row per 512-channel body-part block).  A HIP-backed ``MDM`` then runs all V
variants of all clips as ONE fused batch sharing x_t, and folds the combination into the operand of
the output GEMM (engine / syn_denoise_step).  Any other wrapped module falls back to calling it once
per variant (same arithmetic, torch ops) so the wrappers stay usable with arbitrary models.

Quirk kept from the reference: ``ClassifierFreeSampleModel.forward`` sets ``y['uncond_audio'] = True``
on the CALLER's dict (cfg_sampler.py:18).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .denoiser import MDM, unwrap

PART_BLOCK = {"upper_mask": 0, "hands_mask": 1, "lower_mask": 2}      # cfg_sampler.py:171-186


def _scalar(v, what):
    """A guidance scale as the reference's wrappers take it - `y['scale'].view(-1, 1, 1, 1)`, one value for the batch or one per clip
    (cfg_sampler.py:28,54,167) - as a python float when it is one value, else as a 1-d fp32 tensor of per-clip scales."""
    if torch.is_tensor(v):
        if v.numel() == 1 or not bool((v != v.flatten()[0]).any()):
            return float(v.flatten()[0])
        return v.detach().reshape(-1).float()
    return float(v)


def _is_zero(w) -> bool:
    return (not torch.is_tensor(w)) and w == 0.0


class _Plan:
    """variants: list of (uncond, uncond_audio, style_override|None); weights: (3, V) lists of python floats or, for per-clip guidance scales,
    1-d tensors of one weight per clip."""

    def __init__(self, null_style_on_uncond: bool = True):
        # MDM replaces the style vector by its null row when `uncond` is set (denoiser_h3d.py:116-124), so a
        # style override is irrelevant there and such variants merge; for arbitrary wrapped modules they do not.
        self.variants, self._index, self.weights = [], {}, [[], [], []]
        self.merge = null_style_on_uncond

    def add(self, uncond, uncond_audio, style, w3):
        if all(_is_zero(w) for w in w3):            # a zero-weight evaluation cannot change the result: skip it
            return
        drop = uncond and self.merge
        key = (bool(uncond), bool(uncond_audio), None if (style is None or drop) else id(style))
        if key not in self._index:
            self._index[key] = len(self.variants)
            self.variants.append((bool(uncond), bool(uncond_audio), None if drop else style))
            for r in self.weights:
                r.append(0.0)
        v = self._index[key]
        for c in range(3):
            self.weights[c][v] += w3[c]

    def tensor(self, device):
        """(3, V) weights, or (B, 3, V) when a scale came per clip."""
        flat = [w for row in self.weights for w in row]
        if not any(torch.is_tensor(w) for w in flat):
            return torch.tensor(self.weights, dtype=torch.float32, device=device)
        B = max(w.numel() for w in flat if torch.is_tensor(w))
        cols = []
        for w in flat:
            if torch.is_tensor(w):
                if w.numel() != B:
                    raise ValueError(f"per-clip guidance scales of different lengths ({w.numel()} and {B})")
                cols.append(w.to(device=device, dtype=torch.float32))
            else:
                cols.append(torch.full((B,), float(w), dtype=torch.float32, device=device))
        return torch.stack(cols, 1).view(B, 3, len(self.variants))


def _run(model, x, timesteps, y, plan: _Plan):
    inner = unwrap(model)
    if isinstance(inner, MDM):
        return inner.forward_variants(x, timesteps, y, plan.variants, plan.tensor(x.device))
    out = None                                     # generic module: one call per variant
    W = plan.tensor(x.device)
    for v, (uncond, uncond_audio, style) in enumerate(plan.variants):
        yy = dict(y)
        if uncond:
            yy["uncond"] = True
        if uncond_audio:
            yy["uncond_audio"] = True
        if style is not None:
            yy["style_feature"] = style
        o = model(x, timesteps, yy)
        w = (W[:, v].repeat_interleave(o.shape[1] // 3).view(1, -1, 1, 1) if W.dim() == 2
             else W[:, :, v].repeat_interleave(o.shape[1] // 3, dim=1).view(W.shape[0], -1, 1, 1))
        out = o * w if out is None else out + o * w
    return out


class ClassifierFreeSampleModel(nn.Module):
    """out_u + scale * (out_c - out_u), audio masked on both passes (cfg_sampler.py:10-28)."""

    def __init__(self, model, eval=False):
        super().__init__()
        self.model = model
        self.eval_metric = eval

    def plan(self, y, mdm=True):
        p = _Plan(mdm)
        if self.eval_metric:
            p.add(True, True, None, [1.0] * 3)
        else:
            s = _scalar(y["scale"], "scale")
            p.add(bool(y.get("uncond", False)), True, None, [s] * 3)
            p.add(True, True, None, [1.0 - s] * 3)
        return p

    def forward(self, x, timesteps, y=None):
        y["uncond_audio"] = True
        return _run(self.model, x, timesteps, y, self.plan(y, isinstance(unwrap(self.model), MDM)))


def _two_cfg_into(p: _Plan, y, style, sa, sp, blocks):
    """out_u + sa*(out_{text masked} - out_u) + sp*(out_{audio masked} - out_u)  (cfg_sampler.py:38-54)."""
    w = lambda c: [c if b in blocks else 0.0 for b in range(3)]
    ua0 = bool(y.get("uncond_audio", False))
    u0 = bool(y.get("uncond", False))
    p.add(True, True, style, w(1.0 - sa - sp))
    p.add(u0, True, style, w(sp))
    p.add(True, ua0, style, w(sa))


class TwoClassifierFreeSampleModel(nn.Module):
    def __init__(self, model, eval=False):
        super().__init__()
        self.model = model
        self.eval_metric = eval

    def plan(self, y, mdm=True):
        p = _Plan(mdm)
        _two_cfg_into(p, y, None, _scalar(y["scale_audio"], "scale_audio"), _scalar(y["scale_prompt"], "scale_prompt"),
                      (0, 1, 2))
        return p

    def forward(self, x, timesteps, y=None):
        return _run(self.model, x, timesteps, y, self.plan(y, isinstance(unwrap(self.model), MDM)))


class TwoClassifierFreeSampleModel_Bodypart(nn.Module):
    """Per body part: its own prompt (or none) and its own (audio, prompt) scales; outputs restricted to
    the part's 512 channels and summed (cfg_sampler.py:57-117)."""

    def __init__(self, model, eval=False):
        super().__init__()
        self.model = TwoClassifierFreeSampleModel(model)
        self.latent_dim = 1536
        self.eval_metric = eval
        self.audio_scale = 1
        self.prompt_scale = 4

    def plan(self, y, mdm=True):
        p = _Plan(mdm)
        if self.eval_metric:       # y_uncond: uncond=True, style <- lower prompt (irrelevant once uncond), scales (audio, 0)
            _two_cfg_into(p, dict(y, uncond=True), None, float(self.audio_scale), 0.0, (0, 1, 2))
            return p
        for key, value in y["style_feature"].items():
            blk = (PART_BLOCK[key],)
            if value is None:
                if getattr(self, "_zero", None) is None or self._zero.device != y["seed"].device:
                    self._zero = torch.zeros(1, 256, device=y["seed"].device)      # cfg_sampler.py:84
                zero = self._zero
                _two_cfg_into(p, y, zero, float(self.audio_scale), 0.0, blk)
            else:
                sa = 1.0 if key in "upper_mask" else 0.0
                _two_cfg_into(p, y, value, sa, float(self.prompt_scale), blk)
        return p

    def forward(self, x, timesteps, y=None):
        return _run(self.model.model, x, timesteps, y, self.plan(y, isinstance(unwrap(self.model.model), MDM)))


class ClassifierFreeSampleModel_Bodypart(nn.Module):
    """cfg_sampler.py:125-167: prompted parts use (style=part prompt, audio masked); the rest and the
    unconditional pass use the learned null style with audio on."""

    def __init__(self, model, eval=False):
        super().__init__()
        self.model = model
        self.latent_dim = 1536
        self.eval_metric = eval

    def plan(self, y, mdm=True):
        p = _Plan(mdm)
        ua0 = bool(y.get("uncond_audio", False))
        if self.eval_metric:
            p.add(True, ua0, None, [1.0] * 3)
            return p
        s = _scalar(y["scale"], "scale")
        seen = set()
        for key, value in y["style_feature"].items():
            if value is None:
                continue
            b = PART_BLOCK[key]
            seen.add(b)
            p.add(bool(y.get("uncond", False)), True, value, [s if c == b else 0.0 for c in range(3)])
        # unconditional pass: weight (1-s) on prompted blocks, 1 on the others
        p.add(True, ua0, None, [(1.0 - s) if c in seen else 1.0 for c in range(3)])
        return p

    def forward(self, x, timesteps, y=None):
        return _run(self.model, x, timesteps, y, self.plan(y, isinstance(unwrap(self.model), MDM)))


def resolve(model):
    """-> (MDM, plan_fn) if `model` is a HIP MDM possibly under guidance / DataParallel / respacing
    wrappers, else (None, None).  plan_fn(y) -> _Plan (and applies the wrapper's side effects on y)."""
    m = unwrap(model)
    if isinstance(m, MDM):
        def single(y):
            p = _Plan()
            p.add(*m.own_variant(y)[:2], None, [1.0] * 3)
            return p
        return m, single
    if isinstance(m, ClassifierFreeSampleModel):
        inner = unwrap(m.model)
        if isinstance(inner, MDM):
            def plan_cfg(y):
                y["uncond_audio"] = True
                return m.plan(y)
            return inner, plan_cfg
    if isinstance(m, (TwoClassifierFreeSampleModel, ClassifierFreeSampleModel_Bodypart)):
        inner = unwrap(m.model)
        if isinstance(inner, MDM):
            return inner, m.plan
    if isinstance(m, TwoClassifierFreeSampleModel_Bodypart):
        inner = unwrap(m.model.model)
        if isinstance(inner, MDM):
            return inner, m.plan
    return None, None
