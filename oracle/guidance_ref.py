"""Classifier-free-guidance combinations restated.  Oracle: test infrastructure only.

Follows reference diffusion/cfg_sampler.py:10-186.  Each function takes ``model_fn(x, t, y)`` and
performs the same number of evaluations, on the same flag sets, as the reference wrapper.
"""
from copy import deepcopy

import torch

PART_CHANNELS = {"upper_mask": (0, 512), "hands_mask": (512, 1024), "lower_mask": (1024, 1536)}


def _scale(v):
    return v.view(-1, 1, 1, 1)


def cfg(model_fn, x, t, y, eval_metric=False):
    # cfg_sampler.py:17-28 — note: mutates the CALLER's dict (audio masked on both passes)
    y["uncond_audio"] = True
    out = model_fn(x, t, y)
    yu = deepcopy(y)
    yu["uncond_audio"] = True
    yu["uncond"] = True
    out_u = model_fn(x, t, yu)
    if eval_metric:
        return out_u
    return out_u + _scale(y["scale"]) * (out - out_u)


def two_cfg(model_fn, x, t, y):
    # cfg_sampler.py:38-54
    yu = deepcopy(y); yu["uncond_audio"] = True; yu["uncond"] = True
    out_u = model_fn(x, t, yu)
    ya = deepcopy(y); ya["uncond_audio"] = True
    out_ua = model_fn(x, t, ya)
    yt = deepcopy(y); yt["uncond"] = True
    out_ut = model_fn(x, t, yt)
    return out_u + _scale(y["scale_audio"]) * (out_ut - out_u) + _scale(y["scale_prompt"]) * (out_ua - out_u)


def _keep_channels(out, lo, hi):
    m = torch.zeros(out.shape[1], dtype=torch.bool)
    m[lo:hi] = True
    return out * m.view(1, -1, 1, 1)


def two_cfg_bodypart(model_fn, x, t, y, audio_scale=1.0, prompt_scale=4.0, eval_metric=False):
    # cfg_sampler.py:67-117
    if eval_metric:
        yu = deepcopy(y); yu["uncond"] = True
        yu["scale_audio"] = torch.ones(1) * audio_scale
        yu["scale_prompt"] = torch.zeros(1)
        yu["style_feature"] = yu["style_feature"]["lower_mask"]
        return two_cfg(model_fn, x, t, yu)
    out = torch.zeros_like(x)
    for key, value in y["style_feature"].items():
        yp = deepcopy(y)
        if value is None:
            yp["style_feature"] = torch.zeros(1, 256)
            yp["scale_audio"] = torch.ones(1) * audio_scale
            yp["scale_prompt"] = torch.zeros(1)
        else:
            yp["style_feature"] = value
            yp["scale_audio"] = torch.ones(1) if key in "upper_mask" else torch.zeros(1)
            yp["scale_prompt"] = torch.ones(1) * prompt_scale
        out = out + _keep_channels(two_cfg(model_fn, x, t, yp), *PART_CHANNELS[key])
    return out


def cfg_bodypart(model_fn, x, t, y, eval_metric=False):
    # cfg_sampler.py:134-167
    yu = deepcopy(y); yu["uncond"] = True
    if eval_metric:
        yu["style_feature"] = yu["style_feature"]["lower_mask"]
        return model_fn(x, t, yu)
    out = torch.zeros_like(x)
    seen = torch.zeros(x.shape[1], dtype=torch.bool)
    for key, value in y["style_feature"].items():
        if value is None:
            continue
        yp = deepcopy(y); yp["style_feature"] = value; yp["uncond_audio"] = True
        lo, hi = PART_CHANNELS[key]
        seen[lo:hi] = True
        out = out + _keep_channels(model_fn(x, t, yp), lo, hi)
    yu["style_feature"] = torch.zeros(1, 256)
    out_u = model_fn(x, t, yu)
    out = out + out_u * (~seen).view(1, -1, 1, 1)
    return out_u + _scale(y["scale"]) * (out - out_u)
