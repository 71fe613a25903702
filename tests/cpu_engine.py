"""CPU stand-ins for the device engine (`syntalker_amd.engine.PackedModel / StepBuffers / StepGraph`), TEST INFRASTRUCTURE ONLY.

The product has no CPU path: `MDM` raises on CPU tensors and the three classes above are thin owners of device memory around the
C ABI.  The multi-process tests (gloo, no GPU in the build container) still have to drive the PRODUCT's host logic - `MDM.packed()`
/ `step_buffers()` / `variant_conds()` caching, `guidance.resolve`, `process._fused` (schedule hand-over between the 10-step and the
single-step graph, `first_clip` bookkeeping, x_T handling), `sharding.sample_sharded` - so these doubles implement the same
interfaces with the oracle's folded forward (`oracle/denoiser_ref.mdm_forward_folded`) as the step and a counter-based noise keyed
exactly like the kernels' (seed, step = t_coef, GLOBAL clip index).  `install(monkeypatch)` swaps them in.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import denoiser_ref as dr

T, CH, D = 32, 1536, 512


class _Conditioner:
    def __init__(self, pm):
        self.pm = pm

    def cond(self, y, uncond=False, uncond_audio=False, frame_cache=None):
        yy = dict(y)
        if uncond:
            yy["uncond"] = True
        if uncond_audio:
            yy["uncond_audio"] = True
        with torch.no_grad():
            return dr.clip_conditioning(self.pm.sd, yy, self.pm.fw, self.pm.variant, self.pm.use_motionclip)


class CpuPackedModel:
    """engine.PackedModel: the oracle's fold of the same state_dict."""

    def __init__(self, sd, variant, use_style, n_te=1000):
        self.sd = {k: (v.detach().float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in sd.items()}
        self.device = sd["input_process2.weight"].device
        self.variant, self.use_style = variant, use_style
        self.use_motionclip = bool(use_style and variant == "beatx")
        self.fw = dr.fold_weights(self.sd, variant, self.use_motionclip)
        self.te = dr.time_table(self.sd, self.fw, n_te)
        self.conditioner = _Conditioner(self)


class CpuStepBuffers:
    """engine.StepBuffers: the latent token-major (B, 32, 1536), the conditioning rows, the timestep vectors, the generator's key."""

    def __init__(self, B, V, device, want_x0=False, m_tile=0, layer_mode=0):
        self.B, self.V, self.fragment = B, V, False
        self.x = torch.zeros(B * T, CH)
        self.noise = torch.zeros(B * T, CH)
        self.x0 = torch.zeros(B * T, CH) if want_x0 else None
        self.cond = torch.zeros(V * B * T, D)
        self.t_model = torch.zeros(V * B, dtype=torch.int32)
        self.t_coef = torch.zeros(B, dtype=torch.int32)
        self.cfg_w = torch.zeros(B, 3, V) if V > 1 else None
        self.rng = torch.zeros(2, dtype=torch.int64)

    def load_x(self, x_bct):
        self.x.copy_(x_bct.detach().float().reshape(self.B, CH, T).transpose(1, 2).reshape(self.B * T, CH))

    def load_noise(self, eps_bct):
        self.noise.copy_(eps_bct.detach().float().reshape(self.B, CH, T).transpose(1, 2).reshape(self.B * T, CH))

    def set_rng(self, seed, first_clip=0):
        self.rng.copy_(torch.tensor([seed, first_clip], dtype=torch.int64))

    def check_sync(self):
        pass

    def read(self, src):
        return src.reshape(self.B, T, CH).transpose(1, 2).reshape(self.B, CH, 1, T).clone()


def keyed_noise(seed: int, step: int, clip: int) -> torch.Tensor:
    """N(0, 1) for one clip's (32, 1536) latent, a pure function of (seed, step, GLOBAL clip index) - the keying of syn_randn and
    of the in-epilogue generator (the values differ from Philox's: the tests compare rank counts, not generators)."""
    g = torch.Generator().manual_seed((int(seed) * 1000003 + int(step) * 7919 + int(clip)) % (2 ** 63 - 1))
    return torch.randn(T, CH, generator=g)


class CpuStepGraph:
    """engine.StepGraph: `replay()` advances `steps` steps of the device-side schedule."""

    MAX_STEPS = 1024

    def __init__(self, pm, sb, coef, use_noise=True, fused_rng=False, scheduled=False, steps=1):
        assert steps == 1 or scheduled
        self.pm, self.sb, self.coef, self.use_noise, self.fused_rng = pm, sb, coef, use_noise, fused_rng
        self.scheduled, self.steps = scheduled, steps
        self.sched = torch.zeros(self.MAX_STEPS, 2, dtype=torch.int32)
        self.counter = torch.zeros(1, dtype=torch.int32)
        self.replays = 0

    def set_schedule(self, t_coef_rows, t_model_rows):
        n = len(t_coef_rows)
        if n > self.MAX_STEPS:
            raise ValueError(f"schedule of {n} steps exceeds {self.MAX_STEPS}")
        self.sched[:n].copy_(torch.tensor(list(zip(t_coef_rows, t_model_rows)), dtype=torch.int32).reshape(-1, 2))
        self.counter.zero_()

    def copy_schedule_from(self, other):
        self.sched.copy_(other.sched)
        self.counter.copy_(other.counter)

    def _step(self):
        sb, pm = self.sb, self.pm
        B, V = sb.B, sb.V
        if self.scheduled:
            k = int(self.counter)
            sb.t_coef.fill_(int(self.sched[k, 0])); sb.t_model.fill_(int(self.sched[k, 1]))
            self.counter += 1
        x_bct = sb.read(sb.x)
        with torch.no_grad():
            outs = [dr.mdm_forward_folded(pm.sd, pm.fw, sb.cond.view(V, B, T, D)[v], pm.te, x_bct, sb.t_model.view(V, B)[v].long())
                    for v in range(V)]
        if V == 1:
            x0 = outs[0]
        else:                                               # the guidance combination, one weight row per 512-channel block
            x0 = torch.zeros_like(outs[0])
            for v in range(V):
                x0 = x0 + outs[v] * sb.cfg_w[:, :, v].repeat_interleave(CH // 3, dim=1).view(-1, CH, 1, 1)
        x0 = x0.reshape(B, CH, T).transpose(1, 2).reshape(B * T, CH)
        c = self.coef[sb.t_coef.long()].repeat_interleave(T, 0)                  # (B*T, 4)
        nxt = c[:, 0:1] * x0 + c[:, 1:2] * sb.x
        if self.use_noise:
            if self.fused_rng:
                seed, first = int(sb.rng[0]), int(sb.rng[1])
                eps = torch.cat([keyed_noise(seed, int(sb.t_coef[b]), first + b) for b in range(B)], 0)
            else:
                eps = sb.noise
            nxt = nxt + c[:, 2:3] * eps
        if sb.x0 is not None:
            sb.x0.copy_(x0)
        sb.x.copy_(nxt)

    def replay(self):
        self.replays += 1
        for _ in range(self.steps):
            self._step()


def install(monkeypatch):
    """Swap the device engine for the doubles above and let `MDM` accept CPU tensors (product code paths otherwise untouched)."""
    from syntalker_amd import engine
    monkeypatch.setattr(engine, "PackedModel", CpuPackedModel)
    monkeypatch.setattr(engine, "StepBuffers", CpuStepBuffers)
    monkeypatch.setattr(engine, "StepGraph", CpuStepGraph)
    monkeypatch.setattr(engine, "_require_cuda", lambda t, what: None)


def install_plain():
    """The same without pytest's monkeypatch (spawned worker processes): returns nothing, never undone."""
    from syntalker_amd import engine
    engine.PackedModel, engine.StepBuffers, engine.StepGraph = CpuPackedModel, CpuStepBuffers, CpuStepGraph
    engine._require_cuda = lambda t, what: None
