"""Device-side engine: packed weights, workspaces and the fused denoising step / sampling loop.

Layout in HBM (one GPU, everything resident for the whole loop):
    packed weights   bf16 MFMA-fragment order, 19.1 M params = 38 MB          (syn_pack_weight)
    x                token-major fp32 [B*32][1536] + bf16 shadow copy (GEMM operand), updated in place
    cond             fp32 [V*B*32][512], computed once per clip (conditioning.py)
    workspace        h fp32 [R][512]; xn/q/k/o bf16 [R][512]; vt bf16 [R*512]; hid bf16 [R][1024]   (R = V*B*32)
One step = one ``syn_denoise_step`` call = one kernel launch (k_stack for large batches, the persistent k_lat for
small ones), hipGraph-captured and replayed; the timestep enters through two device int32 vectors so the same graph
serves every step, and in the sampling loops those vectors are advanced on the device (`syn_step_advance`), so a loop
iteration is one graph replay and nothing else.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from . import tape as _tape
from .conditioning import ClipConditioner, fold_input_stage, rotary_tables, time_table

T, CH, D, FF, LAYERS = 32, 1536, 512, 1024, 8


# Parameters and buffers are also written where PyTorch's in-place version counters do not see it: `ClipAdam`'s update and the BatchNorm running
# statistics go through raw pointers, and a replayed `GraphedTrainStep` is one opaque launch.  Every such writer calls `note_raw_write()`; the caches of
# derived weights (`MDM.packed()`, `RVQVAE.packed()`) key on this count next to the tensors' versions, so the first sampling call after any training
# step folds and packs the weights again (the reference's trainer samples between epochs: diffusion_rvqvae_trainer.py `val` / `test`).
_raw_writes = 0


def note_raw_write():
    global _raw_writes
    _raw_writes += 1


def raw_write_epoch() -> int:
    return _raw_writes


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _lib.SynHipError(
            f"{what} is on {t.device}: the denoising path runs only on the MI355X HIP kernels "
            "(no CPU fallback). Move the model and inputs to a cuda device.")


def seq_pass_clips(n_variants: int, device) -> int:
    """Clips one full pass of the wave-per-sequence kernel takes: a workgroup per CU, its four waves = the variants of
    4 // V clips (V = 3 leaves a wave idle).  0 where that kernel does not apply."""
    device = torch.device(device)
    if device.type != "cuda" or not 1 <= n_variants <= 4:
        return 0
    return torch.cuda.get_device_properties(device).multi_processor_count * (4 // n_variants)


def plan_slices(n_clips: int, n_variants: int, device) -> list[tuple[int, int]]:
    """Contiguous clip ranges a batch is run as, one after the other (sequences never interact, noise is keyed by the
    global clip index: the result is the unsliced one).  Both whole-step kernels quantise to passes over the CUs - `k_seq`
    to passes of P = 1024 / V clips, `k_stack` to passes of 512 sequences - and `k_seq`'s pass is the cheaper per clip only
    when it is more than half full, so the library picks `k_stack` for q P + r clips with 0 < r <= P / 2 (1025-1536, ...):
    three `k_stack` passes where one `k_seq` pass and the best kernel for the r clips left do
    (1536 clips: 1.95 ms per step as one batch, 1.13 + 0.64 ms as 1024 + 512; `profiles/r03_diag_batch_sweep.txt`)."""
    P = seq_pass_clips(n_variants, device)
    if P == 0 or n_clips <= P or n_variants == 3:
        return [(0, n_clips)]
    r = n_clips % P
    if r == 0 or 2 * r > P:
        return [(0, n_clips)]
    return [(0, n_clips - r), (n_clips - r, n_clips)]


def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """fp32 (n, k) nn.Linear weight -> packed bf16 MFMA fragments (uint8 view of n*k*2 bytes)."""
    _require_cuda(w, "weight")
    w = w.detach().float().contiguous()
    n, k = w.shape
    out = torch.empty(n * k * 2, dtype=torch.uint8, device=w.device)
    _lib.check(_lib.load().syn_pack_weight(w.data_ptr(), n, k, out.data_ptr(), _lib.current_stream(w.device)), "syn_pack_weight")
    return out


class PackedModel:
    """Folded + packed, step-resident view of an MDM state_dict (inference; eval-mode semantics)."""

    def __init__(self, sd: dict, variant: str, use_style: bool, n_te: int = 1000):
        dev = sd["input_process2.weight"].device
        _require_cuda(sd["input_process2.weight"], "model")
        self.device, self.variant, self.use_style = dev, variant, use_style
        f32 = lambda t: t.detach().float().contiguous()
        folded = fold_input_stage(sd, use_style)
        self.folded = folded
        self.keep = []                       # every tensor the C struct points at
        m = _lib.SynModel()

        def hold(t):
            self.keep.append(t)
            return t.data_ptr()

        m.w_in = hold(pack_weight(folded["A"].float()))
        self.te = time_table(sd, folded["W2a"], n_te)
        m.te, m.n_te = hold(self.te), n_te
        rc, rs = rotary_tables(f32(sd["rel_pos.inv_freq"]), T)
        m.rot_cos, m.rot_sin = hold(rc), hold(rs)
        for i in range(LAYERS):
            p, L = f"mytimmblocks.{i}.", m.layer[i]
            L.ln1_g, L.ln1_b = hold(f32(sd[p + "norm1.weight"])), hold(f32(sd[p + "norm1.bias"]))
            L.w_qkv = hold(pack_weight(sd[p + "attn.qkv.weight"]))
            L.w_proj, L.b_proj = hold(pack_weight(sd[p + "attn.proj.weight"])), hold(f32(sd[p + "attn.proj.bias"]))
            L.ln2_g, L.ln2_b = hold(f32(sd[p + "norm2.weight"])), hold(f32(sd[p + "norm2.bias"]))
            L.w_fc1, L.b_fc1 = hold(pack_weight(sd[p + "mlp.fc1.weight"])), hold(f32(sd[p + "mlp.fc1.bias"]))
            L.w_fc2, L.b_fc2 = hold(pack_weight(sd[p + "mlp.fc2.weight"])), hold(f32(sd[p + "mlp.fc2.bias"]))
        m.w_out = hold(pack_weight(sd["output_process.poseFinal.weight"]))
        m.b_out = hold(f32(sd["output_process.poseFinal.bias"]))
        # the same weights as one tape of MFMA fragments in consumption order: the wave-per-sequence kernel (large batches)
        tp, tb = _tape.build_tape(sd, folded["A"])
        m.tape, m.tape_bias, m.tape_chunks = hold(tp), hold(tb), _tape.TAPE_FRAGS // _tape.CHUNK_FRAGS   # (+ LOOK_CHUNKS repeated chunks behind them)
        self.c = m
        self.conditioner = ClipConditioner({k: (v.detach() if torch.is_tensor(v) else v) for k, v in sd.items()},
                                           folded, variant, use_style)
        torch.cuda.current_stream(dev).synchronize()


class StepBuffers:
    """State + workspace for B clips x V conditioning variants; owns the syn_step struct."""

    def __init__(self, B: int, V: int, device, want_x0: bool = False, m_tile: int = 0, layer_mode: int = 0):
        self.B, self.V = B, V
        # latent layout: fragment order when the wave-per-sequence kernel runs the steps (the library's choice for large
        # single-variant batches, or pinned with layer_mode 5), token-major otherwise
        lib = _lib.load()
        self.fragment = bool(layer_mode == 5 or (layer_mode == 0 and m_tile == 0 and lib.syn_prefers_fragment_order(B, V)))
        R, Mb = V * B * T, B * T
        e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=device)
        bf = torch.bfloat16
        self.x, self.xb = e(Mb, CH), e(Mb, CH, dt=bf)
        self.noise = e(Mb, CH)
        self.rng = torch.zeros(2, dtype=torch.int64, device=device)     # {seed, first_clip} of the in-epilogue generator
        self.x0 = e(Mb, CH) if want_x0 else None
        self.cond = e(R, D)
        self.t_model = torch.zeros(V * B, dtype=torch.int32, device=device)
        self.t_coef = torch.zeros(B, dtype=torch.int32, device=device)
        self.cfg_w = e(B, 3, V) if V > 1 else None      # one weight table per clip (`copy_` broadcasts a (3, V) table over the clips)
        self.h = e(R, D)
        self.xn, self.q, self.k, self.o = e(R, D, dt=bf), e(R, D, dt=bf), e(R, D, dt=bf), e(R, D, dt=bf)
        self.vt, self.hid = e(R * D, dt=bf), e(R, FF, dt=bf)
        self.hc = e(3, Mb, D, dt=bf) if V > 1 else None
        self.sync = torch.zeros(320, dtype=torch.int32, device=device)   # small-batch path: barrier counters + error flag
        s = _lib.SynStep()
        s.n_clips, s.n_variants, s.m_tile = B, V, m_tile
        # 0 auto (small-batch kernel for few sequences, whole-step kernel otherwise); 4 / 3 / 5 pin one of them;
        # 1 = five kernels per block (the restatement the whole-step kernel is checked against bit for bit)
        s.reserved = layer_mode
        s.cond, s.t_model, s.cfg_w = self.cond.data_ptr(), self.t_model.data_ptr(), _lib.ptr(self.cfg_w)
        s.x_t, s.x_t_bf16, s.noise = self.x.data_ptr(), self.xb.data_ptr(), self.noise.data_ptr()
        s.t_coef = self.t_coef.data_ptr()
        s.rng = None
        s.x_next, s.x_next_bf16, s.pred_x0 = self.x.data_ptr(), self.xb.data_ptr(), _lib.ptr(self.x0)
        s.ws_h, s.ws_xn, s.ws_q, s.ws_k = self.h.data_ptr(), self.xn.data_ptr(), self.q.data_ptr(), self.k.data_ptr()
        s.ws_vt, s.ws_o, s.ws_hid, s.ws_hc = self.vt.data_ptr(), self.o.data_ptr(), self.hid.data_ptr(), _lib.ptr(self.hc)
        s.ws_sync = self.sync.data_ptr()
        # guided small batches: per-variant x0_hat, so that the variants of a clip can run on different XCDs
        self.x0v = e(R, CH) if (V > 1 and B * V <= 32) else None
        s.ws_x0v = _lib.ptr(self.x0v)
        # 9..128 sequences: exchange slots for the whole-step kernel's tensor-parallel mode (a tile split over 2 / 4 CUs)
        self.xch = e(V * B + 1, 8, 32 * D) if 8 < V * B <= 256 else None      # (129..256 sequences: 64-row tiles split in two, the same bytes per sequence)
        s.ws_xch = _lib.ptr(self.xch)
        s.x_fragment_order = int(self.fragment)
        s.cfg_w_clip_stride = 3 * V if V > 1 else 0
        self.c = s

    # layout ------------------------------------------------------------------------------------
    def _import(self, src: torch.Tensor, dst_f32, dst_bf16):
        lib = _lib.load()
        fn, name = (lib.syn_x_to_fragment, "syn_x_to_fragment") if self.fragment else (lib.syn_to_token_major, "syn_to_token_major")
        _lib.check(fn(src.data_ptr(), self.B, dst_f32.data_ptr(), _lib.ptr(dst_bf16), _lib.current_stream(src.device)), name)

    def load_x(self, x_bct: torch.Tensor):
        """(B,1536,1,32) fp32 -> x in the step kernel's layout + bf16 shadow."""
        self._import(x_bct.detach().float().contiguous(), self.x, self.xb)

    def load_noise(self, eps_bct: torch.Tensor):
        self._import(eps_bct.detach().float().contiguous(), self.noise, None)

    def set_rng(self, seed: int, first_clip: int = 0):
        """Key of the in-epilogue generator for the whole loop; the per-step stream id is the clip's t_coef."""
        self.rng.copy_(torch.tensor([seed, first_clip], dtype=torch.int64))

    def draw_noise(self, seed: int, step: int, first_clip: int = 0):
        """N(0,1) keyed by (seed, step, global element index): identical for any sharding of the batch."""
        n = self.B * T * CH
        dst = torch.empty_like(self.noise) if self.fragment else self.noise       # the generator's index space is token-major
        _lib.check(_lib.load().syn_randn(dst.data_ptr(), n, seed, step, first_clip * T * CH,
                                         _lib.current_stream(dst.device)), "syn_randn")
        if self.fragment:
            self.noise.view(-1).copy_(_tape.to_fragment_order(dst.view(self.B, T, CH)).view(-1))

    def check_sync(self):
        """The small-batch kernel's group barrier is bounded; a wait that ran out leaves a sticky flag."""
        flag = int(self.sync[256].item())
        if flag:
            self.sync.zero_()
            raise _lib.SynHipError(f"small-batch step kernel: group barrier wait ran out (flag {flag}); "
                                   "is another kernel occupying CUs of this device?")

    def read(self, src: torch.Tensor) -> torch.Tensor:
        self.check_sync()
        out = torch.empty(self.B, CH, 1, T, dtype=torch.float32, device=src.device)
        lib = _lib.load()
        fn, name = (lib.syn_x_from_fragment, "syn_x_from_fragment") if self.fragment else (lib.syn_from_token_major, "syn_from_token_major")
        _lib.check(fn(src.data_ptr(), self.B, out.data_ptr(), _lib.current_stream(src.device)), name)
        return out


def run_step(pm: PackedModel, sb: StepBuffers, coef: torch.Tensor, use_noise: bool = True, fused_rng: bool = False, steps: int = 1):
    """use_noise + fused_rng: the output GEMM's epilogue draws the noise, keyed by sb.rng = {seed, first_clip} and t_coef;
    use_noise only: the noise is read from sb.noise (injected or drawn by draw_noise)."""
    sb.c.coef = coef.data_ptr()
    sb.c.noise = sb.noise.data_ptr() if (use_noise and not fused_rng) else None
    sb.c.rng = sb.rng.data_ptr() if (use_noise and fused_rng) else None
    if steps > 1:        # sb.c.t_model / t_coef point at [steps][n] rows (syn_steps_advance fills them)
        if use_noise and not fused_rng:
            raise ValueError("injected noise is per step: multi-step launches draw theirs (fused_rng) or run without")
        _lib.check(_lib.load().syn_denoise_steps(C.byref(pm.c), C.byref(sb.c), steps, sb.t_model.numel(), sb.t_coef.numel(),
                                                 _lib.current_stream(pm.device)), "syn_denoise_steps")
        return
    _lib.check(_lib.load().syn_denoise_step(C.byref(pm.c), C.byref(sb.c), _lib.current_stream(pm.device)), "syn_denoise_step")


class StepGraph:
    """hipGraph of one step; replays read the timestep from sb.t_model / sb.t_coef (device memory).
    With ``scheduled=True`` the graph starts with `syn_step_advance`: the timestep vectors come from a device-resident
    schedule (`set_schedule`) and a loop iteration is nothing but `replay()`."""

    MAX_STEPS = 1024

    def __init__(self, pm: PackedModel, sb: StepBuffers, coef: torch.Tensor, use_noise: bool = True, fused_rng: bool = False,
                 scheduled: bool = False, steps: int = 1):
        """``steps`` > 1 (scheduled graphs only): that many consecutive steps per replay - no launch gaps between them."""
        assert steps == 1 or scheduled
        self.pm, self.sb, self.coef, self.scheduled, self.steps = pm, sb, coef, scheduled, steps
        if scheduled:
            self.sched = torch.zeros(self.MAX_STEPS, 2, dtype=torch.int32, device=pm.device)
            self.counter = torch.zeros(1, dtype=torch.int32, device=pm.device)
        if steps > 1:
            self.tm_rows = torch.zeros(steps, sb.t_model.numel(), dtype=torch.int32, device=pm.device)
            self.tc_rows = torch.zeros(steps, sb.t_coef.numel(), dtype=torch.int32, device=pm.device)
        if scheduled:
            # The warm-up launch below reads the timestep vectors as they stand, and a cached StepBuffers may carry the indices of
            # an earlier loop over a LONGER coefficient table (p_sample_loop's 1000 rows, then ddim_sample_loop's 50: row 999 of a
            # 50-row table is an out-of-bounds read).  A scheduled graph overwrites them from its schedule before every step anyway.
            sb.t_model.zero_(); sb.t_coef.zero_()
        side = torch.cuda.Stream(device=pm.device)
        side.wait_stream(torch.cuda.current_stream(pm.device))
        with torch.cuda.stream(side):          # warm-up launch outside capture (module load, etc.), the same launch(es) as captured
            x_save, xb_save = sb.x.clone(), sb.xb.clone()
            if steps > 1:
                try:
                    sb.c.t_model, sb.c.t_coef = self.tm_rows.data_ptr(), self.tc_rows.data_ptr()
                    run_step(pm, sb, coef, use_noise, fused_rng, steps=steps)
                finally:
                    sb.c.t_model, sb.c.t_coef = sb.t_model.data_ptr(), sb.t_coef.data_ptr()
            else:
                run_step(pm, sb, coef, use_noise, fused_rng)
            sb.x.copy_(x_save); sb.xb.copy_(xb_save)
        torch.cuda.current_stream(pm.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            if steps > 1:
                # one schedule kernel per replay: it fills a row of timestep vectors per captured step, and every captured
                # launch points at its own row (same-box A/B at B = 1: 160.5 -> 158.3 us per step)
                _lib.check(_lib.load().syn_steps_advance(self.sched.data_ptr(), self.counter.data_ptr(), self.tm_rows.data_ptr(),
                                                         sb.t_model.numel(), self.tc_rows.data_ptr(), sb.t_coef.numel(), steps,
                                                         _lib.current_stream(pm.device)), "syn_steps_advance")
                # syn_denoise_steps: `steps` launches for token-major latents; ONE persistent launch for fragment-order latents
                # (every workgroup takes its sequences through all the steps, the workgroups drift out of phase)
                try:
                    sb.c.t_model, sb.c.t_coef = self.tm_rows.data_ptr(), self.tc_rows.data_ptr()
                    run_step(pm, sb, coef, use_noise, fused_rng, steps=steps)
                finally:
                    sb.c.t_model, sb.c.t_coef = sb.t_model.data_ptr(), sb.t_coef.data_ptr()
            else:
                if scheduled:
                    _lib.check(_lib.load().syn_step_advance(self.sched.data_ptr(), self.counter.data_ptr(), sb.t_model.data_ptr(),
                                                            sb.t_model.numel(), sb.t_coef.data_ptr(), sb.t_coef.numel(),
                                                            _lib.current_stream(pm.device)), "syn_step_advance")
                run_step(pm, sb, coef, use_noise, fused_rng)
        # The first launch of an instantiated hipGraph uploads it to the device (measured at 1024 clips: the first 10-step replay takes
        # 12.6 ms, the following ones 11.1 - the stream idles ~1.5 ms while the host sets the launch up).  That belongs to building the
        # graph, like the capture: one replay here, on a snapshot of the latent (scheduled graphs read schedule row 0; the counter is
        # put back).
        x_save, xb_save = sb.x.clone(), sb.xb.clone()
        tm_save, tc_save = sb.t_model.clone(), sb.t_coef.clone()
        self.graph.replay()
        sb.x.copy_(x_save); sb.xb.copy_(xb_save); sb.t_model.copy_(tm_save); sb.t_coef.copy_(tc_save)
        if scheduled:
            self.counter.zero_()

    def set_schedule(self, t_coef_rows, t_model_rows):
        """Rows of the coefficient table and original timesteps of the coming replays, in order (<= MAX_STEPS)."""
        n = len(t_coef_rows)
        if n > self.MAX_STEPS:
            raise ValueError(f"schedule of {n} steps exceeds {self.MAX_STEPS}")
        host = torch.tensor(list(zip(t_coef_rows, t_model_rows)), dtype=torch.int32).reshape(-1, 2)
        self.sched[:n].copy_(host)
        self.counter.zero_()

    def copy_schedule_from(self, other: "StepGraph"):
        """Continue `other`'s schedule where it stands (a multi-step graph handing over to the single-step one)."""
        self.sched.copy_(other.sched)
        self.counter.copy_(other.counter)

    def replay(self):
        self.graph.replay()


# ---------------------------------------------------------------------------------------------
def posterior_coefs(tab: dict, device) -> torch.Tensor:
    """DDPM ancestral step as x_next = c0*x0 + c1*x_t + sigma*eps (gaussian_diffusion.py:255-277, 546-556)."""
    n = len(tab["betas"])
    c = np.zeros((n, 4), np.float64)
    c[:, 0], c[:, 1] = tab["posterior_mean_coef1"], tab["posterior_mean_coef2"]
    c[:, 2] = np.exp(0.5 * tab["posterior_log_variance_clipped"])
    c[0, 2] = 0.0                                         # nonzero_mask: no noise at t == 0
    return torch.tensor(c, dtype=torch.float32, device=device)


def ddim_coefs(tab: dict, eta: float, device) -> torch.Tensor:
    """DDIM step in the same linear form (gaussian_diffusion.py:771-791):
         eps_hat = (sqrt(1/ab) x_t - x0) / sqrt(1/ab - 1)
         x_next  = sqrt(ab_prev) x0 + sqrt(1 - ab_prev - sigma^2) eps_hat + sigma eps."""
    ab, abp = tab["alphas_cumprod"], tab["alphas_cumprod_prev"]
    sigma = eta * np.sqrt((1 - abp) / (1 - ab)) * np.sqrt(1 - ab / abp)
    dirc = np.sqrt(1 - abp - sigma ** 2)
    r, rm1 = tab["sqrt_recip_alphas_cumprod"], tab["sqrt_recipm1_alphas_cumprod"]
    c = np.zeros((len(ab), 4), np.float64)
    c[:, 0] = np.sqrt(abp) - dirc / rm1
    c[:, 1] = dirc * r / rm1
    c[:, 2] = sigma
    c[0, 2] = 0.0
    return torch.tensor(c, dtype=torch.float32, device=device)


def identity_coefs(device) -> torch.Tensor:
    """x_next = x0: a bare model evaluation (MDM.forward)."""
    return torch.tensor([[1.0, 0.0, 0.0, 0.0]], dtype=torch.float32, device=device)
