"""Training path of the denoiser (e): `training_losses` forward + backward, every op on hand-written HIP kernels.

  * every nn.Linear (time MLP, embed_text, the input stage's five, the blocks' four, poseFinal) runs forward, data gradient and weight gradient on
    the MFMA GEMM through the C ABI (bf16 operands, fp32 accumulate / output): `HipLinearFn` for a Linear on its own, `InputStageFn` for the
    input stage as one autograd node, `StackFn` / `AttnBranchFn` / `MlpBranchFn` for the eight pre-LN blocks;
  * the seams between them - the reference's torch.cat / avg_pool1d / permute / embedding / masked_l2 (models/denoiser.py:147-176,
    gaussian_diffusion.py:202-215) - are the kernels of csrc/syn_glue.inc: GEMM operands are written in the form the GEMM takes them, gradients
    are read where the GEMM left them;
  * the WavEncoder (78 % of the training FLOPs, cannot be hoisted in training): every Conv1d forward / data gradient / weight gradient on
    split-operand MFMA kernels (fp32-grade), the 1-2-channel first layer on fp32 FMAs / the fp32 matrix pipe, BatchNorm on batch statistics
    folded into its neighbours (`WavBlockFn`: a BasicBlock as one autograd node; `ConvSplitFn` / `BnActFn` / `SyncBnActFn`: the per-convolution
    nodes SyncBatchNorm and eval-mode statistics take).  No library convolution is on this path: a layer geometry the kernels do not cover raises;
  * data parallelism, optimizer, captured step: `optim.py` (re-exported here).
Train-mode semantics follow the reference: BatchNorm batch statistics, DropPath(0.1) per sample with scale-by-keep
(timm_transformer/transformer.py:21-38), h3d Bernoulli(0.3) style dropout (denoiser_h3d.py:116-124).  There is no CPU fallback: CPU tensors raise.
Batches whose size is not a multiple of 4 are padded with empty clips behind the audio encoder (the kernels work on 128-row tiles); the padding
rows carry zero gradients and are dropped from the output.
"""
from __future__ import annotations

import ctypes as C
import math
import os as _os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, engine
from .optim import (DDP_BUCKET_MB, ClipAdam, GraphedTrainStep, _avg_comm_hook, _check_clip, _grad_out, _into_bound_buffers,   # noqa: F401 (re-exports)
                    _reset_handed, bind_grad_buffers, ddp_bucket_sizes, direct_grad_report, make_ddp, train_step, unbind_grad_buffers,
                    unused_in_forward)


def _ceil_to(a: int, m: int) -> int:
    return (a + m - 1) // m * m


def _f32c(t):
    t = t.detach()
    if t.dtype is not torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# ---- nn.Linear on the MFMA GEMM --------------------------------------------------------------------------------------------------------------
# A weight W [N][K] (fp32 master) is used through two bf16 fragment sets: W for the forward GEMM x W^T and W^T for the data-gradient GEMM dy W.
# K is zero-padded to the GEMM's 128-column granularity (text_encoder_body: 300 -> 384); N % 128 == 0 for every Linear of the model.

def _pack_t(src: torch.Tensor, n: int, k: int) -> torch.Tensor:
    """Packed fragments of W = src^T for a row-major src [k][n] (fp32 or bf16), no transposed copy."""
    out = torch.empty(n * k * 2, dtype=torch.uint8, device=src.device)
    _lib.check(_lib.load().syn_pack_weight_t(src.data_ptr(), int(src.dtype is torch.bfloat16), n, k, out.data_ptr(),
                                             _lib.current_stream(src.device)), "syn_pack_weight_t")
    return out


class WeightPacks:
    """The bf16 fragment sets the step's Linear layers take - W for the forward GEMM, W^T for the data-gradient GEMM - packed from
    the fp32 master weights in ONE launch per step (`syn_pack_weights`) instead of one launch per use: the weights only change in
    optimizer.step().  `refresh()` at the top of every training forward; a weight is looked up by object identity and in-place version,
    so a stale or foreign tensor simply takes the pack-on-the-spot path (`_packs_of`)."""

    def __init__(self, weights):
        import numpy as np
        self.owner = lambda: None                                # the model the cache belongs to (weak reference, set by its user)
        self.items = {}
        jobs, self.max_frag = [], 0
        ok = []
        for w in weights:
            if w.is_cuda and w.dtype is torch.float32 and w.is_contiguous() and w.dim() == 2 and w.shape[0] % 128 == 0 and not any(w is o for o in ok):
                ok.append(w)
        # two arenas - all forward sets, all transposed sets - so that either can be pulled into the memory-side cache with one pass (`prefetch`)
        nb = sum(w.shape[0] * _ceil_to(w.shape[1], 128) * 2 for w in ok)
        dev = ok[0].device if ok else None
        self.arena_fwd = torch.empty(nb, dtype=torch.uint8, device=dev) if ok else None
        self.arena_tr = torch.empty(nb, dtype=torch.uint8, device=dev) if ok else None
        of = 0
        for w in ok:
            N, K = w.shape
            Kp = _ceil_to(K, 128)
            fwd, tr = self.arena_fwd[of:of + N * Kp * 2], self.arena_tr[of:of + N * Kp * 2]
            of += N * Kp * 2
            pad = K if Kp != K else 0
            jobs.append((w.data_ptr(), fwd.data_ptr(), N, Kp, 0, pad))       # fragments of W [N][Kp]
            jobs.append((w.data_ptr(), tr.data_ptr(), Kp, N, 1, pad))        # fragments of W^T [Kp][N] from the row-major [N][K]
            self.max_frag = max(self.max_frag, (N // 16) * (Kp // 32))
            self.items[id(w)] = [weakref.ref(w), w.data_ptr(), -1, fwd, tr]
        self.n_jobs = len(jobs)
        if jobs:
            arr = np.array(jobs, dtype=np.dtype([("src", "<u8"), ("out", "<u8"), ("n", "<i4"), ("k", "<i4"), ("t", "<i4"), ("src_dim", "<i4")]))
            self.jobs = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)

    def valid(self) -> bool:
        return all(r() is not None and r().data_ptr() == ptr for r, ptr, *_ in self.items.values())

    def prefetch(self, transposed: bool):
        """One read pass over the forward / transposed fragment sets: whatever has been evicted from the memory-side cache since the
        pack comes back before a latency-bound consumer asks for it fragment by fragment."""
        a = self.arena_tr if transposed else self.arena_fwd
        if a is not None and a.numel() >= 64:
            _lib.check(_lib.load().syn_touch(a.data_ptr(), a.numel(), _lib.current_stream(a.device)), "syn_touch")

    def refresh(self):
        if not self.n_jobs:
            return
        _lib.check(_lib.load().syn_pack_weights(self.jobs.data_ptr(), self.n_jobs, self.max_frag, _lib.current_stream(self.jobs.device)),
                   "syn_pack_weights")
        for it in self.items.values():
            it[2] = it[0]()._version

    def lookup(self, w):
        it = self.items.get(id(w))
        if it is None or it[0]() is not w or it[2] != w._version or it[1] != w.data_ptr():
            return None, None
        return it[3], it[4]


_packs: "WeightPacks | None" = None             # the step's packs of every Linear outside the blocks
_packs_blocks: "WeightPacks | None" = None      # the transformer blocks' Linears: packed right in front of the blocks (see train_forward)


def _lookup_packs(w):
    for pk in (_packs, _packs_blocks):
        if pk is not None:
            r = pk.lookup(w)
            if r[0] is not None:
                return r
    return None, None


def _packs_of(w):
    """(W fragments, W^T fragments) of a Linear weight [N][K], K zero-padded to 128: the step's packs, or packed on the spot."""
    fwd, tr = _lookup_packs(w)
    if fwd is not None:
        return fwd, tr
    N, K = w.shape
    if N % 128 or w.dim() != 2:
        raise _lib.SynHipError(f"Linear weight {tuple(w.shape)}: the GEMM kernels take output widths that are multiples of 128 (every Linear of the "
                               "reference's denoiser is: models/denoiser.py:86-109)")
    engine._require_cuda(w, "Linear weight")
    Kp = _ceil_to(K, 128)
    wf = w.detach().float()
    if Kp != K:
        wf = F.pad(wf, (0, Kp - K))
    wf = wf.contiguous()
    return engine.pack_weight(wf), _pack_t(wf, Kp, N)


def _bf16_rows(x, K: int, Kp: int):
    """(..., K) -> bf16 contiguous [M][Kp] GEMM operand (columns K .. Kp zero)."""
    xb = x.detach().reshape(-1, K)
    if xb.dtype is not torch.bfloat16:
        xb = xb.to(torch.bfloat16)
    if Kp != K:
        xb = F.pad(xb, (0, Kp - K))
    return xb.contiguous()


def _gemm(xb, wp, n: int, k: int, bias=None):
    """fp32 y[M][n] = xb[M][k] (bf16, contiguous) . W^T (+ bias) for packed W[n][k]; n % 128 == 0, k % 128 == 0."""
    y = torch.empty(xb.shape[0], n, dtype=torch.float32, device=xb.device)
    _lib.check(_lib.load().syn_linear(xb.data_ptr(), wp.data_ptr(), _lib.ptr(bias), xb.shape[0], n, k, y.data_ptr(), _lib.current_stream(y.device)),
               "syn_linear")
    return y


def _lin_fwd(xb, w, b, res=None, scale=None, rows_per_scale=1, gelu_out=None):
    """bf16 rows [M][Kp] -> fp32 [M][N] = x W^T + b, or res + scale[row // rows_per_scale] * (x W^T + b), or (gelu_out given) also
    bf16(GELU(.)) into gelu_out; and what the backward takes: the x^T fragments of the weight-gradient GEMM (packed by the same launch
    while M <= 2048) and the W^T fragments as of THIS forward (the step's pack cache may have moved on by then).  M % 32 == 0."""
    pk, wt = _packs_of(w)
    M, K = xb.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=xb.device)
    xt = torch.empty(K * M * 2, dtype=torch.uint8, device=xb.device)
    lib, st = _lib.load(), _lib.current_stream(xb.device)
    bc = None if b is None else _f32c(b)
    if gelu_out is not None:                   # fc1: y stays fp32 for the backward, gelu_out receives bf16(GELU(y)) - fc2's operand
        _lib.check(lib.syn_linear_gelu(xb.data_ptr(), pk.data_ptr(), _lib.ptr(bc), M, N, K, y.data_ptr(), gelu_out.data_ptr(), xt.data_ptr(), st),
                   "syn_linear_gelu")
    elif res is not None:
        _lib.check(lib.syn_linear_res(xb.data_ptr(), pk.data_ptr(), _lib.ptr(bc), res.data_ptr(), _lib.ptr(scale), rows_per_scale, M, N, K,
                                      y.data_ptr(), xt.data_ptr(), st), "syn_linear_res")
    else:
        _lib.check(lib.syn_linear_and_pack(xb.data_ptr(), pk.data_ptr(), _lib.ptr(bc), M, N, K, y.data_ptr(), xt.data_ptr(), st), "syn_linear_and_pack")
    return y, (xt, wt)


def _lin_bwd(dy, packs, w, has_bias, scale=None, rows_per_scale=1, owners=(None, None), ld=None, row_div=1, cscale=1.0, rows=None, want_dx=True):
    """Backward of y = x W^T + b from its output gradient: dx [M][Kp] (None unless want_dx), dW [N][Kp], db [N].  The gradient is `dy`, fp32
    [M][N] contiguous - or, with ld, a DEVICE POINTER to rows of pitch ld floats of which row r / row_div is this Linear's row r (`rows` of
    them) - times cscale, times its rows' factors (scale).  packs: (x^T fragments, W^T fragments) from `_lin_fwd`.  M % 128 == 0."""
    xt, wt = packs
    N = w.shape[0]
    K = wt.numel() // (2 * N)                    # the padded width the forward ran at
    if ld is None:
        M, ptr_, ld = dy.shape[0], dy.data_ptr(), N
        dev = dy.device
    else:
        M, ptr_, dev = rows, dy, w.device
    lib, st = _lib.load(), _lib.current_stream(dev)
    dyb = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    dybt = torch.empty(N, M, dtype=torch.bfloat16, device=dev)
    part = torch.empty(M // 64, N, dtype=torch.float32, device=dev) if has_bias else None
    _lib.check(lib.syn_linear_bwd_prep(ptr_, ld, row_div, float(cscale), M, N, _lib.ptr(scale), rows_per_scale, dyb.data_ptr(), dybt.data_ptr(), _lib.ptr(part),
                                       st), "syn_linear_bwd_prep")
    exact = K == w.shape[1]                      # (a zero-padded K: the weight gradient comes out padded and its columns are cut below)
    dw = _grad_out(owners[0], (N, K)) if (owners[0] is not None and exact) else torch.empty(N, K, dtype=torch.float32, device=dev)
    db = (_grad_out(owners[1], (N,)) if owners[1] is not None else torch.empty(N, dtype=torch.float32, device=dev)) if has_bias else None
    dx = None
    if want_dx:
        dx = torch.empty(M, K, dtype=torch.float32, device=dev)
        _lib.check(lib.syn_linear_pair(dyb.data_ptr(), wt.data_ptr(), M, K, N, dx.data_ptr(), dybt.data_ptr(), xt.data_ptr(), N, K, M, dw.data_ptr(),
                                       _lib.ptr(part), M // 64, N, _lib.ptr(db), st), "syn_linear_pair")
    else:
        _lib.check(lib.syn_linear(dybt.data_ptr(), xt.data_ptr(), None, N, K, M, dw.data_ptr(), st), "syn_linear")
        if has_bias:
            _lib.check(lib.syn_colsum_parts(part.data_ptr(), M // 64, N, db.data_ptr(), st), "syn_colsum_parts")
    if not exact:
        dw = dw[:, :w.shape[1]].contiguous()
    return dx, dw, db


class HipLinearFn(torch.autograd.Function):
    """y = x W^T + b for x (..., K): forward, data gradient and weight gradient on the MFMA GEMM (bf16 operands, fp32 accumulate).
    Up to 64 rows (the timestep MLP, embed_text: one row per clip) the weight / bias gradient is one fp32 launch (`syn_linear_wgrad_rows`)."""

    @staticmethod
    def forward(ctx, x, w, b):
        engine._require_cuda(x, "Linear input")
        N, K = w.shape
        Kp = _ceil_to(K, 128)
        pk, ctx.wt = _packs_of(w)
        xb = _bf16_rows(x, K, Kp)
        M = xb.shape[0]
        bc = None if b is None else _f32c(b)
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        ctx.xt = None
        lib, st = _lib.load(), _lib.current_stream(x.device)
        if ctx.needs_input_grad[1] and M % 128 == 0 and M <= 2048:
            # the weight gradient's B operand (x^T as fragments) packed in the shadow of this GEMM instead of by a launch in the backward
            ctx.xt = torch.empty(Kp * M * 2, dtype=torch.uint8, device=x.device)
            _lib.check(lib.syn_linear_and_pack(xb.data_ptr(), pk.data_ptr(), _lib.ptr(bc), M, N, Kp, y.data_ptr(), ctx.xt.data_ptr(), st), "syn_linear_and_pack")
        else:
            _lib.check(lib.syn_linear(xb.data_ptr(), pk.data_ptr(), _lib.ptr(bc), M, N, Kp, y.data_ptr(), st), "syn_linear")
        ctx.save_for_backward(xb, w)
        ctx.owners, ctx.has_bias, ctx.in_shape = (w, b), b is not None, x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xb, w = ctx.saved_tensors
        N, K = w.shape
        M, Kp = xb.shape
        want_dx, want_dw, want_db = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dy2 = dy.reshape(-1, N)
        lib, st = _lib.load(), _lib.current_stream(dy.device)
        if M <= 64:
            dyc = _f32c(dy2)
            dx = dw = db = None
            if want_dw or want_db:
                dw = _grad_out(ctx.owners[0], (N, K)) if Kp == K else torch.empty(N, Kp, dtype=torch.float32, device=dy.device)
                db = _grad_out(ctx.owners[1], (N,)) if want_db else None
                _lib.check(lib.syn_linear_wgrad_rows(dyc.data_ptr(), xb.data_ptr(), M, N, Kp, dw.data_ptr(), _lib.ptr(db), st), "syn_linear_wgrad_rows")
                if Kp != K:
                    dw = dw[:, :K].contiguous()
                if not want_dw:
                    dw = None
            if want_dx:
                dx = _gemm(dyc.to(torch.bfloat16), ctx.wt, Kp, N)[:, :K].reshape(ctx.in_shape)
            return dx, dw, db
        # the GEMM pair works on 128-row tiles: other row counts (a batch that is not a multiple of 4 clips) are zero-padded by PyTorch ops
        Mp = _ceil_to(M, 128)
        xt = ctx.xt
        if Mp != M:
            src = F.pad(_f32c(dy2), (0, 0, 0, Mp - M))
            xb, xt = F.pad(xb, (0, 0, 0, Mp - M)), None
            ld = N
        elif dy2.dtype is torch.float32 and dy2.stride(1) == 1 and dy2.stride(0) >= N and dy2.stride(0) % 4 == 0 and dy2.data_ptr() % 16 == 0:
            src, ld = dy2, dy2.stride(0)                     # read in place (contiguous, or a column slice of a wider gradient)
        else:
            src, ld = _f32c(dy2), N
        if xt is None:
            xt = _pack_t(xb, Kp, Mp)
        dx, dw, db = _lin_bwd(src.data_ptr(), (xt, ctx.wt), w, want_db, owners=ctx.owners, ld=ld, rows=Mp, want_dx=want_dx)
        if dx is not None:
            dx = dx[:M, :K].reshape(ctx.in_shape)
        return dx, (dw if want_dw else None), db


def lin(x, module: nn.Linear):
    return HipLinearFn.apply(x, module.weight, module.bias)


# ---- a pre-LN residual branch as ONE autograd node ---------------------------------------------------------------------------
# x + drop_path(attn(norm1(x))) and x + drop_path(mlp(norm2(x))) (timm_transformer/transformer.py:195-198):
#   forward   LayerNorm -> bf16 rows | GEMM (+ x^T pack; fc1: + GELU -> bf16) | attention -> bf16 | GEMM with `x + factor * (.)` in its epilogue
#   backward  prep (factor * dy -> bf16, bf16^T, bias partials) | GEMM pair (+ bias sum) | attention backward | GELU' | prep | GEMM pair |
#             LayerNorm backward with dy as its addend (the residual path)
# The fp32 LayerNorm / attention / GELU outputs are never written: the Linear behind each takes bf16 operands and nothing else reads them.
# Batches of more than 64 clips run the blocks as these nodes; up to 64 the persistent `StackFn` below.

def engine_has_xcd_groups(device) -> bool:
    """The tile-split kernels deal the workgroups of an XCD by hardware id: 256 CUs in 8 XCDs of 32 (MI355X)."""
    return torch.cuda.get_device_properties(device).multi_processor_count == 256


def _ln_rows_bf16(hc, g, b):
    rows = hc.numel() // 512
    zb = torch.empty(rows, 512, dtype=torch.bfloat16, device=hc.device)
    mean, rstd = torch.empty(rows, device=hc.device), torch.empty(rows, device=hc.device)
    _lib.check(_lib.load().syn_ln_fwd(hc.data_ptr(), g.data_ptr(), b.data_ptr(), None, zb.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows,
                                      _lib.current_stream(hc.device)), "syn_ln_fwd")
    return zb, mean, rstd


def _ln_bwd_rows(dz, hc, g, mean, rstd, add, owners=(None, None)):
    rows = hc.numel() // 512
    dh = torch.empty_like(hc)
    dg = _grad_out(owners[0], (512,)) if owners[0] is not None else torch.empty(512, device=hc.device)
    db = _grad_out(owners[1], (512,)) if owners[1] is not None else torch.empty(512, device=hc.device)
    scratch = torch.empty((rows + 15) // 16 * 1024, device=hc.device)
    _lib.check(_lib.load().syn_ln_bwd(dz.data_ptr(), hc.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), add.data_ptr(), dh.data_ptr(),
                                      dg.data_ptr(), db.data_ptr(), scratch.data_ptr(), rows, _lib.current_stream(hc.device)), "syn_ln_bwd")
    return dh, dg, db


class AttnBranchFn(torch.autograd.Function):
    """h + factor * proj(attention(qkv(LayerNorm(h)))) for h (B, 32, 512), B % 4 == 0; factor (B, 1, 1) or None."""

    @staticmethod
    def forward(ctx, h, g, b, wqkv, bqkv, wproj, bproj, factor):
        hc, gc, bc = _f32c(h), _f32c(g), _f32c(b)
        B, T, _ = hc.shape
        zb, mean, rstd = _ln_rows_bf16(hc, gc, bc)
        qkv, xt1 = _lin_fwd(zb, wqkv, bqkv)
        ob = torch.empty(B * T, 512, dtype=torch.bfloat16, device=hc.device)
        _lib.check(_lib.load().syn_attn_fwd(qkv.data_ptr(), None, ob.data_ptr(), B, _lib.current_stream(hc.device)), "syn_attn_fwd")
        out, xt2 = _lin_fwd(ob, wproj, bproj, hc, factor, T)
        ctx.save_for_backward(hc, gc, mean, rstd, qkv, wqkv, wproj, factor)
        ctx.packs = (xt1, xt2)
        ctx.bias = (bqkv is not None, bproj is not None)
        ctx.owners = (g, b, wqkv, bqkv, wproj, bproj)       # the parameter objects (their bound gradient buffers, `_grad_out`)
        return out.view(B, T, 512)

    @staticmethod
    def backward(ctx, dout):
        hc, gc, mean, rstd, qkv, wqkv, wproj, factor = ctx.saved_tensors
        B, T, _ = hc.shape
        xt1, xt2 = ctx.packs
        og, ob, owq, obq, owp, obp = ctx.owners
        d = _f32c(dout).view(B * T, 512)
        do, dwp, dbp = _lin_bwd(d, xt2, wproj, ctx.bias[1], factor, T, owners=(owp, obp))
        dqkv = torch.empty_like(qkv)
        _lib.check(_lib.load().syn_attn_bwd(qkv.data_ptr(), do.data_ptr(), dqkv.data_ptr(), B, _lib.current_stream(d.device)), "syn_attn_bwd")
        dz, dwq, dbq = _lin_bwd(dqkv.view(B * T, -1), xt1, wqkv, ctx.bias[0], owners=(owq, obq))
        dh, dg, db = _ln_bwd_rows(dz, hc, gc, mean, rstd, d, owners=(og, ob))
        return dh.view(B, T, 512), dg, db, dwq, dbq, dwp, dbp, None


class MlpBranchFn(torch.autograd.Function):
    """h + factor * fc2(GELU(fc1(LayerNorm(h)))), GELU in fc1's epilogue."""

    @staticmethod
    def forward(ctx, h, g, b, w1, b1, w2, b2, factor):
        hc, gc, bc = _f32c(h), _f32c(g), _f32c(b)
        B, T, _ = hc.shape
        zb, mean, rstd = _ln_rows_bf16(hc, gc, bc)
        ab = torch.empty(B * T, w1.shape[0], dtype=torch.bfloat16, device=hc.device)
        pre, xt1 = _lin_fwd(zb, w1, b1, gelu_out=ab)
        out, xt2 = _lin_fwd(ab, w2, b2, hc, factor, T)
        ctx.save_for_backward(hc, gc, mean, rstd, pre, w1, w2, factor)
        ctx.packs = (xt1, xt2)
        ctx.bias = (b1 is not None, b2 is not None)
        ctx.owners = (g, b, w1, b1, w2, b2)
        return out.view(B, T, 512)

    @staticmethod
    def backward(ctx, dout):
        hc, gc, mean, rstd, pre, w1, w2, factor = ctx.saved_tensors
        B, T, _ = hc.shape
        xt1, xt2 = ctx.packs
        og, ob, ow1, ob1, ow2, ob2 = ctx.owners
        d = _f32c(dout).view(B * T, 512)
        da, dw2, db2 = _lin_bwd(d, xt2, w2, ctx.bias[1], factor, T, owners=(ow2, ob2))
        dpre = torch.empty_like(pre)
        _lib.check(_lib.load().syn_gelu_bwd(pre.data_ptr(), da.data_ptr(), dpre.data_ptr(), pre.numel(), _lib.current_stream(d.device)), "syn_gelu_bwd")
        dz, dw1, db1 = _lin_bwd(dpre, xt1, w1, ctx.bias[0], owners=(ow1, ob1))
        dh, dg, db = _ln_bwd_rows(dz, hc, gc, mean, rstd, d, owners=(og, ob))
        return dh.view(B, T, 512), dg, db, dw1, db1, dw2, db2, None


def _blocks_ok(m) -> bool:
    """The reference's block geometry (timm_transformer/transformer.py:154-198 as models/denoiser.py:94-98 builds it): what the fused nodes implement."""
    for blk in m.mytimmblocks:
        if blk.attn.qkv.bias is not None or blk.attn.proj.bias is None or blk.mlp.fc1.bias is None or blk.mlp.fc2.bias is None:
            return False
        if (tuple(blk.attn.qkv.weight.shape), tuple(blk.attn.proj.weight.shape), tuple(blk.mlp.fc1.weight.shape), tuple(blk.mlp.fc2.weight.shape)) != \
                ((1536, 512), (512, 512), (1024, 512), (512, 1024)):
            return False
    return True


# ---- the eight blocks as two persistent launches (csrc/syn_stack_train.inc) -----------------------------------------------------------------------
# `syn_train_stack_fwd`: the sampling path's whole-step kernel in its tile-split mode, writing what the backward takes; `syn_train_stack_bwd`: the
# data-gradient chain as one launch; `syn_train_stack_wgrad`: the 32 weight-gradient GEMMs four per launch.  4 | batch <= 64 clips on a 256-CU device.
_stack_ws = {}


def _stack_workspace(device, n_seq):
    ws = _stack_ws.get(device)
    if ws is None or ws[1].shape[0] < n_seq:
        ws = _stack_ws[device] = (torch.zeros(320, dtype=torch.int32, device=device), torch.empty(max(n_seq, 64), 8, 32 * 512, dtype=torch.float32, device=device))
    return ws


def stack_sync_flag(device):
    """The persistent block kernels' sticky error flag (int32 device tensor of one element), or None if they never ran on `device`.  Their XCD-local
    barrier waits are bounded (`lat::group_wait`); a wait that ran out - CUs taken by another kernel - sets the flag and the kernel carries on with
    whatever partial sums it found.  The loss kernel turns every loss into NaN while the flag is set (`MaskedSmoothL1Fn`), `check_stack_sync` raises."""
    ws = _stack_ws.get(torch.device(device))
    return None if ws is None else ws[0][256:257]


def check_stack_sync(device=None):
    """Host check of the persistent block kernels' barrier flag (a device read: call it once per N steps or after a graph replay, not per launch).
    Raises SynHipError and clears the flag."""
    for dev, ws in list(_stack_ws.items()):
        if device is not None and torch.device(device) != dev:
            continue
        flag = int(ws[0][256].item())
        if flag:
            ws[0].zero_()
            raise _lib.SynHipError(f"persistent block kernels on {dev}: an XCD-local barrier wait ran out (flag {flag}) - activations / gradients of the "
                                   "steps since are wrong; is another kernel occupying CUs of this device?")


class StackFn(torch.autograd.Function):
    """mytimmblocks[0..7] on h (B, 32, 512), 4 | B <= 64; dp: DropPath factors (16, B, 1, 1) or None; params: per block norm1.weight, norm1.bias,
    attn.qkv.weight, attn.proj.weight, attn.proj.bias, norm2.weight, norm2.bias, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias."""

    NP = 11

    @staticmethod
    def forward(ctx, h, dp, *params):
        lib = _lib.load()
        hc = _f32c(h)
        B, T, _ = hc.shape
        M, dev = B * T, hc.device
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
        a = _lib.SynTrainStack()
        out = f32(B, T, 512)
        a.h_in, a.h_out, a.n_seq = hc.data_ptr(), out.data_ptr(), B
        dpc = None if dp is None else _f32c(dp).view(-1, B)
        a.drop_path = _lib.ptr(dpc)
        sync, xch = _stack_workspace(dev, B)
        a.sync, a.xch = sync.data_ptr(), xch.data_ptr()
        saves, packs, keep = [], [], [hc, dpc]
        NP = StackFn.NP
        for l in range(len(params) // NP):
            g1, b1, wq, wp, bp, g2, b2, w1, bb1, w2, bb2 = params[l * NP:(l + 1) * NP]
            L = a.layer[l]
            fr = [_packs_of(w) for w in (wq, wp, w1, w2)]
            vecs = [_f32c(v) for v in (g1, b1, bp, g2, b2, bb1, bb2)]
            keep += vecs
            L.ln1_g, L.ln1_b, L.b_proj, L.ln2_g, L.ln2_b, L.b_fc1, L.b_fc2 = (v.data_ptr() for v in vecs)
            L.w_qkv, L.w_proj, L.w_fc1, L.w_fc2 = (f[0].data_ptr() for f in fr)
            sv = dict(h_attn=f32(B, T, 512), mean_attn=f32(M), rstd_attn=f32(M), qkv=f32(B, T, 1536), xt_ln1=u8(512 * M * 2), xt_attn=u8(512 * M * 2),
                      h_mlp=f32(B, T, 512), mean_mlp=f32(M), rstd_mlp=f32(M), pre=f32(M, 1024), xt_ln2=u8(512 * M * 2), xt_gelu=u8(1024 * M * 2))
            for k, v in sv.items():
                setattr(a.save[l], k, v.data_ptr())
            saves.append(sv)
            packs.append(tuple(f[1] for f in fr))            # W^T fragment sets as of this forward (qkv, proj, fc1, fc2)
        _lib.check(lib.syn_train_stack_fwd(C.byref(a), _lib.current_stream(dev)), "syn_train_stack_fwd")
        ctx.saves, ctx.packs, ctx.params, ctx.keep, ctx.fwd = saves, packs, params, keep, a
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        params, NP = ctx.params, StackFn.NP
        d = _f32c(dout)
        B, T, _ = d.shape
        M, dev = B * T, d.device
        g = _lib.SynTrainStackGrad()
        g.fwd = C.pointer(ctx.fwd)
        dh_in = torch.empty(B, T, 512, dtype=torch.float32, device=dev)
        stash = torch.empty(B, 4, 32 * 512, dtype=torch.float32, device=dev)
        g.dh_out, g.dh_in, g.stash = d.data_ptr(), dh_in.data_ptr(), stash.data_ptr()
        bf = lambda n: torch.empty(n, M, dtype=torch.bfloat16, device=dev)
        keep, grads = [d, stash], [None] * len(params)
        n_blk = len(params) // NP
        for l in range(n_blk):
            g1, b1, wq, wp, bp, g2, b2, w1, bb1, w2, bb2 = params[l * NP:(l + 1) * NP]
            tq, tp_, t1, t2 = ctx.packs[l]
            L = g.layer_t[l]
            gains = (_f32c(g1), _f32c(g2))
            L.ln1_g, L.ln2_g = gains[0].data_ptr(), gains[1].data_ptr()
            L.w_qkv, L.w_proj, L.w_fc1, L.w_fc2 = tq.data_ptr(), tp_.data_ptr(), t1.data_ptr(), t2.data_ptr()
            G = g.grad[l]
            ten = dict(dyt_fc2=bf(512), dyt_fc1=bf(1024), dyt_proj=bf(512), dyt_qkv=bf(1536), part=torch.empty(B, 4096, dtype=torch.float32, device=dev),
                       dw_fc2=_grad_out(w2), dw_fc1=_grad_out(w1), dw_proj=_grad_out(wp), dw_qkv=_grad_out(wq),
                       d_ln2_g=_grad_out(g2), d_ln2_b=_grad_out(b2), d_fc2_b=_grad_out(bb2), d_fc1_b=_grad_out(bb1), d_ln1_g=_grad_out(g1), d_ln1_b=_grad_out(b1),
                       d_proj_b=_grad_out(bp))
            for k, v in ten.items():
                setattr(G, k, v.data_ptr())
            # (no second reference to a gradient tensor may survive this function: AccumulateGrad adopts a gradient only if it is the sole owner,
            # and clones it otherwise - 88 device copies per step)
            keep += [gains, [ten[k] for k in ("dyt_fc2", "dyt_fc1", "dyt_proj", "dyt_qkv", "part")]]
            grads[l * NP:(l + 1) * NP] = [ten["d_ln1_g"], ten["d_ln1_b"], ten["dw_qkv"], ten["dw_proj"], ten["d_proj_b"], ten["d_ln2_g"], ten["d_ln2_b"],
                                          ten["dw_fc1"], ten["d_fc1_b"], ten["dw_fc2"], ten["d_fc2_b"]]
        if _packs_blocks is not None:
            _packs_blocks.prefetch(True)                   # (the transposed sets were packed in front of the forward; 150 MB of saved tensors went by since)
        st = _lib.current_stream(dev)
        g.first_block, g.last_block = n_blk - 1, 0
        _lib.check(lib.syn_train_stack_bwd(C.byref(g), st), "syn_train_stack_bwd")
        _lib.check(lib.syn_train_stack_wgrad(C.byref(g), st), "syn_train_stack_wgrad")
        del ten, keep                             # (stream-ordered allocator: the launches above are enqueued, later work on this stream comes after them)
        return (dh_in, None, *grads)


def _stack_ok(m, bs, T, device) -> bool:
    return T == 32 and bs % 4 == 0 and 4 <= bs <= 64 and len(m.mytimmblocks) == 8 and engine_has_xcd_groups(device)


# ---- the loss ------------------------------------------------------------------------------------------------------------------------------------
class MaskedSmoothL1Fn(torch.autograd.Function):
    """`masked_l2` of the reference's training_losses (gaussian_diffusion.py:202-215: SmoothL1 x mask, summed per sample, / (sum(mask) x C)):
    (target, out (B, C, 1, T), mask (B, 1, 1, T) bool) -> (B,).  `out` is read where the output Linear left it - as (B, C, 1, T), or as the
    [B][T][C] rows that `train_forward`'s result is a permuted view of - and the gradient, with the incoming per-sample factors folded in, is
    written the same way by one pass in the backward: neither direction materialises a permuted copy."""

    @staticmethod
    def forward(ctx, target, out, mask):
        B, Cc, _, T = out.shape
        rows = _is_rows_view(out)
        oc = out.detach() if (rows or out.is_contiguous()) else out.detach().contiguous()
        tc = _f32c(target)
        mk = mask.detach().reshape(B, T).contiguous().view(torch.uint8)
        loss = torch.empty(B, dtype=torch.float32, device=out.device)
        part = torch.empty(B, Cc // 64, dtype=torch.float32, device=out.device)
        flag = stack_sync_flag(out.device)                    # a block-kernel barrier that gave up poisons the loss (sticky until check_stack_sync)
        _lib.check(_lib.load().syn_masked_smooth_l1(tc.data_ptr(), oc.data_ptr(), mk.data_ptr(), B, Cc, T, int(rows), part.data_ptr(), _lib.ptr(flag),
                                                    loss.data_ptr(), _lib.current_stream(out.device)), "syn_masked_smooth_l1")
        ctx.save_for_backward(tc, oc, mk)
        ctx.rows = rows
        return loss

    @staticmethod
    def backward(ctx, g):
        tc, oc, mk = ctx.saved_tensors
        B, Cc, _, T = oc.shape
        gc = _f32c(g)
        if ctx.rows:
            buf = torch.empty(B, T, Cc, dtype=torch.float32, device=oc.device)
            dout = buf.permute(0, 2, 1).unsqueeze(2)
        else:
            buf = dout = torch.empty_like(oc)
        _lib.check(_lib.load().syn_masked_smooth_l1_grad(tc.data_ptr(), oc.data_ptr(), mk.data_ptr(), B, Cc, T, int(ctx.rows), gc.data_ptr(), buf.data_ptr(),
                                                         _lib.current_stream(oc.device)), "syn_masked_smooth_l1_grad")
        return None, dout, None


def _is_rows_view(out) -> bool:
    """`out` (B, C, 1, T) is a permuted view of contiguous [B][T][C] rows (what `train_forward` returns)."""
    B, Cc, _, T = out.shape
    return out.stride(1) == 1 and out.stride(3) == Cc and (B == 1 or out.stride(0) == T * Cc) and Cc > 1


def masked_smooth_l1(target, out, mask):
    """The fused loss when it applies (device fp32 tensors, one mask row per sample, no gradient asked for the target), else None."""
    if not (out.is_cuda and out.dim() == 4 and out.shape[2] == 1 and out.dtype is torch.float32 and target.shape == out.shape and not target.requires_grad
            and target.is_cuda and target.device == out.device
            and torch.is_tensor(mask) and mask.is_cuda and mask.device == out.device and mask.dtype is torch.bool and tuple(mask.shape) == (out.shape[0], 1, 1, out.shape[-1])
            and out.shape[-1] <= 64 and out.shape[1] % 64 == 0):
        return None
    return MaskedSmoothL1Fn.apply(target, out, mask)


# ---- rotary ----------------------------------------------------------------------------------------------------------------------------------------
def _rotary_tables(m, T, device):
    """(T, 32) tables cos / sin(position x inv_freq), fp32 like the reference's buffer (models/denoiser.py:324-343)."""
    inv = m.rel_pos.inv_freq
    key = (inv.data_ptr(), inv._version, inv.device, T)
    tab = m.__dict__.get("_syn_rotary_tables")
    if tab is None or tab[0] != key:
        with torch.no_grad():
            fr = torch.einsum("i,j->ij", torch.arange(T, device=device).type_as(inv), inv)
            tab = m.__dict__["_syn_rotary_tables"] = (key, fr.cos().contiguous(), fr.sin().contiguous())
    return tab[1], tab[2]


def _rotary_launch(x, cs, sn, inverse):
    y = torch.empty_like(x)
    _lib.check(_lib.load().syn_rotary(x.data_ptr(), cs.data_ptr(), sn.data_ptr(), x.numel() // (32 * 512), int(inverse), y.data_ptr(), _lib.current_stream(x.device)),
               "syn_rotary")
    return y


# ---- the input stage as ONE autograd node ------------------------------------------------------------------------------------------------------------
# models/denoiser.py:151-186 (denoiser_h3d.py:180-210) behind the audio encoder, rows in (clip, frame) order:
#   w_feat = text_encoder_body(text_pre_encoder_body(word))                  embedding rows written as the Linear's bf16 operand (300 -> 384 columns)
#   at     = avg_pool(mix_audio_text(cat[a_feat, w_feat]))                   the pool commutes with the Linear: the operand is cat[pool(a_feat), pool(w_feat)],
#                                                                            a quarter of the rows through the GEMM and its backward
#   x_     = poseEmbedding(x permuted to rows)                               transpose + bf16 rounding in one pass
#   seq    = input_process2(cat[emb_seed + emb_t (per clip), x_, at])        one concat kernel writes the operand
#   [seq   = input_process3(cat[seq, style (per clip)])]
#   h      = rotary(seq)
# Backward: every Linear is prep -> GEMM pair; the pieces of a cat's gradient are read in place as column slices of the data gradient (prep's ld /
# row_div / scale), per-clip vectors get their rows summed, the audio features' gradient is the pool's backward of its slice.
# 9 launches forward, ~20 backward, where the op-by-op composition took ~30 / ~70 (PyTorch casts, pads, cats, slice copies, reductions).
def _concat_bf16(srcs, M, out_ld, device):
    """srcs: (tensor, width, ld, row_div, pool, addend | None) -> bf16 [M][out_ld]."""
    arr = (_lib.SynConcatSrc * len(srcs))()
    for i, (t, width, ld, row_div, pool, t2) in enumerate(srcs):
        arr[i].p, arr[i].p2, arr[i].width, arr[i].ld, arr[i].row_div, arr[i].pool = t.data_ptr(), _lib.ptr(t2), width, ld, row_div, pool
    out = torch.empty(M, out_ld, dtype=torch.bfloat16, device=device)
    _lib.check(_lib.load().syn_rows_concat_bf16(arr, len(srcs), M, out_ld, out.data_ptr(), _lib.current_stream(device)), "syn_rows_concat_bf16")
    return out


class InputStageFn(torch.autograd.Function):
    """(a_rows (B, F, A) audio features, x (B, C, 1, T), emb_seed (B, D), emb_t (B, D), style (B, S) | None, word ids (B, F), rotary cos, sin,
    embedding table, text_encoder_body w / b, mix_audio_text w / b, poseEmbedding w / b, input_process2 w / b, input_process3 w / b | None)
    -> h (B, T, D) after the rotary embedding.  F = pool x T; 4 | B."""

    @staticmethod
    def forward(ctx, a_rows, x, emb_seed, emb_t, style, ids, cs, sn, table, wt, bt, wm, bm, wp, bp, w2, b2, w3, b3):
        lib = _lib.load()
        dev = x.device
        st = _lib.current_stream(dev)
        B, Fr, A = a_rows.shape
        Cx, T = x.shape[1], x.shape[-1]
        pool = Fr // T
        M, MF = B * T, B * Fr
        D = w2.shape[0]
        ac, xc, es, et = _f32c(a_rows), _f32c(x), _f32c(emb_seed), _f32c(emb_t)
        idc = ids.detach().reshape(-1).to(torch.int64).contiguous()
        V, DW = table.shape
        DWp = _ceil_to(DW, 128)
        E = torch.empty(MF, DWp, dtype=torch.bfloat16, device=dev)
        _lib.check(lib.syn_embed_rows_bf16(idc.data_ptr(), table.detach().data_ptr(), V, DW, MF, DWp, E.data_ptr(), st), "syn_embed_rows_bf16")
        w_feat, pk_t = _lin_fwd(E, wt, bt)                                                   # (B F, word_f)
        WF = w_feat.shape[1]
        P = _concat_bf16([(ac, A, A, 1, pool, None), (w_feat, WF, WF, 1, pool, None)], M, _ceil_to(A + WF, 128), dev)
        at, pk_m = _lin_fwd(P, wm, bm)                                                        # (B T, audio_f) - pooled
        AT = at.shape[1]
        xt = torch.empty(M, Cx, dtype=torch.bfloat16, device=dev)
        _lib.check(lib.syn_bct_to_rows_bf16(xc.data_ptr(), B, Cx, T, xt.data_ptr(), st), "syn_bct_to_rows_bf16")
        x_, pk_p = _lin_fwd(xt, wp, bp)                                                       # (B T, D)
        I = _concat_bf16([(es, D, D, T, 1, et), (x_, D, D, 1, 1, None), (at, AT, AT, 1, 1, None)], M, _ceil_to(2 * D + AT, 128), dev)
        seq, pk_2 = _lin_fwd(I, w2, b2)
        pk_3, S = None, 0
        if w3 is not None:
            sc = _f32c(style)
            S = sc.shape[1]
            I3 = _concat_bf16([(seq, D, D, 1, 1, None), (sc, S, S, T, 1, None)], M, _ceil_to(D + S, 128), dev)
            seq, pk_3 = _lin_fwd(I3, w3, b3)
        h = _rotary_launch(seq, cs, sn, False)
        ctx.save_for_backward(idc, cs, sn, wt, wm, wp, w2, w3)
        ctx.packs = (pk_t, pk_m, pk_p, pk_2, pk_3)
        ctx.dims = (B, Fr, A, T, pool, D, AT, WF, S, V, DW, DWp)
        ctx.owners = (table, wt, bt, wm, bm, wp, bp, w2, b2, w3, b3)
        ctx.style_shape = None if style is None else style.shape
        return h.view(B, T, D)

    @staticmethod
    def backward(ctx, dh):
        lib = _lib.load()
        idc, cs, sn, wt, wm, wp, w2, w3 = ctx.saved_tensors
        pk_t, pk_m, pk_p, pk_2, pk_3 = ctx.packs
        B, Fr, A, T, pool, D, AT, WF, S, V, DW, DWp = ctx.dims
        table, owt, obt, owm, obm, owp, obp, ow2, ob2, ow3, ob3 = ctx.owners
        dev = dh.device
        st = _lib.current_stream(dev)
        M, MF = B * T, B * Fr
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        d = _rotary_launch(_f32c(dh).view(M, D), cs, sn, True)                                # gradient at the last Linear's output
        src, ld = d.data_ptr(), D
        d_style = dw3 = db3 = None
        keep = [d]
        if w3 is not None:
            dI3, dw3, db3 = _lin_bwd(src, pk_3, w3, ob3 is not None, owners=(ow3, ob3), ld=ld, rows=M)
            K3 = dI3.shape[1]
            d_style = f32(B, S)
            _lib.check(lib.syn_rows_group_sum(dI3.data_ptr() + 4 * D, K3, S, T, B, d_style.data_ptr(), st), "syn_rows_group_sum")
            d_style = d_style.view(ctx.style_shape)
            src, ld = dI3.data_ptr(), K3
            keep.append(dI3)
        dI, dw2, db2 = _lin_bwd(src, pk_2, w2, ob2 is not None, owners=(ow2, ob2), ld=ld, rows=M)   # [M][2 D + AT (+ padding)]
        K2 = dI.shape[1]
        d_emb = f32(B, D)
        _lib.check(lib.syn_rows_group_sum(dI.data_ptr(), K2, D, T, B, d_emb.data_ptr(), st), "syn_rows_group_sum")
        # poseEmbedding: the latent takes no gradient - weight and bias gradient only
        _, dwp, dbp = _lin_bwd(dI.data_ptr() + 4 * D, pk_p, wp, obp is not None, owners=(owp, obp), ld=K2, rows=M, want_dx=False)
        # mix_audio_text on the pooled rows
        dP, dwm, dbm = _lin_bwd(dI.data_ptr() + 4 * 2 * D, pk_m, wm, obm is not None, owners=(owm, obm), ld=K2, rows=M)
        KP = dP.shape[1]
        d_a = f32(B, Fr, A)
        _lib.check(lib.syn_rows_expand(dP.data_ptr(), KP, A, pool, 1.0 / pool, MF, d_a.data_ptr(), st), "syn_rows_expand")
        # text_encoder_body: its output gradient = the pool's backward of dP's second slice, formed by the prep pass
        dE, dwt, dbt = _lin_bwd(dP.data_ptr() + 4 * A, pk_t, wt, obt is not None, owners=(owt, obt), ld=KP, row_div=pool, cscale=1.0 / pool, rows=MF)
        d_table = None
        if ctx.needs_input_grad[8]:
            d_table = _grad_out(table, (V, DW))
            first = True
            for lo in range(0, MF, 8192):                       # (the bench's 32 clips x 128 frames are one call)
                n = min(8192, MF - lo)
                tgt = d_table if first else torch.empty_like(d_table)
                _lib.check(lib.syn_embedding_wgrad(idc[lo:].data_ptr(), dE[lo:].data_ptr(), DWp, n, V, DW, tgt.data_ptr(), st), "syn_embedding_wgrad")
                if not first:
                    d_table += tgt
                first = False
            del tgt                                             # (a second reference would make AccumulateGrad clone the 13 MB gradient instead of adopting it)
        del keep
        return (d_a, None, d_emb, d_emb.clone() if ctx.needs_input_grad[3] and ctx.needs_input_grad[2] else d_emb, d_style, None, None, None, d_table,
                dwt, dbt, dwm, dbm, dwp, dbp, dw2, db2, dw3, db3)


class ConvPacks:
    """The hi / lo bf16 fragment sets of the audio encoder's Conv1d(k = 15) layers - the forward form of every layer and the
    data-gradient form of every layer but the first of the chain - packed in ONE launch per training forward
    (`syn_conv1d_pack_split_many`) instead of one launch per use (28 per step): the weights only change in optimizer.step().
    `ConvSplitFn` looks a weight up by storage address and in-place version; a miss takes the per-call pack."""

    def __init__(self, convs):
        lib = _lib.load()
        self.items, reqs, self.keep = {}, [], []
        for conv in convs:
            w = conv.weight
            cout, cin, stride = conv.out_channels, conv.in_channels, conv.stride[0]
            if not (w.is_cuda and w.dtype is torch.float32 and w.is_contiguous() and (cin, stride, cout) in ConvSplitFn.SUPPORTED):
                continue
            ent = {"ref": __import__("weakref").ref(w), "ptr": w.data_ptr(), "version": -1}
            for transposed in (0, 1):
                nb = lib.syn_conv1d_pack_bytes(cout, cin, stride, transposed)
                hi, lo = torch.empty(nb, dtype=torch.uint8, device=w.device), torch.empty(nb, dtype=torch.uint8, device=w.device)
                ent[transposed] = (hi, lo)
                reqs.append((w.data_ptr(), hi.data_ptr(), lo.data_ptr(), cout, cin, stride, transposed))
            self.items[w.data_ptr()] = ent
        self.lists = []
        for lo in range(0, len(reqs), _lib.SYN_CONV_PACK_MAX):
            chunk = reqs[lo:lo + _lib.SYN_CONV_PACK_MAX]
            arr = (_lib.SynConvPackReq * len(chunk))(*[_lib.SynConvPackReq(*r) for r in chunk])
            self.lists.append(arr)
        self.device = next((e["ref"]().device for e in self.items.values()), None)

    def valid(self) -> bool:
        return all(e["ref"]() is not None and e["ref"]().data_ptr() == e["ptr"] for e in self.items.values())

    def refresh(self):
        for arr in self.lists:
            _lib.check(_lib.load().syn_conv1d_pack_split_many(C.cast(arr, C.c_void_p), len(arr), _lib.current_stream(self.device)),
                       "syn_conv1d_pack_split_many")
        for e in self.items.values():
            e["version"] = e["ref"]()._version

    def lookup(self, w, transposed):
        e = self.items.get(w.data_ptr())
        if e is None or e["ref"]() is None or e["version"] != w._version or e["ptr"] != e["ref"]().data_ptr():
            return None
        return e[int(bool(transposed))]


_conv_packs: "ConvPacks | None" = None


def _lookup_conv_pack(w, transposed):
    return _conv_packs.lookup(w, transposed) if _conv_packs is not None else None


def _unsupported_conv(what, cin, stride, pad, cout):
    return _lib.SynHipError(f"{what}: no hand-written kernel covers Conv1d({cin} -> {cout}, k 15, stride {stride}, padding {pad}) of the audio "
                            "encoder (covered: the reference's WavEncoder, models/denoiser.py:304-322); there is no library fallback")


# Which cross products of the hi / lo operand split each convolution role issues (`syn_debug_conv_terms`; bit 0: A_lo . B_hi, bit 1: A_hi . B_lo,
# 3 = both = fp32-grade).  "forward,data-gradient,weight-gradient"; A / B = (W, x), (W^T, dy), (dy, x).  An A/B switch.
def _parse_conv_terms(text: str) -> tuple:
    parts = text.split(",")
    if len(parts) != 3 or not all(p.strip() in ("0", "1", "2", "3") for p in parts):
        raise ValueError(f"SYN_CONV_TERMS must be three masks 0..3 'forward,data-gradient,weight-gradient' (default 3,3,1), got {text!r}")
    return tuple(int(p) for p in parts)


CONV_TERMS = _parse_conv_terms(_os.environ.get("SYN_CONV_TERMS", "3,3,1"))


def _conv_terms(role: int):
    """Diagnostics only: with the default (3, 3, 1 = the library's own: both cross products forward and in the data gradient, one in the weight
    gradient) the library's switch is never touched - the product path makes no `syn_debug_*` call.  An A/B run (SYN_CONV_TERMS set to something
    else) selects the role's mask in front of a convolution launch; `_conv_terms_done` puts the library's own choice back behind it, so the
    process-wide switch never outlives the launch it was set for."""
    if CONV_TERMS != (3, 3, 1):
        _lib.load().syn_debug_conv_terms(CONV_TERMS[role])


def _conv_terms_done():
    if CONV_TERMS != (3, 3, 1):
        _lib.load().syn_debug_conv_terms(-1)


def _conv_pack_of(w, cout, cin, stride, transposed):
    """hi / lo fragment sets of a Conv1d(k 15) weight (cout, cin, [1,] 15): the step's pack, or packed on the spot."""
    pk = _lookup_conv_pack(w, transposed)
    if pk is not None:
        return pk
    lib = _lib.load()
    wc = w.detach().float().contiguous()
    nb = lib.syn_conv1d_pack_bytes(cout, cin, stride, int(bool(transposed)))
    whi = torch.empty(nb, dtype=torch.uint8, device=w.device)
    wlo = torch.empty_like(whi)
    _lib.check(lib.syn_conv1d_pack_split(wc.data_ptr(), cout, cin, stride, int(bool(transposed)), whi.data_ptr(), wlo.data_ptr(), _lib.current_stream(w.device)),
               "syn_conv1d_pack_split")
    return whi, wlo


class ConvSplitFn(torch.autograd.Function):
    """(N, C, 1, L) channels_last convolution of the audio encoder, forward on the hand-written implicit-GEMM kernel with
    operands split into bf16 hi + lo halves (`syn_conv1d_train_fwd`: three MFMAs per product, fp32-grade - the plain bf16
    forward moves the gradients of the first blocks by 14 %); data and weight gradients on the same family of kernels.
    Covers the encoder's Conv1d(k = 15) layers from block 0's conv2 on (block 0's conv1 / shortcut have 1-2 input channels and
    1 % of the FLOPs)."""

    SUPPORTED = {(64, 1, 64), (128, 1, 128), (256, 1, 256), (64, 6, 64), (64, 6, 128), (128, 3, 256)}

    @staticmethod
    def run(x, w, stride, pad, transposed=False, want_stats=False):
        """y = conv(x, w) for x (N, Cin, 1, L) channels_last fp32 and the module's weight w (Cout, Cin, 1, 15);
        transposed: y = the data gradient of that (stride-1, padding-7) convolution for x = dy (N, Cout, 1, L)."""
        lib = _lib.load()
        n, cx, _, l_in = x.shape
        wc = w.detach().float().contiguous()
        co_w, ci_w = wc.shape[0], wc.shape[1]
        cin, cout = (co_w, ci_w) if transposed else (ci_w, co_w)
        assert cx == cin, (x.shape, w.shape, transposed)
        xc = x.contiguous(memory_format=torch.channels_last)                # physically [n][l][cin]
        whi, wlo = _conv_pack_of(w, co_w, ci_w, stride, transposed)
        l_out = (l_in + 2 * pad - 15) // stride + 1
        y = torch.empty(n, cout, 1, l_out, device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        part = None
        if want_stats:
            # BatchNorm's per-channel sums from the convolution's accumulators: the BatchNorm that follows skips its pass over y
            tiles = lib.syn_conv1d_train_fwd_tiles(n, l_in, cin, stride, pad, cout)
            part = torch.empty(tiles, 2, cout, device=x.device, dtype=torch.float32)
        _conv_terms(1 if transposed else 0)
        _lib.check(lib.syn_conv1d_train_fwd(xc.data_ptr(), n, l_in, cin, stride, pad, whi.data_ptr(), wlo.data_ptr(), None, cout,
                                                    y.data_ptr(), _lib.ptr(part), _lib.current_stream(x.device)), "syn_conv1d_train_fwd")
        _conv_terms_done()
        if part is not None:
            y._syn_bn_part = part                                            # picked up by BnActFn.forward (same tensor object)
        return xc, y

    @staticmethod
    def forward(ctx, x, w, stride, pad):
        xc, y = ConvSplitFn.run(x, w, stride, pad, want_stats=True)
        ctx.save_for_backward(xc, w)
        ctx.geom = (stride, pad)
        ctx.owner = getattr(w, "_syn_owner", None)        # the Conv1d's (Cout, Cin, 15) parameter behind the 4-d view (`_conv_raw`)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, pad = ctx.geom
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = gw = None
        cout, cin = w.shape[0], w.shape[1]
        if ctx.needs_input_grad[0]:
            if stride == 1 and pad == 7 and (cout, 1, cin) in ConvSplitFn.SUPPORTED:
                # the data gradient of a stride-1 'same' convolution is the same convolution with the taps reversed and the
                # channel roles swapped: the same kernel, fp32-grade like the forward
                gx = ConvSplitFn.run(gy, w, 1, 7, transposed=True)[1]
            elif pad == 0 and (cout, stride) in ((64, 6), (128, 6), (256, 3)) and (stride * cin) % 128 == 0:
                # a strided convolution's data gradient = a stride-1 convolution over dy whose output rows are `stride` consecutive
                # positions x cin channels (the forward kernel, all stride x cin columns in one launch)
                lib = _lib.load()
                n, _, _, l_in = x.shape
                whi, wlo = _conv_pack_of(w, cout, cin, stride, True)
                gx = torch.empty(n, cin, 1, l_in, device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
                _conv_terms(1)
                _lib.check(lib.syn_conv1d_train_dgrad_sum(gy.data_ptr(), whi.data_ptr(), wlo.data_ptr(), None, None, None, None, n, l_in, cin, stride, 0, cout,
                                                          gx.data_ptr(), _lib.current_stream(x.device)), "syn_conv1d_train_dgrad_sum")
                _conv_terms_done()
            else:
                raise _unsupported_conv("data gradient", cin, stride, pad, cout)
        if ctx.needs_input_grad[1]:
            if (cin, stride, cout) in ConvSplitFn.SUPPORTED and ((stride == 1 and pad == 7) or (stride > 1 and pad == 0)):
                # contraction over positions of two channels-last tensors: transposed through LDS inside the kernel (a strided layer
                # as a stride-1 one over rows of stride x cin channels, a wave per 16 of them)
                lib = _lib.load()
                n, _, _, l = x.shape
                l_out, kts = gy.shape[-1], -(-15 // stride) * stride
                ws = torch.empty(lib.syn_conv1d_wgrad_shares(n, l_out, stride * cin, cout) * cout * kts * cin, device=x.device, dtype=torch.float32)
                gw = _grad_out(ctx.owner, (cout, cin, 1, 15)) if ctx.owner is not None else torch.empty(cout, cin, 1, 15, device=x.device, dtype=torch.float32)
                _conv_terms(2)
                _lib.check(lib.syn_conv1d_train_wgrad(x.data_ptr(), gy.data_ptr(), n, l, cin, stride, pad, cout, ws.data_ptr(), gw.data_ptr(),
                                                      _lib.current_stream(x.device)), "syn_conv1d_train_wgrad")
                _conv_terms_done()
                gw = gw.to(w.dtype)
            else:
                raise _unsupported_conv("weight gradient", cin, stride, pad, cout)
        return gx, gw, None, None


class ConvFirstFn(torch.autograd.Function):
    """The encoder's first layer, Conv1d(1 | 2 -> 64, k 15, stride 5, padding 1700) of block 0's conv1 and of its shortcut
    (models/denoiser.py:308), on the waveform as the reference passes it, (N, L, cin) fp32: forward and weight gradient on plain
    fp32 FMAs (`syn_conv1d_first_fwd` / `_wgrad`); the waveform takes no gradient.  Returns (N, 64, 1, L_out) channels_last."""

    @staticmethod
    def forward(ctx, wav, w, stride, pad):
        lib = _lib.load()
        n, l_in, cin = wav.shape
        wavc, wc = wav.detach().float().contiguous(), w.detach().float().contiguous()
        l_out = (l_in + 2 * pad - 15) // stride + 1
        y = torch.empty(n, 64, 1, l_out, device=wav.device, dtype=torch.float32, memory_format=torch.channels_last)
        _lib.check(lib.syn_conv1d_first_fwd(wavc.data_ptr(), n, l_in, cin, stride, pad, wc.data_ptr(), y.data_ptr(),
                                            _lib.current_stream(wav.device)), "syn_conv1d_first_fwd")
        ctx.save_for_backward(wavc)
        ctx.geom = (stride, pad, w.shape, w.dtype)
        ctx.owner = w
        return y

    @staticmethod
    def backward(ctx, gy):
        (wavc,) = ctx.saved_tensors
        stride, pad, wshape, wdtype = ctx.geom
        if not ctx.needs_input_grad[1]:
            return None, None, None, None
        lib = _lib.load()
        n, l_in, cin = wavc.shape
        gy = gy.contiguous(memory_format=torch.channels_last)
        ws = torch.empty(lib.syn_conv1d_first_parts(n, gy.shape[-1]) * 64 * cin * 15, device=gy.device, dtype=torch.float32)
        gw = _grad_out(ctx.owner, (64, cin, 15))
        _lib.check(lib.syn_conv1d_first_wgrad(wavc.data_ptr(), gy.data_ptr(), n, l_in, cin, stride, pad, ws.data_ptr(), gw.data_ptr(),
                                              _lib.current_stream(gy.device)), "syn_conv1d_first_wgrad")
        return None, gw.reshape(wshape).to(wdtype), None, None


# (Every convolution of the encoder runs on the hand-written kernels: split-operand MFMA forward / data gradient / weight
# gradient for the Conv1d(k = 15) layers from block 0's conv2 on, plain fp32 FMAs for the 1-2-channel first layer.  A geometry
# none of them covers raises instead of dropping to a library convolution.  Measured and rejected: bf16 operands for these
# convolutions - the gradients of the first encoder blocks move by up to 14 %)


class BnActFn(torch.autograd.Function):
    """act(BatchNorm1d(y) [+ shortcut]) on batch statistics for channels_last (N, C, 1, L) fp32 tensors - the tail of every
    convolution of the encoder's BasicBlock (models/utils/layer.py:171-184) in training mode, as three launches forward and three
    backward (`syn_bn_act_fwd` / `_bwd`) instead of BatchNorm + add + LeakyReLU passes.  ``conv_bias``: the bias of the convolution
    that produced y, which the caller has NOT added (it only shifts the running mean; its gradient is exactly zero)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, conv_bias, shortcut, run_mean, run_var, momentum, eps, act):
        lib = _lib.load()
        n, c, _, l = y.shape
        yc = y.contiguous(memory_format=torch.channels_last)
        sc = None if shortcut is None else shortcut.contiguous(memory_format=torch.channels_last)
        rows = n * l
        part = getattr(y, "_syn_bn_part", None)                  # the producing convolution's per-tile sums (ConvSplitFn.run)
        if part is not None and part.shape[2] == c and part.device == y.device:
            ws, ws_chunks = part, part.shape[0]
        else:
            ws, ws_chunks = torch.empty(2 * lib.syn_bn_chunks(rows) * c, device=y.device, dtype=torch.float32), 0
        stats = torch.empty(2, c, device=y.device, dtype=torch.float32)
        z = torch.empty_like(yc, memory_format=torch.channels_last)
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        cb = None if conv_bias is None else conv_bias.detach().float().contiguous()
        _lib.check(lib.syn_bn_act_fwd(yc.data_ptr(), _lib.ptr(sc), rows, c, g.data_ptr(), b.data_ptr(), float(eps), float(momentum),
                                      _lib.ptr(run_mean), _lib.ptr(run_var), _lib.ptr(cb), int(act), ws.data_ptr(), ws_chunks, stats.data_ptr(),
                                      z.data_ptr(), _lib.current_stream(y.device)), "syn_bn_act_fwd")
        # (without a shortcut the backward recomputes the activation's sign from y and does not read z)
        ctx.save_for_backward(yc, z if (act and shortcut is not None) else None, stats, g, b)
        ctx.act, ctx.has_short, ctx.has_cb = bool(act), shortcut is not None, conv_bias is not None
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        yc, z, stats, g, b = ctx.saved_tensors
        n, c, _, l = yc.shape
        rows = n * l
        dzc = dz.contiguous(memory_format=torch.channels_last)
        ws = torch.empty(2 * lib.syn_bn_chunks(rows) * c, device=dz.device, dtype=torch.float32)
        dgb = torch.empty(3, c, device=dz.device, dtype=torch.float32)          # [dgamma | dbeta | 0 = the conv bias's gradient]
        dy = torch.empty_like(yc, memory_format=torch.channels_last)
        dsh = torch.empty_like(yc, memory_format=torch.channels_last) if ctx.has_short else None
        _lib.check(lib.syn_bn_act_bwd(dzc.data_ptr(), _lib.ptr(z), yc.data_ptr(), stats.data_ptr(), g.data_ptr(), b.data_ptr(), rows, c, int(ctx.act),
                                      ws.data_ptr(), dgb.data_ptr(), dy.data_ptr(), _lib.ptr(dsh), _lib.current_stream(dz.device)),
                   "syn_bn_act_bwd")
        dcb = dgb[2] if ctx.has_cb else None
        return dy, dgb[0], dgb[1], dcb, dsh, None, None, None, None, None


SYNC_BN_RAGGED = False      # True: ranks may bring different numbers of rows to a SyncBatchNorm (costs a host synchronisation per BatchNorm)


def _all_reduce_sum(t, group, tag=None):
    """SUM all-reduce of a small device tensor over `group` (RCCL).  A module-level function so that single-GPU tests can stand in for
    the other ranks (tests/test_gpu_kernels.py); `tag` = (the BatchNorm's `_syn_test_key`, "fwd" | "bwd") identifies the call for them."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return t


class SyncBnActFn(torch.autograd.Function):
    """`BnActFn` with the batch statistics reduced over the ranks of a process group = nn.SyncBatchNorm, which the reference's DDP branch
    converts every BatchNorm of the model to (train.py:90).  Same kernels; the per-channel sums (fp64) make a round trip through one
    small all-reduce in each direction: forward [sum y, sum y^2, rows], backward [sum d, sum d xhat] (torch/nn/modules/_functions.py:
    weight / bias gradients stay local - DDP averages them -, the data gradient and the running statistics use the global numbers)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, conv_bias, shortcut, run_mean, run_var, momentum, eps, act, group, key):
        lib = _lib.load()
        n, c, _, l = y.shape
        yc = y.contiguous(memory_format=torch.channels_last)
        sc = None if shortcut is None else shortcut.contiguous(memory_format=torch.channels_last)
        rows = n * l
        part = getattr(y, "_syn_bn_part", None)
        if part is not None and part.shape[2] == c and part.device == y.device:
            ws, ws_chunks = part, part.shape[0]
        else:
            ws, ws_chunks = torch.empty(2 * lib.syn_bn_chunks(rows) * c, device=y.device, dtype=torch.float32), 0
        sums = torch.empty(2 * c + 1, device=y.device, dtype=torch.float64)          # [sum y | sum y^2 | rows]
        sums[2 * c] = rows
        _lib.check(lib.syn_bn_sums(yc.data_ptr(), rows, c, ws.data_ptr(), ws_chunks, sums.data_ptr(), _lib.current_stream(y.device)), "syn_bn_sums")
        total = _all_reduce_sum(sums.clone(), group, (key, "fwd"))
        if SYNC_BN_RAGGED or key is not None:
            rows_total = int(round(float(total[2 * c])))    # ranks with different numbers of rows: the all-reduced count (a host read)
        else:                                               # DDP's case, equal batches per rank: no host read, so the step stays graph-capturable
            import torch.distributed as dist
            rows_total = rows * (dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1)
        stats = torch.empty(2, c, device=y.device, dtype=torch.float32)
        z = torch.empty_like(yc, memory_format=torch.channels_last)
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        cb = None if conv_bias is None else conv_bias.detach().float().contiguous()
        _lib.check(lib.syn_bn_act_apply(yc.data_ptr(), _lib.ptr(sc), rows, rows_total, c, g.data_ptr(), b.data_ptr(), float(eps), float(momentum),
                                        _lib.ptr(run_mean), _lib.ptr(run_var), _lib.ptr(cb), int(act), total.data_ptr(), stats.data_ptr(), z.data_ptr(),
                                        _lib.current_stream(y.device)), "syn_bn_act_apply")
        ctx.save_for_backward(yc, z if (act and shortcut is not None) else None, stats, g, b)
        ctx.act, ctx.has_short, ctx.has_cb, ctx.group, ctx.rows_total, ctx.key = bool(act), shortcut is not None, conv_bias is not None, group, rows_total, key
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        yc, z, stats, g, b = ctx.saved_tensors
        n, c, _, l = yc.shape
        rows = n * l
        dzc = dz.contiguous(memory_format=torch.channels_last)
        ws = torch.empty(2 * lib.syn_bn_chunks(rows) * c, device=dz.device, dtype=torch.float32)
        local = torch.empty(2 * c, device=dz.device, dtype=torch.float64)
        _lib.check(lib.syn_bn_bwd_sums(dzc.data_ptr(), _lib.ptr(z), yc.data_ptr(), stats.data_ptr(), g.data_ptr(), b.data_ptr(), rows, c, int(ctx.act),
                                       ws.data_ptr(), local.data_ptr(), _lib.current_stream(dz.device)), "syn_bn_bwd_sums")
        total = _all_reduce_sum(local.clone(), ctx.group, (ctx.key, "bwd"))
        dgb = torch.empty(3, c, device=dz.device, dtype=torch.float32)          # [dgamma | dbeta | 0 = the conv bias's gradient]
        scratch = torch.empty(2, c, device=dz.device, dtype=torch.float32)
        dy = torch.empty_like(yc, memory_format=torch.channels_last)
        dsh = torch.empty_like(yc, memory_format=torch.channels_last) if ctx.has_short else None
        _lib.check(lib.syn_bn_act_bwd_apply(dzc.data_ptr(), _lib.ptr(z), yc.data_ptr(), stats.data_ptr(), g.data_ptr(), b.data_ptr(), local.data_ptr(),
                                            total.data_ptr(), rows, ctx.rows_total, c, int(ctx.act), dgb.data_ptr(), scratch.data_ptr(), dy.data_ptr(),
                                            _lib.ptr(dsh), _lib.current_stream(dz.device)), "syn_bn_act_bwd_apply")
        dcb = dgb[2] if ctx.has_cb else None
        return dy, dgb[0], dgb[1], dcb, dsh, None, None, None, None, None, None, None


def _conv_raw(conv, x):
    """The convolution alone (no bias) on channels_last (N, C, 1, L): the split-operand kernel for the Conv1d(k = 15) layers from
    block 0's conv2 on, the plain-fp32 one for the 1-2-channel first layer.  Anything else raises."""
    engine._require_cuda(x, "audio encoder input")
    cin, stride, pad, cout = conv.in_channels, conv.stride[0], conv.padding[0], conv.out_channels
    if x.dim() != 4 or conv.kernel_size[0] != 15 or conv.dilation[0] != 1:
        raise _unsupported_conv("forward", cin, stride, pad, cout)
    if (cin, stride, cout) in ConvSplitFn.SUPPORTED and pad % stride == 0:
        w4 = conv.weight.unsqueeze(2)
        w4._syn_owner = conv.weight
        return ConvSplitFn.apply(x, w4, stride, pad)
    if cin in (1, 2) and cout == 64 and 1 <= stride <= 8 and not x.requires_grad:
        n, _, _, l = x.shape                                                 # channels_last (N, cin, 1, L) = the waveform (N, L, cin)
        return ConvFirstFn.apply(x.permute(0, 2, 3, 1).reshape(n, l, cin), conv.weight, stride, pad)
    raise _unsupported_conv("forward", cin, stride, pad, cout)


_tracked: list = []


def _conv_bn_act(conv, bn, x, shortcut, act):
    """Training-mode conv -> BatchNorm (batch statistics) [+ shortcut] [-> LeakyReLU] with the fused tail."""
    if bn.momentum is None:
        raise _lib.SynHipError("the fused BatchNorm of the audio encoder implements the exponential running average (momentum = 0.1 in "
                               "the reference, models/utils/layer.py:160); momentum=None asks for a cumulative average")
    y = _conv_raw(conv, x)
    if isinstance(bn, nn.SyncBatchNorm):                     # train.py:90: statistics over all ranks of the module's process group
        z = SyncBnActFn.apply(y, bn.weight, bn.bias, conv.bias, shortcut, bn.running_mean, bn.running_var, bn.momentum, bn.eps, act,
                              bn.process_group, getattr(bn, "_syn_test_key", None))
    else:
        z = BnActFn.apply(y, bn.weight, bn.bias, conv.bias, shortcut, bn.running_mean, bn.running_var, bn.momentum, bn.eps, act)
    if bn.num_batches_tracked is not None:
        _tracked.append(bn.num_batches_tracked)              # +1 as nn.BatchNorm1d.forward does in train() mode (checkpoints carry it):
    return z                                                 # one launch for all of the encoder's counters, at the end of its forward


def _conv_bn_eval(conv, bn, x):
    """Conv1d + BatchNorm1d on the module's RUNNING statistics (eval() with autograd on: the gradient tests' deterministic mode)."""
    y = _conv_raw(conv, x)
    if conv.bias is not None:
        y = y + conv.bias.view(1, -1, 1, 1)
    return F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)


# ---- a BasicBlock of the audio encoder as ONE autograd node (round 5) -------------------------------------------------------------------
# models/utils/layer.py:171-184 in train() mode.  Built from `_conv_bn_act` above a block is 3 convolutions + 3 BatchNorm tails of three
# launches each way, and every tail is a pass of its own over tensors that are 117 MB each in block 0 at 32 clips: 13 tensor passes forward
# and 24 backward in block 0 alone (BatchNorm kernels: 1.0 ms of a 6.4 ms step, profiles/r04_train_step_split.txt).  As one node:
#   forward   conv1 (+ statistics from its epilogue) | finalize -> per-channel affine | [shortcut conv (+ statistics) | finalize] |
#             conv2 reading act(bn1(y1)) AS IT STAGES ITS TILE (z1 is never written) | finalize | ONE elementwise pass:
#             out = act(bn2(y2) + (bn_s(y_sc) | x))                                                  -> 7 tensor passes in block 0
#   backward  ONE statistics + ONE apply pass for bn2 AND the shortcut's BatchNorm over (dout, y2, y_sc | x), the activation's sign from the
#             recomputed pre-activation (`out` is neither saved nor read) | conv2 data gradient | conv2 weight gradient recomputing
#             act(bn1(y1)) as it stages | bn1 backward | conv1 (and shortcut) gradients                 -> 19 tensor passes in block 0
# Saved for the backward: x, y1, y2, y_sc and four small per-channel tables - not z1, not the normalised shortcut, not the output.


def _rows3(t):
    """(N, C, 1, L) channels_last fp32 -> the same memory as (N, L, C) contiguous (a copy only if the memory is laid out otherwise)."""
    t = t.detach()
    if t.dtype is not torch.float32:
        t = t.float()
    n, c, _, l = t.shape
    if t.stride(1) == 1 and t.stride(3) == c and (n == 1 or t.stride(0) == l * c):           # (the size-1 axis' stride is arbitrary)
        return t.as_strided((n, l, c), (l * c, c, 1))
    return t.contiguous(memory_format=torch.channels_last).permute(0, 3, 1, 2).reshape(n, l, c)


def _as4(t3):
    """(N, L, C) contiguous -> (N, C, 1, L) channels_last view of the same memory."""
    n, l, c = t3.shape
    return t3.view(n, 1, l, c).permute(0, 3, 1, 2)


def _wb_conv_fwd(x3, conv, first, in_aff=None, in_act=0):
    """The convolution alone (no bias) on (N, L, Cin) -> y (N, L_out, Cout), its BatchNorm partial sums, their chunk count."""
    lib, dev = _lib.load(), x3.device
    n, l_in, cin = x3.shape
    stride, pad, cout = conv.stride[0], conv.padding[0], conv.out_channels
    l_out = (l_in + 2 * pad - 15) // stride + 1
    y = torch.empty(n, l_out, cout, device=dev, dtype=torch.float32)
    st = _lib.current_stream(dev)
    if first:
        chunks = lib.syn_conv1d_first_tiles(n, l_out)
        part = torch.empty(chunks, 2, 64, device=dev, dtype=torch.float32)
        wc = conv.weight.detach()
        _lib.check(lib.syn_conv1d_first_fwd_stats(x3.data_ptr(), n, l_in, cin, stride, pad, wc.data_ptr(), y.data_ptr(), part.data_ptr(), st),
                   "syn_conv1d_first_fwd_stats")
        return y, part, chunks
    whi, wlo = _wb_pack(conv, False)
    chunks = lib.syn_conv1d_train_fwd_tiles(n, l_in, cin, stride, pad, cout)
    part = torch.empty(chunks, 2, cout, device=dev, dtype=torch.float32)
    _conv_terms(0)
    if in_aff is None:
        _lib.check(lib.syn_conv1d_train_fwd(x3.data_ptr(), n, l_in, cin, stride, pad, whi.data_ptr(), wlo.data_ptr(), None, cout, y.data_ptr(),
                                            part.data_ptr(), st), "syn_conv1d_train_fwd")
    else:
        _lib.check(lib.syn_conv1d_train_fwd_norm(x3.data_ptr(), n, l_in, cin, stride, pad, whi.data_ptr(), wlo.data_ptr(), cout, in_aff.data_ptr(),
                                                 int(in_act), y.data_ptr(), part.data_ptr(), st), "syn_conv1d_train_fwd_norm")
    _conv_terms_done()
    return y, part, chunks


def _wb_conv_fwd_pair(x3, conv_a, conv_b):
    """conv1 and the shortcut convolution of a down-sampling block - the same input and geometry - as one launch (`syn_conv1d_train_fwd_pair`: the
    input tile is staged once): -> (y_a, part_a, y_b, part_b, chunks)."""
    lib, dev = _lib.load(), x3.device
    n, l_in, cin = x3.shape
    stride, pad, cout = conv_a.stride[0], conv_a.padding[0], conv_a.out_channels
    l_out = (l_in + 2 * pad - 15) // stride + 1
    chunks = lib.syn_conv1d_train_fwd_tiles(n, l_in, cin, stride, pad, cout)
    ya, yb = (torch.empty(n, l_out, cout, device=dev, dtype=torch.float32) for _ in range(2))
    pa, pb = (torch.empty(chunks, 2, cout, device=dev, dtype=torch.float32) for _ in range(2))
    (ah, al), (bh, bl) = _wb_pack(conv_a, False), _wb_pack(conv_b, False)
    _conv_terms(0)
    _lib.check(lib.syn_conv1d_train_fwd_pair(x3.data_ptr(), n, l_in, cin, stride, pad, cout, ah.data_ptr(), al.data_ptr(), ya.data_ptr(), pa.data_ptr(),
                                             bh.data_ptr(), bl.data_ptr(), yb.data_ptr(), pb.data_ptr(), _lib.current_stream(dev)), "syn_conv1d_train_fwd_pair")
    _conv_terms_done()
    return ya, pa, yb, pb, chunks


def _wb_pack(conv, transposed):
    return _conv_pack_of(conv.weight, conv.out_channels, conv.in_channels, conv.stride[0], transposed)


def _wb_finalize(part, chunks, rows, bn, conv_bias, pair=None):
    """Per-channel sums -> (mean, rstd) and the affine (scale, shift) of a batch-statistics BatchNorm, running statistics updated (`syn_bn_finalize`).
    pair: a second (part, chunks, bn, conv_bias) of the same rows finalised by the same launch -> both results."""
    dev = part.device
    jobs, outs = [], []
    for pt, ch, b, cb in [(part, chunks, bn, conv_bias)] + ([pair] if pair is not None else []):
        c = b.num_features
        stats, aff = torch.empty(2, c, device=dev, dtype=torch.float32), torch.empty(2, c, device=dev, dtype=torch.float32)
        jobs.append((pt, ch, c, b, None if cb is None else cb.detach(), stats, aff))
        outs.append((stats, aff))
        if b.num_batches_tracked is not None:
            _tracked.append(b.num_batches_tracked)
    lib, st = _lib.load(), _lib.current_stream(dev)
    if pair is None:
        pt, ch, c, b, cb, stats, aff = jobs[0]
        _lib.check(lib.syn_bn_finalize(pt.data_ptr(), ch, rows, c, b.weight.detach().data_ptr(), b.bias.detach().data_ptr(), float(b.eps), float(b.momentum),
                                       _lib.ptr(b.running_mean), _lib.ptr(b.running_var), _lib.ptr(cb), stats.data_ptr(), aff.data_ptr(), st), "syn_bn_finalize")
        return outs[0]
    arr = []
    for pt, ch, c, b, cb, stats, aff in jobs:
        j = _lib.SynBnFinalizeJob()
        j.part, j.chunks, j.channels, j.rows, j.gamma, j.beta, j.eps, j.momentum = pt.data_ptr(), ch, c, rows, b.weight.detach().data_ptr(), b.bias.detach().data_ptr(), float(b.eps), float(b.momentum)
        j.run_mean, j.run_var, j.conv_bias, j.stats, j.affine = _lib.ptr(b.running_mean), _lib.ptr(b.running_var), _lib.ptr(cb), stats.data_ptr(), aff.data_ptr()
        arr.append(j)
    _lib.check(lib.syn_bn_finalize_pair(C.byref(arr[0]), C.byref(arr[1]), st), "syn_bn_finalize_pair")
    return outs[0], outs[1]


def _wb_dgrad(dy3, conv, l_in, dy3b=None, conv_b=None, residual=None):
    """Data gradient of `conv` (no bias): dy (N, L_out, Cout) -> dx (N, L_in, Cin); with (dy3b, conv_b) - a second convolution of the same
    geometry on the same input (a down-sampling block's shortcut) - the sum of both, with `residual` (N, L_in, Cin) that added: one launch,
    dx written once (`syn_conv1d_train_dgrad_sum`)."""
    lib, dev = _lib.load(), dy3.device
    n, l_out, cout = dy3.shape
    cin, stride, pad = conv.in_channels, conv.stride[0], conv.padding[0]
    whi, wlo = _wb_pack(conv, True)
    wb = _wb_pack(conv_b, True) if conv_b is not None else (None, None)
    if conv_b is not None and (conv_b.in_channels, conv_b.stride[0], conv_b.padding[0], conv_b.out_channels) != (cin, stride, pad, cout):
        raise _lib.SynHipError("the two convolutions of a summed data gradient must share their geometry")
    ok = ((stride == 1 and pad == 7 and (cout, 1, cin) in ConvSplitFn.SUPPORTED and conv_b is None)
          or (pad == 0 and (cout, stride) in ((64, 6), (128, 6), (256, 3)) and (stride * cin) % 128 == 0 and residual is None))
    if not ok:
        raise _unsupported_conv("data gradient", cin, stride, pad, cout)
    dx = torch.empty(n, l_in, cin, device=dev, dtype=torch.float32)
    _conv_terms(1)
    _lib.check(lib.syn_conv1d_train_dgrad_sum(dy3.data_ptr(), whi.data_ptr(), wlo.data_ptr(), _lib.ptr(dy3b), _lib.ptr(wb[0]), _lib.ptr(wb[1]),
                                              _lib.ptr(residual), n, l_in, cin, stride, pad, cout, dx.data_ptr(), _lib.current_stream(dev)),
               "syn_conv1d_train_dgrad_sum")
    _conv_terms_done()
    return dx


def _wb_wgrad(x3, dy3, conv, first, in_aff=None, in_act=0, sums=None):
    """Weight gradient (Cout, Cin, 15) of `conv` from its input x (N, L_in, Cin) and dy (N, L_out, Cout).  With `sums` (a list) the launch leaves its
    partial sums and the job is appended: `_wb_wgrad_sums` adds up all of a block's gradients in one launch."""
    lib, dev = _lib.load(), x3.device
    n, l_in, cin = x3.shape
    stride, pad, cout = conv.stride[0], conv.padding[0], conv.out_channels
    l_out = dy3.shape[1]
    gw = _grad_out(conv.weight, (cout, cin, 15))
    st = _lib.current_stream(dev)
    target = None if sums is not None else gw.data_ptr()
    if first:
        ws = torch.empty(lib.syn_conv1d_first_parts(n, l_out) * 64 * cin * 15, device=dev, dtype=torch.float32)
        _lib.check(lib.syn_conv1d_first_wgrad(x3.data_ptr(), dy3.data_ptr(), n, l_in, cin, stride, pad, ws.data_ptr(), target, st), "syn_conv1d_first_wgrad")
    else:
        kts = -(-15 // stride) * stride
        ws = torch.empty(lib.syn_conv1d_wgrad_shares(n, l_out, stride * cin, cout) * cout * kts * cin, device=dev, dtype=torch.float32)
        _conv_terms(2)
        if in_aff is None:
            _lib.check(lib.syn_conv1d_train_wgrad(x3.data_ptr(), dy3.data_ptr(), n, l_in, cin, stride, pad, cout, ws.data_ptr(), target, st), "syn_conv1d_train_wgrad")
        else:
            _lib.check(lib.syn_conv1d_train_wgrad_norm(x3.data_ptr(), dy3.data_ptr(), n, l_in, cin, stride, pad, cout, in_aff.data_ptr(), int(in_act),
                                                       ws.data_ptr(), target, st), "syn_conv1d_train_wgrad_norm")
        _conv_terms_done()
    if sums is not None:
        sums.append((ws, gw, n, l_out, cin, stride, cout, int(bool(first))))
    return gw


def _wb_wgrad_pair(x3, dy_a, dy_b, conv_a, conv_b, sums):
    """conv1's and the shortcut convolution's weight gradients of block 1 ((64, stride 6, 64), unpadded: the same 117 MB input) from one launch that stages the
    input once (`syn_conv1d_train_wgrad_pair`); None where the pair kernel does not apply.  Partial sums only: two jobs appended to `sums`."""
    n, l_in, cin = x3.shape
    key = lambda c: (c.in_channels, c.stride[0], c.padding[0], c.out_channels)
    if not (key(conv_a) == key(conv_b) == (64, 6, 0, 64) and cin == 64):
        return None
    lib, dev = _lib.load(), x3.device
    l_out = dy_a.shape[1]
    per = 64 * 18 * 64                                            # cout x (ceil(15 / 6) x 6 taps) x cin
    ws = torch.empty(lib.syn_conv1d_wgrad_shares(n, l_out, 6 * 64, 64) * 2 * per, device=dev, dtype=torch.float32)
    ga, gb = _grad_out(conv_a.weight, (64, 64, 15)), _grad_out(conv_b.weight, (64, 64, 15))
    _conv_terms(2)
    _lib.check(lib.syn_conv1d_train_wgrad_pair(x3.data_ptr(), dy_a.data_ptr(), dy_b.data_ptr(), n, l_in, 64, 6, 0, 64, ws.data_ptr(), _lib.current_stream(dev)),
               "syn_conv1d_train_wgrad_pair")
    _conv_terms_done()
    sums.append((ws, ga, n, l_out, 64, 6, 64, 0, 2 * per))
    sums.append((ws[per:], gb, n, l_out, 64, 6, 64, 0, 2 * per))
    return ga, gb


def _wb_wgrad_sums(sums, device):
    """The partial-sum reductions of a block's weight gradients as one launch (`syn_conv1d_wgrad_sums`)."""
    if not sums:
        return
    arr = (_lib.SynWgradSumJob * len(sums))()
    for i, (ws, gw, n, l_out, cin, stride, cout, first, *pitch) in enumerate(sums):
        arr[i].part, arr[i].dw, arr[i].n_clips, arr[i].l_out, arr[i].cin, arr[i].stride, arr[i].cout, arr[i].first_layer = ws.data_ptr(), gw.data_ptr(), n, l_out, cin, stride, cout, first
        arr[i].share_pitch = pitch[0] if pitch else 0            # (a share that holds two gradients: `_wb_wgrad_pair`)
    _lib.check(_lib.load().syn_conv1d_wgrad_sums(arr, len(sums), _lib.current_stream(device)), "syn_conv1d_wgrad_sums")
    sums.clear()


class WavBlockFn(torch.autograd.Function):
    """act(bn2(conv2(act(bn1(conv1(x))))) + shortcut(x)) of one BasicBlock on batch statistics (see the comment above).
    x: (N, Cin, 1, L) channels_last, or for the encoder's first block the waveform (N, L, cin); returns (N, Cout, 1, L_out) channels_last.
    params: conv1.weight, conv1.bias, bn1.weight, bn1.bias, conv2.weight, conv2.bias, bn2.weight, bn2.bias [, shortcut conv.weight, .bias,
    shortcut bn.weight, .bias] - listed so that autograd routes their gradients; the modules themselves come through `blk`."""

    @staticmethod
    def forward(ctx, x, blk, first, *params):
        lib = _lib.load()
        x3 = x.detach().float().contiguous() if first else _rows3(x)
        n = x3.shape[0]
        ds = blk.downsample is not None
        ysc = sts = afs = None
        if first and ds:
            # conv1 and the shortcut convolution of block 0 read the same waveform window: one launch
            c0, c1m = blk.conv1, blk.downsample[0]
            l_in, cin = x3.shape[1], x3.shape[2]
            l_out = (l_in + 2 * c0.padding[0] - 15) // c0.stride[0] + 1
            c1 = cs = lib.syn_conv1d_first_tiles(n, l_out)
            y1, ysc = (torch.empty(n, l_out, 64, device=x3.device, dtype=torch.float32) for _ in range(2))
            p1, ps = (torch.empty(c1, 2, 64, device=x3.device, dtype=torch.float32) for _ in range(2))
            _lib.check(lib.syn_conv1d_first_fwd2(x3.data_ptr(), n, l_in, cin, c0.stride[0], c0.padding[0], c0.weight.detach().data_ptr(),
                                                 c1m.weight.detach().data_ptr(), y1.data_ptr(), ysc.data_ptr(), p1.data_ptr(), ps.data_ptr(),
                                                 _lib.current_stream(x3.device)), "syn_conv1d_first_fwd2")
        elif ds:
            y1, p1, ysc, ps, c1 = _wb_conv_fwd_pair(x3, blk.conv1, blk.downsample[0])
            cs = c1
        else:
            y1, p1, c1 = _wb_conv_fwd(x3, blk.conv1, first)
        rows = n * y1.shape[1]
        if ds:
            (st1, af1), (sts, afs) = _wb_finalize(p1, c1, rows, blk.bn1, blk.conv1.bias, pair=(ps, cs, blk.downsample[1], blk.downsample[0].bias))
        else:
            st1, af1 = _wb_finalize(p1, c1, rows, blk.bn1, blk.conv1.bias)
        y2, p2, c2 = _wb_conv_fwd(y1, blk.conv2, False, in_aff=af1, in_act=1)
        st2, af2 = _wb_finalize(p2, c2, rows, blk.bn2, blk.conv2.bias)
        c = y2.shape[2]
        short = ysc if ds else x3
        if not ds and tuple(x3.shape) != tuple(y2.shape):
            raise _lib.SynHipError(f"identity shortcut of shape {tuple(x3.shape)} on a block output of shape {tuple(y2.shape)}")
        out = torch.empty_like(y2)
        _lib.check(lib.syn_bn_apply2(y2.data_ptr(), af2.data_ptr(), short.data_ptr(), _lib.ptr(afs), rows, c, 1, out.data_ptr(),
                                     _lib.current_stream(out.device)), "syn_bn_apply2")
        ctx.save_for_backward(x3, y1, y2, ysc, st1, af1, st2, af2, sts, afs)
        ctx.blk, ctx.first = blk, bool(first)
        return _as4(out)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x3, y1, y2, ysc, st1, af1, st2, af2, sts, afs = ctx.saved_tensors
        blk, first = ctx.blk, ctx.first
        ds = blk.downsample is not None
        n, l1, c = y2.shape
        rows = n * l1
        dev = y2.device
        d3 = _rows3(dout)
        ws = torch.empty(3 * lib.syn_bn_chunks(rows) * c, device=dev, dtype=torch.float32)
        dgb2 = torch.empty(3, c, device=dev, dtype=torch.float32)
        dgbs = torch.empty(3, c, device=dev, dtype=torch.float32) if ds else None
        short = ysc if ds else x3
        sums = []
        # block 0 with its shortcut convolution: the tail's apply pass and the shortcut's weight gradient are ONE kernel - it reads (dout, y2, y_sc) once,
        # writes dy2 and feeds the shortcut's dy straight into the weight gradient's matrix products (that tensor is never written or read back)
        tail0 = first and ds and blk.conv1.stride[0] == 5
        dy2 = torch.empty_like(y2)
        dsh = None if tail0 else torch.empty_like(y2)
        _lib.check(lib.syn_bn_block_bwd(d3.data_ptr(), y2.data_ptr(), short.data_ptr(), st2.data_ptr(), af2.data_ptr(), _lib.ptr(sts), _lib.ptr(afs),
                                        rows, c, 1, ws.data_ptr(), dgb2.data_ptr(), _lib.ptr(dgbs), None if tail0 else dy2.data_ptr(), _lib.ptr(dsh),
                                        _lib.current_stream(dev)), "syn_bn_block_bwd")
        gws = None
        if tail0:
            cs_ = blk.downsample[0]
            nn_, l_in, cin = x3.shape
            wss = torch.empty(lib.syn_conv1d_first_parts(nn_, l1) * 64 * cin * 15, device=dev, dtype=torch.float32)
            gws = _grad_out(cs_.weight, (64, cin, 15))
            _lib.check(lib.syn_conv1d_first_wgrad_tail(x3.data_ptr(), d3.data_ptr(), y2.data_ptr(), ysc.data_ptr(), st2.data_ptr(), af2.data_ptr(), sts.data_ptr(),
                                                       afs.data_ptr(), dgb2.data_ptr(), dgbs.data_ptr(), 1, nn_, l_in, cin, cs_.stride[0], cs_.padding[0],
                                                       wss.data_ptr(), dy2.data_ptr(), _lib.current_stream(dev)), "syn_conv1d_first_wgrad_tail")
            sums.append((wss, gws, nn_, l1, cin, cs_.stride[0], 64, 1))
        # conv2: data gradient to z1 = act(bn1(y1)), weight gradient with z1 recomputed from y1 while its rows are staged
        dz1 = _wb_dgrad(dy2, blk.conv2, l1)
        gw2 = _wb_wgrad(y1, dy2, blk.conv2, False, in_aff=af1, in_act=1, sums=sums)
        # bn1 + activation (no shortcut entered it: the sign comes from y1)
        dgb1 = torch.empty(3, c, device=dev, dtype=torch.float32)
        if first and blk.conv1.stride[0] == 5:
            # block 0: nothing but conv1's weight gradient reads dy1 (the waveform takes no gradient), and the BatchNorm backward is linear in bn1's two sums:
            # ONE pass over (dz1, y1) accumulates the gradient's three terms and the sums side by side (`syn_conv1d_first_wgrad_bn_lin`)
            cv = blk.conv1
            nn_, l_in, cin = x3.shape
            wsg = torch.empty(lib.syn_conv1d_first_parts(nn_, l1) * (2 * 64 * cin * 15 + 160), device=dev, dtype=torch.float32)
            gw1 = _grad_out(cv.weight, (64, cin, 15))
            _lib.check(lib.syn_conv1d_first_wgrad_bn_lin(x3.data_ptr(), dz1.data_ptr(), y1.data_ptr(), st1.data_ptr(), af1.data_ptr(), 1, nn_, l_in, cin,
                                                         cv.stride[0], cv.padding[0], wsg.data_ptr(), gw1.data_ptr(), dgb1.data_ptr(), _lib.current_stream(dev)),
                       "syn_conv1d_first_wgrad_bn_lin")
            dy1 = None
        else:
            dy1 = torch.empty_like(y1)
            ws1 = torch.empty(2 * lib.syn_bn_chunks(rows) * c, device=dev, dtype=torch.float32)
            g1, b1 = blk.bn1.weight.detach(), blk.bn1.bias.detach()
            _lib.check(lib.syn_bn_act_bwd(dz1.data_ptr(), None, y1.data_ptr(), st1.data_ptr(), g1.data_ptr(), b1.data_ptr(), rows, c, 1, ws1.data_ptr(),
                                          dgb1.data_ptr(), dy1.data_ptr(), None, _lib.current_stream(dev)), "syn_bn_act_bwd")
            pair = _wb_wgrad_pair(x3, dy1, dsh, blk.conv1, blk.downsample[0], sums) if (ds and not first) else None
            if pair is not None:
                gw1, gws = pair
            else:
                gw1 = _wb_wgrad(x3, dy1, blk.conv1, first, sums=sums)
        dx = None
        if not first and ctx.needs_input_grad[0]:
            # what reaches the block's input, written once: conv1^T dy1 + (shortcut^T dy_sc | the gradient along the identity shortcut)
            dx = (_wb_dgrad(dy1, blk.conv1, x3.shape[1], dy3b=dsh, conv_b=blk.downsample[0]) if ds
                  else _wb_dgrad(dy1, blk.conv1, x3.shape[1], residual=dsh))
        grads = [gw1, dgb1[2] if blk.conv1.bias is not None else None, dgb1[0], dgb1[1],
                 gw2, dgb2[2] if blk.conv2.bias is not None else None, dgb2[0], dgb2[1]]
        owners = [None, blk.conv1.bias, blk.bn1.weight, blk.bn1.bias, None, blk.conv2.bias, blk.bn2.weight, blk.bn2.bias]
        if ds:
            if gws is None:
                gws = _wb_wgrad(x3, dsh, blk.downsample[0], first, sums=sums)
            grads += [gws, dgbs[2] if blk.downsample[0].bias is not None else None, dgbs[0], dgbs[1]]
            owners += [None, blk.downsample[0].bias, blk.downsample[1].weight, blk.downsample[1].bias]
        _wb_wgrad_sums(sums, dev)                              # the block's weight gradients: their partial sums added up in one launch
        _into_bound_buffers(grads, owners)
        return (None if dx is None else _as4(dx), None, None, *grads)


def _wav_block_fused_ok(blk, x, first) -> bool:
    if not (blk.training and torch.is_grad_enabled() and x.is_cuda):
        return False
    bns = [blk.bn1, blk.bn2] + ([blk.downsample[1]] if blk.downsample is not None else [])
    if any(isinstance(b, nn.SyncBatchNorm) or not b.track_running_stats or b.momentum is None or b.weight is None for b in bns):
        return False                                          # (SyncBatchNorm: the per-convolution path with its all-reduces, `_conv_bn_act`)
    convs = [blk.conv1, blk.conv2] + ([blk.downsample[0]] if blk.downsample is not None else [])
    for i, cv in enumerate(convs):
        if cv.kernel_size[0] != 15 or cv.dilation[0] != 1 or cv.weight.dtype is not torch.float32:
            return False
        key = (cv.in_channels, cv.stride[0], cv.out_channels)
        if first and i != 1:
            if not (cv.in_channels in (1, 2) and cv.out_channels == 64 and 1 <= cv.stride[0] <= 8):
                return False
        elif key not in ConvSplitFn.SUPPORTED or cv.padding[0] % cv.stride[0]:
            return False
    c1, c2 = blk.conv1, blk.conv2
    if not (c2.stride[0] == 1 and c2.padding[0] == 7 and c2.in_channels == c2.out_channels):
        return False
    if first:                                                  # block 0: conv1 and the shortcut convolution read one waveform window (`syn_conv1d_first_fwd2`)
        if blk.downsample is None:
            return True
        sc = blk.downsample[0]
        return (sc.stride[0], sc.padding[0], sc.out_channels, sc.in_channels) == (c1.stride[0], c1.padding[0], 64, c1.in_channels) and c1.stride[0] == 5
    # what the backward's summed data gradient covers (`_wb_dgrad`): an identity block's stride-1 'same' conv1, or an unpadded strided conv1 + shortcut pair
    if blk.downsample is None:
        return c1.stride[0] == 1 and c1.padding[0] == 7 and (c1.out_channels, 1, c1.in_channels) in ConvSplitFn.SUPPORTED
    sc = blk.downsample[0]
    return (c1.padding[0] == 0 and (c1.out_channels, c1.stride[0]) in ((64, 6), (128, 6), (256, 3)) and (c1.stride[0] * c1.in_channels) % 128 == 0
            and (sc.in_channels, sc.stride[0], sc.padding[0], sc.out_channels) == (c1.in_channels, c1.stride[0], c1.padding[0], c1.out_channels))


def _wav_block_params(blk):
    ps = [blk.conv1.weight, blk.conv1.bias, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.conv2.bias, blk.bn2.weight, blk.bn2.bias]
    if blk.downsample is not None:
        ps += [blk.downsample[0].weight, blk.downsample[0].bias, blk.downsample[1].weight, blk.downsample[1].bias]
    return ps


def _wav_block(blk, x):
    """models/utils/layer.py:171-184 on channels_last (N, C, 1, L), train or eval statistics as the module says."""
    if blk.training:
        if not blk.bn1.track_running_stats:
            raise _lib.SynHipError("the audio encoder's BatchNorms must track running statistics (the reference's do)")
        first = blk.conv1.in_channels in (1, 2) and not x.requires_grad
        if _wav_block_fused_ok(blk, x, first):
            if first:
                n, cin, _, l = x.shape                           # channels_last (N, cin, 1, L) = the waveform (N, L, cin)
                x = x.permute(0, 2, 3, 1).reshape(n, l, cin)
            return WavBlockFn.apply(x, blk, first, *_wav_block_params(blk))
        z = _conv_bn_act(blk.conv1, blk.bn1, x, None, True)
        short = x if blk.downsample is None else _conv_bn_act(blk.downsample[0], blk.downsample[1], x, None, False)
        return _conv_bn_act(blk.conv2, blk.bn2, z, short, True)
    z = F.leaky_relu(_conv_bn_eval(blk.conv1, blk.bn1, x), 0.01)
    z = _conv_bn_eval(blk.conv2, blk.bn2, z)
    short = x if blk.downsample is None else _conv_bn_eval(blk.downsample[0], blk.downsample[1], x)
    return F.leaky_relu(z + short, 0.01)



def _step_packs(m, training: bool):
    """Refresh the step's fragment sets: every Linear outside the blocks now (one launch), the audio encoder's convolutions (one launch); the blocks'
    Linears are packed later, right in front of the blocks (`train_forward`)."""
    global _packs, _packs_blocks, _conv_packs
    pk = m.__dict__.get("_syn_weight_packs")
    if pk is None or pk[0].owner() is not m or not (pk[0].valid() and pk[1].valid()):    # (a deep copy of the model brings the original's cache along)
        in_blocks = {id(mod.weight) for blk in m.mytimmblocks for mod in blk.modules() if isinstance(mod, nn.Linear)}
        lin_w = [mod.weight for mod in m.modules() if isinstance(mod, nn.Linear)]
        pk = m.__dict__["_syn_weight_packs"] = (WeightPacks([w for w in lin_w if id(w) not in in_blocks]), WeightPacks([w for w in lin_w if id(w) in in_blocks]))
        pk[0].owner = pk[1].owner = weakref.ref(m)
    pk[0].refresh()
    _packs, _packs_blocks = pk
    cp = m.__dict__.get("_syn_conv_packs")
    if training and (cp is None or cp.owner() is not m or not cp.valid()):
        cp = m.__dict__["_syn_conv_packs"] = ConvPacks([mod for mod in m.WavEncoder.modules() if isinstance(mod, nn.Conv1d)])
        cp.owner = weakref.ref(m)
    if training and cp.lists:
        cp.refresh()
        _conv_packs = cp
    else:
        _conv_packs = None


def train_forward(m, x, timesteps, y, drop_path: float = 0.1):
    """Differentiable MDM.forward, op-for-op with models/denoiser.py:132-196 (denoiser_h3d.py:148-221), with the
    module's current train()/eval() semantics.  x (B,1536,1,T) -> (B,1536,1,T) (a permuted view of the output Linear's [B][T][C] rows)."""
    engine._require_cuda(x, "x")
    if x.requires_grad and torch.is_grad_enabled():
        raise _lib.SynHipError("the training path provides no gradient with respect to x_t (training_losses never asks for one, "
                               "gaussian_diffusion.py:1236-1363): detach the latent")
    bs, C_, _, T = x.shape
    training = m.training
    if torch.is_grad_enabled() and any(getattr(p, "_syn_grad_handed", False) for p in m.parameters()):
        _reset_handed(m)                                       # a new step: every bound gradient buffer may be handed out again (`_grad_out`)
    _step_packs(m, training)
    if training:
        engine.note_raw_write()                                # (the BatchNorm running statistics are written by the finalize kernels, not by a tensor op)
    h3d = m.variant == "h3d"
    te = m.embed_timestep
    e = te.sequence_pos_encoder.pe[timesteps]                                   # (B,1,512)
    emb_t = lin(F.silu(lin(e, te.time_embed[0])), te.time_embed[2]).reshape(bs, -1)
    emb_seed = lin(y["seed"].reshape(bs, -1), m.embed_text)
    audio, word = y["audio"], y["word"]
    if h3d and y.get("uncond_audio", False):
        audio, word = torch.zeros_like(audio), torch.zeros_like(word)
    a = audio.unsqueeze(1) if audio.dim() == 2 else audio.transpose(1, 2)
    # (train.py:90 may have converted the BatchNorms to SyncBatchNorm: `_conv_bn_act` then reduces the statistics over the module's
    # process group - the same kernels, one small all-reduce per BatchNorm and direction)
    a = a.unsqueeze(2).contiguous(memory_format=torch.channels_last)       # (B, C, 1, L), channel innermost
    _tracked.clear()
    for blk in m.WavEncoder.feat_extractor:
        a = _wav_block(blk, a)
    if _tracked:
        torch._foreach_add_(list(_tracked), 1)
        _tracked.clear()
    # From here on rows are (clip, frame), not the reference's (frame, clip) (denoiser.py:151-176): every op below is row-wise or acts along the
    # frame axis of one clip, so the order is free - and this one needs no transposing copy between the encoder, the blocks and the output.
    n, ca, _, fr = a.shape
    a_rows = a.permute(0, 3, 1, 2).reshape(n, fr, ca)        # (B, 128, 256): the channels_last output of the encoder as it lies in memory
    pool = getattr(m.args, "vqvae_squeeze_scale", 4) if not h3d else 4
    if fr // pool != T:
        raise _lib.SynHipError(f"{fr} audio frames pooled by {pool} do not give the latent's {T} frames (models/denoiser.py:157)")
    if fr % pool:
        a_rows = a_rows[:, :fr // pool * pool]               # (F.avg_pool1d drops the incomplete window)
        word = word[:, :fr // pool * pool]
    style = None
    if m.uses_style:
        style = y["style_feature"]
        force = bool(y.get("uncond", False))
        null = m.uncon_text_embeddings.repeat(bs, 1) if h3d else torch.zeros_like(style)
        if force:
            style = null
        elif training and m.cond_mask_prob > 0.:
            mask = torch.bernoulli(torch.ones(bs, device=style.device) * m.cond_mask_prob).view(bs, 1)
            style = style * (1. - mask) + null * mask
    # DropPath (timm_transformer/transformer.py:21-38: one Bernoulli(keep) / keep factor per sample and residual branch): all the
    # step's factors from one draw, applied as x + branch * factor in the branch's last GEMM
    dp = None
    if training and drop_path > 0.:
        keep = 1. - drop_path
        dp = x.new_empty(2 * len(m.mytimmblocks), bs, 1, 1).bernoulli_(keep).div_(keep)
    # The block kernels work on tiles of 4 clips: other batch sizes get empty clips appended here (zero rows, factor 1); their rows carry zero
    # gradients back and are cut from the output.  (PyTorch ops: the odd-batch path is a convenience, not a tuned one.)
    padn = (-bs) % 4
    xin, word_in = x.detach(), word
    if padn:
        zr = lambda t: F.pad(t, (0, 0) * (t.dim() - 1) + (0, padn))
        a_rows, xin, emb_seed, emb_t, word_in = zr(a_rows), zr(xin), zr(emb_seed), zr(emb_t), zr(word)
        style = None if style is None else zr(style)
        dp = None if dp is None else F.pad(dp, (0, 0, 0, 0, 0, padn), value=1.0)
    B = bs + padn
    cs, sn = _rotary_tables(m, T, x.device)
    ip3 = getattr(m, "input_process3", None) if m.uses_style else None
    tb, mx, pe_, ip2 = m.text_encoder_body, m.mix_audio_text, m.input_process.poseEmbedding, m.input_process2
    h = InputStageFn.apply(a_rows, xin, emb_seed, emb_t, style, word_in, cs, sn, m.text_pre_encoder_body.weight, tb.weight, tb.bias, mx.weight, mx.bias,
                           pe_.weight, pe_.bias, ip2.weight, ip2.bias, None if ip3 is None else ip3.weight, None if ip3 is None else ip3.bias)
    if not _blocks_ok(m):
        raise _lib.SynHipError("the block kernels implement the reference's Block (timm_transformer/transformer.py:154-198: 512 wide, 4 heads, qkv without "
                               "bias, MLP 1024); this model's blocks differ")
    # The blocks' fragment sets are packed HERE, not at the top of the forward: the audio encoder in between moves ~1 GB through the memory-side cache,
    # and the persistent block kernel below - latency-bound, every phase waits for its first weight fragments - then finds them in HBM
    _packs_blocks.refresh()
    if _stack_ok(m, B, T, h.device):
        ps = []
        for blk in m.mytimmblocks:
            ps += [blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.proj.weight, blk.attn.proj.bias, blk.norm2.weight, blk.norm2.bias,
                   blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias]
        h = StackFn.apply(h, dp, *ps)
    else:
        for i, blk in enumerate(m.mytimmblocks):
            h = AttnBranchFn.apply(h, blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.proj.weight,
                                   blk.attn.proj.bias, None if dp is None else dp[2 * i])
            h = MlpBranchFn.apply(h, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight,
                                  blk.mlp.fc2.bias, None if dp is None else dp[2 * i + 1])
    out = lin(h, m.output_process.poseFinal)                # (B, T, C) rows
    if padn:
        out = out[:bs]
    return out.permute(0, 2, 1).unsqueeze(2)                # (B, C, 1, T) as the reference returns it: a view - `MaskedSmoothL1Fn` reads the rows in place
