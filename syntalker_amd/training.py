"""Training path of the denoiser (e): `training_losses` forward + backward.

  * every nn.Linear of the denoiser (qkv, proj, fc1, fc2, poseEmbedding, input_process2/3, poseFinal, embed_text,
    time MLP, word/mix projections) runs forward, dgrad and wgrad on the hand-written MFMA GEMM through the C ABI
    (`syn_linear`, `syn_linear_pair`: bf16 operands, fp32 accumulate/output) — `HipLinearFn`;
  * LayerNorm, the 32-token softmax attention and GELU of the 8 blocks run forward and backward on fp32 HIP kernels
    (`syn_ln_*`, `syn_attn_*`, `syn_gelu_*`; `HipLnForkFn`, `HipAttentionFn`, `HipGeluFn`);
  * the WavEncoder (78 % of the training FLOPs, cannot be hoisted in training): every Conv1d forward / data gradient / weight
    gradient on split-operand MFMA kernels (fp32-grade), the 1-2-channel first layer on plain fp32 FMAs, BatchNorm on batch
    statistics + shortcut + LeakyReLU fused (`ConvSplitFn`, `ConvFirstFn`, `BnActFn`).  No library convolution is on this
    path: a layer geometry the kernels do not cover raises;
  * rotary, DropPath's multiply-add and the SmoothL1 loss are fp32 PyTorch elementwise ops;
  * data parallelism: one process per GPU, torch DDP over RCCL (`make_ddp`), gradients averaged by bucketed
    all-reduce overlapped with backward.  SyncBatchNorm (the reference's DDP branch converts every BatchNorm, train.py:90) runs on the
    same kernels: per-channel fp64 sums -> one small all-reduce -> finalise (`SyncBnActFn`, `syn_bn_sums` / `syn_bn_act_apply` / ...).
Train-mode semantics follow the reference: BatchNorm batch statistics, DropPath(0.1) per sample with
scale-by-keep (timm_transformer/transformer.py:21-38), h3d Bernoulli(0.3) style dropout
(denoiser_h3d.py:116-124).  There is no CPU fallback: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, engine


def _ceil_to(a: int, m: int) -> int:
    return (a + m - 1) // m * m


def hip_matmul_nt(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """fp32 y[M, N] = x[M, K] @ w[N, K]^T (+ bias) on the MFMA GEMM (operands rounded to bf16).
    Shapes are zero-padded to the kernel's granularity (N % 512, K % 128)."""
    engine._require_cuda(x, "training input")
    M, K = x.shape
    N = w.shape[0]
    Kp, Np = _ceil_to(K, 128), _ceil_to(N, 512)
    xb = x.to(torch.bfloat16)
    wf = w.float()
    if Kp != K:
        xb = F.pad(xb, (0, Kp - K))
        wf = F.pad(wf, (0, Kp - K))
    if Np != N:
        wf = F.pad(wf, (0, 0, 0, Np - N))
    xb = xb.contiguous()
    wp = engine.pack_weight(wf.contiguous())
    b = None
    if bias is not None:
        b = bias.float()
        if Np != N:
            b = F.pad(b, (0, Np - N))
        b = b.contiguous()
    y = torch.empty(M, Np, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().syn_linear(xb.data_ptr(), wp.data_ptr(), _lib.ptr(b), M, Np, Kp, y.data_ptr(),
                                      _lib.current_stream(y.device)), "syn_linear")
    return y if Np == N else y[:, :N]


def _pack_t(src: torch.Tensor, n: int, k: int) -> torch.Tensor:
    """Packed fragments of W = src^T for a row-major src [k][n] (fp32 or bf16), no transposed copy."""
    out = torch.empty(n * k * 2, dtype=torch.uint8, device=src.device)
    _lib.check(_lib.load().syn_pack_weight_t(src.data_ptr(), int(src.dtype is torch.bfloat16), n, k, out.data_ptr(),
                                             _lib.current_stream(src.device)), "syn_pack_weight_t")
    return out


def _gemm_packed(xb: torch.Tensor, wp: torch.Tensor, n: int, k: int) -> torch.Tensor:
    """fp32 y[M][n] = xb[M][k] (bf16, contiguous) . W^T for packed W[n][k]; n % 512 == 0, k % 128 == 0."""
    y = torch.empty(xb.shape[0], n, dtype=torch.float32, device=xb.device)
    _lib.check(_lib.load().syn_linear(xb.data_ptr(), wp.data_ptr(), None, xb.shape[0], n, k, y.data_ptr(), _lib.current_stream(y.device)),
               "syn_linear")
    return y


class WeightPacks:
    """The bf16 fragment sets the step's Linear layers take - W for the forward GEMM, W^T for the data-gradient GEMM - packed from
    the fp32 master weights in ONE launch per step (`syn_pack_weights`) instead of one launch per use (82 per step): the weights
    only change in optimizer.step().  `refresh()` at the top of every training forward; HipLinearFn looks a weight up by object
    identity and in-place version, so a stale or foreign tensor simply takes the per-call packing path."""

    def __init__(self, weights):
        import numpy as np
        self.owner = lambda: None                                # the model the cache belongs to (weak reference, set by its user)
        self.items = {}
        jobs, self.max_frag = [], 0
        dev = None
        # two arenas - all forward sets, all transposed sets - so that either can be pulled into the memory-side cache with one pass (`prefetch`)
        ok = [w for w in weights if w.is_cuda and w.dtype is torch.float32 and w.is_contiguous() and w.dim() == 2]
        nb_f = sum(w.numel() * 2 for w in ok if w.shape[0] % 128 == 0 and w.shape[1] % 128 == 0)
        nb_t = sum(w.numel() * 2 for w in ok if w.shape[1] % 512 == 0 and w.shape[0] % 128 == 0)
        self.arena_fwd = torch.empty(nb_f, dtype=torch.uint8, device=ok[0].device) if ok and nb_f else None
        self.arena_tr = torch.empty(nb_t, dtype=torch.uint8, device=ok[0].device) if ok and nb_t else None
        of = ot = 0
        for w in weights:
            if not (w.is_cuda and w.dtype is torch.float32 and w.is_contiguous() and w.dim() == 2) or id(w) in self.items:
                continue
            dev = w.device
            N, K = w.shape
            fwd = tr = None
            if N % 128 == 0 and K % 128 == 0:                        # (N % 512 != 0: the 128-column GEMM)
                fwd = self.arena_fwd[of:of + N * K * 2]
                of += N * K * 2
            if K % 512 == 0 and N % 128 == 0:
                tr = self.arena_tr[ot:ot + N * K * 2]
                ot += N * K * 2
            if fwd is None and tr is None:
                continue
            if fwd is not None:
                jobs.append((w.data_ptr(), fwd.data_ptr(), N, K, 0, 0))
            if tr is not None:
                jobs.append((w.data_ptr(), tr.data_ptr(), K, N, 1, 0))       # fragments of W^T [K][N] from the row-major [N][K]
            self.max_frag = max(self.max_frag, (N // 16) * (K // 32))
            self.items[id(w)] = [__import__("weakref").ref(w), w.data_ptr(), -1, fwd, tr]
        self.n_jobs = len(jobs)
        if jobs:
            arr = np.array(jobs, dtype=np.dtype([("src", "<u8"), ("out", "<u8"), ("n", "<i4"), ("k", "<i4"), ("t", "<i4"), ("pad", "<i4")]))
            self.jobs = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)

    def valid(self) -> bool:
        return all(r() is not None and r().data_ptr() == ptr for r, ptr, *_ in self.items.values())

    def prefetch(self, transposed: bool):
        """One read pass over the forward / transposed fragment sets (a byte per 64): whatever has been evicted from the memory-side cache since the
        pack comes back before a latency-bound consumer asks for it fragment by fragment."""
        a = self.arena_tr if transposed else self.arena_fwd
        if a is not None and a.numel() >= 64:
            n = a.numel() // 64 * 64
            return a[:n].view(-1, 64)[:, 0].sum(dtype=torch.int32)
        return None

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


WEIGHT_PACKS = bool(int(__import__("os").environ.get("SYN_WEIGHT_PACKS", "1")))       # training forward: pack every Linear weight (and its transpose) in one launch per step
_packs: "WeightPacks | None" = None


_packs_blocks: "WeightPacks | None" = None      # the transformer blocks' Linears: packed right in front of the blocks (see train_forward)


def _lookup_packs(w):
    if not WEIGHT_PACKS:
        return None, None
    for pk in (_packs, _packs_blocks):
        if pk is not None:
            r = pk.lookup(w)
            if r[0] is not None or r[1] is not None:
                return r
    return None, None


_bias_counters = {}


def _counters(device):
    """Arrival counters of syn_linear_bwd_prep's in-launch bias-gradient sum: zero before the first use, left zero by every launch."""
    c = _bias_counters.get(device)
    if c is None:
        c = _bias_counters[device] = torch.zeros(1024, dtype=torch.int32, device=device)
    return c


class HipLinearFn(torch.autograd.Function):
    """y = x W^T + b with forward, dgrad and wgrad on syn_linear."""

    @staticmethod
    def forward(ctx, x, w, b):
        K = x.shape[-1]
        xb = x.reshape(-1, K).to(torch.bfloat16)
        pk, ctx.pack_t = _lookup_packs(w)
        xt = None
        if pk is not None and (b is None or (b.dtype is torch.float32 and b.is_contiguous())):
            xb = xb.contiguous()
            M = xb.shape[0]
            y = torch.empty(M, w.shape[0], dtype=torch.float32, device=x.device)
            if LINEAR_FWD_PACK and ctx.needs_input_grad[1] and K % 512 == 0 and M % 128 == 0 and M <= 2048:
                # the weight gradient's B operand (x^T as fragments) packed in the shadow of this GEMM instead of by a launch in the backward
                xt = torch.empty(K * M * 2, dtype=torch.uint8, device=x.device)
                _lib.check(_lib.load().syn_linear_and_pack(xb.data_ptr(), pk.data_ptr(), _lib.ptr(b), M, w.shape[0], K, y.data_ptr(), xt.data_ptr(),
                                                           _lib.current_stream(y.device)), "syn_linear_and_pack")
            else:
                _lib.check(_lib.load().syn_linear(xb.data_ptr(), pk.data_ptr(), _lib.ptr(b), M, w.shape[0], K, y.data_ptr(),
                                                  _lib.current_stream(y.device)), "syn_linear")
        else:
            y = hip_matmul_nt(xb, w, b)
        ctx.xt = xt
        ctx.save_for_backward(xb, w)
        ctx.owners = (w, b)
        ctx.has_bias = b is not None
        ctx.in_shape = x.shape
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xb, w = ctx.saved_tensors
        N, K = w.shape
        M = xb.shape[0]
        dy2 = dy.reshape(-1, N)
        dx = dw = db = dybt = None
        if SMALL_M_WGRAD and 0 < M <= 64 and N % 16 == 0 and dy2.is_cuda and ctx.needs_input_grad[1] and w.dtype is torch.float32:
            # a Linear that saw one row per clip (timestep MLP, embed_text): weight + bias gradient in ONE fp32 launch instead of the GEMM path's
            # pads / transposes / packs around a nearly empty MFMA tile
            dyc, xc = _f32c(dy2), xb.contiguous()
            dw = _grad_out(ctx.owners[0], (N, K))
            want_db = ctx.has_bias and ctx.needs_input_grad[2]
            db = _grad_out(ctx.owners[1], (N,)) if want_db else None
            _lib.check(_lib.load().syn_linear_wgrad_rows(dyc.data_ptr(), xc.data_ptr(), M, N, K, dw.data_ptr(), _lib.ptr(db), _lib.current_stream(dy.device)),
                       "syn_linear_wgrad_rows")
            if ctx.needs_input_grad[0]:
                dyb = dyc.to(torch.bfloat16)
                if K % 512 == 0 and N % 128 == 0 and w.is_contiguous():
                    dx = _gemm_packed(dyb, ctx.pack_t if ctx.pack_t is not None else _pack_t(w, K, N), K, N).reshape(ctx.in_shape)
                else:
                    dx = hip_matmul_nt(dyb, w.t()).reshape(ctx.in_shape)
            return dx, dw, db
        pair = (LINEAR_BWD_PAIR and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and K % 512 == 0 and N % 128 == 0
                and M % 128 == 0 and w.dtype is torch.float32 and w.is_contiguous() and xb.is_contiguous())
        part = None
        if LINEAR_BWD_PREP and M % 64 == 0 and N % 64 == 0 and dy2.dtype is torch.float32 and dy2.is_cuda:
            # one pass over dy: bf16 copy, bf16 transpose (weight gradient) and the bias gradient's per-64-row partial sums
            dy2 = dy2.contiguous()
            dyb = torch.empty(M, N, dtype=torch.bfloat16, device=dy.device)
            dybt = torch.empty(N, M, dtype=torch.bfloat16, device=dy.device)
            want_db = ctx.has_bias and ctx.needs_input_grad[2] and LINEAR_BWD_PREP > 1
            part = torch.empty(M // 64, N, dtype=torch.float32, device=dy.device) if want_db else None
            in_launch = want_db and LINEAR_BWD_PREP > 2 and N // 64 <= 1024
            if in_launch:
                db = torch.empty(N, dtype=torch.float32, device=dy.device)
            _lib.check(_lib.load().syn_linear_bwd_prep(dy2.data_ptr(), M, N, None, 0, None, dyb.data_ptr(), dybt.data_ptr(), _lib.ptr(part),
                                                       _lib.ptr(_counters(dy.device)) if in_launch else None, _lib.ptr(db) if in_launch else None,
                                                       _lib.current_stream(dy.device)), "syn_linear_bwd_prep")
            if want_db and not in_launch and not (pair and M <= 2048):
                db = part.sum(0)
        else:
            dyb = dy2.to(torch.bfloat16).contiguous()
        if pair and dybt is not None:
            # dy . W and dy^T . x - independent, half a chip each - as one launch, which also adds the bias gradient's partial sums up
            wt = ctx.pack_t if ctx.pack_t is not None else _pack_t(w, K, N)
            dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
            dw = _grad_out(ctx.owners[0], (N, K))
            sum_here = part is not None and db is None
            if sum_here:
                db = _grad_out(ctx.owners[1], (N,))
            _lib.check(_lib.load().syn_linear_pair(dyb.data_ptr(), wt.data_ptr(), M, K, N, dx.data_ptr(),
                                                   dybt.data_ptr(), (ctx.xt if ctx.xt is not None else _pack_t(xb, K, M)).data_ptr(), N, K, M, dw.data_ptr(),
                                                   _lib.ptr(part) if sum_here else None, M // 64, N, _lib.ptr(db) if sum_here else None,
                                                   _lib.current_stream(dy.device)), "syn_linear_pair")
            dx = dx.reshape(ctx.in_shape)
            if ctx.has_bias and ctx.needs_input_grad[2] and db is None:
                db = dy2.sum(0)
            return dx, dw.to(w.dtype), db
        if part is not None and db is None:
            db = part.sum(0)
        if ctx.needs_input_grad[0]:
            if K % 512 == 0 and N % 128 == 0 and w.dtype is torch.float32 and w.is_contiguous():
                wt = ctx.pack_t if ctx.pack_t is not None else _pack_t(w, K, N)          # W^T: the step's pack, or packed in place
                dx = _gemm_packed(dyb, wt, K, N).reshape(ctx.in_shape)                    # dy . W
            else:
                dx = hip_matmul_nt(dyb, w.t()).reshape(ctx.in_shape)
        if ctx.needs_input_grad[1]:
            if K % 512 == 0 and M % 128 == 0 and xb.is_contiguous():
                dw = _gemm_packed(dybt if dybt is not None else dyb.t().contiguous(), _pack_t(xb, K, M), K, M).to(w.dtype)   # dy^T . x, x^T packed in place
            else:
                dw = hip_matmul_nt(dyb.t(), xb.t()).to(w.dtype)                    # contraction over tokens
        if ctx.has_bias and ctx.needs_input_grad[2] and db is None:
            db = dy2.sum(0)
        return dx, dw, db


class EmbeddingFn(torch.autograd.Function):
    """nn.Embedding lookup (models/denoiser.py:72,147: the word ids in front of text_encoder_body) whose table gradient comes from
    `syn_embedding_wgrad` - one deterministic launch - instead of PyTorch-ROCm's embedding_dense_backward (a chain of ~15 sort /
    scan / segment launches, and the op that made the captured training step abort in the HIP runtime)."""

    @staticmethod
    def forward(ctx, ids, weight):
        ctx.save_for_backward(ids)
        ctx.wshape, ctx.wdtype = weight.shape, weight.dtype
        ctx.owner = weight
        return weight.detach()[ids]

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        V, D = ctx.wshape
        idc = ids.reshape(-1).to(torch.int64).contiguous()
        dyc = dy.reshape(-1, D).float().contiguous()
        dw = _grad_out(ctx.owner, (V, D))
        lib = _lib.load()
        first = True
        for lo in range(0, idc.numel(), 8192):                  # (the bench's 32 clips x 128 frames are one call)
            n = min(8192, idc.numel() - lo)
            tgt = dw if first else torch.empty_like(dw)
            _lib.check(lib.syn_embedding_wgrad(idc[lo:].data_ptr(), dyc[lo:].data_ptr(), n, V, D, tgt.data_ptr(), _lib.current_stream(dy.device)),
                       "syn_embedding_wgrad")
            if not first:
                dw += tgt
            first = False
        return None, dw.to(ctx.wdtype)


def _embed(module: nn.Embedding, ids):
    w = module.weight
    if (w.is_cuda and w.requires_grad and torch.is_grad_enabled() and module.padding_idx is None and module.max_norm is None
            and not _os.environ.get("SYN_TORCH_EMBEDDING_GRAD")):   # (set: PyTorch's op, to reproduce the captured step's abort)
        return EmbeddingFn.apply(ids, w)
    return module(ids)


import os as _os
SMALL_M_WGRAD = bool(int(_os.environ.get("SYN_SMALL_M_WGRAD", "1")))          # Linears with <= 64 input rows: weight / bias gradient on syn_linear_wgrad_rows
LINEAR_FWD_PACK = bool(int(_os.environ.get("SYN_LINEAR_FWD_PACK", "1")))   # x^T fragments for the weight gradient from the forward GEMM's launch
LINEAR_BWD_PAIR = bool(int(_os.environ.get("SYN_LINEAR_BWD_PAIR", "1")))   # a Linear's two backward GEMMs as one launch (syn_linear_pair)
LINEAR_BWD_PREP = int(_os.environ.get("SYN_LINEAR_BWD_PREP", "2"))    # 0: PyTorch cast / transpose / sum; 1: fused cast + transpose (syn_linear_bwd_prep);
                                                                      # 2: + per-64-row partial column sums from the same pass, the bias gradient = their (16-row) sum;
                                                                      # 3: the bias gradient itself from that launch (last block of a column block adds the partial
                                                                      #    sums) - correct (test_step_weight_packs_and_in_launch_bias_gradient) but 21 us per launch
                                                                      #    against 4 + 4: the agent-scope release in front of the arrival counter writes the XCD's L2 back.
                                                                      # (2 and 3 used to abort the captured bench-size step: that was PyTorch-ROCm's
                                                                      #  embedding_dense_backward inside the graph, not these - see EmbeddingFn and DESIGN.md 7)


def lin(x, module: nn.Linear):
    return HipLinearFn.apply(x, module.weight, module.bias)


# (the fp32 block ops - LayerNorm, attention, GELU - run on the hand-written kernels below; their PyTorch-op twins live in
# tests/test_gpu_kernels.py::test_training_block_ops_vs_torch_autograd, which checks them against each other to 1e-5)


def _grad_out(param, shape=None):
    """The tensor a parameter's gradient is written into.  Normally a fresh buffer.  When the parameter carries a bound gradient buffer
    (`bind_grad_buffers`: DDP's bucket view of it, inside `GraphedTrainStep`) and holds no gradient yet, a NEW tensor object over that
    buffer: autograd's AccumulateGrad adopts it without a copy, and DDP's reducer, finding the gradient already inside its bucket, skips its
    per-parameter copy-and-divide launch (169 launches of ~2 us per step; the division moves into the collective, `_avg_comm_hook`)."""
    shape = tuple(shape if shape is not None else param.shape)
    buf = getattr(param, "_syn_grad_buf", None) if param is not None else None
    if (buf is not None and param.grad is None and buf.dtype is torch.float32 and buf.is_contiguous() and buf.numel() == math.prod(shape)
            and torch.is_grad_enabled() is False and not getattr(param, "_syn_grad_handed", False)):
        # handed out ONCE per backward: a parameter used by two nodes of one backward (tied weights, a module applied twice) gets a fresh
        # tensor the second time, which autograd accumulates as usual (cleared by `_reset_handed` at the top of the next step)
        param._syn_grad_handed = True
        return buf.view(shape)
    dev = param.device if param is not None else None
    return torch.empty(shape, dtype=torch.float32, device=dev)


def _into_bound_buffers(grads, owners):
    """The small per-channel gradients of a block (BatchNorm gains / shifts, the convolution biases' zeros) come out of the kernels as rows of
    [3][C] tensors.  Where their parameters carry bound gradient buffers (DDP's bucket views, `_grad_out`) they are moved there in ONE
    multi-tensor launch per block and the bound tensors are returned in their place: the reducer, finding a gradient already inside its
    bucket, skips its own copy - which is a hipMemcpyAsync node of ~4 us per parameter in the captured step (48 of them: 0.2 ms, the whole
    difference between the DDP-wrapped and the plain step of round 5).  Without bound buffers nothing happens."""
    dst, src, at = [], [], []
    for i, (g, p) in enumerate(zip(grads, owners)):
        if g is None or p is None or getattr(p, "_syn_grad_buf", None) is None:
            continue
        t = _grad_out(p, g.shape)
        if t.data_ptr() == p._syn_grad_buf.data_ptr():        # (a fresh tensor otherwise: handed out already, or the parameter still holds a gradient)
            dst.append(t); src.append(g); at.append(i)
    if dst:
        torch._foreach_copy_(dst, src)
        for i, t in zip(at, dst):
            grads[i] = t


def bind_grad_buffers(model) -> int:
    """Make the CURRENT gradient tensors of the model's parameters the buffers their next gradients are written into (see `_grad_out`).
    Under `make_ddp(..., capturable=True)` those are views of the reducer's buckets once it has rebuilt them (after its second iteration).
    Only valid while every step starts from `zero_grad(set_to_none=True)` - `GraphedTrainStep` - since a bound buffer is overwritten, not
    accumulated into (a parameter that still holds a gradient is never given its bound buffer).  Returns the number of parameters bound."""
    n = 0
    for p in model.parameters():
        g = p.grad
        if g is not None and g.dtype is torch.float32 and g.is_contiguous() and g.shape == p.shape:
            p._syn_grad_buf = g
            n += 1
    return n


def direct_grad_report(model):
    """(gradients the last backward wrote straight into their bound buffers, parameters with a bound buffer, names of the others - those
    the DDP reducer still copies into its buckets, one memcpy node each in the captured step)."""
    bound = [(n, p) for n, p in model.named_parameters() if getattr(p, "_syn_grad_buf", None) is not None]
    rest = [n for n, p in bound if not getattr(p, "_syn_grad_handed", False)]
    return len(bound) - len(rest), len(bound), rest


def unbind_grad_buffers(model):
    for p in model.parameters():
        if hasattr(p, "_syn_grad_buf"):
            del p._syn_grad_buf
        if hasattr(p, "_syn_grad_handed"):
            del p._syn_grad_handed


def _reset_handed(model):
    """Start of a step: every bound gradient buffer may be handed out again (`_grad_out`)."""
    for p in model.parameters():
        if getattr(p, "_syn_grad_handed", False):
            p._syn_grad_handed = False


def _f32c(t):
    t = t.detach()
    if t.dtype is not torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


class HipLayerNormFn(torch.autograd.Function):
    """nn.LayerNorm(512, eps 1e-5): fp32 forward / backward kernels (syn_ln_fwd / syn_ln_bwd)."""

    @staticmethod
    def forward(ctx, x, g, b):
        engine._require_cuda(x, "LayerNorm input")
        xc, gc, bc = _f32c(x).view(-1, 512), _f32c(g), _f32c(b)
        rows = xc.shape[0]
        y = torch.empty_like(xc)
        mean, rstd = torch.empty(rows, device=x.device), torch.empty(rows, device=x.device)
        _lib.check(_lib.load().syn_ln_fwd(xc.data_ptr(), gc.data_ptr(), bc.data_ptr(), y.data_ptr(), None, mean.data_ptr(), rstd.data_ptr(),
                                          rows, _lib.current_stream(y.device)), "syn_ln_fwd")
        ctx.save_for_backward(xc, gc, mean, rstd)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        dx, dg, db = HipLayerNormFn._bwd(ctx, dy, None)
        return dx, dg, db

    @staticmethod
    def _bwd(ctx, dy, dres):
        xc, gc, mean, rstd = ctx.saved_tensors
        rows = xc.shape[0]
        dyc = _f32c(dy).view(-1, 512)
        add = None if dres is None else _f32c(dres).view(-1, 512)
        dx = torch.empty_like(xc)
        dg, db = torch.empty(512, device=dy.device), torch.empty(512, device=dy.device)
        scratch = torch.empty((rows + 15) // 16 * 1024, device=dy.device)
        _lib.check(_lib.load().syn_ln_bwd(dyc.data_ptr(), xc.data_ptr(), gc.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _lib.ptr(add),
                                          dx.data_ptr(), dg.data_ptr(), db.data_ptr(), scratch.data_ptr(), rows, _lib.current_stream(dx.device)),
                   "syn_ln_bwd")
        return dx.view(dy.shape), dg, db


class HipLnForkFn(torch.autograd.Function):
    """A pre-LN residual block's entry (transformer.py:195-198: x + f(norm(x))): returns (LayerNorm(x), x).  x feeds both the
    norm and the residual add, and autograd would add the two gradients with a copy and an in-place add per block half; here the
    residual path's gradient goes into the LayerNorm backward kernel as its addend - one gradient for x, no extra launch."""

    @staticmethod
    def forward(ctx, x, g, b):
        y = HipLayerNormFn.forward(ctx, x, g, b)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        return HipLayerNormFn._bwd(ctx, dy, dres)


class HipGeluFn(torch.autograd.Function):
    """nn.GELU() (exact erf form)."""

    @staticmethod
    def forward(ctx, x):
        engine._require_cuda(x, "GELU input")
        xc = _f32c(x)
        y = torch.empty_like(xc)
        _lib.check(_lib.load().syn_gelu_fwd(xc.data_ptr(), y.data_ptr(), None, xc.numel(), _lib.current_stream(y.device)), "syn_gelu_fwd")
        ctx.save_for_backward(xc)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, = ctx.saved_tensors
        dyc = _f32c(dy)
        dx = torch.empty_like(xc)
        _lib.check(_lib.load().syn_gelu_bwd(xc.data_ptr(), dyc.data_ptr(), dx.data_ptr(), xc.numel(), _lib.current_stream(dx.device)), "syn_gelu_bwd")
        return dx


class HipAttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(128)) v for 4 heads x 128 dims over 32 tokens, on the packed (B, 32, 1536) output of the qkv Linear
    (models/timm_transformer/transformer.py:83-104; no mask, attention dropout 0)."""

    @staticmethod
    def forward(ctx, qkv):
        engine._require_cuda(qkv, "attention input")
        q = _f32c(qkv)
        bs, T, _ = q.shape
        assert T == 32 and q.shape[2] == 1536, q.shape
        o = torch.empty(bs, T, 512, device=q.device)
        _lib.check(_lib.load().syn_attn_fwd(q.data_ptr(), o.data_ptr(), None, bs, _lib.current_stream(o.device)), "syn_attn_fwd")
        ctx.save_for_backward(q)
        return o

    @staticmethod
    def backward(ctx, do):
        q, = ctx.saved_tensors
        doc = _f32c(do)
        dqkv = torch.empty_like(q)
        _lib.check(_lib.load().syn_attn_bwd(q.data_ptr(), doc.data_ptr(), dqkv.data_ptr(), q.shape[0], _lib.current_stream(dqkv.device)), "syn_attn_bwd")
        return dqkv


# ---- a pre-LN residual branch as ONE autograd node ---------------------------------------------------------------------------
# x + drop_path(attn(norm1(x))) and x + drop_path(mlp(norm2(x))) (timm_transformer/transformer.py:195-198).  Built from the ops above
# these are 7 / 6 autograd nodes per branch with PyTorch glue between them - a bf16 cast in front of every Linear, addcmul for the
# DropPath factor and its mul in the backward, a sum for every bias gradient, copies where a gradient is not contiguous: ~25 launches
# of 3-5 us per branch and direction around kernels that run 5-15 us.  As one node the branch is the kernels and nothing else:
#   forward   LayerNorm -> bf16 rows | GEMM (+ x^T pack; fc1: + GELU -> bf16) | attention -> bf16 | GEMM with `x + factor * (.)` in its epilogue
#   backward  prep (factor * dy -> bf16, bf16^T, bias partials) | GEMM pair (+ bias sum) | attention backward | prep (fc1: x GELU') | GEMM pair |
#             LayerNorm backward with dy as its addend (the residual path)
# The fp32 LayerNorm / attention / GELU outputs are never written: the Linear behind each takes bf16 operands and nothing else reads them.
BLOCK_FUSED = bool(int(_os.environ.get("SYN_TRAIN_BLOCK_FUSED", "1")))
GELU_FUSED = int(_os.environ.get("SYN_TRAIN_GELU_FUSED", "1"))         # bit 0: GELU in fc1's epilogue (-0.1 ms per step); bit 1: GELU' in the backward's
                                                                       # operand pass (same-box A/B: no gain - erf + exp in the transposing pass cost what the launch did)


def engine_has_xcd_groups(device) -> bool:
    """The tile-split kernels deal the workgroups of an XCD by hardware id: 256 CUs in 8 XCDs of 32 (MI355X)."""
    return torch.cuda.get_device_properties(device).multi_processor_count == 256


def _fused_ok(M, *layers) -> bool:
    """Every Linear of the branch has its step packs (W and W^T fragments) and the GEMM pair's shape constraints hold."""
    if not (BLOCK_FUSED and M % 128 == 0):
        return False
    for l in layers:
        pk, tr = _lookup_packs(l.weight)
        if pk is None or tr is None or l.weight.shape[1] % 512 or l.weight.shape[0] % 128:
            return False
        if l.bias is not None and not (l.bias.dtype is torch.float32 and l.bias.is_contiguous()):
            return False
    return True


def _lin_fwd(xb, w, b, res=None, scale=None, rows_per_scale=1, gelu_out=None):
    """bf16 rows [M][K] -> fp32 [M][N] = x W^T + b, or res + scale[row // rows_per_scale] * (x W^T + b); also the x^T fragments the
    weight-gradient GEMM will take (packed by the same launch while M <= 2048)."""
    pk, wt = _lookup_packs(w)
    M, K = xb.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=xb.device)
    xt = torch.empty(K * M * 2, dtype=torch.uint8, device=xb.device)
    lib, st = _lib.load(), _lib.current_stream(xb.device)
    if gelu_out is not None:                   # fc1: y stays fp32 for the backward, gelu_out receives bf16(GELU(y)) - fc2's operand
        _lib.check(lib.syn_linear_gelu(xb.data_ptr(), pk.data_ptr(), _lib.ptr(b), M, N, K, y.data_ptr(), gelu_out.data_ptr(), xt.data_ptr(), st),
                   "syn_linear_gelu")
    elif res is not None:
        _lib.check(lib.syn_linear_res(xb.data_ptr(), pk.data_ptr(), _lib.ptr(b), res.data_ptr(), _lib.ptr(scale), rows_per_scale, M, N, K,
                                      y.data_ptr(), xt.data_ptr(), st), "syn_linear_res")
    else:
        _lib.check(lib.syn_linear_and_pack(xb.data_ptr(), pk.data_ptr(), _lib.ptr(b), M, N, K, y.data_ptr(), xt.data_ptr(), st), "syn_linear_and_pack")
    return y, (xt, wt)          # what the backward takes: x^T and W^T fragments as of THIS forward (the step's pack cache may have moved on by then)


def _lin_bwd(dy2, packs, w, has_bias, scale=None, rows_per_scale=1, gelu_pre=None, owners=(None, None)):
    """fp32 dy [M][N] (contiguous) -> dx [M][K], dW [N][K], db [N] of y = x W^T + b; dy is first multiplied by its rows' factors
    (scale) or by GELU'(gelu_pre) (dy taken behind a GELU of y).  packs: (x^T fragments, W^T fragments) from `_lin_fwd`."""
    xt, wt = packs
    M, N = dy2.shape
    K = w.shape[1]
    dev = dy2.device
    lib, st = _lib.load(), _lib.current_stream(dev)
    dyb = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    dybt = torch.empty(N, M, dtype=torch.bfloat16, device=dev)
    part = torch.empty(M // 64, N, dtype=torch.float32, device=dev) if has_bias else None
    _lib.check(lib.syn_linear_bwd_prep(dy2.data_ptr(), M, N, _lib.ptr(scale), rows_per_scale, _lib.ptr(gelu_pre), dyb.data_ptr(), dybt.data_ptr(), _lib.ptr(part),
                                       None, None, st), "syn_linear_bwd_prep")
    dx = torch.empty(M, K, dtype=torch.float32, device=dev)
    dw = _grad_out(owners[0], (N, K)) if owners[0] is not None else torch.empty(N, K, dtype=torch.float32, device=dev)
    in_pair = has_bias and M <= 2048
    db = (_grad_out(owners[1], (N,)) if owners[1] is not None else torch.empty(N, dtype=torch.float32, device=dev)) if in_pair else None
    _lib.check(lib.syn_linear_pair(dyb.data_ptr(), wt.data_ptr(), M, K, N, dx.data_ptr(), dybt.data_ptr(), xt.data_ptr(), N, K, M, dw.data_ptr(),
                                   _lib.ptr(part) if in_pair else None, M // 64, N, _lib.ptr(db), st), "syn_linear_pair")
    if has_bias and db is None:
        db = part.sum(0)
    return dx, dw, db


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
    """h + factor * proj(attention(qkv(LayerNorm(h)))) for h (B, 32, 512); factor (B, 1, 1) or None."""

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
        dh, dg, db, dwq, dbq, dwp, dbp = _attn_branch_bwd(_f32c(dout).view(B * T, 512), hc, gc, mean, rstd, qkv, wqkv, wproj, factor, ctx.packs, ctx.bias,
                                                          ctx.owners)
        return dh.view(B, T, 512), dg, db, dwq, dbq, dwp, dbp, None


def _attn_branch_bwd(d, hc, gc, mean, rstd, qkv, wqkv, wproj, factor, packs, bias, owners):
    """Backward of h + factor * proj(attention(qkv(LayerNorm(h)))) from d = the gradient at its output, (B * 32, 512) fp32 contiguous."""
    xt1, xt2 = packs
    B, T = hc.shape[0], hc.shape[1]
    og, ob, owq, obq, owp, obp = owners
    do, dwp, dbp = _lin_bwd(d, xt2, wproj, bias[1], factor, T, owners=(owp, obp))
    dqkv = torch.empty_like(qkv)
    _lib.check(_lib.load().syn_attn_bwd(qkv.data_ptr(), do.data_ptr(), dqkv.data_ptr(), B, _lib.current_stream(d.device)), "syn_attn_bwd")
    dz, dwq, dbq = _lin_bwd(dqkv.view(B * T, -1), xt1, wqkv, bias[0], owners=(owq, obq))
    dh, dg, db = _ln_bwd_rows(dz, hc, gc, mean, rstd, d, owners=(og, ob))
    return dh, dg, db, dwq, dbq, dwp, dbp


class MlpBranchFn(torch.autograd.Function):
    """h + factor * fc2(GELU(fc1(LayerNorm(h))))."""

    @staticmethod
    def forward(ctx, h, g, b, w1, b1, w2, b2, factor):
        hc, gc, bc = _f32c(h), _f32c(g), _f32c(b)
        B, T, _ = hc.shape
        zb, mean, rstd = _ln_rows_bf16(hc, gc, bc)
        ab = torch.empty(B * T, w1.shape[0], dtype=torch.bfloat16, device=hc.device)
        if GELU_FUSED & 1:
            pre, xt1 = _lin_fwd(zb, w1, b1, gelu_out=ab)                 # GELU in fc1's epilogue
        else:
            pre, xt1 = _lin_fwd(zb, w1, b1)
            _lib.check(_lib.load().syn_gelu_fwd(pre.data_ptr(), None, ab.data_ptr(), pre.numel(), _lib.current_stream(hc.device)), "syn_gelu_fwd")
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
        dh, dg, db, dw1, db1, dw2, db2 = _mlp_branch_bwd(_f32c(dout).view(B * T, 512), hc, gc, mean, rstd, pre, w1, w2, factor, ctx.packs, ctx.bias, ctx.owners)
        return dh.view(B, T, 512), dg, db, dw1, db1, dw2, db2, None


def _mlp_branch_bwd(d, hc, gc, mean, rstd, pre, w1, w2, factor, packs, bias, owners):
    """Backward of h + factor * fc2(GELU(fc1(LayerNorm(h)))) from d = the gradient at its output, (B * 32, 512) fp32 contiguous."""
    xt1, xt2 = packs
    T = hc.shape[1]
    og, ob, ow1, ob1, ow2, ob2 = owners
    da, dw2, db2 = _lin_bwd(d, xt2, w2, bias[1], factor, T, owners=(ow2, ob2))
    if GELU_FUSED & 2:
        dz, dw1, db1 = _lin_bwd(da, xt1, w1, bias[0], gelu_pre=pre, owners=(ow1, ob1))    # GELU' in the operand pass of fc1's backward
    else:
        dpre = torch.empty_like(pre)
        _lib.check(_lib.load().syn_gelu_bwd(pre.data_ptr(), da.data_ptr(), dpre.data_ptr(), pre.numel(), _lib.current_stream(d.device)), "syn_gelu_bwd")
        dz, dw1, db1 = _lin_bwd(dpre, xt1, w1, bias[0], owners=(ow1, ob1))
    dh, dg, db = _ln_bwd_rows(dz, hc, gc, mean, rstd, d, owners=(og, ob))
    return dh, dg, db, dw1, db1, dw2, db2


# ---- the eight blocks' forward as ONE persistent launch (round 5) ------------------------------------------------------------------------------
# `syn_train_stack_fwd` (csrc/syn_stack_train.inc): the sampling path's whole-step kernel in its tile-split mode, writing what the branch
# backwards above take.  56 launches at their latency floor (0.55 ms at 32 clips) become one; the backward is the per-branch chain unchanged.
PACK_BLOCKS_LATE = bool(int(_os.environ.get("SYN_TRAIN_PACK_BLOCKS_LATE", "1")))     # (A/B: 0 = all Linears packed at the top of the forward)
STACK_FUSED = bool(int(_os.environ.get("SYN_TRAIN_STACK_FUSED", "1")))
STACK_BWD_PIECES = int(_os.environ.get("SYN_TRAIN_STACK_BWD_PIECES", "1"))        # the backward chain in this many launches, the finished pieces' weight-gradient GEMMs on a
                                                                                 # second stream.  Measured (one box, captured step): 1 -> 5.25 ms, 2 -> 5.28, 4 -> 5.27, 8 -> 5.42: the replayed
                                                                                 # graph does not run the GEMMs beside the chain (as the side-stream weight gradients of round 4): default 1
_side_streams = {}


def _side_stream(device):
    s = _side_streams.get(device)
    if s is None:
        s = _side_streams[device] = torch.cuda.Stream(device=device)
    return s


STACK_BWD_PREFETCH = bool(int(_os.environ.get("SYN_TRAIN_STACK_BWD_PREFETCH", "1")))
STACK_BWD_FUSED = bool(int(_os.environ.get("SYN_TRAIN_STACK_BWD_FUSED", "1")))   # (A/B: 0 = the per-branch backward chain behind the persistent forward)       # (A/B: 0 = one autograd node per residual branch, `AttnBranchFn` / `MlpBranchFn`)
_stack_ws = {}


def _stack_workspace(device, n_seq):
    ws = _stack_ws.get(device)
    if ws is None or ws[1].shape[0] < n_seq:
        ws = _stack_ws[device] = (torch.zeros(320, dtype=torch.int32, device=device), torch.empty(max(n_seq, 64), 8, 32 * 512, dtype=torch.float32, device=device))
    return ws


class StackFn(torch.autograd.Function):
    """mytimmblocks[0..7] on h (B, 32, 512), B <= 64; dp: DropPath factors (16, B, 1, 1) or None; params: per block norm1.weight, norm1.bias,
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
            fr = [_lookup_packs(w) for w in (wq, wp, w1, w2)]
            if any(f[0] is None or f[1] is None for f in fr):
                raise _lib.SynHipError("StackFn: every Linear of the blocks needs its step packs (training.WeightPacks)")
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
        ctx.saves, ctx.packs, ctx.dp, ctx.params, ctx.keep, ctx.fwd = saves, packs, dpc, params, keep, a
        return out

    @staticmethod
    def _backward_persistent(ctx, dout):
        """The data-gradient chain as one launch (`syn_train_stack_bwd`), then the 32 weight-gradient GEMMs four per launch (`syn_train_stack_wgrad`)."""
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
        for l in range(len(params) // NP):
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
        if STACK_BWD_PREFETCH and _packs_blocks is not None:
            ctx_pf = _packs_blocks.prefetch(True)              # (the transposed sets were packed in front of the forward; 150 MB of saved tensors went by since)
        # The chain occupies half the chip (32 sequences x 4 workgroups on 256 CUs) and the weight-gradient GEMMs of a block need nothing but that block's
        # piece of it: the chain goes out in STACK_BWD_PIECES pieces on this stream, and the GEMMs of a finished piece on a second stream beside the next
        # piece (a fork / join in the captured graph).
        n_blk = len(params) // NP
        pieces = max(1, min(STACK_BWD_PIECES, n_blk))
        per = -(-n_blk // pieces)
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if pieces > 1 else main
        cur, hi = d, n_blk - 1
        while hi >= 0:
            lo = max(0, hi - per + 1)
            nxt = dh_in if lo == 0 else torch.empty(B, T, 512, dtype=torch.float32, device=dev)
            g.dh_out, g.dh_in, g.first_block, g.last_block = cur.data_ptr(), nxt.data_ptr(), hi, lo
            _lib.check(lib.syn_train_stack_bwd(C.byref(g), main.cuda_stream), "syn_train_stack_bwd")
            if side is not main:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
            _lib.check(lib.syn_train_stack_wgrad(C.byref(g), side.cuda_stream), "syn_train_stack_wgrad")
            keep.append(cur)
            cur, hi = nxt, lo - 1
        if side is not main:
            main.wait_stream(side)
        del ten, keep                             # (stream-ordered allocator: the launches above are enqueued, later work on this stream comes after them)
        return (dh_in, None, *grads)

    @staticmethod
    def backward(ctx, dout):
        if STACK_BWD_FUSED and dout.shape[0] % 4 == 0:
            return StackFn._backward_persistent(ctx, dout)
        params, NP = ctx.params, StackFn.NP
        n = len(params) // NP
        d = _f32c(dout)
        B, T, _ = d.shape
        d = d.view(B * T, 512)
        grads = [None] * len(params)
        for l in reversed(range(n)):
            g1, b1, wq, wp, bp, g2, b2, w1, bb1, w2, bb2 = params[l * NP:(l + 1) * NP]
            sv, (tq, tp_, t1, t2) = ctx.saves[l], ctx.packs[l]
            fa = None if ctx.dp is None else ctx.dp[2 * l]
            fm = None if ctx.dp is None else ctx.dp[2 * l + 1]
            d, dg2, db2, dw1, dbb1, dw2, dbb2 = _mlp_branch_bwd(d, sv["h_mlp"], _f32c(g2), sv["mean_mlp"], sv["rstd_mlp"], sv["pre"], w1, w2, fm,
                                                                  ((sv["xt_ln2"], t1), (sv["xt_gelu"], t2)), (True, True), (g2, b2, w1, bb1, w2, bb2))
            d = d.view(B * T, 512)
            d, dg1, db1, dwq, _, dwp, dbp = _attn_branch_bwd(d, sv["h_attn"], _f32c(g1), sv["mean_attn"], sv["rstd_attn"], sv["qkv"], wq, wp, fa,
                                                             ((sv["xt_ln1"], tq), (sv["xt_attn"], tp_)), (False, True), (g1, b1, wq, None, wp, bp))
            d = d.view(B * T, 512)
            grads[l * NP:(l + 1) * NP] = [dg1, db1, dwq, dwp, dbp, dg2, db2, dw1, dbb1, dw2, dbb2]
        return (d.view(B, T, 512), None, *grads)


def _stack_ok(m, bs, T) -> bool:
    if not (STACK_FUSED and torch.is_grad_enabled() and T == 32 and 1 <= bs <= 64 and len(m.mytimmblocks) == 8):
        return False
    for blk in m.mytimmblocks:
        if blk.attn.qkv.bias is not None or blk.attn.proj.bias is None or blk.mlp.fc1.bias is None or blk.mlp.fc2.bias is None:
            return False
        if not _fused_ok(bs * T, blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2):
            return False
        if (tuple(blk.attn.qkv.weight.shape), tuple(blk.mlp.fc1.weight.shape)) != ((1536, 512), (1024, 512)):
            return False
    return True


class MaskedSmoothL1Fn(torch.autograd.Function):
    """`masked_l2` of the reference's training_losses (gaussian_diffusion.py:202-215: SmoothL1 x mask, summed per sample, / (sum(mask) x C)) with its
    gradient from the same launch (`syn_masked_smooth_l1`): (target, out (B, C, 1, T), mask (B, 1, 1, T) bool) -> (B,)."""

    @staticmethod
    def forward(ctx, target, out, mask):
        B = out.shape[0]
        T = out.shape[-1]
        tc, oc = _f32c(target), _f32c(out)
        mk = mask.detach().reshape(B, T).to(torch.uint8).contiguous()
        loss = torch.empty(B, dtype=torch.float32, device=out.device)
        dout = torch.empty_like(oc)
        _lib.check(_lib.load().syn_masked_smooth_l1(tc.data_ptr(), oc.data_ptr(), mk.data_ptr(), B, oc.numel() // B, T, loss.data_ptr(), dout.data_ptr(),
                                                    _lib.current_stream(out.device)), "syn_masked_smooth_l1")
        ctx.save_for_backward(dout)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dout,) = ctx.saved_tensors
        return None, dout * g.reshape(-1, *([1] * (dout.dim() - 1))), None


LOSS_FUSED = bool(int(_os.environ.get("SYN_TRAIN_LOSS_FUSED", "1")))


def masked_smooth_l1(target, out, mask):
    """The fused loss when it applies (device fp32 tensors, one mask row per sample, no gradient asked for the target), else None."""
    if not (LOSS_FUSED and out.is_cuda and out.dim() == 4 and out.dtype is torch.float32 and target.shape == out.shape and not target.requires_grad
            and target.is_cuda and target.device == out.device
            and torch.is_tensor(mask) and mask.is_cuda and mask.device == out.device and mask.dtype is torch.bool and tuple(mask.shape) == (out.shape[0], 1, 1, out.shape[-1])
            and out.shape[-1] <= 64 and (out.numel() // out.shape[0]) % 4 == 0):
        return None
    return MaskedSmoothL1Fn.apply(target, out, mask)


class RotaryFn(torch.autograd.Function):
    """The rotary embedding of the hidden state as one launch each way (`syn_rotary`); the backward is the transposed rotation."""

    @staticmethod
    def forward(ctx, h, cs, sn):
        hc = _f32c(h)
        y = torch.empty_like(hc)
        _lib.check(_lib.load().syn_rotary(hc.data_ptr(), cs.data_ptr(), sn.data_ptr(), hc.shape[0], 0, y.data_ptr(), _lib.current_stream(hc.device)), "syn_rotary")
        ctx.save_for_backward(cs, sn)
        return y

    @staticmethod
    def backward(ctx, dy):
        cs, sn = ctx.saved_tensors
        d = _f32c(dy)
        dx = torch.empty_like(d)
        _lib.check(_lib.load().syn_rotary(d.data_ptr(), cs.data_ptr(), sn.data_ptr(), d.shape[0], 1, dx.data_ptr(), _lib.current_stream(d.device)), "syn_rotary")
        return dx, None, None


ROTARY_FUSED = bool(int(_os.environ.get("SYN_TRAIN_ROTARY_FUSED", "1")))


def _rotary(m, h):
    """models/denoiser.py:178-186,324-343 on (B, T, 512)."""
    B, T, _ = h.shape
    if ROTARY_FUSED and h.is_cuda and T == 32 and h.shape[2] == 512:
        inv = m.rel_pos.inv_freq
        key = (inv.data_ptr(), inv._version, inv.device)
        tab = m.__dict__.get("_syn_rotary_tables")
        if tab is None or tab[0] != key:                     # (T, 32) tables: cos / sin(position x inv_freq), fp32 like the reference's buffer
            with torch.no_grad():
                fr = torch.einsum("i,j->ij", torch.arange(T, device=h.device).type_as(inv), inv)
                tab = m.__dict__["_syn_rotary_tables"] = (key, fr.cos().contiguous(), fr.sin().contiguous())
        return RotaryFn.apply(h, tab[1], tab[2])
    g = h.view(B, T, 8, -1).permute(0, 2, 1, 3).reshape(B * 8, T, -1)
    pos = torch.arange(T, device=h.device).type_as(m.rel_pos.inv_freq)
    fr = torch.einsum("i,j->ij", pos, m.rel_pos.inv_freq)
    fr = torch.cat((fr, fr), dim=-1)
    half = g.shape[-1] // 2
    g = g * fr.cos() + torch.cat((-g[..., half:], g[..., :half]), dim=-1) * fr.sin()
    return g.reshape(B, 8, T, -1).permute(0, 2, 1, 3).reshape(B, T, -1)


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


CONV_PACKS = bool(int(_os.environ.get("SYN_CONV_PACKS", "1")))        # (A/B: 0 = one pack launch per use)
_conv_packs: "ConvPacks | None" = None


def _lookup_conv_pack(w, transposed):
    return _conv_packs.lookup(w, transposed) if (_conv_packs is not None and WEIGHT_PACKS and CONV_PACKS) else None


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
    gradient) the library's switch is never touched - the product path makes no `syn_debug_*` call.
    An A/B run (SYN_CONV_TERMS set to something else) selects the role's mask in front of every convolution launch."""
    if CONV_TERMS != (3, 3, 1):
        _lib.load().syn_debug_conv_terms(CONV_TERMS[role])


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
        pk = _lookup_conv_pack(w, transposed)
        if pk is not None:
            whi, wlo = pk
        else:
            kts = -(-15 // stride) * stride
            whi = torch.empty(cout * kts * cin * 2, dtype=torch.uint8, device=x.device)
            wlo = torch.empty_like(whi)
            _lib.check(lib.syn_conv1d_pack_split(wc.data_ptr(), co_w, ci_w, stride, int(transposed), whi.data_ptr(), wlo.data_ptr(),
                                                 _lib.current_stream(x.device)), "syn_conv1d_pack_split")
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
                # positions x cin channels (three 128-column launches of the forward kernel)
                lib = _lib.load()
                n, _, _, l_in = x.shape
                pk = _lookup_conv_pack(w, True)
                if pk is not None:
                    whi, wlo = pk
                else:
                    wc = w.detach().float().contiguous()
                    nb = lib.syn_conv1d_pack_bytes(cout, cin, stride, 1)
                    whi = torch.empty(nb, dtype=torch.uint8, device=x.device)
                    wlo = torch.empty_like(whi)
                    _lib.check(lib.syn_conv1d_pack_split(wc.data_ptr(), cout, cin, stride, 1, whi.data_ptr(), wlo.data_ptr(),
                                                         _lib.current_stream(x.device)), "syn_conv1d_pack_split")
                gx = torch.empty(n, cin, 1, l_in, device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
                _conv_terms(1)
                _lib.check(lib.syn_conv1d_train_dgrad_strided(gy.data_ptr(), n, l_in, cin, stride, cout, whi.data_ptr(), wlo.data_ptr(),
                                                              gx.data_ptr(), _lib.current_stream(x.device)), "syn_conv1d_train_dgrad_strided")
            else:
                raise _unsupported_conv("data gradient", cin, stride, pad, cout)
        if ctx.needs_input_grad[1]:
            if (cin, stride, cout) in ConvSplitFn.SUPPORTED and ((stride == 1 and pad == 7) or (stride > 1 and pad == 0)):
                # contraction over positions of two channels-last tensors: transposed through LDS inside the kernel (a strided layer
                # as a stride-1 one over rows of stride x cin channels, a wave per 16 of them)
                lib = _lib.load()
                n, _, _, l = x.shape
                l_out, kts = gy.shape[-1], -(-15 // stride) * stride
                ws = torch.empty(lib.syn_conv1d_wgrad_shares(n, l_out, stride * cin) * cout * kts * cin, device=x.device, dtype=torch.float32)
                gw = _grad_out(ctx.owner, (cout, cin, 1, 15)) if ctx.owner is not None else torch.empty(cout, cin, 1, 15, device=x.device, dtype=torch.float32)
                _conv_terms(2)
                _lib.check(lib.syn_conv1d_train_wgrad(x.data_ptr(), gy.data_ptr(), n, l, cin, stride, pad, cout, ws.data_ptr(), gw.data_ptr(),
                                                      _lib.current_stream(x.device)), "syn_conv1d_train_wgrad")
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
WAV_BLOCK_FUSED = bool(int(_os.environ.get("SYN_TRAIN_WAV_BLOCK_FUSED", "1")))     # (A/B: 0 = the per-convolution nodes above)
FIRST_PAIR = bool(int(_os.environ.get("SYN_TRAIN_FIRST_PAIR", "1")))               # block 0: conv1 + shortcut convolution as one launch (A/B: 0)
FIRST_WGRAD_BN = bool(int(_os.environ.get("SYN_TRAIN_FIRST_WGRAD_BN", "1")))       # block 0: bn1's backward apply folded into conv1's weight gradient (A/B: 0)


def _rows3(t):
    """(N, C, 1, L) channels_last fp32 -> the same memory as (N, L, C) contiguous."""
    t = t.detach()
    if t.dtype is not torch.float32:
        t = t.float()
    return t.contiguous(memory_format=torch.channels_last).permute(0, 3, 1, 2).reshape(t.shape[0], t.shape[3], t.shape[1])


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
    return y, part, chunks


def _wb_pack(conv, transposed):
    """hi / lo fragment sets of a Conv1d(k 15) weight: the step's pack, or packed on the spot."""
    pk = _lookup_conv_pack(conv.weight, transposed)
    if pk is not None:
        return pk
    lib = _lib.load()
    w = conv.weight.detach()
    cout, cin, stride = conv.out_channels, conv.in_channels, conv.stride[0]
    nb = lib.syn_conv1d_pack_bytes(cout, cin, stride, int(transposed))
    whi = torch.empty(nb, dtype=torch.uint8, device=w.device)
    wlo = torch.empty_like(whi)
    _lib.check(lib.syn_conv1d_pack_split(w.data_ptr(), cout, cin, stride, int(transposed), whi.data_ptr(), wlo.data_ptr(), _lib.current_stream(w.device)),
               "syn_conv1d_pack_split")
    return whi, wlo


def _wb_finalize(part, chunks, rows, bn, conv_bias):
    c = bn.num_features
    stats = torch.empty(2, c, device=part.device, dtype=torch.float32)
    aff = torch.empty(2, c, device=part.device, dtype=torch.float32)
    cb = None if conv_bias is None else conv_bias.detach()
    _lib.check(_lib.load().syn_bn_finalize(part.data_ptr(), chunks, rows, c, bn.weight.detach().data_ptr(), bn.bias.detach().data_ptr(), float(bn.eps),
                                           float(bn.momentum), _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var), _lib.ptr(cb), stats.data_ptr(),
                                           aff.data_ptr(), _lib.current_stream(part.device)), "syn_bn_finalize")
    if bn.num_batches_tracked is not None:
        _tracked.append(bn.num_batches_tracked)
    return stats, aff


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
    return dx


def _wb_wgrad(x3, dy3, conv, first, in_aff=None, in_act=0):
    """Weight gradient (Cout, Cin, 15) of `conv` from its input x (N, L_in, Cin) and dy (N, L_out, Cout)."""
    lib, dev = _lib.load(), x3.device
    n, l_in, cin = x3.shape
    stride, pad, cout = conv.stride[0], conv.padding[0], conv.out_channels
    l_out = dy3.shape[1]
    gw = _grad_out(conv.weight, (cout, cin, 15))
    st = _lib.current_stream(dev)
    if first:
        ws = torch.empty(lib.syn_conv1d_first_parts(n, l_out) * 64 * cin * 15, device=dev, dtype=torch.float32)
        _lib.check(lib.syn_conv1d_first_wgrad(x3.data_ptr(), dy3.data_ptr(), n, l_in, cin, stride, pad, ws.data_ptr(), gw.data_ptr(), st),
                   "syn_conv1d_first_wgrad")
        return gw
    kts = -(-15 // stride) * stride
    ws = torch.empty(lib.syn_conv1d_wgrad_shares(n, l_out, stride * cin) * cout * kts * cin, device=dev, dtype=torch.float32)
    _conv_terms(2)
    if in_aff is None:
        _lib.check(lib.syn_conv1d_train_wgrad(x3.data_ptr(), dy3.data_ptr(), n, l_in, cin, stride, pad, cout, ws.data_ptr(), gw.data_ptr(), st),
                   "syn_conv1d_train_wgrad")
    else:
        _lib.check(lib.syn_conv1d_train_wgrad_norm(x3.data_ptr(), dy3.data_ptr(), n, l_in, cin, stride, pad, cout, in_aff.data_ptr(), int(in_act),
                                                   ws.data_ptr(), gw.data_ptr(), st), "syn_conv1d_train_wgrad_norm")
    return gw


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
        if first and ds and FIRST_PAIR:
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
        else:
            y1, p1, c1 = _wb_conv_fwd(x3, blk.conv1, first)
            if ds:
                ysc, ps, cs = _wb_conv_fwd(x3, blk.downsample[0], first)
        rows = n * y1.shape[1]
        st1, af1 = _wb_finalize(p1, c1, rows, blk.bn1, blk.conv1.bias)
        if ds:
            sts, afs = _wb_finalize(ps, cs, rows, blk.downsample[1], blk.downsample[0].bias)
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
        dy2, dsh = torch.empty_like(y2), torch.empty_like(y2)
        short = ysc if ds else x3
        _lib.check(lib.syn_bn_block_bwd(d3.data_ptr(), y2.data_ptr(), short.data_ptr(), st2.data_ptr(), af2.data_ptr(), _lib.ptr(sts), _lib.ptr(afs),
                                        rows, c, 1, ws.data_ptr(), dgb2.data_ptr(), _lib.ptr(dgbs), dy2.data_ptr(), dsh.data_ptr(),
                                        _lib.current_stream(dev)), "syn_bn_block_bwd")
        # conv2: data gradient to z1 = act(bn1(y1)), weight gradient with z1 recomputed from y1 while its rows are staged
        dz1 = _wb_dgrad(dy2, blk.conv2, l1)
        gw2 = _wb_wgrad(y1, dy2, blk.conv2, False, in_aff=af1, in_act=1)
        # bn1 + activation (no shortcut entered it: the sign comes from y1)
        ws1 = torch.empty(2 * lib.syn_bn_chunks(rows) * c, device=dev, dtype=torch.float32)
        dgb1 = torch.empty(3, c, device=dev, dtype=torch.float32)
        g1, b1 = blk.bn1.weight.detach(), blk.bn1.bias.detach()
        if first and FIRST_WGRAD_BN and blk.conv1.stride[0] == 5:
            # block 0: nothing but conv1's weight gradient reads dy1 (the waveform takes no gradient) - it forms dy1 itself from (dz1, y1)
            _lib.check(lib.syn_bn_bwd_stats(dz1.data_ptr(), None, y1.data_ptr(), st1.data_ptr(), g1.data_ptr(), b1.data_ptr(), rows, c, 1, ws1.data_ptr(),
                                            dgb1.data_ptr(), _lib.current_stream(dev)), "syn_bn_bwd_stats")
            cv = blk.conv1
            nn_, l_in, cin = x3.shape
            wsg = torch.empty(lib.syn_conv1d_first_parts(nn_, l1) * 64 * cin * 15, device=dev, dtype=torch.float32)
            gw1 = _grad_out(cv.weight, (64, cin, 15))
            _lib.check(lib.syn_conv1d_first_wgrad_bn(x3.data_ptr(), dz1.data_ptr(), y1.data_ptr(), st1.data_ptr(), af1.data_ptr(), dgb1.data_ptr(), 1,
                                                     nn_, l_in, cin, cv.stride[0], cv.padding[0], wsg.data_ptr(), gw1.data_ptr(), _lib.current_stream(dev)),
                       "syn_conv1d_first_wgrad_bn")
            dy1 = None
        else:
            dy1 = torch.empty_like(y1)
            _lib.check(lib.syn_bn_act_bwd(dz1.data_ptr(), None, y1.data_ptr(), st1.data_ptr(), g1.data_ptr(), b1.data_ptr(), rows, c, 1, ws1.data_ptr(),
                                          dgb1.data_ptr(), dy1.data_ptr(), None, _lib.current_stream(dev)), "syn_bn_act_bwd")
            gw1 = _wb_wgrad(x3, dy1, blk.conv1, first)
        dx = None
        if not first and ctx.needs_input_grad[0]:
            # what reaches the block's input, written once: conv1^T dy1 + (shortcut^T dy_sc | the gradient along the identity shortcut)
            dx = (_wb_dgrad(dy1, blk.conv1, x3.shape[1], dy3b=dsh, conv_b=blk.downsample[0]) if ds
                  else _wb_dgrad(dy1, blk.conv1, x3.shape[1], residual=dsh))
        grads = [gw1, dgb1[2] if blk.conv1.bias is not None else None, dgb1[0], dgb1[1],
                 gw2, dgb2[2] if blk.conv2.bias is not None else None, dgb2[0], dgb2[1]]
        owners = [None, blk.conv1.bias, blk.bn1.weight, blk.bn1.bias, None, blk.conv2.bias, blk.bn2.weight, blk.bn2.bias]
        if ds:
            gws = _wb_wgrad(x3, dsh, blk.downsample[0], first)
            grads += [gws, dgbs[2] if blk.downsample[0].bias is not None else None, dgbs[0], dgbs[1]]
            owners += [None, blk.downsample[0].bias, blk.downsample[1].weight, blk.downsample[1].bias]
        _into_bound_buffers(grads, owners)
        return (None if dx is None else _as4(dx), None, None, *grads)


def _wav_block_fused_ok(blk, x, first) -> bool:
    if not (WAV_BLOCK_FUSED and blk.training and torch.is_grad_enabled() and x.is_cuda):
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
    c2 = blk.conv2
    return c2.stride[0] == 1 and c2.padding[0] == 7 and c2.in_channels == c2.out_channels


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


def train_forward(m, x, timesteps, y, drop_path: float = 0.1):
    """Differentiable MDM.forward, op-for-op with models/denoiser.py:132-196 (denoiser_h3d.py:148-221), with the
    module's current train()/eval() semantics.  x (B,1536,1,T) -> (B,1536,1,T)."""
    engine._require_cuda(x, "x")
    bs, C, _, T = x.shape
    training = m.training
    global _packs
    if WEIGHT_PACKS and torch.is_grad_enabled():
        pk = m.__dict__.get("_syn_weight_packs")
        if pk is None or pk[0].owner() is not m or not (pk[0].valid() and pk[1].valid()):    # (a deep copy of the model brings the original's cache along)
            in_blocks = {id(mod.weight) for blk in m.mytimmblocks for mod in blk.modules() if isinstance(mod, nn.Linear)}
            lin_w = [mod.weight for mod in m.modules() if isinstance(mod, nn.Linear)]
            pk = m.__dict__["_syn_weight_packs"] = (WeightPacks([w for w in lin_w if id(w) not in in_blocks]), WeightPacks([w for w in lin_w if id(w) in in_blocks]))
            pk[0].owner = pk[1].owner = __import__("weakref").ref(m)
        pk[0].refresh()
        global _packs_blocks
        _packs, _packs_blocks = pk
        if not PACK_BLOCKS_LATE:
            _packs_blocks.refresh()
        global _conv_packs
        cp = m.__dict__.get("_syn_conv_packs")
        if training and (cp is None or cp.owner() is not m or not cp.valid()):
            cp = m.__dict__["_syn_conv_packs"] = ConvPacks([mod for mod in m.WavEncoder.modules() if isinstance(mod, nn.Conv1d)])
            cp.owner = __import__("weakref").ref(m)
        if training and cp.lists:
            cp.refresh()
            _conv_packs = cp
        else:
            _conv_packs = None
    h3d = m.variant == "h3d"
    te = m.embed_timestep
    e = te.sequence_pos_encoder.pe[timesteps]                                   # (B,1,512)
    emb_t = lin(F.silu(lin(e, te.time_embed[0])), te.time_embed[2]).permute(1, 0, 2)
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
    a_feat = a.squeeze(2).transpose(1, 2)                  # (B, 128, 256): the channels_last output of the encoder as it lies in memory
    w_feat = lin(_embed(m.text_pre_encoder_body, word), m.text_encoder_body)        # (B, 128, 256)
    at = lin(torch.cat([a_feat, w_feat], dim=2), m.mix_audio_text)
    pool = getattr(m.args, "vqvae_squeeze_scale", 4) if not h3d else 4
    if at.shape[1] % pool:
        at = at[:, :at.shape[1] // pool * pool]              # (F.avg_pool1d drops the incomplete window)
    at = at.reshape(bs, at.shape[1] // pool, pool, at.shape[2]).mean(dim=2)         # F.avg_pool1d over the frame axis (denoiser.py:157) -> (B, T, 256)
    xt = torch.empty(bs, T, C, dtype=torch.bfloat16, device=x.device)
    xt.copy_(x.detach().reshape(bs, C, T).transpose(1, 2))                          # (B, T, C) GEMM operand: transpose + bf16 rounding as one pass
    x_ = lin(xt, m.input_process.poseEmbedding)
    emb = (emb_seed + emb_t.reshape(bs, -1)).unsqueeze(1).expand(bs, T, emb_seed.shape[-1])
    seq = lin(torch.cat((emb, x_, at), dim=2), m.input_process2)
    if m.uses_style:
        st = y["style_feature"]
        force = bool(y.get("uncond", False))
        null = m.uncon_text_embeddings.repeat(bs, 1) if h3d else torch.zeros_like(st)
        if force:
            st = null
        elif training and m.cond_mask_prob > 0.:
            mask = torch.bernoulli(torch.ones(bs, device=st.device) * m.cond_mask_prob).view(bs, 1)
            st = st * (1. - mask) + null * mask
        seq = lin(torch.cat((seq, st.unsqueeze(1).expand(bs, T, st.shape[-1])), dim=2), m.input_process3)
    h = _rotary(m, seq)
    # DropPath (timm_transformer/transformer.py:21-38: one Bernoulli(keep) / keep factor per sample and residual branch): all the
    # step's factors from one draw, and x + branch * factor as one fused multiply-add instead of bernoulli, div, mul, add per branch
    dp = None
    if training and drop_path > 0.:
        keep = 1. - drop_path
        dp = h.new_empty(2 * len(m.mytimmblocks), bs, 1, 1).bernoulli_(keep).div_(keep)
    if WEIGHT_PACKS and torch.is_grad_enabled() and PACK_BLOCKS_LATE and _packs_blocks is not None:
        # The blocks' fragment sets are packed HERE, not at the top of the forward: the audio encoder in between moves ~1 GB through the memory-side cache,
        # and the persistent block kernel below - latency-bound, every phase waits for its first weight fragments - then finds them in HBM
        _packs_blocks.refresh()
    stack = _stack_ok(m, bs, T) and engine_has_xcd_groups(h.device)
    if stack:
        ps = []
        for blk in m.mytimmblocks:
            ps += [blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.proj.weight, blk.attn.proj.bias, blk.norm2.weight, blk.norm2.bias,
                   blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias]
        h = StackFn.apply(h, dp, *ps)
    for i, blk in enumerate(() if stack else m.mytimmblocks):
        if torch.is_grad_enabled() and _fused_ok(bs * T, blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2):
            h = AttnBranchFn.apply(h, blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.proj.weight,
                                   blk.attn.proj.bias, None if dp is None else dp[2 * i])
            h = MlpBranchFn.apply(h, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight,
                                  blk.mlp.fc2.bias, None if dp is None else dp[2 * i + 1])
            continue
        z, h = HipLnForkFn.apply(h, blk.norm1.weight, blk.norm1.bias)
        o = HipAttentionFn.apply(lin(z, blk.attn.qkv))           # (B, T, 3 x 4 heads x 128) -> (B, T, 512)
        br = lin(o, blk.attn.proj)
        h = h + br if dp is None else torch.addcmul(h, br, dp[2 * i])
        z, h = HipLnForkFn.apply(h, blk.norm2.weight, blk.norm2.bias)
        br = lin(HipGeluFn.apply(lin(z, blk.mlp.fc1)), blk.mlp.fc2)
        h = h + br if dp is None else torch.addcmul(h, br, dp[2 * i + 1])
    out = lin(h, m.output_process.poseFinal)                # (B, T, C)
    return out.permute(0, 2, 1).unsqueeze(2)                # (B, C, 1, T) as the reference returns it (a view: the loss kernel reads either layout)


def unused_in_forward(model) -> tuple:
    """Top-level parameter groups the training forward never reads: `embed_style` in both variants and the h3d
    `uncon_audio_embeddings` (denoiser_h3d.py:63).  `uncon_text_embeddings` IS read by the h3d forward - the null prompt of its
    cond-mask dropout (denoiser_h3d.py:119-122; `train_forward` above) - and is trained."""
    m = getattr(model, "module", model)
    return ("embed_style", "uncon_audio_embeddings") if getattr(m, "variant", "beatx") == "h3d" else ("embed_style",)


DDP_AVG_HOOK = bool(int(_os.environ.get("SYN_DDP_AVG_HOOK", "1")))
DIRECT_GRADS = bool(int(_os.environ.get("SYN_DDP_DIRECT_GRADS", "1")))     # captured DDP step: gradients written into the buckets (A/B: 0)
DDP_BUCKET_MB = 32      # 118 MB of fp32 gradients -> 4 all-reduces (+ PyTorch's small first bucket, which starts the stream of collectives
                        # as soon as the output projection's gradients exist).  xGMI is point-to-point, a ring all-reduce is per-link
                        # bound (7 links x ~153 GB/s per GPU): at 8 GPUs a 32 MB bucket is ~0.4 ms on the wire, short enough to overlap
                        # with a ~5 ms backward in four pieces, long enough that RCCL's launch latency (tens of us) stays below 10 %.
                        # (round 2 used 64 MB = two buckets: the second all-reduce could only start when backward was nearly over.)


def make_ddp(model, local_rank: int | None = None, sync_bn: bool = False, capturable: bool = False):
    """One process per GPU, gradients all-reduced over RCCL (backend "nccl"); `embed_style` and the h3d
    `uncon_audio_embeddings` never receive gradients (`unused_in_forward`), hence find_unused_parameters.
    capturable=True prepares the wrapper for `GraphedTrainStep`: the unused-parameter search is a host-side walk plus
    a blocking all-reduce in every backward, which cannot be captured, so those parameters are frozen instead and the
    search is switched off (same gradients: they are None either way)."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if sync_bn:
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
    if capturable:
        frozen = unused_in_forward(model)
        for n, p in model.named_parameters():
            if n.split(".")[0] in frozen:
                p.requires_grad_(False)
    dev_ids = None if local_rank is None else [local_rank]
    ddp = DDP(model, device_ids=dev_ids, broadcast_buffers=False, find_unused_parameters=not capturable,
              gradient_as_bucket_view=True, bucket_cap_mb=DDP_BUCKET_MB)
    if capturable and DDP_AVG_HOOK:
        ddp.register_comm_hook(None, _avg_comm_hook)
    return ddp


def _avg_comm_hook(state, bucket):
    """DDP communication hook of the captured step: ONE collective per bucket that also averages (RCCL's `ncclAvg`), so the reducer neither
    divides a gradient as it copies it into the bucket nor - for a gradient that was written into the bucket directly (`_grad_out`) -
    launches anything per parameter.  Backends without an averaging reduction (gloo, the CPU tests): torch's default hook (divide the
    bucket, all-reduce)."""
    import torch.distributed as dist
    buf = bucket.buffer()
    if dist.get_backend() == "nccl":
        op = dist.ReduceOp.AVG if dist.get_world_size() > 1 else dist.ReduceOp.SUM      # (one rank: RCCL's in-place SUM launches nothing, AVG a pre-multiply kernel)
        fut = dist.all_reduce(buf, op=op, async_op=True).get_future()
        return fut.then(lambda f: f.value()[0])
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    return default_hooks.allreduce_hook(dist.group.WORLD, bucket)


def ddp_bucket_sizes(ddp) -> list[int]:
    """Bytes of gradient per all-reduce bucket of a DDP wrapper (tests, bench report).  Without the unused-parameter search DDP
    runs its FIRST iteration on a single bucket and re-buckets by `bucket_cap_mb` in the order the gradients really arrived;
    this returns the rebuilt plan once it exists (i.e. after the first backward)."""
    try:
        d = ddp._get_ddp_logging_data()
        txt = d.get("rebuilt_bucket_sizes") or d.get("bucket_sizes", "")
        return [int(b) for b in str(txt).split(",") if b.strip()]
    except Exception:
        return []


class ClipAdam(torch.optim.Optimizer):
    """`clip_grad_norm_(params, max_norm)` + `torch.optim.Adam(params, lr, betas, eps, weight_decay).step()` (the reference's
    optimizer step, diffusion_rvqvae_trainer.py:351-356 with optimizers/optim_factory.py's Adam) as `2 + 2 n` launches for `64 n` tensors
    (`syn_opt_sqnorm` / `syn_opt_scalars` / `syn_opt_adam`): the gradients are read twice and never rewritten - the clip factor is applied
    inside the update - where PyTorch's foreach norm + multiply + fused Adam read them three times and write them once.
    State layout and `state_dict()` are torch.optim.Adam's ("step" / "exp_avg" / "exp_avg_sq" per parameter; the step count is ONE device
    tensor per group that every parameter's "step" aliases), so its checkpoints load here and the other way round.  The step count and,
    when `lr` is a tensor, the learning rate live on the device: the step is capturable in a hipGraph.  `last_norm()` = the total gradient
    norm of the latest step (what clip_grad_norm_ returns), a device tensor.
    Differences from the two PyTorch calls: p.grad keeps the UNCLIPPED gradient after the step; amsgrad / maximize are not offered."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_norm = float(max_norm) if max_norm else 0.0
        self._lists = {}            # (pointers of the tensors with gradients) -> prepared argument lists
        self._scal = None

    def last_norm(self):
        return None if self._scal is None else self._scal[0][0, 3]

    def _group_state(self, group, dev):
        ps = [p for p in group["params"] if p.grad is not None]
        for p in ps:
            if p.dtype is not torch.float32 or not p.is_contiguous() or p.grad.dtype is not torch.float32 or not p.grad.is_contiguous() or p.grad.is_sparse:
                raise _lib.SynHipError("ClipAdam takes contiguous fp32 parameters with dense contiguous fp32 gradients")
        step = None
        for p in group["params"]:
            st = self.state.get(p)
            if st and "step" in st:
                step = st["step"] if step is None else step
        if step is None or not torch.is_tensor(step) or step.device != dev or step.dtype is not torch.float32 or step.dim() != 0:
            step = torch.tensor(float(step) if step is not None else 0.0, dtype=torch.float32, device=dev)
        for p in ps:
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format), torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] = step                                           # (one count per group; a loaded state_dict's copies are re-aliased here)
        return ps, step

    def _prepare(self, ps):
        # (every pointer a prepared list holds is part of its key: load_state_dict replaces the moment tensors, backward the gradients)
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p in ps)
        ent = self._lists.get(key)
        if ent is None:
            if len(self._lists) > 4:
                self._lists.clear()
            lib, lists, blocks = _lib.load(), [], []
            for lo in range(0, len(ps), _lib.SYN_OPT_MAX):
                L = _lib.SynOptList()
                chunk = ps[lo:lo + _lib.SYN_OPT_MAX]
                for i, p in enumerate(chunk):
                    st = self.state[p]
                    L.p[i], L.g[i], L.m[i], L.v[i], L.numel[i] = p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
                L.n = len(chunk)
                lists.append(L)
                blocks.append(int(lib.syn_opt_blocks(C.byref(L))))
            ent = self._lists[key] = (lists, blocks)
        return ent

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        groups = []
        for group in self.param_groups:
            first = next((p for p in group["params"] if p.grad is not None), None)
            if first is None:
                continue
            engine._require_cuda(first, "parameter")
            ps, step = self._group_state(group, first.device)
            groups.append((group, ps, step) + self._prepare(ps))
        if not groups:
            return loss
        dev = groups[0][1][0].device
        lib, st = _lib.load(), _lib.current_stream(dev)
        total = sum(sum(b) for *_, b in groups)
        if self._scal is None or self._scal[0].device != dev or self._scal[1].numel() < total or len(self._scal[0]) < len(groups):
            self._scal = (torch.zeros(max(len(groups), 1), 4, device=dev), torch.empty(max(total, 1), device=dev))
        scal, partials = self._scal
        if self.max_norm > 0:                                           # the norm is over ALL parameters, whatever their group
            off = 0
            for _, _, _, lists, blocks in groups:
                for L, b in zip(lists, blocks):
                    _lib.check(lib.syn_opt_sqnorm(C.byref(L), partials[off:].data_ptr(), st), "syn_opt_sqnorm")
                    off += b
        for gi, (group, ps, step, lists, blocks) in enumerate(groups):
            lr, (b1, b2) = group["lr"], group["betas"]
            lr_dev = lr if torch.is_tensor(lr) else None
            if lr_dev is not None and (lr_dev.device != dev or lr_dev.dtype is not torch.float32):
                raise _lib.SynHipError("ClipAdam: a tensor learning rate must be an fp32 tensor on the parameters' device")
            _lib.check(lib.syn_opt_scalars(partials.data_ptr(), total if self.max_norm > 0 else 0, self.max_norm, _lib.ptr(lr_dev),
                                           0.0 if lr_dev is not None else float(lr), b1, b2, step.data_ptr(), scal[gi].data_ptr(), st), "syn_opt_scalars")
            for L in lists:
                _lib.check(lib.syn_opt_adam(C.byref(L), scal[gi].data_ptr(), b1, b2, group["eps"], group["weight_decay"], st), "syn_opt_adam")
        return loss


def _check_clip(optimizer, grad_norm):
    """A ClipAdam clips inside its step with ITS max_norm; a `grad_norm` argument that says something else must not pass silently."""
    if isinstance(optimizer, ClipAdam) and abs(float(grad_norm or 0.0) - optimizer.max_norm) > 1e-12:
        raise ValueError(f"grad_norm={grad_norm} but the ClipAdam optimizer was built with max_norm={optimizer.max_norm or None}: "
                         "construct it with max_norm=grad_norm (the clip is part of its step), or pass grad_norm=optimizer.max_norm")


def train_step(model, diffusion, sampler, optimizer, x0, model_kwargs, grad_norm: float = 0.99):
    """The body of the reference's hot training loop (diffusion_rvqvae_trainer.py:339-356, 555-560)."""
    _check_clip(optimizer, grad_norm)
    t, _ = sampler.sample(x0.shape[0], x0.device)
    optimizer.zero_grad(set_to_none=True)
    loss = diffusion.training_losses(model, x0, t, model_kwargs=model_kwargs)["loss"].mean()
    loss.backward()
    if grad_norm and not isinstance(optimizer, ClipAdam):           # (ClipAdam carries its max_norm: the clip is part of its step)
        torch.nn.utils.clip_grad_norm_(model.parameters(), grad_norm)
    optimizer.step()
    return loss.detach()



class GraphedTrainStep:
    """`train_step` captured once in a hipGraph and replayed.  The step is ~1 000 kernel launches long and host-bound
    when issued from Python (device time 15 ms, wall 17-21 ms at 32 clips); a replay costs the device time (15.5 ms).
    Static shapes: every call must bring tensors of the shapes seen at construction.  The optimizer must be constructed
    with ``capturable=True``.  With DDP: wrap with ``make_ddp(..., capturable=True)`` inside ``torch.cuda.stream(s)``,
    pass ``stream=s`` and ``warmup=11`` (DDP needs that many eager iterations before a capture), and set
    ``TORCH_NCCL_ASYNC_ERROR_HANDLING=0`` before ``init_process_group`` - the bucketed all-reduces are then nodes of the
    graph (`scripts/bench_train_ddp.py`; checked with one rank over RCCL: 22.6 ms eager -> 18.0 ms replayed).

        step = GraphedTrainStep(model, diffusion, optimizer, x0, {"y": y})
        loss = step(x0, t, {"y": y})          # t from the schedule sampler (host RNG, as in the reference)

    Replays are stream-ordered like any launch; nothing waits for them.  (Round 1 synchronised after every replay because
    back-to-back replays aborted in the HIP runtime with an HSA memory-aperture violation.  Root cause, found in round 2: the
    word-embedding gradient - PyTorch-ROCm's embedding_dense_backward, a chain of ~15 sort / scan / segment kernels - does not
    survive being replayed at the bench size; with `EmbeddingFn` in its place 300 un-synchronised replays run clean, and so do the
    launch variants that used to trip the same abort.  SYN_TORCH_EMBEDDING_GRAD=1 brings the op back to reproduce it.)
    Call `close()` (or let the object die) before interpreter shutdown."""

    def __init__(self, model, diffusion, optimizer, x0, model_kwargs, grad_norm: float = 0.99, warmup: int = 3, stream=None,
                 keep_warmup_updates: bool = False, noise=None):
        engine._require_cuda(x0, "x0")
        _check_clip(optimizer, grad_norm)
        self.model, self.opt, self.grad_norm, self.diffusion = model, optimizer, grad_norm, diffusion
        self.sync = bool(_os.environ.get("SYN_TRAIN_GRAPH_SYNC"))   # wait for every replay (not needed: see the class docstring)
        self.wrapped = diffusion._wrap_model(model)          # its timestep map is uploaded once, outside the capture
        self.x0 = x0.detach().clone()
        self.t = torch.zeros(x0.shape[0], dtype=torch.long, device=x0.device)
        # `noise` (a tensor like x0): the step takes its q_sample noise from a static buffer the caller fills per call (`__call__(..., noise=)`,
        # the `noise=` argument of training_losses - parity tests with injected noise) instead of drawing it inside the graph
        self.noise = None if noise is None else noise.detach().clone()
        self.y = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in model_kwargs["y"].items()}
        side = stream if stream is not None else torch.cuda.Stream(device=x0.device)   # DDP: the stream the wrapper was built on
        side.wait_stream(torch.cuda.current_stream(x0.device))
        with torch.cuda.stream(side):                         # warm-up: lazy state, Adam state, DDP's bucket rebuild
            # The warm-up iterations are real optimizer steps on the construction batch at t = 0.  They must not count as training:
            # parameters, buffers (BatchNorm statistics) and the optimizer's state are put back IN PLACE afterwards (the capture
            # holds their addresses), so the first replay is update number 1 of the run - or number n + 1 after a resume.
            snap = None if keep_warmup_updates else self._snapshot()
            self.bound = 0
            for i in range(warmup):
                self._body()
                if i == 2 and hasattr(model, "reducer") and DIRECT_GRADS and DDP_AVG_HOOK:
                    # DDP has rebuilt its buckets by now and every parameter's .grad is a view of one: from here on the backward kernels
                    # write weight gradients straight into those views (the remaining warm-up iterations already run that way)
                    self.bound = bind_grad_buffers(model)
        torch.cuda.current_stream(x0.device).wait_stream(side)
        mode = "global"
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # A process group's watchdog thread polls the events of the warm-up's collectives (hipEventQuery, then their release).  Let it
            # retire them before the capture starts - the device is drained, the thread sweeps every ~100 ms - and keep other threads'
            # runtime calls out of this capture's error domain (thread-local mode: launches on the capturing stream are captured whichever
            # thread issues them - the autograd engine's do - but a foreign thread's query cannot invalidate the capture).
            # Deterministic part of the drain: every rank has issued its warm-up collectives (barrier) and the device has finished them
            # (synchronize).  What is left is the watchdog thread's sweep, which PyTorch does not expose: it polls every ~100 ms, so the wait
            # is three periods by default (SYN_GRAPH_WATCHDOG_DRAIN_S to change it, 0 to skip).
            if torch.distributed.get_world_size() > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize(x0.device)
            drain = float(_os.environ.get("SYN_GRAPH_WATCHDOG_DRAIN_S", "0.3"))
            if drain > 0:
                __import__("time").sleep(drain)
            mode = "thread_local"
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode=mode):
            self.loss = self._body()
        if snap is not None:
            with torch.cuda.stream(side):
                self._restore(snap)
            torch.cuda.current_stream(x0.device).wait_stream(side)

    def _snapshot(self):
        tensors = [p for p in self.model.parameters()] + list(self.model.buffers())
        state = {}
        for group in self.opt.param_groups:
            for p in group["params"]:
                st = self.opt.state.get(p)
                if st:
                    state[p] = {k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)}
        return [(t, t.detach().clone()) for t in tensors], state

    @torch.no_grad()
    def _restore(self, snap):
        for t, saved in snap[0]:
            t.copy_(saved)
        for group in self.opt.param_groups:
            for p in group["params"]:
                for k, v in self.opt.state.get(p, {}).items():
                    if torch.is_tensor(v):               # state born during the warm-up goes back to its initial value: zero
                        v.copy_(snap[1][p][k]) if p in snap[1] and k in snap[1][p] else v.zero_()

    def _body(self):
        self.opt.zero_grad(set_to_none=True)
        if self.bound:
            _reset_handed(self.model)
        loss = self.diffusion.training_losses(self.wrapped, self.x0, self.t, model_kwargs={"y": self.y}, noise=self.noise)["loss"].mean()
        loss.backward()
        if self.grad_norm and not isinstance(self.opt, ClipAdam):
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_norm)
        self.opt.step()
        return loss.detach()

    def __call__(self, x0, t, model_kwargs, noise=None):
        self.x0.copy_(x0)
        self.t.copy_(t)
        if (noise is None) != (self.noise is None):
            raise ValueError("GraphedTrainStep: pass `noise` to every call if and only if the step was constructed with a noise buffer")
        if noise is not None:
            self.noise.copy_(noise)
        for k, v in model_kwargs["y"].items():
            if torch.is_tensor(v):
                self.y[k].copy_(v)
        self.graph.replay()
        if self.sync:
            torch.cuda.current_stream(self.x0.device).synchronize()
        return self.loss.clone()        # (stream-ordered copy: the static tensor is overwritten by the next replay)

    def close(self):
        if getattr(self, "graph", None) is not None:
            torch.cuda.synchronize()
            self.graph, self.loss = None, None
            if getattr(self, "bound", 0):
                unbind_grad_buffers(self.model)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
