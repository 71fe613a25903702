"""LayerNorm, the 32-token attention and GELU of a transformer block as autograd nodes of their own, on the fp32 kernels the product's fused branch
nodes (`training.AttnBranchFn` / `MlpBranchFn`) and the persistent block kernels are built from.  Test infrastructure: tests/test_gpu_kernels.py checks
these against PyTorch's own ops (1e-5) and composes the op-by-op reference of a fused branch from them.  (reference: models/timm_transformer/
transformer.py:56-104, 117-151, 154-198)"""
import torch

from syntalker_amd import _lib, engine
from syntalker_amd.training import _f32c


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
    norm and the residual add; the residual path's gradient goes into the LayerNorm backward kernel as its addend."""

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
