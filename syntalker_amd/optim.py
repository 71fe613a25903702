"""The optimizer step and the data-parallel plumbing of the training path (e): `ClipAdam` (clip_grad_norm + Adam as a few
multi-tensor launches, reference diffusion_rvqvae_trainer.py:351-356), `train_step` / `GraphedTrainStep` (the body of the reference's hot loop,
:339-356, eager or as one hipGraph), `make_ddp` (one process per GPU over RCCL, train.py:87-94) and the bound gradient buffers that let the
backward kernels write straight into DDP's bucket views.  `syntalker_amd.training` re-exports every name here."""
from __future__ import annotations

import ctypes as C
import math
import os as _os

import torch
import torch.nn as nn

from . import _lib, engine

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



def unused_in_forward(model) -> tuple:
    """Top-level parameter groups the training forward never reads: `embed_style` in both variants and the h3d
    `uncon_audio_embeddings` (denoiser_h3d.py:63).  `uncon_text_embeddings` IS read by the h3d forward - the null prompt of its
    cond-mask dropout (denoiser_h3d.py:119-122; `train_forward` above) - and is trained."""
    m = getattr(model, "module", model)
    return ("embed_style", "uncon_audio_embeddings") if getattr(m, "variant", "beatx") == "h3d" else ("embed_style",)


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
    if capturable:
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
        engine.note_raw_write()                                          # (the parameters moved and their version counters did not)
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
    survive being replayed at the bench size; with `syn_embedding_wgrad` in its place 300 un-synchronised replays run clean, and so do the
    launch variants that used to trip the same abort.)
    Call `close()` (or let the object die) before interpreter shutdown."""

    def __init__(self, model, diffusion, optimizer, x0, model_kwargs, grad_norm: float = 0.99, warmup: int = 3, stream=None,
                 keep_warmup_updates: bool = False, noise=None):
        engine._require_cuda(x0, "x0")
        _check_clip(optimizer, grad_norm)
        self.model, self.opt, self.grad_norm, self.diffusion = model, optimizer, grad_norm, diffusion
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
                if i == 2 and hasattr(model, "reducer"):
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
        self._baked0 = self._baked(self.y)

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

    _BAKED = ("optimizer hyper-parameters held as Python numbers (lr, betas, eps, weight_decay: pass lr as a device tensor to change it between replays)",
              "keys, non-tensor values or tensor shapes of y", "model.training", "drop_path", "cond_mask_prob", "the parameters' requires_grad pattern")

    def _baked(self, y):
        """Everything the capture turned into constants of the graph.  A call that brings something else would replay the OLD behaviour silently:
        `__call__` compares and raises instead (construct a new GraphedTrainStep for the new configuration)."""
        num = lambda v: None if torch.is_tensor(v) else v
        hyper = [tuple(num(g.get(k)) for k in ("lr", "betas", "eps", "weight_decay")) for g in self.opt.param_groups]
        core = getattr(self.model, "module", self.model)
        ys = sorted((k, tuple(v.shape) if torch.is_tensor(v) else v) for k, v in y.items())
        return [hyper, ys, bool(self.model.training), getattr(core, "drop_path", None), getattr(core, "cond_mask_prob", None),
                [p.requires_grad for p in self.model.parameters()]]

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
        now = self._baked(model_kwargs["y"])
        if now != self._baked0:
            what = [w for w, a, b in zip(self._BAKED, now, self._baked0) if a != b]
            raise RuntimeError("GraphedTrainStep: the captured step is fixed in " + "; ".join(what) + " - this call differs from the construction. "
                               "Build a new GraphedTrainStep for the new configuration.")
        if tuple(x0.shape) != tuple(self.x0.shape) or tuple(t.shape) != tuple(self.t.shape):
            raise RuntimeError(f"GraphedTrainStep: static shapes - constructed for x0 {tuple(self.x0.shape)}, called with {tuple(x0.shape)}")
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
        engine.note_raw_write()                                          # (parameters, BatchNorm statistics: one opaque launch to the version counters)
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
