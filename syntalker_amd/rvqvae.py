"""RVQ-VAE either side of the denoising loop: the reference's ``models.vq.model.RVQVAE`` surface on the
HIP kernels of csrc/syn_rvq.inc.

Same constructor arguments, same ``state_dict`` keys (a trained ``net`` checkpoint of the reference loads with
``load_state_dict``, diffusion_rvqvae_trainer.py:153-155), same entry points and shapes:
  map2latent(pose (N,T,D))            -> (N, T/4, 512)                      models/vq/model.py:95-100
  latent2origin(latent (N,T/4,512))   -> ((N,T,D), commit loss, perplexity) :102-109
  encode(pose)                        -> (indices (N,T/4,6), codes (6,N,512,T/4))   :53-65
  forward_decoder(indices)            -> (N,T,D)                            :86-93
Inference only (the reference keeps these models in eval(), trainer :159-161, :174-177).  CUDA(ROCm) tensors only:
there is no CPU fallback, a missing extension raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib, engine, synth

NUM_Q, NB_CODE, CODE_DIM = 6, 512, 512


def _conv_specs(dim, width=512, down_t=2, depth=3, growth=3):
    """(key, cin, cout, taps, stride, dil, pad) of every Conv1d, in state_dict order (models/vq/encdec.py, resnet.py)."""
    enc = [("encoder.model.0", dim, width, 3, 1, 1, 1)]
    dec = [("decoder.model.0", width, width, 3, 1, 1, 1)]
    for i in range(down_t):
        enc.append((f"encoder.model.{2 + i}.0", width, width, 4, 2, 1, 1))
        for j in range(depth):
            d = growth ** (depth - 1 - j)                                  # reverse_dilation: 9, 3, 1
            enc.append((f"encoder.model.{2 + i}.1.model.{j}.conv1", width, width, 3, 1, d, d))
            enc.append((f"encoder.model.{2 + i}.1.model.{j}.conv2", width, width, 1, 1, 1, 0))
            dec.append((f"decoder.model.{2 + i}.0.model.{j}.conv1", width, width, 3, 1, d, d))
            dec.append((f"decoder.model.{2 + i}.0.model.{j}.conv2", width, width, 1, 1, 1, 0))
        dec.append((f"decoder.model.{2 + i}.2", width, width, 3, 1, 1, 1))
    enc.append((f"encoder.model.{2 + down_t}", width, width, 3, 1, 1, 1))
    dec.append((f"decoder.model.{2 + down_t}", width, width, 3, 1, 1, 1))
    dec.append((f"decoder.model.{4 + down_t}", width, dim, 3, 1, 1, 1))
    return enc, dec


def _register(root: nn.Module, key: str, value: torch.Tensor, buffer=False):
    """Create the nested (empty) modules a dotted state_dict key implies and hang the tensor on the leaf."""
    *path, leaf = key.split(".")
    m = root
    for p in path:
        if p not in m._modules:
            m.add_module(p, nn.Module())
        m = m._modules[p]
    if buffer:
        m.register_buffer(leaf, value)
    else:
        m.register_parameter(leaf, nn.Parameter(value, requires_grad=False))


def pack_conv(w: torch.Tensor, cin_p: int, cout_p: int) -> torch.Tensor:
    """Conv1d weight (cout, cin, taps) -> MFMA A-operand fragments [taps][cout_p/16][cin_p/32][lane = g*16 + r][8] bf16:
    lane (r, g) of fragment (tap, co/16, ci/32) holds W[co = 16*f + r][ci = 32*ks + 8*g .. +7][tap]."""
    cout, cin, taps = w.shape
    wp = torch.zeros(taps, cout_p, cin_p, dtype=torch.float32, device=w.device)
    wp[:, :cout, :cin] = w.permute(2, 0, 1)
    wp = wp.view(taps, cout_p // 16, 16, cin_p // 32, 4, 8).permute(0, 1, 3, 4, 2, 5)
    return wp.contiguous().to(torch.bfloat16)


def _up(n, m):
    return (n + m - 1) // m * m


class RVQVAE(nn.Module):
    def __init__(self, args, input_width=263, nb_code=1024, code_dim=512, output_emb_width=512, down_t=3, stride_t=2,
                 width=512, depth=3, dilation_growth_rate=3, activation="relu", norm=None):
        super().__init__()
        if (nb_code, code_dim, output_emb_width, width, stride_t) != (NB_CODE, CODE_DIM, CODE_DIM, 512, 2) or down_t != 2 \
                or depth != 3 or dilation_growth_rate != 3 or activation != "relu" or norm is not None \
                or getattr(args, "num_quantizers", NUM_Q) != NUM_Q or getattr(args, "shared_codebook", False):
            raise NotImplementedError("RVQVAE: built for the configuration diffusion_rvqvae_trainer.py:89-103 constructs "
                                      "(6 x 512 codes of 512 dims, width 512, down_t 2, depth 3, ReLU, no norm)")
        self.code_dim, self.num_code, self.input_width = code_dim, nb_code, input_width
        self._enc, self._dec = _conv_specs(input_width)
        for key, cin, cout, taps, *_ in self._enc + self._dec:
            _register(self, key + ".weight", torch.zeros(cout, cin, taps))
            _register(self, key + ".bias", torch.zeros(cout))
        for q in range(NUM_Q):
            _register(self, f"quantizer.layers.{q}.codebook", torch.zeros(nb_code, code_dim), buffer=True)
        self._packed = None
        self.eval()

    def __getstate__(self):
        st = self.__dict__.copy()               # (deepcopy / torch.save: without the packed copy - ctypes pointers into this module's tensors)
        st["_packed"] = None
        return st

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("RVQVAE is an inference module here (the reference keeps it in eval(), trainer :159-161)")
        return super().train(False)

    # ---- device-side parameters ----------------------------------------------------------------------------------
    def packed(self):
        """syn_vq_model of this module: fragment-packed conv weights, padded biases, codebook views.  Rebuilt when a
        parameter changes (version counters) or moves."""
        tensors = list(self.parameters()) + list(self.buffers())
        ver = (engine.raw_write_epoch(),) + tuple((v._version, v.data_ptr()) for v in tensors)
        if self._packed is not None and self._packed["ver"] == ver:
            return self._packed
        sd = self.state_dict()
        dev = sd["decoder.model.0.weight"].device
        if dev.type != "cuda":
            raise _lib.SynHipError("RVQVAE runs on the HIP kernels only: move the module to the GPU (no CPU fallback)")
        _lib.load()
        vm, keep = _lib.SynVqModel(), []
        vm.pose_dim = self.input_width
        for arr, specs in ((vm.enc, self._enc), (vm.dec, self._dec)):
            for i, (key, cin, cout, taps, stride, dil, pad) in enumerate(specs):
                cin_p, cout_p = _up(cin, 32), _up(cout, 128)
                w = pack_conv(sd[key + ".weight"].float(), cin_p, cout_p)
                b = torch.zeros(cout_p, device=dev)
                b[:cout] = sd[key + ".bias"].float()
                arr[i] = _lib.SynVqConv(w.data_ptr(), b.data_ptr(), cin_p, cout_p, cout, taps, stride, dil, pad, 0, 0, 0)
                keep += [w, b]
        cb = torch.stack([sd[f"quantizer.layers.{q}.codebook"].float() for q in range(NUM_Q)]).contiguous()
        cbt = cb.transpose(1, 2).contiguous()
        cc = torch.sum(cbt ** 2, dim=1).contiguous()                        # quantizer.py:66: sum(k_w**2, dim=0)
        vm.codebooks, vm.codebooks_t, vm.code_sq = cb.data_ptr(), cbt.data_ptr(), cc.data_ptr()
        self._packed = {"ver": ver, "model": vm, "keep": keep, "cb": cb, "cbt": cbt, "cc": cc, "ws": {}}
        return self._packed

    def _workspace(self, p, clips, t_pose, dev):
        key = (clips, t_pose)
        if key not in p["ws"]:
            if len(p["ws"]) > 4:
                p["ws"].clear()
            nbytes = _lib.load().syn_vq_workspace_bytes(clips, t_pose, self.input_width)
            p["ws"][key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return p["ws"][key]

    def _quantize(self, lat):
        """lat (N, T', 512) fp32 -> quantised fp32 rows, indices (N, T', 6), commit loss, perplexity (quantiser alone)."""
        p = self.packed()
        lib = _lib.load()
        n, t, c = lat.shape
        rows = n * t
        x = lat.contiguous().float()
        qf = torch.empty(rows, c, device=x.device)
        idx = torch.empty(rows, NUM_Q, device=x.device, dtype=torch.int32)
        sq = torch.empty(lib.syn_vq_quantize_groups(rows), NUM_Q, device=x.device)
        hist = torch.zeros(NUM_Q, NB_CODE, device=x.device, dtype=torch.int32)
        _lib.check(lib.syn_vq_quantize(x.data_ptr(), p["cb"].data_ptr(), p["cbt"].data_ptr(), p["cc"].data_ptr(), qf.data_ptr(),
                                       None, idx.data_ptr(), sq.data_ptr(), hist.data_ptr(), rows,
                                       _lib.current_stream(x.device)), "syn_vq_quantize")
        commit, perp = self._stats(sq, hist, rows)
        return qf.view(n, t, c), idx.view(n, t, NUM_Q).long(), commit, perp

    @staticmethod
    def _stats(sq, hist, rows):
        commit = (sq.sum(0) / (rows * CODE_DIM)).mean()                     # F.mse_loss per layer, mean over layers (residual_vq.py:137)
        prob = hist.float() / rows
        perp = torch.exp(-(prob * torch.log(prob + 1e-7)).sum(1)).mean()   # quantizer.py:84-90, residual_vq.py:138
        return commit, perp

    # ---- the reference's entry points ------------------------------------------------------------------------------
    @torch.no_grad()
    def map2latent(self, x):
        """(N, T, D) -> (N, T/4, 512).  models/vq/model.py:95-100."""
        p = self.packed()
        n, t, d = x.shape
        if d != self.input_width or t % 4:
            raise ValueError(f"RVQVAE.map2latent: expected (N, T % 4 == 0, {self.input_width}), got {tuple(x.shape)}")
        x = x.contiguous().float()
        lat = torch.empty(n, t // 4, CODE_DIM, device=x.device)
        _lib.check(_lib.load().syn_vq_map2latent(C.byref(p["model"]), x.data_ptr(), n, t, self._workspace(p, n, t, x.device).data_ptr(),
                                                 lat.data_ptr(), _lib.current_stream(x.device)), "syn_vq_map2latent")
        return lat

    @torch.no_grad()
    def latent2origin(self, x):
        """(N, T', 512) -> ((N, 4 T', D), commit loss, perplexity).  models/vq/model.py:102-109."""
        p = self.packed()
        lib = _lib.load()
        n, t, _ = x.shape
        x = x.contiguous().float()
        out = torch.empty(n, 4 * t, self.input_width, device=x.device)
        idx = torch.empty(n * t, NUM_Q, device=x.device, dtype=torch.int32)
        sq = torch.empty(lib.syn_vq_quantize_groups(n * t), NUM_Q, device=x.device)
        hist = torch.zeros(NUM_Q, NB_CODE, device=x.device, dtype=torch.int32)
        _lib.check(lib.syn_vq_latent2origin(C.byref(p["model"]), x.data_ptr(), n, t, self._workspace(p, n, 4 * t, x.device).data_ptr(),
                                            out.data_ptr(), idx.data_ptr(), sq.data_ptr(), hist.data_ptr(),
                                            _lib.current_stream(x.device)), "syn_vq_latent2origin")
        commit, perp = self._stats(sq, hist, n * t)
        return out, commit, perp

    @torch.no_grad()
    def encode(self, x):
        """(N, T, D) -> (indices (N, T/4, 6), per-layer codes (6, N, 512, T/4)).  models/vq/model.py:53-65."""
        lat = self.map2latent(x)
        n, t, c = lat.shape
        _, idx, _, _ = self._quantize(lat)
        cb = self.packed()["cb"]
        res, codes = lat.reshape(n * t, c), []
        for q in range(NUM_Q):                                              # the per-layer straight-through outputs (residual_vq.py:150-160)
            cq = cb[q][idx[..., q].reshape(-1)]
            qd = res + (cq - res)
            res = res - qd
            codes.append(qd.view(n, t, c).permute(0, 2, 1))
        return idx, torch.stack(codes, dim=0)

    @torch.no_grad()
    def forward_decoder(self, x):
        """indices (N, T', Q <= 6) -> (N, 4 T', D).  models/vq/model.py:86-93."""
        p = self.packed()
        n, t, nq = x.shape
        idx = x.to(torch.int32).contiguous()
        out = torch.empty(n, 4 * t, self.input_width, device=x.device)
        _lib.check(_lib.load().syn_vq_forward_decoder(C.byref(p["model"]), idx.data_ptr(), nq, n, t,
                                                      self._workspace(p, n, 4 * t, x.device).data_ptr(), out.data_ptr(),
                                                      _lib.current_stream(x.device)), "syn_vq_forward_decoder")
        return out

    @torch.no_grad()
    def forward(self, x):
        """models/vq/model.py:67-83 in eval mode."""
        y, commit, perp = self.latent2origin(self.map2latent(x))
        return {"rec_pose": y, "commit_loss": commit, "perplexity": perp}


# ---- construction as the trainer does it (synthetic weights / poses for tests and benches live in synth.py) ----
def vq_args():
    from types import SimpleNamespace
    return SimpleNamespace(num_quantizers=NUM_Q, shared_codebook=False, quantize_dropout_prob=0.2, mu=0.99)   # trainer :89-92


def build(dim: int) -> "RVQVAE":
    """The constructor call of diffusion_rvqvae_trainer.py:107-150 for a body part of `dim` pose channels."""
    return RVQVAE(vq_args(), dim, NB_CODE, CODE_DIM, CODE_DIM, 2, 2, 512, 3, 3, "relu", None)
