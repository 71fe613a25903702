"""CPU restatement of the reference's RVQ-VAE (eval mode) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(syntalker_amd/rvqvae.py) runs the HIP kernels and never falls back to it.

Follows, on a reference-keyed state_dict (fp32, functional PyTorch on CPU):
  encoder / decoder stacks        models/vq/encdec.py:4-67
  residual block (no norm, ReLU)  models/vq/resnet.py:12-68   (dilations 9, 3, 1: reverse_dilation, :71-83)
  one quantiser layer, eval mode  models/vq/quantizer.py:62-69 (distance), :143-171 (forward; straight-through form)
  residual VQ                     models/vq/residual_vq.py:91-140 (forward), :142-165 (quantize), :54-70 (codes from indices)
  RVQVAE entry points             models/vq/model.py:53-109
Pinned against tests/golden/vq_outputs.npz (outputs of the reference itself, tests/golden/make_vq_golden.py).
"""
import torch
import torch.nn.functional as F

DOWN_T, DEPTH, GROWTH = 2, 3, 3          # diffusion_rvqvae_trainer.py:97-101
NUM_Q = 6                                # diffusion_rvqvae_trainer.py:89


def _conv(sd, key, x, stride=1, pad=1, dil=1):
    return F.conv1d(x, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=pad, dilation=dil)


def _resnet(sd, key, x):
    # models/vq/resnet.py:71-83: blocks built with dilation 3**depth, then reversed -> 9, 3, 1
    for i in range(DEPTH):
        d = GROWTH ** (DEPTH - 1 - i)
        h = _conv(sd, f"{key}.model.{i}.conv1", F.relu(x), pad=d, dil=d)      # resnet.py:52-58 (norm = Identity)
        h = _conv(sd, f"{key}.model.{i}.conv2", F.relu(h), pad=0)             # :60-66; dropout is identity in eval
        x = h + x
    return x


def encoder(sd, x):
    """x: (N, D, T) -> (N, 512, T / 4).  models/vq/encdec.py:4-33."""
    x = F.relu(_conv(sd, "encoder.model.0", x))
    for i in range(DOWN_T):
        x = _conv(sd, f"encoder.model.{2 + i}.0", x, stride=2, pad=1)         # filter_t = 4, pad_t = 1 (:19)
        x = _resnet(sd, f"encoder.model.{2 + i}.1", x)
    return _conv(sd, f"encoder.model.{2 + DOWN_T}", x)


def decoder(sd, x):
    """x: (N, 512, T) -> (N, 4 T, D).  models/vq/encdec.py:36-67."""
    x = F.relu(_conv(sd, "decoder.model.0", x))
    for i in range(DOWN_T):
        x = _resnet(sd, f"decoder.model.{2 + i}.0", x)
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = _conv(sd, f"decoder.model.{2 + i}.2", x)
    x = F.relu(_conv(sd, f"decoder.model.{2 + DOWN_T}", x))
    return _conv(sd, f"decoder.model.{4 + DOWN_T}", x).permute(0, 2, 1)


def quantize_layer(cb, x):
    """One QuantizeEMAReset.forward in eval mode on rows x (M, 512): returns (straight-through output, indices).
    quantizer.py:62-69: distance = |x|^2 - 2 x.C^T + |c|^2, index = argmax(-distance) (no Gumbel noise in eval)."""
    kw = cb.t()
    dist = torch.sum(x ** 2, dim=-1, keepdim=True) - 2 * torch.matmul(x, kw) + torch.sum(kw ** 2, dim=0, keepdim=True)
    idx = (-dist).argmax(dim=-1)
    xd = F.embedding(idx, cb)
    return x + (xd - x), idx             # quantizer.py:163 (the straight-through expression, kept for its rounding)


def residual_vq(sd, x):
    """x: (N, 512, T) -> (quantised (N, 512, T), indices (N, T, Q), commit loss, perplexity).  residual_vq.py:91-140."""
    n, c, t = x.shape
    res = x.permute(0, 2, 1).reshape(n * t, c)
    out = torch.zeros_like(res)
    idxs, losses, perps = [], [], []
    for q in range(NUM_Q):
        cb = sd[f"quantizer.layers.{q}.codebook"]
        rows_in = res
        qd, idx = quantize_layer(cb, rows_in)
        losses.append(F.mse_loss(rows_in, F.embedding(idx, cb)))
        onehot_cnt = torch.zeros(cb.shape[0], device=idx.device).scatter_add_(0, idx, torch.ones(idx.shape[0], device=idx.device))
        prob = onehot_cnt / onehot_cnt.sum()
        perps.append(torch.exp(-torch.sum(prob * torch.log(prob + 1e-7))))
        res = res - qd
        out = out + qd
        idxs.append(idx.view(n, t))
    back = lambda r: r.view(n, t, c).permute(0, 2, 1).contiguous()
    return back(out), torch.stack(idxs, dim=-1), sum(losses) / NUM_Q, sum(perps) / NUM_Q


def codes_from_indices(sd, idx):
    """idx (N, T, Q) -> summed codes (N, 512, T).  residual_vq.py:54-70 + model.py:86-89."""
    out = 0
    for q in range(idx.shape[-1]):
        out = out + F.embedding(idx[..., q], sd[f"quantizer.layers.{q}.codebook"])
    return out.permute(0, 2, 1)


def map2latent(sd, pose):
    """(N, T, D) -> (N, T/4, 512).  model.py:95-100."""
    return encoder(sd, pose.permute(0, 2, 1).float()).permute(0, 2, 1)


def latent2origin(sd, lat):
    """(N, T/4, 512) -> ((N, T, D), commit loss, perplexity).  model.py:102-109."""
    xq, _, commit, perp = residual_vq(sd, lat.permute(0, 2, 1))
    return decoder(sd, xq), commit, perp


def encode(sd, pose):
    """(N, T, D) -> indices (N, T/4, Q).  model.py:53-65."""
    return residual_vq(sd, encoder(sd, pose.permute(0, 2, 1).float()))[1]


def forward_decoder(sd, idx):
    """indices (N, T/4, Q) -> (N, T, D).  model.py:86-93."""
    return decoder(sd, codes_from_indices(sd, idx))
