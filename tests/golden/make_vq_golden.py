#!/usr/bin/env python3
"""Golden vectors for the RVQ-VAE either side of the denoising loop, produced by running the
REFERENCE modules (models/vq/model.py:RVQVAE) on CPU in the build container.

The reference's quantiser registers its codebook with `.cuda()` at construction (models/vq/quantizer.py:43); this
container has no GPU, so Tensor.cuda is made the identity for the duration of the script.  Weights are the
name-keyed deterministic tensors of syntalker_amd.synth; only seeds and the reference's OUTPUTS are stored.
    python tests/golden/make_vq_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
from syntalker_amd import rvqvae, synth  # noqa: E402

PARTS = (("upper", 78), ("hands", 180), ("lower", 57))   # diffusion_rvqvae_trainer.py:105-150 (use_trans: 57)


def vq_args():
    # diffusion_rvqvae_trainer.py:89-103
    return types.SimpleNamespace(num_quantizers=6, shared_codebook=False, quantize_dropout_prob=0.2, mu=0.99)


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self
    from models.vq.model import RVQVAE
    torch.set_grad_enabled(False)
    out = {}
    for part, dim in PARTS:
        m = RVQVAE(vq_args(), dim, 512, 512, 512, 2, 2, 512, 3, 3, "relu", None).eval()
        synth.synth_fill_(m, seed=11)
        pose = synth.synth_vq_pose(part, dim)                                                        # (2, 64, dim)
        lat = m.map2latent(pose)                                                                   # (2, 16, 512)
        out[f"{part}.map2latent"] = lat.numpy()
        idx, all_codes = m.encode(pose)                                                            # (2,16,6), (6,2,512,16)
        out[f"{part}.encode.idx"] = idx.numpy()
        out[f"{part}.forward_decoder"] = m.forward_decoder(idx).numpy()                            # (2, 64, dim)
        rec = synth.synth_vq_rec_latent(m.state_dict(), part)
        xq, qidx, _, _ = m.quantizer(rec.clone().permute(0, 2, 1), sample_codebook_temp=0.5)
        out[f"{part}.quantizer.idx"] = qidx.numpy()
        out[f"{part}.quantizer.out"] = xq.numpy()                                                  # (2, 512, 16)
        y, commit, perp = m.latent2origin(rec.clone())
        out[f"{part}.latent2origin"] = y.numpy()                                                   # (2, 64, dim)
        out[f"{part}.commit"] = np.float32(commit)
        out[f"{part}.perplexity"] = np.float32(perp)
        out[f"{part}.state_keys"] = np.array([f"{k}:{'x'.join(map(str, v.shape))}" for k, v in m.state_dict().items()])
        print(part, "latent rms", float(lat.pow(2).mean().sqrt()), "rec rms", float(rec.pow(2).mean().sqrt()),
              "quantised rms", float(xq.pow(2).mean().sqrt()), "out rms", float(y.pow(2).mean().sqrt()),
              "distinct codes/layer", [int(qidx[..., q].unique().numel()) for q in range(6)])
    np.savez_compressed(os.path.join(HERE, "vq_outputs.npz"), **out)
    print("wrote", os.path.join(HERE, "vq_outputs.npz"), sum(v.nbytes for v in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
