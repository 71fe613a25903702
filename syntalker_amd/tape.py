"""Weight tape of the wave-per-sequence step kernel `k_seq` (syn_seq.inc).

`k_seq` gives every wave ONE sequence (32 tokens x 512 features, fp32 residual stream in registers for the whole
step) and lets the four waves of a workgroup share ONE weight stream: the step's weights are laid out once, at load
time, as a single contiguous "tape" of 1 KB MFMA operand fragments in exactly the order the kernel consumes them,
so the producer side of the kernel is a linear DMA (`global_load_lds`) of the tape into a ring in LDS.

Fragment = A operand of `v_mfma_f32_32x32x16_bf16`: 64 lanes x 8 bf16; lane l = 32*hi + i holds row i of a
32-feature tile.  Every activation the kernel multiplies with is the D (accumulator) layout of a previous MFMA used
directly as the next B operand: lane (hi, token) holds features {8q + 4hi + r}, so slot e of lane hi of the k-chunk
kc carries  k = 16 kc + 8 (e >> 2) + 4 hi + (e & 3)  instead of the natural 16 kc + 8 hi + e; the tape applies the
same permutation to the weights' K index (a contraction does not care in which order k is visited).

Fragment orders inside a piece (every piece is a multiple of 16 fragments = one 16 KB ring chunk):
  "wide16" [kc][tile 0..15]       all 16 tiles of the residual stream are live accumulators (input stage)
  "wide"   [kc][tile 0..15], the last k chunk as [tile 12..15][tile 0..11]: inside the blocks the wave keeps residual tiles
                                  12..15 in a private LDS slab between the pieces (the accumulator file holds 256 registers: twelve
                                  residual tiles + the q / k / v / fc1 accumulators); a wide piece loads them early and, with
                                  them first in its last k chunk, stores them in the shadow of its last twelve MFMAs.  fc2 pieces
                                  (split = 4) defer the slab tiles of their first four k chunks: [kc < 4][0..11], [kc < 4][12..15], ...
  "pair"   [pair][kc][u = 0, 1]   two tiles in flight, each finished tile is consumed at once (q, k, v, fc1, output)
Tape order:
    input   A (512 x 1536)                                                   wide16  96 kc x 16 = 1536
    block l           proj bias as a rank-1 MFMA update (K chunk 0 = [hi | lo] bf16 split of the bias)  wide, 1 kc = 16
    block l, head h   q_h, k_h, v_h (128 x 512 each, LayerNorm-1 gain folded in)   pair  3 x 128
                      proj[:, 128h:128h+128] (512 x 128)                           wide  8 kc = 128
    block l           fc2 bias, rank-1 update                                      wide, 1 kc = 16
    block l, slice s  fc1[128s:128s+128] (128 x 512, LayerNorm-2 gain folded in)  pair  128          (8 slices)
                      fc2[:, 128s:128s+128] (512 x 128)                           wide  8 kc = 128
    output  Wout (1536 x 512)                                                pair  24 pairs x 32 kc x 2 = 1536
  = 1536 + 8 * (4096 + 32) + 1536 = 36 096 fragments (35.25 MB), followed by the first LOOK_CHUNKS chunks once more: the
kernel's DMA runs LOOK_CHUNKS chunks ahead of the MFMAs, and in a multi-step launch the look-ahead of a step's tail is the
next step's head - with the head repeated behind the tail the producer never tests for the wrap, it restarts behind the
chunks the ring already holds when a step begins.

Biases.  Those that start an accumulator (q, fc1, output) are read from LDS-resident sets (fp32, 2048 floats = 8 KB
each, set l for block l, set 8 for the output stage):
    [0:512)     q bias of the 4 heads = Wq . beta1   (LayerNorm-1 shift folded; the k bias cancels in the softmax and
                the v bias passes through the attention unchanged, so it is folded into the proj bias)
    [512:1536)  b_fc1 + W1 . beta2
    set 8: [0:1536) b_out
Those added to the residual stream (proj: b_proj + Wproj . (Wv . beta1); fc2: b_fc2) are applied by the matrix cores as a
rank-1 update h += bias (x) 1 (the "bias" pieces of the tape): no accumulator round trip through the vector ALU.
(reference: models/timm_transformer/transformer.py:83-104,145-151,195-198; models/denoiser.py:188-195.)
"""
from __future__ import annotations

import os as _os

import torch

D, FF, C, HEADS, LAYERS = 512, 1024, 1536, 4, 8
CHUNK_FRAGS = 16       # fragments per ring chunk of the kernel (syn_seq.inc SEQ_CHUNK); syn_model.tape_chunks counts these
TAPE_FRAGS = 1536 + LAYERS * (4096 + 32) + 1536
LOOK_CHUNKS = 4         # ring slots ahead of the one being consumed (syn_seq.inc kLook): the tape's head is repeated at its end
TAPE_ALLOC_FRAGS = TAPE_FRAGS + LOOK_CHUNKS * CHUNK_FRAGS
BIAS_SET = 2048
N_BIAS_SETS = LAYERS + 1


def frag_tiles(w: torch.Tensor) -> torch.Tensor:
    """w [32*NT][16*KC] (any float dtype) -> bf16 fragments [KC][NT][64 lanes][8]."""
    n, k = w.shape
    nt, kc = n // 32, k // 16
    assert nt * 32 == n and kc * 16 == k, (n, k)
    # k = 16 kc + 8 eh + 4 hi + el ; lane = 32 hi + i ; slot e = 4 eh + el
    v = w.reshape(nt, 32, kc, 2, 2, 4)                 # [nt][i][kc][eh][hi][el]
    v = v.permute(2, 0, 4, 1, 3, 5)                    # [kc][nt][hi][i][eh][el]
    return v.reshape(kc, nt, 64, 8).to(torch.bfloat16).contiguous()


def wide16(w: torch.Tensor) -> torch.Tensor:
    """[kc][tile] order, flattened to [n][64][8]."""
    return frag_tiles(w).reshape(-1, 64, 8)


REG_TILES = 12        # residual tiles the kernel keeps in registers; tiles 12..15 live in the wave's LDS slab (syn_seq.inc kRegTiles)


def _wide_order(f: torch.Tensor, split: int = 0) -> torch.Tensor:
    """fragments [kc][16 tiles] -> the order `wide<KC, SPLIT>` of the kernel consumes: [kc < split][register tiles 0..11], then
    [kc < split][slab tiles 12..15], then whole k chunks [tile 0..15] - the last one slab tiles first (their stores ride in the gaps of
    the twelve MFMAs that follow)."""
    kc = f.shape[0]
    out = []
    if split:
        out += [f[:split, :REG_TILES].reshape(-1, 64, 8), f[:split, REG_TILES:].reshape(-1, 64, 8)]
    for k in range(split, kc):
        if kc > 1 and k == kc - 1:
            out += [f[k, REG_TILES:], f[k, :REG_TILES]]
        else:
            out.append(f[k])
    return torch.cat(out, 0)


def wide(w: torch.Tensor, split: int = 0) -> torch.Tensor:
    """512-row weight slice accumulated into the residual stream."""
    assert w.shape[0] == 512
    return _wide_order(frag_tiles(w), split)


def bias_piece(b: torch.Tensor) -> torch.Tensor:
    """h += b (x) 1 on the matrix cores: one K chunk whose slots k = 0, 1 (natural order: lane hi = 0, e = 0, 1) carry the
    bf16 [hi | lo] split of b (16 mantissa bits); the kernel's B operand for it is 1.0 in those two slots."""
    hi = b.to(torch.bfloat16)
    lo = (b - hi.double()).to(torch.bfloat16)
    f = torch.zeros(1, 16, 64, 8, dtype=torch.bfloat16, device=b.device)
    f[0, :, :32, 0] = hi.reshape(16, 32)
    f[0, :, :32, 1] = lo.reshape(16, 32)
    return _wide_order(f)


def pair(w: torch.Tensor) -> torch.Tensor:
    """[pair][kc][u] order: tiles 2p and 2p + 1 interleaved k-chunk by k-chunk."""
    f = frag_tiles(w)                                   # [kc][nt][64][8]
    kc, nt = f.shape[0], f.shape[1]
    return f.reshape(kc, nt // 2, 2, 64, 8).permute(1, 0, 2, 3, 4).reshape(-1, 64, 8)


def build_tape(sd: dict, A: torch.Tensor):
    """-> (tape bf16 [TAPE_ALLOC_FRAGS][64][8]: the TAPE_FRAGS fragments of a step + its first LOOK_CHUNKS chunks again,
    bias fp32 [9][2048]) on the device of the weights.

    sd: MDM state_dict view (mytimmblocks.*, output_process.poseFinal.*); A: folded input matrix (512 x 1536)."""
    dev = A.device
    f64 = lambda t: t.detach().to(dev).double()
    pieces = [wide16(f64(A))]
    bias = torch.zeros(N_BIAS_SETS, BIAS_SET, dtype=torch.float64, device=dev)
    for l in range(LAYERS):
        p = f"mytimmblocks.{l}."
        g1, b1n = f64(sd[p + "norm1.weight"]), f64(sd[p + "norm1.bias"])
        g2, b2n = f64(sd[p + "norm2.weight"]), f64(sd[p + "norm2.bias"])
        wqkv, wproj = f64(sd[p + "attn.qkv.weight"]), f64(sd[p + "attn.proj.weight"])
        w1, w2 = f64(sd[p + "mlp.fc1.weight"]), f64(sd[p + "mlp.fc2.weight"])
        wq, wk, wv = wqkv[:D], wqkv[D:2 * D], wqkv[2 * D:]
        pieces.append(bias_piece(f64(sd[p + "attn.proj.bias"]) + wproj @ (wv @ b1n)))
        for h in range(HEADS):
            r = slice(128 * h, 128 * h + 128)
            pieces += [pair(wq[r] * g1[None, :]), pair(wk[r] * g1[None, :]), pair(wv[r] * g1[None, :]), wide(wproj[:, r])]
        pieces.append(bias_piece(f64(sd[p + "mlp.fc2.bias"])))
        for c in range(FF // 128):
            r = slice(128 * c, 128 * c + 128)
            pieces += [pair(w1[r] * g2[None, :]), wide(w2[:, r], split=4)]
        bias[l, 0:512] = wq @ b1n
        bias[l, 512:1536] = f64(sd[p + "mlp.fc1.bias"]) + w1 @ b2n
    wout = f64(sd["output_process.poseFinal.weight"])
    pieces.append(pair(wout))
    bias[LAYERS, 0:C] = f64(sd["output_process.poseFinal.bias"])
    tape = torch.cat(pieces, 0)
    assert tape.shape[0] == TAPE_FRAGS, tape.shape
    tape = torch.cat([tape, tape[:LOOK_CHUNKS * CHUNK_FRAGS]], 0).contiguous()
    return tape, bias.float().contiguous()


# ---- latent layout of the kernel ("fragment order") -----------------------------------------------------------
# fp32: [seq][nf = channel/32][q][lane = 32 hi + token][r]     channel = 32 nf + 8 q + 4 hi + r
# bf16: [seq][nf][c][lane][e]                                   channel = 32 nf + 16 c + 8 (e >> 2) + 4 hi + (e & 3)
# i.e. exactly what a lane holds after the output GEMM (fp32) and what it needs as B operand of the input GEMM (bf16).
def to_fragment_order(x_btc: torch.Tensor) -> torch.Tensor:
    """token-major (B, 32, 1536) -> fp32 fragment order (B, 48, 4, 64, 4).  (Host-side reference of `syn_x_import`.)"""
    b = x_btc.shape[0]
    v = x_btc.reshape(b, 32, C // 32, 4, 2, 4)          # [b][token][nf][q][hi][r]
    return v.permute(0, 2, 3, 4, 1, 5).reshape(b, C // 32, 4, 64, 4).contiguous()


def from_fragment_order(x_frag: torch.Tensor) -> torch.Tensor:
    b = x_frag.shape[0]
    v = x_frag.reshape(b, C // 32, 4, 2, 32, 4)         # [b][nf][q][hi][token][r]
    return v.permute(0, 4, 1, 2, 3, 5).reshape(b, 32, C).contiguous()


def to_fragment_order_bf16(x_btc: torch.Tensor) -> torch.Tensor:
    """token-major (B, 32, 1536) -> bf16 B-operand fragments (B, 48, 2, 64, 8)."""
    b = x_btc.shape[0]
    v = x_btc.reshape(b, 32, C // 32, 2, 2, 2, 4)       # [b][token][nf][c][eh][hi][el]
    return v.permute(0, 2, 3, 5, 1, 4, 6).reshape(b, C // 32, 2, 64, 8).to(torch.bfloat16).contiguous()
