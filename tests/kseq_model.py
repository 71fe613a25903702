"""Lane-level model of the wave-per-sequence step kernel `k_seq` (syntalker_amd/csrc/syn_seq.inc): test infrastructure.

It executes, for ONE sequence, exactly the dataflow the HIP kernel executes in one wave - the weight tape consumed
fragment by fragment in tape order, `v_mfma_f32_32x32x16_bf16` with its register layouts, accumulator registers used
directly as the next B operand (K permutation), attention entirely in registers, LayerNorm / softmax reductions over
registers + one lane^32 exchange, the fragment-order latent - with numpy arrays indexed [lane][register].  The CPU
tests run it against the oracle: that pins the tape packer (`syntalker_amd/tape.py`) and every layout decision of the
kernel before the GPU sees it.
"""
import numpy as np
import torch

LANES = 64


def bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def mfma32(a, b, c):
    """D = A.B + C for v_mfma_f32_32x32x16_bf16.  a, b: [64][8] (bf16 values as fp32), c: [64][16] fp32.
    A[i][k]: lane = 32*(k//8) + i, slot k%8.  B[k][j]: lane = 32*(k//8) + j, slot k%8.
    D[i][j]: lane = 32*((i//4)%2) + j, reg = (i%4) + 4*(i//8)."""
    A = np.zeros((32, 16), np.float32)
    B = np.zeros((16, 32), np.float32)
    for hi in range(2):
        A[:, 8 * hi:8 * hi + 8] = a[32 * hi:32 * hi + 32]
        B[8 * hi:8 * hi + 8, :] = b[32 * hi:32 * hi + 32].T
    Dm = A @ B
    out = c.copy()
    for hi in range(2):
        for v in range(16):
            i = (v & 3) + 8 * (v >> 2) + 4 * hi
            out[32 * hi:32 * hi + 32, v] += Dm[i, :]
    return out


def d_pair_as_operand(t, c):
    """Registers of quads 2c, 2c+1 of a D tile [64][16] -> bf16 operand fragment [64][8] (the accumulator IS the operand)."""
    return bf16(t[:, 8 * c:8 * c + 8])


class Wave:
    """One wave = one sequence.  tape: [n][64][8] float32 view of the bf16 tape; bias: [9][4096]."""

    def __init__(self, tape, bias):
        self.tape, self.bias, self.pos = tape, bias, 0
        self.lane = np.arange(LANES)
        self.hi, self.tok = self.lane >> 5, self.lane & 31

    def frag(self):
        f = self.tape[self.pos]
        self.pos += 1
        return f

    def bias_tile(self, set_, off, tile):
        """accumulator initial value of a non-swapped 32-feature tile: bias[off + 32 tile + 8q + 4hi + r]"""
        out = np.zeros((LANES, 16), np.float32)
        for v in range(16):
            out[:, v] = self.bias[set_, off + 32 * tile + 8 * (v >> 2) + 4 * self.hi + (v & 3)]
        return out

    def wide16(self, acc, bfrag, n_kc):
        """acc[t] += W-tile(t, kc) x bfrag(kc) for all 16 tiles; fragments in tape order [kc][tile]."""
        for kc in range(n_kc):
            b = bfrag(kc)
            for t in range(16):
                acc[t] = mfma32(self.frag(), b, acc[t])

    def wide(self, acc, bfrag, n_kc, split=0):
        """The same inside the blocks, in the fragment order of the kernel's `wide<KC, SPLIT>` (tape._wide_order): tiles 12..15 are
        the LDS-resident ones; [kc < split][0..11], [kc < split][12..15], whole k chunks, the last one with tiles 12..15 first."""
        def run(kc, tiles):
            b = bfrag(kc)
            for t in tiles:
                acc[t] = mfma32(self.frag(), b, acc[t])
        for kc in range(split):
            run(kc, range(12))
        for kc in range(split):
            run(kc, range(12, 16))
        for kc in range(split, n_kc):
            if n_kc > 1 and kc == n_kc - 1:
                run(kc, range(12, 16)); run(kc, range(12))
            else:
                run(kc, range(16))

    def pairs(self, init, bfrag, n_pairs, n_kc, done, swap=False):
        """Two tiles in flight, tape order [pair][kc][u]; done(tile index, tile) consumes each finished tile.
        swap: activation as A operand -> the tile comes out [token][feature] instead of [feature][token]."""
        for p in range(n_pairs):
            acc = [init(2 * p), init(2 * p + 1)]
            for kc in range(n_kc):
                b = bfrag(kc)
                for u in range(2):
                    w = self.frag()
                    acc[u] = mfma32(b, w, acc[u]) if swap else mfma32(w, b, acc[u])
            done(2 * p, acc[0]); done(2 * p + 1, acc[1])

    def row_stats(self, h):
        """h: 16 tiles [64][16] (feature-major D layout).  Per-token mean / rstd over the 512 features."""
        s = sum(t.sum(1) for t in h)
        q = sum((t * t).sum(1) for t in h)
        s = s + s[self.lane ^ 32]
        q = q + q[self.lane ^ 32]
        mean = s / 512.0
        var = np.maximum(q / 512.0 - mean * mean, 0.0)
        return mean.astype(np.float32), (1.0 / np.sqrt(var + 1e-5)).astype(np.float32)

    def step(self, xb_frag, cond_tok, te_row, rcos, rsin, gelu):
        """xb_frag: [48][2][64][8] bf16 latent fragments of the sequence; cond_tok: [32][512]; te_row: [512].
        Returns x0_hat in fp32 fragment order [48][4][64][4]."""
        # ---- input stage: h = rotary(x A^T + cond + te) ---------------------------------------------------------
        h = []
        for t in range(16):
            init = np.zeros((LANES, 16), np.float32)
            for v in range(16):
                f = 32 * t + 8 * (v >> 2) + 4 * self.hi + (v & 3)
                init[:, v] = cond_tok[self.tok, f] + te_row[f]
            h.append(init)
        self.wide16(h, lambda kc: xb_frag[kc >> 1][kc & 1], 96)
        for g in range(8):                       # 64-wide groups: pairs (j, j + 32) = tiles 2g and 2g + 1, same register
            u, w = h[2 * g], h[2 * g + 1]
            cs, sn = np.zeros_like(u), np.zeros_like(u)
            for v in range(16):
                jj = 8 * (v >> 2) + 4 * self.hi + (v & 3)
                cs[:, v], sn[:, v] = rcos[self.tok, jj], rsin[self.tok, jj]
            h[2 * g], h[2 * g + 1] = u * cs - w * sn, w * cs + u * sn
        scale = np.float32(0.08838834764831845 * 1.4426950408889634)
        zero = lambda t: np.zeros((LANES, 16), np.float32)
        for l in range(8):
            # ---- attention: XN = (h - mean) * rstd materialised as bf16 operand fragments (gain / shift are in the tape)
            mean, rstd = self.row_stats(h)
            xn = [d_pair_as_operand((h[kc >> 1] - mean[:, None]) * rstd[:, None], kc & 1) for kc in range(32)]
            ones = np.zeros((LANES, 8), np.float32)
            ones[:32, 0:2] = 1.0                                         # slots k = 0, 1 of the rank-1 bias update
            self.wide(h, lambda kc: ones, 1)                             # h += proj bias (+ folded v bias)
            for head in range(4):
                qb = {}
                self.pairs(lambda t: self.bias_tile(l, 128 * head, t), lambda kc: xn[kc], 2, 32,
                           lambda t, tile: qb.update({(t, 0): d_pair_as_operand(tile, 0), (t, 1): d_pair_as_operand(tile, 1)}))
                # S^T[key][query] = sum_d K[key][d] Q[query][d]: k tile as A (row = lane&31 = token = key), q tile as B
                st = {"s": zero(0)}
                def k_done(t, tile):
                    for c in range(2):
                        st["s"] = mfma32(d_pair_as_operand(tile, c), qb[(t, c)], st["s"])
                self.pairs(zero, lambda kc: xn[kc], 2, 32, k_done)
                s_ = st["s"]
                mx = s_.max(1)
                mx = np.maximum(mx, mx[self.lane ^ 32])
                p = bf16(np.exp2((s_ - mx[:, None]) * scale))
                sm = p.sum(1)
                sm = sm + sm[self.lane ^ 32]
                inv = (1.0 / sm).astype(np.float32)
                # O^T[d][query] = sum_key V^T[d][key] P^T[key][query]: v tile (swapped: [token=key][d]) as A, P as B
                ob = {}
                def v_done(t, tile):
                    ot = zero(0)
                    for c in range(2):
                        ot = mfma32(d_pair_as_operand(tile, c), p[:, 8 * c:8 * c + 8], ot)
                    ot = ot * inv[:, None]
                    ob[(t, 0)], ob[(t, 1)] = d_pair_as_operand(ot, 0), d_pair_as_operand(ot, 1)
                self.pairs(zero, lambda kc: xn[kc], 2, 32, v_done, swap=True)
                self.wide(h, lambda kc: ob[(kc >> 1, kc & 1)], 8)
            # ---- MLP -------------------------------------------------------------------------------------------
            mean, rstd = self.row_stats(h)
            xn = [d_pair_as_operand((h[kc >> 1] - mean[:, None]) * rstd[:, None], kc & 1) for kc in range(32)]
            self.wide(h, lambda kc: ones, 1)                             # h += fc2 bias
            for sl in range(8):
                hb = {}
                def f_done(t, tile):
                    g = gelu(tile)
                    hb[(t, 0)], hb[(t, 1)] = d_pair_as_operand(g, 0), d_pair_as_operand(g, 1)
                self.pairs(lambda t: self.bias_tile(l, 512 + 128 * sl, t), lambda kc: xn[kc], 2, 32, f_done)
                self.wide(h, lambda kc: hb[(kc >> 1, kc & 1)], 8, split=4)
        # ---- output stage ---------------------------------------------------------------------------------------
        out = np.zeros((48, 4, LANES, 4), np.float32)
        hbf = [d_pair_as_operand(h[kc >> 1], kc & 1) for kc in range(32)]
        def o_done(t, tile):
            for qd in range(4):
                out[t, qd] = tile[:, 4 * qd:4 * qd + 4]
        self.pairs(lambda t: self.bias_tile(8, 0, t), lambda kc: hbf[kc], 24, 32, o_done)
        assert self.pos == self.tape.shape[0], (self.pos, self.tape.shape)
        return out
