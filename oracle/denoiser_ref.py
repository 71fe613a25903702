"""MDM denoiser restated on CPU torch, functional over a state_dict.  Oracle: test infra only.

Two restatements of reference models/denoiser.py:132-196 (and models/denoiser_h3d.py:148-221):

``mdm_forward``         op-for-op, "as written": every call re-runs the audio / word / seed
                        conditioning exactly like the reference does.  This is the CPU baseline
                        bench.py times, and the function pinned against the golden vectors.
``clip_conditioning`` + ``mdm_forward_folded``
                        the algebra the HIP kernels implement: the
                        timestep-independent conditioning is computed once per clip, and the affine
                        chain poseEmbedding -> input_process2 [-> input_process3] is folded into
                        one matrix ``A`` plus per-clip / per-step bias terms.

``variant``: "beatx" = models/denoiser.py (style only if use_motionclip), "h3d" = denoiser_h3d.py.
Eval-mode semantics by default (BatchNorm running stats, no DropPath, no Bernoulli cond-masking); ``train_bn=True`` gives the
audio encoder's BatchNorms their train() semantics - batch statistics, running buffers updated with momentum 0.1 - which is
what the reference's training step runs (DropPath and the style dropout are the random elements of train(): the golden
vectors of that mode are taken with DropPath's probability set to 0, tests/golden/make_golden.py).
"""
import torch
import torch.nn.functional as F

N_LAYERS, N_HEADS, D, ROT_GROUPS = 8, 4, 512, 8


# ------------------------------------------------------------------ pieces
def wav_encoder(sd, wav, prefix="WavEncoder.feat_extractor.", train_bn=False, new_buffers=None):
    """models/denoiser.py:304-322 + models/utils/layer.py:144-184.  wav (B, L, 2) -> (B, 128, 256).
    train_bn: nn.BatchNorm1d in train() mode (biased batch variance in the normalisation; the running buffers move by
    momentum 0.1 towards the batch mean / UNBIASED batch variance - written to ``new_buffers`` if given, sd is not modified)."""
    cfg = [(5, 1700, True), (6, 0, True), (1, 7, False), (6, 0, True), (1, 7, False), (3, 0, True)]
    x = wav.unsqueeze(1) if wav.dim() == 2 else wav.transpose(1, 2)

    def bn(z, p):
        if not train_bn:
            return F.batch_norm(z, sd[p + ".running_mean"], sd[p + ".running_var"],
                                sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)
        rm, rv = sd[p + ".running_mean"].detach().clone(), sd[p + ".running_var"].detach().clone()
        out = F.batch_norm(z, rm, rv, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, 1e-5)
        if new_buffers is not None:
            new_buffers[p + ".running_mean"], new_buffers[p + ".running_var"] = rm, rv
            new_buffers[p + ".num_batches_tracked"] = sd.get(p + ".num_batches_tracked", torch.zeros((), dtype=torch.long)) + 1
        return out

    for i, (stride, pad, down) in enumerate(cfg):
        p = f"{prefix}{i}."
        short = x
        z = F.conv1d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=stride, padding=pad)
        z = F.leaky_relu(bn(z, p + "bn1"), 0.01)
        z = bn(F.conv1d(z, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=7), p + "bn2")
        if down:
            short = bn(F.conv1d(short, sd[p + "downsample.0.weight"], sd[p + "downsample.0.bias"],
                                stride=stride, padding=pad), p + "downsample.1")
        x = F.leaky_relu(z + short, 0.01)
    return x.transpose(1, 2)


def time_embedding(sd, timesteps):
    """models/denoiser.py:231-245 -> (1, B, 512)."""
    e = sd["embed_timestep.sequence_pos_encoder.pe"][timesteps]          # (B, 1, 512)
    e = F.linear(e, sd["embed_timestep.time_embed.0.weight"], sd["embed_timestep.time_embed.0.bias"])
    e = F.linear(F.silu(e), sd["embed_timestep.time_embed.2.weight"], sd["embed_timestep.time_embed.2.bias"])
    return e.permute(1, 0, 2)


def rotary(sd, h):
    """models/denoiser.py:178-186,324-343.  h (B, T, 512): eight 64-wide groups, pairs (j, j+32)."""
    B, T, _ = h.shape
    g = h.view(B, T, ROT_GROUPS, -1).permute(0, 2, 1, 3).reshape(B * ROT_GROUPS, T, -1)
    pos = torch.arange(T).type_as(sd["rel_pos.inv_freq"])
    fr = torch.einsum("i,j->ij", pos, sd["rel_pos.inv_freq"])
    fr = torch.cat((fr, fr), dim=-1).to(h.dtype)
    half = g.shape[-1] // 2
    rot = torch.cat((-g[..., half:], g[..., :half]), dim=-1)
    g = g * fr.cos() + rot * fr.sin()
    return g.reshape(B, ROT_GROUPS, T, -1).permute(0, 2, 1, 3).reshape(B, T, -1)


def block(sd, x, i):
    """models/timm_transformer/transformer.py:83-104,145-151,195-198 (pre-LN ViT block, eval)."""
    p = f"mytimmblocks.{i}."
    B, N, C = x.shape
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    qkv = F.linear(h, sd[p + "attn.qkv.weight"]).reshape(B, N, 3, N_HEADS, C // N_HEADS).permute(2, 0, 3, 1, 4)
    a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], dropout_p=0.0)
    a = a.transpose(1, 2).reshape(B, N, C)
    x = x + F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def _style(sd, y, variant, use_motionclip, bs):
    """mask_cond at eval: denoiser.py:110-119 / denoiser_h3d.py:116-124."""
    if variant == "h3d":
        if y.get("uncond", False):
            return sd["uncon_text_embeddings"].repeat(bs, 1)
        s = y["style_feature"]
        return s.expand(bs, -1) if s.shape[0] == 1 and bs > 1 else s
    if use_motionclip:
        s = y["style_feature"]
        return torch.zeros_like(s) if y.get("uncond", False) else s
    return None


def _audio_word(y, variant):
    """denoiser_h3d.py:173-180: uncond_audio zeroes the waveform and the word ids (-> row 0)."""
    a, w = y["audio"], y["word"]
    if variant == "h3d" and y.get("uncond_audio", False):
        a, w = torch.zeros_like(a), torch.zeros_like(w)
    return a, w


# ------------------------------------------------------------------ as written
def mdm_forward(sd, x, timesteps, y, variant="beatx", use_motionclip=False, pool=4, taps=None, train_bn=False, new_buffers=None):
    """x (B, 1536, 1, T), timesteps (B,) int64 -> (B, 1536, 1, T).  denoiser.py:132-196.
    train_bn / new_buffers: the audio encoder's BatchNorms in train() mode, see wav_encoder."""
    bs, C, _, T = x.shape
    emb_t = time_embedding(sd, timesteps)                                           # :142
    emb_seed = F.linear(y["seed"].reshape(bs, -1), sd["embed_text.weight"], sd["embed_text.bias"])
    audio, word = _audio_word(y, variant)
    a_feat = wav_encoder(sd, audio, train_bn=train_bn, new_buffers=new_buffers).permute(1, 0, 2)   # :151
    w_feat = F.embedding(word, sd["text_pre_encoder_body.weight"])
    w_feat = F.linear(w_feat, sd["text_encoder_body.weight"], sd["text_encoder_body.bias"]).permute(1, 0, 2)
    at = F.linear(torch.cat([a_feat, w_feat], dim=2), sd["mix_audio_text.weight"], sd["mix_audio_text.bias"])
    at = F.avg_pool1d(at.permute(1, 2, 0), pool).permute(2, 0, 1)                   # :157 (T, B, 256)
    xt = x.reshape(bs, C, 1, T).permute(3, 0, 1, 2).reshape(T, bs, C)               # :258-264
    x_ = F.linear(xt, sd["input_process.poseEmbedding.weight"], sd["input_process.poseEmbedding.bias"])
    seq = torch.cat(((emb_seed + emb_t).repeat(T, 1, 1), x_, at), dim=2)            # :166-169
    seq = F.linear(seq, sd["input_process2.weight"], sd["input_process2.bias"])
    st = _style(sd, y, variant, use_motionclip, bs)
    if st is not None:                                                              # :172-174 / h3d :199
        seq = torch.cat((seq, st.unsqueeze(0).repeat(T, 1, 1)), dim=2)
        seq = F.linear(seq, sd["input_process3.weight"], sd["input_process3.bias"])
    h = rotary(sd, seq.permute(1, 0, 2))
    if taps is not None:
        taps["h0"] = h.clone()
    for i in range(N_LAYERS):
        h = block(sd, h, i)
        if taps is not None:
            taps[f"h{i + 1}"] = h.clone()
    out = F.linear(h.permute(1, 0, 2), sd["output_process.poseFinal.weight"], sd["output_process.poseFinal.bias"])
    return out.reshape(T, bs, C, 1).permute(1, 2, 3, 0)[..., :T]                    # :287-301,196


# ------------------------------------------------------------------ hoisted + folded
def fold_weights(sd, variant="beatx", use_motionclip=False):
    """SURVEY.md §8 a17.  With W2 = [W2a | W2b | W2c] (cols 0:512, 512:1024, 1024:1280):
        h = x_t^T A^T + cbias + at·W2c^T + (seed_emb + emb_t)·W2a^T
      A = W2b·Wp, cbias = W2b·bp + b2;  with input_process3 = [W3a | W3s], b3 everything above is
      left-multiplied by W3a and  style·W3s^T + b3  joins the per-clip term."""
    W2, b2 = sd["input_process2.weight"], sd["input_process2.bias"]
    Wp, bp = sd["input_process.poseEmbedding.weight"], sd["input_process.poseEmbedding.bias"]
    W2a, W2b, W2c = W2[:, :D], W2[:, D:2 * D], W2[:, 2 * D:]
    A, cbias = W2b @ Wp, W2b @ bp + b2
    has3 = variant == "h3d" or use_motionclip
    if has3:
        W3, b3 = sd["input_process3.weight"], sd["input_process3.bias"]
        W3a, W3s = W3[:, :D], W3[:, D:]
        A, cbias, W2a, W2c = W3a @ A, W3a @ cbias + b3, W3a @ W2a, W3a @ W2c
    else:
        W3s = None
    return {"A": A, "cbias": cbias, "W2a": W2a, "W2c": W2c, "W3s": W3s}


def clip_conditioning(sd, y, fw, variant="beatx", use_motionclip=False, pool=4):
    """Everything that does not depend on x_t or t, once per clip:
       -> cond (B, T, 512) = cbias + c_frame + seed term [+ style term]."""
    bs = y["seed"].shape[0]
    audio, word = _audio_word(y, variant)
    a_feat = wav_encoder(sd, audio)
    w_feat = F.linear(F.embedding(word, sd["text_pre_encoder_body.weight"]),
                      sd["text_encoder_body.weight"], sd["text_encoder_body.bias"])
    at = F.linear(torch.cat([a_feat, w_feat], dim=2), sd["mix_audio_text.weight"], sd["mix_audio_text.bias"])
    at = F.avg_pool1d(at.transpose(1, 2), pool).transpose(1, 2)                     # (B, T, 256)
    c_frame = at @ fw["W2c"].T
    seed_emb = F.linear(y["seed"].reshape(bs, -1), sd["embed_text.weight"], sd["embed_text.bias"])
    d = seed_emb @ fw["W2a"].T
    st = _style(sd, y, variant, use_motionclip, bs)
    if st is not None:
        d = d + st @ fw["W3s"].T
    return c_frame + (d + fw["cbias"]).unsqueeze(1)


def time_table(sd, fw, n=1000):
    """TE[t] = time_embed(pe[t])·W2a^T for every t: (n, 512)."""
    e = time_embedding(sd, torch.arange(n)).squeeze(0)
    return e @ fw["W2a"].T


def mdm_forward_folded(sd, fw, cond, te, x, timesteps, taps=None):
    bs, C, _, T = x.shape
    xt = x.reshape(bs, C, T).transpose(1, 2)                                        # (B, T, C)
    h = xt @ fw["A"].T + cond + te[timesteps].unsqueeze(1)
    h = rotary(sd, h)
    if taps is not None:
        taps["h0"] = h.clone()
    for i in range(N_LAYERS):
        h = block(sd, h, i)
        if taps is not None:
            taps[f"h{i + 1}"] = h.clone()
    out = F.linear(h, sd["output_process.poseFinal.weight"], sd["output_process.poseFinal.bias"])
    return out.transpose(1, 2).reshape(bs, C, 1, T)


def cast_sd(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
