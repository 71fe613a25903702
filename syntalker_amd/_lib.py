"""ctypes binding of libsyn_hip.so (C ABI: include/syn_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``syntalker_amd/csrc/build.sh`` with
``hipcc --offload-arch=gfx950``.  There is NO fallback: if the shared object is missing or an entry
point fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SYN_HIP_LIB") or os.path.join(_HERE, "csrc", "libsyn_hip.so")   # env override: kernel A/B builds

SYN_LAYERS = 8
ABI_VERSION = 9             # include/syn_hip.h SYN_ABI_VERSION: a library built from other sources is refused at load time
EXPORTS = ("syn_version", "syn_last_error", "syn_denoise_step", "syn_denoise_steps", "syn_denoise_step_profile", "syn_pack_weight", "syn_pack_weight_t", "syn_to_token_major",
           "syn_from_token_major", "syn_axpby_rows", "syn_randn", "syn_linear", "syn_linear_pair", "syn_linear_and_pack", "syn_linear_res", "syn_linear_gelu", "syn_opt_blocks", "syn_opt_sqnorm", "syn_opt_scalars", "syn_opt_adam", "syn_test_gemm", "syn_test_attention", "syn_test_handoff",
           "syn_wav_encode", "syn_wav_workspace_bytes", "syn_wav_out_frames", "syn_linear_bwd_prep", "syn_embedding_wgrad", "syn_pack_weights", "syn_bn_chunks", "syn_bn_act_fwd", "syn_bn_act_bwd", "syn_bn_sums", "syn_bn_act_apply", "syn_bn_bwd_sums", "syn_bn_act_bwd_apply", "syn_conv1d_train_fwd", "syn_conv1d_train_fwd_tiles", "syn_conv1d_pack_split", "syn_conv1d_pack_split_many", "syn_conv1d_pack_bytes", "syn_conv1d_train_wgrad", "syn_conv1d_train_wgrad_pair", "syn_conv1d_wgrad_shares", "syn_conv1d_first_parts", "syn_conv1d_first_fwd", "syn_conv1d_first_wgrad", "syn_conv1d_first_wgrad_tail", "syn_conv1d_first_wgrad_bn_lin", "syn_cond_encode",
           "syn_vq_conv1d", "syn_vq_quantize", "syn_vq_quantize_groups", "syn_vq_codes",
           "syn_vq_workspace_bytes", "syn_vq_map2latent", "syn_vq_latent2origin", "syn_vq_forward_decoder",
           "syn_step_advance", "syn_steps_advance", "syn_prefers_fragment_order", "syn_x_to_fragment", "syn_x_from_fragment", "syn_ln_fwd", "syn_ln_bwd", "syn_gelu_fwd", "syn_gelu_bwd", "syn_attn_fwd", "syn_attn_bwd",
           "syn_axis_angle_to_rot6d", "syn_rot6d_to_axis_angle", "syn_rotary", "syn_linear_wgrad_rows", "syn_masked_smooth_l1",
           "syn_bn_finalize", "syn_bn_apply2", "syn_bn_block_bwd", "syn_conv1d_train_fwd_norm", "syn_conv1d_train_wgrad_norm", "syn_conv1d_first_tiles",
           "syn_conv1d_first_fwd_stats", "syn_test_mfma_rate", "syn_conv1d_train_dgrad_sum", "syn_conv1d_first_fwd2", "syn_conv1d_first_wgrad_bn",
           "syn_bn_bwd_stats", "syn_train_stack_fwd", "syn_train_stack_bwd", "syn_train_stack_wgrad",
           "syn_masked_smooth_l1_grad", "syn_rows_concat_bf16", "syn_embed_rows_bf16", "syn_bct_to_rows_bf16", "syn_rows_group_sum", "syn_rows_expand",
           "syn_colsum_parts", "syn_touch", "syn_conv1d_wgrad_sums", "syn_bn_finalize_pair", "syn_conv1d_train_fwd_pair")

# the `void syn_debug_*` switches of the header's diagnostics section (process-wide, A/B runs and scripts/ only)
DIAGNOSTICS = ("syn_debug_timing", "syn_debug_gemm_resident", "syn_debug_linear_tile", "syn_debug_conv_terms", "syn_debug_seq_skew", "syn_debug_seq_step")

vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64


class SynLayer(C.Structure):
    _fields_ = [("ln1_g", vp), ("ln1_b", vp), ("w_qkv", vp), ("w_proj", vp), ("b_proj", vp),
                ("ln2_g", vp), ("ln2_b", vp), ("w_fc1", vp), ("b_fc1", vp), ("w_fc2", vp), ("b_fc2", vp)]


class SynTrainBlockSave(C.Structure):
    _fields_ = [(n, vp) for n in ("h_attn", "mean_attn", "rstd_attn", "qkv", "xt_ln1", "xt_attn", "h_mlp", "mean_mlp", "rstd_mlp", "pre", "xt_ln2", "xt_gelu")]


class SynTrainStack(C.Structure):
    _fields_ = [("h_in", vp), ("h_out", vp), ("layer", SynLayer * SYN_LAYERS), ("save", SynTrainBlockSave * SYN_LAYERS), ("drop_path", vp),
                ("n_seq", i32), ("reserved", i32), ("sync", vp), ("xch", vp)]


class SynTrainBlockGrad(C.Structure):
    _fields_ = [(n, vp) for n in ("dyt_fc2", "dyt_fc1", "dyt_proj", "dyt_qkv", "part", "dw_fc2", "dw_fc1", "dw_proj", "dw_qkv",
                                   "d_ln2_g", "d_ln2_b", "d_fc2_b", "d_fc1_b", "d_ln1_g", "d_ln1_b", "d_proj_b")]


class SynTrainStackGrad(C.Structure):
    _fields_ = [("fwd", C.POINTER(SynTrainStack)), ("dh_out", vp), ("dh_in", vp), ("layer_t", SynLayer * SYN_LAYERS), ("grad", SynTrainBlockGrad * SYN_LAYERS),
                ("stash", vp), ("first_block", i32), ("last_block", i32)]


class SynModel(C.Structure):
    _fields_ = [("w_in", vp), ("te", vp), ("n_te", i32), ("rot_cos", vp), ("rot_sin", vp),
                ("layer", SynLayer * SYN_LAYERS), ("w_out", vp), ("b_out", vp),
                ("tape", vp), ("tape_bias", vp), ("tape_chunks", i32)]


class SynStep(C.Structure):
    _fields_ = [("n_clips", i32), ("n_variants", i32), ("m_tile", i32), ("reserved", i32),
                ("cond", vp), ("t_model", vp), ("cfg_w", vp),
                ("x_t", vp), ("x_t_bf16", vp), ("noise", vp), ("rng", vp), ("coef", vp), ("t_coef", vp),
                ("x_next", vp), ("x_next_bf16", vp), ("pred_x0", vp),
                ("ws_h", vp), ("ws_xn", vp), ("ws_q", vp), ("ws_k", vp), ("ws_vt", vp), ("ws_o", vp),
                ("ws_hid", vp), ("ws_hc", vp), ("ws_sync", vp), ("ws_x0v", vp), ("ws_xch", vp),
                ("x_fragment_order", i32), ("cfg_w_clip_stride", i32)]


SYN_OPT_MAX = 64
SYN_CONV_PACK_MAX = 40


class SynOptList(C.Structure):
    _fields_ = [("p", vp * SYN_OPT_MAX), ("g", vp * SYN_OPT_MAX), ("m", vp * SYN_OPT_MAX), ("v", vp * SYN_OPT_MAX),
                ("numel", i32 * SYN_OPT_MAX), ("n", i32)]


class SynConcatSrc(C.Structure):
    _fields_ = [("p", vp), ("p2", vp), ("width", i32), ("ld", i32), ("row_div", i32), ("pool", i32)]


class SynWgradSumJob(C.Structure):
    _fields_ = [("part", vp), ("dw", vp), ("n_clips", i32), ("l_out", i32), ("cin", i32), ("stride", i32), ("cout", i32), ("first_layer", i32),
                ("share_pitch", i32), ("reserved", i32)]


class SynBnFinalizeJob(C.Structure):
    _fields_ = [("part", vp), ("chunks", i32), ("channels", i32), ("rows", i64), ("gamma", vp), ("beta", vp), ("eps", C.c_float), ("momentum", C.c_float),
                ("run_mean", vp), ("run_var", vp), ("conv_bias", vp), ("stats", vp), ("affine", vp)]


class SynConvPackReq(C.Structure):
    _fields_ = [("w", vp), ("out_hi", vp), ("out_lo", vp), ("cout", i32), ("cin", i32), ("stride", i32), ("transposed", i32)]


class SynWavConv(C.Structure):
    _fields_ = [("w", vp), ("bias", vp)]


class SynWavEnc(C.Structure):
    _fields_ = [("cin", i32), ("reserved", i32), ("w_first", vp), ("conv", SynWavConv * 11)]


class SynCondWeights(C.Structure):
    _fields_ = [("gt", vp), ("tw", vp), ("st", vp), ("c0", vp), ("vocab", i32), ("seed_dim", i32), ("style_dim", i32), ("reserved", i32)]


class SynVqConv(C.Structure):
    _fields_ = [("w_packed", vp), ("bias", vp), ("cin", i32), ("cout", i32), ("cout_valid", i32), ("taps", i32),
                ("stride", i32), ("dil", i32), ("pad", i32), ("up", i32), ("relu_in", i32), ("relu_out", i32)]


class SynVqModel(C.Structure):
    _fields_ = [("pose_dim", i32), ("reserved", i32), ("enc", SynVqConv * 16), ("dec", SynVqConv * 17),
                ("codebooks", vp), ("codebooks_t", vp), ("code_sq", vp)]


class SynHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library (once).  Raises SynHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SynHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback for the denoising path.")
    lib = C.CDLL(LIB_PATH)
    lib.syn_version.restype = C.c_int
    if lib.syn_version() != ABI_VERSION:
        raise SynHipError(f"{LIB_PATH} has ABI version {lib.syn_version()}, this package needs {ABI_VERSION}: rebuild it "
                          "(`python -c 'import __graft_entry__ as g; g.build()'`)")
    lib.syn_last_error.restype = C.c_char_p
    lib.syn_denoise_step.argtypes = [C.POINTER(SynModel), C.POINTER(SynStep), vp]
    lib.syn_denoise_steps.argtypes = [C.POINTER(SynModel), C.POINTER(SynStep), C.c_int32, C.c_int32, C.c_int32, vp]
    lib.syn_denoise_step_profile.argtypes = [C.POINTER(SynModel), C.POINTER(SynStep), vp, vp, vp]
    lib.syn_pack_weight.argtypes = [vp, i32, i32, vp, vp]
    lib.syn_pack_weight_t.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.syn_to_token_major.argtypes = [vp, i32, vp, vp, vp]
    lib.syn_from_token_major.argtypes = [vp, i32, vp, vp]
    lib.syn_x_to_fragment.argtypes = [vp, i32, vp, vp, vp]
    lib.syn_x_from_fragment.argtypes = [vp, i32, vp, vp]
    lib.syn_prefers_fragment_order.argtypes = [i32, i32]
    lib.syn_axpby_rows.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp]
    lib.syn_randn.argtypes = [vp, i64, u64, u64, i64, vp]
    lib.syn_linear.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    lib.syn_linear_and_pack.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp]
    lib.syn_linear_pair.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, i32, i32, i32, vp, vp, i32, i32, vp, vp]
    lib.syn_linear_res.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
    lib.syn_linear_gelu.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    f32 = C.c_float
    lib.syn_conv1d_pack_split_many.argtypes = [vp, i32, vp]
    lib.syn_opt_blocks.argtypes = [C.POINTER(SynOptList)]
    lib.syn_opt_blocks.restype = i32
    lib.syn_opt_sqnorm.argtypes = [C.POINTER(SynOptList), vp, vp]
    lib.syn_opt_scalars.argtypes = [vp, i32, f32, vp, f32, f32, f32, vp, vp, vp]
    lib.syn_opt_adam.argtypes = [C.POINTER(SynOptList), vp, f32, f32, f32, f32, vp]
    lib.syn_test_gemm.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.syn_test_attention.argtypes = [vp, vp, vp, i32, vp, vp]
    lib.syn_test_handoff.argtypes = [vp, vp, vp, i64, vp, i32, i32, i32, vp]
    lib.syn_step_advance.argtypes = [vp, vp, vp, i32, vp, i32, vp]
    lib.syn_steps_advance.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp]
    lib.syn_ln_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.syn_ln_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.syn_gelu_fwd.argtypes = [vp, vp, vp, i64, vp]
    lib.syn_gelu_bwd.argtypes = [vp, vp, vp, i64, vp]
    lib.syn_linear_wgrad_rows.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.syn_masked_smooth_l1.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.syn_masked_smooth_l1_grad.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
    lib.syn_rows_concat_bf16.argtypes = [C.POINTER(SynConcatSrc), i32, i32, i32, vp, vp]
    lib.syn_embed_rows_bf16.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    lib.syn_bct_to_rows_bf16.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.syn_rows_group_sum.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.syn_rows_expand.argtypes = [vp, i32, i32, i32, C.c_float, i32, vp, vp]
    lib.syn_colsum_parts.argtypes = [vp, i32, i32, vp, vp]
    lib.syn_touch.argtypes = [vp, i64, vp]
    lib.syn_conv1d_wgrad_sums.argtypes = [C.POINTER(SynWgradSumJob), i32, vp]
    lib.syn_conv1d_train_fwd_pair.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.syn_bn_finalize_pair.argtypes = [C.POINTER(SynBnFinalizeJob), C.POINTER(SynBnFinalizeJob), vp]
    lib.syn_rotary.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.syn_axis_angle_to_rot6d.argtypes = [vp, i64, vp, vp]
    lib.syn_rot6d_to_axis_angle.argtypes = [vp, i64, vp, vp]
    lib.syn_debug_conv_terms.argtypes = [i32]
    lib.syn_debug_conv_terms.restype = None
    lib.syn_attn_fwd.argtypes = [vp, vp, vp, i32, vp]
    lib.syn_attn_bwd.argtypes = [vp, vp, vp, i32, vp]
    lib.syn_wav_encode.argtypes = [C.POINTER(SynWavEnc), vp, i32, i32, vp, vp, vp]
    lib.syn_cond_encode.argtypes = [C.POINTER(SynCondWeights), vp, vp, vp, vp, i32, vp, vp, vp]
    lib.syn_bn_chunks.argtypes = [C.c_int64]
    lib.syn_bn_act_fwd.argtypes = [vp, vp, C.c_int64, i32, vp, vp, C.c_float, C.c_float, vp, vp, vp, i32, vp, i32, vp, vp, vp]
    lib.syn_bn_act_bwd.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64, i32, i32, vp, vp, vp, vp, vp]
    lib.syn_bn_sums.argtypes = [vp, C.c_int64, i32, vp, i32, vp, vp]
    lib.syn_bn_act_apply.argtypes = [vp, vp, C.c_int64, C.c_int64, i32, vp, vp, C.c_float, C.c_float, vp, vp, vp, i32, vp, vp, vp, vp]
    lib.syn_bn_bwd_sums.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64, i32, i32, vp, vp, vp]
    lib.syn_bn_act_bwd_apply.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int64, C.c_int64, i32, i32, vp, vp, vp, vp, vp]
    lib.syn_bn_finalize.argtypes = [vp, i32, C.c_int64, i32, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp]
    lib.syn_bn_apply2.argtypes = [vp, vp, vp, vp, C.c_int64, i32, i32, vp, vp]
    lib.syn_bn_block_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int64, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.syn_conv1d_train_fwd_norm.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, vp, vp]
    lib.syn_conv1d_train_wgrad_norm.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp]
    lib.syn_conv1d_first_tiles.argtypes = [i32, i32]
    lib.syn_conv1d_first_fwd_stats.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.syn_test_mfma_rate.argtypes = [i32, vp, vp, vp]
    lib.syn_train_stack_fwd.argtypes = [C.POINTER(SynTrainStack), vp]
    lib.syn_train_stack_bwd.argtypes = [C.POINTER(SynTrainStackGrad), vp]
    lib.syn_train_stack_wgrad.argtypes = [C.POINTER(SynTrainStackGrad), vp]
    lib.syn_conv1d_first_fwd2.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.syn_conv1d_first_wgrad_bn.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.syn_conv1d_first_wgrad_tail.argtypes = [vp] * 10 + [i32] * 6 + [vp, vp, vp]
    lib.syn_conv1d_first_wgrad_bn_lin.argtypes = [vp] * 5 + [i32] * 6 + [vp, vp, vp, vp]
    lib.syn_bn_bwd_stats.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64, i32, i32, vp, vp, vp]
    lib.syn_conv1d_train_dgrad_sum.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.syn_linear_bwd_prep.argtypes = [vp, i32, i32, C.c_float, i32, i32, vp, i32, vp, vp, vp, vp]
    lib.syn_pack_weights.argtypes = [vp, i32, C.c_int64, vp]
    lib.syn_embedding_wgrad.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    lib.syn_conv1d_train_wgrad.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.syn_conv1d_wgrad_shares.argtypes = [i32, i32, i32, i32]
    lib.syn_conv1d_train_wgrad_pair.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.syn_conv1d_first_parts.argtypes = [i32, i32]
    lib.syn_conv1d_first_fwd.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.syn_conv1d_first_wgrad.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.syn_conv1d_pack_bytes.argtypes = [i32, i32, i32, i32]
    lib.syn_conv1d_pack_split.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp]
    lib.syn_conv1d_train_fwd.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp]
    lib.syn_conv1d_train_fwd_tiles.argtypes = [i32, i32, i32, i32, i32, i32]
    lib.syn_wav_workspace_bytes.argtypes = [i32, i32]
    lib.syn_wav_out_frames.argtypes = [i32]
    lib.syn_vq_conv1d.argtypes = [C.POINTER(SynVqConv), vp, vp, vp, i32, vp, i32, i32, i32, vp]
    lib.syn_vq_quantize_groups.argtypes = [i32]
    lib.syn_vq_quantize.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.syn_vq_codes.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    lib.syn_vq_workspace_bytes.argtypes = [i32, i32, i32]
    lib.syn_vq_map2latent.argtypes = [C.POINTER(SynVqModel), vp, i32, i32, vp, vp, vp]
    lib.syn_vq_latent2origin.argtypes = [C.POINTER(SynVqModel), vp, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.syn_vq_forward_decoder.argtypes = [C.POINTER(SynVqModel), vp, i32, i32, i32, vp, vp, vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("syn_version", "syn_last_error", "syn_wav_workspace_bytes", "syn_vq_workspace_bytes", "syn_conv1d_pack_bytes"):
            fn.restype = C.c_int
    lib.syn_wav_workspace_bytes.restype = C.c_int64
    lib.syn_conv1d_pack_bytes.restype = C.c_int64
    lib.syn_vq_workspace_bytes.restype = C.c_int64
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().syn_last_error()
        raise SynHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream(device=None):
    """Raw hipStream_t of torch's current stream on `device` (a tensor's device; default: the current device).
    The library launches on the CURRENT HIP device (kernel attributes, device queries): a tensor that lives on another GPU than
    the one the calling thread has selected is refused here instead of being launched onto a stream that is not current."""
    import torch
    if device is not None:
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
            raise SynHipError(f"tensors on {dev} but the current device is cuda:{torch.cuda.current_device()}: select the device first "
                              "(torch.cuda.set_device / `with torch.cuda.device(...)`, as nn.DataParallel and DDP do)")
    return torch.cuda.current_stream(device).cuda_stream
