/*
 * syn_hip.h — C ABI of libsyn_hip.so: the MI355X (gfx950) denoising-step kernels for SynTalker.
 *
 * The reference (RobinWitch/SynTalker) is pure Python: it has no FFI of its own.  This ABI sits
 * beneath the two Python seams the reference's drivers use:
 *     models/denoiser.py:132      MDM.forward(x, timesteps, y)            -> syn_denoise_step (c_x0=1, c_xt=0, sigma=0)
 *     diffusion/gaussian_diffusion.py:505   GaussianDiffusion.p_sample    -> syn_denoise_step (DDPM coefficients)
 *     diffusion/gaussian_diffusion.py:741   GaussianDiffusion.ddim_sample -> syn_denoise_step (DDIM coefficients)
 *     diffusion/cfg_sampler.py:10-167       the four CFG wrappers         -> syn_denoise_step with n_variants > 1
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no exceptions; every entry point returns 0 on success, <0 on error;
 *     syn_last_error() returns a thread-local message for the last failure.
 *   - every pointer is a DEVICE pointer owned by the caller unless stated otherwise;
 *     nothing is allocated, freed or synchronised inside an entry point (syn_denoise_step_profile excepted), so
 *     all of them can be captured in a hipGraph.  `stream` is a hipStream_t (NULL = default stream).
 *   - "token-major" latent = [clip][frame 0..31][channel 0..1535]; the reference's layout
 *     (B, 1536, 1, 32) is "channel-major".  The sampling loop keeps x token-major across steps.
 *   - bf16 buffers are passed as void*; "packed" weights are in MFMA fragment order, produced
 *     by syn_pack_weight.
 */
#ifndef SYN_HIP_H
#define SYN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYN_ABI_VERSION 9     /* 9: syn_conv1d_wgrad_shares takes the layer's output channels (the share count follows the number of gradient slices a layer is cut in);
                                * syn_conv1d_first_wgrad_tail, syn_bn_block_bwd with dy = NULL, syn_conv1d_train_wgrad_pair, syn_wgrad_sum_job.share_pitch, syn_conv1d_first_wgrad_bn_lin;
                                * 8: the seams of the training step (syn_rows_concat_bf16, syn_embed_rows_bf16, syn_bct_to_rows_bf16, syn_rows_group_sum, syn_touch,
                                * syn_masked_smooth_l1 / _grad on the output Linear's rows), syn_linear_bwd_prep reads a strided / row-repeated dy, syn_pack_job.src_dim,
                                * syn_linear_pair on 128-column tiles, syn_embedding_wgrad ld; superseded kernels and their switches removed;
                                * 7: fused tail of the audio encoder's BasicBlock in training (syn_bn_finalize / syn_bn_apply2 / syn_bn_block_bwd,
                                * syn_conv1d_train_fwd_norm / _wgrad_norm, syn_conv1d_first_fwd_stats / _tiles), syn_test_mfma_rate;
                                * 6: pose formats either side of the RVQ-VAEs - syn_axis_angle_to_rot6d, syn_rot6d_to_axis_angle;
                                * 5: training block entry points - bf16 outputs of syn_ln_fwd / syn_gelu_fwd / syn_attn_fwd, syn_linear_res (residual + DropPath factor
                                * in the GEMM's epilogue), syn_linear_bwd_prep row_scale, syn_linear_pair bias_grad;
                                * 4: syn_model.tape carries its first 4 chunks again behind the last (no wrap test in k_seq's weight stream);
                                * 3: training entry points reworked (syn_ln_bwd add, syn_bn_act_* ws_chunks / beta, syn_conv1d_train_fwd bn_part,
                                * syn_linear_bwd_prep colsum; new: syn_linear_pair / _and_pack, syn_pack_weights, syn_embedding_wgrad, syn_conv1d_first_*) */
#define SYN_D        512   /* hidden width               (models/denoiser.py:19)  */
#define SYN_T        32    /* latent frames per clip     (128 pose frames / 4)    */
#define SYN_C        1536  /* latent channels            (models/denoiser.py:37)  */
#define SYN_FF       1024  /* MLP width                  (models/denoiser.py:20)  */
#define SYN_HEADS    4     /* attention heads x 128      (models/denoiser.py:22)  */
#define SYN_LAYERS   8     /* transformer blocks         (models/denoiser.py:21)  */

int         syn_version(void);
const char* syn_last_error(void);

/* One transformer block (models/timm_transformer/transformer.py:154-198). */
typedef struct syn_layer {
    const float* ln1_g;  const float* ln1_b;      /* norm1 (512)                               */
    const void*  w_qkv;                           /* packed bf16, attn.qkv.weight (1536 x 512)  */
    const void*  w_proj; const float* b_proj;     /* packed bf16 (512 x 512), bias (512)        */
    const float* ln2_g;  const float* ln2_b;      /* norm2 (512)                                */
    const void*  w_fc1;  const float* b_fc1;      /* packed bf16 (1024 x 512), bias (1024)      */
    const void*  w_fc2;  const float* b_fc2;      /* packed bf16 (512 x 1024), bias (512)       */
} syn_layer;

/* Everything the step needs that depends only on the weights. */
typedef struct syn_model {
    const void*  w_in;      /* packed bf16 (512 x 1536): folded input matrix A   */
    const float* te;        /* [n_te][512] time_embed(pe[t]) . W2a^T, row = ORIGINAL timestep    */
    int32_t      n_te;
    const float* rot_cos;   /* [32][32] cos(pos * inv_freq[j])   (models/denoiser.py:324-343)    */
    const float* rot_sin;   /* [32][32]                                                          */
    syn_layer    layer[SYN_LAYERS];
    const void*  w_out;     /* packed bf16 (1536 x 512): output_process.poseFinal.weight         */
    const float* b_out;     /* (1536)                                                            */
    /* the same weights as ONE contiguous tape of 1 KB MFMA fragments in consumption order + per-block bias sets, for the
     * wave-per-sequence step kernel (large batches; host: syntalker_amd/tape.py).  NULL = that kernel is never chosen. */
    const void*  tape;        /* bf16 [tape_chunks + 4][16 fragments][64 lanes][8] (1 KB fragments): the step's chunks followed by the
                               * first 4 of them once more (ABI 4: the kernel's DMA look-ahead runs across the step boundary)      */
    const float* tape_bias;   /* [9][4096]                                                        */
    int32_t      tape_chunks; /* 2256 chunks of 16 fragments (36 096 fragments, 35.25 MB)           */
} syn_model;

/* One denoising step over n_clips clips, each evaluated under n_variants conditionings
 * (classifier-free guidance as ONE fused batch of n_variants*n_clips sequences). */
typedef struct syn_step {
    int32_t n_clips;        /* B                                                                 */
    int32_t n_variants;     /* V >= 1                                                            */
    int32_t m_tile;         /* rows per workgroup: 0 = auto, else 32 / 64 / 128                  */
    int32_t reserved;       /* kernel selection: 0 = auto (small-batch kernel up to 8 sequences when ws_sync != NULL, the whole-step
                               kernel with split tiles at 9..128 sequences when ws_xch != NULL, else the whole-step kernel); 4 = whole-step kernel always;
                               3 = small-batch kernel always; 1 = five kernels per block (the plain restatement the whole-step kernel is checked against bit for bit);
                               5 = wave-per-sequence kernel always (needs x_fragment_order = 1);
                               +8 = never split a tile over several workgroups (see ws_xch)               */
    /* conditioning, row (v*B + b)*32 + frame */
    const float*   cond;    /* [V*B*32][512] per-clip term: cbias + c_frame + seed/style term    */
    const int32_t* t_model; /* [V*B] ORIGINAL timestep -> row of syn_model.te                    */
    const float*   cfg_w;   /* [3][V] weights of the variants for output channels 0:512, 512:1024,
                               1024:1536; NULL iff V == 1.  With cfg_w_clip_stride = 3 V (below):
                               [B][3][V], one table per clip - the reference's per-sample guidance
                               scales, y['scale'].view(-1, 1, 1, 1) (diffusion/cfg_sampler.py:28,54,167) */
    /* state, token-major */
    const float*   x_t;       /* [B*32][1536] fp32                                               */
    const void*    x_t_bf16;  /* [B*32][1536] bf16 copy of x_t (GEMM operand)                    */
    const float*   noise;     /* [B*32][1536] fp32 N(0,1) to inject, or NULL                     */
    const uint64_t* rng;      /* used when noise == NULL: device {seed, first_clip} -> the output GEMM's epilogue
                                 draws N(0,1) itself, identical to syn_randn(seed, stream_id = t_coef[clip],
                                 first_index = first_clip*32*1536); NULL = no noise term            */
    const float*   coef;      /* [n][4] rows (c_x0, c_xt, sigma, unused)                         */
    const int32_t* t_coef;    /* [B] row of coef used by each clip                               */
    float*         x_next;      /* [B*32][1536] = c_x0*x0_hat + c_xt*x_t + sigma*noise; may alias x_t */
    void*          x_next_bf16; /* bf16 copy of x_next; may alias x_t_bf16                       */
    float*         pred_x0;     /* [B*32][1536] x0_hat, or NULL                                  */
    /* workspace (caller-owned, contents undefined on return); R = V*B*32 rows */
    float* ws_h;      /* [R][512]  fp32 residual stream                                          */
    void*  ws_xn;     /* [R][512]  bf16 normalised activations                                   */
    void*  ws_q;      /* [R][512]  bf16                                                          */
    void*  ws_k;      /* [R][512]  bf16                                                          */
    void*  ws_vt;     /* [V*B*4*128][32] bf16, V transposed per (clip, head)                     */
    void*  ws_o;      /* [R][512]  bf16 attention output                                         */
    void*  ws_hid;    /* [R][1024] bf16 MLP hidden                                               */
    void*  ws_hc;     /* [3][B*32][512] bf16 guidance-combined stream (only if V > 1)            */
    uint32_t* ws_sync; /* [320] u32, zeroed ONCE by the caller; small-batch path only (group-barrier
                          counters; word 256 = sticky error flag: a barrier wait ran out)        */
    float* ws_x0v;     /* [V*B*32][1536] fp32 or NULL; small-batch path with V > 1: lets the variants of a clip run on
                          different XCDs (each writes its x0_hat here, a small second kernel combines them)   */
    float* ws_xch;     /* [V*B][8][32*512] fp32 or NULL; with it (and ws_sync) batches of 9..128 sequences run the whole-step
                          kernel with every 32-row tile split over 2 or 4 workgroups of one XCD (heads / MLP slices / output
                          chunks dealt to the members, partial residual streams exchanged through these slots)      */
    int32_t x_fragment_order; /* 0: x_t / x_t_bf16 / noise / x_next / pred_x0 are token-major (above); 1: they are in the
                          wave-per-sequence kernel's fragment order (syn_x_to_fragment), and that kernel runs the step
                          (n_variants <= 4, syn_model.tape non-NULL; of the workspaces only cfg_w is used).  syn_prefers_fragment_order() says when the
                          library would like a caller to keep its latent that way.                                   */
    int32_t cfg_w_clip_stride; /* (ABI 8) floats between the cfg_w tables of consecutive clips: 0 = one table for the batch, 3 * n_variants = one per clip */
} syn_step;

/* 1 when a step over n_clips x n_variants is best run by the wave-per-sequence kernel, i.e. the caller should keep the
 * latent in fragment order for the whole loop (batches whose passes of 4 sequences per CU beat the token-resident kernel's passes
 * of 2: 513..1024, 1537..2048, ... sequences; the <= 4
 * variants of a guided clip are the waves of one workgroup and meet in the output stage through LDS); 0 otherwise. */
int32_t syn_prefers_fragment_order(int32_t n_clips, int32_t n_variants);

/* Enqueue one full step on `stream`: one kernel (k_stack, or k_lat for small batches; + k_guided_update when the
 * variants of a small guided batch were dealt to different XCDs), or the 42-kernel A/B path (reserved = 1). */
int syn_denoise_step(const syn_model* model, const syn_step* step, void* stream);

/* Loop helper: fills t_coef[0..n_t_coef) with sched[2i] and t_model[0..n_t_model) with sched[2i + 1], i = *counter, then
 * advances *counter - a whole p_sample_loop iteration (gaussian_diffusion.py:714-739) becomes one graph replay
 * (this launch + syn_denoise_step) with no host work in between. */
int syn_step_advance(const int32_t* sched, int32_t* counter, int32_t* t_model, int32_t n_t_model, int32_t* t_coef, int32_t n_t_coef,
                     void* stream);
/* The same for the next n_steps steps at once: row s of t_model [n_steps][n_t_model] / t_coef [n_steps][n_t_coef] belongs
 * to step *counter + s; *counter advances by n_steps.  A graph of n_steps captured steps, each pointing at its own row,
 * needs one of these per replay instead of one per step. */
int syn_steps_advance(const int32_t* sched, int32_t* counter, int32_t* t_model, int32_t n_t_model, int32_t* t_coef,
                      int32_t n_t_coef, int32_t n_steps, void* stream);

/* n_steps consecutive steps of a hook-free stretch of p_sample_loop / ddim_sample_loop (gaussian_diffusion.py:714-739,
 * 905-931), in place (x_next = x_t): step j takes its timesteps from t_model + j * t_model_stride and
 * t_coef + j * t_coef_stride (the rows syn_steps_advance fills), its noise from rng (noise must be NULL when n_steps > 1).
 * Fragment-order latents: ONE persistent launch per CU-filling slice of the batch - every workgroup carries its four sequences through all the steps, so the
 * workgroups drift out of phase and the HBM traffic of the input / output stages of some overlaps the matrix work of the
 * others.  Token-major latents: the launches of syn_denoise_step, n_steps times. */
int syn_denoise_steps(const syn_model* model, const syn_step* step, int32_t n_steps, int32_t t_model_stride,
                      int32_t t_coef_stride, void* stream);

/* Same step, eagerly, with a hipEvent after every launch: fills ms_out[8] / count_out[8] with the elapsed
 * milliseconds and launch count per stage class {0 input GEMM, 1 qkv GEMM, 2 attention, 3 proj GEMM,
 * 4 fc1 GEMM, 5 fc2 GEMM, 6 guidance combine, 7 output GEMM}.  Synchronises `stream`; not graph-capturable. */
int syn_denoise_step_profile(const syn_model* model, const syn_step* step, void* stream, float* ms_out, int32_t* count_out);

/* Training-mode forward of one Conv1d(k = 15) of the WavEncoder (models/utils/layer.py:144-184 conv1 / conv2 / downsample[0],
 * called from models/denoiser.py:304-322 with BatchNorm on batch statistics, so nothing is folded): x fp32 channels-last
 * [n_clips][l_in][cin] (what PyTorch calls an (N, C, 1, L) channels_last tensor), y fp32 [n_clips][l_out][cout],
 * l_out = (l_in + 2 pad - 15) / stride + 1.  w_hi / w_lo: syn_pack_weight of the hi / lo bf16 halves of the GEMM matrix
 * W'[cout][tap][cin] (taps zero-padded to a multiple of the stride); products are hi.hi + lo.hi + hi.lo on the bf16 matrix
 * pipe, i.e. fp32-grade.  bias may be NULL.  Supported (cin, stride, cout): the encoder's own, see the error text. */
/* Weight gradient of a stride-1, padding-7 Conv1d(k = 15) of the encoder (cout 64 / 128 / 256; cin a multiple of 16; the
 * encoder's unpadded strided layers too - (cin, stride, cout) = (64, 6, 64), (64, 6, 128), (128, 3, 256), read as stride-1 convolutions
 * over rows of stride x cin = 384 channels): dw [cout][cin][15] = sum over clips and output
 * positions of dy[l][co] x[l stride + t - pad][ci], x fp32 channels-last [n_clips][l_in][cin], dy [n_clips][l_out][cout].
 * Split operands like the forward (fp32-grade); workgroup partial sums go through ws -
 * syn_conv1d_wgrad_shares(n_clips, l_out, stride * cin, cout) * cout * ceil(15 / stride) * stride * cin floats - and are added in a
 * fixed order (no atomics).  (ABI 9: cout - a workgroup owns a slice of the gradient (output-channel tile x input-channel block) and a share of the
 * positions; the more slices a layer is cut in, the fewer shares fill the chip, and every share is a full-size partial gradient through HBM.) */
int32_t syn_conv1d_wgrad_shares(int32_t n_clips, int32_t l_out, int32_t cin_rows, int32_t cout);
int syn_conv1d_train_wgrad(const float* x, const float* dy, int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad,
                           int32_t cout, float* ws, float* dw, void* stream);

/* The encoder's first layer in training mode - Conv1d(cin = 1 | 2 -> 64, k 15, stride, padding) of block 0's conv1 and of its
 * shortcut (models/denoiser.py:308, models/utils/layer.py:150,158; stride 5, padding 1700 there) - without the bias (see
 * syn_bn_act_fwd): exact fp32 - the forward on fp32 FMAs, the weight gradient on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, a persistent stream over the batch).  x fp32 [n_clips][l_in][cin] (the waveform as the reference passes it), w the module's weight
 * [64][cin][15], y / dy fp32 channels-last [n_clips][l_out][64], l_out = (l_in + 2 pad - 15) / stride + 1.  The weight gradient
 * needs ws of syn_conv1d_first_parts(n_clips, l_out) * 64 * cin * 15 floats (workgroup partial sums, added in a fixed order) and
 * writes dw [64][cin][15].  There is no data gradient: the waveform is an input. */
int32_t syn_conv1d_first_parts(int32_t n_clips, int32_t l_out);
int syn_conv1d_first_fwd(const float* x, int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad, const float* w, float* y,
                         void* stream);
int syn_conv1d_first_wgrad(const float* x, const float* dy, int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad, float* ws,
                           float* dw, void* stream);
/* (ABI 7) the forward with the BatchNorm statistics of its output from the same launch: bn_part (NULL, or syn_conv1d_first_tiles(n_clips, l_out) x 2 x 64
 * floats) = per-workgroup sum and sum of squares of every output channel, what syn_bn_finalize takes (chunks = that tile count). */
int32_t syn_conv1d_first_tiles(int32_t n_clips, int32_t l_out);
/* (ABI 7) block 0's conv1 and its shortcut convolution (same input, kernel, stride, padding: models/utils/layer.py:150,158) as ONE launch over the waveform
 * window: y_a / y_b and their statistics (both NULL or both given, syn_conv1d_first_tiles(...) x 2 x 64 floats each). */
int syn_conv1d_first_fwd2(const float* x, int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad, const float* w_a, const float* w_b,
                          float* y_a, float* y_b, float* bn_part_a, float* bn_part_b, void* stream);
/* (ABI 7) the first layer's weight gradient with the backward of the BatchNorm + LeakyReLU behind it folded in: dz = the gradient at the activation's output,
 * y = the convolution's raw output, stats / affine from syn_bn_finalize, dgamma_dbeta from syn_bn_bwd_stats (below); dy = scale (dp - dbeta / M - xhat dgamma / M),
 * dp = dz act'(a(y)), is formed as the values are loaded and never written.  stride 5 (the encoder's). */
int syn_conv1d_first_wgrad_bn(const float* x, const float* dz, const float* y, const float* stats, const float* affine, const float* dgamma_dbeta,
                              int32_t act, int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad, float* ws, float* dw, void* stream);
/* (ABI 9) Block 0's conv1: its weight gradient AND bn1's backward sums from ONE pass over (dz, y) - the BatchNorm backward is linear in dgamma / dbeta and only a
 * reduction consumes dy here (the waveform takes no gradient): dW = scale (S1 - dbeta / M S2 - dgamma / M S3) with S1 = sum dp win, S2 = sum win, S3 = sum xhat win
 * accumulated side by side.  Replaces syn_bn_bwd_stats + syn_conv1d_first_wgrad_bn (a second read of both tensors).  dw [64][cin][15] and dgamma_dbeta [3][64] are
 * written by this call (no partial sums left for syn_conv1d_wgrad_sums); ws: syn_conv1d_first_parts(n_clips, l_out) * (2 * 64 * cin * 15 + 160) floats.  stride 5. */
int syn_conv1d_first_wgrad_bn_lin(const float* x, const float* dz, const float* y, const float* stats, const float* affine, int32_t act,
                                  int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad, float* ws, float* dw, float* dgamma_dbeta, void* stream);
/* (ABI 9) Block 0's SHORTCUT convolution: its weight-gradient partial sums (ws as syn_conv1d_first_wgrad) with the block tail's backward folded into the loads -
 * dout = the gradient at the block's output, y2 / y_short the raw outputs of conv2 and of the shortcut convolution with the statistics, affines and
 * [dgamma | dbeta | 0] of their BatchNorms (syn_bn_block_bwd with dy = NULL leaves exactly those).  d = dout act'(a2(y2) + a_s(y_short)); the shortcut's
 * dy_s = its BatchNorm's backward of d feeds the matrix pipe and is never written; dy2 = bn2's backward of d IS written ([n_clips][l_out][64]: conv2's data
 * and weight gradients read it).  Against syn_bn_block_bwd's apply pass + syn_conv1d_first_wgrad: one write and one read of a 117 MB tensor less.  stride 5. */
int syn_conv1d_first_wgrad_tail(const float* x, const float* dout, const float* y2, const float* y_short, const float* stats2, const float* affine2,
                                const float* short_stats, const float* short_affine, const float* dgb2, const float* short_dgb, int32_t act,
                                int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad, float* ws, float* dy2, void* stream);
/* (ABI 7) the statistics half of syn_bn_act_bwd: dgamma_dbeta [3][channels] only (ws as there). */
int syn_bn_bwd_stats(const float* dz, const float* z, const float* y, const float* stats, const float* gamma, const float* beta, int64_t rows,
                     int32_t channels, int32_t act, float* ws, float* dgamma_dbeta, void* stream);
int syn_conv1d_first_fwd_stats(const float* x, int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad, const float* w, float* y,
                               float* bn_part, void* stream);

/* The two fragment sets syn_conv1d_train_fwd takes, from the module's weight w [cout][cin][15] (fp32) in one launch; each
 * output holds syn_conv1d_pack_bytes(...) bytes.  transposed = 1: the matrix of the DATA GRADIENT (stride 1; for a strided
 * convolution the form syn_conv1d_train_dgrad_sum takes: a stride-1 convolution over dy whose output rows are `stride` consecutive positions x cin channels)
 * of that convolution (taps reversed, channel roles swapped: N = cin, C = cout) - the gradient then is
 * syn_conv1d_train_fwd(dy, ..., cin = cout, stride 1, pad 7, ..., cout = cin). */
int syn_conv1d_pack_split(const float* w, int32_t cout, int32_t cin, int32_t stride, int32_t transposed, void* out_hi, void* out_lo,
                          void* stream);
/* (ABI 5) Up to SYN_CONV_PACK_MAX syn_conv1d_pack_split calls as ONE launch (requests: a host array, read at call time): the training step packs every
 * fragment set its convolutions take - forward and data-gradient forms - once per step, the weights only change in optimizer.step(). */
#define SYN_CONV_PACK_MAX 40
typedef struct syn_conv_pack_req { const float* w; void* out_hi; void* out_lo; int32_t cout, cin, stride, transposed; } syn_conv_pack_req;
int syn_conv1d_pack_split_many(const syn_conv_pack_req* reqs, int32_t n_reqs, void* stream);
/* Bytes of each of syn_conv1d_pack_split's two outputs. */
int64_t syn_conv1d_pack_bytes(int32_t cout, int32_t cin, int32_t stride, int32_t transposed);
/* (ABI 8) syn_conv1d_train_wgrad / _wgrad_norm / syn_conv1d_first_wgrad / _wgrad_bn with dw = NULL leave their per-share partial sums in ws; this
 * adds up to SYN_WGRAD_SUM_MAX such gradients up in ONE launch (the three convolutions of a BasicBlock), each in the order its own launch would have used.
 * A job: part = that call's ws, dw = the module-layout gradient [cout][cin][15], the call's n_clips / cin / stride / cout and ITS l_out;
 * first_layer != 0: the job is a syn_conv1d_first_wgrad* call (cin 1 | 2, cout 64). */
#define SYN_WGRAD_SUM_MAX 4
typedef struct syn_wgrad_sum_job { const float* part; float* dw; int32_t n_clips, l_out, cin, stride, cout, first_layer;
                                   int32_t share_pitch;   /* (ABI 9) floats between consecutive shares of `part`; 0 = the gradient's own size (a share that holds two
                                                           * gradients - syn_conv1d_train_wgrad_pair - has twice that, and the second job's `part` starts one gradient in) */
                                   int32_t reserved; } syn_wgrad_sum_job;
int syn_conv1d_wgrad_sums(const syn_wgrad_sum_job* jobs, int32_t n_jobs, void* stream);
/* (ABI 7) The data gradient that reaches a BasicBlock's input, written once: dx [n_clips][l_in][cin] = conv^T dy [+ conv2^T dy2] [+ residual].
 * Strided, unpadded layers (a down-sampling block: conv1 and the shortcut convolution share input and geometry): dy / dy2 [n_clips][l_out][cout] with their
 * transposed fragment sets, both accumulated in one launch instead of two tensors and an add; residual must be NULL.  Stride 1, padding 7 (a block with an
 * identity shortcut): dy2 NULL, residual (laid out like dx; the gradient arriving along the shortcut) added in the epilogue. */
int syn_conv1d_train_dgrad_sum(const float* dy, const void* w_hi, const void* w_lo, const float* dy2, const void* w2_hi, const void* w2_lo,
                               const float* residual, int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad, int32_t cout, float* dx,
                               void* stream);
int syn_conv1d_train_fwd(const float* x, int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad,
                         const void* w_hi, const void* w_lo, const float* bias, int32_t cout, float* y, float* bn_part, void* stream);
/* bn_part (NULL, or syn_conv1d_train_fwd_tiles(...) x 2 x cout floats; bias must then be NULL): per-workgroup sum and sum of squares
 * of every output channel, straight from the accumulators - the BatchNorm that follows takes them as syn_bn_act_fwd's ws with
 * ws_chunks = that tile count and skips its own pass over y. */
int32_t syn_conv1d_train_fwd_tiles(int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad, int32_t cout);
/* (ABI 7) the stride-1 layers with the BatchNorm (+ LeakyReLU) of the convolution in front applied to the input as it is staged - x is that
 * convolution's raw output, the kernel reads act(x * in_affine[0][c] + in_affine[1][c]) (in_affine [2][cin] from syn_bn_finalize; in_act != 0:
 * LeakyReLU(0.01)), zero outside the clip; no bias.  The weight gradient with the same view of x. */
/* (ABI 8) conv1 and the shortcut convolution of a down-sampling BasicBlock (models/utils/layer.py:150-158: the same input, stride, padding and
 * width) as ONE launch: the input tile is staged (and split into its bf16 halves) once for both weight sets.  The encoder's strided layers only
 * (64x6->64, 64x6->128, 128x3->256); y_a / y_b and the statistics as syn_conv1d_train_fwd's. */
int syn_conv1d_train_fwd_pair(const float* x, int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad, int32_t cout,
                              const void* wa_hi, const void* wa_lo, float* y_a, float* bn_part_a,
                              const void* wb_hi, const void* wb_lo, float* y_b, float* bn_part_b, void* stream);
int syn_conv1d_train_fwd_norm(const float* x, int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad,
                              const void* w_hi, const void* w_lo, int32_t cout, const float* in_affine, int32_t in_act, float* y, float* bn_part,
                              void* stream);
int syn_conv1d_train_wgrad_norm(const float* x, const float* dy, int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad,
                                int32_t cout, const float* in_affine, int32_t in_act, float* ws, float* dw, void* stream);
/* (ABI 9) conv1 and the shortcut convolution of a down-sampling block - same input, stride, padding and width - leave BOTH weight gradients' partial sums from one
 * launch that stages the input once: ws [shares][2 cout][taps][stride cin] with shares = syn_conv1d_wgrad_shares(n_clips, l_out, stride cin, cout); rows 0 .. cout - 1 of
 * a share are dy_a's gradient, the rest dy_b's (two syn_wgrad_sum_job with share_pitch = 2 cout taps stride cin, the second's part one gradient in).
 * (cin, stride, pad, cout) = (64, 6, 0, 64): block 1, whose input is the encoder's largest tensor. */
int syn_conv1d_train_wgrad_pair(const float* x, const float* dy_a, const float* dy_b, int32_t n_clips, int32_t l_in, int32_t cin, int32_t stride, int32_t pad,
                                int32_t cout, float* ws, void* stream);

/* Every bf16 fragment set the training step's Linear layers need, from the fp32 master weights in ONE launch (the weights change
 * once per step, in optimizer.step()): job j packs src (n x k row-major, transposed = 0: as syn_pack_weight; k x n row-major,
 * transposed = 1: as syn_pack_weight_t) into out.  jobs_dev: device array; max_fragments = max over the jobs of n / 16 * k / 32. */
/* (ABI 8) src_dim > 0: the source's real extent along a zero-PADDED dimension - k of a row-major [n][src_dim] source (transposed = 0), n of a
 * row-major [k][src_dim] source (transposed = 1): text_encoder_body's 300 input features as 384 columns. */
typedef struct syn_pack_job { const float* src; void* out; int32_t n, k, transposed, src_dim; } syn_pack_job;
int syn_pack_weights(const syn_pack_job* jobs_dev, int32_t n_jobs, int64_t max_fragments, void* stream);

/* ---- training path: fp32 forward / backward of the non-GEMM pieces of a transformer block ----
 * LayerNorm(512, eps 1e-5) of `rows` rows (models/timm_transformer/transformer.py:160,162,183,193); the backward
 * needs scratch of ceil(rows/16)*1024 floats and writes dgamma[512], dbeta[512] (deterministic two-stage sums); `add` (NULL or
 * [rows][512]) is added to dx: the gradient that reaches x along the block's residual connection (transformer.py:195-198). */
/* (ABI 5) y fp32 and / or y_bf16 [rows][512] (either may be NULL): the Linear that follows takes bf16 operands, and nothing else
 * reads a pre-LN block's normalised rows - the training step asks for the bf16 copy only. */
int syn_ln_fwd(const float* x, const float* gamma, const float* beta, float* y, void* y_bf16, float* mean, float* rstd, int32_t rows,
               void* stream);
int syn_ln_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* add, float* dx,
               float* dgamma, float* dbeta, float* scratch, int32_t rows, void* stream);
/* nn.GELU() (exact erf form, transformer.py:117-151), n % 4 == 0 elements. */
/* (ABI 5) y fp32 and / or y_bf16 (either may be NULL), as syn_ln_fwd. */
int syn_gelu_fwd(const float* x, float* y, void* y_bf16, int64_t n, void* stream);
int syn_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* (ABI 6) Rotary position embedding on the hidden state (models/denoiser.py:178-186, 324-343: SinusoidalEmbeddings + apply_rotary_pos_emb on the
 * (B x 8, 32, 64) view): x, y fp32 [n_seq][32][512] (may alias), a token's features = 8 groups of 64, (u, v) = (first, last 32 of a group) ->
 * (u cos - v sin, v cos + u sin); cos_t / sin_t fp32 [32 positions][32] = cos / sin(position x inv_freq[j]).  inverse != 0: the transposed
 * rotation, i.e. the gradient with respect to x. */
int syn_rotary(const float* x, const float* cos_t, const float* sin_t, int32_t n_seq, int32_t inverse, float* y, void* stream);
/* (ABI 6) Weight / bias gradient of an nn.Linear whose input has few rows (the timestep MLP and embed_text see one row per clip, models/denoiser.py:92,
 * 231-245): dw fp32 [n][k] = sum_m dy[m][n] x[m][k], db [n] = sum_m dy[m][n] (NULL to skip); dy fp32 [m_rows][n], x bf16 [m_rows][k] (the operand
 * the forward GEMM took), m_rows <= 64, n % 16 == 0.  fp32 FMAs in row order. */
int syn_linear_wgrad_rows(const float* dy, const void* x_bf16, int32_t m_rows, int32_t n, int32_t k, float* dw, float* db, void* stream);
/* (ABI 8) masked_l2 of training_losses (gaussian_diffusion.py:202-215, 1307-1314: SmoothL1(beta 1) x mask, summed, / (sum(mask) x C)).
 * target fp32 (B, C, 1, T) as the reference lays it out; out either the same (out_rows = 0) or the output Linear's own rows [B][T][C] (out_rows = 1:
 * what models/denoiser.py:294-300's reshape / permute is a view of); mask bytes [batch][t_len]; part: [batch][channels / 64] floats of scratch;
 * loss [batch].  channels % 64 == 0, t_len <= 64.  Two launches (tile sums, then their sum in tile order).
 * poison_flag (NULL or one device int): while it is nonzero every loss is NaN - pass syn_train_stack.sync + 256, the persistent block kernels' sticky
 * barrier-timeout flag (see syn_train_stack_fwd), so that a step computed from partial sums that never arrived cannot pass for a step. */
int syn_masked_smooth_l1(const float* target, const float* out, const uint8_t* mask, int32_t batch, int32_t channels, int32_t t_len, int32_t out_rows,
                         float* part, const int32_t* poison_flag, float* loss, void* stream);
/* Its gradient: dout (laid out like out) = sample_scale[b] (NULL: 1) x d loss[b] / d out - one pass in the backward, the incoming gradient of
 * loss[b] folded in. */
int syn_masked_smooth_l1_grad(const float* target, const float* out, const uint8_t* mask, int32_t batch, int32_t channels, int32_t t_len, int32_t out_rows,
                              const float* sample_scale, float* dout, void* stream);
/* (ABI 8) bf16 GEMM operand [m_rows][out_ld] = up to SYN_CONCAT_MAX fp32 sources side by side, columns beyond them zero (torch.cat along the feature
 * axis + the operand rounding + the zero padding to the GEMM's 128-column granularity, models/denoiser.py:155-174, denoiser_h3d.py:187-200).  Source i:
 * `width` columns of rows of pitch ld; operand row r reads source row r / row_div (a per-clip vector repeated over the clip's frames), p2 (NULL or
 * like p) added first; or pool > 1: the mean of source rows r * pool .. + pool - 1 (F.avg_pool1d over the frame axis, denoiser.py:157).
 * width % 4 == 0, ld % 4 == 0. */
#define SYN_CONCAT_MAX 4
typedef struct syn_concat_src { const float* p; const float* p2; int32_t width, ld, row_div, pool; } syn_concat_src;
int syn_rows_concat_bf16(const syn_concat_src* srcs, int32_t n_src, int32_t m_rows, int32_t out_ld, void* out_bf16, void* stream);
/* (ABI 8) nn.Embedding lookup (models/denoiser.py:72,152) written as the next Linear's operand: out bf16 [m_rows][out_ld], columns dim .. out_ld zero. */
int syn_embed_rows_bf16(const int64_t* ids, const float* table, int32_t vocab, int32_t dim, int32_t m_rows, int32_t out_ld, void* out_bf16, void* stream);
/* (ABI 8) (B, C, 1, T) fp32 -> bf16 rows [B * T][C]: the permute of models/denoiser.py:160 + the operand rounding; channels % 64 == 0, t_len == 32. */
int syn_bct_to_rows_bf16(const float* x_bct, int32_t n_clips, int32_t channels, int32_t t_len, void* out_bf16, void* stream);
/* (ABI 8) out [n_groups][width] = sums over the `group` consecutive rows (pitch ld) of each group, in row order: the gradient of a vector the forward
 * repeated over a clip's frames. */
int syn_rows_group_sum(const float* src, int32_t ld, int32_t width, int32_t group, int32_t n_groups, float* out, void* stream);
/* (ABI 8) out fp32 [m_rows][width] = scale x src[r / row_div][0:width] (src rows of pitch ld): the backward of an average pool over row_div rows. */
int syn_rows_expand(const float* src, int32_t ld, int32_t width, int32_t row_div, float scale, int32_t m_rows, float* out, void* stream);
/* (ABI 8) out [n] = sum over i < rows of part [i][n], in order: a Linear's bias gradient from syn_linear_bwd_prep's colsum_part when no syn_linear_pair
 * launch carries it (a Linear whose input takes no gradient). */
int syn_colsum_parts(const float* part, int32_t rows, int32_t n, float* out, void* stream);
/* (ABI 8) One read pass over a buffer (a dword per 64-byte line): pulls it back into the memory-side cache ahead of a latency-bound consumer. */
int syn_touch(const void* p, int64_t bytes, void* stream);
/* nn.BatchNorm1d in training mode (batch statistics, running statistics updated as PyTorch does) [+ shortcut] [+ LeakyReLU(0.01)]
 * of the audio encoder's BasicBlock (models/utils/layer.py:171-184) on channels-last fp32 [rows][channels], rows = clips x
 * positions: z = act(gamma (y - mean) rstd + beta [+ shortcut]).  ws: 2 * syn_bn_chunks(rows) * channels floats; stats
 * [2][channels] receives mean / rstd for the backward; run_mean / run_var may be NULL; conv_bias (or NULL): the bias of the
 * convolution that produced y, NOT added to y by the caller - under batch statistics it only shifts the running mean.  Backward: dgamma_dbeta [3][channels]
 * (ABI 5: the third row is written with zeros - the gradient of conv_bias, which the mean subtraction cancels), dy, and dshortcut (NULL when there is none) from dz, the saved z and y.  z may be NULL when there was no shortcut (beta then
 * required): the activation's sign is recomputed from y, which the passes read anyway. */
int32_t syn_bn_chunks(int64_t rows);
int syn_bn_act_fwd(const float* y, const float* shortcut, int64_t rows, int32_t channels, const float* gamma, const float* beta, float eps,
                   float momentum, float* run_mean, float* run_var, const float* conv_bias, int32_t act, float* ws, int32_t ws_chunks, float* stats,
                   float* z, void* stream);
int syn_bn_act_bwd(const float* dz, const float* z, const float* y, const float* stats, const float* gamma, const float* beta, int64_t rows,
                   int32_t channels, int32_t act, float* ws, float* dgamma_dbeta, float* dy, float* dshortcut, void* stream);
/* The same with the statistics reduced over several ranks - nn.SyncBatchNorm, which the reference's DDP branch converts every BatchNorm to
 * (train.py:90; torch/nn/modules/_functions.py SyncBatchNorm).  The library reduces to per-channel sums in fp64 ([2][channels]: forward
 * sum y, sum y^2; backward sum d, sum d * xhat), the CALLER all-reduces them over its process group (RCCL) together with the row count, and the
 * second call finalises with the global numbers: normalisation and the data gradient use the global sums, dgamma / dbeta stay this rank's own
 * (DDP averages parameter gradients), the running statistics move towards the GLOBAL batch statistics.  scratch: 2 * channels floats. */
int syn_bn_sums(const float* y, int64_t rows, int32_t channels, float* ws, int32_t ws_chunks, double* sums, void* stream);
int syn_bn_act_apply(const float* y, const float* shortcut, int64_t rows, int64_t rows_total, int32_t channels, const float* gamma, const float* beta,
                     float eps, float momentum, float* run_mean, float* run_var, const float* conv_bias, int32_t act, const double* sums_total,
                     float* stats, float* z, void* stream);
int syn_bn_bwd_sums(const float* dz, const float* z, const float* y, const float* stats, const float* gamma, const float* beta, int64_t rows,
                    int32_t channels, int32_t act, float* ws, double* sums, void* stream);
int syn_bn_act_bwd_apply(const float* dz, const float* z, const float* y, const float* stats, const float* gamma, const float* beta,
                         const double* sums_local, const double* sums_total, int64_t rows, int64_t rows_total, int32_t channels, int32_t act,
                         float* dgamma_dbeta, float* scratch, float* dy, float* dshortcut, void* stream);

/* (ABI 7) The same block tail with fewer passes over its tensors (models/utils/layer.py:171-184: conv1 -> bn1 -> act -> conv2 -> bn2 (+ shortcut | + bn(conv(x)))
 * -> act).  A batch-statistics BatchNorm is a per-channel affine map a(v) = v * scale + shift once its statistics exist:
 *   syn_bn_finalize   part [chunks][2][channels] (sum, sum of squares: a convolution's bn_part) -> stats [2][channels] = mean, rstd; affine [2][channels] =
 *                     scale = rstd gamma, shift = beta - mean rstd gamma; running statistics as nn.BatchNorm1d updates them (conv_bias as in syn_bn_act_fwd)
 *   syn_bn_apply2     z = act(a(y) + s), s = a_s(shortcut) when short_affine is given (the down-sampling branch's BatchNorm, applied here instead of in a
 *                     pass of its own), the raw shortcut otherwise, nothing when shortcut is NULL
 *   syn_bn_block_bwd  from dz = d loss / d z: dp = dz act'(p) with the pre-activation p RECOMPUTED from y and the shortcut (the block's output is neither
 *                     read nor needed by the backward); dgb [3][channels] = dgamma, dbeta, 0 of a; dy; with a normalised shortcut also short_dgb and
 *                     dshortcut = the data gradient of ITS BatchNorm, with a raw one dshortcut = dp.  One statistics pass and one apply pass over
 *                     (dz, y, shortcut) for both BatchNorms.  ws: 3 * syn_bn_chunks(rows) * channels floats.  (ABI 9) dy = NULL (and dshortcut NULL): the
 *                     statistics pass and [dgamma | dbeta | 0] only - the caller's next kernel forms the gradients itself (syn_conv1d_first_wgrad_tail).
 * bn1 + LeakyReLU in front of conv2 is applied by conv2 itself while it stages its input (syn_conv1d_train_fwd_norm; the weight gradient recomputes it the
 * same way, syn_conv1d_train_wgrad_norm): in_affine [2][cin] = bn1's affine, in_act != 0: LeakyReLU(0.01); positions outside the clip stay zero. */
int syn_bn_finalize(const float* part, int32_t chunks, int64_t rows, int32_t channels, const float* gamma, const float* beta, float eps, float momentum,
                    float* run_mean, float* run_var, const float* conv_bias, float* stats, float* affine, void* stream);
/* (ABI 8) Two syn_bn_finalize calls as one launch: bn1 and the shortcut's BatchNorm of a down-sampling block, whose partial sums exist together. */
typedef struct syn_bn_finalize_job { const float* part; int32_t chunks, channels; int64_t rows; const float* gamma; const float* beta; float eps, momentum;
                                     float* run_mean; float* run_var; const float* conv_bias; float* stats; float* affine; } syn_bn_finalize_job;
int syn_bn_finalize_pair(const syn_bn_finalize_job* a, const syn_bn_finalize_job* b, void* stream);
int syn_bn_apply2(const float* y, const float* affine, const float* shortcut, const float* short_affine, int64_t rows, int32_t channels, int32_t act,
                  float* z, void* stream);
int syn_bn_block_bwd(const float* dz, const float* y, const float* shortcut, const float* stats, const float* affine, const float* short_stats,
                     const float* short_affine, int64_t rows, int32_t channels, int32_t act, float* ws, float* dgb, float* short_dgb, float* dy,
                     float* dshortcut, void* stream);

/* (ABI 7) The eight transformer blocks of the TRAINING forward as one persistent launch (models/timm_transformer/transformer.py:154-198 in train() mode:
 * h + drop_path(attn(norm1(h))), then h + drop_path(mlp(norm2(h))), x 8), 1 .. 64 sequences of 32 tokens: a sequence's 32 rows are shared by 4 workgroups of
 * one XCD (head j / MLP slice j each, partial sums exchanged through that XCD's L2: the whole-step sampling kernel's tile-split mode).  Weights: the packed
 * fragment sets of syn_layer (syn_pack_weight / syn_pack_weights).  drop_path: NULL, or the factors [16][n_seq] (row 2 l: attention branch of block l, row
 * 2 l + 1: its MLP branch; 0 or 1 / keep).  Written for the backward, per block, with M = 32 n_seq rows: the two branch inputs h (fp32 [M][512]) and their LayerNorm
 * mean / rstd [M]; qkv fp32 [M][1536]; the fc1 output + bias before the GELU, fp32 [M][1024]; and the transposed bf16 operands of the four weight-gradient GEMMs as
 * packed fragments (what syn_linear_and_pack's xt_packed holds): LN1(h)^T, attention output^T, LN2(h)^T (512 x M each), GELU output^T (1024 x M).
 * sync: 320 zeroed uint32 (left zeroed); xch: n_seq x 8 x 16384 floats of scratch.
 * sync[256] is a STICKY ERROR FLAG shared by syn_train_stack_fwd and syn_train_stack_bwd: the XCD-local barrier waits of the member workgroups are
 * bounded, and a wait that runs out (another kernel holding CUs of the XCD) sets it and lets the kernel finish on whatever partial sums it found - the
 * results of that launch, and of every later one that finds the flag set (its waits give up after 256 polls), are wrong.  Nothing in the library clears it:
 * the host reads it back now and then and zeroes `sync` (syntalker_amd.training.check_stack_sync), and syn_masked_smooth_l1's poison_flag turns the step's
 * loss into NaN while it is set. */
typedef struct syn_train_block_save {
    float* h_attn; float* mean_attn; float* rstd_attn; float* qkv; void* xt_ln1; void* xt_attn;
    float* h_mlp; float* mean_mlp; float* rstd_mlp; float* pre; void* xt_ln2; void* xt_gelu;
} syn_train_block_save;
typedef struct syn_train_stack {
    const float* h_in; float* h_out;                 /* fp32 [32 n_seq][512] */
    syn_layer layer[SYN_LAYERS];
    syn_train_block_save save[SYN_LAYERS];
    const float* drop_path;
    int32_t n_seq, reserved;
    uint32_t* sync; float* xch;
} syn_train_stack;
int syn_train_stack_fwd(const syn_train_stack* a, void* stream);
/* (ABI 7) The backward of those eight blocks.  syn_train_stack_bwd: the data-gradient chain as one persistent launch (same decomposition: member = head / MLP
 * slice, two exchanges per block): dh_in [M][512] from dh_out, reading what the forward wrote (`fwd`: the forward's own struct, unchanged) and the TRANSPOSED
 * fragment sets of the weights (layer_t[l].w_* = fragments of W^T as syn_pack_weight_t / syn_pack_weights(transposed = 1) make them; ln1_g / ln2_g the gains;
 * the other members unused).  It leaves, per block: the gradients at the outputs of fc2, fc1, proj, qkv TRANSPOSED in bf16 ([512 | 1024 | 512 | 1536][M]:
 * the left operands of the weight-gradient GEMMs) and `part` [n_seq][4096], per-sequence partial sums of [dLN2 gain 512 | dLN2 shift 512 | dfc2 bias 512 |
 * dfc1 bias 1024 | dLN1 gain 512 | dLN1 shift 512 | dproj bias 512].  stash: n_seq x 4 x 16384 floats of scratch.
 * syn_train_stack_wgrad: the 32 weight-gradient GEMMs dW [n][k] = dY^T . X, four per launch (dw_* fp32, the module's layout), and the sums of `part` over the
 * sequences, each of its seven segments into its own tensor (d_*). */
typedef struct syn_train_block_grad {
    void* dyt_fc2; void* dyt_fc1; void* dyt_proj; void* dyt_qkv; float* part;
    float* dw_fc2; float* dw_fc1; float* dw_proj; float* dw_qkv;
    float* d_ln2_g; float* d_ln2_b; float* d_fc2_b; float* d_fc1_b; float* d_ln1_g; float* d_ln1_b; float* d_proj_b;    /* [512] each, d_fc1_b [1024] */
} syn_train_block_grad;
typedef struct syn_train_stack_grad {
    const syn_train_stack* fwd;
    const float* dh_out; float* dh_in;
    syn_layer layer_t[SYN_LAYERS];
    syn_train_block_grad grad[SYN_LAYERS];
    float* stash;
    int32_t first_block, last_block;          /* the blocks a call covers, SYN_LAYERS > first_block >= last_block >= 0 (the chain walks them downwards): cut in pieces,
                                               * the weight-gradient GEMMs of a finished piece can run on another stream beside the next piece of the chain */
} syn_train_stack_grad;
int syn_train_stack_bwd(const syn_train_stack_grad* a, void* stream);
int syn_train_stack_wgrad(const syn_train_stack_grad* a, void* stream);

/* (ABI 5) The optimizer step of the reference's training loop (diffusion_rvqvae_trainer.py:351-356: clip_grad_norm_(grad_norm), then Adam)
 * over lists of <= SYN_OPT_MAX fp32 tensors, pointers as kernel arguments (a captured step holds them by value):
 *   syn_opt_sqnorm    partials[b] = sum of g^2 over workgroup b's 8192-element chunk, b < syn_opt_blocks(list)        (one read of the gradients)
 *   syn_opt_scalars   *step_dev += 1; scal4 = {clip = min(1, max_norm / (sqrt(sum partials) + 1e-6)) (1 if max_norm <= 0), lr / (1 - beta1^step),
 *                     1 / sqrt(1 - beta2^step), total norm}; lr read from lr_dev when that is not NULL (a scheduler's device-resident rate)
 *   syn_opt_adam      g' = clip * g (+ weight_decay * p); m += (1 - beta1)(g' - m); v = beta2 v + (1 - beta2) g'^2;
 *                     p -= scal4[1] * m / (sqrt(v) * scal4[2] + eps)       (torch.optim.Adam, amsgrad off; the gradients are not rewritten) */
#define SYN_OPT_MAX 64
typedef struct syn_opt_list { float* p[SYN_OPT_MAX]; const float* g[SYN_OPT_MAX]; float* m[SYN_OPT_MAX]; float* v[SYN_OPT_MAX];
                              int32_t numel[SYN_OPT_MAX]; int32_t n; } syn_opt_list;
int32_t syn_opt_blocks(const syn_opt_list* l);
int syn_opt_sqnorm(const syn_opt_list* l, float* partials, void* stream);
int syn_opt_scalars(const float* partials, int32_t n_partials, float max_norm, const float* lr_dev, float lr, float beta1, float beta2, float* step_dev,
                    float* scal4, void* stream);
int syn_opt_adam(const syn_opt_list* l, const float* scal4, float beta1, float beta2, float eps, float weight_decay, void* stream);
/* Gradient of an nn.Embedding table (models/denoiser.py:72, the word embedding in front of text_encoder_body): dw [vocab][dim] =
 * sum over the positions p with ids[p] == v of dy[p][:], added in increasing p (no atomics on the result; every row written by the one launch).
 * ids int64 [n_pos] (n_pos <= 8192 per call), dy fp32 rows of pitch ld >= dim (ABI 8: a column slice of the next Linear's data gradient), dim <= 512,
 * vocab <= 65536. */
int syn_embedding_wgrad(const int64_t* ids, const float* dy, int32_t ld, int32_t n_pos, int32_t vocab, int32_t dim, float* dw, void* stream);
/* Backward of an nn.Linear, the operand preparation in one pass over its output gradient [m_rows][n] fp32: the bf16 copy (data-gradient GEMM),
 * the bf16 transpose [n][m_rows] (weight-gradient GEMM) and colsum_part [m_rows / 64][n] = column sums of every 64-row block
 * (NULL to skip; the bias gradient is their sum over the first index: syn_linear_pair's bias_grad).
 * (ABI 8) The gradient is read where its producer left it: row r = row r / row_div of `dy`, rows of pitch ld floats (a column slice of a wider
 * data gradient = one piece of a torch.cat's backward; row_div > 1 = the backward of an average pool over row_div rows with const_scale = 1 / row_div),
 * times const_scale.
 * (ABI 5) row_scale (NULL or one float per rows_per_scale rows): times its row's factor - the backward of
 * syn_linear_res's `residual + row_scale * (x W^T + b)`, i.e. of x + DropPath(branch) (timm_transformer/transformer.py:21-38,195-198). */
int syn_linear_bwd_prep(const float* dy, int32_t ld, int32_t row_div, float const_scale, int32_t m_rows, int32_t n, const float* row_scale,
                        int32_t rows_per_scale, void* dy_bf16, void* dy_bf16_t, float* colsum_part, void* stream);
/* Attention core (transformer.py:83-104, 4 heads x 128, 32 tokens, no mask, no dropout) on the packed output of the
 * qkv Linear: qkv [n_seq][32][3][4][128] -> o [n_seq][32][512]; backward recomputes the probabilities. */
/* (ABI 5) o fp32 and / or o_bf16 (either may be NULL), as syn_ln_fwd. */
int syn_attn_fwd(const float* qkv, float* o, void* o_bf16, int32_t n_seq, void* stream);
int syn_attn_bwd(const float* qkv, const float* d_o, float* dqkv, int32_t n_seq, void* stream);

/* ---- per-clip conditioning: audio encoder -------------------------------------
 * WavEncoder.forward in eval mode (models/denoiser.py:304-322; BasicBlock models/utils/layer.py:144-184) with the
 * BatchNorms folded into the convolutions by the caller.  wav [n_clips][n_samples][cin] fp32 ->
 * out [n_clips][syn_wav_out_frames(n_samples)][256] fp32 (128 frames for the 68224 / 68266-sample clips).
 * conv[i]: fragment-packed (syn_pack_weight) bf16 weights W'[cout][tap][cin'] and fp32 bias of, in order:
 *   b0.conv2, b1.conv1|shortcut, b1.conv2, b2.conv1, b2.conv2, b3.conv1|shortcut, b3.conv2, b4.conv1, b4.conv2,
 *   b5.conv1|shortcut, b5.conv2  ("conv1|shortcut": output channels concatenated; strided convs as stride-1 convs
 *   over s-row groups: taps ceil(15/s), cin' = s*cin, taps >= 15 zero - syntalker_amd/conditioning.py builds them).
 * w_first: block 0's conv1 and shortcut, fp32 [2][15*cin][64] (tap-major) followed by the biases [2][64].
 * workspace: syn_wav_workspace_bytes() bytes, zeroed ONCE by the caller for a given (n_clips, n_samples). */
typedef struct syn_wav_conv { const void* w; const float* bias; } syn_wav_conv;
typedef struct syn_wavenc {
    int32_t cin;            /* waveform channels: 1 or 2 */
    int32_t reserved;
    const float* w_first;
    syn_wav_conv conv[11];
} syn_wavenc;
int32_t syn_wav_out_frames(int32_t n_samples);
int64_t syn_wav_workspace_bytes(int32_t n_clips, int32_t n_samples);
int syn_wav_encode(const syn_wavenc* enc, const float* wav, int32_t n_clips, int32_t n_samples, void* workspace, float* out,
                   void* stream);

/* ---- per-clip conditioning behind the audio encoder --------------------------------
 * models/denoiser.py:147-157,160-174: word embedding -> Linear 300->256, cat with the audio features -> mix_audio_text
 * (512->256) -> avg_pool1d(4) -> . W2c^T, plus embed_text(seed) . W2a^T [+ style . W3s^T] and every bias: the additive
 * tensor `cond` of syn_step.  All of it is affine, so the caller folds it once per weight set (syntalker_amd/conditioning.py
 * CondWeights, fp64) into
 *   gt [256][512] = (W2c Wm_a)^T        tw [vocab][512] = word_embedding (W2c Wm_w Wt)^T      (the word path is a lookup)
 *   st [seed_dim + style_dim][512] = [W2a W_embed_text | W3s]^T        c0 [512] = all constant terms
 * audio_feat [n_clips][128][256] fp32 (syn_wav_encode), word [n_clips][128] int64, seed [n_clips][seed_dim], style
 * [n_clips][style_dim] or NULL, d_scratch [SYN_COND_SCRATCH_ROWS][n_clips][512] (ABI 4: the seed / style GEMM is split over K,
 * its partial sums are added in a fixed order) -> cond [n_clips][32][512].  Two launches, fp32 arithmetic. */
#define SYN_COND_SCRATCH_ROWS 8
typedef struct syn_cond_weights {
    const float* gt; const float* tw; const float* st; const float* c0;
    int32_t vocab, seed_dim, style_dim, reserved;
} syn_cond_weights;
int syn_cond_encode(const syn_cond_weights* w, const float* audio_feat, const int64_t* word, const float* seed, const float* style,
                    int32_t n_clips, float* d_scratch, float* cond, void* stream);

/* ---- load-time helpers -------------------------------------------------------------------- */
/* fp32 row-major W[n][k] (nn.Linear.weight layout) -> packed bf16 fragments (n*k*2 bytes).
 * n % 16 == 0, k % 32 == 0. */
int syn_pack_weight(const float* w, int32_t n, int32_t k, void* out_packed, void* stream);
/* Same fragments for W = S^T, S row-major [k][n], fp32 (is_bf16 = 0) or bf16: the training path packs W^T (dgrad) and
 * x^T (wgrad) without materialising transposed copies. */
int syn_pack_weight_t(const void* s_kn, int32_t is_bf16, int32_t n, int32_t k, void* out_packed, void* stream);

/* ---- layout at loop entry / exit ----------------------------------------------------------- */
/* Fragment order of the wave-per-sequence kernel: fp32 [clip][channel/32][q][lane = 32 hi + frame][r] holds channel
 * 32 nf + 8 q + 4 hi + r (what a lane owns after the output GEMM); bf16 [clip][channel/32][c][lane][e] holds channel
 * 32 nf + 16 c + 8 (e >> 2) + 4 hi + (e & 3) (the B operand of the input GEMM).  (B,1536,1,32) fp32 -> both (nullable). */
int syn_x_to_fragment(const float* x_bct, int32_t n_clips, float* out_f32, void* out_bf16, void* stream);
/* fragment-order fp32 -> (B,1536,1,32) fp32. */
int syn_x_from_fragment(const float* x_frag, int32_t n_clips, float* out_bct, void* stream);
/* (B,1536,1,32) fp32 -> token-major fp32 (nullable) and bf16 (nullable). */
int syn_to_token_major(const float* x_bct, int32_t n_clips, float* out_f32, void* out_bf16, void* stream);
/* token-major fp32 -> (B,1536,1,32) fp32. */
int syn_from_token_major(const float* x_btc, int32_t n_clips, float* out_bct, void* stream);

/* q_sample / generic axpby on flat fp32 buffers: out = a[t_row[clip]]*x + b[t_row[clip]]*y, clip = i / per_clip.
 * (diffusion/gaussian_diffusion.py:235-253). */
int syn_axpby_rows(const float* x, const float* y, const float* coef_ab /*[n][2]*/, const int32_t* t_row,
                   int32_t n_clips, int32_t per_clip, float* out, void* stream);

/* Philox4x32-10 + Box-Muller N(0,1): out[i] depends only on (seed, stream_id, i) — the same values
 * for any batch sharding across GPUs.  n % 4 == 0. */
int syn_randn(float* out, int64_t n, uint64_t seed, uint64_t stream_id, int64_t first_index, void* stream);

/* ---- RVQ-VAE either side of the loop (eval mode) ---------------------------------------------
 * One Conv1d of models/vq/encdec.py / resnet.py on channels-last activations:
 *   out[t][co] = bias[co] + sum_{tap,ci} W[co][ci][tap] * relu_in?(in[(t*stride + tap*dil - pad) >> up][ci])  (+ resid[t][co])
 * x_bf16 [clips][t_in][cin] (MFMA operand), outputs fp32 [clips][t_out][ldy] (first cout_valid channels) and/or bf16
 * [clips][t_out][cout].  w_packed: fragments [taps][cout/16][cin/32][64 lanes][8 bf16] (host: rvqvae.pack_conv).
 * up = 1 folds nn.Upsample(scale_factor=2, nearest) (encdec.py:56) into the read.  cin % 32 == 0, cout % 128 == 0
 * (callers zero-pad; cout_valid = real channel count). */
typedef struct syn_vq_conv {
    const void*  w_packed;
    const float* bias;            /* [cout] */
    int32_t cin, cout, cout_valid, taps, stride, dil, pad, up, relu_in, relu_out;
} syn_vq_conv;
int syn_vq_conv1d(const syn_vq_conv* cv, const void* x_bf16, const float* resid, float* y_f32, int32_t ldy, void* y_bf16,
                  int32_t clips, int32_t t_in, int32_t t_out, void* stream);

/* The whole model of one body part (models/vq/model.py:RVQVAE as diffusion_rvqvae_trainer.py:105-150 builds it):
 * enc = model.0, then per down stage { k4 s2 conv, 3 x (conv1, conv2) with dilations 9, 3, 1 }, final conv (16 convs);
 * dec = model.0, per up stage { 3 x (conv1, conv2), the conv behind nn.Upsample }, model.4, model.6 (17 convs). */
typedef struct syn_vq_model {
    int32_t pose_dim, reserved;
    syn_vq_conv enc[16];
    syn_vq_conv dec[17];
    const float* codebooks;       /* [6][512][512] */
    const float* codebooks_t;     /* [6][dim][code] */
    const float* code_sq;         /* [6][512] */
} syn_vq_model;
/* bytes of scratch for `clips` clips of `t_pose` pose frames (t_pose = 4 x latent rows) */
int64_t syn_vq_workspace_bytes(int32_t clips, int32_t t_pose, int32_t pose_dim);
/* RVQVAE.map2latent (models/vq/model.py:95-100): pose fp32 [clips][t_pose][pose_dim] -> latent fp32 [clips][t_pose/4][512] */
int syn_vq_map2latent(const syn_vq_model* m, const float* pose, int32_t clips, int32_t t_pose, void* workspace, float* latent,
                      void* stream);
/* RVQVAE.latent2origin (:102-109): latent fp32 [clips][t_lat][512] -> pose fp32 [clips][4 t_lat][pose_dim]; idx, sqerr,
 * hist as in syn_vq_quantize (hist zeroed by the caller). */
int syn_vq_latent2origin(const syn_vq_model* m, const float* latent, int32_t clips, int32_t t_lat, void* workspace, float* pose_out,
                         int32_t* idx, float* sqerr, int32_t* hist, void* stream);
/* RVQVAE.forward_decoder (:86-93): indices [clips][t_lat][n_q] -> pose fp32 [clips][4 t_lat][pose_dim] */
int syn_vq_forward_decoder(const syn_vq_model* m, const int32_t* idx, int32_t n_q, int32_t clips, int32_t t_lat, void* workspace,
                           float* pose_out, void* stream);

/* ---- pose formats either side of the RVQ-VAEs (diffusion_rvqvae_trainer.py:257-272 `_load_data`, :503-531 `_g_test`) ---------------------------
 * SMPL-X joint rotations: axis-angle fp32 [n_joints][3] <-> the 6D representation fp32 [n_joints][6] (first two rows of the rotation matrix).
 * syn_axis_angle_to_rot6d replaces rc.matrix_to_rotation_6d(rc.axis_angle_to_matrix(x)) (utils/rotation_conversions.py:416-430, 448-477, 36-64,
 * 535-550); syn_rot6d_to_axis_angle replaces rc.matrix_to_axis_angle(rc.rotation_6d_to_matrix(x)) (:511-533, 96-118, 432-446, 480-508).  The
 * reference's arithmetic operation for operation (small-angle series below 1e-6 rad, sqrt-of-positive-part + copysign quaternion); in and
 * out may not alias; n_joints = 0 is a no-op. */
int syn_axis_angle_to_rot6d(const float* axis_angle, int64_t n_joints, float* rot6d, void* stream);
int syn_rot6d_to_axis_angle(const float* rot6d, int64_t n_joints, float* axis_angle, void* stream);

/* ResidualVQ.forward in eval mode (models/vq/residual_vq.py:91-140 over quantizer.py:62-69,143-171), fp32, 6 layers
 * of 512 codes x 512 dims: x [rows][512] -> q_f32 / q_bf16 [rows][512] (sum of the straight-through outputs), idx
 * [rows][6], sqerr [syn_vq_quantize_groups(rows)][6] (per-group sums of |residual - code|^2: commit loss numerators),
 * hist [6][512] (code usage for the perplexity; the caller zeroes it).  codebooks [6][512][512], codebooks_t
 * [6][dim][code] (transposed), code_sq [6][512] = |code|^2. */
int32_t syn_vq_quantize_groups(int32_t rows);
int syn_vq_quantize(const float* x, const float* codebooks, const float* codebooks_t, const float* code_sq, float* q_f32,
                    void* q_bf16, int32_t* idx, float* sqerr, int32_t* hist, int32_t rows, void* stream);
/* Sum of the codes of given indices (RVQVAE.forward_decoder, models/vq/model.py:86-89): idx [rows][n_q], -1 = no code. */
int syn_vq_codes(const int32_t* idx, const float* codebooks, float* q_f32, void* q_bf16, int32_t rows, int32_t n_q, void* stream);

/* ---- training building block ------------------------------------------------------------------------
 * y[m][n] = sum_k x[m][k] * W[n][k] (+ bias[n]): nn.Linear forward on the MFMA GEMM (bf16 operands, fp32 accumulate
 * and output).  The same entry point serves the backward passes with re-packed operands:
 *   dgrad  dx = dy . W      -> x := dy (m x n_out),   w_packed := pack(W^T)   (n_in x n_out)
 *   wgrad  dW = dy^T . x    -> x := dy^T (n_out x m), w_packed := pack(x^T)   (n_in x m)
 * n % 128 == 0, k % 128 == 0 (callers zero-pad; n % 512 != 0 runs on the 128-column tiles, ABI 7).  Replaces torch.nn.functional.linear for
 * models/timm_transformer/transformer.py:85,102,146,149 and models/denoiser.py:148,162,170,195 in training. */
int syn_linear(const void* x_bf16, const void* w_packed, const float* bias, int32_t m_rows, int32_t n, int32_t k, float* y,
               void* stream);
/* syn_linear and, in the same launch, syn_pack_weight_t(x_bf16, 1, k, m_rows, xt_packed): the fragments of x^T the weight-gradient
 * GEMM of the backward will take, packed in the shadow of the forward GEMM (which fills a quarter to three quarters of the chip). */
int syn_linear_and_pack(const void* x_bf16, const void* w_packed, const float* bias, int32_t m_rows, int32_t n, int32_t k, float* y,
                        void* xt_packed, void* stream);
/* Two independent syn_linear calls (no bias) as ONE launch: the data-gradient GEMM dy . W and the weight-gradient GEMM dy^T . x of
 * an nn.Linear's backward fill half the chip each and do not depend on each other.  Same result as two calls, bitwise.
 * (ABI 8) n % 128 == 0 (512-column or 128-column tiles; shapes neither form takes as a pair go out as two launches + the bias sum). */
/* (ABI 5) bias_grad [part_n] (NULL to skip) = sum over i < part_rows of bias_parts [i][part_n], added in order by the same launch: the
 * Linear's bias gradient from syn_linear_bwd_prep's colsum_part without a reduction launch of its own. */
int syn_linear_pair(const void* x1_bf16, const void* w1_packed, int32_t m1, int32_t n1, int32_t k1, float* y1,
                    const void* x2_bf16, const void* w2_packed, int32_t m2, int32_t n2, int32_t k2, float* y2,
                    const float* bias_parts, int32_t part_rows, int32_t part_n, float* bias_grad, void* stream);
/* (ABI 5) The last Linear of a pre-LN residual branch with the branch's tail in its epilogue (transformer.py:195-198,
 * x = x + drop_path(attn(norm1(x))) / x + drop_path(mlp(norm2(x)))): y = residual + row_scale[m / rows_per_scale] * (x W^T + bias),
 * residual and y fp32 [m_rows][n] (may alias), row_scale NULL = 1 (DropPath off), one factor per sample = rows_per_scale rows.
 * xt_packed (NULL to skip): as syn_linear_and_pack. */
int syn_linear_res(const void* x_bf16, const void* w_packed, const float* bias, const float* residual, const float* row_scale,
                   int32_t rows_per_scale, int32_t m_rows, int32_t n, int32_t k, float* y, void* xt_packed, void* stream);
/* (ABI 5) fc1 of the MLP (transformer.py:117-151) with the GELU behind it in the epilogue: y fp32 [m_rows][n] = x W^T + bias (kept for the
 * backward) and y_gelu_bf16 = bf16(GELU(y)) (exact erf form), the operand fc2 takes.  xt_packed (NULL to skip): as syn_linear_and_pack. */
int syn_linear_gelu(const void* x_bf16, const void* w_packed, const float* bias, int32_t m_rows, int32_t n, int32_t k, float* y, void* y_gelu_bf16,
                    void* xt_packed, void* stream);

/* ---- single stages, exported for unit tests and bisecting ----------------------------------- */
/* Y[m][n] = sum_k X[m][k] * W[n][k] (+ bias[n]); X bf16 [m_rows][k], W packed, Y fp32 [m_rows][n]. n % 512 == 0. */
int syn_test_gemm(const void* x_bf16, const void* w_packed, const float* bias, int32_t m_rows, int32_t n, int32_t k,
                  int32_t m_tile, float* y, void* stream);
/* Adversarial check of the XCD-group barrier's acquire (syn_latency.inc): on every XCD one workgroup republishes `words`
 * (<= 4096) 32-bit words `rounds` times with a new payload each round, a second workgroup of the same XCD re-reads them
 * (L1-warm) after the flag, under mode 0: no invalidate, 1: `buffer_inv sc0` (what the barrier uses), 2: `buffer_inv sc1`,
 * while the other 30 workgroups of the XCD stream from `stream` (NULL: idle).  stale[xcd] (+= ) counts words read with an
 * old value.  sync [320] and stale [9] zeroed by the caller; buf [8][4096]. */
int syn_test_handoff(uint32_t* sync_320_zeroed, uint32_t* buf_8x4096, const float* stream, int64_t stream_n, uint32_t* stale_9_zeroed,
                     int32_t words, int32_t rounds, int32_t mode, void* stream_h);
/* (ABI 7) the matrix pipe alone in k_seq's occupancy (one wave per SIMD, 12 independent v_mfma_f32_32x32x16_bf16 accumulators, register operands): iters x 192
 * MFMAs per wave on every CU; out: 256 floats per CU (keeps the loop alive); *flops (host, may be NULL) = the floating-point operations of the launch.
 * Timed by bench.py as roofline.practical_peak: what the chip delivers on this instruction at the clock it holds under that load. */
int syn_test_mfma_rate(int32_t iters, float* out, int64_t* flops, void* stream);
/* attention over ws_q/ws_k/ws_vt -> ws_o for n_seq sequences of 32 tokens, 4 heads x 128. */
int syn_test_attention(const void* q, const void* k, const void* vt, int32_t n_seq, void* o, void* stream);

/* ---- diagnostics (scripts/diag_*.py, scripts/ubench_*.py, A/B environment switches; never called by the product path in its default configuration:
 * syn_debug_conv_terms is reached from syntalker_amd/training.py only when SYN_CONV_TERMS asks for something other than 3,3,3) ------
 * Process-wide switches of the library, not thread-safe, no status to return.  They replace nothing in the reference. */
/* per-phase cycle counters of the step kernels: the attention / MLP phases write s_memtime stamps into these device buffers (NULL: off).
 * The first buffer also takes the training convolutions' stamps while it is set: k_conv_train 8 slots per workgroup (start, tile staged, barrier passed,
 * k loop done, end, HW_ID: scripts/diag_conv_phases.py), k_conv_wgrad 8 per workgroup (cycles in the tile stores, in the k loop, in all, chunks:
 * scripts/diag_wgrad_phases.py) - size it for the launch's workgroups. */
void syn_debug_timing(long long* attn_buf, long long* mlp_buf);
/* 16-row plain GEMMs of the training step: 1 (default) = activation block resident in the LDS, 0 = the streaming loop (A/B, bitwise equal) */
void syn_debug_gemm_resident(int on);
/* pin syn_linear's row tile (16 / 32 / 64 / 128); 0 = automatic */
void syn_debug_linear_tile(int rows);
/* training-mode convolutions (syn_conv1d_train_*): which cross products of the hi / lo operand split the NEXT launches issue - bit 0: A_lo . B_hi
 * (A = weights, or dy in the weight gradient), bit 1: A_hi . B_lo (B = the activations: x, or dy in the data gradient); 3 = both, fp32-grade; a negative mask restores the defaults (3 forward / data gradient,
 * 1 weight gradient: x is read rounded to bf16 there, the sum over positions averages it out) */
void syn_debug_conv_terms(int mask);
/* k_seq: delay workgroup i by (i % 8) * units_of_64_cycles * 64 cycles at launch (phase experiments); -1 = off */
void syn_debug_seq_skew(int units_of_64_cycles);
/* k_seq: which step of a multi-step launch syn_debug_timing's stamps are taken in */
void syn_debug_seq_step(int step);

#ifdef __cplusplus
}
#endif
#endif /* SYN_HIP_H */
