/*
 * d4hip — C-ABI of the MI355X-native (gfx950) implementation of dreamer4's imagination hot path.
 *
 * The reference (lucidrains/dreamer4, /root/reference/dreamer4/dreamer4.py = "D4") has no FFI or
 * operator boundary: its API is the Python nn.Module surface.  This header is the boundary a
 * maintainer would bind instead of the ATen op sequences of
 *
 *   DynamicsWorldModel.generate               D4:6308-6774   -> d4_rollout
 *   DynamicsWorldModel.forward (inference)    D4:6792-7295   -> d4_wm_forward
 *   DynamicsWorldModel.learn_from_experience  D4:5893-6305   -> d4_learn (+ d4_optim_step)
 *
 * Conventions
 *   - plain C, no torch types: device pointers + sizes; `stream` is a hipStream_t passed as void*.
 *   - every function returns 0 on success; otherwise a non-zero code and d4_last_error() holds the
 *     message (shape / config / HIP errors).  Nothing is retried or silently replaced by a fallback.
 *   - all tensors are dense fp32 row-major unless stated; integer tensors are int64 (actions, lens,
 *     tasks) to match torch.long on the Python side; booleans are uint8.
 *   - the engine never allocates device memory: the caller owns one workspace buffer
 *     (d4_engine_workspace_bytes) and all weight / IO tensors.  Kernels are enqueued on `stream`
 *     with no host synchronisation, so calls are hipGraph-capturable.
 *   - weights are bound by the reference's state_dict key names (weight interchange = flat
 *     {key: tensor}), see d4_engine_bind.
 */
#ifndef D4HIP_H
#define D4HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D4_MAX_ACTION_TYPES 8
#define D4_MODE_DYNAMICS 0
#define D4_MODE_DECODER 1
#define D4_MODE_ENCODER 2
#define D4_MLP_PRE_RMS 0        /* RMSNorm -> Linear -> SiLU                        (x_mlps_pytorch create_mlp: recipe unpinned) */
#define D4_MLP_POST_LAYER 1     /* Linear -> LayerNorm -> SiLU, bare last Linear */
/* Link of the Beta policy head's raw outputs (discrete_continuous_embed_readout BetaDist, unimodal=True at D4:1172-1173: alpha, beta >= 1).
 * The package is absent from the image, so the link is a descriptor like the MLP recipe (csrc/beta.h). */
#define D4_BETA_SOFTPLUS_P1 0   /* alpha = softplus(raw0) + 1, beta = softplus(raw1) + 1 */
#define D4_BETA_EXP_P1 1        /* alpha = exp(raw0) + 1,      beta = exp(raw1) + 1 */

/* Constructor arguments of DynamicsWorldModel (supported subset; names as D4:4662-4778). */
typedef struct d4_config {
    int32_t dim, dim_latent, num_latent_tokens, depth, time_block_every;
    int32_t attn_heads, attn_dim_head;          /* attn_dim_head: 16, 32 or 64 (a head row lives in one wavefront) */
    float attn_softclamp_value;
    int32_t num_spatial_tokens, num_register_tokens, max_steps, num_tasks;
    int32_t num_discrete_action_types;
    int32_t num_discrete_actions[D4_MAX_ACTION_TYPES];
    int32_t num_continuous_actions;             /* Beta policy head: continuous_dist_type='beta', no norm stats (D4:1131, 1172-1196) */
    int32_t multi_token_pred_len;
    int32_t policy_head_mlp_depth, value_head_mlp_depth, terminal_mlp_depth, predict_terminals;
    int32_t reward_num_bins, value_num_bins;
    int32_t reward_encoder_type;                /* 0 = hl_gauss (default, D4:1041), 1 = symexp_two_hot (D4:947): bins = symexp(linspace), two-hot targets */
    int32_t matmul_bf16;                        /* trunk GEMM arithmetic.  0: fp32 on the f32-input MFMA; 2: fp32 on the bf16 matrix cores by operand
                                                   splitting (three bf16 planes per operand, six products, fp32 accumulate: fp32 accuracy,
                                                   csrc/gemm_x3.hip) — the Python mirror's default; 1: bf16 MFMA (bf16-rounded weights +
                                                   activations, fp32 accumulate, fp32 norms / softmax / residual stream); 3: opt-in fp32-class mode on
                                                   the fp16 matrix cores (two fp16 planes per operand under exact row scales, three products:
                                                   csrc/gemm_h2.hip, see d4_gemm_split2) */
    int32_t head_mlp_recipe;                    /* D4_MLP_PRE_RMS / D4_MLP_POST_LAYER: layer recipe of the policy / value / terminal MLPs (engine.h) */
    int32_t continuous_beta_param;              /* D4_BETA_SOFTPLUS_P1 / D4_BETA_EXP_P1: link of the Beta head's raw parameters (csrc/beta.h) */
    int32_t pool_heads, pool_dim_head;          /* AttentionPool defaults 4 x 64 (D4:2147-2148) */
    /* learn_from_experience hyper-parameters (D4:4731-4744) */
    float gae_discount_factor, gae_lambda, ppo_eps_clip, policy_entropy_weight;
    int32_t use_delight_gating;
    float delight_temperature, pmpo_pos_to_neg_weight, pmpo_kl_div_loss_weight;
    int32_t pmpo_reverse_kl;
    float hl_gauss_sigma_to_bin_ratio, hl_gauss_eps, value_min, value_max;
    /* mode D4_MODE_DECODER / D4_MODE_ENCODER: the engine is the video tokenizer's decoder (VideoDecoderNetwork, D4:3490-3682) or its
     * encoder (encoder_transformer, D4:3912-3933) instead of the dynamics model — the same trunk kernels over [patches | latent tokens] per frame; constructor arguments of VideoTokenizer (D4:3686-3764). */
    int32_t mode;
    int32_t patch_size, channels, image_height, image_width, decoder_flow_steps, decoder_pos_mlp_depth;
    /* capacities the workspace is sized for */
    int32_t max_batch;            /* trajectories per call */
    int32_t max_frames;           /* KV-cache capacity in frames (prompt + generated) */
    int32_t max_parallel_frames;  /* frames evaluated in one parallel pass (1 = cached decode only) */
    int32_t max_learn_rows;       /* batch * time rows of one learn_from_experience call (0 = no learner) */
} d4_config;

typedef struct d4_engine d4_engine;

const char* d4_last_error(void);
int d4_version(void);

int d4_engine_create(const d4_config* cfg, d4_engine** out);
void d4_engine_destroy(d4_engine* e);

/* Bytes of device workspace the engine needs (prepared weights + activations + KV cache). */
size_t d4_engine_workspace_bytes(const d4_engine* e);
int d4_engine_set_workspace(d4_engine* e, void* device_ptr, size_t bytes);

/* Bind one parameter/buffer by its reference state_dict key, e.g.
 * "transformer.layers.3.2.fn.to_q.weight" (layout list: DESIGN.md "Weight interchange").
 * `grad` may be NULL; when given, d4_learn writes d(loss)/d(param) there (policy / value heads).
 * Pseudo-keys for buffers the reference builds in its constructor:
 *   "reward_encoder.centers" [reward_num_bins], "value_encoder.centers" [value_num_bins],
 *   "value_encoder.support" [value_num_bins + 1]   (hl_gauss);
 *   symexp_two_hot reads the reference's own buffers "reward_encoder.bin_values" / "value_encoder.bin_values" [num_bins]. */
int d4_engine_bind(d4_engine* e, const char* key, const float* device_ptr, float* grad, int64_t numel);

/* Build the fused / gamma-folded weight images in the workspace.  Call after binding, and again
 * whenever a trunk weight changes (head MLP weights are read in place and need no re-prepare). */
int d4_engine_prepare(d4_engine* e, void* stream);

/* Number of frames currently held by the time KV cache (token_count of D4:3261). */
int d4_engine_cache_frames(const d4_engine* e);
int d4_engine_cache_reset(d4_engine* e, int frames);   /* set the frame counter: 0 = empty, < current = truncate; moving it forward
                                                          again is valid while the slots in between have not been rewritten */
/* Export / import the cache in the reference layout (time_layers, 2, B*S, heads, t, 64)  D4:2075, 3256. */
int d4_engine_cache_export(d4_engine* e, float* dst, int batch, void* stream);
int d4_engine_cache_import(d4_engine* e, const float* src, int batch, int frames, void* stream);

/* DynamicsWorldModel.forward(latents=..., signal_levels=..., step_sizes=..., discrete_actions=...,
 * time_cache=..., latent_is_noised=True, return_pred_only=True, return_intermediates=True)
 * D4:6792-7295.  Evaluates `frames` new frames per trajectory (row order b, t):
 *   latents        [batch][frames][n][dl]
 *   signal_levels  [batch][frames] int32 in [0, max_steps)
 *   prev_actions   [batch][frames][action_types] int64: the action token of each frame, i.e. the
 *                  action taken at the previous frame; a negative first entry means "no action"
 *                  (zero token, D4:7110-7126).  NULL = zero tokens everywhere (models with discrete actions).
 *   prev_cont      [batch][frames][num_continuous_actions] float, same pairing; "no action" = NaN in the first
 *                  entry (only consulted when the model has no discrete actions).  NULL allowed as above.
 *   tasks          [batch] int64 or NULL
 *   use_cache      attend over the frames already in the KV cache (their count is the rotary offset)
 *   commit_cache   keep the new frames' K/V in the cache (the "extra clean step" of D4:6545)
 * Outputs: pred [batch][frames][n][dl], agent_embed [batch][frames][dim]. */
int d4_wm_forward(d4_engine* e, const float* latents, const int32_t* signal_levels, int step_size,
                  const int64_t* prev_actions, const float* prev_cont, const int64_t* tasks, int batch, int frames,
                  int use_cache, int commit_cache, float* pred, float* agent_embed, void* stream);

/* VideoTokenizer.decode_step (D4:4137-4184) on an engine created with mode = D4_MODE_DECODER: one evaluation of the decoder.
 *   latents       [batch][frames][n][dl]
 *   noised_video  [batch][channels][frames][H][W]   the flow sample being denoised (pure noise on the first step, D4:4212)
 *   time_index    flow step i in [0, decoder_flow_steps)   (time_embed row, D4:4155)
 * Output: pred_video [batch][channels][frames][H][W] (the predicted clean video).  VideoTokenizer.decode's Euler update between
 * steps (D4:4226-4230) is d4_euler_step. */
int d4_decoder_forward(d4_engine* e, const float* latents, const float* noised_video, int time_index, int batch, int frames,
                       float* pred_video, void* stream);
/* VideoTokenizer.tokenize (D4:4107-4113 -> forward(return_latents=True) in eval mode, D4:4239-4433) on an engine created with
 * mode = D4_MODE_ENCODER (depth = encoder_depth): video [batch][channels][frames][H][W] -> latents [batch][frames][n][dl] in (-1, 1). */
int d4_encoder_forward(d4_engine* e, const float* video, int batch, int frames, float* latents, void* stream);
/* x += (pred - x) / one_minus_t * dt, elementwise over n floats: the flow-matching Euler step of generate (D4:6567-6580) and of
 * VideoTokenizer.decode (D4:4226-4230). */
int d4_euler_step(float* x, const float* pred, int64_t n, float one_minus_t, float dt, void* stream);

/* DynamicsWorldModel.generate(...) D4:6308-6774 with every random draw injected. */
typedef struct d4_rollout_io {
    int32_t batch;
    int32_t time_steps;            /* total frames incl. prompt (while latents.shape[1] < time_steps) */
    int32_t prompt_frames;         /* frames already present in `latents` (teacher forcing, D4:6392) */
    int32_t num_steps;             /* denoising steps K (power of two) */
    int32_t use_time_cache;        /* D4:6321 */
    int32_t sample_terminals;      /* return_terminals && predict_terminals (D4:6454) */
    int32_t sample_actions;        /* return_agent_actions (D4:6625): policy head + Gumbel sample + value head */
    float context_signal_noise;    /* D4:6319 */
    float discrete_temperature;
    float continuous_temperature;
    /* injected noise, one slice per generated frame f = 0 .. time_steps - prompt_frames - 1 */
    const float* noise_latent;     /* [F][batch][n][dl] normal   (D4:6475) */
    const float* noise_context;    /* [F][batch][n][dl] normal   (D4:6670); may be NULL when use_time_cache */
    const float* gumbel_u;         /* [F][batch][A] uniform (0,1) (MultiCategorical.sample) */
    const float* bern_u;           /* [F][batch] uniform         (D4:6611); NULL unless sample_terminals */
    const float* beta_noise;       /* [F][batch][nc][2][6][2] (normal, uniform) per gamma rejection round (Readout.sample_continuous) */
    const int64_t* tasks;          /* [batch] or NULL */
    /* in/out histories, time-major stride = time_steps; prompt entries pre-filled by the caller */
    float* latents;                /* [batch][time_steps][n][dl]  (unclamped; caller clamps, D4:6686) */
    int64_t* actions;              /* [batch][time_steps][action_types] */
    float* actions_cont;           /* [batch][time_steps][nc] continuous actions in the Beta's native (0, 1) range */
    float* rewards;                /* [batch][time_steps] */
    float* ctx_hist;               /* [batch][time_steps][n][dl] fixed context noise per frame (prompt entries =
                                      the prompt latents, D4:6400); NULL allowed when use_time_cache */
    /* outputs for generated frames only, index = frame - prompt_frames, stride = F */
    float* agent_embed;            /* [batch][F][dim] */
    float* log_probs;              /* [batch][F][action_types] */
    float* log_probs_cont;         /* [batch][F][nc] */
    float* cont_params;            /* [batch][F][nc][2] raw Beta parameters (old_action_unembeds.continuous, D4:6749-6750) */
    float* values;                 /* [batch][F] */
    float* action_logits;          /* [batch][F][A]   (old_action_unembeds, D4:6749-6750) */
    int64_t* lens;                 /* [batch]  pre-filled with time_steps */
    uint8_t* terminals;            /* [batch]  pre-filled with 0 */
} d4_rollout_io;

int d4_rollout(d4_engine* e, const d4_rollout_io* io, void* stream);

/* DynamicsWorldModel.learn_from_experience(experience, only_learn_policy_value_heads=True,
 * objective=...) D4:5893-6305 with stored agent embeds: losses AND gradients of the policy head
 * (policy MLP + discrete_action_unembed) and the value head into the `grad` buffers given at bind. */
typedef struct d4_learn_io {
    int32_t batch, time;
    int32_t objective;             /* 0 = ppo, 1 = spo, 2 = pmpo */
    int32_t normalize_advantages;  /* -1 = default (objective != pmpo) */
    float eps;                     /* z-score epsilon (1e-6, D4:5903) */
    int32_t use_delight_gating;    /* -1 = engine config default */
    float delight_temperature;     /* <= 0 = engine config default */
    const float* agent_embed;      /* [batch][time][dim] */
    const int64_t* actions;        /* [batch][time][action_types] */
    const float* old_log_probs;    /* [batch][time][action_types] */
    const float* actions_cont;     /* [batch][time][nc] */
    const float* old_log_probs_cont;   /* [batch][time][nc] */
    const float* old_cont_params;  /* [batch][time][nc][2] (pmpo KL) or NULL */
    const float* old_values;       /* [batch][time] */
    const float* rewards;          /* [batch][time] */
    const float* old_action_logits;/* [batch][time][A] (pmpo KL) or NULL */
    const int64_t* lens;           /* [batch] */
    const uint8_t* is_truncated;   /* [batch] */
    const uint8_t* terminals;      /* [batch] */
    /* global-batch statistics for data parallel runs: host callback that sum-reduces `n` floats in
     * place across ranks (NULL = single process).  Called on the host between kernel phases. */
    int (*allreduce_sum)(float* device_buf, int n, void* user);
    void* allreduce_user;
    float* losses;                 /* device [2]: total_policy_loss, value_loss */
    float* returns;                /* optional device [batch][time] */
    /* optional device [batch][time][dim] each: d total_policy_loss / d agent_embed and d value_loss / d agent_embed, for
     * learn_from_experience(only_learn_policy_value_heads=False) (D4:6045-6075): the caller backpropagates them through the world model
     * that produced agent_embed (dreamer4_amd/trunk_ops.py).  NULL (the default path): not computed. */
    float* d_agent_embed_policy;
    float* d_agent_embed_value;
} d4_learn_io;

int d4_learn(d4_engine* e, const d4_learn_io* io, void* stream);

/* clip_grad_norm_(params, max_norm) followed by one torch.optim.AdamW step on a flat parameter group,
 * as DreamTrainer does per head (trainers.py:1436-1452).  `grad_scale` multiplies the gradient first
 * (1/world_size after a SUM all-reduce of per-rank-mean gradients; 1 with global statistics).
 * exp_avg / exp_avg_sq: caller-owned optimiser state (zero-initialised); scratch: >= 1025 floats;
 * scratch[0] receives the (scaled) total gradient norm. */
int d4_adamw_clip(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                  float grad_scale, float* scratch, void* stream);

/* Measurement hook (bench.py roofline leg): `mask` bit c (c < 27) enables tile configuration c of the GEMM kernels; every
 * (mask >> 27)-th launch (0 -> every launch) of an enabled configuration carries a HIP event pair on its dispatch (launch stream); d4_profile_read sums elapsed ms / algorithmic flops /
 * launches per configuration and clears the log.  d4_profile_classes() configurations exist; d4_profile_class_name(c) is the
 * prefix of the kernel name rocprofv3 reports for configuration c ("gemm_kernel<BM, BN, WGM, WGN, BK, 1"). */
/* bf16 path: every `stride`-th bf16 GEMM launch carries an event pair (0 = off); read sums ms / flops / launches and clears. */
int d4_profile_bf16_enable(int stride);
int d4_profile_bf16_read(double* ms, double* flops, int64_t* count);
int d4_profile_enable(int mask);
int d4_profile_read(double* ms, double* flops, int64_t* count, int nclass);
int d4_profile_classes(void);
const char* d4_profile_class_name(int c);
/* The same for the non-GEMM kernel classes of the rollout (within-frame / time attention, KV append, attention-pool mix, the small
 * attention forms, token assembly, split-K reduce): d4_profile_glue_read sums elapsed ms / ALGORITHMIC HBM bytes / launches per class.
 * Mask and stride as d4_profile_enable.  (bench.py's HBM roofline leg; replaces nothing in the reference.) */
int d4_profile_glue_enable(int mask);
int d4_profile_glue_read(double* ms, double* bytes, int64_t* count, int nclass);
int d4_profile_glue_classes(void);
int d4_profile_glue_read_flops(double* flops, int nclass);      /* matrix work the per-frame fused classes carry; call before d4_profile_glue_read */
const char* d4_profile_glue_class_name(int c);

/* Measured denominators for the roofline fractions (nothing of the reference is replaced: it has no measurement layer): what THIS device
 * sustains on a float4 stream copy through `scratch` (two halves; >= 1 GiB so that the 256 MB Infinity Cache cannot serve it; read + write GB/s) and on
 * bare v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x16_bf16 streams with random operands (TFLOP/s).  Well under a second; synchronises the stream. */
int d4_measure_peaks(void* scratch, size_t scratch_bytes, double* hbm_copy_gbs, double* mfma_f32_tflops, double* mfma_bf16_tflops, void* stream);

/* Test hook: run GEMM tile configuration `id` wherever it is valid instead of the tuned / static choice (-1 restores it);
 * 100 + c: configuration c of the second fp32 family (gemm2.hip); 200 + c: configuration c of the bf16 kernel; 300 + c: tile c of the
 * split-operand fp32 family (gemm_x3.hip); 400 + c: tile c of the fp16x2 family (gemm_h2.hip); 500 + c: tile c of the bf16-activation kernel (gemm_bf16a.hip).
 * 199: the few-row / long-K form of the second family (gemm2_ksplit_kernel) on every call it can run (normally taken by a rule on the shape).
 * ONE family is forced at a time: every call first clears the hooks of the other families (so forcing, say, 400 + c also switches off the special forms
 * that only run un-forced: the persistent split-operand kernel, the pair launches).  Returns the number of configurations of the family addressed
 * (id = -1: of the first family).  Every configuration of a family must produce the same bits (tests/test_gpu_kernels.py). */
/* Test hook for the per-frame fused block tails (csrc/frame_fused.hip; default 1): 0 separate kernels, 1 fused, 2 fused tails with the
 * pool mix as its own kernel.  Returns the previous mode.  Which path runs is otherwise a rule on the call's shape. */
int d4_frame_fused_set(int mode);
/* Test hook for the two bit-identical launch fusions of the cached decode (each is asserted bitwise against its two-launch form): name =
 * "time_attn_fused_append" (KV append inside the time attention, csrc/attn.hip) or "attn_out_cols" (attention inside the column-split output
 * projection at <= 4 frames, csrc/frame_fused.hip); value 1 (default) fused, 0 two launches.  Returns the previous value, -1 for an unknown name.
 * "pool_wide_keys" (bf16 engine only; NOT bit-identical: the queries become bf16): 1 (default) a hidden is projected once, when produced, onto the key
 * weights of every later attention pool and the query weights of the pool it feeds; 0 a query launch + a key launch over the whole stack per pool.
 * Read when a frame is ENQUEUED: a captured hipGraph keeps the form it was captured with. */
int d4_debug_switch(const char* name, int value);
int d4_gemm_force_config(int id);

/* Test hook: device address of an engine-internal activation buffer (names: engine.hip d4_debug_buffer). */
int d4_debug_buffer(d4_engine* e, const char* name, float** ptr);

/* ---- single-kernel entry points (parity tests call the same launchers the engine uses) ---- */
int d4_gemm(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias,
            const float* R, int ldr, int M, int N, int K, int flags, float rms_eps, void* stream);
/* Two independent Linear layers of equal in-features (the AttentionPool's query and key projections, dreamer4.py:2143-2160) in ONE
 * launch: C1 = f(A1 W1^T), C2 = f(A2 W2^T), W row-major [N][K] (ldw = K), `flags` as d4_gemm (row scale only).  Bit-identical to two
 * d4_gemm calls; falls back to them when the grouped form does not apply. */
int d4_gemm_pair(const float* A1, int lda1, const float* W1, float* C1, int ldc1, int M1, int N1, const float* A2, int lda2, const float* W2, float* C2,
                 int ldc2, int M2, int N2, int K, int flags, float rms_eps, void* stream);
/* Gradient of a Linear's weight (backward of dreamer4.py:2079-2116, 1968-2075): C[M][N] = A^T B, A [K][lda] (the output gradient, M columns),
 * B [K][ldb] (the layer input, N columns), both row-major with the contraction over their ROWS.  M, N, lda, ldb, ldc multiples of 4, 16-byte
 * aligned operands.  `part` (part_floats floats, or NULL) holds the partial products when the rows are split into slices; tile_n / slices = 0
 * take the shape rule (csrc/gemm_tn.hip), other values force the 64 x (64 tile_n) tile and the slice count (benchmarks).  Deterministic. */
int d4_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, float* part, int64_t part_floats,
               int tile_n, int slices, void* stream);
/* strided-batch form (the AttentionPool's per-head value projection, D4:2143-2177): problem b reads A + b*strideA,
 * W + b*strideW and writes C (and R) + b*strideC   (strides in elements). */
int d4_gemm_batched(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias,
                    const float* R, int ldr, int M, int N, int K, int flags, float rms_eps, int batch,
                    int64_t strideA, int64_t strideW, int64_t strideC, void* stream);
/* the bf16 MFMA kernel alone: A fp32 [M][lda] (rounded to bf16 on the way in), Wb bf16 [N][ldw] (raw 16-bit patterns), C fp32 */
int d4_gemm_bf16(const float* A, int lda, const uint16_t* Wb, int ldw, float* C, int ldc, const float* bias,
                 const float* R, int ldr, int M, int N, int K, int flags, float rms_eps, void* stream);
/* The same Linear with the activations already in bf16 (the producer's bf16 copy): Ab bf16 [M][lda], Wb bf16 [N][ldw], C fp32 [M][ldc], Cb (optional)
 * = bf16(C) at the same leading dimension for the next layer.  K % 64 == 0, lda / ldw % 8 == 0.  LDS-DMA ring kernel (csrc/gemm_bf16a.hip); `config`
 * = -1: tile by the shape rule, else one of its configurations.  d4_cvt_bf16: fp32 -> bf16 (round to nearest even), n elements. */
int d4_gemm_bf16a(const uint16_t* Ab, int lda, const uint16_t* Wb, int ldw, float* C, int ldc, uint16_t* Cb, const float* bias, const float* R, int ldr,
                  int M, int N, int K, int flags, float rms_eps, int config, void* stream);
int d4_cvt_bf16(const float* src, uint16_t* dst, int64_t n, void* stream);
/* The producer -> consumer hand-off of the bf16 engine (the reference keeps one activation tensor per nn.Linear input, dreamer4.py:1983-2068,
 * 2105-2116, 2143-2177; here a bf16 IMAGE of it is what the next Linear reads): C may be NULL when only the bf16 image Cb is wanted; `batch` > 1 runs
 * `batch` independent products A + b * strideA, Wb + b * strideW -> C / Cb + b * strideC (the attention pool's per-head value projection).
 * d4_cvt_rows_bf16: rows x cols of a strided fp32 matrix -> bf16 (round to nearest even), the pass behind producers that cannot write the image. */
/* ... with the engine's second, ROW-COMPACTED copy of the output (the rows the final attention pool and the latent head read: token rows s = m % c2_S with
 * c2_lo <= s < c2_hi, plus the last token of a frame when c2_last; fp32 `C2` and optionally its bf16 image `C2b`, [frames * (c2_hi - c2_lo + c2_last)][ldc2]).
 * It replaces the reference's indexing of the layer hiddens at D4:3242-3246 / 7251 (the trunk keeps every token row; only these are consumed). */
int d4_gemm_bf16a_compact(const uint16_t* Ab, int lda, const uint16_t* Wb, int ldw, float* C, int ldc, uint16_t* Cb, const float* bias, const float* R, int ldr,
                          int M, int N, int K, int flags, float rms_eps, float* C2, uint16_t* C2b, int ldc2, int c2_S, int c2_lo, int c2_hi, int c2_last,
                          int config, void* stream);
int d4_gemm_bf16a_batched(const uint16_t* Ab, int lda, const uint16_t* Wb, int ldw, float* C, int ldc, uint16_t* Cb, const float* bias, const float* R, int ldr,
                          int M, int N, int K, int flags, float rms_eps, int batch, int64_t strideA, int64_t strideW, int64_t strideC, int config, void* stream);
int d4_cvt_rows_bf16(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int rows, int cols, void* stream);
/* fp32 GEMM on the bf16 matrix cores (csrc/gemm_x3.hip; the trunk's default `Linear` arithmetic): every fp32 operand is the exact sum of three
 * bf16 numbers and a product is accumulated from its six leading bf16 x bf16 terms in fp32 — fp32 accuracy (error against float64 no
 * larger than the f32-input MFMA kernels'), 6/16 of their matrix-pipe time.  d4_split_bf16x3 writes the three planes of W
 * (dst[p * plane_stride + i], p = 0..2; plane_stride % 8 == 0); d4_gemm_split takes A in fp32 and splits it on the fly.  `config` = -1: the
 * dispatcher's choice, 0..5 one tile configuration of the family (all give the same bits), 6 the persistent form the engine uses (csrc/gemm_x3sk.hip:
 * one workgroup per CU, a last partial round of at most half the workgroups as 128 x 64 half tiles — the same bits). */
int d4_split_bf16x3(const float* src, uint16_t* dst, int64_t n, int64_t plane_stride, void* stream);
int d4_gemm_split(const float* A, int lda, const uint16_t* W3, int64_t plane_stride, int ldw, float* C, int ldc, const float* bias,
                  const float* R, int ldr, int M, int N, int K, int flags, float rms_eps, int config, void* stream);
/* fp32-class GEMM on the fp16 matrix cores (csrc/gemm_h2.hip; round 5; the engine's OPT-IN `matmul_bf16 = 3` arithmetic for every nn.Linear of
 * D4:1983-2068, 2105-2116, 2143-2210): each operand row is scaled by an exact power of two into fp16's range and written as two fp16 planes
 * (hi, lo 2^11: a 23-bit image of the operand), a product is accumulated from its three leading fp16 x fp16 terms in fp32 and the scales are undone
 * by one exponent shift.  Against float64 its error on dot products of >= 16 terms is BELOW the f32-input MFMA kernels' (0.45x at K = 512) at
 * x1.2-2.0 their speed, but a single product carries a relative error of up to 2^-21 where fp32 has 2^-24: on operands spanning 2^120 inside a
 * row (every output one product) it reads 5.5e-7 of sum |a w| against the f32-input MFMA's 3.1e-7 — which is why it is NOT the default fp32 path
 * (profiles/r05_x3_products.txt).  d4_split_f16x2 writes the two planes of a weight matrix W [rows][ld] (dst[p * plane_stride + r * ld + c], p = 0..1;
 * ld % 8 == 0, plane_stride % 8 == 0) and inv_scale[r] = the power of two that undoes row r's scale; d4_row_scale_exp the scale exponents of an
 * activation matrix's rows (once for every GEMM that reads it); d4_gemm_split2 takes A in fp32 and splits it on the fly (a_exp null: the kernel
 * finds the exponents itself, a pass over A per column tile).  `config` = -1: the family's static choice, 0..6 one tile (all give the same bits). */
int d4_split_f16x2(const float* src, uint16_t* dst, int rows, int cols, int ld, int64_t plane_stride, float* inv_scale, void* stream);
int d4_row_scale_exp(const float* A, int64_t lda, int rows, int K, int32_t* exp_out, void* stream);
int d4_gemm_split2(const float* A, int lda, const uint16_t* W2, int64_t plane_stride, int ldw, const float* w_inv_scale, float* C, int ldc, const float* bias,
                   const float* R, int ldr, int M, int N, int K, int flags, float rms_eps, int config, const int32_t* a_exp, void* stream);
/* ---- trunk backward, first slice (SURVEY.md 8f-3 groundwork; not on the imagination path) ----
 * FeedForward block (dreamer4.py:2079-2116) on the reference parameter layout: y = proj_out(a * silu(g)) with [a | g] = proj_in(RMSNorm(x));
 * x / y / dy / dx [rows][dim], norm_w [dim], w_in [2*inner][dim], b_in [2*inner], w_out [dim][inner], b_out [dim].  The backward
 * recomputes the forward intermediates (nothing is saved between the two calls).  workspace: 256-byte aligned device memory.
 * Every `*_backward_saved` entry (FeedForward and the three attention blocks below) takes the SAME arguments as its `*_backward` but requires
 * `workspace` to be the very buffer the matching `*_forward` call ran in, untouched since: the forward intermediates it holds (normalised input,
 * concatenated weight images, projections) are used as they are and nothing is recomputed (one GEMM in four of a block's backward). */
size_t d4_ff_workspace_bytes(int rows, int dim, int inner);
int d4_ff_forward(const float* x, const float* norm_w, const float* w_in, const float* b_in, const float* w_out, const float* b_out,
                  int rows, int dim, int inner, float* y, float* workspace, size_t workspace_bytes, void* stream);
int d4_ff_backward(const float* x, const float* dy, const float* norm_w, const float* w_in, const float* b_in, const float* w_out,
                   int rows, int dim, int inner, float* dx, float* d_norm_w, float* d_w_in, float* d_b_in, float* d_w_out, float* d_b_out,
                   float* workspace, size_t workspace_bytes, void* stream);
int d4_ff_backward_saved(const float* x, const float* dy, const float* norm_w, const float* w_in, const float* b_in, const float* w_out,
                   int rows, int dim, int inner, float* dx, float* d_norm_w, float* d_w_in, float* d_b_in, float* d_w_out, float* d_b_out,
                   float* workspace, size_t workspace_bytes, void* stream);
/* Space attention block (Attention.forward, dreamer4.py:1968-2075, self attention within a frame): x / y [frames*tokens][dim],
 * residual_values [frames*tokens][heads*dim_head] or null (then w_mix / b_mix and their gradients are unused), wq / wk / wv
 * [heads*dim_head][dim], wo [dim][heads*dim_head], w_gates / w_mix [heads][dim], b_mix [heads], k_gamma [heads][dim_head];
 * tokens <= 64 per frame, dim_head 16 / 32 / 64; num_special trailing tokens are hidden from the ordinary queries (dreamer4.py:1769-1783). */
size_t d4_attn_workspace_bytes(int frames, int tokens, int dim, int heads, int dim_head);
int d4_space_attn_forward(const float* x, const float* residual_values, const float* norm_w, const float* wq, const float* wk, const float* wv,
                          const float* wo, const float* w_gates, const float* w_mix, const float* b_mix, const float* k_gamma,
                          int frames, int tokens, int dim, int heads, int dim_head, float softclamp, int num_special, int belief,
                          float* y, float* workspace, size_t workspace_bytes, void* stream);
int d4_space_attn_backward(const float* x, const float* residual_values, const float* dy, const float* norm_w, const float* wq, const float* wk,
                           const float* wv, const float* wo, const float* w_gates, const float* w_mix, const float* b_mix, const float* k_gamma,
                           int frames, int tokens, int dim, int heads, int dim_head, float softclamp, int num_special, int belief,
                           float* dx, float* d_residual_values, float* d_norm_w, float* d_wq, float* d_wk, float* d_wv, float* d_wo,
                           float* d_w_gates, float* d_w_mix, float* d_b_mix, float* d_k_gamma,
                           float* workspace, size_t workspace_bytes, void* stream);
int d4_space_attn_backward_saved(const float* x, const float* residual_values, const float* dy, const float* norm_w, const float* wq, const float* wk,
                           const float* wv, const float* wo, const float* w_gates, const float* w_mix, const float* b_mix, const float* k_gamma,
                           int frames, int tokens, int dim, int heads, int dim_head, float softclamp, int num_special, int belief,
                           float* dx, float* d_residual_values, float* d_norm_w, float* d_wq, float* d_wk, float* d_wv, float* d_wo,
                           float* d_w_gates, float* d_w_mix, float* d_b_mix, float* d_k_gamma,
                           float* workspace, size_t workspace_bytes, void* stream);
/* Time attention block (the same Attention with rotary positions and a causal mask along time, one problem per token column,
 * dreamer4.py:3176-3215 / 1626-1659): x / y [batch][frames][tokens][dim] row-major, frames <= 64 (no KV cache: the training form);
 * inv_freq [dim_head / 2] = time_rotary.inv_freq. */
size_t d4_time_attn_workspace_bytes(int batch, int frames, int tokens, int dim, int heads, int dim_head);
int d4_time_attn_forward(const float* x, const float* residual_values, const float* norm_w, const float* wq, const float* wk, const float* wv,
                         const float* wo, const float* w_gates, const float* w_mix, const float* b_mix, const float* k_gamma, const float* inv_freq,
                         int batch, int frames, int tokens, int dim, int heads, int dim_head, float softclamp, int belief,
                         float* y, float* workspace, size_t workspace_bytes, void* stream);
int d4_time_attn_backward(const float* x, const float* residual_values, const float* dy, const float* norm_w, const float* wq, const float* wk,
                          const float* wv, const float* wo, const float* w_gates, const float* w_mix, const float* b_mix, const float* k_gamma,
                          const float* inv_freq, int batch, int frames, int tokens, int dim, int heads, int dim_head, float softclamp, int belief,
                          float* dx, float* d_residual_values, float* d_norm_w, float* d_wq, float* d_wk, float* d_wv, float* d_wo,
                          float* d_w_gates, float* d_w_mix, float* d_b_mix, float* d_k_gamma,
                          float* workspace, size_t workspace_bytes, void* stream);
int d4_time_attn_backward_saved(const float* x, const float* residual_values, const float* dy, const float* norm_w, const float* wq, const float* wk,
                          const float* wv, const float* wo, const float* w_gates, const float* w_mix, const float* b_mix, const float* k_gamma,
                          const float* inv_freq, int batch, int frames, int tokens, int dim, int heads, int dim_head, float softclamp, int belief,
                          float* dx, float* d_residual_values, float* d_norm_w, float* d_wq, float* d_wk, float* d_wv, float* d_wo,
                          float* d_w_gates, float* d_w_mix, float* d_b_mix, float* d_k_gamma,
                          float* workspace, size_t workspace_bytes, void* stream);
/* Cross-attention block (Attention.forward with a context: the AttentionPool over the layer hiddens dreamer4.py:2143-2177, the final
 * special-token cross attention :3227-3234, the learned-query pools :2179-2210): q_tokens [groups*nq][dim], ctx [groups*nk][dim_ctx] with key
 * j of group g at row g*nk + j, or at row j*groups + g when ctx_item_major (the stack of hiddens); norm_ctx_w may be null (context not
 * normalised); wk / wv [heads*dim_head][dim_ctx]; nq, nk <= 64.  No value residual and no belief projection (as the reference with a context). */
size_t d4_cross_attn_workspace_bytes(int groups, int nq, int nk, int dim, int dim_ctx, int heads, int dim_head);
int d4_cross_attn_forward(const float* q_tokens, const float* ctx, const float* norm_w, const float* norm_ctx_w, const float* wq, const float* wk,
                          const float* wv, const float* wo, const float* w_gates, const float* k_gamma, int groups, int nq, int nk, int ctx_item_major,
                          int dim, int dim_ctx, int heads, int dim_head, float softclamp, float* y, float* workspace, size_t workspace_bytes, void* stream);
int d4_cross_attn_backward(const float* q_tokens, const float* ctx, const float* dy, const float* norm_w, const float* norm_ctx_w, const float* wq,
                           const float* wk, const float* wv, const float* wo, const float* w_gates, const float* k_gamma, int groups, int nq, int nk,
                           int ctx_item_major, int dim, int dim_ctx, int heads, int dim_head, float softclamp,
                           float* d_q_tokens, float* d_ctx, float* d_norm_w, float* d_norm_ctx_w, float* d_wq, float* d_wk, float* d_wv, float* d_wo,
                           float* d_w_gates, float* d_k_gamma, float* workspace, size_t workspace_bytes, void* stream);
int d4_cross_attn_backward_saved(const float* q_tokens, const float* ctx, const float* dy, const float* norm_w, const float* norm_ctx_w, const float* wq,
                           const float* wk, const float* wv, const float* wo, const float* w_gates, const float* k_gamma, int groups, int nq, int nk,
                           int ctx_item_major, int dim, int dim_ctx, int heads, int dim_head, float softclamp,
                           float* d_q_tokens, float* d_ctx, float* d_norm_w, float* d_norm_ctx_w, float* d_wq, float* d_wk, float* d_wv, float* d_wo,
                           float* d_w_gates, float* d_k_gamma, float* workspace, size_t workspace_bytes, void* stream);
int d4_rmsnorm(const float* x, int ldx, const float* gamma, float* y, int ldy, int rows, int dim,
               float eps, void* stream);
/* backward of nn.RMSNorm (autograd of y = x / rms(x) * gamma): dx [rows][dim], d_gamma [dim]; scratch = rows * dim floats. */
int d4_rmsnorm_backward(const float* x, const float* dy, const float* gamma, float* dx, float* d_gamma, float* scratch, int rows, int dim, float eps,
                        void* stream);
int d4_hl_gauss_scalar(const float* logits, int ld, const float* centers, float* out, int rows,
                       int bins, void* stream);
/* MultiCategorical.sample + log_prob of the sample (dreamer4.py:485-497, 1374-1376, 1422-1423), stateless: Gumbel-max per action type from
 * injected uniforms [rows][ld_u] (the layout of the logits), log-softmax gather.  action_sizes: device int32 [na].  Indices are bit-exact
 * against the reference whenever the top-2 (logit / T + Gumbel) margin exceeds fp32 rounding. */
int d4_categorical_sample_logp(const float* logits, int ld, const float* uniform, int ld_u, const int32_t* action_sizes, int rows, int na,
                               float temperature, int64_t* actions, float* log_probs, void* stream);
/* The value branch's loss, stateless (dreamer4.py:6254-6295; HLGaussLoss / SymExpTwoHot targets): HL-Gauss (two_hot = 0: support = [bins + 1] bin
 * edges, sigma in value units) or two-hot (two_hot = 1: support = [bins] bin values) targets of `targets`, cross entropy against `logits`, mean
 * over the rows with mask != 0 (mask null: all rows).  loss[0] and dlogits [rows][ld] = d loss / d logits in one call.
 * scratch >= 2 * rows + 64 floats. */
int d4_hl_gauss_ce(const float* logits, int ld, const float* targets, const float* mask, const float* support, int rows, int bins, float vmin,
                   float vmax, float sigma, float eps, int two_hot, float* loss, float* dlogits, float* scratch, void* stream);
/* The policy branch's loss for discrete actions, stateless (dreamer4.py:6077-6242: log-prob of the stored actions under the MultiCategorical
 * logits, PPO clipped surrogate (objective 0, dreamer4.py:6204-6212) or SPO (1, dreamer4.py:6188-6198) against the behaviour log-probs
 * [rows][na], entropy bonus, masked mean): loss[0] and dlogits [rows][ld] in one call — the fused kernel d4_learn runs, the advantages taken as
 * given (normalise them before, dreamer4.py:5985-5999).  action_sizes: device int32 [na], `total` their sum.  scratch >= 5 * rows + 64 floats. */
int d4_ppo_policy_loss(const float* logits, int ld, const int64_t* actions, const float* old_log_probs, const float* advantages, const float* mask,
                       const int32_t* action_sizes, int rows, int na, int total, int objective, float eps_clip, float entropy_weight, float* loss,
                       float* dlogits, float* scratch, void* stream);
int d4_gae(const float* rewards, const float* values, const int64_t* lens, const uint8_t* is_truncated,
           const uint8_t* terminals, float gamma, float lam, int batch, int time, float* returns,
           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* D4HIP_H */
