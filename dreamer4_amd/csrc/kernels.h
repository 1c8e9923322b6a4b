// Internal launcher interface shared by the engine (engine.hip) and the C-ABI test entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace d4 {

// ------------------------------------------------------------------------------------ GEMM
enum : int {
    GEMM_RMS_ROWSCALE = 1,   // C = (1/rms(A row)) * (A W^T): RMSNorm folded (gamma pre-multiplied into W)
    GEMM_SILU = 2,           // C = silu(.)
    GEMM_SWIGLU = 4,         // packed (value, gate) column pairs -> C[:, N/2] = value * silu(gate)
    GEMM_TRANS_A = 8,        // A(m, k) read from A[k * lda + m]
    GEMM_TRANS_B = 16,       // W(n, k) read from W[k * ldw + n]
    GEMM_ACCUMULATE = 32,    // C += result
};

struct GemmArgs {
    const float* A; int lda;
    const float* W; int ldw;
    float* C; int ldc;
    const float* bias;          // [N] or null
    const float* R; int ldr;    // residual added after activation, or null
    int M, N, K;
    int flags;
    float rms_eps;
    double algo_flops = 0;
    // optional second, row-compacted copy of the output: token rows s = gm % c2_S with c2_lo <= s < c2_hi or s == c2_S - 1
    // land in C2 at row (gm / c2_S) * (c2_hi - c2_lo + 1) + rank  (the rows the final pool / latent head need)
    float* C2 = nullptr; int ldc2 = 0, c2_S = 0, c2_lo = 0, c2_hi = 0, c2_last = 1;    // c2_last = 0: the frame has no trailing agent row
    // strided batch (blockIdx.y): A += b * strideA, W += b * strideW, C/R += b * strideC   (elements)
    int batch = 1; int64_t strideA = 0, strideW = 0, strideC = 0;      // algorithmic flops of this launch when padding makes 2MNK an over-count (profiling only)
    const uint16_t* Wb = nullptr;      // bf16 image of W (same [N][ldw] layout): when set the call runs on the bf16 MFMA kernel (gemm_bf16.hip)
    const uint16_t* Ab = nullptr;      // bf16 image of A (same [M][lda] layout, written by the producer): with Wb set the call runs on the
                                       // bf16-activation kernel (gemm_bf16a.hip) and never touches the fp32 A
    uint16_t* Cb = nullptr;            // optional bf16 copy of the output ([M][ldc]; SiLU-GLU: [M][N/2] at ldc), for the next GEMM's Ab
    uint16_t* C2b = nullptr;           // ... and of the row-compacted second output ([rows][ldc2])
    int group_m = 0;                   // > 0 (gemm_bf16a.hip): tiles are walked in groups of `group_m` row panels, column-major inside a group, so the
                                       // ~32 tiles an XCD runs at a time share few operand panels (its 4 MB L2 then holds them); 0: row-major
    int64_t wplane = 0;                // > 0: Wb holds THREE bf16 planes (W = W1 + W2 + W3, plane stride in elements) and the call runs as
                                       // an fp32 GEMM on the bf16 matrix cores (split operands, six products: gemm_x3.hip)
    const float* wscale = nullptr;     // set (with wplane > 0): Wb holds TWO fp16 planes of W scaled row by row (hi, lo 2^11) and wscale[n] = the exact
                                       // power of two that undoes row n's scale -> fp32 GEMM on the fp16 matrix cores, three products (gemm_h2.hip)
    int64_t strideWs = 0;              // ... batch stride of wscale (elements)
    const int* aexp = nullptr;         // optional (gemm_h2.hip): scale exponent of every A row ([batch][M]) from a producer that knows it; null: the kernel finds it
};

int gemm(const GemmArgs& p, hipStream_t stream);
// bf16 MFMA path (gemm_bf16.hip): A fp32 rounded to bf16 on the way into LDS, W pre-converted, fp32 accumulate / epilogue
bool gemm_bf16_applicable(const GemmArgs& p);
int gemm_bf16(const GemmArgs& p, hipStream_t stream);
int gemm_bf16_force_config(int id);                          // test / microbenchmark hook; returns the number of configurations
void gemm_bf16a_force_config(int id);                        // ... of the bf16-activation kernel (-1: by rule)
int cvt_f32_to_bf16(const float* src, uint16_t* dst, int64_t n, hipStream_t s);
int cvt_rows_bf16(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int rows, int cols, hipStream_t s);      // strided rows -> bf16
int gemm_bf16a(const GemmArgs& p, hipStream_t stream);        // bf16 activations + bf16 weights: tile by rule, optional event pair (gemm_bf16.hip)
// bf16 GEMM with bf16 activations, LDS-DMA ring (gemm_bf16a.hip): p.Ab / p.Wb in, fp32 C (+ bf16 copy p.Cb) out
bool gemm_bf16a_applicable(const GemmArgs& p);
bool gemm_bf16a_config_valid(int c, const GemmArgs& p);
int gemm_bf16a_configs();
int gemm_bf16a_rule(const GemmArgs& p);
int gemm_bf16a_launch(int c, const GemmArgs& p, hipStream_t stream, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);
int gemm_bf16p_launch(const GemmArgs& p, hipStream_t stream, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);      // gemm_bf16p.hip: the phased 256 x 256 form (configuration 6 of gemm_bf16a_launch)
bool gemm_bf16a_pair_applicable(const GemmArgs& a, const GemmArgs& b);      // two RMS-folded products of equal K in one grid
int gemm_bf16a_pair_launch(const GemmArgs& a, const GemmArgs& b, hipStream_t stream, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);
int gemm_bf16a_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t stream);      // ... with gemm_bf16.hip's optional event pair
void gemm_bf16_profile_enable(int stride);
int gemm_bf16_profile_read(double* ms, double* flops, int64_t* count);
// third fp32 family (gemm_x3.hip): fp32 operands split into three bf16 numbers, six bf16 MFMA products, fp32 accumulate (fp32 accuracy)
bool gemm_x3_applicable(const GemmArgs& p);
bool gemm_x3_config_valid(int c, const GemmArgs& p);
int gemm_x3_configs();
const char* gemm_x3_config_name(int c);
void gemm_x3_config_tile(int c, int* bm, int* bn);
int gemm_x3_heuristic(const GemmArgs& p);
int gemm_x3_launch(int c, const GemmArgs& p, hipStream_t stream, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);
int split_bf16x3(const float* src, uint16_t* dst, int64_t n, int64_t plane, hipStream_t s);
// fourth fp32 family (gemm_h2.hip): fp32 operands as two fp16 planes under exact power-of-two row scales, three fp16 MFMA products, fp32 accumulate
bool gemm_h2_applicable(const GemmArgs& p);
bool gemm_h2_config_valid(int c, const GemmArgs& p);
int gemm_h2_configs();
const char* gemm_h2_config_name(int c);
void gemm_h2_config_tile(int c, int* bm, int* bn);
int gemm_h2_heuristic(const GemmArgs& p);
bool gemm_h2_takes(const GemmArgs& p);                     // the dispatcher's rule (gemm.hip): the calls of an fp16x2 engine that run on this family
int gemm_h2_launch(int c, const GemmArgs& p, hipStream_t stream, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);
int split_f16x2_rows(const float* src, uint16_t* dst, int rows, int cols, int ld, int64_t plane, float* inv_scale, hipStream_t s);
int row_scale_exp(const float* A, int64_t lda, int rows, int K, int* out, hipStream_t s);      // GemmArgs::aexp of an activation matrix
// persistent form of the 128 x 128 split-operand kernel (gemm_x3sk.hip): one workgroup per CU, the tiles of the last partial round as half tiles
bool gemm_x3sk_applicable(const GemmArgs& p);
bool gemm_x3sk_rule(const GemmArgs& p);                    // the calls that take it (a rule on the shape and the CU count)
const char* gemm_x3sk_name();
int gemm_x3sk_launch(const GemmArgs& p, hipStream_t stream, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);
// second fp32 family (gemm2.hip): 16x16x4 MFMA fed by an LDS-DMA ring; non-transposed operands, K % 32 == 0
bool gemm2_applicable(const GemmArgs& p);
bool gemm2_config_valid(int c, const GemmArgs& p);
int gemm2_configs();
const char* gemm2_config_name(int c);
void gemm2_config_tile(int c, int* bm, int* bn);
int gemm2_launch(int c, const GemmArgs& p, hipStream_t stream, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);
bool gemm2_ksplit_applicable(const GemmArgs& p);
bool gemm2_ksplit_rule(const GemmArgs& p);               // few rows x long K: the contraction cut four ways inside the workgroup (gemm2_ksplit_kernel)
int gemm2_ksplit_launch(const GemmArgs& p, hipStream_t stream, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);
bool gemm2_pair_config_ok(int c);
bool gemm2_pair_applicable(const GemmArgs& a, const GemmArgs& b);     // two independent GEMMs in one grid (gemm2_pair_kernel)
int gemm2_pair_launch(int c, const GemmArgs& a, const GemmArgs& b, hipStream_t stream, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);
int gemm_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t stream);   // gemm.hip: one launch when the pair form applies, else two
bool gemm_skinny_applicable(const GemmArgs& p);            // M <= 16 rows: VALU kernel that streams W once (gemm_skinny.hip)
int gemm_skinny(const GemmArgs& p, hipStream_t stream);
bool gemm_skinny_pair_applicable(const GemmArgs& a, const GemmArgs& b);   // two few-row GEMMs of equal K in one launch
int gemm_skinny_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t stream);
int gemm_profile_enable(int on);
bool gemm_profile_active();
int gemm_profile_read(double* ms, double* flops, int64_t* count, int nclass);
int gemm_profile_classes();
extern int g_time_attn_fused_append;     // attn.hip: 1 (default) the cached decode's KV append rides in its time attention launch; 0 two launches (test hook)
extern int g_pool_wide_keys;             // engine.hip: 1 (default) bf16 engine projects a hidden once for every later pool's keys; 0 one key projection per pool (test hook)
extern int g_attn_out_cols;              // frame_fused.hip: 1 (default) attention inside the column-split output projection at <= 4 frames; 0 two launches (test hook)
int gemm_force_config(int id);                             // test hook; returns the number of configurations
const char* gemm_profile_class_name(int c);

// ------------------------------------------------------------------------------------ small attention
// One wave per (group, head); head dim 64 (lane = feature).  Covers the space attention of the
// trunk, the learned-query pools, the per-token layer pool and the agent-token cross attention.
struct SmallAttnArgs {
    const float* q;      int64_t q_group_stride, q_item_stride;      // + h * 64 + lane
    const float* k;      int64_t k_group_stride, k_item_stride;
    const float* v;      int64_t v_group_stride, v_item_stride;
    const float* gate;   int64_t g_group_stride, g_item_stride;      // pre-sigmoid, + h ; may be null
    const float* k_gamma;                                             // [H][64] K-head-RMSNorm gamma
    const float* vres;   int64_t r_group_stride, r_item_stride;      // value residual (+ h*64 + lane) or null
    const float* mix;    int64_t m_group_stride, m_item_stride;      // pre-sigmoid mix logits (+ h) (with vres)
    float* out;          int64_t o_group_stride, o_item_stride;
    int groups, heads, nq, nk;
    float softclamp;      // <= 0 : none
    int mask_special;     // number of trailing "special" keys ordinary queries may not see
    int belief;           // subtract the component of out along l2norm(v_i) (requires nq == nk, self attention)
    // space_attn only: restrict the QUERIES to tokens [q_lo, q_hi) plus the last token; outputs are written at item
    // rank (i - q_lo, or q_hi - q_lo for the last token).  q_hi == 0 -> all queries, natural order.
    int q_lo = 0, q_hi = 0, q_last = 1;   // q_last = 0: do not add the last token to the restricted query set
    int dh = 64;                          // head dim (16 / 32 / 64): lanes >= dh of the wavefront idle; rows are packed h * dh + lane
    uint16_t* out_b = nullptr;            // optional bf16 copy of the output (same strides as `out`): the next GEMM's bf16 activation image (bf16 engine)
};
int small_attn(const SmallAttnArgs& p, hipStream_t stream);

// AttentionPool core, value-side restructured: scores over the L hiddens from projected keys, then the softmax-
// weighted (and gated) sum of the NORMALISED hiddens per head, u[m][h][:] = sigmoid(gate) * sum_l p[l][h] * h_l[m] / rms(h_l[m]);
// the value projection is then ONE [D -> 64] GEMM per head on u instead of L projections (linear in the hiddens).
struct PoolMixArgs {
    const float* q; int ldq;           // [M][ldq]: projected queries (heads*64)
    const float* x; int ldx;           // [M][D] pool input tokens (the head gates are sigmoid(RMSNorm(x) . gate_w[h]), computed here:
    const float* gate_w;               //  [heads][D], norm gamma folded — 4 extra columns would cost the query GEMM a whole tile column)
    const float* k; int ldk;           // [L*M][ldk]: projected keys, row l*M + m
    const float* hid; int D;           // [L*M][D] hiddens
    const float* k_gamma;              // [heads][64]
    float* u;                          // [M][heads][D]; may be null when u_b is set
    int M, L, heads;
    float eps;
    uint16_t* u_b = nullptr;           // optional bf16 image of u (bf16 engine: the per-head value GEMM reads this one)
    const uint16_t* k_b = nullptr;     // optional bf16 image of the projected keys (same [L*M][ldk] layout; bf16 engine): read INSTEAD of `k`
    const uint16_t* q_b = nullptr;     // ... and of the projected queries ([M][ldq]; only with k_b): read instead of `q`
    const uint16_t* hid_b = nullptr;   // optional bf16 image of the hiddens (bf16 engine): read instead of `hid` by the one-wave-per-row form — the
                                       // values the pool's key GEMM consumed, at half the bytes of the kernel's dominant stream
};
int pool_mix(const PoolMixArgs& p, hipStream_t stream);

// ------------------------------------------------------------------------------------ per-frame fused block tails (frame_fused.hip)
// W [N][K] -> Wt [N / 16][K / 4][16][4]: the weight image the per-frame kernels stream (a 16-row tile is one contiguous run)
int tile16_weights(const float* W, int ldw, float* Wt, int N, int K, hipStream_t s);
bool frame_fused_frames_ok(int frames);
int frame_fused_mode();                 // 0 off, 1 on (default), 2 tails only
int frame_fused_set(int mode);          // test hook; returns the previous mode
// within-frame attention -> output projection + residual (+ row-compacted copy), one workgroup per frame
bool frame_attn_out_applicable(const SmallAttnArgs& sa, int D);
int frame_attn_out(const SmallAttnArgs& sa, const float* wo_t, int D, const float* resid, int ldr, float* out, int ldo, float* c2, int ldc2, int c2_lo,
                   int c2_hi, int c2_last, hipStream_t s);
// few frames (launch-bound decode): the same pair as ONE column-split launch that recomputes the frame's attention per 16 output columns
bool attn_out_cols_applicable(const SmallAttnArgs& sa, int D);
int attn_out_cols(const SmallAttnArgs& sa, const float* W, int ldw, int D, const float* resid, int ldr, float* out, int ldo, float* c2, int ldc2, int c2_lo,
                  int c2_hi, int c2_last, hipStream_t s);
// AttentionPool tail: per-head value projection of the mixes -> output projection + residual (+ row-compacted copy)
bool frame_pool_tail_applicable(int frames, int S, int D, int pool_heads);
int frame_pool(const PoolMixArgs& pm, const float* wv_t, const float* wo_t, int frames, int S, const float* resid, int ldr, float* out, int ldo, float* c2,
               int ldc2, int c2_lo, int c2_hi, int c2_last, hipStream_t s);      // mix + tail in one kernel
int frame_pool_tail(const float* u, const float* wv_t, const float* wo_t, int frames, int S, int D, int pool_heads, const float* resid, int ldr, float* out,
                    int ldo, float* c2, int ldc2, int c2_lo, int c2_hi, int c2_last, hipStream_t s);

// ------------------------------------------------------------------------------------ time attention
// Causal attention along time for every token column (b, s) with a preallocated KV cache.
//   proj rows are ordered (b, tq, s); columns: q @ 0, k @ hd, v @ 2hd, gate @ 3hd, mix @ 3hd + h.
//   cache layout: [2][B*S][H][Tcap][64]
struct TimeAttnArgs {
    const float* proj; int ldp;
    const float* vres; int ldv;          // value residual rows (same row order), + h*64 + lane
    const float* k_gamma;                // [H][64]
    const float* inv_freq;               // [32]
    float* cache;                        // this layer's cache
    float* out; int ldo;                 // rows (b, tq, s), cols h*64 + lane
    int B, S, H, Tq, t0, Tcap;
    int cache_batch;                     // batch capacity the cache was laid out for
    int cache_S = 0;                     // tokens per frame the cache was laid out for (0: same as S)
    const int* t0_dev = nullptr;         // when set, the frame offset is read from device memory (hipGraph replay); t0 must then lie in the same
                                         // time_history_bucket as the value read (the launcher picks its kernel by the bucket of t0)
    float softclamp;
    int dh = 64;                         // head dim (16 / 32 / 64); cache rows are dh wide
    uint16_t* out_b = nullptr;           // optional bf16 copy of the output rows (same ldo)
};
// History-length class the cached-decode launcher picks its kernel by: 0: <= 8 keys (t0 < 8), 1: <= 16 keys, 2: general.
inline int time_history_bucket(int t0) { return t0 < 8 ? 0 : (t0 < 16 ? 1 : 2); }
int time_kv_append(const TimeAttnArgs& p, hipStream_t stream);   // normalise/rotate/mix new K,V -> cache[t0 .. t0+Tq)
int time_attn(const TimeAttnArgs& p, hipStream_t stream);        // attend over cache[0 .. t0+i], belief + gates
int time_attn_append(const TimeAttnArgs& p, hipStream_t stream); // both; ONE launch for the cached decode of one frame (head dim 64, aligned rows)

// ------------------------------------------------------------------------------------ elementwise / glue
int fold_rows(const float* W, const float* gamma, float* out, int rows, int K, int ld_out, hipStream_t s);
int copy_rows(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t s);
int fill_f32(float* dst, float v, int64_t n, hipStream_t s);
int swiglu_pack_rows(const float* W, const float* bias, const float* gamma, float* Wp, float* bp,
                     int inner, int inner_pad, int K, hipStream_t s);
int pad_cols(const float* W, float* out, int rows, int cols, int cols_pad, hipStream_t s);
// learn.hip: out[c] = sum_r x[r][c] (fixed order); RMSNorm backward (tg = dxhat * x * rstd, column-summed by the caller -> dgamma)
int colsum(const float* x, int ld, int rows, int cols, float* out, hipStream_t s, float* scratch = nullptr, size_t scratch_floats = 0);
struct ColsumBatch {                   // up to four independent column sums as one launch (learn.hip: colsum_batch)
    enum { MAX = 4 };
    const float* x[MAX]; int ld[MAX], rows[MAX], cols[MAX]; float* out[MAX]; int first[MAX]; int n = 0;
    void add(const float* x_, int ld_, int rows_, int cols_, float* out_) { x[n] = x_; ld[n] = ld_; rows[n] = rows_; cols[n] = cols_; out[n] = out_; ++n; }
};
int colsum_batch(ColsumBatch& b, hipStream_t s);
int rmsnorm_bwd(const float* x, const float* dxhat, const float* gamma, float* tg, float* dx, int rows, int d, float eps, hipStream_t s);
int rmsnorm_rows(const float* x, int ldx, const float* gamma, float* y, int ldy, int rows, int D, float eps, hipStream_t s);
int layernorm_rows(const float* x, int ldx, const float* g, const float* b, float* y, int ldy, int rows, int D, float eps, int silu, hipStream_t s);

struct AssembleArgs {
    float* tokens;                 // [B*Tq][S][D]
    const float* space;            // [B*Tq][ns][D]
    const float* signal_embed;     // [max_steps][D/2]
    const float* step_embed;       // [log2 max_steps][D/2]
    const float* registers;        // [nr][D]
    const float* agent_embed;      // [D]
    const float* task_embed;       // [num_tasks][D] or null
    const float* action_embed;     // [total_actions][D]
    const float* action_learned;   // [D]
    const int32_t* signal_levels;  // [B*Tq], or null: every frame is at signal_uniform
    int signal_uniform = 0;
    const int64_t* prev_actions;   // [B*Tq][na] (-1 in slot 0 => zero token) or null (=> zero token)
    const float* prev_cont;        // [B*Tq][nc] continuous actions (NaN in slot 0 => zero token when na == 0) or null
    const float* cont_embed;       // [nc][D]  action_embedder.continuous_action_embed.weight
    const int64_t* tasks;          // [B] or null
    const int32_t* action_offsets; // [na] (device)
    int B, Tq, S, D, ns, nr, na, nc, step_log2;
    float* compact;                // optional [B*Tq][ns (+ 1)][D]: spatial (+ agent) rows only
    int has_agent;                 // 0: the frame is packed without its trailing agent token (S = tokens actually present)
};
int assemble_tokens(const AssembleArgs& p, hipStream_t s);

// gather the spatial-token rows of every frame, RMSNorm(gamma0) then (statistics only) second RMS for the
// LQAP context norm folded downstream is NOT possible -> both norms applied here.
int gather_space_double_norm(const float* tokens, float* out, const float* g0, const float* g1,
                             int frames, int S, int first, int D, int ns, float eps, hipStream_t s);

int euler_step(float* x, int ldx, const float* pred, int ldp, int B, int n_el, float one_minus_t, float dt, hipStream_t s);
int silu_rows(const float* z, float* y, int64_t n, hipStream_t s);
// y[m][n] = act(sum_s part[s][m][n] + bias[n])   (split-K combine, fixed order)
int splitk_reduce(const float* part, int S, int M, int N, const float* bias, int silu, float* y, int ldy, hipStream_t s);
// weight-gradient GEMM C[M][N] = A^T B with A [K][lda], B [K][ldb] row-major (contraction over rows), operands read straight into the MFMA
// layout (gemm_tn.hip); `part` = scratch for the k-slices' partial products (part_floats floats) or null; forced_* = 0: the shape rule
// input gradient dX[M][N] = dY[M][K] W[K][N] (W: a Linear's weight [out = K][in = N]) through a transposed weight image `wt` (N * K floats scratch)
// on the LDS-DMA family; falls back to the transposed-operand form of gemm() when wt is null, K % 32 != 0 or M < 256
int gemm_dx(const float* dY, int ldy, const float* W, int ldw, float* dX, int ldx, int M, int N, int K, float* wt, hipStream_t s);
bool gemm_tn_applicable(const float* A, int lda, const float* B, int ldb, const float* C, int ldc, int M, int N, int K);
int gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, float* part, size_t part_floats, hipStream_t s,
            int forced_tn = 0, int forced_slices = 0);
int prep_eval_inputs(int32_t* sig, int64_t* pact, const int64_t* actions_hist, int B, int Tq, int na, int frame_base,
                     int hist_stride, int sig_val, int ctx_sig, hipStream_t s, float* pcont = nullptr, const float* cont_hist = nullptr, int nc = 0);
int fill_sig(int32_t* sig, int n, int value, hipStream_t s);
int set_frame_state(int* state, int t0, hipStream_t s);
int cache_transfer(float* cache, float* ext, int Lt, int cache_batch, int B, int S, int H, int Tcap, int frames, int to_ext, int dh, hipStream_t s);

// latents input for a parallel (multi-frame) evaluation: context frames lerp(history, ctx_noise, w), last frame = x
int build_latent_input(float* out, const float* hist, const float* ctx_noise, const float* x,
                       int B, int Tq, int n_el, int hist_t_stride, float w, hipStream_t s);

// ------------------------------------------------------------------------------------ heads
int hl_gauss_scalar(const float* logits, int ld, const float* centers, float* out, int out_stride, int rows, int bins, hipStream_t s);
int mean_tokens(const float* x, float* out, int B, int n, int d, hipStream_t s);
struct SampleArgs {
    const float* logits; int ld;    // [B][A]
    const float* gumbel_u; int ld_u; // [B][A] uniform
    const float* term_logit;        // [B] or null
    const float* bern_u;            // [B] or null
    int64_t* actions;  int act_stride;      // out [B][na] at stride
    float* log_probs;  int lp_stride;       // out [B][na]
    uint8_t* terminals;                     // in/out [B] or null
    int64_t* lens;                          // in/out [B] or null
    const int32_t* action_sizes;            // [na] device
    int B, na, frame_index;
    float temperature;
    // continuous actions (Beta head): raw parameters [B][nc][2] at stride ld_c, injected gamma noise [B][nc][2][6][2]
    const float* cont_params = nullptr; int ld_c = 0;
    const float* beta_noise = nullptr;
    float* actions_cont = nullptr; int actc_stride = 0;     // out [B][nc]
    float* log_probs_cont = nullptr; int lpc_stride = 0;    // out [B][nc]
    int nc = 0;
    float cont_temperature = 1.f;
    int beta_param = 0;                                     // d4_config.continuous_beta_param (beta.h)
};
int sample_actions_terminals(const SampleArgs& p, hipStream_t s);
// tokenizer decoder glue: video <-> patch rows ('b c t (h p1) (w p2) <-> (b t h w) (p1 p2 c)', D4:3556, 3896), token packing, coordinate grid
int video_to_patches(const float* video, float* patches, int B, int C, int T, int nh, int nw, int ps, hipStream_t s);
int patches_to_video(const float* patches, float* video, int B, int C, int T, int nh, int nw, int ps, hipStream_t s);
int decoder_pack_tokens(float* tokens, float* compact, const float* pos, const float* img, const float* lat, int frames, int P, int n_lat, int n_total, int D, hipStream_t s);
int coord_grid(float* out, int nh, int nw, int ld, hipStream_t s);
int encoder_pack_tokens(float* tokens, float* compact, const float* img, const float* latent_tokens, int frames, int P, int n, int D, hipStream_t s);
int tanh_rows(const float* x, float* y, int64_t n, hipStream_t s);
int cunembed_gather(const float* U, float* w, int nc, int mtp, int d, hipStream_t s);
int cunembed_scatter_grad(const float* g, float* dU, int nc, int mtp, int d, hipStream_t s);

}  // namespace d4
