// Engine state of the MI355X imagination runtime (one per process / GPU).
#pragma once
#include "../../include/d4hip.h"
#include "kernels.h"
#include <string>
#include <unordered_map>
#include <vector>

namespace d4 {
constexpr size_t L_DWPART_FLOATS = (size_t)8 << 20;   // learner: scratch for the k-slice partial products of the weight-gradient GEMMs (32 MB)

struct Bound { const float* p; float* g; int64_t n; };

struct AttnW {            // a reference `Attention` module (D4:1887)
    const float *norm = nullptr, *norm_ctx = nullptr, *to_q = nullptr, *to_k = nullptr, *to_v = nullptr,
                *to_out = nullptr, *to_gates = nullptr, *k_gamma = nullptr, *mix_w = nullptr, *mix_b = nullptr;
};
struct FfW { const float *norm, *in_w, *in_b, *out_w, *out_b; };

struct FfPrep { float *w1, *b1, *w2; };

// Normed MLP head (x_mlps_pytorch.normed_mlp.create_mlp, D4:4950, 5083, 5095).  The package is absent from the image, so the layer
// recipe is a descriptor honoured by every layer of the stack (oracle shim, restatement, host module tree, these kernels):
//   D4_MLP_PRE_RMS      layer = RMSNorm(d_in) -> Linear(d_in, d_out, bias) -> SiLU ; no activation on the last layer
//                       keys   layers.{i}.0.weight (norm)  layers.{i}.1.{weight,bias} (linear)
//   D4_MLP_POST_LAYER   layer = Linear(d_in, d_out, bias) -> LayerNorm(d_out) -> SiLU ; the last layer is a bare Linear
//                       keys   layers.{i}.0.{weight,bias} (linear)  layers.{i}.1.{weight,bias} (norm) ; last: layers.{i}.{weight,bias}
struct Mlp {
    int nl = 0;
    int recipe = 0;
    int dims[10];
    const float *g[9], *nb[9], *w[9], *b[9];      // g / nb: norm weight / bias (bias: LayerNorm only)
    float *dg[9], *dnb[9], *dw[9], *db[9];        // gradient buffers (may be null)
    bool post_norm(int i) const { return recipe == 1 && i < nl - 1; }
    // learner save area, per layer: x [R][din] | xhat [R][din] | z [R][ldz]; every block 256-byte aligned
    static size_t al(size_t n) { return (n + 63) / 64 * 64; }
    int ldz(int i) const { return (dims[i + 1] + 3) / 4 * 4; }
    size_t save_floats(size_t R) const {
        size_t t = 0;
        for (int i = 0; i < nl; ++i) t += 2 * al(R * dims[i]) + al(R * (size_t)ldz(i));
        return t;
    }
    void save_ptrs(float* base, size_t R, int i, float** x, float** xhat, float** z) const {
        float* p = base;
        for (int j = 0; j < i; ++j) p += 2 * al(R * dims[j]) + al(R * (size_t)ldz(j));
        *x = p; *xhat = p + al(R * dims[i]); *z = *xhat + al(R * dims[i]);
    }
};

}  // namespace d4

struct d4_engine {
    d4_config c;
    std::unordered_map<std::string, d4::Bound> bound;

    // derived sizes
    int S, hd, hp, php, D, Nproj, Nproj0, inner, inner_pad, Lt, A, na, nc, ldpq, ldcq, nslab;
    std::vector<int> is_time, time_index;
    int maxB, maxTq, Tcap, Mmax, Fr;

    // workspace
    char* ws = nullptr;
    size_t ws_bytes = 0, ws_need = 0;
    bool prepared = false;
    int cache_frames = 0;
    // captured decode frames: key = (batch, num_steps, step_log2, has_tasks, history bucket) -> executable graph of the K+1 evaluations.
    // The history bucket (d4::time_history_bucket of the frame offset) is part of the key because the time-attention launcher picks its
    // kernel by it: a replayed frame runs exactly the kernels the same frame runs when enqueued eagerly.
    struct FrameGraph { int B, K, sl, tasks, bucket; hipGraphExec_t exec; };
    std::vector<FrameGraph> graphs;
    int graph_max_rows = 4096;
    bool warm = false;
    hipStream_t capture_stream = nullptr;             // use graphs when batch * tokens_per_frame <= this (launch-bound regime)

    // ---- decoder mode (video tokenizer's decoder, D4:3490-3682): tokens per frame = [P patches | n latent tokens]
    bool decoder = false, encoder = false;
    const float *ptt_w = nullptr, *ptt_b = nullptr, *ptt_ln = nullptr, *latent_tokens = nullptr, *e2l_w = nullptr;    // encoder mode
    float* enc_out = nullptr;
    int P = 0, dim_patch = 0, nph = 0, npw = 0;
    int keep_lo = 1, keep_hi = 1;          // token rows of a frame the final stage needs (dynamics: the spatial tokens; decoder: the patches)
    const float *ld_w = nullptr, *time_embed = nullptr, *npt_w = nullptr, *npt_b = nullptr, *npt_ln = nullptr, *t2p_w = nullptr, *t2p_b = nullptr, *final_norm = nullptr;
    d4::Mlp posmlp;
    float *pos_emb = nullptr, *t2p_wf = nullptr, *dec_in = nullptr, *img_tok = nullptr, *lat_tok = nullptr, *dec_out = nullptr, *posA = nullptr, *posB = nullptr, *zerosD = nullptr;

    // ---- bf16 compute (opt-in): bf16 mirrors of every weight the trunk GEMMs read, carved from one arena of the workspace
    bool bf16 = false;                     // the trunk's GEMMs read mirrors of their weights (bf16 mode, or the three planes of the split-operand fp32 mode)
    bool split = false;                    // mirrors are three bf16 planes (plane stride = bf16_cap): fp32 GEMMs on the bf16 matrix cores (gemm_x3.hip)
    bool h2 = false;                       // opt-in `fp32_fp16x2` mode: mirrors are two fp16 planes + an exact power-of-two scale per weight row (gemm_h2.hip:
                                           // 23-bit operand images, three products, fp32 accumulate); activations stay fp32, their row exponents in `aexp`
    bool fp32_planes() const { return split || h2; }      // an fp32-class engine whose GEMMs read plane mirrors of their weights
    uint16_t* bf16_arena = nullptr; size_t bf16_cap = 0, bf16_used = 0;
    float* wscale_arena = nullptr; size_t wscale_cap = 0, wscale_used = 0;      // h2 mode: the inverse row scales, one float per mirrored weight row
    // h2 mode: scale exponents of activation rows (GemmArgs::aexp).  The residual-stream slabs keep theirs for the whole evaluation (every later attention
    // pool re-reads every earlier slab): aexp_slab [nslab][M] + one validity flag per slab, cleared when an evaluation starts; any other A gets its
    // exponents into aexp_tmp right before the GEMM that reads it
    int* aexp_slab = nullptr; int* aexp_tmp = nullptr; size_t aexp_tmp_rows = 0, aexp_M = 0;      // aexp_M: token rows per slab in the running evaluation
    std::vector<char> aexp_valid;
    struct Mirror { const float* src; size_t n; uint16_t* dst; int ld; float* scales; };
    std::vector<Mirror> mirrors;
    // bf16 mode (not split): bf16 IMAGES of the activation buffers the trunk GEMMs read (same element offsets / leading dimensions as the fp32
    // buffer).  Producers write them (GEMM epilogue `Cb`, attention / pool-mix `out_b`, a conversion pass elsewhere) and consumers read them
    // through gemm_bf16a.hip (both operands by LDS-DMA) — for the GEMMs numerically the rounding the fp32-activation kernel applied on its way into LDS.
    // One NON-GEMM consumer reads images too at D > 512: the attention pool's mix takes the bf16 images of the layer hiddens for its softmax-weighted
    // VALUE mix and its in-kernel RMS (fp32 before round 4) — an extra rounding of bf16 mode (a precision trade-off of that mode, not of the kernel), inside the
    // engine-level bf16-vs-fp32 bound of tests/test_gpu_bf16.py (config-5 shape: latents 7.7e-3 max, asserted < 3e-2).
    // `only`: nothing reads the fp32 buffer when the image exists (the producer may skip the fp32 store).
    struct Shadow { const float* src; size_t n; uint16_t* dst; bool only; };
    std::vector<Shadow> shadows;
    uint16_t* shadow_of(const float* p) const {
        for (const auto& sh : shadows) if (p >= sh.src && p < sh.src + sh.n) return sh.dst + (p - sh.src);
        return nullptr;
    }
    bool shadow_only(const float* p) const {
        for (const auto& sh : shadows) if (p >= sh.src && p < sh.src + sh.n) return sh.only;
        return false;
    }

    // ---- bound (raw) weights
    std::vector<d4::AttnW> layer_attn;
    std::vector<d4::FfW> layer_ff;
    std::vector<d4::AttnW> pools;          // depth-1 layer pools, then the final pool
    d4::AttnW cross, lq_in, lq_out;
    d4::FfW sff;
    const float *vres_norm, *vres_w, *inv_freq, *lq_in_queries, *lq_out_queries, *latent_norm, *latent_w;
    const float *lin_w = nullptr, *lin_b = nullptr;      // num_spatial_tokens == num_latent_tokens: Linear(dim_latent -> dim)
    const float *registers, *signal_embed, *step_embed, *agent_learned, *action_learned, *task_embed, *action_embed;
    const float *action_unembed; float* action_unembed_grad;
    const float *cont_embed = nullptr, *cont_unembed = nullptr; float* cont_unembed_grad = nullptr;      // continuous actions (Beta head)
    const float *reward_norm, *reward_w, *reward_centers, *value_centers, *value_support;
    d4::Mlp policy, value, terminal;

    // ---- prepared weight images (workspace)
    std::vector<float*> proj_w, proj_b;
    std::vector<d4::FfPrep> ffp;           // depth layers + special ff at index depth
    std::vector<float*> pq_w, pkv_w;
    // bf16 engine, wide key projection (round 6): pkq_w[l] = the folded key weights of pools l .. depth-2 followed by pool l's query weights,
    // [(depth - l) * hp][D]; kall_b[slab][row][kall_ld] = the keys of every later pool for a hidden (columns p * hp) and, in the last hp columns,
    // the queries of the pool whose input the slab is — a hidden is projected ONCE, when it is produced, for all the pools that will read it
    std::vector<float*> pkq_w;
    uint16_t* kall_b = nullptr;
    int kall_ld = 0;
    // tiled images ([N / 16][K / 4][16][4]) of the small projections the per-frame fused kernels stream (frame_fused.hip); empty vectors:
    // the configuration does not take that path
    std::vector<float*> wo_t, pv_t, po_t;
    float *cq_w, *ckv_w;
    float *lin_kv_w, *lin_q, *lin_gate, *lout_kv_w, *lout_q, *lout_gate, *qtmp;
    float *lout_w;                         // [dl][hd] = to_latent_pred.2.weight @ to_latent_pred.1.attn.to_out.weight
    int32_t *action_offsets, *action_sizes;

    // ---- activations (workspace)
    float *cslabs, *xfc, *xpool_c, *att_c, *pool_u;                   // row-compacted hiddens (spatial + agent rows) and final tokens
    float *slabs, *xpool, *proj0, *proj, *att, *ffh, *pool_q, *pool_kv, *pool_att, *cq, *ckv, *catt;
    float *lat_in, *lkv, *latt, *space, *gs, *okv, *oatt, *oproj, *pred, *x_lat;
    int32_t* sig;
    int64_t* pact;
    float *pcont, *cu_w, *cparams;          // previous continuous actions per frame, head-0 unembed as a GEMM weight [2 nc][4 D], raw Beta parameters
    int* fstate;                           // device frame state {t0} read by the time-attention kernels under graph replay
    int64_t* tasks_dev;                    // engine-owned copy of the task ids (stable address for captured graphs)
    float* cache;
    float *agent_c, *hbuf[2], *hnorm, *rlogits, *term_pool, *term_logit, *l_dwpart = nullptr;

    // ---- learner (workspace; sized by max_learn_rows)
    int LR = 0;
    float *l_save;                         // per-layer saved activations for both MLP heads
    float *l_tmp[4];
    float *l_cparams, *l_dcparams, *l_cu_g;
    float *l_logits, *l_dlogits, *l_vbins, *l_dvbins, *l_returns, *l_adv, *l_scal, *l_mask, *l_rows, *l_dpe;
};

namespace d4 {
int engine_layout(d4_engine* e, bool assign);
int engine_resolve(d4_engine* e);
int engine_prepare(d4_engine* e, hipStream_t s);
int engine_forward(d4_engine* e, const float* latents, int B, int Tq, int t0, int step_log2,
                   const int64_t* tasks, bool need_agent, hipStream_t s, const int* t0_dev = nullptr);
void engine_drop_graphs(d4_engine* e);
int mlp_forward(d4_engine* e, const Mlp& m, const float* x, int ldx, int rows, float* out, int ldo,
                float* save, hipStream_t s);
int learn(d4_engine* e, const d4_learn_io* io, hipStream_t s);
const char* last_error();
}  // namespace d4
