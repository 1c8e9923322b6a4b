// Native runtime of the imagination path: weight binding by reference state_dict key, weight
// preparation, the world-model forward (reference DynamicsWorldModel.forward inference branch +
// AxialSpaceTimeTransformer.forward, D4:6792-7295, 2927-3267) and the rollout driver (reference
// DynamicsWorldModel.generate, D4:6308-6774).  Everything is enqueued on one HIP stream without
// host synchronisation; the only host-side state is the KV-cache frame counter.
#include "common.h"
#include "engine.h"
#include "prof.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

using namespace d4;

namespace d4 {

static constexpr float RMS_EPS = 1.1920928955078125e-07f;   // torch.finfo(float32).eps (nn.RMSNorm eps=None)

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ------------------------------------------------------------------------------------- layout
int engine_layout(d4_engine* e, bool assign) {
    const d4_config& c = e->c;
    size_t off = 0;
    auto alloc_bytes = [&](size_t bytes) -> char* {
        off = (off + 255) / 256 * 256;
        char* p = assign ? e->ws + off : nullptr;
        off += bytes;
        return p;
    };
    auto fl = [&](size_t n) { return reinterpret_cast<float*>(alloc_bytes(n * sizeof(float))); };

    const int D = e->D, hd = e->hd, hp = e->hp, depth = c.depth, S = e->S;
    const int n = c.num_latent_tokens, dl = c.dim_latent, ns = c.num_spatial_tokens;
    const size_t M = e->Mmax, Fr = e->Fr;

    e->proj_w.assign(depth, nullptr); e->proj_b.assign(depth, nullptr);
    for (int l = 0; l < depth; ++l) {
        const int N = l == 0 ? e->Nproj0 : e->Nproj;
        e->proj_w[l] = fl((size_t)N * D);
        e->proj_b[l] = fl(N);
    }
    e->ffp.assign(depth + 1, FfPrep{});
    for (int l = 0; l <= depth; ++l) {
        e->ffp[l].w1 = fl((size_t)2 * e->inner_pad * D);
        e->ffp[l].b1 = fl((size_t)2 * e->inner_pad);
        e->ffp[l].w2 = fl((size_t)D * e->inner_pad);
    }
    e->pq_w.assign(depth, nullptr); e->pkv_w.assign(depth, nullptr);
    for (int p = 0; p < depth; ++p) {
        e->pq_w[p] = fl((size_t)(hp + e->php) * D);
        e->pkv_w[p] = fl((size_t)2 * hp * D);
    }
    // wide key projection (bf16 activation images only; see engine.h): the weights of pool step l = keys of pools l .. depth-2, then queries of pool l
    const bool wide_keys = e->bf16 && !e->fp32_planes() && !e->decoder && !e->encoder && c.pool_heads == 4 && hp == 256 && D <= 1024 && D % 64 == 0 && depth >= 2;
    e->pkq_w.clear();
    if (wide_keys) {
        e->pkq_w.assign(depth - 1, nullptr);
        for (int l = 0; l < depth - 1; ++l) e->pkq_w[l] = fl((size_t)(depth - l) * hp * D);
    }
    // per-frame fused tails (frame_fused.hip): 8 heads x 64 and, for the pool tail, dim 512 with 4 pool heads — dynamics mode only
    e->wo_t.clear(); e->pv_t.clear(); e->po_t.clear();
    if (!e->encoder && !e->decoder && c.attn_dim_head == 64 && c.attn_heads == 8 && D % 32 == 0 && D >= 256) {
        e->wo_t.assign(depth, nullptr);
        for (int l = 0; l < depth; ++l) e->wo_t[l] = fl((size_t)D * hd);
        if (D == 512 && c.pool_heads == 4) {
            e->pv_t.assign(depth, nullptr); e->po_t.assign(depth, nullptr);
            for (int p = 0; p < depth; ++p) { e->pv_t[p] = fl((size_t)hp * D); e->po_t[p] = fl((size_t)D * hp); }
        }
    }
    e->cq_w = fl((size_t)(hd + c.attn_heads) * D);
    e->ckv_w = fl((size_t)2 * hd * D);
    e->lin_kv_w = fl((size_t)2 * hd * dl);
    e->lin_q = fl((size_t)ns * hd);
    e->lin_gate = fl((size_t)ns * c.attn_heads);
    e->lout_kv_w = fl((size_t)2 * hd * D);
    e->lout_q = fl((size_t)n * hd);
    e->lout_gate = fl((size_t)n * c.attn_heads);
    e->qtmp = fl((size_t)(n > ns ? n : ns) * D);
    e->lout_w = fl((size_t)dl * (hd > D ? hd : D));      // LQAP path: [dl][hd]; same-length path: [dl][D] (gamma folded)
    e->action_offsets = reinterpret_cast<int32_t*>(alloc_bytes(sizeof(int32_t) * D4_MAX_ACTION_TYPES));
    e->action_sizes = reinterpret_cast<int32_t*>(alloc_bytes(sizeof(int32_t) * D4_MAX_ACTION_TYPES));

    if (e->bf16) {
        // every GEMM weight of the trunk once more in bf16: prepared images + the raw output projections (upper bound from the config)
        const size_t per_layer = (size_t)(e->Nproj0 + hd) * D + (size_t)3 * e->inner_pad * D;
        const size_t per_pool = (size_t)(hp + e->php + 2 * hp) * D + (size_t)D * hp;
        const size_t extra = (size_t)(hd + c.attn_heads + 2 * hd + hd) * D + (size_t)2 * hd * dl + (size_t)2 * hd * D + (size_t)dl * (hd > D ? hd : D) + (size_t)D * hd + (size_t)D * dl;
        size_t wide = 0;
        for (int l = 0; l < (int)e->pkq_w.size(); ++l) wide += (size_t)(depth - l) * hp * D + 8;
        e->bf16_cap = ((size_t)(depth + 1) * per_layer + (size_t)depth * per_pool + extra + wide + 4096 + 7) / 8 * 8;
        e->bf16_arena = reinterpret_cast<uint16_t*>(alloc_bytes(e->bf16_cap * (e->split ? 3 : (e->h2 ? 2 : 1)) * sizeof(uint16_t)));     // split mode: three planes; fp16x2 mode: two
        if (e->h2) {
            e->wscale_cap = e->bf16_cap / 16 + 64;           // one float per weight row; the narrowest mirrored matrix has >= 16 columns
            e->wscale_arena = reinterpret_cast<float*>(alloc_bytes(e->wscale_cap * sizeof(float)));
        }
    }
    const size_t KR = e->decoder ? (size_t)e->P : (e->encoder ? (size_t)n : (size_t)ns + 1);        // token rows per frame the final stage keeps (compact copies)
    const size_t KQ = e->encoder ? (size_t)n : 1;                        // special tokens per frame that cross-attend (D4:3227-3238)
    e->slabs = fl((size_t)e->nslab * M * D);
    e->xpool = fl(M * D);
    e->cslabs = fl((size_t)e->nslab * Fr * KR * D);
    e->xfc = fl(Fr * KR * D);
    e->xpool_c = fl(Fr * KR * D);
    e->att_c = fl(Fr * KR * hd);
    if (e->encoder) {
        const size_t P = e->P, dp = e->dim_patch;
        e->t2p_wf = fl((size_t)dl * D);
        e->dec_in = fl(Fr * P * dp);
        e->img_tok = fl(Fr * P * D);
        e->enc_out = fl(Fr * (size_t)n * dl);
    }
    if (e->decoder) {
        const size_t P = e->P, dp = e->dim_patch;
        e->pos_emb = fl(P * D);
        e->t2p_wf = fl(dp * D);
        e->dec_in = fl(Fr * P * dp);
        e->dec_out = fl(Fr * P * dp);
        e->img_tok = fl(Fr * P * D);
        e->lat_tok = fl(Fr * (size_t)n * D);
        e->posA = fl(P * (size_t)(2 * D > 4 ? 2 * D : 4));
        e->posB = fl(P * (size_t)(2 * D > 4 ? 2 * D : 4));
        e->zerosD = fl((size_t)(2 * D > (int)dp ? 2 * D : dp));
    }
    e->proj0 = fl(M * e->Nproj0);
    e->proj = fl(M * e->Nproj);
    e->att = fl(M * hd);
    e->ffh = fl(M * e->inner_pad);
    e->pool_q = fl(M * e->ldpq);
    e->pool_kv = fl((size_t)e->nslab * M * 2 * hp);
    e->pool_att = fl(M * hp);
    e->pool_u = fl(M * (size_t)e->php * D);
    if (e->h2) {
        e->aexp_slab = reinterpret_cast<int*>(alloc_bytes((size_t)e->nslab * M * sizeof(int)));
        e->aexp_tmp_rows = (size_t)e->nslab * M;
        e->aexp_tmp = reinterpret_cast<int*>(alloc_bytes(e->aexp_tmp_rows * sizeof(int)));
        e->aexp_valid.assign(e->nslab, 0);
    }
    e->shadows.clear();
    if (e->bf16 && !e->fp32_planes() && !e->decoder && !e->encoder) {
        auto sh = [&](const float* src, size_t n, bool only) {
            uint16_t* dst = reinterpret_cast<uint16_t*>(alloc_bytes(n * sizeof(uint16_t)));
            e->shadows.push_back({src, n, dst, only});
        };
        sh(e->slabs, (size_t)e->nslab * M * D, false);
        sh(e->xpool, M * D, false);
        sh(e->att, M * hd, false);
        // `only` (the producer skips the fp32 store) where the ONE consumer is certain to take the bf16-activation kernel — decided by the SAME predicate
        // that consumer call evaluates (gemm_bf16a_applicable on its leading dimensions, strides and contraction; operands from this allocator are
        // 256-byte aligned), not by K % 64 alone: ffh -> the SiLU-GLU output projection, pool_att -> the pool's output projection,
        // pool_u -> the per-head value projection (batch = pool heads, strideA = D, strideW = 64 D)
        auto consumer_takes_images = [&](int K, int lda, int ldw, int batch, int64_t strideA, int64_t strideW) {
            alignas(16) static const uint16_t probe[8] = {0};
            GemmArgs g{nullptr, lda, nullptr, ldw, nullptr, 0, nullptr, nullptr, 0, 1, 64, K, 0, RMS_EPS};
            g.Ab = probe; g.Wb = probe; g.batch = batch; g.strideA = strideA; g.strideW = strideW;
            return gemm_bf16a_applicable(g);
        };
        sh(e->ffh, M * e->inner_pad, consumer_takes_images(e->inner_pad, e->inner_pad, e->inner_pad, 1, 0, 0));
        sh(e->pool_att, M * hp, consumer_takes_images(hp, hp, hp, 1, 0, 0));
        sh(e->pool_u, M * (size_t)e->php * D, consumer_takes_images(D, e->php * D, D, e->php, D, (int64_t)64 * D));
        // the attention pools' projected KEYS (mix path): written as a bf16 image only — the pool's key projection at N = 256 is bound by its output
        // bytes (170 flop per byte with fp32 keys), and the pool mix reads every key row once: half the bytes on both sides.  The scores then see keys
        // rounded to bf16 like every other activation of this mode.
        if (c.pool_heads == 4 && D <= 1024) sh(e->pool_kv, (size_t)e->nslab * M * 2 * hp, true);
    }
    e->kall_b = nullptr; e->kall_ld = 0;
    if (!e->pkq_w.empty()) {
        e->kall_ld = depth * hp;                       // depth - 1 pools' keys + one query block
        e->kall_b = reinterpret_cast<uint16_t*>(alloc_bytes((size_t)(2 * depth - 1) * M * e->kall_ld * sizeof(uint16_t)));
    }
    e->cq = fl(Fr * KQ * e->ldcq);
    e->ckv = fl(M * 2 * hd);
    e->catt = fl(Fr * KQ * hd);
    e->lat_in = fl(Fr * n * dl);
    e->lkv = fl(Fr * n * 2 * hd);
    e->latt = fl(Fr * ns * hd);
    e->space = fl(Fr * ns * D);
    e->gs = fl(Fr * ns * D);
    e->okv = fl(Fr * ns * 2 * hd);
    e->oatt = fl(Fr * n * hd);
    e->oproj = fl(Fr * n * D);
    e->pred = fl(Fr * n * dl);
    e->x_lat = fl((size_t)e->maxB * n * dl);
    e->sig = reinterpret_cast<int32_t*>(alloc_bytes(sizeof(int32_t) * Fr));
    e->pact = reinterpret_cast<int64_t*>(alloc_bytes(sizeof(int64_t) * Fr * (e->na > 0 ? e->na : 1)));
    e->pcont = fl(Fr * (size_t)(e->nc > 0 ? e->nc : 1));
    e->cu_w = fl((size_t)2 * (e->nc > 0 ? e->nc : 1) * 4 * D);
    e->cparams = fl((size_t)e->maxB * (2 * e->nc + 4));
    e->fstate = reinterpret_cast<int*>(alloc_bytes(sizeof(int) * 16));
    e->tasks_dev = reinterpret_cast<int64_t*>(alloc_bytes(sizeof(int64_t) * e->maxB));
    e->cache = fl((size_t)(e->Lt > 0 ? e->Lt : 1) * 2 * e->maxB * S * c.attn_heads * e->Tcap * c.attn_dim_head);

    int maxdim = 4 * D;
    if (c.reward_num_bins > maxdim) maxdim = c.reward_num_bins;
    if (c.value_num_bins > maxdim) maxdim = c.value_num_bins;
    if (4 * dl > maxdim) maxdim = 4 * dl;
    const size_t hb = (size_t)e->maxB * maxdim;
    e->agent_c = fl((size_t)e->maxB * D);
    e->hbuf[0] = fl(hb); e->hbuf[1] = fl(hb); e->hnorm = fl(hb);
    e->rlogits = fl((size_t)e->maxB * c.reward_num_bins);
    e->term_pool = fl((size_t)e->maxB * dl);
    e->term_logit = fl(e->maxB);

    // learner
    e->LR = c.max_learn_rows;
    if (e->LR > 0) {
        const size_t R = e->LR;
        // saved per layer: x (input), xhat (normalised * gamma input of the linear), z (pre-activation)
        e->l_save = fl(e->policy.save_floats(R) + e->value.save_floats(R));
        for (int i = 0; i < 4; ++i) e->l_tmp[i] = fl(R * maxdim);
        e->l_dwpart = fl(L_DWPART_FLOATS);       // partial products of the weight-gradient GEMMs' k-slices (gemm_tn.hip)
        e->l_cparams = fl(R * (size_t)(2 * e->nc + 4)); e->l_dcparams = fl(R * (size_t)(2 * e->nc + 4)); e->l_cu_g = fl((size_t)(2 * e->nc + 4) * 4 * D);
        e->l_logits = fl(R * (size_t)((e->A + 3) / 4 * 4 + 4));
        e->l_dlogits = fl(R * (size_t)((e->A + 3) / 4 * 4 + 4));
        e->l_vbins = fl(R * (size_t)((c.value_num_bins + 3) / 4 * 4));
        e->l_dvbins = fl(R * (size_t)((c.value_num_bins + 3) / 4 * 4));
        e->l_returns = fl(R);
        e->l_adv = fl(R);
        e->l_mask = fl(R);
        e->l_rows = fl(4 * R);
        e->l_dpe = fl(R * 4 * (size_t)D);
        e->l_scal = fl(64);
    }
    e->ws_need = off + 256;
    return 0;
}

// ------------------------------------------------------------------------------------- key lookup
static thread_local char g_key[256];
static const char* keyf(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_key, sizeof(g_key), fmt, ap);
    va_end(ap);
    return g_key;
}

struct Resolver {
    d4_engine* e;
    int rc = 0;
    const float* get(const char* key, int64_t numel, float** grad = nullptr) {
        auto it = e->bound.find(key);
        if (it == e->bound.end() && numel == 0) return nullptr;      // empty parameter (e.g. no register tokens): nothing to bind
        if (it == e->bound.end()) {
            if (!rc) set_error("weight '%s' is not bound (expected %lld elements)", key, (long long)numel);
            rc = 4;
            return nullptr;
        }
        if (it->second.n != numel) {
            if (!rc) set_error("weight '%s' has %lld elements, expected %lld", key, (long long)it->second.n, (long long)numel);
            rc = 4;
            return nullptr;
        }
        if (grad) *grad = it->second.g;
        return it->second.p;
    }
    void attn(AttnW& a, const std::string& pre, int dim_q, int dim_kv, int heads, bool ctx_norm, bool mix, int dh = 64) {
        const int inner = heads * dh;
        a.norm = get((pre + "norm.weight").c_str(), dim_q);
        if (ctx_norm) a.norm_ctx = get((pre + "norm_context.weight").c_str(), dim_kv);
        a.to_q = get((pre + "to_q.weight").c_str(), (int64_t)inner * dim_q);
        a.to_k = get((pre + "to_k.weight").c_str(), (int64_t)inner * dim_kv);
        a.to_v = get((pre + "to_v.weight").c_str(), (int64_t)inner * dim_kv);
        a.to_out = get((pre + "to_out.weight").c_str(), (int64_t)dim_q * inner);
        a.to_gates = get((pre + "to_gates.0.weight").c_str(), (int64_t)heads * dim_q);
        a.k_gamma = get((pre + "k_heads_rmsnorm.gamma").c_str(), (int64_t)heads * dh);
        if (mix) {
            a.mix_w = get((pre + "to_learned_value_residual_mix.0.weight").c_str(), (int64_t)heads * dim_q);
            a.mix_b = get((pre + "to_learned_value_residual_mix.0.bias").c_str(), heads);
        }
    }
    void ff(FfW& f, const std::string& pre, int D, int inner) {
        f.norm = get((pre + "norm.weight").c_str(), D);
        f.in_w = get((pre + "proj_in.weight").c_str(), (int64_t)2 * inner * D);
        f.in_b = get((pre + "proj_in.bias").c_str(), 2 * inner);
        f.out_w = get((pre + "proj_out.weight").c_str(), (int64_t)D * inner);
        f.out_b = get((pre + "proj_out.bias").c_str(), D);
    }
    void mlp(Mlp& m, const std::string& pre) {
        for (int i = 0; i < m.nl; ++i) {
            const int64_t nw = (int64_t)m.dims[i + 1] * m.dims[i];
            if (m.recipe == D4_MLP_PRE_RMS) {
                m.g[i] = get(keyf("%slayers.%d.0.weight", pre.c_str(), i), m.dims[i], &m.dg[i]);
                m.w[i] = get(keyf("%slayers.%d.1.weight", pre.c_str(), i), nw, &m.dw[i]);
                m.b[i] = get(keyf("%slayers.%d.1.bias", pre.c_str(), i), m.dims[i + 1], &m.db[i]);
            } else if (i < m.nl - 1) {
                m.w[i] = get(keyf("%slayers.%d.0.weight", pre.c_str(), i), nw, &m.dw[i]);
                m.b[i] = get(keyf("%slayers.%d.0.bias", pre.c_str(), i), m.dims[i + 1], &m.db[i]);
                m.g[i] = get(keyf("%slayers.%d.1.weight", pre.c_str(), i), m.dims[i + 1], &m.dg[i]);
                m.nb[i] = get(keyf("%slayers.%d.1.bias", pre.c_str(), i), m.dims[i + 1], &m.dnb[i]);
            } else {
                m.w[i] = get(keyf("%slayers.%d.weight", pre.c_str(), i), nw, &m.dw[i]);
                m.b[i] = get(keyf("%slayers.%d.bias", pre.c_str(), i), m.dims[i + 1], &m.db[i]);
            }
        }
    }
};

static void mlp_dims(Mlp& m, int dim_in, int dim, int dim_out, int depth, int recipe) {
    // create_mlp(dim, depth, dim_in, dim_out): widths (dim_in, dim x (depth + 1), dim_out)  [recipe: DESIGN.md]
    int k = 0;
    m.recipe = recipe;
    m.dims[k++] = dim_in;
    for (int i = 0; i <= depth; ++i) m.dims[k++] = dim;
    m.dims[k++] = dim_out;
    m.nl = k - 1;
    for (int i = 0; i < 9; ++i) { m.dg[i] = m.dnb[i] = m.dw[i] = m.db[i] = nullptr; m.g[i] = m.nb[i] = m.w[i] = m.b[i] = nullptr; }
}

int engine_resolve(d4_engine* e) {
    const d4_config& c = e->c;
    Resolver r{e};
    const int D = e->D, h = c.attn_heads;
    e->layer_attn.assign(c.depth, AttnW{});
    e->layer_ff.assign(c.depth, FfW{});
    const std::string tp = e->decoder ? "decoder.transformer." : (e->encoder ? "encoder_transformer." : "transformer.");
    for (int l = 0; l < c.depth; ++l) {
        r.attn(e->layer_attn[l], tp + keyf("layers.%d.2.fn.", l), D, D, h, false, true, c.attn_dim_head);
        r.ff(e->layer_ff[l], tp + keyf("layers.%d.3.fn.", l), D, e->inner);
    }
    e->pools.assign(c.depth, AttnW{});
    for (int p = 0; p < c.depth - 1; ++p)
        r.attn(e->pools[p], tp + keyf("attn_pools.%d.fn.attn.", p), D, D, c.pool_heads, true, false);
    r.attn(e->pools[c.depth - 1], tp + "final_attn_pool.fn.attn.", D, D, c.pool_heads, true, false);
    e->vres_norm = r.get((tp + "to_value_residual.0.weight").c_str(), D);
    e->vres_w = r.get((tp + "to_value_residual.1.weight").c_str(), (int64_t)e->hd * D);
    e->inv_freq = r.get((tp + "time_rotary.inv_freq").c_str(), c.attn_dim_head / 2);
    if (e->decoder) {
        // VideoTokenizer / VideoDecoderNetwork pieces around the trunk (D4:3526-3557, 3894-3899, 3938).  The trunk's final special
        // cross-attention / feedforward only update the one special token (the last latent token), which nothing reads: not bound.
        const int dp = e->dim_patch;
        e->ld_w = r.get("latents_to_decoder.weight", (int64_t)D * c.dim_latent);
        e->time_embed = r.get("time_embed.weight", (int64_t)c.decoder_flow_steps * D);
        e->npt_w = r.get("noised_patch_to_tokens.1.weight", (int64_t)D * dp);
        e->npt_b = r.get("noised_patch_to_tokens.1.bias", D);
        e->npt_ln = r.get("noised_patch_to_tokens.2.weight", D);
        e->t2p_w = r.get("decoder.tokens_to_patch.0.weight", (int64_t)dp * D);
        e->t2p_b = r.get("decoder.tokens_to_patch.0.bias", dp);
        e->final_norm = r.get((tp + "final_norm.weight").c_str(), D);
        r.mlp(e->posmlp, "decoder.to_decoder_pos_emb.");
        return r.rc;
    }
    r.attn(e->cross, tp + "final_special_cross_attn.fn.", D, D, h, true, false, c.attn_dim_head);
    r.ff(e->sff, tp + "final_special_ff.fn.", D, e->inner);
    if (e->encoder) {
        // VideoTokenizer pieces around the encoder trunk (D4:3838-3848, 3796, 3936)
        const int dp = e->dim_patch;
        e->ptt_w = r.get("patch_to_tokens.1.weight", (int64_t)D * dp);
        e->ptt_b = r.get("patch_to_tokens.1.bias", D);
        e->ptt_ln = r.get("patch_to_tokens.2.weight", D);
        e->latent_tokens = r.get("latent_tokens", (int64_t)c.num_latent_tokens * D);
        e->e2l_w = r.get("encoded_to_latents.weight", (int64_t)c.dim_latent * D);
        e->final_norm = r.get((tp + "final_norm.weight").c_str(), D);
        return r.rc;
    }
    e->latent_norm = r.get("to_latent_pred.0.weight", D);
    e->latent_w = r.get("to_latent_pred.2.weight", (int64_t)c.dim_latent * D);
    if (c.num_spatial_tokens == c.num_latent_tokens) {
        // one spatial token per latent token: Linear in, RMSNorm -> Linear out, no learned-query pools   D4:4816-4834
        e->lin_w = r.get("latents_to_spatial_tokens.weight", (int64_t)D * c.dim_latent);
        e->lin_b = r.get("latents_to_spatial_tokens.bias", D);
    } else {
        r.attn(e->lq_in, "latents_to_spatial_tokens.attn.", D, c.dim_latent, h, true, false, c.attn_dim_head);
        e->lq_in_queries = r.get("latents_to_spatial_tokens.queries", (int64_t)c.num_spatial_tokens * D);
        r.attn(e->lq_out, "to_latent_pred.1.attn.", D, D, h, true, false, c.attn_dim_head);
        e->lq_out_queries = r.get("to_latent_pred.1.queries", (int64_t)c.num_latent_tokens * D);
    }
    e->registers = r.get("register_tokens", (int64_t)c.num_register_tokens * D);
    e->signal_embed = r.get("signal_levels_embed.weight", (int64_t)c.max_steps * (D / 2));
    e->step_embed = r.get("step_size_embed.weight", (int64_t)(int)round(log2((double)c.max_steps)) * (D / 2));
    e->agent_learned = r.get("agent_learned_embed", D);
    e->action_learned = r.get("action_learned_embed", D);
    e->task_embed = c.num_tasks > 0 ? r.get("task_embed.weight", (int64_t)c.num_tasks * D) : nullptr;
    e->action_embed = e->A > 0 ? r.get("action_embedder.discrete_action_embed.weight", (int64_t)e->A * D) : nullptr;
    e->action_unembed = e->A > 0 ? r.get("action_embedder.discrete_action_unembed",
                                         (int64_t)e->A * c.multi_token_pred_len * 4 * D, &e->action_unembed_grad) : nullptr;
    if (e->nc > 0) {
        e->cont_embed = r.get("action_embedder.continuous_action_embed.weight", (int64_t)e->nc * D);
        e->cont_unembed = r.get("action_embedder.continuous_action_unembed", (int64_t)e->nc * c.multi_token_pred_len * 4 * D * 2, &e->cont_unembed_grad);
    }
    e->reward_norm = r.get("to_reward_pred.params.0", (int64_t)c.multi_token_pred_len * D);
    e->reward_w = r.get("to_reward_pred.params.1", (int64_t)c.multi_token_pred_len * c.reward_num_bins * D);
    if (c.reward_encoder_type == 1) {       // symexp_two_hot: softmax . bin_values; the learner's targets are two-hot over the same values
        e->reward_centers = r.get("reward_encoder.bin_values", c.reward_num_bins);
        e->value_centers = r.get("value_encoder.bin_values", c.value_num_bins);
        e->value_support = e->value_centers;
    } else {
        e->reward_centers = r.get("reward_encoder.centers", c.reward_num_bins);
        e->value_centers = r.get("value_encoder.centers", c.value_num_bins);
        e->value_support = r.get("value_encoder.support", c.value_num_bins + 1);
    }
    r.mlp(e->policy, "policy_head.");
    r.mlp(e->value, "value_head.");
    if (c.predict_terminals) r.mlp(e->terminal, "to_state_terminal_pred.0.");
    return r.rc;
}

// ------------------------------------------------------------------------------------- bf16 mirrors
// The trunk's GEMMs run on the bf16 MFMA kernel when the engine was created with matmul_bf16: engine_forward sets the thread's
// active engine, and every GEMM issued below it swaps its weight pointer for the bf16 mirror made at prepare time.
static thread_local d4_engine* t_bf16 = nullptr;
static thread_local int t_sig_uniform = -1;    // >= 0: every frame of the next engine_forward is at this signal level (decode loop: no fill kernel)
static thread_local int t_keep_lo = 1;         // first token row of a frame the compacted copies keep (set by engine_forward)

static int mirror_weight(d4_engine* e, const float* src, size_t rows, int ld, hipStream_t s) {
    size_t n = rows * (size_t)ld;
    if (!e->bf16 || !src || n == 0) return 0;
    n = (n + 7) / 8 * 8;
    D4_REQUIRE(e->bf16_used + n <= e->bf16_cap, "bf16 weight arena exhausted (%zu + %zu > %zu)", e->bf16_used, n, e->bf16_cap);
    uint16_t* dst = e->bf16_arena + e->bf16_used;
    if (e->h2) {
        // two fp16 planes under one exact power-of-two scale per row (gemm_h2.hip); a matrix too narrow for that kernel's 16-byte plane rows
        // (ld % 8) gets no mirror: its GEMMs stay on the f32-input kernels
        if ((ld % 8) != 0) return 0;
        D4_REQUIRE(e->wscale_used + rows <= e->wscale_cap, "weight scale arena exhausted");
        float* sc = e->wscale_arena + e->wscale_used;
        e->wscale_used += rows;
        e->bf16_used += n;
        e->mirrors.push_back({src, n, dst, ld, sc});
        return split_f16x2_rows(src, dst, (int)rows, ld, ld, (int64_t)e->bf16_cap, sc, s);
    }
    e->bf16_used += n;
    e->mirrors.push_back({src, n, dst, ld, nullptr});
    if (e->split) return split_bf16x3(src, dst, (int64_t)n, (int64_t)e->bf16_cap, s);
    return cvt_f32_to_bf16(src, dst, (int64_t)n, s);
}

// fp16x2 mode: attach the fp16 planes and row scales of g's weight view (whole rows of a mirrored matrix at its leading dimension) and — when the
// dispatcher's rule sends the call to gemm_h2.hip — the scale exponents of A's rows: a residual-stream slab keeps them for the evaluation
// (computed once, by the first GEMM that reads the slab), any other activation buffer gets them right here (row_scale_exp: one small launch)
static int attach_h2(d4_engine* e, GemmArgs& g, hipStream_t s) {
    g.Wb = nullptr; g.wplane = 0; g.wscale = nullptr; g.strideWs = 0; g.aexp = nullptr;
    const int nb = g.batch > 1 ? g.batch : 1;
    for (const auto& m : e->mirrors) {
        if (g.W < m.src || g.W >= m.src + m.n) continue;
        const size_t off = (size_t)(g.W - m.src);
        if (g.ldw != m.ld || (off % (size_t)m.ld) != 0 || (nb > 1 && (g.strideW % m.ld) != 0)) return 0;     // not a row view: f32-input kernels
        g.Wb = m.dst + off; g.wplane = (int64_t)e->bf16_cap; g.wscale = m.scales + off / (size_t)m.ld; g.strideWs = nb > 1 ? g.strideW / m.ld : 0;
        break;
    }
    if (!g.Wb || !gemm_h2_takes(g)) { g.Wb = nullptr; g.wplane = 0; g.wscale = nullptr; g.strideWs = 0; return 0; }
    if (nb > 1) return 0;                                   // batched views: the kernel finds the row exponents itself
    const size_t M = e->aexp_M;
    if (M > 0 && g.A >= e->slabs && g.A < e->slabs + (size_t)e->nslab * M * e->D && g.lda == e->D && g.K == e->D && ((size_t)(g.A - e->slabs) % (size_t)e->D) == 0) {
        const size_t row0 = (size_t)(g.A - e->slabs) / e->D;
        if (row0 + g.M <= (size_t)e->nslab * M) {
            for (size_t sl = row0 / M; sl <= (row0 + g.M - 1) / M; ++sl) {
                if (e->aexp_valid[sl]) continue;
                if (int rc = row_scale_exp(e->slabs + sl * M * e->D, e->D, (int)M, e->D, e->aexp_slab + sl * M, s)) return rc;
                e->aexp_valid[sl] = 1;
            }
            g.aexp = e->aexp_slab + row0;
            return 0;
        }
    }
    if ((size_t)g.M <= e->aexp_tmp_rows && (g.K % 4) == 0) {
        if (int rc = row_scale_exp(g.A, g.lda, g.M, g.K, e->aexp_tmp, s)) return rc;
        g.aexp = e->aexp_tmp;
    }
    return 0;
}

static int engine_gemm(GemmArgs& g, hipStream_t s) {
    if (d4_engine* e = t_bf16) {
        if (e->h2) {
            if (int rc = attach_h2(e, g, s)) return rc;
            return gemm(g, s);
        }
        g.Wb = nullptr;
        for (const auto& m : e->mirrors)
            if (g.W >= m.src && g.W < m.src + m.n) { g.Wb = m.dst + (g.W - m.src); break; }
        D4_REQUIRE(g.Wb != nullptr, "bf16 engine: a trunk GEMM weight has no bf16 mirror (M=%d N=%d K=%d)", g.M, g.N, g.K);
        if (e->split) {
            g.wplane = (int64_t)e->bf16_cap;
            if (!gemm_x3_applicable(g)) { g.Wb = nullptr; g.wplane = 0; }
        } else
        if (!gemm_bf16_applicable(g)) g.Wb = nullptr;       // e.g. K not a multiple of 32: this call stays on the fp32 kernel
        if (!e->fp32_planes() && !e->shadows.empty()) {
            // bf16 activation images: read A's when it has one (and the bf16-activation kernel takes the call), refresh C's either in the
            // epilogue of that kernel or by a conversion pass after any other kernel
            uint16_t* cb = (!g.C && g.Cb) ? g.Cb : e->shadow_of(g.C);     // (a caller may name a bf16-only output itself: the wide key projection)
            g.Ab = g.Wb ? e->shadow_of(g.A) : nullptr;
            if (g.Ab && !gemm_bf16a_applicable(g)) g.Ab = nullptr;
            D4_REQUIRE(g.Ab || g.C, "bf16 engine: a GEMM (M=%d N=%d K=%d) with a bf16-only output cannot take the bf16-activation kernel", g.M, g.N, g.K);
            D4_REQUIRE(g.Ab || !e->shadow_only(g.A), "bf16 engine: a GEMM (M=%d N=%d K=%d) reads a bf16-only activation buffer but cannot take the bf16-activation kernel", g.M, g.N, g.K);
            if (g.Ab) {
                g.Cb = cb;
                if (cb && e->shadow_only(g.C) && !(g.flags & GEMM_ACCUMULATE)) g.C = nullptr;
                return gemm(g, s);
            }
            if (int rc = gemm(g, s)) return rc;
            if (!cb) return 0;
            const int nb = g.batch > 1 ? g.batch : 1;
            const int ncol = (g.flags & GEMM_SWIGLU) ? g.N / 2 : g.N;
            if (nb == 1) return cvt_rows_bf16(g.C, g.ldc, cb, g.ldc, g.M, ncol, s);
            for (int b = 0; b < nb; ++b)
                if (int rc = cvt_rows_bf16(g.C + b * g.strideC, g.ldc, cb + b * g.strideC, g.ldc, g.M, ncol, s)) return rc;
            return 0;
        }
    }
    return gemm(g, s);
}

// ------------------------------------------------------------------------------------- prepare
static int fold_attn_rows(float* dst, int D, const float* w, const float* gamma, int rows, hipStream_t s) {
    return fold_rows(w, gamma, dst, rows, D, D, s);
}

static int prep_ff(const FfW& f, FfPrep& p, d4_engine* e, hipStream_t s) {
    int rc;
    if ((rc = swiglu_pack_rows(f.in_w, f.in_b, f.norm, p.w1, p.b1, e->inner, e->inner_pad, e->D, s))) return rc;
    return pad_cols(f.out_w, p.w2, e->D, e->inner, e->inner_pad, s);
}

static int gemm_simple(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                       int flags, const float* bias, const float* R, int ldr, hipStream_t s) {
    GemmArgs g{A, lda, W, ldw, C, ldc, bias, R, ldr, M, N, K, flags, RMS_EPS};
    return engine_gemm(g, s);
}

// Two independent projections of equal K: one launch when both are few-row problems (fp32 engine), else one after the other.
static int gemm_two(GemmArgs a, GemmArgs b, hipStream_t s) {
    if ((!t_bf16 || t_bf16->fp32_planes()) && gemm_skinny_pair_applicable(a, b)) return gemm_skinny_pair(a, b, s);
    if (t_bf16 && t_bf16->h2) {
        // fp16x2 mode: each call by the rule (the key projection of many rows on gemm_h2.hip, the small query projection on the f32-input kernels)
        int rc;
        if ((rc = engine_gemm(a, s))) return rc;
        return engine_gemm(b, s);
    }
    if (!t_bf16 || t_bf16->split) {
        // fp32 engine: one grid for both when the dispatcher's pair form applies (gemm_pair falls back to two launches itself)
        if (d4_engine* e = t_bf16) {
            for (GemmArgs* g : {&a, &b}) {
                g->Wb = nullptr;
                for (const auto& m : e->mirrors)
                    if (g->W >= m.src && g->W < m.src + m.n) { g->Wb = m.dst + (g->W - m.src); break; }
                g->wplane = g->Wb ? (int64_t)e->bf16_cap : 0;
                if (!g->Wb || !gemm_x3_applicable(*g)) { g->Wb = nullptr; g->wplane = 0; }
            }
        }
        return gemm_pair(a, b, s);
    }
    // bf16 engine: both activations have bf16 images and neither output has one -> one grid on the bf16-activation kernel (the pool's query
    // projection rides in the key projection's launch)
    if (d4_engine* e = t_bf16) {
        if (!e->fp32_planes() && !e->shadows.empty()) {
            GemmArgs pa = a, pb = b;
            bool ok = true;
            for (GemmArgs* g : {&pa, &pb}) {
                g->Wb = nullptr;
                for (const auto& m : e->mirrors)
                    if (g->W >= m.src && g->W < m.src + m.n) { g->Wb = m.dst + (g->W - m.src); break; }
                g->Ab = e->shadow_of(g->A);
                ok = ok && g->Wb && g->Ab;
                g->Cb = e->shadow_of(g->C);                                   // an output with an image: written by the same epilogue,
                if (g->Cb && e->shadow_only(g->C)) g->C = nullptr;            // ... alone when nothing reads the fp32 values (the pools' keys)
            }
            if (ok && gemm_bf16a_pair_applicable(pa, pb)) return gemm_bf16a_pair(pa, pb, s);
        }
    }
    int rc;
    if ((rc = engine_gemm(a, s))) return rc;
    return engine_gemm(b, s);
}

static int gemm_c2(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K, int flags,
                   const float* bias, const float* R, int ldr, float* C2, int S, int ns, int has_agent, hipStream_t s) {
    GemmArgs g{A, lda, W, ldw, C, ldc, bias, R, ldr, M, N, K, flags, RMS_EPS, 0};
    g.C2 = C2; g.ldc2 = N; g.c2_S = S; g.c2_lo = t_keep_lo; g.c2_hi = t_keep_lo + ns; g.c2_last = has_agent;
    return engine_gemm(g, s);
}

int engine_prepare(d4_engine* e, hipStream_t s) {
    const d4_config& c = e->c;
    D4_REQUIRE(e->ws != nullptr, "workspace not set");
    int rc;
    if ((rc = engine_resolve(e))) return rc;
    const int D = e->D, hd = e->hd, h = c.attn_heads, hp = e->hp;
    for (int l = 0; l < c.depth; ++l) {
        const AttnW& a = e->layer_attn[l];
        float* w = e->proj_w[l];
        const int N = l == 0 ? e->Nproj0 : e->Nproj;
        if ((rc = fill_f32(e->proj_b[l], 0.f, N, s))) return rc;
        if ((rc = fold_attn_rows(w, D, a.to_q, a.norm, hd, s))) return rc;
        if ((rc = fold_attn_rows(w + (size_t)hd * D, D, a.to_k, a.norm, hd, s))) return rc;
        if ((rc = fold_attn_rows(w + (size_t)2 * hd * D, D, a.to_v, a.norm, hd, s))) return rc;
        if ((rc = fold_attn_rows(w + (size_t)3 * hd * D, D, a.to_gates, a.norm, h, s))) return rc;
        if ((rc = fold_attn_rows(w + (size_t)(3 * hd + h) * D, D, a.mix_w, a.norm, h, s))) return rc;
        if ((rc = copy_rows(a.mix_b, h, e->proj_b[l] + 3 * hd + h, h, 1, h, s))) return rc;
        if (l == 0 && (rc = fold_attn_rows(w + (size_t)(3 * hd + 2 * h) * D, D, e->vres_w, e->vres_norm, hd, s))) return rc;
        if ((rc = prep_ff(e->layer_ff[l], e->ffp[l], e, s))) return rc;
    }
    if (!e->decoder && (rc = prep_ff(e->sff, e->ffp[c.depth], e, s))) return rc;
    for (int p = 0; p < c.depth; ++p) {
        const AttnW& a = e->pools[p];
        if ((rc = fold_attn_rows(e->pq_w[p], D, a.to_q, a.norm, hp, s))) return rc;
        if ((rc = fold_attn_rows(e->pq_w[p] + (size_t)hp * D, D, a.to_gates, a.norm, e->php, s))) return rc;
        if ((rc = fold_attn_rows(e->pkv_w[p], D, a.to_k, a.norm_ctx, hp, s))) return rc;
        if ((rc = fold_attn_rows(e->pkv_w[p] + (size_t)hp * D, D, a.to_v, a.norm_ctx, hp, s))) return rc;
        if (!e->pv_t.empty()) {
            if ((rc = tile16_weights(e->pkv_w[p] + (size_t)hp * D, D, e->pv_t[p], hp, D, s))) return rc;
            if ((rc = tile16_weights(a.to_out, hp, e->po_t[p], D, hp, s))) return rc;
        }
    }
    for (int l = 0; l < (int)e->pkq_w.size(); ++l) {
        for (int p = l; p < c.depth - 1; ++p)
            if ((rc = copy_rows(e->pkv_w[p], D, e->pkq_w[l] + (size_t)(p - l) * hp * D, D, hp, D, s))) return rc;
        if ((rc = copy_rows(e->pq_w[l], D, e->pkq_w[l] + (size_t)(c.depth - 1 - l) * hp * D, D, hp, D, s))) return rc;
    }
    for (int l = 0; l < (int)e->wo_t.size(); ++l)
        if ((rc = tile16_weights(e->layer_attn[l].to_out, hd, e->wo_t[l], D, hd, s))) return rc;
    if (e->decoder) {
        // positional embedding of the patch grid: the normed MLP (recipe: engine.h) of the (row, column) coordinates — batch independent,
        // evaluated once here (D4:3617-3625).  Its first layer has K = 2: coordinates / weight are zero-padded to 4 columns.
        const Mlp& m = e->posmlp;
        const int P = e->P;
        float* cur = e->posA;
        if ((rc = coord_grid(cur, e->nph, e->npw, 4, s))) return rc;
        float* other = e->posB;
        for (int i = 0; i < m.nl; ++i) {
            const int din = m.dims[i], dout = m.dims[i + 1], ld_in = i == 0 ? 4 : din;
            const bool last = i == m.nl - 1;
            const float* w = m.w[i];
            if (i == 0) {                                     // [dout][2] -> [dout][4]
                if ((rc = pad_cols(m.w[0], e->t2p_wf, dout, 2, 4, s))) return rc;      // (t2p_wf is free until the end of prepare)
                w = e->t2p_wf;
            }
            const float* lin_in = cur;
            if (m.recipe == D4_MLP_PRE_RMS) {
                // RMSNorm over the din real columns (for i == 0: 2 of the 4), written back in place with the padding kept zero
                if ((rc = rmsnorm_rows(cur, ld_in, m.g[i], cur, ld_in, P, din, RMS_EPS, s))) return rc;
            }
            float* out = last ? e->pos_emb : other;
            const int act = (!last && !m.post_norm(i)) ? GEMM_SILU : 0;
            if ((rc = gemm_simple(lin_in, ld_in, w, ld_in, out, dout, P, dout, ld_in, act, m.b[i], nullptr, 0, s))) return rc;
            if (m.post_norm(i) && (rc = layernorm_rows(out, dout, m.g[i], m.nb[i], out, dout, P, dout, 1e-5f, 1, s))) return rc;
            other = cur; cur = out;
        }
        // final RMSNorm of the trunk folded into the patch projection: W . diag(gamma), 1 / rms applied inside the GEMM   D4:3246, 3555
        if ((rc = fold_rows(e->t2p_w, e->final_norm, e->t2p_wf, e->dim_patch, D, D, s))) return rc;
        if ((rc = fill_f32(e->zerosD, 0.f, 2 * D > e->dim_patch ? 2 * D : e->dim_patch, s))) return rc;
        D4_HIP(hipStreamSynchronize(s));
        e->prepared = true;
        return 0;
    }
    if ((rc = fold_attn_rows(e->cq_w, D, e->cross.to_q, e->cross.norm, hd, s))) return rc;
    if ((rc = fold_attn_rows(e->cq_w + (size_t)hd * D, D, e->cross.to_gates, e->cross.norm, h, s))) return rc;
    if ((rc = fold_attn_rows(e->ckv_w, D, e->cross.to_k, e->cross.norm_ctx, hd, s))) return rc;
    if ((rc = fold_attn_rows(e->ckv_w + (size_t)hd * D, D, e->cross.to_v, e->cross.norm_ctx, hd, s))) return rc;
    if (e->encoder) {
        // final RMSNorm of the trunk folded into the latent bottleneck: W . diag(gamma), 1 / rms inside the GEMM   D4:3246, 4408
        if ((rc = fold_rows(e->e2l_w, e->final_norm, e->t2p_wf, c.dim_latent, D, D, s))) return rc;
        D4_HIP(hipStreamSynchronize(s));
        e->prepared = true;
        return 0;
    }

    // learned-query pools: the query side is batch independent -> evaluate once        D4:2189, 2206
    const int dl = c.dim_latent, ns = c.num_spatial_tokens, n = c.num_latent_tokens;
    if (ns == n) {
        if ((rc = fold_rows(e->latent_w, e->latent_norm, e->lout_w, dl, D, D, s))) return rc;       // W_latent . diag(gamma)
    } else {
    if ((rc = fold_rows(e->lq_in.to_k, e->lq_in.norm_ctx, e->lin_kv_w, hd, dl, dl, s))) return rc;
    if ((rc = fold_rows(e->lq_in.to_v, e->lq_in.norm_ctx, e->lin_kv_w + (size_t)hd * dl, hd, dl, dl, s))) return rc;
    if ((rc = rmsnorm_rows(e->lq_in_queries, D, e->lq_in.norm, e->qtmp, D, ns, D, RMS_EPS, s))) return rc;
    if ((rc = gemm_simple(e->qtmp, D, e->lq_in.to_q, D, e->lin_q, hd, ns, hd, D, 0, nullptr, nullptr, 0, s))) return rc;
    if ((rc = gemm_simple(e->qtmp, D, e->lq_in.to_gates, D, e->lin_gate, h, ns, h, D, 0, nullptr, nullptr, 0, s))) return rc;
    // to_latent_pred: both norms are applied explicitly on the gathered rows -> plain K/V weights
    if ((rc = fold_rows(e->lq_out.to_k, nullptr, e->lout_kv_w, hd, D, D, s))) return rc;
    if ((rc = fold_rows(e->lq_out.to_v, nullptr, e->lout_kv_w + (size_t)hd * D, hd, D, D, s))) return rc;
    if ((rc = rmsnorm_rows(e->lq_out_queries, D, e->lq_out.norm, e->qtmp, D, n, D, RMS_EPS, s))) return rc;
    if ((rc = gemm_simple(e->qtmp, D, e->lq_out.to_q, D, e->lout_q, hd, n, hd, D, 0, nullptr, nullptr, 0, s))) return rc;
    if ((rc = gemm_simple(e->qtmp, D, e->lq_out.to_gates, D, e->lout_gate, h, n, h, D, 0, nullptr, nullptr, 0, s))) return rc;
    // the pool's output projection and the latent Linear are back-to-back linear maps with nothing in between
    // (D4:2068, 4833): fold them, W[dl][hd] = W_latent[dl][D] . W_out[D][hd]
    if ((rc = gemm_simple(e->latent_w, D, e->lq_out.to_out, hd, e->lout_w, hd, dl, hd, D, GEMM_TRANS_B, nullptr, nullptr, 0, s))) return rc;
    }

    // ---- bf16 mirrors of everything the trunk's GEMMs read (prepared images above + the raw output projections)
    if (e->bf16) {
        e->mirrors.clear(); e->bf16_used = 0;
        e->wscale_used = 0;
        auto mir = [&](const float* w, size_t rows, int ld) { return mirror_weight(e, w, rows, ld, s); };
        for (int l = 0; l < c.depth; ++l) {
            if ((rc = mir(e->proj_w[l], (size_t)(l == 0 ? e->Nproj0 : e->Nproj), D))) return rc;
            if ((rc = mir(e->layer_attn[l].to_out, (size_t)D, hd))) return rc;
        }
        for (int l = 0; l <= c.depth; ++l) {
            if ((rc = mir(e->ffp[l].w1, (size_t)2 * e->inner_pad, D))) return rc;
            if ((rc = mir(e->ffp[l].w2, (size_t)D, e->inner_pad))) return rc;
        }
        for (int p = 0; p < c.depth; ++p) {
            if ((rc = mir(e->pq_w[p], (size_t)(hp + e->php), D))) return rc;
            if ((rc = mir(e->pkv_w[p], (size_t)2 * hp, D))) return rc;
            if ((rc = mir(e->pools[p].to_out, (size_t)D, hp))) return rc;
        }
        for (int l = 0; l < (int)e->pkq_w.size(); ++l)
            if ((rc = mir(e->pkq_w[l], (size_t)(c.depth - l) * hp, D))) return rc;
        if ((rc = mir(e->cq_w, (size_t)(hd + h), D))) return rc;
        if ((rc = mir(e->ckv_w, (size_t)2 * hd, D))) return rc;
        if ((rc = mir(e->cross.to_out, (size_t)D, hd))) return rc;
        if (ns == n) {
            if ((rc = mir(e->lin_w, (size_t)D, dl))) return rc;
            if ((rc = mir(e->lout_w, (size_t)dl, D))) return rc;
        } else {
            if ((rc = mir(e->lin_kv_w, (size_t)2 * hd, dl))) return rc;
            if ((rc = mir(e->lq_in.to_out, (size_t)D, hd))) return rc;
            if ((rc = mir(e->lout_kv_w, (size_t)2 * hd, D))) return rc;
            if ((rc = mir(e->lout_w, (size_t)dl, hd))) return rc;
        }
    }

    int32_t offs[D4_MAX_ACTION_TYPES] = {0}, sizes[D4_MAX_ACTION_TYPES] = {0};
    int o = 0;
    for (int a = 0; a < e->na; ++a) { offs[a] = o; sizes[a] = c.num_discrete_actions[a]; o += sizes[a]; }
    D4_HIP(hipMemcpyAsync(e->action_offsets, offs, sizeof(offs), hipMemcpyHostToDevice, s));
    D4_HIP(hipMemcpyAsync(e->action_sizes, sizes, sizeof(sizes), hipMemcpyHostToDevice, s));
    D4_HIP(hipStreamSynchronize(s));    // the two host arrays above live on this stack frame
    e->prepared = true;
    return 0;
}

// ------------------------------------------------------------------------------------- forward
static int ff_block(d4_engine* e, const FfPrep& fp, const float* out_b, const float* x, int ldx, float* y, int ldy,
                    int rows, hipStream_t s, float* y_compact = nullptr, int S = 0, int has_agent = 1) {
    int rc;
    GemmArgs g1{x, ldx, fp.w1, e->D, e->ffh, e->inner_pad, fp.b1, nullptr, 0, rows, 2 * e->inner_pad, e->D,
                GEMM_RMS_ROWSCALE | GEMM_SWIGLU, RMS_EPS, 2.0 * rows * (2.0 * e->inner) * e->D};
    if ((rc = engine_gemm(g1, s))) return rc;
    GemmArgs g2{e->ffh, e->inner_pad, fp.w2, e->inner_pad, y, ldy, out_b, x, ldx, rows, e->D, e->inner_pad, 0, RMS_EPS,
                2.0 * rows * (double)e->D * e->inner};
    if (y_compact) { g2.C2 = y_compact; g2.ldc2 = e->D; g2.c2_S = S; g2.c2_lo = e->keep_lo; g2.c2_hi = e->keep_hi; g2.c2_last = has_agent; }
    return engine_gemm(g2, s);
}

int g_pool_wide_keys = 1;             // test hook d4_debug_switch("pool_wide_keys")
static int pool_block(d4_engine* e, int p, const float* x, float* y, int L, int M, hipStream_t s, const float* hiddens = nullptr,
                      float* y_compact = nullptr, int S = 0, int has_agent = 1) {
    if (!hiddens) hiddens = e->slabs;
    const d4_config& c = e->c;
    const int D = e->D, hp = e->hp;
    const AttnW& a = e->pools[p];
    int rc;
    const bool mix_path = c.pool_heads == 4 && D <= 1024;
    // (mix path: the head-gate logits are computed inside pool_mix; hp = 256 columns are exactly two / four tile columns)
    const GemmArgs gq{x, D, e->pq_w[p], D, e->pool_q, e->ldpq, nullptr, nullptr, 0, M, mix_path ? hp : hp + e->php, D, GEMM_RMS_ROWSCALE, RMS_EPS};
    if (mix_path) {
        // keys only: [L*M][hp]; values come from ONE per-head projection of the softmax-weighted normalised hiddens
        // bf16 engine, in-loop pools: the hiddens produced since the previous pool (slabs 2p+1, 2p+2; pool 0: slab 0 too) are projected ONCE, onto the
        // key weights of this and every later pool plus this pool's query weights — one launch of N = (depth - p) * 256 per pool instead of a query
        // launch and a key launch of L M rows x 256 (every hidden re-read by every pool: quadratic in depth).  Same products, same row scales
        // (1 / rms of the hidden; both norms' gammas are folded into the weight rows); the queries become bf16 like the keys.
        const bool wide = g_pool_wide_keys && t_bf16 && e->kall_b && hiddens == e->slabs && p < c.depth - 1 && L == 2 * p + 3 && x == e->slabs + (size_t)(L - 1) * M * D;
        PoolMixArgs pm{};
        pm.q = e->pool_q; pm.ldq = e->ldpq; pm.k = e->pool_kv; pm.ldk = hp;
        if (wide) {
            const int j0 = p == 0 ? 0 : 2 * p + 1, ld = e->kall_ld;
            GemmArgs gw{hiddens + (size_t)j0 * M * D, D, e->pkq_w[p], D, nullptr, ld, nullptr, nullptr, 0, (L - j0) * M, (c.depth - p) * hp, D, GEMM_RMS_ROWSCALE, RMS_EPS};
            gw.algo_flops = 2.0 * M * hp * (double)D * ((double)(L - j0) * (c.depth - 1 - p) + 1.0);       // (the query columns of the other slabs are not wanted)
            gw.Cb = e->kall_b + (size_t)j0 * M * ld + (size_t)p * hp;
            if ((rc = engine_gemm(gw, s))) return rc;
            pm.k = nullptr; pm.k_b = e->kall_b + (size_t)p * hp; pm.ldk = ld;
            pm.q = nullptr; pm.q_b = e->kall_b + (size_t)(L - 1) * M * ld + (size_t)(c.depth - 1) * hp; pm.ldq = ld;
        } else {
        const GemmArgs gk{hiddens, D, e->pkv_w[p], D, e->pool_kv, hp, nullptr, nullptr, 0, L * M, hp, D, GEMM_RMS_ROWSCALE, RMS_EPS};
        if ((rc = gemm_two(gq, gk, s))) return rc;
        if (t_bf16) pm.k_b = t_bf16->shadow_of(e->pool_kv);               // bf16 keys (the only copy of them in this mode)
        }
        pm.x = x; pm.ldx = D; pm.gate_w = e->pq_w[p] + (size_t)hp * D; pm.hid = hiddens; pm.D = D; pm.k_gamma = a.k_gamma;
        pm.u = e->pool_u; pm.M = M; pm.L = L; pm.heads = c.pool_heads; pm.eps = RMS_EPS;
        if (t_bf16 && D > 512) pm.hid_b = t_bf16->shadow_of(hiddens);     // (D <= 512 may take the block-per-row form, which reads fp32)
        if (t_bf16) if (uint16_t* ub = t_bf16->shadow_of(e->pool_u)) { pm.u_b = ub; if (t_bf16->shadow_only(e->pool_u)) pm.u = nullptr; }   // only the value GEMM reads the mixes
        // per-frame fused form (mix -> value projection -> output projection + residual in one kernel) where a frame per workgroup fills the chip;
        // frame_fused mode 2 (test hook): the mix stays its own kernel and only the tail is fused
        if (S > 0 && M % S == 0 && !e->pv_t.empty() && (!t_bf16 || t_bf16->fp32_planes()) && frame_pool_tail_applicable(M / S, S, D, c.pool_heads)) {
            const bool tail_only = frame_fused_mode() == 2;
            if (!tail_only) return frame_pool(pm, e->pv_t[p], e->po_t[p], M / S, S, x, D, y, D, y_compact, D, e->keep_lo, e->keep_hi, has_agent, s);
            if ((rc = pool_mix(pm, s))) return rc;
            return frame_pool_tail(e->pool_u, e->pv_t[p], e->po_t[p], M / S, S, D, c.pool_heads, x, D, y, D, y_compact, D, e->keep_lo, e->keep_hi, has_agent, s);
        }
        if ((rc = pool_mix(pm, s))) return rc;
        GemmArgs gv{e->pool_u, c.pool_heads * D, e->pkv_w[p] + (size_t)hp * D, D, e->pool_att, hp, nullptr, nullptr, 0, M, 64, D, 0, RMS_EPS};
        gv.batch = c.pool_heads; gv.strideA = D; gv.strideW = (int64_t)64 * D; gv.strideC = 64;
        if ((rc = engine_gemm(gv, s))) return rc;
    } else {
    const GemmArgs gkv{hiddens, D, e->pkv_w[p], D, e->pool_kv, 2 * hp, nullptr, nullptr, 0, L * M, 2 * hp, D, GEMM_RMS_ROWSCALE, RMS_EPS};
    if ((rc = gemm_two(gq, gkv, s))) return rc;
    SmallAttnArgs sa{};
    sa.q = e->pool_q; sa.q_group_stride = e->ldpq; sa.q_item_stride = 0;
    sa.k = e->pool_kv; sa.k_group_stride = 2 * hp; sa.k_item_stride = (int64_t)M * 2 * hp;
    sa.v = e->pool_kv + hp; sa.v_group_stride = 2 * hp; sa.v_item_stride = (int64_t)M * 2 * hp;
    sa.gate = e->pool_q + hp; sa.g_group_stride = e->ldpq; sa.g_item_stride = 0;
    sa.k_gamma = a.k_gamma;
    sa.out = e->pool_att; sa.o_group_stride = hp; sa.o_item_stride = 0;
    sa.groups = M; sa.heads = c.pool_heads; sa.nq = 1; sa.nk = L;
    sa.out_b = t_bf16 ? t_bf16->shadow_of(e->pool_att) : nullptr;
    if ((rc = small_attn(sa, s))) return rc;
    }
    if (y_compact) return gemm_c2(e->pool_att, hp, a.to_out, hp, y, D, M, D, hp, 0, nullptr, x, D, y_compact, S, e->keep_hi - e->keep_lo, has_agent, s);
    return gemm_simple(e->pool_att, hp, a.to_out, hp, y, D, M, D, hp, 0, nullptr, x, D, s);
}

// Inputs expected in e->sig / e->pact (device).  Results: e->pred [B*Tq][n][dl], e->xfc [B*Tq][ns+1][D] (agent = row ns;
// only valid when need_agent).
void engine_drop_graphs(d4_engine* e) {
    for (auto& g : e->graphs) (void)hipGraphExecDestroy(g.exec);
    e->graphs.clear();
}

static int engine_forward_impl(d4_engine* e, const float* latents, int B, int Tq, int t0, int step_log2,
                               const int64_t* tasks, bool need_agent, hipStream_t s, const int* t0_dev);

int engine_forward(d4_engine* e, const float* latents, int B, int Tq, int t0, int step_log2,
                   const int64_t* tasks, bool need_agent, hipStream_t s, const int* t0_dev) {
    t_bf16 = e->bf16 ? e : nullptr;           // the trunk's GEMMs below pick their bf16 weight mirrors; heads and learner stay fp32
    t_keep_lo = e->keep_lo;
    const int rc = engine_forward_impl(e, latents, B, Tq, t0, step_log2, tasks, need_agent, s, t0_dev);
    t_bf16 = nullptr;
    return rc;
}

static int engine_forward_impl(d4_engine* e, const float* latents, int B, int Tq, int t0, int step_log2,
                               const int64_t* tasks, bool need_agent, hipStream_t s, const int* t0_dev) {
    const d4_config& c = e->c;
    D4_REQUIRE(e->prepared, "engine not prepared");
    D4_REQUIRE(B >= 1 && B <= e->maxB, "batch %d exceeds max_batch %d", B, e->maxB);
    D4_REQUIRE(Tq >= 1 && Tq <= e->maxTq, "parallel frames %d exceed max_parallel_frames %d", Tq, e->maxTq);
    D4_REQUIRE(t0 + Tq <= e->Tcap || e->Lt == 0, "KV cache capacity %d exceeded (%d + %d frames)", e->Tcap, t0, Tq);
    // Denoise evaluations (no agent embedding wanted) drop the agent token altogether: it is the one special token, no
    // ordinary query may attend to it (D4:1781), time attention is per token column, and its own outputs are only read
    // by the heads of the clean step -> nothing consumed downstream depends on it.  S = tokens present per frame.
    const int has_agent = (need_agent && !e->encoder) ? 1 : 0;
    const int D = e->D, hd = e->hd, h = c.attn_heads, S = (need_agent || e->encoder) ? e->S : e->S - 1;
    const int n = c.num_latent_tokens, dl = c.dim_latent, ns = c.num_spatial_tokens;
    const int Fr = B * Tq, M = Fr * S;
    const bool denoise_only = !need_agent && !e->encoder && !e->decoder;        // dynamics evaluation whose last layer only feeds the spatial rows
    int rc;
    if (e->h2) {                              // fp16x2 mode: every slab is rewritten by this evaluation -> no row exponent of the last one is valid
        e->aexp_M = (size_t)M;
        std::fill(e->aexp_valid.begin(), e->aexp_valid.end(), 0);
    }

    float* slab0 = e->slabs;
    auto slab = [&](int j) { return e->slabs + (size_t)j * M * D; };
    const int nkeep = e->decoder ? e->P : (e->encoder ? n : ns + has_agent), Mc = Fr * nkeep;          // rows the final stage reads
    auto cslab = [&](int j) { return e->cslabs + (size_t)j * Mc * D; };
    const bool same_len = ns == n;
    if (e->encoder) {
        // tokens = [patch tokens of the video | learned latent tokens] (D4:4307, 4361-4376); patch rows were laid out in e->dec_in
        const int P = e->P, dp = e->dim_patch;
        if ((rc = gemm_simple(e->dec_in, dp, e->ptt_w, dp, e->img_tok, D, Fr * P, D, dp, 0, e->ptt_b, nullptr, 0, s))) return rc;
        if ((rc = layernorm_rows(e->img_tok, D, e->ptt_ln, nullptr, e->img_tok, D, Fr * P, D, 1e-5f, 0, s))) return rc;
        if ((rc = encoder_pack_tokens(slab0, e->cslabs, e->img_tok, e->latent_tokens, Fr, P, n, D, s))) return rc;
    } else if (e->decoder) {
        // tokens = [pos_emb + patch tokens of the noised video | latent tokens] (D4:3625-3654); `latents` here are the latent TOKENS
        // input row-major [Fr][n][dl]; the noised video's patch rows were laid out in e->dec_in by d4_decoder_forward
        const int P = e->P, dp = e->dim_patch;
        if ((rc = gemm_simple(e->dec_in, dp, e->npt_w, dp, e->img_tok, D, Fr * P, D, dp, 0, e->npt_b, nullptr, 0, s))) return rc;
        if ((rc = layernorm_rows(e->img_tok, D, e->npt_ln, nullptr, e->img_tok, D, Fr * P, D, 1e-5f, 0, s))) return rc;
        if ((rc = gemm_simple(latents, dl, e->ld_w, dl, e->lat_tok, D, Fr * n, D, dl, 0, e->time_embed + (size_t)step_log2 * D, nullptr, 0, s))) return rc;
        if ((rc = decoder_pack_tokens(slab0, e->cslabs, e->pos_emb, e->img_tok, e->lat_tok, Fr, P, n - 1, n, D, s))) return rc;
    } else {
    // ---- latents -> spatial tokens (LearnedQueriesAttentionPool, D4:7168; a plain Linear when there is one per latent)
    if (same_len) {
        if ((rc = gemm_simple(latents, dl, e->lin_w, dl, e->space, D, Fr * n, D, dl, 0, e->lin_b, nullptr, 0, s))) return rc;
    } else {
    if ((rc = gemm_simple(latents, dl, e->lin_kv_w, dl, e->lkv, 2 * hd, Fr * n, 2 * hd, dl, GEMM_RMS_ROWSCALE, nullptr, nullptr, 0, s))) return rc;
    {
        SmallAttnArgs sa{};
        sa.dh = c.attn_dim_head;
        sa.q = e->lin_q; sa.q_group_stride = 0; sa.q_item_stride = hd;
        sa.k = e->lkv; sa.k_group_stride = (int64_t)n * 2 * hd; sa.k_item_stride = 2 * hd;
        sa.v = e->lkv + hd; sa.v_group_stride = (int64_t)n * 2 * hd; sa.v_item_stride = 2 * hd;
        sa.gate = e->lin_gate; sa.g_group_stride = 0; sa.g_item_stride = h;
        sa.k_gamma = e->lq_in.k_gamma;
        sa.out = e->latt; sa.o_group_stride = (int64_t)ns * hd; sa.o_item_stride = hd;
        sa.groups = Fr; sa.heads = h; sa.nq = ns; sa.nk = n;
        if ((rc = small_attn(sa, s))) return rc;
    }
    if ((rc = gemm_simple(e->latt, hd, e->lq_in.to_out, hd, e->space, D, Fr * ns, D, hd, 0, nullptr, nullptr, 0, s))) return rc;
    }

    // ---- pack tokens (D4:7182-7222)
    {
        AssembleArgs a{};
        a.tokens = slab0; a.space = e->space; a.signal_embed = e->signal_embed; a.step_embed = e->step_embed;
        a.registers = e->registers; a.agent_embed = e->agent_learned; a.task_embed = e->task_embed;
        a.action_embed = e->action_embed; a.action_learned = e->action_learned;
        a.signal_levels = t_sig_uniform >= 0 ? nullptr : e->sig; a.signal_uniform = t_sig_uniform; a.prev_actions = e->na > 0 ? e->pact : nullptr; a.tasks = tasks;
        a.prev_cont = e->nc > 0 ? e->pcont : nullptr; a.cont_embed = e->cont_embed; a.nc = e->nc;
        a.action_offsets = e->action_offsets;
        a.B = B; a.Tq = Tq; a.S = S; a.D = D; a.ns = ns; a.nr = c.num_register_tokens; a.na = e->na; a.step_log2 = step_log2;
        a.compact = e->cslabs; a.has_agent = has_agent;
        D4_REQUIRE(tasks == nullptr || c.num_tasks > 0, "tasks given but num_tasks == 0");
        if ((rc = assemble_tokens(a, s))) return rc;
    }
    }
    if (t_bf16) if (uint16_t* sb = t_bf16->shadow_of(slab0)) { if ((rc = cvt_rows_bf16(slab0, D, sb, D, M, D, s))) return rc; }

    // ---- trunk (AxialSpaceTimeTransformer.forward, D4:3040-3223)
    const float* x_in = slab0;
    const float* vres = e->proj0 + 3 * hd + 2 * h;
    for (int l = 0; l < c.depth; ++l) {
        const AttnW& a = e->layer_attn[l];
        float* P = l == 0 ? e->proj0 : e->proj;
        const int ldp = l == 0 ? e->Nproj0 : e->Nproj;
        if ((rc = gemm_simple(x_in, D, e->proj_w[l], D, P, ldp, M, ldp, D, GEMM_RMS_ROWSCALE, e->proj_b[l], nullptr, 0, s))) return rc;
        bool fused_out = false;
        if (e->is_time[l]) {
            TimeAttnArgs ta{};
            ta.proj = P; ta.ldp = ldp; ta.vres = vres; ta.ldv = e->Nproj0; ta.k_gamma = a.k_gamma; ta.inv_freq = e->inv_freq;
            ta.cache = e->cache + (size_t)e->time_index[l] * 2 * e->maxB * e->S * h * e->Tcap * c.attn_dim_head;
            ta.dh = c.attn_dim_head;
            ta.out = e->att; ta.ldo = hd; ta.B = B; ta.S = S; ta.H = h; ta.Tq = Tq; ta.t0 = t0; ta.Tcap = e->Tcap;
            ta.softclamp = c.attn_softclamp_value;
            ta.cache_batch = e->maxB; ta.cache_S = e->S;
            ta.t0_dev = t0_dev;
            ta.out_b = t_bf16 ? t_bf16->shadow_of(e->att) : nullptr;
            if ((rc = time_attn_append(ta, s))) return rc;
        } else {
            SmallAttnArgs sa{};
            sa.dh = c.attn_dim_head;
            const int64_t gs = (int64_t)S * ldp;
            sa.q = P; sa.q_group_stride = gs; sa.q_item_stride = ldp;
            sa.k = P + hd; sa.k_group_stride = gs; sa.k_item_stride = ldp;
            sa.v = P + 2 * hd; sa.v_group_stride = gs; sa.v_item_stride = ldp;
            sa.gate = P + 3 * hd; sa.g_group_stride = gs; sa.g_item_stride = ldp;
            sa.mix = P + 3 * hd + h; sa.m_group_stride = gs; sa.m_item_stride = ldp;
            sa.vres = vres; sa.r_group_stride = (int64_t)S * e->Nproj0; sa.r_item_stride = e->Nproj0;
            sa.k_gamma = a.k_gamma;
            sa.out = e->att; sa.o_group_stride = (int64_t)S * hd; sa.o_item_stride = hd;
            sa.groups = Fr; sa.heads = h; sa.nq = S; sa.nk = S;
            sa.softclamp = c.attn_softclamp_value; sa.mask_special = e->encoder ? n : has_agent; sa.belief = 1;
            sa.out_b = t_bf16 ? t_bf16->shadow_of(e->att) : nullptr;
            if (denoise_only && l == c.depth - 1 && c.depth >= 2 && S <= 16 && S >= 8) {
                sa.q_lo = 1; sa.q_hi = 1 + ns; sa.q_last = 0;
                sa.out = e->att_c; sa.o_group_stride = (int64_t)nkeep * hd; sa.out_b = nullptr;
            }
            // per-frame fused form: attention of all heads of a frame, then its output projection + residual, in one kernel
            const bool tail_compact = denoise_only && l == c.depth - 1 && c.depth >= 2 && S <= 16 && S >= 8;
            if (!tail_compact && !e->wo_t.empty() && (!t_bf16 || t_bf16->fp32_planes()) && frame_attn_out_applicable(sa, D)) {
                if ((rc = frame_attn_out(sa, e->wo_t[l], D, x_in, D, slab(2 * l + 1), D, cslab(2 * l + 1), D, e->keep_lo, e->keep_hi, has_agent, s))) return rc;
                fused_out = true;
            } else if (!tail_compact && hd == 512 && (!t_bf16 || t_bf16->fp32_planes()) && attn_out_cols_applicable(sa, D)) {
                // few frames (launch-bound decode): attention recomputed inside every column workgroup of the output projection, one launch for two
                if ((rc = attn_out_cols(sa, a.to_out, hd, D, x_in, D, slab(2 * l + 1), D, cslab(2 * l + 1), D, e->keep_lo, e->keep_hi, has_agent, s))) return rc;
                fused_out = true;
            } else
            if ((rc = small_attn(sa, s))) return rc;
        }
        // Denoise steps (no agent embedding wanted): nothing downstream reads the last layer's flow / register /
        // action rows, so its output projection and feedforward run on the compacted rows only.
        const bool compact_tail = denoise_only && l == c.depth - 1 && !e->is_time[l] && c.depth >= 2 && S <= 16 && S >= 8;
        if (compact_tail) {
            if ((rc = gemm_simple(e->att_c, hd, a.to_out, hd, cslab(2 * l + 1), D, Mc, D, hd, 0, nullptr, e->xpool_c, D, s))) return rc;
            if ((rc = ff_block(e, e->ffp[l], e->layer_ff[l].out_b, cslab(2 * l + 1), D, cslab(2 * l + 2), D, Mc, s))) return rc;
            break;
        }
        float* h1 = slab(2 * l + 1);
        float* h2 = slab(2 * l + 2);
        if (!fused_out && (rc = gemm_c2(e->att, hd, a.to_out, hd, h1, D, M, D, hd, 0, nullptr, x_in, D, cslab(2 * l + 1), S, e->keep_hi - e->keep_lo, has_agent, s))) return rc;
        if ((rc = ff_block(e, e->ffp[l], e->layer_ff[l].out_b, h1, D, h2, D, M, s, cslab(2 * l + 2), S, has_agent))) return rc;
        if (l != c.depth - 1) {
            float* xc = (denoise_only && l == c.depth - 2 && !e->is_time[c.depth - 1] && S <= 16 && S >= 8) ? e->xpool_c : nullptr;
            if ((rc = pool_block(e, l, h2, e->xpool, 2 * l + 3, M, s, nullptr, xc, S, has_agent))) return rc;
            x_in = e->xpool;
        }
    }

    // ---- final stage on the compacted rows.  Only the ns spatial tokens (-> latent prediction) and the agent token
    // (-> heads) of each frame are consumed downstream, and every remaining op is per token, so the final attention
    // pool runs on (ns + 1) of the S rows per frame — the K/V projection of its 2*depth+1 hiddens is the largest
    // single GEMM of an evaluation.  xfc [Fr][ns+1][D]: rows 0..ns-1 spatial tokens, row ns the agent token.
    float* xfc = e->xfc;
    const float* last = slab(2 * c.depth);
    if ((rc = copy_rows(cslab(2 * c.depth), D, xfc, D, Mc, D, s))) return rc;
    if (e->encoder) {
        // the n latent (special) tokens of each frame cross-attend its P patch tokens, then their own feedforward (D4:3227-3238);
        // xfc = compact latent rows [Fr][n][D]
        const int P = e->P;
        const float* spec = cslab(2 * c.depth);
        if ((rc = gemm_simple(spec, D, e->cq_w, D, e->cq, e->ldcq, Mc, hd + h, D, GEMM_RMS_ROWSCALE, nullptr, nullptr, 0, s))) return rc;
        if ((rc = gemm_simple(last, D, e->ckv_w, D, e->ckv, 2 * hd, M, 2 * hd, D, GEMM_RMS_ROWSCALE, nullptr, nullptr, 0, s))) return rc;
        SmallAttnArgs sa{};
        sa.dh = c.attn_dim_head;
        sa.q = e->cq; sa.q_group_stride = (int64_t)n * e->ldcq; sa.q_item_stride = e->ldcq;
        sa.k = e->ckv; sa.k_group_stride = (int64_t)S * 2 * hd; sa.k_item_stride = 2 * hd;
        sa.v = e->ckv + hd; sa.v_group_stride = (int64_t)S * 2 * hd; sa.v_item_stride = 2 * hd;
        sa.gate = e->cq + hd; sa.g_group_stride = (int64_t)n * e->ldcq; sa.g_item_stride = e->ldcq;
        sa.k_gamma = e->cross.k_gamma;
        sa.out = e->catt; sa.o_group_stride = (int64_t)n * hd; sa.o_item_stride = hd;
        sa.groups = Fr; sa.heads = h; sa.nq = n; sa.nk = P;
        if ((rc = small_attn(sa, s))) return rc;
        if ((rc = gemm_simple(e->catt, hd, e->cross.to_out, hd, xfc, D, Mc, D, hd, 0, nullptr, spec, D, s))) return rc;
        if ((rc = ff_block(e, e->ffp[c.depth], e->sff.out_b, xfc, D, xfc, D, Mc, s))) return rc;
        if ((rc = pool_block(e, c.depth - 1, xfc, xfc, e->nslab, Mc, s, e->cslabs))) return rc;
        // final RMSNorm (folded) -> encoded_to_latents -> tanh                                   D4:3246, 4408, 4417
        if ((rc = gemm_simple(xfc, D, e->t2p_wf, D, e->enc_out, c.dim_latent, Mc, c.dim_latent, D, GEMM_RMS_ROWSCALE, nullptr, nullptr, 0, s))) return rc;
        return tanh_rows(e->enc_out, e->enc_out, (int64_t)Mc * c.dim_latent, s);
    }
    if (e->decoder) {
        // final attention pool on the patch rows, then final RMSNorm (folded) -> Linear(dim, channels * patch^2)   D4:3242-3246, 3555
        if ((rc = pool_block(e, c.depth - 1, xfc, xfc, e->nslab, Mc, s, e->cslabs))) return rc;
        return gemm_simple(xfc, D, e->t2p_wf, D, e->dec_out, e->dim_patch, Mc, e->dim_patch, D, GEMM_RMS_ROWSCALE, e->t2p_b, nullptr, 0, s);
    }
    if (need_agent) {
        // agent token cross-attends the non-special tokens of its frame, then its own feedforward (D4:3227-3238)
        const float* agent_in = last + (size_t)(S - 1) * D;
        float* agent_rows = xfc + (size_t)ns * D;
        const int lda = S * D, ldc = nkeep * D;
        {
            const GemmArgs gq{agent_in, lda, e->cq_w, D, e->cq, e->ldcq, nullptr, nullptr, 0, Fr, hd + h, D, GEMM_RMS_ROWSCALE, RMS_EPS};
            const GemmArgs gkv{last, D, e->ckv_w, D, e->ckv, 2 * hd, nullptr, nullptr, 0, M, 2 * hd, D, GEMM_RMS_ROWSCALE, RMS_EPS};
            if ((rc = gemm_two(gq, gkv, s))) return rc;
        }
        SmallAttnArgs sa{};
        sa.dh = c.attn_dim_head;
        sa.q = e->cq; sa.q_group_stride = e->ldcq; sa.q_item_stride = 0;
        sa.k = e->ckv; sa.k_group_stride = (int64_t)S * 2 * hd; sa.k_item_stride = 2 * hd;
        sa.v = e->ckv + hd; sa.v_group_stride = (int64_t)S * 2 * hd; sa.v_item_stride = 2 * hd;
        sa.gate = e->cq + hd; sa.g_group_stride = e->ldcq; sa.g_item_stride = 0;
        sa.k_gamma = e->cross.k_gamma;
        sa.out = e->catt; sa.o_group_stride = hd; sa.o_item_stride = 0;
        sa.groups = Fr; sa.heads = h; sa.nq = 1; sa.nk = S - 1;
        if ((rc = small_attn(sa, s))) return rc;
        if ((rc = gemm_simple(e->catt, hd, e->cross.to_out, hd, agent_rows, ldc, Fr, D, hd, 0, nullptr, agent_in, lda, s))) return rc;
        if ((rc = ff_block(e, e->ffp[c.depth], e->sff.out_b, agent_rows, ldc, agent_rows, ldc, Fr, s))) return rc;
    }
    // ---- final attention pool over every layer hidden (D4:3242-3243), compact rows
    if ((rc = pool_block(e, c.depth - 1, xfc, xfc, e->nslab, Mc, s, e->cslabs))) return rc;

    // ---- to_latent_pred: RMSNorm -> LQAP (n queries over the ns spatial tokens) -> Linear  (D4:7251)
    if (same_len) {
        // RMSNorm -> Linear on the spatial rows (gamma folded into the weight, 1/rms in the GEMM)
        if ((rc = copy_rows(xfc, nkeep * D, e->gs, ns * D, Fr, ns * D, s))) return rc;
        return gemm_simple(e->gs, D, e->lout_w, D, e->pred, dl, Fr * n, dl, D, GEMM_RMS_ROWSCALE, nullptr, nullptr, 0, s);
    }
    if ((rc = gather_space_double_norm(xfc, e->gs, e->latent_norm, e->lq_out.norm_ctx, Fr, nkeep, 0, D, ns, RMS_EPS, s))) return rc;
    if ((rc = gemm_simple(e->gs, D, e->lout_kv_w, D, e->okv, 2 * hd, Fr * ns, 2 * hd, D, 0, nullptr, nullptr, 0, s))) return rc;
    {
        SmallAttnArgs sa{};
        sa.dh = c.attn_dim_head;
        sa.q = e->lout_q; sa.q_group_stride = 0; sa.q_item_stride = hd;
        sa.k = e->okv; sa.k_group_stride = (int64_t)ns * 2 * hd; sa.k_item_stride = 2 * hd;
        sa.v = e->okv + hd; sa.v_group_stride = (int64_t)ns * 2 * hd; sa.v_item_stride = 2 * hd;
        sa.gate = e->lout_gate; sa.g_group_stride = 0; sa.g_item_stride = h;
        sa.k_gamma = e->lq_out.k_gamma;
        sa.out = e->oatt; sa.o_group_stride = (int64_t)n * hd; sa.o_item_stride = hd;
        sa.groups = Fr; sa.heads = h; sa.nq = n; sa.nk = ns;
        if ((rc = small_attn(sa, s))) return rc;
    }
    if ((rc = gemm_simple(e->oatt, hd, e->lout_w, hd, e->pred, dl, Fr * n, dl, hd, 0, nullptr, nullptr, 0, s))) return rc;
    return 0;
}

// normed MLP head (recipe: engine.h).  `save` (optional): per layer [x | xhat | z] for the learner's backward.
static constexpr float LN_EPS = 1e-5f;      // nn.LayerNorm default

int mlp_forward(d4_engine* e, const Mlp& m, const float* x, int ldx, int rows, float* out, int ldo,
                float* save, hipStream_t s) {
    int rc;
    const float* cur = x;
    int ld = ldx;
    const bool pre = m.recipe == D4_MLP_PRE_RMS;
    for (int i = 0; i < m.nl; ++i) {
        const int din = m.dims[i], dout = m.dims[i + 1];
        const bool last = i == m.nl - 1;
        const bool post = m.post_norm(i);
        if (save) {
            // learner path (rows may exceed max_batch: only the save area is used, never hbuf / hnorm):
            // x_i -> sx[i], (pre-norm: xhat_i -> sxh[i]), linear output -> sz[i]; the next layer's input is written straight
            // into sx[i+1] (silu(z) or silu(LayerNorm(z))).
            float *sx, *sxh, *sz;
            m.save_ptrs(save, rows, i, &sx, &sxh, &sz);
            const int ldz = m.ldz(i);
            if (i == 0 && (rc = copy_rows(x, ldx, sx, din, rows, din, s))) return rc;
            const float* lin_in = sx;
            if (pre) {
                if ((rc = rmsnorm_rows(sx, din, m.g[i], sxh, din, rows, din, RMS_EPS, s))) return rc;
                lin_in = sxh;
            }
            if (ldz != dout && (rc = fill_f32(sz, 0.f, (int64_t)rows * ldz, s))) return rc;
            if ((rc = gemm_simple(lin_in, din, m.w[i], din, sz, ldz, rows, dout, din, 0, m.b[i], nullptr, 0, s))) return rc;
            if (last) { if ((rc = copy_rows(sz, ldz, out, ldo, rows, dout, s))) return rc; }
            else {
                float *nx, *nxh, *nz;
                m.save_ptrs(save, rows, i + 1, &nx, &nxh, &nz);
                if (post) { if ((rc = layernorm_rows(sz, ldz, m.g[i], m.nb[i], nx, dout, rows, dout, LN_EPS, 1, s))) return rc; }
                else if ((rc = silu_rows(sz, nx, (int64_t)rows * dout, s))) return rc;
            }
            continue;
        }
        float* y = last ? out : e->hbuf[i & 1];
        const int ldy = last ? ldo : dout;
        const float* lin_in = cur;
        int ld_in = ld;
        if (pre) {
            if ((rc = rmsnorm_rows(cur, ld, m.g[i], e->hnorm, din, rows, din, RMS_EPS, s))) return rc;
            lin_in = e->hnorm; ld_in = din;
        }
        const int act = (!last && !post) ? 1 : 0;          // SiLU fused into the linear's epilogue unless a LayerNorm sits in between
        // ONE launch per layer (bias + SiLU in the epilogue).  Rounds 1-4 sliced K over the grid and combined the slices in a reduce kernel (few rows x
        // long K on 64 x 64 tiles); with the 32 x 32 tiles of the LDS-DMA family the direct form is level on the 2048 x 2048 layers (30.1 us vs 22.8 + a
        // 5 us reduce + a boundary) and ahead on the first and last layers (K = 512: 9.7 vs 13.7 us) — tools/splitk_probe.py — so the reduce launches are gone
        if ((rc = gemm_simple(lin_in, ld_in, m.w[i], din, y, ldy, rows, dout, din, act ? GEMM_SILU : 0, m.b[i], nullptr, 0, s))) return rc;
        if (post && (rc = layernorm_rows(y, ldy, m.g[i], m.nb[i], y, ldy, rows, dout, LN_EPS, 1, s))) return rc;
        cur = y; ld = ldy;
    }
    return 0;
}

}  // namespace d4

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* d4_last_error(void) { return d4::last_error(); }
int d4_version(void) { return 2; }

int d4_engine_create(const d4_config* cfg, d4_engine** out) {
    D4_REQUIRE(cfg && out, "null argument");
    const d4_config& c = *cfg;
    D4_REQUIRE(c.attn_dim_head == 16 || c.attn_dim_head == 32 || c.attn_dim_head == 64,
               "attn_dim_head=%d: 16, 32 or 64 (a head row lives in one wavefront) is implemented", c.attn_dim_head);
    D4_REQUIRE(c.pool_dim_head == 64, "pool_dim_head=%d: only 64 is implemented", c.pool_dim_head);
    D4_REQUIRE(c.dim % 4 == 0 && c.dim_latent % 4 == 0, "dim and dim_latent must be multiples of 4");
    D4_REQUIRE(c.depth >= 1 && c.time_block_every >= 1, "bad depth/time_block_every");
    D4_REQUIRE(c.num_discrete_action_types >= 0 && c.num_discrete_action_types <= D4_MAX_ACTION_TYPES, "too many action types");
    const bool encoder = c.mode == D4_MODE_ENCODER;
    const bool decoder = c.mode == D4_MODE_DECODER || encoder;          // (shared geometry checks below; e->decoder is set for the decoder only)
    D4_REQUIRE(c.mode == D4_MODE_DYNAMICS || decoder, "unknown engine mode %d", c.mode);
    D4_REQUIRE(c.num_latent_tokens <= 64 && (decoder || c.num_spatial_tokens <= 64), "at most 64 latent / spatial tokens");
    if (decoder) {
        D4_REQUIRE(c.attn_dim_head == 64, "decoder mode: attn_dim_head must be 64 (the wide attention kernel)");
        D4_REQUIRE(c.patch_size >= 1 && c.channels >= 1 && c.image_height % c.patch_size == 0 && c.image_width % c.patch_size == 0 && c.image_height > 0 && c.image_width > 0,
                   "decoder mode: image %d x %d must be a positive multiple of the patch size %d", c.image_height, c.image_width, c.patch_size);
        D4_REQUIRE((c.channels * c.patch_size * c.patch_size) % 4 == 0, "decoder mode: channels * patch_size^2 must be a multiple of 4");
        D4_REQUIRE(encoder || c.decoder_flow_steps >= 1, "decoder mode: decoder_flow_steps >= 1 (the flow decoder is the reference's default, D4:3875)");
        D4_REQUIRE(c.num_discrete_action_types == 0 && c.num_continuous_actions == 0 && c.matmul_bf16 == 0, "decoder mode: no actions / bf16");
        D4_REQUIRE(c.decoder_pos_mlp_depth >= 0 && c.decoder_pos_mlp_depth <= 6, "decoder_pos_mlp_depth out of range");
    }
    D4_REQUIRE((c.max_steps & (c.max_steps - 1)) == 0, "max_steps must be a power of two");
    D4_REQUIRE(c.policy_head_mlp_depth <= 6 && c.value_head_mlp_depth <= 6 && c.terminal_mlp_depth <= 6, "mlp depth > 6");
    d4_engine* e = new d4_engine();
    e->c = c;
    e->D = c.dim;
    e->S = 1 + c.num_spatial_tokens + c.num_register_tokens + ((c.num_discrete_action_types > 0 || c.num_continuous_actions > 0) ? 1 : 0) + 1;   // no action token without an action space
    e->keep_lo = 1; e->keep_hi = 1 + c.num_spatial_tokens;
    e->decoder = decoder && !encoder;
    e->encoder = encoder;
    if (decoder) {
        e->nph = c.image_height / c.patch_size; e->npw = c.image_width / c.patch_size;
        e->P = e->nph * e->npw;
        e->dim_patch = c.channels * c.patch_size * c.patch_size;
        e->S = e->P + c.num_latent_tokens;                     // [patches | latent tokens]; the last latent token is the trunk's one special token
        e->keep_lo = 0; e->keep_hi = e->P;
        if (encoder) { e->keep_lo = e->P; e->keep_hi = e->P + c.num_latent_tokens; }       // the latent (special) tokens are the encoder's output rows
        D4_REQUIRE(e->S - 1 <= 160, "decoder mode: %d tokens per frame exceed the wide attention kernel's 160", e->S - 1);
    }
    D4_REQUIRE((decoder || e->S <= 64) && 2 * c.depth + 1 <= 64, "tokens per frame / pooled hiddens exceed 64");
    e->hd = c.attn_heads * c.attn_dim_head;
    e->php = c.pool_heads;
    e->hp = c.pool_heads * 64;
    e->Nproj = 3 * e->hd + 2 * c.attn_heads;
    e->Nproj = (e->Nproj + 3) / 4 * 4;
    e->Nproj0 = 3 * e->hd + 2 * c.attn_heads + e->hd;
    e->Nproj0 = (e->Nproj0 + 3) / 4 * 4;
    e->inner = (int)((double)c.dim * 4 * 2 / 3);               // int(dim * 4 * 2 / 3)  D4:2094
    e->inner_pad = (e->inner + 31) / 32 * 32;
    e->ldpq = (e->hp + e->php + 3) / 4 * 4;
    e->ldcq = (e->hd + c.attn_heads + 3) / 4 * 4;
    e->nslab = 2 * c.depth + 1;
    e->na = c.num_discrete_action_types;
    e->nc = c.num_continuous_actions;
    e->bf16 = c.matmul_bf16 != 0;
    e->split = c.matmul_bf16 == 2;
    e->h2 = c.matmul_bf16 == 3;
    D4_REQUIRE(e->nc >= 0 && e->nc <= 64, "num_continuous_actions out of range");
    e->A = 0;
    for (int a = 0; a < e->na; ++a) e->A += c.num_discrete_actions[a];
    e->is_time.assign(c.depth, 0);
    e->time_index.assign(c.depth, -1);
    e->Lt = 0;
    for (int l = 0; l < c.depth; ++l)
        if ((l + 1) % c.time_block_every == 0) { e->is_time[l] = 1; e->time_index[l] = e->Lt++; }
    e->maxB = c.max_batch > 0 ? c.max_batch : 1;
    e->maxTq = c.max_parallel_frames > 0 ? c.max_parallel_frames : 1;
    e->Tcap = c.max_frames > e->maxTq ? c.max_frames : e->maxTq;
    e->Fr = e->maxB * e->maxTq;
    e->Mmax = e->Fr * e->S;
    D4_REQUIRE(c.reward_encoder_type == 0 || c.reward_encoder_type == 1, "unknown reward_encoder_type %d", c.reward_encoder_type);
    D4_REQUIRE(c.head_mlp_recipe == D4_MLP_PRE_RMS || c.head_mlp_recipe == D4_MLP_POST_LAYER, "unknown head_mlp_recipe %d", c.head_mlp_recipe);
    D4_REQUIRE(c.continuous_beta_param == D4_BETA_SOFTPLUS_P1 || c.continuous_beta_param == D4_BETA_EXP_P1, "unknown continuous_beta_param %d", c.continuous_beta_param);
    d4::mlp_dims(e->policy, c.dim, 4 * c.dim, 4 * c.dim, c.policy_head_mlp_depth, c.head_mlp_recipe);
    d4::mlp_dims(e->value, c.dim, 4 * c.dim, c.value_num_bins, c.value_head_mlp_depth, c.head_mlp_recipe);
    d4::mlp_dims(e->terminal, c.dim_latent, 4 * c.dim_latent, 1, c.terminal_mlp_depth, c.head_mlp_recipe);
    d4::mlp_dims(e->posmlp, 2, 2 * c.dim, c.dim, (decoder && !encoder) ? c.decoder_pos_mlp_depth : 0, c.head_mlp_recipe);      // D4:3526-3532
    if (const char* gm = getenv("D4_GRAPH_MAX_ROWS")) e->graph_max_rows = atoi(gm);     // 0 disables graph replay
    d4::engine_layout(e, false);
    *out = e;
    return 0;
}

void d4_engine_destroy(d4_engine* e) {
    if (e) {
        d4::engine_drop_graphs(e);
        if (e->capture_stream) (void)hipStreamDestroy(e->capture_stream);
    }
    delete e;
}

size_t d4_engine_workspace_bytes(const d4_engine* e) { return e ? e->ws_need : 0; }

int d4_engine_set_workspace(d4_engine* e, void* p, size_t bytes) {
    D4_REQUIRE(e && p, "null argument");
    D4_REQUIRE(bytes >= e->ws_need, "workspace too small: %zu < %zu bytes", bytes, e->ws_need);
    D4_REQUIRE(((uintptr_t)p % 256) == 0, "workspace must be 256-byte aligned");
    d4::engine_drop_graphs(e);
    e->ws = static_cast<char*>(p);
    e->ws_bytes = bytes;
    e->prepared = false;
    e->cache_frames = 0;
    return d4::engine_layout(e, true);
}

int d4_engine_bind(d4_engine* e, const char* key, const float* p, float* grad, int64_t numel) {
    D4_REQUIRE(e && key && p, "null argument");
    D4_REQUIRE(((uintptr_t)p % 16) == 0, "tensor '%s' is not 16-byte aligned", key);
    e->bound[key] = d4::Bound{p, grad, numel};
    e->prepared = false;
    d4::engine_drop_graphs(e);          // captured graphs bake weight addresses
    return 0;
}

int d4_engine_prepare(d4_engine* e, void* stream) {
    D4_REQUIRE(e, "null engine");
    return d4::engine_prepare(e, static_cast<hipStream_t>(stream));
}

int d4_engine_cache_frames(const d4_engine* e) { return e ? e->cache_frames : -1; }

int d4_engine_cache_reset(d4_engine* e, int frames) {
    // the K/V of frame t sit in slot t of the ring whatever the counter says: the caller may rewind, or move forward again over
    // slots it knows to be intact (dreamer4_amd.world_model.TimeCache keeps that book)
    D4_REQUIRE(e && frames >= 0 && frames <= e->Tcap, "cache_reset: bad frame count %d (capacity %d)", frames, e ? e->Tcap : -1);
    e->cache_frames = frames;
    return 0;
}

int d4_engine_cache_export(d4_engine* e, float* dst, int batch, void* stream) {
    D4_REQUIRE(e && dst, "null argument");
    return d4::cache_transfer(e->cache, dst, e->Lt, e->maxB, batch, e->S, e->c.attn_heads, e->Tcap, e->cache_frames, 1, e->c.attn_dim_head,
                              static_cast<hipStream_t>(stream));
}

int d4_engine_cache_import(d4_engine* e, const float* src, int batch, int frames, void* stream) {
    D4_REQUIRE(e && src, "null argument");
    D4_REQUIRE(frames <= e->Tcap && batch <= e->maxB, "cache_import: exceeds capacity");
    int rc = d4::cache_transfer(e->cache, const_cast<float*>(src), e->Lt, e->maxB, batch, e->S, e->c.attn_heads, e->Tcap, frames, 0, e->c.attn_dim_head,
                                static_cast<hipStream_t>(stream));
    if (!rc) e->cache_frames = frames;
    return rc;
}

static int step_log2_of(int step_size, int* out) {
    D4_REQUIRE(step_size >= 1 && (step_size & (step_size - 1)) == 0, "step_sizes must be powers of 2 (got %d)  [D4:6944]", step_size);
    int l = 0;
    while ((1 << l) < step_size) ++l;
    *out = l;
    return 0;
}

static int check_step_log2(const d4_engine* e, int sl) {
    int nlog = 0;
    while ((1 << nlog) < e->c.max_steps) ++nlog;
    // step_size_embed has log2(max_steps) rows (D4:4896): the reference raises IndexError for step_size == max_steps
    D4_REQUIRE(sl < nlog, "step size 2^%d has no step_size_embed row (max_steps = %d allows at most 2^%d)", sl, e->c.max_steps, nlog - 1);
    return 0;
}

int d4_wm_forward(d4_engine* e, const float* latents, const int32_t* signal_levels, int step_size,
                  const int64_t* prev_actions, const float* prev_cont, const int64_t* tasks, int batch, int frames,
                  int use_cache, int commit_cache, float* pred, float* agent_embed, void* stream) {
    D4_REQUIRE(e && latents && signal_levels, "null argument");
    D4_REQUIRE(!e->decoder && !e->encoder, "d4_wm_forward needs a dynamics engine");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int sl, rc;
    if ((rc = step_log2_of(step_size, &sl))) return rc;
    if ((rc = check_step_log2(e, sl))) return rc;
    D4_REQUIRE(batch <= e->maxB && frames <= e->maxTq, "batch/frames exceed engine capacity");
    const int Fr = batch * frames;
    D4_HIP(hipMemcpyAsync(e->sig, signal_levels, sizeof(int32_t) * Fr, hipMemcpyDeviceToDevice, s));
    if (e->na > 0) {
        if (prev_actions) D4_HIP(hipMemcpyAsync(e->pact, prev_actions, sizeof(int64_t) * Fr * e->na, hipMemcpyDeviceToDevice, s));
        else D4_HIP(hipMemsetAsync(e->pact, 0xFF, sizeof(int64_t) * Fr * e->na, s));     // -1 => zero token
    }
    if (e->nc > 0) {
        if (prev_cont) D4_HIP(hipMemcpyAsync(e->pcont, prev_cont, sizeof(float) * Fr * e->nc, hipMemcpyDeviceToDevice, s));
        else if (e->na > 0) D4_HIP(hipMemsetAsync(e->pcont, 0, sizeof(float) * Fr * e->nc, s));      // validity comes from the discrete side
        else D4_HIP(hipMemsetAsync(e->pcont, 0xFF, sizeof(float) * Fr * e->nc, s));                    // NaN => zero token
    }
    const int t0 = use_cache ? e->cache_frames : 0;
    if ((rc = d4::engine_forward(e, latents, batch, frames, t0, sl, tasks, true, s))) return rc;
    if (commit_cache) e->cache_frames = t0 + frames;
    const int n_el = e->c.num_latent_tokens * e->c.dim_latent;
    if (pred && (rc = d4::copy_rows(e->pred, n_el, pred, n_el, Fr, n_el, s))) return rc;
    const int nkeep = e->c.num_spatial_tokens + 1;
    if (agent_embed && (rc = d4::copy_rows(e->xfc + (size_t)(nkeep - 1) * e->D, nkeep * e->D, agent_embed, e->D, Fr, e->D, s))) return rc;
    return 0;
}

int d4_decoder_forward(d4_engine* e, const float* latents, const float* noised_video, int time_index, int batch, int frames,
                       float* pred_video, void* stream) {
    D4_REQUIRE(e && latents && noised_video && pred_video, "null argument");
    D4_REQUIRE(e->decoder, "d4_decoder_forward needs an engine created with mode = D4_MODE_DECODER");
    D4_REQUIRE(batch >= 1 && batch <= e->maxB && frames >= 1 && frames <= e->maxTq, "batch / frames exceed the decoder engine's capacity (%d x %d)", e->maxB, e->maxTq);
    D4_REQUIRE(time_index >= 0 && time_index < e->c.decoder_flow_steps, "flow step %d outside [0, %d)", time_index, e->c.decoder_flow_steps);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const d4_config& c = e->c;
    int rc;
    if ((rc = d4::video_to_patches(noised_video, e->dec_in, batch, c.channels, frames, e->nph, e->npw, c.patch_size, s))) return rc;
    if ((rc = d4::engine_forward(e, latents, batch, frames, 0, time_index, nullptr, false, s))) return rc;
    return d4::patches_to_video(e->dec_out, pred_video, batch, c.channels, frames, e->nph, e->npw, c.patch_size, s);
}

int d4_encoder_forward(d4_engine* e, const float* video, int batch, int frames, float* latents, void* stream) {
    D4_REQUIRE(e && video && latents, "null argument");
    D4_REQUIRE(e->encoder, "d4_encoder_forward needs an engine created with mode = D4_MODE_ENCODER");
    D4_REQUIRE(batch >= 1 && batch <= e->maxB && frames >= 1 && frames <= e->maxTq, "batch / frames exceed the encoder engine's capacity (%d x %d)", e->maxB, e->maxTq);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const d4_config& c = e->c;
    int rc;
    if ((rc = d4::video_to_patches(video, e->dec_in, batch, c.channels, frames, e->nph, e->npw, c.patch_size, s))) return rc;
    if ((rc = d4::engine_forward(e, nullptr, batch, frames, 0, 0, nullptr, false, s))) return rc;
    const int nl = c.num_latent_tokens * c.dim_latent;
    return d4::copy_rows(e->enc_out, nl, latents, nl, batch * frames, nl, s);
}

int d4_euler_step(float* x, const float* pred, int64_t n, float one_minus_t, float dt, void* stream) {
    D4_REQUIRE(x && pred && n >= 0 && n < (int64_t)1 << 31, "euler_step: bad arguments");
    return d4::euler_step(x, (int)n, pred, (int)n, 1, (int)n, one_minus_t, dt, static_cast<hipStream_t>(stream));
}

int d4_rollout(d4_engine* e, const d4_rollout_io* io, void* stream) {
    D4_REQUIRE(e && io, "null argument");
    D4_REQUIRE(!e->decoder && !e->encoder, "d4_rollout needs a dynamics engine");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const d4_config& c = e->c;
    const int B = io->batch, T = io->time_steps, P = io->prompt_frames, K = io->num_steps;
    D4_REQUIRE(B >= 1 && B <= e->maxB, "batch %d exceeds max_batch %d", B, e->maxB);
    D4_REQUIRE(K >= 1 && (K & (K - 1)) == 0 && K <= c.max_steps, "number of steps %d must be a power of 2 in (0, %d]  [D4:6357-6358]", K, c.max_steps);
    D4_REQUIRE(P >= 0 && P <= T, "bad prompt_frames");
    D4_REQUIRE(!io->sample_actions || e->na > 0 || e->nc > 0, "the model has no actions to sample  [D4:6626]");
    D4_REQUIRE(!io->sample_actions || e->na == 0 || (io->actions && io->gumbel_u && io->log_probs && io->action_logits), "sample_actions needs the action outputs and gumbel_u");
    D4_REQUIRE(!io->sample_actions || e->nc == 0 || (io->actions_cont && io->beta_noise && io->log_probs_cont && io->cont_params), "sample_actions needs the continuous action outputs and beta_noise");
    D4_REQUIRE(!io->sample_actions || io->values, "sample_actions needs the values output");
    D4_REQUIRE(!io->sample_terminals || (c.predict_terminals && io->bern_u), "sample_terminals needs predict_terminals and bern_u");
    D4_REQUIRE(io->use_time_cache || io->noise_context || T - P <= 1, "noise_context is required without the time cache");
    const int F = T - P;
    const int step_size = c.max_steps / K;
    int sl, rc;
    if ((rc = step_log2_of(step_size, &sl))) return rc;
    if ((rc = check_step_log2(e, sl))) return rc;
    const int n_el = c.num_latent_tokens * c.dim_latent;
    const int D = e->D, S = e->S, A = e->A, na = e->na, nc = e->nc;
    if (!io->use_time_cache) e->cache_frames = 0;

    for (int f = 0; f < F; ++f) {
        const int cur = P + f;                       // frames of history in this call
        int Tq, t0;
        bool commit;
        if (io->use_time_cache && (e->cache_frames > 0 || cur == 0)) { Tq = 1; t0 = e->cache_frames; commit = true; }
        else if (io->use_time_cache) { Tq = cur + 1; t0 = 0; commit = true; }
        else { Tq = cur + 1; t0 = 0; commit = false; }
        D4_REQUIRE(Tq <= e->maxTq, "a parallel pass over %d frames exceeds max_parallel_frames %d", Tq, e->maxTq);
        if ((rc = d4::copy_rows(io->noise_latent + (size_t)f * B * n_el, n_el, e->x_lat, n_el, B, n_el, s))) return rc;

        // Launch-bound regime (small batch, cached decode): the K+1 evaluations of a frame are replayed from a captured
        // hipGraph; everything frame dependent they read (rotary / cache position) comes from device memory.
        const bool graphable = Tq == 1 && B * S <= e->graph_max_rows && !d4::gemm_profile_active() && e->graph_max_rows > 0;
        if (graphable) {
            D4_REQUIRE(t0 + 1 <= e->Tcap || e->Lt == 0, "KV cache capacity %d exceeded (%d + 1 frames)", e->Tcap, t0);
            if ((rc = d4::set_frame_state(e->fstate, t0, s))) return rc;
            if ((rc = d4::prep_eval_inputs(e->sig, e->pact, io->actions, B, 1, na, cur, T, 0, c.max_steps - 1, s, nc > 0 ? e->pcont : nullptr, io->actions_cont, nc))) return rc;
            const int64_t* tasks = nullptr;
            if (io->tasks) {
                D4_HIP(hipMemcpyAsync(e->tasks_dev, io->tasks, sizeof(int64_t) * B, hipMemcpyDeviceToDevice, s));
                tasks = e->tasks_dev;
            }
            hipGraphExec_t exec = nullptr;
            const int bucket = e->Lt > 0 ? d4::time_history_bucket(t0) : 0;
            for (auto& g : e->graphs) if (g.B == B && g.K == K && g.sl == sl && g.tasks == (tasks != nullptr) && g.bucket == bucket) exec = g.exec;
            if (!exec) {
                if (!e->warm) {
                    // first decode frame of this engine: run it eagerly once (function attributes, lazy module load)
                    e->warm = true;
                } else {
                    // capture on an engine-owned stream (the caller's may be the legacy default stream, which cannot
                    // capture); the instantiated graph is then launched on the caller's stream
                    if (!e->capture_stream) D4_HIP(hipStreamCreateWithFlags(&e->capture_stream, hipStreamNonBlocking));
                    hipStream_t cs = e->capture_stream;
                    hipGraph_t graph;
                    D4_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
                    int crc = 0;
                    for (int step = 0; step <= K && !crc; ++step) {
                        const int sig_val = step * step_size < c.max_steps - 1 ? step * step_size : c.max_steps - 1;
                        d4::t_sig_uniform = sig_val;
                        crc = d4::engine_forward(e, e->x_lat, B, 1, t0, sl, tasks, step == K, cs, e->fstate);
                        d4::t_sig_uniform = -1;
                        if (crc) break;
                        if (step == K) break;
                        const float tt = (float)sig_val / (float)c.max_steps;
                        crc = d4::euler_step(e->x_lat, n_el, e->pred, n_el, B, n_el, 1.f - tt, (float)step_size / (float)c.max_steps, cs);
                    }
                    hipError_t ce = hipStreamEndCapture(cs, &graph);
                    if (crc) { if (ce == hipSuccess) (void)hipGraphDestroy(graph); return crc; }
                    D4_HIP(ce);
                    D4_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                    (void)hipGraphDestroy(graph);
                    e->graphs.push_back({B, K, sl, tasks != nullptr, bucket, exec});
                }
            }
            if (exec) {
                D4_HIP(hipGraphLaunch(exec, s));
            } else {
                for (int step = 0; step <= K; ++step) {
                    const int sig_val = step * step_size < c.max_steps - 1 ? step * step_size : c.max_steps - 1;
                    d4::t_sig_uniform = sig_val;
                    rc = d4::engine_forward(e, e->x_lat, B, 1, t0, sl, tasks, step == K, s, e->fstate);
                    d4::t_sig_uniform = -1;
                    if (rc) return rc;
                    if (step == K) break;
                    const float tt = (float)sig_val / (float)c.max_steps;
                    if ((rc = d4::euler_step(e->x_lat, n_el, e->pred, n_el, B, n_el, 1.f - tt, (float)step_size / (float)c.max_steps, s))) return rc;
                }
            }
            if (commit) e->cache_frames = t0 + 1;
        } else
        for (int step = 0; step <= K; ++step) {
            const bool last = step == K;
            const int sig_val = step * step_size < c.max_steps - 1 ? step * step_size : c.max_steps - 1;   // D4:6492
            const float* lat = e->x_lat;
            if (Tq > 1) {
                // context frames noised with their fixed noise; prompt frames carry noise == themselves (D4:6400, 6497)
                if ((rc = d4::build_latent_input(e->lat_in, io->latents, io->ctx_hist, e->x_lat, B, Tq, n_el, T, io->context_signal_noise, s))) return rc;
                lat = e->lat_in;
            }
            if ((rc = d4::prep_eval_inputs(e->sig, e->pact, io->actions, B, Tq, na, cur + 1 - Tq, T, sig_val, c.max_steps - 1, s, nc > 0 ? e->pcont : nullptr, io->actions_cont, nc))) return rc;
            if ((rc = d4::engine_forward(e, lat, B, Tq, t0, sl, io->tasks, last, s))) return rc;
            if (last) { if (commit) e->cache_frames = t0 + Tq; break; }
            const float tt = (float)sig_val / (float)c.max_steps;
            if ((rc = d4::euler_step(e->x_lat, n_el, e->pred + (size_t)(Tq - 1) * n_el, Tq * n_el, B, n_el,
                                     1.f - tt, (float)step_size / (float)c.max_steps, s))) return rc;
        }

        // ---- heads on the agent embedding of the clean step (D4:6595-6662)
        const int nkeep = c.num_spatial_tokens + 1;
        const float* agent_row = e->xfc + ((size_t)(Tq - 1) * nkeep + (nkeep - 1)) * D;
        if ((rc = d4::copy_rows(agent_row, Tq * nkeep * D, e->agent_c, D, B, D, s))) return rc;
        if (io->agent_embed && (rc = d4::copy_rows(e->agent_c, D, io->agent_embed + (size_t)f * D, F * D, B, D, s))) return rc;
        // reward: Ensemble member 0 of [RMSNorm -> Linear]                     D4:6598-6601
        if ((rc = d4::rmsnorm_rows(e->agent_c, D, e->reward_norm, e->hnorm, D, B, D, d4::RMS_EPS, s))) return rc;
        if ((rc = d4::gemm_simple(e->hnorm, D, e->reward_w, D, e->rlogits, c.reward_num_bins, B, c.reward_num_bins, D, 0, nullptr, nullptr, 0, s))) return rc;
        if ((rc = d4::hl_gauss_scalar(e->rlogits, c.reward_num_bins, e->reward_centers, io->rewards + cur, T, B, c.reward_num_bins, s))) return rc;
        // terminal logit from the mean-pooled denoised latent                    D4:6605-6607
        const float* term_logit = nullptr;
        if (io->sample_terminals) {
            if ((rc = d4::mean_tokens(e->x_lat, e->term_pool, B, c.num_latent_tokens, c.dim_latent, s))) return rc;
            if ((rc = d4::mlp_forward(e, e->terminal, e->term_pool, c.dim_latent, B, e->term_logit, 1, nullptr, s))) return rc;
            term_logit = e->term_logit;
        }
        if (io->sample_actions) {
            // policy embed -> logits of prediction head 0                            D4:6628-6643
            float* pe = e->hbuf[(e->policy.nl - 1) & 1];   // the buffer mlp_forward's last hidden does not occupy
            if ((rc = d4::mlp_forward(e, e->policy, e->agent_c, D, B, pe, 4 * D, nullptr, s))) return rc;
            float* logits = na > 0 ? io->action_logits + (size_t)f * A : nullptr;
            if (na > 0 && (rc = d4::gemm_simple(pe, 4 * D, e->action_unembed, c.multi_token_pred_len * 4 * D, logits, F * A, B, A, 4 * D, 0, nullptr, nullptr, 0, s))) return rc;
            float* cparams = nc > 0 ? io->cont_params + (size_t)f * nc * 2 : nullptr;
            if (nc > 0) {
                // raw Beta parameters of prediction head 0: the [nc][mtp][4D][2] parameter is read through a K-contiguous copy of
                // its head-0 slice (a trained head changes every optimiser step, so the copy is refreshed per frame: 2 nc x 4D floats)
                if ((rc = d4::cunembed_gather(e->cont_unembed, e->cu_w, nc, c.multi_token_pred_len, 4 * D, s))) return rc;
                if ((rc = d4::gemm_simple(pe, 4 * D, e->cu_w, 4 * D, cparams, F * nc * 2, B, 2 * nc, 4 * D, 0, nullptr, nullptr, 0, s))) return rc;
            }
            // value                                                                    D4:6659-6662
            float* vb = e->hbuf[(e->value.nl - 1) & 1];
            if ((rc = d4::mlp_forward(e, e->value, e->agent_c, D, B, vb, c.value_num_bins, nullptr, s))) return rc;
            if ((rc = d4::hl_gauss_scalar(vb, c.value_num_bins, e->value_centers, io->values + f, F, B, c.value_num_bins, s))) return rc;
            // sample action, log-prob, terminal                                        D4:6611-6616, 6637-6657
            d4::SampleArgs sa{};
            sa.logits = logits; sa.ld = F * A;
            sa.gumbel_u = na > 0 ? io->gumbel_u + (size_t)f * B * A : nullptr; sa.ld_u = A;
            sa.term_logit = term_logit; sa.bern_u = io->sample_terminals ? io->bern_u + (size_t)f * B : nullptr;
            sa.actions = na > 0 ? io->actions + (size_t)cur * na : nullptr; sa.act_stride = T * na;
            sa.log_probs = na > 0 ? io->log_probs + (size_t)f * na : nullptr; sa.lp_stride = F * na;
            sa.terminals = io->terminals; sa.lens = io->lens; sa.action_sizes = e->action_sizes;
            sa.B = B; sa.na = na; sa.frame_index = cur; sa.temperature = io->discrete_temperature;
            if (nc > 0) {
                sa.nc = nc; sa.cont_params = cparams; sa.ld_c = F * nc * 2;
                sa.beta_noise = io->beta_noise + (size_t)f * B * nc * 4 * 6;
                sa.actions_cont = io->actions_cont + (size_t)cur * nc; sa.actc_stride = T * nc;
                sa.log_probs_cont = io->log_probs_cont + (size_t)f * nc; sa.lpc_stride = F * nc;
                sa.cont_temperature = io->continuous_temperature; sa.beta_param = e->c.continuous_beta_param;
            }
            if ((rc = d4::sample_actions_terminals(sa, s))) return rc;
        } else if (term_logit) {
            d4::SampleArgs sa{};
            sa.term_logit = term_logit; sa.bern_u = io->bern_u + (size_t)f * B;
            sa.terminals = io->terminals; sa.lens = io->lens; sa.B = B; sa.na = 0; sa.frame_index = cur;
            if ((rc = d4::sample_actions_terminals(sa, s))) return rc;
        }
        // history
        if ((rc = d4::copy_rows(e->x_lat, n_el, io->latents + (size_t)cur * n_el, T * n_el, B, n_el, s))) return rc;
        if (io->ctx_hist && io->noise_context &&
            (rc = d4::copy_rows(io->noise_context + (size_t)f * B * n_el, n_el, io->ctx_hist + (size_t)cur * n_el, T * n_el, B, n_el, s))) return rc;
    }
    return 0;
}

int d4_profile_bf16_enable(int stride) { d4::gemm_bf16_profile_enable(stride); return 0; }
int d4_profile_bf16_read(double* ms, double* flops, int64_t* count) { return d4::gemm_bf16_profile_read(ms, flops, count); }
int d4_profile_enable(int on) { return d4::gemm_profile_enable(on); }
int d4_profile_read(double* ms, double* flops, int64_t* count, int nclass) { return d4::gemm_profile_read(ms, flops, count, nclass); }
int d4_profile_classes(void) { return d4::gemm_profile_classes(); }
int d4_profile_glue_enable(int mask) { return d4::glue_profile_enable(mask); }
int d4_profile_glue_read(double* ms, double* bytes, int64_t* count, int nclass) { return d4::glue_profile_read(ms, bytes, count, nclass); }
int d4_profile_glue_classes(void) { return d4::GL_N; }
int d4_profile_glue_read_flops(double* flops, int nclass) { return d4::glue_profile_read_flops(flops, nclass); }
const char* d4_profile_glue_class_name(int c) { return d4::glue_class_name(c); }
int d4_frame_fused_set(int mode) { return d4::frame_fused_set(mode); }

int d4_debug_switch(const char* name, int value) {
    int* sw = nullptr;
    if (name && !strcmp(name, "time_attn_fused_append")) sw = &d4::g_time_attn_fused_append;
    else if (name && !strcmp(name, "attn_out_cols")) sw = &d4::g_attn_out_cols;
    else if (name && !strcmp(name, "pool_wide_keys")) sw = &d4::g_pool_wide_keys;
    if (!sw) return -1;
    const int old = *sw;
    *sw = value;
    return old;
}
int d4_gemm_force_config(int id) {
    // one family is forced at a time; every call first clears the hooks of the other families, and returns the number of configurations of the family
    // it addressed (id < 0: everything back to the rules / the tuner, returns the first family's count)
    d4::gemm_bf16a_force_config(-1);
    d4::gemm_bf16_force_config(-1);
    const int n_main = d4::gemm_force_config(-1);
    if (id < 0) return n_main;
    if (id >= 500) { d4::gemm_bf16a_force_config(id - 500); return d4::gemm_bf16a_configs(); }     // tile c of the bf16-activation kernel (gemm_bf16a.hip)
    if (id >= 400) { d4::gemm_force_config(id); return d4::gemm_h2_configs(); }                     // tile c of the fp16x2 family (gemm_h2.hip)
    if (id >= 300) { d4::gemm_force_config(id); return d4::gemm_x3_configs(); }                     // tile c of the split-operand fp32 family (gemm_x3.hip)
    if (id >= 200) return d4::gemm_bf16_force_config(id - 200);                                     // configuration c of the fp32-activation bf16 kernel
    d4::gemm_force_config(id);                                                                      // < 100 first fp32 family, 100 + c second (gemm2.hip)
    return id >= 100 ? d4::gemm2_configs() : n_main;
}
const char* d4_profile_class_name(int c) { return d4::gemm_profile_class_name(c); }

int d4_debug_buffer(d4_engine* e, const char* name, float** ptr) {
    D4_REQUIRE(e && name && ptr, "null argument");
    struct { const char* n; float* p; } tbl[] = {
        {"slabs", e->slabs}, {"xpool", e->xpool}, {"xfc", e->xfc}, {"cslabs", e->cslabs}, {"proj0", e->proj0}, {"proj", e->proj}, {"att", e->att},
        {"ffh", e->ffh}, {"pool_q", e->pool_q}, {"pool_kv", e->pool_kv}, {"pool_att", e->pool_att}, {"cq", e->cq},
        {"ckv", e->ckv}, {"catt", e->catt}, {"lkv", e->lkv}, {"latt", e->latt}, {"space", e->space}, {"gs", e->gs},
        {"okv", e->okv}, {"oatt", e->oatt}, {"oproj", e->oproj}, {"pred", e->pred}, {"x_lat", e->x_lat},
        {"cache", e->cache}, {"lin_q", e->lin_q}, {"lin_gate", e->lin_gate}, {"lout_q", e->lout_q},
        {"l_logits", e->l_logits}, {"l_dlogits", e->l_dlogits}, {"l_adv", e->l_adv}, {"l_returns", e->l_returns}, {"l_mask", e->l_mask},
    };
    for (auto& t : tbl) if (!strcmp(t.n, name)) {
        // a bf16 engine keeps some activation buffers ONLY as bf16 images (nothing writes the fp32 buffer): handing out the stale fp32 view would mislead
        D4_REQUIRE(!e->shadow_only(t.p), "debug buffer '%s' exists only as a bf16 image in this engine (matmul_bf16 = 1)", name);
        *ptr = t.p; return 0;
    }
    D4_REQUIRE(false, "unknown debug buffer '%s'", name);
}

int d4_learn(d4_engine* e, const d4_learn_io* io, void* stream) {
    D4_REQUIRE(e && io, "null argument");
    return d4::learn(e, io, static_cast<hipStream_t>(stream));
}

int d4_gemm(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias,
            const float* R, int ldr, int M, int N, int K, int flags, float rms_eps, void* stream) {
    d4::GemmArgs g{A, lda, W, ldw, C, ldc, bias, R, ldr, M, N, K, flags, rms_eps};
    return d4::gemm(g, static_cast<hipStream_t>(stream));
}

int d4_gemm_pair(const float* A1, int lda1, const float* W1, float* C1, int ldc1, int M1, int N1, const float* A2, int lda2, const float* W2, float* C2,
                 int ldc2, int M2, int N2, int K, int flags, float rms_eps, void* stream) {
    d4::GemmArgs a{A1, lda1, W1, K, C1, ldc1, nullptr, nullptr, 0, M1, N1, K, flags, rms_eps};
    d4::GemmArgs b{A2, lda2, W2, K, C2, ldc2, nullptr, nullptr, 0, M2, N2, K, flags, rms_eps};
    return d4::gemm_pair(a, b, static_cast<hipStream_t>(stream));
}

int d4_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, float* part, int64_t part_floats,
               int tile_n, int slices, void* stream) {
    return d4::gemm_tn(A, lda, B, ldb, C, ldc, M, N, K, part, (size_t)part_floats, static_cast<hipStream_t>(stream), tile_n, slices);
}

int d4_gemm_batched(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias,
                    const float* R, int ldr, int M, int N, int K, int flags, float rms_eps, int batch,
                    int64_t strideA, int64_t strideW, int64_t strideC, void* stream) {
    d4::GemmArgs g{A, lda, W, ldw, C, ldc, bias, R, ldr, M, N, K, flags, rms_eps};
    g.batch = batch; g.strideA = strideA; g.strideW = strideW; g.strideC = strideC;
    return d4::gemm(g, static_cast<hipStream_t>(stream));
}

int d4_split_bf16x3(const float* src, uint16_t* dst, int64_t n, int64_t plane_stride, void* stream) {
    D4_REQUIRE(src && dst && n >= 0 && plane_stride >= n && (plane_stride % 8) == 0, "d4_split_bf16x3: bad arguments");
    return d4::split_bf16x3(src, dst, n, plane_stride, static_cast<hipStream_t>(stream));
}
int d4_gemm_split(const float* A, int lda, const uint16_t* W3, int64_t plane_stride, int ldw, float* C, int ldc, const float* bias,
                  const float* R, int ldr, int M, int N, int K, int flags, float rms_eps, int config, void* stream) {
    d4::GemmArgs g{A, lda, reinterpret_cast<const float*>(W3), ldw, C, ldc, bias, R, ldr, M, N, K, flags, rms_eps};
    g.Wb = W3; g.wplane = plane_stride;
    D4_REQUIRE(d4::gemm_x3_applicable(g), "d4_gemm_split: call not supported (M=%d N=%d K=%d flags=%d: K %% 32, ldw %% 8, plane_stride %% 8, 16-byte alignment)", M, N, K, flags);
    if (M == 0) return 0;
    if (config == d4::gemm_x3_configs()) return d4::gemm_x3sk_launch(g, static_cast<hipStream_t>(stream));      // 6: the persistent form (the engine's)
    if (config >= 0) return d4::gemm_x3_launch(config, g, static_cast<hipStream_t>(stream));
    return d4::gemm_x3_launch(d4::gemm_x3_heuristic(g), g, static_cast<hipStream_t>(stream));
}
int d4_split_f16x2(const float* src, uint16_t* dst, int rows, int cols, int ld, int64_t plane_stride, float* inv_scale, void* stream) {
    D4_REQUIRE(src && dst && inv_scale && rows >= 0 && cols >= 1 && ld >= cols && (ld % 8) == 0 && plane_stride >= (int64_t)rows * ld && (plane_stride % 8) == 0,
               "d4_split_f16x2: bad arguments");
    return d4::split_f16x2_rows(src, dst, rows, cols, ld, plane_stride, inv_scale, static_cast<hipStream_t>(stream));
}
int d4_row_scale_exp(const float* A, int64_t lda, int rows, int K, int32_t* exp_out, void* stream) {
    D4_REQUIRE(A && exp_out && rows >= 0 && K >= 4 && lda >= K, "d4_row_scale_exp: bad arguments");
    return d4::row_scale_exp(A, lda, rows, K, exp_out, static_cast<hipStream_t>(stream));
}
int d4_gemm_split2(const float* A, int lda, const uint16_t* W2, int64_t plane_stride, int ldw, const float* w_inv_scale, float* C, int ldc, const float* bias,
                   const float* R, int ldr, int M, int N, int K, int flags, float rms_eps, int config, const int32_t* a_exp, void* stream) {
    d4::GemmArgs g{A, lda, reinterpret_cast<const float*>(W2), ldw, C, ldc, bias, R, ldr, M, N, K, flags, rms_eps};
    g.Wb = W2; g.wplane = plane_stride; g.wscale = w_inv_scale; g.aexp = a_exp;
    D4_REQUIRE(d4::gemm_h2_applicable(g), "d4_gemm_split2: call not supported (M=%d N=%d K=%d flags=%d: K %% 32, lda %% 4, ldw %% 8, plane_stride %% 8, 16-byte alignment)", M, N, K, flags);
    if (M == 0) return 0;
    return d4::gemm_h2_launch(config >= 0 ? config : d4::gemm_h2_heuristic(g), g, static_cast<hipStream_t>(stream));
}
int d4_gemm_bf16(const float* A, int lda, const uint16_t* Wb, int ldw, float* C, int ldc, const float* bias,
                 const float* R, int ldr, int M, int N, int K, int flags, float rms_eps, void* stream) {
    d4::GemmArgs g{A, lda, reinterpret_cast<const float*>(Wb), ldw, C, ldc, bias, R, ldr, M, N, K, flags, rms_eps};
    g.Wb = Wb;
    return d4::gemm_bf16(g, static_cast<hipStream_t>(stream));
}

int d4_cvt_bf16(const float* src, uint16_t* dst, int64_t n, void* stream) { return d4::cvt_f32_to_bf16(src, dst, n, static_cast<hipStream_t>(stream)); }

int d4_gemm_bf16a(const uint16_t* Ab, int lda, const uint16_t* Wb, int ldw, float* C, int ldc, uint16_t* Cb, const float* bias, const float* R, int ldr,
                  int M, int N, int K, int flags, float rms_eps, int config, void* stream) {
    d4::GemmArgs g{nullptr, lda, nullptr, ldw, C, ldc, bias, R, ldr, M, N, K, flags, rms_eps};
    g.Ab = Ab; g.Wb = Wb; g.Cb = Cb;
    D4_REQUIRE(d4::gemm_bf16a_applicable(g), "d4_gemm_bf16a: call not supported (K %% 64, lda / ldw %% 8, 16-byte aligned operands)");
    if (config >= 100) { g.group_m = -1; config -= 100; }       // 100 + c: configuration c with the plain row-major tile order (A/B of the grouped order)
    return d4::gemm_bf16a_launch(config >= 0 ? config : d4::gemm_bf16a_rule(g), g, static_cast<hipStream_t>(stream));
}

int d4_gemm_bf16a_compact(const uint16_t* Ab, int lda, const uint16_t* Wb, int ldw, float* C, int ldc, uint16_t* Cb, const float* bias, const float* R, int ldr,
                          int M, int N, int K, int flags, float rms_eps, float* C2, uint16_t* C2b, int ldc2, int c2_S, int c2_lo, int c2_hi, int c2_last,
                          int config, void* stream) {
    d4::GemmArgs g{nullptr, lda, nullptr, ldw, C, ldc, bias, R, ldr, M, N, K, flags, rms_eps};
    g.Ab = Ab; g.Wb = Wb; g.Cb = Cb;
    g.C2 = C2; g.C2b = C2b; g.ldc2 = ldc2; g.c2_S = c2_S; g.c2_lo = c2_lo; g.c2_hi = c2_hi; g.c2_last = c2_last;
    D4_REQUIRE(C2 && c2_S >= 1 && c2_lo >= 0 && c2_hi >= c2_lo && c2_hi <= c2_S, "d4_gemm_bf16a_compact: bad compaction arguments");
    D4_REQUIRE(d4::gemm_bf16a_applicable(g), "d4_gemm_bf16a_compact: call not supported (K %% 64, lda / ldw %% 8, 16-byte aligned operands)");
    return d4::gemm_bf16a_launch(config >= 0 ? config : d4::gemm_bf16a_rule(g), g, static_cast<hipStream_t>(stream));
}

int d4_gemm_bf16a_batched(const uint16_t* Ab, int lda, const uint16_t* Wb, int ldw, float* C, int ldc, uint16_t* Cb, const float* bias, const float* R, int ldr,
                          int M, int N, int K, int flags, float rms_eps, int batch, int64_t strideA, int64_t strideW, int64_t strideC, int config, void* stream) {
    d4::GemmArgs g{nullptr, lda, nullptr, ldw, C, ldc, bias, R, ldr, M, N, K, flags, rms_eps};
    g.Ab = Ab; g.Wb = Wb; g.Cb = Cb;
    g.batch = batch; g.strideA = strideA; g.strideW = strideW; g.strideC = strideC;
    D4_REQUIRE(C || Cb, "d4_gemm_bf16a_batched: no output");
    if (M == 0) return 0;
    D4_REQUIRE(d4::gemm_bf16a_applicable(g), "d4_gemm_bf16a_batched: call not supported (K %% 64, lda / ldw / strides %% 8, 16-byte aligned operands)");
    return d4::gemm_bf16a_launch(config >= 0 ? config : d4::gemm_bf16a_rule(g), g, static_cast<hipStream_t>(stream));
}
int d4_cvt_rows_bf16(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int rows, int cols, void* stream) {
    D4_REQUIRE(src && dst && rows >= 0 && cols >= 0 && lds >= cols && ldd >= cols, "d4_cvt_rows_bf16: bad arguments");
    return d4::cvt_rows_bf16(src, lds, dst, ldd, rows, cols, static_cast<hipStream_t>(stream));
}

int d4_rmsnorm(const float* x, int ldx, const float* gamma, float* y, int ldy, int rows, int dim, float eps, void* stream) {
    return d4::rmsnorm_rows(x, ldx, gamma, y, ldy, rows, dim, eps, static_cast<hipStream_t>(stream));
}

int d4_rmsnorm_backward(const float* x, const float* dy, const float* gamma, float* dx, float* d_gamma, float* scratch, int rows, int dim, float eps, void* stream) {
    D4_REQUIRE(x && dy && gamma && dx && d_gamma && scratch, "d4_rmsnorm_backward: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (rows == 0) return 0;
    if (int rc = d4::rmsnorm_bwd(x, dy, gamma, scratch, dx, rows, dim, eps, s)) return rc;
    return d4::colsum(scratch, dim, rows, dim, d_gamma, s);
}

// MultiCategorical.sample + log_prob of the sample (D4:485-497, 1374-1376, 1422-1423), stateless: Gumbel-max per action type from injected
// uniforms (same shape as the logits), log-softmax gather.  action_sizes: device [na].
int d4_categorical_sample_logp(const float* logits, int ld, const float* uniform, int ld_u, const int32_t* action_sizes, int rows, int na,
                               float temperature, int64_t* actions, float* log_probs, void* stream) {
    D4_REQUIRE(logits && uniform && action_sizes && actions && log_probs && rows >= 0 && na >= 1, "d4_categorical_sample_logp: bad arguments");
    d4::SampleArgs sa{};
    sa.logits = logits; sa.ld = ld; sa.gumbel_u = uniform; sa.ld_u = ld_u; sa.actions = actions; sa.act_stride = na; sa.log_probs = log_probs;
    sa.lp_stride = na; sa.action_sizes = action_sizes; sa.B = rows; sa.na = na; sa.temperature = temperature;
    return d4::sample_actions_terminals(sa, static_cast<hipStream_t>(stream));
}
int d4_hl_gauss_scalar(const float* logits, int ld, const float* centers, float* out, int rows, int bins, void* stream) {
    return d4::hl_gauss_scalar(logits, ld, centers, out, 1, rows, bins, static_cast<hipStream_t>(stream));
}

}  // extern "C"
