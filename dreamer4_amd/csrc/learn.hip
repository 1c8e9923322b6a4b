// Actor / critic learner: reference DynamicsWorldModel.learn_from_experience with
// only_learn_policy_value_heads=True and stored agent embeddings (D4:5893-6305), forward AND
// backward, plus the clipped AdamW step DreamTrainer applies (trainers.py:1436-1452).
//
//   calc_gae (D4:1566-1600) -> masked z-score (D4:404-410, 6024) -> policy MLP + unembed head 0 ->
//   log-prob / entropy (D4:1422-1426) -> PPO / SPO / PMPO surrogate with delight gating
//   (D4:6119-6242) ;  value MLP -> HL-Gauss cross entropy against transform_to_probs(returns)
//   (D4:6268-6295).
//
// The reference builds these losses from ~40 elementwise ATen ops and lets autograd derive the
// gradients; here each loss has one fused forward+backward kernel and the MLP backward is explicit
// (GEMMs on the MFMA kernel in its transposed-operand modes, deterministic column reductions).
// Every reduction has a fixed order: results are bitwise reproducible run to run.
#include "common.h"
#include "engine.h"
#include "beta.h"
#include <float.h>
#include <math.h>

namespace d4 {

static constexpr float RMS_EPS_L = 1.1920928955078125e-07f;

// ------------------------------------------------------------------------------------ GAE
// one thread per trajectory: masks (D4:5943-5967) + reverse recurrence
__global__ void gae_kernel(const float* rewards, const float* values, const int64_t* lens, const uint8_t* is_truncated,
                           const uint8_t* terminals, float gamma, float lam, int B, int T,
                           float* returns, float* adv, float* mask) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int len = lens ? (int)lens[b] : T;
    const int trunc = is_truncated ? (is_truncated[b] != 0) : 1;
    const int term = terminals ? (terminals[b] != 0) : 0;
    const int learn_len = len - trunc;
    const int last = len - 1 > 0 ? len - 1 : 0;
    float g = 0.f, v_next = 0.f;
    for (int t = T - 1; t >= 0; --t) {
        const bool in_len = t < len;
        const float r = in_len ? rewards[(int64_t)b * T + t] : 0.f;
        const float v = in_len ? values[(int64_t)b * T + t] : 0.f;
        bool cont = t < last;
        if (term && t == last) cont = false;
        const float m = cont ? 1.f : 0.f;
        const bool learn = t < learn_len;
        float delta = r + gamma * v_next * m - v;
        if (!learn) delta = 0.f;
        g = (gamma * lam * m) * g + delta;
        const float ret = g + v;
        returns[(int64_t)b * T + t] = ret;
        if (adv) adv[(int64_t)b * T + t] = ret - v;
        if (mask) mask[(int64_t)b * T + t] = learn ? 1.f : 0.f;
        v_next = v;
    }
}

// ------------------------------------------------------------------------------------ deterministic reductions
// out[slot] = sum_i f(a[i], b[i]) with one 1024-thread block (fixed tree order).
enum { RED_PROD = 0, RED_SQDIFF_MASKED = 1, RED_SUM = 2 };
template <int MODE>
__global__ __launch_bounds__(1024) void reduce1_kernel(const float* a, const float* b, const float* scal, int64_t n, float* out) {
    __shared__ float sh[1024];
    float s = 0.f;
    float mean = 0.f;
    if (MODE == RED_SQDIFF_MASKED) mean = scal[0] / fmaxf(scal[1], 1.f);
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        if (MODE == RED_PROD) s += a[i] * b[i];
        else if (MODE == RED_SUM) s += a[i];
        else { float d = a[i] - mean; s += d * d * b[i]; }
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

// column sums: out[c] = sum_r x[r][c]; block = 16 columns x 64 row lanes (4x more blocks than a 64-column block: the
// [4096][2048] bias-gradient reductions are latency bound, not bandwidth bound), 16 loads in flight per thread (a 3840-row reduction is
// four round trips to memory instead of eight: 14.7 -> ~8 us), fixed order
__device__ __forceinline__ void colsum_block(const float* x, int ld, int rows, int cols, float* out, int bx) {
    __shared__ float sh[64][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = bx * 16 + cl;
    float s = 0.f;
    if (c < cols) {
        for (int r = rl; r < rows; r += 16 * 64) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int rr = r + u * 64;
                v[u] = rr < rows ? x[(int64_t)rr * ld + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
    }
    sh[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < cols) {
        float t = 0.f;
        for (int i = 0; i < 64; ++i) t += sh[i][cl];
        out[c] = t;
    }
}
__global__ __launch_bounds__(1024) void colsum_kernel(const float* x, int ld, int rows_all, int cols, float* out_all, int chunk) {
    // blockIdx.y: row chunk [y * chunk, (y + 1) * chunk) summed into row y of out (one chunk = the whole matrix in the plain form)
    const int r_lo = blockIdx.y * chunk, rows = min(rows_all, r_lo + chunk) - r_lo;
    colsum_block(x + (int64_t)r_lo * ld, ld, rows, cols, out_all + (int64_t)blockIdx.y * cols, blockIdx.x);
}
// several independent column sums in ONE launch (a block backward has three or four: bias, norm-gain and key-gain gradients — each alone is a
// 32-block, latency-bound launch of ~14 us); same per-column arithmetic and order as colsum_kernel
__global__ __launch_bounds__(1024) void colsum_batch_kernel(ColsumBatch b) {
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.first[i + 1]) ++i;
    colsum_block(b.x[i], b.ld[i], b.rows[i], b.cols[i], b.out[i], blockIdx.x - b.first[i]);
}
int colsum_batch(ColsumBatch& b, hipStream_t s) {
    if (b.n == 0) return 0;
    int nb = 0;
    for (int i = 0; i < b.n; ++i) { b.first[i] = nb; nb += cdiv(b.cols[i], 16); }
    hipLaunchKernelGGL(colsum_batch_kernel, dim3(nb), dim3(1024), 0, s, b);
    D4_LAUNCH_CHECK();
    return 0;
}
// scratch (optional, >= 16 * cols floats): tall matrices (the attention pools' context gradients: up to 13 x the token rows) are summed in 16
// row chunks over 16 x the blocks, then the 16 partial rows in a second, tiny launch — fixed order either way
int colsum(const float* x, int ld, int rows, int cols, float* out, hipStream_t s, float* scratch, size_t scratch_floats) {
    constexpr int SY = 16;
    if (scratch && rows >= 8192 && scratch_floats >= (size_t)SY * cols) {
        const int chunk = cdiv(cdiv(rows, SY), 64) * 64;
        const int ny = cdiv(rows, chunk);
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 16), ny), dim3(1024), 0, s, x, ld, rows, cols, scratch, chunk);
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 16)), dim3(1024), 0, s, scratch, cols, ny, cols, out, ny);
        D4_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 16)), dim3(1024), 0, s, x, ld, rows, cols, out, rows);
    D4_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ policy loss fwd + bwd
struct PolicyLossArgs {
    const float* logits; int ld;        // [R][Apad]
    const int64_t* actions;             // [R][na]
    const float* old_lp;                // [R][na]
    const float* old_logits; int ldo;   // [R][A] (pmpo) or null
    const float* adv_raw;               // [R]
    const float* mask;                  // [R]
    const float* scal;                  // stats: [0] sum adv*mask, [1] count, [2] sum sq diff
    const int32_t* action_sizes;
    float* dlogits;                     // [R][Apad]
    float* row_pl; float* row_ent; float* row_aux;   // per-row masked terms
    // continuous actions (Beta head): raw parameters [R][ldc] (2 per action), targets, old log-probs, old parameters (pmpo KL)
    const float* cparams; int ldc;
    const float* actions_cont; const float* old_lp_cont; const float* old_cparams;
    float* dcparams;                    // [R][ldc]
    int beta_param;                     // d4_config.continuous_beta_param
    int nc;
    int R, na, A, objective, normalize, use_gate, reverse_kl;
    float eps, clip, ent_w, gate_temp, pmpo_alpha, kl_w;
};

__global__ void policy_loss_kernel(PolicyLossArgs p) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.R) return;
    const float count = fmaxf(p.scal[1], 1.f);
    const float mk = p.mask[r];
    float adv = p.adv_raw[r];
    if (p.normalize) {
        const float mean = p.scal[0] / count;
        const float var = p.scal[2] / count;
        adv = (adv - mean) / sqrtf(fmaxf(var, p.eps));
    }
    const float* lg = p.logits + (int64_t)r * p.ld;
    float* dl = p.dlogits + (int64_t)r * p.ld;
    // pass 1: joint log-prob, entropy
    float lp = 0.f, old = 0.f, ent = 0.f;
    int o = 0;
    for (int a = 0; a < p.na; ++a) {
        const int n = p.action_sizes[a];
        float mx = -FLT_MAX;
        for (int j = 0; j < n; ++j) mx = fmaxf(mx, lg[o + j]);
        float se = 0.f;
        for (int j = 0; j < n; ++j) se += expf(lg[o + j] - mx);
        const float lse = mx + logf(se);
        const int act = (int)p.actions[(int64_t)r * p.na + a];
        lp += lg[o + act] - lse;
        old += p.old_lp[(int64_t)r * p.na + a];
        float h = 0.f;
        for (int j = 0; j < n; ++j) { const float l = lg[o + j] - lse; h -= expf(l) * l; }
        ent += h;
        o += n;
    }
    // continuous actions: log-prob of the stored action and entropy of each Beta join the same sums  D4:6090-6111
    for (int c = 0; c < p.nc; ++c) {
        const float* raw = p.cparams + (int64_t)r * p.ldc + 2 * c;
        const BetaAB ab = beta_ab(raw[0], raw[1], p.beta_param);
        lp += beta_log_prob(ab.a, ab.b, p.actions_cont[(int64_t)r * p.nc + c]);
        old += p.old_lp_cont[(int64_t)r * p.nc + c];
        ent += lbetaf(ab.a, ab.b) - (ab.a - 1.f) * digammaf(ab.a) - (ab.b - 1.f) * digammaf(ab.b) + (ab.a + ab.b - 2.f) * digammaf(ab.a + ab.b);
    }
    const float gate = p.use_gate ? sigmoidf(-lp * adv / p.gate_temp) : 1.f;
    float pl = 0.f, dpl_dlp = 0.f, aux = 0.f;
    if (p.objective == 0) {             // ppo  D4:6204-6212
        const float ratio = expf(lp - old);
        const float cr = fminf(fmaxf(ratio, 1.f - p.clip), 1.f + p.clip);
        const float s1 = ratio * adv, s2 = cr * adv;
        pl = -fminf(s1, s2) * gate;
        const bool inside = ratio >= 1.f - p.clip && ratio <= 1.f + p.clip;
        const float dmin = (s1 < s2 || (s1 == s2 && inside)) ? s1 : ((s1 == s2) ? 0.5f * s1 : 0.f);
        dpl_dlp = -dmin * gate;
    } else if (p.objective == 1) {      // spo  D4:6188-6198
        const float ratio = expf(lp - old);
        const float q = fabsf(adv) * (ratio - 1.f) / p.clip;
        pl = -(ratio * adv - fabsf(adv) * (ratio - 1.f) * (ratio - 1.f) / (2.f * p.clip)) * gate;
        dpl_dlp = -(ratio * adv - q * ratio) * gate;
    } else {                            // pmpo D4:6127-6154
        const float w = fabsf(tanhf(adv)) * gate;
        const float sgn = adv >= 0.f ? 1.f : -1.f;
        pl = -p.pmpo_alpha * sgn * lp * w;              // summed / num below (mask applied there)
        dpl_dlp = -p.pmpo_alpha * sgn * w;
    }
    // pass 2: gradients wrt logits
    const float scale = mk / count;
    o = 0;
    for (int a = 0; a < p.na; ++a) {
        const int n = p.action_sizes[a];
        float mx = -FLT_MAX;
        for (int j = 0; j < n; ++j) mx = fmaxf(mx, lg[o + j]);
        float se = 0.f;
        for (int j = 0; j < n; ++j) se += expf(lg[o + j] - mx);
        const float lse = mx + logf(se);
        const int act = (int)p.actions[(int64_t)r * p.na + a];
        float h = 0.f;
        for (int j = 0; j < n; ++j) { const float l = lg[o + j] - lse; h -= expf(l) * l; }
        float kl = 0.f, lse_o = 0.f;
        if (p.objective == 2 && p.kl_w > 0.f && p.old_logits) {
            const float* og = p.old_logits + (int64_t)r * p.ldo;
            float mo = -FLT_MAX;
            for (int j = 0; j < n; ++j) mo = fmaxf(mo, og[o + j]);
            float so = 0.f;
            for (int j = 0; j < n; ++j) so += expf(og[o + j] - mo);
            lse_o = mo + logf(so);
            for (int j = 0; j < n; ++j) {
                const float ln = lg[o + j] - lse, lo = og[o + j] - lse_o;
                kl += p.reverse_kl ? expf(lo) * (lo - ln) : expf(ln) * (ln - lo);
            }
            aux += kl;
        }
        for (int j = 0; j < n; ++j) {
            const float l = lg[o + j] - lse;
            const float pj = expf(l);
            float g = dpl_dlp * ((j == act ? 1.f : 0.f) - pj);          // d lp / d logit
            g += p.ent_w * pj * (l + h);                                  // d(-H)/d logit
            if (p.objective == 2 && p.kl_w > 0.f && p.old_logits) {
                const float lo = p.old_logits[(int64_t)r * p.ldo + o + j] - lse_o;
                g += p.kl_w * (p.reverse_kl ? (pj - expf(lo)) : pj * ((l - lo) - kl));
            }
            dl[o + j] = g * scale;
        }
        o += n;
    }
    for (int j = p.A; j < p.ld; ++j) dl[j] = 0.f;
    // gradients wrt the raw Beta parameters: d lp, d(-H) and (pmpo) the KL against the behaviour parameters
    for (int c = 0; c < p.nc; ++c) {
        const float* raw = p.cparams + (int64_t)r * p.ldc + 2 * c;
        const BetaAB ab = beta_ab(raw[0], raw[1], p.beta_param);
        const float a = ab.a, b = ab.b, x = p.actions_cont[(int64_t)r * p.nc + c];
        const float psi_a = digammaf(a), psi_b = digammaf(b), psi_ab = digammaf(a + b);
        const float tri_ab = trigammaf(a + b);
        float ga = dpl_dlp * (logf(x) - psi_a + psi_ab);                              // d lp / d alpha
        float gb = dpl_dlp * (log1pf(-x) - psi_b + psi_ab);
        // H = lbeta - (a-1) psi(a) - (b-1) psi(b) + (a+b-2) psi(a+b);  dH/da = -(a-1) psi'(a) + (a+b-2) psi'(a+b)
        ga += p.ent_w * ((a - 1.f) * trigammaf(a) - (a + b - 2.f) * tri_ab);          // d(-H) / d alpha
        gb += p.ent_w * ((b - 1.f) * trigammaf(b) - (a + b - 2.f) * tri_ab);
        if (p.objective == 2 && p.kl_w > 0.f && p.old_cparams) {
            const float* oraw = p.old_cparams + ((int64_t)r * p.nc + c) * 2;
            const BetaAB ob = beta_ab(oraw[0], oraw[1], p.beta_param);
            const float a2 = ob.a, b2 = ob.b;
            float kl, ka, kb;
            if (p.reverse_kl) {
                // KL(old || new): lbeta(a,b) - lbeta(a2,b2) + (a2-a) psi(a2) + (b2-b) psi(b2) + (a-a2+b-b2) psi(a2+b2)
                const float q_a = digammaf(a2), q_b = digammaf(b2), q_ab = digammaf(a2 + b2);
                kl = lbetaf(a, b) - lbetaf(a2, b2) + (a2 - a) * q_a + (b2 - b) * q_b + (a - a2 + b - b2) * q_ab;
                ka = (psi_a - psi_ab) - q_a + q_ab;
                kb = (psi_b - psi_ab) - q_b + q_ab;
            } else {
                // KL(new || old): lbeta(a2,b2) - lbeta(a,b) + (a-a2) psi(a) + (b-b2) psi(b) + (a2-a+b2-b) psi(a+b)
                kl = lbetaf(a2, b2) - lbetaf(a, b) + (a - a2) * psi_a + (b - b2) * psi_b + (a2 - a + b2 - b) * psi_ab;
                ka = -(psi_a - psi_ab) + psi_a + (a - a2) * trigammaf(a) - psi_ab + (a2 - a + b2 - b) * tri_ab;
                kb = -(psi_b - psi_ab) + psi_b + (b - b2) * trigammaf(b) - psi_ab + (a2 - a + b2 - b) * tri_ab;
            }
            aux += kl;
            ga += p.kl_w * ka; gb += p.kl_w * kb;
        }
        float* dc = p.dcparams + (int64_t)r * p.ldc + 2 * c;
        dc[0] = ga * ab.da * scale;
        dc[1] = gb * ab.db * scale;
    }
    for (int j = 2 * p.nc; j < p.ldc; ++j) p.dcparams[(int64_t)r * p.ldc + j] = 0.f;
    p.row_pl[r] = pl * mk;
    p.row_ent[r] = -ent * mk;
    p.row_aux[r] = aux * mk;
}

// ------------------------------------------------------------------------------------ value loss fwd + bwd
// one wave per row: HL-Gauss target probs (erf), CE against log_softmax(value bins)
// (TWO_HOT: SymExpTwoHot.forward D4:1001-1040 — `support` holds the `bins` bin values; the target puts (right - v) / (right - left) on the
//  bin to the left of v and the rest on its right neighbour)
template <bool TWO_HOT>
__global__ __launch_bounds__(256) void value_loss_kernel(const float* vbins, int ld, const float* returns, const float* mask,
                                                         const float* scal, const float* support, int R, int bins,
                                                         float sigma_sqrt2, float hl_eps, float vmin, float vmax,
                                                         float* dv, float* row_loss) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63;
    const float count = fmaxf(scal[1], 1.f);
    const float mk = mask[r];
    const float* v = vbins + (int64_t)r * ld;
    float ret = fminf(fmaxf(returns[r], TWO_HOT ? support[0] : vmin), TWO_HOT ? support[bins - 1] : vmax);
    int li = 0, ri = 0;
    float wl = 0.f;
    if (TWO_HOT) {
        // torch.searchsorted(bin_values, v): first index with bin_values[i] >= v (binary search, wave-uniform)
        int lo = 0, hi = bins;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (support[mid] < ret) lo = mid + 1; else hi = mid; }
        li = lo - 1 > 0 ? lo - 1 : 0;
        ri = li + 1 < bins - 1 ? li + 1 : bins - 1;
        wl = (support[ri] - ret) / (support[ri] - support[li]);
    }
    const float z = TWO_HOT ? 1.f : erff((support[bins] - ret) / sigma_sqrt2) - erff((support[0] - ret) / sigma_sqrt2);
    const float zc = fmaxf(z, hl_eps);
    auto target = [&](int c) -> float {
        if (TWO_HOT) return c == ri ? 1.f - wl : (c == li ? wl : 0.f);            // (scatter order of the reference: right written last)
        return (erff((support[c + 1] - ret) / sigma_sqrt2) - erff((support[c] - ret) / sigma_sqrt2)) / zc;
    };
    float mx = -FLT_MAX;
    for (int c = lane; c < bins; c += 64) mx = fmaxf(mx, v[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < bins; c += 64) se += expf(v[c] - mx);
    const float lse = mx + logf(wave_sum(se));
    float loss = 0.f, psum = 0.f;
    for (int c = lane; c < bins; c += 64) {
        const float pr = target(c);
        loss -= pr * (v[c] - lse);
        psum += pr;
    }
    loss = wave_sum(loss);
    psum = wave_sum(psum);
    const float scale = mk / count;
    for (int c = lane; c < bins; c += 64) {
        const float pr = target(c);
        dv[(int64_t)r * ld + c] = (expf(v[c] - lse) * psum - pr) * scale;
    }
    for (int c = bins + lane; c < ld; c += 64) dv[(int64_t)r * ld + c] = 0.f;
    if (lane == 0) row_loss[r] = loss * mk;
}

__global__ void masked_mean_final_kernel(const float* scal, float* out) { out[0] = scal[7] / fmaxf(scal[1], 1.f); }

__global__ void finalize_losses_kernel(const float* scal, float* losses, int objective, float ent_w, float kl_w) {
    // scal: [1] count, [4] sum pl, [5] sum ent, [6] sum aux(kl), [7] sum value loss
    const float count = fmaxf(scal[1], 1.f);
    float pl = scal[4] / count;
    if (objective == 2) pl += kl_w * scal[6] / count;
    losses[0] = pl + ent_w * scal[5] / count;
    losses[1] = scal[7] / count;
}

// ------------------------------------------------------------------------------------ MLP backward pieces
__global__ void silu_bwd_kernel(const float* dy, const float* z, float* dz, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = sigmoidf(z[i]);
        dz[i] = dy[i] * s * (1.f + z[i] * (1.f - s));
    }
}

// RMSNorm backward, one wave per row:  xhat = x * rstd * gamma
//   tg[r][k] = dxhat * x * rstd   (column-summed afterwards -> dgamma)
//   dx[r][k] = rstd * (gamma * dxhat - x * rstd^2 * mean_k(gamma * dxhat * x))
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* x, const float* dxhat, const float* gamma, float* tg, float* dx,
                                                          int rows, int d, float eps) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (int64_t)r * d;
    const float* gr = dxhat + (int64_t)r * d;
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < d; c += 64) { const float v = xr[c]; ss += v * v; dot += gamma[c] * gr[c] * v; }
    const float rstd = rsqrtf(wave_sum(ss) / (float)d + eps);
    dot = wave_sum(dot) / (float)d;
    for (int c = lane; c < d; c += 64) {
        const float v = xr[c], g = gr[c];
        tg[(int64_t)r * d + c] = g * v * rstd;
        if (dx) dx[(int64_t)r * d + c] = rstd * (gamma[c] * g - v * rstd * rstd * dot);
    }
}

int rmsnorm_bwd(const float* x, const float* dxhat, const float* gamma, float* tg, float* dx, int rows, int d, float eps, hipStream_t s) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, dxhat, gamma, tg, dx, rows, d, eps);
    D4_LAUNCH_CHECK();
    return 0;
}

// Backward through a = silu(y), y = LayerNorm(z) * g + b (one wave per row, z recomputed into zhat):
//   dy = da * silu'(y);  tg = dy * zhat, tb = dy (column-summed afterwards -> dg, db_norm)
//   dz = rstd * (g dy - mean(g dy) - zhat * mean(g dy zhat))
__global__ __launch_bounds__(256) void layernorm_silu_bwd_kernel(const float* z, int ldz, const float* da, const float* g, const float* b,
                                                                 float* tg, float* tb, float* dz, int rows, int d, float eps) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* zr = z + (int64_t)r * ldz;
    float sum = 0.f;
    for (int c = lane; c < d; c += 64) sum += zr[c];
    const float mean = wave_sum(sum) / (float)d;
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) { const float t = zr[c] - mean; ss += t * t; }
    const float rstd = rsqrtf(wave_sum(ss) / (float)d + eps);
    float m1 = 0.f, m2 = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float zh = (zr[c] - mean) * rstd;
        const float y = zh * g[c] + b[c];
        const float sg = sigmoidf(y);
        const float dy = da[(int64_t)r * d + c] * sg * (1.f + y * (1.f - sg));
        tg[(int64_t)r * d + c] = dy * zh;
        tb[(int64_t)r * d + c] = dy;
        const float gd = g[c] * dy;
        m1 += gd; m2 += gd * zh;
    }
    m1 = wave_sum(m1) / (float)d; m2 = wave_sum(m2) / (float)d;
    for (int c = lane; c < d; c += 64) {
        const float zh = (zr[c] - mean) * rstd;
        const float gd = g[c] * tb[(int64_t)r * d + c];
        dz[(int64_t)r * ldz + c] = rstd * (gd - m1 - zh * m2);
    }
}

static inline dim3 grid1d(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}

// dW[M][N] = A^T B over R rows (weight gradient of a head Linear)
static int gemm_dw_l(d4_engine* e, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int R, hipStream_t s) {
    if (gemm_tn_applicable(A, lda, B, ldb, C, ldc, M, N, R)) return gemm_tn(A, lda, B, ldb, C, ldc, M, N, R, e->l_dwpart, e->l_dwpart ? L_DWPART_FLOATS : 0, s);
    GemmArgs g{A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, M, N, R, GEMM_TRANS_A | GEMM_TRANS_B, 0.f};
    return gemm(g, s);
}

static int gemm_l(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K, int flags, hipStream_t s) {
    GemmArgs g{A, lda, W, ldw, C, ldc, nullptr, nullptr, 0, M, N, K, flags, 0.f};
    return gemm(g, s);
}

// Backward of mlp_forward(save=...).  dout: [R][dims[nl]] gradient of the (un-activated) output.
// dx0 (optional, [R][dims[0]]): gradient with respect to the MLP's INPUT (the agent embeddings), for fine-tuning the whole world model
static int mlp_backward(d4_engine* e, const Mlp& m, const float* save, int R, const float* dout, int ld_dout, hipStream_t s, float* dx0 = nullptr) {
    int rc;
    float* sx[9]; float* sxh[9]; float* sz[9];
    for (int i = 0; i < m.nl; ++i) m.save_ptrs(const_cast<float*>(save), R, i, &sx[i], &sxh[i], &sz[i]);
    float* dy = e->l_tmp[0]; float* dz = e->l_tmp[1]; float* dxh = e->l_tmp[2];
    const float* cur = dout;
    const bool pre = m.recipe == D4_MLP_PRE_RMS;
    for (int i = m.nl - 1; i >= 0; --i) {
        const int din = m.dims[i], dout_i = m.dims[i + 1];
        const float* dzp;
        int ldd = dout_i;
        D4_REQUIRE(m.db[i] && m.dw[i], "learner: head parameters were bound without gradient buffers");
        if (i == m.nl - 1) { dzp = cur; ldd = ld_dout; }
        else if (m.post_norm(i)) {
            // through silu and the LayerNorm: tg / tb (per-row terms of the norm's weight / bias gradients) use dxh and dy's buffers
            D4_REQUIRE(m.dg[i] && m.dnb[i], "learner: head parameters were bound without gradient buffers");
            const int ldz = m.ldz(i);
            float* tg = dxh; float* tb = dz; float* dzo = e->l_tmp[3];        // cur (= dy, the incoming gradient) stays intact while it is read
            hipLaunchKernelGGL(layernorm_silu_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, sz[i], ldz, cur, m.g[i], m.nb[i], tg, tb, dzo, R, dout_i, 1e-5f);
            D4_LAUNCH_CHECK();
            if ((rc = colsum(tg, dout_i, R, dout_i, m.dg[i], s))) return rc;
            if ((rc = colsum(tb, dout_i, R, dout_i, m.dnb[i], s))) return rc;
            dzp = dzo; ldd = ldz;
        } else {
            hipLaunchKernelGGL(silu_bwd_kernel, grid1d((int64_t)R * dout_i), dim3(256), 0, s, cur, sz[i], dz, (int64_t)R * dout_i);
            D4_LAUNCH_CHECK();
            dzp = dz;
        }
        if ((rc = colsum(dzp, ldd, R, dout_i, m.db[i], s))) return rc;
        const float* lin_in = pre ? sxh[i] : sx[i];
        // input gradients go through a transposed image of the weight (gemm_dx; scratch shared with the weight gradient's partial products,
        // used one after the other on the stream)
        float* wt = (e->l_dwpart && (size_t)dout_i * din <= L_DWPART_FLOATS) ? e->l_dwpart : nullptr;
        // dW[n][k] = sum_r dz[r][n] * lin_in[r][k]
        if ((rc = gemm_dw_l(e, dzp, ldd, lin_in, din, m.dw[i], din, dout_i, din, R, s))) return rc;
        if (!pre) {
            // dx[r][k] = sum_n dz[r][n] * W[n][k] is the previous layer's activation gradient directly
            if (i > 0 && (rc = gemm_dx(dzp, ldd, m.w[i], din, dy, din, R, din, dout_i, wt, s))) return rc;
            if (i == 0 && dx0 && (rc = gemm_dx(dzp, ldd, m.w[i], din, dx0, din, R, din, dout_i, wt, s))) return rc;
            cur = dy;
            continue;
        }
        D4_REQUIRE(m.dg[i], "learner: head parameters were bound without gradient buffers");
        // dxhat[r][k] = sum_n dz[r][n] * W[n][k]
        if ((rc = gemm_dx(dzp, ldd, m.w[i], din, dxh, din, R, din, dout_i, wt, s))) return rc;
        // through the RMSNorm; dy of the previous layer overwrites e->l_tmp[0]; tg reuses dz (dz is dead after the GEMMs)
        float* tg = dz;
        hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, sx[i], dxh, m.g[i], tg, i > 0 ? dy : dx0, R, din, RMS_EPS_L);
        D4_LAUNCH_CHECK();
        if ((rc = colsum(tg, din, R, din, m.dg[i], s))) return rc;
        cur = dy;
    }
    return 0;
}

// A rank whose trajectory shard is EMPTY (a global batch smaller than the number of ranks, or an uneven split): it has nothing to learn from but must
// take part in every collective of the step with a zero contribution, in the order the other ranks issue them (three statistics all-reduces here, the
// gradient all-reduce of each head in the caller), and ends up with the global losses and zero local gradients.
static int learn_empty_shard(d4_engine* e, const d4_learn_io* io, hipStream_t s) {
    const d4_config& c = e->c;
    int rc;
    float* scal = e->l_scal;
    D4_REQUIRE(e->LR > 0 && scal, "learn: an empty shard still needs an engine created with max_learn_rows >= 1 (the learner's scalar workspace)");
    D4_HIP(hipMemsetAsync(scal, 0, 64 * sizeof(float), s));
    if ((rc = io->allreduce_sum(scal, 2, io->allreduce_user)) || (rc = io->allreduce_sum(scal + 2, 1, io->allreduce_user)) ||
        (rc = io->allreduce_sum(scal + 4, 4, io->allreduce_user))) { set_error("allreduce callback failed (%d)", rc); return 5; }
    hipLaunchKernelGGL(finalize_losses_kernel, dim3(1), dim3(1), 0, s, scal, io->losses, io->objective, c.policy_entropy_weight,
                       io->objective == 2 ? c.pmpo_kl_div_loss_weight : 0.f);
    D4_LAUNCH_CHECK();
    for (const Mlp* m : {&e->policy, &e->value})
        for (int i = 0; i < m->nl; ++i) {
            const int din = m->dims[i], dout = m->dims[i + 1];
            D4_REQUIRE(m->db[i] && m->dw[i], "learner: head parameters were bound without gradient buffers");
            if ((rc = fill_f32(m->dw[i], 0.f, (int64_t)dout * din, s)) || (rc = fill_f32(m->db[i], 0.f, dout, s))) return rc;
            if (m->dg[i] && (rc = fill_f32(m->dg[i], 0.f, m->post_norm(i) ? dout : din, s))) return rc;
            if (m->dnb[i] && (rc = fill_f32(m->dnb[i], 0.f, dout, s))) return rc;
        }
    const int64_t mtp4d = (int64_t)c.multi_token_pred_len * 4 * e->D;
    if (e->na > 0 && e->action_unembed_grad && (rc = fill_f32(e->action_unembed_grad, 0.f, e->A * mtp4d, s))) return rc;
    if (e->nc > 0 && e->cont_unembed_grad && (rc = fill_f32(e->cont_unembed_grad, 0.f, 2 * e->nc * mtp4d, s))) return rc;
    return 0;
}

int learn(d4_engine* e, const d4_learn_io* io, hipStream_t s) {
    const d4_config& c = e->c;
    D4_REQUIRE(e->prepared, "engine not prepared");
    const int B = io->batch, T = io->time, R = B * T;
    D4_REQUIRE(io->objective >= 0 && io->objective <= 2, "unknown objective %d  [D4:6215]", io->objective);
    if (R == 0 && io->allreduce_sum && io->losses) return learn_empty_shard(e, io, s);
    D4_REQUIRE(R > 0 && R <= e->LR, "learn: %d rows exceed max_learn_rows %d", R, e->LR);
    D4_REQUIRE(io->agent_embed && io->old_values && io->rewards && io->losses && (e->na == 0 || (io->actions && io->old_log_probs)) &&
               (e->nc == 0 || (io->actions_cont && io->old_log_probs_cont)),
               "the generations need to contain the log probs, values, and rewards for policy optimization  [D4:5935]");
    D4_REQUIRE(io->objective >= 0 && io->objective <= 2, "unknown objective %d  [D4:6215]", io->objective);
    const int D = e->D, A = e->A, Apad = (A + 3) / 4 * 4 + (A == 0 ? 4 : 0), na = e->na, nc = e->nc, Cpad = (2 * nc + 3) / 4 * 4 + (nc == 0 ? 4 : 0);
    int rc;
    float* scal = e->l_scal;
    D4_HIP(hipMemsetAsync(scal, 0, 64 * sizeof(float), s));

    // ---- returns, advantages, masks
    float* row_a = e->l_adv;                        // raw advantage [R]
    float* mask_keep = e->l_mask;                   // learnable-step mask [R]
    hipLaunchKernelGGL(gae_kernel, dim3(cdiv(B, 128)), dim3(128), 0, s, io->rewards, io->old_values, io->lens, io->is_truncated,
                       io->terminals, c.gae_discount_factor, c.gae_lambda, B, T, e->l_returns, row_a, mask_keep);
    D4_LAUNCH_CHECK();
    if (io->returns) D4_HIP(hipMemcpyAsync(io->returns, e->l_returns, sizeof(float) * R, hipMemcpyDeviceToDevice, s));

    hipLaunchKernelGGL(reduce1_kernel<RED_PROD>, dim3(1), dim3(1024), 0, s, row_a, mask_keep, scal, (int64_t)R, scal + 0);
    hipLaunchKernelGGL(reduce1_kernel<RED_SUM>, dim3(1), dim3(1024), 0, s, mask_keep, nullptr, scal, (int64_t)R, scal + 1);
    D4_LAUNCH_CHECK();
    if (io->allreduce_sum && (rc = io->allreduce_sum(scal, 2, io->allreduce_user))) { set_error("allreduce callback failed (%d)", rc); return 5; }
    hipLaunchKernelGGL(reduce1_kernel<RED_SQDIFF_MASKED>, dim3(1), dim3(1024), 0, s, row_a, mask_keep, scal, (int64_t)R, scal + 2);
    D4_LAUNCH_CHECK();
    if (io->allreduce_sum && (rc = io->allreduce_sum(scal + 2, 1, io->allreduce_user))) { set_error("allreduce callback failed (%d)", rc); return 5; }

    const int normalize = io->normalize_advantages < 0 ? (io->objective != 2) : io->normalize_advantages;

    // ---- policy head forward (saved), logits of prediction head 0
    float* save_p = e->l_save;
    float* save_v = e->l_save + e->policy.save_floats(R);
    float* pe = e->l_tmp[0];
    if ((rc = mlp_forward(e, e->policy, io->agent_embed, D, R, pe, 4 * D, save_p, s))) return rc;
    if ((rc = fill_f32(e->l_logits, 0.f, (int64_t)R * Apad, s))) return rc;
    if (na > 0) {
        GemmArgs g{pe, 4 * D, e->action_unembed, c.multi_token_pred_len * 4 * D, e->l_logits, Apad, nullptr, nullptr, 0, R, A, 4 * D, 0, 0.f};
        if ((rc = gemm(g, s))) return rc;
    }
    if (nc > 0) {      // raw Beta parameters of prediction head 0 (head-0 slice of the [nc][mtp][4D][2] parameter as a GEMM weight)
        if ((rc = cunembed_gather(e->cont_unembed, e->cu_w, nc, c.multi_token_pred_len, 4 * D, s))) return rc;
        if ((rc = fill_f32(e->l_cparams, 0.f, (int64_t)R * Cpad, s))) return rc;
        GemmArgs g{pe, 4 * D, e->cu_w, 4 * D, e->l_cparams, Cpad, nullptr, nullptr, 0, R, 2 * nc, 4 * D, 0, 0.f};
        if ((rc = gemm(g, s))) return rc;
    }
    float* row_pl = e->l_rows, *row_ent = e->l_rows + R, *row_aux = e->l_rows + 2 * R, *row_vl = e->l_rows + 3 * R;
    {
        PolicyLossArgs p{};
        p.logits = e->l_logits; p.ld = Apad; p.actions = io->actions; p.old_lp = io->old_log_probs;
        p.old_logits = io->old_action_logits; p.ldo = A; p.adv_raw = row_a; p.mask = mask_keep; p.scal = scal;
        p.action_sizes = e->action_sizes; p.dlogits = e->l_dlogits; p.row_pl = row_pl; p.row_ent = row_ent; p.row_aux = row_aux;
        p.cparams = e->l_cparams; p.ldc = Cpad; p.actions_cont = io->actions_cont; p.old_lp_cont = io->old_log_probs_cont;
        p.old_cparams = io->old_cont_params; p.dcparams = e->l_dcparams; p.nc = nc; p.beta_param = e->c.continuous_beta_param;
        p.R = R; p.na = na; p.A = A; p.objective = io->objective; p.normalize = normalize;
        p.use_gate = io->use_delight_gating < 0 ? c.use_delight_gating : io->use_delight_gating;
        p.reverse_kl = c.pmpo_reverse_kl;
        p.eps = io->eps; p.clip = c.ppo_eps_clip; p.ent_w = c.policy_entropy_weight;
        p.gate_temp = io->delight_temperature > 0.f ? io->delight_temperature : c.delight_temperature;
        p.pmpo_alpha = c.pmpo_pos_to_neg_weight;
        p.kl_w = io->objective == 2 ? c.pmpo_kl_div_loss_weight : 0.f;
        D4_REQUIRE(!(p.kl_w > 0.f) || ((na == 0 || io->old_action_logits) && (nc == 0 || io->old_cont_params)), "pmpo with a KL weight needs old_action_unembeds  [D4:6160-6170]");
        hipLaunchKernelGGL(policy_loss_kernel, dim3(cdiv(R, 128)), dim3(128), 0, s, p);
        D4_LAUNCH_CHECK();
    }
    // ---- value head forward (saved) + HL-Gauss cross entropy
    const int vld = (c.value_num_bins + 3) / 4 * 4;
    if ((rc = mlp_forward(e, e->value, io->agent_embed, D, R, e->l_vbins, vld, save_v, s))) return rc;
    {
        const float sigma = c.hl_gauss_sigma_to_bin_ratio * (c.value_max - c.value_min) / (float)c.value_num_bins;
        if (c.reward_encoder_type == 1)
            hipLaunchKernelGGL(value_loss_kernel<true>, dim3(cdiv(R, 4)), dim3(256), 0, s, e->l_vbins, vld, e->l_returns, mask_keep, scal,
                               e->value_support, R, c.value_num_bins, sqrtf(2.f) * sigma, c.hl_gauss_eps, c.value_min, c.value_max, e->l_dvbins, row_vl);
        else
            hipLaunchKernelGGL(value_loss_kernel<false>, dim3(cdiv(R, 4)), dim3(256), 0, s, e->l_vbins, vld, e->l_returns, mask_keep, scal,
                               e->value_support, R, c.value_num_bins, sqrtf(2.f) * sigma, c.hl_gauss_eps, c.value_min, c.value_max, e->l_dvbins, row_vl);
        D4_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(reduce1_kernel<RED_SUM>, dim3(1), dim3(1024), 0, s, row_pl, nullptr, scal, (int64_t)R, scal + 4);
    hipLaunchKernelGGL(reduce1_kernel<RED_SUM>, dim3(1), dim3(1024), 0, s, row_ent, nullptr, scal, (int64_t)R, scal + 5);
    hipLaunchKernelGGL(reduce1_kernel<RED_SUM>, dim3(1), dim3(1024), 0, s, row_aux, nullptr, scal, (int64_t)R, scal + 6);
    hipLaunchKernelGGL(reduce1_kernel<RED_SUM>, dim3(1), dim3(1024), 0, s, row_vl, nullptr, scal, (int64_t)R, scal + 7);
    D4_LAUNCH_CHECK();
    if (io->allreduce_sum && (rc = io->allreduce_sum(scal + 4, 4, io->allreduce_user))) { set_error("allreduce callback failed (%d)", rc); return 5; }
    hipLaunchKernelGGL(finalize_losses_kernel, dim3(1), dim3(1), 0, s, scal, io->losses, io->objective, c.policy_entropy_weight,
                       io->objective == 2 ? c.pmpo_kl_div_loss_weight : 0.f);
    D4_LAUNCH_CHECK();

    // ---- backward: unembed head 0, policy MLP, value MLP
    const int mtp4d = c.multi_token_pred_len * 4 * D;
    // pe (policy MLP output) was saved as the last layer's z
    float *pe_x, *pe_xh, *pe_saved;
    e->policy.save_ptrs(save_p, R, e->policy.nl - 1, &pe_x, &pe_xh, &pe_saved);
    float* dpe = e->l_dpe;
    if (na > 0) {
        D4_REQUIRE(e->action_unembed_grad, "learner: discrete_action_unembed was bound without a gradient buffer");
        if ((rc = fill_f32(e->action_unembed_grad, 0.f, (int64_t)A * mtp4d, s))) return rc;
        // dU0[a][k] = sum_r dlogits[r][a] * pe[r][k]
        if ((rc = gemm_dw_l(e, e->l_dlogits, Apad, pe_saved, 4 * D, e->action_unembed_grad, mtp4d, A, 4 * D, R, s))) return rc;
        // dpe[r][k] = sum_a dlogits[r][a] * U0[a][k]      (K = Apad; pad rows of U0 are never read: mask K to A)
        if ((rc = gemm_l(e->l_dlogits, Apad, e->action_unembed, mtp4d, dpe, 4 * D, R, 4 * D, A, GEMM_TRANS_B, s))) return rc;
    }
    if (nc > 0) {
        D4_REQUIRE(e->cont_unembed_grad, "learner: continuous_action_unembed was bound without a gradient buffer");
        // d cu_w[j][k] = sum_r dcparams[r][j] * pe[r][k], scattered back into the [nc][mtp][4D][2] layout (other heads: zero)
        if ((rc = gemm_dw_l(e, e->l_dcparams, Cpad, pe_saved, 4 * D, e->l_cu_g, 4 * D, 2 * nc, 4 * D, R, s))) return rc;
        if ((rc = cunembed_scatter_grad(e->l_cu_g, e->cont_unembed_grad, nc, c.multi_token_pred_len, 4 * D, s))) return rc;
        // dpe (+)= dcparams . cu_w
        if ((rc = gemm_l(e->l_dcparams, Cpad, e->cu_w, 4 * D, dpe, 4 * D, R, 4 * D, 2 * nc, GEMM_TRANS_B | (na > 0 ? GEMM_ACCUMULATE : 0), s))) return rc;
    }
    if ((rc = mlp_backward(e, e->policy, save_p, R, dpe, 4 * D, s, io->d_agent_embed_policy))) return rc;
    if ((rc = mlp_backward(e, e->value, save_v, R, e->l_dvbins, vld, s, io->d_agent_embed_value))) return rc;
    return 0;
}

// ------------------------------------------------------------------------------------ clip_grad_norm_ + AdamW
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* g, int64_t n, float* partial) {
    __shared__ float sh[256];
    float s = 0.f;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += g[i] * g[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void norm_final_kernel(const float* partial, int np, float grad_scale, float* norm_out) {
    __shared__ float sh[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *norm_out = sqrtf(sh[0]) * grad_scale;
}

__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n, const float* norm, float max_norm,
                             float grad_scale, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2) {
    float coef = grad_scale;
    if (max_norm > 0.f) {
        const float c = max_norm / (*norm + 1e-6f);           // torch.nn.utils.clip_grad_norm_
        coef *= c < 1.f ? c : 1.f;
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

}  // namespace d4

extern "C" {

int d4_gae(const float* rewards, const float* values, const int64_t* lens, const uint8_t* is_truncated,
           const uint8_t* terminals, float gamma, float lam, int batch, int time, float* returns, void* stream) {
    D4_REQUIRE(rewards && values && returns, "null argument");
    hipLaunchKernelGGL(d4::gae_kernel, dim3(d4::cdiv(batch, 128)), dim3(128), 0, static_cast<hipStream_t>(stream), rewards, values, lens,
                       is_truncated, terminals, gamma, lam, batch, time, returns, (float*)nullptr, (float*)nullptr);
    D4_LAUNCH_CHECK();
    return 0;
}

// Stateless form of the value branch's loss (D4:6254-6295: HL-Gauss / two-hot targets of the returns, cross entropy against the value bins,
// mean over the learnable steps), forward + backward in one call:  loss[0] = sum_r mask[r] * CE_r / max(sum mask, 1),  dlogits = d loss / d logits.
// support: [bins + 1] bin edges (HL-Gauss) or [bins] bin values (two-hot), device.  mask: [rows] or null (all rows).  scratch >= 2 * rows + 64 floats.
int d4_hl_gauss_ce(const float* logits, int ld, const float* targets, const float* mask, const float* support, int rows, int bins, float vmin,
                   float vmax, float sigma, float eps, int two_hot, float* loss, float* dlogits, float* scratch, void* stream) {
    D4_REQUIRE(logits && targets && support && loss && dlogits && scratch && rows >= 1 && bins >= 1 && ld >= bins, "d4_hl_gauss_ce: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* scal = scratch;                     // [64]: [1] count, [7] summed row losses
    float* ones = scratch + 64;                // [rows]
    float* row_loss = ones + rows;             // [rows]
    D4_HIP(hipMemsetAsync(scal, 0, 64 * sizeof(float), s));
    if (!mask) { if (int rc = d4::fill_f32(ones, 1.f, rows, s)) return rc; mask = ones; }
    hipLaunchKernelGGL(d4::reduce1_kernel<d4::RED_SUM>, dim3(1), dim3(1024), 0, s, mask, (const float*)nullptr, scal, (int64_t)rows, scal + 1);
    if (two_hot)
        hipLaunchKernelGGL(d4::value_loss_kernel<true>, dim3(d4::cdiv(rows, 4)), dim3(256), 0, s, logits, ld, targets, mask, scal, support, rows, bins,
                           sqrtf(2.f) * sigma, eps, vmin, vmax, dlogits, row_loss);
    else
        hipLaunchKernelGGL(d4::value_loss_kernel<false>, dim3(d4::cdiv(rows, 4)), dim3(256), 0, s, logits, ld, targets, mask, scal, support, rows, bins,
                           sqrtf(2.f) * sigma, eps, vmin, vmax, dlogits, row_loss);
    hipLaunchKernelGGL(d4::reduce1_kernel<d4::RED_SUM>, dim3(1), dim3(1024), 0, s, row_loss, (const float*)nullptr, scal, (int64_t)rows, scal + 7);
    hipLaunchKernelGGL(d4::masked_mean_final_kernel, dim3(1), dim3(1), 0, s, scal, loss);
    D4_LAUNCH_CHECK();
    return 0;
}

// Stateless form of the policy branch's loss for discrete actions (D4:6077-6242: joint log-prob of the stored actions under MultiCategorical
// logits, PPO clipped surrogate (objective 0, D4:6204-6212) or SPO (1, D4:6188-6198) against the behaviour log-probs, entropy bonus, masked mean
// over the learnable steps), forward + backward in one call — the fused kernel d4_learn runs, with the advantages taken as given (normalise them
// before):  loss[0] = sum_r mask[r] (pl_r - entropy_weight * H_r) / max(sum mask, 1),  dlogits [rows][ld] = d loss / d logits.
// action_sizes: device int32 [na], total = their sum <= ld.  mask: [rows] or null.  scratch >= 5 * rows + 64 floats.
int d4_ppo_policy_loss(const float* logits, int ld, const int64_t* actions, const float* old_log_probs, const float* advantages, const float* mask,
                       const int32_t* action_sizes, int rows, int na, int total, int objective, float eps_clip, float entropy_weight, float* loss,
                       float* dlogits, float* scratch, void* stream) {
    D4_REQUIRE(logits && actions && old_log_probs && advantages && action_sizes && loss && dlogits && scratch && rows >= 1 && na >= 1 && total >= 1 && ld >= total,
               "d4_ppo_policy_loss: bad arguments");
    D4_REQUIRE(objective == 0 || objective == 1, "d4_ppo_policy_loss: objective 0 (ppo) or 1 (spo)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* scal = scratch;                     // [64]: [1] count, [4] sum pl, [5] sum -entropy
    float* ones = scratch + 64;                // [rows]
    float* row_pl = ones + rows, *row_ent = row_pl + rows, *row_aux = row_ent + rows;
    D4_HIP(hipMemsetAsync(scal, 0, 64 * sizeof(float), s));
    if (!mask) { if (int rc = d4::fill_f32(ones, 1.f, rows, s)) return rc; mask = ones; }
    hipLaunchKernelGGL(d4::reduce1_kernel<d4::RED_SUM>, dim3(1), dim3(1024), 0, s, mask, (const float*)nullptr, scal, (int64_t)rows, scal + 1);
    d4::PolicyLossArgs p{};
    p.logits = logits; p.ld = ld; p.actions = actions; p.old_lp = old_log_probs; p.adv_raw = advantages; p.mask = mask; p.scal = scal;
    p.action_sizes = action_sizes; p.dlogits = dlogits; p.row_pl = row_pl; p.row_ent = row_ent; p.row_aux = row_aux;
    p.R = rows; p.na = na; p.A = total; p.objective = objective; p.normalize = 0; p.use_gate = 0; p.clip = eps_clip; p.ent_w = entropy_weight;
    p.eps = 1e-6f; p.gate_temp = 1.f;
    hipLaunchKernelGGL(d4::policy_loss_kernel, dim3(d4::cdiv(rows, 128)), dim3(128), 0, s, p);
    hipLaunchKernelGGL(d4::reduce1_kernel<d4::RED_SUM>, dim3(1), dim3(1024), 0, s, row_pl, (const float*)nullptr, scal, (int64_t)rows, scal + 4);
    hipLaunchKernelGGL(d4::reduce1_kernel<d4::RED_SUM>, dim3(1), dim3(1024), 0, s, row_ent, (const float*)nullptr, scal, (int64_t)rows, scal + 5);
    hipLaunchKernelGGL(d4::finalize_losses_kernel, dim3(1), dim3(1), 0, s, scal, scratch + 32, objective, entropy_weight, 0.f);
    D4_HIP(hipMemcpyAsync(loss, scratch + 32, sizeof(float), hipMemcpyDeviceToDevice, s));
    D4_LAUNCH_CHECK();
    return 0;
}

// clip_grad_norm_(params, max_norm) followed by one AdamW step on a flat parameter group
// (trainers.py:1436-1452; torch.optim.AdamW semantics).  scratch: >= 1025 floats.
int d4_adamw_clip(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                  float grad_scale, float* scratch, void* stream) {
    D4_REQUIRE(params && grads && exp_avg && exp_avg_sq && scratch && n > 0 && step >= 1, "adamw: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int np = 1024;
    hipLaunchKernelGGL(d4::sumsq_partial_kernel, dim3(np), dim3(256), 0, s, grads, n, scratch + 1);
    hipLaunchKernelGGL(d4::norm_final_kernel, dim3(1), dim3(256), 0, s, scratch + 1, np, grad_scale, scratch);
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(d4::adamw_kernel, dim3((unsigned)g), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, n, scratch, max_grad_norm,
                       grad_scale, lr, beta1, beta2, eps, weight_decay, bc1, bc2);
    D4_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
