// Attention cores of the imagination path: everything between the Q/K/V projection GEMM and the
// output projection GEMM of reference Attention.forward (D4:2003-2064) in one kernel each.
//
//   value-residual lerp (D4:2005-2012) -> K head-RMSNorm (D4:1663-1679, 2017) -> rotary (time
//   layers, D4:1626-1659) -> [KV cache] -> q.k^T * dh^-1/2 -> softclamp 50*tanh(s/50) (D4:527) ->
//   special-token / causal mask (D4:1738, 1781) -> softmax -> .v -> belief projection
//   (D4:2049-2054) -> sigmoid head gates (D4:2058-2060)
//
// Head dim is 64 == one wavefront: lane = feature index, one wave per (sequence, head).  Row dot
// products are DPP row reductions + readlane; scores therefore live in SGPRs.  These kernels are
// HBM/latency bound (S = 15 tokens per frame): each q/k/v element is read exactly once per wave.
#include "common.h"
#include <type_traits>
#include "kernels.h"
#include "prof.h"
#include "attn_mfma.h"
#include "pool_mix_row.h"
#include <float.h>
#include <stdlib.h>

namespace d4 {

// Generic form: one block (4 waves) per (group, head).  Wave w prepares keys w, w+4, ... (value-residual mix, key l2-norm)
// into LDS, then owns queries w, w+4, ...; scores are wave reductions, so they live in SGPRs.
template <int NKM, int DH>
__global__ __launch_bounds__(256) void small_attn_kernel(SmallAttnArgs p) {
    __shared__ float Ks[NKM * 64], Vs[NKM * 64];
    const int w = threadIdx.x >> 6;
    const int g = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int lane = threadIdx.x & 63;
    const int nk = p.nk, nq = p.nq;
    constexpr int dh = DH;                                          // compile-time: the 64-wide case keeps no lane predicate
    const bool act = DH == 64 || lane < dh;                         // head dims below 64: the upper lanes carry zeros
    const int hl = h * dh + (act ? lane : 0);
    const float kscale = act ? (p.k_gamma[hl] + 1.f) * sqrtf((float)dh) : 0.f;   // (gamma + 1) * sqrt(dh)
    const float qscale = rsqrtf((float)dh);

    {
        constexpr int PER = (NKM + 3) / 4;
        float kr[PER], vr[PER], rr[PER], wm[PER];
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int j = w + 4 * t;
            kr[t] = vr[t] = rr[t] = wm[t] = 0.f;
            if (j < nk) {
                kr[t] = act ? p.k[g * p.k_group_stride + j * p.k_item_stride + hl] : 0.f;
                vr[t] = act ? p.v[g * p.v_group_stride + j * p.v_item_stride + hl] : 0.f;
                if (p.vres) {
                    rr[t] = act ? p.vres[g * p.r_group_stride + j * p.r_item_stride + hl] : 0.f;
                    wm[t] = p.mix[g * p.m_group_stride + j * p.m_item_stride + h];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int j = w + 4 * t;
            if (j < nk) {
                float vj = vr[t];
                if (p.vres) vj = lerp_torch(vj, rr[t], sigmoidf(wm[t]));
                const float nrm = sqrtf(wave_sum(kr[t] * kr[t]));
                Ks[j * 64 + lane] = kr[t] / fmaxf(nrm, 1e-12f) * kscale;
                Vs[j * 64 + lane] = vj;
            }
        }
    }
    __syncthreads();
    if (w >= nq) return;

    float K[NKM], V[NKM];
#pragma unroll
    for (int j = 0; j < NKM; ++j) {
        K[j] = j < nk ? Ks[j * 64 + lane] : 0.f;
        V[j] = j < nk ? Vs[j * 64 + lane] : 0.f;
    }

    for (int i = w; i < nq; i += 4) {
        const float qi = act ? p.q[g * p.q_group_stride + i * p.q_item_stride + hl] : 0.f;
        float s[NKM];
        float m = -FLT_MAX;
        const bool ordinary_q = p.mask_special > 0 && i < nq - p.mask_special;
#pragma unroll
        for (int j = 0; j < NKM; ++j) {
            s[j] = -FLT_MAX;
            if (j < nk) {
                float sc = wave_sum(qi * K[j]) * qscale;
                if (p.softclamp > 0.f) sc = tanhf(sc / p.softclamp) * p.softclamp;
                if (ordinary_q && j >= nk - p.mask_special) sc = -FLT_MAX;
                s[j] = sc;
                m = fmaxf(m, sc);
            }
        }
        float l = 0.f, acc = 0.f;
#pragma unroll
        for (int j = 0; j < NKM; ++j) {
            if (j < nk) {
                float e = expf(s[j] - m);
                l += e;
                acc += e * V[j];
            }
        }
        float o = acc / l;
        if (p.belief) {
            // v_i (self attention): the mixed value row of token i
            const float vi = Vs[i * 64 + lane];
            float vn = vi / fmaxf(sqrtf(wave_sum(vi * vi)), 1e-12f);
            o -= wave_sum(o * vn) * vn;
        }
        if (p.gate) o *= sigmoidf(p.gate[g * p.g_group_stride + i * p.g_item_stride + h]);
        if (act) p.out[g * p.o_group_stride + i * p.o_item_stride + hl] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// Self attention over the <= 16 tokens of one frame (the trunk's space layers), LDS-staged.
// The generic kernel above keeps scores in SGPRs, so softclamp's tanh and the softmax exp are evaluated
// once per (query, key) by ALL 64 lanes redundantly (225 x tanhf x 64 lanes): it is issue-bound.  Here the
// 16x16 score matrix is spread over the lanes (lane -> 4 (i, j) pairs), so every transcendental is useful
// work, and a softmax row is one 16-lane DPP row (row_max16 / row_sum16).
//   per wave: one (frame, head);  LDS per wave: Q,K [16][68] (+4 pad: conflict-free ds_read_b128), P [16][16]
constexpr int SA_LD = 68;

// One block (4 waves) per (frame, head); n = nq = nk <= 16 tokens.  Wave w prepares tokens w, w+4, w+8, w+12 (value-residual
// mix, key l2-norm, 1/|v| for the belief projection) into LDS, then owns query rows 4w..4w+3: scores for the four rows in one
// pass (lane = (row, key) pair), softmax over 16-lane rows, P.V with lane = feature.  8192 short waves instead of 2048 long
// ones: the kernel is a chain of dependent reductions, so the win is latency hiding, not bandwidth.
template <int DH>
__global__ __launch_bounds__(256) void space_attn_kernel(SmallAttnArgs p) {
    __shared__ __attribute__((aligned(16))) float Qs[16 * SA_LD];
    __shared__ __attribute__((aligned(16))) float Ks[16 * SA_LD];
    __shared__ __attribute__((aligned(16))) float Vs[16 * SA_LD];
    __shared__ __attribute__((aligned(16))) float Ps[16 * 16];
    __shared__ float vinv_s[16];
    const int w = threadIdx.x >> 6;
    const int g = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int lane = threadIdx.x & 63;
    const int n = p.nk;                                   // == nq <= 16
    constexpr int dh = DH;                                // compile-time: the 64-wide case keeps no lane predicate
    const bool act = DH == 64 || lane < dh;               // head dims below 64: the upper lanes carry zeros
    const int hl = h * dh + (act ? lane : 0);
    const float kscale = act ? (p.k_gamma[hl] + 1.f) * sqrtf((float)dh) : 0.f;
    const float qscale = rsqrtf((float)dh);

    // phase 1: all global loads of this wave's tokens before the first dependent reduction
    float V[4], Kr[4], Qr[4], Rr[4], Wm[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = w + 4 * t;
        V[t] = Kr[t] = Qr[t] = Rr[t] = Wm[t] = 0.f;
        if (j < n) {
            Kr[t] = act ? p.k[g * p.k_group_stride + j * p.k_item_stride + hl] : 0.f;
            Qr[t] = act ? p.q[g * p.q_group_stride + j * p.q_item_stride + hl] : 0.f;
            V[t] = act ? p.v[g * p.v_group_stride + j * p.v_item_stride + hl] : 0.f;
            if (p.vres) {
                Rr[t] = act ? p.vres[g * p.r_group_stride + j * p.r_item_stride + hl] : 0.f;
                Wm[t] = p.mix[g * p.m_group_stride + j * p.m_item_stride + h];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = w + 4 * t;
        float kj = Kr[t], vi = 0.f;
        if (j < n) {
            if (p.vres) V[t] = lerp_torch(V[t], Rr[t], sigmoidf(Wm[t]));
            const float nrm = sqrtf(wave_sum(kj * kj));
            kj = kj / fmaxf(nrm, 1e-12f) * kscale;
            if (p.belief) vi = 1.f / fmaxf(sqrtf(wave_sum(V[t] * V[t])), 1e-12f);
        }
        Ks[j * SA_LD + lane] = kj;
        Qs[j * SA_LD + lane] = Qr[t];
        Vs[j * SA_LD + lane] = V[t];
        if (lane == 0) vinv_s[j] = vi;
    }
    __syncthreads();

    // scores of query rows 4w .. 4w+3: pair (i, j) = (4w + lane/16, lane%16)
    const int jl = lane & 15, il = lane >> 4;
    {
        const int i = 4 * w + il;
        const f32x4* qrow = reinterpret_cast<const f32x4*>(Qs + i * SA_LD);
        const f32x4* krow = reinterpret_cast<const f32x4*>(Ks + jl * SA_LD);
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 a = qrow[c], b = krow[c];
            acc += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
        }
        float sc = acc * qscale;
        if (p.softclamp > 0.f) sc = tanhf(sc / p.softclamp) * p.softclamp;
        const bool ordinary_q = p.mask_special > 0 && i < n - p.mask_special;
        if (ordinary_q && jl >= n - p.mask_special) sc = -FLT_MAX;
        const bool valid = jl < n && i < n;
        const float m = row_max16(valid ? sc : -FLT_MAX);
        const float e = valid ? expf(sc - m) : 0.f;
        const float l = row_sum16(e);
        Ps[i * 16 + jl] = (i < n) ? e / l : 0.f;
    }
    // (each wave reads back only the P rows it wrote: program order + lgkmcnt is enough)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // out[i][d] = sum_j P[i][j] V[j][d]   (lane = d; P rows are LDS broadcasts)
    float Vr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) Vr[j] = Vs[j * SA_LD + lane];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = 4 * w + t;
        if (i >= n) break;
        int orank = i;
        if (p.q_hi > 0) {
            if (i >= p.q_lo && i < p.q_hi) orank = i - p.q_lo;
            else if (p.q_last && i == n - 1) orank = p.q_hi - p.q_lo;
            else continue;
        }
        const f32x4* prow = reinterpret_cast<const f32x4*>(Ps + i * 16);
        float o = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 pv = prow[c];
            o += pv[0] * Vr[4 * c] + pv[1] * Vr[4 * c + 1] + pv[2] * Vr[4 * c + 2] + pv[3] * Vr[4 * c + 3];
        }
        if (p.belief) {
            const float vn = Vs[i * SA_LD + lane] * vinv_s[i];
            o -= wave_sum(o * vn) * vn;
        }
        if (p.gate) o *= sigmoidf(p.gate[g * p.g_group_stride + i * p.g_item_stride + h]);
        if (act) p.out[g * p.o_group_stride + orank * p.o_item_stride + hl] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// The same self attention (<= 16 tokens of one frame, head dim 64) on the matrix pipe — round 3.  The LDS-staged kernel above is
// INSTRUCTION-bound, not memory-bound (rocprofv3: 33 k wave-cycles per wave, ~1100 issued instructions per wave x 4 waves per (frame, head), most of
// them per-token scalar work replicated over lanes): it runs at 0.21 of the HBM rate it could stream at.  Here ONE wave owns a (frame, head):
//   * q, k, v, value residual are read as float4 per lane in the MFMA operand layout (lane = (token = l & 15, feature quarter kq = l >> 4):
//     features 16 s + 4 kq .. + 3 for s = 0..3), so the 16 x 16 score matrix is 16 v_mfma_f32_16x16x4_f32 and P.V another 16 — exact fp32 FMAs;
//   * the key l2-norm is a per-key scale of the score COLUMN ((gamma + 1) goes onto q, sqrt(dh) cancels against the query scale), so no key is
//     ever rewritten; S^T = K Q^T is computed, whose accumulator layout (lane holds P[i = l & 15][4 kq .. + 3]) IS the A-operand layout of P.V;
//   * per-token reductions (|k|, |v|) are 16 FMAs + two cross-row shuffles for all 16 tokens at once; softmax is 4 values per lane + two shuffles.
// ~10x fewer issued instructions per (frame, head); V' (the mixed values) is the only LDS-staged operand.
// The kernel is a template over the number of 16-row query tiles QT and key tiles KT: <1, 1> is the within-frame self attention (value residual,
// special-token mask, belief projection, restricted query set), <1, 2> / <2, 1> / <1, 1> the small cross forms (learned-query pools in / out with up
// to 32 latents or queries, the agent token's cross attention).
template <int QT, int KT, int NWB = 4>            // NWB waves (= (group, head) units) per block: 2 for the 64-key form, whose V' tiles are 17 KB per wave
__global__ __launch_bounds__(NWB * 64) void attn_mfma_kernel(SmallAttnArgs p) {
    __shared__ __attribute__((aligned(16))) float Vs_all[NWB][KT * 16 * SM_LDV];
    __shared__ __attribute__((aligned(16))) float kinv_all[NWB][KT * 16];
    __shared__ __attribute__((aligned(16))) float vinv_all[NWB][KT * 16];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int unit = blockIdx.x * NWB + w;
    if (unit >= p.groups * p.heads) return;
    const int g = unit / p.heads, h = unit % p.heads;
    float* op = p.out ? p.out + g * p.o_group_stride + h * 64 : nullptr;
    uint16_t* ob = p.out_b ? p.out_b + g * p.o_group_stride + h * 64 : nullptr;
    const int64_t ois = p.o_item_stride;
    attn_mfma_unit<QT, KT>(p, g, h, lane, Vs_all[w], kinv_all[w], vinv_all[w], [&](int orank, int t, int tok, float v) {
        if (op) op[(int64_t)orank * ois + 16 * t + tok] = v;
        if (ob) ob[(int64_t)orank * ois + 16 * t + tok] = bf16_bits(v);
    });
}

// ---------------------------------------------------------------------------------------------
// Attention over up to ATTN_MAXK keys per (group, head) with head dim 64: the space layers of the video tokenizer's decoder
// (~100 tokens per frame: patches + latents, D4:3654-3668).  One 4-wave block per (group, head):
//   phase 1  the keys are prepared ONCE, cooperatively, into LDS: value-residual lerp on V, K l2-norm * (gamma + 1) * sqrt(dh), 1 / |v|
//   phase 2  wave w owns queries w, w + 4, ...; lanes are spread over (key residue j & 3, feature group of 4) exactly as in
//            time_attn64_kernel, so one pass scores four keys with a 16-lane DPP row reduction, softclamp / exp run for four keys at
//            once, and K / V fragments are conflict-free 16-byte LDS reads; keys go in chunks of 64 with an online softmax.
constexpr int ATTN_MAXK = 160;
__global__ __launch_bounds__(256) void attn_wide_kernel(SmallAttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) float wide_s[];            // K [nk][64] | V [nk][64] | 1/|v| [nk]
    const int nk = p.nk, nq = p.nq;
    float* Ks = wide_s;
    float* Vs = wide_s + nk * 64;
    float* vinv_s = Vs + nk * 64;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int hl = h * 64 + lane;
    const float kscale = (p.k_gamma[hl] + 1.f) * 8.f;
    for (int j = w; j < nk; j += 4) {
        float kj = p.k[g * p.k_group_stride + j * p.k_item_stride + hl];
        float vj = p.v[g * p.v_group_stride + j * p.v_item_stride + hl];
        if (p.vres) vj = lerp_torch(vj, p.vres[g * p.r_group_stride + j * p.r_item_stride + hl], sigmoidf(p.mix[g * p.m_group_stride + j * p.m_item_stride + h]));
        const float nrm = sqrtf(wave_sum(kj * kj));
        Ks[j * 64 + lane] = kj / fmaxf(nrm, 1e-12f) * kscale;
        Vs[j * 64 + lane] = vj;
        if (p.belief) { const float vn = sqrtf(wave_sum(vj * vj)); if (lane == 0) vinv_s[j] = 1.f / fmaxf(vn, 1e-12f); }
    }
    __syncthreads();

    const int fg = lane & 15, kr = lane >> 4;
    const f32x4* kt = reinterpret_cast<const f32x4*>(Ks);
    const f32x4* vt = reinterpret_cast<const f32x4*>(Vs);
    for (int i = w; i < nq; i += 4) {
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(p.q + g * p.q_group_stride + i * p.q_item_stride + h * 64 + fg * 4);
        // ordinary queries may not see the trailing special keys (D4:1781)
        const int vis = (p.mask_special > 0 && i < nq - p.mask_special) ? nk - p.mask_special : nk;
        float m = -FLT_MAX, l = 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < vis; c0 += 64) {
            const int cn = min(64, vis - c0);
            const int passes = (cn + 3) / 4;
            float sc[16];
            float cm = -FLT_MAX;
#pragma unroll
            for (int ps = 0; ps < 16; ++ps) {
                sc[ps] = -FLT_MAX;
                if (ps < passes) {
                    const int j = ps * 4 + kr;
                    const bool ok = j < cn;
                    const f32x4 k4 = ok ? kt[(c0 + j) * 16 + fg] : f32x4{0.f, 0.f, 0.f, 0.f};
                    float d = row_sum16(q4[0] * k4[0] + q4[1] * k4[1] + q4[2] * k4[2] + q4[3] * k4[3]) * 0.125f;
                    if (p.softclamp > 0.f) d = tanhf(d / p.softclamp) * p.softclamp;
                    sc[ps] = ok ? d : -FLT_MAX;
                    cm = fmaxf(cm, sc[ps]);
                }
            }
            cm = fmaxf(cm, __shfl_xor(cm, 16));
            cm = fmaxf(cm, __shfl_xor(cm, 32));
            const float mn = fmaxf(m, cm);
            const float alpha = expf(m - mn);
            l *= alpha;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] *= alpha;
#pragma unroll
            for (int ps = 0; ps < 16; ++ps) {
                if (ps < passes && sc[ps] > -FLT_MAX) {
                    const float e_ = expf(sc[ps] - mn);
                    const f32x4 v4 = vt[(c0 + ps * 4 + kr) * 16 + fg];
                    l += e_;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += e_ * v4[e];
                }
            }
            m = mn;
        }
        l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] += __shfl_xor(acc[e], 16); acc[e] += __shfl_xor(acc[e], 32); }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[e] / l;
        if (p.belief) {                                       // self attention: orthogonalise against token i's own (mixed) value
            const f32x4 vi = vt[i * 16 + fg];
            const float inv = vinv_s[i];
            const float dot = row_sum16(o[0] * vi[0] + o[1] * vi[1] + o[2] * vi[2] + o[3] * vi[3]) * inv;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] -= dot * (vi[e] * inv);
        }
        if (p.gate) {
            const float gt = sigmoidf(p.gate[g * p.g_group_stride + i * p.g_item_stride + h]);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] *= gt;
        }
        if (kr == 0) *reinterpret_cast<f32x4*>(p.out + g * p.o_group_stride + i * p.o_item_stride + h * 64 + fg * 4) = o;
    }
}

static int small_attn_impl(const SmallAttnArgs& p, hipStream_t stream, bool* wrote_b) {
    *wrote_b = false;
    if (p.nk > 64 || (p.nq == p.nk && p.nk > 16 && p.belief && p.dh == 64)) {
        // wide form (tokenizer decoder): needs 16-byte aligned rows for the float4 q loads / out stores
        D4_REQUIRE(p.nk <= ATTN_MAXK && p.dh == 64, "attention: %d keys (max %d) / head dim %d (64) not supported by the wide kernel", p.nk, ATTN_MAXK, p.dh);
        D4_REQUIRE(p.q_lo == 0 && p.q_hi == 0, "attention: query restriction is not implemented in the wide kernel");
        const size_t lds = (size_t)(2 * p.nk * 64 + p.nk) * sizeof(float);
        static DeviceOnce attr_set;
        if (attr_set.need()) {
            D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((2 * ATTN_MAXK * 64 + ATTN_MAXK) * sizeof(float))));
            attr_set.done();
        }
        if (p.groups * p.heads == 0) return 0;
        const double wide_bytes = 4.0 * p.groups * p.heads * p.dh * ((double)p.nq * 2 + (double)p.nk * (p.vres ? 3 : 2));
        D4_GLUE_LAUNCH(GL_ATTN_WIDE, wide_bytes, attn_wide_kernel, dim3(p.groups * p.heads), dim3(256), lds, stream, p);
        D4_LAUNCH_CHECK();
        return 0;
    }
    D4_REQUIRE(p.nk >= 1 && p.nk <= 64, "small_attn: nk=%d out of range [1,64]", p.nk);
    D4_REQUIRE(p.dh == 16 || p.dh == 32 || p.dh == 64, "small_attn: head dim %d (16, 32 or 64)", p.dh);
    D4_REQUIRE(!p.belief || p.nq == p.nk, "small_attn: belief needs self attention");
    const int waves = p.groups * p.heads;
    if (waves == 0) return 0;
    dim3 block(256);
    if (p.nq == p.nk && p.nk <= 16 && p.nq >= 8 && p.q_group_stride != 0) {
        // algorithmic bytes: q, k, v (+ value residual) rows of every (frame, head) read once, the kept query rows written once
        const int nq_out = p.q_hi > 0 ? (p.q_hi - p.q_lo + p.q_last) : p.nq;
        const double sp_bytes = 4.0 * p.groups * p.heads * (p.dh * ((double)p.nk * (p.vres ? 4 : 3) + nq_out) + 2.0 * p.nk);
        // head dim 64 with 16-byte aligned rows: one wave per (frame, head) on the matrix pipe (else the LDS-staged VALU form)
        auto al4 = [](const void* q, int64_t a, int64_t b) { return ((uintptr_t)q % 16) == 0 && (a % 4) == 0 && (b % 4) == 0; };
        const bool mfma_ok = p.dh == 64 && al4(p.q, p.q_group_stride, p.q_item_stride) && al4(p.k, p.k_group_stride, p.k_item_stride) &&
                             al4(p.v, p.v_group_stride, p.v_item_stride) && (!p.vres || al4(p.vres, p.r_group_stride, p.r_item_stride)) &&
                             ((uintptr_t)p.k_gamma % 16) == 0;
        *wrote_b = mfma_ok;
        if (mfma_ok) D4_GLUE_LAUNCH(GL_SPACE_ATTN, sp_bytes, (attn_mfma_kernel<1, 1>), dim3(cdiv(waves, 4)), block, 0, stream, p);
        else if (p.dh == 64) D4_GLUE_LAUNCH(GL_SPACE_ATTN, sp_bytes, space_attn_kernel<64>, dim3(waves), block, 0, stream, p);
        else if (p.dh == 32) hipLaunchKernelGGL(space_attn_kernel<32>, dim3(waves), block, 0, stream, p);
        else hipLaunchKernelGGL(space_attn_kernel<16>, dim3(waves), block, 0, stream, p);
        D4_LAUNCH_CHECK();
        return 0;
    }
    // algorithmic bytes: a batch-independent operand (group stride 0) is counted once
    auto al4s = [](const void* q, int64_t a, int64_t b) { return ((uintptr_t)q % 16) == 0 && (a % 4) == 0 && (b % 4) == 0; };
    const bool mfma_small = p.dh == 64 && p.nq <= 64 && p.nk <= 64 && (p.nq <= 16 || p.nk <= 16) && p.q_hi == 0 &&
                            al4s(p.q, p.q_group_stride, p.q_item_stride) && al4s(p.k, p.k_group_stride, p.k_item_stride) &&
                            al4s(p.v, p.v_group_stride, p.v_item_stride) && (!p.vres || al4s(p.vres, p.r_group_stride, p.r_item_stride)) &&
                            ((uintptr_t)p.k_gamma % 16) == 0;
    const double sm_bytes = 4.0 * p.heads * p.dh * ((p.q_group_stride ? (double)p.groups : 1.0) * p.nq + (p.k_group_stride ? (double)p.groups : 1.0) * p.nk * 2 + (double)p.groups * p.nq);
#define D4_SMALL_ATTN(NK)                                                                                       \
    do {                                                                                                          \
        if (p.dh == 64) D4_GLUE_LAUNCH(GL_SMALL_ATTN, sm_bytes, (small_attn_kernel<NK, 64>), dim3(waves), block, 0, stream, p);        \
        else if (p.dh == 32) hipLaunchKernelGGL((small_attn_kernel<NK, 32>), dim3(waves), block, 0, stream, p);   \
        else hipLaunchKernelGGL((small_attn_kernel<NK, 16>), dim3(waves), block, 0, stream, p);                   \
    } while (0)
    *wrote_b = mfma_small;
    if (mfma_small) {         // one wave per (group, head) on the matrix pipe (attn_mfma_kernel): up to 32 x 16 or 16 x 32 (queries x keys)
        if (p.nq <= 16 && p.nk <= 16) D4_GLUE_LAUNCH(GL_SMALL_ATTN, sm_bytes, (attn_mfma_kernel<1, 1>), dim3(cdiv(waves, 4)), block, 0, stream, p);
        else if (p.nq <= 16 && p.nk <= 32) D4_GLUE_LAUNCH(GL_SMALL_ATTN, sm_bytes, (attn_mfma_kernel<1, 2>), dim3(cdiv(waves, 4)), block, 0, stream, p);
        else if (p.nq <= 16) D4_GLUE_LAUNCH(GL_SMALL_ATTN, sm_bytes, (attn_mfma_kernel<1, 4, 2>), dim3(cdiv(waves, 2)), dim3(128), 0, stream, p);    // <= 64 latents -> spatial tokens (cfg 5)
        else if (p.nq <= 32) D4_GLUE_LAUNCH(GL_SMALL_ATTN, sm_bytes, (attn_mfma_kernel<2, 1>), dim3(cdiv(waves, 4)), block, 0, stream, p);
        else D4_GLUE_LAUNCH(GL_SMALL_ATTN, sm_bytes, (attn_mfma_kernel<4, 1>), dim3(cdiv(waves, 4)), block, 0, stream, p);                           // spatial tokens -> <= 64 latents
    }
    else if (p.nk <= 16) D4_SMALL_ATTN(16);
    else if (p.nk <= 32) D4_SMALL_ATTN(32);
    else D4_SMALL_ATTN(64);
#undef D4_SMALL_ATTN
    D4_LAUNCH_CHECK();
    return 0;
}

// bf16 engine: `out_b` asks for a bf16 copy of the output (the next GEMM's activation image).  The matrix-pipe kernels write it themselves; the
// other forms are followed by one conversion pass over the (contiguous) output rows.
int small_attn(const SmallAttnArgs& p, hipStream_t stream) {
    bool wrote_b = false;
    D4_REQUIRE(p.out != nullptr || p.out_b != nullptr, "small_attn: no output");
    if (p.out_b && !p.out) {
        // only the kernels that write the bf16 copy themselves can run without the fp32 output
        D4_REQUIRE(false, "small_attn: out_b without out is not supported");
    }
    if (int rc = small_attn_impl(p, stream, &wrote_b)) return rc;
    if (!p.out_b || wrote_b || p.groups * p.heads == 0) return 0;
    const int nq_out = p.q_hi > 0 ? (p.q_hi - p.q_lo + p.q_last) : p.nq;
    const int cols = p.heads * p.dh;
    if (nq_out == 1 || p.o_item_stride == 0) return cvt_rows_bf16(p.out, p.o_group_stride, p.out_b, p.o_group_stride, p.groups, cols, stream);
    D4_REQUIRE(p.o_group_stride == (int64_t)nq_out * p.o_item_stride, "small_attn: bf16 output copy needs contiguous output rows");
    return cvt_rows_bf16(p.out, p.o_item_stride, p.out_b, p.o_item_stride, p.groups * nq_out, cols, stream);
}

// ---------------------------------------------------------------------------------------------
// AttentionPool (D4:2143-2177) with the value side restructured (see PoolMixArgs).  One wave per token row.
template <int ITER, bool DEEP = false, bool KB16 = false>
__global__ __launch_bounds__(256) void pool_mix_kernel(PoolMixArgs p) {
    constexpr int PH = 4, LMAX = 64;
    __shared__ float psh[4][LMAX * PH];
    __shared__ f32x4 gws[PH * ITER * 64];                 // head-gate weights [PH][D] of the pool (norm gamma folded)
    const int L = p.L, D = p.D;
    const int nf4 = D / 4;
    for (int i = threadIdx.x; i < PH * ITER * 64; i += 256) {        // (columns past D are zero: they meet zero-padded rows)
        const int h = i / (ITER * 64), c4 = i % (ITER * 64);
        gws[i] = c4 < nf4 ? reinterpret_cast<const f32x4*>(p.gate_w)[h * nf4 + c4] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int wslot = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wslot;
    if (m >= p.M) return;
    const int lane = threadIdx.x & 63;
    pool_mix_row<ITER, DEEP, KB16>(p, m, lane, psh[wslot], gws, [&](int h, int c4, const f32x4& v) {
        if (p.u) reinterpret_cast<f32x4*>(p.u + ((int64_t)m * PH + h) * D)[c4] = v;
        if (p.u_b) store_bf16x4(p.u_b + ((int64_t)m * PH + h) * D + 4 * c4, v);
    });
}

// Few token rows (BASELINE config 4's decode regime: one trajectory = 11 rows): one BLOCK per token row, its four waves split the L
// hiddens (l = w, w + 4, ...) so every wave has all of its loads in flight at once — with one wave per row the kernel is L
// dependent memory round trips long (12 us at L = 13).  The partial mixes are folded through LDS in wave order (fixed).  Chosen by
// M alone (pool_mix below), so a given shape always takes the same arithmetic path.
template <int ITER, bool KB16 = false>
__global__ __launch_bounds__(256) void pool_mix_rows_kernel(PoolMixArgs p) {
    constexpr int PH = 4, LMAX = 64;
    __shared__ float ps[LMAX * PH];
    __shared__ float gsh[PH];
    __shared__ f32x4 accs[4][PH][ITER * 64];              // [wave][head][D / 4]
    const int L = p.L, D = p.D, nf4 = D / 4;
    const int m = blockIdx.x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hh = lane >> 4;
    // every global load of the first round (4 hiddens per wave: all of them up to L = 16) is issued before anything is computed:
    // query, key rows, hidden rows, and for the gating wave the gate weights (and x when it is not the last hidden)
    const f32x4 q4 = pool_query4<KB16>(p, m, lane);
    f32x4 g4 = *reinterpret_cast<const f32x4*>(p.k_gamma + lane * 4);
    auto load_keys = [&](int l0, f32x4 (&kv)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = l0 + 4 * j;
            if (l >= L) kv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            else if constexpr (KB16) {                   // bf16 keys (bf16 engine: the only copy)
                const uint2 raw = *reinterpret_cast<const uint2*>(p.k_b + ((int64_t)l * p.M + m) * p.ldk + lane * 4);
                kv[j] = f32x4{__builtin_bit_cast(float, raw.x << 16), __builtin_bit_cast(float, raw.x & 0xFFFF0000u),
                              __builtin_bit_cast(float, raw.y << 16), __builtin_bit_cast(float, raw.y & 0xFFFF0000u)};
            } else kv[j] = *reinterpret_cast<const f32x4*>(p.k + ((int64_t)l * p.M + m) * p.ldk + lane * 4);
        }
    };
    auto load_hid = [&](int l0, f32x4 (&v)[4][ITER]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = l0 + 4 * j;
            const f32x4* hr = reinterpret_cast<const f32x4*>(p.hid + ((int64_t)(l < L ? l : 0) * p.M + m) * D);
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                const int c4 = lane + 64 * i;
                v[j][i] = (l < L && c4 < nf4) ? hr[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    const bool x_is_last_hidden = p.x == p.hid + (int64_t)(L - 1) * p.M * D && p.ldx == D;
    const bool gating_wave = w == (x_is_last_hidden ? ((L - 1) & 3) : 0);          // l = w (mod 4): the wave that meets hidden L - 1
    f32x4 kv[4], v[4][ITER], gwv[PH][ITER], xv[ITER];
    load_keys(w, kv);
    load_hid(w, v);
    if (gating_wave) {
        const f32x4* gw = reinterpret_cast<const f32x4*>(p.gate_w);
        const f32x4* xr = reinterpret_cast<const f32x4*>(p.x + (int64_t)m * p.ldx);
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int c4 = lane + 64 * i;
#pragma unroll
            for (int h = 0; h < PH; ++h) gwv[h][i] = c4 < nf4 ? gw[h * nf4 + c4] : f32x4{0.f, 0.f, 0.f, 0.f};
            xv[i] = (!x_is_last_hidden && c4 < nf4) ? xr[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) g4[e] = (g4[e] + 1.f) * 8.f;
    for (int l0 = w; l0 < L; l0 += 16) {                  // scores of 4 key rows of this wave per round
        if (l0 != w) load_keys(l0, kv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = l0 + 4 * j;
            const float nrm = sqrtf(row_sum16(kv[j][0] * kv[j][0] + kv[j][1] * kv[j][1] + kv[j][2] * kv[j][2] + kv[j][3] * kv[j][3]));
            const float inv = 1.f / fmaxf(nrm, 1e-12f);
            const float sc = row_sum16(q4[0] * (kv[j][0] * inv * g4[0]) + q4[1] * (kv[j][1] * inv * g4[1]) +
                                       q4[2] * (kv[j][2] * inv * g4[2]) + q4[3] * (kv[j][3] * inv * g4[3])) * 0.125f;
            if (l < L && (lane & 15) == 0) ps[l * PH + hh] = sc;
        }
    }
    __syncthreads();
    // softmax over l per head, once (wave 0): entry idx = l * 4 + h sits in lane idx & 63, so a lane's entries share its head
    // (lane & 3) and the per-head max / sum are butterflies over lane bits 2..5.  ps <- exp(s - max) / sum.
    if (w == 0) {
        float sc[4], mm = -FLT_MAX;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = lane + 64 * j;
            sc[j] = idx < L * PH ? ps[idx] : -FLT_MAX;
            mm = fmaxf(mm, sc[j]);
        }
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) mm = fmaxf(mm, __shfl_xor(mm, o));
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc[j] = lane + 64 * j < L * PH ? expf(sc[j] - mm) : 0.f; d += sc[j]; }
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) d += __shfl_xor(d, o);
        const float inv = 1.f / d;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane + 64 * j < L * PH) ps[lane + 64 * j] = sc[j] * inv;
    }
    __syncthreads();
    f32x4 acc[PH][ITER];
#pragma unroll
    for (int h = 0; h < PH; ++h)
#pragma unroll
        for (int i = 0; i < ITER; ++i) acc[h][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto gate_logits = [&](const f32x4 (&r)[ITER], float rstd) {       // gate_h = sigmoid(RMSNorm(x) . gate_w[h])
#pragma unroll
        for (int h = 0; h < PH; ++h) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < ITER; ++i) d += r[i][0] * gwv[h][i][0] + r[i][1] * gwv[h][i][1] + r[i][2] * gwv[h][i][2] + r[i][3] * gwv[h][i][3];
            d = wave_sum(d) * rstd;
            if (lane == 0) gsh[h] = d;
        }
    };
    for (int l0 = w; l0 < L; l0 += 16) {
        if (l0 != w) load_hid(l0, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = l0 + 4 * j;
            if (l >= L) break;
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < ITER; ++i) ss += v[j][i][0] * v[j][i][0] + v[j][i][1] * v[j][i][1] + v[j][i][2] * v[j][i][2] + v[j][i][3] * v[j][i][3];
            const float rstd = rsqrtf(wave_sum(ss) / (float)D + p.eps);
#pragma unroll
            for (int h = 0; h < PH; ++h) {
                const float wt = ps[l * PH + h] * rstd;
#pragma unroll
                for (int i = 0; i < ITER; ++i) acc[h][i] += v[j][i] * wt;
            }
            if (l == L - 1 && x_is_last_hidden) gate_logits(v[j], rstd);
        }
    }
    if (!x_is_last_hidden && w == 0) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < ITER; ++i) ss += xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1] + xv[i][2] * xv[i][2] + xv[i][3] * xv[i][3];
        gate_logits(xv, rsqrtf(wave_sum(ss) / (float)D + p.eps));
    }
#pragma unroll
    for (int h = 0; h < PH; ++h)
#pragma unroll
        for (int i = 0; i < ITER; ++i) accs[w][h][lane + 64 * i] = acc[h][i];
    __syncthreads();
    for (int idx = threadIdx.x; idx < PH * ITER * 64; idx += 256) {
        const int h = idx / (ITER * 64), c4 = idx % (ITER * 64);
        if (c4 >= nf4) continue;
        const f32x4 sum = ((accs[0][h][c4] + accs[1][h][c4]) + accs[2][h][c4]) + accs[3][h][c4];
        const f32x4 gated = sum * sigmoidf(gsh[h]);
        if (p.u) reinterpret_cast<f32x4*>(p.u + ((int64_t)m * PH + h) * D)[c4] = gated;
        if (p.u_b) store_bf16x4(p.u_b + ((int64_t)m * PH + h) * D + 4 * c4, gated);
    }
}

int pool_mix(const PoolMixArgs& p, hipStream_t stream) {
    D4_REQUIRE(p.heads == 4, "pool_mix: 4 pool heads expected (AttentionPool default, D4:2147)");
    D4_REQUIRE(p.L >= 1 && p.L <= 64 && p.D % 4 == 0 && p.D <= 1024, "pool_mix: L=%d D=%d out of range", p.L, p.D);
    D4_REQUIRE(p.k_b ? (p.q != nullptr || p.q_b != nullptr) : (p.k != nullptr && p.q != nullptr && p.q_b == nullptr), "pool_mix: keys / queries: fp32 (k, q) or the bf16 images (k_b with q or q_b)");
    if (p.M == 0) return 0;
    dim3 grid(cdiv(p.M, 4)), block(256);
    // one block per row (its four waves split the hiddens) while that leaves the CUs short of waves — by M alone: measured at B = 256, L = 13:
    // M = 1280 (the final stage's compacted rows) 24.1 -> 14.3 us, M = 3584 25.3 -> 27.8 us (the wave-per-row form wins once it fills the chip)
    constexpr int rows_max = 2048;
    const bool kb = p.k_b != nullptr;                // bf16 engine: keys from their bf16 image (template flag of both kernels)
    if (p.M <= rows_max && p.D <= 512) {
        const double rb = 4.0 * p.M * ((double)p.L * (p.D + p.ldk) + p.ldq + p.D + (double)p.heads * p.D);
        if (p.D <= 256) { if (kb) hipLaunchKernelGGL((pool_mix_rows_kernel<1, true>), dim3(p.M), block, 0, stream, p); else hipLaunchKernelGGL(pool_mix_rows_kernel<1>, dim3(p.M), block, 0, stream, p); }
        else if (kb) hipLaunchKernelGGL((pool_mix_rows_kernel<2, true>), dim3(p.M), block, 0, stream, p);
        else D4_GLUE_LAUNCH(GL_POOL_MIX, rb, pool_mix_rows_kernel<2>, dim3(p.M), block, 0, stream, p);
        D4_LAUNCH_CHECK();
        return 0;
    }
    // algorithmic bytes: L hiddens + L projected keys per token row, queries + the row itself, the per-head mixes written
    const double pm_bytes = 4.0 * p.M * ((double)p.L * (p.D + p.ldk) + p.ldq + p.D + (double)p.heads * p.D);
    if (p.D <= 256) { if (kb) hipLaunchKernelGGL((pool_mix_kernel<1, false, true>), grid, block, 0, stream, p); else hipLaunchKernelGGL(pool_mix_kernel<1>, grid, block, 0, stream, p); }
    else if (p.D <= 512) { if (kb) hipLaunchKernelGGL((pool_mix_kernel<2, false, true>), grid, block, 0, stream, p); else D4_GLUE_LAUNCH(GL_POOL_MIX, pm_bytes, pool_mix_kernel<2>, grid, block, 0, stream, p); }
    // D > 512 (BASELINE config 5: dim 1024, 1792 rows x up to 25 hiddens: 32 us per launch = ~3.6 TB/s of hiddens + keys).  Measured in round 4 and NOT
    // adopted, all level with this form: the deep-prefetch variant, reading the bf16 hidden images (kept: half the bytes), and a block-per-row kernel
    // whose four waves split the features with every load issued up front — the launch is bound by what the memory side delivers for rows that
    // were written many kernels ago, not by the wave structure
    else if (kb) hipLaunchKernelGGL((pool_mix_kernel<4, false, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL(pool_mix_kernel<4>, grid, block, 0, stream, p);
    D4_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// time axis

__device__ __forceinline__ float rotate_half_lane(float x, int lane, float pos, const float* inv_freq, int dh) {
    // freqs = cat(f, f); rotated = x * cos + cat(-x2, x1) * sin      (D4:1624, 1653-1658); halves are dh / 2 lanes wide
    const int hw = dh >> 1;
    const float f = pos * inv_freq[lane & (hw - 1)];
    const float partner = __shfl_xor(x, hw);
    const float half = (lane < hw) ? -partner : partner;
    float sn, cs;
    sincosf(f, &sn, &cs);
    return x * cs + half * sn;
}

template <int DH>
__global__ __launch_bounds__(256) void time_kv_append_kernel(TimeAttnArgs p) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rows = p.B * p.Tq * p.S;
    if (wid >= rows * p.H) return;
    const int row = wid / p.H, h = wid % p.H;
    const int lane = threadIdx.x & 63;
    const int s = row % p.S, tq = (row / p.S) % p.Tq, b = row / (p.S * p.Tq);
    constexpr int dh = DH;
    const int hd = p.H * dh;
    const bool act = DH == 64 || lane < dh;
    const int hl = h * dh + (act ? lane : 0);
    const float* pr = p.proj + (int64_t)row * p.ldp;
    float k = act ? pr[hd + hl] : 0.f;
    float v = act ? pr[2 * hd + hl] : 0.f;
    const float vr = act ? p.vres[(int64_t)row * p.ldv + hl] : 0.f;
    const float w = sigmoidf(pr[3 * hd + p.H + h]);
    v = lerp_torch(v, vr, w);
    const float nrm = sqrtf(wave_sum(k * k));
    k = k / fmaxf(nrm, 1e-12f) * (act ? (p.k_gamma[hl] + 1.f) * sqrtf((float)dh) : 0.f);
    const int pos = (p.t0_dev ? *p.t0_dev : p.t0) + tq;
    k = rotate_half_lane(k, lane, (float)pos, p.inv_freq, dh);
    const int cS = p.cache_S > 0 ? p.cache_S : p.S;
    const int64_t col = (int64_t)b * cS + s;
    const int64_t cols = (int64_t)p.cache_batch * cS;
    if (!act) return;
    const int64_t off = ((col * p.H + h) * p.Tcap + pos) * dh + lane;
    p.cache[off] = k;
    p.cache[cols * p.H * p.Tcap * dh + off] = v;
}

// Head dim 64, four heads per wave: lane = (head of the group, feature quarter-row), a float4 of k / v / value residual per lane, so a wave
// instruction moves 1 KB instead of 256 B and a quarter of the waves carry the same bytes (the one-head-per-wave form above is a chain of
// 256-byte round trips: 0.36 of the HBM rate).  The key norm is a 16-lane DPP row reduction, the rotary partner (feature ^ 32) sits 8 lanes away.
__global__ __launch_bounds__(256) void time_kv_append4_kernel(TimeAttnArgs p) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rows = p.B * p.Tq * p.S, HG = p.H >> 2;
    if (wid >= rows * HG) return;
    const int row = wid / HG, hg = wid % HG;
    const int lane = threadIdx.x & 63, fg = lane & 15;
    const int h = hg * 4 + (lane >> 4);
    const int s = row % p.S, tq = (row / p.S) % p.Tq, b = row / (p.S * p.Tq);
    const int hd = p.H * 64;
    const int hl = h * 64 + 4 * fg;
    const float* pr = p.proj + (int64_t)row * p.ldp;
    f32x4 k = *reinterpret_cast<const f32x4*>(pr + hd + hl);
    f32x4 v = *reinterpret_cast<const f32x4*>(pr + 2 * hd + hl);
    const f32x4 vr = *reinterpret_cast<const f32x4*>(p.vres + (int64_t)row * p.ldv + hl);
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.k_gamma + hl);
    const float w = sigmoidf(pr[3 * hd + p.H + h]);
    const int pos = (p.t0_dev ? *p.t0_dev : p.t0) + tq;
    const f32x4 fr = *reinterpret_cast<const f32x4*>(p.inv_freq + 4 * (fg & 7));
    const float nrm = sqrtf(row_sum16(((k[0] * k[0] + k[1] * k[1]) + k[2] * k[2]) + k[3] * k[3]));
    const float den = fmaxf(nrm, 1e-12f);
    f32x4 ko;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = lerp_torch(v[e], vr[e], w);
        const float kn = k[e] / den * ((g4[e] + 1.f) * 8.f);
        const float partner = __shfl_xor(kn, 8);                   // feature ^ 32: the other half of the head
        const float half = fg < 8 ? -partner : partner;
        float sn, cs;
        sincosf((float)pos * fr[e], &sn, &cs);
        ko[e] = kn * cs + half * sn;
    }
    const int cS = p.cache_S > 0 ? p.cache_S : p.S;
    const int64_t col = (int64_t)b * cS + s;
    const int64_t cols = (int64_t)p.cache_batch * cS;
    const int64_t off = ((col * p.H + h) * p.Tcap + pos) * 64 + 4 * fg;
    *reinterpret_cast<f32x4*>(p.cache + off) = ko;
    *reinterpret_cast<f32x4*>(p.cache + cols * p.H * p.Tcap * 64 + off) = v;
}

// The new K / V row of (token row, head h) at frame `pos`, features 4 fg .. 4 fg + 3, on a 16-lane group (fg = lane & 15): the arithmetic of
// time_kv_append4_kernel — value-residual lerp, K l2-norm * (gamma + 1) * sqrt(64), rotary — shared with the cached-decode kernels that append
// their own row before they attend (one frame per pass: a (column, head) needs no other row of this pass).
__device__ __forceinline__ void time_new_kv(const TimeAttnArgs& p, const float* pr, int row, int h, int fg, int pos, f32x4& ko, f32x4& vo) {
    const int hd = p.H * 64, hl = h * 64 + 4 * fg;
    const f32x4 k = *reinterpret_cast<const f32x4*>(pr + hd + hl);
    f32x4 v = *reinterpret_cast<const f32x4*>(pr + 2 * hd + hl);
    const f32x4 vr = *reinterpret_cast<const f32x4*>(p.vres + (int64_t)row * p.ldv + hl);
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.k_gamma + hl);
    const float w = sigmoidf(pr[3 * hd + p.H + h]);
    const f32x4 fr = *reinterpret_cast<const f32x4*>(p.inv_freq + 4 * (fg & 7));
    const float nrm = sqrtf(row_sum16(((k[0] * k[0] + k[1] * k[1]) + k[2] * k[2]) + k[3] * k[3]));
    const float den = fmaxf(nrm, 1e-12f);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        vo[e] = lerp_torch(v[e], vr[e], w);
        const float kn = k[e] / den * ((g4[e] + 1.f) * 8.f);
        const float partner = __shfl_xor(kn, 8);                   // feature ^ 32: the other half of the head
        const float half = fg < 8 ? -partner : partner;
        float sn, cs;
        sincosf((float)pos * fr[e], &sn, &cs);
        ko[e] = kn * cs + half * sn;
    }
}

template <int DH>
__global__ __launch_bounds__(256) void time_attn_kernel(TimeAttnArgs p) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rows = p.B * p.Tq * p.S;
    if (wid >= rows * p.H) return;
    const int row = wid / p.H, h = wid % p.H;
    const int lane = threadIdx.x & 63;
    const int s = row % p.S, tq = (row / p.S) % p.Tq, b = row / (p.S * p.Tq);
    constexpr int dh = DH;
    const int hd = p.H * dh;
    const bool act = DH == 64 || lane < dh;
    const int hl = h * dh + (act ? lane : 0);
    const float* pr = p.proj + (int64_t)row * p.ldp;
    const int pos = (p.t0_dev ? *p.t0_dev : p.t0) + tq;
    float q = rotate_half_lane(act ? pr[hl] : 0.f, lane, (float)pos, p.inv_freq, dh);
    const int cS = p.cache_S > 0 ? p.cache_S : p.S;
    const int64_t col = (int64_t)b * cS + s;
    const int64_t cols = (int64_t)p.cache_batch * cS;
    const float* ck = p.cache + ((col * p.H + h) * p.Tcap) * dh + (act ? lane : 0);
    const float* cv = ck + cols * p.H * p.Tcap * dh;
    const float qscale = rsqrtf((float)dh);

    float m = -FLT_MAX, l = 0.f, acc = 0.f;
    for (int j = 0; j <= pos; ++j) {
        float sc = wave_sum(act ? q * ck[j * dh] : 0.f) * qscale;
        if (p.softclamp > 0.f) sc = tanhf(sc / p.softclamp) * p.softclamp;
        const float mn = fmaxf(m, sc);
        const float alpha = expf(m - mn);
        const float e = expf(sc - mn);
        l = l * alpha + e;
        acc = acc * alpha + e * (act ? cv[j * dh] : 0.f);
        m = mn;
    }
    float o = acc / l;
    // belief: orthogonalise against this step's (mixed) value            D4:2049-2054
    const float vi = act ? cv[pos * dh] : 0.f;
    const float vn = vi / fmaxf(sqrtf(wave_sum(vi * vi)), 1e-12f);
    o -= wave_sum(o * vn) * vn;
    o *= sigmoidf(pr[3 * hd + h]);
    if (act) p.out[(int64_t)row * p.ldo + hl] = o;
}

// Head dim 64, lanes spread over (key, feature group): lane = (key j & 3) * 16 + feature group, a float4 of K / V per lane, so ONE
// pass of the wave scores four keys at once (q . k is a 16-lane DPP row reduction) instead of one key per full-wave reduction, and
// tanh / exp are evaluated for four keys per pass.  Keys are processed in chunks of 64 (16 passes) with an online softmax across
// chunks, so the dependent chain is ceil((pos + 1) / 64) chunk steps + 16 independent passes each, not pos + 1 serial reductions.
//   STAGE = true  (several queries of one (column, head): the parallel multi-frame evaluation): the chunk's K / V tiles are staged
//                 ONCE in LDS by the block's waves (coalesced float4), every wave (= query frame) then reads them from LDS;
//   STAGE = false (cached decode, one query per (column, head)): each K / V float4 is read exactly once — straight to registers.
constexpr int TA_CHUNK = 64;
template <bool STAGE, bool APPEND = false>
__global__ __launch_bounds__(256) void time_attn64_kernel(TimeAttnArgs p) {
    static_assert(!(STAGE && APPEND), "the appending form is the cached decode (one query per column and head)");
    extern __shared__ __attribute__((aligned(16))) float kv_s[];              // STAGE: [2][TA_CHUNK][64]
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int unit = STAGE ? blockIdx.x : blockIdx.x * nw + w;                // (b, s, h)
    const int units = p.B * p.S * p.H;
    if (!STAGE && unit >= units) return;
    const int h = unit % p.H, s = (unit / p.H) % p.S, b = unit / (p.H * p.S);
    const int t0 = p.t0_dev ? *p.t0_dev : p.t0;
    const int hd = p.H * 64;
    const int cS = p.cache_S > 0 ? p.cache_S : p.S;
    const int64_t col = (int64_t)b * cS + s, cols = (int64_t)p.cache_batch * cS;
    const float* ck = p.cache + ((col * p.H + h) * p.Tcap) * 64;
    const float* cv = ck + cols * p.H * p.Tcap * 64;
    const int fg = lane & 15, kr = lane >> 4;                                  // feature group (4 floats), key residue
    const float qscale = 0.125f;

    const int rounds = STAGE ? (p.Tq + nw - 1) / nw : 1;
    for (int rd = 0; rd < rounds; ++rd) {
        const int tq = STAGE ? rd * nw + w : 0;
        const bool active = tq < p.Tq;                                         // (STAGE: idle waves still help staging / hit the barriers)
        const int row = (b * p.Tq + (active ? tq : 0)) * p.S + s;
        const float* pr = p.proj + (int64_t)row * p.ldp;
        const int pos = t0 + (active ? tq : 0);
        // q: rotate-half rotary on features 4 fg .. 4 fg + 3 (partner features +-32 live 8 lanes away in the row)
        f32x4 q4 = *reinterpret_cast<const f32x4*>(pr + h * 64 + fg * 4);
        {
            f32x4 part;
#pragma unroll
            for (int e = 0; e < 4; ++e) part[e] = __shfl_xor(q4[e], 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = fg * 4 + e;
                float sn, cs;
                sincosf((float)pos * p.inv_freq[d & 31], &sn, &cs);
                q4[e] = q4[e] * cs + (d < 32 ? -part[e] : part[e]) * sn;
            }
        }
        // APPEND: this (column, head)'s new K / V row is computed here (every 16-lane key-residue group computes the same quarter-rows), written
        // to the cache by the first group, and used from registers as key `pos`
        f32x4 knew = {0.f, 0.f, 0.f, 0.f}, vnew = {0.f, 0.f, 0.f, 0.f};
        if constexpr (APPEND) {
            time_new_kv(p, pr, row, h, fg, pos, knew, vnew);
            if (kr == 0) {
                *reinterpret_cast<f32x4*>(const_cast<float*>(ck) + (int64_t)pos * 64 + fg * 4) = knew;
                *reinterpret_cast<f32x4*>(const_cast<float*>(cv) + (int64_t)pos * 64 + fg * 4) = vnew;
            }
        }
        // the belief projection's own value row and the head gate: requested here, consumed at the very end
        const f32x4 vi = APPEND ? vnew : *reinterpret_cast<const f32x4*>(cv + (int64_t)pos * 64 + fg * 4);
        const float gate_logit = pr[3 * hd + h];
        float m = -FLT_MAX, l = 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int nkeys = STAGE ? t0 + p.Tq : pos + 1;                          // keys any query of this block needs
        for (int c0 = 0; c0 < nkeys; c0 += TA_CHUNK) {
            const int cn = min(TA_CHUNK, nkeys - c0);
            if (STAGE) {
                __syncthreads();                                               // previous chunk fully consumed
                for (int i = threadIdx.x; i < cn * 16; i += blockDim.x) {
                    reinterpret_cast<f32x4*>(kv_s)[i] = reinterpret_cast<const f32x4*>(ck + (int64_t)c0 * 64)[i];
                    reinterpret_cast<f32x4*>(kv_s + TA_CHUNK * 64)[i] = reinterpret_cast<const f32x4*>(cv + (int64_t)c0 * 64)[i];
                }
                __syncthreads();
            }
            if (!active) continue;
            const f32x4* kt = STAGE ? reinterpret_cast<const f32x4*>(kv_s) : reinterpret_cast<const f32x4*>(ck + (int64_t)c0 * 64);
            const f32x4* vt = STAGE ? reinterpret_cast<const f32x4*>(kv_s + TA_CHUNK * 64) : reinterpret_cast<const f32x4*>(cv + (int64_t)c0 * 64);
            float sc[TA_CHUNK / 4];
            float cm = -FLT_MAX;
            const int last = min(cn - 1, pos - c0);                            // last key of this chunk the query may see (causal)
            const int passes = __builtin_amdgcn_readfirstlane(last < 0 ? 0 : last / 4 + 1);      // wave-uniform: skipped passes cost one scalar branch
            // Cached decode: the K and V rows of the chunk's first TA_PRE passes (16 keys = a whole 15-frame horizon) are requested in ONE
            // batch, lane-predicated, before anything depends on them — with a load inside each pass (behind a wave-uniform branch) a wave paid
            // one memory round trip per pass and per operand, 8 in a row at t = 15 (5.5 us per wave measured, 55 % of it waiting).
            constexpr int TA_PRE = STAGE ? 0 : 4;
            f32x4 kpre[TA_PRE > 0 ? TA_PRE : 1], vpre[TA_PRE > 0 ? TA_PRE : 1];
            if constexpr (TA_PRE > 0) {
#pragma unroll
                for (int ps = 0; ps < TA_PRE; ++ps) {
                    const int j = ps * 4 + kr;
                    const bool ok = j <= last;
                    kpre[ps] = ok ? kt[j * 16 + fg] : f32x4{0.f, 0.f, 0.f, 0.f};
                    vpre[ps] = ok ? vt[j * 16 + fg] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int ps = 0; ps < TA_CHUNK / 4; ++ps) {
                sc[ps] = -FLT_MAX;
                if (ps < passes) {
                    const int j = ps * 4 + kr;
                    const bool ok = j <= last;
                    f32x4 k4 = ps < TA_PRE ? kpre[ps < TA_PRE ? ps : 0] : (ok ? kt[j * 16 + fg] : f32x4{0.f, 0.f, 0.f, 0.f});
                    if (APPEND && c0 + j == pos) k4 = knew;
                    float d = row_sum16(q4[0] * k4[0] + q4[1] * k4[1] + q4[2] * k4[2] + q4[3] * k4[3]) * qscale;
                    if (p.softclamp > 0.f) d = tanhf(d / p.softclamp) * p.softclamp;
                    sc[ps] = ok ? d : -FLT_MAX;
                    cm = fmaxf(cm, sc[ps]);
                }
            }
            cm = fmaxf(cm, __shfl_xor(cm, 16));
            cm = fmaxf(cm, __shfl_xor(cm, 32));                                // chunk max over the four key residues
            const float mn = fmaxf(m, cm);
            const float alpha = expf(m - mn);
            l *= alpha;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] *= alpha;
#pragma unroll
            for (int ps = 0; ps < TA_CHUNK / 4; ++ps) {
                const int j = ps * 4 + kr;
                if (ps < passes && sc[ps] > -FLT_MAX) {
                    const float e_ = expf(sc[ps] - mn);
                    f32x4 v4 = ps < TA_PRE ? vpre[ps < TA_PRE ? ps : 0] : vt[j * 16 + fg];
                    if (APPEND && c0 + j == pos) v4 = vnew;
                    l += e_;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += e_ * v4[e];
                }
            }
            m = mn;
        }
        if (!active) continue;
        // fold the four key residues (lanes fg, fg + 16, fg + 32, fg + 48)
        l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] += __shfl_xor(acc[e], 16); acc[e] += __shfl_xor(acc[e], 32); }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[e] / l;
        // belief: orthogonalise against this step's (mixed) value            D4:2049-2054
        const float vn2 = row_sum16(vi[0] * vi[0] + vi[1] * vi[1] + vi[2] * vi[2] + vi[3] * vi[3]);
        const float inv = 1.f / fmaxf(sqrtf(vn2), 1e-12f);
        const float dot = row_sum16(o[0] * vi[0] + o[1] * vi[1] + o[2] * vi[2] + o[3] * vi[3]) * inv;
        const float gate = sigmoidf(gate_logit);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (o[e] - dot * (vi[e] * inv)) * gate;
        if (kr == 0) {
            *reinterpret_cast<f32x4*>(p.out + (int64_t)row * p.ldo + h * 64 + fg * 4) = o;
            if (p.out_b) store_bf16x4(p.out_b + (int64_t)row * p.ldo + h * 64 + fg * 4, o);
        }
    }
}

// Cached decode over a SHORT history (<= TA_FEW keys; one instantiation for up to 8 keys, one for 9 .. 16: a single 16-key form measured slower on the
// first frames — 4 / 8 / 16 keys for every frame: 39.0 / 36.8 / 39.7 us averaged over a 15-frame horizon), head dim 64: four heads per wave — lane = (head of the
// group, feature quarter-row) — and the few keys walked in sequence, all K / V rows requested before anything depends on them.  The four-keys-per-
// pass kernel above leaves three quarters of its lanes idle at t < 4 and runs one wave per head: 32 us per launch at t = 0 for 15 MB.
template <int TA_FEW, bool APPEND = false>
__global__ __launch_bounds__(256) void time_attn64_few_kernel(TimeAttnArgs p) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int HG = p.H >> 2;
    if (wid >= p.B * p.S * HG) return;
    const int hg = wid % HG, s = (wid / HG) % p.S, b = wid / (HG * p.S);
    const int lane = threadIdx.x & 63, fg = lane & 15;
    const int h = hg * 4 + (lane >> 4);
    const int pos = p.t0_dev ? *p.t0_dev : p.t0;                               // Tq == 1; pos < TA_FEW (the launcher picks by time_history_bucket(p.t0), which a replayed graph is keyed on)
    const int hd = p.H * 64;
    const int cS = p.cache_S > 0 ? p.cache_S : p.S;
    const int64_t col = (int64_t)b * cS + s, cols = (int64_t)p.cache_batch * cS;
    const f32x4* ck = reinterpret_cast<const f32x4*>(p.cache + ((col * p.H + h) * p.Tcap) * 64) + fg;
    const f32x4* cv = reinterpret_cast<const f32x4*>(p.cache + cols * p.H * p.Tcap * 64 + ((col * p.H + h) * p.Tcap) * 64) + fg;
    const int row = b * p.S + s;
    const float* pr = p.proj + (int64_t)row * p.ldp;
    f32x4 q4 = *reinterpret_cast<const f32x4*>(pr + h * 64 + fg * 4);
    const float gate_logit = pr[3 * hd + h];
    f32x4 k4[TA_FEW], v4[TA_FEW];
#pragma unroll
    for (int j = 0; j < TA_FEW; ++j) {
        const bool ok = APPEND ? j < pos : j <= pos;                           // (APPEND: row `pos` is this kernel's own, below)
        k4[j] = ok ? ck[j * 16] : f32x4{0.f, 0.f, 0.f, 0.f};
        v4[j] = ok ? cv[j * 16] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (APPEND) {
        // this (column, head)'s new K / V row: computed here (the lane layout is time_kv_append4_kernel's), written to the cache, used from registers
        f32x4 knew, vnew;
        time_new_kv(p, pr, row, h, fg, pos, knew, vnew);
        const int64_t off = (int64_t)pos * 64;
        *reinterpret_cast<f32x4*>(const_cast<float*>(reinterpret_cast<const float*>(ck)) + off) = knew;
        *reinterpret_cast<f32x4*>(const_cast<float*>(reinterpret_cast<const float*>(cv)) + off) = vnew;
#pragma unroll
        for (int j = 0; j < TA_FEW; ++j) if (j == pos) { k4[j] = knew; v4[j] = vnew; }
    }
    {
        const f32x4 fr = *reinterpret_cast<const f32x4*>(p.inv_freq + 4 * (fg & 7));
        f32x4 part;
#pragma unroll
        for (int e = 0; e < 4; ++e) part[e] = __shfl_xor(q4[e], 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sn, cs;
            sincosf((float)pos * fr[e], &sn, &cs);
            q4[e] = q4[e] * cs + (fg < 8 ? -part[e] : part[e]) * sn;
        }
    }
    float sc[TA_FEW], m = -FLT_MAX;
#pragma unroll
    for (int j = 0; j < TA_FEW; ++j) {
        float d = row_sum16(q4[0] * k4[j][0] + q4[1] * k4[j][1] + q4[2] * k4[j][2] + q4[3] * k4[j][3]) * 0.125f;
        if (p.softclamp > 0.f) d = tanhf(d / p.softclamp) * p.softclamp;
        sc[j] = j <= pos ? d : -FLT_MAX;
        m = fmaxf(m, sc[j]);
    }
    float l = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TA_FEW; ++j) {
        if (j <= pos) {
            const float e_ = expf(sc[j] - m);
            l += e_;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += e_ * v4[j][e];
        }
    }
    f32x4 vi = v4[0];                                                          // this step's (mixed) value row = key `pos`
#pragma unroll
    for (int j = 1; j < TA_FEW; ++j) if (j == pos) vi = v4[j];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = acc[e] / l;
    // belief: orthogonalise against this step's (mixed) value            D4:2049-2054
    const float vn2 = row_sum16(vi[0] * vi[0] + vi[1] * vi[1] + vi[2] * vi[2] + vi[3] * vi[3]);
    const float inv = 1.f / fmaxf(sqrtf(vn2), 1e-12f);
    const float dot = row_sum16(o[0] * vi[0] + o[1] * vi[1] + o[2] * vi[2] + o[3] * vi[3]) * inv;
    const float gate = sigmoidf(gate_logit);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (o[e] - dot * (vi[e] * inv)) * gate;
    *reinterpret_cast<f32x4*>(p.out + (int64_t)row * p.ldo + h * 64 + fg * 4) = o;
    if (p.out_b) store_bf16x4(p.out_b + (int64_t)row * p.ldo + h * 64 + fg * 4, o);
}

int time_kv_append(const TimeAttnArgs& p, hipStream_t stream) {
    D4_REQUIRE(p.t0 + p.Tq <= p.Tcap, "time attention: cache capacity %d exceeded (t0=%d, Tq=%d)", p.Tcap, p.t0, p.Tq);
    const int waves = p.B * p.Tq * p.S * p.H;
    if (waves == 0) return 0;
    // algorithmic bytes: k, v, value residual read from the projection rows; K and V written into the cache
    const double ka_bytes = 4.0 * waves * p.dh * 5.0;
    const bool al4 = (p.ldp % 4) == 0 && (p.ldv % 4) == 0 && ((uintptr_t)p.proj % 16) == 0 && ((uintptr_t)p.vres % 16) == 0 && ((uintptr_t)p.cache % 16) == 0 &&
                     ((uintptr_t)p.k_gamma % 16) == 0 && ((uintptr_t)p.inv_freq % 16) == 0 && (((int64_t)p.cache_batch * (p.cache_S > 0 ? p.cache_S : p.S) * p.H * p.Tcap) % 4) == 0;
    if (p.dh == 64 && (p.H % 4) == 0 && al4)               // four heads per wave; else the one-head-per-wave form
        D4_GLUE_LAUNCH(GL_TIME_KV_APPEND, ka_bytes, time_kv_append4_kernel, dim3(cdiv(waves / 4, 4)), dim3(256), 0, stream, p);
    else if (p.dh == 64) D4_GLUE_LAUNCH(GL_TIME_KV_APPEND, ka_bytes, time_kv_append_kernel<64>, dim3(cdiv(waves, 4)), dim3(256), 0, stream, p);
    else if (p.dh == 32) hipLaunchKernelGGL(time_kv_append_kernel<32>, dim3(cdiv(waves, 4)), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(time_kv_append_kernel<16>, dim3(cdiv(waves, 4)), dim3(256), 0, stream, p);
    D4_LAUNCH_CHECK();
    return 0;
}

// Cached decode of ONE frame (Tq == 1): append + attend in one launch — every (column, head) computes its own new K / V row, stores it and
// attends over its history with the new row taken from registers.  Same arithmetic and kernel choice (history bucket) as the two launches;
// anything the fused forms do not cover (several frames per pass, head dims 16 / 32, unaligned rows) runs as the two launches.
int g_time_attn_fused_append = 1;
int time_attn_append(const TimeAttnArgs& p, hipStream_t stream) {
    const bool fused_on = g_time_attn_fused_append != 0;           // (test hook d4_debug_switch("time_attn_fused_append"): the bitwise A/B of the two forms)
    auto al = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
    const bool al4 = (p.ldp % 4) == 0 && (p.ldv % 4) == 0 && (p.ldo % 4) == 0 && al(p.proj) && al(p.vres) && al(p.cache) && al(p.out) && al(p.k_gamma) && al(p.inv_freq) &&
                     (((int64_t)p.cache_batch * (p.cache_S > 0 ? p.cache_S : p.S) * p.H * p.Tcap) % 4) == 0;
    if (!(fused_on && p.Tq == 1 && p.dh == 64 && al4)) {
        if (int rc = time_kv_append(p, stream)) return rc;
        return time_attn(p, stream);
    }
    D4_REQUIRE(p.t0 + 1 <= p.Tcap, "time attention: cache capacity %d exceeded (t0=%d)", p.Tcap, p.t0);
    const int units = p.B * p.S * p.H;
    if (units == 0) return 0;
    // algorithmic bytes: the append's (k, v, value residual read; K, V written) + the attention's (history K / V, q read, out written)
    const double bytes = 4.0 * units * 64.0 * 5.0 + 4.0 * units * 64.0 * (2.0 * (p.t0 + 1) + 2.0);
    const int bucket = time_history_bucket(p.t0);
    if (bucket == 0 && (p.H % 4) == 0)
        D4_GLUE_LAUNCH(GL_TIME_ATTN, bytes, (time_attn64_few_kernel<8, true>), dim3(cdiv(units / 4, 4)), dim3(256), 0, stream, p);
    else if (bucket == 1 && (p.H % 4) == 0)
        D4_GLUE_LAUNCH(GL_TIME_ATTN, bytes, (time_attn64_few_kernel<16, true>), dim3(cdiv(units / 4, 4)), dim3(256), 0, stream, p);
    else D4_GLUE_LAUNCH(GL_TIME_ATTN, bytes, (time_attn64_kernel<false, true>), dim3(cdiv(units, 4)), dim3(256), 0, stream, p);
    D4_LAUNCH_CHECK();
    return 0;
}

int time_attn(const TimeAttnArgs& p, hipStream_t stream) {
    const int waves = p.B * p.Tq * p.S * p.H;
    if (waves == 0) return 0;
    if (p.dh == 64 && (p.ldp % 4) == 0 && (p.ldo % 4) == 0) {         // (else the one-key-per-reduction form: head dims 16 / 32, unaligned rows)
        const int units = p.B * p.S * p.H;
        // algorithmic bytes (cached decode): the K and V of frames 0..t0 of every (column, head) read once + q read + out written
        const double ta_bytes = 4.0 * units * 64.0 * (2.0 * (p.t0 + 1) + 2.0);
        const bool al4 = ((uintptr_t)p.proj % 16) == 0 && ((uintptr_t)p.cache % 16) == 0 && ((uintptr_t)p.out % 16) == 0 && ((uintptr_t)p.inv_freq % 16) == 0 &&
                         (((int64_t)p.cache_batch * (p.cache_S > 0 ? p.cache_S : p.S) * p.H * p.Tcap) % 4) == 0;
        const int bucket = time_history_bucket(p.t0);                           // the same rule eagerly and under graph replay (graphs are keyed on it)
        if (p.Tq == 1 && bucket == 0 && (p.H % 4) == 0 && al4)           // a short history: four heads per wave
            D4_GLUE_LAUNCH(GL_TIME_ATTN, ta_bytes, time_attn64_few_kernel<8>, dim3(cdiv(units / 4, 4)), dim3(256), 0, stream, p);
        else if (p.Tq == 1 && bucket == 1 && (p.H % 4) == 0 && al4)
            D4_GLUE_LAUNCH(GL_TIME_ATTN, ta_bytes, time_attn64_few_kernel<16>, dim3(cdiv(units / 4, 4)), dim3(256), 0, stream, p);
        else if (p.Tq == 1) D4_GLUE_LAUNCH(GL_TIME_ATTN, ta_bytes, time_attn64_kernel<false>, dim3(cdiv(units, 4)), dim3(256), 0, stream, p);
        else {
            // one block per (column, head): min(Tq, 4) waves = query frames, the chunk's K / V staged once in LDS
            const int nwv = p.Tq < 4 ? p.Tq : 4;
            hipLaunchKernelGGL(time_attn64_kernel<true>, dim3(units), dim3(64 * nwv), (size_t)2 * TA_CHUNK * 64 * sizeof(float), stream, p);
        }
    }
    else {
        if (p.dh == 64) hipLaunchKernelGGL(time_attn_kernel<64>, dim3(cdiv(waves, 4)), dim3(256), 0, stream, p);
        else if (p.dh == 32) hipLaunchKernelGGL(time_attn_kernel<32>, dim3(cdiv(waves, 4)), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(time_attn_kernel<16>, dim3(cdiv(waves, 4)), dim3(256), 0, stream, p);
        D4_LAUNCH_CHECK();
        if (p.out_b) return cvt_rows_bf16(p.out, p.ldo, p.out_b, p.ldo, p.B * p.Tq * p.S, p.H * p.dh, stream);       // (the one-key-per-reduction form writes fp32 only)
        return 0;
    }
    D4_LAUNCH_CHECK();
    return 0;
}

}  // namespace d4
