// Attention of one (group, head) on ONE wavefront, head dim 64, on the matrix pipe (see attn.hip for the derivation): shared by the
// stand-alone kernel (attn.hip: attn_mfma_kernel) and the per-frame fused kernels (frame_fused.hip).
#pragma once
#include "common.h"
#include "kernels.h"
#include <float.h>

namespace d4 {

__device__ __forceinline__ float lerp_torch(float a, float b, float w) {
    // at::native::lerp: two-branch form
    float d = b - a;
    return (fabsf(w) < 0.5f) ? a + w * d : b - d * (1.f - w);
}

constexpr int SM_LDV = 68;
// One (group g, head h) on the calling wave.  Vs [KT * 16][SM_LDV], kinv_s / vinv_s [KT * 16]: this wave's LDS scratch.
// store(orank, t, tok, value): output row `orank` (after the query-set restriction), feature 16 t + tok of head h.
template <int QT, int KT, class Store>
__device__ __forceinline__ void attn_mfma_unit(const SmallAttnArgs& p, int g, int h, int lane, float* Vs, float* kinv_s, float* vinv_s, Store store) {
    const int nq = p.nq, nk = p.nk;
    const int tok = lane & 15, kq = lane >> 4;
    const int hoff = h * 64 + 4 * kq;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    // ---- operand loads: token row 16 tile + tok, features 16 s + 4 kq .. + 3
    f32x4 q4[QT][4], k4[KT][4];
    float gate_logit[QT][4];                 // the head gates of this lane's output rows: requested with the operands
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int i = 16 * qt + tok;
        const float* qp = p.q + g * p.q_group_stride + (int64_t)i * p.q_item_stride + hoff;
#pragma unroll
        for (int s = 0; s < 4; ++s) q4[qt][s] = i < nq ? *reinterpret_cast<const f32x4*>(qp + 16 * s) : zero;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = 16 * qt + 4 * kq + r;
            gate_logit[qt][r] = (p.gate && qi < nq) ? p.gate[g * p.g_group_stride + (int64_t)qi * p.g_item_stride + h] : 0.f;
        }
    }
    f32x4 gm[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) gm[s] = *reinterpret_cast<const f32x4*>(p.k_gamma + h * 64 + 16 * s + 4 * kq);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int j = 16 * kt + tok;
        const bool ok = j < nk;
        const float* kp = p.k + g * p.k_group_stride + (int64_t)j * p.k_item_stride + hoff;
        const float* vp = p.v + g * p.v_group_stride + (int64_t)j * p.v_item_stride + hoff;
        const float* rp = p.vres ? p.vres + g * p.r_group_stride + (int64_t)j * p.r_item_stride + hoff : nullptr;
        f32x4 v4[4], r4[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            k4[kt][s] = ok ? *reinterpret_cast<const f32x4*>(kp + 16 * s) : zero;
            v4[s] = ok ? *reinterpret_cast<const f32x4*>(vp + 16 * s) : zero;
            r4[s] = (ok && rp) ? *reinterpret_cast<const f32x4*>(rp + 16 * s) : zero;
        }
        float wmix = 0.f;
        if (p.vres && ok) wmix = sigmoidf(p.mix[g * p.m_group_stride + (int64_t)j * p.m_item_stride + h]);
        float ksq = 0.f, vsq = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ksq = __builtin_fmaf(k4[kt][s][e], k4[kt][s][e], ksq);
                if (p.vres) v4[s][e] = lerp_torch(v4[s][e], r4[s][e], wmix);
                vsq = __builtin_fmaf(v4[s][e], v4[s][e], vsq);
            }
            *reinterpret_cast<f32x4*>(Vs + j * SM_LDV + 16 * s + 4 * kq) = v4[s];
        }
        ksq += __shfl_xor(ksq, 16); ksq += __shfl_xor(ksq, 32);
        vsq += __shfl_xor(vsq, 16); vsq += __shfl_xor(vsq, 32);
        if (kq == 0) {
            kinv_s[j] = 1.f / fmaxf(sqrtf(ksq), 1e-12f);
            vinv_s[j] = 1.f / fmaxf(sqrtf(vsq), 1e-12f);
        }
    }
    // (gamma + 1) onto q: score = sum_f q_f (gamma_f + 1) k_f / |k|   (the sqrt(dh) of the key scale cancels the 1 / sqrt(dh) of the query scale)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) q4[qt][s][e] *= gm[s][e] + 1.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                       // this wave's LDS writes (V', 1/|k|, 1/|v|) are read back by this wave only

#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        // ---- S^T = K Q'^T per key tile: A operand = K rows (key 16 kt + (l & 15)), B operand = Q' rows; acc[r] = S[i = 16 qt + (l & 15)][j = 16 kt + 4 kq + r]
        const int i = 16 * qt + tok;
        const bool ordinary_q = p.mask_special > 0 && i < nq - p.mask_special;
        f32x4 pr[KT];
        float m = -FLT_MAX;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f32x4 st = zero;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[kt][s][e], q4[qt][s][e], st, 0, 0, 0);
            const f32x4 kinv = *reinterpret_cast<const f32x4*>(kinv_s + 16 * kt + 4 * kq);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * kt + 4 * kq + r;
                float v = st[r] * kinv[r];
                if (p.softclamp > 0.f) v = tanhf(v / p.softclamp) * p.softclamp;
                const bool valid = j < nk && i < nq && !(ordinary_q && j >= nk - p.mask_special);
                pr[kt][r] = valid ? v : -FLT_MAX;
                m = fmaxf(m, pr[kt][r]);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { pr[kt][r] = pr[kt][r] > -FLT_MAX ? expf(pr[kt][r] - m) : 0.f; l += pr[kt][r]; }
        l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
        const float linv = l > 0.f ? 1.f / l : 0.f;

        // ---- out = P V': A operand = P (lane holds P[i][16 kt + 4 kq + e]), B operand = V'[16 kt + 4 kq + e][16 t + (l & 15)] from LDS
        f32x4 o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            o[t] = zero;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[kt][e] * linv, Vs[(16 * kt + 4 * kq + e) * SM_LDV + 16 * t + tok], o[t], 0, 0, 0);
        }
        // o[t][r] = out[i = 16 qt + 4 kq + r][d = 16 t + (l & 15)]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = 16 * qt + 4 * kq + r;
            if (qi >= nq) continue;                                             // (uniform over each 16-lane row group)
            float vn[4] = {0.f, 0.f, 0.f, 0.f}, dot = 0.f;
            if (p.belief) {                                                     // self attention only (nq == nk): the query's own mixed value row
                const float vinv = vinv_s[qi];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    vn[t] = Vs[qi * SM_LDV + 16 * t + tok] * vinv;
                    dot = __builtin_fmaf(o[t][r], vn[t], dot);
                }
                dot = row_sum16(dot);
            }
            int orank = qi;
            if (p.q_hi > 0) {
                if (qi >= p.q_lo && qi < p.q_hi) orank = qi - p.q_lo;
                else if (p.q_last && qi == nq - 1) orank = p.q_hi - p.q_lo;
                else continue;
            }
            const float gate = p.gate ? sigmoidf(gate_logit[qt][r]) : 1.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) store(orank, t, tok, (o[t][r] - dot * vn[t]) * gate);
        }
    }
}

}  // namespace d4
