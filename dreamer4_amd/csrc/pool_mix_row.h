// AttentionPool core for ONE token row on one wavefront (value side restructured: see PoolMixArgs in kernels.h): shared by the stand-alone
// kernel (attn.hip: pool_mix_kernel) and the per-frame fused pool kernel (frame_fused.hip).
#pragma once
#include "common.h"
#include "kernels.h"
#include <float.h>

namespace d4 {

// ps: LMAX * 4 floats of this wave's LDS scratch; gws: the pool's head-gate weights [4][ITER * 64] float4 staged in LDS (norm gamma folded,
// columns past D zero); store(h, c4, value): the gated mix of head h, features 4 c4 .. 4 c4 + 3 of row m.
// DEEP: key rows requested in batches of 8 and hidden rows 3 ahead (instead of one load per iteration / one row ahead) — for callers that run
// few waves per SIMD (the per-frame fused kernel: 2), where a row is otherwise 2 L dependent memory round trips; with 3.5 waves per SIMD the
// stand-alone kernel is bandwidth-bound and the deeper form measured 5 % slower there.
// KB16: the projected keys are read from their bf16 image p.k_b (bf16 engine) instead of p.k — a template flag, not a run-time branch: a branch
// inside the unrolled load batches keeps the compiler from issuing them together (measured: +26 % on the block-per-row kernel).
// the lane's four query features of row m: fp32, or (bf16 engine, queries projected by the same launch as the keys) their bf16 image
template <bool KB16>
__device__ __forceinline__ f32x4 pool_query4(const PoolMixArgs& p, int m, int lane) {
    if constexpr (KB16) {
        if (p.q_b) {
            const uint2 raw = *reinterpret_cast<const uint2*>(p.q_b + (int64_t)m * p.ldq + lane * 4);
            return f32x4{__builtin_bit_cast(float, raw.x << 16), __builtin_bit_cast(float, raw.x & 0xFFFF0000u),
                         __builtin_bit_cast(float, raw.y << 16), __builtin_bit_cast(float, raw.y & 0xFFFF0000u)};
        }
    }
    return *reinterpret_cast<const f32x4*>(p.q + (int64_t)m * p.ldq + lane * 4);
}

template <int ITER, bool DEEP = false, bool KB16 = false, class Store>
__device__ __forceinline__ void pool_mix_row(const PoolMixArgs& p, int m, int lane, float* ps, const f32x4* gws, Store store) {
    constexpr int PH = 4;
    const int L = p.L, D = p.D;
    const int nf4 = D / 4;

    // gate_h = sigmoid(RMSNorm(x) . gate_w[h]) scales the whole head output, so it is applied once after the mix; for the
    // in-loop pools x IS the last hidden row of the loop.  (gate weights: staged once per block in LDS, see above)
    const bool x_is_last_hidden = p.x == p.hid + (int64_t)(L - 1) * p.M * D && p.ldx == D;
    float glog[PH] = {0.f, 0.f, 0.f, 0.f};

    // scores: the 4 heads x 64 features of a key row are exactly one float4 per lane (head = lane / 16), so the
    // per-head reductions are 16-lane DPP row reductions and all four heads are scored at once
    const int hh = lane >> 4;
    const f32x4 q4 = pool_query4<KB16>(p, m, lane);
    f32x4 g4 = *reinterpret_cast<const f32x4*>(p.k_gamma + lane * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) g4[e] = (g4[e] + 1.f) * 8.f;
    float mxl = -FLT_MAX;
    constexpr int KB = DEEP ? 8 : 1;
    for (int l0 = 0; l0 < L; l0 += KB) {
        f32x4 kb[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j)
            if (l0 + j >= L) kb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            else if constexpr (KB16) {
                const uint2 raw = *reinterpret_cast<const uint2*>(p.k_b + ((int64_t)(l0 + j) * p.M + m) * p.ldk + lane * 4);
                kb[j] = f32x4{__builtin_bit_cast(float, raw.x << 16), __builtin_bit_cast(float, raw.x & 0xFFFF0000u),
                              __builtin_bit_cast(float, raw.y << 16), __builtin_bit_cast(float, raw.y & 0xFFFF0000u)};
            } else kb[j] = *reinterpret_cast<const f32x4*>(p.k + ((int64_t)(l0 + j) * p.M + m) * p.ldk + lane * 4);
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const int l = l0 + j;
            if (l >= L) break;
            const f32x4 kv = kb[j];
            const float nrm = sqrtf(row_sum16(kv[0] * kv[0] + kv[1] * kv[1] + kv[2] * kv[2] + kv[3] * kv[3]));
            const float inv = 1.f / fmaxf(nrm, 1e-12f);
            const float sc = row_sum16(q4[0] * (kv[0] * inv * g4[0]) + q4[1] * (kv[1] * inv * g4[1]) +
                                       q4[2] * (kv[2] * inv * g4[2]) + q4[3] * (kv[3] * inv * g4[3])) * 0.125f;
            mxl = fmaxf(mxl, sc);
            if ((lane & 15) == 0) ps[l * PH + hh] = sc;
        }
    }
    float mx[PH];
#pragma unroll
    for (int h = 0; h < PH; ++h) mx[h] = readlane_f(mxl, h * 16);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // Softmax weights once per (hidden, head), spread over the lanes — the L x 4 exponentials and divisions are the same for every lane of the
    // wave (they were the bulk of this function's instruction stream: 2 L x 4 exponentials + L x 4 divisions per LANE; the per-frame fused
    // kernel's mix phase, 60 % of its time, is bound by VALU issue — removing every key load or deepening the prefetch changed nothing).
    // Same values in the same order as before: e = exp(s - max), den = sum over l in order, w = e / den.
    for (int idx = lane; idx < L * PH; idx += 64) ps[idx] = expf(ps[idx] - mx[idx & (PH - 1)]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float den[PH];
#pragma unroll
    for (int h = 0; h < PH; ++h) {
        float d = 0.f;
        for (int l = 0; l < L; ++l) d += ps[l * PH + h];
        den[h] = d;
    }
    __builtin_amdgcn_wave_barrier();
    for (int idx = lane; idx < L * PH; idx += 64) ps[idx] = ps[idx] / den[idx & (PH - 1)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x4 acc[PH][ITER];
#pragma unroll
    for (int h = 0; h < PH; ++h)
#pragma unroll
        for (int i = 0; i < ITER; ++i) acc[h][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // hidden rows are software-pipelined RD ahead: the loads of the next RD rows are in flight while this row is reduced / mixed
    constexpr int RD = DEEP ? 3 : 1;
    f32x4 vn[RD][ITER];
    auto load_row = [&](int l, f32x4 (&dst)[ITER]) {
        if (p.hid_b) {                                   // bf16 image of the hiddens: 8 bytes per lane and group
            const uint2* hb = reinterpret_cast<const uint2*>(p.hid_b + ((int64_t)l * p.M + m) * D);
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                const int c4 = lane + 64 * i;
                const uint2 raw = c4 < nf4 ? hb[c4] : uint2{0u, 0u};
                dst[i] = f32x4{__builtin_bit_cast(float, raw.x << 16), __builtin_bit_cast(float, raw.x & 0xFFFF0000u),
                               __builtin_bit_cast(float, raw.y << 16), __builtin_bit_cast(float, raw.y & 0xFFFF0000u)};
            }
            return;
        }
        const f32x4* hr = reinterpret_cast<const f32x4*>(p.hid + ((int64_t)l * p.M + m) * D);
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int c4 = lane + 64 * i;
            dst[i] = c4 < nf4 ? hr[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
#pragma unroll
    for (int j = 0; j < RD; ++j)
        if (j < L) load_row(j, vn[j]);
    for (int l0 = 0; l0 < L; l0 += RD) {
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            const int l = l0 + j;
            if (l >= L) break;
            f32x4 v[ITER];
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                v[i] = vn[j][i];
                ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
            }
            if (l + RD < L) load_row(l + RD, vn[j]);
            const float rstd = rsqrtf(wave_sum(ss) / (float)D + p.eps);
#pragma unroll
            for (int h = 0; h < PH; ++h) {
                const float w = ps[l * PH + h] * rstd;
#pragma unroll
                for (int i = 0; i < ITER; ++i) acc[h][i] += v[i] * w;
            }
            if (l == L - 1 && x_is_last_hidden) {
#pragma unroll
                for (int h = 0; h < PH; ++h) {
                    float d = 0.f;
#pragma unroll
                    for (int i = 0; i < ITER; ++i) { const f32x4 g = gws[h * (ITER * 64) + lane + 64 * i]; d += v[i][0] * g[0] + v[i][1] * g[1] + v[i][2] * g[2] + v[i][3] * g[3]; }
                    glog[h] = wave_sum(d) * rstd;
                }
            }
        }
    }
    if (!x_is_last_hidden) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(p.x + (int64_t)m * p.ldx);
        f32x4 xv[ITER];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int c4 = lane + 64 * i;
            xv[i] = c4 < nf4 ? xr[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
            ss += xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1] + xv[i][2] * xv[i][2] + xv[i][3] * xv[i][3];
        }
        const float rstd = rsqrtf(wave_sum(ss) / (float)D + p.eps);
#pragma unroll
        for (int h = 0; h < PH; ++h) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < ITER; ++i) { const f32x4 g = gws[h * (ITER * 64) + lane + 64 * i]; d += xv[i][0] * g[0] + xv[i][1] * g[1] + xv[i][2] * g[2] + xv[i][3] * g[3]; }
            glog[h] = wave_sum(d) * rstd;
        }
    }
#pragma unroll
    for (int h = 0; h < PH; ++h) {
        const float gate = sigmoidf(glog[h]);
#pragma unroll
        for (int i = 0; i < ITER; ++i) acc[h][i] = acc[h][i] * gate;
    }
#pragma unroll
    for (int h = 0; h < PH; ++h)
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nf4) store(h, c4, acc[h][i]);
        }
}

}  // namespace d4
