// Frame-resident fp32 GEMM on the bf16 matrix cores (split operands, six products — the arithmetic of gemm_x3.hip) for the per-frame
// trunk kernel: the <= 16 token rows of one frame stay in LDS as three bf16 planes, every weight matrix is streamed global -> VGPR
// exactly once per workgroup, already split into its three bf16 planes at prepare time.
//
//   out[m][n] = sum_k A[m][k] * W[n][k]     A = A1 + A2 + A3, W = W1 + W2 + W3 (bf16 each, exact), six products of weight >= 2^-16:
//   hi += A1.W1 ;  lo += A3.W1 + A2.W2 + A1.W3 + A2.W1 + A1.W2 ;  out = hi + lo         (fp32 accumulation, fp32 accuracy)
//
// Why: with 16 rows per CU a weight element is used for 32 flops, so the CU-side feed (64 B/clk/CU of L1 bandwidth) — not the matrix
// pipe — is the bound; the f32-input MFMA would cap the same stream at 8 weights/clk/CU (256 flop/clk/CU), the bf16 pipe takes the six
// products of a weight in 6 x 16 / 512 clk and leaves the bound with the loads: 64 B/clk / 6 B per weight = 10.7 weights/clk/CU.
//
// v_mfma_f32_16x16x32_bf16, operands swapped (weight tile on the row side): lane l = (i = l & 15, kq = l >> 4) supplies
// W[n0 + i][32 kb + 8 kq .. + 7] and A[i][32 kb + 8 kq .. + 7] (16 bytes each per plane) in k-block kb and ends up holding
// out[m = l & 15][n0 + 4 (l >> 4) + r], r = 0..3.
// Weight image (fg3_tile_weights / the engine's prepare): Wt[n / 16][k / 32][plane][lane = 16 kq + i][8 bf16] — the three planes of one
// (tile, k-block) are 3 KB contiguous and every wave-load is one contiguous KB.
// A image in LDS: three planes [16][lda] bf16, lda = K + 8 (16-byte aligned rows, fragment reads spread over the banks).
#pragma once
#include "common.h"

namespace d4 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int FG3_R = 4;          // k-blocks (of 32) the weight loads run ahead: 2 tiles x 3 planes x 4 x 1 KB = 24 KB in flight per wave

struct Fg3Ring {
    f32x4 w[FG3_R][2][3];         // [k-block % R][tile of the unit][plane], 16 raw bytes = 8 bf16
};

struct Fg3Unit {                  // this lane's addresses (in floats = 4 bytes) of the unit's two weight tiles at k-block 0
    const float* wa;
    const float* wb;
};
// Wt as float*: one (tile, k-block) is 3 planes x 64 lanes x 4 floats = 768 floats; a tile of K columns is (K / 32) * 768 floats
__device__ __forceinline__ Fg3Unit fg3_make_unit(const float* Wt, int K, int tile_a, int tile_b, int lane) {
    const size_t tile = (size_t)(K / 32) * 768;
    return Fg3Unit{Wt + tile_a * tile + 4 * lane, Wt + tile_b * tile + 4 * lane};
}

__device__ __forceinline__ f32x4 fg3_load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__device__ __forceinline__ void fg3_prefetch(Fg3Ring& r, const Fg3Unit& u) {
#pragma unroll
    for (int j = 0; j < FG3_R; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            r.w[j][0][pl] = fg3_load(u.wa + j * 768 + pl * 256);
            r.w[j][1][pl] = fg3_load(u.wb + j * 768 + pl * 256);
        }
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void fg3_split(float a, __bf16& h1, __bf16& h2, __bf16& h3) {
    h1 = (__bf16)a;
    const float r = a - (float)h1;
    h2 = (__bf16)r;
    h3 = (__bf16)(r - (float)h2);
}

// One unit: K / 32 k-blocks.  a_lds: this lane's fp32 address = A + (lane & 15) * lda + 8 * (lane >> 4): the activations stay fp32 in
// LDS and are split into their three bf16 planes on the way into the MFMAs (44 VALU ops per k-block, beside the MFMAs of the previous
// k-block; the stream is load-bound, the vector unit has the slack).  Software pipeline per k-block b: raw fragment read in step b - 2,
// split in step b - 1, multiplied in step b.  Results: out = hi + lo per tile.
struct Fg3Frag { bf16x8 p[3]; };
__device__ __forceinline__ Fg3Frag fg3_split_frag(const f32x4& v0, const f32x4& v1) {
    Fg3Frag f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        __bf16 h1, h2, h3;
        fg3_split(v0[e], h1, h2, h3); f.p[0][e] = h1; f.p[1][e] = h2; f.p[2][e] = h3;
        fg3_split(v1[e], h1, h2, h3); f.p[0][e + 4] = h1; f.p[1][e + 4] = h2; f.p[2][e + 4] = h3;
    }
    return f;
}

template <int K>
__device__ __forceinline__ void fg3_unit(Fg3Ring& r, const Fg3Unit& cur, const Fg3Unit& next, const float* a_lds, f32x4& out0, f32x4& out1) {
    static_assert(K % (32 * FG3_R) == 0 && FG3_R % 2 == 0, "K must be a multiple of 128");
    constexpr int TRIPS = K / 32 / FG3_R, NKB = K / 32;
    f32x4 hi0 = {0.f, 0.f, 0.f, 0.f}, lo0 = hi0, hi1 = hi0, lo1 = hi0;
    f32x4 xr[2][2];
    Fg3Frag xp[2];
    xr[0][0] = *reinterpret_cast<const f32x4*>(a_lds); xr[0][1] = *reinterpret_cast<const f32x4*>(a_lds + 4);
    xr[1][0] = *reinterpret_cast<const f32x4*>(a_lds + 32); xr[1][1] = *reinterpret_cast<const f32x4*>(a_lds + 36);
    xp[0] = fg3_split_frag(xr[0][0], xr[0][1]);
#pragma unroll 1
    for (int t = 0; t < TRIPS; ++t) {
        const bool last = t == TRIPS - 1;
        const float* la = last ? next.wa : cur.wa + (t + 1) * FG3_R * 768;
        const float* lb = last ? next.wb : cur.wb + (t + 1) * FG3_R * 768;
#pragma unroll
        for (int j = 0; j < FG3_R; ++j) {
            // raw fragment of k-block b + 2 (past the end of the unit: k-blocks 0, 1 again — harmless)
            const int b2 = t * FG3_R + j + 2;
            const float* an = a_lds + (b2 < NKB ? b2 : b2 - NKB) * 32;
            xr[j & 1][0] = *reinterpret_cast<const f32x4*>(an); xr[j & 1][1] = *reinterpret_cast<const f32x4*>(an + 4);
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 w0[3], w1[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                w0[pl] = __builtin_bit_cast(bf16x8, r.w[j][0][pl]);
                w1[pl] = __builtin_bit_cast(bf16x8, r.w[j][1][pl]);
            }
            const bf16x8* x = xp[j & 1].p;
            // small terms first into `lo` (a3.w1, a2.w2, a1.w3, a2.w1, a1.w2), the leading term into `hi`; the two tiles alternate
#define D4_FG3_TERM(PA, PW, ACC0, ACC1)                                                  \
    ACC0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[PW], x[PA], ACC0, 0, 0, 0);       \
    ACC1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[PW], x[PA], ACC1, 0, 0, 0);
            D4_FG3_TERM(2, 0, lo0, lo1)
            D4_FG3_TERM(0, 0, hi0, hi1)
            D4_FG3_TERM(1, 1, lo0, lo1)
            D4_FG3_TERM(0, 2, lo0, lo1)
            D4_FG3_TERM(1, 0, lo0, lo1)
            D4_FG3_TERM(0, 1, lo0, lo1)
#undef D4_FG3_TERM
            // beside them: the split of k-block b + 1's fragment (read one step ago)
            xp[(j + 1) & 1] = fg3_split_frag(xr[(j + 1) & 1][0], xr[(j + 1) & 1][1]);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);     // 4 VALU
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                r.w[j][0][pl] = fg3_load(la + j * 768 + pl * 256);
                r.w[j][1][pl] = fg3_load(lb + j * 768 + pl * 256);
            }
            __builtin_amdgcn_sched_barrier(0);          // keep the loads where they are: R k-blocks ahead of their use
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { out0[e] = hi0[e] + lo0[e]; out1[e] = hi1[e] + lo1[e]; }
}

}  // namespace d4
