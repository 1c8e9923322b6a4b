// Frame-resident GEMM primitive for the per-frame trunk kernel (trunk_frame.hip): one workgroup owns the <= 16 token rows of ONE
// frame for a whole chain of layers; activations live in LDS, every weight matrix is streamed global -> VGPR exactly once per
// workgroup and goes straight into the matrix pipe.
//
//   out[m][n] = sum_k A[m][k] * W[n][k]        m < 16 (LDS, row stride lda floats), n < N (global, row stride ldw floats)
//
// Why this shape on MI355X: the imagination rollout at B = 256 trajectories has exactly one frame (14 / 15 token rows) per CU and no
// dependency between trajectories, so a CU can carry its frame through every layer with no kernel boundary, no grid barrier and no
// tail; what it needs from the memory system is the weight stream, 4 bytes per 32 flops = 32 B/clk/CU at the fp32 matrix rate, half of
// what a CU's L1 accepts (tools/micro/l2_feed_bench.hip: 25-30 TB/s of L2-resident data reach the CUs through either load path).
//
// Weight layout: the stream wants every wave-load to be ONE contiguous KB (a row-major tile would be 64 separate 16-byte pieces per
// instruction — measured 2x slower: the address unit, not the cache, limits it), so the matrices are re-tiled once at prepare time:
//   Wt[n / 16][k / 4][n % 16][k % 4]          (fg_tile_weights; a 16-row tile of K columns is K * 16 contiguous floats)
// and lane l = (i, kq) reads the 16 bytes at  tile + 256 s + 4 l  in k-step s.
//
// v_mfma_f32_16x16x4_f32 with the operands swapped (the weight tile on the MFMA's row side), as gemm2.hip: lane l = (i = l & 15,
// kq = l >> 4) supplies W[n0 + i][k] and A[i][k] for k = 16 s + 4 kq + e in MFMA e of k-step s — one 16-byte load per operand per
// k-step of 16 — and ends up holding out[m = l & 15][n0 + 4 (l >> 4) + r], r = 0..3: four consecutive columns of one row.
// A wave works on a UNIT of two column tiles at a time (two independent accumulators, the A fragment shared), its weight loads run
// R k-steps ahead of the MFMAs in a register ring that is kept full across unit boundaries (the loads of the next unit's first R
// k-steps are issued during the last R k-steps of the current one), so the stream never drains between tiles or between GEMMs
// of a chain (`prefetch` / `first` below).
#pragma once
#include "common.h"

namespace d4 {

constexpr int FG_R = 8;          // k-steps (of 16) the weight loads run ahead: 2 tiles x 8 x 1 KB = 16 KB in flight per wave

struct FgRing {                  // the weight ring of one wave: [k-step % R][tile of the unit]
    f32x4 w[FG_R][2];
};

// This lane's addresses of a unit's two weight tiles (16 rows of W each, k-step 0) in the tiled image.
struct FgUnit {
    const float* wa;             // Wt + tile_a * K * 16 + 4 * lane
    const float* wb;             // same for the second tile (== wa when the unit has one tile: the loads stay in bounds, the results are dropped)
};
__device__ __forceinline__ FgUnit fg_make_unit(const float* Wt, int K, int tile_a, int tile_b, int lane) {
    return FgUnit{Wt + (size_t)tile_a * K * 16 + 4 * lane, Wt + (size_t)tile_b * K * 16 + 4 * lane};
}

__device__ __forceinline__ f32x4 fg_load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Issue the loads of k-steps [0, R) of a unit (pipeline fill, or called by the previous GEMM's last unit as its look-ahead).
__device__ __forceinline__ void fg_prefetch(FgRing& r, const FgUnit& u) {
#pragma unroll
    for (int j = 0; j < FG_R; ++j) {
        r.w[j][0] = fg_load(u.wa + j * 256);
        r.w[j][1] = fg_load(u.wb + j * 256);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// One unit: K / 16 k-steps over the ring; while the last R k-steps are multiplied the first R of `next` are loaded.
//   a_lds: this lane's A pointer = A + (lane & 15) * lda + 4 * (lane >> 4)      (`next` may belong to another GEMM: any row stride)
template <int K, int EXPERIMENT = 0>      // EXPERIMENT (microbenchmarks only): 1 = no weight loads, 2 = no A fragment reads
__device__ __forceinline__ void fg_unit(FgRing& r, const FgUnit& cur, const FgUnit& next, const float* a_lds, f32x4& acc0, f32x4& acc1) {
    static_assert(K % (16 * FG_R) == 0, "K must be a multiple of 128");
    constexpr int TRIPS = K / 16 / FG_R;
    acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    // A fragments ping-pong between two register sets: the fragment of k-step j + 1 is requested from LDS before the MFMAs of k-step j
    f32x4 xa[2];
    xa[0] = *reinterpret_cast<const f32x4*>(a_lds);
#pragma unroll 1
    for (int t = 0; t < TRIPS; ++t) {
        const bool last = t == TRIPS - 1;
        const float* la = last ? next.wa : cur.wa + (t + 1) * FG_R * 256;
        const float* lb = last ? next.wb : cur.wb + (t + 1) * FG_R * 256;
        const float* ap = a_lds + t * FG_R * 16;
#pragma unroll
        for (int j = 0; j < FG_R; ++j) {
            // (the last fragment read of a unit re-reads k-step 0: harmless, keeps the loop uniform)
            if constexpr (EXPERIMENT != 2) xa[(j + 1) & 1] = *reinterpret_cast<const f32x4*>((j == FG_R - 1 && last) ? a_lds : ap + (j + 1) * 16);
            else xa[(j + 1) & 1] = xa[j & 1];
            __builtin_amdgcn_sched_barrier(0);          // ... and stays before them (left alone, hipcc sinks it behind the MFMAs into the same registers)
            const f32x4 wa = r.w[j][0], wb = r.w[j][1];
            const f32x4 x = xa[j & 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[e], x[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[e], x[e], acc1, 0, 0, 0);
            }
            if constexpr (EXPERIMENT != 1) {
                r.w[j][0] = fg_load(la + j * 256);
                r.w[j][1] = fg_load(lb + j * 256);
            }
            __builtin_amdgcn_sched_barrier(0);          // keep the loads where they are: R k-steps ahead of their use
        }
    }
}

}  // namespace d4
