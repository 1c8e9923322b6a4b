// bf16 GEMM, 256 x 256 tile, PHASED k-loop (BASELINE config 5, round 6): the large-tile form of gemm_bf16a.hip rebuilt around what bounds that
// file's 256 x 256 kernel — one barrier per k-tile with every wave reading fragments and multiplying in lock-step, the ring refilled a single
// k-tile ahead (35.6 us for a 16-k-tile tile whose matrix work is 13.7 us).
//
//   C[m, n] = epilogue( rowscale[m] * sum_k Ab[m, k] * Wb[n, k] )      same operands, epilogues and k order per accumulator as gemm_bf16a.hip
//
// Structure (one workgroup = 8 waves = 2 (M) x 4 (N), wave tile 128 x 64, one workgroup per CU):
//   * The operand stream is cut into HALF-TILES of 128 rows x 64 k (16 KB): a k-tile is { A-lo, B-lo, B-hi, A-hi }, where A-lo holds rows [0, 64) of
//     BOTH wave rows' 128-row panels, A-hi rows [64, 128), B-lo columns [0, 32) of all four wave columns' 64-column panels, B-hi columns [32, 64).
//     LDS-DMA gathers any rows, so a half-tile is exactly what every wave reads in ONE phase; its ring slot is free right after that phase.
//   * A phase = a LOAD segment (the ds_read_b128 of the half-tiles it consumes, the 2 DMA pieces of the half-tile SIX half-tiles ahead, the folded
//     RMSNorm's sum of squares from fragments already in registers) then a COMPUTE segment (16 v_mfma_f32_16x16x32_bf16 = one 64 x 32 quadrant of the
//     wave tile over the whole k-tile), each closed by a raw s_barrier.  Four phases per k-tile: (a0,b0) (a0,b1) (a1,b1) (a1,b0), reading 12 / 4 / 8 / 0
//     fragments: a0 and a1 share ONE register set (64 fragment registers beside the 128 accumulators: the two waves of a SIMD have 256 each).
//   * The two wave rows run HALF A PHASE APART (waves 4-7 execute one extra barrier up front): waves w and w + 4 share a SIMD, so while one of them is in
//     its compute segment the other is in its load segment — the matrix pipe of every SIMD always has a wave feeding it, and LDS reads / DMA issue
//     ride under the partner's MFMAs instead of in front of the wave's own.
//   * Ring of 8 half-tile slots (128 KB); phase G issues half-tile G + 6 (1.5 k-tiles ahead) and ends its load segment with one counted
//     s_waitcnt vmcnt(8 | 10) (never 0 in the loop): a piece has four phases to land.  Half-tile H = 4 kt + j is read in phase r(H) = 4 kt + {0,0,1,2}[j].
//     Hazards (group 0 runs the load segment of phase G after its barrier #2G + 1, group 1 after its #2G + 2):
//       RAW  every wave waited for its own pieces of the half-tiles read in phase G at the end of ITS load segment of phase G - 1 and then arrived at
//            a barrier (its #2G resp. #2G + 1), which any reader of phase G has passed.
//       WAR  phase G refills the slot of half-tile G - 2, read in a phase <= G - 2; those reads are retired (lgkmcnt(0) at the head of every compute
//            segment) before the reader's barrier #2G at the latest, which both writers of phase G have passed.
//     Half-tiles past the end of K are "loaded" through a descriptor of zero records (no memory traffic, zeros into a dead slot) so the counted waits
//     stay uniform to the last phase.
#include "common.h"
#include <hip/hip_ext.h>
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace d4 {

typedef __attribute__((address_space(3))) void* lds_void_ptr_p;
typedef __bf16 bf16x8_p __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_p __attribute__((ext_vector_type(4)));

#define D4P_FENCE() asm volatile("" ::: "memory")
#define D4P_BARRIER()                          \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        D4P_FENCE();                           \
        __builtin_amdgcn_s_barrier();          \
        D4P_FENCE();                           \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

// identity the optimiser cannot see through: an address built from it is recomputed where it is used (one VALU add) instead of being hoisted out of
// the k-loop into a register of its own — the loop runs at the 256-register limit of two waves per SIMD
__device__ __forceinline__ int opaque_v(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ uint32_t opaque_v(uint32_t v) { asm volatile("" : "+v"(v)); return v; }

template <bool RMS>
__global__ __launch_bounds__(512) void gemm_bf16p_kernel(GemmArgs p) {
    constexpr int BM = 256, BN = 256;
    constexpr int HALF_B = 16384;                   // bytes per half-tile slot: 128 rows x 128 B
    constexpr int NSLOT = 8;
    constexpr int STAGE_RS = 256 + 16, STAGE_BYTES = 8 * 64 * STAGE_RS;      // epilogue staging: per wave [64 rows][64 fp32 + 16 B] (139 264 B >= the 128 KB ring)
    extern __shared__ __attribute__((aligned(16))) char smem_p[];   // ring [8][128 rows][128 B] (the epilogue's staging area afterwards) | rowscale[256] floats

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;        // wave row (= ping-pong group), wave column

    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM, ntiles = nbm * nbn;
    auto tile_origin = [&](int bid, int& bm0, int& bn0) {
        {
            const int nx = 8;
            const int q = ntiles / nx, r = ntiles % nx, x = bid % nx, o = bid / nx;
            bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
        }
        int tr = bid / nbn, tc = bid % nbn;
        if (p.group_m > 0) {
            const int per = p.group_m * nbn, g = bid / per, r = bid % per;
            const int gm_eff = min(p.group_m, nbm - g * p.group_m);
            tr = g * p.group_m + r % gm_eff; tc = r / gm_eff;
        }
        bm0 = tr * BM; bn0 = tc * BN;
    };
    int bm0, bn0;
    tile_origin(blockIdx.x, bm0, bn0);
    const int bz = blockIdx.y;
    const uint16_t* Ab = p.Ab + bz * p.strideA;
    const uint16_t* Wb = p.Wb + bz * p.strideW;

    auto uniform_rsrc = [](const void* base, int64_t bytes) {
        const uint64_t b = reinterpret_cast<uint64_t>(base);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        const int nb = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7FFFFFFF ? bytes : 0x7FFFFFFF));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
    };
    // rows past the matrix edge fall outside num_records (the row offset rides in the VGPR offset, which IS range-checked) and arrive as zeros
    auto rsrc_a = [&](int m0) { return uniform_rsrc(Ab + (int64_t)m0 * p.lda, ((int64_t)(min(BM, p.M - m0) - 1) * p.lda + p.K) * 2); };
    auto rsrc_b = [&](int n0) { return uniform_rsrc(Wb + (int64_t)n0 * p.ldw, ((int64_t)(min(BN, p.N - n0) - 1) * p.ldw + p.K) * 2); };
    const __amdgpu_buffer_rsrc_t rsA = rsrc_a(bm0), rsB = rsrc_b(bn0);
    const __amdgpu_buffer_rsrc_t rs0 = uniform_rsrc(Ab, 0);        // zero records: every access is out of range

    // DMA geometry: a half-tile is 16 pieces of 1 KB (8 slot rows x 128 B); wave w moves pieces w and w + 8.  Slot row s of piece (w, i):
    //   s = (w + 8 i) * 8 + lane / 8;  the 16-byte chunk at LDS position lane % 8 is source chunk (lane % 8) ^ (s & 7)
    //   A-lo / A-hi : tile row = (s >> 6) * 128 + (s & 63) (+ 64 for hi)       B-lo / B-hi : tile column = (s >> 5) * 64 + (s & 31) (+ 32 for hi)
    // s of piece i = 1 is s of piece 0 + 64: tile row + 128 (A), tile column + 128 (B).
    const int s0 = wave * 8 + (lane >> 3);
    const int srcc = ((lane & 7) ^ (s0 & 7)) << 4;
    const uint32_t voffA = (uint32_t)(s0 * p.lda * 2 + srcc);
    const uint32_t voffB = (uint32_t)((((s0 >> 5) * 64 + (s0 & 31)) * p.ldw) * 2 + srcc);
    const uint32_t stepA_piece = (uint32_t)(128 * p.lda * 2), stepA_hi = (uint32_t)(64 * p.lda * 2);
    const uint32_t stepB_piece = (uint32_t)(128 * p.ldw * 2), stepB_hi = (uint32_t)(32 * p.ldw * 2);

    const int nk = p.K / 64;
    // half-tile `type` (0 A-lo, 1 B-lo, 2 B-hi, 3 A-hi) of k-tile kt into ring slot `slot`; k-tiles past the end of K are "loaded" through the
    // zero-record descriptor (no memory traffic, zeros into a dead slot): the counted waits stay uniform to the last phase
    auto issue_half = [&](auto type_tag, int kt, auto slot_tag) {
        constexpr int TYPE = decltype(type_tag)::value, SLOT = decltype(slot_tag)::value;
        constexpr bool IS_A = TYPE == 0 || TYPE == 3, IS_HI = TYPE >= 2;
        const __amdgpu_buffer_rsrc_t rs = kt < nk ? (IS_A ? rsA : rsB) : rs0;
        const int ko = kt * 128;
        const uint32_t v0 = (IS_A ? voffA : voffB) + (IS_HI ? (IS_A ? stepA_hi : stepB_hi) : 0u);
        const uint32_t v1 = v0 + (IS_A ? stepA_piece : stepB_piece);
        char* dst = smem_p + SLOT * HALF_B + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr_p)dst, 16, v0, ko, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr_p)(dst + 8 * 1024), 16, v1, ko, 0, 0);
    };

    f32x4 acc[8][4];
    // fragment addresses (bytes inside a slot): lane (row = lane & 15, kq = lane >> 4) reads chunk kq of its row, then chunk kq + 4 (offset ^ 64)
    const int kq = lane >> 4, frow = lane & 15;
    const int foff0 = frow * 128 + ((kq ^ (frow & 7)) << 4);
    const int aoff = grp * 64 * 128, boff = wc * 32 * 128;
    bf16x8_p af[4][2], b0[2][2], b1[2][2];          // ONE set of A fragments: rows [0, 64) of the wave tile in phases 0 - 1, rows [64, 128) in phases 2 - 3
    // A fragments are held in an m-tile order ROTATED by the wave column: register index i = m-tile (i + wc) & 3 of the 64-row half, so that index 0 is
    // the m-tile whose row sums (folded RMSNorm) this wave owns — a compile-time register, a run-time (wave-uniform) LDS offset
    int rot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rot[i] = __builtin_amdgcn_readfirstlane(((i + wc) & 3) * 2048);

    const int fa0 = aoff + foff0, fa1 = aoff + (foff0 ^ 64), fb0 = boff + foff0, fb1 = boff + (foff0 ^ 64);
    auto read_a = [&](bf16x8_p (&dst)[4][2], auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        const int f0 = fa0, f1 = fa1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[i][0] = *reinterpret_cast<const bf16x8_p*>(smem_p + (f0 + (SLOT * HALF_B + rot[i])));
            dst[i][1] = *reinterpret_cast<const bf16x8_p*>(smem_p + (f1 + (SLOT * HALF_B + rot[i])));
        }
    };
    auto read_b = [&](bf16x8_p (&dst)[2][2], auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        const int f0 = fb0, f1 = fb1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            dst[j][0] = *reinterpret_cast<const bf16x8_p*>(smem_p + (f0 + (SLOT * HALF_B + j * 2048)));
            dst[j][1] = *reinterpret_cast<const bf16x8_p*>(smem_p + (f1 + (SLOT * HALF_B + j * 2048)));
        }
    };
    auto quadrant = [&](auto mi_tag, auto nj_tag, const bf16x8_p (&af)[4][2], const bf16x8_p (&bf)[2][2]) {
        constexpr int MI = decltype(mi_tag)::value, NJ = decltype(nj_tag)::value;       // accumulator block: rows MI * 4 .. + 4, columns NJ * 2 .. + 2
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[MI * 4 + i][NJ * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][h], af[i][h], acc[MI * 4 + i][NJ * 2 + j], 0, 0, 0);
    };
    // folded RMSNorm: the wave column wc owns the row sums of m-tile wc of a0 and of a1 (register index 0 of each, see `rot`): v_dot2c_f32_bf16 squares
    // and adds two bf16 per instruction in fp32
    float ssq[2] = {0.f, 0.f};
    auto sumsq = [&](const bf16x8_p (&af)[4][2], auto which_tag) {
        constexpr int WHICH = decltype(which_tag)::value;
        if constexpr (RMS) {
            typedef __bf16 bf16x2_p __attribute__((ext_vector_type(2)));
            float s = ssq[WHICH];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const bf16x2_p v = {af[0][h][e], af[0][h][e + 1]};
                    s = __builtin_amdgcn_fdot2_f32_bf16(v, v, s, false);
                }
            ssq[WHICH] = s;
        }
    };

    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>; using T3 = std::integral_constant<int, 3>;
#define D4P_SLOT(N) std::integral_constant<int, (N) % NSLOT>{}

    // One phase, P = phase index mod 8 (compile time), kb = the k-tile of phase 0 of this group of eight.  Half-tile H = 4 kt + j (j: 0 A-lo, 1 B-lo,
    // 2 B-hi, 3 A-hi) lives in slot H % 8 and is read in phase 4 kt + {0, 0, 1, 2}[j]; phase G issues half-tile G + 6.
    auto phase = [&](auto p_tag, int kb) {
        constexpr int P = decltype(p_tag)::value;
        constexpr int IT = (P + 6) % 4;                      // type issued in this phase
        const int kt_issue = kb + (P + 6) / 4;
        // ---- load segment
        if constexpr (P % 4 == 0) { read_b(b0, D4P_SLOT(P + 1)); read_a(af, D4P_SLOT(P)); }
        else if constexpr (P % 4 == 1) read_b(b1, D4P_SLOT(P + 1));
        else if constexpr (P % 4 == 2) read_a(af, D4P_SLOT(P + 1));
        issue_half(std::integral_constant<int, IT>{}, kt_issue, D4P_SLOT(P + 6));
        if constexpr (P % 4 == 1) sumsq(af, T0{});           // rows [0, 64): read in phase 0, retired at the head of its compute segment
        if constexpr (P % 4 == 3) sumsq(af, T1{});           // rows [64, 128): read in phase 2
        // everything read in the NEXT phase has landed (this wave's pieces): half-tiles up to G + 2, i.e. all but the last four issued — except ahead of
        // a phase 3, which reads nothing: there A-lo of the next k-tile (G + 2) may stay in flight one phase longer
        if constexpr (P % 4 == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        D4P_BARRIER();
        // ---- compute segment
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (P % 4 == 0) quadrant(T0{}, T0{}, af, b0);
        else if constexpr (P % 4 == 1) quadrant(T0{}, T1{}, af, b1);
        else if constexpr (P % 4 == 2) quadrant(T1{}, T1{}, af, b1);
        else quadrant(T1{}, T0{}, af, b0);
        __builtin_amdgcn_s_setprio(0);
        D4P_BARRIER();
    };

    // ---- prologue: half-tiles 0 .. 5 in flight, the first two (A-lo, B-lo of k-tile 0) landed
    issue_half(T0{}, 0, D4P_SLOT(0)); issue_half(T1{}, 0, D4P_SLOT(1)); issue_half(T2{}, 0, D4P_SLOT(2)); issue_half(T3{}, 0, D4P_SLOT(3));
    issue_half(T0{}, 1, D4P_SLOT(4)); issue_half(T1{}, 1, D4P_SLOT(5));
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    D4P_BARRIER();
    if (grp == 1) D4P_BARRIER();                              // the second wave row runs half a phase behind the first

#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int kb = 0;
    for (; kb + 1 < nk; kb += 2) {
        phase(std::integral_constant<int, 0>{}, kb); phase(std::integral_constant<int, 1>{}, kb);
        phase(std::integral_constant<int, 2>{}, kb); phase(std::integral_constant<int, 3>{}, kb);
        phase(std::integral_constant<int, 4>{}, kb); phase(std::integral_constant<int, 5>{}, kb);
        phase(std::integral_constant<int, 6>{}, kb); phase(std::integral_constant<int, 7>{}, kb);
    }
    if (kb < nk) {                                            // odd k-tile count: one more k-tile on the even slots
        phase(std::integral_constant<int, 0>{}, kb); phase(std::integral_constant<int, 1>{}, kb);
        phase(std::integral_constant<int, 2>{}, kb); phase(std::integral_constant<int, 3>{}, kb);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (only zero-record pieces are left: the ring must be quiet before the epilogue reuses it)
    if (grp == 0) D4P_BARRIER();                              // the wave rows meet again: every wave is past its last fragment read

    GemmArgs& q = p;
    if (q.C) q.C += bz * q.strideC;
    if (q.Cb) q.Cb += bz * q.strideC;
    if (q.R) q.R += bz * q.strideC;
    float* rowscale_s = reinterpret_cast<float*>(smem_p + STAGE_BYTES);
    if constexpr (RMS) {
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            float s = ssq[w];                                 // chunks kq and kq + 4 of row frow; the other six chunks sit in lanes frow + 16 q
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (kq == 0) rowscale_s[grp * 128 + w * 64 + wc * 16 + frow] = rsqrtf(s / (float)q.K + q.rms_eps);
        }
    }
    __syncthreads();

    // ---- epilogue (gemm_bf16a.hip's): D[i][j]: i = W row (n) = 4 * (lane >> 4) + reg, j = A row (m) = lane & 15 -> four consecutive n per lane
    const bool swiglu = (q.flags & GEMM_SWIGLU) != 0;
    const bool vecC = q.C && (q.ldc % 4) == 0 && ((uintptr_t)q.C % 16) == 0;
    const bool vecR = q.R && (q.ldr % 4) == 0 && ((uintptr_t)q.R % 16) == 0;
    const bool vecC2 = q.C2 && (q.ldc2 % 4) == 0 && ((uintptr_t)q.C2 % 16) == 0;
    auto store_b = [&](int64_t row, int col, const f32x4& v, bool full, int ncols) {          // bf16 copy for the next GEMM
        if (!q.Cb) return;
        uint16_t* cb = q.Cb + row * q.ldc + col;
        if (full && (q.ldc % 4) == 0 && ((uintptr_t)q.Cb % 8) == 0) {
            bf16x4_p o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x4_p*>(cb) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (col + e < ncols) { const __bf16 h = (__bf16)v[e]; cb[e] = __builtin_bit_cast(uint16_t, h); }
        }
    };
    // SiLU-GLU into the bf16 image only (the engine's form of the feedforward's input projection): the products are staged through a wave-private
    // [128 rows][64 + 16 B] image in the ring (quiet since the __syncthreads above: every wave had drained its DMA pieces before it) and leave as
    // 16-byte stores — 8 per lane in 64-byte row segments instead of 16 eight-byte ones in 32-byte segments.  LDS accesses of one wave execute in order.
    if (swiglu && !q.C && q.Cb && !q.C2 && (q.ldc % 8) == 0 && ((uintptr_t)q.Cb % 16) == 0) {
        const int gn0 = bn0 + wc * 64;
        if (gn0 >= q.N) return;
        constexpr int RS = 80;
        char* stg = smem_p + wave * (128 * RS);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = (i & 4) * 16 + (((i & 3) + wc) & 3) * 16 + frow;
            const float rs = RMS ? rowscale_s[grp * 128 + rl] : 1.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gn = gn0 + j * 16 + kq * 4;
                bf16x4_p o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float val = acc[i][j][e] * rs, gate = acc[i][j + 2][e] * rs;
                    if (q.bias) { val += q.bias[gn + e]; gate += q.bias[gn + e + 32]; }
                    o[e] = (__bf16)(val * siluf_fast(gate));
                }
                *reinterpret_cast<bf16x4_p*>(stg + rl * RS + (j * 16 + kq * 4) * 2) = o;
            }
        }
        uint16_t* cb = q.Cb + (gn0 / 64) * 32 + (lane & 3) * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int rl = r * 16 + (lane >> 2);
            const bf16x8_p v = *reinterpret_cast<const bf16x8_p*>(stg + rl * RS + (lane & 3) * 16);
            const int gm = bm0 + grp * 128 + rl;
            if (gm < q.M) *reinterpret_cast<bf16x8_p*>(cb + (int64_t)gm * q.ldc) = v;
        }
        return;
    }
    // Every other epilogue with whole 4-column groups: the scaled accumulators of a 64-row half of the wave tile are staged through a wave-private
    // [64 rows][64 fp32 + 16 B] image and re-read ROW-contiguous (lane = 4 columns of one row, a wave instruction = 4 rows x 256 B), so that the residual
    // / accumulate loads, the fp32 store and the bf16 image move whole 128-byte lines — and, what matters more, the 16 residual loads of a half are all
    // requested before the first is used: in the MFMA layout each of the 32 sub-tiles of a wave exposed one memory round trip in turn (in situ the
    // feedforward's output projection at 14336 rows ran 131 us against 76 us without a residual).
    if (!swiglu && (q.N % 4) == 0 && (!q.C || vecC) && (!q.R || vecR) && (!q.C2 || vecC2) && (!q.Cb || ((q.ldc % 4) == 0 && ((uintptr_t)q.Cb % 8) == 0)) &&
        (!q.C2b || ((q.ldc2 % 4) == 0 && ((uintptr_t)q.C2b % 8) == 0)) && (!q.bias || ((uintptr_t)q.bias % 16) == 0)) {
        const int col = (lane & 15) * 4, gn = bn0 + wc * 64 + col;
        if (bn0 + wc * 64 >= q.N) return;
        const bool col_ok = gn < q.N;                                     // (N % 4 == 0: a lane's four columns are in or out together)
        char* stg = smem_p + wave * (64 * STAGE_RS);
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (q.bias && col_ok) bias4 = *reinterpret_cast<const f32x4*>(q.bias + gn);
        const int keep = q.c2_hi - q.c2_lo;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int row0 = bm0 + grp * 128 + half * 64;                 // first matrix row of this half
            f32x4 rr[16];
            if (q.R) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = row0 + r * 4 + (lane >> 4);
                    rr[r] = (col_ok && gm < q.M) ? *reinterpret_cast<const f32x4*>(q.R + (int64_t)gm * q.ldr + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int rl = ((i4 + wc) & 3) * 16 + frow;               // register index i = m-tile (i + wc) & 3 of its 64-row half
                const float rs = RMS ? rowscale_s[grp * 128 + half * 64 + rl] : 1.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[half * 4 + i4][j][e] * rs;
                    *reinterpret_cast<f32x4*>(stg + rl * STAGE_RS + (j * 16 + kq * 4) * 4) = v;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = r * 4 + (lane >> 4), gm = row0 + rl;
                f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * STAGE_RS + col * 4);
                if (!col_ok || gm >= q.M) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bias4[e];
                if (q.flags & GEMM_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = siluf(v[e]);
                }
                if (q.R) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rr[r][e];
                }
                if (q.C) {
                    float* cp = q.C + (int64_t)gm * q.ldc + gn;
                    if (q.flags & GEMM_ACCUMULATE) { const f32x4 c4 = *reinterpret_cast<const f32x4*>(cp); v[0] += c4[0]; v[1] += c4[1]; v[2] += c4[2]; v[3] += c4[3]; }
                    *reinterpret_cast<f32x4*>(cp) = v;
                }
                bf16x4_p vb;
#pragma unroll
                for (int e = 0; e < 4; ++e) vb[e] = (__bf16)v[e];
                if (q.Cb) *reinterpret_cast<bf16x4_p*>(q.Cb + (int64_t)gm * q.ldc + gn) = vb;
                if (q.C2) {
                    const int ts = gm % q.c2_S;
                    const int rank = (ts >= q.c2_lo && ts < q.c2_hi) ? ts - q.c2_lo : ((q.c2_last && ts == q.c2_S - 1) ? keep : -1);
                    if (rank >= 0) {
                        const int64_t c2row = (int64_t)(gm / q.c2_S) * (keep + q.c2_last) + rank;
                        *reinterpret_cast<f32x4*>(q.C2 + c2row * q.ldc2 + gn) = v;
                        if (q.C2b) *reinterpret_cast<bf16x4_p*>(q.C2b + c2row * q.ldc2 + gn) = vb;
                    }
                }
            }
        }
        return;
    }
    // MFMA-layout fallback (ragged column counts, unaligned outputs, a SiLU-GLU that also wants fp32): gemm_bf16a.hip's epilogue
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ml = grp * 128 + (i & 4) * 16 + (((i & 3) + wc) & 3) * 16 + frow;          // register index i = m-tile (i + wc) & 3 of its 64-row half
        const int gm = bm0 + ml;
        if (gm >= q.M) continue;
        const float rs = RMS ? rowscale_s[ml] : 1.f;
        int64_t c2row = -1;
        if (q.C2) {
            const int ts = gm % q.c2_S, keep = q.c2_hi - q.c2_lo;
            const int rank = (ts >= q.c2_lo && ts < q.c2_hi) ? ts - q.c2_lo : ((q.c2_last && ts == q.c2_S - 1) ? keep : -1);
            if (rank >= 0) c2row = (int64_t)(gm / q.c2_S) * (keep + q.c2_last) + rank;
        }
        if (swiglu) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gn = bn0 + wc * 64 + j * 16 + kq * 4;
                if (gn >= q.N) continue;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float val = acc[i][j][e] * rs, gate = acc[i][j + 2][e] * rs;
                    if (q.bias) { val += q.bias[gn + e]; gate += q.bias[gn + e + 32]; }
                    o[e] = val * siluf_fast(gate);
                }
                const int on = (gn / 64) * 32 + (gn % 64);
                float* cp = q.C + (int64_t)gm * q.ldc + on;
                if (vecC) *reinterpret_cast<f32x4*>(cp) = o;
                else if (q.C) { cp[0] = o[0]; cp[1] = o[1]; cp[2] = o[2]; cp[3] = o[3]; }
                store_b(gm, on, o, true, q.N / 2);
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = bn0 + wc * 64 + j * 16 + kq * 4;
            if (gn >= q.N) continue;
            const bool full = gn + 3 < q.N;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * rs;
            if (q.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (full || gn + e < q.N) v[e] += q.bias[gn + e];
            }
            if (q.flags & GEMM_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = siluf(v[e]);
            }
            if (q.R) {
                const float* rp = q.R + (int64_t)gm * q.ldr + gn;
                if (vecR && full) { const f32x4 r4 = *reinterpret_cast<const f32x4*>(rp); v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3]; }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (full || gn + e < q.N) v[e] += rp[e];
                }
            }
            float* cp = q.C + (int64_t)gm * q.ldc + gn;
            if ((q.flags & GEMM_ACCUMULATE) && q.C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (full || gn + e < q.N) v[e] += cp[e];
            }
            if (vecC && full) *reinterpret_cast<f32x4*>(cp) = v;
            else if (q.C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (full || gn + e < q.N) cp[e] = v[e];
            }
            store_b(gm, gn, v, full, q.N);
            if (c2row >= 0) {
                float* c2 = q.C2 + c2row * q.ldc2 + gn;
                if (vecC2 && full) *reinterpret_cast<f32x4*>(c2) = v;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (full || gn + e < q.N) c2[e] = v[e];
                }
                if (q.C2b) {
                    uint16_t* cb = q.C2b + c2row * q.ldc2 + gn;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (full || gn + e < q.N) { const __bf16 h = (__bf16)v[e]; cb[e] = __builtin_bit_cast(uint16_t, h); }
                }
            }
        }
    }
}

#undef D4P_SLOT

int gemm_bf16p_launch(const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    constexpr int BM = 256, BN = 256;
    const size_t lds = (size_t)8 * 64 * (256 + 16) + BM * sizeof(float);          // max(ring 8 x 16 KB, epilogue staging) + row scales
    const bool rms = (p.flags & GEMM_RMS_ROWSCALE) != 0;
    auto k = rms ? gemm_bf16p_kernel<true> : gemm_bf16p_kernel<false>;
    static DeviceOnce attr_set[2];
    if (attr_set[rms].need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[rms].done();
    }
    const dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), p.batch > 0 ? p.batch : 1), block(512);
    GemmArgs q = p;
    if (q.group_m == 0) {
        const int nbm = cdiv(p.M, BM), nbn = cdiv(p.N, BN);
        int best = 1; long best_cost = -1;
        for (int gm = 1; gm <= 16 && gm <= nbm; ++gm) {
            const int cols = (32 + gm - 1) / gm;
            const long cost = (long)gm * BM + (long)(cols < nbn ? cols : nbn) * BN;
            if (best_cost < 0 || cost < best_cost) { best = gm; best_cost = cost; }
        }
        q.group_m = best;
    } else if (q.group_m < 0) q.group_m = 0;
    if (ea) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)lds, stream, ea, eb, 0, q);
    else hipLaunchKernelGGL(k, grid, block, lds, stream, q);
    D4_LAUNCH_CHECK();
    return 0;
}

}  // namespace d4
