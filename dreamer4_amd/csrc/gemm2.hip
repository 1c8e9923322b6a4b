// fp32 GEMM, second kernel family: v_mfma_f32_16x16x4_f32 fed by an LDS-DMA ring.
//
//   C[m, n] = epilogue( rowscale[m] * sum_k A[m, k] * W[n, k] )      (both operands K-contiguous, K % 32 == 0; k-tiles of 32 or 16)
//
// Same contract and epilogues as gemm_kernel (gemm.hip: RMSNorm folded as a row scale, bias, SiLU, SiLU-GLU pairing, residual,
// accumulate, row-compacted second output); what differs is how the matrix pipes are fed:
//   * operand tiles go global -> LDS directly (`buffer_load_dwordx4 ... lds`, 1 KB per wave-instruction) into a ring of NS stages
//     with counted `s_waitcnt vmcnt(N)` and one raw `s_barrier` per k-tile, so NS-2 k-tiles are in flight beyond the one being
//     multiplied and the next one (whose first fragments are read from LDS under the current MFMAs: the fragment registers are
//     software-pipelined by half k-tiles), and no VGPR or ds_write is spent on staging.  The LDS image is the plain [rows][32 floats] tile with the 16-byte chunks of a row XOR-swizzled by
//     (row & 7) on the SOURCE address (the DMA writes lane-linear), which makes every ds_read_b128 fragment read conflict-free.
//   * 16x16x4 MFMA (32-cycle issue, 4 accumulator registers per 16x16 sub-tile): block tiles are multiples of 16 rows, so a shape
//     can be cut into exactly 256 / 512 tiles (e.g. 3584 x 512 -> 32 x 8 tiles of 112 x 64: one per CU) instead of the 1.75 tiles per
//     CU a 64 x 64 grid gives.  The operands are swapped (W rows on the MFMA's row side) so that a lane ends up holding four
//     CONSECUTIVE columns of one output row: 16-byte epilogue loads / stores.
// Within this family every configuration walks k in the same order (chunk kq + 4h of each 32-k tile on lane group kq), so all of
// them give the same bits and the choice among them can be made by timing; the family itself is chosen by rule (shape only).
// The folded RMSNorm's row sums use the same canonical order as gemm_kernel (eight running sums per row, one per 16-byte chunk
// position, fused a0^2+a1^2+a2^2+a3^2, then the tree ((0+1)+(2+3))+((4+5)+(6+7))), so the row scale is bit-identical across families.
#include "common.h"
#include <hip/hip_ext.h>
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace d4 {

typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// swizzle of a row's 16-byte chunks: BK = 32 (8 chunks) chunk ^ (row & 7); BK = 16 (4 chunks) chunk ^ perm[(row >> 2) & 3] with
// perm = {0, 2, 3, 1}.  Either way the 16 lanes a ds_read_b128 services together ({0-3, 12-15, 20-27}, ...) hit 16 distinct slots.
template <int BK>
__device__ __forceinline__ int chunk_swizzle(int row) {
    if constexpr (BK == 32) return row & 7;
    else return (0x78 >> (((row >> 2) & 3) * 2)) & 3;          // {0, 2, 3, 1} packed two bits each: 0b01'11'10'00
}

template <int WGM, int WGN, int TM, int TN, int NS, int BK, bool RMS>
__device__ __forceinline__ void gemm2_body(GemmArgs& p, int bid, const int bz) {
    static_assert(BK == 16 || BK == 32, "k-tile");
    constexpr int CH = BK / 4;                      // 16-byte chunks per tile row
    constexpr int RPP = 64 / CH;                    // rows per 1 KB DMA piece (8 or 16)
    constexpr int H = BK / 16;                      // fragment reads (16 k each) per k-tile
    constexpr int NW = WGM * WGN, NT = NW * 64;
    constexpr int BM = WGM * TM * 16, BN = WGN * TN * 16;
    constexpr int STAGE_ROWS = BM + BN;
    constexpr int STAGE_F = STAGE_ROWS * BK;        // floats per ring stage
    constexpr int NSLOT = STAGE_ROWS / RPP;         // DMA pieces per stage
    constexpr int LPW = (NSLOT + NW - 1) / NW;      // pieces per wave per stage (a short last round re-sends the last piece)
    static_assert(NS >= 3 && NS <= 4 && 2 * LPW <= 63, "ring depth / vmcnt range");
    extern __shared__ __attribute__((aligned(16))) float smem[];    // [NS][STAGE_ROWS][BK] | rowscale[BM]   (ONE LDS object)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    // XCD-aware block order (as gemm_kernel): consecutive blocks on one XCD share an A row-panel
    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    {
        const int nblk = nbm * nbn, nx = 8;
        const int q = nblk / nx, r = nblk % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    }
    const int bm0 = (bid / nbn) * BM, bn0 = (bid % nbn) * BN;
    p.A += bz * p.strideA; p.W += bz * p.strideW; p.C += bz * p.strideC;
    if (p.R) p.R += bz * p.strideC;

    const int rowsA = min(BM, p.M - bm0), rowsB = min(BN, p.N - bn0);
    auto uniform_rsrc = [](const float* base, int64_t bytes) {
        const uint64_t b = reinterpret_cast<uint64_t>(base);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        const int nb = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7FFFFFFF ? bytes : 0x7FFFFFFF));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
    };
    // rows past the matrix edge fall outside num_records and arrive as zeros
    const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc(p.A + (int64_t)bm0 * p.lda, ((int64_t)(rowsA - 1) * p.lda + p.K) * 4);
    const __amdgpu_buffer_rsrc_t rsB = uniform_rsrc(p.W + (int64_t)bn0 * p.ldw, ((int64_t)(rowsB - 1) * p.ldw + p.K) * 4);

    // this lane's part of each DMA piece: row (lane / CH) of the piece's RPP rows, LDS chunk (lane % CH) <- source chunk ^ swizzle(row)
    uint32_t voff[LPW];
    const int prow = lane / CH;
    const int src_chunk = (lane % CH) ^ chunk_swizzle<BK>(prow);
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int slot = min(wave + NW * i, NSLOT - 1);
        const int r = slot * RPP + prow;
        voff[i] = (uint32_t)((slot * RPP < BM ? r * p.lda : (r - BM) * p.ldw) * 4 + src_chunk * 16);
    }
    // one ring stage: this wave's LPW pieces of k-tile KT, straight into LDS buffer BUF.  (A macro: the offset operand must be
    // passed as an rvalue — an lvalue there makes the host pass of the compilation drop the kernel stub without a diagnostic.)
#define D4_ISSUE_STAGE(KT, BUF)                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < LPW; ++i_) {                                                                      \
        const int slot_ = min(wave + NW * i_, NSLOT - 1);                                                                     \
        float* dst_ = smem + (BUF) * STAGE_F + slot_ * 256;                                                                   \
        if (slot_ * RPP < BM) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_ptr)dst_, 16, (uint32_t)voff[i_], (KT) * BK * 4, 0, 0); \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_ptr)dst_, 16, (uint32_t)voff[i_], (KT) * BK * 4, 0, 0);              \
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: lane (row = lane & 15, kq = lane >> 4) reads chunk kq (BK = 32: + 4 for the second half) of its row
    const int kq = lane >> 4, frow = lane & 15;
    const int foff0 = frow * BK + ((kq ^ chunk_swizzle<BK>(frow)) << 2);       // floats; BK = 32 second half: ^ 16 (chunk ^ 4)
    const int a_base = wm * TM * 16 * BK, b_base = BM * BK + wn * TN * 16 * BK;

    constexpr int NPAR = 32 / BK;                                     // BK = 16: chunk positions 0..3 on even k-tiles, 4..7 on odd ones
    constexpr int SQI = RMS ? (BM * CH + NT - 1) / NT : 1;            // 16-byte chunks of the A tile per thread (row sums)
    float ssq[SQI][NPAR];
#pragma unroll
    for (int i = 0; i < SQI; ++i)
#pragma unroll
        for (int h = 0; h < NPAR; ++h) ssq[i][h] = 0.f;

    const int nk = p.K / BK;
    // at most `stages` later ring stages of this wave may still be in flight
    auto wait_allow = [&](int stages) {
        if (stages >= 2) wait_vmcnt<2 * LPW>();
        else if (stages == 1) wait_vmcnt<LPW>();
        else wait_vmcnt<0>();
    };
    // fragment registers, software-pipelined: the set an MFMA group reads was loaded from LDS while the previous group ran.
    //   BK = 32: set 0 = chunks kq of the current k-tile (read during the previous iteration), set 1 = its chunks kq + 4
    //   BK = 16: one read per k-tile; the sets alternate with the k-tile parity
    f32x4 af[2][TM], bf[2][TN];
    auto read_frags = [&](const float* st, auto set_tag, int second_half) {
        constexpr int SET = decltype(set_tag)::value;
        const int fo = second_half ? (foff0 ^ 16) : foff0;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[SET][i] = *reinterpret_cast<const f32x4*>(st + a_base + i * 16 * BK + fo);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[SET][j] = *reinterpret_cast<const f32x4*>(st + b_base + j * 16 * BK + fo);
    };
    auto mfma_set = [&](auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[SET][j][e], af[SET][i][e], acc[i][j], 0, 0, 0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) { D4_ISSUE_STAGE(s, s) }
    wait_allow(min(NS - 2, nk - 1));
    __builtin_amdgcn_s_barrier();                           // k-tile 0 is visible to every wave
    read_frags(smem, S0{}, 0);

    auto k_tile = [&](int kt, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;       // kt & 1
        if (kt + 1 < nk) {
            wait_allow(min(kt + NS - 2, nk - 1) - (kt + 1));   // this wave's pieces of k-tile kt+1 have landed
            __builtin_amdgcn_s_barrier();                       // ... everyone's have; and everyone is done reading k-tile kt-1
            if (kt + NS - 1 < nk) { D4_ISSUE_STAGE(kt + NS - 1, (kt + NS - 1) % NS) }
        }
        const float* st = smem + (kt % NS) * STAGE_F;
        const float* nxt = smem + ((kt + 1) % NS) * STAGE_F;
        if constexpr (RMS) {
#pragma unroll
            for (int i = 0; i < SQI; ++i) {
                const int idx = tid + i * NT;
                if (BM * CH % NT == 0 || idx < BM * CH) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(st + idx * 4);
                    ssq[i][PAR % NPAR] = ssq[i][PAR % NPAR] + __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], v[0] * v[0])));
                }
            }
        }
        if constexpr (H == 2) {
            read_frags(st, S1{}, 1);
            mfma_set(S0{});
            if (kt + 1 < nk) read_frags(nxt, S0{}, 0);
            mfma_set(S1{});
        } else {
            using CUR = std::integral_constant<int, PAR>;
            using NXT = std::integral_constant<int, PAR ^ 1>;
            if (kt + 1 < nk) read_frags(nxt, NXT{}, 0);
            mfma_set(CUR{});
        }
    };
    for (int kt = 0; kt < nk; kt += 2) {
        k_tile(kt, S0{});
        if (kt + 1 < nk) k_tile(kt + 1, S1{});
    }

#undef D4_ISSUE_STAGE
    float* rowscale_s = smem + NS * STAGE_F;
    if constexpr (RMS) {
#pragma unroll
        for (int i = 0; i < SQI; ++i) {
            float s = ssq[i][0];
            s += dpp_f<0xB1>(s);
            s += dpp_f<0x4E>(s);
            if constexpr (NPAR == 1) {
                s += dpp_f<0x141>(s);                      // the other quad of the row's 8 lanes: (0123) + (4567)
            } else {
                float s1 = ssq[i][1];
                s1 += dpp_f<0xB1>(s1);
                s1 += dpp_f<0x4E>(s1);
                s = s + s1;                                // (0123) + (4567)
            }
            const int idx = tid + i * NT;
            if ((idx % CH) == 0 && idx < BM * CH) rowscale_s[idx / CH] = rsqrtf(s / (float)p.K + p.rms_eps);
        }
        __syncthreads();
    }

    // ---- epilogue.  D[i][j]: i = W row (n) = 4 * (lane >> 4) + reg, j = A row (m) = lane & 15  ->  four consecutive n per lane
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    const bool vecC = (p.ldc % 4) == 0 && ((uintptr_t)p.C % 16) == 0;
    const bool vecR = p.R && (p.ldr % 4) == 0 && ((uintptr_t)p.R % 16) == 0;
    const bool vecC2 = p.C2 && (p.ldc2 % 4) == 0 && ((uintptr_t)p.C2 % 16) == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm * TM * 16 + i * 16 + frow;
        const int gm = bm0 + ml;
        if (gm >= p.M) continue;
        const float rs = RMS ? rowscale_s[ml] : 1.f;
        int64_t c2row = -1;
        if (p.C2) {
            const int ts = gm % p.c2_S, keep = p.c2_hi - p.c2_lo;
            const int rank = (ts >= p.c2_lo && ts < p.c2_hi) ? ts - p.c2_lo : ((p.c2_last && ts == p.c2_S - 1) ? keep : -1);
            if (rank >= 0) c2row = (int64_t)(gm / p.c2_S) * (keep + p.c2_last) + rank;
        }
        if (swiglu) {
            if constexpr (TN % 4 == 0) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if ((j & 3) >= 2) continue;                                    // sub-tiles 0, 1 of a 64-column group hold values; 2, 3 their gates
                    const int gn = bn0 + wn * TN * 16 + j * 16 + kq * 4;           // packed column of the first value
                    if (gn >= p.N) continue;
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float val = acc[i][j][e] * rs, gate = acc[i][j + 2][e] * rs;
                        if (p.bias) { val += p.bias[gn + e]; gate += p.bias[gn + e + 32]; }
                        o[e] = val * siluf(gate);
                    }
                    const int on = (gn / 64) * 32 + (gn % 64);
                    float* cp = p.C + (int64_t)gm * p.ldc + on;
                    if (vecC) *reinterpret_cast<f32x4*>(cp) = o;
                    else { cp[0] = o[0]; cp[1] = o[1]; cp[2] = o[2]; cp[3] = o[3]; }
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = bn0 + wn * TN * 16 + j * 16 + kq * 4;
            if (gn >= p.N) continue;
            const bool full = gn + 3 < p.N;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * rs;
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (full || gn + e < p.N) v[e] += p.bias[gn + e];
            }
            if (p.flags & GEMM_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = siluf(v[e]);
            }
            if (p.R) {
                const float* rp = p.R + (int64_t)gm * p.ldr + gn;
                if (vecR && full) { const f32x4 r4 = *reinterpret_cast<const f32x4*>(rp); v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3]; }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (full || gn + e < p.N) v[e] += rp[e];
                }
            }
            float* cp = p.C + (int64_t)gm * p.ldc + gn;
            if (p.flags & GEMM_ACCUMULATE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (full || gn + e < p.N) v[e] += cp[e];
            }
            if (vecC && full) *reinterpret_cast<f32x4*>(cp) = v;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (full || gn + e < p.N) cp[e] = v[e];
            }
            if (c2row >= 0) {
                float* c2 = p.C2 + c2row * p.ldc2 + gn;
                if (vecC2 && full) *reinterpret_cast<f32x4*>(c2) = v;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (full || gn + e < p.N) c2[e] = v[e];
                }
            }
        }
    }
}

template <int WGM, int WGN, int TM, int TN, int NS, int BK, bool RMS>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm2_kernel(GemmArgs p) {
    gemm2_body<WGM, WGN, TM, TN, NS, BK, RMS>(p, blockIdx.x, blockIdx.y);
}

// Two independent GEMMs in ONE grid (blocks [0, nblk_a) work on `a`, the rest on `b`): launch ramp and tail are paid once.  Used for the
// attention pool's query and key projections (same K, same epilogue class); each problem keeps its own XCD-aware tile order and the bits of
// every output are those of the separate launches (same body, same k order).
template <int WGM, int WGN, int TM, int TN, int NS, int BK, bool RMS>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm2_pair_kernel(GemmArgs a, GemmArgs b, int nblk_a) {
    const int bid = blockIdx.x;
    if (bid < nblk_a) gemm2_body<WGM, WGN, TM, TN, NS, BK, RMS>(a, bid, 0);
    else gemm2_body<WGM, WGN, TM, TN, NS, BK, RMS>(b, bid - nblk_a, 0);
}

// ---- few rows x long K: the contraction cut FOUR ways inside the workgroup (round 6) ---------------------------------------------------
// The heads' hidden layers at rollout batch (256 x 2048 x 2048, 128 x 4096 x 4096 at config 5) are 512 tiles of 32 x 32 whose 64 k-tiles run one after
// the other: 40 us of k-loop LATENCY at 54 TF/s with half the chip idle.  Here a workgroup of 16 waves owns one 32 x 64 tile and its four wave groups
// (each exactly the 32 x 64 configuration above: 2 x 2 waves, wave tile 16 x 32, its own ring of three stages) each multiply one QUARTER of K; the four
// partial tiles meet in LDS (the rings are free by then) and are summed in group order — a fixed order, nothing crosses between workgroups, no atomics:
// deterministic, but NOT the bits of the family's other configurations (their k order is one chain), so this form is taken by a RULE on the shape
// (gemm2_ksplit_rule), never by the tuner.  Epilogue on the row-contiguous image: bias, SiLU, residual, 16-byte stores.
constexpr int KS_GROUPS = 4, KS_NS = 3, KS_BM = 32, KS_BN = 64, KS_BK = 32;
constexpr int KS_STAGE_F = (KS_BM + KS_BN) * KS_BK;                 // floats per ring stage of one group
constexpr int KS_RING_F = KS_NS * KS_STAGE_F;                       // 9216 floats = 36 KB per group
constexpr int KS_LDP = KS_BN + 4;                                   // partial tile row stride (floats)

__global__ __launch_bounds__(1024) void gemm2_ksplit_kernel(GemmArgs p) {
    constexpr int BK = KS_BK, CH = 8, RPP = 8, BM = KS_BM, BN = KS_BN, NS = KS_NS, LPW = 3, NSLOT = (BM + BN) / RPP;      // 12 pieces per stage, 3 per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];    // [4 groups][3 stages][96 rows][32]; afterwards [4][32][68] partial tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w = wave & 3, wm = w >> 1, wn = w & 1;
    int bid = blockIdx.x;
    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    {
        const int nblk = nbm * nbn, nx = 8;
        const int q = nblk / nx, r = nblk % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    }
    const int bm0 = (bid / nbn) * BM, bn0 = (bid % nbn) * BN;
    const int rowsA = min(BM, p.M - bm0), rowsB = min(BN, p.N - bn0);
    auto uniform_rsrc = [](const float* base, int64_t bytes) {
        const uint64_t b = reinterpret_cast<uint64_t>(base);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        const int nb = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7FFFFFFF ? bytes : 0x7FFFFFFF));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc(p.A + (int64_t)bm0 * p.lda, ((int64_t)(rowsA - 1) * p.lda + p.K) * 4);
    const __amdgpu_buffer_rsrc_t rsB = uniform_rsrc(p.W + (int64_t)bn0 * p.ldw, ((int64_t)(rowsB - 1) * p.ldw + p.K) * 4);
    uint32_t voff[LPW];
    const int prow = lane / CH;
    const int src_chunk = (lane % CH) ^ chunk_swizzle<BK>(prow);
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int slot = w + 4 * i;
        const int r = slot * RPP + prow;
        voff[i] = (uint32_t)((slot * RPP < BM ? r * p.lda : (r - BM) * p.ldw) * 4 + src_chunk * 16);
    }
    float* ring = smem + grp * KS_RING_F;
    // this group's k-tiles: the first (nkt % 4) groups take one more; every group RUNS nk = ceil(nkt / 4) iterations (same barrier sequence) and a
    // group's tile past its share is "loaded" through a zero-record descriptor: zeros in LDS, its MFMAs add nothing
    const int nkt = p.K / BK, nk = (nkt + KS_GROUPS - 1) / KS_GROUPS;
    const int mine = nkt / KS_GROUPS + (grp < nkt % KS_GROUPS ? 1 : 0);
    const int kt0 = grp * (nkt / KS_GROUPS) + min(grp, nkt % KS_GROUPS);
    const __amdgpu_buffer_rsrc_t rs0 = uniform_rsrc(p.A, 0);
#define D4_KS_ISSUE(KT, BUF)                                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < LPW; ++i_) {                                                                           \
        const int slot_ = w + 4 * i_;                                                                                               \
        float* dst_ = ring + (BUF) * KS_STAGE_F + slot_ * 256;                                                                      \
        const __amdgpu_buffer_rsrc_t rs_ = (KT) < mine ? (slot_ * RPP < BM ? rsA : rsB) : rs0;                                    \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void_ptr)dst_, 16, (uint32_t)voff[i_], (kt0 + (KT)) * BK * 4, 0, 0);           \
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int kq = lane >> 4, frow = lane & 15;
    const int foff0 = frow * BK + ((kq ^ chunk_swizzle<BK>(frow)) << 2);
    const int a_base = wm * 16 * BK, b_base = BM * BK + wn * 32 * BK;
    auto wait_allow = [&](int stages) {
        if (stages >= 2) wait_vmcnt<2 * LPW>();
        else if (stages == 1) wait_vmcnt<LPW>();
        else wait_vmcnt<0>();
    };
    f32x4 af[2], bf[2][2];
    auto read_frags = [&](const float* st, int set, int second_half) {
        const int fo = second_half ? (foff0 ^ 16) : foff0;
        af[set] = *reinterpret_cast<const f32x4*>(st + a_base + fo);
        bf[set][0] = *reinterpret_cast<const f32x4*>(st + b_base + fo);
        bf[set][1] = *reinterpret_cast<const f32x4*>(st + b_base + 16 * BK + fo);
    };
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nk) { D4_KS_ISSUE(st, st) }
    wait_allow(min(NS - 2, nk - 1));
    __builtin_amdgcn_s_barrier();
    read_frags(ring, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            wait_allow(min(kt + NS - 2, nk - 1) - (kt + 1));
            __builtin_amdgcn_s_barrier();
            if (kt + NS - 1 < nk) { D4_KS_ISSUE(kt + NS - 1, (kt + NS - 1) % NS) }
        }
        const float* st = ring + (kt % NS) * KS_STAGE_F;
        const float* nxt = ring + ((kt + 1) % NS) * KS_STAGE_F;
        read_frags(st, 1, 1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[0][j][e], af[0][e], acc[j], 0, 0, 0);
        if (kt + 1 < nk) read_frags(nxt, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[1][j][e], af[1][e], acc[j], 0, 0, 0);
    }
#undef D4_KS_ISSUE
    __syncthreads();                                                 // every group is past its last fragment read: the rings become the partial tiles
    float* part = smem + grp * KS_RING_F;                            // [32][KS_LDP]
#pragma unroll
    for (int j = 0; j < 2; ++j)
        *reinterpret_cast<f32x4*>(part + (wm * 16 + frow) * KS_LDP + wn * 32 + j * 16 + kq * 4) = acc[j];
    __syncthreads();
    if (tid >= BM * BN / 4) return;
    const int row = tid / (BN / 4), c4 = (tid % (BN / 4)) * 4;
    const int gm = bm0 + row, gn = bn0 + c4;
    if (gm >= p.M || gn >= p.N) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * KS_LDP + c4);
#pragma unroll
    for (int g = 1; g < KS_GROUPS; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(smem + g * KS_RING_F + row * KS_LDP + c4);
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    const bool full = gn + 3 < p.N;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (!full && gn + e >= p.N) continue;
        float x = v[e];
        if (p.bias) x += p.bias[gn + e];
        if (p.flags & GEMM_SILU) x = siluf(x);
        if (p.R) x += p.R[(int64_t)gm * p.ldr + gn + e];
        v[e] = x;
    }
    float* cp = p.C + (int64_t)gm * p.ldc + gn;
    if (full && (p.ldc % 4) == 0 && ((uintptr_t)p.C % 16) == 0) *reinterpret_cast<f32x4*>(cp) = v;
    else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (full || gn + e < p.N) cp[e] = v[e];
    }
}

// Measured (tools/gemm2_ksplit_probe.py, profiles/r06_gemm2_ksplit.txt): 256 x 2048 x 2048 27.2 us against 30.8 us for the best tiled configuration (40 us in
// situ), 128 x 4096 x 4096 46 against 55 us; 512 rows lose (51.5 vs 44.8 us) -> at most 256 rows.  Same-box A/B of the whole rollout: cfg 2 181.75 -> 180.32 ms,
// cfg 5 (bf16, B = 128, 6 frames) 78.0 -> 76.9 ms.
// the calls that take it: few rows (at most 256), a long contraction in whole quarters of 32-k tiles, at most ~4 workgroups per CU, plain epilogue
bool gemm2_ksplit_applicable(const GemmArgs& p) {               // what the kernel can run at all (test hook: d4_gemm_force_config(199))
    return gemm2_applicable(p) && p.batch <= 1 && !p.C2 && !p.Wb && !p.Ab && !(p.flags & ~GEMM_SILU) && p.K >= 4 * KS_BK;
}
bool gemm2_ksplit_rule(const GemmArgs& p) {
    if (!gemm2_ksplit_applicable(p)) return false;
    const int64_t tiles = (int64_t)cdiv(p.M, KS_BM) * cdiv(p.N, KS_BN);
    if (p.M <= 256) return p.K >= 1024 && tiles >= 64 && tiles <= 1024;
    return false;
}

int gemm2_ksplit_launch(const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    D4_REQUIRE(gemm2_ksplit_applicable(p), "gemm2_ksplit: call not supported (M=%d N=%d K=%d flags=%d)", p.M, p.N, p.K, p.flags);
    const size_t lds = (size_t)KS_GROUPS * KS_RING_F * sizeof(float);
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm2_ksplit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.done();
    }
    const dim3 grid(cdiv(p.M, KS_BM) * cdiv(p.N, KS_BN)), block(1024);
    if (ea) hipExtLaunchKernelGGL(gemm2_ksplit_kernel, grid, block, (uint32_t)lds, stream, ea, eb, 0, p);
    else hipLaunchKernelGGL(gemm2_ksplit_kernel, grid, block, lds, stream, p);
    D4_LAUNCH_CHECK();
    return 0;
}

// ---- configurations -------------------------------------------------------------------------------------------
// name          waves   wave tile   block tile  BK  LDS ring     blocks (waves) / CU
// 64x64         2 x 2     32 x 32     64 x 64   32  3 x 16 KB    3 (12)
// 64x64/s       4 x 1     16 x 64     64 x 64   32  3 x 16 KB    3 (12)   SiLU-GLU capable: a wave spans 64 columns
// 128x64/8      4 x 2     32 x 32    128 x 64   32  3 x 24 KB    2 (16)
// 128x64/s      4 x 1     32 x 64    128 x 64   32  3 x 24 KB    2 (8)    SiLU-GLU capable
// 128x64/k16    4 x 2     32 x 32    128 x 64   16  4 x 12 KB    3 (24)
// 128x128/k16   4 x 2     32 x 64    128 x 128  16  3 x 16 KB    3 (24)   SiLU-GLU capable
// 32x64         2 x 2     16 x 32     32 x 64   32  3 x 12 KB    4 (16)   small GEMMs (< 2 blocks of 64 x 64 per CU): twice the blocks
// 32x64/s       2 x 1     16 x 64     32 x 64   32  3 x 12 KB    4 (8)    SiLU-GLU capable
// Measured on the engine's shapes (tools/gemm2_bench.py, profiles/r02_gemm_families.txt) and dropped: 128x128 with 4 or 8 waves and
// 256x128 at either BK (one or two blocks per CU leave every barrier exposed: 1.3-3x slower on the N <= 512 shapes), 112x64 /
// 128x64 with 4 waves, 64x64 with a 4-deep ring or at BK = 16, 32x128 (SiLU-GLU capable) and 32x64 at BK = 16 (never ahead of the forms
// above).  Waves per CU is what pays: the 32-row tiles added last (4-6 blocks per CU) win every N <= 512 shape.
enum { V2_64x64 = 0, V2_64x64_s, V2_128x64_8, V2_128x64_s, V2_128x64_k16, V2_128x128_k16, V2_32x64, V2_32x64_s, V2_32x32, V2_64x32, V2_N };
static const char* const kV2Kernel[V2_N] = {"gemm2_kernel<2, 2, 2, 2, 3, 32", "gemm2_kernel<4, 1, 1, 4, 3, 32", "gemm2_kernel<4, 2, 2, 2, 3, 32", "gemm2_kernel<4, 1, 2, 4, 3, 32",
                                            "gemm2_kernel<4, 2, 2, 2, 4, 16", "gemm2_kernel<4, 2, 2, 4, 3, 16", "gemm2_kernel<2, 2, 1, 2, 3, 32", "gemm2_kernel<2, 1, 1, 4, 3, 32", "gemm2_kernel<2, 2, 1, 1, 3, 32", "gemm2_kernel<4, 1, 1, 2, 3, 32"};
static const int kV2BM[V2_N] = {64, 64, 128, 128, 128, 128, 32, 32, 32, 64}, kV2BN[V2_N] = {64, 64, 64, 64, 64, 128, 64, 64, 32, 32};

int gemm2_configs() { return V2_N; }
const char* gemm2_config_name(int c) { return c >= 0 && c < V2_N ? kV2Kernel[c] : ""; }
void gemm2_config_tile(int c, int* bm, int* bn) { *bm = kV2BM[c]; *bn = kV2BN[c]; }

bool gemm2_applicable(const GemmArgs& p) {
    return !(p.flags & (GEMM_TRANS_A | GEMM_TRANS_B)) && (p.K % 32) == 0 && (p.lda % 4) == 0 && (p.ldw % 4) == 0 && p.M >= 1;
}

bool gemm2_config_valid(int c, const GemmArgs& p) {
    if (c < 0 || c >= V2_N || !gemm2_applicable(p)) return false;
    if (p.flags & GEMM_SWIGLU) return c == V2_64x64_s || c == V2_128x64_s || c == V2_128x128_k16 || c == V2_32x64_s;
    return true;
}

template <int WGM, int WGN, int TM, int TN, int NS, int BK>
static int launch2(const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    constexpr int BM = WGM * TM * 16, BN = WGN * TN * 16;
    const size_t lds = (size_t)(NS * (BM + BN) * BK + BM) * sizeof(float);
    const bool rms = (p.flags & GEMM_RMS_ROWSCALE) != 0;
    auto k = rms ? gemm2_kernel<WGM, WGN, TM, TN, NS, BK, true> : gemm2_kernel<WGM, WGN, TM, TN, NS, BK, false>;
    static DeviceOnce attr_set[2];
    if (attr_set[rms].need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[rms].done();
    }
    const dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), p.batch > 0 ? p.batch : 1), block(WGM * WGN * 64);
    if (ea) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)lds, stream, ea, eb, 0, p);
    else hipLaunchKernelGGL(k, grid, block, lds, stream, p);
    D4_LAUNCH_CHECK();
    return 0;
}

template <int WGM, int WGN, int TM, int TN, int NS, int BK>
static int launch2_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    constexpr int BM = WGM * TM * 16, BN = WGN * TN * 16;
    const size_t lds = (size_t)(NS * (BM + BN) * BK + BM) * sizeof(float);
    const bool rms = (a.flags & GEMM_RMS_ROWSCALE) != 0;
    auto k = rms ? gemm2_pair_kernel<WGM, WGN, TM, TN, NS, BK, true> : gemm2_pair_kernel<WGM, WGN, TM, TN, NS, BK, false>;
    static DeviceOnce attr_set[2];
    if (attr_set[rms].need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[rms].done();
    }
    const int na = cdiv(a.M, BM) * cdiv(a.N, BN), nb = cdiv(b.M, BM) * cdiv(b.N, BN);
    const dim3 grid(na + nb), block(WGM * WGN * 64);
    if (ea) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)lds, stream, ea, eb, 0, a, b, na);
    else hipLaunchKernelGGL(k, grid, block, lds, stream, a, b, na);
    D4_LAUNCH_CHECK();
    return 0;
}

bool gemm2_pair_config_ok(int c) { return c == V2_64x64 || c == V2_128x64_8 || c == V2_128x64_k16 || c == V2_32x64 || c == V2_32x32 || c == V2_64x32; }

bool gemm2_pair_applicable(const GemmArgs& a, const GemmArgs& b) {
    const int fa = a.flags & ~GEMM_RMS_ROWSCALE, fb = b.flags & ~GEMM_RMS_ROWSCALE;
    return gemm2_applicable(a) && gemm2_applicable(b) && a.batch <= 1 && b.batch <= 1 && ((a.flags ^ b.flags) & GEMM_RMS_ROWSCALE) == 0 &&
           !(fa & GEMM_SWIGLU) && !(fb & GEMM_SWIGLU) && a.M > 0 && b.M > 0;
}

int gemm2_pair_launch(int c, const GemmArgs& a, const GemmArgs& b, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    D4_REQUIRE(gemm2_pair_applicable(a, b) && gemm2_config_valid(c, a) && gemm2_config_valid(c, b), "gemm2 pair: configuration %d is not valid for these calls", c);
    switch (c) {
        case V2_64x64: return launch2_pair<2, 2, 2, 2, 3, 32>(a, b, stream, ea, eb);
        case V2_128x64_8: return launch2_pair<4, 2, 2, 2, 3, 32>(a, b, stream, ea, eb);
        case V2_128x64_k16: return launch2_pair<4, 2, 2, 2, 4, 16>(a, b, stream, ea, eb);
        case V2_32x64: return launch2_pair<2, 2, 1, 2, 3, 32>(a, b, stream, ea, eb);
        case V2_32x32: return launch2_pair<2, 2, 1, 1, 3, 32>(a, b, stream, ea, eb);
        case V2_64x32: return launch2_pair<4, 1, 1, 2, 3, 32>(a, b, stream, ea, eb);
    }
    return 2;       // the remaining configurations are not instantiated in pair form
}

int gemm2_launch(int c, const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    D4_REQUIRE(gemm2_config_valid(c, p), "gemm2: configuration %d is not valid for this call", c);
    switch (c) {
        case V2_64x64: return launch2<2, 2, 2, 2, 3, 32>(p, stream, ea, eb);
        case V2_64x64_s: return launch2<4, 1, 1, 4, 3, 32>(p, stream, ea, eb);
        case V2_128x64_8: return launch2<4, 2, 2, 2, 3, 32>(p, stream, ea, eb);
        case V2_128x64_s: return launch2<4, 1, 2, 4, 3, 32>(p, stream, ea, eb);
        case V2_128x64_k16: return launch2<4, 2, 2, 2, 4, 16>(p, stream, ea, eb);
        case V2_128x128_k16: return launch2<4, 2, 2, 4, 3, 16>(p, stream, ea, eb);
        case V2_32x64: return launch2<2, 2, 1, 2, 3, 32>(p, stream, ea, eb);
        case V2_32x64_s: return launch2<2, 1, 1, 4, 3, 32>(p, stream, ea, eb);
        case V2_32x32: return launch2<2, 2, 1, 1, 3, 32>(p, stream, ea, eb);
        case V2_64x32: return launch2<4, 1, 1, 2, 3, 32>(p, stream, ea, eb);
    }
    return 2;
}

}  // namespace d4
