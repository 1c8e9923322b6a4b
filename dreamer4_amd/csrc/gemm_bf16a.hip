// bf16 GEMM with bf16 ACTIVATIONS, LDS-DMA ring (BASELINE config 5, round 3): both operands arrive in HBM as bf16 and go global -> LDS
// directly, nothing is converted or staged through registers.
//
//   C[m, n] = epilogue( rowscale[m] * sum_k Ab[m, k] * Wb[n, k] )      Ab bf16 [M][K] (the producer's bf16 copy of the activation),
//                                                                      Wb bf16 [N][K], C fp32 (+ optionally Cb = bf16(C), the copy the
//                                                                      next GEMM reads)
//
// This is gemm2.hip's structure with 2-byte elements: a k-tile is 64 bf16 = 128 bytes per row, so the DMA geometry (8 rows per 1 KB
// wave-instruction, chunks XOR-swizzled by row & 7 on the source address, conflict-free ds_read_b128 fragments), the ring (3-4 stages,
// counted s_waitcnt vmcnt, one raw s_barrier per k-tile) and the software-pipelined fragment sets are the same; what changes is the
// matrix instruction — v_mfma_f32_16x16x32_bf16 takes a lane's whole 16-byte chunk (8 consecutive k) in ONE instruction where the
// f32 kernel issues four — so a k-tile of a 16 x 16 sub-tile is 2 MFMAs (32 cycles) instead of 8 (256 cycles) and the kernel is bound by
// operand delivery (DMA + LDS reads), not by the matrix pipe.  Halving the activation bytes (the register-staged kernel fetches fp32
// activations and rounds them on the way into LDS) is therefore what pays; the producer side is the epilogue's Cb store here and one
// conversion pass after the non-GEMM producers (engine.hip).
// The folded RMSNorm's 1/rms is accumulated from the bf16 tile in LDS (the register-staged kernel takes it from the fp32 registers:
// the two differ by the bf16 rounding of the activations, inside this mode's stated tolerance).
#include "common.h"
#include <hip/hip_ext.h>
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace d4 {

typedef __attribute__((address_space(3))) void* lds_void_ptr_b;
typedef __bf16 bf16x8_b __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_b __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vmcnt_b() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WGM, int WGN, int TM, int TN, int NS, bool RMS>
__device__ __forceinline__ void gemm_bf16a_body(GemmArgs p, const int block_x, const int block_y) {
    constexpr int BKB = 128;                        // bytes per tile row = 64 bf16
    constexpr int CH = 8, RPP = 8;                  // 16-byte chunks per row; rows per 1 KB DMA piece
    constexpr int NW = WGM * WGN, NT = NW * 64;
    constexpr int BM = WGM * TM * 16, BN = WGN * TN * 16;
    constexpr int STAGE_ROWS = BM + BN;
    constexpr int STAGE_B = STAGE_ROWS * BKB;       // bytes per ring stage
    constexpr int NSLOT = STAGE_ROWS / RPP;
    constexpr int LPW = (NSLOT + NW - 1) / NW;
    static_assert(NS >= 2 && NS <= 8 && (NS - 2) * LPW <= 63, "ring depth / vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem_b[];   // [NS][STAGE_ROWS][128 B] | rowscale[BM] floats

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    constexpr bool LATE = false;             // (measured: issuing the ring refill behind the first half's MFMAs changes nothing)

    int bid = block_x;
    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    {
        const int nblk = nbm * nbn, nx = 8;
        const int q = nblk / nx, r = nblk % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    }
    int tr = bid / nbn, tc = bid % nbn;
    if (p.group_m > 0) {                                     // grouped order: bid -> (group of row panels, column, row inside the group)
        const int per = p.group_m * nbn, g = bid / per, r = bid % per;
        const int gm_eff = min(p.group_m, nbm - g * p.group_m);
        tr = g * p.group_m + r % gm_eff; tc = r / gm_eff;
    }
    const int bm0 = tr * BM, bn0 = tc * BN;
    const int bz = block_y;
    const uint16_t* Ab = p.Ab + bz * p.strideA;
    const uint16_t* Wb = p.Wb + bz * p.strideW;
    if (p.C) p.C += bz * p.strideC;                  // (C may be null: only the bf16 copy Cb is wanted)
    if (p.Cb) p.Cb += bz * p.strideC;
    if (p.R) p.R += bz * p.strideC;

    const int rowsA = min(BM, p.M - bm0), rowsB = min(BN, p.N - bn0);
    auto uniform_rsrc = [](const void* base, int64_t bytes) {
        const uint64_t b = reinterpret_cast<uint64_t>(base);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        const int nb = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7FFFFFFF ? bytes : 0x7FFFFFFF));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
    };
    // rows past the matrix edge fall outside num_records and arrive as zeros
    const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc(Ab + (int64_t)bm0 * p.lda, ((int64_t)(rowsA - 1) * p.lda + p.K) * 2);
    const __amdgpu_buffer_rsrc_t rsB = uniform_rsrc(Wb + (int64_t)bn0 * p.ldw, ((int64_t)(rowsB - 1) * p.ldw + p.K) * 2);

    uint32_t voff[LPW];
    const int prow = lane / CH;
    const int src_chunk = (lane % CH) ^ (prow & 7);
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int slot = min(wave + NW * i, NSLOT - 1);
        const int r = slot * RPP + prow;
        voff[i] = (uint32_t)((slot * RPP < BM ? r * p.lda : (r - BM) * p.ldw) * 2 + src_chunk * 16);
    }
#define D4_ISSUE_STAGE_B(KT, BUF)                                                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < LPW; ++i_) {                                                                      \
        const int slot_ = min(wave + NW * i_, NSLOT - 1);                                                                     \
        char* dst_ = smem_b + (BUF) * STAGE_B + slot_ * 1024;                                                                 \
        if (slot_ * RPP < BM) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_ptr_b)dst_, 16, (uint32_t)voff[i_], (KT) * BKB, 0, 0); \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_ptr_b)dst_, 16, (uint32_t)voff[i_], (KT) * BKB, 0, 0);              \
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses (bytes): lane (row = lane & 15, kq = lane >> 4) reads chunk kq of its row, then chunk kq + 4 (offset ^ 64)
    const int kq = lane >> 4, frow = lane & 15;
    const int foff0 = frow * BKB + ((kq ^ (frow & 7)) << 4);
    const int a_base = wm * TM * 16 * BKB, b_base = BM * BKB + wn * TN * 16 * BKB;

    constexpr int SQI = RMS ? (BM * CH + NT - 1) / NT : 1;
    float ssq[SQI];
#pragma unroll
    for (int i = 0; i < SQI; ++i) ssq[i] = 0.f;

    const int nk = p.K / 64;
    auto wait_allow = [&](int stages) {              // at most NS - 2 stages are ever allowed to stay in flight
        if (NS >= 8 && stages >= 6) wait_vmcnt_b<(NS >= 8 ? 6 : 0) * LPW>();
        else if (NS >= 7 && stages == 5) wait_vmcnt_b<(NS >= 7 ? 5 : 0) * LPW>();
        else if (NS >= 6 && stages == 4) wait_vmcnt_b<(NS >= 6 ? 4 : 0) * LPW>();
        else if (NS >= 5 && stages == 3) wait_vmcnt_b<(NS >= 5 ? 3 : 0) * LPW>();
        else if (stages >= 2) wait_vmcnt_b<2 * LPW>();
        else if (stages == 1) wait_vmcnt_b<LPW>();
        else wait_vmcnt_b<0>();
    };
    bf16x8_b af[2][TM], bf[2][TN];
    auto read_frags = [&](const char* st, auto set_tag, int second_half) {
        constexpr int SET = decltype(set_tag)::value;
        const int fo = second_half ? (foff0 ^ 64) : foff0;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[SET][i] = *reinterpret_cast<const bf16x8_b*>(st + a_base + i * 16 * BKB + fo);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[SET][j] = *reinterpret_cast<const bf16x8_b*>(st + b_base + j * 16 * BKB + fo);
    };
    auto mfma_set = [&](auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[SET][j], af[SET][i], acc[i][j], 0, 0, 0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    if constexpr (NS == 2) {
        // Two slots (the 256 x 256 tile: a stage is 64 KB): stage kt + 1 is in flight while stage kt is multiplied; one barrier per k-tile — it says
        // "stage kt has landed in every wave's pieces AND everyone is done reading the other slot".  No fragment pre-read across the barrier.
        D4_ISSUE_STAGE_B(0, 0)
        for (int kt = 0; kt < nk; ++kt) {
            wait_vmcnt_b<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < nk) { D4_ISSUE_STAGE_B(kt + 1, (kt + 1) & 1) }
            const char* st = smem_b + (kt & 1) * STAGE_B;
            if constexpr (RMS) {
#pragma unroll
                for (int i = 0; i < SQI; ++i) {
                    const int idx = tid + i * NT;
                    if (BM * CH % NT == 0 || idx < BM * CH) {
                        const bf16x8_b v = *reinterpret_cast<const bf16x8_b*>(st + idx * 16);
                        float s = ssq[i];
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s = __builtin_fmaf(f, f, s); }
                        ssq[i] = s;
                    }
                }
            }
            read_frags(st, S0{}, 0);
            mfma_set(S0{});
            read_frags(st, S0{}, 1);
            mfma_set(S0{});
        }
        __builtin_amdgcn_s_barrier();                        // (the row scales below reuse nothing of the ring, but keep the waves together for the epilogue)
    } else {
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) { D4_ISSUE_STAGE_B(s, s) }
    wait_allow(min(NS - 2, nk - 1));
    __builtin_amdgcn_s_barrier();
    read_frags(smem_b, S0{}, 0);

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            wait_allow(min(kt + NS - 2, nk - 1) - (kt + 1));
            __builtin_amdgcn_s_barrier();
            if (!LATE && kt + NS - 1 < nk) { D4_ISSUE_STAGE_B(kt + NS - 1, (kt + NS - 1) % NS) }
        }
        const char* st = smem_b + (kt % NS) * STAGE_B;
        const char* nxt = smem_b + ((kt + 1) % NS) * STAGE_B;
        if constexpr (RMS) {
#pragma unroll
            for (int i = 0; i < SQI; ++i) {
                const int idx = tid + i * NT;
                if (BM * CH % NT == 0 || idx < BM * CH) {
                    const bf16x8_b v = *reinterpret_cast<const bf16x8_b*>(st + idx * 16);
                    float s = ssq[i];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s = __builtin_fmaf(f, f, s); }
                    ssq[i] = s;
                }
            }
        }
        read_frags(st, S1{}, 1);
        mfma_set(S0{});
        // LATE: the ring refill is issued BEHIND the first half's MFMAs, so the matrix pipe has work queued while the DMA pieces issue
        if (LATE && kt + 1 < nk && kt + NS - 1 < nk) { D4_ISSUE_STAGE_B(kt + NS - 1, (kt + NS - 1) % NS) }
        if (kt + 1 < nk) read_frags(nxt, S0{}, 0);
        mfma_set(S1{});
    }
    }
#undef D4_ISSUE_STAGE_B

    float* rowscale_s = reinterpret_cast<float*>(smem_b + NS * STAGE_B);
    if constexpr (RMS) {
#pragma unroll
        for (int i = 0; i < SQI; ++i) {
            float s = ssq[i];
            s += dpp_f<0xB1>(s);
            s += dpp_f<0x4E>(s);
            s += dpp_f<0x141>(s);                      // the 8 lanes (chunks) of a row: (0123) + (4567)
            const int idx = tid + i * NT;
            if ((idx % CH) == 0 && idx < BM * CH) rowscale_s[idx / CH] = rsqrtf(s / (float)p.K + p.rms_eps);
        }
        __syncthreads();
    }

    // ---- epilogue (gemm2.hip's): D[i][j]: i = W row (n) = 4 * (lane >> 4) + reg, j = A row (m) = lane & 15 -> four consecutive n per lane
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    const bool vecC = p.C && (p.ldc % 4) == 0 && ((uintptr_t)p.C % 16) == 0;
    const bool vecR = p.R && (p.ldr % 4) == 0 && ((uintptr_t)p.R % 16) == 0;
    const bool vecC2 = p.C2 && (p.ldc2 % 4) == 0 && ((uintptr_t)p.C2 % 16) == 0;
    auto store_b = [&](int64_t row, int col, const f32x4& v, bool full, int ncols) {          // bf16 copy for the next GEMM
        if (!p.Cb) return;
        uint16_t* cb = p.Cb + row * p.ldc + col;
        if (full && (p.ldc % 4) == 0 && ((uintptr_t)p.Cb % 8) == 0) {
            bf16x4_b o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x4_b*>(cb) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (col + e < ncols) { const __bf16 h = (__bf16)v[e]; cb[e] = __builtin_bit_cast(uint16_t, h); }
        }
    };
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
                    if ((j & 3) >= 2) continue;
                    const int gn = bn0 + wn * TN * 16 + j * 16 + kq * 4;
                    if (gn >= p.N) continue;
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float val = acc[i][j][e] * rs, gate = acc[i][j + 2][e] * rs;
                        if (p.bias) { val += p.bias[gn + e]; gate += p.bias[gn + e + 32]; }
                        o[e] = val * siluf_fast(gate);
                    }
                    const int on = (gn / 64) * 32 + (gn % 64);
                    float* cp = p.C + (int64_t)gm * p.ldc + on;
                    if (vecC) *reinterpret_cast<f32x4*>(cp) = o;
                    else if (p.C) { cp[0] = o[0]; cp[1] = o[1]; cp[2] = o[2]; cp[3] = o[3]; }
                    store_b(gm, on, o, true, p.N / 2);
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
            if ((p.flags & GEMM_ACCUMULATE) && p.C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (full || gn + e < p.N) v[e] += cp[e];
            }
            if (vecC && full) *reinterpret_cast<f32x4*>(cp) = v;
            else if (p.C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (full || gn + e < p.N) cp[e] = v[e];
            }
            store_b(gm, gn, v, full, p.N);
            if (c2row >= 0) {
                float* c2 = p.C2 + c2row * p.ldc2 + gn;
                if (vecC2 && full) *reinterpret_cast<f32x4*>(c2) = v;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (full || gn + e < p.N) c2[e] = v[e];
                }
                if (p.C2b) {                                   // bf16 copy of the compacted rows
                    uint16_t* cb = p.C2b + c2row * p.ldc2 + gn;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (full || gn + e < p.N) { const __bf16 h = (__bf16)v[e]; cb[e] = __builtin_bit_cast(uint16_t, h); }
                }
            }
        }
    }
}

template <int WGM, int WGN, int TM, int TN, int NS, bool RMS>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_bf16a_kernel(GemmArgs p) {
    gemm_bf16a_body<WGM, WGN, TM, TN, NS, RMS>(p, blockIdx.x, blockIdx.y);
}

// Two independent products of equal K in ONE grid (blocks [0, nblk_a) work on `a`, the rest on `b`; each keeps its own tile order and the bits of
// the separate launches): the attention pool's query projection (a few dozen tiles) rides in the key projection's launch.
template <int WGM, int WGN, int TM, int TN, int NS, bool RMS>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_bf16a_pair_kernel(GemmArgs a, GemmArgs b, int nblk_a) {
    const int bid = blockIdx.x;
    if (bid < nblk_a) gemm_bf16a_body<WGM, WGN, TM, TN, NS, RMS>(a, bid, 0);
    else gemm_bf16a_body<WGM, WGN, TM, TN, NS, RMS>(b, bid - nblk_a, 0);
}

// ---- configurations --------------------------------------------------------------------------------------------
// name        waves   wave tile   block tile   ring            blocks / CU
// 128x128     4 x 2    32 x 64    128 x 128    3 x 32 KB       1          SiLU-GLU capable
// 128x64      4 x 2    32 x 32    128 x 64     3 x 24 KB       2
// 64x64       2 x 2    32 x 32     64 x 64     4 x 16 KB       2
// 64x64/s     4 x 1    16 x 64     64 x 64     4 x 16 KB       2          SiLU-GLU capable
// 32x64       2 x 2    16 x 32     32 x 64     4 x 12 KB       3
// 256x128     4 x 2    64 x 64    256 x 128    3 x 48 KB       1          SiLU-GLU capable (large problems)
// 256x192     4 x 3    64 x 64    256 x 192    2 x 56 KB       1          SiLU-GLU capable; 12 waves; where its tile count fits the machine's rounds better
// 256x256     2 x 4   128 x 64    256 x 256    8 x 16 KB       1          SiLU-GLU capable; gemm_bf16p.hip (round 6): 8 waves, phased k-loop with the two wave rows half a
//                                                                        phase apart, ring of half-tiles 1.5 k-tiles deep.  It replaced the 16-wave two-slot form of
//                                                                        round 4 (4 x 4 waves of 64 x 64, one barrier per k-tile): cube 8192 1174 -> 1298 TF/s, the
//                                                                        SiLU-GLU input projection at 14336 rows 180 -> 162 us on one box (profiles/r06_bf16p_probe.txt)
// Measured and retired (round 4, tools/bf16a_probe.py): rings of 4 - 8 stages at one workgroup per CU (the whole LDS in flight) are level or slower
// than these — two co-resident workgroups hide more than a deeper ring; the k-loop step (~0.6 - 0.7 us for 128 x 128 x 64 even on a quarter of
// the CUs) is bound inside the workgroup (barrier / DMA issue / fragment reads per k-tile), not by bytes in flight.
enum { VA_128x128 = 0, VA_128x64, VA_64x64, VA_64x64_s, VA_32x64, VA_256x128, VA_256x256, VA_256x192, VA_N };
static const int kVaBM[VA_N] = {128, 128, 64, 64, 32, 256, 256, 256}, kVaBN[VA_N] = {128, 64, 64, 64, 64, 128, 256, 192};

int gemm_bf16a_configs() { return VA_N; }

bool gemm_bf16a_applicable(const GemmArgs& p) {
    return p.Ab != nullptr && p.Wb != nullptr && p.wplane == 0 && !(p.flags & (GEMM_TRANS_A | GEMM_TRANS_B)) && (p.K % 64) == 0 && (p.lda % 8) == 0 &&
           (p.ldw % 8) == 0 && ((uintptr_t)p.Ab % 16) == 0 && ((uintptr_t)p.Wb % 16) == 0 && (p.strideW % 8) == 0 && (p.strideA % 8) == 0 && p.M >= 1;
}

bool gemm_bf16a_config_valid(int c, const GemmArgs& p) {
    if (c < 0 || c >= VA_N || !gemm_bf16a_applicable(p)) return false;
    if (p.flags & GEMM_SWIGLU) return c == VA_128x128 || c == VA_64x64_s || c == VA_256x128 || c == VA_256x256 || c == VA_256x192;
    return true;
}

template <int WGM, int WGN, int TM, int TN, int NS>
static int launch_va(const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    constexpr int BM = WGM * TM * 16, BN = WGN * TN * 16;
    const size_t lds = (size_t)NS * (BM + BN) * 128 + BM * sizeof(float);
    const bool rms = (p.flags & GEMM_RMS_ROWSCALE) != 0;
    auto k = rms ? gemm_bf16a_kernel<WGM, WGN, TM, TN, NS, true> : gemm_bf16a_kernel<WGM, WGN, TM, TN, NS, false>;
    static DeviceOnce attr_set[2];
    if (attr_set[rms].need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[rms].done();
    }
    const dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), p.batch > 0 ? p.batch : 1), block(WGM * WGN * 64);
    GemmArgs q = p;
    if (q.group_m == 0) {
        // rows of operand panels the ~32 concurrent tiles of one XCD touch: gm * BM + ceil(32 / gm) * BN, minimised over the group height
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

int gemm_bf16a_launch(int c, const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    D4_REQUIRE(gemm_bf16a_config_valid(c, p), "gemm_bf16a: configuration %d is not valid for this call", c);
    switch (c) {
        case VA_128x128: return launch_va<4, 2, 2, 4, 3>(p, stream, ea, eb);
        case VA_128x64: return launch_va<4, 2, 2, 2, 3>(p, stream, ea, eb);
        case VA_64x64: return launch_va<2, 2, 2, 2, 4>(p, stream, ea, eb);
        case VA_64x64_s: return launch_va<4, 1, 1, 4, 4>(p, stream, ea, eb);
        case VA_32x64: return launch_va<2, 2, 1, 2, 4>(p, stream, ea, eb);
        case VA_256x128: return launch_va<4, 2, 4, 4, 3>(p, stream, ea, eb);
        case VA_256x256: return gemm_bf16p_launch(p, stream, ea, eb);          // (round 6: the phased kernel of gemm_bf16p.hip replaced the 16-wave two-slot form)
        case VA_256x192: return launch_va<4, 3, 4, 4, 2>(p, stream, ea, eb);
    }
    return 2;
}

int gemm_bf16a_rule(const GemmArgs& p);
static void bf16a_group(GemmArgs& q, int BM, int BN) {
    const int nbm = cdiv(q.M, BM), nbn = cdiv(q.N, BN);
    int best = 1; long best_cost = -1;
    for (int gm = 1; gm <= 16 && gm <= nbm; ++gm) {
        const int cols = (32 + gm - 1) / gm;
        const long cost = (long)gm * BM + (long)(cols < nbn ? cols : nbn) * BN;
        if (best_cost < 0 || cost < best_cost) { best = gm; best_cost = cost; }
    }
    q.group_m = best;
}

template <int WGM, int WGN, int TM, int TN, int NS>
static int launch_va_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    constexpr int BM = WGM * TM * 16, BN = WGN * TN * 16;
    const size_t lds = (size_t)NS * (BM + BN) * 128 + BM * sizeof(float);
    auto k = gemm_bf16a_pair_kernel<WGM, WGN, TM, TN, NS, true>;
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.done();
    }
    GemmArgs qa = a, qb = b;
    bf16a_group(qa, BM, BN); bf16a_group(qb, BM, BN);
    const int na = cdiv(a.M, BM) * cdiv(a.N, BN), nb = cdiv(b.M, BM) * cdiv(b.N, BN);
    const dim3 grid(na + nb), block(WGM * WGN * 64);
    if (ea) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)lds, stream, ea, eb, 0, qa, qb, na);
    else hipLaunchKernelGGL(k, grid, block, lds, stream, qa, qb, na);
    D4_LAUNCH_CHECK();
    return 0;
}

// both: bf16 activations + weights, the folded RMSNorm, no batch, equal K; the tile is the LARGER problem's (by the shape rule), restricted to the
// two 8-wave forms that carry the pool projections
bool gemm_bf16a_pair_applicable(const GemmArgs& a, const GemmArgs& b) {
    auto ok = [](const GemmArgs& p) { return gemm_bf16a_applicable(p) && (p.flags == GEMM_RMS_ROWSCALE) && p.batch <= 1 && !p.C2 && !p.R && !p.bias && p.M > 0; };
    if (!ok(a) || !ok(b) || a.K != b.K) return false;
    const GemmArgs& big = (double)a.M * a.N >= (double)b.M * b.N ? a : b;
    const int c = gemm_bf16a_rule(big);
    return c == VA_128x64 || c == VA_64x64;
}
int gemm_bf16a_pair_launch(const GemmArgs& a, const GemmArgs& b, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    D4_REQUIRE(gemm_bf16a_pair_applicable(a, b), "gemm_bf16a_pair: call not supported");
    const GemmArgs& big = (double)a.M * a.N >= (double)b.M * b.N ? a : b;
    if (gemm_bf16a_rule(big) == VA_128x64) return launch_va_pair<4, 2, 2, 2, 3>(a, b, stream, ea, eb);
    return launch_va_pair<2, 2, 2, 2, 4>(a, b, stream, ea, eb);
}

// tile by shape (a rule, never a timing)
int gemm_bf16a_rule(const GemmArgs& p) {
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    const int nb = p.batch > 0 ? p.batch : 1;
    const int64_t t128 = (int64_t)cdiv(p.M, 128) * cdiv(p.N, 128) * nb;
    const int64_t t128x64 = (int64_t)cdiv(p.M, 128) * cdiv(p.N, 64) * nb;
    const int64_t t64 = (int64_t)cdiv(p.M, 64) * cdiv(p.N, 64) * nb;
    // 256 x 256 (half the operand bytes per flop) once its tiles cover more than half the CUs — measured (tools/bf16a_probe.py, round 4): the SiLU-GLU
    // input projection at 1792 rows (154 tiles) 46.6 -> 36.8 us, at 14336 rows 290 -> 202 us; the SiLU-GLU output projection at 14336 rows 131 -> 80 us;
    // the pool key projection at 114688 rows 143 -> 91 us.  Below that it leaves CUs idle and loses (28 tiles: 65 vs 25 us).
    const int64_t t256 = (int64_t)cdiv(p.M, 256) * cdiv(p.N, 256) * nb;
    const int64_t t192 = (int64_t)cdiv(p.M, 256) * cdiv(p.N, 192) * nb;
    // one nearly full round of 256 x 128 tiles where the 256 x 256 form would leave half the CUs idle (the wide key projection at 2 x 1792 rows,
    // N = 1792 .. 2304: 27-28 us against 30-31 us on 256 x 192 tiles, tools/bf16a_widekeys_probe.py)
    const int64_t t256x128 = (int64_t)cdiv(p.M, 256) * cdiv(p.N, 128) * nb;
    if (!swiglu && t256 < 140 && t256x128 >= 190 && t256x128 <= 256) return VA_256x128;
    if ((t256 >= 140 || t192 >= 140) && p.N >= 256) {
        // 256 x 192 (12 waves) when its rounds of the machine cost less tile area than the 256 x 256 form's: the SiLU-GLU input projection at 1792 rows
        // (203 tiles in one round against 154: 36.5 -> 31.8 us), the fused q/k/v projection at 14336 rows (N = 1552: 73.6 -> 62.4 us); at 14336 x 5504
        // the square tile stays (193 vs 216 us)
        const int64_t c256 = t256 >= 140 ? (int64_t)cdiv(t256, 256) * 256 * 256 : INT64_MAX;
        const int64_t c192 = t192 >= 140 ? (int64_t)cdiv(t192, 256) * 256 * 192 : INT64_MAX;
        // (round 6: the 256 x 256 form is the phased kernel, ~15 % faster per tile area than the two-slot structure the 256 x 192 form still has —
        //  tools/bf16p_probe.py: 14336 x 1552 68 vs 67 us, 14336 x 5504 154 vs 196 us, 1792 x 5504 27.5 vs 26.8 us)
        return (double)c192 < 0.85 * (double)c256 ? VA_256x192 : VA_256x256;
    }
    if (swiglu) return t128 >= 200 ? VA_128x128 : VA_64x64_s;
    if (t128 >= 400 && p.N >= 128) return VA_128x128;
    if (t128x64 >= 300) return VA_128x64;
    if (t64 >= 300) return VA_64x64;
    return VA_32x64;
}

}  // namespace d4
