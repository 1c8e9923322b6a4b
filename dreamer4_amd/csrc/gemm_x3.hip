// fp32 GEMM on the bf16 matrix cores by operand splitting ("bf16x3 operands, six products") — third fp32 GEMM family.
//
//   C[m, n] = epilogue( rowscale[m] * sum_k A[m, k] * W[n, k] )        A fp32 [M][K], W fp32 given as three bf16 planes, C fp32
//
// On gfx950 the f32-input MFMA runs at 1/16 of the bf16 MFMA rate (64 vs 1024 FLOP/clk/SIMD), so an fp32 product is cheaper as
// SIX bf16 products than as one fp32 one.  Every fp32 operand is written as a sum of three bf16 numbers,
//       a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)          (round to nearest even; 3 x 8 = 24 bits)
// and a.w is accumulated from the six products of weight >= 2^-16,
//       a1.w1                                  -> accumulator `hi`
//       a1.w2 + a2.w1 + a2.w2 + a1.w3 + a3.w1  -> accumulator `lo`            (the dropped a2.w3 + a3.w2 + a3.w3 are <= 2^-24 |a.w|)
// with v_mfma_f32_32x32x16_bf16: bf16 x bf16 products are exact in fp32 and the sums are fp32.  `hi` takes K / 16 rounded additions
// (the f32-input MFMA chain takes K / 4 or K / 2), `lo` carries terms 2^-8 smaller, and C = hi + lo at the end.  Measured against
// float64 the result is as accurate as the f32-input MFMA kernels or better (tests/test_gpu_kernels.py::test_gemm_split_*; on the
// host model of both, rms error 1.3e-7 vs 2.5e-7 of the result scale at K = 512): this is fp32 arithmetic re-associated, not a
// reduced-precision mode — there is no bf16 rounding of any operand or result anywhere.
//
// Non-finite operands: an infinite a gives a2 = bf16(inf - inf) = NaN, so the result is NaN where the f32-input MFMA would give +-inf (both are
// failures upstream: every operand on this path is finite); fp32 subnormals are below bf16's three-plane range and count as zero.
//
// Weights are split ONCE (engine prepare: split_bf16x3) into three planes [3][N][ldw] (plane stride p.wplane elements);
// activations are split in registers on their way into LDS (11 VALU ops per pair of elements, beside the MFMAs).  The folded
// RMSNorm's 1/rms is accumulated from the fp32 registers in the canonical order of the other families (one running sum per 16-byte
// chunk position, fused a0^2+a1^2+a2^2+a3^2, tree ((0+1)+(2+3))+((4+5)+(6+7))): bit-identical row scales.  Same epilogues as
// gemm_bf16_kernel.  Every tile configuration walks k in the same order with the same instruction: identical bits.
//
// Staging: global -> registers -> LDS, double-buffered, k-tiles of 32; LDS rows are [32 + 8] bf16 per plane (conflict-free
// ds_read_b128 fragments, as gemm_bf16.hip).
#include "common.h"
#include <hip/hip_ext.h>
#include "kernels.h"
#include <stdlib.h>
#include <map>
#include <tuple>
#include <type_traits>

namespace d4 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float a, __bf16& h1, __bf16& h2, __bf16& h3) {
    h1 = (__bf16)a;
    const float r = a - (float)h1;
    h2 = (__bf16)r;
    h3 = (__bf16)(r - (float)h2);
}

template <int BM, int BN, int WGM, int WGN, int D, int NBUF, int OCC, bool STAG>
__global__ __launch_bounds__(WGM* WGN * 64, OCC) void gemm_x3_kernel(GemmArgs p) {      // OCC: waves per SIMD the register budget must allow
    static_assert(!STAG || (NBUF == 2 && WGM * WGN == 8), "staggered form: 8 waves, two LDS buffers");
    static_assert(D >= 1 && D <= 3 && (NBUF == 1 || NBUF == 2), "register staging sets / LDS buffers");
    constexpr int BK = 32, LDS_LD = BK + 8, NT = WGM * WGN * 64;
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int G = BK / 8;                       // 8-element groups per tile row
    constexpr int A_G = BM * G / NT, B_G = BN * G / NT;
    static_assert(TM >= 1 && TN >= 1 && A_G >= 1 && B_G >= 1, "tile");
    constexpr int APL = BM * LDS_LD, BPL = BN * LDS_LD;      // one plane of one buffer (elements)

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __bf16* As = reinterpret_cast<__bf16*>(smem_raw);                 // [NBUF][3][BM][LDS_LD]
    __bf16* Bs = As + NBUF * 3 * APL;                                 // [NBUF][3][BN][LDS_LD]
    float* rowscale_s = reinterpret_cast<float*>(Bs + NBUF * 3 * BPL);   // [BM]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    int bid = blockIdx.x;                           // XCD-aware order: consecutive blocks on one XCD share an A row-panel
    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    {
        const int nblk = nbm * nbn, nx = 8;
        const int q = nblk / nx, r = nblk % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    }
    // within an XCD's share the tiles are walked column-major inside bands of RB row-panels: the ~32 blocks in flight on an XCD then
    // cover RB A row-panels x ~8 W column tiles (RB x BM x K x 4 + 8 x BN x K x 6 bytes: fits the 4 MB L2) instead of one row-panel x every
    // column tile (all three W planes streamed from the MALL once per row-panel: measured 282 MB of fabric reads per FF1 launch, 18x the operands)
    int tm, tn;
    {
        constexpr int RB = 4;
        const int band = bid / (RB * nbn), j = bid % (RB * nbn);
        const int rows = min(RB, nbm - band * RB);
        tm = band * RB + j % rows; tn = j / rows;
    }
    const int bm0 = tm * BM, bn0 = tn * BN;
    const int bz = blockIdx.y;
    const __bf16* Wb = reinterpret_cast<const __bf16*>(p.Wb) + bz * p.strideW;
    p.A += bz * p.strideA; p.C += bz * p.strideC;
    if (p.R) p.R += bz * p.strideC;

    const int rowsA = min(BM, p.M - bm0), rowsB = min(BN, p.N - bn0);
    auto uniform_rsrc = [](const void* base, int64_t bytes) {
        const uint64_t b = reinterpret_cast<uint64_t>(base);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        const int nb = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7FFFFFFF ? bytes : 0x7FFFFFFF));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
    };
    // rows past the matrix edge fall outside num_records and read as zeros
    const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc(p.A + (int64_t)bm0 * p.lda, ((int64_t)(rowsA - 1) * p.lda + p.K) * 4);
    const int64_t wbytes = ((int64_t)(rowsB - 1) * p.ldw + p.K) * 2;
    const __amdgpu_buffer_rsrc_t rsB0 = uniform_rsrc(Wb + (int64_t)bn0 * p.ldw, wbytes);
    const __amdgpu_buffer_rsrc_t rsB1 = uniform_rsrc(Wb + p.wplane + (int64_t)bn0 * p.ldw, wbytes);
    const __amdgpu_buffer_rsrc_t rsB2 = uniform_rsrc(Wb + 2 * p.wplane + (int64_t)bn0 * p.ldw, wbytes);

    // D register staging sets: the loads of k-tile kt + D are issued while k-tile kt is multiplied (an L2 hit takes ~1.5 us under load,
    // two to three k-tile times).  The loads are UNCONDITIONAL (the k offset is clamped to the last tile) so that the counted
    // s_waitcnt vmcnt before a set is consumed is exact: a load skipped on one path makes the compiler wait for the newest set.
    f32x4 ra[D][A_G][2];
    f32x4 rb[D][B_G][3];                            // 8 bf16 of each plane as 16 raw bytes
    float ssq[A_G][2];
#pragma unroll
    for (int i = 0; i < A_G; ++i) ssq[i][0] = ssq[i][1] = 0.f;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;

    auto load_tile = [&](auto set_tag, int k0) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < A_G; ++i) {
            const int idx = tid + i * NT, r = idx / G, c = (idx % G) * 8;
            const uint32_t off = (uint32_t)((r * p.lda + k0 + c) * 4);
            ra[S][i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
            ra[S][i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off + 16, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < B_G; ++i) {
            const int idx = tid + i * NT, r = idx / G, c = (idx % G) * 8;
            const uint32_t off = (uint32_t)((r * p.ldw + k0 + c) * 2);
            rb[S][i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB0, off, 0, 0));
            rb[S][i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB1, off, 0, 0));
            rb[S][i][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB2, off, 0, 0));
        }
    };
    auto store_tile = [&](auto set_tag, int buf) {
        constexpr int S = decltype(set_tag)::value;
        __bf16* as = As + buf * 3 * APL;
        __bf16* bs = Bs + buf * 3 * BPL;
#pragma unroll
        for (int i = 0; i < B_G; ++i) {
            const int idx = tid + i * NT, r = idx / G, c = (idx % G) * 8;
            *reinterpret_cast<f32x4*>(bs + r * LDS_LD + c) = rb[S][i][0];
            *reinterpret_cast<f32x4*>(bs + BPL + r * LDS_LD + c) = rb[S][i][1];
            *reinterpret_cast<f32x4*>(bs + 2 * BPL + r * LDS_LD + c) = rb[S][i][2];
        }
#pragma unroll
        for (int i = 0; i < A_G; ++i) {
            const int idx = tid + i * NT, r = idx / G, c = (idx % G) * 8;
            const f32x4 v0 = ra[S][i][0], v1 = ra[S][i][1];
            bf16x8 o1, o2, o3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 h1, h2, h3;
                split3(v0[e], h1, h2, h3); o1[e] = h1; o2[e] = h2; o3[e] = h3;
                split3(v1[e], h1, h2, h3); o1[e + 4] = h1; o2[e + 4] = h2; o3[e + 4] = h3;
            }
            *reinterpret_cast<bf16x8*>(as + r * LDS_LD + c) = o1;
            *reinterpret_cast<bf16x8*>(as + APL + r * LDS_LD + c) = o2;
            *reinterpret_cast<bf16x8*>(as + 2 * APL + r * LDS_LD + c) = o3;
            ssq[i][0] = ssq[i][0] + __builtin_fmaf(v0[3], v0[3], __builtin_fmaf(v0[2], v0[2], __builtin_fmaf(v0[1], v0[1], v0[0] * v0[0])));
            ssq[i][1] = ssq[i][1] + __builtin_fmaf(v1[3], v1[3], __builtin_fmaf(v1[2], v1[2], __builtin_fmaf(v1[1], v1[1], v1[0] * v1[0])));
        }
    };

    f32x16 hi[TM][TN], lo[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { hi[i][j][e] = 0.f; lo[i][j][e] = 0.f; }

    const int lrow = lane & 31, lhalf = lane >> 5;
    // one 16-k step of the current LDS buffer: small terms first into `lo` (a3.w1, a2.w2, a1.w3, then a2.w1, a1.w2), the leading term into
    // `hi`; the (i, j) loops are innermost so that neighbouring MFMAs are independent
    auto mma = [&](int buf, int ks) {
        const __bf16* as = As + buf * 3 * APL + (wm * TM * 32 + lrow) * LDS_LD + lhalf * 8 + ks * 16;
        const __bf16* bs = Bs + buf * 3 * BPL + (wn * TN * 32 + lrow) * LDS_LD + lhalf * 8 + ks * 16;
        bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(as + pl * APL + i * 32 * LDS_LD);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(bs + pl * BPL + j * 32 * LDS_LD);
        }
#define D4_X3_TERM(PA, PB, ACC)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                           \
        ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA][i], bf[PB][j], ACC[i][j], 0, 0, 0);
        D4_X3_TERM(2, 0, lo)
        D4_X3_TERM(0, 0, hi)
        D4_X3_TERM(1, 1, lo)
        D4_X3_TERM(0, 2, lo)
        D4_X3_TERM(1, 0, lo)
        D4_X3_TERM(0, 1, lo)
#undef D4_X3_TERM
    };

    const int nk = p.K / BK;
    const int klast = (nk - 1) * BK;
    load_tile(S0{}, 0);
    if constexpr (D >= 2) load_tile(S1{}, min(BK, klast));
    if constexpr (D == 3) load_tile(S2{}, min(2 * BK, klast));
    store_tile(S0{}, 0);
    __syncthreads();
    // k-tile kt came through register set kt % D.  NBUF = 2: it lives in LDS buffer kt & 1, the split + LDS store of k-tile kt + 1 sits
    // between the two MFMA groups of k-tile kt (VALU / LDS work beside the matrix pipe), one barrier per k-tile.  NBUF = 1 (half the LDS:
    // two or three co-resident blocks per CU, whose phases interleave on the matrix pipe): multiply, barrier, store the next tile, barrier.
    // STAG (8 waves = two per SIMD): the block's waves form two groups (waves 0-3 / 4-7, one wave of each on every SIMD) that run half a
    // k-tile out of phase — while one group multiplies k-tile kt (24 MFMAs per wave), the other splits and stores its half of k-tile
    // kt + 1; then they swap.  Without this both waves of a SIMD reach the barrier together, queue on the matrix pipe together and
    // leave it idle together (measured: 26 % MFMA busy).  Two barriers per k-tile.
    const bool grp_b = STAG && wave >= 4;
    auto k_tile = [&](int kt, auto set_tag, auto store_tag) {
        constexpr int S = decltype(set_tag)::value;
        const int buf = NBUF == 2 ? (kt & 1) : 0;
        load_tile(set_tag, min((kt + D) * BK, klast));
        const bool store = decltype(store_tag)::value || kt + 1 < nk;
        if constexpr (STAG) {
            if (!grp_b) { mma(buf, 0); mma(buf, 1); }
            else if (store) store_tile(std::integral_constant<int, (S + 1) % D>{}, buf ^ 1);
            __syncthreads();
            if (grp_b) { mma(buf, 0); mma(buf, 1); }
            else if (store) store_tile(std::integral_constant<int, (S + 1) % D>{}, buf ^ 1);
        } else if constexpr (NBUF == 2) {
            mma(buf, 0);
            if (store) store_tile(std::integral_constant<int, (S + 1) % D>{}, buf ^ 1);
            mma(buf, 1);
            if constexpr (decltype(store_tag)::value) {
                // ask the scheduler for one interleaved stream: behind every MFMA a few of the split's VALU ops and one LDS access (left to
                // itself hipcc emits MFMA group | split + stores | MFMA group, and the two waves of a SIMD stall on the same phase together)
                constexpr int NMFMA = 2 * 6 * TM * TN;
#pragma unroll
                for (int i = 0; i < NMFMA; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);     // 4 VALU
                    __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);     // 1 DS
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // 1 VMEM read
                }
            }
        } else {
            mma(buf, 0);
            mma(buf, 1);
            __syncthreads();
            if (store) store_tile(std::integral_constant<int, (S + 1) % D>{}, 0);
        }
        __syncthreads();
    };
    // full trips of D k-tiles, none of them the last k-tile: straight-line code (no branch inside the trip — with one, the compiler's
    // s_waitcnt pass merges the paths conservatively and drains every load at the loop head); then the 1 .. D remaining k-tiles
    using Always = std::true_type;
    using Check = std::false_type;
    const int nfull = (nk - 1) / D;
    int kt = 0;
    for (int t = 0; t < nfull; ++t, kt += D) {
        k_tile(kt, S0{}, Always{});
        if constexpr (D >= 2) k_tile(kt + 1, S1{}, Always{});
        if constexpr (D == 3) k_tile(kt + 2, S2{}, Always{});
    }
    k_tile(kt, S0{}, Check{});
    if constexpr (D >= 2) { if (kt + 1 < nk) k_tile(kt + 1, S1{}, Check{}); }
    if constexpr (D == 3) { if (kt + 2 < nk) k_tile(kt + 2, S2{}, Check{}); }

    if (p.flags & GEMM_RMS_ROWSCALE) {
#pragma unroll
        for (int i = 0; i < A_G; ++i) {
            float s = ssq[i][0] + ssq[i][1];           // chunks (2g) + (2g + 1)
            s += dpp_f<0xB1>(s);                       // ((0+1)+(2+3)), ((4+5)+(6+7))
            s += dpp_f<0x4E>(s);                       // the four lanes of a row
            const int idx = tid + i * NT;
            if ((idx % G) == 0) rowscale_s[idx / G] = rsqrtf(s / (float)p.K + p.rms_eps);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int lr = wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
            const int gm = bm0 + lr;
            if (gm >= p.M) continue;
            const float rs = (p.flags & GEMM_RMS_ROWSCALE) ? rowscale_s[lr] : 1.f;
            if (swiglu) {
                if constexpr (TN % 2 == 0) {
#pragma unroll
                    for (int j = 0; j < TN; j += 2) {
                        const int gn = bn0 + wn * TN * 32 + j * 32 + lrow;       // packed column of the value
                        if (gn >= p.N) continue;
                        float val = (hi[i][j][e] + lo[i][j][e]) * rs, gate = (hi[i][j + 1][e] + lo[i][j + 1][e]) * rs;
                        if (p.bias) { val += p.bias[gn]; gate += p.bias[gn + 32]; }
                        const int on = (gn / 64) * 32 + (gn % 64);
                        p.C[(int64_t)gm * p.ldc + on] = val * siluf(gate);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int gn = bn0 + wn * TN * 32 + j * 32 + lrow;
                    if (gn >= p.N) continue;
                    float v = (hi[i][j][e] + lo[i][j][e]) * rs;
                    if (p.bias) v += p.bias[gn];
                    if (p.flags & GEMM_SILU) v = siluf(v);
                    if (p.R) v += p.R[(int64_t)gm * p.ldr + gn];
                    if (p.flags & GEMM_ACCUMULATE) v += p.C[(int64_t)gm * p.ldc + gn];
                    p.C[(int64_t)gm * p.ldc + gn] = v;
                    if (p.C2) {
                        const int ts = gm % p.c2_S;
                        const int keep = p.c2_hi - p.c2_lo;
                        const int rank = (ts >= p.c2_lo && ts < p.c2_hi) ? ts - p.c2_lo : ((p.c2_last && ts == p.c2_S - 1) ? keep : -1);
                        if (rank >= 0) p.C2[((int64_t)(gm / p.c2_S) * (keep + p.c2_last) + rank) * p.ldc2 + gn] = v;
                    }
                }
            }
        }
    }
}

// ---- fp32 -> three bf16 planes (engine prepare) -------------------------------------------------------------------
__global__ void split_bf16x3_kernel(const float* src, __bf16* dst, int64_t n, int64_t plane) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        __bf16 h1, h2, h3;
        split3(src[i], h1, h2, h3);
        dst[i] = h1; dst[plane + i] = h2; dst[2 * plane + i] = h3;
    }
}
int split_bf16x3(const float* src, uint16_t* dst, int64_t n, int64_t plane, hipStream_t s) {
    if (n == 0) return 0;
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)g), dim3(256), 0, s, src, reinterpret_cast<__bf16*>(dst), n, plane);
    D4_LAUNCH_CHECK();
    return 0;
}

// ---- configurations ---------------------------------------------------------------------------------------------
// name        waves   wave tile   register sets   LDS (buffers x 2 operands x 3 planes)   blocks (waves) / CU
// 64x64       2 x 2    32 x 32          3            1 x 30 KB                             3 (12)
// 128x64      2 x 2    64 x 32          2            1 x 45 KB                             2 (8)
// 64x128      2 x 2    32 x 64          2            1 x 45 KB                             2 (8)     SiLU-GLU capable
// 128x128     2 x 2    64 x 64          1            1 x 60 KB                             2 (8)     SiLU-GLU capable
// 128x128/8   4 x 2    32 x 64          3            2 x 60 KB                             1 (8)     SiLU-GLU capable
// 32x64       1 x 2    32 x 32          3            1 x 22 KB                             3+ (6+)   small GEMMs
enum { X3_64x64 = 0, X3_128x64, X3_64x128, X3_128x128, X3_128x128_8, X3_32x64, X3_N };
static const char* const kX3Name[X3_N] = {"gemm_x3_kernel<64, 64, 2, 2", "gemm_x3_kernel<128, 64, 2, 2", "gemm_x3_kernel<64, 128, 2, 2",
                                          "gemm_x3_kernel<128, 128, 2, 2", "gemm_x3_kernel<128, 128, 4, 2", "gemm_x3_kernel<32, 64, 1, 2"};
static const int kX3BM[X3_N] = {64, 128, 64, 128, 128, 32}, kX3BN[X3_N] = {64, 64, 128, 128, 128, 64};

int gemm_x3_configs() { return X3_N; }
const char* gemm_x3_config_name(int c) { return c >= 0 && c < X3_N ? kX3Name[c] : ""; }
void gemm_x3_config_tile(int c, int* bm, int* bn) { *bm = kX3BM[c]; *bn = kX3BN[c]; }

bool gemm_x3_applicable(const GemmArgs& p) {
    return p.Wb != nullptr && p.wplane > 0 && !(p.flags & (GEMM_TRANS_A | GEMM_TRANS_B)) && (p.K % 32) == 0 && (p.lda % 4) == 0 &&
           (p.ldw % 8) == 0 && ((uintptr_t)p.Wb % 16) == 0 && (p.strideW % 8) == 0 && (p.wplane % 8) == 0;
}

bool gemm_x3_config_valid(int c, const GemmArgs& p) {
    if (c < 0 || c >= X3_N || !gemm_x3_applicable(p)) return false;
    if (p.flags & GEMM_SWIGLU) return c == X3_64x128 || c == X3_128x128 || c == X3_128x128_8;     // the pairing needs two 32-column sub-tiles per wave
    return true;
}

template <int BM, int BN, int WGM, int WGN, int D, int NBUF, int OCC, bool STAG = false>
static int launch_x3(const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    constexpr int LDS_LD = 32 + 8;
    const size_t lds = (size_t)(NBUF * 3 * (BM + BN) * LDS_LD) * 2 + BM * sizeof(float);
    auto k = gemm_x3_kernel<BM, BN, WGM, WGN, D, NBUF, OCC, STAG>;
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.done();
    }
    const dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), p.batch > 0 ? p.batch : 1), block(WGM * WGN * 64);
    if (ea) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)lds, stream, ea, eb, 0, p);
    else hipLaunchKernelGGL(k, grid, block, lds, stream, p);
    D4_LAUNCH_CHECK();
    return 0;
}

int gemm_x3_launch(int c, const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    D4_REQUIRE(gemm_x3_config_valid(c, p), "gemm_x3: configuration %d is not valid for this call", c);
    switch (c) {
        case X3_64x64: return launch_x3<64, 64, 2, 2, 3, 1, 3>(p, stream, ea, eb);
        case X3_128x64: return launch_x3<128, 64, 2, 2, 2, 1, 2>(p, stream, ea, eb);
        case X3_64x128: return launch_x3<64, 128, 2, 2, 2, 1, 2>(p, stream, ea, eb);
        case X3_128x128: return launch_x3<128, 128, 2, 2, 1, 1, 2>(p, stream, ea, eb);
        case X3_128x128_8: return launch_x3<128, 128, 4, 2, 3, 2, 2, false>(p, stream, ea, eb);
        case X3_32x64: return launch_x3<32, 64, 1, 2, 2, 1, 3>(p, stream, ea, eb);
    }
    return 2;
}

// static choice by shape (the timed choice in gemm.hip refines it where a call can be repeated)
int gemm_x3_heuristic(const GemmArgs& p) {
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    const int nb = p.batch > 0 ? p.batch : 1;
    const int64_t t128 = (int64_t)cdiv(p.M, 128) * cdiv(p.N, 128) * nb;
    if (t128 >= 256 && p.N > 64) return X3_128x128_8;
    if (swiglu) return X3_64x128;
    if ((int64_t)cdiv(p.M, 64) * cdiv(p.N, 64) * nb >= 512) return X3_64x64;
    return p.N >= 64 ? X3_32x64 : X3_64x64;
}

}  // namespace d4
