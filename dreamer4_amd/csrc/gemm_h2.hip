// fp32-CLASS GEMM on the fp16 matrix cores by operand splitting with exact power-of-two scales ("fp16x2 operands, three products") — round 5, the
// engine's OPT-IN `matmul_bf16 = 3` mode (Python: matmul_dtype='fp32_fp16x2').  Against the default bf16x3 / six-product scheme of gemm_x3.hip: half the
// MFMAs, two thirds of the planes through LDS (tools/micro/x3_planes_probe.hip, profiles/r05b_x3_planes_probe.txt: x1.3-1.65 on every cfg-2 shape in
// the same kernel body; whole rollout 176.5 -> 152.0 ms).  NOT the default and NOT what bench.py's `dtype: f32` is measured on: see "What this is".
//
//   C[m, n] = epilogue( rowscale[m] * sum_k A[m, k] * W[n, k] )        A fp32 [M][K], W fp32 given as two fp16 planes + a scale per row, C fp32
//
// Every fp32 operand row gets an exact power-of-two scale s that puts its largest magnitude into [2^14, 2^15) (the top of fp16's range), and each
// scaled element x = a s is written as
//       x = hi + lo 2^-11 + r,    hi = fp16(x),  lo = fp16((x - hi) 2^11),   |r| <= 2^-23 |x|     (x - hi and the 2^11 are exact in fp32)
// A product is accumulated from its three leading fp16 x fp16 terms — each EXACT in fp32 (11 x 11 significant bits) — with v_mfma_f32_32x32x16_f16:
//       hi_a.hi_w                 -> accumulator `hi`        (K / 16 rounded additions; the f32-input MFMA chain takes K)
//       hi_a.lo_w + lo_a.hi_w     -> accumulator `lo`        (terms 2^-11 smaller; the dropped lo_a.lo_w is up to 2^-22 of the product)
// and C = ldexp(hi + lo 2^-11, -(ea + ew)) undoes the two scales by ONE exact exponent shift (never a product of two floats: no intermediate
// overflow).
//
// What this is: fp32 arithmetic re-associated, carried on 23-bit operand images.  On dot products its error against float64 is BELOW the
// f32-input MFMA's for every K (the fmaf chain of K rounded additions dominates that one: 0.43-0.45x at K = 512 on the GPU) and non-finite operands
// poison exactly the outputs that depend on them — two of the three criteria the bf16x3 scheme is held to (tests/test_gpu_kernels.py::test_gemm_h2_*).
// The third it FAILS: with operands spanning 2^120 inside a row every output is one product, and one product carries up to 2^-21 relative error where
// fp32 has 2^-24: 5.5e-7 of sum |a w| against 3.1e-7 for the f32-input MFMA (allowed: 1.25 x + 6e-8).  Completing the product (lo.lo) or a third
// plane of W does not fix it (5.0e-7, 4.95e-7: the fp16 MFMA's own accumulation is the rest): profiles/r05_x3_products.txt.
//
// Range: an element more than 2^28 below its row's largest keeps fewer bits (fp16 subnormal hi, lo), one more than ~2^38 below counts as zero:
// its absolute error is <= 2^-50 of the row maximum — invisible against sum_k |a_k w_k| unless the OTHER operand is >= 2^26 above its own
// row's typical size at exactly that k (the wide-exponent test measures that case).
// Non-finite operands: a row (column) holding an inf / NaN gets scale 1, the element becomes hi = +-inf / NaN, lo = NaN, and every output of
// that row (column) is non-finite; other outputs never see it.  fp32 subnormal rows are scaled by 2^126 and keep their leading bits.
//
// The row maximum of A: each workgroup computes it for ITS rows in a prologue pass over the whole A panel (BM x K floats, coalesced 1 KB wave
// loads, L2 hits: the same panel is read by every column tile) unless the caller hands in per-row exponents (p.aexp: a producer that already
// knows them).  The scale of a row is a function of the row alone — every tile configuration gives the same bits.
// W is split ONCE at engine prepare (split_f16x2_rows: one scale per weight row = output column).
//
// Staging: as gemm_x3.hip — global -> registers -> LDS, k-tiles of 32, LDS rows [32 + 8] fp16 per plane (conflict-free ds_read_b128), register
// staging sets D deep with counted vmcnt; folded RMSNorm row sums from the fp32 registers in the canonical order of the other families.
#include "common.h"
#include <hip/hip_ext.h>
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace d4 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// scale exponent of a row whose largest magnitude has the fp32 bit pattern `mx` (sign cleared): e such that max * 2^e lies in [2^14, 2^15).
// inf / NaN rows: 0.  The result is clamped so that 2^e is a normal float (subnormal / zero rows: 2^126).
__host__ __device__ __forceinline__ int h2_scale_exp(uint32_t mx) {
    const int be = (int)(mx >> 23);             // biased exponent of the maximum
    if (be == 255) return 0;
    int e = 14 - (be - 127);
    return e > 126 ? 126 : e;                   // (be >= 1 -> e <= 140 -> clamp; be = 254 -> e = -113: normal)
}
__device__ __forceinline__ float h2_pow2(int e) { return __builtin_bit_cast(float, (uint32_t)(e + 127) << 23); }      // -126 <= e <= 127

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    auto dpp = [](uint32_t x, auto ctrl) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, decltype(ctrl)::value, 0xF, 0xF, false);
    };
    using C1 = std::integral_constant<int, 0xB1>; using C2 = std::integral_constant<int, 0x4E>;
    using C3 = std::integral_constant<int, 0x141>; using C4 = std::integral_constant<int, 0x140>;
    uint32_t t;
    t = dpp(v, C1{}); v = v > t ? v : t;
    t = dpp(v, C2{}); v = v > t ? v : t;
    t = dpp(v, C3{}); v = v > t ? v : t;
    t = dpp(v, C4{}); v = v > t ? v : t;
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

template <int BM, int BN, int WGM, int WGN, int D, int NBUF, int OCC>
__global__ __launch_bounds__(WGM* WGN * 64, OCC) void gemm_h2_kernel(GemmArgs p) {      // OCC: waves per SIMD the register budget must allow
    static_assert(D >= 1 && D <= 3 && (NBUF == 1 || NBUF == 2), "register staging sets / LDS buffers");
    constexpr int BK = 32, LDS_LD = BK + 8, NT = WGM * WGN * 64, NW = WGM * WGN;
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int G = BK / 8;                       // 8-element groups per tile row
    constexpr int A_G = BM * G / NT, B_G = BN * G / NT;
    static_assert(TM >= 1 && TN >= 1 && A_G >= 1 && B_G >= 1, "tile");
    constexpr int APL = BM * LDS_LD, BPL = BN * LDS_LD;      // one plane of one buffer (elements)

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    _Float16* As = reinterpret_cast<_Float16*>(smem_raw);             // [NBUF][2][BM][LDS_LD]
    _Float16* Bs = As + NBUF * 2 * APL;                               // [NBUF][2][BN][LDS_LD]
    float* rowscale_s = reinterpret_cast<float*>(Bs + NBUF * 2 * BPL);   // [BM] folded RMSNorm 1 / rms
    int* aexp_s = reinterpret_cast<int*>(rowscale_s + BM);               // [BM] scale exponent of each A row

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
    // within an XCD's share the tiles are walked column-major inside bands of RB row-panels (as gemm_x3.hip: the ~32 blocks in flight on an
    // XCD cover RB A row-panels x ~8 W column tiles, which its 4 MB L2 holds)
    int tm, tn;
    {
        constexpr int RB = 4;
        const int band = bid / (RB * nbn), j = bid % (RB * nbn);
        const int rows = min(RB, nbm - band * RB);
        tm = band * RB + j % rows; tn = j / rows;
    }
    const int bm0 = tm * BM, bn0 = tn * BN;
    const int bz = blockIdx.y;
    const _Float16* Wb = reinterpret_cast<const _Float16*>(p.Wb) + bz * p.strideW;
    const float* wscale = p.wscale + (int64_t)bz * p.strideWs;
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

    // ---- prologue: the scale exponent of each of this tile's A rows.  A wave takes rows wave, wave + NW, ...; a row is read as 1 KB wave loads
    // (lane l: floats 4 l .. 4 l + 3 of every 256-float chunk), up to sixteen rows' loads in flight together; integer maximum of the sign-cleared
    // bit patterns (a NaN outranks inf outranks every finite number).
    if (p.aexp) {
        for (int r = tid; r < BM; r += NT) aexp_s[r] = (bm0 + r < p.M) ? p.aexp[(int64_t)bz * p.M + bm0 + r] : 0;
    } else {
        constexpr int RPW = BM / NW;                 // rows per wave
        constexpr int RU = RPW >= 16 ? 16 : RPW;     // rows in flight per wave: ONE memory round trip for a 64-row tile on four waves (a load inside a
                                                     // dependent chain costs a wave ~1-2 us under load; four batches of four rows measured +15 us on a 50 us launch)
        static_assert(RPW * NW == BM && RPW % RU == 0, "row split");
        const int nchunk = (p.K + 255) / 256;
        for (int r0 = 0; r0 < RPW; r0 += RU) {
            uint32_t mx[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) mx[u] = 0u;
            for (int ch = 0; ch < nchunk; ++ch) {
                const int k = ch * 256 + lane * 4;
                u32x4 v[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int r = wave + (r0 + u) * NW;
                    // past the row's end (k >= K) the offset is pushed outside num_records: reads as zero
                    const uint32_t off = k < p.K ? (uint32_t)((r * p.lda + k) * 4) : 0xFFFFFFF0u;
                    v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
                }
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const u32x4 w = v[u] & 0x7FFFFFFFu;
                    const uint32_t m01 = w[0] > w[1] ? w[0] : w[1], m23 = w[2] > w[3] ? w[2] : w[3];
                    const uint32_t m = m01 > m23 ? m01 : m23;
                    mx[u] = mx[u] > m ? mx[u] : m;
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const uint32_t m = wave_max_u32(mx[u]);
                if (lane == 0) aexp_s[wave + (r0 + u) * NW] = h2_scale_exp(m);
            }
        }
    }
    __syncthreads();

    // D register staging sets: the loads of k-tile kt + D are issued while k-tile kt is multiplied.  The loads are UNCONDITIONAL (the k offset
    // is clamped to the last tile) so that the counted s_waitcnt vmcnt before a set is consumed is exact.
    f32x4 ra[D][A_G][2];
    f32x4 rb[D][B_G][2];                            // 8 fp16 of each plane as 16 raw bytes
    float ssq[A_G][2], sa[A_G];
#pragma unroll
    for (int i = 0; i < A_G; ++i) {
        ssq[i][0] = ssq[i][1] = 0.f;
        sa[i] = h2_pow2(aexp_s[(tid + i * NT) / G]);       // a thread stages the same rows in every k-tile
    }
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
        }
    };
    auto store_tile = [&](auto set_tag, int buf) {
        constexpr int S = decltype(set_tag)::value;
        _Float16* as = As + buf * 2 * APL;
        _Float16* bs = Bs + buf * 2 * BPL;
#pragma unroll
        for (int i = 0; i < B_G; ++i) {
            const int idx = tid + i * NT, r = idx / G, c = (idx % G) * 8;
            *reinterpret_cast<f32x4*>(bs + r * LDS_LD + c) = rb[S][i][0];
            *reinterpret_cast<f32x4*>(bs + BPL + r * LDS_LD + c) = rb[S][i][1];
        }
#pragma unroll
        for (int i = 0; i < A_G; ++i) {
            const int idx = tid + i * NT, r = idx / G, c = (idx % G) * 8;
            const f32x4 v0 = ra[S][i][0], v1 = ra[S][i][1];
            const float s = sa[i];
            f16x8 o1, o2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = v0[e] * s;
                _Float16 h = (_Float16)x; o1[e] = h; o2[e] = (_Float16)((x - (float)h) * 2048.f);
                x = v1[e] * s;
                h = (_Float16)x; o1[e + 4] = h; o2[e + 4] = (_Float16)((x - (float)h) * 2048.f);
            }
            *reinterpret_cast<f16x8*>(as + r * LDS_LD + c) = o1;
            *reinterpret_cast<f16x8*>(as + APL + r * LDS_LD + c) = o2;
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
    // one 16-k step of the current LDS buffer: a cross term into `lo`, the leading term into `hi`, the other cross term into `lo`; the (i, j)
    // loops are innermost so that neighbouring MFMAs are independent
    auto mma = [&](int buf, int ks) {
        const _Float16* as = As + buf * 2 * APL + (wm * TM * 32 + lrow) * LDS_LD + lhalf * 8 + ks * 16;
        const _Float16* bs = Bs + buf * 2 * BPL + (wn * TN * 32 + lrow) * LDS_LD + lhalf * 8 + ks * 16;
        f16x8 af[2][TM], bf[2][TN];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[pl][i] = *reinterpret_cast<const f16x8*>(as + pl * APL + i * 32 * LDS_LD);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const f16x8*>(bs + pl * BPL + j * 32 * LDS_LD);
        }
#define D4_H2_TERM(PA, PB, ACC)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                           \
        ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[PA][i], bf[PB][j], ACC[i][j], 0, 0, 0);
        D4_H2_TERM(1, 0, lo)
        D4_H2_TERM(0, 0, hi)
        D4_H2_TERM(0, 1, lo)
#undef D4_H2_TERM
    };

    const int nk = p.K / BK;
    const int klast = (nk - 1) * BK;
    load_tile(S0{}, 0);
    if constexpr (D >= 2) load_tile(S1{}, min(BK, klast));
    if constexpr (D == 3) load_tile(S2{}, min(2 * BK, klast));
    store_tile(S0{}, 0);
    __syncthreads();
    // k-tile kt came through register set kt % D.  NBUF = 2: it lives in LDS buffer kt & 1, the split + LDS store of k-tile kt + 1 sits
    // between the two MFMA groups of k-tile kt, one barrier per k-tile.  NBUF = 1 (half the LDS: more co-resident blocks per CU, whose phases
    // interleave on the matrix pipe): multiply, barrier, store the next tile, barrier.
    auto k_tile = [&](int kt, auto set_tag, auto store_tag) {
        constexpr int S = decltype(set_tag)::value;
        const int buf = NBUF == 2 ? (kt & 1) : 0;
        load_tile(set_tag, min((kt + D) * BK, klast));
        const bool store = decltype(store_tag)::value || kt + 1 < nk;
        if constexpr (NBUF == 2) {
            mma(buf, 0);
            if (store) store_tile(std::integral_constant<int, (S + 1) % D>{}, buf ^ 1);
            mma(buf, 1);
            if constexpr (decltype(store_tag)::value) {
                constexpr int NMFMA = 2 * 3 * TM * TN;
#pragma unroll
                for (int i = 0; i < NMFMA; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);     // 6 VALU
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
    // full trips of D k-tiles, none of them the last k-tile: straight-line code; then the 1 .. D remaining k-tiles
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

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // value = ldexp(hi + lo 2^-11, -(ea + ew)): ONE exact exponent shift undoes both scales
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    auto wexp = [&](int gn) { return (int)((__builtin_bit_cast(uint32_t, wscale[gn]) >> 23) & 0xFFu) - 127; };    // exponent of the stored 2^-ew
    auto comb = [&](int i, int j, int e) { return __builtin_fmaf(lo[i][j][e], 1.f / 2048.f, hi[i][j][e]); };
    int we[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gn = bn0 + wn * TN * 32 + j * 32 + lrow;
        we[j] = gn < p.N ? wexp(gn) : 0;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int lr = wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
            const int gm = bm0 + lr;
            if (gm >= p.M) continue;
            const float rs = (p.flags & GEMM_RMS_ROWSCALE) ? rowscale_s[lr] : 1.f;
            const int ae = -aexp_s[lr];
            if (swiglu) {
                if constexpr (TN % 2 == 0) {
#pragma unroll
                    for (int j = 0; j < TN; j += 2) {
                        const int gn = bn0 + wn * TN * 32 + j * 32 + lrow;       // packed column of the value
                        if (gn >= p.N) continue;
                        float val = ldexpf(comb(i, j, e), ae + we[j]) * rs;
                        float gate = ldexpf(comb(i, j + 1, e), ae + we[j + 1]) * rs;
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
                    float v = ldexpf(comb(i, j, e), ae + we[j]) * rs;
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

// ---- fp32 weight rows -> two fp16 planes + the inverse scale of each row (engine prepare; one workgroup per row) ----------------------------
// dst[r * ld + c] = hi, dst[plane + r * ld + c] = lo of src[r * ld + c] * 2^e_r;  inv_scale[r] = 2^-e_r.  Columns cols .. ld - 1 (row padding) are copied
// through the same arithmetic (they are zeros or ignored: the GEMM's K never reaches them) but do not enter the row maximum.
__global__ __launch_bounds__(256) void split_f16x2_rows_kernel(const float* src, _Float16* dst, int cols, int ld, int64_t plane, float* inv_scale) {
    __shared__ uint32_t red[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* s = src + (int64_t)r * ld;
    uint32_t mx = 0u;
    for (int c = tid; c < cols; c += 256) {
        const uint32_t b = __builtin_bit_cast(uint32_t, s[c]) & 0x7FFFFFFFu;
        mx = mx > b ? mx : b;
    }
    mx = wave_max_u32(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    const uint32_t a = red[0] > red[1] ? red[0] : red[1], b = red[2] > red[3] ? red[2] : red[3];
    const int e = h2_scale_exp(a > b ? a : b);
    const float sc = h2_pow2(e);
    if (tid == 0) inv_scale[r] = h2_pow2(-e);
    _Float16* d = dst + (int64_t)r * ld;
    for (int c = tid; c < ld; c += 256) {
        const float x = s[c] * sc;
        const _Float16 h = (_Float16)x;
        d[c] = h;
        d[plane + c] = (_Float16)((x - (float)h) * 2048.f);
    }
}
int split_f16x2_rows(const float* src, uint16_t* dst, int rows, int cols, int ld, int64_t plane, float* inv_scale, hipStream_t s) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(split_f16x2_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, src, reinterpret_cast<_Float16*>(dst), cols, ld, plane, inv_scale);
    D4_LAUNCH_CHECK();
    return 0;
}

// ---- the scale exponents of the rows of an activation matrix, ONCE for all the GEMMs (and all their column tiles) that read it: a wave per row,
// 1 KB wave loads, integer maximum of the sign-cleared bit patterns.  out[r] = h2_scale_exp(max |A[r][0 .. K)|).
__global__ __launch_bounds__(256) void row_scale_exp_kernel(const float* A, int64_t lda, int rows, int K, int* out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* a = A + (int64_t)r * lda;
    uint32_t mx = 0u;
    for (int k = lane * 4; k < K; k += 256) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(a + k) & 0x7FFFFFFFu;
        const uint32_t m01 = w[0] > w[1] ? w[0] : w[1], m23 = w[2] > w[3] ? w[2] : w[3];
        const uint32_t m = m01 > m23 ? m01 : m23;
        mx = mx > m ? mx : m;
    }
    mx = wave_max_u32(mx);
    if (lane == 0) out[r] = h2_scale_exp(mx);
}
int row_scale_exp(const float* A, int64_t lda, int rows, int K, int* out, hipStream_t s) {
    if (rows == 0) return 0;
    D4_REQUIRE((K % 4) == 0 && (lda % 4) == 0 && ((uintptr_t)A % 16) == 0, "row_scale_exp: K and lda must be multiples of 4, A 16-byte aligned");
    hipLaunchKernelGGL(row_scale_exp_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, s, A, lda, rows, K, out);
    D4_LAUNCH_CHECK();
    return 0;
}

// ---- configurations ---------------------------------------------------------------------------------------------
// name        waves   wave tile   register sets   LDS (buffers x 2 operands x 2 planes)   blocks (waves) / CU
// 64x64       2 x 2    32 x 32          3            1 x 20 KB                             3 (12)
// 128x64      2 x 2    64 x 32          2            1 x 30 KB                             2 (8)
// 64x128      2 x 2    32 x 64          2            1 x 30 KB                             2-3 (8-12)   SiLU-GLU capable
// 128x128     2 x 2    64 x 64          1            1 x 40 KB                             2 (8)        SiLU-GLU capable
// 128x128/8   4 x 2    32 x 64          3            2 x 40 KB                             1 (8)        SiLU-GLU capable
// 32x64       1 x 2    32 x 32          3            1 x 15 KB                             3+ (6+)      small GEMMs
// 64x128/o3   2 x 2    32 x 64          2            1 x 30 KB                             3 (12)       SiLU-GLU capable; 168 registers
enum { H2_64x64 = 0, H2_128x64, H2_64x128, H2_128x128, H2_128x128_8, H2_32x64, H2_64x128_O3, H2_N };
static const char* const kH2Name[H2_N] = {"gemm_h2_kernel<64, 64, 2, 2", "gemm_h2_kernel<128, 64, 2, 2", "gemm_h2_kernel<64, 128, 2, 2",
                                          "gemm_h2_kernel<128, 128, 2, 2", "gemm_h2_kernel<128, 128, 4, 2", "gemm_h2_kernel<32, 64, 1, 2", "gemm_h2_kernel<64, 128, 2, 2, 2, 1, 3"};
static const int kH2BM[H2_N] = {64, 128, 64, 128, 128, 32, 64}, kH2BN[H2_N] = {64, 64, 128, 128, 128, 64, 128};

int gemm_h2_configs() { return H2_N; }
const char* gemm_h2_config_name(int c) { return c >= 0 && c < H2_N ? kH2Name[c] : ""; }
void gemm_h2_config_tile(int c, int* bm, int* bn) { *bm = kH2BM[c]; *bn = kH2BN[c]; }

bool gemm_h2_applicable(const GemmArgs& p) {
    return p.Wb != nullptr && p.wplane > 0 && p.wscale != nullptr && !(p.flags & (GEMM_TRANS_A | GEMM_TRANS_B)) && (p.K % 32) == 0 && (p.lda % 4) == 0 &&
           (p.ldw % 8) == 0 && ((uintptr_t)p.Wb % 16) == 0 && (p.strideW % 8) == 0 && (p.wplane % 8) == 0 && ((uintptr_t)p.A % 16) == 0;
}

bool gemm_h2_config_valid(int c, const GemmArgs& p) {
    if (c < 0 || c >= H2_N || !gemm_h2_applicable(p)) return false;
    if (p.flags & GEMM_SWIGLU) return c == H2_64x128 || c == H2_128x128 || c == H2_128x128_8 || c == H2_64x128_O3;     // the pairing needs two 32-column sub-tiles per wave
    return true;
}

template <int BM, int BN, int WGM, int WGN, int D, int NBUF, int OCC>
static int launch_h2(const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    constexpr int LDS_LD = 32 + 8;
    const size_t lds = (size_t)(NBUF * 2 * (BM + BN) * LDS_LD) * 2 + BM * (sizeof(float) + sizeof(int));
    auto k = gemm_h2_kernel<BM, BN, WGM, WGN, D, NBUF, OCC>;
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

int gemm_h2_launch(int c, const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    D4_REQUIRE(gemm_h2_config_valid(c, p), "gemm_h2: configuration %d is not valid for this call", c);
    switch (c) {
        case H2_64x64: return launch_h2<64, 64, 2, 2, 3, 1, 3>(p, stream, ea, eb);
        case H2_128x64: return launch_h2<128, 64, 2, 2, 2, 1, 2>(p, stream, ea, eb);
        case H2_64x128: return launch_h2<64, 128, 2, 2, 2, 1, 2>(p, stream, ea, eb);
        case H2_128x128: return launch_h2<128, 128, 2, 2, 1, 1, 2>(p, stream, ea, eb);
        case H2_128x128_8: return launch_h2<128, 128, 4, 2, 3, 2, 2>(p, stream, ea, eb);
        case H2_32x64: return launch_h2<32, 64, 1, 2, 2, 1, 3>(p, stream, ea, eb);
        case H2_64x128_O3: return launch_h2<64, 128, 2, 2, 2, 1, 3>(p, stream, ea, eb);      // the same tile held to 168 registers: three blocks per CU
    }
    return 2;
}

// static choice by shape (the timed choice in gemm.hip refines it where a call can be repeated; every configuration gives the same bits)
int gemm_h2_heuristic(const GemmArgs& p) {
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    const int nb = p.batch > 0 ? p.batch : 1;
    if (swiglu || (int64_t)cdiv(p.M, 64) * cdiv(p.N, 128) * nb >= 256) return p.N >= 128 ? H2_64x128 : H2_64x64;
    if ((int64_t)cdiv(p.M, 64) * cdiv(p.N, 64) * nb >= 256) return H2_64x64;
    return p.N >= 64 ? H2_32x64 : H2_64x64;
}

}  // namespace d4
