// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
//   C[m, n] = epilogue( rowscale[m] * sum_k A[m, k] * W[n, k] )            ("NT": both K-contiguous)
//
// This one kernel family carries every dense projection of the imagination path
// (reference: Attention.to_q/k/v/out D4:1985-2068, FeedForward.proj_in/out D4:2105-2116,
// AttentionPool / LQAP projections D4:2143-2210, head MLPs D4:4950, 5095).  The reference runs
// RMSNorm -> Linear -> activation -> residual as separate ATen ops; here
//   * RMSNorm is folded: gamma is pre-multiplied into W (engine prepare), and the per-row
//     1/rms is accumulated from the A tiles as they stream through LDS and applied in the epilogue,
//   * bias, SiLU, SiLU-GLU pairing and the residual add run in the epilogue.
//
// Tiling: BM x BN block tile, BK = 32 or 16, WGM x WGN waves (4, 8 or 16), each wave owns
// TM x TN MFMA 32x32 sub-tiles.  LDS tiles are row-major [rows][BK + 4] (the +4 keeps 16-byte
// alignment and spreads ds_read_b128 over the 64 banks), double-buffered, one barrier per k-tile.
// Staging is global -> registers -> LDS with a 2-deep register prefetch: k-tile j+2 is in flight
// while k-tile j+1 (loaded an iteration earlier) is written to the other LDS buffer between the
// MFMA groups of k-tile j.  Eight tile configurations exist (enum TileCfg); the first launch of a
// shape times the valid ones and the choice is cached (all of them produce identical bits).
// The MFMA consumes k in the order lanes<32: {8q+e}, lanes>=32: {8q+4+e} so that each lane's
// operand for four consecutive k-steps is one ds_read_b128 (summation order inside a k-tile is a
// fixed permutation; results are deterministic).
#include "common.h"
#include <string.h>
#include <hip/hip_ext.h>
#include "kernels.h"
#include <mutex>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <tuple>
#include <type_traits>
#include <vector>

namespace d4 {


template <int BM, int BN, int WGM, int WGN, int BK, int KS, bool TA, bool TB, bool KTAIL>
__global__ __launch_bounds__(WGM * WGN * 64 * KS) void gemm_kernel(GemmArgs p) {
    constexpr int LDS_LD = BK + 4;
    constexpr int RF4 = BK / 4;                 // float4 per tile row
    constexpr int NT = WGM * WGN * 64 * KS;     // threads per block (KS > 1: intra-block split of each k-tile across wave groups)
    constexpr int TM = BM / WGM / 32;
    constexpr int TN = BN / WGN / 32;
    static_assert(TM >= 1 && TN >= 1, "tile");
    constexpr int A_F4 = BM * BK / 4 / NT;    // float4 per thread per A tile
    constexpr int B_F4 = BN * BK / 4 / NT;
    static_assert(A_F4 >= 1 && B_F4 >= 1, "tile too small for the thread count");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                 // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;               // [2][BN][LDS_LD]
    float* rowscale_s = Bs + 2 * BN * LDS_LD;         // [BM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = (tid >> 6) % (WGM * WGN);
    const int ks = (tid >> 6) / (WGM * WGN);    // which share of the k-steps of every tile this wave group takes
    const int wm = wave / WGN, wn = wave % WGN;

    // XCD-aware block order: consecutive blocks on one XCD share the same A row-panel.
    int bid = blockIdx.x;
    const int nbn = (p.N + BN - 1) / BN;
    const int nbm = (p.M + BM - 1) / BM;
    const int nblk = nbm * nbn;
    {
        const int nx = 8;
        int q = nblk / nx, r = nblk % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    }
    const int bm0 = (bid / nbn) * BM;
    const int bn0 = (bid % nbn) * BN;

    // Operands are read through buffer descriptors (wave-uniform, built from kernel arguments and the block
    // index only): out-of-range rows fall outside num_records and return 0 in hardware, so the staging loads are
    // branch-free; logically-invalid lanes (k tail, partial float4 of a transposed operand) are steered to an
    // out-of-range offset / masked with selects instead of exec-mask branches around each load.
    constexpr uint32_t OOB = 0x80000000u;
    const int rowsA = min(BM, p.M - bm0), rowsB = min(BN, p.N - bn0);
    const int bz = blockIdx.y;
    p.A += bz * p.strideA; p.W += bz * p.strideW; p.C += bz * p.strideC;
    if (p.R) p.R += bz * p.strideC;
    const float* baseA = TA ? p.A + bm0 : p.A + (int64_t)bm0 * p.lda;
    const float* baseB = TB ? p.W + bn0 : p.W + (int64_t)bn0 * p.ldw;
    const int64_t elemsA = TA ? (int64_t)(p.K - 1) * p.lda + rowsA : (int64_t)(rowsA - 1) * p.lda + p.K;
    const int64_t elemsB = TB ? (int64_t)(p.K - 1) * p.ldw + rowsB : (int64_t)(rowsB - 1) * p.ldw + p.K;
    // make the descriptor inputs PROVABLY wave-uniform (readfirstlane), otherwise hipcc wraps every buffer load
    // in a waterfall loop (cdna_hip_programming.md T20)
    auto uniform_rsrc = [](const float* base, int64_t bytes) {
        const uint64_t b = reinterpret_cast<uint64_t>(base);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        const int nb = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7FFFFFFF ? bytes : 0x7FFFFFFF));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc(baseA, elemsA * 4);
    const __amdgpu_buffer_rsrc_t rsB = uniform_rsrc(baseB, elemsB * 4);

    // Two register sets for the global->LDS staging: the loads of k-tile j+2 are in flight while k-tile j+1 (loaded one
    // iteration earlier, certainly arrived) is written to the other LDS buffer in the shadow of the MFMAs of tile j.
    f32x4 ra[2][A_F4], rb[2][B_F4];
    // Row sums of squares for the folded RMSNorm, in ONE canonical order whatever the tile configuration: eight running sums
    // per row (16-byte chunk index mod 8 along k), each fed in k order with an explicitly fused a0^2 + a1^2 + a2^2 + a3^2, then
    // the tree ((0+1)+(2+3)) + ((4+5)+(6+7)).  With BK = 32 a thread owns one chunk position; with BK = 16 it owns position c of
    // even k-tiles and 4 + c of odd ones.  (Needed for the tuner's freedom: every configuration must give the same bits.)
    static_assert(BK == 16 || BK == 32, "canonical row-sum order is defined for BK = 16 / 32");
    constexpr int NPAR = 32 / BK;
    float ssq[A_F4][NPAR];
#pragma unroll
    for (int i = 0; i < A_F4; ++i)
#pragma unroll
        for (int h = 0; h < NPAR; ++h) ssq[i][h] = 0.f;

    // one operand tile -> registers.  Non-transposed: memory [rows][K], float4 along K.  Transposed: memory
    // [K][rows], float4 along rows (scattered into LDS by store_tile).
    auto load_operand = [&](auto& regs, const __amdgpu_buffer_rsrc_t rs, int ld, int rows_valid, int k0, auto trans_tag, auto rows_tag) {
        constexpr bool T = decltype(trans_tag)::value;
        constexpr int ROWS = decltype(rows_tag)::value;
        constexpr int NF4 = ROWS * BK / 4 / NT;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int idx = tid + i * NT;
            if constexpr (!T) {
                const int r = idx / RF4, c = (idx % RF4) * 4;
                const int gk = k0 + c;
                uint32_t off = (uint32_t)((r * ld + gk) * 4);
                if (KTAIL) off = gk < p.K ? off : OOB;
                f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
                if (KTAIL) {
#pragma unroll
                    for (int e = 1; e < 4; ++e) v[e] = gk + e < p.K ? v[e] : 0.f;
                }
                regs[i] = v;
            } else {
                const int kk = idx / (ROWS / 4), c = (idx % (ROWS / 4)) * 4;
                const int gk = k0 + kk;
                uint32_t off = (uint32_t)((gk * ld + c) * 4);
                off = (gk < p.K && c < rows_valid) ? off : OOB;
                f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
#pragma unroll
                for (int e = 1; e < 4; ++e) v[e] = c + e < rows_valid ? v[e] : 0.f;
                regs[i] = v;
            }
        }
    };
    auto load_tile = [&](int k0, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        load_operand(ra[SET], rsA, p.lda, rowsA, k0, std::integral_constant<bool, TA>{}, std::integral_constant<int, BM>{});
        load_operand(rb[SET], rsB, p.ldw, rowsB, k0, std::integral_constant<bool, TB>{}, std::integral_constant<int, BN>{});
    };

    auto store_tile = [&](int buf, auto set_tag, auto parity_tag) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr int PAR = decltype(parity_tag)::value % NPAR;
        float* as = As + buf * BM * LDS_LD;
        float* bs = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            int idx = tid + i * NT;
            const f32x4 v = ra[SET][i];
            if constexpr (!TA) {
                int r = idx / RF4, c = (idx % RF4) * 4;
                *reinterpret_cast<f32x4*>(as + r * LDS_LD + c) = v;
                ssq[i][PAR] = ssq[i][PAR] + __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], v[0] * v[0])));
            } else {
                int kk = idx / (BM / 4), c = (idx % (BM / 4)) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) as[(c + e) * LDS_LD + kk] = v[e];
            }
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            int idx = tid + i * NT;
            const f32x4 v = rb[SET][i];
            if constexpr (!TB) {
                int r = idx / RF4, c = (idx % RF4) * 4;
                *reinterpret_cast<f32x4*>(bs + r * LDS_LD + c) = v;
            } else {
                int kk = idx / (BN / 4), c = (idx % (BN / 4)) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) bs[(c + e) * LDS_LD + kk] = v[e];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    const int nk = (p.K + BK - 1) / BK;
    load_tile(0, Set0{});
    store_tile(0, Set0{}, Set0{});
    if (nk > 1) load_tile(BK, Set0{});            // k-tile j lives in register set (j - 1) & 1 until it is staged
    if (nk > 2) load_tile(2 * BK, Set1{});
    __syncthreads();

    const int lrow = lane & 31;
    const int lhalf = lane >> 5;

    auto mfma_steps = [&](int cur, int q0, int q1) {
        const float* as = As + cur * BM * LDS_LD + (wm * TM * 32 + lrow) * LDS_LD + lhalf * 4;
        const float* bs = Bs + cur * BN * LDS_LD + (wn * TN * 32 + lrow) * LDS_LD + lhalf * 4;
#pragma unroll
        for (int q = q0; q < q1; ++q) {
            if (KS > 1 && (q % KS) != ks) continue;
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDS_LD + q * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDS_LD + q * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
    };
    auto iteration = [&](int kt, auto set_tag, auto next_parity) {
        const int cur = kt & 1;
        mfma_steps(cur, 0, 1);
        if (kt + 1 < nk) store_tile(cur ^ 1, set_tag, next_parity);   // k-tile kt+1: in registers since the previous iteration
        if (kt + 3 < nk) load_tile((kt + 3) * BK, set_tag);     // refill the set just drained
        mfma_steps(cur, 1, BK / 8);
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        iteration(kt, Set0{}, Set1{});                            // kt even: the tile staged is odd
        if (kt + 1 < nk) iteration(kt + 1, Set1{}, Set0{});
    }

    // ---- intra-block split-K: fold the partner group's accumulators through LDS (fixed order: group 0 + group 1)
    if constexpr (KS > 1) {
        static_assert(KS == 2, "");
        constexpr int GT = WGM * WGN * 64;
        float* xch = smem;                       // the operand tiles are dead after the last barrier of the main loop
        const int gt = tid % GT;
        if (ks == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) xch[((i * TN + j) * 16 + e) * GT + gt] = acc[i][j][e];
        }
        __syncthreads();
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] += xch[((i * TN + j) * 16 + e) * GT + gt];
        }
    }

    // ---- per-row 1/rms of A (RMSNorm folded into the GEMM) -----------------------------------
    if (p.flags & GEMM_RMS_ROWSCALE) {
        if constexpr (!TA) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                float s = ssq[i][0];
                s += dpp_f<0xB1>(s);
                s += dpp_f<0x4E>(s);    // RF4 (4 or 8) consecutive lanes share one row
                if constexpr (NPAR == 1) {
                    s += dpp_f<0x141>(s);                          // (0123) + (4567): the other half of the row's 8 lanes
                } else {
                    float s1 = ssq[i][1];
                    s1 += dpp_f<0xB1>(s1);
                    s1 += dpp_f<0x4E>(s1);
                    s = s + s1;                                    // (0123) + (4567)
                }
                int r = (tid + i * NT) / RF4;
                if ((tid % RF4) == 0) rowscale_s[r] = rsqrtf(s / (float)p.K + p.rms_eps);
            }
        }
        __syncthreads();
    }

    // ---- epilogue ----------------------------------------------------------------------------
    if (KS > 1 && ks != 0) return;
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
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
                static_assert(TN >= 1, "");
                if constexpr (TN % 2 == 0) {
#pragma unroll
                    for (int j = 0; j < TN; j += 2) {
                        const int gn = bn0 + wn * TN * 32 + j * 32 + lrow;       // packed column of the value
                        if (gn + 32 >= p.N + 32 || gn >= p.N) continue;
                        float val = acc[i][j][e] * rs, gate = acc[i][j + 1][e] * rs;
                        if (p.bias) { val += p.bias[gn]; gate += p.bias[gn + 32]; }
                        const int on = (gn / 64) * 32 + (gn % 64);                 // output (hidden) column
                        p.C[(int64_t)gm * p.ldc + on] = val * siluf(gate);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int gn = bn0 + wn * TN * 32 + j * 32 + lrow;
                    if (gn >= p.N) continue;
                    float v = acc[i][j][e] * rs;
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

// ---- optional per-launch timing (bench.py roofline leg): HIP events on the launch stream --------------------
struct ProfRec { hipEvent_t a, b; int cls; double flops; int M, N, K, flags, batch, bm, bn; };
static int g_prof_mask = 0;            // bit c set: time launches of tile configuration c
static int g_prof_stride = 1;          // ... every g_prof_stride-th of them
static int g_prof_tick = 0;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_event_pool;

static hipEvent_t prof_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

bool gemm_bf16_profile_active();
bool glue_profile_active();            // prof.cpp: event-timed launches cannot be captured into a graph either
bool gemm_profile_active() { return g_prof_mask != 0 || gemm_bf16_profile_active() || glue_profile_active(); }

static std::recursive_mutex& gemm_mutex();
int gemm_profile_enable(int mask) {
    std::lock_guard<std::recursive_mutex> lock(gemm_mutex());
    g_prof_stride = (mask >> 27) > 0 ? (mask >> 27) : 1;          // bits 27..30: stride; bits 0..26: configurations (27 classes since round 6)
    mask &= 0x7FFFFFF;
    g_prof_tick = 0;
    g_prof_mask = mask;
    return 0;
}

// Sums elapsed time / algorithmic flops / launch count per tile class (0: 128x128, 1: 64x128, 2: 64x64) and clears the log.
int gemm_profile_read(double* ms, double* flops, int64_t* count, int nclass) {
    std::lock_guard<std::recursive_mutex> lock(gemm_mutex());
    for (int i = 0; i < nclass; ++i) { ms[i] = 0; flops[i] = 0; count[i] = 0; }
    // D4_GEMM_LOG=1: also print a per-shape table (launches, total ms, TFLOP/s) to stderr
    static const bool log_shapes = getenv("D4_GEMM_LOG") != nullptr;
    struct Agg { ProfRec r; double ms, fl; int64_t n; };
    std::vector<Agg> shapes;
    for (auto& r : g_prof) {
        D4_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        D4_HIP(hipEventElapsedTime(&t, r.a, r.b));
        if (r.cls < nclass) { ms[r.cls] += t; flops[r.cls] += r.flops; count[r.cls] += 1; }
        if (log_shapes) {
            bool hit = false;
            for (auto& a : shapes)
                if (a.r.M == r.M && a.r.N == r.N && a.r.K == r.K && a.r.flags == r.flags && a.r.batch == r.batch) { a.ms += t; a.fl += r.flops; a.n += 1; hit = true; break; }
            if (!hit) shapes.push_back(Agg{r, t, r.flops, 1});
        }
        g_event_pool.push_back(r.a);
        g_event_pool.push_back(r.b);
    }
    if (log_shapes) {
        double tot = 0;
        for (auto& a : shapes) tot += a.ms;
        for (auto& a : shapes)
            fprintf(stderr, "[d4 gemm] M %6d N %5d K %5d batch %2d flags %3d tile %3dx%-3d : %6lld launches %9.3f ms (%5.1f %%) avg %7.1f us %6.1f TF/s\n",
                    a.r.M, a.r.N, a.r.K, a.r.batch, a.r.flags, a.r.bm, a.r.bn, (long long)a.n, a.ms, 100 * a.ms / tot, 1e3 * a.ms / a.n, a.fl / a.ms / 1e9);
    }
    g_prof.clear();
    return 0;
}


static inline double tile_cost(const GemmArgs& p, int BM, int BN, int slots) {
    // co-resident blocks share a CU's matrix pipes: a CU's time ~ (blocks it hosts) x (tile area)
    const double blocks = (double)cdiv(p.M, BM) * cdiv(p.N, BN) * (p.batch > 0 ? p.batch : 1);
    return ceil(blocks / slots) * BM * BN;
}

template <int BM, int BN, int WGM, int WGN, int BK, int KS, bool TA, bool TB>
static int launch_cfg(const GemmArgs& p, hipStream_t stream, int cls) {
    constexpr int LDS_LD = BK + 4;
    const int nblk = cdiv(p.M, BM) * cdiv(p.N, BN);
    const size_t lds = (size_t)(2 * BM * LDS_LD + 2 * BN * LDS_LD + BM) * sizeof(float);
    const bool ktail = (p.K % BK) != 0;
    auto k = ktail ? gemm_kernel<BM, BN, WGM, WGN, BK, KS, TA, TB, true> : gemm_kernel<BM, BN, WGM, WGN, BK, KS, TA, TB, false>;
    static DeviceOnce attr_set[2];
    if (attr_set[ktail].need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[ktail].done();
    }
    ProfRec rec{};
    const bool g_prof_on = ((g_prof_mask >> cls) & 1) && (g_prof_tick++ % g_prof_stride) == 0;
    if (g_prof_on) {
        rec.a = prof_event(); rec.b = prof_event(); rec.cls = cls;
        rec.M = p.M; rec.N = p.N; rec.K = p.K; rec.flags = p.flags; rec.batch = p.batch; rec.bm = BM; rec.bn = BN;
        rec.flops = p.algo_flops > 0 ? p.algo_flops : 2.0 * p.M * p.N * p.K * (p.batch > 0 ? p.batch : 1);
        // the two events ride on the dispatch itself (its begin / end timestamps): no extra barrier packets on the stream,
        // and the elapsed time is the kernel's own duration (what rocprofv3 --kernel-trace reports)
        hipExtLaunchKernelGGL(k, dim3(nblk, p.batch > 0 ? p.batch : 1), dim3(WGM * WGN * 64 * KS), (uint32_t)lds, stream, rec.a, rec.b, 0, p);
        g_prof.push_back(rec);
    } else {
        hipLaunchKernelGGL(k, dim3(nblk, p.batch > 0 ? p.batch : 1), dim3(WGM * WGN * 64 * KS), lds, stream, p);
    }
    D4_LAUNCH_CHECK();
    return 0;
}

// ---- tile configurations and their selection ---------------------------------------------------------------
// All configurations walk k in the same order with the same MFMA, so every output element is the same
// bit pattern whichever one runs: the choice is a pure performance decision.
enum TileCfg { T128x128_2x4 = 0, T128x128_4x2, T128x96_4x1, T64x128_2x2, T64x64_2x2, T256x128_4x4, T64x128_k16, T64x64_k16, N_TILE_CFG };
static const char* const kTileName[N_TILE_CFG] = {"128x128/2x4", "128x128/4x2", "128x96/4x1", "64x128/2x2", "64x64/2x2", "256x128/4x4", "64x128/2x2/k16", "64x64/2x2/k16"};

// rocprofv3 shows the instantiation as d4::gemm_kernel<BM, BN, WGM, WGN, BK, 1, transA, transB, ktail>
static const char* const kTileKernel[N_TILE_CFG] = {
    "gemm_kernel<128, 128, 2, 4, 32, 1", "gemm_kernel<128, 128, 4, 2, 32, 1", "gemm_kernel<128, 96, 4, 1, 32, 1", "gemm_kernel<64, 128, 2, 2, 32, 1",
    "gemm_kernel<64, 64, 2, 2, 32, 1", "gemm_kernel<256, 128, 4, 4, 32, 1", "gemm_kernel<64, 128, 2, 2, 16, 1", "gemm_kernel<64, 64, 2, 2, 16, 1"};
// profile classes: the N_TILE_CFG configurations of this family, then the configurations of the second family (gemm2.hip)
int gemm_profile_classes() { return N_TILE_CFG + gemm2_configs() + gemm_x3_configs() + 3; }     // + the persistent split-operand form (gemm_x3sk.hip) + the fp16x2 family (gemm_h2.hip, one class) + the few-row k-split form (gemm2.hip)
const char* gemm_profile_class_name(int c) {
    if (c == N_TILE_CFG + gemm2_configs() + gemm_x3_configs() + 2) return "gemm2_ksplit_kernel";
    if (c == N_TILE_CFG + gemm2_configs() + gemm_x3_configs() + 1) return "gemm_h2_kernel";
    if (c == N_TILE_CFG + gemm2_configs() + gemm_x3_configs()) return gemm_x3sk_name();
    if (c >= N_TILE_CFG + gemm2_configs()) return gemm_x3_config_name(c - N_TILE_CFG - gemm2_configs());
    if (c >= N_TILE_CFG) return gemm2_config_name(c - N_TILE_CFG);
    return c >= 0 ? kTileKernel[c] : "";
}

template <bool TA, bool TB>
static bool cfg_valid(int id, const GemmArgs& p) {
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    switch (id) {
        case T128x128_2x4: return !swiglu;
        case T128x128_4x2: return true;
        case T128x96_4x1: return !swiglu && !TA && !TB;
        case T64x128_2x2: return true;
        case T64x64_2x2: return !swiglu;
        case T256x128_4x4: return !swiglu && !TA && !TB;
        case T64x128_k16: return !TA && !TB;
        case T64x64_k16: return !swiglu && !TA && !TB;
    }
    return false;
}

template <bool TA, bool TB>
static int launch_id(int id, const GemmArgs& p, hipStream_t stream) {
    switch (id) {
        // 8 waves (2 x 4, wave tile 64 x 32): two co-resident blocks put 4 waves on every SIMD
        case T128x128_2x4: return launch_cfg<128, 128, 2, 4, 32, 1, TA, TB>(p, stream, id);
        // (the SiLU-GLU epilogue pairs two N sub-tiles inside one wave -> 4 x 2 waves, wave tile 32 x 64)
        case T128x128_4x2: return launch_cfg<128, 128, 4, 2, 32, 1, TA, TB>(p, stream, id);
        case T128x96_4x1:
            if constexpr (!TA && !TB) return launch_cfg<128, 96, 4, 1, 32, 1, TA, TB>(p, stream, id);
            break;
        case T64x128_2x2: return launch_cfg<64, 128, 2, 2, 32, 1, TA, TB>(p, stream, id);
        case T64x64_2x2: return launch_cfg<64, 64, 2, 2, 32, 1, TA, TB>(p, stream, id);
        case T256x128_4x4:
            if constexpr (!TA && !TB) return launch_cfg<256, 128, 4, 4, 32, 1, TA, TB>(p, stream, id);
            break;
        case T64x128_k16:
            if constexpr (!TA && !TB) return launch_cfg<64, 128, 2, 2, 16, 1, TA, TB>(p, stream, id);
            break;
        case T64x64_k16:
            if constexpr (!TA && !TB) return launch_cfg<64, 64, 2, 2, 16, 1, TA, TB>(p, stream, id);
            break;
    }
    D4_REQUIRE(false, "gemm: tile configuration %d not available for this operand layout", id);
}

// Static choice (used for shapes that cannot be timed: tiny, non-idempotent, or first seen under stream capture).
template <bool TA, bool TB>
static int heuristic_cfg(const GemmArgs& p) {
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    const int nb = p.batch > 0 ? p.batch : 1;
    const int64_t b128 = (int64_t)cdiv(p.M, 128) * cdiv(p.N, 128) * nb;
    // 256 CUs: prefer the big tile once it yields >= ~1.5 blocks per CU; long-K backward GEMMs (dW = dZ^T X with
    // K = batch*time rows) take it earlier: one big tile per CU beats two rounds of small ones.
    if (b128 >= 360 || (b128 >= 192 && p.K >= 2048)) {
        if (swiglu) return T128x128_4x2;
        // 128 x 96 tiles when they quantise better onto the 256 CUs (the fused q|k|v projection, N = 1552: 30 x 13 = 390
        // tiles of 128 x 128 leave half the CUs with one block and half with two; 30 x 17 = 510 fit once).
        if (!TA && !TB && tile_cost(p, 128, 96, 256) < 0.8 * tile_cost(p, 128, 128, 256)) return T128x96_4x1;
        return T128x128_2x4;
    }
    if (swiglu || (p.N > 64 && (int64_t)cdiv(p.M, 64) * cdiv(p.N, 128) * nb >= 256)) return T64x128_2x2;
    return T64x64_2x2;
}

// Shape -> configuration, filled by timing every valid configuration the first time a shape is seen (the engine's
// shapes are fixed by (batch, frames, config), a few dozen in all).  D4_GEMM_AUTOTUNE=0 keeps the static choice.
struct TuneKey {
    int M, N, K, flags, batch;
    bool operator<(const TuneKey& o) const { return std::tie(M, N, K, flags, batch) < std::tie(o.M, o.N, o.K, o.flags, o.batch); }
};
static std::map<TuneKey, int> g_tuned, g_tuned2, g_tuned3, g_tuned4;
static int g_forced_cfg = -1;          // test hook (d4_gemm_force_config): run this configuration wherever it is valid;
                                       // 100 + c: configuration c of the second family (gemm2.hip)
int gemm_force_config(int id) { g_forced_cfg = id; return N_TILE_CFG; }

// D4_GEMM_TUNE_CACHE=<file>: the shape -> configuration table is read at first use and every new entry is appended, so a
// later process (a profiler pass, a restarted trainer) starts with the choices already made and never re-times.
static const char* tune_cache_path() {
    static const char* path = getenv("D4_GEMM_TUNE_CACHE");
    return path;
}
static void tune_cache_read(const char* path) {
    if (!path) return;
    if (FILE* f = fopen(path, "r")) {
        char line[256];
        while (fgets(line, sizeof(line), f)) {
            TuneKey k; int id;
            if (line[0] == '#' || sscanf(line, "%d %d %d %d %d %d", &k.M, &k.N, &k.K, &k.flags, &k.batch, &id) != 6) continue;
            if (id >= 0 && id < N_TILE_CFG) g_tuned[k] = id;
            else if (id >= 100 && id < 100 + gemm2_configs()) g_tuned2[k] = id - 100;      // second family (gemm2.hip)
            else if (id >= 300 && id < 300 + gemm_x3_configs()) g_tuned3[k] = id - 300;    // split-operand family (gemm_x3.hip)
            else if (id >= 400 && id < 400 + gemm_h2_configs()) g_tuned4[k] = id - 400;    // fp16x2 split-operand family (gemm_h2.hip)
        }
        fclose(f);
    }
}
// D4_GEMM_TUNE_DEFAULT (set by dreamer4_amd._lib to the table shipped in the package): preloaded read-only, so the shapes of the
// benchmark configurations never time anything (no first-call synchronisation, the same choice on every run);
// D4_GEMM_TUNE_CACHE=<file>: read as well, and every newly tuned shape is appended to it.
static void tune_cache_load() {
    static bool loaded = false;
    if (loaded) return;
    loaded = true;
    tune_cache_read(getenv("D4_GEMM_TUNE_DEFAULT"));
    tune_cache_read(tune_cache_path());
}
static void tune_cache_append(const TuneKey& k, int id) {
    const char* path = tune_cache_path();
    if (!path) return;
    if (FILE* f = fopen(path, "a")) {
        fprintf(f, "%d %d %d %d %d %d\n", k.M, k.N, k.K, k.flags, k.batch, id);
        fclose(f);
    }
}

// D4_GEMM_AUTOTUNE: 1 / unset = time a new shape's configurations at first use; 0 = never time (static choice); strict = never time AND fail
// on a shape that would have been timed (the multi-rank bench: every rank must take its choices from the shipped table, not from its own clock).
static int tune_mode() {
    static const int mode = [] {
        const char* v = getenv("D4_GEMM_AUTOTUNE");
        if (!v) return 1;
        if (!strcmp(v, "strict")) return 2;
        return atoi(v) == 0 ? 0 : 1;
    }();
    return mode;
}
#define D4_TUNE_STRICT_CHECK(p, nb)                                                                                                            \
    D4_REQUIRE(tune_mode() != 2, "gemm: shape M=%d N=%d K=%d flags=%d batch=%d is not in the shipped tile table (dreamer4_amd/gemm_tune_default.txt) and " \
               "D4_GEMM_AUTOTUNE=strict forbids timing it here: run the workload once on one GPU with D4_GEMM_TUNE_CACHE=<file> and merge the new lines",  \
               (p).M, (p).N, (p).K, (p).flags, (nb))

template <bool TA, bool TB>
static int autotune(const GemmArgs& p, hipStream_t stream, int* best_out) {
    hipEvent_t e0 = prof_event(), e1 = prof_event();
    const int saved_mask = g_prof_mask;
    g_prof_mask = 0;
    int best = -1, rc = 0;
    float best_ms = 0.f;
    for (int id = 0; id < N_TILE_CFG && !rc; ++id) {
        if (!cfg_valid<TA, TB>(id, p)) continue;
        if ((id == T256x128_4x4 || id == T128x128_2x4 || id == T128x128_4x2) && (int64_t)cdiv(p.M, 128) * cdiv(p.N, 128) * (p.batch > 0 ? p.batch : 1) < 64) continue;
        if ((rc = launch_id<TA, TB>(id, p, stream))) break;              // warm (attribute set, code resident)
        float ms = 1e30f;
        for (int rep = 0; rep < 2 && !rc; ++rep) {                       // best of two timed pairs
            (void)hipEventRecord(e0, stream);
            if ((rc = launch_id<TA, TB>(id, p, stream))) break;
            if ((rc = launch_id<TA, TB>(id, p, stream))) break;
            (void)hipEventRecord(e1, stream);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = 1; break; }
            float t = 0.f;
            (void)hipEventElapsedTime(&t, e0, e1);
            ms = t < ms ? t : ms;
        }
        if (!rc && (best < 0 || ms < best_ms)) { best = id; best_ms = ms; }
    }
    g_prof_mask = saved_mask;
    g_event_pool.push_back(e0);
    g_event_pool.push_back(e1);
    if (rc) return rc;
    D4_REQUIRE(best >= 0, "gemm: no tile configuration for M=%d N=%d K=%d flags=%d", p.M, p.N, p.K, p.flags);
    if (getenv("D4_GEMM_LOG"))
        fprintf(stderr, "[d4 gemm] tuned M %6d N %5d K %5d batch %2d flags %3d -> %s (%.1f us)\n", p.M, p.N, p.K, p.batch, p.flags, kTileName[best], 500.f * best_ms);
    *best_out = best;
    return 0;
}

template <bool TA, bool TB>
static int launch_t(const GemmArgs& p, hipStream_t stream) {
    const bool tune_on = tune_mode() != 0;
    const int nb = p.batch > 0 ? p.batch : 1;
    const TuneKey key{p.M, p.N, p.K, p.flags, p.batch};
    if (g_forced_cfg >= 0 && g_forced_cfg < N_TILE_CFG && cfg_valid<TA, TB>(g_forced_cfg, p)) return launch_id<TA, TB>(g_forced_cfg, p, stream);
    tune_cache_load();
    auto it = g_tuned.find(key);
    if (it != g_tuned.end() && cfg_valid<TA, TB>(it->second, p)) return launch_id<TA, TB>(it->second, p, stream);
    // timing repeats the launch, so the call must be idempotent (no accumulate, no in-place residual), worth it
    // (>= 0.1 GFLOP), and the stream must not be capturing
    const bool idempotent = !(p.flags & GEMM_ACCUMULATE) && p.R != p.C && p.A != p.C;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    if (!tune_on || !idempotent || cap != hipStreamCaptureStatusNone || 2.0 * p.M * p.N * p.K * nb < 1e8)
        return launch_id<TA, TB>(heuristic_cfg<TA, TB>(p), p, stream);
    // strict mode: a FEW-ROW product missing from the table (the heads' MLPs at a per-rank batch other than the shipped 256: M = the batch) takes the static
    // choice — a rule on the shape, so every rank still makes the same one without a clock; anything larger must be in the table
    if (tune_mode() == 2 && p.M <= 512) return launch_id<TA, TB>(heuristic_cfg<TA, TB>(p), p, stream);
    D4_TUNE_STRICT_CHECK(p, nb);
    int best = 0;
    if (int rc = autotune<TA, TB>(p, stream, &best)) return rc;
    g_tuned[key] = best;
    tune_cache_append(key, best);
    return launch_id<TA, TB>(best, p, stream);
}

// ---- second family: launch (with the optional event pair), per-shape choice among its configurations by timing ----
static int launch_v2(int c, const GemmArgs& p, hipStream_t stream) {
    const int cls = N_TILE_CFG + c;
    const bool timed = ((g_prof_mask >> cls) & 1) && (g_prof_tick++ % g_prof_stride) == 0;
    if (!timed) return gemm2_launch(c, p, stream);
    ProfRec rec{};
    rec.a = prof_event(); rec.b = prof_event(); rec.cls = cls;
    rec.M = p.M; rec.N = p.N; rec.K = p.K; rec.flags = p.flags; rec.batch = p.batch;
    gemm2_config_tile(c, &rec.bm, &rec.bn);
    rec.flops = p.algo_flops > 0 ? p.algo_flops : 2.0 * p.M * p.N * p.K * (p.batch > 0 ? p.batch : 1);
    if (int rc = gemm2_launch(c, p, stream, rec.a, rec.b)) return rc;
    g_prof.push_back(rec);
    return 0;
}

static int launch_v2ks(const GemmArgs& p, hipStream_t stream) {
    const int cls = N_TILE_CFG + gemm2_configs() + gemm_x3_configs() + 2;
    const bool timed = ((g_prof_mask >> cls) & 1) && (g_prof_tick++ % g_prof_stride) == 0;
    if (!timed) return gemm2_ksplit_launch(p, stream);
    ProfRec rec{};
    rec.a = prof_event(); rec.b = prof_event(); rec.cls = cls;
    rec.M = p.M; rec.N = p.N; rec.K = p.K; rec.flags = p.flags; rec.batch = p.batch;
    rec.bm = 32; rec.bn = 64;
    rec.flops = p.algo_flops > 0 ? p.algo_flops : 2.0 * p.M * p.N * p.K;
    if (int rc = gemm2_ksplit_launch(p, stream, rec.a, rec.b)) return rc;
    g_prof.push_back(rec);
    return 0;
}

static int heuristic_v2(const GemmArgs& p) {
    // (gemm2.hip's enum) 0 = 64x64 with three blocks per CU, 1 = its SiLU-GLU capable form: the best or within a few per cent
    // of it on every shape measured; the timed choice refines this where a call can be repeated
    return (p.flags & GEMM_SWIGLU) ? 1 : 0;
}

static int autotune_v2(const GemmArgs& p, hipStream_t stream, int* best_out) {
    hipEvent_t e0 = prof_event(), e1 = prof_event();
    const int saved_mask = g_prof_mask;
    g_prof_mask = 0;
    int best = -1, rc = 0;
    float best_ms = 0.f;
    for (int c = 0; c < gemm2_configs() && !rc; ++c) {
        if (!gemm2_config_valid(c, p)) continue;
        if ((rc = gemm2_launch(c, p, stream))) break;
        float ms = 1e30f;
        for (int rep = 0; rep < 2 && !rc; ++rep) {
            (void)hipEventRecord(e0, stream);
            if ((rc = gemm2_launch(c, p, stream))) break;
            if ((rc = gemm2_launch(c, p, stream))) break;
            (void)hipEventRecord(e1, stream);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = 1; break; }
            float t = 0.f;
            (void)hipEventElapsedTime(&t, e0, e1);
            ms = t < ms ? t : ms;
        }
        if (!rc && (best < 0 || ms < best_ms)) { best = c; best_ms = ms; }
    }
    g_prof_mask = saved_mask;
    g_event_pool.push_back(e0);
    g_event_pool.push_back(e1);
    if (rc) return rc;
    D4_REQUIRE(best >= 0, "gemm2: no configuration for M=%d N=%d K=%d flags=%d", p.M, p.N, p.K, p.flags);
    if (getenv("D4_GEMM_LOG"))
        fprintf(stderr, "[d4 gemm2] tuned M %6d N %5d K %5d batch %2d flags %3d -> %s (%.1f us)\n", p.M, p.N, p.K, p.batch, p.flags, gemm2_config_name(best), 500.f * best_ms);
    *best_out = best;
    return 0;
}

// Which family runs a call is a RULE on the call's shape / layout (never a timing): the two families sum k in different orders,
// so a timing-dependent choice between them would make results depend on the tuner.
static bool use_v2(const GemmArgs& p) { return gemm2_applicable(p); }

static int gemm_v2(const GemmArgs& p, hipStream_t stream) {
    const bool tune_on = tune_mode() != 0;
    const int nb = p.batch > 0 ? p.batch : 1;
    const TuneKey key{p.M, p.N, p.K, p.flags, p.batch};
    tune_cache_load();
    auto it = g_tuned2.find(key);
    if (it != g_tuned2.end() && gemm2_config_valid(it->second, p)) return launch_v2(it->second, p, stream);
    const bool idempotent = !(p.flags & GEMM_ACCUMULATE) && p.R != p.C && p.A != p.C;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    if (!tune_on || !idempotent || cap != hipStreamCaptureStatusNone || 2.0 * p.M * p.N * p.K * nb < 1e8)
        return launch_v2(heuristic_v2(p), p, stream);
    if (tune_mode() == 2 && p.M <= 512) return launch_v2(heuristic_v2(p), p, stream);      // strict mode, few-row product missing from the table: the static rule (see launch_t)
    D4_TUNE_STRICT_CHECK(p, nb);
    int best = 0;
    if (int rc = autotune_v2(p, stream, &best)) return rc;
    g_tuned2[key] = best;
    tune_cache_append(key, 100 + best);
    return launch_v2(best, p, stream);
}

// ---- third family (split operands on the bf16 matrix cores): launch with the optional event pair, timed choice among its tiles ----
static int launch_v3(int c, const GemmArgs& p, hipStream_t stream) {
    const int cls = N_TILE_CFG + gemm2_configs() + c;
    const bool timed = ((g_prof_mask >> cls) & 1) && (g_prof_tick++ % g_prof_stride) == 0;
    if (!timed) return gemm_x3_launch(c, p, stream);
    ProfRec rec{};
    rec.a = prof_event(); rec.b = prof_event(); rec.cls = cls;
    rec.M = p.M; rec.N = p.N; rec.K = p.K; rec.flags = p.flags; rec.batch = p.batch;
    gemm_x3_config_tile(c, &rec.bm, &rec.bn);
    rec.flops = p.algo_flops > 0 ? p.algo_flops : 2.0 * p.M * p.N * p.K * (p.batch > 0 ? p.batch : 1);
    if (int rc = gemm_x3_launch(c, p, stream, rec.a, rec.b)) return rc;
    g_prof.push_back(rec);
    return 0;
}

static int launch_v3sk(const GemmArgs& p, hipStream_t stream) {
    const int cls = N_TILE_CFG + gemm2_configs() + gemm_x3_configs();
    const bool timed = ((g_prof_mask >> cls) & 1) && (g_prof_tick++ % g_prof_stride) == 0;
    if (!timed) return gemm_x3sk_launch(p, stream);
    ProfRec rec{};
    rec.a = prof_event(); rec.b = prof_event(); rec.cls = cls;
    rec.M = p.M; rec.N = p.N; rec.K = p.K; rec.flags = p.flags; rec.batch = p.batch;
    rec.bm = 128; rec.bn = 128;
    rec.flops = p.algo_flops > 0 ? p.algo_flops : 2.0 * p.M * p.N * p.K;
    if (int rc = gemm_x3sk_launch(p, stream, rec.a, rec.b)) return rc;
    g_prof.push_back(rec);
    return 0;
}

static int autotune_v3(const GemmArgs& p, hipStream_t stream, int* best_out) {
    hipEvent_t e0 = prof_event(), e1 = prof_event();
    const int saved_mask = g_prof_mask;
    g_prof_mask = 0;
    int best = -1, rc = 0;
    float best_ms = 0.f;
    for (int c = 0; c < gemm_x3_configs() && !rc; ++c) {
        if (!gemm_x3_config_valid(c, p)) continue;
        if ((rc = gemm_x3_launch(c, p, stream))) break;
        float ms = 1e30f;
        for (int rep = 0; rep < 2 && !rc; ++rep) {
            (void)hipEventRecord(e0, stream);
            if ((rc = gemm_x3_launch(c, p, stream))) break;
            if ((rc = gemm_x3_launch(c, p, stream))) break;
            (void)hipEventRecord(e1, stream);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = 1; break; }
            float t = 0.f;
            (void)hipEventElapsedTime(&t, e0, e1);
            ms = t < ms ? t : ms;
        }
        if (!rc && (best < 0 || ms < best_ms)) { best = c; best_ms = ms; }
    }
    g_prof_mask = saved_mask;
    g_event_pool.push_back(e0);
    g_event_pool.push_back(e1);
    if (rc) return rc;
    D4_REQUIRE(best >= 0, "gemm_x3: no configuration for M=%d N=%d K=%d flags=%d", p.M, p.N, p.K, p.flags);
    if (getenv("D4_GEMM_LOG"))
        fprintf(stderr, "[d4 gemm_x3] tuned M %6d N %5d K %5d batch %2d flags %3d -> %s (%.1f us)\n", p.M, p.N, p.K, p.batch, p.flags, gemm_x3_config_name(best), 500.f * best_ms);
    *best_out = best;
    return 0;
}

static int gemm_v3(const GemmArgs& p, hipStream_t stream) {
    const bool tune_on = tune_mode() != 0;
    const int nb = p.batch > 0 ? p.batch : 1;
    if (g_forced_cfg >= 300 && gemm_x3_config_valid(g_forced_cfg - 300, p)) return launch_v3(g_forced_cfg - 300, p, stream);
    const TuneKey key{p.M, p.N, p.K, p.flags, p.batch};
    tune_cache_load();
    auto it = g_tuned3.find(key);
    if (it != g_tuned3.end() && gemm_x3_config_valid(it->second, p)) return launch_v3(it->second, p, stream);
    const bool idempotent = !(p.flags & GEMM_ACCUMULATE) && p.R != p.C && p.A != p.C;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    if (!tune_on || !idempotent || cap != hipStreamCaptureStatusNone || 2.0 * p.M * p.N * p.K * nb < 1e8)
        return launch_v3(gemm_x3_heuristic(p), p, stream);
    D4_TUNE_STRICT_CHECK(p, nb);
    int best = 0;
    if (int rc = autotune_v3(p, stream, &best)) return rc;
    g_tuned3[key] = best;
    tune_cache_append(key, 300 + best);
    return launch_v3(best, p, stream);
}

// ---- fp16x2 split-operand family (gemm_h2.hip; the opt-in `fp32_fp16x2` engine mode): launch with the optional event pair (ONE profile class for
// all its tiles), timed choice among its tiles (every tile gives the same bits) ----
static int launch_v4(int c, const GemmArgs& p, hipStream_t stream) {
    const int cls = N_TILE_CFG + gemm2_configs() + gemm_x3_configs() + 1;
    const bool timed = ((g_prof_mask >> cls) & 1) && (g_prof_tick++ % g_prof_stride) == 0;
    if (!timed) return gemm_h2_launch(c, p, stream);
    ProfRec rec{};
    rec.a = prof_event(); rec.b = prof_event(); rec.cls = cls;
    rec.M = p.M; rec.N = p.N; rec.K = p.K; rec.flags = p.flags; rec.batch = p.batch;
    gemm_h2_config_tile(c, &rec.bm, &rec.bn);
    rec.flops = p.algo_flops > 0 ? p.algo_flops : 2.0 * p.M * p.N * p.K * (p.batch > 0 ? p.batch : 1);
    if (int rc = gemm_h2_launch(c, p, stream, rec.a, rec.b)) return rc;
    g_prof.push_back(rec);
    return 0;
}

static int autotune_v4(const GemmArgs& p, hipStream_t stream, int* best_out) {
    hipEvent_t e0 = prof_event(), e1 = prof_event();
    const int saved_mask = g_prof_mask;
    g_prof_mask = 0;
    int best = -1, rc = 0;
    float best_ms = 0.f;
    for (int c = 0; c < gemm_h2_configs() && !rc; ++c) {
        if (!gemm_h2_config_valid(c, p)) continue;
        if ((rc = gemm_h2_launch(c, p, stream))) break;
        float ms = 1e30f;
        for (int rep = 0; rep < 2 && !rc; ++rep) {
            (void)hipEventRecord(e0, stream);
            if ((rc = gemm_h2_launch(c, p, stream))) break;
            if ((rc = gemm_h2_launch(c, p, stream))) break;
            (void)hipEventRecord(e1, stream);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = 1; break; }
            float t = 0.f;
            (void)hipEventElapsedTime(&t, e0, e1);
            ms = t < ms ? t : ms;
        }
        if (!rc && (best < 0 || ms < best_ms)) { best = c; best_ms = ms; }
    }
    g_prof_mask = saved_mask;
    g_event_pool.push_back(e0);
    g_event_pool.push_back(e1);
    if (rc) return rc;
    D4_REQUIRE(best >= 0, "gemm_h2: no configuration for M=%d N=%d K=%d flags=%d", p.M, p.N, p.K, p.flags);
    if (getenv("D4_GEMM_LOG"))
        fprintf(stderr, "[d4 gemm_h2] tuned M %6d N %5d K %5d batch %2d flags %3d -> %s (%.1f us)\n", p.M, p.N, p.K, p.batch, p.flags, gemm_h2_config_name(best), 500.f * best_ms);
    *best_out = best;
    return 0;
}

static int gemm_v4(const GemmArgs& p, hipStream_t stream) {
    const bool tune_on = tune_mode() != 0;
    const int nb = p.batch > 0 ? p.batch : 1;
    if (g_forced_cfg >= 400 && gemm_h2_config_valid(g_forced_cfg - 400, p)) return launch_v4(g_forced_cfg - 400, p, stream);
    const TuneKey key{p.M, p.N, p.K, p.flags, p.batch};
    tune_cache_load();
    auto it = g_tuned4.find(key);
    if (it != g_tuned4.end() && gemm_h2_config_valid(it->second, p)) return launch_v4(it->second, p, stream);
    const bool idempotent = !(p.flags & GEMM_ACCUMULATE) && p.R != p.C && p.A != p.C;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    if (!tune_on || !idempotent || cap != hipStreamCaptureStatusNone || 2.0 * p.M * p.N * p.K * nb < 1e8)
        return launch_v4(gemm_h2_heuristic(p), p, stream);
    D4_TUNE_STRICT_CHECK(p, nb);
    int best = 0;
    if (int rc = autotune_v4(p, stream, &best)) return rc;
    g_tuned4[key] = best;
    tune_cache_append(key, 400 + best);
    return launch_v4(best, p, stream);
}

// Which calls of an fp16x2 engine the family takes is a RULE on the call's shape (never a timing): the families differ in their bits.  Measured on
// MI355X (tools/gemm_h2_bench.py, profiles/r05c_*): x1.2-2.0 the f32-input MFMA kernels wherever a launch has a few hundred rows, N >= 256 and a k
// loop of some length; level or behind on launches of under ~1.2 G multiply-adds (the 512 x 512 / 256 x 512 projections at 3584 rows) and at
// K < 128; few-row calls stay on the few-row kernel.
bool gemm_h2_takes(const GemmArgs& p) {
    if (!p.Wb || p.wplane <= 0 || !p.wscale || !gemm_h2_applicable(p) || gemm_skinny_applicable(p)) return false;
    if (g_forced_cfg >= 0 && g_forced_cfg < 400) return false;       // a forced tile of another family
    if (g_forced_cfg >= 400) return true;
    return p.M >= 256 && p.K >= 128 && p.N >= 64 && (double)p.M * p.N * p.K * (p.batch > 0 ? p.batch : 1) >= 1.2e9;
}

// Two independent non-transposed fp32 GEMMs (the attention pool's query and key projections): ONE grid on the LDS-DMA family when both calls
// would have run there anyway (same family, same k order, same bits as the two separate launches), else one after the other.  The tile
// configuration is the one chosen for the larger problem.
int gemm_pair(const GemmArgs& a_in, const GemmArgs& b_in, hipStream_t stream) {
    std::lock_guard<std::recursive_mutex> lock(gemm_mutex());
    GemmArgs a = a_in, b = b_in;
    const bool fp32_rule = (!a.Wb || (a.wplane > 0 && a.N < 2048)) && (!b.Wb || (b.wplane > 0 && b.N < 2048)) &&     // neither goes to a bf16 / split-operand kernel
                           !gemm_h2_takes(a) && !gemm_h2_takes(b);
    if (fp32_rule && g_forced_cfg < 0 && a.M > 0 && b.M > 0 && a.K == b.K && use_v2(a) && use_v2(b) && !gemm_skinny_applicable(a) &&
        !gemm_skinny_applicable(b) && gemm2_pair_applicable(a, b)) {
        a.Wb = nullptr; a.wplane = 0; a.wscale = nullptr; b.Wb = nullptr; b.wplane = 0; b.wscale = nullptr;
        const GemmArgs& big = (double)a.M * a.N >= (double)b.M * b.N ? a : b;
        tune_cache_load();
        auto it = g_tuned2.find(TuneKey{big.M, big.N, big.K, big.flags, big.batch});
        const int c = it != g_tuned2.end() ? it->second : heuristic_v2(big);
        if (gemm2_pair_config_ok(c) && gemm2_config_valid(c, a) && gemm2_config_valid(c, b)) {
            const int cls = N_TILE_CFG + c;
            const bool timed = ((g_prof_mask >> cls) & 1) && (g_prof_tick++ % g_prof_stride) == 0;
            if (!timed) return gemm2_pair_launch(c, a, b, stream);
            ProfRec rec{};
            rec.a = prof_event(); rec.b = prof_event(); rec.cls = cls;
            rec.M = a.M + b.M; rec.N = big.N; rec.K = big.K; rec.flags = big.flags; rec.batch = 2;
            gemm2_config_tile(c, &rec.bm, &rec.bn);
            rec.flops = 2.0 * a.M * a.N * a.K + 2.0 * b.M * b.N * b.K;
            if (int rc = gemm2_pair_launch(c, a, b, stream, rec.a, rec.b)) return rc;
            g_prof.push_back(rec);
            return 0;
        }
    }
    if (int rc = gemm(a_in, stream)) return rc;
    return gemm(b_in, stream);
}

// The dispatcher's process-global state (tile choices per shape, the tuning-cache file, the profiling log and its event pool, the static
// per-instantiation attribute flags) is guarded by ONE recursive mutex taken at this entry: ctypes releases the GIL, and two engines may be
// driven from two host threads (tools/two_stream_rollout.py).  Launches are asynchronous, so the lock is held for microseconds except
// while a shape is being timed for the first time.
static std::recursive_mutex g_gemm_mu;
static std::recursive_mutex& gemm_mutex() { return g_gemm_mu; }

int gemm(const GemmArgs& p, hipStream_t stream) {
    std::lock_guard<std::recursive_mutex> lock(g_gemm_mu);
    D4_REQUIRE(p.M >= 0 && p.N > 0 && p.K > 0, "gemm: bad sizes M=%d N=%d K=%d", p.M, p.N, p.K);
    if (p.M == 0) return 0;
    const bool ta = p.flags & GEMM_TRANS_A, tb = p.flags & GEMM_TRANS_B;
    D4_REQUIRE((p.lda % 4) == 0 && (p.ldw % 4) == 0, "gemm: lda/ldw must be multiples of 4 (got %d, %d)", p.lda, p.ldw);
    D4_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0, "gemm: operands must be 16-byte aligned");
    D4_REQUIRE(!((p.flags & GEMM_RMS_ROWSCALE) && ta), "gemm: rms rowscale needs a non-transposed A");
    D4_REQUIRE(!((p.flags & GEMM_SWIGLU) && (p.N % 64) != 0), "gemm: swiglu needs N %% 64 == 0 (packed pairs)");
    D4_REQUIRE(!((p.flags & GEMM_SWIGLU) && (ta || tb)), "gemm: swiglu epilogue is forward-only");
    // fp16x2 engine mode (two fp16 planes of W + row scales, gemm_h2.hip): the rule gemm_h2_takes names the calls; every other call drops the
    // planes and runs on the f32-input kernels
    if (p.Wb && p.wplane > 0 && p.wscale) {
        if (gemm_h2_takes(p)) return gemm_v4(p, stream);
        GemmArgs q = p;
        q.Wb = nullptr; q.wplane = 0; q.wscale = nullptr; q.aexp = nullptr;
        return gemm(q, stream);
    }
    // split-operand fp32 (three bf16 planes of W): few-row calls stay on the few-row kernel (a rule on the shape, like every family choice)
    if (p.Wb && p.wplane > 0) {
        // Which calls take it is a RULE on the call's shape (never a timing).  Measured on MI355X (tools/gemm_x3_bench.py,
        // profiles/r02_gemm_split_operands.txt): it wins where a launch has enough wide tiles to hide its heavier staging — the SiLU-GLU
        // input projections (N = 2 x 1376 ... 5504: 1.2-1.35x the f32-input MFMA kernels) and other N >= 2048 projections (level to 1.3x) —
        // and loses on the N <= 1552 shapes (0.8-0.95x; with the whole rollout as the clock +3 .. +6 ms: profiles/r05b_x3_every_call_ab.txt).
        // Every such call of >= 1024 rows runs on the PERSISTENT form (gemm_x3sk.hip: whole rounds as plain tiles, a last round that is at most
        // half full as 128 x 64 half tiles — bit-identical to the plain kernel; same-box A/B pairs, profiles/r03l_ab_late_changes.txt: 188.8 vs
        // 191.8-192.4 (half-tile calls only) vs 199.3-200.0 ms (never) per step on a mid-speed box, level on the fastest one).
        const bool wide = ((p.flags & GEMM_SWIGLU) || p.N >= 2048) && !gemm_skinny_applicable(p);
        if (wide && g_forced_cfg < 0 && (gemm_x3sk_rule(p) || (gemm_x3sk_applicable(p) && p.M >= 1024))) return launch_v3sk(p, stream);
        if (wide && p.M >= 256 && gemm_x3_applicable(p)) return gemm_v3(p, stream);
        GemmArgs q = p;
        q.Wb = nullptr; q.wplane = 0;
        return gemm(q, stream);
    }
    if (p.Wb && p.Ab && gemm_bf16a_applicable(p)) return gemm_bf16a(p, stream);     // bf16 engine, the activation has a bf16 image: both operands by LDS-DMA
    if (p.Wb) return gemm_bf16(p, stream);
    // test hook: 100 + c forces configuration c of the second family, 0 .. N_TILE_CFG-1 a configuration of this one
    if (g_forced_cfg >= 100 && g_forced_cfg != 199 && gemm2_config_valid(g_forced_cfg - 100, p)) return launch_v2(g_forced_cfg - 100, p, stream);
    if (gemm_skinny_applicable(p)) return gemm_skinny(p, stream);
    // few rows x long K (the heads' hidden layers at rollout batch): the contraction cut four ways inside the workgroup — a rule on the shape
    if ((g_forced_cfg < 0 && gemm2_ksplit_rule(p)) || (g_forced_cfg == 199 && gemm2_ksplit_applicable(p))) return launch_v2ks(p, stream);      // (199: test hook, any shape it can run)
    if ((g_forced_cfg < 0 || g_forced_cfg >= 300) && use_v2(p)) return gemm_v2(p, stream);      // (300 + c forces a tile of the split-operand family only)
    if (!ta && !tb) return launch_t<false, false>(p, stream);
    if (!ta && tb) return launch_t<false, true>(p, stream);
    if (ta && tb) return launch_t<true, true>(p, stream);
    return launch_t<true, false>(p, stream);
}

}  // namespace d4
