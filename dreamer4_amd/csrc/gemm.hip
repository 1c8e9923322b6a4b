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
// Tiling: BM x BN block tile, BK = 32, 256 threads = 4 waves (WGM x WGN), each wave owns
// TM x TN MFMA 32x32 sub-tiles.  LDS tiles are row-major [rows][BK + 4] (the +4 keeps 16-byte
// alignment and spreads ds_read_b128 over the 64 banks), double-buffered, one barrier per k-tile;
// the next tile's global loads are issued before the current tile's MFMAs (register staging).
// The MFMA consumes k in the order lanes<32: {8q+e}, lanes>=32: {8q+4+e} so that each lane's
// operand for four consecutive k-steps is one ds_read_b128 (summation order inside a k-tile is a
// fixed permutation; results are deterministic).
#include "common.h"
#include "kernels.h"
#include <math.h>
#include <stdlib.h>
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

    f32x4 ra[A_F4], rb[B_F4];
    float ssq[A_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) ssq[i] = 0.f;

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
    auto load_tile = [&](int k0) {
        load_operand(ra, rsA, p.lda, rowsA, k0, std::integral_constant<bool, TA>{}, std::integral_constant<int, BM>{});
        load_operand(rb, rsB, p.ldw, rowsB, k0, std::integral_constant<bool, TB>{}, std::integral_constant<int, BN>{});
    };

    auto store_tile = [&](int buf) {
        float* as = As + buf * BM * LDS_LD;
        float* bs = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            int idx = tid + i * NT;
            if constexpr (!TA) {
                int r = idx / RF4, c = (idx % RF4) * 4;
                *reinterpret_cast<f32x4*>(as + r * LDS_LD + c) = ra[i];
                ssq[i] += ra[i][0] * ra[i][0] + ra[i][1] * ra[i][1] + ra[i][2] * ra[i][2] + ra[i][3] * ra[i][3];
            } else {
                int kk = idx / (BM / 4), c = (idx % (BM / 4)) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) as[(c + e) * LDS_LD + kk] = ra[i][e];
            }
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            int idx = tid + i * NT;
            if constexpr (!TB) {
                int r = idx / RF4, c = (idx % RF4) * 4;
                *reinterpret_cast<f32x4*>(bs + r * LDS_LD + c) = rb[i];
            } else {
                int kk = idx / (BN / 4), c = (idx % (BN / 4)) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) bs[(c + e) * LDS_LD + kk] = rb[i][e];
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

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int lrow = lane & 31;
    const int lhalf = lane >> 5;

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);

        const float* as = As + cur * BM * LDS_LD + (wm * TM * 32 + lrow) * LDS_LD + lhalf * 4;
        const float* bs = Bs + cur * BN * LDS_LD + (wn * TN * 32 + lrow) * LDS_LD + lhalf * 4;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
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

        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
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
                float s = ssq[i];
                s += dpp_f<0xB1>(s);
                s += dpp_f<0x4E>(s);
                s += dpp_f<0x141>(s);   // RF4 (8 or 16) consecutive lanes share one row
                if (RF4 == 16) s += dpp_f<0x140>(s);
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
struct ProfRec { hipEvent_t a, b; int cls; double flops; };
static int g_prof_mask = 0;            // bit c set: time launches of tile class c
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_event_pool;

static hipEvent_t prof_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

bool gemm_profile_active() { return g_prof_mask != 0; }

int gemm_profile_enable(int mask) {
    g_prof_mask = mask;
    return 0;
}

// Sums elapsed time / algorithmic flops / launch count per tile class (0: 128x128, 1: 64x128, 2: 64x64) and clears the log.
int gemm_profile_read(double* ms, double* flops, int64_t* count, int nclass) {
    for (int i = 0; i < nclass; ++i) { ms[i] = 0; flops[i] = 0; count[i] = 0; }
    for (auto& r : g_prof) {
        D4_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        D4_HIP(hipEventElapsedTime(&t, r.a, r.b));
        if (r.cls < nclass) { ms[r.cls] += t; flops[r.cls] += r.flops; count[r.cls] += 1; }
        g_event_pool.push_back(r.a);
        g_event_pool.push_back(r.b);
    }
    g_prof.clear();
    return 0;
}

template <int BM, int BN> struct TileClass { static constexpr int value = BM == 128 ? 0 : (BN == 128 ? 1 : 2); };

static inline double tile_cost(const GemmArgs& p, int BM, int BN, int slots) {
    // co-resident blocks share a CU's matrix pipes: a CU's time ~ (blocks it hosts) x (tile area)
    const double blocks = (double)cdiv(p.M, BM) * cdiv(p.N, BN) * (p.batch > 0 ? p.batch : 1);
    return ceil(blocks / slots) * BM * BN;
}

template <int BM, int BN, int WGM, int WGN, int BK, int KS, bool TA, bool TB>
static int launch_cfg(const GemmArgs& p, hipStream_t stream) {
    constexpr int LDS_LD = BK + 4;
    const int nblk = cdiv(p.M, BM) * cdiv(p.N, BN);
    const size_t lds = (size_t)(2 * BM * LDS_LD + 2 * BN * LDS_LD + BM) * sizeof(float);
    const bool ktail = (p.K % BK) != 0;
    auto k = ktail ? gemm_kernel<BM, BN, WGM, WGN, BK, KS, TA, TB, true> : gemm_kernel<BM, BN, WGM, WGN, BK, KS, TA, TB, false>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[ktail]) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[ktail] = true;
    }
    ProfRec rec{};
    const bool g_prof_on = (g_prof_mask >> TileClass<BM, BN>::value) & 1;
    if (g_prof_on) {
        rec.a = prof_event(); rec.b = prof_event(); rec.cls = TileClass<BM, BN>::value;
        rec.flops = p.algo_flops > 0 ? p.algo_flops : 2.0 * p.M * p.N * p.K * (p.batch > 0 ? p.batch : 1);
        hipEventRecord(rec.a, stream);
    }
    hipLaunchKernelGGL(k, dim3(nblk, p.batch > 0 ? p.batch : 1), dim3(WGM * WGN * 64 * KS), lds, stream, p);
    if (g_prof_on) { hipEventRecord(rec.b, stream); g_prof.push_back(rec); }
    D4_LAUNCH_CHECK();
    return 0;
}

template <bool TA, bool TB>
static int launch_t(const GemmArgs& p, hipStream_t stream) {
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    const int nb = p.batch > 0 ? p.batch : 1;
    const int64_t b128 = (int64_t)cdiv(p.M, 128) * cdiv(p.N, 128) * nb;
    // 256 CUs: prefer the big tile once it yields >= ~1.5 blocks per CU.
    // (long-K backward GEMMs, dW = dZ^T X with K = batch*time rows: one big tile per CU beats two rounds of small ones)
    if (b128 >= 384 || (b128 >= 192 && p.K >= 2048)) {
        // 8 waves (2 x 4, wave tile 64 x 32): two co-resident blocks put 4 waves on every SIMD at the same LDS
        // footprint as the 4-wave form -> +6..8 % on the K=512 projections (measured, scratch/gpu_gemm_bench.py)
        // (the SiLU-GLU epilogue pairs two N sub-tiles inside one wave -> 4 x 2 waves, wave tile 32 x 64)
        if (swiglu) return launch_cfg<128, 128, 4, 2, 32, 1, TA, TB>(p, stream);
        // 128 x 96 tiles when they quantise better onto the 256 CUs (e.g. the fused q|k|v projection, N = 1552:
        // 30 x 13 = 390 tiles of 128 x 128 leave half the CUs with one block and half with two; 30 x 17 = 510 fit once).
        // The 4-wave 128 x 96 form is ~20 % less efficient per flop than the 8-wave 128 x 128 one, hence the margin.
        if (!TA && !TB && tile_cost(p, 128, 96, 256) < 0.8 * tile_cost(p, 128, 128, 256)) return launch_cfg<128, 96, 4, 1, 32, 1, TA, TB>(p, stream);
        return launch_cfg<128, 128, 2, 4, 32, 1, TA, TB>(p, stream);
    }
    // (measured on MI355X, scratch/gpu_gemm_bench.py: BK = 64 and an intra-block split of the k-steps over 8 waves
    //  (KS = 2) change these small-tile shapes by < 2 % — they are bound by the ~5 us fixed cost per launch.)
    if (swiglu || (p.N > 64 && (int64_t)cdiv(p.M, 64) * cdiv(p.N, 128) * nb >= 256)) return launch_cfg<64, 128, 2, 2, 32, 1, TA, TB>(p, stream);
    return launch_cfg<64, 64, 2, 2, 32, 1, TA, TB>(p, stream);
}

int gemm(const GemmArgs& p, hipStream_t stream) {
    D4_REQUIRE(p.M >= 0 && p.N > 0 && p.K > 0, "gemm: bad sizes M=%d N=%d K=%d", p.M, p.N, p.K);
    if (p.M == 0) return 0;
    const bool ta = p.flags & GEMM_TRANS_A, tb = p.flags & GEMM_TRANS_B;
    D4_REQUIRE((p.lda % 4) == 0 && (p.ldw % 4) == 0, "gemm: lda/ldw must be multiples of 4 (got %d, %d)", p.lda, p.ldw);
    D4_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0, "gemm: operands must be 16-byte aligned");
    D4_REQUIRE(!((p.flags & GEMM_RMS_ROWSCALE) && ta), "gemm: rms rowscale needs a non-transposed A");
    D4_REQUIRE(!((p.flags & GEMM_SWIGLU) && (p.N % 64) != 0), "gemm: swiglu needs N %% 64 == 0 (packed pairs)");
    D4_REQUIRE(!((p.flags & GEMM_SWIGLU) && (ta || tb)), "gemm: swiglu epilogue is forward-only");
    if (!ta && !tb) return launch_t<false, false>(p, stream);
    if (!ta && tb) return launch_t<false, true>(p, stream);
    if (ta && tb) return launch_t<true, true>(p, stream);
    return launch_t<true, false>(p, stream);
}

}  // namespace d4
