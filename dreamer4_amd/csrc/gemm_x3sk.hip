// Persistent form of the split-operand fp32 GEMM (gemm_x3.hip, 128 x 128 tiles on 8 waves): ONE workgroup per CU walks a static list of work items.
//
// Why: this tile needs 120 KB of LDS, so exactly one workgroup fits a CU and a launch proceeds in rounds of 256 tiles
// (profiles/r03g_x3_quantisation_sweep.txt is a staircase in steps of 256 tiles; a k-tile step takes ~1.8 us).  The cfg-2 shapes sit badly on
// that staircase: the SiLU-GLU input projection is 616 tiles = 2.4 rounds and pays 3.
//
// IN THE ENGINE every split-operand call of >= 1024 rows runs here (gemm.hip): whole tiles walked by 256 long-lived workgroups are
// bit-identical to gemm_x3_kernel and, measured in situ on mid-speed boxes, leave the FOLLOWING kernels ~3-4 % faster than 476-660 short-lived
// 120 KB workgroups do (profiles/r03l_ab_late_changes.txt: 188 / 191 / 199 ms per step for all calls / half-tile calls only / none).
//
// HALF TILES: when the tiles of the last partial round number at most half the workgroups, they run as 2 R work items of 128 x 64 (the 8 waves as
// 4 x 2 with 32 x 32 wave tiles; a SiLU-GLU column group is one such item: its value wave and its gate wave meet through LDS in the epilogue), one
// per workgroup, after the whole rounds.  Nothing is exchanged between workgroups and every element keeps its k order: bit-identical to
// gemm_x3_kernel, 2.4 rounds -> 2 rounds + one shorter one (a half tile takes ~0.88 of a whole tile's time — the k-tile step is bound by the
// per-wave split / LDS / barrier chain, not by the MFMAs: 83.3 vs 86.5 us on the SiLU-GLU input projection).
//
// (Rounds 3-4 also carried a k-cut of the last round — "stream-K" with an integer split, partial tiles exchanged through agent-scope atomics.  Correct on
// every epilogue but a measured no-go: a 64 KB slice takes 10-20 us to cross between workgroups, which is what the cut saves at K = 512
// (profiles/r03i_gemm_x3sk.txt).  Removed in round 5 together with its workspace allocation.)
#include "common.h"
#include <hip/hip_ext.h>
#include "kernels.h"
#include <type_traits>

namespace d4 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// (file-local names carry an sk_ / SK_ prefix; the kernel keeps the plain name d4::gemm_x3sk_kernel for the profiler)

constexpr int SK_BM = 128, SK_BN = 128, SK_WGM = 4, SK_WGN = 2, SK_D = 3;
constexpr int SK_BK = 32, SK_LD = SK_BK + 8, SK_NT = SK_WGM * SK_WGN * 64;
constexpr int SK_G = SK_BK / 8;
constexpr int SK_APL = SK_BM * SK_LD, SK_BPL = SK_BN * SK_LD;   // one plane of one buffer (elements); half tiles use the first 64 rows of a B plane
constexpr size_t SK_LDS = (size_t)(2 * 3 * (SK_BM + SK_BN) * SK_LD) * 2 + SK_BM * sizeof(float) + 16;
static_assert(SK_BM * SK_G == SK_NT && SK_BN * SK_G == SK_NT, "one 8-element group of each operand per thread and k-tile");

struct SkArgs {
    GemmArgs p;
    int P, F, R;              // grid; whole rounds; tiles of the last partial round
    int half;                 // 1: the remaining tiles run as 2 R half tiles of 128 x 64
};

__device__ __forceinline__ void split3(float a, __bf16& h1, __bf16& h2, __bf16& h3) {
    h1 = (__bf16)a;
    const float r = a - (float)h1;
    h2 = (__bf16)r;
    h3 = (__bf16)(r - (float)h2);
}


// One work item on the calling workgroup: rows bm0 .. bm0 + 127, columns bn0 .. bn0 + 64 TNV - 1, all of K.
// TNV = 2: a 128 x 128 tile (wave tile 32 x 64); TNV = 1: a half tile of 128 x 64 (wave tile 32 x 32).
template <int TNV>
__device__ __forceinline__ void sk_item(const SkArgs& s, const int bm0, const int bn0, __bf16* As, __bf16* Bs, float* rowscale_s) {
    const GemmArgs& p = s.p;
    const int kt0 = 0, kt1 = p.K / SK_BK;
    constexpr int BNV = 64 * TNV;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / SK_WGN, wn = wave % SK_WGN;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const __bf16* Wb = reinterpret_cast<const __bf16*>(p.Wb);
    const int rowsA = min(SK_BM, p.M - bm0), rowsB = min(BNV, p.N - bn0);
    auto uniform_rsrc = [](const void* base, int64_t bytes) {
        const uint64_t bb = reinterpret_cast<uint64_t>(base);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)bb);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(bb >> 32));
        const int nb = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7FFFFFFF ? bytes : 0x7FFFFFFF));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
    };
    // rows past the matrix edge fall outside num_records and read as zeros
    const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc(p.A + (int64_t)bm0 * p.lda, ((int64_t)(rowsA - 1) * p.lda + p.K) * 4);
    const int64_t wbytes = ((int64_t)(rowsB - 1) * p.ldw + p.K) * 2;
    const __amdgpu_buffer_rsrc_t rsB0 = uniform_rsrc(Wb + (int64_t)bn0 * p.ldw, wbytes);
    const __amdgpu_buffer_rsrc_t rsB1 = uniform_rsrc(Wb + p.wplane + (int64_t)bn0 * p.ldw, wbytes);
    const __amdgpu_buffer_rsrc_t rsB2 = uniform_rsrc(Wb + 2 * p.wplane + (int64_t)bn0 * p.ldw, wbytes);

    // register staging as gemm_x3_kernel<128, 128, 4, 2, D = 3, two LDS buffers>; k runs over this item's k-tiles only.  Half tiles have 64
    // W rows for 128 staging rows: the upper half of the workgroup loads the same rows again (unconditional loads keep the counted waits
    // exact) and does not store them.
    f32x4 ra[SK_D][2];
    f32x4 rb[SK_D][3];
    float ssq0 = 0.f, ssq1 = 0.f;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    const int srow = tid / SK_G, scol = (tid % SK_G) * 8;
    const int browB = TNV == 2 ? srow : (srow & 63);
    const bool b_store = TNV == 2 || srow < 64;
    const uint32_t offA = (uint32_t)((srow * p.lda + scol) * 4), offB = (uint32_t)((browB * p.ldw + scol) * 2);
    auto load_tile = [&](auto set_tag, int k0) {
        constexpr int S = decltype(set_tag)::value;
        ra[S][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, offA + (uint32_t)k0 * 4, 0, 0));
        ra[S][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, offA + (uint32_t)k0 * 4 + 16, 0, 0));
        rb[S][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB0, offB + (uint32_t)k0 * 2, 0, 0));
        rb[S][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB1, offB + (uint32_t)k0 * 2, 0, 0));
        rb[S][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB2, offB + (uint32_t)k0 * 2, 0, 0));
    };
    auto store_tile = [&](auto set_tag, int buf) {
        constexpr int S = decltype(set_tag)::value;
        __bf16* as = As + buf * 3 * SK_APL + srow * SK_LD + scol;
        __bf16* bs = Bs + buf * 3 * SK_BPL + browB * SK_LD + scol;
        if (b_store) {
            *reinterpret_cast<f32x4*>(bs) = rb[S][0];
            *reinterpret_cast<f32x4*>(bs + SK_BPL) = rb[S][1];
            *reinterpret_cast<f32x4*>(bs + 2 * SK_BPL) = rb[S][2];
        }
        const f32x4 v0 = ra[S][0], v1 = ra[S][1];
        bf16x8 o1, o2, o3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 h1, h2, h3;
            split3(v0[e], h1, h2, h3); o1[e] = h1; o2[e] = h2; o3[e] = h3;
            split3(v1[e], h1, h2, h3); o1[e + 4] = h1; o2[e + 4] = h2; o3[e + 4] = h3;
        }
        *reinterpret_cast<bf16x8*>(as) = o1;
        *reinterpret_cast<bf16x8*>(as + SK_APL) = o2;
        *reinterpret_cast<bf16x8*>(as + 2 * SK_APL) = o3;
        ssq0 = ssq0 + __builtin_fmaf(v0[3], v0[3], __builtin_fmaf(v0[2], v0[2], __builtin_fmaf(v0[1], v0[1], v0[0] * v0[0])));
        ssq1 = ssq1 + __builtin_fmaf(v1[3], v1[3], __builtin_fmaf(v1[2], v1[2], __builtin_fmaf(v1[1], v1[1], v1[0] * v1[0])));
    };

    f32x16 hi[TNV], lo[TNV];
#pragma unroll
    for (int j = 0; j < TNV; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { hi[j][e] = 0.f; lo[j][e] = 0.f; }

    auto mma = [&](int buf, int ks) {
        const __bf16* as = As + buf * 3 * SK_APL + (wm * 32 + lrow) * SK_LD + lhalf * 8 + ks * 16;
        const __bf16* bs = Bs + buf * 3 * SK_BPL + (wn * TNV * 32 + lrow) * SK_LD + lhalf * 8 + ks * 16;
        bf16x8 af[3], bf[3][TNV];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            af[pl] = *reinterpret_cast<const bf16x8*>(as + pl * SK_APL);
#pragma unroll
            for (int j = 0; j < TNV; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(bs + pl * SK_BPL + j * 32 * SK_LD);
        }
#define D4_SK_TERM(PA, PB, ACC) \
    _Pragma("unroll") for (int j = 0; j < TNV; ++j) ACC[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA], bf[PB][j], ACC[j], 0, 0, 0);
        D4_SK_TERM(2, 0, lo)
        D4_SK_TERM(0, 0, hi)
        D4_SK_TERM(1, 1, lo)
        D4_SK_TERM(0, 2, lo)
        D4_SK_TERM(1, 0, lo)
        D4_SK_TERM(0, 1, lo)
#undef D4_SK_TERM
    };

    const int nk = kt1 - kt0, kbeg = kt0 * SK_BK;
    const int klast = kbeg + (nk - 1) * SK_BK;
    {
        load_tile(S0{}, kbeg);
        load_tile(S1{}, min(kbeg + SK_BK, klast));
        load_tile(S2{}, min(kbeg + 2 * SK_BK, klast));
        store_tile(S0{}, 0);
        __syncthreads();
        auto k_tile = [&](int kt, auto set_tag, auto store_tag) {
            constexpr int S = decltype(set_tag)::value;
            const int buf = kt & 1;
            load_tile(set_tag, min(kbeg + (kt + SK_D) * SK_BK, klast));
            const bool store = decltype(store_tag)::value || kt + 1 < nk;
            mma(buf, 0);
            if (store) store_tile(std::integral_constant<int, (S + 1) % SK_D>{}, buf ^ 1);
            mma(buf, 1);
            if constexpr (decltype(store_tag)::value) {
                constexpr int NMFMA = 2 * 6 * TNV;
#pragma unroll
                for (int i = 0; i < NMFMA; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 4 * (3 - TNV), 0); // 4 (8: half tiles have half the MFMAs for the same split) VALU
                    __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);             // 1 DS
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);             // 1 VMEM read
                }
            }
            __syncthreads();
        };
        using Always = std::true_type;
        using Check = std::false_type;
        const int nfull = (nk - 1) / SK_D;
        int kt = 0;
        for (int t = 0; t < nfull; ++t, kt += SK_D) {
            k_tile(kt, S0{}, Always{});
            k_tile(kt + 1, S1{}, Always{});
            k_tile(kt + 2, S2{}, Always{});
        }
        k_tile(kt, S0{}, Check{});
        if (kt + 1 < nk) k_tile(kt + 1, S1{}, Check{});
        if (kt + 2 < nk) k_tile(kt + 2, S2{}, Check{});
    }

    // ---- this item's sums: hi + lo per element, the row's sum of squares on the first lane of each 4-lane row group
#pragma unroll
    for (int j = 0; j < TNV; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) hi[j][e] += lo[j][e];
    const bool rms = (p.flags & GEMM_RMS_ROWSCALE) != 0;
    float rsum = 0.f;
    if (rms) {
        rsum = ssq0 + ssq1;                        // chunks (2g) + (2g + 1)
        rsum += dpp_f<0xB1>(rsum);                 // ((0+1)+(2+3)), ((4+5)+(6+7))
        rsum += dpp_f<0x4E>(rsum);                 // the four lanes of a row
    }
    const bool row_lane = (tid % SK_G) == 0;

    if (rms) {
        if (row_lane) rowscale_s[srow] = rsqrtf(rsum / (float)p.K + p.rms_eps);
        __syncthreads();
    }

    // ---- epilogue (as gemm_x3_kernel): C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    if (TNV == 1 && swiglu) {
        // half tile = one packed column group: wave column 0 holds its 32 values, wave column 1 their 32 gates; silu(gate) crosses through LDS
        // (the A buffers are free: the k-loop ended on a barrier)
        float* xs = reinterpret_cast<float*>(As);                  // [128][33]
        const int gn = bn0 + lrow;                                 // packed column of the value
        if (wn == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int lr = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
                float gate = hi[0][e] * (rms ? rowscale_s[lr] : 1.f);
                if (p.bias) gate += p.bias[gn + 32];
                xs[lr * 33 + lrow] = siluf(gate);
            }
        }
        __syncthreads();
        if (wn == 0 && gn < p.N) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int lr = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
                const int gm = bm0 + lr;
                if (gm >= p.M) continue;
                float val = hi[0][e] * (rms ? rowscale_s[lr] : 1.f);
                if (p.bias) val += p.bias[gn];
                const int on = (gn / 64) * 32 + (gn % 64);
                p.C[(int64_t)gm * p.ldc + on] = val * xs[lr * 33 + lrow];
            }
        }
        __syncthreads();
        return;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int lr = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
        const int gm = bm0 + lr;
        if (gm >= p.M) continue;
        const float rs = rms ? rowscale_s[lr] : 1.f;
        if (swiglu) {
            if constexpr (TNV == 2) {
                const int gn = bn0 + wn * 64 + lrow;               // packed column of the value
                if (gn >= p.N) continue;
                float val = hi[0][e] * rs, gate = hi[1][e] * rs;
                if (p.bias) { val += p.bias[gn]; gate += p.bias[gn + 32]; }
                const int on = (gn / 64) * 32 + (gn % 64);
                p.C[(int64_t)gm * p.ldc + on] = val * siluf(gate);
            }
        } else {
#pragma unroll
            for (int j = 0; j < TNV; ++j) {
                const int gn = bn0 + wn * TNV * 32 + j * 32 + lrow;
                if (gn >= p.N) continue;
                float v = hi[j][e] * rs;
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
    __syncthreads();                                   // rowscale_s / the LDS tiles are rewritten by the next item
}

__global__ __launch_bounds__(SK_NT, 2) void gemm_x3sk_kernel(SkArgs s) {
    const GemmArgs& p = s.p;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __bf16* As = reinterpret_cast<__bf16*>(smem_raw);                 // [2][3][BM][LD]
    __bf16* Bs = As + 2 * 3 * SK_APL;                                 // [2][3][BN][LD]
    float* rowscale_s = reinterpret_cast<float*>(Bs + 2 * 3 * SK_BPL);   // [BM]

    const int b = blockIdx.x;
    const int nbn = (p.N + SK_BN - 1) / SK_BN, nbm = (p.M + SK_BM - 1) / SK_BM;
    const int nk_all = p.K / SK_BK;
    // tile index -> (tm, tn): the XCD-aware banded walk of gemm_x3_kernel (workgroup b sits on XCD b % 8 and b + i * P keeps it)
    auto tile_of = [&](int tile, int& tm, int& tn) {
        int bid = tile;
        const int nblk = nbm * nbn, nx = 8;
        const int q = nblk / nx, r = nblk % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
        constexpr int RB = 4;
        const int band = bid / (RB * nbn), j = bid % (RB * nbn);
        const int rows = min(RB, nbm - band * RB);
        tm = band * RB + j % rows; tn = j / rows;
    };
    int tm, tn;
    if (s.half) {
        // whole rounds, then the remaining tiles as 2 R half tiles: item h = (remaining tile h / 2, column half h & 1) on workgroup h % P
        for (int i = 0; i < s.F; ++i) {
            tile_of(b + i * s.P, tm, tn);
            sk_item<2>(s, tm * SK_BM, tn * SK_BN, As, Bs, rowscale_s);
        }
        // remaining tile j's two halves go to workgroups j and R8 + j (R8 = R rounded up to 8): both sit on XCD j % 8, the XCD whose L2 the banded
        // walk gave tile F P + j's neighbours to (with h / 2 on workgroup h the halves landed on other XCDs: +18 MB of fabric reads per launch)
        const int R8 = (s.R + 7) & ~7;
        if (R8 + s.R <= s.P) {
            const int j = b < s.R ? b : b - R8, half = b < s.R ? 0 : 1;
            if (j >= 0 && j < s.R && (b < s.R || b >= R8)) {
                tile_of(s.F * s.P + j, tm, tn);
                const int bn0 = tn * SK_BN + half * 64;
                if (bn0 < p.N) sk_item<1>(s, tm * SK_BM, bn0, As, Bs, rowscale_s);
            }
            return;
        }
        for (int h = b; h < 2 * s.R; h += s.P) {
            tile_of(s.F * s.P + h / 2, tm, tn);
            const int bn0 = tn * SK_BN + (h & 1) * 64;
            if (bn0 < p.N) sk_item<1>(s, tm * SK_BM, bn0, As, Bs, rowscale_s);
        }
        return;
    }
    // whole tiles only: F rounds, then the last partial round's R tiles on the first R workgroups
    for (int i = 0; i < s.F + (b < s.R ? 1 : 0); ++i) {
        tile_of(b + i * s.P, tm, tn);
        sk_item<2>(s, tm * SK_BM, tn * SK_BN, As, Bs, rowscale_s);
    }
}

static std::atomic<int> g_sk_cus[64];            // per device id (a process may drive several devices); 0 = not queried yet

static int sk_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::atomic<int>& c = g_sk_cus[dev & 63];
    int n = c.load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        c.store(n, std::memory_order_relaxed);
    }
    return n;
}


const char* gemm_x3sk_name() { return "gemm_x3sk_kernel"; }

bool gemm_x3sk_applicable(const GemmArgs& p) {
    return gemm_x3_applicable(p) && p.batch <= 1 && p.M > 0 && (!(p.flags & GEMM_SWIGLU) || (p.N % 64) == 0);
}

// Half tiles: the remaining tiles (at most half the workgroups) run as 2 R items of 128 x 64 after the F whole rounds
bool gemm_x3sk_half_plan(const GemmArgs& p, int* F, int* R, int* P_out) {
    const int P = sk_cus();
    const int T = cdiv(p.M, SK_BM) * cdiv(p.N, SK_BN);
    const int f = T / P, r = T - f * P;
    *F = f; *R = r;
    if (P_out) *P_out = P;
    return r > 0 && 2 * r <= P;
}

// The shape rule (never a timing): the persistent form with half tiles runs a call that has at least one whole round and whose last
// round is at most half full — the 128 x 64 items then take ~0.7 of a round (measured, profiles/r03i_gemm_x3sk.txt) instead of a whole one.
// Tall N = 256 products stream their A operand from HBM and sit on the f32-input kernels (N >= 512 here).
bool gemm_x3sk_rule(const GemmArgs& p) {
    if (!gemm_x3sk_applicable(p) || p.M < 1024 || p.N < 512) return false;
    int F, R, P;
    return gemm_x3sk_half_plan(p, &F, &R, &P) && F >= 1 && F <= 3;
}

// half tiles where the plan allows them, else whole tiles only: nothing crosses between workgroups
int gemm_x3sk_launch(const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    D4_REQUIRE(gemm_x3sk_applicable(p), "gemm_x3sk: call not supported (M=%d N=%d K=%d flags=%d batch=%d)", p.M, p.N, p.K, p.flags, p.batch);
    SkArgs a;
    a.p = p;
    a.half = gemm_x3sk_half_plan(p, &a.F, &a.R, &a.P) ? 1 : 0;
    const int T = a.F * a.P + a.R;
    if (T < a.P && !a.half) a.P = T;                   // a small call: one tile per workgroup, nothing persistent
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3sk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SK_LDS));
        attr_set.done();
    }
    const dim3 grid(a.P), block(SK_NT);
    if (ea) hipExtLaunchKernelGGL(gemm_x3sk_kernel, grid, block, (uint32_t)SK_LDS, stream, ea, eb, 0, a);
    else hipLaunchKernelGGL(gemm_x3sk_kernel, grid, block, SK_LDS, stream, a);
    D4_LAUNCH_CHECK();
    return 0;
}

}  // namespace d4
