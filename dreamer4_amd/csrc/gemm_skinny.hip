// Few-row GEMM: C[M][N] = epilogue(A[M][K] . W[N][K]^T) when M is so small that the tiled kernel would have fewer than 64
// tiles (a quarter of the CUs) — the small-batch decode regime (BASELINE config 4: one trajectory = 15 token rows per frame).
// There the tiled MFMA kernel is pure latency: one 64-row tile per block walks all of K serially (16 k-tiles x ~0.7 us) with the
// matrix pipes 75 % empty, while the real cost is streaming W once.
//
// Here a block owns 16 output columns x 16 rows and all of K; its 4 / 8 / 16 waves split K and each runs the 16x16x4 fp32
// MFMA (16 rows tall) with BOTH operands streamed straight from global memory: every W element is used once per row group and A
// is a few tens of KB that stays in L2, so neither touches LDS.  One 16-byte load per lane feeds four MFMA k-steps (the same
// permuted-k trick as gemm_kernel).  The partial tiles (and the rows' sums of squares for the folded RMSNorm) are combined
// through LDS in a fixed order (deterministic), then the same epilogue as gemm_kernel runs (1/rms row scale, bias, SiLU, SiLU-GLU
// pairing, residual, accumulate, compact copy).
// Selection is by shape only, never by timing, so a given call always takes the same arithmetic path.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

namespace d4 {

constexpr int SKN = 16;          // output columns per block
constexpr int SK_ROWS = 16;      // MFMA tile height
constexpr int SK_RED = SKN + 1;  // per (k-quarter, row): 16 column partials + the row's sum of squares

// NW waves split K; U k-steps of loads are issued together.  The launcher picks (NW, U) from K alone so that a wave issues all of
// its loads in one round where it can: at these sizes a launch is a chain of dependent memory round trips (~1.6 us each, weights
// stream from HBM / the memory-side cache), not bandwidth — tools/skinny_bench.py.
template <bool SWIGLU, int NW, int U>
__device__ __forceinline__ void skinny_body(GemmArgs& p, const int bx, const int by, const int bz, float* red) {
    constexpr int RB = SK_ROWS;                                  // rows per block (by = row group)
    const int m0 = by * RB;
    {   // strided batch (bz), as in gemm_kernel
        p.A += bz * p.strideA; p.W += bz * p.strideW; p.C += bz * p.strideC;
        if (p.R) p.R += bz * p.strideC;
    }
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int nl = lane & 15, kk = lane >> 4;

    // ---- this lane's W row (MFMA B operand: column nl of the block, k sub-slot kk)
    int n;
    if (SWIGLU) {       // packed pairs: 32 value columns then their 32 gate columns per group of 64; a block takes 8 + 8
        const int g = bx >> 2, c0 = (bx & 3) * 8;
        n = g * 64 + (nl < 8 ? c0 + nl : 32 + c0 + (nl - 8));
    } else {
        n = bx * SKN + nl;
    }
    const bool valid = n < p.N;
    const float* wrow = p.W + (int64_t)(valid ? n : 0) * p.ldw;
    // MFMA A operand: row nl (lane & 15) of the 16-row tile, same k sub-slot; rows past M read as zero
    const int am = m0 + nl;
    const bool arow_ok = am < p.M;
    const float* arow = p.A + (int64_t)(arow_ok ? am : 0) * p.lda;

    // the epilogue's own operands (bias, residual, accumulate) are fetched now, with the operand loads, not after the fold: at these
    // sizes every dependent memory round trip is a visible fraction of the launch
    float e_bias = 0.f, e_gate_bias = 0.f, e_res = 0.f, e_acc = 0.f;
    {
        const int c = tid & 15, m = m0 + (tid >> 4);
        if (tid < 256 && m < p.M) {
            if (SWIGLU) {
                const int nv = (bx >> 2) * 64 + (bx & 3) * 8 + c;
                if (c < 8 && nv < p.N && p.bias) { e_bias = p.bias[nv]; e_gate_bias = p.bias[nv + 32]; }
            } else {
                const int gn = bx * SKN + c;
                if (gn < p.N) {
                    if (p.bias) e_bias = p.bias[gn];
                    if (p.R) e_res = p.R[(int64_t)m * p.ldr + gn];
                    if (p.flags & GEMM_ACCUMULATE) e_acc = p.C[(int64_t)m * p.ldc + gn];
                }
            }
        }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float ssq = 0.f;
    const int steps = (p.K + 15) >> 4;                           // 16 k per step (4 MFMAs of k = 4)
    const int per = (steps + NW - 1) / NW;
    const int s1 = min((q + 1) * per, steps);
    for (int s0 = q * per; s0 < s1; s0 += U) {
        f32x4 w4[U], a4[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = 16 * (s0 + u) + 4 * kk;
            const bool in = s0 + u < s1 && k < p.K;
            w4[u] = (valid && in) ? *reinterpret_cast<const f32x4*>(wrow + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            a4[u] = (arow_ok && in) ? *reinterpret_cast<const f32x4*>(arow + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4 a = a4[u];
            ssq += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w4[u][e], acc, 0, 0, 0);
        }
    }
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg.  The sum of squares of row nl is spread over
    // the 4 k sub-slots of this wave: fold those with two lane swaps (xor 16, xor 32).
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(q * RB + 4 * kk + r) * SK_RED + nl] = acc[r];
    {
        float s = ssq;
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (kk == 0) red[(q * RB + nl) * SK_RED + SKN] = s;
    }
    __syncthreads();

    // ---- fold the NW k-slices in a fixed order; thread = (row, column)
    if (tid >= 256) return;
    const int c = tid & 15;
    const int ml = tid >> 4, m = m0 + ml;
    if (m >= p.M) return;
    auto fold = [&](int col) {
        float v = red[ml * SK_RED + col];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[(w * RB + ml) * SK_RED + col];
        return v;
    };
    const float rscale = (p.flags & GEMM_RMS_ROWSCALE) ? rsqrtf(fold(SKN) / (float)p.K + p.rms_eps) : 1.f;
    if (SWIGLU) {
        if (c >= 8) return;
        const int g = bx >> 2, c0 = (bx & 3) * 8;
        const int nv = g * 64 + c0 + c;                          // packed column of the value; its gate is nv + 32
        if (nv >= p.N) return;
        const float val = fold(c) * rscale + e_bias, gate = fold(c + 8) * rscale + e_gate_bias;
        p.C[(int64_t)m * p.ldc + g * 32 + c0 + c] = val * siluf(gate);
        return;
    }
    const int gn = bx * SKN + c;
    if (gn >= p.N) return;
    float v = fold(c) * rscale + e_bias;
    if (p.flags & GEMM_SILU) v = siluf(v);
    if (p.R) v += e_res;
    if (p.flags & GEMM_ACCUMULATE) v += e_acc;
    p.C[(int64_t)m * p.ldc + gn] = v;
    if (p.C2) {
        const int ts = m % p.c2_S;
        const int keep = p.c2_hi - p.c2_lo;
        const int rank = (ts >= p.c2_lo && ts < p.c2_hi) ? ts - p.c2_lo : ((p.c2_last && ts == p.c2_S - 1) ? keep : -1);
        if (rank >= 0) p.C2[((int64_t)(m / p.c2_S) * (keep + p.c2_last) + rank) * p.ldc2 + gn] = v;
    }
}

template <bool SWIGLU, int NW, int U>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(GemmArgs p) {
    __shared__ float red[NW * SK_ROWS * SK_RED];                 // [NW k-slices][16 rows][16 cols | ssq]
    skinny_body<SWIGLU, NW, U>(p, blockIdx.x, blockIdx.y, blockIdx.z, red);
}

// Two independent few-row GEMMs with the same K in ONE launch (blockIdx.x < nba: problem a): at these sizes a launch costs more
// than the arithmetic (a kernel boundary inside a graph is ~1.8 us, the kernel's own fixed latency ~2.6 us), and the attention
// pools' query / key projections and the final cross attention's q / kv projections are such pairs.
template <int NW, int U>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_pair_kernel(GemmArgs a, GemmArgs b, int nba) {
    __shared__ float red[NW * SK_ROWS * SK_RED];
    if ((int)blockIdx.x < nba) {
        if ((int)blockIdx.y * SK_ROWS < a.M) skinny_body<false, NW, U>(a, blockIdx.x, blockIdx.y, 0, red);
    } else {
        if ((int)blockIdx.y * SK_ROWS < b.M) skinny_body<false, NW, U>(b, blockIdx.x - nba, blockIdx.y, 0, red);
    }
}

// Few rows: M <= 32 rows, or the tiled kernel would have fewer than 64 tiles of 64 x 64 (a quarter of the CUs) to work with.  By shape only,
// never by timing.
bool gemm_skinny_applicable(const GemmArgs& p) {
    constexpr int max_tiles = 64, max_m = 32;
    const bool few = p.M <= max_m || (int64_t)cdiv(p.M, 64) * cdiv(p.N, 64) * (p.batch > 0 ? p.batch : 1) < max_tiles;
    return p.M >= 1 && p.M <= 256 && few &&
           !(p.flags & (GEMM_TRANS_A | GEMM_TRANS_B)) && (p.K % 4) == 0 && (p.C2 == nullptr || p.batch <= 1);
}

template <bool SWIGLU, int NW, int U>
static int launch_skinny(const GemmArgs& p, hipStream_t stream) {
    const int blocks = SWIGLU ? (p.N / 64) * 4 : cdiv(p.N, SKN);
    hipLaunchKernelGGL((gemm_skinny_kernel<SWIGLU, NW, U>), dim3(blocks, cdiv(p.M, SK_ROWS), p.batch > 0 ? p.batch : 1), dim3(64 * NW), 0, stream, p);
    D4_LAUNCH_CHECK();
    return 0;
}

// waves per block from K: one round of <= 4 (6) load steps per wave up to K = 1024 (1536)
static void skinny_shape(int K, int& nw, bool& u6) {
    const int steps = (K + 15) >> 4;
    nw = steps <= 16 ? 4 : (steps <= 32 ? 8 : 16);
    u6 = nw == 16 && steps > 64 && steps <= 96;
}

bool gemm_skinny_pair_applicable(const GemmArgs& a, const GemmArgs& b) {
    return gemm_skinny_applicable(a) && gemm_skinny_applicable(b) && a.K == b.K && !((a.flags | b.flags) & GEMM_SWIGLU) &&
           a.batch <= 1 && b.batch <= 1 && a.Wb == nullptr && b.Wb == nullptr;
}

int gemm_skinny_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t stream) {
    int nw; bool u6;
    skinny_shape(a.K, nw, u6);
    const int nba = cdiv(a.N, SKN), nbb = cdiv(b.N, SKN);
    const dim3 grid(nba + nbb, cdiv(a.M > b.M ? a.M : b.M, SK_ROWS));
#define D4_SKP(NW_, U_) hipLaunchKernelGGL((gemm_skinny_pair_kernel<NW_, U_>), grid, dim3(64 * NW_), 0, stream, a, b, nba)
    if (nw == 4) D4_SKP(4, 4);
    else if (nw == 8) D4_SKP(8, 4);
    else if (u6) D4_SKP(16, 6);
    else D4_SKP(16, 4);
#undef D4_SKP
    D4_LAUNCH_CHECK();
    return 0;
}

int gemm_skinny(const GemmArgs& p, hipStream_t stream) {
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    int nw; bool u6;
    skinny_shape(p.K, nw, u6);
#define D4_SK(NW_, U_) (swiglu ? launch_skinny<true, NW_, U_>(p, stream) : launch_skinny<false, NW_, U_>(p, stream))
    if (nw == 4) return D4_SK(4, 4);
    if (nw == 8) return D4_SK(8, 4);
    return u6 ? D4_SK(16, 6) : D4_SK(16, 4);
#undef D4_SK
}

}  // namespace d4
