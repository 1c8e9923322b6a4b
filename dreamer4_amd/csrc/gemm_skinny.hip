// Few-row GEMM: C[M][N] = epilogue(A[M][K] . W[N][K]^T) when M is so small that the tiled kernel would have fewer than 64
// tiles (a quarter of the CUs) — the small-batch decode regime (BASELINE config 4: one trajectory = 15 token rows per frame).
// There the tiled MFMA kernel is pure latency: one 64-row tile per block walks all of K serially (16 k-tiles x ~0.7 us) with the
// matrix pipes 75 % empty, while the real cost is streaming W once.
//
// Here a block owns 16 output columns x up to 64 rows and all of K; its 4 waves split K four ways and each runs the 16x16x4 fp32
// MFMA (16 rows tall) with BOTH operands streamed straight from global memory: every W element is used once per row group and A
// is a few tens of KB that stays in L2, so neither touches LDS.  One 16-byte load per lane feeds four MFMA k-steps (the same
// permuted-k trick as gemm_kernel).  The four partial tiles (and the rows' sums of squares for the folded RMSNorm) are combined
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

template <bool SWIGLU, int TM>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p) {
    constexpr int RB = SK_ROWS * TM;                             // rows per block (blockIdx.y = row group)
    const int m0 = blockIdx.y * RB;
    {   // strided batch (blockIdx.z), as in gemm_kernel
        const int bz = blockIdx.z;
        p.A += bz * p.strideA; p.W += bz * p.strideW; p.C += bz * p.strideC;
        if (p.R) p.R += bz * p.strideC;
    }
    __shared__ float red[4 * RB * SK_RED];                       // [4 k-quarters][RB rows][16 cols | ssq]
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int nl = lane & 15, kk = lane >> 4;

    // ---- this lane's W row (MFMA B operand: column nl of the block, k sub-slot kk)
    int n;
    if (SWIGLU) {       // packed pairs: 32 value columns then their 32 gate columns per group of 64; a block takes 8 + 8
        const int g = blockIdx.x >> 2, c0 = (blockIdx.x & 3) * 8;
        n = g * 64 + (nl < 8 ? c0 + nl : 32 + c0 + (nl - 8));
    } else {
        n = blockIdx.x * SKN + nl;
    }
    const bool valid = n < p.N;
    const float* wrow = p.W + (int64_t)(valid ? n : 0) * p.ldw;
    // MFMA A operand: row nl (lane & 15) of each 16-row tile, same k sub-slot; rows past M read as zero
    const float* arow[TM];
    bool arow_ok[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int m = m0 + t * SK_ROWS + nl;
        arow_ok[t] = m < p.M;
        arow[t] = p.A + (int64_t)(arow_ok[t] ? m : 0) * p.lda;
    }

    f32x4 acc[TM];
    float ssq[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; ssq[t] = 0.f; }
    const int steps = (p.K + 15) >> 4;                           // 16 k per step (4 MFMAs of k = 4)
    const int per = (steps + 3) >> 2;
    const int s1 = min((q + 1) * per, steps);
    constexpr int U = TM >= 4 ? 2 : 4;                           // steps whose loads are issued together (8 / 4 measured: no gain at M = 15)
    for (int s0 = q * per; s0 < s1; s0 += U) {
        f32x4 w4[U], a4[U][TM];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = 16 * (s0 + u) + 4 * kk;
            const bool in = s0 + u < s1 && k < p.K;
            w4[u] = (valid && in) ? *reinterpret_cast<const f32x4*>(wrow + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < TM; ++t)
                a4[u][t] = (arow_ok[t] && in) ? *reinterpret_cast<const f32x4*>(arow[t] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const f32x4 a = a4[u][t];
                ssq[t] += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w4[u][e], acc[t], 0, 0, 0);
            }
    }
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg.  The sum of squares of row nl is spread over
    // the 4 k sub-slots of this wave: fold those with two DPP-free shuffles through LDS-less lane swaps (xor 16, xor 32).
#pragma unroll
    for (int t = 0; t < TM; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((q * RB) + t * SK_ROWS + 4 * kk + r) * SK_RED + nl] = acc[t][r];
        float s = ssq[t];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (kk == 0) red[((q * RB) + t * SK_ROWS + nl) * SK_RED + SKN] = s;
    }
    __syncthreads();

    // ---- fold the 4 k-quarters in a fixed order; thread = (row, column), TM passes of 16 rows
    const int c = tid & 15;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
    const int ml = t * SK_ROWS + (tid >> 4), m = m0 + ml;
    if (m >= p.M) return;
    auto fold = [&](int col) {
        return ((red[(0 * RB + ml) * SK_RED + col] + red[(1 * RB + ml) * SK_RED + col]) + red[(2 * RB + ml) * SK_RED + col]) +
               red[(3 * RB + ml) * SK_RED + col];
    };
    const float rscale = (p.flags & GEMM_RMS_ROWSCALE) ? rsqrtf(fold(SKN) / (float)p.K + p.rms_eps) : 1.f;
    if (SWIGLU) {
        if (c >= 8) return;
        const int g = blockIdx.x >> 2, c0 = (blockIdx.x & 3) * 8;
        const int nv = g * 64 + c0 + c;                          // packed column of the value; its gate is nv + 32
        if (nv >= p.N) return;
        float val = fold(c) * rscale, gate = fold(c + 8) * rscale;
        if (p.bias) { val += p.bias[nv]; gate += p.bias[nv + 32]; }
        p.C[(int64_t)m * p.ldc + g * 32 + c0 + c] = val * siluf(gate);
        continue;
    }
    const int gn = blockIdx.x * SKN + c;
    if (gn >= p.N) return;
    float v = fold(c) * rscale;
    if (p.bias) v += p.bias[gn];
    if (p.flags & GEMM_SILU) v = siluf(v);
    if (p.R) v += p.R[(int64_t)m * p.ldr + gn];
    if (p.flags & GEMM_ACCUMULATE) v += p.C[(int64_t)m * p.ldc + gn];
    p.C[(int64_t)m * p.ldc + gn] = v;
    if (p.C2) {
        const int ts = m % p.c2_S;
        const int keep = p.c2_hi - p.c2_lo;
        const int rank = (ts >= p.c2_lo && ts < p.c2_hi) ? ts - p.c2_lo : ((p.c2_last && ts == p.c2_S - 1) ? keep : -1);
        if (rank >= 0) p.C2[((int64_t)(m / p.c2_S) * (keep + p.c2_last) + rank) * p.ldc2 + gn] = v;
    }
    }
}

// rows per block = 16 * tm
static int skinny_tm(const GemmArgs& p) { return p.M > 32 ? 4 : (p.M > 16 ? 2 : 1); }

// Few rows = the tiled kernel would have fewer than 64 tiles of 64 x 64 (a quarter of the CUs) to work with.
bool gemm_skinny_applicable(const GemmArgs& p) {
    static const bool on = !(getenv("D4_GEMM_SKINNY") && atoi(getenv("D4_GEMM_SKINNY")) == 0);
    static const int max_tiles = getenv("D4_SKINNY_TILES") ? atoi(getenv("D4_SKINNY_TILES")) : 64;
    return on && p.M >= 1 && p.M <= 256 && (int64_t)cdiv(p.M, 64) * cdiv(p.N, 64) * (p.batch > 0 ? p.batch : 1) < max_tiles &&
           !(p.flags & (GEMM_TRANS_A | GEMM_TRANS_B)) && (p.K % 4) == 0 && (p.C2 == nullptr || p.batch <= 1);
}

template <bool SWIGLU, int TM>
static int launch_skinny(const GemmArgs& p, hipStream_t stream) {
    const int blocks = SWIGLU ? (p.N / 64) * 4 : cdiv(p.N, SKN);
    hipLaunchKernelGGL((gemm_skinny_kernel<SWIGLU, TM>), dim3(blocks, cdiv(p.M, SK_ROWS * TM), p.batch > 0 ? p.batch : 1), dim3(256), 0, stream, p);
    D4_LAUNCH_CHECK();
    return 0;
}

int gemm_skinny(const GemmArgs& p, hipStream_t stream) {
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    switch (skinny_tm(p)) {
        case 4: return swiglu ? launch_skinny<true, 4>(p, stream) : launch_skinny<false, 4>(p, stream);
        case 2: return swiglu ? launch_skinny<true, 2>(p, stream) : launch_skinny<false, 2>(p, stream);
        default: return swiglu ? launch_skinny<true, 1>(p, stream) : launch_skinny<false, 1>(p, stream);
    }
}

}  // namespace d4
