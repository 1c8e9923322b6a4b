// Weight-gradient GEMM of the trunk backward:  C[M][N] = sum_k A[k][M] * B[k][N]
// with A = dY [rows][M] and B = X [rows][N] row-major (the gradient of a Linear's weight, dreamer4.py:2079-2116 / 1968-2075 backward):
// the contraction index k (token rows, thousands) is the SLOW index of both operands.
//
// Layout choice.  v_mfma_f32_16x16x4_f32 wants lane (i = lane % 16, kk = lane / 16) to hold A[i][kk] / B[kk][j]; a float4 read along the
// fast index of row kk therefore gives a lane FOUR different i (or j) at one kk.  Nothing forces the 16 rows of an MFMA tile to be adjacent:
// component c of the float4 feeds MFMA (c, d), whose tile is rows {m0 + 4 i + c}, columns {n0 + 4 j + d}.  So both operands go
// global -> VGPR -> MFMA as 16-byte loads, a wave instruction fetching 4 rows x 256 contiguous bytes — no LDS staging, no transposes, no
// barriers in the k loop — and the accumulator of (c, .) holds four adjacent output columns per lane: the result is written as float4 rows.
// A wave owns a 64 x 64 (x TM x TN) output tile = 16 TM TN accumulators; per 4 contraction rows it issues TM + TN loads for 16 TM TN MFMAs.
//
// The output is one weight matrix (tens to hundreds of 64 x 64 tiles) while k is long: the four waves of a workgroup take interleaved
// 4-row groups of the SAME tile and are summed through LDS in a fixed order (a 4-way split of k that costs no memory traffic), and the grid's
// y dimension splits k further into slices whose partial products are summed in slice order by splitk_reduce.  Both split factors are rules
// on the shape (never a timing), so a gradient is reproducible bit for bit.
#include "common.h"
#include "kernels.h"

namespace d4 {

struct TnArgs {
    const float* A; int lda;       // [K][lda], M columns used
    const float* B; int ldb;       // [K][ldb], N columns used
    float* C; int ldc;             // slice z writes C + z * strideC
    int64_t strideC;
    int M, N, K, kslice;           // slice z covers rows [z * kslice, min(K, (z + 1) * kslice)); kslice % 16 == 0
};

typedef float f4 __attribute__((ext_vector_type(4)));

template <int TM, int TN, int DEPTH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_tn_kernel(TnArgs p) {
    constexpr int RC = 8;                          // accumulators per reduction pass: 3 x 8 x 64 float4 = 24 KB of LDS (four workgroups per CU)
    __shared__ f4 red[3 * RC * 64];                // [3 waves][RC accumulators][64 lanes]
    constexpr int NACC = 16 * TM * TN;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l16 = lane & 15, lk = lane >> 4;
    const int tiles_n = (p.N + 64 * TN - 1) / (64 * TN);
    const int m0 = (blockIdx.x / tiles_n) * 64 * TM, n0 = (blockIdx.x % tiles_n) * 64 * TN;
    const int kbeg = blockIdx.y * p.kslice, kend = min(p.K, kbeg + p.kslice);
    const int steps = (kend - kbeg + 15) / 16;

    // column of this lane in each 64-wide unit (clamped inside the matrix; out-of-range lanes load a valid address and are zeroed)
    int ca[TM], cb[TN];
    bool oka[TM], okb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int c = m0 + 64 * i + 4 * l16; oka[i] = c < p.M; ca[i] = oka[i] ? c : 0; }
#pragma unroll
    for (int i = 0; i < TN; ++i) { const int c = n0 + 64 * i + 4 * l16; okb[i] = c < p.N; cb[i] = okb[i] ? c : 0; }

    f4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};

    f4 ra[DEPTH][TM], rb[DEPTH][TN];
    // raw loads only (the zeroing of out-of-range rows / columns happens where the values are consumed, so that a load never waits)
    auto load = [&](f4 (&a)[TM], f4 (&b)[TN], int t) {
        const int row = kbeg + 16 * t + 4 * w + lk;
        const int64_t r = row < kend ? row : kbeg;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f4*>(p.A + r * p.lda + ca[i]);
#pragma unroll
        for (int i = 0; i < TN; ++i) b[i] = *reinterpret_cast<const f4*>(p.B + r * p.ldb + cb[i]);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(ra[d], rb[d], d);          // (rows past the slice read row kbeg and are zeroed at use: no branches)

    const f4 zero{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int t = 0; t < steps; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            {
                const bool rok = kbeg + 16 * (t + d) + 4 * w + lk < kend;
                f4 xa[TM], xb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) xa[i] = (rok && oka[i]) ? ra[d][i] : zero;
#pragma unroll
                for (int i = 0; i < TN; ++i) xb[i] = (rok && okb[i]) ? rb[d][i] : zero;
                __builtin_amdgcn_sched_barrier(0);
                load(ra[d], rb[d], t + d + DEPTH);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int im = 0; im < TM; ++im)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int in = 0; in < TN; ++in)
#pragma unroll
                            for (int dd = 0; dd < 4; ++dd) {
                                const int idx = ((im * 4 + c) * TN + in) * 4 + dd;
                                acc[idx] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[im][c], xb[in][dd], acc[idx], 0, 0, 0);
                            }
            }
        }
    }

    // fixed-order sum of the four waves' partial tiles: waves 1..3 park theirs in LDS (8 accumulators = 24 KB per pass), wave 0 adds them
    // in wave order; then wave 0 stores
#pragma unroll
    for (int pass = 0; pass < NACC / RC; ++pass) {
        if (pass > 0) __syncthreads();
        if (w > 0) {
#pragma unroll
            for (int i = 0; i < RC; ++i) red[((w - 1) * RC + i) * 64 + lane] = acc[pass * RC + i];
        }
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < RC; ++i)
#pragma unroll
                for (int o = 0; o < 3; ++o) acc[pass * RC + i] += red[(o * RC + i) * 64 + lane];
        }
    }
    if (w > 0) return;
    float* C = p.C + blockIdx.y * p.strideC;
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 64 * im + 16 * lk + 4 * r + c;
                if (row >= p.M) continue;
#pragma unroll
                for (int in = 0; in < TN; ++in) {
                    const int col = n0 + 64 * in + 4 * l16;
                    if (col >= p.N) continue;
                    const int base = ((im * 4 + c) * TN + in) * 4;
                    const f4 v{acc[base + 0][r], acc[base + 1][r], acc[base + 2][r], acc[base + 3][r]};
                    *reinterpret_cast<f4*>(C + (int64_t)row * p.ldc + col) = v;
                }
            }
}

bool gemm_tn_applicable(const float* A, int lda, const float* B, int ldb, const float* C, int ldc, int M, int N, int K) {
    return M % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && M >= 4 && N >= 4 && K >= 1 &&
           (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
}

// Shape rule for the slice count (measured on the weight gradients of a cfg-2 training step, tools/gemm_tn_bench.py: 3840 to 49920 rows,
// 32 to 344 tiles): tiles x slices of about 768 workgroups (two to three per CU at the kernel's register budget), at most 8 slices, each
// slice >= 256 contraction rows, partials within `part`.  64 x 64 tile per workgroup throughout (the 64 x 128 form needs twice the registers,
// runs one wave per SIMD and measured slower on every shape).
void gemm_tn_plan(int M, int N, int K, size_t part_floats, int* tile_n, int* slices, int forced_tn, int forced_slices) {
    const int tn = 1;
    (void)forced_tn;
    const int64_t tiles = (int64_t)cdiv(M, 64) * cdiv(N, 64 * tn);
    int S = (int)((768 + tiles / 2) / tiles);
    if (S > 8) S = 8;
    if (S > K / 256) S = K / 256;
    if (S < 1) S = 1;
    if (forced_slices > 0) S = forced_slices;
    if (S > 1 && (size_t)S * M * N > part_floats) S = (int)(part_floats / ((size_t)M * N));
    if (S < 1) S = 1;
    *tile_n = tn; *slices = S;
}

int gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, float* part, size_t part_floats, hipStream_t s,
            int forced_tn, int forced_slices) {
    D4_REQUIRE(gemm_tn_applicable(A, lda, B, ldb, C, ldc, M, N, K), "gemm_tn: M / N / leading dimensions must be multiples of 4 and the operands 16-byte aligned");
    int tn = 1, S = 1;
    gemm_tn_plan(M, N, K, part ? part_floats : 0, &tn, &S, forced_tn, forced_slices);
    int ks = cdiv(cdiv(K, S), 16) * 16;
    S = cdiv(K, ks);
    TnArgs p{A, lda, B, ldb, S > 1 ? part : C, S > 1 ? N : ldc, S > 1 ? (int64_t)M * N : 0, M, N, K, ks};
    const dim3 grid(cdiv(M, 64) * cdiv(N, 64 * tn), S), block(256);
    if (forced_tn == 3) hipLaunchKernelGGL((gemm_tn_kernel<1, 1, 3>), grid, block, 0, s, p);
    else if (forced_tn == 4) hipLaunchKernelGGL((gemm_tn_kernel<1, 1, 2>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_tn_kernel<1, 1, 4>), grid, block, 0, s, p);       // 116 registers: four waves per SIMD, four workgroups per CU
    D4_LAUNCH_CHECK();
    if (S > 1) return splitk_reduce(part, S, M, N, nullptr, 0, C, ldc, s);
    return 0;
}

// Input gradient dX[M][N] = dY[M][K] W[K][N] (W = the Linear's weight, row-major [out = K][in = N]): W's contraction index is its slow one.
// The weight is transposed once per call into `wt` [N][K] (a few MB: microseconds) so that the product runs as the plain row-major form on the
// LDS-DMA family (gemm2.hip) instead of the transposed-operand form of the first family (measured 95-105 vs 65-77 TF/s on these shapes).
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, int rows, int cols) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        t[ty + 8 * i][tx] = (r < rows && c < cols) ? src[(int64_t)r * lds_ + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < cols && r < rows) dst[(int64_t)c * ldd + r] = t[tx][ty + 8 * i];
    }
}
int gemm_dx(const float* dY, int ldy, const float* W, int ldw, float* dX, int ldx, int M, int N, int K, float* wt, hipStream_t s) {
    if (!wt || K % 32 != 0 || ldy % 4 != 0 || M < 256) { GemmArgs g{dY, ldy, W, ldw, dX, ldx, nullptr, nullptr, 0, M, N, K, GEMM_TRANS_B, 0.f}; return gemm(g, s); }
    hipLaunchKernelGGL(transpose_kernel, dim3((N + 31) / 32, (K + 31) / 32), dim3(256), 0, s, W, ldw, wt, K, K, N);
    D4_LAUNCH_CHECK();
    GemmArgs g{dY, ldy, wt, K, dX, ldx, nullptr, nullptr, 0, M, N, K, 0, 0.f};
    return gemm(g, s);
}


}  // namespace d4
