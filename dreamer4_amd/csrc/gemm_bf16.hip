// bf16 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate) — the opt-in compute type of the trunk
// (BASELINE config 5: "MFMA bf16"; the headline config 2 stays fp32, as the reference computes).
//
//   C[m, n] = epilogue( rowscale[m] * sum_k bf16(A[m, k]) * Wb[n, k] )          A fp32 [M][K], Wb bf16 [N][K], C fp32
//
// Weights are converted to bf16 ONCE at engine prepare (gamma-folded images and the raw output projections alike);
// activations stay fp32 in HBM (residual stream, attention-pool context, every glue kernel is unchanged) and are rounded to
// bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on their way into LDS, which is numerically the same as storing them in
// bf16.  Norms stay fp32: the folded RMSNorm's 1/rms is accumulated from the fp32 registers before the rounding.
// Same epilogues as gemm_kernel (row scale, bias, SiLU, SiLU-GLU pairing, residual, accumulate, row-compacted second output);
// the 32x32 C/D layout is dtype independent on gfx950, so the epilogue code is the fp32 kernel's.
//
// Staging is global -> registers -> LDS, double-buffered in LDS with the loads of k-tile j+1 in flight under the MFMAs of
// k-tile j.  LDS rows are [BK + 8] bf16 (16 bytes of padding: the 16 lanes a ds_read_b128 services together land on 16
// distinct 16-byte slots).  A 32x32x16 MFMA takes 8 consecutive k per lane (lane >> 5 selects the k half), so fragments
// are single ds_read_b128 and the fp32 -> bf16 staging writes are ds_write_b128 of 8 converted values.
#include "common.h"
#include <hip/hip_ext.h>
#include "kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace d4 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int BM, int BN, int WGM, int WGN, int BK>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_bf16_kernel(GemmArgs p) {
    static_assert(BK == 32 || BK == 64, "k-tile");
    constexpr int LDS_LD = BK + 8;                  // bf16 elements per LDS row
    constexpr int NT = WGM * WGN * 64;
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int G = BK / 8;                       // 8-element groups per tile row
    constexpr int A_G = BM * G / NT, B_G = BN * G / NT;
    static_assert(TM >= 1 && TN >= 1 && A_G >= 1 && B_G >= 1, "tile");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __bf16* As = reinterpret_cast<__bf16*>(smem_raw);                 // [2][BM][LDS_LD]
    __bf16* Bs = As + 2 * BM * LDS_LD;                                // [2][BN][LDS_LD]
    float* rowscale_s = reinterpret_cast<float*>(Bs + 2 * BN * LDS_LD);   // [BM]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    int bid = blockIdx.x;
    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    {
        const int nblk = nbm * nbn, nx = 8;
        const int q = nblk / nx, r = nblk % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    }
    const int bm0 = (bid / nbn) * BM, bn0 = (bid % nbn) * BN;
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
    const __amdgpu_buffer_rsrc_t rsB = uniform_rsrc(Wb + (int64_t)bn0 * p.ldw, ((int64_t)(rowsB - 1) * p.ldw + p.K) * 2);

    f32x4 ra[A_G][2];
    f32x4 rb[B_G];                                  // 8 bf16 as 16 raw bytes
    float ssq[A_G];
#pragma unroll
    for (int i = 0; i < A_G; ++i) ssq[i] = 0.f;

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_G; ++i) {
            const int idx = tid + i * NT, r = idx / G, c = (idx % G) * 8;
            const uint32_t off = (uint32_t)((r * p.lda + k0 + c) * 4);
            ra[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
            ra[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off + 16, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < B_G; ++i) {
            const int idx = tid + i * NT, r = idx / G, c = (idx % G) * 8;
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (uint32_t)((r * p.ldw + k0 + c) * 2), 0, 0));
        }
    };
    auto store_tile = [&](int buf) {
        __bf16* as = As + buf * BM * LDS_LD;
        __bf16* bs = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < A_G; ++i) {
            const int idx = tid + i * NT, r = idx / G, c = (idx % G) * 8;
            const f32x4 v0 = ra[i][0], v1 = ra[i][1];
            bf16x8 o;
            o[0] = (__bf16)v0[0]; o[1] = (__bf16)v0[1]; o[2] = (__bf16)v0[2]; o[3] = (__bf16)v0[3];
            o[4] = (__bf16)v1[0]; o[5] = (__bf16)v1[1]; o[6] = (__bf16)v1[2]; o[7] = (__bf16)v1[3];
            *reinterpret_cast<bf16x8*>(as + r * LDS_LD + c) = o;
            ssq[i] = ssq[i] + ((v0[0] * v0[0] + v0[1] * v0[1]) + (v0[2] * v0[2] + v0[3] * v0[3])) + ((v1[0] * v1[0] + v1[1] * v1[1]) + (v1[2] * v1[2] + v1[3] * v1[3]));
        }
#pragma unroll
        for (int i = 0; i < B_G; ++i) {
            const int idx = tid + i * NT, r = idx / G, c = (idx % G) * 8;
            *reinterpret_cast<f32x4*>(bs + r * LDS_LD + c) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = p.K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int lrow = lane & 31, lhalf = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        const __bf16* as = As + cur * BM * LDS_LD + (wm * TM * 32 + lrow) * LDS_LD + lhalf * 8;
        const __bf16* bs = Bs + cur * BN * LDS_LD + (wn * TN * 32 + lrow) * LDS_LD + lhalf * 8;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(as + i * 32 * LDS_LD + ks * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(bs + j * 32 * LDS_LD + ks * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    if (p.flags & GEMM_RMS_ROWSCALE) {
#pragma unroll
        for (int i = 0; i < A_G; ++i) {
            float s = ssq[i];
            s += dpp_f<0xB1>(s);
            s += dpp_f<0x4E>(s);                       // G (4 or 8) consecutive lanes share one row
            if constexpr (G == 8) s += dpp_f<0x141>(s);
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
                        float val = acc[i][j][e] * rs, gate = acc[i][j + 1][e] * rs;
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

// ---- fp32 -> bf16 weight images (engine prepare) ---------------------------------------------------------------
__global__ void cvt_bf16_kernel(const float* src, __bf16* dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (__bf16)src[i];
}
int cvt_f32_to_bf16(const float* src, uint16_t* dst, int64_t n, hipStream_t s) {
    if (n == 0) return 0;
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(cvt_bf16_kernel, dim3((unsigned)g), dim3(256), 0, s, src, reinterpret_cast<__bf16*>(dst), n);
    D4_LAUNCH_CHECK();
    return 0;
}

// strided rows -> bf16 (the activation images of the bf16 engine where a producer cannot write them itself); cols % 4 == 0 takes the 16-byte path
__global__ __launch_bounds__(256) void cvt_rows_bf16_kernel(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int rows, int cols) {
    const int c4 = cols >> 2;
    const int64_t n4 = (int64_t)rows * c4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c4; const int c = (int)(i % c4) * 4;
        store_bf16x4(dst + r * ldd + c, *reinterpret_cast<const f32x4*>(src + r * lds + c));
    }
}
__global__ __launch_bounds__(256) void cvt_rows_bf16_scalar_kernel(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int rows, int cols) {
    const int64_t n = (int64_t)rows * cols;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[(i / cols) * ldd + i % cols] = bf16_bits(src[(i / cols) * lds + i % cols]);
}
int cvt_rows_bf16(const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int rows, int cols, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return 0;
    const bool v4 = (cols % 4) == 0 && (lds % 4) == 0 && (ldd % 4) == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0;
    const int64_t n = (int64_t)rows * (v4 ? cols / 4 : cols);
    int64_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    if (v4) hipLaunchKernelGGL(cvt_rows_bf16_kernel, dim3((unsigned)g), dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
    else hipLaunchKernelGGL(cvt_rows_bf16_scalar_kernel, dim3((unsigned)g), dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
    D4_LAUNCH_CHECK();
    return 0;
}

// ---- per-launch timing (bench.py's cfg-5 leg) ----------------------------------------------------------------------------------
struct Bf16Prof { hipEvent_t a, b; double flops; int M, N, K, flags, batch; };
static std::vector<Bf16Prof> g_bprof;
static int g_bprof_stride = 0, g_bprof_tick = 0;        // 0 = off; n = every n-th launch carries an event pair
void gemm_bf16_profile_enable(int stride) { g_bprof_stride = stride; g_bprof_tick = 0; }
bool gemm_bf16_profile_active() { return g_bprof_stride > 0; }
int gemm_bf16_profile_read(double* ms, double* flops, int64_t* count) {
    *ms = 0; *flops = 0; *count = 0;
    static const bool log_shapes = getenv("D4_GEMM_LOG") != nullptr;       // per-shape table on stderr
    struct Agg { Bf16Prof r; double ms, fl; int64_t n; };
    std::vector<Agg> shapes;
    for (auto& r : g_bprof) {
        D4_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        D4_HIP(hipEventElapsedTime(&t, r.a, r.b));
        *ms += t; *flops += r.flops; *count += 1;
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
        if (log_shapes) {
            bool hit = false;
            for (auto& a : shapes)
                if (a.r.M == r.M && a.r.N == r.N && a.r.K == r.K && a.r.flags == r.flags && a.r.batch == r.batch) { a.ms += t; a.fl += r.flops; a.n += 1; hit = true; break; }
            if (!hit) shapes.push_back(Agg{r, t, r.flops, 1});
        }
    }
    if (log_shapes)
        for (auto& a : shapes)
            fprintf(stderr, "[d4 gemm bf16] M %6d N %5d K %5d batch %2d flags %3d : %6lld launches %9.3f ms (%5.1f %%) avg %7.1f us %6.1f TF/s\n", a.r.M, a.r.N, a.r.K,
                    a.r.batch, a.r.flags, (long long)a.n, a.ms, 100 * a.ms / (*ms > 0 ? *ms : 1), 1e3 * a.ms / a.n, a.fl / a.ms / 1e9);
    g_bprof.clear();
    return 0;
}

template <int BM, int BN, int WGM, int WGN, int BK>
static int launch_bf16(const GemmArgs& p, hipStream_t stream) {
    constexpr int LDS_LD = BK + 8;
    const size_t lds = (size_t)(2 * BM * LDS_LD + 2 * BN * LDS_LD) * 2 + BM * sizeof(float);
    auto k = gemm_bf16_kernel<BM, BN, WGM, WGN, BK>;
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.done();
    }
    const dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), p.batch > 0 ? p.batch : 1), block(WGM * WGN * 64);
    if (g_bprof_stride > 0 && (g_bprof_tick++ % g_bprof_stride) == 0) {
        Bf16Prof r{};
        D4_HIP(hipEventCreate(&r.a)); D4_HIP(hipEventCreate(&r.b));
        r.flops = p.algo_flops > 0 ? p.algo_flops : 2.0 * p.M * p.N * p.K * (p.batch > 0 ? p.batch : 1);
        r.M = p.M; r.N = p.N; r.K = p.K; r.flags = p.flags; r.batch = p.batch;
        hipExtLaunchKernelGGL(k, grid, block, (uint32_t)lds, stream, r.a, r.b, 0, p);
        g_bprof.push_back(r);
    } else {
        hipLaunchKernelGGL(k, grid, block, lds, stream, p);
    }
    D4_LAUNCH_CHECK();
    return 0;
}

// bf16 activations + bf16 weights (gemm_bf16a.hip): tile by shape rule, with this file's optional per-launch event pair
static int g_bf16a_forced = -1;          // test / microbenchmark hook (d4_gemm_force_config(500 + c); -1 resets)
void gemm_bf16a_force_config(int id) { g_bf16a_forced = id; }
int gemm_bf16a(const GemmArgs& p, hipStream_t stream) {
    D4_REQUIRE(gemm_bf16a_applicable(p), "gemm_bf16a: call not supported (M=%d N=%d K=%d flags=%d)", p.M, p.N, p.K, p.flags);
    D4_REQUIRE(p.C || p.Cb, "gemm_bf16a: no output");
    const int c = gemm_bf16a_config_valid(g_bf16a_forced, p) ? g_bf16a_forced : gemm_bf16a_rule(p);
    if (g_bprof_stride > 0 && (g_bprof_tick++ % g_bprof_stride) == 0) {
        Bf16Prof r{};
        D4_HIP(hipEventCreate(&r.a)); D4_HIP(hipEventCreate(&r.b));
        r.flops = p.algo_flops > 0 ? p.algo_flops : 2.0 * p.M * p.N * p.K * (p.batch > 0 ? p.batch : 1);
        r.M = p.M; r.N = p.N; r.K = p.K; r.flags = p.flags | 1024; r.batch = p.batch;          // (1024 marks the bf16-activation kernel in the shape log)
        const int rc = gemm_bf16a_launch(c, p, stream, r.a, r.b);
        g_bprof.push_back(r);
        return rc;
    }
    return gemm_bf16a_launch(c, p, stream, nullptr, nullptr);
}

int gemm_bf16a_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t stream) {
    if (g_bprof_stride > 0 && (g_bprof_tick++ % g_bprof_stride) == 0) {
        Bf16Prof r{};
        D4_HIP(hipEventCreate(&r.a)); D4_HIP(hipEventCreate(&r.b));
        r.flops = 2.0 * a.M * a.N * a.K + 2.0 * b.M * b.N * b.K;
        r.M = a.M + b.M; r.N = a.N; r.K = a.K; r.flags = a.flags | 1024 | 2048; r.batch = 2;       // (2048 marks the pair form in the shape log)
        const int rc = gemm_bf16a_pair_launch(a, b, stream, r.a, r.b);
        g_bprof.push_back(r);
        return rc;
    }
    return gemm_bf16a_pair_launch(a, b, stream, nullptr, nullptr);
}

bool gemm_bf16_applicable(const GemmArgs& p) {
    return p.Wb != nullptr && !(p.flags & (GEMM_TRANS_A | GEMM_TRANS_B)) && (p.K % 32) == 0 && (p.lda % 4) == 0 && (p.ldw % 8) == 0 &&
           ((uintptr_t)p.Wb % 16) == 0 && (p.strideW % 8) == 0;
}

// tile configurations (every one sums k in the same order with the same MFMA: identical bits, so the choice is free)
enum { B16_128x128 = 0, B16_128x128_8, B16_64x128, B16_64x64, B16_256x128_8, B16_128x64, B16_N };
static int g_bf16_forced = -1;          // test / microbenchmark hook
int gemm_bf16_force_config(int id) { g_bf16_forced = id; return B16_N; }

template <int BK>
static int launch_bf16_id(int id, const GemmArgs& p, hipStream_t stream) {
    switch (id) {
        case B16_128x128: return launch_bf16<128, 128, 2, 2, BK>(p, stream);
        case B16_128x128_8: return launch_bf16<128, 128, 4, 2, BK>(p, stream);
        case B16_64x128: return launch_bf16<64, 128, 2, 2, BK>(p, stream);
        case B16_64x64: return launch_bf16<64, 64, 2, 2, BK>(p, stream);
        case B16_256x128_8: return launch_bf16<256, 128, 4, 2, BK>(p, stream);
        case B16_128x64: return launch_bf16<128, 64, 2, 2, BK>(p, stream);
    }
    return 2;
}

static bool bf16_cfg_valid(int id, const GemmArgs& p) {
    if (id < 0 || id >= B16_N) return false;
    if (p.flags & GEMM_SWIGLU) return id != B16_64x64 && id != B16_128x64;        // the pairing needs two 32-column sub-tiles per wave
    return true;
}

// tile choice by rule
int gemm_bf16(const GemmArgs& p, hipStream_t stream) {
    D4_REQUIRE(gemm_bf16_applicable(p), "gemm_bf16: call not supported (M=%d N=%d K=%d flags=%d)", p.M, p.N, p.K, p.flags);
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    const int nb = p.batch > 0 ? p.batch : 1;
    const bool k64 = (p.K % 64) == 0;
    int id;
    if (bf16_cfg_valid(g_bf16_forced, p)) id = g_bf16_forced;
    else {
        // measured on the config-5 shapes (tools/gemm_bf16_bench.py, profiles/r02_gemm_bf16_tiles.txt): 64x64 with 4 waves wins while a
        // GEMM has fewer than ~300 tiles of 128x128 (it is launch / first-tile latency, not matrix work); above that 128x128 with 8 waves
        const int64_t t128 = (int64_t)cdiv(p.M, 128) * cdiv(p.N, 128) * nb;
        if (t128 >= 300 && p.N > 64) id = B16_128x128_8;
        else if (swiglu) id = t128 >= 150 ? B16_128x128_8 : B16_64x128;
        else id = B16_64x64;
    }
    return k64 ? launch_bf16_id<64>(id, p, stream) : launch_bf16_id<32>(id, p, stream);
}

}  // namespace d4
