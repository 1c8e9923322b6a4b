// Split-operand fp32 GEMM (the arithmetic of gemm_x3.hip: three bf16 planes per operand, six bf16 MFMA products, fp32 accumulate — bit-identical
// results), A-STATIONARY / W-STREAMING form, round 5.
//
// What the tiled forms (gemm_x3.hip, gemm_x3sk.hip) pay per k-tile of 32: both operands through LDS (six plane stores, nine fragment reads), the split
// of the activation tile repeated in EVERY column tile of its row panel (22 times for the SiLU-GLU input projection) and a barrier — the k-tile
// step is bound by that chain, not by the matrix pipe (profiles/r02_gemm_split_operands_ablation.txt: everything but the MFMAs is 45 of 91 us).
// Here a workgroup (8 waves) owns 64 rows x 256 columns:
//   * A: the 64-row panel is split ONCE per 128-deep k pass into three bf16 planes in LDS (double buffered: the split of pass p + 1 rides between
//     the MFMAs of pass p: 16 elements per thread per 96 MFMAs), fragments by conflict-free ds_read_b128;
//   * W: never touches LDS.  The weights are re-tiled once at prepare time (split_bf16x3_tiled) into the B-operand fragment order of
//     v_mfma_f32_32x32x16_bf16 — [N / 32][K / 16][plane][lane][8 bf16]: a wave's three plane fragments of one (column tile, k step) are ONE
//     contiguous 3 KB — and stream L2 -> VGPR -> MFMA, three k steps ahead;
//   * ONE barrier per 128-deep pass (96 MFMAs per wave), none inside it.
// A wave owns 32 columns x the 64 rows (two 32 x 32 accumulator pairs hi / lo): every W fragment feeds two row tiles, every A fragment six MFMAs.
//
// Same term order per 16-k step and the same accumulators as gemm_x3_kernel (small terms into `lo`, a1.w1 into `hi`), the folded RMSNorm's row
// sums in the canonical order of the other families: BIT-IDENTICAL to gemm_x3_kernel on every shape (tests/test_gpu_kernels.py).
#include "common.h"
#include <hip/hip_ext.h>
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace d4 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int XW_BM = 64, XW_BN = 256, XW_KP = 128, XW_NW = 8, XW_NT = XW_NW * 64;
constexpr int XW_LDA = XW_KP + 8;                         // bf16 elements per LDS row (272 bytes: rows 16 bytes apart modulo 256 -> conflict-free b128)
constexpr int XW_PLANE = XW_BM * XW_LDA;                  // one plane of one buffer
constexpr size_t XW_LDS = (size_t)(2 * 3 * XW_PLANE) * 2 + XW_BM * sizeof(float);

__device__ __forceinline__ void xw_split3(float a, __bf16& h1, __bf16& h2, __bf16& h3) {
    h1 = (__bf16)a;
    const float r = a - (float)h1;
    h2 = (__bf16)r;
    h3 = (__bf16)(r - (float)h2);
}

__global__ __launch_bounds__(XW_NT, 2) void gemm_x3w_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __bf16* As = reinterpret_cast<__bf16*>(smem_raw);                       // [2][3][64][XW_LDA]
    float* rowscale_s = reinterpret_cast<float*>(As + 2 * 3 * XW_PLANE);    // [64]; the epilogue's SiLU-GLU exchange reuses the plane area

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbn = (p.N + XW_BN - 1) / XW_BN, nbm = (p.M + XW_BM - 1) / XW_BM;
    int bid = blockIdx.x;                                                    // XCD-aware order (as gemm_x3.hip): an XCD's blocks share A row panels
    {
        const int nblk = nbm * nbn, nx = 8;
        const int q = nblk / nx, r = nblk % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    }
    int tm, tn;
    {
        constexpr int RB = 8;
        const int band = bid / (RB * nbn), j = bid % (RB * nbn);
        const int rows = min(RB, nbm - band * RB);
        tm = band * RB + j % rows; tn = j / rows;
    }
    const int bm0 = tm * XW_BM, bn0 = tn * XW_BN;
    const int rowsA = min(XW_BM, p.M - bm0);
    auto uniform_rsrc = [](const void* base, int64_t bytes) {
        const uint64_t b = reinterpret_cast<uint64_t>(base);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        const int nb = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7FFFFFFF ? bytes : 0x7FFFFFFF));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
    };
    // rows past the matrix edge fall outside num_records and read as zeros
    const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc(p.A + (int64_t)bm0 * p.lda, ((int64_t)(rowsA - 1) * p.lda + p.K) * 4);
    // this wave's 32-column tile of the fragment-ordered weight image: [K / 16][3][64 lanes][8 bf16]; a column tile past N reads zeros
    const int nt = (bn0 >> 5) + wave;
    const int nk16 = p.K >> 4;
    const int ntiles = (p.N + 31) >> 5;
    const __bf16* Wt = reinterpret_cast<const __bf16*>(p.Wb);
    const __amdgpu_buffer_rsrc_t rsW = uniform_rsrc(Wt + (int64_t)(nt < ntiles ? nt : 0) * nk16 * 3 * 512, nt < ntiles ? (int64_t)nk16 * 3 * 512 * 2 : 0);

    // ---- A staging: thread -> (row = tid / 8, 16 consecutive k of the pass): four 16-byte loads, six 16-byte plane stores
    const int srow = tid >> 3, sj = tid & 7;
    f32x4 ra[4];
    auto load_a = [&](int k0) {                       // k0: first k of the pass; columns past K (a shorter last pass) read as zeros
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = k0 + sj * 16 + c * 4;
            const uint32_t off = k < p.K ? (uint32_t)((srow * p.lda + k) * 4) : 0xFFFFFFF0u;
            ra[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
        }
    };
    // folded RMSNorm: running sums per 16-byte chunk position (k mod 32) / 4, k-tiles of 32 added in k order (the canonical order of the other
    // families).  Thread sj holds k-tile sj / 2 of the pass, chunk positions 4 (sj & 1) .. + 3: the lanes sj = 0, 1 own the row's eight running
    // sums and take the later k-tiles' values from their neighbours 2, 4, 6 lanes up, in k order.
    float ssq[4] = {0.f, 0.f, 0.f, 0.f};
    auto store_a = [&](int buf) {
        __bf16* as = As + buf * 3 * XW_PLANE + srow * XW_LDA + sj * 16;
        bf16x8 o[3][2];
        float f[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 v = ra[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 h1, h2, h3;
                xw_split3(v[e], h1, h2, h3);
                o[0][c >> 1][(c & 1) * 4 + e] = h1; o[1][c >> 1][(c & 1) * 4 + e] = h2; o[2][c >> 1][(c & 1) * 4 + e] = h3;
            }
            f[c] = __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], v[0] * v[0])));
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            *reinterpret_cast<bf16x8*>(as + pl * XW_PLANE) = o[pl][0];
            *reinterpret_cast<bf16x8*>(as + pl * XW_PLANE + 8) = o[pl][1];
        }
        if (p.flags & GEMM_RMS_ROWSCALE) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {              // k-tile kt of the pass lives in lanes sj = 2 kt, 2 kt + 1
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float g = kt == 0 ? f[c] : __shfl_down(f[c], 2 * kt, 8);
                    if (sj < 2) ssq[c] += g;
                }
            }
        }
    };

    f32x16 hi[2], lo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { hi[i][e] = 0.f; lo[i][e] = 0.f; }

    const int lrow = lane & 31, lhalf = lane >> 5;
    // W fragments: a whole pass (eight k steps) in flight — ring of 8 slots x 3 planes, a step's slot is a compile-time constant; an L2 hit takes
    // ~1.5 us under load, four steps of cover (1.5 us of MFMAs) measured 43 % MFMA utilisation
    f32x4 wf[8][3];
    auto load_w = [&](auto slot_tag, int ks) {        // ks: global 16-k step; past the end: clamped (never used)
        constexpr int slot = decltype(slot_tag)::value;
        const int kk = ks < nk16 ? ks : nk16 - 1;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            wf[slot][pl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (uint32_t)(((kk * 3 + pl) * 64 + lane) * 16), 0, 0));
    };
    auto mma = [&](int buf, int ksl, auto slot_tag) {       // ksl: 16-k step inside the pass
        constexpr int slot = decltype(slot_tag)::value;
        const __bf16* as = As + buf * 3 * XW_PLANE + lrow * XW_LDA + lhalf * 8 + ksl * 16;
        bf16x8 af[3][2], bf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            bf[pl] = __builtin_bit_cast(bf16x8, wf[slot][pl]);
#pragma unroll
            for (int i = 0; i < 2; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(as + pl * XW_PLANE + i * 32 * XW_LDA);
        }
#define D4_XW_TERM(PA, PB, ACC) \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) ACC[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA][i], bf[PB], ACC[i], 0, 0, 0);
        D4_XW_TERM(2, 0, lo)
        D4_XW_TERM(0, 0, hi)
        D4_XW_TERM(1, 1, lo)
        D4_XW_TERM(0, 2, lo)
        D4_XW_TERM(1, 0, lo)
        D4_XW_TERM(0, 1, lo)
#undef D4_XW_TERM
    };

    const int npass = (p.K + XW_KP - 1) / XW_KP;
    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>; using T3 = std::integral_constant<int, 3>;
    load_w(T0{}, 0); load_w(T1{}, 1); load_w(T2{}, 2); load_w(T3{}, 3);
    load_w(std::integral_constant<int, 4>{}, 4); load_w(std::integral_constant<int, 5>{}, 5); load_w(std::integral_constant<int, 6>{}, 6); load_w(std::integral_constant<int, 7>{}, 7);
    load_a(0);
    store_a(0);
    if (npass > 1) load_a(XW_KP);
    __syncthreads();
    for (int ps = 0; ps < npass; ++ps) {
        const int buf = ps & 1;
        const int ks0 = ps * (XW_KP / 16);                              // first global 16-k step of the pass
        const int steps = min(XW_KP, p.K - ps * XW_KP) >> 4;          // 8; 2, 4 or 6 in a shorter last pass (K % 32 == 0)
        auto step = [&](auto s_tag) {
            constexpr int S = decltype(s_tag)::value;
            if (S < steps) {
                mma(buf, S, std::integral_constant<int, S>{});
                load_w(std::integral_constant<int, S>{}, ks0 + S + 8);
            }
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
        if (ps + 1 < npass) {
            store_a(buf ^ 1);                          // (buf ^ 1 was last read in pass ps - 1: every wave is past that pass's barrier)
            if (ps + 2 < npass) load_a((ps + 2) * XW_KP);
        }
        step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
        __syncthreads();
    }

    if (p.flags & GEMM_RMS_ROWSCALE) {
        // the row's eight running sums sit in lanes sj = 0 (positions 0-3) and sj = 1 (4-7): ((0+1)+(2+3)) + ((4+5)+(6+7))
        const float a = (ssq[0] + ssq[1]) + (ssq[2] + ssq[3]);
        const float b = __shfl_down(a, 1, 8);
        if (sj == 0) rowscale_s[srow] = rsqrtf((a + b) / (float)p.K + p.rms_eps);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
    const int gn = bn0 + wave * 32 + lrow;
    if (swiglu) {
        // packed layout: 32 value columns then their 32 gate columns: wave 2 g holds the values, wave 2 g + 1 the gates of group g -> the gate
        // wave parks (gate + bias) in LDS (the plane area is free now), the value wave finishes
        float* xch = reinterpret_cast<float*>(As) + (wave >> 1) * (XW_BM * 32);
        if (wave & 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int lr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
                    const float rs = (p.flags & GEMM_RMS_ROWSCALE) ? rowscale_s[lr] : 1.f;
                    float gate = (hi[i][e] + lo[i][e]) * rs;
                    if (p.bias && gn < p.N) gate += p.bias[gn];
                    xch[lr * 32 + lrow] = gate;
                }
        }
        __syncthreads();
        if (!(wave & 1) && gn < p.N) {
            const int on = (gn / 64) * 32 + (gn % 64);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int lr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
                    const int gm = bm0 + lr;
                    if (gm >= p.M) continue;
                    const float rs = (p.flags & GEMM_RMS_ROWSCALE) ? rowscale_s[lr] : 1.f;
                    float val = (hi[i][e] + lo[i][e]) * rs;
                    if (p.bias) val += p.bias[gn];
                    p.C[(int64_t)gm * p.ldc + on] = val * siluf(xch[lr * 32 + lrow]);
                }
        }
        return;
    }
    if (gn >= p.N) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int lr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
            const int gm = bm0 + lr;
            if (gm >= p.M) continue;
            const float rs = (p.flags & GEMM_RMS_ROWSCALE) ? rowscale_s[lr] : 1.f;
            float v = (hi[i][e] + lo[i][e]) * rs;
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

// ---- fp32 W [N][ldw] -> the fragment-ordered three-plane image [ceil(N / 32)][K / 16][3][64][8] (rows past N: zeros)
__global__ void split_bf16x3_tiled_kernel(const float* src, __bf16* dst, int N, int K, int ldw) {
    const int nk16 = K >> 4;
    const int64_t total = (int64_t)((N + 31) >> 5) * nk16 * 64;           // (tile, k step, lane) triples
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(i & 63);
        const int64_t tk = i >> 6;
        const int ks = (int)(tk % nk16), nt = (int)(tk / nk16);
        const int n = nt * 32 + (l & 31), k = ks * 16 + (l >> 5) * 8;
        bf16x8 o[3];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = n < N ? src[(int64_t)n * ldw + k + e] : 0.f;
            __bf16 h1, h2, h3;
            xw_split3(a, h1, h2, h3);
            o[0][e] = h1; o[1][e] = h2; o[2][e] = h3;
        }
        __bf16* d = dst + ((int64_t)tk * 3) * 512 + l * 8;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(d + pl * 512) = o[pl];
    }
}
int64_t split_bf16x3_tiled_elems(int N, int K) { return (int64_t)((N + 31) >> 5) * (K >> 4) * 3 * 512; }
int split_bf16x3_tiled(const float* src, uint16_t* dst, int N, int K, int ldw, hipStream_t s) {
    if (N == 0) return 0;
    D4_REQUIRE((K % 16) == 0 && ((uintptr_t)dst % 16) == 0, "split_bf16x3_tiled: K %% 16, 16-byte aligned image");
    const int64_t total = (int64_t)((N + 31) >> 5) * (K >> 4) * 64;
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(split_bf16x3_tiled_kernel, dim3((unsigned)g), dim3(256), 0, s, src, reinterpret_cast<__bf16*>(dst), N, K, ldw);
    D4_LAUNCH_CHECK();
    return 0;
}

// p.Wb = the fragment-ordered image of W (split_bf16x3_tiled); p.ldw / p.wplane unused
bool gemm_x3w_applicable(const GemmArgs& p) {
    return p.Wb != nullptr && !(p.flags & (GEMM_TRANS_A | GEMM_TRANS_B)) && (p.K % 32) == 0 && (p.lda % 4) == 0 && ((uintptr_t)p.Wb % 16) == 0 &&
           ((uintptr_t)p.A % 16) == 0 && p.batch <= 1 && (!(p.flags & GEMM_SWIGLU) || (p.N % 64) == 0);
}

int gemm_x3w_launch(const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    D4_REQUIRE(gemm_x3w_applicable(p), "gemm_x3w: call not supported (M=%d N=%d K=%d flags=%d)", p.M, p.N, p.K, p.flags);
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XW_LDS));
        attr_set.done();
    }
    const dim3 grid(cdiv(p.M, XW_BM) * cdiv(p.N, XW_BN)), block(XW_NT);
    if (ea) hipExtLaunchKernelGGL(gemm_x3w_kernel, grid, block, (uint32_t)XW_LDS, stream, ea, eb, 0, p);
    else hipLaunchKernelGGL(gemm_x3w_kernel, grid, block, XW_LDS, stream, p);
    D4_LAUNCH_CHECK();
    return 0;
}

}  // namespace d4
