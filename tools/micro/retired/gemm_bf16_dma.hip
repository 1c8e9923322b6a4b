// bf16 GEMM, second form: v_mfma_f32_32x32x16_bf16 fed by an LDS-DMA ring (the structure of gemm2.hip's fp32 kernel).
//
//   C[m, n] = epilogue( rowscale[m] * sum_k bf16(A[m, k]) * Wb[n, k] )          A fp32 [M][K], Wb bf16 [N][K], C fp32, K % 32 == 0
//
// The first form (gemm_bf16.hip) stages both operands global -> registers -> LDS and converts A on the way in: 9 % of the bf16 matrix
// peak on config 5's shapes, bound by that staging (VALU converts, ds_write_b128 at ~80 B/clk, two barriers per k-tile).  Here both
// tiles go global -> LDS directly (`buffer_load_dwordx4 ... lds`, no VGPR, no ds_write): the W tile as bf16 [rows][32] (64-byte rows),
// the A tile as the fp32 it is in HBM [rows][32] (128-byte rows), into a ring of NS stages with counted `s_waitcnt vmcnt(N)` and one
// raw s_barrier per k-tile.  A fragment (8 consecutive k of one row per lane) is read as two ds_read_b128 of fp32 and rounded to bf16
// in registers (round-to-nearest-even, the same values the first form stores), a W fragment is one ds_read_b128.  The folded RMSNorm's
// row sums are taken from the fp32 tile in LDS.  16-byte chunks are XOR-swizzled on the DMA's SOURCE address so that the 16 lanes a
// ds_read_b128 services together land on 16 distinct slots: A chunk ^ ((row >> 1) & 7), W chunk ^ ((row >> 2) & 3).
// Same epilogue as the first form.  Which form runs a call is a rule on its shape (gemm_bf16.hip), never a timing.
#include "common.h"
#include <hip/hip_ext.h>
#include "kernels.h"
#include <stdlib.h>

namespace d4 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int N>
__device__ __forceinline__ void wait_vmcnt_b() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WGM, int WGN, int TM, int TN, int NS, bool RMS>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_bf16_dma_kernel(GemmArgs p) {
    constexpr int BK = 32;
    constexpr int NW = WGM * WGN, NT = NW * 64;
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int A_BYTES = BM * BK * 4, W_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int NSLOT_A = A_BYTES / 1024, NSLOT = STAGE_BYTES / 1024;        // 1 KB DMA pieces: 8 A rows or 16 W rows each
    constexpr int LPW = (NSLOT + NW - 1) / NW;
    static_assert(NS >= 3 && NS <= 4 && 2 * LPW <= 63, "ring depth / vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem_b[];              // [NS][A fp32 tile | W bf16 tile] | rowscale[BM]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc(p.A + (int64_t)bm0 * p.lda, ((int64_t)(rowsA - 1) * p.lda + p.K) * 4);
    const __amdgpu_buffer_rsrc_t rsB = uniform_rsrc(Wb + (int64_t)bn0 * p.ldw, ((int64_t)(rowsB - 1) * p.ldw + p.K) * 2);

    // this lane's part of each DMA piece.  A piece: row lane / 8 of 8 rows, LDS chunk lane % 8 <- source chunk ^ ((row >> 1) & 7);
    // W piece: row lane / 4 of 16 rows, LDS chunk lane % 4 <- source chunk ^ ((row >> 2) & 3)
    uint32_t voff[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int slot = min(wave + NW * i, NSLOT - 1);
        if (slot < NSLOT_A) {
            const int r = slot * 8 + lane / 8;
            voff[i] = (uint32_t)(r * p.lda * 4 + (((lane % 8) ^ ((r >> 1) & 7)) * 16));
        } else {
            const int r = (slot - NSLOT_A) * 16 + lane / 4;
            voff[i] = (uint32_t)(r * p.ldw * 2 + (((lane % 4) ^ ((r >> 2) & 3)) * 16));
        }
    }
#define D4_ISSUE_STAGE_B(KT, BUF)                                                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < LPW; ++i_) {                                                                      \
        const int slot_ = min(wave + NW * i_, NSLOT - 1);                                                                     \
        char* dst_ = smem_b + (BUF) * STAGE_BYTES + slot_ * 1024;                                                             \
        if (slot_ < NSLOT_A) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_ptr)dst_, 16, (uint32_t)voff[i_], (KT) * BK * 4, 0, 0); \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_ptr)dst_, 16, (uint32_t)voff[i_], (KT) * BK * 2, 0, 0);  \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addresses (bytes inside a stage): lane (row = lane & 31, g = lane >> 5); k-step s of the tile covers k = 16 s + 8 g .. + 8
    const int lrow = lane & 31, lg = lane >> 5;
    int a_off[2][TM], w_off[2][TN];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm * TM * 32 + i * 32 + lrow;
            a_off[s][i] = r * 128 + (((s * 4 + lg * 2) ^ ((r >> 1) & 7)) * 16);          // second half of the 8 k: chunk ^ 1 -> byte offset ^ 16
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = wn * TN * 32 + j * 32 + lrow;
            w_off[s][j] = A_BYTES + r * 64 + (((s * 2 + lg) ^ ((r >> 2) & 3)) * 16);
        }
    }

    constexpr int SQI = RMS ? (BM * 8 + NT - 1) / NT : 1;          // 16-byte chunks of the fp32 A tile per thread (row sums)
    float ssq[SQI];
#pragma unroll
    for (int i = 0; i < SQI; ++i) ssq[i] = 0.f;

    const int nk = p.K / BK;
    auto wait_allow = [&](int stages) {
        if (stages >= 2) wait_vmcnt_b<2 * LPW>();
        else if (stages == 1) wait_vmcnt_b<LPW>();
        else wait_vmcnt_b<0>();
    };
    auto cvt8 = [](const f32x4 lo, const f32x4 hi) {
        bf16x8 o;
        o[0] = (__bf16)lo[0]; o[1] = (__bf16)lo[1]; o[2] = (__bf16)lo[2]; o[3] = (__bf16)lo[3];
        o[4] = (__bf16)hi[0]; o[5] = (__bf16)hi[1]; o[6] = (__bf16)hi[2]; o[7] = (__bf16)hi[3];
        return o;
    };

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) { D4_ISSUE_STAGE_B(s, s) }

    for (int kt = 0; kt < nk; ++kt) {
        wait_allow(min(kt + NS - 2, nk - 1) - kt);              // this wave's pieces of k-tile kt have landed
        __builtin_amdgcn_s_barrier();                           // ... everyone's have; and everyone is done reading k-tile kt-1
        if (kt + NS - 1 < nk) { D4_ISSUE_STAGE_B(kt + NS - 1, (kt + NS - 1) % NS) }
        const char* st = smem_b + (kt % NS) * STAGE_BYTES;
        if constexpr (RMS) {
#pragma unroll
            for (int i = 0; i < SQI; ++i) {
                const int idx = tid + i * NT;
                if (BM * 8 % NT == 0 || idx < BM * 8) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(st + idx * 16);
                    ssq[i] = ssq[i] + ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = cvt8(*reinterpret_cast<const f32x4*>(st + a_off[s][i]), *reinterpret_cast<const f32x4*>(st + (a_off[s][i] ^ 16)));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(st + w_off[s][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
#undef D4_ISSUE_STAGE_B

    float* rowscale_s = reinterpret_cast<float*>(smem_b + NS * STAGE_BYTES);
    if constexpr (RMS) {
        // a row's 8 chunks sit on 8 consecutive threads (idx = row * 8 + chunk slot): fold them in a fixed order
#pragma unroll
        for (int i = 0; i < SQI; ++i) {
            float s = ssq[i];
            s += dpp_f<0xB1>(s);
            s += dpp_f<0x4E>(s);
            s += dpp_f<0x141>(s);
            const int idx = tid + i * NT;
            if ((idx & 7) == 0 && idx < BM * 8) rowscale_s[idx >> 3] = rsqrtf(s / (float)p.K + p.rms_eps);
        }
        __syncthreads();
    }

    // ---- epilogue.  32x32 C/D layout: col = lane & 31, row = 8 * (r / 4) + 4 * (lane >> 5) + (r & 3), r = 0 .. 15
    const int ccol = lane & 31, chalf = lane >> 5;
    const bool swiglu = (p.flags & GEMM_SWIGLU) != 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * TM * 32 + i * 32 + 8 * (r >> 2) + 4 * chalf + (r & 3);
            const int gm = bm0 + ml;
            if (gm >= p.M) continue;
            const float rs = RMS ? rowscale_s[ml] : 1.f;
            int64_t c2row = -1;
            if (p.C2) {
                const int ts = gm % p.c2_S, keep = p.c2_hi - p.c2_lo;
                const int rank = (ts >= p.c2_lo && ts < p.c2_hi) ? ts - p.c2_lo : ((p.c2_last && ts == p.c2_S - 1) ? keep : -1);
                if (rank >= 0) c2row = (int64_t)(gm / p.c2_S) * (keep + p.c2_last) + rank;
            }
            if (swiglu) {
                if constexpr (TN % 2 == 0) {
#pragma unroll
                    for (int j = 0; j < TN; j += 2) {              // sub-tile j = 32 values, j + 1 = their 32 gates (packed pairs)
                        const int gn = bn0 + wn * TN * 32 + j * 32 + ccol;
                        if (gn >= p.N) continue;
                        float val = acc[i][j][r] * rs, gate = acc[i][j + 1][r] * rs;
                        if (p.bias) { val += p.bias[gn]; gate += p.bias[gn + 32]; }
                        p.C[(int64_t)gm * p.ldc + (gn / 64) * 32 + (gn % 64)] = val * siluf(gate);
                    }
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int gn = bn0 + wn * TN * 32 + j * 32 + ccol;
                if (gn >= p.N) continue;
                float v = acc[i][j][r] * rs;
                if (p.bias) v += p.bias[gn];
                if (p.flags & GEMM_SILU) v = siluf(v);
                if (p.R) v += p.R[(int64_t)gm * p.ldr + gn];
                float* cp = p.C + (int64_t)gm * p.ldc + gn;
                if (p.flags & GEMM_ACCUMULATE) v += *cp;
                *cp = v;
                if (c2row >= 0) p.C2[c2row * p.ldc2 + gn] = v;
            }
        }
}

// ---- configurations: name, waves (M x N), wave tile, block tile, LDS ring
//   D128x128     2 x 2   64 x 64   128 x 128   3 x 24 KB  (2 blocks / CU)
//   D128x256     2 x 2   64 x 128  128 x 256   3 x 32 KB  (1 block / CU)
//   D64x128      1 x 4   64 x 32    64 x 128   4 x 16 KB  (2 blocks / CU)
//   D256x128_8   4 x 2   64 x 64   256 x 128   3 x 40 KB  (1 block / CU, 8 waves)
//   D128x128_8   4 x 2   32 x 64   128 x 128   3 x 24 KB  (2 blocks / CU, 8 waves each)
enum { BD_128x128 = 0, BD_128x256, BD_64x128, BD_256x128_8, BD_128x128_8, BD_N };
static const char* const kBdName[BD_N] = {"gemm_bf16_dma_kernel<2, 2, 2, 2, 3", "gemm_bf16_dma_kernel<2, 2, 2, 4, 3", "gemm_bf16_dma_kernel<1, 4, 2, 1, 4",
                                          "gemm_bf16_dma_kernel<4, 2, 2, 2, 3", "gemm_bf16_dma_kernel<4, 2, 1, 2, 3"};
int gemm_bf16_dma_configs() { return BD_N; }
const char* gemm_bf16_dma_config_name(int c) { return c >= 0 && c < BD_N ? kBdName[c] : ""; }

bool gemm_bf16_dma_applicable(const GemmArgs& p) {
    return p.Wb != nullptr && !(p.flags & (GEMM_TRANS_A | GEMM_TRANS_B)) && (p.K % 32) == 0 && (p.lda % 4) == 0 && (p.ldw % 8) == 0 &&
           ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.Wb % 16) == 0 && p.M >= 1;
}

bool gemm_bf16_dma_config_valid(int c, const GemmArgs& p) {
    if (c < 0 || c >= BD_N || !gemm_bf16_dma_applicable(p)) return false;
    if (p.flags & GEMM_SWIGLU) return c != BD_64x128;                // the SiLU-GLU pairing needs a wave to span 64 columns
    return true;
}

template <int WGM, int WGN, int TM, int TN, int NS>
static int launch_bd(const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    const size_t lds = (size_t)NS * (BM * 128 + BN * 64) + BM * sizeof(float);
    const bool rms = (p.flags & GEMM_RMS_ROWSCALE) != 0;
    auto k = rms ? gemm_bf16_dma_kernel<WGM, WGN, TM, TN, NS, true> : gemm_bf16_dma_kernel<WGM, WGN, TM, TN, NS, false>;
    static DeviceOnce attr_set[2];
    if (attr_set[rms].need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[rms].done();
    }
    const dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), p.batch > 0 ? p.batch : 1), block(WGM * WGN * 64);
    if (ea) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)lds, stream, ea, eb, 0, p);
    else hipLaunchKernelGGL(k, grid, block, lds, stream, p);
    D4_LAUNCH_CHECK();
    return 0;
}

int gemm_bf16_dma_launch(int c, const GemmArgs& p, hipStream_t stream, hipEvent_t ea, hipEvent_t eb) {
    D4_REQUIRE(gemm_bf16_dma_config_valid(c, p), "gemm_bf16_dma: configuration %d is not valid for this call", c);
    switch (c) {
        case BD_128x128: return launch_bd<2, 2, 2, 2, 3>(p, stream, ea, eb);
        case BD_128x256: return launch_bd<2, 2, 2, 4, 3>(p, stream, ea, eb);
        case BD_64x128: return launch_bd<1, 4, 2, 1, 4>(p, stream, ea, eb);
        case BD_256x128_8: return launch_bd<4, 2, 2, 2, 3>(p, stream, ea, eb);
        case BD_128x128_8: return launch_bd<4, 2, 1, 2, 3>(p, stream, ea, eb);
    }
    return 2;
}

}  // namespace d4
