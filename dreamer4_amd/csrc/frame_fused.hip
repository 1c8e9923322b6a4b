// Per-frame fused tails of the trunk's blocks (round 3): one workgroup owns the <= 16 token rows of ONE frame.
//
// At B = 256 trajectories a frame is 14-15 token rows and there is one frame per CU, so the SMALL GEMMs of a layer (output projection
// 512 x 512, the attention pool's value / output projections) are launches of 1-2 GFLOP whose fixed cost (ramp, first-tile latency,
// epilogue: ~6.5 us) is a third of their run time: the tiled kernels reach 0.35-0.45 of the matrix peak on them.  Here such a GEMM runs
// as the tail of the kernel that produces its input, per frame: the activations stay in LDS, the weight matrix is streamed global ->
// VGPR -> v_mfma_f32_16x16x4_f32 by frame_gemm.h (0.70 of the nominal peak measured, profiles/r03_frame_ff_bench.txt), re-tiled once
// at prepare time (tile16_weights) so that every wave-load is one contiguous KB, and its stream is requested BEFORE the producing phase
// starts.  The large GEMMs (fused q|k|v projection, SiLU-GLU feedforward, pool key projection) stay on the tiled kernels, which beat
// the stream there.
//
//   frame_attn_out_kernel   within-frame attention (attn_mfma_unit, one wave per head) -> output projection + residual (+ the
//                           row-compacted copy)                                              D4:2005-2068
//   frame_pool_tail_kernel  AttentionPool tail: per-head value projection of the mixed hiddens -> output projection + residual
//                           (+ the row-compacted copy)                                       D4:2160-2177
#include "attn_mfma.h"
#include "pool_mix_row.h"
#include "frame_gemm.h"
#include "prof.h"
#include <hip/hip_ext.h>
#include <stdlib.h>

namespace d4 {

// ---- W [N][K] row-major -> Wt [N / 16][K / 4][16][4]  (N % 16 == 0, K % 4 == 0)
__global__ void tile16_kernel(const float* W, float* Wt, int N, int K, int ldw) {
    const int64_t n4 = (int64_t)N * (K / 4);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i % 16), c = (int)((i / 16) % (K / 4)), t = (int)(i / (16 * (int64_t)(K / 4)));
        reinterpret_cast<f32x4*>(Wt)[i] = *reinterpret_cast<const f32x4*>(W + (int64_t)(t * 16 + r) * ldw + c * 4);
    }
}
int tile16_weights(const float* W, int ldw, float* Wt, int N, int K, hipStream_t s) {
    D4_REQUIRE(N % 16 == 0 && K % 4 == 0 && ldw % 4 == 0, "tile16_weights: N %% 16, K %% 4, ldw %% 4");
    const int64_t n4 = (int64_t)N * (K / 4);
    hipLaunchKernelGGL(tile16_kernel, dim3((unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048)), dim3(256), 0, s, W, Wt, N, K, ldw);
    D4_LAUNCH_CHECK();
    return 0;
}

// row `ts` of a frame of c2_S tokens -> its row in the compacted copy, or -1 (the rule of the GEMM epilogues' C2 output)
__device__ __forceinline__ int compact_rank(int ts, int c2_S, int lo, int hi, int last) {
    return (ts >= lo && ts < hi) ? ts - lo : ((last && ts == c2_S - 1) ? hi - lo : -1);
}

// out[frame rows][n .. n + 3] = acc + residual, plus the compacted copy: shared epilogue of both kernels
struct FrameOut {
    const float* resid; int ldr;       // [frames * S][ldr]
    float* out; int ldo;               // [frames * S][ldo]
    float* c2; int ldc2, c2_lo, c2_hi, c2_last;      // optional compacted copy [frames * (hi - lo + last)][ldc2]
    int S;
};

constexpr int FF_NW = 8;

template <int HD>
__global__ __launch_bounds__(FF_NW * 64) void frame_attn_out_kernel(SmallAttnArgs sa, const float* __restrict__ wo_t, int D, FrameOut fo) {
    constexpr int LDA = HD + 4;
    extern __shared__ __attribute__((aligned(16))) float smem_ao[];
    float* As = smem_ao;                                                               // [16][LDA] attention output of the frame, all heads
    float* Vs_all = As + 16 * LDA;                                                     // [waves][16][SM_LDV]
    float* kinv_all = Vs_all + FF_NW * 16 * SM_LDV;                                    // [waves][16]
    float* vinv_all = kinv_all + FF_NW * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x;
    const int li = lane & 15, kq = lane >> 4;
    const int NU = D / 32;                                                             // units of two 16-column tiles
    auto unit = [&](int u) { return fg_make_unit(wo_t, HD, 2 * u, 2 * u + 1, lane); };
    FgRing ring;
    if (wave < NU) fg_prefetch(ring, unit(wave));                                      // the weight stream starts before the attention does

    for (int h = wave; h < sa.heads; h += FF_NW)
        attn_mfma_unit<1, 1>(sa, g, h, lane, Vs_all + wave * 16 * SM_LDV, kinv_all + wave * 16, vinv_all + wave * 16,
                             [&](int orank, int t, int tok, float v) { As[orank * LDA + h * 64 + 16 * t + tok] = v; });
    __syncthreads();

    const float* a_lds = As + li * LDA + 4 * kq;
    const int64_t grow = (int64_t)g * fo.S + li;
    const bool row_ok = li < fo.S;
    const int rank = (fo.c2 && row_ok) ? compact_rank(li, fo.S, fo.c2_lo, fo.c2_hi, fo.c2_last) : -1;
    const int64_t c2row = (int64_t)g * (fo.c2_hi - fo.c2_lo + fo.c2_last) + rank;
    for (int u = wave; u < NU; u += FF_NW) {
        const FgUnit nxt = unit(u + FF_NW < NU ? u + FF_NW : u);
        f32x4 r4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)                                                    // residual: requested before the unit's weight stream
            r4[t] = row_ok ? *reinterpret_cast<const f32x4*>(fo.resid + grow * fo.ldr + 32 * u + 16 * t + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 acc0, acc1;
        fg_unit<HD>(ring, unit(u), nxt, a_lds, acc0, acc1);
        if (row_ok) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = 32 * u + 16 * t + 4 * kq;
                const f32x4 a = t ? acc1 : acc0;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = a[e] + r4[t][e];
                *reinterpret_cast<f32x4*>(fo.out + grow * fo.ldo + n) = o;
                if (rank >= 0) *reinterpret_cast<f32x4*>(fo.c2 + c2row * fo.ldc2 + n) = o;
            }
        }
    }
}

// AttentionPool tail.  u [frames * S][heads][D] (the gated softmax-weighted mixes of the normalised hiddens, pool_mix_kernel);
// wv_t: the value projection [heads * 64][D] tiled (K = D); wo_t: the pool's output projection [D][heads * 64] tiled (K = heads * 64).
template <int DD, int PH>
__global__ __launch_bounds__(FF_NW * 64) void frame_pool_tail_kernel(const float* __restrict__ u, const float* __restrict__ wv_t, const float* __restrict__ wo_t, FrameOut fo) {
    constexpr int LDU = DD + 4, HP = PH * 64, LDP = HP + 4;
    extern __shared__ __attribute__((aligned(16))) float smem_pt[];
    float* Us = smem_pt;                          // [PH][16][LDU]
    float* Ps = Us + PH * 16 * LDU;               // [16][LDP]   pooled attention output of the frame
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x;
    const int li = lane & 15, kq = lane >> 4;
    constexpr int NU1 = HP / 32;                  // value-projection units: two 16-row tiles of ONE head each (64 rows per head = 2 units)
    const int NU2 = DD / 32;
    auto unit1 = [&](int v) { return fg_make_unit(wv_t, DD, 2 * v, 2 * v + 1, lane); };
    auto unit2 = [&](int v) { return fg_make_unit(wo_t, HP, 2 * v, 2 * v + 1, lane); };
    FgRing ring;
    if (wave < NU1) fg_prefetch(ring, unit1(wave));
    else fg_prefetch(ring, unit2(wave));          // (never with 4 pool heads: NU1 = 8 = waves)

    // stage the frame's mixes: row m, head h -> Us[h][m][:]   (rows >= S are zero)
    for (int i = tid; i < PH * 16 * (DD / 4); i += FF_NW * 64) {
        const int c4 = i % (DD / 4), m = (i / (DD / 4)) % 16, h = i / (16 * (DD / 4));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < fo.S) v = *reinterpret_cast<const f32x4*>(u + (((int64_t)g * fo.S + m) * PH + h) * DD + c4 * 4);
        *reinterpret_cast<f32x4*>(Us + (h * 16 + m) * LDU + c4 * 4) = v;
    }
    __syncthreads();
    f32x4 acc0, acc1;
    for (int v = wave; v < NU1; v += FF_NW) {
        const int h = v / 2;                                                           // head of this unit (64 value rows per head)
        const FgUnit nxt = v + FF_NW < NU1 ? unit1(v + FF_NW) : unit2(wave);           // then straight on into the output projection's weights
        fg_unit<DD>(ring, unit1(v), nxt, Us + (h * 16 + li) * LDU + 4 * kq, acc0, acc1);
        *reinterpret_cast<f32x4*>(Ps + li * LDP + 32 * v + 4 * kq) = acc0;
        *reinterpret_cast<f32x4*>(Ps + li * LDP + 32 * v + 16 + 4 * kq) = acc1;
    }
    __syncthreads();
    const float* a_lds = Ps + li * LDP + 4 * kq;
    const int64_t grow = (int64_t)g * fo.S + li;
    const bool row_ok = li < fo.S;
    const int rank = (fo.c2 && row_ok) ? compact_rank(li, fo.S, fo.c2_lo, fo.c2_hi, fo.c2_last) : -1;
    const int64_t c2row = (int64_t)g * (fo.c2_hi - fo.c2_lo + fo.c2_last) + rank;
    for (int v = wave; v < NU2; v += FF_NW) {
        const FgUnit nxt = unit2(v + FF_NW < NU2 ? v + FF_NW : v);
        f32x4 r4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
            r4[t] = row_ok ? *reinterpret_cast<const f32x4*>(fo.resid + grow * fo.ldr + 32 * v + 16 * t + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        fg_unit<HP>(ring, unit2(v), nxt, a_lds, acc0, acc1);
        if (row_ok) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = 32 * v + 16 * t + 4 * kq;
                const f32x4 a = t ? acc1 : acc0;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = a[e] + r4[t][e];
                *reinterpret_cast<f32x4*>(fo.out + grow * fo.ldo + n) = o;
                if (rank >= 0) *reinterpret_cast<f32x4*>(fo.c2 + c2row * fo.ldc2 + n) = o;
            }
        }
    }
}

// The whole AttentionPool core + tail per frame: the softmax-weighted, gated mixes of the normalised layer hiddens (pool_mix_row, one wave per
// token row) go straight into LDS instead of through HBM (29 MB written and read back per pool at B = 256), then the tail above.
// NW waves: 8, or 16 (round 6) — with 16 every token row of the frame has a wave of its own in the mix phase (8 waves: two rows one after the other on six of them)
// and the output projection's 16 units are one per wave; LM = rows of the per-wave score scratch (>= the number of hiddens).
template <int DD, int PH, int NW, int LM>
__global__ __launch_bounds__(NW * 64) void frame_pool_kernel(PoolMixArgs pm, const float* __restrict__ wv_t, const float* __restrict__ wo_t, FrameOut fo) {
    constexpr int LDU = DD + 4, HP = PH * 64, LDP = HP + 4, ITER = DD / 256;
    extern __shared__ __attribute__((aligned(16))) float smem_pk[];
    float* Us = smem_pk;                          // [PH][16][LDU]
    float* Ps = Us + PH * 16 * LDU;               // [16][LDP]   pooled attention output of the frame (second phase)
    f32x4* gws = reinterpret_cast<f32x4*>(Ps);    // first phase only: head-gate weights [PH][ITER * 64] float4 ...
    float* psh = Ps + 16 * LDP;                   // ... and, behind the tile, the per-wave score scratch [waves][LM * PH]
    static_assert(PH * ITER * 64 * 4 <= 16 * LDP, "first-phase scratch must fit the second phase's tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x;
    const int li = lane & 15, kq = lane >> 4;
    constexpr int NU1 = HP / 32;
    const int NU2 = DD / 32;
    auto unit1 = [&](int v) { return fg_make_unit(wv_t, DD, 2 * v, 2 * v + 1, lane); };
    auto unit2 = [&](int v) { return fg_make_unit(wo_t, HP, 2 * v, 2 * v + 1, lane); };
    FgRing ring;                                  // (filled AFTER the mix phase: its 64 registers held across the mix cost more — 187 vs 125 VGPRs —
                                                  //  than the head start of the weight stream gives; 180.5 vs 181.3 ms per rollout, round 4)
    constexpr int nf4 = DD / 4;
    for (int i = tid; i < PH * ITER * 64; i += NW * 64) {
        const int h = i / (ITER * 64), c4 = i % (ITER * 64);
        gws[i] = c4 < nf4 ? reinterpret_cast<const f32x4*>(pm.gate_w)[h * nf4 + c4] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = tid; i < PH * (16 - fo.S) * (DD / 4); i += NW * 64) {               // rows >= S of the mixes are zero
        const int c4 = i % (DD / 4), r = (i / (DD / 4)) % (16 - fo.S), h = i / ((16 - fo.S) * (DD / 4));
        *reinterpret_cast<f32x4*>(Us + (h * 16 + fo.S + r) * LDU + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    for (int ml = wave; ml < fo.S; ml += NW)
        pool_mix_row<ITER, true>(pm, g * fo.S + ml, lane, psh + wave * LM * PH, gws,
                           [&](int h, int c4, const f32x4& v) { *reinterpret_cast<f32x4*>(Us + (h * 16 + ml) * LDU + c4 * 4) = v; });
    if (wave < NU1) fg_prefetch(ring, unit1(wave));
    else fg_prefetch(ring, unit2(wave));          // (16 waves: the upper eight go straight to the output projection's weights)
    __syncthreads();

    f32x4 acc0, acc1;
    for (int v = wave; v < NU1; v += NW) {
        const int h = v / 2;
        const FgUnit nxt = v + NW < NU1 ? unit1(v + NW) : unit2(wave);
        fg_unit<DD>(ring, unit1(v), nxt, Us + (h * 16 + li) * LDU + 4 * kq, acc0, acc1);
        *reinterpret_cast<f32x4*>(Ps + li * LDP + 32 * v + 4 * kq) = acc0;
        *reinterpret_cast<f32x4*>(Ps + li * LDP + 32 * v + 16 + 4 * kq) = acc1;
    }
    __syncthreads();
    const float* a_lds = Ps + li * LDP + 4 * kq;
    const int64_t grow = (int64_t)g * fo.S + li;
    const bool row_ok = li < fo.S;
    const int rank = (fo.c2 && row_ok) ? compact_rank(li, fo.S, fo.c2_lo, fo.c2_hi, fo.c2_last) : -1;
    const int64_t c2row = (int64_t)g * (fo.c2_hi - fo.c2_lo + fo.c2_last) + rank;
    for (int v = wave; v < NU2; v += NW) {
        const FgUnit nxt = unit2(v + NW < NU2 ? v + NW : v);
        f32x4 r4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
            r4[t] = row_ok ? *reinterpret_cast<const f32x4*>(fo.resid + grow * fo.ldr + 32 * v + 16 * t + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        fg_unit<HP>(ring, unit2(v), nxt, a_lds, acc0, acc1);
        if (row_ok) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = 32 * v + 16 * t + 4 * kq;
                const f32x4 a = t ? acc1 : acc0;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = a[e] + r4[t][e];
                *reinterpret_cast<f32x4*>(fo.out + grow * fo.ldo + n) = o;
                if (rank >= 0) *reinterpret_cast<f32x4*>(fo.c2 + c2row * fo.ldc2 + n) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Few frames (the launch-bound decode regime, BASELINE config 4: one trajectory = one frame of 11 token rows): within-frame attention ->
// output projection + residual in ONE launch, column-split.  A workgroup owns 16 output columns of ONE frame and all of K = heads * 64:
//   * its weight rows are requested first (8 waves x 4 k-steps x 16 columns: the launch's longest memory round trip);
//   * every wave then runs one head's attention of the frame on the matrix pipe (attn_mfma_unit) into LDS — the attention is RECOMPUTED by
//     each of the D / 16 column workgroups of a frame: a few thousand MFMA cycles that hide under the weight fetch, instead of a launch;
//   * the product runs as gemm_skinny_kernel<false, 8, 4> does (the 8 waves split K, partial tiles folded through LDS in wave order, the
//     same permuted-k MFMA feed), with its A operand read from LDS: BIT-IDENTICAL to attn_mfma_kernel followed by the few-row GEMM.
constexpr int AOC_NW = 8, AOC_U = 4, AOC_RED = 17;
template <int HD>
__global__ __launch_bounds__(AOC_NW * 64) void attn_out_cols_kernel(SmallAttnArgs sa, const float* __restrict__ W, int ldw, int D, FrameOut fo) {
    static_assert(HD == AOC_NW * AOC_U * 16, "8 waves x 4 k-steps of 16 cover the attention width");
    constexpr int LDA = HD + 4;
    extern __shared__ __attribute__((aligned(16))) float smem_aoc[];
    float* As = smem_aoc;                                                              // [16][LDA]
    float* Vs_all = As + 16 * LDA;                                                     // [waves][16][SM_LDV]
    float* kinv_all = Vs_all + AOC_NW * 16 * SM_LDV;
    float* vinv_all = kinv_all + AOC_NW * 16;
    float* red = vinv_all + AOC_NW * 16;                                               // [waves][16][17]
    const int tid = threadIdx.x, lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x, g = blockIdx.y;
    const int nl = lane & 15, kk = lane >> 4;
    const int n = bx * 16 + nl;
    const bool valid = n < D;
    const float* wrow = W + (int64_t)(valid ? n : 0) * ldw;
    f32x4 w4[AOC_U];
#pragma unroll
    for (int u = 0; u < AOC_U; ++u) w4[u] = valid ? *reinterpret_cast<const f32x4*>(wrow + 16 * (q * AOC_U + u) + 4 * kk) : f32x4{0.f, 0.f, 0.f, 0.f};
    // the epilogue's residual: thread = (row, column), requested with the weights
    const int c = tid & 15, ml = tid >> 4;
    const int gn = bx * 16 + c;
    const int64_t grow = (int64_t)g * fo.S + ml;
    float e_res = 0.f;
    if (tid < 256 && ml < fo.S && gn < D) e_res = fo.resid[grow * fo.ldr + gn];
    // rows past the frame's tokens are never written by the attention: they must read as zeros
    for (int i = tid; i < (16 - sa.nq) * LDA; i += AOC_NW * 64) As[sa.nq * LDA + i] = 0.f;

    for (int h = q; h < sa.heads; h += AOC_NW)
        attn_mfma_unit<1, 1>(sa, g, h, lane, Vs_all + q * 16 * SM_LDV, kinv_all + q * 16, vinv_all + q * 16,
                             [&](int orank, int t, int tok, float v) { As[orank * LDA + h * 64 + 16 * t + tok] = v; });
    __syncthreads();

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < AOC_U; ++u) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(As + nl * LDA + 16 * (q * AOC_U + u) + 4 * kk);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w4[u][e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(q * 16 + 4 * kk + r) * AOC_RED + nl] = acc[r];
    __syncthreads();
    if (tid >= 256 || ml >= fo.S || gn >= D) return;
    float v = red[ml * AOC_RED + c];
#pragma unroll
    for (int w = 1; w < AOC_NW; ++w) v += red[(w * 16 + ml) * AOC_RED + c];
    v += e_res;
    fo.out[grow * fo.ldo + gn] = v;
    if (fo.c2) {
        const int rank = compact_rank(ml, fo.S, fo.c2_lo, fo.c2_hi, fo.c2_last);
        if (rank >= 0) fo.c2[((int64_t)g * (fo.c2_hi - fo.c2_lo + fo.c2_last) + rank) * fo.ldc2 + gn] = v;
    }
}

// ---- launchers --------------------------------------------------------------------------------------------------
static int g_frame_fused = 1;          // 0 off; 1 on (default); 2 tails only (the pool mix stays its own kernel) — test hook d4_frame_fused_set
int g_attn_out_cols = 1;               // test hook d4_debug_switch("attn_out_cols"): 0 keeps attention and out-projection as two launches at <= 4 frames
int frame_fused_mode() { return g_frame_fused; }
int frame_fused_set(int mode) { const int old = frame_fused_mode(); g_frame_fused = mode; return old; }
static bool frame_fused_on() { return frame_fused_mode() != 0; }

// by shape only: head dim 64 x 8 heads, <= 16 tokens, at least 3/4 of the CUs get a frame (one workgroup per frame: fewer frames leave CUs
// idle and the tiled kernels win), no more than four rounds of frames
bool frame_fused_frames_ok(int frames) { return frame_fused_on() && frames >= 192 && frames <= 1024; }

bool frame_attn_out_applicable(const SmallAttnArgs& sa, int D) {
    auto al4 = [](const void* q, int64_t a, int64_t b) { return ((uintptr_t)q % 16) == 0 && (a % 4) == 0 && (b % 4) == 0; };
    return frame_fused_frames_ok(sa.groups) && sa.dh == 64 && sa.heads == 8 && sa.nq == sa.nk && sa.nk >= 1 && sa.nk <= 16 && sa.q_hi == 0 && D % 32 == 0 && D >= 256 &&
           al4(sa.q, sa.q_group_stride, sa.q_item_stride) && al4(sa.k, sa.k_group_stride, sa.k_item_stride) && al4(sa.v, sa.v_group_stride, sa.v_item_stride) &&
           (!sa.vres || al4(sa.vres, sa.r_group_stride, sa.r_item_stride)) && ((uintptr_t)sa.k_gamma % 16) == 0;
}

int frame_attn_out(const SmallAttnArgs& sa, const float* wo_t, int D, const float* resid, int ldr, float* out, int ldo, float* c2, int ldc2, int c2_lo,
                   int c2_hi, int c2_last, hipStream_t s) {
    D4_REQUIRE(frame_attn_out_applicable(sa, D) && ldr % 4 == 0 && ldo % 4 == 0 && (!c2 || ldc2 % 4 == 0), "frame_attn_out: call not supported");
    FrameOut fo{resid, ldr, out, ldo, c2, ldc2, c2_lo, c2_hi, c2_last, sa.nk};
    const int hd = sa.heads * 64;
    // algorithmic bytes: q, k, v, value residual and the block input read once, the block output written once (+ the weight matrix once per XCD)
    const double bytes = 4.0 * sa.groups * sa.nk * ((sa.vres ? 4.0 : 3.0) * hd + 2.0 * D) + 8.0 * 4.0 * D * hd;
    const double flops = 2.0 * sa.groups * sa.nk * (double)D * hd;
    const size_t lds = (size_t)(16 * (512 + 4) + FF_NW * 16 * SM_LDV + 2 * FF_NW * 16) * sizeof(float);
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(frame_attn_out_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.done();
    }
    D4_GLUE_LAUNCH_F(GL_FRAME_ATTN_OUT, bytes, flops, (frame_attn_out_kernel<512>), dim3(sa.groups), dim3(FF_NW * 64), lds, s, sa, wo_t, D, fo);
    D4_LAUNCH_CHECK();
    return 0;
}

// few frames: the column-split form above.  By shape only: 8 heads x 64, <= 16 tokens, at most 4 frames — measured on one box (tools/cfg4_ab.sh,
// BASELINE config 4): B = 1 1.73 -> 1.69 ms per env step, but B = 16 2.79 -> 2.87 ms: 16 frames x 32 column workgroups each recomputing a frame's
// attention cost more than the launch they save
bool attn_out_cols_applicable(const SmallAttnArgs& sa, int D) {
    const bool on = g_attn_out_cols != 0;
    auto al4 = [](const void* q, int64_t a, int64_t b) { return ((uintptr_t)q % 16) == 0 && (a % 4) == 0 && (b % 4) == 0; };
    return on && frame_fused_on() && sa.groups >= 1 && sa.groups <= 4 && sa.dh == 64 && sa.heads == 8 && sa.nq == sa.nk && sa.nk >= 1 && sa.nk <= 16 && sa.q_hi == 0 &&
           D % 16 == 0 && !sa.out_b &&
           al4(sa.q, sa.q_group_stride, sa.q_item_stride) && al4(sa.k, sa.k_group_stride, sa.k_item_stride) && al4(sa.v, sa.v_group_stride, sa.v_item_stride) &&
           (!sa.vres || al4(sa.vres, sa.r_group_stride, sa.r_item_stride)) && ((uintptr_t)sa.k_gamma % 16) == 0;
}

int attn_out_cols(const SmallAttnArgs& sa, const float* W, int ldw, int D, const float* resid, int ldr, float* out, int ldo, float* c2, int ldc2, int c2_lo,
                  int c2_hi, int c2_last, hipStream_t s) {
    D4_REQUIRE(attn_out_cols_applicable(sa, D) && ldw % 4 == 0 && ((uintptr_t)W % 16) == 0, "attn_out_cols: call not supported");
    FrameOut fo{resid, ldr, out, ldo, c2, ldc2, c2_lo, c2_hi, c2_last, sa.nk};
    const size_t lds = (size_t)(16 * (512 + 4) + AOC_NW * 16 * SM_LDV + 2 * AOC_NW * 16 + AOC_NW * 16 * AOC_RED) * sizeof(float);
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_out_cols_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.done();
    }
    hipLaunchKernelGGL(attn_out_cols_kernel<512>, dim3(D / 16, sa.groups), dim3(AOC_NW * 64), lds, s, sa, W, ldw, D, fo);
    D4_LAUNCH_CHECK();
    return 0;
}

bool frame_pool_tail_applicable(int frames, int S, int D, int pool_heads) {
    return frame_fused_frames_ok(frames) && D == 512 && pool_heads == 4 && S >= 1 && S <= 16;
}

int frame_pool_tail(const float* u, const float* wv_t, const float* wo_t, int frames, int S, int D, int pool_heads, const float* resid, int ldr, float* out,
                    int ldo, float* c2, int ldc2, int c2_lo, int c2_hi, int c2_last, hipStream_t s) {
    D4_REQUIRE(frame_pool_tail_applicable(frames, S, D, pool_heads) && ldr % 4 == 0 && ldo % 4 == 0 && (!c2 || ldc2 % 4 == 0), "frame_pool_tail: call not supported");
    FrameOut fo{resid, ldr, out, ldo, c2, ldc2, c2_lo, c2_hi, c2_last, S};
    constexpr int DD = 512, PH = 4;
    const size_t lds = (size_t)(PH * 16 * (DD + 4) + 16 * (PH * 64 + 4)) * sizeof(float);
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(frame_pool_tail_kernel<DD, PH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.done();
    }
    const double bytes = 4.0 * frames * S * ((double)PH * D + 2.0 * D) + 8.0 * 4.0 * 2.0 * PH * 64 * D;
    const double flops = 2.0 * frames * S * ((double)PH * 64 * D + (double)D * PH * 64);
    D4_GLUE_LAUNCH_F(GL_FRAME_POOL_TAIL, bytes, flops, (frame_pool_tail_kernel<DD, PH>), dim3(frames), dim3(FF_NW * 64), lds, s, u, wv_t, wo_t, fo);
    D4_LAUNCH_CHECK();
    return 0;
}

// mix + tail in one kernel (pm.u is not written); same applicability as the tail alone
int frame_pool(const PoolMixArgs& pm, const float* wv_t, const float* wo_t, int frames, int S, const float* resid, int ldr, float* out, int ldo, float* c2,
               int ldc2, int c2_lo, int c2_hi, int c2_last, hipStream_t s) {
    D4_REQUIRE(frame_pool_tail_applicable(frames, S, pm.D, pm.heads) && pm.M == frames * S && pm.L >= 1 && pm.L <= 64 && ldr % 4 == 0 && ldo % 4 == 0 &&
               (!c2 || ldc2 % 4 == 0) && pm.k && pm.q && !pm.k_b && !pm.q_b, "frame_pool: call not supported (fp32 keys / queries only)");
    FrameOut fo{resid, ldr, out, ldo, c2, ldc2, c2_lo, c2_hi, c2_last, S};
    constexpr int DD = 512, PH = 4;
    // algorithmic bytes: L hiddens + L projected keys per token row, queries + the row itself, the block output (+ the two weight matrices per XCD)
    const double bytes = 4.0 * pm.M * ((double)pm.L * (pm.D + pm.ldk) + pm.ldq + 2.0 * pm.D) + 8.0 * 4.0 * 2.0 * PH * 64 * pm.D;
    const double flops = 2.0 * pm.M * ((double)PH * 64 * pm.D + (double)pm.D * PH * 64);
    if (pm.L <= 32) {
        // 16 waves: a wave per token row in the mix phase (the score scratch of 16 waves fits the 160 KB beside the mixes for up to 32 hiddens)
        constexpr int NW = 16, LM = 32;
        const size_t lds = (size_t)(PH * 16 * (DD + 4) + 16 * (PH * 64 + 4) + NW * LM * PH) * sizeof(float);
        static DeviceOnce attr_set;
        if (attr_set.need()) {
            D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(frame_pool_kernel<DD, PH, NW, LM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set.done();
        }
        D4_GLUE_LAUNCH_F(GL_FRAME_POOL_TAIL, bytes, flops, (frame_pool_kernel<DD, PH, NW, LM>), dim3(frames), dim3(NW * 64), lds, s, pm, wv_t, wo_t, fo);
    } else {
        constexpr int NW = 8, LM = 64;
        const size_t lds = (size_t)(PH * 16 * (DD + 4) + 16 * (PH * 64 + 4) + NW * LM * PH) * sizeof(float);
        static DeviceOnce attr_set;
        if (attr_set.need()) {
            D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(frame_pool_kernel<DD, PH, NW, LM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set.done();
        }
        D4_GLUE_LAUNCH_F(GL_FRAME_POOL_TAIL, bytes, flops, (frame_pool_kernel<DD, PH, NW, LM>), dim3(frames), dim3(NW * 64), lds, s, pm, wv_t, wo_t, fo);
    }
    D4_LAUNCH_CHECK();
    return 0;
}

}  // namespace d4
