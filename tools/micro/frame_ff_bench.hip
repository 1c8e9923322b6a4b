// MI355X microbenchmark / go-no-go for the per-frame trunk kernel: the SiLU-GLU feedforward of ONE frame (<= 16 token rows) per
// workgroup, weights streamed global -> VGPR -> MFMA by dreamer4_amd/csrc/frame_gemm.h, hidden activations in LDS.
//   hipcc --offload-arch=gfx950 -O3 -I dreamer4_amd/csrc -o tools/micro/_bin/frame_ff_bench tools/micro/frame_ff_bench.hip
// Prints the time of 256 frames x 14 rows (cfg 2: D = 512, inner 1365 -> 1376, K of the second GEMM padded to 1408) against the
// matrix-pipe time of the padded work (16 rows) and checks three frames against a float64 host reference.
#include "frame_gemm.h"
#include "frame_gemm_x3.h"
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using namespace d4;

constexpr int D = 512, INNER = 1376, INNER_K = 1408;
constexpr int LDX = D + 4, LDU = INNER_K + 4;

template <int NW, int HOT, int EXP = 0>
__global__ __launch_bounds__(NW * 64) void ff_frame_kernel(const float* __restrict__ x, int S, const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ y, float eps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                    // [16][LDX]
    float* us = xs + 16 * LDX;           // [16][LDU]
    float* rs = us + 16 * LDU;           // [16]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.x;
    const int li = lane & 15, kq = lane >> 4;

    // FF1 unit u = (group g = u / 2, j = u % 2): value tile rows 64 g + 16 j, gate tile rows 64 g + 32 + 16 j of the packed W1
    constexpr int NU1 = INNER / 32 * 2, NU2 = D / 32;
    auto unit1 = [&](int u) {
        const int g = u >> 1, j = u & 1;
        return fg_make_unit(w1, D, HOT ? 4 * (g % 4) + j : 4 * g + j, HOT ? 4 * (g % 4) + 2 + j : 4 * g + 2 + j, lane);
    };
    auto unit2 = [&](int u) { return fg_make_unit(w2, INNER_K, HOT ? 2 * (u % 2) : 2 * u, HOT ? 2 * (u % 2) + 1 : 2 * u + 1, lane); };
    FgRing ring;
    fg_prefetch(ring, unit1(wave));            // the weight stream starts before the activations are even staged

    // stage x (rows >= S are zero), zero the K padding of the hidden tile, 1/rms per row
    for (int i = tid; i < 16 * (D / 4); i += NW * 64) {
        const int m = i / (D / 4), c = (i % (D / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < S) v = *reinterpret_cast<const f32x4*>(x + ((size_t)f * S + m) * D + c);
        *reinterpret_cast<f32x4*>(xs + m * LDX + c) = v;
    }
    for (int i = tid; i < 16 * (INNER_K - INNER); i += NW * 64) us[(i / (INNER_K - INNER)) * LDU + INNER + i % (INNER_K - INNER)] = 0.f;
    __syncthreads();
    for (int m = wave; m < 16; m += NW) {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) { const float v = xs[m * LDX + c]; s = __builtin_fmaf(v, v, s); }
        s = wave_sum(s);
        if (lane == 0) rs[m] = rsqrtf(s / (float)D + eps);
    }
    __syncthreads();

    f32x4 acc0, acc1;
    const float* a1 = xs + li * LDX + 4 * kq;
    for (int u = wave; u < NU1; u += NW) {
        const bool more = u + NW < NU1;
        const FgUnit nxt = more ? unit1(u + NW) : unit2(wave);          // last unit of FF1: look ahead into FF2's weights
        const int g = u >> 1, j = u & 1;
        const int col = 32 * g + 16 * j + 4 * kq;                        // hidden column of r = 0
        const int nv = 64 * g + 16 * j + 4 * kq;                         // packed row of the value (gate: + 32)
        // epilogue operands are requested BEFORE the unit's weight stream: at the end they are older than every load of the look-ahead,
        // so waiting for them does not drain the ring
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b1 + nv), bg = *reinterpret_cast<const f32x4*>(b1 + nv + 32);
        fg_unit<D, EXP>(ring, unit1(u), nxt, a1, acc0, acc1);
        const float r = rs[li];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float val = acc0[e] * r + bv[e], gate = acc1[e] * r + bg[e];
            o[e] = val * siluf(gate);
        }
        *reinterpret_cast<f32x4*>(us + li * LDU + col) = o;
    }
    __syncthreads();
    const float* a2 = us + li * LDU + 4 * kq;
    for (int u = wave; u < NU2; u += NW) {
        const bool more = u + NW < NU2;
        const FgUnit nxt = unit2(more ? u + NW : u);                     // nothing follows: the last look-ahead re-reads this unit (dropped)
        const f32x4 bb[2] = {*reinterpret_cast<const f32x4*>(b2 + 32 * u + 4 * kq), *reinterpret_cast<const f32x4*>(b2 + 32 * u + 16 + 4 * kq)};
        fg_unit<INNER_K, EXP>(ring, unit2(u), nxt, a2, acc0, acc1);
        if (li < S) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = 32 * u + 16 * t + 4 * kq;
                const f32x4 a = t ? acc1 : acc0;
                const f32x4 xr = *reinterpret_cast<const f32x4*>(xs + li * LDX + n);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = a[e] + bb[t][e] + xr[e];
                *reinterpret_cast<f32x4*>(y + ((size_t)f * S + li) * D + n) = o;
            }
        }
    }
}


// the same feedforward on the bf16 matrix cores by split operands (frame_gemm_x3.h): weights as tiled bf16 planes, activations fp32 in LDS
template <int NW>
__global__ __launch_bounds__(NW * 64) void ff_frame_x3_kernel(const float* __restrict__ x, int S, const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ y, float eps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                    // [16][LDX]
    float* us = xs + 16 * LDX;           // [16][LDU]
    float* rs = us + 16 * LDU;           // [16]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.x;
    const int li = lane & 15, kq = lane >> 4;
    constexpr int NU1 = INNER / 32 * 2, NU2 = D / 32;
    auto unit1 = [&](int u) { const int g = u >> 1, j = u & 1; return fg3_make_unit(w1, D, 4 * g + j, 4 * g + 2 + j, lane); };
    auto unit2 = [&](int u) { return fg3_make_unit(w2, INNER_K, 2 * u, 2 * u + 1, lane); };
    Fg3Ring ring;
    fg3_prefetch(ring, unit1(wave));
    for (int i = tid; i < 16 * (D / 4); i += NW * 64) {
        const int m = i / (D / 4), c = (i % (D / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < S) v = *reinterpret_cast<const f32x4*>(x + ((size_t)f * S + m) * D + c);
        *reinterpret_cast<f32x4*>(xs + m * LDX + c) = v;
    }
    for (int i = tid; i < 16 * (INNER_K - INNER); i += NW * 64) us[(i / (INNER_K - INNER)) * LDU + INNER + i % (INNER_K - INNER)] = 0.f;
    __syncthreads();
    for (int m = wave; m < 16; m += NW) {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) { const float v = xs[m * LDX + c]; s = __builtin_fmaf(v, v, s); }
        s = wave_sum(s);
        if (lane == 0) rs[m] = rsqrtf(s / (float)D + eps);
    }
    __syncthreads();
    f32x4 acc0, acc1;
    const float* a1 = xs + li * LDX + 8 * kq;
    for (int u = wave; u < NU1; u += NW) {
        const bool more = u + NW < NU1;
        const Fg3Unit nxt = more ? unit1(u + NW) : unit2(wave);
        const int g = u >> 1, j = u & 1;
        const int col = 32 * g + 16 * j + 4 * kq, nv = 64 * g + 16 * j + 4 * kq;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b1 + nv), bg = *reinterpret_cast<const f32x4*>(b1 + nv + 32);
        fg3_unit<D>(ring, unit1(u), nxt, a1, acc0, acc1);
        const float r = rs[li];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float val = acc0[e] * r + bv[e], gate = acc1[e] * r + bg[e];
            o[e] = val * siluf(gate);
        }
        *reinterpret_cast<f32x4*>(us + li * LDU + col) = o;
    }
    __syncthreads();
    const float* a2 = us + li * LDU + 8 * kq;
    for (int u = wave; u < NU2; u += NW) {
        const bool more = u + NW < NU2;
        const Fg3Unit nxt = unit2(more ? u + NW : u);
        const f32x4 bb[2] = {*reinterpret_cast<const f32x4*>(b2 + 32 * u + 4 * kq), *reinterpret_cast<const f32x4*>(b2 + 32 * u + 16 + 4 * kq)};
        fg3_unit<INNER_K>(ring, unit2(u), nxt, a2, acc0, acc1);
        if (li < S) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = 32 * u + 16 * t + 4 * kq;
                const f32x4 a = t ? acc1 : acc0;
                const f32x4 xr = *reinterpret_cast<const f32x4*>(xs + li * LDX + n);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = a[e] + bb[t][e] + xr[e];
                *reinterpret_cast<f32x4*>(y + ((size_t)f * S + li) * D + n) = o;
            }
        }
    }
}

static unsigned short bf16_rne(float v) { unsigned u; memcpy(&u, &v, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float v; memcpy(&v, &u, 4); return v; }

static double silu_d(double v) { return v / (1.0 + exp(-v)); }

int main() {
    const int F = 256, S = 14;
    const float eps = 1.1920928955078125e-07f;
    std::vector<float> hx((size_t)F * S * D), hw1((size_t)2 * INNER * D), hb1(2 * INNER), hw2((size_t)D * INNER_K, 0.f), hb2(D);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.f / 16777216.f) - 0.5f) * 2.f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw1) v = rnd() * 0.05f;
    for (auto& v : hb1) v = rnd() * 0.1f;
    for (int n = 0; n < D; ++n) for (int k = 0; k < INNER; ++k) hw2[(size_t)n * INNER_K + k] = rnd() * 0.03f;
    for (auto& v : hb2) v = rnd() * 0.1f;
    float *x, *w1, *b1, *w2, *b2, *y;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&w1, hw1.size() * 4); hipMalloc(&b1, hb1.size() * 4); hipMalloc(&w2, hw2.size() * 4); hipMalloc(&b2, hb2.size() * 4);
    hipMalloc(&y, hx.size() * 4);
    auto tiled = [](const std::vector<float>& w, int N, int K) {          // [N][K] -> [N / 16][K / 4][16][4]
        std::vector<float> t(w.size());
        for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) t[(((size_t)(n / 16) * (K / 4) + k / 4) * 16 + n % 16) * 4 + k % 4] = w[(size_t)n * K + k];
        return t;
    };
    const std::vector<float> tw1 = tiled(hw1, 2 * INNER, D), tw2 = tiled(hw2, D, INNER_K);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w1, tw1.data(), hw1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b1, hb1.data(), hb1.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w2, tw2.data(), hw2.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b2, hb2.data(), hb2.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = (size_t)(16 * LDX + 16 * LDU + 16) * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto bench = [&](auto kern, int nw, const char* what) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(F), dim3(nw * 64), lds, 0, x, S, w1, b1, w2, b2, y, eps);
        const int reps = 20;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(F), dim3(nw * 64), lds, 0, x, S, w1, b1, w2, b2, y, eps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us_per = ms * 1e3 / reps;
        const double flop_pad = 2.0 * 16 * D * (2.0 * INNER + INNER_K) * F, flop_alg = 2.0 * S * D * (2.0 * 1365 + 1365) * F;
        printf("ff_frame_kernel %-22s: %d frames x %d rows, %2d waves per CU: %.1f us per launch  | executed (16 rows, padded K) %.1f TF/s = %.3f of 157.3 | "
               "algorithmic (%d rows, inner 1365) %.1f TF/s = %.3f\n", what, F, S, nw, us_per, flop_pad / us_per / 1e6, flop_pad / us_per / 1e6 / 157.3, S,
               flop_alg / us_per / 1e6, flop_alg / us_per / 1e6 / 157.3);
    };
    bench(ff_frame_kernel<8, 0>, 8, "(warm-up)");
    bench(ff_frame_kernel<8, 0, 1>, 8, "NO weight loads");
    bench(ff_frame_kernel<4, 0, 1>, 4, "NO weight loads");
    bench(ff_frame_kernel<8, 0, 2>, 8, "NO A-fragment reads");
    bench(ff_frame_kernel<4, 0>, 4, "");
    bench(ff_frame_kernel<8, 0>, 8, "");
    auto check = [&](const char* what) {
    std::vector<float> hy(hx.size());
    hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int f : {0, 100, 255}) {
        for (int m = 0; m < S; ++m) {
            const float* xr = &hx[((size_t)f * S + m) * D];
            double ss = 0; for (int k = 0; k < D; ++k) ss += (double)xr[k] * xr[k];
            const double r = 1.0 / sqrt(ss / D + eps);
            std::vector<double> u(INNER);
            for (int c = 0; c < INNER; ++c) {
                const int g = c / 32, nv = 64 * g + c % 32;
                double v = 0, gt = 0;
                for (int k = 0; k < D; ++k) { v += (double)xr[k] * hw1[(size_t)nv * D + k]; gt += (double)xr[k] * hw1[(size_t)(nv + 32) * D + k]; }
                u[c] = (v * r + hb1[nv]) * silu_d(gt * r + hb1[nv + 32]);
            }
            for (int n = 0; n < D; ++n) {
                double o = hb2[n] + xr[n];
                for (int c = 0; c < INNER; ++c) o += u[c] * hw2[(size_t)n * INNER_K + c];
                const double d = fabs(o - hy[((size_t)f * S + m) * D + n]);
                if (d > maxerr) maxerr = d;
                if (fabs(o) > maxref) maxref = fabs(o);
            }
        }
    }
    printf("%s: max |err| vs float64 reference %.3e at scale %.3f (%s)\n", what, maxerr, maxref, maxerr < 2e-5 * maxref + 1e-5 ? "OK" : "MISMATCH");
    };
    check("fp32 MFMA");
    {   // split-operand form: tiled bf16 planes  Wt[n / 16][k / 32][plane][16 kq + i][8]
        auto planes = [&](const std::vector<float>& w, int N, int K) {
            std::vector<unsigned short> t((size_t)N * K * 3);
            for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) {
                const float a = w[(size_t)n * K + k];
                const unsigned short h1 = bf16_rne(a); const float r1 = a - bf16_f(h1);
                const unsigned short h2 = bf16_rne(r1); const unsigned short h3 = bf16_rne(r1 - bf16_f(h2));
                const size_t base = ((size_t)(n / 16) * (K / 32) + k / 32) * 3;
                const int lane = 16 * ((k % 32) / 8) + n % 16, e = k % 8;
                t[((base + 0) * 64 + lane) * 8 + e] = h1; t[((base + 1) * 64 + lane) * 8 + e] = h2; t[((base + 2) * 64 + lane) * 8 + e] = h3;
            }
            return t;
        };
        const auto p1 = planes(hw1, 2 * INNER, D), p2 = planes(hw2, D, INNER_K);
        float *w1p, *w2p;
        hipMalloc(&w1p, p1.size() * 2); hipMalloc(&w2p, p2.size() * 2);
        hipMemcpy(w1p, p1.data(), p1.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w2p, p2.data(), p2.size() * 2, hipMemcpyHostToDevice);
        float* sw1 = w1; float* sw2 = w2; w1 = w1p; w2 = w2p;
        hipMemset(y, 0, hx.size() * 4);
        bench(ff_frame_x3_kernel<8>, 8, "x3 (bf16 planes)");
        bench(ff_frame_x3_kernel<4>, 4, "x3 (bf16 planes)");
        bench(ff_frame_x3_kernel<8>, 8, "x3 (bf16 planes)");
        check("split operands on the bf16 MFMA");
        w1 = sw1; w2 = sw2;
    }
    return 0;
}
