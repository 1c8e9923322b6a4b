// MI355X microbenchmark: issue rate of v_mfma_f32_32x32x16_bf16 under the accumulator patterns of gemm_x3.hip.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/micro/mfma_bf16_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// PATTERN 0: 12 MFMAs on 12 independent accumulators; 1: the 8-wave tile's order (lo0 lo1 hi0 hi1 lo0 lo1 lo0 lo1 lo0 lo1 lo0 lo1);
// 2: one accumulator (fully dependent chain); 3: two accumulators alternating
template <int PATTERN>
__global__ __launch_bounds__(512) void k(float* out, int iters, const float* rnd) {
    bf16x8 a, b;       // rnd == nullptr: constant small integers (few toggling bits); else random normal operands (what a GEMM feeds the pipe)
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(rnd ? rnd[(threadIdx.x * 16 + e) % 8192] : (float)(threadIdx.x + e));
        b[e] = (__bf16)(rnd ? rnd[(threadIdx.x * 16 + 8 + e) % 8192] : (float)(e + 1));
    }
    f32x16 acc[12];
    for (int i = 0; i < 12; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#define M(i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        if (PATTERN == 0) { M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) }
        if (PATTERN == 1) { M(0) M(1) M(2) M(3) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) }
        if (PATTERN == 2) { M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) }
        if (PATTERN == 3) { M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) }
    }
    float s = 0;
    for (int i = 0; i < 12; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int P>
void run(const char* name, int threads, bool random = false) {
    float* out; hipMalloc(&out, 256 * 512 * 4 * 2);
    float* rnd = nullptr;
    if (random) {
        static float host[8192];
        unsigned s = 12345u;
        for (int i = 0; i < 8192; ++i) { float acc = 0; for (int j = 0; j < 12; ++j) { s = s * 1664525u + 1013904223u; acc += (s >> 8) * (1.f / 16777216.f); } host[i] = acc - 6.f; }
        hipMalloc(&rnd, sizeof(host)); hipMemcpy(rnd, host, sizeof(host), hipMemcpyHostToDevice);
    }
    const int iters = 40000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<P>, dim3(256), dim3(threads), 0, 0, out, 10, rnd);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<P>, dim3(256), dim3(threads), 0, 0, out, iters, rnd);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 12 * (threads / 64) / 4.0;
    const double tf = (double)iters * 12 * (threads / 64) * 256 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-34s %d waves/CU: %.3f ms  %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)  %.0f TF/s\n", name, threads / 64, ms, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, tf);
    hipFree(out);
}

int main() {
    for (int threads : {256, 512}) {
        run<0>("12 independent accumulators", threads);
        run<1>("gemm_x3 8-wave order (4 acc)", threads);
        run<3>("2 accumulators alternating", threads);
        run<2>("1 accumulator (dependent chain)", threads);
        run<0>("12 independent, RANDOM operands", threads, true);
        run<1>("gemm_x3 order, RANDOM operands", threads, true);
    }
    return 0;
}
