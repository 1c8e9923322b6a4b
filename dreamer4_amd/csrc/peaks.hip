// Measured denominators for the roofline fractions (BASELINE.md "measured peaks beside the datasheet ones"): what THIS device sustains on
//   * a float4 stream copy between two buffers far larger than the 256 MB Infinity Cache   (HBM, read + write bytes per second),
//   * bare v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x16_bf16 streams on random operands, 12 independent accumulators, 2 waves per SIMD
//     (the matrix pipes at the clock the power budget allows with real data: constant operands run 15 - 20 % faster and are not what a GEMM feeds).
// d4_measure_peaks runs in well under a second, outside every timed region; bench.py prints the three numbers next to the datasheet peaks.
#include "common.h"
#include <hip/hip_runtime.h>

namespace d4 {

__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n4) {
    // a block moves contiguous 16 KB pieces (4 x 256 lanes x 16 B), four 16-byte loads in flight per lane, non-temporal both ways
    const int64_t step = (int64_t)gridDim.x * 1024;
    int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    for (; i + 768 < n4; i += step) {
        const f32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256);
        const f32x4 c = __builtin_nontemporal_load(src + i + 512), d = __builtin_nontemporal_load(src + i + 768);
        __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + 256);
        __builtin_nontemporal_store(c, dst + i + 512); __builtin_nontemporal_store(d, dst + i + 768);
    }
}

typedef __bf16 bf16x8_k __attribute__((ext_vector_type(8)));

template <bool BF16>
__global__ __launch_bounds__(512) void mfma_rate_kernel(float* out, int iters, const float* rnd) {
    float s = 0.f;
    if constexpr (BF16) {
        bf16x8_k a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)rnd[(threadIdx.x * 16 + e) % 8192]; b[e] = (__bf16)rnd[(threadIdx.x * 16 + 8 + e) % 8192]; }
        f32x16 acc[12];
        for (int i = 0; i < 12; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 12; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        const float a = rnd[(threadIdx.x * 2) % 8192], b = rnd[(threadIdx.x * 2 + 1) % 8192];
        f32x4 acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 12; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace d4

// scratch: caller-owned device buffer of `scratch_bytes` (>= 1 GiB recommended: two halves, source and destination of the copy); results in GB/s, TFLOP/s
extern "C" int d4_measure_peaks(void* scratch, size_t scratch_bytes, double* hbm_copy_gbs, double* mfma_f32_tflops, double* mfma_bf16_tflops, void* stream) {
    using namespace d4;
    D4_REQUIRE(scratch && scratch_bytes >= ((size_t)64 << 20) && hbm_copy_gbs && mfma_f32_tflops && mfma_bf16_tflops, "d4_measure_peaks: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int dev = 0, n_cu = 0;
    D4_HIP(hipGetDevice(&dev));
    D4_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    hipEvent_t e0, e1;
    D4_HIP(hipEventCreate(&e0)); D4_HIP(hipEventCreate(&e1));
    float ms = 0.f;
    // ---- stream copy
    const size_t half = (scratch_bytes / 2) & ~(size_t)255;
    const int64_t n4 = (int64_t)(half / 16) & ~(int64_t)1023;          // whole 16 KB pieces
    const f32x4* src = static_cast<const f32x4*>(scratch);
    f32x4* dst = reinterpret_cast<f32x4*>(static_cast<char*>(scratch) + half);
    D4_HIP(hipMemsetAsync(scratch, 0x3c, half, s));
    const int reps = 4;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(n_cu * 16), dim3(256), 0, s, src, dst, n4);
    D4_HIP(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream_copy_kernel, dim3(n_cu * 16), dim3(256), 0, s, src, dst, n4);
    D4_HIP(hipEventRecord(e1, s));
    D4_HIP(hipEventSynchronize(e1));
    D4_LAUNCH_CHECK();
    D4_HIP(hipEventElapsedTime(&ms, e0, e1));
    *hbm_copy_gbs = 2.0 * (double)n4 * 16.0 * reps / (ms * 1e-3) / 1e9;
    // ---- matrix pipes: random operands from a small table at the start of the scratch buffer, results behind it
    static float host[8192];
    unsigned seed = 12345u;
    for (int i = 0; i < 8192; ++i) {
        float acc = 0.f;
        for (int j = 0; j < 12; ++j) { seed = seed * 1664525u + 1013904223u; acc += (seed >> 8) * (1.f / 16777216.f); }
        host[i] = acc - 6.f;                                             // ~ N(0, 1)
    }
    float* rnd = static_cast<float*>(scratch);
    float* out = rnd + 8192;
    D4_HIP(hipMemcpyAsync(rnd, host, sizeof(host), hipMemcpyHostToDevice, s));
    const int iters_bf16 = 20000, iters_f32 = 4000;
    for (int which = 0; which < 2; ++which) {
        const int iters = which ? iters_bf16 : iters_f32;
        if (which) hipLaunchKernelGGL(mfma_rate_kernel<true>, dim3(n_cu), dim3(512), 0, s, out, 10, rnd);
        else hipLaunchKernelGGL(mfma_rate_kernel<false>, dim3(n_cu), dim3(512), 0, s, out, 10, rnd);
        D4_HIP(hipEventRecord(e0, s));
        if (which) hipLaunchKernelGGL(mfma_rate_kernel<true>, dim3(n_cu), dim3(512), 0, s, out, iters, rnd);
        else hipLaunchKernelGGL(mfma_rate_kernel<false>, dim3(n_cu), dim3(512), 0, s, out, iters, rnd);
        D4_HIP(hipEventRecord(e1, s));
        D4_HIP(hipEventSynchronize(e1));
        D4_LAUNCH_CHECK();
        D4_HIP(hipEventElapsedTime(&ms, e0, e1));
        const double flop_per_mfma = which ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 4;
        const double tf = (double)iters * 12 * 8 * n_cu * flop_per_mfma / (ms * 1e-3) / 1e12;
        *(which ? mfma_bf16_tflops : mfma_f32_tflops) = tf;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}
