// MI355X microbenchmark / go-no-go (round 6): the floor of a PER-FRAME fused out-projection tail at config 5 on the bf16 pipe.
// One workgroup owns the 14 (padded 16) token rows of a frame x `cols` output columns: its A operand (16 x 512 bf16) sits in LDS, the weight
// image [cols][512] bf16, re-tiled so that every wave-load is one contiguous KB, is streamed global (L2) -> VGPR -> v_mfma_f32_16x16x32_bf16.
// No attention phase, no residual, no epilogue traffic beyond one fp32 store: what is timed is the stream + the MFMAs, i.e. less than the fused
// kernel would cost.  Forms: 128 workgroups x 1024 columns (a frame per workgroup) and 256 x 512 (a frame's columns over two workgroups).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/_bin/frame_tail_bf16_floor tools/micro/frame_tail_bf16_floor.hip && tools/micro/_bin/frame_tail_bf16_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 512, NW = 8;

// Wt: [cols / 16][K / 32][64 lanes][8 bf16]  (lane = (n = lane & 15, kq = lane >> 4): W[16 t + n][32 s + 8 kq .. + 8])
template <int DEPTH>
__global__ __launch_bounds__(NW * 64) void tail_kernel(const uint16_t* __restrict__ a, const bf16x8* __restrict__ wt, float* __restrict__ out, int cols, int halves) {
    __shared__ __attribute__((aligned(16))) uint16_t As[16 * (K + 8)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frame = blockIdx.x / halves, half = blockIdx.x % halves;
    for (int i = tid; i < 16 * K / 8; i += NW * 64) {
        const int r = i / (K / 8), c = (i % (K / 8)) * 8;
        *reinterpret_cast<bf16x8*>(As + r * (K + 8) + c) = *reinterpret_cast<const bf16x8*>(a + ((size_t)frame * 16 + r) * K + c);
    }
    __syncthreads();
    const int li = lane & 15, kq = lane >> 4;
    const int ntile = cols / 16;                         // 16-column tiles of this workgroup
    const bf16x8* w0 = wt + (size_t)half * ntile * (K / 32) * 64;
    for (int t = wave; t < ntile; t += NW) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bf16x8* wp = w0 + (size_t)t * (K / 32) * 64 + lane;
        bf16x8 wf[K / 32];
#pragma unroll
        for (int s = 0; s < K / 32; ++s) wf[s] = wp[s * 64];                              // the tile's whole K: 16 loads of 16 B per lane in flight
#pragma unroll
        for (int s = 0; s < K / 32; ++s) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(As + li * (K + 8) + 32 * s + 8 * kq);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], af, acc, 0, 0, 0);
        }
        *reinterpret_cast<f32x4*>(out + ((size_t)frame * 16 + li) * (cols * halves) + (size_t)half * cols + 16 * t + 4 * kq) = acc;
    }
}

int main() {
    const int frames = 128, N = 1024;
    uint16_t *a, *w; float* out;
    hipMalloc(&a, (size_t)frames * 16 * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&out, (size_t)frames * 16 * N * 4);
    std::vector<uint16_t> h((size_t)N * K);
    unsigned s = 1u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((s >> 9) & 0x3ff)); }
    hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(a, h.data(), (size_t)frames * 16 * K * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int halves : {1, 2, 4}) {
        const int cols = N / halves;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(tail_kernel<16>, dim3(frames * halves), dim3(NW * 64), 0, 0, a, reinterpret_cast<const bf16x8*>(w), out, cols, halves);
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(tail_kernel<16>, dim3(frames * halves), dim3(NW * 64), 0, 0, a, reinterpret_cast<const bf16x8*>(w), out, cols, halves);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / 20, wbytes = (double)frames * halves * cols * K * 2;
        printf("%d frames x %d workgroup(s) per frame x %4d columns (K = %d): %6.2f us per launch; weight stream %5.1f MB through L2 -> VGPR = %5.2f TB/s; matrix work %.2f GFLOP = %5.0f TF/s\n",
               frames, halves, cols, K, us, wbytes / 1e6, wbytes / us / 1e6, 2.0 * frames * 16 * N * K / 1e9, 2.0 * frames * 16 * N * K / us / 1e6);
    }
    return 0;
}
