// MI355X microbenchmark (round 6): what the memory side delivers for the attention-pool mix's ACCESS PATTERN at config 5, with the arithmetic removed.
// One wave per token row m (4 per workgroup, as pool_mix_kernel): for l = 0 .. L-1 it reads the row's hidden image hid[l][m][D] bf16 (2 KB) and,
// optionally, its key row keys[l][m] (512 B at a leading dimension `ldk`: 256 = contiguous rows, 3072 = the wide layout), RD rows in flight, and, optionally,
// writes the 4 x D bf16 mixes of the row.  Before every timed launch a 1 GB buffer is rewritten, so the rows come from HBM as they do in the rollout
// (written many kernels earlier), not from the 256 MB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/_bin/pool_mix_stream_ceiling tools/micro/pool_mix_stream_ceiling.hip && tools/micro/_bin/pool_mix_stream_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int D = 1024;

template <int RD, bool KEYS, bool WRITE>
__global__ __launch_bounds__(256) void stream_kernel(const uint2* __restrict__ hid, const uint2* __restrict__ keys, uint2* __restrict__ u, int M, int L, int ldk, unsigned* sink) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    unsigned acc = 0;
    if (KEYS) {
        for (int l0 = 0; l0 < L; l0 += 8) {
            uint2 kr[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) kr[j] = l0 + j < L ? keys[(((size_t)(l0 + j) * M + m) * ldk) / 4 + lane] : uint2{0u, 0u};
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += kr[j].x ^ kr[j].y;
        }
    }
    uint2 raw[RD][4];
#pragma unroll
    for (int j = 0; j < RD; ++j)
        if (j < L)
#pragma unroll
            for (int i = 0; i < 4; ++i) raw[j][i] = hid[((size_t)j * M + m) * (D / 4) + lane + 64 * i];
    for (int l0 = 0; l0 < L; l0 += RD) {
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            const int l = l0 + j;
            if (l >= L) break;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += raw[j][i].x + raw[j][i].y;
            if (l + RD < L)
#pragma unroll
                for (int i = 0; i < 4; ++i) raw[j][i] = hid[((size_t)(l + RD) * M + m) * (D / 4) + lane + 64 * i];
        }
    }
    if (WRITE) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) u[((size_t)m * 4 + h) * (D / 4) + lane + 64 * i] = uint2{acc + h, acc + i};
    } else if (acc == 0x12345u) sink[0] = acc;
}

// the same traffic with 16-byte accesses: a lane owns 8 consecutive bf16 features (two chunks of a 1024-wide row) instead of four groups of 4
template <int RD, bool WRITE>
__global__ __launch_bounds__(256) void stream16_kernel(const uint4* __restrict__ hid, const uint2* __restrict__ keys, uint4* __restrict__ u, int M, int L, int ldk, unsigned* sink) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    unsigned acc = 0;
    for (int l0 = 0; l0 < L; l0 += 8) {
        uint2 kr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) kr[j] = l0 + j < L ? keys[(((size_t)(l0 + j) * M + m) * ldk) / 4 + lane] : uint2{0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += kr[j].x ^ kr[j].y;
    }
    uint4 raw[RD][2];
#pragma unroll
    for (int j = 0; j < RD; ++j)
        if (j < L)
#pragma unroll
            for (int i = 0; i < 2; ++i) raw[j][i] = hid[((size_t)j * M + m) * (D / 8) + lane + 64 * i];
    for (int l0 = 0; l0 < L; l0 += RD) {
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            const int l = l0 + j;
            if (l >= L) break;
#pragma unroll
            for (int i = 0; i < 2; ++i) acc += raw[j][i].x + raw[j][i].y + raw[j][i].z + raw[j][i].w;
            if (l + RD < L)
#pragma unroll
                for (int i = 0; i < 2; ++i) raw[j][i] = hid[((size_t)(l + RD) * M + m) * (D / 8) + lane + 64 * i];
        }
    }
    if (WRITE) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) u[((size_t)m * 4 + h) * (D / 8) + lane + 64 * i] = uint4{acc + h, acc + i, acc, acc};
    } else if (acc == 0x12345u) sink[0] = acc;
}

__global__ void fill_kernel(uint4* p, size_t n, unsigned v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = uint4{v, v + 1, v + 2, v + 3};
}

template <int RD, bool KEYS, bool WRITE>
static void run(const char* name, const uint2* hid, const uint2* keys, uint2* u, int M, int L, int ldk, unsigned* sink, uint4* flush, size_t flush_n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum = 0.f;
    const int reps = 6;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, flush, flush_n, (unsigned)r);      // evict the Infinity Cache
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<RD, KEYS, WRITE>), dim3((M + 3) / 4), dim3(256), 0, 0, hid, keys, u, M, L, ldk, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r > 0) { sum += ms; best = ms < best ? ms : best; }
    }
    const double bytes = (double)L * M * (2.0 * D + (KEYS ? 512.0 : 0.0)) + (WRITE ? 4.0 * M * D * 2.0 : 0.0);
    printf("%-58s M %6d L %2d: avg %7.1f us  min %7.1f us   %6.1f MB -> %5.2f TB/s (avg)\n", name, M, L, sum / (reps - 1) * 1e3, best * 1e3, bytes / 1e6, bytes / (sum / (reps - 1) * 1e-3) / 1e12);
}

template <int RD, bool WRITE>
static void run16(const char* name, const uint2* hid, const uint2* keys, uint2* u, int M, int L, int ldk, unsigned* sink, uint4* flush, size_t flush_n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum = 0.f;
    const int reps = 6;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, flush, flush_n, (unsigned)r);
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream16_kernel<RD, WRITE>), dim3((M + 3) / 4), dim3(256), 0, 0, reinterpret_cast<const uint4*>(hid), keys, reinterpret_cast<uint4*>(u), M, L, ldk, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r > 0) { sum += ms; best = ms < best ? ms : best; }
    }
    const double bytes = (double)L * M * (2.0 * D + 512.0) + (WRITE ? 4.0 * M * D * 2.0 : 0.0);
    printf("%-58s M %6d L %2d: avg %7.1f us  min %7.1f us   %6.1f MB -> %5.2f TB/s (avg)\n", name, M, L, sum / (reps - 1) * 1e3, best * 1e3, bytes / 1e6, bytes / (sum / (reps - 1) * 1e-3) / 1e12);
}

int main() {
    const int Lmax = 23, NP = 12;
    for (int M : {1792, 14336}) {
        uint2 *hid, *keys, *u; unsigned* sink; uint4* flush;
        const size_t flush_n = (size_t)1 << 26;                   // 1 GiB
        hipMalloc(&hid, (size_t)Lmax * M * D * 2); hipMalloc(&keys, (size_t)Lmax * M * NP * 256 * 2); hipMalloc(&u, (size_t)M * 4 * D * 2); hipMalloc(&sink, 64);
        hipMalloc(&flush, flush_n * 16);
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<uint4*>(hid), (size_t)Lmax * M * D * 2 / 16, 7u);
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<uint4*>(keys), (size_t)Lmax * M * NP * 256 * 2 / 16, 9u);
        for (int L : {13, 23}) {
            run<1, false, false>("hiddens only, 1 row ahead", hid, keys, u, M, L, 256, sink, flush, flush_n);
            run<3, false, false>("hiddens only, 3 rows ahead", hid, keys, u, M, L, 256, sink, flush, flush_n);
            run<3, true, false>("hiddens + keys (contiguous key rows), 3 ahead", hid, keys, u, M, L, 256, sink, flush, flush_n);
            run<3, true, false>("hiddens + keys (wide layout, ld 3072), 3 ahead", hid, keys, u, M, L, NP * 256, sink, flush, flush_n);
            run<3, true, true>("hiddens + keys (wide) + the mixes written, 3 ahead", hid, keys, u, M, L, NP * 256, sink, flush, flush_n);
            run<1, true, true>("hiddens + keys (wide) + the mixes written, 1 ahead", hid, keys, u, M, L, NP * 256, sink, flush, flush_n);
            run16<3, true>("16-byte accesses: hiddens + keys (wide) + mixes, 3 ahead", hid, keys, u, M, L, NP * 256, sink, flush, flush_n);
            run16<1, true>("16-byte accesses: hiddens + keys (wide) + mixes, 1 ahead", hid, keys, u, M, L, NP * 256, sink, flush, flush_n);
            run16<3, false>("16-byte accesses: hiddens + keys (wide), no write, 3 ahead", hid, keys, u, M, L, NP * 256, sink, flush, flush_n);
        }
        hipFree(hid); hipFree(keys); hipFree(u); hipFree(sink); hipFree(flush);
    }
    return 0;
}
