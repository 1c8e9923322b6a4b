// Microbenchmark: what does a software grid barrier cost on MI355X, against a kernel boundary inside a hipGraph?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/_bin/grid_barrier_bench tools/micro/grid_barrier_bench.hip
// Variants: (a) all G workgroups, agent-scope release/acquire (L2 write-back + invalidate every barrier);
//           (b) data exchanged with agent-scope relaxed atomics (bypass the non-coherent L2s), barrier counter relaxed + s_waitcnt only;
//           (c) only the workgroups of one XCD (blockIdx % 8 == 0) take part.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned& target, unsigned G) {
    __syncthreads();
    if (threadIdx.x == 0) {
        target += G;
        if (MODE == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        } else {
            __builtin_amdgcn_s_waitcnt(0);          // this thread's own stores; the block's were ordered by the barrier above + write-through L1
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

// every iteration: each participating block writes `words` floats, barrier, reads a neighbour's words and checks them
template <int MODE, int XCD_ONLY>
__global__ __launch_bounds__(256) void barrier_kernel(float* buf, unsigned* ctr, int iters, int words, int* errors) {
    int wg = blockIdx.x, G = gridDim.x;
    if (XCD_ONLY) { if (blockIdx.x & 7) return; wg = blockIdx.x >> 3; G = gridDim.x >> 3; }
    unsigned target = 0;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        float* mine = buf + ((size_t)(it & 1) * G + wg) * words;
        const float* theirs = buf + ((size_t)(it & 1) * G + (wg + 5) % G) * words;
        const float expect = (float)(it * 1000 + (wg + 5) % G);
        for (int i = threadIdx.x; i < words; i += 256) {
            if (MODE == 0) mine[i] = (float)(it * 1000 + wg);
            else __hip_atomic_store(mine + i, (float)(it * 1000 + wg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        grid_barrier<MODE>(ctr, target, (unsigned)G);
        for (int i = threadIdx.x; i < words; i += 256) {
            const float v = MODE == 0 ? theirs[i] : __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bad += v != expect;
        }
    }
    if (bad) atomicAdd(errors, bad);
}

__global__ void tiny_kernel(float* buf, int it) { if (threadIdx.x == 0) buf[blockIdx.x] += 1.f; }

template <int MODE, int XCD_ONLY>
static void run(const char* name, int G, int iters, int words, float* buf, unsigned* ctr, int* err) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(ctr, 0, 4)); CK(hipMemset(err, 0, 4));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((barrier_kernel<MODE, XCD_ONLY>), dim3(G), dim3(256), 0, 0, buf, ctr, iters, words, err);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    }
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    int h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
    printf("%-44s G=%3d words=%5d : %.3f us / barrier   (visibility errors %d)\n", name, XCD_ONLY ? G / 8 : G, words, 1e3f * ms / iters, h);
}

int main() {
    float* buf; unsigned* ctr; int* err;
    CK(hipMalloc(&buf, 64 << 20)); CK(hipMemset(buf, 0, 64 << 20)); CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&err, 4));
    const int iters = 4000;
    for (int words : {256, 4096}) {
        run<0, 0>("agent release/acquire, all CUs", 256, iters, words, buf, ctr, err);
        run<0, 0>("agent release/acquire, 128 blocks", 128, iters, words, buf, ctr, err);
        run<0, 0>("agent release/acquire, 64 blocks", 64, iters, words, buf, ctr, err);
        run<1, 0>("relaxed atomics data + counter, all CUs", 256, iters, words, buf, ctr, err);
        run<1, 0>("relaxed atomics data + counter, 64 blocks", 64, iters, words, buf, ctr, err);
        run<0, 1>("one XCD, release/acquire", 256, iters, words, buf, ctr, err);
        run<1, 1>("one XCD, relaxed atomics", 256, iters, words, buf, ctr, err);
    }
    // kernel boundaries in a graph
    hipStream_t s; CK(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    const int n = 1000;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(64), dim3(256), 0, s, buf, i);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); }
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("hipGraph of %d dependent tiny kernels: %.3f us / kernel\n", n, 1e3f * ms / n);
    for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(a, s)); for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(64), dim3(256), 0, s, buf, i); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); }
    CK(hipEventElapsedTime(&ms, a, b));
    printf("stream of %d dependent tiny kernels:   %.3f us / kernel\n", n, 1e3f * ms / n);
    return 0;
}
