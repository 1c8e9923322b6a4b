// Microbenchmark: what sets the per-kernel floor of a dependent kernel chain replayed from a hipGraph on MI355X?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/_bin/kernel_floor_bench tools/micro/kernel_floor_bench.hip
// Variants of a chain of 960 tiny kernels: one kernel repeated / 16 distinct kernels cycled (instruction cache), a 256-byte by-value
// argument struct (kernarg fetch), every kernel also streaming `mb` MB through the L2s (write-back / invalidate at the boundary),
// and grids of 1 / 64 / 256 blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Big { float* buf; const float* stream; int words; int pad[59]; };   // 256 bytes

template <int V>
__global__ __launch_bounds__(256) void k_small(float* buf, const float* stream, int words) {
    float acc = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < words; i += gridDim.x * 256) acc += stream[i];
    // V distinct code bodies of a few hundred instructions each
#pragma unroll
    for (int j = 0; j < 48; ++j) acc = acc * (1.0001f + 0.001f * (V + 1)) + (float)(j * (V + 3));
    if (threadIdx.x == 0) buf[blockIdx.x] += acc * 1e-30f + 1.f;
}
template <int V>
__global__ __launch_bounds__(256) void k_big(Big a) {
    float acc = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.words; i += gridDim.x * 256) acc += a.stream[i];
#pragma unroll
    for (int j = 0; j < 48; ++j) acc = acc * (1.0001f + 0.001f * (V + 1)) + (float)(j * (V + 3)) + (float)a.pad[j];
    if (threadIdx.x == 0) a.buf[blockIdx.x] += acc * 1e-30f + 1.f;
}

template <int V> static void launch_small(int grid, hipStream_t s, float* buf, const float* st, int words) { hipLaunchKernelGGL(k_small<V>, dim3(grid), dim3(256), 0, s, buf, st, words); }
template <int V> static void launch_big(int grid, hipStream_t s, float* buf, const float* st, int words) { Big a{}; a.buf = buf; a.stream = st; a.words = words; hipLaunchKernelGGL(k_big<V>, dim3(grid), dim3(256), 0, s, a); }
typedef void (*Launch)(int, hipStream_t, float*, const float*, int);
static Launch smalls[16] = {launch_small<0>, launch_small<1>, launch_small<2>, launch_small<3>, launch_small<4>, launch_small<5>, launch_small<6>, launch_small<7>,
                            launch_small<8>, launch_small<9>, launch_small<10>, launch_small<11>, launch_small<12>, launch_small<13>, launch_small<14>, launch_small<15>};
static Launch bigs[16] = {launch_big<0>, launch_big<1>, launch_big<2>, launch_big<3>, launch_big<4>, launch_big<5>, launch_big<6>, launch_big<7>,
                          launch_big<8>, launch_big<9>, launch_big<10>, launch_big<11>, launch_big<12>, launch_big<13>, launch_big<14>, launch_big<15>};

static void run(const char* name, Launch* table, int distinct, int grid, int words, float* buf, const float* st, size_t st_words) {
    hipStream_t s; CK(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    const int n = 960;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) table[i % distinct](grid, s, buf, st + ((size_t)i * 1315423911u % (st_words - words - 1) & ~63ull), words);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) { CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); }
    printf("%-60s grid %3d, %7d words streamed: %.2f us / kernel\n", name, grid, words, 1e3f * ms / n);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
}

int main() {
    float *buf, *st; const size_t st_words = (size_t)256 << 20;      // 1 GB stream source
    CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20)); CK(hipMalloc(&st, st_words * 4)); CK(hipMemset(st, 0, st_words * 4));
    for (int grid : {1, 64, 256}) {
        run("one kernel, 3 scalar args", smalls, 1, grid, 0, buf, st, st_words);
        run("16 distinct kernels, 3 scalar args", smalls, 16, grid, 0, buf, st, st_words);
        run("one kernel, 256-byte struct arg", bigs, 1, grid, 0, buf, st, st_words);
        run("16 distinct kernels, 256-byte struct arg", bigs, 16, grid, 0, buf, st, st_words);
    }
    for (int words : {1 << 16, 1 << 20, 4 << 20}) {
        run("one kernel + stream", smalls, 1, 256, words, buf, st, st_words);
        run("16 distinct kernels + stream", smalls, 16, 256, words, buf, st, st_words);
    }
    return 0;
}
