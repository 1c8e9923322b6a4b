// MI355X microbenchmark: how fast can the CUs pull L2-resident (and L1-resident) data, by path and by access shape?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_feed tools/micro/l2_feed_bench.hip && /tmp/l2_feed
// Question it answers: the GEMM kernels of this repository all stop near 9-10 TB/s of operand delivery; the
// guide's L2 figure is ~34.5 TB/s.  Is the ceiling the path (buffer_load ... lds vs loads to registers), the access shape (a k-tile of a
// row-major [M][512] fp32 matrix = 8 rows x 128 B at a 2 KB stride per 1 KB wave-instruction), or concurrency (waves per CU x loads in
// flight per wave)?
//   MODE 0: buffer_load_dwordx4 ... lds (LDS-DMA, what gemm2.hip does)      MODE 1: buffer_load_dwordx4 to VGPRs (what gemm_x3.hip does)
//   PAT 0: 1 KB contiguous per wave-instruction        PAT 1: 8 rows x 128 B, row stride 2 KB (fp32 k-tile of 32)
//   PAT 2: PAT 1 with the 16-byte chunks XOR-swizzled  PAT 3: 4 rows x 256 B, row stride 2 KB (k-tile of 64)
//   PAT 4: 16 rows x 64 B, row stride 1 KB (a bf16 k-tile of 32 of a [N][512] bf16 plane)
//   WS: bytes each XCD's blocks walk (2 MB: L2-resident, far over the 32 KB L1; 16 KB per block: L1-resident)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int PAT, int DEPTH>
__global__ __launch_bounds__(256) void feed(const float* base, uint32_t ws_bytes, int private_ws, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];     // [4 waves][DEPTH][256 floats]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, inx = blockIdx.x >> 3;
    // L2-resident case: the XCD's blocks share one window of ws_bytes and start at different places; L1-resident: a private window per block
    const float* win = base + (size_t)xcd * (4u << 20) / 4 + (private_ws ? (size_t)(inx % 64) * ws_bytes / 4 : 0);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(win), 0, (int)ws_bytes, 0x00020000);
    uint32_t loff;          // this lane's offset inside one 1 KB wave-instruction footprint
    uint32_t step;          // bytes the footprint advances per instruction inside a row group
    uint32_t group_bytes;   // bytes covered by one row group (rows x row stride)
    if (PAT == 0) { loff = lane * 16; step = 1024; group_bytes = 1024; }
    else if (PAT == 1) { loff = (lane >> 3) * 2048 + (lane & 7) * 16; step = 128; group_bytes = 8 * 2048; }
    else if (PAT == 2) { loff = (lane >> 3) * 2048 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16); step = 128; group_bytes = 8 * 2048; }
    else if (PAT == 3) { loff = (lane >> 4) * 2048 + (lane & 15) * 16; step = 256; group_bytes = 4 * 2048; }
    else { loff = (lane >> 2) * 1024 + (lane & 3) * 16; step = 64; group_bytes = 16 * 1024; }
    const uint32_t steps_per_group = (PAT == 0) ? 1 : (PAT == 4 ? 1024u : 2048u) / step;
    // instruction n of this wave: group g = n / steps_per_group, k-step = n % steps_per_group
    uint32_t n = (uint32_t)(inx * 4 + wave) * 37u;       // different starting places
    const uint32_t ngroups = ws_bytes / group_bytes;
    auto addr = [&](uint32_t i) {
        const uint32_t g = (i / steps_per_group) % ngroups, ks = i % steps_per_group;
        return g * group_bytes + ks * step + loff;
    };
    f32x4 sink = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == 0) {
        float* my = smem + wave * DEPTH * 256;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(my + d * 256), 16, (uint32_t)addr(n + d), 0, 0, 0);
        n += DEPTH;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                wait_vmcnt<DEPTH - 1>();
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(my + d * 256), 16, (uint32_t)addr(n + d), 0, 0, 0);
            }
            n += DEPTH;
        }
        wait_vmcnt<0>();
        sink[0] = my[lane];
    } else {
        f32x4 r[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) r[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, addr(n + d), 0, 0));
        n += DEPTH;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                sink[0] += r[d][0]; sink[1] += r[d][1]; sink[2] += r[d][2]; sink[3] += r[d][3];
                r[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, addr(n + d), 0, 0));
            }
            n += DEPTH;
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) sink[0] += r[d][0];
    }
    if (sink[0] + sink[1] + sink[2] + sink[3] == 12345.678f) out[blockIdx.x] = sink[0];
}

static float* g_buf; static float* g_out;

template <int MODE, int PAT, int DEPTH>
void run(int blocks_per_cu, uint32_t ws_bytes, int private_ws) {
    const int iters = 400;
    const int grid = 256 * blocks_per_cu;
    const size_t lds = MODE == 0 ? 4 * DEPTH * 1024 : 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((feed<MODE, PAT, DEPTH>), dim3(grid), dim3(256), lds, 0, g_buf, ws_bytes, private_ws, 20, g_out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((feed<MODE, PAT, DEPTH>), dim3(grid), dim3(256), lds, 0, g_buf, ws_bytes, private_ws, iters, g_out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 4 * (iters + 1) * DEPTH * 1024.0;
    static const char* pn[] = {"1KB contiguous", "8 rows x 128B @2KB", "8x128B swizzled", "4 rows x 256B @2KB", "16 rows x 64B @1KB"};
    printf("%-10s %-20s depth %2d  %2d waves/CU  ws %7u B %-8s : %7.3f ms  %6.2f TB/s  (%5.1f B/clk/CU at 2.1 GHz)\n", MODE ? "to VGPRs" : "LDS-DMA", pn[PAT], DEPTH,
           blocks_per_cu * 4, ws_bytes, private_ws ? "(L1)" : "(L2)", ms, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 2.1e9);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int MODE, int PAT>
void sweep() {
    for (int bpc : {1, 2, 4}) run<MODE, PAT, 4>(bpc, 2u << 20, 0);
    for (int bpc : {1, 2, 4}) run<MODE, PAT, 8>(bpc, 2u << 20, 0);
    run<MODE, PAT, 16>(2, 2u << 20, 0);
    run<MODE, PAT, 8>(4, 16u << 10, 1);          // L1-resident
}

int main() {
    hipMalloc(&g_buf, 64u << 20); hipMemset(g_buf, 0, 64u << 20); hipMalloc(&g_out, 1 << 20);
    sweep<0, 0>(); sweep<0, 1>(); sweep<0, 2>(); sweep<0, 3>(); sweep<0, 4>();
    sweep<1, 0>(); sweep<1, 1>(); sweep<1, 3>(); sweep<1, 4>();
    return 0;
}
