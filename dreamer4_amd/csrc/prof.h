// Per-launch HIP-event timing of the non-GEMM ("glue") kernel classes of the rollout, for bench.py's HBM roofline leg:
// the event pair rides on the dispatch itself (hipExtLaunchKernelGGL), so the elapsed time is the kernel's own duration
// (what rocprofv3 --kernel-trace reports) and no barrier packet is added to the stream.  Off by default: one branch per launch.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace d4 {

enum GlueClass : int { GL_SPACE_ATTN = 0, GL_TIME_ATTN, GL_TIME_KV_APPEND, GL_POOL_MIX, GL_SMALL_ATTN, GL_ASSEMBLE, GL_SPLITK_REDUCE, GL_ATTN_WIDE,
                       GL_FRAME_ATTN_OUT, GL_FRAME_POOL_TAIL, GL_N };

// true: this launch is timed — launch with hipExtLaunchKernelGGL(..., *a, *b, 0, ...); `bytes` = algorithmic HBM bytes of the launch
bool glue_prof_begin(int cls, double bytes, hipEvent_t* a, hipEvent_t* b, double flops = 0.);      // flops: matrix work a fused class carries (per-frame kernels)
bool glue_profile_active();
int glue_profile_enable(int mask_and_stride);                     // bits 0..23: classes, bits 24..30: time every n-th launch of a class
int glue_profile_read(double* ms, double* bytes, int64_t* count, int nclass);
int glue_profile_read_flops(double* flops, int nclass);           // call BEFORE glue_profile_read (which clears the log)
const char* glue_class_name(int c);

}  // namespace d4

#define D4_GLUE_LAUNCH(CLS, BYTES, KERNEL, GRID, BLOCK, LDS, STREAM, ...) D4_GLUE_LAUNCH_F(CLS, BYTES, 0., KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__)
#define D4_GLUE_LAUNCH_F(CLS, BYTES, FLOPS, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                 \
    do {                                                                                                           \
        hipEvent_t ea__, eb__;                                                                                     \
        if (d4::glue_prof_begin((CLS), (BYTES), &ea__, &eb__, (FLOPS)))                                                  \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, (uint32_t)(LDS), STREAM, ea__, eb__, 0, __VA_ARGS__);       \
        else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                                    \
    } while (0)
