// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the dreamer4 imagination path.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define D4_WAVE 64

namespace d4 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- last-error plumbing (host) -------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define D4_HIP(expr)                                             \
    do {                                                         \
        hipError_t e__ = (expr);                                 \
        if (e__ != hipSuccess) return d4::hip_fail(e__, #expr);  \
    } while (0)

#define D4_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            d4::set_error(__VA_ARGS__);       \
            return 2;                         \
        }                                     \
    } while (0)

#define D4_LAUNCH_CHECK() D4_HIP(hipGetLastError())

// ---- wave64 reductions -----------------------------------------------------------------------
// Row (16-lane) butterfly with DPP, then the four row totals are combined through readlane.
// Every lane ends up holding the full 64-lane sum (or max).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);   // row_half_mirror
    v += dpp_f<0x140>(v);   // row_mirror
    return v;
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
    // the builtin is int -> int: go through the bit pattern, never a value conversion
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

__device__ __forceinline__ float wave_sum(float v) {
    v = row_sum16(v);
    float a = readlane_f(v, 0);
    float b = readlane_f(v, 16);
    float c = readlane_f(v, 32);
    float d = readlane_f(v, 48);
    return (a + b) + (c + d);
}

__device__ __forceinline__ float row_max16(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
    v = row_max16(v);
    float a = readlane_f(v, 0);
    float b = readlane_f(v, 16);
    float c = readlane_f(v, 32);
    float d = readlane_f(v, 48);
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float siluf(float x) { return x / (1.f + expf(-x)); }
// bf16 engine only (its GEMM epilogues round to bf16 right after): v_exp_f32 + v_rcp_f32, ~1e-7 relative, 5 instructions instead of ~25
__device__ __forceinline__ float siluf_fast(float x) { return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }

// fp32 -> bf16 (round to nearest even) bit patterns, for the bf16 activation images the bf16 engine's GEMMs read
__device__ __forceinline__ uint16_t bf16_bits(float v) { const __bf16 h = (__bf16)v; return __builtin_bit_cast(uint16_t, h); }
__device__ __forceinline__ void store_bf16x4(uint16_t* dst, const f32x4& v) {      // dst 8-byte aligned
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t o;
    o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
    *reinterpret_cast<bf16x4_t*>(dst) = o;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Once-per-DEVICE guard for function attributes (hipFuncSetAttribute applies to the current device only): bit d = done on device d.
// Two host threads racing set the same value twice, which is harmless.
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
    bool need() const { return !((mask.load(std::memory_order_acquire) >> dev()) & 1); }
    void done() { mask.fetch_or((uint64_t)1 << dev(), std::memory_order_release); }
};

}  // namespace d4
