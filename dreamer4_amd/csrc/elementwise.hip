// Glue kernels of the imagination path: weight preparation (RMSNorm gamma folding, SiLU-GLU pair
// packing), token assembly (D4:7182-7222), latent-pred post-processing, shortcut-flow Euler step
// (D4:6567-6580), and the small per-frame heads (D4:6595-6662).  All HBM-bound, fully coalesced
// (float4 where the layout allows), grid-stride.
#include "common.h"
#include "kernels.h"
#include "prof.h"
#include "beta.h"
#include <float.h>

namespace d4 {

static inline dim3 grid1d(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}

// ---------------------------------------------------------------------------------- weight prep
__global__ void fold_rows_kernel(const float* W, const float* gamma, float* out, int rows, int K, int ld_out) {
    const int64_t n = (int64_t)rows * K;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int r = (int)(i / K), k = (int)(i % K);
        out[(int64_t)r * ld_out + k] = W[i] * (gamma ? gamma[k] : 1.f);
    }
}
int fold_rows(const float* W, const float* gamma, float* out, int rows, int K, int ld_out, hipStream_t s) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(fold_rows_kernel, grid1d((int64_t)rows * K), dim3(256), 0, s, W, gamma, out, rows, K, ld_out);
    D4_LAUNCH_CHECK();
    return 0;
}

__global__ void copy_rows_kernel(const float* src, int lds, float* dst, int ldd, int rows, int cols) {
    const int64_t n = (int64_t)rows * cols;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int r = (int)(i / cols), c = (int)(i % cols);
        dst[(int64_t)r * ldd + c] = src[(int64_t)r * lds + c];
    }
}
int copy_rows(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t s) {
    if (rows == 0 || cols == 0) return 0;
    hipLaunchKernelGGL(copy_rows_kernel, grid1d((int64_t)rows * cols), dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
    D4_LAUNCH_CHECK();
    return 0;
}

__global__ void fill_kernel(float* dst, float v, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = v;
}
int fill_f32(float* dst, float v, int64_t n, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_kernel, grid1d(n), dim3(256), 0, s, dst, v, n);
    D4_LAUNCH_CHECK();
    return 0;
}

// proj_in [2*inner][K] (value rows first, gate rows second, D4:2111) -> packed [2*inner_pad][K]:
// per 64 packed rows: 32 value rows then the matching 32 gate rows; gamma folded; zero rows for padding.
__global__ void swiglu_pack_kernel(const float* W, const float* bias, const float* gamma, float* Wp, float* bp,
                                   int inner, int inner_pad, int K) {
    const int64_t n = (int64_t)2 * inner_pad * K;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int pr = (int)(i / K), k = (int)(i % K);
        int blk = pr / 64, e = pr % 64;
        int hidden = blk * 32 + (e % 32);
        bool is_gate = e >= 32;
        float w = 0.f, b = 0.f;
        if (hidden < inner) {
            int src = hidden + (is_gate ? inner : 0);
            w = W[(int64_t)src * K + k] * gamma[k];
            b = bias[src];
        }
        Wp[i] = w;
        if (k == 0) bp[pr] = b;
    }
}
int swiglu_pack_rows(const float* W, const float* bias, const float* gamma, float* Wp, float* bp,
                     int inner, int inner_pad, int K, hipStream_t s) {
    hipLaunchKernelGGL(swiglu_pack_kernel, grid1d((int64_t)2 * inner_pad * K), dim3(256), 0, s, W, bias, gamma, Wp, bp, inner, inner_pad, K);
    D4_LAUNCH_CHECK();
    return 0;
}

__global__ void pad_cols_kernel(const float* W, float* out, int rows, int cols, int cols_pad) {
    const int64_t n = (int64_t)rows * cols_pad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int r = (int)(i / cols_pad), c = (int)(i % cols_pad);
        out[i] = c < cols ? W[(int64_t)r * cols + c] : 0.f;
    }
}
int pad_cols(const float* W, float* out, int rows, int cols, int cols_pad, hipStream_t s) {
    hipLaunchKernelGGL(pad_cols_kernel, grid1d((int64_t)rows * cols_pad), dim3(256), 0, s, W, out, rows, cols, cols_pad);
    D4_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------- RMSNorm (explicit; head MLPs)
// one wave per row; the row is held in registers (16-byte loads, all in flight at once) when it is aligned and D <= 4096
template <bool VEC>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const float* x, int ldx, const float* gamma, float* y, int ldy,
                                                           int rows, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (int64_t)row * ldx;
    float* yr = y + (int64_t)row * ldy;
    if (VEC) {
        constexpr int MAXI = 16;
        const int nf4 = D >> 2;
        f32x4 v[MAXI];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int c4 = lane + 64 * i;
            v[i] = c4 < nf4 ? reinterpret_cast<const f32x4*>(xr)[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
            ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
        }
        const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nf4) {
                const f32x4 g = gamma ? reinterpret_cast<const f32x4*>(gamma)[c4] : f32x4{1.f, 1.f, 1.f, 1.f};
                reinterpret_cast<f32x4*>(yr)[c4] = f32x4{v[i][0] * rstd * g[0], v[i][1] * rstd * g[1], v[i][2] * rstd * g[2], v[i][3] * rstd * g[3]};
            }
        }
        return;
    }
    float ss = 0.f;
    for (int c = lane; c < D; c += 64) { float v = xr[c]; ss += v * v; }
    const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
    for (int c = lane; c < D; c += 64) yr[c] = xr[c] * rstd * (gamma ? gamma[c] : 1.f);
}
int rmsnorm_rows(const float* x, int ldx, const float* gamma, float* y, int ldy, int rows, int D, float eps, hipStream_t s) {
    if (rows == 0) return 0;
    const bool vec = (D % 4) == 0 && D <= 4096 && (ldx % 4) == 0 && (ldy % 4) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                     (gamma == nullptr || ((uintptr_t)gamma % 16) == 0);
    if (vec) hipLaunchKernelGGL(rmsnorm_rows_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, gamma, y, ldy, rows, D, eps);
    else hipLaunchKernelGGL(rmsnorm_rows_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, gamma, y, ldy, rows, D, eps);
    D4_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------- LayerNorm (+ SiLU), head MLPs of the
// Linear -> LayerNorm -> activation recipe.  One wave per row held in registers (D <= 4096); biased variance, two-pass.
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* x, int ldx, const float* g, const float* b, float* y, int ldy,
                                                             int rows, int D, float eps, int silu) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (int64_t)row * ldx;
    float* yr = y + (int64_t)row * ldy;
    constexpr int MAXI = 64;                    // 64 lanes x 64 = 4096 columns
    float v[MAXI];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < D ? xr[c] : 0.f;
        sum += v[i];
    }
    const float mean = wave_sum(sum) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = lane + 64 * i;
        const float d = c < D ? v[i] - mean : 0.f;
        ss += d * d;
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = lane + 64 * i;
        if (c < D) {
            float o = (v[i] - mean) * rstd * g[c] + (b ? b[c] : 0.f);
            if (silu) o = siluf(o);
            yr[c] = o;
        }
    }
}
int layernorm_rows(const float* x, int ldx, const float* g, const float* b, float* y, int ldy, int rows, int D, float eps, int silu, hipStream_t s) {
    if (rows == 0) return 0;
    D4_REQUIRE(D <= 4096, "layernorm: row width %d > 4096", D);
    hipLaunchKernelGGL(layernorm_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, g, b, y, ldy, rows, D, eps, silu);
    D4_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------- token assembly
// tokens[(b,t)][s] = [flow | space x ns | registers x nr | action (if the model has actions) | agent]          D4:7182-7222
__global__ void assemble_kernel(AssembleArgs p) {
    const int D = p.D;
    const int64_t n = (int64_t)p.B * p.Tq * p.S * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int s = (int)((i / D) % p.S);
        const int64_t f = i / ((int64_t)D * p.S);         // frame index (b * Tq + t)
        const int b = (int)(f / p.Tq);
        float v;
        if (s == 0) {
            const int half = D / 2;
            v = d < half ? p.signal_embed[(int64_t)(p.signal_levels ? p.signal_levels[f] : p.signal_uniform) * half + d]
                         : p.step_embed[(int64_t)p.step_log2 * half + (d - half)];
        } else if (s <= p.ns) {
            v = p.space[(f * p.ns + (s - 1)) * D + d];
        } else if (s <= p.ns + p.nr) {
            v = p.registers[(int64_t)(s - 1 - p.ns) * D + d];
        } else if ((p.na > 0 || p.nc > 0) && s == p.ns + p.nr + 1) {        // (a model without an action space has no action token, D4:7124-7130)
            v = 0.f;
            const bool have = p.na > 0 ? (p.prev_actions && p.prev_actions[f * p.na] >= 0) : (p.prev_cont && p.prev_cont[f * p.nc] == p.prev_cont[f * p.nc]);
            if (have) {
                for (int a = 0; a < p.na; ++a)
                    v += p.action_embed[(p.prev_actions[f * p.na + a] + p.action_offsets[a]) * D + d];
                if (p.prev_cont)                                  // embed[type] * value, summed over the types  D4:1535-1545
                    for (int c = 0; c < p.nc; ++c) v += p.cont_embed[(int64_t)c * D + d] * p.prev_cont[f * p.nc + c];
                v += p.action_learned[d];
            }
        } else {
            v = p.agent_embed[d];
            if (p.tasks) v += p.task_embed[p.tasks[b] * D + d];
        }
        p.tokens[i] = v;
        if (p.compact) {
            const int rank = (s >= 1 && s <= p.ns) ? s - 1 : ((p.has_agent && s == p.S - 1) ? p.ns : -1);
            if (rank >= 0) p.compact[(f * (p.ns + p.has_agent) + rank) * D + d] = v;
        }
    }
}
// the same, four features per thread (16-byte loads / stores, one set of index divisions per four elements): D % 8 == 0, 16-byte aligned tables
__global__ __launch_bounds__(256) void assemble4_kernel(AssembleArgs p) {
    const int D4 = p.D / 4;
    const int64_t n4 = (int64_t)p.B * p.Tq * p.S * D4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D4) * 4;
        const int s = (int)((i / D4) % p.S);
        const int64_t f = i / ((int64_t)D4 * p.S);        // frame index (b * Tq + t)
        const int b = (int)(f / p.Tq);
        auto ld4 = [](const float* q) { return *reinterpret_cast<const f32x4*>(q); };
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (s == 0) {
            const int half = p.D / 2;
            v = d < half ? ld4(p.signal_embed + (int64_t)(p.signal_levels ? p.signal_levels[f] : p.signal_uniform) * half + d)
                         : ld4(p.step_embed + (int64_t)p.step_log2 * half + (d - half));
        } else if (s <= p.ns) {
            v = ld4(p.space + (f * p.ns + (s - 1)) * p.D + d);
        } else if (s <= p.ns + p.nr) {
            v = ld4(p.registers + (int64_t)(s - 1 - p.ns) * p.D + d);
        } else if ((p.na > 0 || p.nc > 0) && s == p.ns + p.nr + 1) {
            const bool have = p.na > 0 ? (p.prev_actions && p.prev_actions[f * p.na] >= 0) : (p.prev_cont && p.prev_cont[f * p.nc] == p.prev_cont[f * p.nc]);
            if (have) {
                for (int a = 0; a < p.na; ++a) {
                    const f32x4 e = ld4(p.action_embed + (p.prev_actions[f * p.na + a] + p.action_offsets[a]) * p.D + d);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] += e[c];
                }
                if (p.prev_cont)
                    for (int c = 0; c < p.nc; ++c) {
                        const f32x4 e = ld4(p.cont_embed + (int64_t)c * p.D + d);
                        const float x = p.prev_cont[f * p.nc + c];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += e[q] * x;
                    }
                const f32x4 e = ld4(p.action_learned + d);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] += e[c];
            }
        } else {
            v = ld4(p.agent_embed + d);
            if (p.tasks) {
                const f32x4 e = ld4(p.task_embed + p.tasks[b] * p.D + d);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] += e[c];
            }
        }
        *reinterpret_cast<f32x4*>(p.tokens + i * 4) = v;
        if (p.compact) {
            const int rank = (s >= 1 && s <= p.ns) ? s - 1 : ((p.has_agent && s == p.S - 1) ? p.ns : -1);
            if (rank >= 0) *reinterpret_cast<f32x4*>(p.compact + (f * (p.ns + p.has_agent) + rank) * p.D + d) = v;
        }
    }
}
int assemble_tokens(const AssembleArgs& p, hipStream_t s) {
    const int64_t n = (int64_t)p.B * p.Tq * p.S * p.D;
    if (n == 0) return 0;
    auto al = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
    const bool vec = (p.D % 8) == 0 && al(p.tokens) && al(p.compact) && al(p.signal_embed) && al(p.step_embed) && al(p.space) && al(p.registers) &&
                     al(p.action_embed) && al(p.cont_embed) && al(p.action_learned) && al(p.agent_embed) && al(p.task_embed);
    if (vec) D4_GLUE_LAUNCH(GL_ASSEMBLE, 4.0 * (double)n + 4.0 * p.B * p.Tq * p.ns * p.D, assemble4_kernel, grid1d(n / 4), dim3(256), 0, s, p);
    else D4_GLUE_LAUNCH(GL_ASSEMBLE, 4.0 * (double)n + 4.0 * p.B * p.Tq * p.ns * p.D, assemble_kernel, grid1d(n), dim3(256), 0, s, p);
    D4_LAUNCH_CHECK();
    return 0;
}

// to_latent_pred.0 RMSNorm then the LQAP context RMSNorm (D4:4830-4834, 1993): gathered rows only.
__global__ __launch_bounds__(256) void gather_space_kernel(const float* tokens, float* out, const float* g0, const float* g1,
                                                           int frames, int S, int first, int D, int ns, float eps) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= frames * ns) return;
    const int lane = threadIdx.x & 63;
    const int f = r / ns, j = r % ns;
    const float* xr = tokens + ((int64_t)f * S + first + j) * D;
    float ss = 0.f;
    for (int c = lane; c < D; c += 64) { float v = xr[c]; ss += v * v; }
    const float r0 = rsqrtf(wave_sum(ss) / (float)D + eps);
    float ss1 = 0.f;
    for (int c = lane; c < D; c += 64) { float v = xr[c] * r0 * g0[c]; ss1 += v * v; }
    const float r1 = rsqrtf(wave_sum(ss1) / (float)D + eps);
    float* yr = out + (int64_t)r * D;
    for (int c = lane; c < D; c += 64) yr[c] = (xr[c] * r0 * g0[c]) * r1 * g1[c];
}
// the same with the row held in registers (ITER float4 per lane: D = 256 ITER), one read of the row instead of three
template <int ITER>
__global__ __launch_bounds__(256) void gather_space4_kernel(const float* tokens, float* out, const float* g0, const float* g1,
                                                            int frames, int S, int first, int ns, float eps) {
    constexpr int D = 256 * ITER;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= frames * ns) return;
    const int lane = threadIdx.x & 63;
    const int f = r / ns, j = r % ns;
    const float* xr = tokens + ((int64_t)f * S + first + j) * D;
    f32x4 x[ITER], a[ITER], b[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        x[k] = *reinterpret_cast<const f32x4*>(xr + k * 256 + lane * 4);
        a[k] = *reinterpret_cast<const f32x4*>(g0 + k * 256 + lane * 4);
        b[k] = *reinterpret_cast<const f32x4*>(g1 + k * 256 + lane * 4);
    }
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < ITER; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += x[k][e] * x[k][e];
    const float r0 = rsqrtf(wave_sum(ss) / (float)D + eps);
    float ss1 = 0.f;
#pragma unroll
    for (int k = 0; k < ITER; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[k][e] = x[k][e] * r0 * a[k][e]; ss1 += x[k][e] * x[k][e]; }
    const float r1 = rsqrtf(wave_sum(ss1) / (float)D + eps);
    float* yr = out + (int64_t)r * D;
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = x[k][e] * r1 * b[k][e];
        *reinterpret_cast<f32x4*>(yr + k * 256 + lane * 4) = y;
    }
}
int gather_space_double_norm(const float* tokens, float* out, const float* g0, const float* g1,
                             int frames, int S, int first, int D, int ns, float eps, hipStream_t s) {
    if (frames == 0) return 0;
    const bool al = ((uintptr_t)tokens % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)g0 % 16) == 0 && ((uintptr_t)g1 % 16) == 0;
    if (al && D == 512) hipLaunchKernelGGL(gather_space4_kernel<2>, dim3(cdiv(frames * ns, 4)), dim3(256), 0, s, tokens, out, g0, g1, frames, S, first, ns, eps);
    else if (al && D == 1024) hipLaunchKernelGGL(gather_space4_kernel<4>, dim3(cdiv(frames * ns, 4)), dim3(256), 0, s, tokens, out, g0, g1, frames, S, first, ns, eps);
    else if (al && D == 256) hipLaunchKernelGGL(gather_space4_kernel<1>, dim3(cdiv(frames * ns, 4)), dim3(256), 0, s, tokens, out, g0, g1, frames, S, first, ns, eps);
    else
    hipLaunchKernelGGL(gather_space_kernel, dim3(cdiv(frames * ns, 4)), dim3(256), 0, s, tokens, out, g0, g1, frames, S, first, D, ns, eps);
    D4_LAUNCH_CHECK();
    return 0;
}

// x += (pred - x) / (1 - t) * (step_size / max_steps)                                  D4:6567-6580
__global__ void euler_kernel(float* x, int ldx, const float* pred, int ldp, int B, int n_el, float one_minus_t, float dt) {
    const int64_t n = (int64_t)B * n_el;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / n_el), e = (int)(i % n_el);
        float* xp = x + (int64_t)b * ldx + e;
        const float xi = *xp;
        *xp = xi + (pred[(int64_t)b * ldp + e] - xi) / one_minus_t * dt;
    }
}
int euler_step(float* x, int ldx, const float* pred, int ldp, int B, int n_el, float one_minus_t, float dt, hipStream_t s) {
    if (B == 0) return 0;
    hipLaunchKernelGGL(euler_kernel, grid1d((int64_t)B * n_el), dim3(256), 0, s, x, ldx, pred, ldp, B, n_el, one_minus_t, dt);
    D4_LAUNCH_CHECK();
    return 0;
}

__global__ void silu_kernel(const float* z, float* y, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = siluf(z[i]);
}
int silu_rows(const float* z, float* y, int64_t n, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(silu_kernel, grid1d(n), dim3(256), 0, s, z, y, n);
    D4_LAUNCH_CHECK();
    return 0;
}

__global__ void splitk_reduce_kernel(const float* part, int S, int M, int N, const float* bias, int silu, float* y, int ldy) {
    const int64_t n = (int64_t)M * N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / N), c = (int)(i % N);
        float v = 0.f;
        for (int k = 0; k < S; ++k) v += part[(int64_t)k * n + i];
        if (bias) v += bias[c];
        if (silu) v = siluf(v);
        y[(int64_t)r * ldy + c] = v;
    }
}
int splitk_reduce(const float* part, int S, int M, int N, const float* bias, int silu, float* y, int ldy, hipStream_t s) {
    if (M == 0) return 0;
    D4_GLUE_LAUNCH(GL_SPLITK_REDUCE, 4.0 * (double)M * N * (S + 1), splitk_reduce_kernel, grid1d((int64_t)M * N), dim3(256), 0, s, part, S, M, N, bias, silu, y, ldy);
    D4_LAUNCH_CHECK();
    return 0;
}

// per-evaluation integer inputs: signal level of every frame and the action token source   D4:6492-6523
__global__ void prep_inputs_kernel(int32_t* sig, int64_t* pact, const int64_t* hist, int B, int Tq, int na, int frame_base,
                                   int hist_stride, int sig_val, int ctx_sig, float* pcont, const float* chist, int nc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Tq) return;
    const int b = i / Tq, t = i % Tq;
    sig[i] = (t == Tq - 1) ? sig_val : ctx_sig;
    const int g = frame_base + t;
    for (int a = 0; a < na; ++a)
        pact[(int64_t)i * na + a] = (g == 0 || !hist) ? -1 : hist[((int64_t)b * hist_stride + (g - 1)) * na + a];
    for (int c = 0; c < nc; ++c)
        pcont[(int64_t)i * nc + c] = (g == 0 || !chist) ? __builtin_nanf("") : chist[((int64_t)b * hist_stride + (g - 1)) * nc + c];
}
int prep_eval_inputs(int32_t* sig, int64_t* pact, const int64_t* actions_hist, int B, int Tq, int na, int frame_base,
                     int hist_stride, int sig_val, int ctx_sig, hipStream_t s, float* pcont, const float* cont_hist, int nc) {
    hipLaunchKernelGGL(prep_inputs_kernel, dim3(cdiv(B * Tq, 128)), dim3(128), 0, s, sig, pact, actions_hist, B, Tq, na, frame_base,
                       hist_stride, sig_val, ctx_sig, pcont, cont_hist, pcont ? nc : 0);
    D4_LAUNCH_CHECK();
    return 0;
}

__global__ void fill_sig_kernel(int32_t* sig, int n, int value) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sig[i] = value;
}
int fill_sig(int32_t* sig, int n, int value, hipStream_t s) {
    hipLaunchKernelGGL(fill_sig_kernel, dim3(cdiv(n, 128)), dim3(128), 0, s, sig, n, value);
    D4_LAUNCH_CHECK();
    return 0;
}
__global__ void set_frame_state_kernel(int* state, int t0) { state[0] = t0; }
int set_frame_state(int* state, int t0, hipStream_t s) {
    hipLaunchKernelGGL(set_frame_state_kernel, dim3(1), dim3(1), 0, s, state, t0);
    D4_LAUNCH_CHECK();
    return 0;
}

// engine cache [Lt][2][cache_batch*S][H][Tcap][dh]  <->  reference layout (Lt, 2, B*S, H, frames, dh)   D4:2075, 3256
__global__ void cache_transfer_kernel(float* cache, float* ext, int Lt, int cache_batch, int B, int S, int H, int Tcap,
                                      int frames, int to_ext, int dh) {
    const int64_t n = (int64_t)Lt * 2 * B * S * H * frames * dh;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int d = (int)(r % dh); r /= dh;
        const int t = (int)(r % frames); r /= frames;
        const int h = (int)(r % H); r /= H;
        const int col = (int)(r % ((int64_t)B * S)); r /= (int64_t)B * S;
        const int kv = (int)(r % 2); r /= 2;
        const int l = (int)r;
        const int64_t ci = ((((int64_t)(l * 2 + kv) * cache_batch * S + col) * H + h) * Tcap + t) * dh + d;
        if (to_ext) ext[i] = cache[ci]; else cache[ci] = ext[i];
    }
}
int cache_transfer(float* cache, float* ext, int Lt, int cache_batch, int B, int S, int H, int Tcap, int frames, int to_ext, int dh, hipStream_t s) {
    const int64_t n = (int64_t)Lt * 2 * B * S * H * frames * dh;
    if (n == 0) return 0;
    hipLaunchKernelGGL(cache_transfer_kernel, grid1d(n), dim3(256), 0, s, cache, ext, Lt, cache_batch, B, S, H, Tcap, frames, to_ext, dh);
    D4_LAUNCH_CHECK();
    return 0;
}

// out[b][t] = t < Tq-1 ? lerp(hist[b][t], ctx_noise[b][t], w) : x[b]                    D4:6497-6499
__global__ void build_latent_kernel(float* out, const float* hist, const float* ctx, const float* x,
                                    int B, int Tq, int n_el, int hist_t_stride, float w) {
    const int64_t n = (int64_t)B * Tq * n_el;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % n_el);
        const int t = (int)((i / n_el) % Tq);
        const int b = (int)(i / ((int64_t)n_el * Tq));
        float v;
        if (t == Tq - 1) v = x[(int64_t)b * n_el + e];
        else {
            const int64_t o = ((int64_t)b * hist_t_stride + t) * n_el + e;
            const float a = hist[o], c = ctx ? ctx[o] : a;
            const float d = c - a;
            v = (fabsf(w) < 0.5f) ? a + w * d : c - d * (1.f - w);
        }
        out[i] = v;
    }
}
int build_latent_input(float* out, const float* hist, const float* ctx_noise, const float* x,
                       int B, int Tq, int n_el, int hist_t_stride, float w, hipStream_t s) {
    const int64_t n = (int64_t)B * Tq * n_el;
    if (n == 0) return 0;
    hipLaunchKernelGGL(build_latent_kernel, grid1d(n), dim3(256), 0, s, out, hist, ctx_noise, x, B, Tq, n_el, hist_t_stride, w);
    D4_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------- tokenizer decoder glue
// 'b c t (h p1) (w p2) -> (b t h w) (p1 p2 c)' (D4:3896) and its inverse (D4:3556); patch row = ((b T + t) nh + h) nw + w
__global__ void video_patch_kernel(const float* src, float* dst, int B, int C, int T, int nh, int nw, int ps, int to_patches) {
    const int H = nh * ps, W = nw * ps, dp = ps * ps * C;
    const int64_t n = (int64_t)B * C * T * H * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        // i enumerates the patch-row layout (contiguous writes when to_patches, contiguous reads otherwise)
        int64_t r = i;
        const int c = (int)(r % C); r /= C;
        const int p2 = (int)(r % ps); r /= ps;
        const int p1 = (int)(r % ps); r /= ps;
        const int w = (int)(r % nw); r /= nw;
        const int h = (int)(r % nh); r /= nh;
        const int t = (int)(r % T); r /= T;
        const int b = (int)r;
        const int64_t v = ((((int64_t)b * C + c) * T + t) * H + (h * ps + p1)) * W + (w * ps + p2);
        if (to_patches) dst[i] = src[v]; else dst[v] = src[i];
        (void)dp;
    }
}
int video_to_patches(const float* video, float* patches, int B, int C, int T, int nh, int nw, int ps, hipStream_t s) {
    const int64_t n = (int64_t)B * C * T * nh * ps * nw * ps;
    if (n == 0) return 0;
    hipLaunchKernelGGL(video_patch_kernel, grid1d(n), dim3(256), 0, s, video, patches, B, C, T, nh, nw, ps, 1);
    D4_LAUNCH_CHECK();
    return 0;
}
int patches_to_video(const float* patches, float* video, int B, int C, int T, int nh, int nw, int ps, hipStream_t s) {
    const int64_t n = (int64_t)B * C * T * nh * ps * nw * ps;
    if (n == 0) return 0;
    hipLaunchKernelGGL(video_patch_kernel, grid1d(n), dim3(256), 0, s, patches, video, B, C, T, nh, nw, ps, 0);
    D4_LAUNCH_CHECK();
    return 0;
}
// tokens[f][s] = s < P ? pos[s] + img[f][s] : lat[f][s - P]   for the n_lat latent tokens kept (the last latent token — the trunk's one
// "special" token — is never visible to another token, D4:1781, and only patch rows are read back: it is dropped); the patch rows
// are also written to the row-compacted copy the final attention pool reads
__global__ void decoder_pack_kernel(float* tokens, float* compact, const float* pos, const float* img, const float* lat, int frames, int P,
                                    int n_lat, int n_total, int D) {
    const int S = P + n_lat;
    const int64_t n = (int64_t)frames * S * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int s = (int)((i / D) % S);
        const int64_t f = i / ((int64_t)D * S);
        float v;
        if (s < P) {
            v = pos[(int64_t)s * D + d] + img[(f * P + s) * D + d];
            compact[(f * P + s) * D + d] = v;
        } else {
            v = lat[(f * n_total + (s - P)) * D + d];
        }
        tokens[i] = v;
    }
}
int decoder_pack_tokens(float* tokens, float* compact, const float* pos, const float* img, const float* lat, int frames, int P, int n_lat, int n_total, int D, hipStream_t s) {
    const int64_t n = (int64_t)frames * (P + n_lat) * D;
    if (n == 0) return 0;
    hipLaunchKernelGGL(decoder_pack_kernel, grid1d(n), dim3(256), 0, s, tokens, compact, pos, img, lat, frames, P, n_lat, n_total, D);
    D4_LAUNCH_CHECK();
    return 0;
}
// encoder: tokens[f][s] = s < P ? img[f][s] : latent_tokens[s - P] (the learned latent tokens are the trunk's special tokens, D4:4361-4376);
// the latent rows also go to the row-compacted copy
__global__ void encoder_pack_kernel(float* tokens, float* compact, const float* img, const float* lt, int frames, int P, int n, int D) {
    const int S = P + n;
    const int64_t tot = (int64_t)frames * S * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int s = (int)((i / D) % S);
        const int64_t f = i / ((int64_t)D * S);
        float v;
        if (s < P) v = img[(f * P + s) * D + d];
        else { v = lt[(int64_t)(s - P) * D + d]; compact[(f * n + (s - P)) * D + d] = v; }
        tokens[i] = v;
    }
}
int encoder_pack_tokens(float* tokens, float* compact, const float* img, const float* latent_tokens, int frames, int P, int n, int D, hipStream_t s) {
    const int64_t tot = (int64_t)frames * (P + n) * D;
    if (tot == 0) return 0;
    hipLaunchKernelGGL(encoder_pack_kernel, grid1d(tot), dim3(256), 0, s, tokens, compact, img, latent_tokens, frames, P, n, D);
    D4_LAUNCH_CHECK();
    return 0;
}
__global__ void tanh_kernel(const float* x, float* y, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = tanhf(x[i]);
}
int tanh_rows(const float* x, float* y, int64_t n, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(tanh_kernel, grid1d(n), dim3(256), 0, s, x, y, n);
    D4_LAUNCH_CHECK();
    return 0;
}
// out[(h, w)][0..1] = (linspace(-1, 1, nh)[h], linspace(-1, 1, nw)[w]), remaining columns of the row zero   D4:3617-3620
__global__ void coord_grid_kernel(float* out, int nh, int nw, int ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nh * nw) return;
    const int h = i / nw, w = i % nw;
    for (int c = 0; c < ld; ++c) out[(int64_t)i * ld + c] = 0.f;
    // torch.linspace(-1, 1, n): start + step * i for the first half, end - step * (n - 1 - i) for the second (n = 1 -> -1)
    auto lin = [](int k, int n) { const float step = n > 1 ? 2.f / (float)(n - 1) : 0.f; return k < n / 2 ? -1.f + step * (float)k : 1.f - step * (float)(n - 1 - k); };
    out[(int64_t)i * ld + 0] = nh > 1 ? lin(h, nh) : -1.f;
    out[(int64_t)i * ld + 1] = nw > 1 ? lin(w, nw) : -1.f;
}
int coord_grid(float* out, int nh, int nw, int ld, hipStream_t s) {
    hipLaunchKernelGGL(coord_grid_kernel, dim3(cdiv(nh * nw, 128)), dim3(128), 0, s, out, nh, nw, ld);
    D4_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------- heads
// HL-Gauss bins -> scalar: softmax(logits) . bin centres                               D4:1088-1096
__global__ __launch_bounds__(256) void hl_gauss_scalar_kernel(const float* logits, int ld, const float* centers, float* out,
                                                              int out_stride, int rows, int bins) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* lr = logits + (int64_t)row * ld;
    float m = -FLT_MAX;
    for (int c = lane; c < bins; c += 64) m = fmaxf(m, lr[c]);
    m = wave_max(m);
    float l = 0.f, acc = 0.f;
    for (int c = lane; c < bins; c += 64) {
        float e = expf(lr[c] - m);
        l += e;
        acc += e * centers[c];
    }
    l = wave_sum(l);
    acc = wave_sum(acc);
    if (lane == 0) out[(int64_t)row * out_stride] = acc / l;
}
int hl_gauss_scalar(const float* logits, int ld, const float* centers, float* out, int out_stride, int rows, int bins, hipStream_t s) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(hl_gauss_scalar_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, logits, ld, centers, out, out_stride, rows, bins);
    D4_LAUNCH_CHECK();
    return 0;
}

// pooled[b][d] = mean_n x[b][n][d]                                                     D4:6606
__global__ void mean_tokens_kernel(const float* x, float* out, int B, int n, int d) {
    const int64_t tot = (int64_t)B * d;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / d), c = (int)(i % d);
        float s = 0.f;
        const float* xb = x + (int64_t)b * n * d + c;
        int j = 0;
        for (; j + 8 <= n; j += 8) {                      // eight loads in flight, summed in token order (the order is part of the result)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = xb[(int64_t)(j + u) * d];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; j < n; ++j) s += xb[(int64_t)j * d];
        out[i] = s / (float)n;
    }
}
int mean_tokens(const float* x, float* out, int B, int n, int d, hipStream_t s) {
    if (B == 0) return 0;
    hipLaunchKernelGGL(mean_tokens_kernel, grid1d((int64_t)B * d), dim3(256), 0, s, x, out, B, n, d);
    D4_LAUNCH_CHECK();
    return 0;
}

// continuous_action_unembed [nc][mtp][d][2] (D4:1244): prediction head 0 as a K-contiguous GEMM weight [2 nc][d] (row 2 n + t), and back
__global__ void cunembed_gather_kernel(const float* U, float* w, int nc, int mtp, int d) {
    const int64_t n = (int64_t)2 * nc * d;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % d), row = (int)(i / d);
        w[i] = U[(((int64_t)(row >> 1) * mtp) * d + k) * 2 + (row & 1)];
    }
}
__global__ void cunembed_scatter_kernel(const float* g, float* dU, int nc, int mtp, int d) {
    const int64_t n = (int64_t)nc * mtp * d * 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i & 1);
        const int64_t r = i >> 1;
        const int k = (int)(r % d), h = (int)((r / d) % mtp), c = (int)(r / ((int64_t)d * mtp));
        dU[i] = h == 0 ? g[((int64_t)(2 * c + t)) * d + k] : 0.f;        // only prediction head 0 is read (pred_head_index = 0)
    }
}
int cunembed_gather(const float* U, float* w, int nc, int mtp, int d, hipStream_t s) {
    if (nc == 0) return 0;
    hipLaunchKernelGGL(cunembed_gather_kernel, grid1d((int64_t)2 * nc * d), dim3(256), 0, s, U, w, nc, mtp, d);
    D4_LAUNCH_CHECK();
    return 0;
}
int cunembed_scatter_grad(const float* g, float* dU, int nc, int mtp, int d, hipStream_t s) {
    if (nc == 0) return 0;
    hipLaunchKernelGGL(cunembed_scatter_kernel, grid1d((int64_t)nc * mtp * d * 2), dim3(256), 0, s, g, dU, nc, mtp, d);
    D4_LAUNCH_CHECK();
    return 0;
}

__device__ __forceinline__ float log_eps(float t) { return logf(fmaxf(t, 1e-20f)); }

// Gumbel-max sampling per action type + log-prob of the sample (D4:485-497, 1374-1376, 1422-1423);
// Bernoulli terminal draw and lens bookkeeping (D4:6611-6616).  One thread per trajectory.
__global__ void sample_kernel(SampleArgs p) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    const float* lg = p.na > 0 ? p.logits + (int64_t)b * p.ld : nullptr;
    const float* u = p.na > 0 ? p.gumbel_u + (int64_t)b * p.ld_u : nullptr;
    const float temp = fmaxf(p.temperature, 1e-10f);
    int o = 0;
    for (int a = 0; a < p.na; ++a) {
        const int n = p.action_sizes[a];
        float best = -FLT_MAX, mx = -FLT_MAX;
        int arg = 0;
        for (int j = 0; j < n; ++j) {
            const float l = lg[o + j];
            const float g = -log_eps(-log_eps(u[o + j]));
            const float sc = l / temp + g;
            if (sc > best) { best = sc; arg = j; }
            mx = fmaxf(mx, l);
        }
        float se = 0.f;
        for (int j = 0; j < n; ++j) se += expf(lg[o + j] - mx);
        p.actions[(int64_t)b * p.act_stride + a] = arg;
        p.log_probs[(int64_t)b * p.lp_stride + a] = (lg[o + arg] - mx) - logf(se);
        o += n;
    }
    // continuous actions: Beta(alpha, beta) sample (tempered) from injected gamma noise + log-prob under the untempered head
    for (int c = 0; c < p.nc; ++c) {
        const float* raw = p.cont_params + (int64_t)b * p.ld_c + 2 * c;
        const BetaAB ab = beta_ab(raw[0], raw[1], p.beta_param);
        const float x = beta_sample(ab.a, ab.b, p.cont_temperature, p.beta_noise + ((int64_t)b * p.nc + c) * (4 * BETA_ROUNDS));
        p.actions_cont[(int64_t)b * p.actc_stride + c] = x;
        p.log_probs_cont[(int64_t)b * p.lpc_stride + c] = beta_log_prob(ab.a, ab.b, x);
    }
    if (p.term_logit) {
        const float pr = sigmoidf(p.term_logit[b]);
        const bool is_term = p.bern_u[b] < pr;
        const bool was = p.terminals[b] != 0;
        if (is_term && !was) p.lens[b] = p.frame_index + 1;
        p.terminals[b] = (was || is_term) ? 1 : 0;
    }
}
int sample_actions_terminals(const SampleArgs& p, hipStream_t s) {
    if (p.B == 0) return 0;
    hipLaunchKernelGGL(sample_kernel, dim3(cdiv(p.B, 128)), dim3(128), 0, s, p);
    D4_LAUNCH_CHECK();
    return 0;
}

}  // namespace d4
