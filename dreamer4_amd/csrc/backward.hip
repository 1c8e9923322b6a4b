// Trunk backward: the blocks of the AxialSpaceTimeTransformer as forward + backward operators on the REFERENCE parameter
// layout — FeedForward (D4:2079-2116) and Attention.forward (D4:1968-2075) in its three uses: within-frame self attention (RMSNorm, q/k/v,
// learned value-residual mix, K-head-RMSNorm, soft clamp, special-token mask, belief projection, head gates, to_out), time attention (the
// same with rotary positions and a causal mask, one problem per token column), and attention over a context (attention pools over the
// stack of layer hiddens, the special tokens' cross attention, learned-query pools) — so that each can be checked directly against
// autograd of the oracle's restatement (tests/test_gpu_backward.py) and composed into the dynamics training forward
// (dreamer4_amd/trunk_ops.py: D4:6956-7743).
//
// The matrix work is the engine's fp32 MFMA GEMM (transposed-operand forms for dX, split-K for dW); what is new here are the two
// attention-core backward kernels (one block per (group, head), everything for <= 32 / 64 items in LDS) and the SiLU-GLU / RMSNorm
// backward glue.  Every backward recomputes its forward: only the block inputs are kept between the two passes.  Nothing here is called
// by the imagination path.
#include "common.h"
#include "kernels.h"
#include "../../include/d4hip.h"
#include <float.h>

namespace d4 {

static constexpr float RMS_EPS = 1.1920928955078125e-07f;   // torch.finfo(float32).eps (nn.RMSNorm eps=None)

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------------------------ SiLU-GLU glue
// hidden layout [rows][2 * Ip]: value half at column 0, gate half at column Ip (Ip = inner rounded up to 16, so that 2 Ip — the contraction
// length of the input-gradient GEMM — is a multiple of 32; pad columns are zero)
static inline int ff_ip(int inner) { return (inner + 15) / 16 * 16; }
__global__ void swiglu_fwd_kernel(const float* h, float* u, int rows, int I, int Ip) {
    const int64_t tot = (int64_t)rows * Ip;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Ip);
        const int64_t r = i / Ip;
        const float a = h[r * 2 * Ip + c], g = h[r * 2 * Ip + Ip + c];
        u[i] = c < I ? a * g * sigm(g) : 0.f;
    }
}
// dh = [du * silu(g) | du * a * silu'(g)]
__global__ void swiglu_bwd_kernel(const float* h, const float* du, float* dh, int rows, int I, int Ip) {
    const int64_t tot = (int64_t)rows * Ip;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Ip);
        const int64_t r = i / Ip;
        const float a = h[r * 2 * Ip + c], g = h[r * 2 * Ip + Ip + c], d = du[i];
        const float s = sigm(g);
        dh[r * 2 * Ip + c] = c < I ? d * g * s : 0.f;
        dh[r * 2 * Ip + Ip + c] = c < I ? d * a * s * (1.f + g * (1.f - s)) : 0.f;
    }
}
static dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)); }

static int gemm_b(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, int M, int N, int K, int flags, hipStream_t s) {
    GemmArgs g{A, lda, W, ldw, C, ldc, bias, nullptr, 0, M, N, K, flags, 0.f};
    return gemm(g, s);
}

// Weight gradient dW[M][N] = A^T B with A [K][M], B [K][N] row-major and K = token rows (thousands to tens of thousands) while M x N is one
// weight matrix (often only 64 tiles): split K over the grid's batch dimension into partial products, then one fixed-order reduce —
// without it these GEMMs run on a quarter of the CUs and were half of a training step (tools/train_step_time.py).
constexpr size_t DW_PART_FLOATS = (size_t)8 << 20;          // partial-product scratch per workspace (32 MB)
static int gemm_dw(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, float* part, hipStream_t s) {
    // operands read straight into the MFMA layout, k split inside the workgroup and over slices (gemm_tn.hip); the transposed-operand form of
    // the general GEMM below remains for leading dimensions that are not multiples of 4
    if (gemm_tn_applicable(A, lda, B, ldb, C, ldc, M, N, K)) return gemm_tn(A, lda, B, ldb, C, ldc, M, N, K, part, part ? DW_PART_FLOATS : 0, s);
    const int64_t tiles = (int64_t)((M + 63) / 64) * ((N + 63) / 64);
    int S = (int)((1024 + tiles - 1) / tiles);
    if (S > K / 256) S = K / 256;
    if ((size_t)S * M * N > DW_PART_FLOATS) S = (int)(DW_PART_FLOATS / ((size_t)M * N));
    if (S <= 1 || !part) return gemm_b(A, lda, B, ldb, C, ldc, nullptr, M, N, K, GEMM_TRANS_A | GEMM_TRANS_B, s);
    const int ks = ((K + S - 1) / S + 31) / 32 * 32;        // k per slice
    const int full = K / ks, rem = K - full * ks;
    int rc;
    GemmArgs g{A, lda, B, ldb, part, N, nullptr, nullptr, 0, M, N, ks, GEMM_TRANS_A | GEMM_TRANS_B, 0.f};
    g.batch = full; g.strideA = (int64_t)ks * lda; g.strideW = (int64_t)ks * ldb; g.strideC = (int64_t)M * N;
    if ((rc = gemm(g, s))) return rc;
    if (rem > 0 && (rc = gemm_b(A + (int64_t)full * ks * lda, lda, B + (int64_t)full * ks * ldb, ldb, part + (int64_t)full * M * N, N, nullptr, M, N, rem,
                                GEMM_TRANS_A | GEMM_TRANS_B, s))) return rc;
    return splitk_reduce(part, full + (rem > 0), M, N, nullptr, 0, C, ldc, s);
}

// Several contiguous copies / zero fills in ONE launch (the concatenated weight images of a block and the pieces of their gradients: what
// used to be a memset and four to six device-to-device copies per block and pass).  Segments must not overlap.
struct MultiCopy {
    static constexpr int MAX = 10;
    float* dst[MAX]; const float* src[MAX]; int64_t n[MAX];       // src null: zero fill
    int count = 0;
    void add(float* d, const float* s, int64_t len) { if (len > 0) { dst[count] = d; src[count] = s; n[count] = len; ++count; } }
};
__global__ __launch_bounds__(256) void multi_copy_kernel(MultiCopy mc) {
    const int seg = blockIdx.y;
    float* __restrict__ d = mc.dst[seg];
    const float* __restrict__ s = mc.src[seg];
    const int64_t n = mc.n[seg];
    const bool vec = ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15) == 0;
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        const int64_t n4 = n / 4;
        for (int64_t i = tid; i < n4; i += stride)
            reinterpret_cast<float4*>(d)[i] = s ? reinterpret_cast<const float4*>(s)[i] : float4{0.f, 0.f, 0.f, 0.f};
        for (int64_t i = n4 * 4 + tid; i < n; i += stride) d[i] = s ? s[i] : 0.f;
    } else {
        for (int64_t i = tid; i < n; i += stride) d[i] = s ? s[i] : 0.f;
    }
}
static int multi_copy(const MultiCopy& mc, hipStream_t s) {
    if (mc.count == 0) return 0;
    D4_REQUIRE(mc.count <= MultiCopy::MAX, "multi_copy: too many segments");
    int64_t big = 0;
    for (int i = 0; i < mc.count; ++i) big = mc.n[i] > big ? mc.n[i] : big;
    const int bx = (int)((big / 4 + 255) / 256 < 1 ? 1 : ((big / 4 + 255) / 256 > 512 ? 512 : (big / 4 + 255) / 256));
    hipLaunchKernelGGL(multi_copy_kernel, dim3(bx, mc.count), dim3(256), 0, s, mc);
    D4_LAUNCH_CHECK();
    return 0;
}

struct FfWs {            // workspace carve-up (floats); all leading dimensions are multiples of 4
    float *xn, *w1p, *b1p, *w2p, *h, *u, *du, *dh, *dxn, *tg, *dw1p, *dw2p, *part, *wt;
    size_t total;
};
static FfWs ff_ws(float* base, int R, int D, int I) {
    const size_t Ip = (size_t)ff_ip(I);
    FfWs w{};
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += (n + 63) / 64 * 64; return p; };
    w.xn = take((size_t)R * D); w.w1p = take(2 * Ip * D); w.b1p = take(2 * Ip); w.w2p = take((size_t)D * Ip);
    w.h = take((size_t)R * 2 * Ip); w.u = take((size_t)R * Ip); w.du = take((size_t)R * Ip); w.dh = take((size_t)R * 2 * Ip);
    w.dxn = take((size_t)R * D); w.tg = take((size_t)R * D); w.dw1p = take(2 * Ip * D); w.dw2p = take((size_t)D * Ip);
    w.part = take(DW_PART_FLOATS);
    w.wt = take(2 * Ip * D);             // transposed weight image of the input-gradient GEMMs (gemm_dx)
    w.total = off;
    return w;
}

// padded copies of proj_in ([a rows | pad | g rows | pad]), its bias, and proj_out (columns padded), then the forward up to u
static int ff_recompute(const FfWs& w, const float* x, const float* norm_w, const float* w_in, const float* b_in, const float* w_out,
                        int R, int D, int I, hipStream_t s) {
    const int Ip = ff_ip(I);
    int rc;
    MultiCopy mc;
    mc.add(w.w1p, w_in, (int64_t)I * D); mc.add(w.w1p + (size_t)I * D, nullptr, (int64_t)(Ip - I) * D);
    mc.add(w.w1p + (size_t)Ip * D, w_in + (size_t)I * D, (int64_t)I * D); mc.add(w.w1p + (size_t)(Ip + I) * D, nullptr, (int64_t)(Ip - I) * D);
    mc.add(w.b1p, b_in, I); mc.add(w.b1p + I, nullptr, Ip - I); mc.add(w.b1p + Ip, b_in + I, I); mc.add(w.b1p + Ip + I, nullptr, Ip - I);
    if ((rc = multi_copy(mc, s))) return rc;
    if ((rc = pad_cols(w_out, w.w2p, D, I, Ip, s))) return rc;
    if ((rc = rmsnorm_rows(x, D, norm_w, w.xn, D, R, D, RMS_EPS, s))) return rc;
    if ((rc = gemm_b(w.xn, D, w.w1p, D, w.h, 2 * Ip, w.b1p, R, 2 * Ip, D, 0, s))) return rc;
    hipLaunchKernelGGL(swiglu_fwd_kernel, grid_for((int64_t)R * Ip), dim3(256), 0, s, w.h, w.u, R, I, Ip);
    D4_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ attention core backward
// One block (4 waves) per (frame, head); S <= 32 tokens, head dim DH <= 64 (lanes >= DH idle).  Rows of proj: q @ 0, k @ hd, v @ 2hd,
// gate logit @ 3hd + head, mix logit @ 3hd + hp4 + head  (hd = heads * DH, hp4 = heads rounded up to 4).
struct AttnBwdArgs {
    const float* proj; int ldp;        // [F*S][ldp] forward projections (mix logits include their bias)
    const float* rv;                   // [F*S][hd] value residual or null
    const float* gamma;                // [heads][DH]
    const float* d_o3;                 // [F*S][hd] gradient of the gated attention output (before to_out); null: forward only
    float* o3;                         // [F*S][hd] out: gated attention output (recomputed forward)
    float* dproj;                      // [F*S][ldp] out: gradients of the projections (same columns)
    float* d_rv;                       // [F*S][hd] out (when rv)
    float* dgamma_part;                // [F][hd] out: per-frame partial of d gamma
    int F, S, heads, hp4;              // F groups of S items
    float softclamp; int num_special, belief;
    // row of item j of group g = (g / g_inner) * g_outer_stride + (g % g_inner) + j * item_stride:
    //   within-frame attention: g_inner 1, g_outer_stride S, item_stride 1;  time attention over [B][T][S] rows: g_inner S, g_outer_stride T * S, item_stride S
    int g_inner = 1; int64_t g_outer_stride = 0, item_stride = 1;
    int causal = 0;                    // item i sees items j <= i
    const float* inv_freq = nullptr;   // [DH / 2] rotary frequencies applied to q and k at position j (time attention), or null
};

constexpr int AB_S = 64, AB_LD = 65;           // items (tokens of a frame / frames of a trajectory) per group: <= 64 (a score row = one wavefront)
// CAP: token capacity of the LDS arrays (16 or 32): at <= 16 tokens per group the block needs 27 KB instead of 60 KB of LDS, so five blocks
// instead of two share a CU — the kernel is a chain of LDS reads and wave reductions, and the extra waves are what hides them
template <int DH, int CAP>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnBwdArgs p) {
    extern __shared__ float ab_s[];             // dynamic: 27 KB (CAP 16), 60 KB (32), 134 KB (64: one block per CU)
    float* qs = ab_s; float* kn = qs + CAP * AB_LD; float* kh = kn + CAP * AB_LD; float* vm = kh + CAP * AB_LD; float* dO = vm + CAP * AB_LD;
    float* dvm = dO + CAP * AB_LD;
    float* P = dvm + CAP * AB_LD; float* dsm = P + CAP * (CAP + 1);
    float* kinv = dsm + CAP * (CAP + 1); float* vinv = kinv + CAP; float* mxs = vinv + CAP; float* gts = mxs + CAP;
    __shared__ float gpart[4][64];
    const int S = p.S, hd = p.heads * DH;
    const int f = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool on = lane < DH;
    const int64_t row0 = (int64_t)(f / p.g_inner) * p.g_outer_stride + (f % p.g_inner);
    const int64_t ist = p.item_stride;
    const float sc = on ? (p.gamma[h * DH + lane] + 1.f) * sqrtf((float)DH) : 0.f;
    const float scale = rsqrtf((float)DH);
    const bool bwd = p.d_o3 != nullptr;

    // rotary (D4:1626-1659): t * cos + rotate_half(t) * sin with rotate_half(t)[d] = d < DH/2 ? -t[d + DH/2] : t[d - DH/2]; the partner
    // feature is lane ^ (DH/2).  rot_t is the transpose (backward): y * cos + rotate_half^T(y * sin), rotate_half^T(z)[d] = d < DH/2 ? z[d + DH/2] : -z[d - DH/2]
    const float freq = (p.inv_freq && on) ? p.inv_freq[lane & (DH / 2 - 1)] : 0.f;
    const bool lo_half = lane < DH / 2;
    auto rot = [&](float t, int pos) {
        if (!p.inv_freq) return t;
        float sn, cs;
        sincosf(freq * (float)pos, &sn, &cs);
        const float partner = __shfl(t, lane ^ (DH / 2));
        return on ? t * cs + (lo_half ? -partner : partner) * sn : 0.f;
    };
    auto rot_t = [&](float y, int pos) {
        if (!p.inv_freq) return y;
        float sn, cs;
        sincosf(freq * (float)pos, &sn, &cs);
        const float partner = __shfl(y * sn, lane ^ (DH / 2));
        return on ? y * cs + (lo_half ? partner : -partner) : 0.f;
    };

    // ---- phase A: per token j: value mix, key normalisation
    for (int j = w; j < S; j += 4) {
        const float* pr = p.proj + (row0 + j * ist) * p.ldp;
        const float qv = on ? pr[h * DH + lane] : 0.f, kv = on ? pr[hd + h * DH + lane] : 0.f;
        float vv = on ? pr[2 * hd + h * DH + lane] : 0.f;
        float mx = 0.f;
        if (p.rv) {
            mx = sigm(pr[3 * hd + p.hp4 + h]);
            const float r = on ? p.rv[(row0 + j * ist) * hd + h * DH + lane] : 0.f;
            vv = vv + mx * (r - vv);
        }
        const float ki = 1.f / fmaxf(sqrtf(wave_sum(kv * kv)), 1e-12f);
        const float vi = 1.f / fmaxf(sqrtf(wave_sum(vv * vv)), 1e-12f);
        qs[j * AB_LD + lane] = rot(qv, j); kh[j * AB_LD + lane] = kv * ki; kn[j * AB_LD + lane] = rot(kv * ki * sc, j);
        vm[j * AB_LD + lane] = vv;
        if (lane == 0) { kinv[j] = ki; vinv[j] = vi; mxs[j] = mx; gts[j] = sigm(pr[3 * hd + h]); }
    }
    __syncthreads();

    // ---- phase B: per query i (lane = key j for the score row, lane = feature for the vectors)
    const int first_special = S - p.num_special;
    for (int i = w; i < S; i += 4) {
        float dot = 0.f;
        if (lane < S) {
#pragma unroll 8
            for (int d = 0; d < DH; ++d) dot += qs[i * AB_LD + d] * kn[lane * AB_LD + d];
        }
        const float sim = dot * scale;
        float th = 0.f, simc = sim;
        if (p.softclamp > 0.f) { th = tanhf(sim / p.softclamp); simc = th * p.softclamp; }
        const bool visible = lane < S && !(i < first_special && lane >= first_special) && !(p.causal && lane > i);
        simc = visible ? simc : -FLT_MAX;
        float m = simc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        const float e = visible ? expf(simc - m) : 0.f;
        const float pij = e / wave_sum(e);
        // o_i = sum_j P_ij vm_j
        float o = 0.f;
        for (int j = 0; j < S; ++j) o += __shfl(pij, j) * vm[j * AB_LD + lane];
        const float vni = vm[i * AB_LD + lane] * vinv[i];
        const float sdot = p.belief ? wave_sum(o * vni) : 0.f;
        const float o2 = o - sdot * vni;
        const float gt = gts[i];
        if (on) p.o3[(row0 + i * ist) * hd + h * DH + lane] = o2 * gt;
        if (!bwd) continue;
        const float d3 = on ? p.d_o3[(row0 + i * ist) * hd + h * DH + lane] : 0.f;
        const float dgl = wave_sum(d3 * o2) * gt * (1.f - gt);
        if (lane == 0) p.dproj[(row0 + i * ist) * p.ldp + 3 * hd + h] = dgl;
        const float d2 = d3 * gt;
        float dOi = d2, dvdir = 0.f;
        if (p.belief) {
            const float c2 = wave_sum(d2 * vni);
            dOi = d2 - c2 * vni;
            const float dvn = -(sdot * d2 + c2 * o);
            dvdir = (dvn - wave_sum(dvn * vni) * vni) * vinv[i];
        }
        dO[i * AB_LD + lane] = dOi;
        dvm[i * AB_LD + lane] = dvdir;
        // dP_ij = dO_i . vm_j  (lane = j)
        float dp = 0.f;
        if (lane < S) {
#pragma unroll 8
            for (int d = 0; d < DH; ++d) dp += dO[i * AB_LD + d] * vm[lane * AB_LD + d];
        }
        const float rowdot = wave_sum(pij * dp);
        float dsim = pij * (dp - rowdot);
        if (p.softclamp > 0.f) dsim *= 1.f - th * th;
        dsim *= scale;
        if (lane < S) { P[i * (CAP + 1) + lane] = pij; dsm[i * (CAP + 1) + lane] = dsim; }
        // dq_i = sum_j dsim_ij kn_j
        float dq = 0.f;
        for (int j = 0; j < S; ++j) dq += __shfl(dsim, j) * kn[j * AB_LD + lane];
        dq = rot_t(dq, i);
        if (on) p.dproj[(row0 + i * ist) * p.ldp + h * DH + lane] = dq;
    }
    if (!bwd) return;
    __syncthreads();

    // ---- phase C: per key / value token j (lane = feature)
    float gacc = 0.f;
    for (int j = w; j < S; j += 4) {
        float dkn = 0.f, dv = dvm[j * AB_LD + lane];
        for (int i = 0; i < S; ++i) {
            dkn += dsm[i * (CAP + 1) + j] * qs[i * AB_LD + lane];
            dv += P[i * (CAP + 1) + j] * dO[i * AB_LD + lane];
        }
        dkn = rot_t(dkn, j);
        const float khj = kh[j * AB_LD + lane];
        gacc += dkn * khj;
        const float dkh = dkn * sc;
        const float dk = (dkh - wave_sum(dkh * khj) * khj) * kinv[j];
        float* dr = p.dproj + (row0 + j * ist) * p.ldp;
        if (on) dr[hd + h * DH + lane] = dk;
        if (p.rv) {
            const float mx = mxs[j];
            const float* pr = p.proj + (row0 + j * ist) * p.ldp;
            const float vraw = on ? pr[2 * hd + h * DH + lane] : 0.f;
            const float r = on ? p.rv[(row0 + j * ist) * hd + h * DH + lane] : 0.f;
            const float dmx = wave_sum(dv * (r - vraw));
            if (on) { dr[2 * hd + h * DH + lane] = dv * (1.f - mx); p.d_rv[(row0 + j * ist) * hd + h * DH + lane] = dv * mx; }
            if (lane == 0) dr[3 * hd + p.hp4 + h] = dmx * mx * (1.f - mx);
        } else {
            if (on) dr[2 * hd + h * DH + lane] = dv;
            if (lane == 0) dr[3 * hd + p.hp4 + h] = 0.f;
        }
    }
    gpart[w][lane] = gacc;
    __syncthreads();
    if (w == 0 && on) p.dgamma_part[(int64_t)f * hd + h * DH + lane] = (((gpart[0][lane] + gpart[1][lane]) + gpart[2][lane]) + gpart[3][lane]) * sqrtf((float)DH);
}

static int attn_core(const AttnBwdArgs& a, int dh, hipStream_t s) {
    if (a.F * a.heads == 0) return 0;
    auto lds_of = [](int cap) { return (size_t)(6 * cap * AB_LD + 2 * cap * (cap + 1) + 4 * cap) * sizeof(float); };
    static bool attr = false;
    if (!attr) {
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<64, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(64)));
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<32, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(64)));
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<16, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(64)));
        attr = true;
    }
    const dim3 grid(a.F * a.heads), block(256);
    const int cap = a.S <= 16 ? 16 : (a.S <= 32 ? 32 : 64);
#define D4_AB_LAUNCH(DH_, CAP_) hipLaunchKernelGGL((attn_bwd_kernel<DH_, CAP_>), grid, block, lds_of(CAP_), s, a)
    if (dh == 64) { if (cap == 16) D4_AB_LAUNCH(64, 16); else if (cap == 32) D4_AB_LAUNCH(64, 32); else D4_AB_LAUNCH(64, 64); }
    else if (dh == 32) { if (cap <= 32) D4_AB_LAUNCH(32, 32); else D4_AB_LAUNCH(32, 64); }
    else { if (cap <= 32) D4_AB_LAUNCH(16, 32); else D4_AB_LAUNCH(16, 64); }
#undef D4_AB_LAUNCH
    D4_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ cross-attention core
// The trunk's cross attentions (D4:1968-2075 with a context): the AttentionPool over the stack of layer hiddens (one query per token,
// D4:2143-2177), the final special-token cross attention (D4:3227-3234) and the learned-query pools (D4:2179-2210).  No value residual,
// no belief projection (the reference skips it when a context is given), optional soft clamp.  One block per (group, head); up to 64
// queries and 64 keys in (dynamic) LDS.  projq rows: q @ 0, gate logit @ hd + head; projk rows: k @ 0, v @ hd.
struct XAttnArgs {
    const float* projq; int ldq;       // [G * nq][ldq], row g * nq + i
    const float* projk; int ldk;       // key j of group g: row g * nk + j (group major) or j * G + g (item major: the stack of hiddens)
    const float* gamma;
    const float* d_o3;                 // [G * nq][hd] or null (forward only)
    float* o3;                         // [G * nq][hd]
    float* dprojq; float* dprojk;      // gradients, same layouts
    float* dgamma_part;                // [G][hd]
    int G, nq, nk, heads, item_major;
    float softclamp;
};

constexpr int XA_N = 64;
template <int DH>
__global__ __launch_bounds__(256) void xattn_bwd_kernel(XAttnArgs p) {
    extern __shared__ float xa_s[];
    const int nq = p.nq, nk = p.nk, hd = p.heads * DH;
    float* kn = xa_s;                           // [nk][65]
    float* kh = kn + nk * AB_LD;
    float* vs = kh + nk * AB_LD;
    float* qs = vs + nk * AB_LD;                // [nq][65]
    float* dO = qs + nq * AB_LD;
    float* P = dO + nq * AB_LD;                 // [nq][65]
    float* dsm = P + nq * AB_LD;
    float* kinv = dsm + nq * AB_LD;             // [nk]
    float* gts = kinv + nk;                     // [nq]
    float* gpart = gts + nq;                    // [4][64]
    const int g = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool on = lane < DH;
    const float sc = on ? (p.gamma[h * DH + lane] + 1.f) * sqrtf((float)DH) : 0.f;
    const float scale = rsqrtf((float)DH);
    const bool bwd = p.d_o3 != nullptr;
    auto krow = [&](int j) { return p.item_major ? (int64_t)j * p.G + g : (int64_t)g * nk + j; };

    for (int j = w; j < nk; j += 4) {
        const float* pr = p.projk + krow(j) * p.ldk;
        const float kv = on ? pr[h * DH + lane] : 0.f, vv = on ? pr[hd + h * DH + lane] : 0.f;
        const float ki = 1.f / fmaxf(sqrtf(wave_sum(kv * kv)), 1e-12f);
        kh[j * AB_LD + lane] = kv * ki; kn[j * AB_LD + lane] = kv * ki * sc; vs[j * AB_LD + lane] = vv;
        if (lane == 0) kinv[j] = ki;
    }
    for (int i = w; i < nq; i += 4) {
        const float* pr = p.projq + ((int64_t)g * nq + i) * p.ldq;
        qs[i * AB_LD + lane] = on ? pr[h * DH + lane] : 0.f;
        if (lane == 0) gts[i] = sigm(pr[hd + h]);
    }
    __syncthreads();

    for (int i = w; i < nq; i += 4) {
        const int64_t qrow = (int64_t)g * nq + i;
        float dot = 0.f;
        if (lane < nk) {
#pragma unroll 8
            for (int d = 0; d < DH; ++d) dot += qs[i * AB_LD + d] * kn[lane * AB_LD + d];
        }
        const float sim = dot * scale;
        float th = 0.f, simc = sim;
        if (p.softclamp > 0.f) { th = tanhf(sim / p.softclamp); simc = th * p.softclamp; }
        simc = lane < nk ? simc : -FLT_MAX;
        float m = simc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        const float e = lane < nk ? expf(simc - m) : 0.f;
        const float pij = e / wave_sum(e);
        float o = 0.f;
        for (int j = 0; j < nk; ++j) o += __shfl(pij, j) * vs[j * AB_LD + lane];
        const float gt = gts[i];
        if (on) p.o3[qrow * hd + h * DH + lane] = o * gt;
        if (!bwd) continue;
        const float d3 = on ? p.d_o3[qrow * hd + h * DH + lane] : 0.f;
        const float dgl = wave_sum(d3 * o) * gt * (1.f - gt);
        if (lane == 0) p.dprojq[qrow * p.ldq + hd + h] = dgl;
        const float dOi = d3 * gt;
        dO[i * AB_LD + lane] = dOi;
        float dp = 0.f;
        if (lane < nk) {
#pragma unroll 8
            for (int d = 0; d < DH; ++d) dp += dO[i * AB_LD + d] * vs[lane * AB_LD + d];
        }
        const float rowdot = wave_sum(pij * dp);
        float dsim = pij * (dp - rowdot);
        if (p.softclamp > 0.f) dsim *= 1.f - th * th;
        dsim *= scale;
        if (lane < nk) { P[i * AB_LD + lane] = pij; dsm[i * AB_LD + lane] = dsim; }
        float dq = 0.f;
        for (int j = 0; j < nk; ++j) dq += __shfl(dsim, j) * kn[j * AB_LD + lane];
        if (on) p.dprojq[qrow * p.ldq + h * DH + lane] = dq;
    }
    if (!bwd) return;
    __syncthreads();

    float gacc = 0.f;
    for (int j = w; j < nk; j += 4) {
        float dkn = 0.f, dv = 0.f;
        for (int i = 0; i < nq; ++i) {
            dkn += dsm[i * AB_LD + j] * qs[i * AB_LD + lane];
            dv += P[i * AB_LD + j] * dO[i * AB_LD + lane];
        }
        const float khj = kh[j * AB_LD + lane];
        gacc += dkn * khj;
        const float dkh = dkn * sc;
        const float dk = (dkh - wave_sum(dkh * khj) * khj) * kinv[j];
        float* dr = p.dprojk + krow(j) * p.ldk;
        if (on) { dr[h * DH + lane] = dk; dr[hd + h * DH + lane] = dv; }
    }
    gpart[w * 64 + lane] = gacc;
    __syncthreads();
    if (w == 0 && on) p.dgamma_part[(int64_t)g * hd + h * DH + lane] = (((gpart[lane] + gpart[64 + lane]) + gpart[128 + lane]) + gpart[192 + lane]) * sqrtf((float)DH);
}

static int xattn_core(const XAttnArgs& a, int dh, hipStream_t s) {
    if (a.G * a.heads == 0) return 0;
    const size_t lds = sizeof(float) * ((size_t)(3 * a.nk + 4 * a.nq) * AB_LD + a.nk + a.nq + 256);
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        const int mx = (int)(sizeof(float) * ((size_t)7 * XA_N * AB_LD + 2 * XA_N + 256));
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_bwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, mx));
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_bwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, mx));
        D4_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_bwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, mx));
        attr_set.done();
    }
    if (dh == 64) hipLaunchKernelGGL(xattn_bwd_kernel<64>, dim3(a.G * a.heads), dim3(256), lds, s, a);
    else if (dh == 32) hipLaunchKernelGGL(xattn_bwd_kernel<32>, dim3(a.G * a.heads), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(xattn_bwd_kernel<16>, dim3(a.G * a.heads), dim3(256), lds, s, a);
    D4_LAUNCH_CHECK();
    return 0;
}

__global__ void zero_pad_cols_kernel(float* x, int rows, int ld, int c0, int c1) {
    const int n = c1 - c0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)rows * n; i += (int64_t)gridDim.x * blockDim.x)
        x[(i / n) * ld + c0 + (int)(i % n)] = 0.f;
}

struct AttnWs {
    float *xn, *wcat, *bcat, *proj, *dproj, *d_o3, *o3, *dwcat, *tg, *dxn, *gpart, *part, *wt;
    size_t total;
    int P, hp4;
};
static AttnWs attn_ws(float* base, int R, int F, int D, int heads, int dh) {
    AttnWs w{};
    const int hd = heads * dh;
    w.hp4 = (heads + 3) / 4 * 4;
    w.P = (3 * hd + 2 * w.hp4 + 31) / 32 * 32;       // a multiple of 32: P is the contraction length of the input-gradient GEMM (gemm_dx)
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += (n + 63) / 64 * 64; return p; };
    w.xn = take((size_t)R * D); w.wcat = take((size_t)w.P * D); w.bcat = take(w.P); w.proj = take((size_t)R * w.P); w.dproj = take((size_t)R * w.P);
    w.d_o3 = take((size_t)R * hd); w.o3 = take((size_t)R * hd); w.dwcat = take((size_t)w.P * D); w.tg = take((size_t)R * D); w.dxn = take((size_t)R * D);
    w.gpart = take((size_t)F * hd);
    w.part = take(DW_PART_FLOATS);
    w.wt = take((size_t)w.P * D);
    w.total = off;
    return w;
}

struct AttnParams { const float *norm_w, *wq, *wk, *wv, *wo, *wg, *wm, *bm, *gamma; };

// concatenated projection [q | k | v | gates (hp4) | mix (hp4)] and the forward up to the projections
static int attn_project(const AttnWs& w, const float* x, const AttnParams& prm, int R, int D, int heads, int dh, bool has_rv, hipStream_t s) {
    const int hd = heads * dh;
    int rc;
    {   // rows of wcat: q | k | v | gates, pad to hp4 | mix (or zero), pad to P;  bcat: zero but for the mix bias
        const int64_t blk = (int64_t)hd * D;
        const int mix0 = 3 * hd + w.hp4;
        MultiCopy mc;
        mc.add(w.wcat, prm.wq, blk); mc.add(w.wcat + blk, prm.wk, blk); mc.add(w.wcat + 2 * blk, prm.wv, blk);
        mc.add(w.wcat + 3 * blk, prm.wg, (int64_t)heads * D); mc.add(w.wcat + 3 * blk + (int64_t)heads * D, nullptr, (int64_t)(w.hp4 - heads) * D);
        mc.add(w.wcat + (int64_t)mix0 * D, has_rv ? prm.wm : nullptr, (int64_t)heads * D);
        mc.add(w.wcat + (int64_t)(mix0 + heads) * D, nullptr, (int64_t)(w.P - mix0 - heads) * D);
        mc.add(w.bcat, nullptr, mix0); mc.add(w.bcat + mix0, has_rv ? prm.bm : nullptr, heads); mc.add(w.bcat + mix0 + heads, nullptr, w.P - mix0 - heads);
        if ((rc = multi_copy(mc, s))) return rc;
    }
    if ((rc = rmsnorm_rows(x, D, prm.norm_w, w.xn, D, R, D, RMS_EPS, s))) return rc;
    return gemm_b(w.xn, D, w.wcat, D, w.proj, w.P, w.bcat, R, w.P, D, 0, s);
}

}  // namespace d4

using namespace d4;

extern "C" {

size_t d4_ff_workspace_bytes(int rows, int dim, int inner) { return ff_ws(nullptr, rows, dim, inner).total * sizeof(float); }

int d4_ff_forward(const float* x, const float* norm_w, const float* w_in, const float* b_in, const float* w_out, const float* b_out,
                  int rows, int dim, int inner, float* y, float* workspace, size_t workspace_bytes, void* stream) {
    D4_REQUIRE(x && norm_w && w_in && b_in && w_out && b_out && y && workspace, "d4_ff_forward: null argument");
    D4_REQUIRE(dim % 4 == 0 && ((uintptr_t)workspace % 256) == 0, "d4_ff_forward: dim must be a multiple of 4 and the workspace 256-byte aligned");
    D4_REQUIRE(workspace_bytes >= d4_ff_workspace_bytes(rows, dim, inner), "d4_ff_forward: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (rows == 0) return 0;
    const FfWs w = ff_ws(workspace, rows, dim, inner);
    const int Ip = ff_ip(inner);
    int rc;
    if ((rc = ff_recompute(w, x, norm_w, w_in, b_in, w_out, rows, dim, inner, s))) return rc;
    return gemm_b(w.u, Ip, w.w2p, Ip, y, dim, b_out, rows, dim, Ip, 0, s);
}

// `reuse`: the workspace is the one the matching forward call ran in and still holds its intermediates (normalised input, padded weight
// images, pre-activation h, hidden u): nothing is recomputed.  Otherwise the forward up to u is recomputed from x first.
static int ff_backward_impl(const float* x, const float* dy, const float* norm_w, const float* w_in, const float* b_in, const float* w_out,
                   int rows, int dim, int inner, float* dx, float* d_norm_w, float* d_w_in, float* d_b_in, float* d_w_out, float* d_b_out,
                   float* workspace, size_t workspace_bytes, void* stream, bool reuse) {
    D4_REQUIRE(x && dy && norm_w && w_in && b_in && w_out && dx && d_norm_w && d_w_in && d_b_in && d_w_out && d_b_out && workspace, "d4_ff_backward: null argument");
    D4_REQUIRE(dim % 4 == 0 && ((uintptr_t)workspace % 256) == 0, "d4_ff_backward: dim must be a multiple of 4 and the workspace 256-byte aligned");
    D4_REQUIRE(workspace_bytes >= d4_ff_workspace_bytes(rows, dim, inner), "d4_ff_backward: workspace too small");
    D4_REQUIRE(rows >= 1, "d4_ff_backward: no rows");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int R = rows, D = dim, I = inner, Ip = ff_ip(inner);
    const FfWs w = ff_ws(workspace, R, D, I);
    int rc;
    if (!reuse && (rc = ff_recompute(w, x, norm_w, w_in, b_in, w_out, R, D, I, s))) return rc;
    // y = u W2^T + b2
    if ((rc = gemm_dw(dy, D, w.u, Ip, w.dw2p, Ip, D, Ip, R, w.part, s))) return rc;                                    // dW2 = dy^T u
    if ((rc = copy_rows(w.dw2p, Ip, d_w_out, I, D, I, s))) return rc;
    if ((rc = gemm_dx(dy, D, w.w2p, Ip, w.du, Ip, R, Ip, D, w.wt, s))) return rc;                                        // du = dy W2
    hipLaunchKernelGGL(swiglu_bwd_kernel, grid_for((int64_t)R * Ip), dim3(256), 0, s, w.h, w.du, w.dh, R, I, Ip);
    D4_LAUNCH_CHECK();
    // h = xn W1^T + b1
    if ((rc = gemm_dw(w.dh, 2 * Ip, w.xn, D, w.dw1p, D, 2 * Ip, D, R, w.part, s))) return rc;                          // dW1 = dh^T xn
    {
        MultiCopy mc;
        mc.add(d_w_in, w.dw1p, (int64_t)I * D); mc.add(d_w_in + (size_t)I * D, w.dw1p + (size_t)Ip * D, (int64_t)I * D);
        if ((rc = multi_copy(mc, s))) return rc;
    }
    if ((rc = gemm_dx(w.dh, 2 * Ip, w.w1p, D, w.dxn, D, R, D, 2 * Ip, w.wt, s))) return rc;                             // dxn = dh W1
    // xn = rmsnorm(x) * gamma
    if ((rc = rmsnorm_bwd(x, w.dxn, norm_w, w.tg, dx, R, D, RMS_EPS, s))) return rc;
    // the four column sums of the block (both bias gradients, the norm gain's) in one launch: their operands are all still in place
    ColsumBatch cb;
    cb.add(dy, D, R, D, d_b_out); cb.add(w.dh, 2 * Ip, R, I, d_b_in); cb.add(w.dh + Ip, 2 * Ip, R, I, d_b_in + I); cb.add(w.tg, D, R, D, d_norm_w);
    return colsum_batch(cb, s);
}

int d4_ff_backward(const float* x, const float* dy, const float* norm_w, const float* w_in, const float* b_in, const float* w_out,
                   int rows, int dim, int inner, float* dx, float* d_norm_w, float* d_w_in, float* d_b_in, float* d_w_out, float* d_b_out,
                   float* workspace, size_t workspace_bytes, void* stream) {
    return ff_backward_impl(x, dy, norm_w, w_in, b_in, w_out, rows, dim, inner, dx, d_norm_w, d_w_in, d_b_in, d_w_out, d_b_out, workspace, workspace_bytes, stream, false);
}
int d4_ff_backward_saved(const float* x, const float* dy, const float* norm_w, const float* w_in, const float* b_in, const float* w_out,
                         int rows, int dim, int inner, float* dx, float* d_norm_w, float* d_w_in, float* d_b_in, float* d_w_out, float* d_b_out,
                         float* workspace, size_t workspace_bytes, void* stream) {
    return ff_backward_impl(x, dy, norm_w, w_in, b_in, w_out, rows, dim, inner, dx, d_norm_w, d_w_in, d_b_in, d_w_out, d_b_out, workspace, workspace_bytes, stream, true);
}

size_t d4_attn_workspace_bytes(int frames, int tokens, int dim, int heads, int dim_head) {
    return attn_ws(nullptr, frames * tokens, frames, dim, heads, dim_head).total * sizeof(float);
}

}  // extern "C"

namespace {

struct AttnGeom {                  // how the rows of x group into attention problems (see AttnBwdArgs)
    int groups, items, g_inner; int64_t g_outer_stride, item_stride;
    int causal, num_special; const float* inv_freq;
};

int attn_check(int rows, const AttnGeom& g, int dim, int heads, int dim_head, const float* workspace, size_t workspace_bytes) {
    D4_REQUIRE(g.items >= 1 && g.items <= AB_S, "attention block: %d items per group (max %d)", g.items, AB_S);
    D4_REQUIRE(dim_head == 16 || dim_head == 32 || dim_head == 64, "attention block: head dim %d (16, 32 or 64)", dim_head);
    D4_REQUIRE(dim % 4 == 0 && ((uintptr_t)workspace % 256) == 0, "attention block: dim must be a multiple of 4 and the workspace 256-byte aligned");
    D4_REQUIRE(workspace_bytes >= attn_ws(nullptr, rows, g.groups, dim, heads, dim_head).total * sizeof(float), "attention block: workspace too small");
    return 0;
}

void set_geom(AttnBwdArgs& a, const AttnGeom& g) {
    a.g_inner = g.g_inner; a.g_outer_stride = g.g_outer_stride; a.item_stride = g.item_stride; a.causal = g.causal; a.inv_freq = g.inv_freq;
}

int attn_block_forward(const float* x, const float* residual_values, const AttnParams& prm, int rows, const AttnGeom& g, int dim, int heads, int dim_head,
                       float softclamp, int belief, float* y, float* workspace, size_t workspace_bytes, hipStream_t s) {
    D4_REQUIRE(x && prm.norm_w && prm.wq && prm.wk && prm.wv && prm.wo && prm.wg && prm.gamma && y && workspace, "attention forward: null argument");
    D4_REQUIRE(!residual_values || (prm.wm && prm.bm), "attention forward: residual values need the mix projection");
    int rc;
    if ((rc = attn_check(rows, g, dim, heads, dim_head, workspace, workspace_bytes))) return rc;
    const int R = rows, hd = heads * dim_head;
    if (R == 0) return 0;
    const AttnWs w = attn_ws(workspace, R, g.groups, dim, heads, dim_head);
    if ((rc = attn_project(w, x, prm, R, dim, heads, dim_head, residual_values != nullptr, s))) return rc;
    AttnBwdArgs a{w.proj, w.P, residual_values, prm.gamma, nullptr, w.o3, nullptr, nullptr, nullptr, g.groups, g.items, heads, w.hp4, softclamp, g.num_special, belief};
    set_geom(a, g);
    if ((rc = attn_core(a, dim_head, s))) return rc;
    return gemm_b(w.o3, hd, prm.wo, hd, y, dim, nullptr, R, dim, hd, 0, s);
}

struct AttnGrads { float *dx, *d_rv, *d_norm_w, *d_wq, *d_wk, *d_wv, *d_wo, *d_wg, *d_wm, *d_bm, *d_gamma; };

int attn_block_backward(const float* x, const float* residual_values, const float* dy, const AttnParams& prm, int rows, const AttnGeom& g, int dim,
                        int heads, int dim_head, float softclamp, int belief, const AttnGrads& o, float* workspace, size_t workspace_bytes, hipStream_t s,
                        bool reuse = false) {
    D4_REQUIRE(x && dy && prm.norm_w && prm.wq && prm.wk && prm.wv && prm.wo && prm.wg && prm.gamma && workspace, "attention backward: null argument");
    D4_REQUIRE(o.dx && o.d_norm_w && o.d_wq && o.d_wk && o.d_wv && o.d_wo && o.d_wg && o.d_gamma, "attention backward: null gradient output");
    D4_REQUIRE(!residual_values || (prm.wm && prm.bm && o.d_rv && o.d_wm && o.d_bm), "attention backward: residual values need the mix projection and its gradients");
    int rc;
    if ((rc = attn_check(rows, g, dim, heads, dim_head, workspace, workspace_bytes))) return rc;
    D4_REQUIRE(rows >= 1, "attention backward: no rows");
    const int R = rows, D = dim, hd = heads * dim_head;
    const bool has_rv = residual_values != nullptr;
    const AttnWs w = attn_ws(workspace, R, g.groups, D, heads, dim_head);
    if (!reuse && (rc = attn_project(w, x, prm, R, D, heads, dim_head, has_rv, s))) return rc;     // reuse: the forward's xn / wcat / proj are still there
    // out = o3 Wo^T
    if ((rc = gemm_dx(dy, D, prm.wo, hd, w.d_o3, hd, R, hd, D, w.wt, s))) return rc;                                     // d_o3 = dy Wo
    AttnBwdArgs a{w.proj, w.P, residual_values, prm.gamma, w.d_o3, w.o3, w.dproj, o.d_rv, w.gpart, g.groups, g.items, heads, w.hp4, softclamp, g.num_special, belief};
    set_geom(a, g);
    if ((rc = attn_core(a, dim_head, s))) return rc;
    // the pad columns of the gate / mix logits and of the row carry no gradient
    if (w.hp4 > heads) hipLaunchKernelGGL(zero_pad_cols_kernel, grid_for((int64_t)R * (w.hp4 - heads)), dim3(256), 0, s, w.dproj, R, w.P, 3 * hd + heads, 3 * hd + w.hp4);
    if (w.P > 3 * hd + w.hp4 + heads)
        hipLaunchKernelGGL(zero_pad_cols_kernel, grid_for((int64_t)R * (w.P - (3 * hd + w.hp4 + heads))), dim3(256), 0, s, w.dproj, R, w.P, 3 * hd + w.hp4 + heads, w.P);
    D4_LAUNCH_CHECK();
    if ((rc = gemm_dw(dy, D, w.o3, hd, o.d_wo, hd, D, hd, R, w.part, s))) return rc;                                     // dWo = dy^T o3
    // projections: dW = dproj^T xn, dxn = dproj Wcat
    if ((rc = gemm_dw(w.dproj, w.P, w.xn, D, w.dwcat, D, w.P, D, R, w.part, s))) return rc;
    {
        const int64_t blk = (int64_t)hd * D;
        MultiCopy mc;
        mc.add(o.d_wq, w.dwcat, blk); mc.add(o.d_wk, w.dwcat + blk, blk); mc.add(o.d_wv, w.dwcat + 2 * blk, blk);
        mc.add(o.d_wg, w.dwcat + 3 * blk, (int64_t)heads * D);
        if (has_rv) mc.add(o.d_wm, w.dwcat + (size_t)(3 * hd + w.hp4) * D, (int64_t)heads * D);
        if ((rc = multi_copy(mc, s))) return rc;
    }
    if ((rc = gemm_dx(w.dproj, w.P, w.wcat, D, w.dxn, D, R, D, w.P, w.wt, s))) return rc;
    if ((rc = rmsnorm_bwd(x, w.dxn, prm.norm_w, w.tg, o.dx, R, D, RMS_EPS, s))) return rc;
    ColsumBatch cb;                                    // key gain, mix bias, norm gain: one launch
    cb.add(w.gpart, hd, g.groups, hd, o.d_gamma);
    if (has_rv) cb.add(w.dproj + 3 * hd + w.hp4, w.P, R, heads, o.d_bm);
    cb.add(w.tg, D, R, D, o.d_norm_w);
    return colsum_batch(cb, s);
}

struct XWs {
    float *qn, *cn, *wqg, *wkv, *projq, *projk, *dprojq, *dprojk, *d_o3, *o3, *dwqg, *dwkv, *tg, *dqn, *tgc, *dcn, *gpart, *part, *wt;
    size_t total; int Pq, Pk, hp4;
};
XWs x_ws(float* base, int Rq, int Rk, int G, int D, int Dc, int heads, int dh) {
    XWs w{};
    const int hd = heads * dh;
    w.hp4 = (heads + 3) / 4 * 4; w.Pq = hd + w.hp4; w.Pk = 2 * hd;
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += (n + 63) / 64 * 64; return p; };
    w.qn = take((size_t)Rq * D); w.cn = take((size_t)Rk * Dc); w.wqg = take((size_t)w.Pq * D); w.wkv = take((size_t)w.Pk * Dc);
    w.projq = take((size_t)Rq * w.Pq); w.projk = take((size_t)Rk * w.Pk); w.dprojq = take((size_t)Rq * w.Pq); w.dprojk = take((size_t)Rk * w.Pk);
    w.d_o3 = take((size_t)Rq * hd); w.o3 = take((size_t)Rq * hd); w.dwqg = take((size_t)w.Pq * D); w.dwkv = take((size_t)w.Pk * Dc);
    w.tg = take((size_t)Rq * D); w.dqn = take((size_t)Rq * D); w.tgc = take((size_t)Rk * Dc); w.dcn = take((size_t)Rk * Dc); w.gpart = take((size_t)G * hd);
    w.part = take(DW_PART_FLOATS);
    w.wt = take((size_t)(w.Pk > w.Pq ? w.Pk : w.Pq) * (D > Dc ? D : Dc));
    w.total = off;
    return w;
}

struct XParams { const float *norm_w, *norm_ctx_w, *wq, *wk, *wv, *wo, *wg, *gamma; };

int x_project(const XWs& w, const float* q_tokens, const float* ctx, const XParams& prm, int Rq, int Rk, int D, int Dc, int heads, int dh, hipStream_t s) {
    const int hd = heads * dh;
    int rc;
    {
        MultiCopy mc;
        mc.add(w.wqg, prm.wq, (int64_t)hd * D); mc.add(w.wqg + (size_t)hd * D, prm.wg, (int64_t)heads * D);
        mc.add(w.wqg + (size_t)(hd + heads) * D, nullptr, (int64_t)(w.Pq - hd - heads) * D);
        mc.add(w.wkv, prm.wk, (int64_t)hd * Dc); mc.add(w.wkv + (size_t)hd * Dc, prm.wv, (int64_t)hd * Dc);
        if ((rc = multi_copy(mc, s))) return rc;
    }
    if ((rc = rmsnorm_rows(q_tokens, D, prm.norm_w, w.qn, D, Rq, D, RMS_EPS, s))) return rc;
    if (prm.norm_ctx_w) { if ((rc = rmsnorm_rows(ctx, Dc, prm.norm_ctx_w, w.cn, Dc, Rk, Dc, RMS_EPS, s))) return rc; }
    else if ((rc = copy_rows(ctx, Dc, w.cn, Dc, Rk, Dc, s))) return rc;
    if ((rc = gemm_b(w.qn, D, w.wqg, D, w.projq, w.Pq, nullptr, Rq, w.Pq, D, 0, s))) return rc;
    return gemm_b(w.cn, Dc, w.wkv, Dc, w.projk, w.Pk, nullptr, Rk, w.Pk, Dc, 0, s);
}

int x_check(int G, int nq, int nk, int D, int Dc, int heads, int dh, const float* workspace, size_t workspace_bytes) {
    D4_REQUIRE(nq >= 1 && nq <= XA_N && nk >= 1 && nk <= XA_N, "cross attention block: %d queries / %d keys per group (max %d)", nq, nk, XA_N);
    D4_REQUIRE(dh == 16 || dh == 32 || dh == 64, "cross attention block: head dim %d (16, 32 or 64)", dh);
    D4_REQUIRE(D % 4 == 0 && Dc % 4 == 0 && ((uintptr_t)workspace % 256) == 0, "cross attention block: dims must be multiples of 4 and the workspace 256-byte aligned");
    D4_REQUIRE(workspace_bytes >= x_ws(nullptr, G * nq, G * nk, G, D, Dc, heads, dh).total * sizeof(float), "cross attention block: workspace too small");
    return 0;
}

}  // namespace

extern "C" {

size_t d4_cross_attn_workspace_bytes(int groups, int nq, int nk, int dim, int dim_ctx, int heads, int dim_head) {
    return x_ws(nullptr, groups * nq, groups * nk, groups, dim, dim_ctx, heads, dim_head).total * sizeof(float);
}

int d4_cross_attn_forward(const float* q_tokens, const float* ctx, const float* norm_w, const float* norm_ctx_w, const float* wq, const float* wk,
                          const float* wv, const float* wo, const float* w_gates, const float* k_gamma, int groups, int nq, int nk, int ctx_item_major,
                          int dim, int dim_ctx, int heads, int dim_head, float softclamp, float* y, float* workspace, size_t workspace_bytes, void* stream) {
    D4_REQUIRE(q_tokens && ctx && norm_w && wq && wk && wv && wo && w_gates && k_gamma && y && workspace, "d4_cross_attn_forward: null argument");
    int rc;
    if ((rc = x_check(groups, nq, nk, dim, dim_ctx, heads, dim_head, workspace, workspace_bytes))) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Rq = groups * nq, Rk = groups * nk, hd = heads * dim_head;
    if (Rq == 0) return 0;
    const XWs w = x_ws(workspace, Rq, Rk, groups, dim, dim_ctx, heads, dim_head);
    const XParams prm{norm_w, norm_ctx_w, wq, wk, wv, wo, w_gates, k_gamma};
    if ((rc = x_project(w, q_tokens, ctx, prm, Rq, Rk, dim, dim_ctx, heads, dim_head, s))) return rc;
    XAttnArgs a{w.projq, w.Pq, w.projk, w.Pk, k_gamma, nullptr, w.o3, nullptr, nullptr, nullptr, groups, nq, nk, heads, ctx_item_major, softclamp};
    if ((rc = xattn_core(a, dim_head, s))) return rc;
    return gemm_b(w.o3, hd, wo, hd, y, dim, nullptr, Rq, dim, hd, 0, s);
}

static int cross_attn_backward_impl(const float* q_tokens, const float* ctx, const float* dy, const float* norm_w, const float* norm_ctx_w, const float* wq,
                           const float* wk, const float* wv, const float* wo, const float* w_gates, const float* k_gamma, int groups, int nq, int nk,
                           int ctx_item_major, int dim, int dim_ctx, int heads, int dim_head, float softclamp,
                           float* d_q_tokens, float* d_ctx, float* d_norm_w, float* d_norm_ctx_w, float* d_wq, float* d_wk, float* d_wv, float* d_wo,
                           float* d_w_gates, float* d_k_gamma, float* workspace, size_t workspace_bytes, void* stream, bool reuse) {
    D4_REQUIRE(q_tokens && ctx && dy && norm_w && wq && wk && wv && wo && w_gates && k_gamma && workspace, "d4_cross_attn_backward: null argument");
    D4_REQUIRE(d_q_tokens && d_ctx && d_norm_w && d_wq && d_wk && d_wv && d_wo && d_w_gates && d_k_gamma && (!norm_ctx_w || d_norm_ctx_w),
               "d4_cross_attn_backward: null gradient output");
    int rc;
    if ((rc = x_check(groups, nq, nk, dim, dim_ctx, heads, dim_head, workspace, workspace_bytes))) return rc;
    D4_REQUIRE(groups >= 1, "d4_cross_attn_backward: no groups");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Rq = groups * nq, Rk = groups * nk, D = dim, Dc = dim_ctx, hd = heads * dim_head;
    const XWs w = x_ws(workspace, Rq, Rk, groups, D, Dc, heads, dim_head);
    const XParams prm{norm_w, norm_ctx_w, wq, wk, wv, wo, w_gates, k_gamma};
    if (!reuse && (rc = x_project(w, q_tokens, ctx, prm, Rq, Rk, D, Dc, heads, dim_head, s))) return rc;
    if ((rc = gemm_dx(dy, D, wo, hd, w.d_o3, hd, Rq, hd, D, w.wt, s))) return rc;
    XAttnArgs a{w.projq, w.Pq, w.projk, w.Pk, k_gamma, w.d_o3, w.o3, w.dprojq, w.dprojk, w.gpart, groups, nq, nk, heads, ctx_item_major, softclamp};
    if ((rc = xattn_core(a, dim_head, s))) return rc;
    if (w.hp4 > heads) {
        hipLaunchKernelGGL(zero_pad_cols_kernel, grid_for((int64_t)Rq * (w.hp4 - heads)), dim3(256), 0, s, w.dprojq, Rq, w.Pq, hd + heads, w.Pq);
        D4_LAUNCH_CHECK();
    }
    if ((rc = gemm_dw(dy, D, w.o3, hd, d_wo, hd, D, hd, Rq, w.part, s))) return rc;
    // query side
    if ((rc = gemm_dw(w.dprojq, w.Pq, w.qn, D, w.dwqg, D, w.Pq, D, Rq, w.part, s))) return rc;
    if ((rc = gemm_dx(w.dprojq, w.Pq, w.wqg, D, w.dqn, D, Rq, D, w.Pq, w.wt, s))) return rc;
    if ((rc = rmsnorm_bwd(q_tokens, w.dqn, norm_w, w.tg, d_q_tokens, Rq, D, RMS_EPS, s))) return rc;
    {
        ColsumBatch cb;                                // key gain and the query norm's gain: one launch
        cb.add(w.gpart, hd, groups, hd, d_k_gamma); cb.add(w.tg, D, Rq, D, d_norm_w);
        if ((rc = colsum_batch(cb, s))) return rc;
    }
    // context side
    if ((rc = gemm_dw(w.dprojk, w.Pk, w.cn, Dc, w.dwkv, Dc, w.Pk, Dc, Rk, w.part, s))) return rc;
    {   // the pieces of both concatenated weight gradients in one launch
        MultiCopy mc;
        mc.add(d_wq, w.dwqg, (int64_t)hd * D); mc.add(d_w_gates, w.dwqg + (size_t)hd * D, (int64_t)heads * D);
        mc.add(d_wk, w.dwkv, (int64_t)hd * Dc); mc.add(d_wv, w.dwkv + (size_t)hd * Dc, (int64_t)hd * Dc);
        if ((rc = multi_copy(mc, s))) return rc;
    }
    if (norm_ctx_w) {
        if ((rc = gemm_dx(w.dprojk, w.Pk, w.wkv, Dc, w.dcn, Dc, Rk, Dc, w.Pk, w.wt, s))) return rc;
        if ((rc = rmsnorm_bwd(ctx, w.dcn, norm_ctx_w, w.tgc, d_ctx, Rk, Dc, RMS_EPS, s))) return rc;
        return colsum(w.tgc, Dc, Rk, Dc, d_norm_ctx_w, s, w.part, DW_PART_FLOATS);
    }
    return gemm_dx(w.dprojk, w.Pk, w.wkv, Dc, d_ctx, Dc, Rk, Dc, w.Pk, w.wt, s);
}

#define D4_XBWD_PARAMS const float* q_tokens, const float* ctx, const float* dy, const float* norm_w, const float* norm_ctx_w, const float* wq,          \
                       const float* wk, const float* wv, const float* wo, const float* w_gates, const float* k_gamma, int groups, int nq, int nk,        \
                       int ctx_item_major, int dim, int dim_ctx, int heads, int dim_head, float softclamp, float* d_q_tokens, float* d_ctx,             \
                       float* d_norm_w, float* d_norm_ctx_w, float* d_wq, float* d_wk, float* d_wv, float* d_wo, float* d_w_gates, float* d_k_gamma,    \
                       float* workspace, size_t workspace_bytes, void* stream
#define D4_XBWD_ARGS q_tokens, ctx, dy, norm_w, norm_ctx_w, wq, wk, wv, wo, w_gates, k_gamma, groups, nq, nk, ctx_item_major, dim, dim_ctx, heads, dim_head, \
                     softclamp, d_q_tokens, d_ctx, d_norm_w, d_norm_ctx_w, d_wq, d_wk, d_wv, d_wo, d_w_gates, d_k_gamma, workspace, workspace_bytes, stream
int d4_cross_attn_backward(D4_XBWD_PARAMS) { return cross_attn_backward_impl(D4_XBWD_ARGS, false); }
int d4_cross_attn_backward_saved(D4_XBWD_PARAMS) { return cross_attn_backward_impl(D4_XBWD_ARGS, true); }

size_t d4_time_attn_workspace_bytes(int batch, int frames, int tokens, int dim, int heads, int dim_head) {
    return attn_ws(nullptr, batch * frames * tokens, batch * tokens, dim, heads, dim_head).total * sizeof(float);
}

int d4_space_attn_forward(const float* x, const float* residual_values, const float* norm_w, const float* wq, const float* wk, const float* wv,
                          const float* wo, const float* w_gates, const float* w_mix, const float* b_mix, const float* k_gamma,
                          int frames, int tokens, int dim, int heads, int dim_head, float softclamp, int num_special, int belief,
                          float* y, float* workspace, size_t workspace_bytes, void* stream) {
    const AttnParams prm{norm_w, wq, wk, wv, wo, w_gates, w_mix, b_mix, k_gamma};
    const AttnGeom g{frames, tokens, 1, tokens, 1, 0, num_special, nullptr};
    return attn_block_forward(x, residual_values, prm, frames * tokens, g, dim, heads, dim_head, softclamp, belief, y, workspace, workspace_bytes,
                              static_cast<hipStream_t>(stream));
}

int d4_space_attn_backward(const float* x, const float* residual_values, const float* dy, const float* norm_w, const float* wq, const float* wk,
                           const float* wv, const float* wo, const float* w_gates, const float* w_mix, const float* b_mix, const float* k_gamma,
                           int frames, int tokens, int dim, int heads, int dim_head, float softclamp, int num_special, int belief,
                           float* dx, float* d_residual_values, float* d_norm_w, float* d_wq, float* d_wk, float* d_wv, float* d_wo,
                           float* d_w_gates, float* d_w_mix, float* d_b_mix, float* d_k_gamma,
                           float* workspace, size_t workspace_bytes, void* stream) {
    const AttnParams prm{norm_w, wq, wk, wv, wo, w_gates, w_mix, b_mix, k_gamma};
    const AttnGeom g{frames, tokens, 1, tokens, 1, 0, num_special, nullptr};
    const AttnGrads o{dx, d_residual_values, d_norm_w, d_wq, d_wk, d_wv, d_wo, d_w_gates, d_w_mix, d_b_mix, d_k_gamma};
    return attn_block_backward(x, residual_values, dy, prm, frames * tokens, g, dim, heads, dim_head, softclamp, belief, o, workspace, workspace_bytes,
                               static_cast<hipStream_t>(stream));
}

int d4_space_attn_backward_saved(const float* x, const float* residual_values, const float* dy, const float* norm_w, const float* wq, const float* wk,
                           const float* wv, const float* wo, const float* w_gates, const float* w_mix, const float* b_mix, const float* k_gamma,
                           int frames, int tokens, int dim, int heads, int dim_head, float softclamp, int num_special, int belief,
                           float* dx, float* d_residual_values, float* d_norm_w, float* d_wq, float* d_wk, float* d_wv, float* d_wo,
                           float* d_w_gates, float* d_w_mix, float* d_b_mix, float* d_k_gamma,
                           float* workspace, size_t workspace_bytes, void* stream) {
    const AttnParams prm{norm_w, wq, wk, wv, wo, w_gates, w_mix, b_mix, k_gamma};
    const AttnGeom g{frames, tokens, 1, tokens, 1, 0, num_special, nullptr};
    const AttnGrads o{dx, d_residual_values, d_norm_w, d_wq, d_wk, d_wv, d_wo, d_w_gates, d_w_mix, d_b_mix, d_k_gamma};
    return attn_block_backward(x, residual_values, dy, prm, frames * tokens, g, dim, heads, dim_head, softclamp, belief, o, workspace, workspace_bytes,
                               static_cast<hipStream_t>(stream), true);
}

int d4_time_attn_forward(const float* x, const float* residual_values, const float* norm_w, const float* wq, const float* wk, const float* wv,
                         const float* wo, const float* w_gates, const float* w_mix, const float* b_mix, const float* k_gamma, const float* inv_freq,
                         int batch, int frames, int tokens, int dim, int heads, int dim_head, float softclamp, int belief,
                         float* y, float* workspace, size_t workspace_bytes, void* stream) {
    D4_REQUIRE(inv_freq, "d4_time_attn_forward: null rotary frequencies");
    const AttnParams prm{norm_w, wq, wk, wv, wo, w_gates, w_mix, b_mix, k_gamma};
    const AttnGeom g{batch * tokens, frames, tokens, (int64_t)frames * tokens, tokens, 1, 0, inv_freq};
    return attn_block_forward(x, residual_values, prm, batch * frames * tokens, g, dim, heads, dim_head, softclamp, belief, y, workspace, workspace_bytes,
                              static_cast<hipStream_t>(stream));
}

int d4_time_attn_backward(const float* x, const float* residual_values, const float* dy, const float* norm_w, const float* wq, const float* wk,
                          const float* wv, const float* wo, const float* w_gates, const float* w_mix, const float* b_mix, const float* k_gamma,
                          const float* inv_freq, int batch, int frames, int tokens, int dim, int heads, int dim_head, float softclamp, int belief,
                          float* dx, float* d_residual_values, float* d_norm_w, float* d_wq, float* d_wk, float* d_wv, float* d_wo,
                          float* d_w_gates, float* d_w_mix, float* d_b_mix, float* d_k_gamma,
                          float* workspace, size_t workspace_bytes, void* stream) {
    D4_REQUIRE(inv_freq, "d4_time_attn_backward: null rotary frequencies");
    const AttnParams prm{norm_w, wq, wk, wv, wo, w_gates, w_mix, b_mix, k_gamma};
    const AttnGeom g{batch * tokens, frames, tokens, (int64_t)frames * tokens, tokens, 1, 0, inv_freq};
    const AttnGrads o{dx, d_residual_values, d_norm_w, d_wq, d_wk, d_wv, d_wo, d_w_gates, d_w_mix, d_b_mix, d_k_gamma};
    return attn_block_backward(x, residual_values, dy, prm, batch * frames * tokens, g, dim, heads, dim_head, softclamp, belief, o, workspace,
                               workspace_bytes, static_cast<hipStream_t>(stream));
}

int d4_time_attn_backward_saved(const float* x, const float* residual_values, const float* dy, const float* norm_w, const float* wq, const float* wk,
                          const float* wv, const float* wo, const float* w_gates, const float* w_mix, const float* b_mix, const float* k_gamma,
                          const float* inv_freq, int batch, int frames, int tokens, int dim, int heads, int dim_head, float softclamp, int belief,
                          float* dx, float* d_residual_values, float* d_norm_w, float* d_wq, float* d_wk, float* d_wv, float* d_wo,
                          float* d_w_gates, float* d_w_mix, float* d_b_mix, float* d_k_gamma,
                          float* workspace, size_t workspace_bytes, void* stream) {
    D4_REQUIRE(inv_freq, "d4_time_attn_backward: null rotary frequencies");
    const AttnParams prm{norm_w, wq, wk, wv, wo, w_gates, w_mix, b_mix, k_gamma};
    const AttnGeom g{batch * tokens, frames, tokens, (int64_t)frames * tokens, tokens, 1, 0, inv_freq};
    const AttnGrads o{dx, d_residual_values, d_norm_w, d_wq, d_wk, d_wv, d_wo, d_w_gates, d_w_mix, d_b_mix, d_k_gamma};
    return attn_block_backward(x, residual_values, dy, prm, batch * frames * tokens, g, dim, heads, dim_head, softclamp, belief, o, workspace,
                               workspace_bytes, static_cast<hipStream_t>(stream), true);
}

}  // extern "C"
