// Beta policy head of the continuous actions (reference: ActionEmbedder.sample / log_probs / kl_div through
// discrete_continuous_embed_readout.Readout + BetaDist, D4:1172-1196, 1379-1389, 1442-1452, 1491-1494).
// The package is absent from the image; parameterisation, tempering and the injectable sampler follow the stand-in
// oracle/shim/discrete_continuous_embed_readout (PARITY UNPINNED against the real package):
//   alpha = link(raw0) + 1, beta = link(raw1) + 1 (unimodal: both >= 1); the link is a DESCRIPTOR (d4_config.continuous_beta_param),
//   softplus by default and exp as the other published choice — a checkpoint of the real package that turns out to use the other
//   link needs the constructor argument, not a kernel change.  sample ~ Beta(1 + (alpha-1)/T, 1 + (beta-1)/T) as Ga / (Ga + Gb)
//   with Marsaglia-Tsang gammas whose rejection rounds consume injected (normal, uniform) pairs.
#pragma once
#include "common.h"

namespace d4 {

constexpr int BETA_ROUNDS = 6;

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }      // F.softplus (beta 1, threshold 20)

// digamma / trigamma for x >= 1: shift to x >= 6 by the recurrences, then the asymptotic series
__device__ __forceinline__ float digammaf(float x) {
    float r = 0.f;
    while (x < 6.f) { r -= 1.f / x; x += 1.f; }
    const float i = 1.f / x, i2 = i * i;
    return r + logf(x) - 0.5f * i - i2 * (1.f / 12.f - i2 * (1.f / 120.f - i2 * (1.f / 252.f)));
}
__device__ __forceinline__ float trigammaf(float x) {
    float r = 0.f;
    while (x < 6.f) { r += 1.f / (x * x); x += 1.f; }
    const float i = 1.f / x, i2 = i * i;
    return r + i * (1.f + 0.5f * i + i2 * (1.f / 6.f - i2 * (1.f / 30.f - i2 * (1.f / 42.f))));
}

struct BetaAB { float a, b, da, db; };        // alpha, beta and d alpha / d raw0, d beta / d raw1
// kind: D4_BETA_SOFTPLUS_P1 (0)  alpha = softplus(raw) + 1, d alpha = sigmoid(raw);  D4_BETA_EXP_P1 (1)  alpha = exp(raw) + 1, d alpha = exp(raw)
__device__ __forceinline__ BetaAB beta_ab(float r0, float r1, int kind) {
    if (kind == 1) { const float e0 = expf(r0), e1 = expf(r1); return BetaAB{e0 + 1.f, e1 + 1.f, e0, e1}; }
    return BetaAB{softplusf(r0) + 1.f, softplusf(r1) + 1.f, sigmoidf(r0), sigmoidf(r1)};
}

__device__ __forceinline__ float lbetaf(float a, float b) { return lgammaf(a) + lgammaf(b) - lgammaf(a + b); }

__device__ __forceinline__ float beta_log_prob(float a, float b, float x) {
    return (a - 1.f) * logf(x) + (b - 1.f) * log1pf(-x) + lgammaf(a + b) - lgammaf(a) - lgammaf(b);
}

// one gamma(shape >= 1) draw; noise: rounds x (normal, uniform) at `stride` floats per round
__device__ __forceinline__ float gamma_from_noise(float shape, const float* noise) {
    const float d = shape - 1.f / 3.f;
    const float c = 1.f / sqrtf(9.f * d);
    float out = 0.f;
    bool done = false;
#pragma unroll
    for (int r = 0; r < BETA_ROUNDS; ++r) {
        const float x = noise[2 * r], u = noise[2 * r + 1];
        const float t = 1.f + c * x;
        const float v = t * t * t;
        const bool ok = v > 0.f && logf(fmaxf(u, 1e-30f)) < 0.5f * x * x + d - d * v + d * logf(fmaxf(v, 1e-30f));
        if (!done && (ok || r == BETA_ROUNDS - 1)) out = d * v;
        done = done || ok;
    }
    return fmaxf(out, 1e-30f);
}

// noise: [2 gammas][BETA_ROUNDS][2]
__device__ __forceinline__ float beta_sample(float a, float b, float temperature, const float* noise) {
    const float t = fmaxf(temperature, 1e-10f);
    const float at = 1.f + (a - 1.f) / t, bt = 1.f + (b - 1.f) / t;
    const float ga = gamma_from_noise(at, noise), gb = gamma_from_noise(bt, noise + 2 * BETA_ROUNDS);
    return ga / (ga + gb);
}

}  // namespace d4
