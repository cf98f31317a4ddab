// ToInterval (bijectors.py:429-557) evaluated in registers inside the flow kernels (float32, hardware
// transcendentals: v_exp / v_log / v_rcp).  The standalone kernels of support_kernels.hip use the libm
// functions and also cover float64; this header is the fused load / store stage of
// NormFlow(..., support_layer=ToInterval) when the rest of the flow is one kernel.
// Constants: 7 rows of DP floats in LDS -- tanh_flg, softplus_flg, tanh_m, tanh_c, softplus_m, softplus_c,
// log(tanh_m) -- DP >= D the row stride.
#pragma once
#include "mfma_tile.h"

namespace tnf {

__device__ __forceinline__ float sm_exp(float x) { return __builtin_amdgcn_exp2f(kLog2e * x); }
__device__ __forceinline__ float sm_log(float x) { return kLn2 * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float sm_tanh(float x) { return 1.f - 2.f * sig2(kTwoLog2e * x); }
// log(sigmoid(x)) = min(x, 0) - log(1 + exp(-|x|))
__device__ __forceinline__ float sm_logsigmoid(float x) { return fminf(x, 0.f) - sm_log(1.f + sm_exp(-fabsf(x))); }

// INV: x is a point of the constrained space; out = its pre-image, ld = the FORWARD log-det there (the
// reference's convention, :529-553).  !INV: forward map and its log-det (:509-527).
template <bool INV>
__device__ __forceinline__ void interval_fast(float x, const float* c, int DP, int d, float& out, float& ld) {
    const float eps = 1e-12f;
    const float tf = c[d], sf = c[DP + d];
    out = x;
    ld = 0.f;
    if (tf != 0.f) {
        const float tm = c[2 * DP + d], tc = c[3 * DP + d], ltm = c[6 * DP + d];
        float zi = x;
        if (INV) {
            const float u = (x - tc) * __builtin_amdgcn_rcpf(tm);
            zi = 0.5f * (sm_log(1.f + u + eps) - sm_log(1.f - u + eps));  // torch_atanh, :555-557
        }
        const float t = sm_tanh(zi);
        ld = ltm + sm_log(1.f - t * t + eps);
        out = INV ? zi : tm * t + tc;
    } else if (sf != 0.f) {
        const float sm = c[4 * DP + d], sc = c[5 * DP + d];
        if (INV) {
            const float zi = sm_log(sm_exp((x - sc) * sm) - 1.f + eps);  // softplus_m is +-1: division = product
            out = zi;
            ld = sm_logsigmoid(zi);
        } else {
            out = sm * (fmaxf(x, 0.f) + sm_log(1.f + sm_exp(-fabsf(x)))) + sc;
            ld = sm_logsigmoid(x);
        }
    }
}

}  // namespace tnf
