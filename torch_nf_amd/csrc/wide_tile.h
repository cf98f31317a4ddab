// Operand layout of the "wide" fp32-MFMA coupling kernels (coupling_wide.hip: forward / inverse, coupling_wide_bwd.hip:
// backward): tile counts, the lane-ordered LDS image of a layer's folded operands and the wave that builds it.
#pragma once
#include "mfma_tile.h"
#include "tnf_common.h"

namespace tnf {

struct WideLayout {
    int UT, HT, L;
    __host__ __device__ int nW0() const { return 2 * UT * HT; }
    __host__ __device__ int nWh() const { return 2 * UT * UT; }
    __host__ __device__ int nW2() const { return 2 * HT * UT; }
    __host__ __device__ int NWG() const { return nW0() + (L - 1) * nWh() + nW2(); }
    __host__ __device__ int NBG() const { return 2 * UT + (L - 1) * 2 * UT + 2 * HT; }
    __host__ __device__ int floats() const { return NWG() * 256 + NBG() * 16; }
    __host__ __device__ int g_w0(int net, int ut, int m) const { return (net * UT + ut) * HT + m; }
    __host__ __device__ int g_wh(int l, int net, int uo, int ui) const { return nW0() + l * nWh() + (net * UT + uo) * UT + ui; }
    __host__ __device__ int g_w2(int net, int mo, int ui) const { return nW0() + (L - 1) * nWh() + (net * HT + mo) * UT + ui; }
    __host__ __device__ int b_b0(int net, int ut) const { return net * UT + ut; }
    __host__ __device__ int b_bh(int l, int net, int uo) const { return 2 * UT + l * 2 * UT + net * UT + uo; }
    __host__ __device__ int b_b2(int net, int mo) const { return 2 * UT + (L - 1) * 2 * UT + net * HT + mo; }
};

// Build the folded operand image of one layer from the reference's packed parameter row
// (bijectors.py:222-235).  One full wave; works for LDS and global destinations.
__device__ inline void build_wide_image(float* img, const float* __restrict__ p, WideLayout wl, int H, int U, int lane) {
    const int r = lane & 15, q = lane >> 4;
    float* wdst = img + lane * 4;
    float* bdst = img + wl.NWG() * 256 + q * 4;
    const bool bias_lane = r == 0;
    // layer 0: H -> U, feeds a tanh: weights and biases scaled by c = 2 log2(e)
    {
        const float* w[2] = {p, p + H * U};
        const float* b[2] = {p + 2 * H * U, p + 2 * H * U + U};
        for (int net = 0; net < 2; ++net)
            for (int ut = 0; ut < wl.UT; ++ut) {
                const int u = 16 * ut + r;
                for (int m = 0; m < wl.HT; ++m) {
                    f4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = 16 * m + 4 * q + j;
                        v[j] = kTwoLog2e * ld_sel(w[net], f * U + u, f < H && u < U);
                    }
                    *reinterpret_cast<f4*>(wdst + wl.g_w0(net, ut, m) * 256) = v;
                }
                f4 bv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ub = 16 * ut + 4 * q + j;
                    bv[j] = kTwoLog2e * ld_sel(b[net], ub, ub < U);
                }
                if (bias_lane) *reinterpret_cast<f4*>(bdst + wl.b_b0(net, ut) * 16) = bv;
            }
        p += 2 * H * U + 2 * U;
    }
    // hidden layers: U -> U, consume r = (1 - tanh)/2, feed a tanh
    for (int l = 0; l < wl.L - 1; ++l) {
        const float* w[2] = {p, p + U * U};
        const float* b[2] = {p + 2 * U * U, p + 2 * U * U + U};
        for (int net = 0; net < 2; ++net)
            for (int uo = 0; uo < wl.UT; ++uo) {
                const int o = 16 * uo + r;
                float csum = 0.f;
                for (int ui = 0; ui < wl.UT; ++ui) {
                    f4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 16 * ui + 4 * q + j;
                        const float raw = ld_sel(w[net], k * U + o, k < U && o < U);
                        csum += raw;
                        v[j] = -2.f * kTwoLog2e * raw;
                    }
                    *reinterpret_cast<f4*>(wdst + wl.g_wh(l, net, uo, ui) * 256) = v;
                }
                csum = reduce_q(csum);  // column sum of W for output unit 16uo + r
                f4 bv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ob = 16 * uo + 4 * q + j;
                    const float cs = __shfl(csum, 4 * q + j);
                    bv[j] = ob < U ? kTwoLog2e * (ld_sel(b[net], ob, ob < U) + cs) : 0.f;
                }
                if (bias_lane) *reinterpret_cast<f4*>(bdst + wl.b_bh(l, net, uo) * 16) = bv;
            }
        p += 2 * U * U + 2 * U;
    }
    // output layer: U -> H, consumes r; t plain, s scaled by log2(e)
    {
        const float* w[2] = {p, p + U * H};
        const float* b[2] = {p + 2 * U * H, p + 2 * U * H + H};
        for (int net = 0; net < 2; ++net) {
            const float sc = net == 0 ? 1.f : kLog2e;
            for (int mo = 0; mo < wl.HT; ++mo) {
                const int o = 16 * mo + r;
                float csum = 0.f;
                for (int ui = 0; ui < wl.UT; ++ui) {
                    f4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 16 * ui + 4 * q + j;
                        const float raw = ld_sel(w[net], k * H + o, k < U && o < H);
                        csum += raw;
                        v[j] = -2.f * sc * raw;
                    }
                    *reinterpret_cast<f4*>(wdst + wl.g_w2(net, mo, ui) * 256) = v;
                }
                csum = reduce_q(csum);
                f4 bv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ob = 16 * mo + 4 * q + j;
                    const float cs = __shfl(csum, 4 * q + j);
                    bv[j] = ob < H ? sc * (ld_sel(b[net], ob, ob < H) + cs) : 0.f;
                }
                if (bias_lane) *reinterpret_cast<f4*>(bdst + wl.b_b2(net, mo) * 16) = bv;
            }
        }
    }
}

inline WideLayout wide_layout(int D, int L, int U) {
    WideLayout wl;
    wl.UT = (U + 15) / 16;
    wl.HT = (D / 2 + 15) / 16;
    wl.L = L;
    return wl;
}

}  // namespace tnf
