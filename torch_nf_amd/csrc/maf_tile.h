// Operand layout of the matrix-pipe MAF kernels (maf_mfma.hip: both directions; coupling_wide_bwd.hip: the two-pass
// backward for D > 32): tile counts, the lane-ordered LDS image of the masked, folded twin MLPs and the waves that build it.
#pragma once
#include "mfma_tile.h"
#include "tnf_common.h"

namespace tnf {

struct MafLayout {
    int UT, DT, L;
    __host__ __device__ int nW0() const { return 2 * UT * DT; }
    __host__ __device__ int nWh() const { return 2 * UT * UT; }
    __host__ __device__ int nW2() const { return 2 * DT * UT; }
    __host__ __device__ int NWG() const { return nW0() + (L - 1) * nWh() + nW2(); }
    __host__ __device__ int NBG() const { return (L - 1) * 2 * UT + 2 * DT; }
    __host__ __device__ int floats() const { return NWG() * 256 + NBG() * 16; }
    __host__ __device__ int g_w0(int net, int ut, int m) const { return (net * UT + ut) * DT + m; }
    __host__ __device__ int g_wh(int l, int net, int uo, int ui) const { return nW0() + l * nWh() + (net * UT + uo) * UT + ui; }
    __host__ __device__ int g_w2(int net, int mo, int ui) const { return nW0() + (L - 1) * nWh() + (net * DT + mo) * UT + ui; }
    __host__ __device__ int b_bh(int l, int net, int uo) const { return l * 2 * UT + net * UT + uo; }
    __host__ __device__ int b_b2(int net, int mo) const { return (L - 1) * 2 * UT + net * DT + mo; }
};

// packed parameters (bijectors.py:698-740): per layer [W_mu | W_alpha], W row-major [in][out]; masks: one
// matrix per layer of the same shape.  One full wave builds the image (LDS destination).
// The (layer, net, out tile) units are dealt round-robin to the workgroup's `nwaves` waves.
__device__ inline void build_maf_image(float* img, const float* __restrict__ p, const float* __restrict__ mk, MafLayout wl,
                                int D, int U, int lane, int wave, int nwaves, int bf) {
    const int r = lane & 15, q = lane >> 4;
    int unit = 0;
    float* wdst = img + lane * 4;
    float* bdst = img + wl.NWG() * 256 + q * 4;
    const bool bias_lane = r == 0;
    {   // layer 0: D -> U, feeds a tanh: scaled by c = 2 log2(e); no bias
        const float* w[2] = {p, p + D * U};
        for (int net = 0; net < 2; ++net)
            for (int ut = 0; ut < wl.UT; ++ut, ++unit) {
                if (unit % nwaves != wave) continue;
                const int u = 16 * ut + r;
                for (int m = 0; m < wl.DT; ++m) {
                    f4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = 16 * m + 4 * q + j;
                        const bool ok = f < D && u < U;
                        v[j] = kTwoLog2e * ld_sel(w[net], f * U + u, ok) * ld_sel(mk, f * U + u, ok);
                    }
                    *reinterpret_cast<f4*>(wdst + wl.g_w0(net, ut, m) * 256) = rbf16_4(v, bf);
                }
            }
        p += 2 * D * U;
        mk += D * U;
    }
    for (int l = 0; l < wl.L - 1; ++l) {  // hidden: U -> U, consume r = (1 - tanh)/2, feed a tanh
        const float* w[2] = {p, p + U * U};
        for (int net = 0; net < 2; ++net)
            for (int uo = 0; uo < wl.UT; ++uo, ++unit) {
                if (unit % nwaves != wave) continue;
                const int o = 16 * uo + r;
                float csum = 0.f;
                for (int ui = 0; ui < wl.UT; ++ui) {
                    f4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 16 * ui + 4 * q + j;
                        const bool ok = k < U && o < U;
                        const float raw = ld_sel(w[net], k * U + o, ok) * ld_sel(mk, k * U + o, ok);
                        csum += raw;
                        v[j] = -2.f * kTwoLog2e * raw;
                    }
                    *reinterpret_cast<f4*>(wdst + wl.g_wh(l, net, uo, ui) * 256) = rbf16_4(v, bf);
                }
                csum = reduce_q(csum);
                f4 bv;
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = kTwoLog2e * __shfl(csum, 4 * q + j);
                if (bias_lane) *reinterpret_cast<f4*>(bdst + wl.b_bh(l, net, uo) * 16) = bv;
            }
        p += 2 * U * U;
        mk += U * U;
    }
    {   // output: U -> D, consumes r; mu plain, alpha scaled by log2(e)
        const float* w[2] = {p, p + U * D};
        for (int net = 0; net < 2; ++net) {
            const float sc = net == 0 ? 1.f : kLog2e;
            for (int mo = 0; mo < wl.DT; ++mo, ++unit) {
                if (unit % nwaves != wave) continue;
                const int o = 16 * mo + r;
                float csum = 0.f;
                for (int ui = 0; ui < wl.UT; ++ui) {
                    f4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 16 * ui + 4 * q + j;
                        const bool ok = k < U && o < D;
                        const float raw = ld_sel(w[net], k * D + o, ok) * ld_sel(mk, k * D + o, ok);
                        csum += raw;
                        v[j] = -2.f * sc * raw;
                    }
                    *reinterpret_cast<f4*>(wdst + wl.g_w2(net, mo, ui) * 256) = rbf16_4(v, bf);
                }
                csum = reduce_q(csum);
                f4 bv;
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = sc * __shfl(csum, 4 * q + j);
                if (bias_lane) *reinterpret_cast<f4*>(bdst + wl.b_b2(net, mo) * 16) = bv;
            }
        }
    }
}

}  // namespace tnf
