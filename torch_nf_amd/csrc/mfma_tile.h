// MFMA formulation of one RealNVP coupling layer on a 16-sample tile (gfx950).
//
// Mapping (v_mfma_f32_16x16x4_f32, exact fp32 == k-ordered fmaf chain):
//   lane l = (s, q), s = l & 15 (sample inside the tile), q = l >> 4.
//   The MLP is evaluated TRANSPOSED: H^T = W^T . X^T, i.e. the weights are the A
//   operand (rows = output units), the activations are the B operand (cols =
//   samples).  The accumulator of a 16-unit output tile then lives as
//       acc[j] (j=0..3) = unit (4q + j) of sample s
//   which is exactly the B-operand layout of K-step j of the NEXT layer, provided
//   that layer's weight operand enumerates its K index in the same permuted order
//   (k = 4q + j at step j).  So the three-layer twin MLP chains accumulator ->
//   operand with no LDS round trip and no cross-lane traffic; the reduction order
//   inside a dot product is a permutation of the reference's, nothing else.
//   The sample features are loaded as float4: lane (s,q) holds features
//   16m + 4q + {0..3} (m = 16-feature tile index) of both halves, which is at
//   once the layer-0 B operand and the layout of the t/s outputs, so the
//   scale-shift happens in registers on the same lanes.
//   log|det J| = sum_f s_f : 8 in-lane adds + two cross-lane adds (q = 0..3).
//
// Hidden width U <= 16 is zero-padded to 16 (padded units see weight 0 / bias 0,
// tanh(0) = 0 exactly, and feed weight-0 columns).
#pragma once
#include <hip/hip_runtime.h>

namespace tnf {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// exp(x) = 2^(x*log2e) with the rounding error of the product fed back:
// t = rn(x*L), d = fma(x, L, -t) + x*L_lo  ->  2^t * (1 + d*ln2).  v_exp_f32 is ~1 ulp.
__device__ __forceinline__ float fast_exp(float x) {
    const float L_hi = 1.44269502162933349609375f;   // rn(log2(e))
    const float L_lo = 1.925963033500011e-08f;       // log2(e) - L_hi
    const float t = x * L_hi;
    const float d = __builtin_fmaf(x, L_hi, -t) + x * L_lo;
    const float e = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(e * d, 0.693147182464599609375f, e);
}

// tanh(x) = sign(x) * (1 - 2/(exp(2|x|) + 1)); abs error <= ~1.5e-7, exact 0 at 0,
// saturates cleanly (exp -> inf -> 2/inf = 0).
__device__ __forceinline__ float fast_tanh(float x) {
    const float ax = __builtin_fabsf(x);
    const float e = __builtin_amdgcn_exp2f(ax * 2.885390081777926815f);  // 2*log2(e)
    const float r = __builtin_amdgcn_rcpf(e + 1.0f);
    const float t = __builtin_fmaf(-2.0f, r, 1.0f);
    return __builtin_copysignf(t, x);
}

__device__ __forceinline__ f4 tanh4(f4 v) {
    f4 r;
    r[0] = fast_tanh(v[0]);
    r[1] = fast_tanh(v[1]);
    r[2] = fast_tanh(v[2]);
    r[3] = fast_tanh(v[3]);
    return r;
}

// Per-lane MFMA operands of one coupling layer (weights = A operands, biases =
// accumulator initial values).  H = half width (d_in = d_out), L = num_layers.
template <int H, int L>
struct LayerW {
    static constexpr int HT = (H + 15) / 16;
    static constexpr int LH = (L > 1) ? (L - 1) : 1;
    float w0[2][HT * 4];
    float wh[LH][2][4];
    float w2[2][HT][4];
    f4 b0[2];
    f4 bh[LH][2];
    f4 b2[2][HT];
};

// Gather the operands of lane (r = lane&15, q = lane>>4) from the reference's packed
// parameter row (bijectors.py:222-235): per MLP layer [W_t | W_s | b_t | b_s], W[in][out].
template <int H, int L>
__device__ __forceinline__ void load_layer_w(LayerW<H, L>& w, const float* __restrict__ p, int U,
                                             int lane) {
    constexpr int HT = LayerW<H, L>::HT;
    const int r = lane & 15, q = lane >> 4;
    const bool r_ok = r < U;
    // layer 0: H -> U
    {
        const float* wt = p;
        const float* ws = p + H * U;
        const float* bt = p + 2 * H * U;
        const float* bs = bt + U;
#pragma unroll
        for (int m = 0; m < HT; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = 16 * m + 4 * q + j;
                const bool ok = r_ok && f < H;
                w.w0[0][m * 4 + j] = ok ? wt[f * U + r] : 0.f;
                w.w0[1][m * 4 + j] = ok ? ws[f * U + r] : 0.f;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = 4 * q + j;
            w.b0[0][j] = u < U ? bt[u] : 0.f;
            w.b0[1][j] = u < U ? bs[u] : 0.f;
        }
        p = bs + U;
    }
    // hidden layers: U -> U
#pragma unroll
    for (int l = 0; l < L - 1; ++l) {
        const float* wt = p;
        const float* ws = p + U * U;
        const float* bt = p + 2 * U * U;
        const float* bs = bt + U;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * q + j;
            const bool ok = r_ok && k < U;
            w.wh[l][0][j] = ok ? wt[k * U + r] : 0.f;
            w.wh[l][1][j] = ok ? ws[k * U + r] : 0.f;
            w.bh[l][0][j] = k < U ? bt[k] : 0.f;
            w.bh[l][1][j] = k < U ? bs[k] : 0.f;
        }
        p = bs + U;
    }
    // output layer: U -> H
    {
        const float* wt = p;
        const float* ws = p + U * H;
        const float* bt = p + 2 * U * H;
        const float* bs = bt + H;
#pragma unroll
        for (int mo = 0; mo < HT; ++mo)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * q + j;
                const int o = 16 * mo + r;
                const bool ok = k < U && o < H;
                w.w2[0][mo][j] = ok ? wt[k * H + o] : 0.f;
                w.w2[1][mo][j] = ok ? ws[k * H + o] : 0.f;
                const int ob = 16 * mo + 4 * q + j;
                w.b2[0][mo][j] = ob < H ? bt[ob] : 0.f;
                w.b2[1][mo][j] = ob < H ? bs[ob] : 0.f;
            }
    }
}

// Operand providers: the same tile code runs with the layer's operands held in
// registers (per-layer kernel) or fetched from an LDS image (whole-flow kernel).
template <int H, int L>
struct RegOperands {
    static constexpr int HT = (H + 15) / 16;
    const LayerW<H, L>& w;
    __device__ __forceinline__ f4 w0(int net, int m) const {
        return f4{w.w0[net][m * 4 + 0], w.w0[net][m * 4 + 1], w.w0[net][m * 4 + 2], w.w0[net][m * 4 + 3]};
    }
    __device__ __forceinline__ f4 wh(int l, int net) const {
        return f4{w.wh[l][net][0], w.wh[l][net][1], w.wh[l][net][2], w.wh[l][net][3]};
    }
    __device__ __forceinline__ f4 w2(int net, int mo) const {
        return f4{w.w2[net][mo][0], w.w2[net][mo][1], w.w2[net][mo][2], w.w2[net][mo][3]};
    }
    __device__ __forceinline__ f4 b0(int net) const { return w.b0[net]; }
    __device__ __forceinline__ f4 bh(int l, int net) const { return w.bh[l][net]; }
    __device__ __forceinline__ f4 b2(int net, int mo) const { return w.b2[net][mo]; }
};

// LDS image of one layer: weight groups [g][lane][4] (one ds_read_b128 per group),
// then bias groups [g][q][4] (wave-broadcast reads).
template <int H, int L>
struct LdsLayerImage {
    static constexpr int HT = (H + 15) / 16;
    static constexpr int NWG = 4 * HT + 2 * (L - 1);       // weight groups
    static constexpr int NBG = 2 + 2 * (L - 1) + 2 * HT;   // bias groups
    static constexpr int FLOATS = NWG * 256 + NBG * 16;
    __device__ static constexpr int g_w0(int net, int m) { return net * HT + m; }
    __device__ static constexpr int g_wh(int l, int net) { return 2 * HT + 2 * l + net; }
    __device__ static constexpr int g_w2(int net, int mo) { return 2 * HT + 2 * (L - 1) + net * HT + mo; }
    __device__ static constexpr int b_b0(int net) { return net; }
    __device__ static constexpr int b_bh(int l, int net) { return 2 + 2 * l + net; }
    __device__ static constexpr int b_b2(int net, int mo) { return 2 + 2 * (L - 1) + net * HT + mo; }
};

template <int H, int L>
struct LdsOperands {
    typedef LdsLayerImage<H, L> Img;
    const float* wl;  // image base + lane*4
    const float* bl;  // image base + NWG*256 + q*4
    __device__ __forceinline__ LdsOperands(const float* img, int lane)
        : wl(img + lane * 4), bl(img + Img::NWG * 256 + (lane >> 4) * 4) {}
    __device__ __forceinline__ f4 w0(int net, int m) const {
        return *reinterpret_cast<const f4*>(wl + Img::g_w0(net, m) * 256);
    }
    __device__ __forceinline__ f4 wh(int l, int net) const {
        return *reinterpret_cast<const f4*>(wl + Img::g_wh(l, net) * 256);
    }
    __device__ __forceinline__ f4 w2(int net, int mo) const {
        return *reinterpret_cast<const f4*>(wl + Img::g_w2(net, mo) * 256);
    }
    __device__ __forceinline__ f4 b0(int net) const {
        return *reinterpret_cast<const f4*>(bl + Img::b_b0(net) * 16);
    }
    __device__ __forceinline__ f4 bh(int l, int net) const {
        return *reinterpret_cast<const f4*>(bl + Img::b_bh(l, net) * 16);
    }
    __device__ __forceinline__ f4 b2(int net, int mo) const {
        return *reinterpret_cast<const f4*>(bl + Img::b_b2(net, mo) * 16);
    }
};

// Write a lane's register operands into the LDS image (called once per layer per block).
template <int H, int L>
__device__ __forceinline__ void store_layer_image(float* img, const LayerW<H, L>& w, int lane) {
    typedef LdsLayerImage<H, L> Img;
    constexpr int HT = Img::HT;
    RegOperands<H, L> r{w};
    float* wl = img + lane * 4;
#pragma unroll
    for (int net = 0; net < 2; ++net) {
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            *reinterpret_cast<f4*>(wl + Img::g_w0(net, m) * 256) = r.w0(net, m);
            *reinterpret_cast<f4*>(wl + Img::g_w2(net, m) * 256) = r.w2(net, m);
        }
#pragma unroll
        for (int l = 0; l < L - 1; ++l) *reinterpret_cast<f4*>(wl + Img::g_wh(l, net) * 256) = r.wh(l, net);
    }
    if ((lane & 15) == 0) {
        float* bl = img + Img::NWG * 256 + (lane >> 4) * 4;
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            *reinterpret_cast<f4*>(bl + Img::b_b0(net) * 16) = r.b0(net);
#pragma unroll
            for (int l = 0; l < L - 1; ++l) *reinterpret_cast<f4*>(bl + Img::b_bh(l, net) * 16) = r.bh(l, net);
#pragma unroll
            for (int m = 0; m < HT; ++m) *reinterpret_cast<f4*>(bl + Img::b_b2(net, m) * 16) = r.b2(net, m);
        }
    }
}

// One coupling layer on NT tiles that share the layer's operands (independent MFMA
// chains -> the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 is covered).
// x = conditioner half (unchanged), y = transformed half (updated in place),
// ssum[t] += this lane's share of sum(s).
//   forward : y = t + y*exp(s)     (bijectors.py:172)
//   inverse : y = (y - t)/exp(s)   (bijectors.py:198), evaluated as (y - t)*exp(-s)
template <int H, int L, bool INV, int NT, class OP>
__device__ __forceinline__ void coupling_tile(const OP& op, const f4 (&x)[NT][(H + 15) / 16],
                                              f4 (&y)[NT][(H + 15) / 16], float (&ssum)[NT]) {
    constexpr int HT = (H + 15) / 16;
    f4 at[NT], as[NT];
    {
        const f4 bt = op.b0(0), bs = op.b0(1);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            at[t] = bt;
            as[t] = bs;
        }
    }
#pragma unroll
    for (int m = 0; m < HT; ++m) {
        const f4 wt = op.w0(0, m), ws = op.w0(1, m);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                at[t] = mfma4(wt[j], x[t][m][j], at[t]);
                as[t] = mfma4(ws[j], x[t][m][j], as[t]);
            }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        at[t] = tanh4(at[t]);
        as[t] = tanh4(as[t]);
    }
#pragma unroll
    for (int l = 0; l < L - 1; ++l) {
        const f4 wt = op.wh(l, 0), ws = op.wh(l, 1);
        const f4 bt = op.bh(l, 0), bs = op.bh(l, 1);
        f4 nt[NT], ns[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            nt[t] = bt;
            ns[t] = bs;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                nt[t] = mfma4(wt[j], at[t][j], nt[t]);
                ns[t] = mfma4(ws[j], as[t][j], ns[t]);
            }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            at[t] = tanh4(nt[t]);
            as[t] = tanh4(ns[t]);
        }
    }
#pragma unroll
    for (int mo = 0; mo < HT; ++mo) {
        const f4 wt = op.w2(0, mo), ws = op.w2(1, mo);
        const f4 bt = op.b2(0, mo), bs = op.b2(1, mo);
        f4 tt[NT], sv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            tt[t] = bt;
            sv[t] = bs;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                tt[t] = mfma4(wt[j], at[t][j], tt[t]);
                sv[t] = mfma4(ws[j], as[t][j], sv[t]);
            }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float s1 = sv[t][j];
                ssum[t] += s1;
                if (INV)
                    y[t][mo][j] = (y[t][mo][j] - tt[t][j]) * fast_exp(-s1);
                else
                    y[t][mo][j] = __builtin_fmaf(y[t][mo][j], fast_exp(s1), tt[t][j]);
            }
    }
}

// sum over the four q-lanes that share a sample (lanes s, s+16, s+32, s+48)
__device__ __forceinline__ float reduce_q(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

}  // namespace tnf
