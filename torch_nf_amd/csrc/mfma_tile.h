// MFMA formulation of one RealNVP coupling layer on 16-sample tiles (gfx950).
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
// Hidden width U <= 16 is zero-padded to 16 (padded units see weight 0 / bias 0).
//
// Activation folding (all done once, when the operands are built):
//   tanh(a) = 1 - 2 r,  r = 1/(2^(c a) + 1),  c = 2 log2(e).
//   * every layer that feeds a tanh has its weights and biases pre-multiplied by c,
//     so the accumulator is already the exp2 argument;
//   * the layer AFTER a tanh consumes r instead of tanh:  W.(1 - 2r) + b
//     = (b + colsum(W)) + (-2W).r, so "1 - 2r" never executes;
//   * the s-net's output layer is additionally scaled by log2(e): the kernel gets
//     s' = s log2(e), uses exp2(+-s') directly and sums s'; the caller multiplies the
//     reduced sum by ln 2.
//   Per tanh that leaves v_exp_f32, v_add_f32, v_rcp_f32; per transformed feature
//   v_sub/v_exp/v_mul (+ the log-det add).
#pragma once
#include <hip/hip_runtime.h>

#ifndef TNF_ABLATE
#define TNF_ABLATE 0
#endif

namespace tnf {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kTwoLog2e = 2.88539008177792681472f;
constexpr float kLn2 = 0.69314718055994530942f;

// r = 1/(2^a + 1)  (a = 2 log2(e) x  ->  tanh(x) = 1 - 2r).  2^a -> inf gives r = 0, -> 0 gives r = 1.
__device__ __forceinline__ float sig2(float a) {
#if TNF_ABLATE == 1
    return a;  // timing experiment only: no transcendental work
#else
    return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a) + 1.0f);
#endif
}

// The bf16 experiment on the fp32-MFMA kernels (TNF_OPT_OPERAND_PREC = 1): operands rounded to bf16 (RNE) and widened
// back.  The products of two such values are exact in fp32, so feeding them to the fp32 MFMA reproduces a bf16 MFMA
// with fp32 accumulation up to summation order -- the accuracy of a bf16 kernel, not its speed.  `on` is wave-uniform.
__device__ __forceinline__ float rbf16(float v, int on) {
    typedef __bf16 bfv2 __attribute__((ext_vector_type(2)));
    typedef float fv2 __attribute__((ext_vector_type(2)));
    if (!on) return v;
    const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(fv2{v, 0.f}, bfv2));
    return __builtin_bit_cast(float, u << 16);
}
__device__ __forceinline__ f4 rbf16_4(f4 v, int on) {
    return f4{rbf16(v[0], on), rbf16(v[1], on), rbf16(v[2], on), rbf16(v[3], on)};
}

__device__ __forceinline__ f4 sig2_4(f4 v) {
    f4 r;
    r[0] = sig2(v[0]);
    r[1] = sig2(v[1]);
    r[2] = sig2(v[2]);
    r[3] = sig2(v[3]);
    return r;
}

// sum over the four q-lanes that share a sample / an operand row (lanes r, r+16, r+32, r+48)
__device__ __forceinline__ float reduce_q(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// Unconditional load + select: a predicated `ok ? p[i] : 0` makes hipcc branch around every
// load and drain vmcnt per element (a chain of dependent L2 round trips); this form keeps
// all loads of the gather in flight together.
__device__ __forceinline__ float ld_sel(const float* __restrict__ p, int idx, bool ok) {
    const float v = p[ok ? idx : 0];
    return ok ? v : 0.f;
}

// Per-lane MFMA operands of one coupling layer (weights = A operands, biases =
// accumulator initial values), already folded as described above.
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
// Must be called by a full wave (the column sums use cross-lane shuffles).
template <int H, int L>
__device__ __forceinline__ void load_layer_w(LayerW<H, L>& w, const float* __restrict__ p, int U,
                                             int lane) {
    constexpr int HT = LayerW<H, L>::HT;
    const int r = lane & 15, q = lane >> 4;
    const bool r_ok = r < U;
    // layer 0: H -> U, feeds a tanh: scale by c
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
                w.w0[0][m * 4 + j] = kTwoLog2e * ld_sel(wt, f * U + r, ok);
                w.w0[1][m * 4 + j] = kTwoLog2e * ld_sel(ws, f * U + r, ok);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = 4 * q + j;
            w.b0[0][j] = kTwoLog2e * ld_sel(bt, u, u < U);
            w.b0[1][j] = kTwoLog2e * ld_sel(bs, u, u < U);
        }
        p = bs + U;
    }
    // hidden layers: U -> U, consume r, feed a tanh
#pragma unroll
    for (int l = 0; l < L - 1; ++l) {
        const float* wt = p;
        const float* ws = p + U * U;
        const float* bt = p + 2 * U * U;
        const float* bs = bt + U;
        float ct = 0.f, cs = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * q + j;
            const bool ok = r_ok && k < U;
            const float a = ld_sel(wt, k * U + r, ok);
            const float b = ld_sel(ws, k * U + r, ok);
            ct += a;
            cs += b;
            w.wh[l][0][j] = -2.f * kTwoLog2e * a;
            w.wh[l][1][j] = -2.f * kTwoLog2e * b;
        }
        ct = reduce_q(ct);  // column sum of W_t for output unit r
        cs = reduce_q(cs);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = 4 * q + j;
            const float st = __shfl(ct, u), ss = __shfl(cs, u);
            w.bh[l][0][j] = u < U ? kTwoLog2e * (ld_sel(bt, u, u < U) + st) : 0.f;
            w.bh[l][1][j] = u < U ? kTwoLog2e * (ld_sel(bs, u, u < U) + ss) : 0.f;
        }
        p = bs + U;
    }
    // output layer: U -> H, consumes r; t plain, s scaled by log2(e)
    {
        const float* wt = p;
        const float* ws = p + U * H;
        const float* bt = p + 2 * U * H;
        const float* bs = bt + H;
#pragma unroll
        for (int mo = 0; mo < HT; ++mo) {
            float ct = 0.f, cs = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * q + j;
                const int o = 16 * mo + r;
                const bool ok = k < U && o < H;
                const float a = ld_sel(wt, k * H + o, ok);
                const float b = ld_sel(ws, k * H + o, ok);
                ct += a;
                cs += b;
                w.w2[0][mo][j] = -2.f * a;
                w.w2[1][mo][j] = -2.f * kLog2e * b;
            }
            ct = reduce_q(ct);  // column sum for output feature 16mo + r
            cs = reduce_q(cs);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = 4 * q + j;  // row inside the 16-feature tile held by this lane
                const int ob = 16 * mo + rr;
                const float st = __shfl(ct, rr), ss = __shfl(cs, rr);
                w.b2[0][mo][j] = ob < H ? ld_sel(bt, ob, ob < H) + st : 0.f;
                w.b2[1][mo][j] = ob < H ? kLog2e * (ld_sel(bs, ob, ob < H) + ss) : 0.f;
            }
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
// ssum2[t] += this lane's share of sum(s)*log2(e)  (multiply the total by ln 2).
//   forward : y = t + y*exp(s)     (bijectors.py:172)
//   inverse : y = (y - t)/exp(s)   (bijectors.py:198), evaluated as (y - t)*2^(-s')
template <int H, int L, bool INV, int NT, class OP>
__device__ __forceinline__ void coupling_tile(const OP& op, const f4 (&x)[NT][(H + 15) / 16],
                                              f4 (&y)[NT][(H + 15) / 16], float (&ssum2)[NT]) {
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
        at[t] = sig2_4(at[t]);
        as[t] = sig2_4(as[t]);
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
            at[t] = sig2_4(nt[t]);
            as[t] = sig2_4(ns[t]);
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
                const float s2 = sv[t][j];
                ssum2[t] += s2;
#if TNF_ABLATE == 1
                if (INV)
                    y[t][mo][j] = (y[t][mo][j] - tt[t][j]) * s2;
#else
                if (INV)
                    y[t][mo][j] = (y[t][mo][j] - tt[t][j]) * __builtin_amdgcn_exp2f(-s2);
#endif
                else
                    y[t][mo][j] = __builtin_fmaf(y[t][mo][j], __builtin_amdgcn_exp2f(s2), tt[t][j]);
            }
    }
}

}  // namespace tnf
