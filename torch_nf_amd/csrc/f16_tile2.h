// Split-f16 coupling layer, second formulation (whole-flow inverse kernel, flow_fused2.hip).
//
// Same lane mapping and accumulator -> operand chaining as mfma_tile.h / f16_tile.h; what changes:
//
// 1. NO per-layer fold instructions.  The BatchNorm^-1 / Affine^-1 maps between coupling layers are per-feature
//    affine maps v -> A v + B.  Instead of applying them to both halves in front of every layer (16 FMAs per tile
//    and layer), every half carries a PENDING map (true value = Ap * register + Bp) that is absorbed into constants:
//      * conditioner half: the layer-0 weights are scaled per input feature and the bias takes W.B, so the MFMAs
//        consume the registers as they are;
//      * transformed half: y' = ((Ay y + By) - t) 2^-s' = fma(y, Ay, -(t - By)) 2^-s' -- By rides in the t-net's
//        output bias, Ay is one constant per feature: the update is ONE fma and ONE mul per feature.
//    After the last layer one fma per feature of the lower half turns the registers into true values.
//
// 2. Power-of-two operand normalisation (exact, free at run time).  A split-f16 operand is fp32-accurate only while
//    its magnitude stays inside the f16 range: absolute resolution 2^-24 (subnormal spacing), overflow at 65520.
//    Every weight operand is therefore scaled by a power of two chosen in the prologue so that its largest element
//    lies in [1, 2), and the activation operand it meets is scaled by the inverse power of two:
//      * sigmoid outputs: r' = r / S = rcp(fma(2^a, S, S)) -- an fma in place of the add;
//      * conditioner inputs: the PREVIOUS layer emits its transformed half already divided by 2^kappa (its t-net
//        output layer and Ay carry the factor), the pending map remembers it.
//    With that, weights of 1e-6 or 1e+3 and inputs up to 65520 / max|W| per conditioner feature product keep fp32
//    accuracy (tests/test_gpu_parity.py::test_operand_range).
//
// 3. Out-of-range inputs are DETECTED, not silently wrong: operands are split with round-to-nearest conversions
//    (v_cvt_pk_f16_f32), so |x'| >= 65520 becomes +-inf, its remainder -+inf, and every product of the three-term
//    contraction NaN; the NaN reaches the tile's log-det sum, the kernel sees it once per tile and re-runs that tile
//    with the first layer's contraction in exact fp32 MFMAs (weights rebuilt as hi + lo, inputs not split at all).
//
// FORWARD direction (sampling, FWD = true): the same machinery with the maps AFTER the layers.  Registers x~, y~ with true
//    values Px x~ and Py y~ (per-feature affine maps); the layer emits  y~' = fma(fma(y~, Ay, By), 2^s', t')  with
//    Ay = Py_A sigma, By = Py_B sigma (By cannot ride in t's bias: it meets the scale), t' = t sigma; the fold F behind the
//    layer stays pending on both halves: next layer's conditioner map (F_A / sigma, F_B) is absorbed by its layer-0
//    weights, the other half's F o Px becomes its Py.  The caller passes foldc = F_{c-1}, foldprev = F_{c-2}.
//
// 4. A compiler hazard found on the way (hipcc / ROCm 7.2, gfx950) and avoided by construction: a
//    v_mfma_f32_16x16x16_f16 whose SrcC is the result of a v_mfma_f32_16x16x32_f16 (or the other way round) gets too
//    few wait states -- the accumulator is read before the first MFMA has written it, and results change from run to
//    run.  (A two-MFMA form of the K = 16 contractions, [r_hi | r_lo] . [w_hi | w_hi] as one K = 32 product followed
//    by r_hi . w_lo as a K = 16 product, was 3.5 % faster and wrong for that reason.)  Every accumulation chain here
//    stays within ONE MFMA shape: K = 32 chains for the first layer at D = 64, K = 16 chains everywhere else.
#pragma once
#include "f16_tile.h"
#include "support_math.h"

#ifndef TNF2_ABL
#define TNF2_ABL 0
#endif
// Explicit pipe interleave of the layer body, all measured and none adopted -- DESIGN.md 3.11.2:
//   1 / 2: __builtin_amdgcn_iglp_opt(0 / 1);  3: one MFMA then five vector instructions, 24 NT times per layer, as
//   __builtin_amdgcn_sched_group_barrier groups;  4: one MFMA, two transcendentals, three other vector instructions.
#ifndef TNF2_SCHED
#define TNF2_SCHED 0
#endif

namespace tnf {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

// hi = rne_f16(v), lo = rne_f16(v - hi) (the remainder is exact in fp32).  |v| >= 65520: hi = +-inf, lo = -+inf.
__device__ __forceinline__ HiLo split2r(float v0, float v1) {
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{v0, v1}, h2));  // v_cvt_pk_f16_f32
#if TNF2_ABL == 4  // timing experiment only: no remainder
    return HiLo{hb, hb};
#endif
    // in place ("+v"), never into a fresh register: the inline-asm rule of f16_tile.h
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(v0) : "v"(hb));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v1) : "v"(hb));
    const unsigned lb = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{v0, v1}, h2));
    return HiLo{hb, lb};
}

// ---- reduced-precision experiment (BASELINE configs[4], "fp32 vs bf16 tolerance sweep"): PREC = 1 evaluates every
// contraction as ONE bf16 MFMA (operands rounded to bf16 once, fp32 accumulate) instead of the three split-f16
// products.  Not a parity path: tools/bf16_sweep.py measures what it costs in accuracy and buys in time.
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, bf2));  // v_cvt_pk_bf16_f32
}
__device__ __forceinline__ f4 mfma16b(u2 a, u2 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), c, 0, 0, 0);
}
__device__ __forceinline__ f4 mfma32b(u4 a, u4 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

// four values -> hi(4), lo(4)
__device__ __forceinline__ void split4r(f4 v, h4& hi, h4& lo) {
    const HiLo a = split2r(v[0], v[1]), b = split2r(v[2], v[3]);
    hi = __builtin_bit_cast(h4, u2{a.hi, b.hi});
    lo = __builtin_bit_cast(h4, u2{a.lo, b.lo});
}

// one K = 16 contraction  c + w . r  on split operands: w_hi.r_hi + w_hi.r_lo + w_lo.r_hi, one MFMA shape throughout
__device__ __forceinline__ f4 contract16(u4 wv, h4 rh, h4 rl, f4 c) {
    const h4 wh = __builtin_bit_cast(h4, u2{wv[0], wv[1]}), wl = __builtin_bit_cast(h4, u2{wv[2], wv[3]});
    return mfma16h(wl, rh, mfma16h(wh, rl, mfma16h(wh, rh, c)));
}

// r / S with r = 1/(2^a + 1):  2^a -> inf gives 0, -> 0 gives 1/S
__device__ __forceinline__ float sig2s(float a, float S) {
#if TNF2_ABL == 1  // timing experiment only: no transcendental work in the sigmoids
    return __builtin_fmaf(a, S, S);
#else
    return __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(a), S, S));
#endif
}
__device__ __forceinline__ f4 sig2s_4(f4 v, float S) {
    return f4{sig2s(v[0], S), sig2s(v[1], S), sig2s(v[2], S), sig2s(v[3], S)};
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
// exponent kappa with mx * 2^kappa in [1, 2), clamped; 0 for mx = 0 / non-finite
__device__ __forceinline__ int norm_exponent(float mx, int lo, int hi) {
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 0;
    int e;
    (void)__builtin_frexpf(mx, &e);  // mx = m 2^e, m in [0.5, 1)
    int k = 1 - e;
    return k < lo ? lo : (k > hi ? hi : k);
}
__device__ __forceinline__ float pow2i(int k) { return __builtin_ldexpf(1.f, k); }

// LDS image of one coupling layer (4-byte units)
template <int H, int L>
struct Img2 {
    static constexpr int HT = H / 16;
    static constexpr int N0 = (H == 32) ? 4 : 0;                            // layer-0 groups of the K = 32 form
    static constexpr int NK = ((H == 16) ? 2 : 0) + 2 * (L - 1) + 2 * HT;   // K = 16 contractions
    static constexpr int NBG = 2 + 2 * (L - 1) + 2 * HT;
    static constexpr int OFF_D = N0 * 256;          // [g][lane] 16 B: [w_hi(4) | w_lo(4)]
    static constexpr int OFF_B = OFF_D + NK * 256;  // [g][q][4] fp32 biases (accumulator initial values)
    static constexpr int OFF_S = OFF_B + NBG * 16;  // 8 floats: sigmoid scales S[stage][net]
    static constexpr int OFF_A = OFF_S + 8;         // H floats: Ay * sigma of the transformed half; H more: By * sigma (forward)
    static constexpr int FLOATS = OFF_A + 2 * H;
    __device__ static constexpr int k_l0(int net) { return net; }  // H == 16 only
    __device__ static constexpr int k_h(int l, int net) { return ((H == 16) ? 2 : 0) + 2 * l + net; }
    __device__ static constexpr int k_o(int net, int mo) { return ((H == 16) ? 2 : 0) + 2 * (L - 1) + net * HT + mo; }
    __device__ static constexpr int b_b0(int net) { return net; }
    __device__ static constexpr int b_bh(int l, int net) { return 2 + 2 * l + net; }
    __device__ static constexpr int b_b2(int net, int mo) { return 2 + 2 * (L - 1) + net * HT + mo; }
};
static_assert(Img2<32, 3>::FLOATS % 4 == 0 && Img2<16, 1>::FLOATS % 4 == 0, "images must stay 16-byte aligned");

// kappa of a layer: its layer-0 weights, with the activation folding (2 log2 e) and the fold A in front of the
// conditioner half, scaled by 2^kappa have their largest magnitude in [1, 2).  One wave.
template <int H>
__device__ __forceinline__ int layer_kappa(const float* __restrict__ p, int U, int lane, const float* foldc, int c) {
    constexpr int D = 2 * H;
    constexpr int HT = H / 16;
    const int r = lane & 15, q = lane >> 4;
    const int coff = (c & 1) ? H : 0;
    const float* wt = p;
    const float* ws = p + H * U;
    float mx = 0.f;
#pragma unroll
    for (int m = 0; m < HT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = 16 * m + 4 * q + j;
            const bool ok = r < U;
            const float a = foldc[coff + f];
            mx = fmaxf(mx, fabsf(kTwoLog2e * ld_sel(wt, f * U + r, ok) * a));
            mx = fmaxf(mx, fabsf(kTwoLog2e * ld_sel(ws, f * U + r, ok) * a));
        }
    (void)D;
    return norm_exponent(wave_max(mx), -30, 30);
}

// K = 16 weight group: four fp32 values -> [hi(4) | lo(4)]  (PREC = 1: [bf16(4) | 0])
template <int PREC>
__device__ __forceinline__ void store_k16(u4* gd, const float (&v)[4]) {
    if constexpr (PREC == 1) {
        *gd = u4{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), 0u, 0u};
    } else {
        const HiLo a = split2r(v[0], v[1]), b = split2r(v[2], v[3]);
        *gd = u4{a.hi, b.hi, a.lo, b.lo};
    }
}

// One wave builds the image of the coupling layer c (walk position k = 2S-1-c of the inverse pass).
//   foldc    [A_c (D) | B_c (D)]: the map in front of layer c (BatchNorm^-1, Affine^-1), both halves
//   foldprev the same of layer c+1 (the layer walked just before), NULL for the first layer walked
//   sc_in    2^kappa_k: the conditioner registers hold (true value before foldc) / sc_in
//   sc_prev  2^kappa_{k-1} (ignored when foldprev == NULL)
//   sig_next 2^-kappa_{k+1}: factor the transformed half is emitted with (1 for the last layer walked)
template <int H, int L, int PREC = 0, bool FWD = false>
__device__ __forceinline__ void build_image2(float* img, const float* __restrict__ p, int U, int lane, const float* foldc,
                                             const float* foldprev, int c, float sc_in, float sc_prev, float sig_next) {
    typedef Img2<H, L> I;
    constexpr int D = 2 * H;
    constexpr int HT = I::HT;
    LayerW<H, L> w;
    load_layer_w<H, L>(w, p, U, lane);
    const int q = lane >> 4;
    const int coff = (c & 1) ? H : 0, toff = (c & 1) ? 0 : H;

    // ---- layer 0: absorb A_c x + B_c (x = sc_in * register) ----
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        float pb = 0.f;
#pragma unroll
        for (int m = 0; m < HT; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = 16 * m + 4 * q + j;
                pb = __builtin_fmaf(w.w0[net][m * 4 + j], foldc[D + coff + f], pb);
                w.w0[net][m * 4 + j] *= foldc[coff + f] * sc_in;
            }
        pb = reduce_q(pb);  // sum over all conditioner features for unit (lane & 15)
#pragma unroll
        for (int j = 0; j < 4; ++j) w.b0[net][j] += __shfl(pb, 4 * q + j);
    }
    // ---- sigmoid scales: S[stage][net] normalises the weights that consume that stage's output ----
    float Sst[L][2];
#pragma unroll
    for (int l = 0; l < L - 1; ++l)
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            float mx = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fabsf(w.wh[l][net][j]));
            const float S = pow2i(norm_exponent(wave_max(mx), -15, 40));
            Sst[l][net] = S;
#pragma unroll
            for (int j = 0; j < 4; ++j) w.wh[l][net][j] *= S;
        }
    // output layer: the t-net also carries sig_next and (through its bias) the pending B of the transformed half
    f4 ays[HT], bys[HT];
#pragma unroll
    for (int mo = 0; mo < HT; ++mo)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = 16 * mo + 4 * q + j;  // bias / accumulator layout: feature held by this lane
            const float Ac = foldc[toff + f], Bc = foldc[D + toff + f];
            const float Ap = foldprev ? foldprev[toff + f] * sc_prev : 1.f;
            const float Bp = foldprev ? foldprev[D + toff + f] : 0.f;
            ays[mo][j] = Ac * Ap * sig_next;
            bys[mo][j] = __builtin_fmaf(Ac, Bp, Bc) * sig_next;
            if constexpr (FWD) w.b2[0][mo][j] *= sig_next;
            else w.b2[0][mo][j] = (w.b2[0][mo][j] - __builtin_fmaf(Ac, Bp, Bc)) * sig_next;
        }
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        float mx = 0.f;
#pragma unroll
        for (int mo = 0; mo < HT; ++mo)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (net == 0) w.w2[0][mo][j] *= sig_next;
                mx = fmaxf(mx, fabsf(w.w2[net][mo][j]));
            }
        const float S = pow2i(norm_exponent(wave_max(mx), -15, 40));
        Sst[L - 1][net] = S;
#pragma unroll
        for (int mo = 0; mo < HT; ++mo)
#pragma unroll
            for (int j = 0; j < 4; ++j) w.w2[net][mo][j] *= S;
    }

    // ---- write the image ----
    u4* g0 = reinterpret_cast<u4*>(img) + lane;
    u4* gd = reinterpret_cast<u4*>(img + I::OFF_D) + lane;
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        if constexpr (H == 32) {
            u4 hi, lo;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                if constexpr (PREC == 1) {
                    hi[pp] = pk_bf16(w.w0[net][2 * pp], w.w0[net][2 * pp + 1]);
                    lo[pp] = 0u;
                } else {
                    const HiLo s = split2r(w.w0[net][2 * pp], w.w0[net][2 * pp + 1]);
                    hi[pp] = s.hi;
                    lo[pp] = s.lo;
                }
            }
            g0[(2 * net) * 64] = hi;
            g0[(2 * net + 1) * 64] = lo;
        } else {
            const float v[4] = {w.w0[net][0], w.w0[net][1], w.w0[net][2], w.w0[net][3]};
            store_k16<PREC>(gd + I::k_l0(net) * 64, v);
        }
#pragma unroll
        for (int l = 0; l < L - 1; ++l) {
            const float v[4] = {w.wh[l][net][0], w.wh[l][net][1], w.wh[l][net][2], w.wh[l][net][3]};
            store_k16<PREC>(gd + I::k_h(l, net) * 64, v);
        }
#pragma unroll
        for (int mo = 0; mo < HT; ++mo) {
            const float v[4] = {w.w2[net][mo][0], w.w2[net][mo][1], w.w2[net][mo][2], w.w2[net][mo][3]};
            store_k16<PREC>(gd + I::k_o(net, mo) * 64, v);
        }
    }
    if ((lane & 15) == 0) {
        float* bl = img + I::OFF_B + q * 4;
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            *reinterpret_cast<f4*>(bl + I::b_b0(net) * 16) = w.b0[net];
#pragma unroll
            for (int l = 0; l < L - 1; ++l) *reinterpret_cast<f4*>(bl + I::b_bh(l, net) * 16) = w.bh[l][net];
#pragma unroll
            for (int mo = 0; mo < HT; ++mo) *reinterpret_cast<f4*>(bl + I::b_b2(net, mo) * 16) = w.b2[net][mo];
        }
#pragma unroll
        for (int mo = 0; mo < HT; ++mo) {
            *reinterpret_cast<f4*>(img + I::OFF_A + 16 * mo + 4 * q) = ays[mo];
            *reinterpret_cast<f4*>(img + I::OFF_A + H + 16 * mo + 4 * q) = bys[mo];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            img[I::OFF_S + 2 * l] = l < L ? Sst[l < L ? l : 0][0] : 1.f;
            img[I::OFF_S + 2 * l + 1] = l < L ? Sst[l < L ? l : 0][1] : 1.f;
        }
    }
}

// One coupling layer (inverse direction) on NT tiles.  x: conditioner registers (unchanged), y: transformed half,
// ssum2[t] += this lane's share of sum(s) log2(e).  SLOW: layer 0 in exact fp32 MFMAs (out-of-range inputs).
// the bf16 experiment: same image layout (the bf16 weights sit in the hi slots), one MFMA per contraction
template <int H, int L, int NT>
__device__ __forceinline__ void coupling_tile2_bf16(const float* img, int lane, const f4 (&x)[NT][H / 16], f4 (&y)[NT][H / 16],
                                                    float (&ssum2)[NT]) {
    typedef Img2<H, L> I;
    constexpr int HT = I::HT;
    const u4* g0 = reinterpret_cast<const u4*>(img) + lane;
    const u4* gd = reinterpret_cast<const u4*>(img + I::OFF_D) + lane;
    const float* bl = img + I::OFF_B + (lane >> 4) * 4;
    auto bias = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(bl + g * 16); };
    auto w16 = [&](int g) -> u2 { const u4 v = gd[g * 64]; return u2{v[0], v[1]}; };
    const f4 Sa = *reinterpret_cast<const f4*>(img + I::OFF_S);
    f4 Sb = {1.f, 1.f, 1.f, 1.f};
    if constexpr (L == 3) Sb = *reinterpret_cast<const f4*>(img + I::OFF_S + 4);
    auto Sc = [&](int stage, int net) -> float { return (2 * stage + net) < 4 ? Sa[2 * stage + net] : Sb[2 * stage + net - 4]; };
    auto b4 = [&](f4 v) -> u2 { return u2{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3])}; };

    f4 acc[NT][2];
    if constexpr (H == 32) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const u2 a = b4(x[t][0]), b = b4(x[t][1]);
            const u4 xb = u4{a[0], a[1], b[0], b[1]};
#pragma unroll
            for (int net = 0; net < 2; ++net) acc[t][net] = mfma32b(g0[(2 * net) * 64], xb, bias(I::b_b0(net)));
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const u2 xb = b4(x[t][0]);
#pragma unroll
            for (int net = 0; net < 2; ++net) acc[t][net] = mfma16b(w16(I::k_l0(net)), xb, bias(I::b_b0(net)));
        }
    }
    u2 rb[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int net = 0; net < 2; ++net) rb[t][net] = b4(sig2s_4(acc[t][net], Sc(0, net)));
#pragma unroll
    for (int l = 0; l < L - 1; ++l) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int net = 0; net < 2; ++net) {
                acc[t][net] = mfma16b(w16(I::k_h(l, net)), rb[t][net], bias(I::b_bh(l, net)));
                rb[t][net] = b4(sig2s_4(acc[t][net], Sc(l + 1, net)));
            }
    }
#pragma unroll
    for (int mo = 0; mo < HT; ++mo) {
        const f4 ay = *reinterpret_cast<const f4*>(img + I::OFF_A + 16 * mo + 4 * (lane >> 4));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f4 tt = mfma16b(w16(I::k_o(0, mo)), rb[t][0], bias(I::b_b2(0, mo)));
            const f4 sv = mfma16b(w16(I::k_o(1, mo)), rb[t][1], bias(I::b_b2(1, mo)));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ssum2[t] += sv[j];
                y[t][mo][j] = __builtin_fmaf(y[t][mo][j], ay[j], -tt[j]) * __builtin_amdgcn_exp2f(-sv[j]);
            }
        }
    }
}

template <int H, int L, int NT, bool SLOW, int PREC = 0, bool FWD = false>
__device__ __forceinline__ void coupling_tile2(const float* img, int lane, const f4 (&x)[NT][H / 16], f4 (&y)[NT][H / 16],
                                               float (&ssum2)[NT]) {
    static_assert(!(FWD && PREC == 1), "the bf16 experiment covers the inverse direction");
    if constexpr (PREC == 1) {
        coupling_tile2_bf16<H, L, NT>(img, lane, x, y, ssum2);
        return;
    }
    typedef Img2<H, L> I;
    constexpr int HT = I::HT;
#if TNF2_SCHED == 3 || TNF2_SCHED == 4
    __builtin_amdgcn_sched_barrier(0);  // one scheduling region per layer (without it the group solver ran > 30 min)
#elif TNF2_SCHED == 1
    __builtin_amdgcn_iglp_opt(0);
#elif TNF2_SCHED == 2
    __builtin_amdgcn_iglp_opt(1);
#endif
    const u4* g0 = reinterpret_cast<const u4*>(img) + lane;
    const u4* gd = reinterpret_cast<const u4*>(img + I::OFF_D) + lane;
    const float* bl = img + I::OFF_B + (lane >> 4) * 4;
    auto bias = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(bl + g * 16); };
    const f4 Sa = *reinterpret_cast<const f4*>(img + I::OFF_S);
    f4 Sb = {1.f, 1.f, 1.f, 1.f};
    if constexpr (L == 3) Sb = *reinterpret_cast<const f4*>(img + I::OFF_S + 4);
    auto Sc = [&](int stage, int net) -> float { return (2 * stage + net) < 4 ? Sa[2 * stage + net] : Sb[2 * stage + net - 4]; };

    f4 acc[NT][2];
    // ---- layer 0 ----
    if constexpr (!SLOW) {
        if constexpr (H == 32) {
            h8 xh[NT], xl[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const HiLo a = split2r(x[t][0][0], x[t][0][1]), b = split2r(x[t][0][2], x[t][0][3]);
                const HiLo c = split2r(x[t][1][0], x[t][1][1]), d = split2r(x[t][1][2], x[t][1][3]);
                xh[t] = __builtin_bit_cast(h8, u4{a.hi, b.hi, c.hi, d.hi});
                xl[t] = __builtin_bit_cast(h8, u4{a.lo, b.lo, c.lo, d.lo});
            }
#pragma unroll
            for (int net = 0; net < 2; ++net) {
                const h8 wh = __builtin_bit_cast(h8, g0[(2 * net) * 64]);
                const h8 wl = __builtin_bit_cast(h8, g0[(2 * net + 1) * 64]);
                const f4 b0 = bias(I::b_b0(net));
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t][net] = mfma32h(wh, xh[t], b0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t][net] = mfma32h(wh, xl[t], acc[t][net]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t][net] = mfma32h(wl, xh[t], acc[t][net]);
            }
        } else {
            h4 xh[NT], xl[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) split4r(x[t][0], xh[t], xl[t]);
#pragma unroll
            for (int net = 0; net < 2; ++net) {
                const u4 wv = gd[I::k_l0(net) * 64];
                const f4 b0 = bias(I::b_b0(net));
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t][net] = contract16(wv, xh[t], xl[t], b0);
            }
        }
    } else {
        // exact path: weights rebuilt as hi + lo (what the split path uses anyway), inputs as they are;
        // K-step (m, j) of v_mfma_f32_16x16x4_f32 contracts features 16m + 4q + j, q = 0..3
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            float wv[H / 4];
            if constexpr (H == 32) {
                const h8 wh = __builtin_bit_cast(h8, g0[(2 * net) * 64]);
                const h8 wl = __builtin_bit_cast(h8, g0[(2 * net + 1) * 64]);
#pragma unroll
                for (int i = 0; i < 8; ++i) wv[i] = (float)wh[i] + (float)wl[i];
            } else {
                const h8 wd = __builtin_bit_cast(h8, gd[I::k_l0(net) * 64]);  // [w_hi(4) | w_lo(4)]
#pragma unroll
                for (int i = 0; i < 4; ++i) wv[i] = (float)wd[i] + (float)wd[4 + i];
            }
            const f4 b0 = bias(I::b_b0(net));
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][net] = b0;
#pragma unroll
            for (int i = 0; i < H / 4; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t][net] = mfma4(wv[i], x[t][i >> 2][i & 3], acc[t][net]);
        }
    }
    // ---- sigmoid of stage 0, split ----
    h4 rh[NT][2], rl[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int net = 0; net < 2; ++net) split4r(sig2s_4(acc[t][net], Sc(0, net)), rh[t][net], rl[t][net]);
    // ---- hidden layers ----
#pragma unroll
    for (int l = 0; l < L - 1; ++l) {
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            const u4 wv = gd[I::k_h(l, net) * 64];
            const f4 bh = bias(I::b_bh(l, net));
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][net] = contract16(wv, rh[t][net], rl[t][net], bh);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int net = 0; net < 2; ++net) split4r(sig2s_4(acc[t][net], Sc(l + 1, net)), rh[t][net], rl[t][net]);
    }
    // ---- output layer and the update of the transformed half ----
#pragma unroll
    for (int mo = 0; mo < HT; ++mo) {
        f4 tt[NT], sv[NT];
        {
            const u4 wv = gd[I::k_o(0, mo) * 64];
            const f4 b2 = bias(I::b_b2(0, mo));
#pragma unroll
            for (int t = 0; t < NT; ++t) tt[t] = contract16(wv, rh[t][0], rl[t][0], b2);
        }
        {
            const u4 wv = gd[I::k_o(1, mo) * 64];
            const f4 b2 = bias(I::b_b2(1, mo));
#pragma unroll
            for (int t = 0; t < NT; ++t) sv[t] = contract16(wv, rh[t][1], rl[t][1], b2);
        }
        const f4 ay = *reinterpret_cast<const f4*>(img + I::OFF_A + 16 * mo + 4 * (lane >> 4));
        f4 by = {0.f, 0.f, 0.f, 0.f};
        if constexpr (FWD) by = *reinterpret_cast<const f4*>(img + I::OFF_A + H + 16 * mo + 4 * (lane >> 4));
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float s2 = sv[t][j];
#if TNF2_ABL == 3  // timing experiment only: one log-det add per tile instead of eight
                if (j == 0 && mo == 0) ssum2[t] += s2;
#else
                ssum2[t] += s2;
#endif
                if constexpr (FWD)
                    y[t][mo][j] = __builtin_fmaf(__builtin_fmaf(y[t][mo][j], ay[j], by[j]), __builtin_amdgcn_exp2f(s2), tt[t][j]);
                else
                    y[t][mo][j] = __builtin_fmaf(y[t][mo][j], ay[j], -tt[t][j]) * __builtin_amdgcn_exp2f(-s2);
            }
    }
#if TNF2_SCHED == 3 || TNF2_SCHED == 4
    // the layer's MFMAs (3 per contraction: 2 x 3 first-layer + (2 (L - 1) + 2 HT) x 3 K = 16 contractions per tile), each
    // followed by its share of the vector work
    constexpr int NMFMA = NT * 3 * (2 + 2 * (L - 1) + 2 * HT);
#pragma unroll
    for (int i = 0; i < NMFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
#if TNF2_SCHED == 3
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);  // VALU
#else
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);  // transcendental
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // VALU
#endif
    }
#endif
}

}  // namespace tnf
