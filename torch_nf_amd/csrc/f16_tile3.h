// Split-f16 coupling layer on 32-sample groups, v_mfma_f32_32x32x16_f16 (whole-flow inverse kernel, flow_fused3.hip).
//
// Same arithmetic as f16_tile2.h (pending per-feature maps, power-of-two operand normalisation, round-to-nearest
// splits with detected overflow, one MFMA shape per accumulation chain); what changes is the tile shape.  The
// whole-flow kernel is bound by the SIMD's vector-instruction issue (profiles/README.md, r02 PMC: VALU active 80 % of
// the kernel, the matrix pipe 36 %), and every MFMA holds that issue port for 8 cycles whatever its shape
// (MI355X_MICROARCH.md, "vector-instruction ISSUE cost").  A 16x16 tile spends 24 MFMAs per 16 samples and layer; here
// one MFMA covers BOTH conditioner nets (32 accumulator rows = 16 t-net + 16 s-net units) and 32 samples:
//
//   lane l = (s, h): s = l & 31 the sample of the group, h = l >> 5.
//   Every per-sample vector of 32 entries (conditioner features, hidden units of both nets, outputs) lives in 16
//   registers per lane in the accumulator layout of the 32x32 MFMA:  register i  <->  entry  R(i, h) = 8 (i >> 2) + 4 h + (i & 3).
//   Registers 8j .. 8j+7, converted pairwise to f16, ARE the B operand of k-step j of the next 32x32x16 MFMA
//   (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"); the weights are the A operand
//   (rows = output units), stored with their k index in that same permuted order, so layer chains to layer with no
//   LDS round trip and no cross-lane traffic.
//
//   layer 0   rows = [t-net units | s-net units], K = H conditioner features           H/16 k-steps x 3 split products
//   hidden    block diagonal: k-step 0 consumes the t-net's units, k-step 1 the s-net's  2 x 3
//   output    H = 32: t (32 feature rows, K = t-net units) and s (K = s-net units)        3 + 3
//             H = 16: rows = [t features | s features], block diagonal like the hidden    2 x 3
//   = 18 (15) MFMAs per 32 samples and layer at D = 64 (32), against 48.  Half of the hidden / output products multiply
//   zeros; the matrix pipe has the room.
//
// A sample's 16 registers of a half hold its features R(i, h), so a lane loads / stores four float4 per half
// (features 8g + 4h .. +3, g = 0..3): two lanes cover 32 contiguous bytes of a row per instruction.  That is a poorer
// global access shape than the 16x16 tile's (64 B), which is why the memory-bound per-layer chain keeps f16_tile2.h.
#pragma once
#include "f16_tile2.h"

namespace tnf {

typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f16v mfma32x32h(h8 a, h8 b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f16v mfma32x32f(float a, float b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// entry held by register i of lane half h
__device__ __forceinline__ constexpr int r_entry(int i, int h) { return 8 * (i >> 2) + 4 * h + (i & 3); }

// LDS image of one coupling layer (4-byte units).  A-operand groups: [g][lane] 16 B = 8 f16 (k-slots 0..7 of the lane).
template <int H, int L>
struct Img3 {
    static constexpr int KS0 = H / 16;                       // k-steps of layer 0
    static constexpr int NA = 2 * KS0 + 4 * (L - 1) + 4;     // groups: {hi, lo} per k-step (H = 32 output: per net)
    static constexpr int NOB = (H == 32) ? 64 : 32;          // output bias entries
    static constexpr int OFF_B = NA * 256;                   // fp32 accumulator initial values, natural row order
    static constexpr int NB = 32 + 32 * (L - 1) + NOB;
    static constexpr int OFF_S = OFF_B + NB;                 // 8 floats: sigmoid scales S[stage][net]
    static constexpr int OFF_A = OFF_S + 8;                  // H floats: Ay * sigma of the transformed half
    static constexpr int FLOATS = OFF_A + H;
    __device__ static constexpr int a_l0(int j, int part) { return 2 * j + part; }
    __device__ static constexpr int a_h(int l, int j, int part) { return 2 * KS0 + 4 * l + 2 * j + part; }
    // g: H = 32: 0 = t-net, 1 = s-net (one k-step each); H = 16: the k-step (0 consumes the t-net's units, 1 the s-net's)
    __device__ static constexpr int a_o(int g, int part) { return 2 * KS0 + 4 * (L - 1) + 2 * g + part; }
    __device__ static constexpr int b_0() { return 0; }
    __device__ static constexpr int b_h(int l) { return 32 + 32 * l; }
    __device__ static constexpr int b_o(int g) { return 32 + 32 * (L - 1) + 32 * g; }
};
static_assert(Img3<32, 3>::FLOATS % 4 == 0 && Img3<16, 1>::FLOATS % 4 == 0, "images must stay 16-byte aligned");

// eight fp32 values (k-slots 0..7) -> hi(8 f16), lo(8 f16)
__device__ __forceinline__ void split8r(const float (&v)[8], u4& hi, u4& lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const HiLo s = split2r(v[2 * p], v[2 * p + 1]);
        hi[p] = s.hi;
        lo[p] = s.lo;
    }
}

// max over the lanes of one net's rows (rows 0..15 / 16..31 of a 32-row operand: lanes with (lane & 16) == 16 net)
__device__ __forceinline__ float net_max(float v, int lane, int net) {
    return wave_max(((lane >> 4) & 1) == net ? v : 0.f);
}

// One wave builds the image of the coupling layer c; arguments as build_image2 (f16_tile2.h).
// Parameter layout (bijectors.py:222-235): per MLP layer [W_t | W_s | b_t | b_s], W[in][out].
template <int H, int L>
__device__ __forceinline__ void build_image3(float* img, const float* __restrict__ p, int U, int lane, const float* foldc,
                                             const float* foldprev, int c, float sc_in, float sc_prev, float sig_next) {
    typedef Img3<H, L> I;
    constexpr int D = 2 * H;
    constexpr int KS0 = I::KS0;
    const int m = lane & 31, h = lane >> 5;
    const int net = m >> 4, u = m & 15;  // this lane's row as a hidden unit
    const bool u_ok = u < U;
    const int coff = (c & 1) ? H : 0, toff = (c & 1) ? 0 : H;
    u4* ga = reinterpret_cast<u4*>(img) + lane;
    float* bias = img + I::OFF_B;
    float Sst[L][2];

    // ---- layer 0: H -> U, feeds a tanh (scale 2 log2 e); absorbs A_c x + B_c with x = sc_in * register ----
    {
        const float* w = p + net * H * U;
        const float* b = p + 2 * H * U + net * U;
        float pb = 0.f;
#pragma unroll
        for (int j = 0; j < KS0; ++j) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int f = 16 * j + r_entry(e, h);
                const float raw = kTwoLog2e * ld_sel(w, f * U + u, u_ok);
                pb = __builtin_fmaf(raw, foldc[D + coff + f], pb);
                v[e] = raw * (foldc[coff + f] * sc_in);
            }
            u4 hi, lo;
            split8r(v, hi, lo);
            ga[I::a_l0(j, 0) * 64] = hi;
            ga[I::a_l0(j, 1) * 64] = lo;
        }
        pb += __shfl_xor(pb, 32);
        const float b0 = kTwoLog2e * ld_sel(b, u, u_ok) + pb;
        if (h == 0) bias[I::b_0() + m] = b0;
        p += 2 * H * U + 2 * U;
    }
    // ---- hidden layers: U -> U per net, consume r' = r / S (weights carry -2 and S), feed a tanh ----
#pragma unroll
    for (int l = 0; l < L - 1; ++l) {
        const float* w = p + net * U * U;
        const float* b = p + 2 * U * U + net * U;
        float v[8], cs = 0.f, mx = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = r_entry(e, h);  // input unit of this lane's own net
            const float raw = ld_sel(w, k * U + u, u_ok && k < U);
            cs += raw;
            v[e] = -2.f * kTwoLog2e * raw;
            mx = fmaxf(mx, fabsf(v[e]));
        }
        cs += __shfl_xor(cs, 32);
        const float St = pow2i(norm_exponent(net_max(mx, lane, 0), -15, 40));
        const float Ss = pow2i(norm_exponent(net_max(mx, lane, 1), -15, 40));
        Sst[l][0] = St;
        Sst[l][1] = Ss;
        const float S = net ? Ss : St;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= S;
        u4 hi, lo;
        split8r(v, hi, lo);
        const u4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 2; ++j) {  // k-step j holds the units of net j: rows of the other net see zeros
            ga[I::a_h(l, j, 0) * 64] = (net == j) ? hi : zero;
            ga[I::a_h(l, j, 1) * 64] = (net == j) ? lo : zero;
        }
        if (h == 0) bias[I::b_h(l) + m] = u_ok ? kTwoLog2e * (ld_sel(b, u, u_ok) + cs) : 0.f;
        p += 2 * U * U + 2 * U;
    }
    // ---- output layer: U -> H per net, consumes r'; t carries sig_next and the pending B of the transformed half ----
    {
        const float* wt = p;
        const float* ws = p + U * H;
        const float* bt = p + 2 * U * H;
        const float* bs = bt + H;
        if constexpr (H == 32) {
            const int fo = m;  // this lane's row: output feature, for both nets
            float vt[8], vs[8], ct = 0.f, cs = 0.f, mt = 0.f, ms = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = r_entry(e, h);
                const float a = ld_sel(wt, k * H + fo, k < U), b = ld_sel(ws, k * H + fo, k < U);
                ct += a;
                cs += b;
                vt[e] = -2.f * a * sig_next;
                vs[e] = -2.f * kLog2e * b;
                mt = fmaxf(mt, fabsf(vt[e]));
                ms = fmaxf(ms, fabsf(vs[e]));
            }
            ct += __shfl_xor(ct, 32);
            cs += __shfl_xor(cs, 32);
            const float St = pow2i(norm_exponent(wave_max(mt), -15, 40));
            const float Ss = pow2i(norm_exponent(wave_max(ms), -15, 40));
            Sst[L - 1][0] = St;
            Sst[L - 1][1] = Ss;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                vt[e] *= St;
                vs[e] *= Ss;
            }
            u4 hi, lo;
            split8r(vt, hi, lo);
            ga[I::a_o(0, 0) * 64] = hi;
            ga[I::a_o(0, 1) * 64] = lo;
            split8r(vs, hi, lo);
            ga[I::a_o(1, 0) * 64] = hi;
            ga[I::a_o(1, 1) * 64] = lo;
            if (h == 0) {
                const float Ac = foldc[toff + fo], Bc = foldc[D + toff + fo];
                const float Ap = foldprev ? foldprev[toff + fo] * sc_prev : 1.f;
                const float Bp = foldprev ? foldprev[D + toff + fo] : 0.f;
                img[I::OFF_A + fo] = Ac * Ap * sig_next;
                bias[I::b_o(0) + fo] = (bt[fo] + ct - __builtin_fmaf(Ac, Bp, Bc)) * sig_next;
                bias[I::b_o(1) + fo] = kLog2e * (bs[fo] + cs);
            }
        } else {
            const int fo = m & 15;  // rows 0..15: t features, 16..31: s features
            const float* w = net ? ws : wt;
            float v[8], cs = 0.f, mx = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = r_entry(e, h);
                const float a = ld_sel(w, k * H + fo, k < U);
                cs += a;
                v[e] = net ? -2.f * kLog2e * a : -2.f * a * sig_next;
                mx = fmaxf(mx, fabsf(v[e]));
            }
            cs += __shfl_xor(cs, 32);
            const float St = pow2i(norm_exponent(net_max(mx, lane, 0), -15, 40));
            const float Ss = pow2i(norm_exponent(net_max(mx, lane, 1), -15, 40));
            Sst[L - 1][0] = St;
            Sst[L - 1][1] = Ss;
            const float S = net ? Ss : St;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= S;
            u4 hi, lo;
            split8r(v, hi, lo);
            const u4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ga[I::a_o(j, 0) * 64] = (net == j) ? hi : zero;
                ga[I::a_o(j, 1) * 64] = (net == j) ? lo : zero;
            }
            if (h == 0) {
                if (net == 0) {
                    const float Ac = foldc[toff + fo], Bc = foldc[D + toff + fo];
                    const float Ap = foldprev ? foldprev[toff + fo] * sc_prev : 1.f;
                    const float Bp = foldprev ? foldprev[D + toff + fo] : 0.f;
                    img[I::OFF_A + fo] = Ac * Ap * sig_next;
                    bias[I::b_o(0) + m] = (bt[fo] + cs - __builtin_fmaf(Ac, Bp, Bc)) * sig_next;
                } else {
                    bias[I::b_o(0) + m] = kLog2e * (bs[fo] + cs);
                }
            }
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

// sixteen accumulator initial values of lane half h: entries R(i, h) of a 32-entry row vector in LDS
__device__ __forceinline__ f16v ld_rows16(const float* v, int h) {
    f16v o;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f4 t = *reinterpret_cast<const f4*>(v + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[4 * g + e] = t[e];
    }
    return o;
}

// registers 8j .. 8j+7 -> the f16 fragments (hi, lo) of k-step j
__device__ __forceinline__ void split_step(const float* v8, h8& hi, h8& lo) {
    const HiLo a = split2r(v8[0], v8[1]), b = split2r(v8[2], v8[3]);
    const HiLo c = split2r(v8[4], v8[5]), d = split2r(v8[6], v8[7]);
    hi = __builtin_bit_cast(h8, u4{a.hi, b.hi, c.hi, d.hi});
    lo = __builtin_bit_cast(h8, u4{a.lo, b.lo, c.lo, d.lo});
}

// One coupling layer (inverse direction) on one 32-sample group.  x: conditioner half (H / 2 registers, unchanged),
// y: transformed half, ssum2 += this lane's share of sum(s) log2(e).  SLOW: layer 0 in exact fp32 MFMAs.
template <int H, int L, bool SLOW>
__device__ __forceinline__ void coupling_tile3(const float* img, int lane, const float (&x)[H / 2], float (&y)[H / 2],
                                               float& ssum2) {
    typedef Img3<H, L> I;
    constexpr int KS0 = I::KS0;
    constexpr int NR = H / 2;
    const int h = lane >> 5;
    const u4* ga = reinterpret_cast<const u4*>(img) + lane;
    auto A = [&](int g) -> h8 { return __builtin_bit_cast(h8, ga[g * 64]); };
    const float* bias = img + I::OFF_B;
    const f4 Sa = *reinterpret_cast<const f4*>(img + I::OFF_S);
    f4 Sb = {1.f, 1.f, 1.f, 1.f};
    if constexpr (L == 3) Sb = *reinterpret_cast<const f4*>(img + I::OFF_S + 4);
    auto Sc = [&](int stage, int net) -> float { return (2 * stage + net) < 4 ? Sa[2 * stage + net] : Sb[2 * stage + net - 4]; };

    // ---- layer 0 ----
    f16v acc = ld_rows16(bias + I::b_0(), h);
    if constexpr (!SLOW) {
        h8 xh[KS0], xl[KS0];
#pragma unroll
        for (int j = 0; j < KS0; ++j) split_step(&x[8 * j], xh[j], xl[j]);
#pragma unroll
        for (int j = 0; j < KS0; ++j) acc = mfma32x32h(A(I::a_l0(j, 0)), xh[j], acc);
#pragma unroll
        for (int j = 0; j < KS0; ++j) acc = mfma32x32h(A(I::a_l0(j, 0)), xl[j], acc);
#pragma unroll
        for (int j = 0; j < KS0; ++j) acc = mfma32x32h(A(I::a_l0(j, 1)), xh[j], acc);
    } else {
        // exact path: weights rebuilt as hi + lo, inputs as they are; step (j, e) of v_mfma_f32_32x32x2_f32 contracts
        // entries 16 j + R(e, h), h = 0, 1 -- the entry register 8j + e holds
#pragma unroll
        for (int j = 0; j < KS0; ++j) {
            const h8 wh = A(I::a_l0(j, 0)), wl = A(I::a_l0(j, 1));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = mfma32x32f((float)wh[e] + (float)wl[e], x[8 * j + e], acc);
        }
    }
    // ---- sigmoids of stage 0 (registers 0..7: t-net units, 8..15: s-net units), split ----
    h8 rh[2], rl[2];
    auto activate = [&](int stage) {
        float r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = sig2s(acc[i], Sc(stage, i >> 3));
        split_step(&r[0], rh[0], rl[0]);
        split_step(&r[8], rh[1], rl[1]);
    };
    activate(0);
    // ---- hidden layers ----
#pragma unroll
    for (int l = 0; l < L - 1; ++l) {
        acc = ld_rows16(bias + I::b_h(l), h);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc = mfma32x32h(A(I::a_h(l, j, 0)), rh[j], acc);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc = mfma32x32h(A(I::a_h(l, j, 0)), rl[j], acc);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc = mfma32x32h(A(I::a_h(l, j, 1)), rh[j], acc);
        activate(l + 1);
    }
    // ---- output layer and the update of the transformed half ----
    if constexpr (H == 32) {
        f16v tt = ld_rows16(bias + I::b_o(0), h), sv = ld_rows16(bias + I::b_o(1), h);
        tt = mfma32x32h(A(I::a_o(0, 0)), rh[0], tt);
        sv = mfma32x32h(A(I::a_o(1, 0)), rh[1], sv);
        tt = mfma32x32h(A(I::a_o(0, 0)), rl[0], tt);
        sv = mfma32x32h(A(I::a_o(1, 0)), rl[1], sv);
        tt = mfma32x32h(A(I::a_o(0, 1)), rh[0], tt);
        sv = mfma32x32h(A(I::a_o(1, 1)), rh[1], sv);
        const f16v ay = ld_rows16(img + I::OFF_A, h);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const float s2 = sv[i];
            ssum2 += s2;
            y[i] = __builtin_fmaf(y[i], ay[i], -tt[i]) * __builtin_amdgcn_exp2f(-s2);
        }
    } else {
        f16v o = ld_rows16(bias + I::b_o(0), h);  // registers 0..7: t of features R(i, h), 8..15: s
#pragma unroll
        for (int j = 0; j < 2; ++j) o = mfma32x32h(A(I::a_o(j, 0)), rh[j], o);
#pragma unroll
        for (int j = 0; j < 2; ++j) o = mfma32x32h(A(I::a_o(j, 0)), rl[j], o);
#pragma unroll
        for (int j = 0; j < 2; ++j) o = mfma32x32h(A(I::a_o(j, 1)), rh[j], o);
        const f4 ay0 = *reinterpret_cast<const f4*>(img + I::OFF_A + 4 * h);
        const f4 ay1 = *reinterpret_cast<const f4*>(img + I::OFF_A + 8 + 4 * h);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const float s2 = o[8 + i];
            const float ay = i < 4 ? ay0[i & 3] : ay1[i & 3];
            ssum2 += s2;
            y[i] = __builtin_fmaf(y[i], ay, -o[i]) * __builtin_amdgcn_exp2f(-s2);
        }
    }
}

}  // namespace tnf
