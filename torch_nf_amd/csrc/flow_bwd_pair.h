// Whole-flow backward of loss = f(NormFlow.log_prob(z)): the round-3 EXPERIMENT, selectable with TNF_OPT_REV_VARIANT = 1.
// (Included by flow_bwd_f16.hip after the shared pieces -- images, transposes, AccLayout, FlowBwdArgs.)
// It did not pay: flow_bwd_f16_kernel stays the default.  Kept because the measurements behind that
// entry are reproducible only with the code (tools/revbench.py), and tests/test_gpu_grad.py holds it to the same oracle.
//
// flow_bwd_f16_kernel (one tile per wave, 12 waves) spends, per tile and layer (ISA count): 184 of ~575 vector
// instructions in hi/lo splits and packs, 114 in the fixed-point conversion of the twelve weight-gradient tiles (mul,
// v_max3, v_cvt_rpi per element) + 46 ds_add_u32, 36 in the LDS transposes of tanh outputs.  This form changes the
// accounting, not the arithmetic:
//   * TNF_PAIR_NT = 2: a wave carries a PAIR of tiles (32 samples).  A weight-gradient outer product contracts over
//     samples, and the contraction does not care which K slot a sample sits in as long as both operands agree: the
//     K = 32 operand of the pair is the concatenation of the two tiles' K = 16 operands, register for register.  Twelve
//     K = 32 products (36 MFMAs) and twelve LDS accumulations per pair replace 24 K = 16 products and 24 accumulations.
//   * the accumulation itself: the MFMA chain starts from C = magic = 1.5 * 2^(23 - f) instead of 0, so its fp32 result
//     IS the fixed-point value -- bits(t) = bits(magic) + round(v 2^f) while |v 2^f| < 2^22 -- and ds_add_u32 takes the
//     raw bits.  A word that received K adds holds K bits(magic) + sum n_i (mod 2^32); K is the same for every word
//     (iterations x waves) and comes off at the flush.  Per gradient tile: 4 min/max (range tracking) + 4 ds_add_u32.
//     Price: the sum is rounded to the 2^-f grid after each of the chain's three MFMAs instead of once.
//   * tanh outputs enter the outer products as r = (1 - tanh)/2, the sigmoid the forward recompute already split for its
//     own MFMAs (transposed on the matrix pipe like the deltas): sum_s tanh_k d_o = sum_s d_o - 2 sum_s r_k d_o, and
//     sum_s d_o is the bias gradient.  With num_units <= 15 the padded unit 15 carries r = 1, so row 15 of the tile is
//     that bias gradient; the reduction over workgroups forms db - 2 G in 64-bit integers.  No LDS transposes of
//     activations, no separate splits.  tanh' = 4 r (1 - r): the 4 rides in the transposed weight image (hscale).
//   * the next layer's operand image reaches the LDS ring by LDS-DMA (adopted by flow_bwd_f16_kernel as well).
// Measured (D = 64, S = 4, N = 2^19, steady state; PMC: gpurun_out -> profiles/r03_pmc_flow_bwd_variants.json):
//   2 tiles x 8 waves  0.575-0.583 ms, SQ_INSTS_VALU -23 %, MFMA -12 %, LDS -50 % vs flow_bwd_f16_kernel at 0.550-0.555 ms:
//     256 VGPRs with 55 spilled, and at two waves per SIMD a vector instruction costs the SIMD 1.25-1.5 x what it costs at
//     three (v_fma 3.25 vs 2.6 cycles, v_cvt_pk 5.1 vs 4.3: DESIGN.md 3.10.2's table) -- the saving is eaten.
//   2 x 4 (512 registers, no spills) 0.714 ms;  1 tile x 12 waves (this file's default) 0.528-0.582 ms;  D = 32: 0.437
//     against 0.417 ms.  Gradient error against the fp32 layer kernels 7e-6 of the largest entry (2^-9 grid, three
//     roundings, the db - 2 G subtraction) against 4e-7.
#pragma once

namespace tnf {

typedef __attribute__((address_space(3))) void lds_void;
// Tiles per wave and waves per workgroup.  Measured at D = 64, N = 2^19 (MI355X, steady state): 2 tiles x 8 waves
// 0.583 ms (256 VGPRs, 55 spilled: the per-instruction issue cost of two waves per SIMD eats what the pairing saves),
// 2 x 4 0.714 ms, the round-2 kernel (1 x 12, different accumulation) 0.555-0.594 ms.
#ifndef TNF_PAIR_NT
#define TNF_PAIR_NT 1
#endif
#ifndef TNF_PAIR_NW
#define TNF_PAIR_NW (TNF_PAIR_NT == 1 ? 12 : 8)
#endif
constexpr int kPairNT = TNF_PAIR_NT;
constexpr int kPairNW = TNF_PAIR_NW;
constexpr int kPairTermBits = 13;                 // budget per accumulated term: 2^13 in units where max |g_log_prob| is in [1, 2)
constexpr int kPairMaxFbits = 22 - kPairTermBits;  // ... which must stay inside the 2^22 range of the magic-number form

struct MagicAcc {
    f4 magic;
    float hi, lo;  // running max / min of the raw sums (range check at the flush)
};

__device__ __forceinline__ h8 cat8(h4 a, h4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

// D[rows of a][rows of b] = c + sum over the pair's 32 samples (slots 0-3 of a lane: tile A, slots 4-7: tile B)
__device__ __forceinline__ f4 outer32h(const T16& aA, const T16& aB, const T16& bA, const T16& bB, f4 c) {
    const h8 ah = cat8(aA.hi, aB.hi), al = cat8(aA.lo, aB.lo), bh = cat8(bA.hi, bB.hi), bl = cat8(bA.lo, bB.lo);
    c = mfma32h(ah, bh, c);
    c = mfma32h(al, bh, c);
    return mfma32h(ah, bl, c);
}
__device__ __forceinline__ f4 rowsum32h(const T16& aA, const T16& aB) {
    const _Float16 o = (_Float16)1.f;
    const h8 ones = {o, o, o, o, o, o, o, o};
    f4 c = mfma32h(cat8(aA.hi, aB.hi), ones, f4{0.f, 0.f, 0.f, 0.f});
    return mfma32h(cat8(aA.lo, aB.lo), ones, c);
}
template <int NT>
__device__ __forceinline__ f4 outer_nt(const T16 (&a)[NT], const T16 (&b)[NT], f4 c) {
    if constexpr (NT == 2) return outer32h(a[0], a[1], b[0], b[1], c);
    else return outer16h_acc(a[0], b[0], c);
}
template <int NT>
__device__ __forceinline__ f4 rowsum_nt(const T16 (&a)[NT]) {
    if constexpr (NT == 2) return rowsum32h(a[0], a[1]);
    else return rowsum16h(a[0]);
}
// one AccLayout tile, p = tile + lane; t = magic + value (the MFMA chain started from magic)
__device__ __forceinline__ void lds_add_magic(int* p, f4 t, MagicAcc& ma) {
#if TNF_PAIR_DEBUG == 1  // bisect: the plain fixed-point path on (t - magic)
    {
        const float fxs = 12582912.f / ma.magic[0];
        atomicAdd(p + 0, __float2int_rn((t[0] - ma.magic[0]) * fxs) + __builtin_bit_cast(int, ma.magic[0]));
        atomicAdd(p + 64, __float2int_rn((t[1] - ma.magic[0]) * fxs) + __builtin_bit_cast(int, ma.magic[0]));
        atomicAdd(p + 128, __float2int_rn((t[2] - ma.magic[0]) * fxs) + __builtin_bit_cast(int, ma.magic[0]));
        atomicAdd(p + 192, __float2int_rn((t[3] - ma.magic[0]) * fxs) + __builtin_bit_cast(int, ma.magic[0]));
        return;
    }
#endif
    ma.hi = __builtin_fmaxf(__builtin_fmaxf(ma.hi, t[0]), t[1]);
    ma.hi = __builtin_fmaxf(__builtin_fmaxf(ma.hi, t[2]), t[3]);
    ma.lo = __builtin_fminf(__builtin_fminf(ma.lo, t[0]), t[1]);
    ma.lo = __builtin_fminf(__builtin_fminf(ma.lo, t[2]), t[3]);
    // (the whole vector is reinterpreted at once: hipcc 7.2 selects element 0 four times for
    //  atomicAdd(p + 64 j, __builtin_bit_cast(int, t[j])) on an MFMA result)
    typedef int i4 __attribute__((ext_vector_type(4)));
    const i4 ti = __builtin_bit_cast(i4, t);
    atomicAdd(p + 0, ti[0]);
    atomicAdd(p + 64, ti[1]);
    atomicAdd(p + 128, ti[2]);
    atomicAdd(p + 192, ti[3]);
}

// One inverse-pass coupling layer backwards on a pair of tiles (MODE 0 of layer_bwd16; same arithmetic per sample).
template <int H, int L, bool SPARE, int NT>
__device__ __forceinline__ void layer_bwd_pair(const float* img, int* acc, MagicAcc& ma, FxAcc& fa, int lane, int U,
                                               const f4 (&x)[NT][(H + 15) / 16], f4 (&y)[NT][(H + 15) / 16],
                                               f4 (&gx)[NT][(H + 15) / 16], f4 (&gy)[NT][(H + 15) / 16],
                                               const float (&gl)[NT]) {
    typedef F16Image<H, L> FImg;
    typedef B16Image<H, L> BImg;
    typedef RevImage<H, L> R;
    typedef AccLayout<H, L> A_;
    constexpr int HT = FImg::HT;
    const int s = lane & 15, q = lane >> 4;
    (void)U;
    const u4* fg = reinterpret_cast<const u4*>(img + R::F_OFF) + lane;
    const u4* bg = reinterpret_cast<const u4*>(img + R::B_OFF) + lane;
    const float* bl = img + R::F_OFF + FImg::NWG * 256 + q * 4;
    auto bias = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(bl + g * 16); };
    auto hl = [&](const u4* base, int g, h4& hi, h4& lo) {
        const u4 wv = base[g * 64];
        hi = __builtin_bit_cast(h4, u2{wv[0], wv[1]});
        lo = __builtin_bit_cast(h4, u2{wv[2], wv[3]});
    };
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    h4 ident;  // B operand of the identity: lane (n = s, q) holds K = 4q + i
#pragma unroll
    for (int i = 0; i < 4; ++i) ident[i] = (4 * q + i == s) ? (_Float16)1.f : (_Float16)0.f;
    auto with_one = [&](f4 v) -> f4 {  // unit 15 (lane group q = 3, register 3) is padding when num_units <= 15
        if (SPARE) v[3] = (q == 3) ? 1.f : v[3];
        return v;
    };
    int* accl = acc + lane;

    // ---- 1. forward recompute.  u = r (1 - r) is what tanh' needs (x 4, in the transposed image); r_t: the sigmoids
    //         as sample-contracting operands ----
    f4 u[NT][L][2];
    T16 r_t[L][2][NT];
    f4 dout[NT][2][HT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f4 r0[2];
        if constexpr (H == 32) {
            u4 a_, b_;
            split2(x[t][0][0], x[t][0][1], a_[0], b_[0]);
            split2(x[t][0][2], x[t][0][3], a_[1], b_[1]);
            split2(x[t][1][0], x[t][1][1], a_[2], b_[2]);
            split2(x[t][1][2], x[t][1][3], a_[3], b_[3]);
            const h8 xh = __builtin_bit_cast(h8, a_), xl = __builtin_bit_cast(h8, b_);
#pragma unroll
            for (int net = 0; net < 2; ++net) {
                const h8 wh = __builtin_bit_cast(h8, fg[FImg::g_w0(net, 0) * 64]);
                const h8 wl = __builtin_bit_cast(h8, fg[FImg::g_w0(net, 1) * 64]);
                f4 a0 = mfma32h(wh, xh, bias(FImg::b_b0(net)));
                a0 = mfma32h(wh, xl, a0);
                a0 = mfma32h(wl, xh, a0);
                r0[net] = with_one(sig2_4(a0));
            }
        } else {
            h4 xh, xl;
            split4(x[t][0], xh, xl);
#pragma unroll
            for (int net = 0; net < 2; ++net) {
                h4 wh, wl;
                hl(fg, FImg::g_w0(net, 0), wh, wl);
                r0[net] = with_one(sig2_4(mm3(wh, wl, xh, xl, bias(FImg::b_b0(net)))));
            }
        }
        h4 rh[2], rl[2];
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int net = 0; net < 2; ++net) {
                const f4 rr = r0[net];
#pragma unroll
                for (int j = 0; j < 4; ++j) u[t][l][net][j] = __builtin_fmaf(-rr[j], rr[j], rr[j]);
                split4(rr, rh[net], rl[net]);
                r_t[l][net][t] = mtrans(rh[net], rl[net], ident);
                if (l + 1 < L) {
                    h4 wh, wl;
                    hl(fg, FImg::g_wh(l, net), wh, wl);
                    r0[net] = with_one(sig2_4(mm3(wh, wl, rh[net], rl[net], bias(FImg::b_bh(l, net)))));
                }
            }
        // output layer, rebuilt input, output deltas
#pragma unroll
        for (int mo = 0; mo < HT; ++mo) {
            h4 wh, wl;
            hl(fg, FImg::g_w2(0, mo), wh, wl);
            const f4 tt = mm3(wh, wl, rh[0], rl[0], bias(FImg::b_b2(0, mo)));
            hl(fg, FImg::g_w2(1, mo), wh, wl);
            const f4 sv = mm3(wh, wl, rh[1], rl[1], bias(FImg::b_b2(1, mo)));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = __builtin_amdgcn_exp2f(sv[j]);
                const float em = __builtin_amdgcn_exp2f(-sv[j]);
                const float g = gy[t][mo][j], yo = y[t][mo][j];
                const float dy = g * em;
                dout[t][0][mo][j] = -dy;
                dout[t][1][mo][j] = __builtin_fmaf(-g, yo, gl[t]);
                y[t][mo][j] = __builtin_fmaf(yo, e, tt[j]);
                gy[t][mo][j] = dy;
            }
        }
    }

    // ---- 2. output layer: G2 = sum_s d_o r_k (the flush turns it into dW2, db2), d h_{L-1} (x 4) ----
    f4 dh[NT][2];
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        h4 dsh[NT][HT], dsl[NT][HT];
        T16 d_t[HT][NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mo = 0; mo < HT; ++mo) {
                split4(dout[t][net][mo], dsh[t][mo], dsl[t][mo]);
                d_t[mo][t] = mtrans(dsh[t][mo], dsl[t][mo], ident);
            }
#pragma unroll
        for (int mo = 0; mo < HT; ++mo) {
            lds_add_magic(accl + A_::o_w2 + (net * A_::HT + mo) * A_::TILE,
                          outer_nt<NT>(d_t[mo], r_t[L - 1][net], ma.magic), ma);
            if (!SPARE) lds_add_rows(acc + A_::o_b2 + net * H + 16 * mo + 4 * q, rowsum_nt<NT>(d_t[mo]), s, fa);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (H == 32) {
                const u2 h0_ = __builtin_bit_cast(u2, dsh[t][0]), h1_ = __builtin_bit_cast(u2, dsh[t][1]);
                const u2 l0_ = __builtin_bit_cast(u2, dsl[t][0]), l1_ = __builtin_bit_cast(u2, dsl[t][1]);
                const h8 dhi = __builtin_bit_cast(h8, u4{h0_[0], h0_[1], h1_[0], h1_[1]});
                const h8 dlo = __builtin_bit_cast(h8, u4{l0_[0], l0_[1], l1_[0], l1_[1]});
                const h8 wh = __builtin_bit_cast(h8, bg[BImg::g_w2(net, 0) * 64]);
                const h8 wl = __builtin_bit_cast(h8, bg[BImg::g_w2(net, 1) * 64]);
                f4 a0 = mfma32h(wh, dhi, zero);
                a0 = mfma32h(wh, dlo, a0);
                dh[t][net] = mfma32h(wl, dhi, a0);
            } else {
                h4 wh, wl;
                hl(bg, BImg::g_w2(net, 0), wh, wl);
                dh[t][net] = mm3(wh, wl, dsh[t][0], dsl[t][0], zero);
            }
        }
    }
    // ---- 3. hidden layers, last to first ----
#pragma unroll
    for (int l = L - 2; l >= 0; --l)
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            h4 dhi[NT], dlo[NT];
            T16 d_t[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f4 da;
#pragma unroll
                for (int j = 0; j < 4; ++j) da[j] = dh[t][net][j] * u[t][l + 1][net][j];
                split4(da, dhi[t], dlo[t]);
                d_t[t] = mtrans(dhi[t], dlo[t], ident);
            }
            lds_add_magic(accl + A_::o_h + l * A_::HID + net * A_::TILE,
                          outer_nt<NT>(d_t, r_t[l][net], ma.magic), ma);
            if (!SPARE) lds_add_rows(acc + A_::o_h + l * A_::HID + 2 * A_::TILE + net * 16 + 4 * q, rowsum_nt<NT>(d_t), s, fa);
            h4 wh, wl;
            hl(bg, BImg::g_wh(l, net), wh, wl);
#pragma unroll
            for (int t = 0; t < NT; ++t) dh[t][net] = mm3(wh, wl, dhi[t], dlo[t], zero);
        }
    // ---- 4. first layer: dW0, db0, d x ----
    // (the conditioner half is unchanged by the layer: splitting it again here costs 8 instructions per 16 features and
    //  frees the registers its halves would hold across steps 2 and 3)
    T16 x_t[HT][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            h4 xh, xl;
            split4(x[t][mm], xh, xl);
            x_t[mm][t] = mtrans(xh, xl, ident);
        }
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        h4 dhi[NT], dlo[NT];
        T16 d_t[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f4 da;
#pragma unroll
            for (int j = 0; j < 4; ++j) da[j] = dh[t][net][j] * u[t][0][net][j];
            split4(da, dhi[t], dlo[t]);
            d_t[t] = mtrans(dhi[t], dlo[t], ident);
        }
        lds_add_rows(acc + A_::o_b0 + net * 16 + 4 * q, rowsum_nt<NT>(d_t), s, fa);
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            lds_add_magic(accl + A_::o_w0 + (net * A_::HT + mm) * A_::TILE,
                          outer_nt<NT>(d_t, x_t[mm], ma.magic), ma);
            h4 wh, wl;
            hl(bg, BImg::g_w0(net, mm), wh, wl);
#pragma unroll
            for (int t = 0; t < NT; ++t) gx[t][mm] = mm3(wh, wl, dhi[t], dlo[t], gx[t][mm]);
        }
    }
}

// unfold_half for a pair: v <- (v - B)/A, g <- g A on both tiles; with AFFINE the pair's contributions to
// dA = sum g v, dB = sum g are summed over both tiles before they are converted and added.
template <int H, bool AFFINE, int NT>
__device__ __forceinline__ void unfold_pair(const float* fc, int* gf, FxAcc& fa, float* scrA, float* scrB, int lane, int f0,
                                            f4 (&v)[NT][(H + 15) / 16], f4 (&g)[NT][(H + 15) / 16]) {
    constexpr int D = 2 * H;
    constexpr int HT = (H + 15) / 16;
    const int s = lane & 15, q = lane >> 4;
#pragma unroll
    for (int mm = 0; mm < HT; ++mm) {
        const int f = f0 + 16 * mm + 4 * q;
        const f4 A = *reinterpret_cast<const f4*>(fc + f);
        const f4 iA = *reinterpret_cast<const f4*>(fc + 2 * D + f);
        const f4 C = *reinterpret_cast<const f4*>(fc + 3 * D + f);
        f4 vp[NT], gv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vp[t][j] = __builtin_fmaf(v[t][mm][j], iA[j], C[j]);
                gv[t][j] = g[t][mm][j] * vp[t][j];
            }
        if (AFFINE) {
            // (the wave's LDS instructions execute in order: a scratch tile is reused as soon as its read was issued)
            float da = 0.f, db = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f4 ta = transpose16(gv[t], (t & 1) ? scrB : scrA, lane);
                const f4 tb = transpose16(g[t][mm], (t & 1) ? scrA : scrB, lane);
                da += (ta[0] + ta[1]) + (ta[2] + ta[3]);
                db += (tb[0] + tb[1]) + (tb[2] + tb[3]);
            }
            da *= fa.fx;
            db *= fa.fx;
            fa.amax = amax3(fa.amax, da, db);
            atomicAdd(gf + f0 + 16 * mm + s, fx_cvt(da));  // feature = row s of the transposed tiles
            atomicAdd(gf + D + f0 + 16 * mm + s, fx_cvt(db));
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g[t][mm][j] *= A[j];
                v[t][mm][j] = vp[t][j];
            }
    }
}

// Raw fixed-point value of element kk of a layer's parameter block (bijectors.py:222-235) from this workgroup's
// accumulators: weight tiles carry kmagic = adds * bits(magic) (mod 2^32).  For a weight behind a tanh this is
// G = sum_s r_k d_o, not the gradient: flow_bwd_reduce_kernel forms db - 2 G after summing the workgroups' rows, in 64 bits.
template <int H, int L, bool SPARE>
__device__ __forceinline__ int acc_value_pair(const int* acc, int kk, int U, unsigned kmagic) {
    typedef AccLayout<H, L> A_;
    auto w = [&](int idx) -> int { return (int)((unsigned)acc[idx] - kmagic); };
    if (kk < 2 * H * U + 2 * U) {
        if (kk < 2 * H * U) return w(A_::w0(kk / (H * U), (kk / U) % H, kk % U));
        return acc[A_::o_b0 + ((kk - 2 * H * U) / U) * 16 + (kk - 2 * H * U) % U];
    }
    kk -= 2 * H * U + 2 * U;
    const int hs = 2 * U * U + 2 * U;
    if (kk < (L - 1) * hs) {
        const int l = kk / hs, r = kk - l * hs;
        if (r < 2 * U * U) {
            const int net = r / (U * U), rr = r - net * U * U;
            return w(A_::wh(l, net, rr / U, rr % U));
        }
        const int net = (r - 2 * U * U) / U, ko = (r - 2 * U * U) % U;
        if (SPARE) return w(A_::wh(l, net, 15, ko));
        return acc[A_::o_h + l * A_::HID + 2 * A_::TILE + net * 16 + ko];
    }
    kk -= (L - 1) * hs;
    if (kk < 2 * U * H) {
        const int net = kk / (U * H), rr = kk - net * U * H;
        return w(A_::w2(net, rr / H, rr % H));
    }
    const int net = (kk - 2 * U * H) / H, f = (kk - 2 * U * H) % H;
    if (SPARE) return w(A_::w2(net, 15, f));
    return acc[A_::o_b2 + net * H + f];
}

template <int H, int L>
struct PairLds {
    typedef RevImage<H, L> R;
    static constexpr int SLOT = (R::FLOATS + 255) & ~255;  // ring slot in floats: whole 1-KB LDS-DMA pieces
    static constexpr int NSCR = kPairNT;                   // transposition scratch tiles per wave (one per tile)
    __host__ __device__ static constexpr int64_t floats(int nl) {
        return 2 * (int64_t)SLOT + (int64_t)nl * AccLayout<H, L>::INTS + (int64_t)kPairNW * NSCR * kScr;
    }
};

template <int H, int L, bool SPARE>
__global__ void __launch_bounds__(kPairNW * 64)
flow_bwd_pair_kernel(FlowBwdArgs a) {
    constexpr int NT = kPairNT;
    typedef RevImage<H, L> R;
    typedef PairLds<H, L> PL;
    constexpr int D = 2 * H;
    constexpr int HT = (H + 15) / 16;
    constexpr int NW = kPairNW;
    constexpr int RU4 = R::FLOATS / 4;
    constexpr int NPIECE = PL::SLOT / 256;  // 1-KB pieces per image
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nl = 2 * a.S;
    const int U = a.U;
    typedef AccLayout<H, L> A_;
    constexpr int ACC = A_::INTS;
    float* ring = lds;                                            // [2][SLOT]
    int* accb = reinterpret_cast<int*>(lds + 2 * PL::SLOT);       // [nl][ACC]
    float* scr = lds + 2 * PL::SLOT + nl * ACC;                   // [NW][NSCR][kScr]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const int64_t m = grid_m();
    if (m >= a.M) return;
    const int64_t mp = a.Mp == 1 ? 0 : m;
    float* scrA = scr + wave * PL::NSCR * kScr;
    float* scrB = scrA + (PL::NSCR - 1) * kScr;
    const u4* isrc = reinterpret_cast<const u4*>(a.rimg + mp * (int64_t)nl * R::FLOATS);

    // layer image c -> ring slot: LDS-DMA, 1 KB per wave-instruction, no staging registers; a piece's tail beyond the
    // image re-reads its last 16 bytes (the slot is padded to whole pieces, the padding is never read)
    auto fetch = [&](int c, float* slot) {
        const u4* src = isrc + (int64_t)c * RU4;
        for (int i = wave; i < NPIECE; i += NW) {
            const int idx = i * 64 + lane;
            __builtin_amdgcn_global_load_lds(src + (idx < RU4 ? idx : RU4 - 1), (lds_void*)(slot + i * 256), 16, 0, 0);
        }
    };
    fetch(0, ring);
    for (int i = threadIdx.x; i < nl * ACC; i += NW * 64) accb[i] = 0;
    float sc = 1.f, isc = 1.f;
    {
        const float gm = __uint_as_float(*a.gmax);
        if (gm > 0.f && gm < 3.0e38f) {
            int e;
            (void)frexpf(gm, &e);  // gm = f 2^e, f in [0.5, 1)
            int k = 1 - e;
            k = k > 120 ? 120 : (k < -120 ? -120 : k);
            sc = ldexpf(1.f, k);
            isc = ldexpf(1.f, -k);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int64_t npairs = (a.N + 16 * NT - 1) / (16 * NT);  // units of NT tiles
    const int64_t stride = (int64_t)gridDim.x * NW;
    const int64_t iters = (npairs + stride - 1) / stride;
    const float* zb = a.z0 + m * a.N * D;
    const float* glb = a.g_lp + m * a.N;
    float* gzb = a.g_z ? a.g_z + m * a.N * D : nullptr;
    float glp_acc = 0.f;
    FxAcc fa{a.fx, 0.f};
    const float magic = 12582912.f / a.fx;  // 1.5 * 2^23 / 2^f
    MagicAcc ma{f4{magic, magic, magic, magic}, magic, magic};
    unsigned bad = 0;  // largest |bit pattern| of an outgoing gradient: inf / NaN anywhere shows up here
    int step = 0;
    const int64_t nsteps = iters * nl;

    for (int64_t it = 0; it < iters; ++it) {
        const int64_t pair = (it * gridDim.x + blockIdx.x) * NW + wave;
        f4 lo[NT][HT], hi[NT][HT], glo[NT][HT], ghi[NT][HT];
        float gl[NT];
        int64_t row[NT];
        bool row_ok[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            row[t] = (pair * NT + t) * 16 + s;
            row_ok[t] = row[t] < a.N;
            const int64_t rowc = row_ok[t] ? row[t] : a.N - 1;
            const float glp = row_ok[t] ? sc * glb[rowc] : 0.f;
            if (q == 0) glp_acc += glp;
            const float* zr = zb + rowc * D + 4 * q;
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                lo[t][mm] = *reinterpret_cast<const f4*>(zr + 16 * mm);
                hi[t][mm] = *reinterpret_cast<const f4*>(zr + H + 16 * mm);
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // d (-|z0|^2 / 2) / d z0, times g_log_prob
                    glo[t][mm][j] = -glp * lo[t][mm][j];
                    ghi[t][mm][j] = -glp * hi[t][mm][j];
                }
            }
            gl[t] = -glp;  // log_prob = base - sum of the layers' log-dets
        }

        for (int c = 0; c < nl; ++c, ++step) {
            const float* img = ring + (step & 1) * PL::SLOT;
            const bool more = (int64_t)step + 1 < nsteps;
#if TNF_PAIR_ABL != 1
            if (more) fetch((c + 1 == nl) ? 0 : c + 1, ring + ((step + 1) & 1) * PL::SLOT);
#endif
            int* acc = accb + c * ACC;
            const float* fc = img + R::C_OFF;
            if ((c & 1) == 0) {  // RealNVP(upper): conditioner = low half
                layer_bwd_pair<H, L, SPARE, NT>(img, acc, ma, fa, lane, U, lo, hi, glo, ghi, gl);
                unfold_pair<H, false, NT>(fc, acc + A_::o_fold, fa, scrA, scrB, lane, 0, lo, glo);
                unfold_pair<H, false, NT>(fc, acc + A_::o_fold, fa, scrA, scrB, lane, H, hi, ghi);
            } else {             // RealNVP(lower) behind BatchNorm + Affine
                layer_bwd_pair<H, L, SPARE, NT>(img, acc, ma, fa, lane, U, hi, lo, ghi, glo, gl);
                unfold_pair<H, true, NT>(fc, acc + A_::o_fold, fa, scrA, scrB, lane, 0, lo, glo);
                unfold_pair<H, true, NT>(fc, acc + A_::o_fold, fa, scrA, scrB, lane, H, hi, ghi);
            }
#if TNF_PAIR_ABL != 1  // (timing experiment: 1 = no barrier between steps, results are wrong)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next image have landed
            __syncthreads();
#endif
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int mm = 0; mm < HT; ++mm)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v0 = glo[t][mm][j], v1 = ghi[t][mm][j];  // (scalars: see lds_add_magic)
                    const unsigned b0 = __float_as_uint(v0) & 0x7fffffffu;
                    const unsigned b1 = __float_as_uint(v1) & 0x7fffffffu;
                    bad = max(bad, max(b0, b1));
                }
            if (gzb && row_ok[t]) {
                float* gr = gzb + row[t] * D + 4 * q;
#pragma unroll
                for (int mm = 0; mm < HT; ++mm) {
                    f4 a0, a1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a0[j] = glo[t][mm][j] * isc;
                        a1[j] = ghi[t][mm][j] * isc;
                    }
                    *reinterpret_cast<f4*>(gr + 16 * mm) = a0;
                    *reinterpret_cast<f4*>(gr + H + 16 * mm) = a1;
                }
            }
        }
    }

    // ---- flush ----
    // A term above the budget may have wrapped an accumulator or left the range of the magic-number form; an inf / NaN in
    // a gradient means an operand did (or the input was not finite).  Either way: flag it, the reduction poisons the
    // result instead of returning a wrong gradient.
    float amax = __builtin_fmaxf(fa.amax, __builtin_fmaxf(ma.hi - magic, magic - ma.lo) * a.fx);
    if (bad >= 0x7f800000u) amax = __builtin_inff();
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    float* red = scr;  // the transposition scratch is free now
    __syncthreads();
    if (lane == 0) red[wave] = amax;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) amax = fmaxf(amax, red[w]);
    const float nadds = (float)(iters * NW);
    // 2^31: the int32 accumulators; 2^22: the magic-number form (also true for NaN / inf terms)
    const bool wrapped = !(amax * nadds < 2147483648.f && amax < 4194304.f);
    if (wrapped && threadIdx.x == 0) atomicOr(a.overflow, 1);
    const int P = 2 * (H * U + U) + (L - 1) * 2 * (U * U + U) + 2 * (U * H + H);
    float tot = glp_acc;
    tot += __shfl_xor(tot, 1);
    tot += __shfl_xor(tot, 2);
    tot += __shfl_xor(tot, 4);
    tot += __shfl_xor(tot, 8);  // lanes 0..15 (q = 0) carried the terms
    __syncthreads();
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    {
        const unsigned kmagic = (unsigned)(iters * NW) * __builtin_bit_cast(unsigned, magic);
        const int64_t nred = (a.Mp == 1 ? a.M : 1) * gridDim.x;
        const int64_t blk = (a.Mp == 1 ? m : 0) * gridDim.x + blockIdx.x;
        const int64_t prow = (int64_t)nl * (P + 2 * D);
        int* dst = a.partials + (mp * nred + blk) * prow;
        for (int i = threadIdx.x; i < nl * (P + 2 * D); i += NW * 64) {
            const int c = i / (P + 2 * D);
            const int k = i - c * (P + 2 * D);
            const int* acc = accb + c * ACC;
            dst[i] = (k >= P) ? ((c & 1) ? acc[A_::o_fold + (k - P)] : 0) : acc_value_pair<H, L, SPARE>(acc, k, U, kmagic);
        }
        if (threadIdx.x == 0) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[w];
            a.glp_part[mp * nred + blk] = t * isc;
        }
    }
}

}  // namespace tnf
