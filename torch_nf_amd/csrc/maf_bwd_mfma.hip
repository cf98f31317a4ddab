// Backward of MAF.inverse_and_log_det on the matrix pipe (float32) -- what training through
// NormFlow('AR').log_prob differentiates (the LFI scripts' inner loop: per-context parameter rows).
//   out = (z - mu(z)) e^-alpha(z),  ld = sum alpha
//   given g_out (M,N,D), g_ld (M,N):  g_z (M,N,D),  g_params (M_p, P)
// Per 16-sample tile and wave, in the transposed MFMA formulation of maf_mfma.hip (units on accumulator
// rows, samples on columns):
//   1. forward recompute from the folded operand image, keeping every layer's r = (1 - tanh)/2;
//   2. output deltas  d_mu = -g e^-alpha,  d_alpha = -g out + g_ld,  g_z = g e^-alpha;
//   3. layers back to front: weight gradients as outer products over the tile's samples
//      (dW[o][k] = sum_s delta[o][s] in[k][s]: both operands transposed through a per-wave LDS scratch,
//      K = 16 samples = four fp32 MFMAs per 16x16 weight tile), added into per-workgroup LDS
//      accumulators with ds_add_f32; delta propagation with the TRANSPOSED weight image
//      (rows = the layer's inputs, K = its outputs);  tanh' = 4 r (1 - r).
// A workgroup owns one parameter row (context) and walks its tiles grid-stride; the accumulators leave
// LDS once, masked, by plain stores (per-context rows, one workgroup per row) or one atomic per
// parameter per workgroup.
#include "mfma_tile.h"
#include "support_math.h"
#include "tnf_common.h"

namespace tnf {

struct MafBLayout {  // group numbering shared by the folded image, the transposed image and the accumulators
    int UT, DT, L;
    __host__ __device__ int n0() const { return 2 * UT * DT; }
    __host__ __device__ int nh() const { return 2 * UT * UT; }
    __host__ __device__ int NWG() const { return n0() + (L - 1) * nh() + n0(); }
    __host__ __device__ int NBG() const { return (L - 1) * 2 * UT + 2 * DT; }
    // (net, out tile, in tile) of each layer
    __host__ __device__ int g0(int net, int ut, int mm) const { return (net * UT + ut) * DT + mm; }
    __host__ __device__ int gh(int l, int net, int uo, int ui) const { return n0() + l * nh() + (net * UT + uo) * UT + ui; }
    __host__ __device__ int g2(int net, int mo, int ui) const { return n0() + (L - 1) * nh() + (net * DT + mo) * UT + ui; }
    __host__ __device__ int bh(int l, int net, int uo) const { return l * 2 * UT + net * UT + uo; }
    __host__ __device__ int b2(int net, int mo) const { return (L - 1) * 2 * UT + net * DT + mo; }
    __host__ __device__ int fwd_floats() const { return NWG() * 256 + NBG() * 16; }
};

// folded forward image (same maths as build_maf_image of maf_mfma.hip) and the plain transposed image:
// group (net, out tile, in tile), lane (r, q): W*M [in = 16 it + r][out = 16 ot + 4q + j].
// The (layer, net, out tile) units are dealt round-robin to the workgroup's `nwaves` waves: with one
// workgroup per context the build is on the critical path of every context.
__device__ void build_maf_bwd_images(float* fimg, float* timg, const float* __restrict__ p0, const float* __restrict__ mk0,
                                     MafBLayout wl, int D, int U, int lane, int wave, int nwaves, int bf) {
    const int r = lane & 15, q = lane >> 4;
    float* fw = fimg + lane * 4;
    float* fb = fimg + wl.NWG() * 256 + q * 4;
    float* tw = timg + lane * 4;
    const bool bias_lane = r == 0;
    int unit = 0;
    const float* p = p0;
    const float* mk = mk0;
    for (int layer = 0; layer <= wl.L; ++layer) {
        const int din = layer == 0 ? D : U, dout = layer == wl.L ? D : U;
        const int IT = layer == 0 ? wl.DT : wl.UT, OT = layer == wl.L ? wl.DT : wl.UT;
        const float* w[2] = {p, p + din * dout};
        for (int net = 0; net < 2; ++net) {
            // scale of the folded weights: layer 0 feeds a tanh (c = 2 log2 e); later layers consume
            // r = (1 - tanh)/2 (factor -2) and feed a tanh (c) or the outputs (1 for mu, log2 e for alpha)
            const float outsc = layer == wl.L ? (net == 0 ? 1.f : kLog2e) : kTwoLog2e;
            const float wsc = layer == 0 ? outsc : -2.f * outsc;
            for (int ot = 0; ot < OT; ++ot, ++unit) {
                if (unit % nwaves != wave) continue;
                const int g_base = layer == 0 ? wl.g0(net, ot, 0) : (layer == wl.L ? wl.g2(net, ot, 0) : wl.gh(layer - 1, net, ot, 0));
                float csum = 0.f;
                for (int it = 0; it < IT; ++it) {
                    f4 vf, vt;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // folded: row = output unit 16 ot + r, K = input 16 it + 4q + j
                        const int o = 16 * ot + r, k = 16 * it + 4 * q + j;
                        const bool ok = k < din && o < dout;
                        const float raw = ld_sel(w[net], k * dout + o, ok) * ld_sel(mk, k * dout + o, ok);
                        csum += raw;
                        vf[j] = wsc * raw;
                        // transposed: row = input 16 it + r, K = output 16 ot + 4q + j
                        const int k2 = 16 * it + r, o2 = 16 * ot + 4 * q + j;
                        const bool ok2 = k2 < din && o2 < dout;
                        vt[j] = ld_sel(w[net], k2 * dout + o2, ok2) * ld_sel(mk, k2 * dout + o2, ok2);
                    }
                    *reinterpret_cast<f4*>(fw + (g_base + it) * 256) = rbf16_4(vf, bf);
                    *reinterpret_cast<f4*>(tw + (g_base + it) * 256) = rbf16_4(vt, bf);
                }
                if (layer > 0) {  // accumulator initial values: column sums (there are no biases in MAF)
                    csum = reduce_q(csum);
                    f4 bv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[j] = outsc * __shfl(csum, 4 * q + j);
                    const int bg = layer == wl.L ? wl.b2(net, ot) : wl.bh(layer - 1, net, ot);
                    if (bias_lane) *reinterpret_cast<f4*>(fb + bg * 16) = bv;
                }
            }
        }
        p += 2 * din * dout;
        mk += din * dout;
    }
}

// 16x16 transpose of an accumulator-layout tile through LDS (see coupling_bwd_mfma.hip)
__device__ __forceinline__ f4 maf_transpose(f4 v, float* scr, int lane) {
    const int s = lane & 15, q = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) scr[(4 * q + j) * 17 + s] = v[j];
    f4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = scr[s * 17 + 4 * i + q];
    return o;
}

struct MafBwdArgs {
    const float* z;
    const float* params;
    const float* masks;
    const float* g_zout;
    const float* g_ld;
    float* g_z;
    float* g_params;
    int64_t M, Mp, N, pstride, gpstride;
    int D, L, U, nacc;
    // fused NormFlow('AR').log_prob training (tnf_ar_flow_log_prob_bwd_f32): z is the flow's input; the kernel
    // applies ToInterval^-1 (iv, may be NULL) and the folded Affine^-1 . BatchNorm^-1 (pre: A | B per context)
    // itself, seeds the backward from g_lp (g_out = -g_lp out, g_ld = -g_lp), and reduces the fold gradients
    // dA = sum g x_pre, dB = sum g (g_fold: (Mp, 2, D)) and sum g_lp (glp_sum: (Mp)); g_zout / g_ld are unused
    // and g_z may be NULL.
    const float* pre;
    const float* iv;
    const float* g_lp;
    float* g_fold;
    float* glp_sum;
    // fused mode with one shared accumulator copy (nacc = 1): 32-bit fixed-point accumulators and ds_add_u32 instead of
    // ds_add_f32 (44 x cheaper, tools/lds_atomic_bench.hip); g_lp is pre-scaled by the power of two that brings
    // max |g_lp| (gmax, float bits) into [1, 2) and fx = 2^f leaves 2^13 per accumulated term inside int32.
    const unsigned* gmax;
    float fx;
    int bf16;  // != 0: the bf16 experiment (every MFMA operand rounded to bf16, mfma_tile.h rbf16)
};

constexpr int kMafLMax = 3;

template <int DT, int UT, bool VEC>
__global__ void __launch_bounds__(256)
maf_bwd_mfma_kernel(MafBwdArgs a, MafBLayout wl) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = a.D, U = a.U, L = wl.L;
    const int NWG = wl.NWG();
    float* fimg = lds;
    float* timg = fimg + wl.fwd_floats();
    // weight-gradient accumulators: nacc = 4 -> one private copy per wave, plain read-add-write (ds_add_f32 costs
    // ~190 cycles per wave-instruction on gfx950, tools/lds_atomic_bench.hip); nacc = 1 (the copies do not fit the
    // LDS) -> one shared copy and float atomics
    const int nacc = a.nacc;
    float* gacc = timg + NWG * 256;
    float* scr_all = gacc + nacc * NWG * 256;
    float* cst = scr_all + 4 * 2 * 272;  // fused mode: fold A | B (2 D), ToInterval constants (7 D), reduction slots (2 D + 1)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    float* scrA = scr_all + wave * 2 * 272;
    float* scrB = scrA + 272;
    const int64_t m = grid_m();
    if (m >= a.M) return;
    const int64_t mp = a.Mp == 1 ? 0 : m;
    for (int i = threadIdx.x; i < nacc * NWG * 256; i += 256) gacc[i] = 0.f;
    const bool fusedm = a.g_lp != nullptr;
    const bool has_iv = a.iv != nullptr;
    const bool fixedp = fusedm && a.gmax != nullptr;
    float sc = 1.f, isc = 1.f, amax = 0.f;
    if (fixedp) {
        const float gm = __uint_as_float(*a.gmax);
        if (gm > 0.f && gm < 3.0e38f) {
            int e;
            (void)frexpf(gm, &e);
            int k = 1 - e;
            k = k > 120 ? 120 : (k < -120 ? -120 : k);
            sc = ldexpf(1.f, k);
            isc = ldexpf(1.f, -k);
        }
    }
    if (fusedm) {
        for (int i = threadIdx.x; i < 2 * D; i += 256) cst[i] = a.pre[mp * 2 * D + i];
        if (has_iv)
            for (int i = threadIdx.x; i < 7 * D; i += 256) cst[2 * D + i] = a.iv[i];
        for (int i = threadIdx.x; i < 2 * D + 1; i += 256) cst[9 * D + i] = 0.f;
    }
    build_maf_bwd_images(fimg, timg, a.params + mp * a.pstride, a.masks, wl, D, U, lane, wave, 4, a.bf16);
    __syncthreads();

    const float* fsrc = fimg + lane * 4;
    const float* bsrc = fimg + NWG * 256 + q * 4;
    const float* tsrc = timg + lane * 4;
    float* gdst = gacc + (nacc > 1 ? wave * NWG * 256 : 0) + lane * 4;
    auto wgrp = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(fsrc + g * 256); };
    auto bgrp = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(bsrc + g * 16); };
    auto tgrp = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(tsrc + g * 256); };
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    auto gadd = [&](int g, f4 v) {
        if (nacc > 1) {
            f4* p = reinterpret_cast<f4*>(gdst + g * 256);
            *p = *p + v;
        } else if (fixedp) {
            int* p = reinterpret_cast<int*>(gdst + g * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = v[j] * a.fx;  // a plain VALU result, converted in place: the inline-asm rule of f16_tile.h
                amax = fmaxf(amax, fabsf(t));
                asm("v_cvt_rpi_i32_f32 %0, %0" : "+v"(t));
                atomicAdd(p + j, __builtin_bit_cast(int, t));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(gdst + g * 256 + j, v[j]);  // ds_add_f32
        }
    };

    const float* zb = a.z + m * a.N * D;
    const float* gob = fusedm ? nullptr : a.g_zout + m * a.N * D;
    const float* glb = (fusedm ? a.g_lp : a.g_ld) + m * a.N;
    float* gzb = a.g_z ? a.g_z + m * a.N * D : nullptr;
    f4 fdA[DT], fdB[DT];  // fused mode: this lane's share of dA, dB
#pragma unroll
    for (int mm = 0; mm < DT; ++mm) fdA[mm] = fdB[mm] = zero;
    float glp_acc = 0.f;

    const int64_t ntiles = (a.N + 15) >> 4;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row = tile * 16 + s;
        const bool row_ok = row < a.N;
        const int64_t rr = row_ok ? row : a.N - 1;
        f4 x[DT], g[DT], xpre[DT];
#pragma unroll
        for (int mm = 0; mm < DT; ++mm) {
            const int f0 = 16 * mm + 4 * q;
            if (VEC) {
                x[mm] = f0 < D ? *reinterpret_cast<const f4*>(zb + rr * D + f0) : zero;
                g[mm] = (!fusedm && f0 < D && row_ok) ? *reinterpret_cast<const f4*>(gob + rr * D + f0) : zero;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[mm][j] = ld_sel(zb + rr * D, f0 + j, f0 + j < D);
                    g[mm][j] = (!fusedm && row_ok) ? ld_sel(gob + rr * D, f0 + j, f0 + j < D) : 0.f;
                }
            }
            xpre[mm] = zero;
            if (fusedm) {  // the bijectors in front of the MAF: ToInterval^-1, then x A + B
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = f0 + j;
                    float v = x[mm][j];
                    if (f < D) {
                        if (has_iv) {
                            float o, l;
                            interval_fast<true>(v, cst + 2 * D, D, f, o, l);
                            v = o;
                        }
                        xpre[mm][j] = v;
                        v = __builtin_fmaf(v, cst[f], cst[D + f]);
                    }
                    x[mm][j] = v;
                }
            }
        }
        const float glv = row_ok ? sc * glb[rr] : 0.f;
        const float gl = fusedm ? -glv : glv;  // log_prob = base - (sum of forward log-dets)
        if (fusedm && q == 0) glp_acc += glv;
        asm volatile("" ::: "memory");

        // ---- 1. forward recompute, keeping r of every hidden level ----
        f4 r[kMafLMax][2][UT];
        const int bf = a.bf16;
        f4 xo[DT];  // the contraction operand (x itself, or its bf16 rounding)
#pragma unroll
        for (int mm = 0; mm < DT; ++mm) xo[mm] = rbf16_4(x[mm], bf);
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            f4 at = zero, as = zero;
#pragma unroll
            for (int mm = 0; mm < DT; ++mm) {
                const f4 wt = wgrp(wl.g0(0, ut, mm)), ws = wgrp(wl.g0(1, ut, mm));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    at = mfma4(wt[j], xo[mm][j], at);
                    as = mfma4(ws[j], xo[mm][j], as);
                }
            }
            r[0][0][ut] = rbf16_4(sig2_4(at), bf);
            r[0][1][ut] = rbf16_4(sig2_4(as), bf);
        }
#pragma unroll
        for (int l = 1; l < kMafLMax; ++l) {
            if (l < L) {
#pragma unroll
                for (int uo = 0; uo < UT; ++uo) {
                    f4 at = bgrp(wl.bh(l - 1, 0, uo)), as = bgrp(wl.bh(l - 1, 1, uo));
#pragma unroll
                    for (int ui = 0; ui < UT; ++ui) {
                        const f4 wt = wgrp(wl.gh(l - 1, 0, uo, ui)), ws = wgrp(wl.gh(l - 1, 1, uo, ui));
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            at = mfma4(wt[j], r[l - 1][0][ui][j], at);
                            as = mfma4(ws[j], r[l - 1][1][ui][j], as);
                        }
                    }
                    r[l][0][uo] = rbf16_4(sig2_4(at), bf);
                    r[l][1][uo] = rbf16_4(sig2_4(as), bf);
                }
            }
        }
        // ---- 2. outputs and their deltas ----
        f4 dlt[2][DT];  // deltas of the current level's outputs: [net][tile]
#pragma unroll
        for (int mo = 0; mo < DT; ++mo) {
            f4 mu = bgrp(wl.b2(0, mo)), al2 = bgrp(wl.b2(1, mo));
#pragma unroll
            for (int l = 0; l < kMafLMax; ++l) {
                if (l == L - 1) {
#pragma unroll
                    for (int ui = 0; ui < UT; ++ui) {
                        const f4 wt = wgrp(wl.g2(0, mo, ui)), ws = wgrp(wl.g2(1, mo, ui));
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            mu = mfma4(wt[j], r[l][0][ui][j], mu);
                            al2 = mfma4(ws[j], r[l][1][ui][j], al2);
                        }
                    }
                }
            }
            f4 gz;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = __builtin_amdgcn_exp2f(-al2[j]);
                const float gup = fusedm ? -glv * ((x[mo][j] - mu[j]) * e) : g[mo][j];  // d(-|out|^2/2)/d out . g_lp
                const float ge = gup * e;
                gz[j] = ge;
                dlt[0][mo][j] = -ge;
                // padded features (f >= D): x = mu = 0 and g = 0, but g_ld must not leak into them
                dlt[1][mo][j] = (16 * mo + 4 * q + j < D) ? (-ge * (x[mo][j] - mu[j]) + gl) : 0.f;
            }
            g[mo] = gz;  // from here on g holds g_z (direct path); the nets' share is added at layer 0
            dlt[0][mo] = rbf16_4(dlt[0][mo], bf);
            dlt[1][mo] = rbf16_4(dlt[1][mo], bf);
        }

        // ---- 3. layers back to front ----
        // inputs of the level-`lev` layer in h = tanh form (true activations), level L = output layer
        auto act_h = [&](int lev, int net, int t) -> f4 {  // lev >= 1: hidden activations of level lev-1
            f4 v = zero;
#pragma unroll
            for (int l = 0; l < kMafLMax; ++l)
                if (l == lev - 1) v = r[l][net][t];
            return 1.f - 2.f * v;
        };
        {   // output layer (level L): inputs = hidden level L-1 (UT tiles), outputs = DT tiles
            f4 dprev[2][UT];
#pragma unroll
            for (int net = 0; net < 2; ++net) {
#pragma unroll
                for (int ui = 0; ui < UT; ++ui) {
                    const f4 hin = act_h(L, net, ui);
                    const f4 h_t = rbf16_4(maf_transpose(hin, scrB, lane), bf);
                    f4 acc = zero;
#pragma unroll
                    for (int mo = 0; mo < DT; ++mo) {
                        const f4 dt_ = maf_transpose(dlt[net][mo], scrA, lane);  // scrA holds one tile at a time
                        f4 dw = zero;
#pragma unroll
                        for (int i = 0; i < 4; ++i) dw = mfma4(dt_[i], h_t[i], dw);  // D[o][k]
                        gadd(wl.g2(net, mo, ui), dw);
                        const f4 wb = tgrp(wl.g2(net, mo, ui));
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc = mfma4(wb[j], dlt[net][mo][j], acc);
                    }
                    const f4 rv = (1.f - hin) * 0.5f;  // r
                    dprev[net][ui] = rbf16_4(acc * (4.f * rv * (1.f - rv)), bf);
                }
            }
            // hidden layers L-1 .. 1
#pragma unroll
            for (int lev = kMafLMax - 1; lev >= 1; --lev) {
                if (lev < L) {
                    f4 dnext[2][UT];
#pragma unroll
                    for (int net = 0; net < 2; ++net) {
#pragma unroll
                        for (int ui = 0; ui < UT; ++ui) {
                            const f4 hin = act_h(lev, net, ui);
                            const f4 h_t = rbf16_4(maf_transpose(hin, scrB, lane), bf);
                            f4 acc = zero;
#pragma unroll
                            for (int uo = 0; uo < UT; ++uo) {
                                const f4 dt_ = maf_transpose(dprev[net][uo], scrA, lane);
                                f4 dw = zero;
#pragma unroll
                                for (int i = 0; i < 4; ++i) dw = mfma4(dt_[i], h_t[i], dw);
                                gadd(wl.gh(lev - 1, net, uo, ui), dw);
                                const f4 wb = tgrp(wl.gh(lev - 1, net, uo, ui));
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc = mfma4(wb[j], dprev[net][uo][j], acc);
                            }
                            const f4 rv = (1.f - hin) * 0.5f;
                            dnext[net][ui] = rbf16_4(acc * (4.f * rv * (1.f - rv)), bf);
                        }
                    }
#pragma unroll
                    for (int net = 0; net < 2; ++net)
#pragma unroll
                        for (int u = 0; u < UT; ++u) dprev[net][u] = dnext[net][u];
                }
            }
            // layer 0: inputs = x (DT tiles), outputs = hidden level 0 (UT tiles)
#pragma unroll
            for (int mm = 0; mm < DT; ++mm) {
                const f4 x_t = maf_transpose(xo[mm], scrB, lane);
                f4 acc = zero;
#pragma unroll
                for (int net = 0; net < 2; ++net) {
#pragma unroll
                    for (int ut = 0; ut < UT; ++ut) {
                        const f4 dt_ = maf_transpose(dprev[net][ut], scrA, lane);
                        f4 dw = zero;
#pragma unroll
                        for (int i = 0; i < 4; ++i) dw = mfma4(dt_[i], x_t[i], dw);
                        gadd(wl.g0(net, ut, mm), dw);
                        const f4 wb = tgrp(wl.g0(net, ut, mm));
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc = mfma4(wb[j], dprev[net][ut][j], acc);
                    }
                }
                g[mm] += acc;
            }
        }
        if (fusedm) {
#pragma unroll
            for (int mm = 0; mm < DT; ++mm)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fdA[mm][j] = __builtin_fmaf(g[mm][j], xpre[mm][j], fdA[mm][j]);
                    fdB[mm][j] += g[mm][j];
                    const int f = 16 * mm + 4 * q + j;
                    if (gzb && f < D) g[mm][j] *= cst[f] * isc;  // g wrt the MAF input -> wrt the fold's input (no ToInterval here)
                }
        }
        if (row_ok && gzb) {
#pragma unroll
            for (int mm = 0; mm < DT; ++mm) {
                const int f0 = 16 * mm + 4 * q;
                if (VEC) {
                    if (f0 < D) *reinterpret_cast<f4*>(gzb + row * D + f0) = g[mm];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (f0 + j < D) gzb[row * D + f0 + j] = g[mm][j];
                }
            }
        }
    }

    if (fusedm) {  // fold gradients: 16 sample lanes -> LDS (once per wave) -> global
        float* red = cst + 9 * D;
#pragma unroll
        for (int mm = 0; mm < DT; ++mm)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float va = fdA[mm][j] * isc, vb = fdB[mm][j] * isc;
                for (int off = 8; off > 0; off >>= 1) {
                    va += __shfl_xor(va, off);
                    vb += __shfl_xor(vb, off);
                }
                const int f = 16 * mm + 4 * q + j;
                if (s == 0 && f < D) {
                    atomicAdd(red + f, va);
                    atomicAdd(red + D + f, vb);
                }
            }
        for (int off = 8; off > 0; off >>= 1) glp_acc += __shfl_xor(glp_acc, off);
        if (lane == 0) atomicAdd(red + 2 * D, glp_acc * isc);
        __syncthreads();
        const bool own_row = a.Mp > 1 && gridDim.x == 1;
        for (int i = threadIdx.x; i < 2 * D; i += 256) {
            if (own_row) a.g_fold[mp * 2 * D + i] = red[i];
            else atomicAdd(a.g_fold + mp * 2 * D + i, red[i]);
        }
        if (threadIdx.x == 0) {
            if (own_row) a.glp_sum[mp] = red[2 * D];
            else atomicAdd(a.glp_sum + mp, red[2 * D]);
        }
    }
    __syncthreads();
    // ---- flush: tile (out tile ot, in tile it) element (o = 16 ot + 4q + j, k = 16 it + (lane & 15)) ----
    {
        const bool own = a.Mp > 1 && gridDim.x == 1;
        float poison = 0.f, unfx = 1.f;
        if (fixedp) {  // a term beyond the budget may have wrapped an accumulator: poison instead of returning it
            for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
            float* pr = scr_all;
            if (lane == 0) pr[wave] = amax;
            __syncthreads();
            amax = fmaxf(fmaxf(pr[0], pr[1]), fmaxf(pr[2], pr[3]));
            const float adds = (float)((ntiles + gridDim.x - 1) / gridDim.x);
            if (amax * adds >= 2147483648.f) poison = __builtin_nanf("");
            unfx = isc / a.fx;
        }
        float* gp = a.g_params + mp * a.gpstride;
        const float* mk = a.masks;
        int64_t off = 0, moff = 0;
        for (int layer = 0; layer <= L; ++layer) {
            const int din = layer == 0 ? D : U, dout = layer == L ? D : U;
            const int IT = layer == 0 ? DT : UT, OT = layer == L ? DT : UT;
            const int64_t nw = (int64_t)din * dout;
            for (int t = wave; t < 2 * OT * IT; t += 4) {
                const int net = t / (OT * IT), rem = t - net * OT * IT, ot = rem / IT, it = rem - ot * IT;
                const int gidx = layer == 0 ? wl.g0(net, ot, it) : (layer == L ? wl.g2(net, ot, it) : wl.gh(layer - 1, net, ot, it));
                f4 v = *reinterpret_cast<const f4*>(gacc + (gidx * 64 + lane) * 4);
                for (int c = 1; c < nacc; ++c) v += *reinterpret_cast<const f4*>(gacc + c * NWG * 256 + (gidx * 64 + lane) * 4);
                if (fixedp) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (float)__float_as_int(v[j]) * unfx + poison;
                }
                const int k = 16 * it + s;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = 16 * ot + 4 * q + j;
                    if (k < din && o < dout) {
                        const int64_t idx = (int64_t)k * dout + o;
                        const float val = v[j] * mk[moff + idx];
                        float* dst = gp + off + net * nw + idx;
                        if (own) *dst = val;
                        else if (val != 0.f) atomicAdd(dst, val);
                    }
                }
            }
            off += 2 * nw;
            moff += nw;
        }
    }
}

static MafBLayout maf_blayout(int D, int L, int U) {
    MafBLayout wl;
    wl.UT = (U + 15) / 16;
    wl.DT = (D + 15) / 16;
    wl.L = L;
    return wl;
}

static size_t maf_bwd_smem(const MafBLayout& wl, int nacc = 1) {
    return (size_t)(wl.fwd_floats() + (1 + nacc) * wl.NWG() * 256 + 4 * 2 * 272 + 11 * 16 * wl.DT + 4) * sizeof(float);
}

bool maf_bwd_mfma_supported(int D, int L, int U) {
    if (D < 1 || D > 32 || L < 1 || L > kMafLMax || U < 1 || U > 64) return false;
    return maf_bwd_smem(maf_blayout(D, L, U)) <= 156 * 1024;
}

template <int DT, int UT>
static int launch_maf_bwd_du(const MafBwdArgs& a, const MafBLayout& wl, dim3 grid, size_t smem, hipStream_t st) {
    const bool vec = (a.D % 4) == 0;
    if (vec) {
        auto k = maf_bwd_mfma_kernel<DT, UT, true>;
        if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return fail(TNF_ELAUNCH, "maf_bwd_mfma: cannot reserve %zu B of LDS", smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, a, wl);
    } else {
        auto k = maf_bwd_mfma_kernel<DT, UT, false>;
        if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return fail(TNF_ELAUNCH, "maf_bwd_mfma: cannot reserve %zu B of LDS", smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, a, wl);
    }
    return TNF_OK;
}

template <int DT>
static int launch_maf_bwd_d(const MafBwdArgs& a, const MafBLayout& wl, dim3 grid, size_t smem, hipStream_t st) {
    switch (wl.UT) {
        case 1: return launch_maf_bwd_du<DT, 1>(a, wl, grid, smem, st);
        case 2: return launch_maf_bwd_du<DT, 2>(a, wl, grid, smem, st);
        case 3: return launch_maf_bwd_du<DT, 3>(a, wl, grid, smem, st);
        default: return launch_maf_bwd_du<DT, 4>(a, wl, grid, smem, st);
    }
}

static int maf_bwd_nacc(int D, int L, int U) { return maf_bwd_smem(maf_blayout(D, L, U), 4) <= 156 * 1024 ? 4 : 1; }

__global__ void __launch_bounds__(256)
maf_gmax_kernel(const float* __restrict__ g, int64_t n, unsigned* __restrict__ out) {
    __shared__ float red[4];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(g[i]));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.f) atomicMax(out, __float_as_uint(m));
    }
}

static int launch_maf_bwd_args(MafBwdArgs& a, hipStream_t st) {
    const MafBLayout wl = maf_blayout(a.D, a.L, a.U);
    diag_count(TNF_DIAG_MAF_BWD_MFMA);
    const int nacc = maf_bwd_nacc(a.D, a.L, a.U);
    const size_t smem = maf_bwd_smem(wl, nacc);
    a.nacc = nacc;
    a.bf16 = g_operand_prec == 1;
    const int64_t ntiles = (a.N + 15) / 16;
    int64_t bx = (ntiles + 3) / 4;
    if (a.Mp > 1) bx = 1;  // one workgroup owns the context's gradient row: plain stores
    else if (bx > 512) bx = 512;
    const dim3 grid = grid_xm(bx, a.M);
    int rc = wl.DT == 1 ? launch_maf_bwd_d<1>(a, wl, grid, smem, st) : launch_maf_bwd_d<2>(a, wl, grid, smem, st);
    if (rc != TNF_OK) return rc;
    return check_launch("maf_bwd_mfma");
}

int launch_maf_backward_mfma(const float* z, const float* params, const float* masks, const float* g_zout,
                             const float* g_ld, float* g_z, float* g_params, int64_t M, int64_t Mp, int64_t N, int D,
                             int L, int U, int64_t pstride, int64_t gpstride, hipStream_t st) {
    if (!maf_bwd_mfma_supported(D, L, U))
        return fail(TNF_EUNSUPPORTED, "maf_bwd_mfma: no kernel for D=%d L=%d U=%d", D, L, U);
    if (N <= 0) return TNF_OK;
    MafBwdArgs a = {};
    a.z = z; a.params = params; a.masks = masks; a.g_zout = g_zout; a.g_ld = g_ld; a.g_z = g_z; a.g_params = g_params;
    a.M = M; a.Mp = Mp; a.N = N; a.pstride = pstride; a.gpstride = gpstride; a.D = D; a.L = L; a.U = U;
    return launch_maf_bwd_args(a, st);
}

// [ToInterval^-1,] Affine^-1, BatchNorm^-1, MAF^-1, base density: gradient of log_prob w.r.t. the parameter rows.
//   fold (Mp, 2, D): A | B of launch_ar_fold(inverse);  g_fold (Mp, 2, D) + glp_sum (Mp): zeroed by the caller.
__global__ void __launch_bounds__(64)
ar_fold_backward_kernel(const float* __restrict__ params, int64_t pstride, int64_t p_maf, const float* __restrict__ fold,
                        const float* __restrict__ g_fold, const float* __restrict__ glp_sum, float* __restrict__ g_params,
                        int64_t gpstride, int D, int accumulate) {
    const int64_t m = blockIdx.x;
    const float* ap = params + m * pstride + p_maf;
    float* gp = g_params + m * gpstride + p_maf;
    const float sg = glp_sum[m];
    for (int d = threadIdx.x; d < D; d += 64) {
        const float A = fold[m * 2 * D + d], sh = ap[D + d];
        const float dA = g_fold[m * 2 * D + d], dB = g_fold[m * 2 * D + D + d];
        // A = alpha_bn e^-a, B = mean_bn - shift A;  log_prob also carries -sum a
        const float ga = -A * dA + sh * A * dB - sg, gs = -A * dB;
        if (accumulate) {
            atomicAdd(gp + d, ga);
            atomicAdd(gp + D + d, gs);
        } else {
            gp[d] = ga;
            gp[D + d] = gs;
        }
    }
}

int launch_ar_flow_backward(const float* z, const float* params, const float* masks, const float* fold,
                            const float* interval_consts, const float* g_lp, float* g_params, float* g_fold,
                            float* glp_sum, int64_t M, int64_t Mp, int64_t N, int D, int L, int U, int64_t pstride,
                            int64_t gpstride, hipStream_t st) {
    if (!maf_bwd_mfma_supported(D, L, U))
        return fail(TNF_EUNSUPPORTED, "ar_flow_backward: no kernel for D=%d L=%d U=%d", D, L, U);
    if (N <= 0) return TNF_OK;
    if (hipMemsetAsync(g_fold, 0, (size_t)(Mp * 2 * D + Mp + 1) * sizeof(float), st) != hipSuccess)
        return fail(TNF_ELAUNCH, "ar_flow_backward: memset failed");
    MafBwdArgs a = {};
    a.z = z; a.params = params; a.masks = masks; a.g_params = g_params;
    a.M = M; a.Mp = Mp; a.N = N; a.pstride = pstride; a.gpstride = gpstride; a.D = D; a.L = L; a.U = U;
    a.pre = fold; a.iv = interval_consts; a.g_lp = g_lp; a.g_fold = g_fold; a.glp_sum = glp_sum;
    if (maf_bwd_nacc(D, L, U) == 1) {  // shared accumulators: fixed point, scaled by the largest upstream gradient
        unsigned* gmax = reinterpret_cast<unsigned*>(glp_sum + Mp);
        const int64_t n = M * N;
        int64_t blocks = (n + 255) / 256;
        if (blocks > 256) blocks = 256;
        hipLaunchKernelGGL(maf_gmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g_lp, n, gmax);
        const int64_t ntiles = (N + 15) / 16;
        int64_t bx = (ntiles + 3) / 4;
        if (Mp > 1) bx = 1;
        else if (bx > 512) bx = 512;
        const int64_t adds = (ntiles + bx - 1) / bx;  // tiles (= terms per accumulator) per workgroup
        int fbits = 31 - 13;
        for (int64_t v = 1; v < adds; v <<= 1) --fbits;
        if (fbits < 0) fbits = 0;
        a.gmax = gmax;
        a.fx = ldexpf(1.f, fbits);
    }
    int rc = launch_maf_bwd_args(a, st);
    if (rc != TNF_OK) return rc;
    const int64_t p_maf = 2 * ((int64_t)D * U + (int64_t)(L - 1) * U * U + (int64_t)U * D) + 0;
    hipLaunchKernelGGL(ar_fold_backward_kernel, dim3((unsigned)Mp), dim3(64), 0, st, params, pstride, p_maf, fold, g_fold,
                       glp_sum, g_params, gpstride, D, 1);
    return check_launch("ar_fold_backward");
}

}  // namespace tnf
