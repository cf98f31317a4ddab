// Conditional flow, one sample per context (the SNPE / APT layout: cde.log_prob(z[:, None, :], x),
// conditional_density_estimator.py:101-104 with N = 1): the last Linear of param_net and the whole
// coupling flow in ONE kernel, so the (M, D_params) parameter tensor -- 82 KB per context at D = 64,
// which IS the traffic of the reference's formulation -- never exists in HBM.
//
//   params[m, :] = W_last . h[m, :] + b_last          (h = output of param_net's last activation)
//   log_prob[m]  = NormFlow.log_prob(z[m], params[m]) (density_estimator.py:390-416)
//
// Work decomposition.  The flow parameters of a context are used exactly once, so the kernel
// generates them tile by tile and consumes them on the spot:
//   * A "tile" is 16 consecutive entries of the parameter row: one input unit k of an MLP layer
//     towards 16 output units (W is row-major [in][out]), 16 biases, or 16 Affine alphas / shifts.
//   * P[16 params x 16 contexts] = Wtile[16 x H] . h^T[H x 16] on the matrix pipe, fp32-accurate
//     through three split-f16 MFMAs per K = 32 step (see flow_fused_f16.hip), b_last as the
//     accumulator's initial value.  Output layout: lane (c = lane & 15, q = lane >> 4) holds
//     params 4q..4q+3 of context c -- i.e. output units 4q..4q+3 of the MLP layer.
//   * Consumption is 4 FMAs per tile: acc[o] += x[c][k] * P[k][o] with the scalar x[c][k] read from
//     the wave's LDS copy of the context's state.  No cross-lane traffic at all.
// The tile sequence is the flow's inverse pass (stages S-1..0: Affine^-1 + BN^-1, RealNVP(low)^-1,
// BN^-1, RealNVP(up)^-1); an image kernel lays W_last / b_last out in exactly that order, split
// into f16 halves, and the main kernel streams it through LDS (double-buffered chunks shared by the
// workgroup's waves).  Each wave owns 16*BT contexts; the matrix pipe is the binding unit
// (2 * H * D_params * 3 f16 flops per context), everything else rides under it.
#include "cond_tile.h"

namespace tnf {

// ---------------------------------------------------------------------------
// Prep 1: largest magnitude of W_last / b_last (operand scaling keeps the f16 halves normal).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
cond_absmax_kernel(const float* __restrict__ W, const float* __restrict__ b, int64_t P, int H, int64_t ldw,
                   unsigned* __restrict__ maxbits) {
    float mx = 0.f;
    const int64_t n = P * H;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / H;
        mx = fmaxf(mx, fabsf(W[r * ldw + (i - r * H)]));
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256)
        mx = fmaxf(mx, fabsf(b[i]));
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) atomicMax(maxbits, __float_as_uint(mx));  // non-negative floats order like uints
}

// Prep 2: the operand image.  Tile t = [ks][part hi/lo][lane] 16-byte groups (the lane's 8 f16 of row
// lane&15, k = 32ks + 8(lane>>4) .. +7) followed by the 16 scaled biases (fp32).
template <int KS>
__global__ void __launch_bounds__(256)
cond_image_kernel(const float* __restrict__ W, const float* __restrict__ b, int64_t ldw, CondCfg cfg,
                  const unsigned* __restrict__ maxbits, float* __restrict__ inv_scale, u4* __restrict__ image,
                  int backward_order) {
    constexpr int TILE_U4 = KS * 128 + 4;
    const float scale = cond_scale(*maxbits);
    if (blockIdx.x == 0 && threadIdx.x == 0) *inv_scale = 1.f / scale;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < cfg.T; t += (int64_t)gridDim.x * 4) {
        int64_t base;
        int count;
        if (backward_order == 2) cond_tile_desc_fwd(cfg, t, base, count);  // the sampling program
        else if (backward_order) cond_tile_desc_bwd(cfg, t, base, count);
        else cond_tile_desc(cfg, t, base, count);
        const bool ok = r < count;
        const float* wr = W + (base + (ok ? r : 0)) * ldw;
        u4* tp = image + t * TILE_U4;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f4 v0 = *reinterpret_cast<const f4*>(wr + 32 * ks + 8 * q);
            f4 v1 = *reinterpret_cast<const f4*>(wr + 32 * ks + 8 * q + 4);
            const float sc = ok ? scale : 0.f;
            v0 *= sc;
            v1 *= sc;
            h8 hi, lo;
            csplit8(v0, v1, hi, lo);
            tp[(ks * 2 + 0) * 64 + lane] = __builtin_bit_cast(u4, hi);
            tp[(ks * 2 + 1) * 64 + lane] = __builtin_bit_cast(u4, lo);
        }
        if (q == 0) reinterpret_cast<float*>(tp + KS * 128)[r] = ok ? b[base + r] * scale : 0.f;
    }
}

// ---------------------------------------------------------------------------
// Main kernel
// ---------------------------------------------------------------------------
struct CondArgs {
    const float* z;         // (M, D)
    const float* h;         // (M, ldh), first H columns used
    const u4* image;
    const float* inv_scale;
    const float* bn_mean;   // (2S, D)
    const float* bn_alpha;
    float* log_prob;        // (M)
    float* z0;              // (M, D) or NULL
    float* sum_log_det;     // (M) or NULL
    // training mode (SAVE): what the backward kernels need, slot-major so that a wave's contexts are
    // contiguous.  acts_aff [S][M][D]: state after Affine^-1 + BN^-1 of each stage (compute order);
    // acts_c [2S][M][CR]: per coupling layer [x1 (D/2) | x2_out (D/2) | s (D/2) | (h_t 16, h_s 16) x L]
    float* acts_aff;
    float* acts_c;
    int64_t M, ldh, T;
    int S, L, U;
};

// FWD = true: the SAMPLING direction (cde(x, N = 1) with frozen statistics, conditional_density_estimator.py:93-99 over
// density_estimator.py:374-388): a.z holds the base draws, the tile stream is the sampling program (cond_tile_desc_fwd),
// every bijector runs forwards; a.z0 receives the samples, a.sum_log_det the forward log-dets, a.log_prob is unused.
template <int DT, int KS, int BT, int NW, bool SAVE, bool FWD = false>
__global__ void __launch_bounds__(64 * NW)
cond_flow_kernel(CondArgs a) {
    static_assert(!(SAVE && FWD), "the training forward is the inverse pass");
    constexpr int D = 16 * DT, Hd = D / 2, HT = DT / 2;
    constexpr int ZS = D + 4, HS = 20, CT = 16 * BT;
    typedef TileStream<KS * 128 + 4, kCondG<KS>, NW, kCondNS<KS, BT, NW>> Stream;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u4* stage = reinterpret_cast<u4*>(smem_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    float* wl = reinterpret_cast<float*>(stage + Stream::LDS_U4) + wave * (CT * (ZS + 2 * HS));
    float* zb = wl;                 // [CT][ZS]  the contexts' running state
    float* hbt = wl + CT * ZS;      // [CT][HS]  hidden activations of the t net
    float* hbs = hbt + CT * HS;     // [CT][HS]  ... of the s net

    const int64_t m0 = ((int64_t)blockIdx.x * NW + wave) * CT;
    // ---- the wave's contexts: z rows into LDS (coalesced), h rows into MFMA B operands ----
#pragma unroll
    for (int i = 0; i < CT * (D / 4) / 64; ++i) {
        const int idx = lane + 64 * i;
        const int row = idx / (D / 4), c4 = idx - row * (D / 4);
        int64_t m = m0 + row;
        m = m < a.M ? m : a.M - 1;
        *reinterpret_cast<f4*>(zb + row * ZS + 4 * c4) = *reinterpret_cast<const f4*>(a.z + m * D + 4 * c4);
    }
    h8 bh[BT][KS], bl[BT][KS];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
        int64_t m = m0 + bt * 16 + c;
        m = m < a.M ? m : a.M - 1;
        const float* hr = a.h + m * a.ldh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            csplit8(*reinterpret_cast<const f4*>(hr + 32 * ks + 8 * q),
                    *reinterpret_cast<const f4*>(hr + 32 * ks + 8 * q + 4), bh[bt][ks], bl[bt][ks]);
    }
    const float inv = *a.inv_scale;
    TilePipe<KS, BT, Stream> pipe;
    pipe.init(a.image, stage, a.T, lane);

    float ld[BT];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) ld[bt] = 0.f;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    const int U = a.U;
    const int CR = 3 * Hd + 32 * a.L;  // floats per saved coupling record
    bool live[BT];                     // context exists (the tail wave carries clamped duplicates)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) live[bt] = m0 + bt * 16 + c < a.M;

    // one RealNVP layer, inverse direction (bijectors.py:183-206): z2 <- (z2 - t(z1)) / exp(s(z1))
    auto coupling = [&](int cond_off, int tr_off, int slot) {
        // saved record of this layer for context (bt, c); only dereferenced when SAVE
        auto rec = [&](int bt) -> float* { return a.acts_c + ((int64_t)slot * a.M + m0 + bt * 16 + c) * CR; };
        f4 at[BT], as[BT], Pt[BT], Ps[BT];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) at[bt] = as[bt] = zero;
        // Software pipeline: the FMAs that consume a tile's result are placed in the shadow of the NEXT
        // tile's MFMAs (t results under the s tile's, s results under the next t tile's) -- with two waves
        // per SIMD in barrier lock-step there is nobody else to fill the matrix pipe meanwhile.
        float xp[BT];  // scaled input of the pending s-net tile
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            xp[bt] = 0.f;
            Ps[bt] = zero;
        }
        for (int k = 0; k < Hd; ++k) {
            float x[BT];
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) x[bt] = zb[(bt * 16 + c) * ZS + cond_off + k] * inv;
            pipe.gemm0(lane, bh, bl, Pt);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) as[bt] += xp[bt] * Ps[bt];
            pipe.gemm1(lane, bh, bl, Ps);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                at[bt] += x[bt] * Pt[bt];
                xp[bt] = x[bt];
            }
            pipe.refill1(lane);
        }
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) as[bt] += xp[bt] * Ps[bt];
        pipe.gemm0(lane, bh, bl, Pt);
        pipe.gemm1(lane, bh, bl, Ps);
        pipe.refill1(lane);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            at[bt] += inv * Pt[bt];
            as[bt] += inv * Ps[bt];
            f4 ht4, hs4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ht4[j] = fast_tanh(at[bt][j]);
                hs4[j] = fast_tanh(as[bt][j]);
            }
            *reinterpret_cast<f4*>(hbt + (bt * 16 + c) * HS + 4 * q) = ht4;
            *reinterpret_cast<f4*>(hbs + (bt * 16 + c) * HS + 4 * q) = hs4;
            if (SAVE && live[bt]) {
                float* r = rec(bt);
#pragma unroll
                for (int t = 0; t < HT; ++t)  // conditioning half, unchanged by this layer
                    *reinterpret_cast<f4*>(r + 16 * t + 4 * q) =
                        *reinterpret_cast<const f4*>(zb + (bt * 16 + c) * ZS + cond_off + 16 * t + 4 * q);
                *reinterpret_cast<f4*>(r + 3 * Hd + 4 * q) = ht4;
                *reinterpret_cast<f4*>(r + 3 * Hd + 16 + 4 * q) = hs4;
            }
        }
        for (int l = 1; l < a.L; ++l) {
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) at[bt] = as[bt] = zero;
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                xp[bt] = 0.f;
                Ps[bt] = zero;
            }
            for (int k = 0; k < U; ++k) {
                float xt[BT], xs[BT];
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    xt[bt] = hbt[(bt * 16 + c) * HS + k] * inv;
                    xs[bt] = hbs[(bt * 16 + c) * HS + k] * inv;
                }
                pipe.gemm0(lane, bh, bl, Pt);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) as[bt] += xp[bt] * Ps[bt];
                pipe.gemm1(lane, bh, bl, Ps);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    at[bt] += xt[bt] * Pt[bt];
                    xp[bt] = xs[bt];
                }
                pipe.refill1(lane);
            }
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) as[bt] += xp[bt] * Ps[bt];
            pipe.gemm0(lane, bh, bl, Pt);
            pipe.gemm1(lane, bh, bl, Ps);
            pipe.refill1(lane);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                at[bt] += inv * Pt[bt];
                as[bt] += inv * Ps[bt];
                f4 ht4, hs4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ht4[j] = fast_tanh(at[bt][j]);
                    hs4[j] = fast_tanh(as[bt][j]);
                }
                *reinterpret_cast<f4*>(hbt + (bt * 16 + c) * HS + 4 * q) = ht4;
                *reinterpret_cast<f4*>(hbs + (bt * 16 + c) * HS + 4 * q) = hs4;
                if (SAVE && live[bt]) {
                    float* r = rec(bt) + 3 * Hd + 32 * l;
                    *reinterpret_cast<f4*>(r + 4 * q) = ht4;
                    *reinterpret_cast<f4*>(r + 16 + 4 * q) = hs4;
                }
            }
        }
        f4 ot[BT][HT], os[BT][HT];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
#pragma unroll
            for (int o = 0; o < HT; ++o) ot[bt][o] = os[bt][o] = zero;
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            xp[bt] = 0.f;
            Ps[bt] = zero;
        }
        for (int k = 0; k < U; ++k) {
            float xt[BT], xs[BT];
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                xt[bt] = hbt[(bt * 16 + c) * HS + k] * inv;
                xs[bt] = hbs[(bt * 16 + c) * HS + k] * inv;
            }
#pragma unroll
            for (int o = 0; o < HT; ++o) {
                pipe.gemm0(lane, bh, bl, Pt);
                if (o > 0) {
#pragma unroll
                    for (int bt = 0; bt < BT; ++bt) os[bt][o - 1] += xs[bt] * Ps[bt];
                } else {
#pragma unroll
                    for (int bt = 0; bt < BT; ++bt) os[bt][HT - 1] += xp[bt] * Ps[bt];  // previous k's last s tile
                }
                pipe.gemm1(lane, bh, bl, Ps);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) ot[bt][o] += xt[bt] * Pt[bt];
                pipe.refill1(lane);
            }
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) xp[bt] = xs[bt];
        }
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) os[bt][HT - 1] += xp[bt] * Ps[bt];
#pragma unroll
        for (int o = 0; o < HT; ++o) {
            pipe.gemm0(lane, bh, bl, Pt);
            pipe.gemm1(lane, bh, bl, Ps);
            pipe.refill1(lane);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                const f4 t4 = ot[bt][o] + inv * Pt[bt];
                const f4 s4 = os[bt][o] + inv * Ps[bt];
                f4* zp = reinterpret_cast<f4*>(zb + (bt * 16 + c) * ZS + tr_off + 16 * o + 4 * q);
                f4 zv = *zp;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (FWD) zv[j] = __builtin_fmaf(zv[j], fast_exp(s4[j]), t4[j]);  // bijectors.py:172
                    else zv[j] = (zv[j] - t4[j]) * fast_exp(-s4[j]);
                    ld[bt] += s4[j];
                }
                *zp = zv;
                if (SAVE && live[bt]) {
                    float* r = rec(bt);
                    *reinterpret_cast<f4*>(r + Hd + 16 * o + 4 * q) = zv;
                    *reinterpret_cast<f4*>(r + 2 * Hd + 16 * o + 4 * q) = s4;
                }
            }
        }
    };

    // BatchNorm with cached statistics, forwards (bijectors.py:397-399): (z - mean) / alpha, log_det = -sum log alpha
    auto bn_forward = [&](int layer) {
        const float* bnA = a.bn_alpha + (int64_t)layer * D;
        const float* bnM = a.bn_mean + (int64_t)layer * D;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const f4 al = *reinterpret_cast<const f4*>(bnA + 16 * t + 4 * q);
            const f4 mu = *reinterpret_cast<const f4*>(bnM + 16 * t + 4 * q);
            float lal = 0.f;
            f4 ia;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lal += fast_log(al[j]);
                ia[j] = 1.f / al[j];
            }
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                f4* zp = reinterpret_cast<f4*>(zb + (bt * 16 + c) * ZS + 16 * t + 4 * q);
                *zp = (*zp - mu) * ia;
                ld[bt] -= lal;
            }
        }
    };
    if constexpr (FWD) {
        for (int st = 0; st < a.S; ++st) {
            coupling(0, Hd, 2 * st);      // RealNVP(transform_upper=True): conditions on the lower half
            bn_forward(2 * st);
            coupling(Hd, 0, 2 * st + 1);  // RealNVP(transform_upper=False)
            bn_forward(2 * st + 1);
            f4 Pa[BT], Psh[BT];           // Affine (bijectors.py:277-298): z e^alpha + shift, log_det = sum alpha
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                pipe.gemm0(lane, bh, bl, Pa);
                pipe.gemm1(lane, bh, bl, Psh);
                pipe.refill1(lane);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    f4* zp = reinterpret_cast<f4*>(zb + (bt * 16 + c) * ZS + 16 * t + 4 * q);
                    f4 zv = *zp;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float aa = Pa[bt][j] * inv, sh = Psh[bt][j] * inv;
                        zv[j] = __builtin_fmaf(zv[j], fast_exp(aa), sh);
                        ld[bt] += aa;
                    }
                    *zp = zv;
                }
            }
        }
    }
    for (int si = 0; si < (FWD ? 0 : a.S); ++si) {
        const int stg_i = a.S - 1 - si;
        {   // Affine^-1 (bijectors.py:300-315) then BatchNorm^-1 (:420-426) of layer 2*stage+1
            const float* bnA = a.bn_alpha + (int64_t)(2 * stg_i + 1) * D;
            const float* bnM = a.bn_mean + (int64_t)(2 * stg_i + 1) * D;
            f4 Pa[BT], Psh[BT];
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                pipe.gemm0(lane, bh, bl, Pa);
                pipe.gemm1(lane, bh, bl, Psh);
                pipe.refill1(lane);
                const f4 al = *reinterpret_cast<const f4*>(bnA + 16 * t + 4 * q);
                const f4 mu = *reinterpret_cast<const f4*>(bnM + 16 * t + 4 * q);
                float lal = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) lal += fast_log(al[j]);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    f4* zp = reinterpret_cast<f4*>(zb + (bt * 16 + c) * ZS + 16 * t + 4 * q);
                    f4 zv = *zp;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float aa = Pa[bt][j] * inv, sh = Psh[bt][j] * inv;
                        zv[j] = (zv[j] - sh) * fast_exp(-aa) * al[j] + mu[j];
                        ld[bt] += aa;
                    }
                    ld[bt] -= lal;
                    *zp = zv;
                    if (SAVE && live[bt])
                        *reinterpret_cast<f4*>(a.acts_aff + ((int64_t)si * a.M + m0 + bt * 16 + c) * D + 16 * t + 4 * q) = zv;
                }
            }
        }
        coupling(Hd, 0, 2 * si);  // RealNVP(transform_upper=False): conditions on the upper half
        {   // BatchNorm^-1 of layer 2*stage
            const float* bnA = a.bn_alpha + (int64_t)(2 * stg_i) * D;
            const float* bnM = a.bn_mean + (int64_t)(2 * stg_i) * D;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const f4 al = *reinterpret_cast<const f4*>(bnA + 16 * t + 4 * q);
                const f4 mu = *reinterpret_cast<const f4*>(bnM + 16 * t + 4 * q);
                float lal = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) lal += fast_log(al[j]);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    f4* zp = reinterpret_cast<f4*>(zb + (bt * 16 + c) * ZS + 16 * t + 4 * q);
                    *zp = *zp * al + mu;
                    ld[bt] -= lal;
                }
            }
        }
        coupling(0, Hd, 2 * si + 1);  // RealNVP(transform_upper=True)
    }

    // ---- log q = -|z0|^2/2 - D log sqrt(2 pi) - sum log_det (density_estimator.py:413-416) ----
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const f4 zv = *reinterpret_cast<const f4*>(zb + (bt * 16 + c) * ZS + 16 * t + 4 * q);
            ss += zv[0] * zv[0] + zv[1] * zv[1] + zv[2] * zv[2] + zv[3] * zv[3];
        }
        ss = reduce_q(ss);
        const float ldt = reduce_q(ld[bt]);
        const int64_t m = m0 + bt * 16 + c;
        if (q == 0 && m < a.M) {
            if (!FWD) a.log_prob[m] = -0.5f * ss - (float)D * 0.91893853320467274178f - ldt;
            if (a.sum_log_det) a.sum_log_det[m] = ldt;
        }
    }
    if (a.z0) {
#pragma unroll
        for (int i = 0; i < CT * (D / 4) / 64; ++i) {
            const int idx = lane + 64 * i;
            const int row = idx / (D / 4), c4 = idx - row * (D / 4);
            const int64_t m = m0 + row;
            if (m < a.M) *reinterpret_cast<f4*>(a.z0 + m * D + 4 * c4) = *reinterpret_cast<const f4*>(zb + row * ZS + 4 * c4);
        }
    }
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
bool cond_flow_supported(int D, int S, int L, int U, int H) {
    return (D == 32 || D == 64) && S >= 1 && L >= 1 && L <= 5 && U >= 1 && U <= 16 && (H == 32 || H == 64 || H == 128);
}

static int64_t cond_image_bytes(const CondCfg& c) { return c.T * (int64_t)(c.H / 32 * 128 + 4) * 16; }

int64_t cond_flow_workspace(int D, int S, int L, int U, int H) {
    return 256 + cond_image_bytes(cond_cfg(D, S, L, U, H));
}

template <int DT, int KS, int BT, int NW, bool SAVE, bool FWD = false>
static int launch_cond_variant(const CondArgs& a, hipStream_t st) {
    typedef TileStream<KS * 128 + 4, kCondG<KS>, NW, kCondNS<KS, BT, NW>> Stream;
    constexpr int D = 16 * DT;
    const size_t smem = (size_t)Stream::LDS_U4 * 16 + (size_t)NW * 16 * BT * (D + 4 + 40) * 4;
    auto k = cond_flow_kernel<DT, KS, BT, NW, SAVE, FWD>;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int64_t per_wg = (int64_t)NW * 16 * BT;
    const int64_t blocks = (a.M + per_wg - 1) / per_wg;
    if (blocks > 0x7fffffff) return fail(TNF_EUNSUPPORTED, "cond_flow: grid too large");
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * NW), smem, st, a);
    return check_launch("cond_flow");
}

thread_local int g_cond_variant = 0;  // testing hook: 0 = by M, 1 = (BT 1, 4 waves), 2 = (BT 1, 8 waves), 3 = (BT 2, 8 waves)

template <int DT, int KS>
static int launch_cond_dk(const CondArgs& a, hipStream_t st) {
    int v = g_cond_variant;
    if (v == 0) v = a.M >= 256 * 256 ? 3 : (a.M >= 256 * 128 ? 2 : 1);
    if (a.log_prob == nullptr) {  // the sampling direction: fewer shapes (compile time)
        if (v >= 3) return launch_cond_variant<DT, KS, 2, 8, false, true>(a, st);
        return launch_cond_variant<DT, KS, 1, 4, false, true>(a, st);
    }
    if (a.acts_c) {  // training forward: fewer shapes (compile time)
        if (v == 4) return launch_cond_variant<DT, KS, 4, 4, true>(a, st);
        if (v >= 2) return launch_cond_variant<DT, KS, 2, 8, true>(a, st);
        return launch_cond_variant<DT, KS, 1, 4, true>(a, st);
    }
    if (v == 4) return launch_cond_variant<DT, KS, 4, 4, false>(a, st);
    if (v == 3) return launch_cond_variant<DT, KS, 2, 8, false>(a, st);
    if (v == 2) return launch_cond_variant<DT, KS, 1, 8, false>(a, st);
    return launch_cond_variant<DT, KS, 1, 4, false>(a, st);
}

// absmax -> power-of-two scale (ws[0]: max bits, ws[1]: 1/scale) and the operand image in forward or
// backward tile order
int launch_cond_image(const float* W, const float* b, int64_t ldw, const CondCfg& cfg, void* ws, void* image,
                      int backward_order, hipStream_t st) {
    unsigned* maxbits = reinterpret_cast<unsigned*>(ws);
    float* inv_scale = reinterpret_cast<float*>(ws) + 1;
    u4* img = reinterpret_cast<u4*>(image);
    if (hipMemsetAsync(maxbits, 0, 8, st) != hipSuccess) return fail(TNF_ELAUNCH, "cond_flow: memset failed");
    hipLaunchKernelGGL(cond_absmax_kernel, dim3(256), dim3(256), 0, st, W, b, cfg.fl.total, cfg.H, ldw, maxbits);
    const unsigned ib = (unsigned)((cfg.T + 3) / 4);
    if (cfg.H == 32)
        hipLaunchKernelGGL(cond_image_kernel<1>, dim3(ib), dim3(256), 0, st, W, b, ldw, cfg, maxbits, inv_scale, img, backward_order);
    else if (cfg.H == 64)
        hipLaunchKernelGGL(cond_image_kernel<2>, dim3(ib), dim3(256), 0, st, W, b, ldw, cfg, maxbits, inv_scale, img, backward_order);
    else
        hipLaunchKernelGGL(cond_image_kernel<4>, dim3(ib), dim3(256), 0, st, W, b, ldw, cfg, maxbits, inv_scale, img, backward_order);
    return check_launch("cond_flow_prep");
}

int launch_cond_flow_log_prob(const float* z, const float* h, const float* W, const float* b, const float* bn_mean,
                              const float* bn_alpha, float* log_prob, float* z0, float* sum_log_det, float* acts,
                              int64_t M, int D, int S, int L, int U, int H, int64_t ldh, int64_t ldw, void* ws,
                              hipStream_t st) {
    const CondCfg cfg = cond_cfg(D, S, L, U, H);
    float* inv_scale = reinterpret_cast<float*>(ws) + 1;
    u4* image = reinterpret_cast<u4*>(reinterpret_cast<char*>(ws) + 256);
    int rc = launch_cond_image(W, b, ldw, cfg, ws, image, 0, st);
    if (rc) return rc;
    CondArgs a;
    a.z = z; a.h = h; a.image = image; a.inv_scale = inv_scale; a.bn_mean = bn_mean; a.bn_alpha = bn_alpha;
    a.log_prob = log_prob; a.z0 = z0; a.sum_log_det = sum_log_det;
    a.acts_aff = acts;
    a.acts_c = acts ? acts + (int64_t)S * M * D : nullptr;
    a.M = M; a.ldh = ldh; a.T = cfg.T; a.S = S; a.L = L; a.U = U;
    if (D == 64) {
        if (H == 32) return launch_cond_dk<4, 1>(a, st);
        if (H == 64) return launch_cond_dk<4, 2>(a, st);
        return launch_cond_dk<4, 4>(a, st);
    }
    if (H == 32) return launch_cond_dk<2, 1>(a, st);
    if (H == 64) return launch_cond_dk<2, 2>(a, st);
    return launch_cond_dk<2, 4>(a, st);
}

// ConditionalDensityEstimator.__call__(x, N = 1) with frozen statistics: omega (M, D) base draws -> z_out (M, D),
// sum_log_det (M); the same kernel with the sampling program.
int launch_cond_flow_forward(const float* omega, const float* h, const float* W, const float* b, const float* bn_mean,
                             const float* bn_alpha, float* z_out, float* sum_log_det, int64_t M, int D, int S, int L, int U,
                             int H, int64_t ldh, int64_t ldw, void* ws, hipStream_t st) {
    const CondCfg cfg = cond_cfg(D, S, L, U, H);
    float* inv_scale = reinterpret_cast<float*>(ws) + 1;
    u4* image = reinterpret_cast<u4*>(reinterpret_cast<char*>(ws) + 256);
    int rc = launch_cond_image(W, b, ldw, cfg, ws, image, 2, st);
    if (rc) return rc;
    CondArgs a;
    a.z = omega; a.h = h; a.image = image; a.inv_scale = inv_scale; a.bn_mean = bn_mean; a.bn_alpha = bn_alpha;
    a.log_prob = nullptr; a.z0 = z_out; a.sum_log_det = sum_log_det;
    a.acts_aff = nullptr;
    a.acts_c = nullptr;
    a.M = M; a.ldh = ldh; a.T = cfg.T; a.S = S; a.L = L; a.U = U;
    if (D == 64) {
        if (H == 32) return launch_cond_dk<4, 1>(a, st);
        if (H == 64) return launch_cond_dk<4, 2>(a, st);
        return launch_cond_dk<4, 4>(a, st);
    }
    if (H == 32) return launch_cond_dk<2, 1>(a, st);
    if (H == 64) return launch_cond_dk<2, 2>(a, st);
    return launch_cond_dk<2, 4>(a, st);
}

}  // namespace tnf
