// Backward of the fused conditioner + flow (cond_flow.hip), for SNPE / APT training:
//   loss -> g_log_prob (M)  ==>  g_W_last (D_params, H), g_b_last (D_params), g_h (M, H)  [, g_z (M, D)]
// with params[m] = W_last h[m] + b_last never materialised, and neither their gradient
// gP[m, (k,o)] = x[m,k] * delta[m,o] (82 KB per context each).  Two kernels:
//
// 1. cond_flow_bwd_kernel walks the flow back (stages 0..S-1: RealNVP(up), BN, RealNVP(low), Affine) for
//    16*BT contexts per wave.  The per-context weights are regenerated tile by tile exactly as in the
//    forward (P = Wtile . h^T, split-f16 MFMA) and used once, for delta propagation (gx[k] = sum_o P[k,o]
//    delta[o]).  In the same sweep the conditioner gradient g_h[m,:] = sum_p gP[m,p] W[p,:] is
//    accumulated on the matrix pipe from the TRANSPOSED operand image (rows = hidden unit j, K = the 32
//    parameters of a [t net, s net] tile pair) against B = x_k * delta, built in registers.
//    Activations come from the records the training forward saved (CondArgs::acts_*); the deltas of
//    every layer are written out (6 KB per context) for kernel 2.
// 2. cond_gw_kernel: g_W[(k,o), j] = sum_m x[m,k] delta[m,o] h[m,j] -- a reduction over contexts, so it is
//    parallelised over parameter tiles (one workgroup = one (layer, net, 16-output tile) x a slice of the
//    contexts; its waves share the staged x / delta / h chunk and each owns a few input units k), with
//    no atomics beyond the final few-way merge of the context slices.
// MFMA work: kernel 1 = 2x the forward, kernel 2 = 1x.
#include <type_traits>

#include "cond_tile.h"

namespace tnf {

// ---------------------------------------------------------------------------
// Transposed image, backward tile order, per PAIR of tiles: [jt < H/16][part hi/lo][lane] 16-byte groups;
// lane (j = lane & 15, q = lane >> 4) holds W[param(8q + e)][16 jt + j], e = 0..7, where params 0..15
// are the rows of the pair's first tile and 16..31 those of the second.
// ---------------------------------------------------------------------------
template <int KS>
__global__ void __launch_bounds__(256)
cond_timage_kernel(const float* __restrict__ W, int64_t ldw, CondCfg cfg, const unsigned* __restrict__ maxbits,
                   u4* __restrict__ image) {
    constexpr int JT = 2 * KS, PAIR_U4 = KS * 256;
    const float scale = cond_scale(*maxbits);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    for (int64_t p = (int64_t)blockIdx.x * 4 + wave; p < cfg.T / 2; p += (int64_t)gridDim.x * 4) {
        int64_t base;
        int count;
        cond_tile_desc_bwd(cfg, 2 * p + (q >> 1), base, count);
        u4* tp = image + p * PAIR_U4;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * (q & 1) + e;
                const bool ok = r < count;
                const float w = W[(base + (ok ? r : 0)) * ldw + 16 * jt + j];
                v[e] = ok ? w * scale : 0.f;
            }
            h8 hi, lo;
            csplit8((f4){v[0], v[1], v[2], v[3]}, (f4){v[4], v[5], v[6], v[7]}, hi, lo);
            tp[(jt * 2 + 0) * 64 + lane] = __builtin_bit_cast(u4, hi);
            tp[(jt * 2 + 1) * 64 + lane] = __builtin_bit_cast(u4, lo);
        }
    }
}

// Upstream gradients are typically 1/M-sized; all of the backward is linear in them, so it runs on
// g_lp * 2^k with max |g_lp| 2^k in [8, 16) -- keeps the f16 halves of x*delta normal -- and every
// output is multiplied by 2^-k at the end.
__device__ __forceinline__ float cond_gscale(unsigned maxbits) {
    const float mx = __uint_as_float(maxbits);
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
    return ldexpf(1.f, 3 - ilogbf(mx));
}

struct CondBwdArgs {
    const float* g_lp;      // (M)
    const unsigned* gmaxbits;  // bits of max |g_lp|
    const float* h;         // (M, ldh)
    const u4* pimg;         // forward-layout tiles in backward order
    const u4* timg;         // transposed pairs in backward order
    const float* inv_scale;
    const float* bn_mean;
    const float* bn_alpha;
    const float* acts_aff;  // [S][M][D]
    const float* acts_c;    // [2S][M][CR]
    float* d_aff;           // [S][M][2D]    g_alpha | g_shift per context
    float* d_c;             // [2S][M][DR]   [delta_out t (D/2) | s (D/2) | (delta_t 16, delta_s 16) x L levels]
    float* g_h;             // (M, ldgh)
    float* g_z;             // (M, D) or NULL
    int64_t M, ldh, ldgh, T;
    int S, L, U;
};

__device__ __forceinline__ float dot4(f4 a, f4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

// lane (c, q) gets elements 8(q&1) .. 8(q&1)+7 of context c's 16-vector of net (q >> 1); the vector lives
// 4 per lane in the MFMA output layout (lane (c, q') holds 4q' .. 4q'+3)
__device__ __forceinline__ void gather8(f4 vt, f4 vs, int c, int q, float (&out)[8]) {
    const int srcA = c + 32 * (q & 1), srcB = srcA + 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float t0 = __shfl(vt[j], srcA), s0 = __shfl(vs[j], srcA);
        const float t1 = __shfl(vt[j], srcB), s1 = __shfl(vs[j], srcB);
        out[j] = q < 2 ? t0 : s0;
        out[4 + j] = q < 2 ? t1 : s1;
    }
}

// element k (wave-uniform, 0..15) of context c's 16-vector held 4 per q-lane: no memory traffic inside the
// k loops (a global load there would make hipcc wait vmcnt(0), i.e. for the LDS-DMA copies in flight)
__device__ __forceinline__ float bcast16(f4 v, int k, int c) {
    const int j = k & 3;
    const float sel = j == 0 ? v[0] : (j == 1 ? v[1] : (j == 2 ? v[2] : v[3]));
    return __shfl(sel, c + 16 * (k >> 2));
}

// GH: accumulate the conditioner gradient g_h in this kernel (transposed image stream + one more MFMA group
// per tile pair).  With 32 contexts per wave that does not fit the registers of two waves per SIMD, and with
// 16 the operand reads saturate the LDS; GH = false leaves g_h to cond_gh_kernel, which reads the deltas
// this kernel writes.
template <int DT, int KS, int BT, int NW, bool GH>
__global__ void __launch_bounds__(64 * NW)
cond_flow_bwd_kernel(CondBwdArgs a) {
    constexpr int D = 16 * DT, Hd = D / 2, HT = DT / 2, JT = 2 * KS;
    constexpr int ZS = D + 4, CT = 16 * BT;
    typedef TileStream<KS * 128 + 4, kCondG<KS>, NW, 2> PStream;
    typedef TileStream<KS * 256, 4 / KS, NW, 2> TStream;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u4* pstage = reinterpret_cast<u4*>(smem_raw);
    u4* tstage = pstage + PStream::LDS_U4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    float* gzb = reinterpret_cast<float*>(tstage + (GH ? TStream::LDS_U4 : 0)) + wave * (CT * ZS);  // [CT][ZS] g_z

    const int64_t m0 = ((int64_t)blockIdx.x * NW + wave) * CT;
    const int CR = 3 * Hd + 32 * a.L, DR = 2 * Hd + 32 * a.L;
    const int U = a.U;
    const float gsc = cond_gscale(*a.gmaxbits);
    int64_t mrow[BT];
    bool live[BT];
    float glp[BT];
    h8 bh[BT][KS], bl[BT][KS];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
        const int64_t m = m0 + bt * 16 + c;
        live[bt] = m < a.M;
        mrow[bt] = live[bt] ? m : a.M - 1;
        glp[bt] = live[bt] ? a.g_lp[mrow[bt]] * gsc : 0.f;
        const float* hr = a.h + mrow[bt] * a.ldh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            csplit8(*reinterpret_cast<const f4*>(hr + 32 * ks + 8 * q),
                    *reinterpret_cast<const f4*>(hr + 32 * ks + 8 * q + 4), bh[bt][ks], bl[bt][ks]);
    }
    {   // g_z0 = -g_lp * z0; z0 = [x1 | x2_out] of the coupling layer computed last (slot 2(S-1)+1, upper)
        const int slot = 2 * (a.S - 1) + 1;
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            const float* r = a.acts_c + ((int64_t)slot * a.M + mrow[bt]) * CR;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const f4 v = *reinterpret_cast<const f4*>(r + 16 * t + 4 * q);  // x1 (Hd) then x2_out (Hd)
                *reinterpret_cast<f4*>(gzb + (bt * 16 + c) * ZS + 16 * t + 4 * q) = -glp[bt] * v;
            }
        }
    }
    const float inv = *a.inv_scale;
    TilePipe<KS, BT, PStream> pipe;
    TStream tq;
    pipe.init(a.pimg, pstage, a.T, lane);
    if constexpr (GH) tq.init(a.timg, tstage, a.T / 2);

    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    f4 gacc[BT][JT];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) gacc[bt][jt] = zero;

    // g_h += Wpair^T (H x 32 params) . B (32 params x 16 contexts), B = xsel * d8 per lane
    auto pair_gh = [&](const float (&xsel)[BT], const float (&d8)[BT][8]) {
        if constexpr (!GH) return;
        const u4* tp = tq.next();
        h8 Bh[BT], Bl[BT];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            const float x = xsel[bt];
            csplit8((f4){x * d8[bt][0], x * d8[bt][1], x * d8[bt][2], x * d8[bt][3]},
                    (f4){x * d8[bt][4], x * d8[bt][5], x * d8[bt][6], x * d8[bt][7]}, Bh[bt], Bl[bt]);
        }
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const h8 ah = __builtin_bit_cast(h8, tp[(jt * 2 + 0) * 64 + lane]);
            const h8 al = __builtin_bit_cast(h8, tp[(jt * 2 + 1) * 64 + lane]);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) gacc[bt][jt] = cmfma32h(ah, Bh[bt], gacc[bt][jt]);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) gacc[bt][jt] = cmfma32h(al, Bh[bt], gacc[bt][jt]);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) gacc[bt][jt] = cmfma32h(ah, Bl[bt], gacc[bt][jt]);
        }
    };
    float ones[BT];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) ones[bt] = 1.f;

    auto coupling_bwd = [&](int cond_off, int tr_off, int slot) {
        const float* rec[BT];
        float* drec[BT];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            rec[bt] = a.acts_c + ((int64_t)slot * a.M + mrow[bt]) * CR;
            drec[bt] = a.d_c + ((int64_t)slot * a.M + mrow[bt]) * DR;
        }
        // ---- through x2_out = (x2_in - t) e^-s and log_det += sum s ----
        f4 dt[HT][BT], ds[HT][BT];
        float d8o[HT][BT][8];
#pragma unroll
        for (int o = 0; o < HT; ++o)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                f4* gp = reinterpret_cast<f4*>(gzb + (bt * 16 + c) * ZS + tr_off + 16 * o + 4 * q);
                const f4 g = *gp;
                const f4 x2 = *reinterpret_cast<const f4*>(rec[bt] + Hd + 16 * o + 4 * q);
                const f4 s4 = *reinterpret_cast<const f4*>(rec[bt] + 2 * Hd + 16 * o + 4 * q);
                f4 gn, vt, vs;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e = fast_exp(-s4[j]);
                    vt[j] = -e * g[j];
                    vs[j] = -x2[j] * g[j] - glp[bt];
                    gn[j] = e * g[j];
                }
                *gp = gn;
                dt[o][bt] = vt;
                ds[o][bt] = vs;
                if (live[bt]) {
                    *reinterpret_cast<f4*>(drec[bt] + 16 * o + 4 * q) = vt;
                    *reinterpret_cast<f4*>(drec[bt] + Hd + 16 * o + 4 * q) = vs;
                }
                if constexpr (GH) gather8(vt, vs, c, q, d8o[o][bt]);
            }
        // ---- output layer: U -> Hd; its inputs are the activations of hidden level L-1 ----
        f4 dht[BT], dhs[BT], Pt[BT], Ps[BT];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) dht[bt] = dhs[bt] = zero;
        {
            const int aoff = 3 * Hd + 32 * (a.L - 1);
            f4 hat[BT], has[BT];  // the layer's inputs: hidden activations, 4 per lane
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                hat[bt] = *reinterpret_cast<const f4*>(rec[bt] + aoff + 4 * q);
                has[bt] = *reinterpret_cast<const f4*>(rec[bt] + aoff + 16 + 4 * q);
            }
            for (int k = 0; k < U; ++k) {
                float xt[BT], xs[BT], xsel[BT], pt[BT], pss[BT];
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    xt[bt] = bcast16(hat[bt], k, c);
                    xs[bt] = bcast16(has[bt], k, c);
                    xsel[bt] = q < 2 ? xt[bt] : xs[bt];
                    pt[bt] = pss[bt] = 0.f;
                }
#pragma unroll
                for (int o = 0; o < HT; ++o) {
                    pipe.gemm0(lane, bh, bl, Pt);
                    pipe.gemm1(lane, bh, bl, Ps);
                    pipe.refill1(lane);
#pragma unroll
                    for (int bt = 0; bt < BT; ++bt) {
                        pt[bt] += dot4(Pt[bt], dt[o][bt]);
                        pss[bt] += dot4(Ps[bt], ds[o][bt]);
                    }
                    pair_gh(xsel, d8o[o]);
                }
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    const float vt = reduce_q(pt[bt]) * inv * (1.f - xt[bt] * xt[bt]);
                    const float vs = reduce_q(pss[bt]) * inv * (1.f - xs[bt] * xs[bt]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool mine = k == 4 * q + j;
                        dht[bt][j] = mine ? vt : dht[bt][j];
                        dhs[bt][j] = mine ? vs : dhs[bt][j];
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < HT; ++o) {  // output-layer biases: gP = delta
                pipe.skip_pair(lane);
                pair_gh(ones, d8o[o]);
            }
        }
        // ---- hidden layers L-1 .. 1 (U -> U), then layer 0 (Hd -> U) ----
        for (int l = a.L - 1; l >= 0; --l) {
            float d8[BT][8];
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                if (live[bt]) {
                    *reinterpret_cast<f4*>(drec[bt] + 2 * Hd + 32 * l + 4 * q) = dht[bt];
                    *reinterpret_cast<f4*>(drec[bt] + 2 * Hd + 32 * l + 16 + 4 * q) = dhs[bt];
                }
                if constexpr (GH) gather8(dht[bt], dhs[bt], c, q, d8[bt]);
            }
            if (l > 0) {
                f4 nt[BT], ns[BT];
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) nt[bt] = ns[bt] = zero;
                const int aoff = 3 * Hd + 32 * (l - 1);
                f4 hat[BT], has[BT];
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    hat[bt] = *reinterpret_cast<const f4*>(rec[bt] + aoff + 4 * q);
                    has[bt] = *reinterpret_cast<const f4*>(rec[bt] + aoff + 16 + 4 * q);
                }
                for (int k = 0; k < U; ++k) {
                    float xt[BT], xs[BT], xsel[BT];
#pragma unroll
                    for (int bt = 0; bt < BT; ++bt) {
                        xt[bt] = bcast16(hat[bt], k, c);
                        xs[bt] = bcast16(has[bt], k, c);
                        xsel[bt] = q < 2 ? xt[bt] : xs[bt];
                    }
                    pipe.gemm0(lane, bh, bl, Pt);
                    pipe.gemm1(lane, bh, bl, Ps);
                    pipe.refill1(lane);
                    pair_gh(xsel, d8);
#pragma unroll
                    for (int bt = 0; bt < BT; ++bt) {
                        const float vt = reduce_q(dot4(Pt[bt], dht[bt])) * inv * (1.f - xt[bt] * xt[bt]);
                        const float vs = reduce_q(dot4(Ps[bt], dhs[bt])) * inv * (1.f - xs[bt] * xs[bt]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool mine = k == 4 * q + j;
                            nt[bt][j] = mine ? vt : nt[bt][j];
                            ns[bt][j] = mine ? vs : ns[bt][j];
                        }
                    }
                }
                pipe.skip_pair(lane);
                pair_gh(ones, d8);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    dht[bt] = nt[bt];
                    dhs[bt] = ns[bt];
                }
            } else {
                f4 x1v[BT][HT];  // x1, the layer's conditioning half: the same input for both nets
#pragma unroll
                for (int bt = 0; bt < BT; ++bt)
#pragma unroll
                    for (int t = 0; t < HT; ++t) x1v[bt][t] = *reinterpret_cast<const f4*>(rec[bt] + 16 * t + 4 * q);
                for (int k = 0; k < Hd; ++k) {
                    float x[BT];
#pragma unroll
                    for (int bt = 0; bt < BT; ++bt) {
                        f4 v = x1v[bt][0];
#pragma unroll
                        for (int t = 1; t < HT; ++t) v = (k >> 4) == t ? x1v[bt][t] : v;
                        x[bt] = bcast16(v, k & 15, c);
                    }
                    pipe.gemm0(lane, bh, bl, Pt);
                    pipe.gemm1(lane, bh, bl, Ps);
                    pipe.refill1(lane);
                    pair_gh(x, d8);
#pragma unroll
                    for (int bt = 0; bt < BT; ++bt) {
                        const float g = reduce_q(dot4(Pt[bt], dht[bt]) + dot4(Ps[bt], dhs[bt])) * inv;
                        if (q == 0) gzb[(bt * 16 + c) * ZS + cond_off + k] += g;
                    }
                }
                pipe.skip_pair(lane);
                pair_gh(ones, d8);
            }
        }
    };

    for (int stage = 0; stage < a.S; ++stage) {
        const int si = a.S - 1 - stage;  // the forward's compute index of this stage
        coupling_bwd(0, Hd, 2 * si + 1);  // RealNVP(transform_upper=True), computed last in the stage
        {   // BatchNorm^-1 of layer 2*stage: z_out = z_in * alpha + mean
            const float* bnA = a.bn_alpha + (int64_t)(2 * stage) * D;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const f4 al = *reinterpret_cast<const f4*>(bnA + 16 * t + 4 * q);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    f4* gp = reinterpret_cast<f4*>(gzb + (bt * 16 + c) * ZS + 16 * t + 4 * q);
                    *gp = *gp * al;
                }
            }
        }
        coupling_bwd(Hd, 0, 2 * si);
        {   // z_out = (z_in - shift) e^-a alpha_bn + mean_bn;  log_det += a
            const float* bnA = a.bn_alpha + (int64_t)(2 * stage + 1) * D;
            const float* bnM = a.bn_mean + (int64_t)(2 * stage + 1) * D;
            f4 Pa[BT];
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                pipe.gemm0(lane, bh, bl, Pa);
                pipe.skip1(lane);  // the shift tile: its values are not needed backwards
                const f4 al = *reinterpret_cast<const f4*>(bnA + 16 * t + 4 * q);
                const f4 mu = *reinterpret_cast<const f4*>(bnM + 16 * t + 4 * q);
                float d8[BT][8];
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    f4* gp = reinterpret_cast<f4*>(gzb + (bt * 16 + c) * ZS + 16 * t + 4 * q);
                    const f4 g = *gp;
                    const f4 zo = *reinterpret_cast<const f4*>(a.acts_aff + ((int64_t)si * a.M + mrow[bt]) * D + 16 * t + 4 * q);
                    f4 ga, gsh, gn;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float e = fast_exp(-Pa[bt][j] * inv) * al[j];
                        ga[j] = -g[j] * (zo[j] - mu[j]) - glp[bt];
                        gsh[j] = -g[j] * e;
                        gn[j] = g[j] * e;
                    }
                    *gp = gn;
                    if (live[bt]) {
                        float* dr = a.d_aff + ((int64_t)si * a.M + mrow[bt]) * 2 * D;
                        *reinterpret_cast<f4*>(dr + 16 * t + 4 * q) = ga;
                        *reinterpret_cast<f4*>(dr + D + 16 * t + 4 * q) = gsh;
                    }
                    if constexpr (GH) gather8(ga, gsh, c, q, d8[bt]);
                }
                pair_gh(ones, d8);
            }
        }
    }

    // ---- outputs: g_h (rows j = 16 jt + 4q + jj of the accumulators, column = context), g_z ----
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
        if (live[bt]) {
            if constexpr (GH) {
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
                    *reinterpret_cast<f4*>(a.g_h + mrow[bt] * a.ldgh + 16 * jt + 4 * q) = gacc[bt][jt] * (inv / gsc);
            }
            if (a.g_z) {
#pragma unroll
                for (int t = 0; t < DT; ++t)
                    *reinterpret_cast<f4*>(a.g_z + mrow[bt] * D + 16 * t + 4 * q) =
                        *reinterpret_cast<const f4*>(gzb + (bt * 16 + c) * ZS + 16 * t + 4 * q) * (1.f / gsc);
            }
        }
    }
}


// ---------------------------------------------------------------------------
// g_h on its own (after cond_flow_bwd_kernel<..., GH = false>): g_h[m,:] = sum_p gP[m,p] W[p,:] with
// gP[m,(k,o)] = x[m,k] delta[m,o] rebuilt in registers from the saved activations and the deltas, the
// transposed operand image streamed through a 4-slot LDS ring.  No per-context state in LDS at all.
// ---------------------------------------------------------------------------
template <int DT, int KS, int BT, int NW>
__global__ void __launch_bounds__(64 * NW)
cond_gh_kernel(CondBwdArgs a) {
    constexpr int D = 16 * DT, Hd = D / 2, HT = DT / 2, JT = 2 * KS;
    constexpr int CT = 16 * BT;
    typedef TileStream<KS * 256, 4 / KS, NW, 4> TStream;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u4* tstage = reinterpret_cast<u4*>(smem_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int64_t m0 = ((int64_t)blockIdx.x * NW + wave) * CT;
    const int CR = 3 * Hd + 32 * a.L, DR = 2 * Hd + 32 * a.L;
    const int U = a.U;
    const float gsc = cond_gscale(*a.gmaxbits);
    const float inv = *a.inv_scale;
    int64_t mrow[BT];
    bool live[BT];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
        const int64_t m = m0 + bt * 16 + c;
        live[bt] = m < a.M;
        mrow[bt] = live[bt] ? m : a.M - 1;
    }
    TStream tq;
    tq.init(a.timg, tstage, a.T / 2);
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    f4 gacc[BT][JT];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) gacc[bt][jt] = zero;
    auto pair_gh = [&](const float (&xsel)[BT], const float (&d8)[BT][8]) {
        const u4* tp = tq.next();
        h8 Bh[BT], Bl[BT];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            const float x = xsel[bt];
            csplit8((f4){x * d8[bt][0], x * d8[bt][1], x * d8[bt][2], x * d8[bt][3]},
                    (f4){x * d8[bt][4], x * d8[bt][5], x * d8[bt][6], x * d8[bt][7]}, Bh[bt], Bl[bt]);
        }
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const h8 ah = __builtin_bit_cast(h8, tp[(jt * 2 + 0) * 64 + lane]);
            const h8 al = __builtin_bit_cast(h8, tp[(jt * 2 + 1) * 64 + lane]);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) gacc[bt][jt] = cmfma32h(ah, Bh[bt], gacc[bt][jt]);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) gacc[bt][jt] = cmfma32h(al, Bh[bt], gacc[bt][jt]);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) gacc[bt][jt] = cmfma32h(ah, Bl[bt], gacc[bt][jt]);
        }
    };
    float ones[BT];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) ones[bt] = 1.f;
    // this lane's 8 deltas of a 16-vector stored at `p` (t net) / `p + sstride` (s net): elements 8(q&1)..+7 of net q>>1
    auto load8 = [&](const float* p, int sstride, float (&out)[8]) {
        const float* src = p + (q >> 1) * sstride + 8 * (q & 1);
        const f4 v0 = *reinterpret_cast<const f4*>(src), v1 = *reinterpret_cast<const f4*>(src + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            out[j] = v0[j];
            out[4 + j] = v1[j];
        }
    };
    auto coupling_gh = [&](int slot) {
        const float* rec[BT];
        const float* drec[BT];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            rec[bt] = a.acts_c + ((int64_t)slot * a.M + mrow[bt]) * CR;
            drec[bt] = a.d_c + ((int64_t)slot * a.M + mrow[bt]) * DR;
        }
        {   // output layer: inputs = hidden level L-1, deltas [t (Hd) | s (Hd)]
            float d8o[HT][BT][8];
            f4 hat[BT], has[BT];
            const int aoff = 3 * Hd + 32 * (a.L - 1);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                hat[bt] = *reinterpret_cast<const f4*>(rec[bt] + aoff + 4 * q);
                has[bt] = *reinterpret_cast<const f4*>(rec[bt] + aoff + 16 + 4 * q);
#pragma unroll
                for (int o = 0; o < HT; ++o) load8(drec[bt] + 16 * o, Hd, d8o[o][bt]);
            }
            for (int k = 0; k < U; ++k) {
                float xsel[BT];
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    const float xt = bcast16(hat[bt], k, c), xs = bcast16(has[bt], k, c);
                    xsel[bt] = q < 2 ? xt : xs;
                }
#pragma unroll
                for (int o = 0; o < HT; ++o) pair_gh(xsel, d8o[o]);
            }
#pragma unroll
            for (int o = 0; o < HT; ++o) pair_gh(ones, d8o[o]);
        }
        for (int l = a.L - 1; l >= 0; --l) {  // hidden layers L-1..1, then layer 0: deltas of level l
            float d8[BT][8];
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) load8(drec[bt] + 2 * Hd + 32 * l, 16, d8[bt]);
            if (l > 0) {
                f4 hat[BT], has[BT];
                const int aoff = 3 * Hd + 32 * (l - 1);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    hat[bt] = *reinterpret_cast<const f4*>(rec[bt] + aoff + 4 * q);
                    has[bt] = *reinterpret_cast<const f4*>(rec[bt] + aoff + 16 + 4 * q);
                }
                for (int k = 0; k < U; ++k) {
                    float xsel[BT];
#pragma unroll
                    for (int bt = 0; bt < BT; ++bt) {
                        const float xt = bcast16(hat[bt], k, c), xs = bcast16(has[bt], k, c);
                        xsel[bt] = q < 2 ? xt : xs;
                    }
                    pair_gh(xsel, d8);
                }
            } else {
                f4 x1v[BT][HT];
#pragma unroll
                for (int bt = 0; bt < BT; ++bt)
#pragma unroll
                    for (int t = 0; t < HT; ++t) x1v[bt][t] = *reinterpret_cast<const f4*>(rec[bt] + 16 * t + 4 * q);
                for (int k = 0; k < Hd; ++k) {
                    float x[BT];
#pragma unroll
                    for (int bt = 0; bt < BT; ++bt) {
                        f4 v = x1v[bt][0];
#pragma unroll
                        for (int t = 1; t < HT; ++t) v = (k >> 4) == t ? x1v[bt][t] : v;
                        x[bt] = bcast16(v, k & 15, c);
                    }
                    pair_gh(x, d8);
                }
            }
            pair_gh(ones, d8);
        }
    };
    for (int stage = 0; stage < a.S; ++stage) {
        const int si = a.S - 1 - stage;
        coupling_gh(2 * si + 1);
        coupling_gh(2 * si);
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            float d8[BT][8];
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) load8(a.d_aff + ((int64_t)si * a.M + mrow[bt]) * 2 * D + 16 * t, D, d8[bt]);
            pair_gh(ones, d8);
        }
    }
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
        if (live[bt]) {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
                *reinterpret_cast<f4*>(a.g_h + mrow[bt] * a.ldgh + 16 * jt + 4 * q) = gacc[bt][jt] * (inv / gsc);
        }
}

// ---------------------------------------------------------------------------
// Kernel 2: g_W, g_b.  blockIdx.x = job (one (coupling layer, MLP layer, net, 16-output tile) or one
// Affine 16-feature tile) x a contiguous slice of the contexts.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cond_gmax_kernel(const float* __restrict__ g, int64_t n, unsigned* __restrict__ maxbits) {
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) mx = fmaxf(mx, fabsf(g[i]));
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) atomicMax(maxbits, __float_as_uint(mx));
}

struct GwArgs {
    const float* h;
    const float* acts_c;
    const float* d_c;
    const float* d_aff;
    float* g_W;   // (D_params, ldgw), zero-initialised
    float* g_b;   // (D_params), zero-initialised
    const unsigned* gmaxbits;
    const u4* himg;  // h as split-f16 MFMA B operands: [32-context group][jt][hi/lo][lane] (cond_hsplit_kernel)
    int64_t M, ldh, ldgw;
    CondCfg cfg;
    int jobs, spp;  // jobs per context slice; 128-context (64 at H = 128) steps per slice
};

// A job = up to two SEGMENTS that share the staged h operands: a segment is one (MLP layer, net, 16-output
// tile) -- d_in weight rows + one bias row, one delta tile -- or one Affine 16-feature tile (bias row only).
// The t and s nets of the hidden / output layers are merged into one job (2 x 17 items), pairs of alpha or
// shift tiles of an Affine too; the 33-item layer-0 nets stay separate.  All segments of a job have the
// same d_in.
struct GwSeg {
    const float* x;   // x[m * xs + k], k < d_in   (NULL: no weight items)
    const float* d;   // d[m * dstr + o], o < count
    int64_t wbase;    // parameter index of weight (k = 0, o = 0): + k * wk + o
    int64_t bbase;    // parameter index of bias o = 0
    int wk, count;
};
struct GwJob {
    GwSeg seg[2];
    int64_t xs, dstr;
    int nseg, d_in;
};

__host__ __device__ inline int gw_jobs_per_stage(const CondCfg& c) { return 2 * (2 + (c.L - 1) + c.HT) + 2 * c.HT; }

__device__ inline GwJob gw_job(const GwArgs& a, int job) {
    const CondCfg& c = a.cfg;
    const int Hd = c.D / 2, U = c.U, L = c.L;
    const int JC = 2 + (L - 1) + c.HT, JS = 2 * JC + 2 * c.HT;
    const int CR = 3 * Hd + 32 * L, DR = 2 * Hd + 32 * L;
    const int si = job / JS, stage = c.S - 1 - si;
    int r = job % JS;
    GwJob j;
    const int64_t so = (int64_t)stage * c.fl.stage;
    if (r >= 2 * JC) {  // Affine: gP = g_alpha / g_shift themselves, one segment per 16 features, two per job
        const int which = (r - 2 * JC) / c.HT, pair = (r - 2 * JC) % c.HT;
        j.nseg = 2;
        j.d_in = 0;
        j.xs = 0;
        j.dstr = 2 * c.D;
        for (int t = 0; t < 2; ++t) {
            GwSeg& g = j.seg[t];
            const int ft = 2 * pair + t;
            g.x = nullptr;
            g.d = a.d_aff + (int64_t)si * a.M * 2 * c.D + which * c.D + 16 * ft;
            g.wbase = 0;
            g.wk = 0;
            g.bbase = so + c.fl.p_up + c.fl.p_low + which * c.D + 16 * ft;
            g.count = 16;
        }
        return j;
    }
    const int low = r < JC;  // slot 2si = RealNVP(lower), 2si+1 = RealNVP(upper)
    if (!low) r -= JC;
    const int slot = 2 * si + (low ? 0 : 1);
    const int64_t off = so + (low ? c.fl.p_up : 0);
    const float* rec = a.acts_c + (int64_t)slot * a.M * CR;
    const float* drec = a.d_c + (int64_t)slot * a.M * DR;
    j.xs = CR;
    j.dstr = DR;
    if (r < 2) {  // layer 0, net r
        const int net = r;
        j.nseg = 1;
        j.d_in = Hd;
        GwSeg& g = j.seg[0];
        g.x = rec;
        g.d = drec + 2 * Hd + net * 16;
        g.wbase = off + (int64_t)net * Hd * U;
        g.wk = U;
        g.bbase = off + 2 * (int64_t)Hd * U + net * U;
        g.count = U;
    } else if (r < 2 + (L - 1)) {  // hidden layer lvl, both nets
        const int lvl = r - 1;
        const int64_t ol = off + 2 * (int64_t)Hd * U + 2 * U + (int64_t)(lvl - 1) * (2 * U * U + 2 * U);
        j.nseg = 2;
        j.d_in = U;
        for (int net = 0; net < 2; ++net) {
            GwSeg& g = j.seg[net];
            g.x = rec + 3 * Hd + 32 * (lvl - 1) + net * 16;
            g.d = drec + 2 * Hd + 32 * lvl + net * 16;
            g.wbase = ol + (int64_t)net * U * U;
            g.wk = U;
            g.bbase = ol + 2 * (int64_t)U * U + net * U;
            g.count = U;
        }
    } else {  // output layer, tile ot, both nets
        const int ot = r - 2 - (L - 1);
        const int64_t oo = off + 2 * (int64_t)Hd * U + 2 * U + (int64_t)(L - 1) * (2 * U * U + 2 * U);
        j.nseg = 2;
        j.d_in = U;
        for (int net = 0; net < 2; ++net) {
            GwSeg& g = j.seg[net];
            g.x = rec + 3 * Hd + 32 * (L - 1) + net * 16;
            g.d = drec + net * Hd + 16 * ot;
            g.wbase = oo + (int64_t)net * U * Hd + 16 * ot;
            g.wk = Hd;
            g.bbase = oo + 2 * (int64_t)U * Hd + net * Hd + 16 * ot;
            g.count = 16;
        }
    }
    return j;
}

// h (M, ldh) -> ready B operands of the g_W contraction, split once for all jobs: group g = contexts
// 32g..32g+31; lane (j = lane & 15, q = lane >> 4) of tile jt holds contexts 32g + 8q .. +7 of hidden unit
// 16 jt + j (zero past M).  One wave per (group, jt).
template <int KS>
__global__ void __launch_bounds__(256)
cond_hsplit_kernel(const float* __restrict__ h, int64_t ldh, int64_t M, u4* __restrict__ himg) {
    constexpr int JT = 2 * KS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int64_t ngroups = (M + 31) / 32;
    for (int64_t w = (int64_t)blockIdx.x * 4 + wave; w < ngroups * JT; w += (int64_t)gridDim.x * 4) {
        const int64_t g = w / JT;
        const int jt = (int)(w - g * JT);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t m = 32 * g + 8 * q + e;
            const float x = h[(m < M ? m : M - 1) * ldh + 16 * jt + j];
            v[e] = m < M ? x : 0.f;
        }
        h8 hi, lo;
        csplit8((f4){v[0], v[1], v[2], v[3]}, (f4){v[4], v[5], v[6], v[7]}, hi, lo);
        himg[((g * JT + jt) * 2 + 0) * 64 + lane] = __builtin_bit_cast(u4, hi);
        himg[((g * JT + jt) * 2 + 1) * 64 + lane] = __builtin_bit_cast(u4, lo);
    }
}

#ifndef TNF_GW_NW
#define TNF_GW_NW 8
#endif
#ifndef TNF_GW_CH
#define TNF_GW_CH 128
#endif
constexpr int kGwNW = TNF_GW_NW;
template <int KS>
__host__ __device__ constexpr int gw_ch() { return KS == 4 ? 64 : TNF_GW_CH; }

// a 16-byte LDS read the compiler does not see as one (no s_waitcnt of its own: the caller waits lgkmcnt(0) before use)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<I + 1, N>(f);
    }
}

template <int OFF>
__device__ __forceinline__ h8 lds_read16_blind(unsigned base) {
    h8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF));
    return v;
}

template <int DT, int KS, int NW>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2)))
cond_gw_kernel(GwArgs a) {
    constexpr int JT = 2 * KS, NT = 64 * NW;
    constexpr int XR = 32;                        // staged x rows: one 32-wide layer-0 input or 2 x 16 hidden units
    constexpr int IPW = (XR + 2 + NW - 1) / NW;   // items (weight rows + bias rows of all segments) per wave
    constexpr int CH = gw_ch<KS>();               // contexts staged per step
    constexpr int CS = CH + 4;                    // padded row stride of the transposed staging buffers
    constexpr int NX = (CH * (XR / 4) + NT - 1) / NT, ND = (CH * 16 + NT - 1) / NT;
    constexpr int HB_U4 = (CH / 32) * JT * 2 * 64;  // one step's h operands (32 KB at H = 64)
    // two staging buffers each: a wave writes the next step's rows right behind its own MFMAs of this step, and ONE
    // barrier per step publishes them (with two barriers around a single buffer every wave waited for the slowest twice)
    constexpr int XTF = (XR + 1) * CS, DTF = 2 * 16 * CS;
    __shared__ __attribute__((aligned(16))) float xT2[2 * XTF];   // row XR: ones (the bias items)
    __shared__ __attribute__((aligned(16))) float dT2[2 * DTF];   // one delta tile per segment
    // h as ready MFMA B operands [sub-step][jt][hi/lo][lane], copied from the pre-split image by LDS-DMA
    // (contiguous), two slots
    // two OBJECTS, and the step loop below unrolled by two so that each access names one of them at compile time: hipcc
    // orders an LDS read behind every LDS-DMA copy in flight that it cannot prove disjoint (s_waitcnt vmcnt(0) in front of
    // the first operand read -- the whole fetch latency exposed once per step when this was one array with a run-time slot)
    __shared__ __attribute__((aligned(16))) u4 hB0[HB_U4];
    __shared__ __attribute__((aligned(16))) u4 hB1[HB_U4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    // XCD-aware placement: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own
    // L2.  All jobs of one context slice are given ids of the same residue mod 8, so the slice's h image
    // (a.spp steps, ~1 MB) is fetched into ONE L2 and re-read there by the other jobs.
    const int xcd = blockIdx.x & 7, tq = blockIdx.x >> 3;
    const int split = xcd + 8 * (tq / a.jobs);
    __shared__ GwJob job;  // in LDS: its segments are indexed dynamically
    if (tid == 0) job = gw_job(a, tq % a.jobs);
    __syncthreads();
    const int ips = job.d_in + 1;                 // items per segment
    const int items = job.nseg * ips;
    const int xq = (job.d_in + 3) >> 2;           // float4 per segment row of x actually read (rows are padded to 4)
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    f4 acc[IPW][JT];
    float gb[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        gb[i] = 0.f;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) acc[i][jt] = zero;
    }
    for (int i = tid; i < CS; i += NT) xT2[XR * CS + i] = xT2[XTF + XR * CS + i] = 1.f;

    // register-staged prefetch of one step: x rows and delta rows of CH contexts (zero past M)
    f4 px[NX], pd[ND];
    const int64_t himg_u4 = ((a.M + 31) / 32) * JT * 2 * 64;
    auto fetch_h = [&](int64_t step, u4* hdst) {  // asynchronous: complete behind the issuing wave's vmcnt(0) + a barrier
        const int64_t base = step * HB_U4;
        for (int i = wave; i < HB_U4 / 64; i += NW) {
            int64_t g = base + i * 64 + lane;
            g = g < himg_u4 ? g : himg_u4 - 1;  // past the last group: never used (x and delta are zero there)
            __builtin_amdgcn_global_load_lds(a.himg + g, (lds_void*)(hdst + i * 64), 16, 0, 0);
        }
    };
    const int xrow4 = job.nseg * xq;  // float4 per context over all segments
    // which (context, segment, float4) of a step this thread stages: the same every step, so the divisions by the job's
    // run-time extents are done once (they were half of the kernel's vector instructions when redone per step)
    const float* xsrc[NX];
    const float* dsrc[ND];
    int xctx[NX], xdst[NX], xne[NX], dctx[ND], ddst[ND], dne[ND];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int idx = tid + i * NT;
        const int ctx = xrow4 ? idx / xrow4 : 0, k4g = xrow4 ? idx - ctx * xrow4 : 0;
        const int sg = xq ? k4g / xq : 0, k4 = xq ? k4g - sg * xq : 0;
        const bool ok = xrow4 && ctx < CH;
        xctx[i] = ok ? ctx : -1;
        xsrc[i] = ok ? job.seg[sg].x + 4 * k4 : a.h;
        xdst[i] = (sg * job.d_in + 4 * k4) * CS + ctx;
        const int left = job.d_in - 4 * k4;
        xne[i] = ok ? (left < 4 ? left : 4) : 0;
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int idx = tid + i * NT;
        const int ctx = idx / (4 * job.nseg), rem = idx - ctx * 4 * job.nseg;
        const int sg = rem >> 2, o4 = rem & 3;
        const bool ok = ctx < CH;
        dctx[i] = ok ? ctx : -1;
        dsrc[i] = job.seg[sg].d + 4 * o4;
        ddst[i] = (sg * 16 + 4 * o4) * CS + ctx;
        const int left = job.seg[sg].count - 4 * o4;
        dne[i] = left < 0 ? 0 : (left < 4 ? left : 4);
    }
    // The loaded registers are not touched before commit() (rows past M are clamped to row M - 1 here and zeroed there): any
    // instruction on them in between would make the wave wait for the loads before its MFMAs instead of behind them.  The
    // pointers come out of the job descriptor in LDS, so they are cast to the global address space by hand (generic `flat`
    // loads would also count on lgkmcnt, against the LDS operand reads).
    typedef const __attribute__((address_space(1))) f4* gf4p;
    const int64_t mlast = a.M - 1;
    auto fetch = [&](int64_t mbase) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            int64_t m = mbase + (xctx[i] >= 0 ? xctx[i] : 0);
            m = m < a.M ? m : mlast;
            px[i] = *(gf4p)(xsrc[i] + (xctx[i] >= 0 ? m * job.xs : 0));
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            int64_t m = mbase + (dctx[i] >= 0 ? dctx[i] : 0);
            m = m < a.M ? m : mlast;
            pd[i] = *(gf4p)(dsrc[i] + m * job.dstr);
        }
    };
    auto commit = [&](int buf, int64_t mbase) {
        float* xT = xT2 + buf * XTF;
        float* dT = dT2 + buf * DTF;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const bool in = mbase + xctx[i] < a.M;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < xne[i]) xT[xdst[i] + e * CS] = in ? px[i][e] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            if (dctx[i] >= 0) {
                const bool in = mbase + dctx[i] < a.M;
#pragma unroll
                for (int e = 0; e < 4; ++e) dT[ddst[i] + e * CS] = (in && e < dne[i]) ? pd[i][e] : 0.f;
            }
        }
    };
    // item -> (segment, x row in xT)
    int iseg[IPW], irow[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int it = wave + i * NW;
        const int sg = it / ips, k = it - sg * ips;
        iseg[i] = sg < job.nseg ? sg : 0;
        irow[i] = k < job.d_in ? sg * job.d_in + k : XR;
    }

    const int64_t nsteps = (a.M + CH - 1) / CH;
    int64_t st = (int64_t)split * a.spp;
    const int64_t st_end = (st + a.spp) < nsteps ? (st + a.spp) : nsteps;
    if (st < st_end) {  // step 0 into buffer 0
        fetch(st * CH);
        fetch_h(st, hB0);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        commit(0, st * CH);
    }
    __syncthreads();
    auto step = [&](auto slot) {
        constexpr int cur = decltype(slot)::value;
        // here buffer `cur` and h slot `cur` are complete and visible, and nobody reads the other ones any more
        if (st + 1 < st_end) {  // both in flight during the MFMAs below
            fetch((st + 1) * CH);
            fetch_h(st + 1, cur ? hB0 : hB1);
        }
        const float* xT = xT2 + cur * XTF;
        const float* dT = dT2 + cur * DTF;
        const unsigned hbase = (unsigned)(uintptr_t)(lds_void*)((cur ? hB1 : hB0) + lane);
        static_for<0, CH / 32>([&](auto sub_c) {
            constexpr int sub = decltype(sub_c)::value;
            const int co = 32 * sub + 8 * q;
            h8 Bh[JT], Bl[JT];
            static_for<0, JT>([&](auto jt_c) {
                constexpr int jt = decltype(jt_c)::value;
#if TNF_GW_ABL == 4  // timing experiment: no B-operand reads
                Bh[jt] = Bl[jt] = __builtin_bit_cast(h8, (u4){(unsigned)lane, (unsigned)st, (unsigned)sub, (unsigned)jt});
#else
                // read by hand: hipcc orders a visible LDS read of this array behind every LDS-DMA copy in flight
                // (s_waitcnt vmcnt(0), i.e. behind the x / delta rows just requested as well); the copy in flight targets the
                // OTHER slot, this one was completed before the barrier that ended the previous step
                Bh[jt] = lds_read16_blind<((sub * JT + jt) * 2 + 0) * 1024>(hbase);
                Bl[jt] = lds_read16_blind<((sub * JT + jt) * 2 + 1) * 1024>(hbase);
#endif
            });
#if TNF_GW_ABL != 4
            if constexpr (JT == 4)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bh[0]), "+v"(Bh[1]), "+v"(Bh[2]), "+v"(Bh[3]), "+v"(Bl[0]), "+v"(Bl[1]),
                             "+v"(Bl[2]), "+v"(Bl[3]));
            else {
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bh[jt]), "+v"(Bl[jt]));
            }
#endif
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                if (wave + i * NW < items) {  // wave-uniform
                    h8 ah, al;
#if TNF_GW_ABL == 2  // timing experiment: no A-operand build (LDS reads, products, split)
                    ah = al = __builtin_bit_cast(h8, (u4){(unsigned)lane, (unsigned)st, (unsigned)sub, (unsigned)i});
#else
                    const float* dr = dT + (iseg[i] * 16 + r) * CS + co;
                    const float* xr = xT + irow[i] * CS + co;
                    const f4 a0 = *reinterpret_cast<const f4*>(xr) * *reinterpret_cast<const f4*>(dr);
                    const f4 a1 = *reinterpret_cast<const f4*>(xr + 4) * *reinterpret_cast<const f4*>(dr + 4);
                    gb[i] += (a0[0] + a0[1]) + (a0[2] + a0[3]) + (a1[0] + a1[1]) + (a1[2] + a1[3]);
#if TNF_COND_ABLATE == 4  // timing experiment: no operand split
                    ah = al = __builtin_bit_cast(h8, (u4){__float_as_uint(a0[0]), __float_as_uint(a0[1]), __float_as_uint(a1[0]), __float_as_uint(a1[1])});
#else
                    csplit8(a0, a1, ah, al);
#endif
#endif
#if TNF_GW_ABL == 1  // timing experiment: no MFMAs
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        const f4 ha = __builtin_bit_cast(f4, ah), la = __builtin_bit_cast(f4, al);
                        const f4 hb = __builtin_bit_cast(f4, Bh[jt]), lb = __builtin_bit_cast(f4, Bl[jt]);
                        acc[i][jt] += ha * hb + la * lb;
                    }
#else
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        acc[i][jt] = cmfma32h(ah, Bh[jt], acc[i][jt]);
                        acc[i][jt] = cmfma32h(al, Bh[jt], acc[i][jt]);
                        acc[i][jt] = cmfma32h(ah, Bl[jt], acc[i][jt]);
                    }
#endif
                }
            }
        });
        if (st + 1 < st_end) {
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the rows fetched above, and this wave's share of the h copy
            commit(cur ^ 1, (st + 1) * CH);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    while (st < st_end) {
        step(std::integral_constant<int, 0>());
        if (++st >= st_end) break;
        step(std::integral_constant<int, 1>());
        ++st;
    }
    const float ig = 1.f / cond_gscale(*a.gmaxbits);  // the deltas were computed on scaled upstream gradients
    // ---- merge: rows of the accumulators = outputs 4q + jj, columns = hidden unit 16 jt + r ----
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int it = wave + i * NW;
        if (it < items) {
            const int sg = it / ips, k = it - sg * ips;
            const GwSeg& g = job.seg[sg];
            const int64_t pb = k < job.d_in ? g.wbase + (int64_t)k * g.wk : g.bbase;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int o = 4 * q + jj;
                if (o < g.count) {
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) atomicAdd(a.g_W + (pb + o) * a.ldgw + 16 * jt + r, acc[i][jt][jj] * ig);
                }
            }
            const float sres = reduce_q(gb[i]);
            if (q == 0 && r < g.count) atomicAdd(a.g_b + pb + r, sres * ig);
        }
    }
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
int64_t cond_acts_floats(int64_t M, int D, int S, int L) { return (int64_t)S * M * D + 2 * (int64_t)S * M * (3 * (D / 2) + 32 * L); }
// deltas of every layer + the pre-split h image of the g_W kernel (H floats per context, 32-context groups)
int64_t cond_deltas_floats(int64_t M, int D, int S, int L, int H) {
    return 2 * (int64_t)S * M * D + 2 * (int64_t)S * M * (D + 32 * L) + ((M + 31) / 32) * 32 * (int64_t)H;
}

int64_t cond_flow_bwd_workspace(int D, int S, int L, int U, int H) {
    const CondCfg c = cond_cfg(D, S, L, U, H);
    const int KS = H / 32;
    return 256 + c.T * (int64_t)(KS * 128 + 4) * 16 + (c.T / 2) * (int64_t)(KS * 256) * 16;
}

template <int DT, int KS, int BT, int NW>
static int launch_gh(const CondBwdArgs& a, hipStream_t st) {
    typedef TileStream<KS * 256, 4 / KS, NW, 4> TStream;
    const size_t smem = (size_t)TStream::LDS_U4 * 16;
    auto k = cond_gh_kernel<DT, KS, BT, NW>;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int64_t per_wg = (int64_t)NW * 16 * BT;
    const int64_t blocks = (a.M + per_wg - 1) / per_wg;
    if (blocks > 0x7fffffff) return fail(TNF_EUNSUPPORTED, "cond_gh: grid too large");
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * NW), smem, st, a);
    return check_launch("cond_gh");
}

template <int DT, int KS, int BT, int NW, bool GH>
static int launch_bwd_variant(const CondBwdArgs& a, hipStream_t st) {
    typedef TileStream<KS * 128 + 4, kCondG<KS>, NW, 2> PStream;
    typedef TileStream<KS * 256, 4 / KS, NW, 2> TStream;
    constexpr int D = 16 * DT;
    const size_t smem = (size_t)(PStream::LDS_U4 + (GH ? TStream::LDS_U4 : 0)) * 16 + (size_t)NW * 16 * BT * (D + 4) * 4;
    auto k = cond_flow_bwd_kernel<DT, KS, BT, NW, GH>;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int64_t per_wg = (int64_t)NW * 16 * BT;
    const int64_t blocks = (a.M + per_wg - 1) / per_wg;
    if (blocks > 0x7fffffff) return fail(TNF_EUNSUPPORTED, "cond_flow_bwd: grid too large");
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * NW), smem, st, a);
    return check_launch("cond_flow_bwd");
}

template <int DT, int KS>
static int launch_bwd_dk(const CondBwdArgs& a, const GwArgs& g, hipStream_t st) {
    int v = g_cond_variant;
    // many contexts: the walk without g_h at 32 contexts per wave + the separate g_h kernel (10.4 + 9.4 ms at
    // 2^20 contexts) beats the single kernel, which at 32 contexts per wave spills (g_h accumulators, two
    // operand slots and the gathered deltas exceed 256 VGPRs) and at 16 saturates the LDS (22 ms)
    if (v == 0) v = a.M >= 256 * 128 ? 5 : 1;
    int rc;
    if (v == 5) {  // the walk without g_h at 32 contexts per wave, then g_h on its own
        rc = launch_bwd_variant<DT, KS, 2, 8, false>(a, st);
        if (!rc) rc = launch_gh<DT, KS, 2, 8>(a, st);
    } else if (v == 4) rc = launch_bwd_variant<DT, KS, 2, 4, true>(a, st);
    else if (v == 3) rc = launch_bwd_variant<DT, KS, 2, 8, true>(a, st);
    else if (v == 2) rc = launch_bwd_variant<DT, KS, 1, 8, true>(a, st);
    else rc = launch_bwd_variant<DT, KS, 1, 4, true>(a, st);
    if (rc) return rc;
    const CondCfg& c = g.cfg;
    const int jobs = c.S * gw_jobs_per_stage(c);  // gw_job(): layer-0 nets apart, the rest merged
    const int64_t nsteps = (a.M + gw_ch<KS>() - 1) / gw_ch<KS>();
    // context slices of ~1 MB of h rows (32 steps), at least 8 (one per XCD) when there is enough work
    int64_t spp = 32;
    while (spp > 1 && (nsteps + spp - 1) / spp < 16) spp >>= 1;
    int64_t split = (nsteps + spp - 1) / spp;
    split = (split + 7) & ~(int64_t)7;
    GwArgs gg = g;
    gg.jobs = jobs;
    gg.spp = (int)spp;
    {
        const int64_t waves = ((a.M + 31) / 32) * 2 * KS;
        int64_t hb = (waves + 3) / 4;
        if (hb > 4096) hb = 4096;
        hipLaunchKernelGGL((cond_hsplit_kernel<KS>), dim3((unsigned)hb), dim3(256), 0, st, g.h, g.ldh, a.M,
                           const_cast<u4*>(g.himg));
    }
    if (split * jobs > 0x7fffffff) return fail(TNF_EUNSUPPORTED, "cond_gw: grid too large");
    hipLaunchKernelGGL((cond_gw_kernel<DT, KS, kGwNW>), dim3((unsigned)(split * jobs)), dim3(64 * kGwNW), 0, st, gg);
    return check_launch("cond_gw");
}

int launch_cond_flow_backward(const float* g_lp, const float* h, const float* W, const float* b, const float* bn_mean,
                              const float* bn_alpha, const float* acts, float* deltas, float* g_h, float* g_W,
                              float* g_b, float* g_z, int64_t M, int D, int S, int L, int U, int H, int64_t ldh,
                              int64_t ldw, int64_t ldgh, int64_t ldgw, void* ws, hipStream_t st) {
    const CondCfg cfg = cond_cfg(D, S, L, U, H);
    const int KS = H / 32;
    char* base = reinterpret_cast<char*>(ws);
    u4* pimg = reinterpret_cast<u4*>(base + 256);
    u4* timg = pimg + cfg.T * (int64_t)(KS * 128 + 4);
    int rc = launch_cond_image(W, b, ldw, cfg, ws, pimg, 1, st);
    if (rc) return rc;
    const unsigned* maxbits = reinterpret_cast<const unsigned*>(ws);
    const unsigned tb = (unsigned)((cfg.T / 2 + 3) / 4);
    if (KS == 1) hipLaunchKernelGGL(cond_timage_kernel<1>, dim3(tb), dim3(256), 0, st, W, ldw, cfg, maxbits, timg);
    else if (KS == 2) hipLaunchKernelGGL(cond_timage_kernel<2>, dim3(tb), dim3(256), 0, st, W, ldw, cfg, maxbits, timg);
    else hipLaunchKernelGGL(cond_timage_kernel<4>, dim3(tb), dim3(256), 0, st, W, ldw, cfg, maxbits, timg);
    unsigned* gmax = reinterpret_cast<unsigned*>(ws) + 2;
    if (hipMemsetAsync(gmax, 0, 4, st) != hipSuccess) return fail(TNF_ELAUNCH, "cond_flow_bwd: memset failed");
    hipLaunchKernelGGL(cond_gmax_kernel, dim3(64), dim3(256), 0, st, g_lp, M, gmax);
    rc = check_launch("cond_flow_bwd_prep");
    if (rc) return rc;
    if (hipMemsetAsync(g_W, 0, (size_t)cfg.fl.total * ldgw * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(g_b, 0, (size_t)cfg.fl.total * sizeof(float), st) != hipSuccess)
        return fail(TNF_ELAUNCH, "cond_flow_bwd: memset failed");
    CondBwdArgs a;
    a.g_lp = g_lp; a.gmaxbits = gmax; a.h = h; a.pimg = pimg; a.timg = timg; a.inv_scale = reinterpret_cast<const float*>(ws) + 1;
    a.bn_mean = bn_mean; a.bn_alpha = bn_alpha;
    a.acts_aff = acts; a.acts_c = acts + (int64_t)S * M * D;
    a.d_aff = deltas; a.d_c = deltas + 2 * (int64_t)S * M * D;
    a.g_h = g_h; a.g_z = g_z; a.M = M; a.ldh = ldh; a.ldgh = ldgh; a.T = cfg.T; a.S = S; a.L = L; a.U = U;
    GwArgs g;
    g.h = h; g.acts_c = a.acts_c; g.d_c = a.d_c; g.d_aff = a.d_aff; g.g_W = g_W; g.g_b = g_b;
    g.M = M; g.ldh = ldh; g.ldgw = ldgw; g.cfg = cfg; g.gmaxbits = gmax;
    g.himg = reinterpret_cast<const u4*>(a.d_c + 2 * (int64_t)S * M * (D + 32 * L));  // 16-byte aligned: all terms are
    if (D == 64) {
        if (KS == 1) return launch_bwd_dk<4, 1>(a, g, st);
        if (KS == 2) return launch_bwd_dk<4, 2>(a, g, st);
        return launch_bwd_dk<4, 4>(a, g, st);
    }
    if (KS == 1) return launch_bwd_dk<2, 1>(a, g, st);
    if (KS == 2) return launch_bwd_dk<2, 2>(a, g, st);
    return launch_bwd_dk<2, 4>(a, g, st);
}

}  // namespace tnf
