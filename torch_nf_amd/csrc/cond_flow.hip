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
#include "mfma_tile.h"
#include "tnf_common.h"

namespace tnf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 cmfma32h(h8 a, h8 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// v = hi + lo, hi = rtz_f16(v), lo = rtz_f16(v - hi)   (pairs packed into one dword each)
__device__ __forceinline__ void csplit2(float v0, float v1, unsigned& hi, unsigned& lo) {
    const auto h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(v1));
    hi = hb;
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
__device__ __forceinline__ void csplit8(f4 v0, f4 v1, h8& hi, h8& lo) {
    unsigned a0, a1, a2, a3, b0, b1, b2, b3;
    csplit2(v0[0], v0[1], a0, b0);
    csplit2(v0[2], v0[3], a1, b1);
    csplit2(v1[0], v1[1], a2, b2);
    csplit2(v1[2], v1[3], a3, b3);
    hi = __builtin_bit_cast(h8, (u4){a0, a1, a2, a3});
    lo = __builtin_bit_cast(h8, (u4){b0, b1, b2, b3});
}

// ---------------------------------------------------------------------------
// The tile program: which 16 parameter-row entries tile t holds, in consumption order.
// ---------------------------------------------------------------------------
struct CondCfg {
    int D, S, L, U, H;  // H: width of the conditioner's last hidden layer (multiple of 32)
    int DT, HT;         // 16-feature tiles of z and of one coupling half
    int64_t TC, TS, T;  // tiles per coupling layer, per stage, in total
    FlowLayout fl;
};
__host__ __device__ inline CondCfg cond_cfg(int D, int S, int L, int U, int H) {
    CondCfg c;
    c.D = D; c.S = S; c.L = L; c.U = U; c.H = H;
    c.DT = D / 16;
    c.HT = D / 32;
    c.TC = 2 * (int64_t)(D / 2) + 2 + (int64_t)(L - 1) * (2 * U + 2) + 2 * (int64_t)U * c.HT + 2 * c.HT;
    c.TS = 2 * c.DT + 2 * c.TC;
    c.T = c.TS * S;
    c.fl = flow_layout(D, S, L, U);
    return c;
}
// (first parameter index, number of valid rows) of tile t
__host__ __device__ inline void cond_tile_desc(const CondCfg& c, int64_t t, int64_t& base, int& count) {
    const int Hd = c.D / 2, U = c.U;
    const int stage = c.S - 1 - (int)(t / c.TS);
    int64_t r = t % c.TS;
    const int64_t so = (int64_t)stage * c.fl.stage;
    if (r < 2 * c.DT) {  // Affine: [alpha tile, shift tile] per 16 features
        base = so + c.fl.p_up + c.fl.p_low + (r & 1) * c.D + 16 * (r >> 1);
        count = 16;
        return;
    }
    r -= 2 * c.DT;
    int64_t off = so + c.fl.p_up;  // RealNVP(lower) first in the inverse pass
    if (r >= c.TC) {
        r -= c.TC;
        off = so;
    }
    const int64_t n0 = 2 * (int64_t)Hd + 2;
    if (r < n0) {  // layer 0: Hd -> U
        count = U;
        if (r < 2 * Hd) base = off + (r & 1) * (int64_t)Hd * U + (r >> 1) * U;
        else base = off + 2 * (int64_t)Hd * U + (r - 2 * Hd) * U;
        return;
    }
    r -= n0;
    off += 2 * (int64_t)Hd * U + 2 * U;
    const int64_t nh = 2 * (int64_t)U + 2;
    if (r < (c.L - 1) * nh) {  // hidden layers: U -> U
        off += (r / nh) * (2 * (int64_t)U * U + 2 * U);
        r %= nh;
        count = U;
        if (r < 2 * U) base = off + (r & 1) * (int64_t)U * U + (r >> 1) * U;
        else base = off + 2 * (int64_t)U * U + (r - 2 * U) * U;
        return;
    }
    r -= (c.L - 1) * nh;
    off += (c.L - 1) * (2 * (int64_t)U * U + 2 * U);
    count = 16;  // output layer: U -> Hd, Hd a multiple of 16
    if (r < 2 * (int64_t)U * c.HT) {
        const int64_t k = r / (2 * c.HT), rem = r % (2 * c.HT);
        base = off + (rem & 1) * (int64_t)U * Hd + k * Hd + 16 * (rem >> 1);
    } else {
        r -= 2 * (int64_t)U * c.HT;
        base = off + 2 * (int64_t)U * Hd + (r & 1) * Hd + 16 * (r >> 1);
    }
}

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

// power-of-two scale that brings the largest magnitude just below 2^14 (f16 max is 65504)
__device__ __forceinline__ float cond_scale(unsigned maxbits) {
    const float mx = __uint_as_float(maxbits);
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
    return ldexpf(1.f, 13 - ilogbf(mx));
}

// Prep 2: the operand image.  Tile t = [ks][part hi/lo][lane] 16-byte groups (the lane's 8 f16 of row
// lane&15, k = 32ks + 8(lane>>4) .. +7) followed by the 16 scaled biases (fp32).
template <int KS>
__global__ void __launch_bounds__(256)
cond_image_kernel(const float* __restrict__ W, const float* __restrict__ b, int64_t ldw, CondCfg cfg,
                  const unsigned* __restrict__ maxbits, float* __restrict__ inv_scale, u4* __restrict__ image) {
    constexpr int TILE_U4 = KS * 128 + 4;
    const float scale = cond_scale(*maxbits);
    if (blockIdx.x == 0 && threadIdx.x == 0) *inv_scale = 1.f / scale;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < cfg.T; t += (int64_t)gridDim.x * 4) {
        int64_t base;
        int count;
        cond_tile_desc(cfg, t, base, count);
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
    int64_t M, ldh, T;
    int S, L, U;
};

// the workgroup's view of the image: chunks of G tiles, double-buffered in LDS
template <int KS, int G, int NTHREADS>
struct TileStream {
    static constexpr int TILE_U4 = KS * 128 + 4;
    static constexpr int CHUNK_U4 = G * TILE_U4;
    static constexpr int PF = (CHUNK_U4 + NTHREADS - 1) / NTHREADS;
    const u4* img;
    u4* stg;
    int64_t total_u4;
    int t;
    u4 pf[PF];

    __device__ __forceinline__ void fetch(int chunk) {
        const int64_t base = (int64_t)chunk * CHUNK_U4;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            int64_t g = base + threadIdx.x + i * NTHREADS;
            g = g < total_u4 ? g : total_u4 - 1;  // clamped, never predicated (see ld_sel)
            pf[i] = img[g];
        }
    }
    __device__ __forceinline__ void commit(int chunk) {
        u4* dst = stg + (chunk & 1) * CHUNK_U4;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int idx = threadIdx.x + i * NTHREADS;
            if (idx < CHUNK_U4) dst[idx] = pf[i];
        }
    }
    __device__ __forceinline__ void init(const u4* image, u4* stage, int64_t tiles) {
        img = image;
        stg = stage;
        total_u4 = tiles * TILE_U4;
        t = 0;
        fetch(0);
        commit(0);
        __syncthreads();
        fetch(1);
    }
    // every wave of the workgroup calls next() the same number of times, in the same order
    __device__ __forceinline__ const u4* next() {
        const int chunk = t / G, in = t - chunk * G;
        if (in == 0 && t > 0) {
            commit(chunk);   // safe: all waves left chunk-2 (same buffer) before the previous barrier
            __syncthreads();
            fetch(chunk + 1);
        }
        ++t;
        return stg + (chunk & 1) * CHUNK_U4 + in * TILE_U4;
    }
};

template <int KS, int BT>
__device__ __forceinline__ void tile_gemm(const u4* tp, int lane, const h8 (&bh)[BT][KS], const h8 (&bl)[BT][KS],
                                          f4 (&P)[BT]) {
    const f4 c0 = *reinterpret_cast<const f4*>(tp + KS * 128 + (lane >> 4));
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) P[bt] = c0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const h8 ah = __builtin_bit_cast(h8, tp[(ks * 2 + 0) * 64 + lane]);
        const h8 al = __builtin_bit_cast(h8, tp[(ks * 2 + 1) * 64 + lane]);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) P[bt] = cmfma32h(ah, bh[bt][ks], P[bt]);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) P[bt] = cmfma32h(ah, bl[bt][ks], P[bt]);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) P[bt] = cmfma32h(al, bh[bt][ks], P[bt]);
    }
}

template <int KS> constexpr int kCondG = 8 / KS;  // tiles per LDS chunk (~16.5 KB)

template <int DT, int KS, int BT, int NW>
__global__ void __launch_bounds__(64 * NW)
cond_flow_kernel(CondArgs a) {
    constexpr int D = 16 * DT, Hd = D / 2, HT = DT / 2;
    constexpr int ZS = D + 4, HS = 20, CT = 16 * BT;
    typedef TileStream<KS, kCondG<KS>, 64 * NW> Stream;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u4* stage = reinterpret_cast<u4*>(smem_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    float* wl = reinterpret_cast<float*>(stage + 2 * Stream::CHUNK_U4) + wave * (CT * (ZS + 2 * HS));
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
    Stream ts;
    ts.init(a.image, stage, a.T);

    float ld[BT];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) ld[bt] = 0.f;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    const int U = a.U;

    // one RealNVP layer, inverse direction (bijectors.py:183-206): z2 <- (z2 - t(z1)) / exp(s(z1))
    auto coupling = [&](int cond_off, int tr_off) {
        f4 at[BT], as[BT], Pt[BT], Ps[BT];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) at[bt] = as[bt] = zero;
        for (int k = 0; k < Hd; ++k) {
            tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Pt);
            tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Ps);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                const float x = zb[(bt * 16 + c) * ZS + cond_off + k] * inv;
                at[bt] += x * Pt[bt];
                as[bt] += x * Ps[bt];
            }
        }
        tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Pt);
        tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Ps);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            at[bt] += inv * Pt[bt];
            as[bt] += inv * Ps[bt];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hbt[(bt * 16 + c) * HS + 4 * q + j] = tanhf(at[bt][j]);
                hbs[(bt * 16 + c) * HS + 4 * q + j] = tanhf(as[bt][j]);
            }
        }
        for (int l = 1; l < a.L; ++l) {
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) at[bt] = as[bt] = zero;
            for (int k = 0; k < U; ++k) {
                tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Pt);
                tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Ps);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    at[bt] += (hbt[(bt * 16 + c) * HS + k] * inv) * Pt[bt];
                    as[bt] += (hbs[(bt * 16 + c) * HS + k] * inv) * Ps[bt];
                }
            }
            tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Pt);
            tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Ps);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                at[bt] += inv * Pt[bt];
                as[bt] += inv * Ps[bt];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hbt[(bt * 16 + c) * HS + 4 * q + j] = tanhf(at[bt][j]);
                    hbs[(bt * 16 + c) * HS + 4 * q + j] = tanhf(as[bt][j]);
                }
            }
        }
        f4 ot[BT][HT], os[BT][HT];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
#pragma unroll
            for (int o = 0; o < HT; ++o) ot[bt][o] = os[bt][o] = zero;
        for (int k = 0; k < U; ++k) {
            float xt[BT], xs[BT];
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                xt[bt] = hbt[(bt * 16 + c) * HS + k] * inv;
                xs[bt] = hbs[(bt * 16 + c) * HS + k] * inv;
            }
#pragma unroll
            for (int o = 0; o < HT; ++o) {
                tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Pt);
                tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Ps);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    ot[bt][o] += xt[bt] * Pt[bt];
                    os[bt][o] += xs[bt] * Ps[bt];
                }
            }
        }
#pragma unroll
        for (int o = 0; o < HT; ++o) {
            tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Pt);
            tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Ps);
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                const f4 t4 = ot[bt][o] + inv * Pt[bt];
                const f4 s4 = os[bt][o] + inv * Ps[bt];
                f4* zp = reinterpret_cast<f4*>(zb + (bt * 16 + c) * ZS + tr_off + 16 * o + 4 * q);
                f4 zv = *zp;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    zv[j] = (zv[j] - t4[j]) * expf(-s4[j]);
                    ld[bt] += s4[j];
                }
                *zp = zv;
            }
        }
    };

    for (int si = 0; si < a.S; ++si) {
        const int stg_i = a.S - 1 - si;
        {   // Affine^-1 (bijectors.py:300-315) then BatchNorm^-1 (:420-426) of layer 2*stage+1
            const float* bnA = a.bn_alpha + (int64_t)(2 * stg_i + 1) * D;
            const float* bnM = a.bn_mean + (int64_t)(2 * stg_i + 1) * D;
            f4 Pa[BT], Psh[BT];
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Pa);
                tile_gemm<KS, BT>(ts.next(), lane, bh, bl, Psh);
                const f4 al = *reinterpret_cast<const f4*>(bnA + 16 * t + 4 * q);
                const f4 mu = *reinterpret_cast<const f4*>(bnM + 16 * t + 4 * q);
                float lal = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) lal += logf(al[j]);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    f4* zp = reinterpret_cast<f4*>(zb + (bt * 16 + c) * ZS + 16 * t + 4 * q);
                    f4 zv = *zp;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float aa = Pa[bt][j] * inv, sh = Psh[bt][j] * inv;
                        zv[j] = (zv[j] - sh) * expf(-aa) * al[j] + mu[j];
                        ld[bt] += aa;
                    }
                    ld[bt] -= lal;
                    *zp = zv;
                }
            }
        }
        coupling(Hd, 0);  // RealNVP(transform_upper=False): conditions on the upper half
        {   // BatchNorm^-1 of layer 2*stage
            const float* bnA = a.bn_alpha + (int64_t)(2 * stg_i) * D;
            const float* bnM = a.bn_mean + (int64_t)(2 * stg_i) * D;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const f4 al = *reinterpret_cast<const f4*>(bnA + 16 * t + 4 * q);
                const f4 mu = *reinterpret_cast<const f4*>(bnM + 16 * t + 4 * q);
                float lal = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) lal += logf(al[j]);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) {
                    f4* zp = reinterpret_cast<f4*>(zb + (bt * 16 + c) * ZS + 16 * t + 4 * q);
                    *zp = *zp * al + mu;
                    ld[bt] -= lal;
                }
            }
        }
        coupling(0, Hd);  // RealNVP(transform_upper=True)
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
            a.log_prob[m] = -0.5f * ss - (float)D * 0.91893853320467274178f - ldt;
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

template <int DT, int KS, int BT, int NW>
static int launch_cond_variant(const CondArgs& a, hipStream_t st) {
    typedef TileStream<KS, kCondG<KS>, 64 * NW> Stream;
    constexpr int D = 16 * DT;
    const size_t smem = (size_t)2 * Stream::CHUNK_U4 * 16 + (size_t)NW * 16 * BT * (D + 4 + 40) * 4;
    auto k = cond_flow_kernel<DT, KS, BT, NW>;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int64_t per_wg = (int64_t)NW * 16 * BT;
    const int64_t blocks = (a.M + per_wg - 1) / per_wg;
    if (blocks > 0x7fffffff) return fail(TNF_EUNSUPPORTED, "cond_flow: grid too large");
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * NW), smem, st, a);
    return check_launch("cond_flow");
}

int g_cond_variant = 0;  // testing hook: 0 = by M, 1 = (BT 1, 4 waves), 2 = (BT 1, 8 waves), 3 = (BT 2, 8 waves)

template <int DT, int KS>
static int launch_cond_dk(const CondArgs& a, hipStream_t st) {
    int v = g_cond_variant;
    if (v == 0) v = a.M >= 256 * 256 ? 3 : (a.M >= 256 * 128 ? 2 : 1);
    if (v == 3) return launch_cond_variant<DT, KS, 2, 8>(a, st);
    if (v == 2) return launch_cond_variant<DT, KS, 1, 8>(a, st);
    return launch_cond_variant<DT, KS, 1, 4>(a, st);
}

int launch_cond_flow_log_prob(const float* z, const float* h, const float* W, const float* b, const float* bn_mean,
                              const float* bn_alpha, float* log_prob, float* z0, float* sum_log_det, int64_t M, int D,
                              int S, int L, int U, int H, int64_t ldh, int64_t ldw, void* ws, hipStream_t st) {
    const CondCfg cfg = cond_cfg(D, S, L, U, H);
    unsigned* maxbits = reinterpret_cast<unsigned*>(ws);
    float* inv_scale = reinterpret_cast<float*>(ws) + 1;
    u4* image = reinterpret_cast<u4*>(reinterpret_cast<char*>(ws) + 256);
    if (hipMemsetAsync(maxbits, 0, 8, st) != hipSuccess) return fail(TNF_ELAUNCH, "cond_flow: memset failed");
    hipLaunchKernelGGL(cond_absmax_kernel, dim3(256), dim3(256), 0, st, W, b, cfg.fl.total, H, ldw, maxbits);
    const unsigned ib = (unsigned)((cfg.T + 3) / 4);
    if (H == 32) hipLaunchKernelGGL(cond_image_kernel<1>, dim3(ib), dim3(256), 0, st, W, b, ldw, cfg, maxbits, inv_scale, image);
    else if (H == 64) hipLaunchKernelGGL(cond_image_kernel<2>, dim3(ib), dim3(256), 0, st, W, b, ldw, cfg, maxbits, inv_scale, image);
    else hipLaunchKernelGGL(cond_image_kernel<4>, dim3(ib), dim3(256), 0, st, W, b, ldw, cfg, maxbits, inv_scale, image);
    int rc = check_launch("cond_flow_prep");
    if (rc) return rc;
    CondArgs a;
    a.z = z; a.h = h; a.image = image; a.inv_scale = inv_scale; a.bn_mean = bn_mean; a.bn_alpha = bn_alpha;
    a.log_prob = log_prob; a.z0 = z0; a.sum_log_det = sum_log_det;
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
