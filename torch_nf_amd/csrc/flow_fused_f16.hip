// Whole-flow kernel, split-f16 matrix path.
//
// Measured on MI355X (scratch microbenchmarks): the fp32 MFMA
// (v_mfma_f32_16x16x4_f32) runs at the fp32 VECTOR rate and does not overlap with VALU or
// transcendental instructions on its SIMD -- time adds up -- whereas an f16 MFMA
// (16x16x32 / 16x16x16) costs ~20 cycles of matrix pipe and hides almost completely under
// the VALU stream of the same SIMD.  The flow is then bound by its sigmoids and exps, not
// by the contractions.  fp32 accuracy is kept by splitting every operand into two f16
// halves, v = hi + lo (hi = rtz_f16(v), lo = rtz_f16(v - hi), |lo| <= 2^-11 |v|), and
// evaluating  x.w ~= xh.wh + xl.wh + xh.wl  in three f16 MFMAs that accumulate in fp32
// (the dropped xl.wl term is 2^-22 relative; f16 subnormals pass through the MFMA
// un-flushed, checked on hardware).  Emulated in numpy on the golden flows the log_prob
// error stays at 2e-7 relative, the same as the fp32 path.
//
// Everything else -- the transposed lane mapping (lane = (sample s, k-quad q)), the
// accumulator->operand chaining without LDS round trips, the activation folding, the folded
// BatchNorm/Affine FMAs, the LDS-resident operand images, the work queue -- is the design of
// mfma_tile.h / flow_fused.hip; only the operand element type and the MFMA shape change:
//   layer 0 (K = H):   v_mfma_f32_16x16x32_f16 when H = 32 (8 f16 per lane), else 16x16x16
//   hidden / output:   v_mfma_f32_16x16x16_f16 (K = 16: lane (s,q) supplies units 4q..4q+3,
//                      which is exactly where the previous accumulator left them)
#include "f16_tile.h"
#include "support_math.h"
#include "tnf_common.h"

namespace tnf {

template <int H, int L>
__global__ void __launch_bounds__(64)
flow_images_f16_kernel(const float* __restrict__ params, float* __restrict__ images, int S, int U,
                       int64_t pstride, int64_t image_floats, int64_t Mp) {
    constexpr int D = 2 * H;
    const int c = blockIdx.x;
    const int64_t m = grid_m();
    if (m >= Mp) return;
    const int64_t pc = coupling_num_params(D, L, U, 1);
    const int64_t stage = 2 * pc + 2 * D;
    build_f16_image<H, L>(images + (m * 2 * S + c) * image_floats, params + m * pstride + (c >> 1) * stage + (c & 1) * pc,
                          U, threadIdx.x);
}

// ---------------------------------------------------------------------------
// One coupling layer on NT tiles, operands from the f16 image in LDS.
// ---------------------------------------------------------------------------
template <int H, int L, bool INV, int NT>
__device__ __forceinline__ void coupling_tile_f16(const float* img, int lane, const f4 (&x)[NT][(H + 15) / 16],
                                                  f4 (&y)[NT][(H + 15) / 16], float (&ssum2)[NT]) {
    typedef F16Image<H, L> Img;
    constexpr int HT = Img::HT;
    const u4* grp = reinterpret_cast<const u4*>(img) + lane;
    const float* bl = img + Img::NWG * 256 + (lane >> 4) * 4;
    auto bias = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(bl + g * 16); };

    f4 acc[NT][2];
    // ---- layer 0: H -> 16 ----
    if constexpr (H == 32) {
        h8 xh[NT], xl[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            u4 a, b;
            split2(x[t][0][0], x[t][0][1], a[0], b[0]);
            split2(x[t][0][2], x[t][0][3], a[1], b[1]);
            split2(x[t][1][0], x[t][1][1], a[2], b[2]);
            split2(x[t][1][2], x[t][1][3], a[3], b[3]);
            xh[t] = __builtin_bit_cast(h8, a);
            xl[t] = __builtin_bit_cast(h8, b);
        }
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            const h8 wh = __builtin_bit_cast(h8, grp[Img::g_w0(net, 0) * 64]);
            const h8 wl = __builtin_bit_cast(h8, grp[Img::g_w0(net, 1) * 64]);
            const f4 b0 = bias(Img::b_b0(net));
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
        for (int t = 0; t < NT; ++t) split4(x[t][0], xh[t], xl[t]);
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            const u4 wv = grp[Img::g_w0(net, 0) * 64];
            const h4 wh = __builtin_bit_cast(h4, u2{wv[0], wv[1]});
            const h4 wl = __builtin_bit_cast(h4, u2{wv[2], wv[3]});
            const f4 b0 = bias(Img::b_b0(net));
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][net] = mfma16h(wh, xh[t], b0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][net] = mfma16h(wh, xl[t], acc[t][net]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][net] = mfma16h(wl, xh[t], acc[t][net]);
        }
    }
    // ---- sigmoid, split ----
    h4 rh[NT][2], rl[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int net = 0; net < 2; ++net) split4(sig2_4(acc[t][net]), rh[t][net], rl[t][net]);
    // ---- hidden layers: 16 -> 16 ----
#pragma unroll
    for (int l = 0; l < L - 1; ++l) {
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            const u4 wv = grp[Img::g_wh(l, net) * 64];
            const h4 wh = __builtin_bit_cast(h4, u2{wv[0], wv[1]});
            const h4 wl = __builtin_bit_cast(h4, u2{wv[2], wv[3]});
            const f4 bh = bias(Img::b_bh(l, net));
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][net] = mfma16h(wh, rh[t][net], bh);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][net] = mfma16h(wh, rl[t][net], acc[t][net]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][net] = mfma16h(wl, rh[t][net], acc[t][net]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int net = 0; net < 2; ++net) split4(sig2_4(acc[t][net]), rh[t][net], rl[t][net]);
    }
    // ---- output layer: 16 -> H, then the scale-shift ----
#pragma unroll
    for (int mo = 0; mo < HT; ++mo) {
        f4 tt[NT], sv[NT];
        {
            const u4 wv = grp[Img::g_w2(0, mo) * 64];
            const h4 wh = __builtin_bit_cast(h4, u2{wv[0], wv[1]});
            const h4 wl = __builtin_bit_cast(h4, u2{wv[2], wv[3]});
            const f4 b2 = bias(Img::b_b2(0, mo));
#pragma unroll
            for (int t = 0; t < NT; ++t) tt[t] = mfma16h(wh, rh[t][0], b2);
#pragma unroll
            for (int t = 0; t < NT; ++t) tt[t] = mfma16h(wh, rl[t][0], tt[t]);
#pragma unroll
            for (int t = 0; t < NT; ++t) tt[t] = mfma16h(wl, rh[t][0], tt[t]);
        }
        {
            const u4 wv = grp[Img::g_w2(1, mo) * 64];
            const h4 wh = __builtin_bit_cast(h4, u2{wv[0], wv[1]});
            const h4 wl = __builtin_bit_cast(h4, u2{wv[2], wv[3]});
            const f4 b2 = bias(Img::b_b2(1, mo));
#pragma unroll
            for (int t = 0; t < NT; ++t) sv[t] = mfma16h(wh, rh[t][1], b2);
#pragma unroll
            for (int t = 0; t < NT; ++t) sv[t] = mfma16h(wh, rl[t][1], sv[t]);
#pragma unroll
            for (int t = 0; t < NT; ++t) sv[t] = mfma16h(wl, rh[t][1], sv[t]);
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

struct FlowF16Args {
    const float* z;
    const float* images;
    const float* fold;
    const float* ldc;
    float* z_out;
    float* sum_log_det;
    float* log_prob;
    int64_t Mz, Mp, N, slot;  // slot = floats between consecutive layer images in `images`
    int S;
    // fold == NULL: the kernel folds BatchNorm / Affine itself in its prologue (one launch less per call)
    const float* params;
    const float* bn_mean;
    const float* bn_alpha;
    int64_t pstride, stage_stride, affine_off;  // floats per stage; offset of the Affine block inside a stage
    int64_t low_off;                            // offset of RealNVP(lower) inside a stage
    int U;
    const float* iv;  // (7, D) constants of a fused ToInterval support layer, or NULL
};

template <int H, int NT>
__device__ __forceinline__ void apply_fold16(const float* fc, int q, f4 (&lo)[NT][(H + 15) / 16],
                                             f4 (&hi)[NT][(H + 15) / 16]) {
    constexpr int D = 2 * H;
    constexpr int HT = (H + 15) / 16;
#pragma unroll
    for (int mm = 0; mm < HT; ++mm) {
        const f4 al = *reinterpret_cast<const f4*>(fc + 16 * mm + 4 * q);
        const f4 bl = *reinterpret_cast<const f4*>(fc + D + 16 * mm + 4 * q);
        const f4 ah = *reinterpret_cast<const f4*>(fc + H + 16 * mm + 4 * q);
        const f4 bh = *reinterpret_cast<const f4*>(fc + D + H + 16 * mm + 4 * q);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[t][mm][j] = __builtin_fmaf(lo[t][mm][j], al[j], bl[j]);
                hi[t][mm][j] = __builtin_fmaf(hi[t][mm][j], ah[j], bh[j]);
            }
    }
}

// SS > 0: the number of stages is fixed at compile time and the layer loop is fully unrolled (every LDS
// offset of the operand reads becomes an immediate); SS = 0: run-time a.S.
template <int H, int L, bool INV, int NT, int NWAVES, int SS = 0>
__global__ void __launch_bounds__(NWAVES * 64)
flow_fused_f16_kernel(FlowF16Args a) {
    constexpr int D = 2 * H;
    constexpr int HT = (H + 15) / 16;
    typedef F16Image<H, L> Img;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nl = 2 * a.S;
    float* img = lds;
    float* fold = lds + nl * Img::FLOATS;
    int* qhead = reinterpret_cast<int*>(fold + nl * 2 * D);
    float* red = fold + nl * 2 * D + 4;  // [NWAVES] partial log-det constants
    float* ivc = red + NWAVES;           // [7][D] ToInterval constants (only read when a.iv)

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const int64_t m = grid_m();
    if (m >= (a.Mz > a.Mp ? a.Mz : a.Mp)) return;
    const int64_t mz = a.Mz == 1 ? 0 : m, mp = a.Mp == 1 ? 0 : m;
    {
        if (a.images) {
            for (int c = 0; c < nl; ++c) {
                const f4* isrc = reinterpret_cast<const f4*>(a.images + (mp * nl + c) * a.slot);
                f4* idst = reinterpret_cast<f4*>(img + c * Img::FLOATS);
                for (int i = threadIdx.x; i < Img::FLOATS / 4; i += NWAVES * 64) idst[i] = isrc[i];
            }
        } else {  // no prepared images: the workgroup's waves build the layers' operands straight into LDS
            for (int c = wave; c < nl; c += NWAVES)
                build_f16_image<H, L>(img + c * Img::FLOATS,
                                      a.params + mp * a.pstride + (c >> 1) * a.stage_stride + (c & 1) * a.low_off, a.U, lane);
        }
        if (a.fold) {
            const float* fsrc = a.fold + mp * (int64_t)nl * 2 * D;
            for (int i = threadIdx.x; i < nl * 2 * D; i += NWAVES * 64) fold[i] = fsrc[i];
        } else {
            // the arithmetic of flow_fold_kernel (coupling_mfma.hip): per layer c and feature d
            //   inverse: A = alpha_bn / e^a, B = mean_bn - shift A;  forward: A = e^a / alpha_bn, B = shift - mean_bn A
            // (a, shift: the Affine in front of odd layers), ldc = sum a - sum log alpha_bn
            const float* prow = a.params + mp * a.pstride;
            float acc = 0.f;
            for (int i = threadIdx.x; i < nl * D; i += NWAVES * 64) {
                const int c = i / D, d = i - c * D;
                const float alpha = a.bn_alpha[c * D + d], mu = a.bn_mean[c * D + d];
                acc -= logf(alpha);
                float ea = 1.f, shift = 0.f;
                if (c & 1) {
                    const float* ap = prow + (c >> 1) * a.stage_stride + a.affine_off;
                    const float av = ap[d];
                    acc += av;
                    ea = expf(av);
                    shift = ap[D + d];
                }
                float A, B;
                if (INV) {
                    A = alpha / ea;
                    B = mu - shift * A;
                } else {
                    A = ea / alpha;
                    B = shift - mu * A;
                }
                fold[c * 2 * D + d] = A;
                fold[c * 2 * D + D + d] = B;
            }
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
            if (lane == 0) red[wave] = acc;
        }
        if (threadIdx.x == 0) *qhead = NWAVES;
        if (a.iv)
            for (int i = threadIdx.x; i < 7 * D; i += NWAVES * 64) ivc[i] = a.iv[i];
    }
    const bool has_iv = a.iv != nullptr;
    __syncthreads();

    const float* zb = a.z + mz * a.N * D;
    float* zo = a.z_out ? a.z_out + m * a.N * D : nullptr;
    float* sldo = a.sum_log_det ? a.sum_log_det + m * a.N : nullptr;
    float* lpo = a.log_prob ? a.log_prob + m * a.N : nullptr;
    float ldc = 0.f;
    if (a.fold) ldc = a.ldc[mp];
    else {
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) ldc += red[w];
    }

    const int64_t ngroups = (a.N + 16 * NT - 1) / (16 * NT);
    const int64_t per_block = (ngroups + gridDim.x - 1) / gridDim.x;
    const int64_t g_lo = (int64_t)blockIdx.x * per_block;
    const int64_t g_hi = (g_lo + per_block < ngroups) ? g_lo + per_block : ngroups;
    int64_t grp = g_lo + wave;
    if (grp >= g_hi) return;

    f4 nlo[NT][HT], nhi[NT][HT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int64_t row = (grp * NT + t) * 16 + s;
        if (row >= a.N) row = a.N - 1;
        const float* zr = zb + row * D + 4 * q;
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            nlo[t][mm] = *reinterpret_cast<const f4*>(zr + 16 * mm);
            nhi[t][mm] = *reinterpret_cast<const f4*>(zr + H + 16 * mm);
        }
    }

    for (;;) {
        int nxt_off = 0;
        if (lane == 0) nxt_off = atomicAdd(qhead, 1);
        const int64_t nxt = g_lo + __builtin_amdgcn_readfirstlane(nxt_off);
        const bool has_next = nxt < g_hi;
        f4 lo[NT][HT], hi[NT][HT];
        float ssum[NT], ssup[NT];  // ssup: log-det of the fused support layer (natural log, this lane's features)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            ssum[t] = 0.f;
            ssup[t] = 0.f;
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                lo[t][mm] = nlo[t][mm];
                hi[t][mm] = nhi[t][mm];
            }
        }
        if (INV && has_iv) {  // the support layer is the first bijector of the inverse pass
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mm = 0; mm < HT; ++mm)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float o, l;
                        interval_fast<true>(lo[t][mm][j], ivc, D, 16 * mm + 4 * q + j, o, l);
                        lo[t][mm][j] = o;
                        ssup[t] += l;
                        interval_fast<true>(hi[t][mm][j], ivc, D, H + 16 * mm + 4 * q + j, o, l);
                        hi[t][mm][j] = o;
                        ssup[t] += l;
                    }
        }
        if (has_next) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                int64_t nrow = (nxt * NT + t) * 16 + s;
                if (nrow >= a.N) nrow = a.N - 1;
                const float* zr = zb + nrow * D + 4 * q;
#pragma unroll
                for (int mm = 0; mm < HT; ++mm) {
                    nlo[t][mm] = *reinterpret_cast<const f4*>(zr + 16 * mm);
                    nhi[t][mm] = *reinterpret_cast<const f4*>(zr + H + 16 * mm);
                }
            }
        }
        auto stage_inv = [&](int st) {
            const int c1 = 2 * st + 1, c0 = 2 * st;
            apply_fold16<H, NT>(fold + c1 * 2 * D, q, lo, hi);
            coupling_tile_f16<H, L, true, NT>(img + c1 * Img::FLOATS, lane, hi, lo, ssum);
            apply_fold16<H, NT>(fold + c0 * 2 * D, q, lo, hi);
            coupling_tile_f16<H, L, true, NT>(img + c0 * Img::FLOATS, lane, lo, hi, ssum);
        };
        auto stage_fwd = [&](int st) {
            const int c0 = 2 * st, c1 = 2 * st + 1;
            coupling_tile_f16<H, L, false, NT>(img + c0 * Img::FLOATS, lane, lo, hi, ssum);
            apply_fold16<H, NT>(fold + c0 * 2 * D, q, lo, hi);
            coupling_tile_f16<H, L, false, NT>(img + c1 * Img::FLOATS, lane, hi, lo, ssum);
            apply_fold16<H, NT>(fold + c1 * 2 * D, q, lo, hi);
        };
        if constexpr (SS > 0) {
#pragma unroll
            for (int i = 0; i < SS; ++i) {
                if (INV) stage_inv(SS - 1 - i); else stage_fwd(i);
            }
        } else if (INV) {
            for (int st = a.S - 1; st >= 0; --st) stage_inv(st);
        } else {
            for (int st = 0; st < a.S; ++st) stage_fwd(st);
        }
        if (!INV && has_iv) {  // ... and the last one of the forward pass
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mm = 0; mm < HT; ++mm)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float o, l;
                        interval_fast<false>(lo[t][mm][j], ivc, D, 16 * mm + 4 * q + j, o, l);
                        lo[t][mm][j] = o;
                        ssup[t] += l;
                        interval_fast<false>(hi[t][mm][j], ivc, D, H + 16 * mm + 4 * q + j, o, l);
                        hi[t][mm][j] = o;
                        ssup[t] += l;
                    }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t row = (grp * NT + t) * 16 + s;
            const bool row_ok = row < a.N;
            const float ld_tot = __builtin_fmaf(reduce_q(ssum[t]), kLn2, ldc) + (has_iv ? reduce_q(ssup[t]) : 0.f);
            if (INV && lpo) {
                float sq = 0.f;
#pragma unroll
                for (int mm = 0; mm < HT; ++mm)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sq = __builtin_fmaf(lo[t][mm][j], lo[t][mm][j], sq);
                        sq = __builtin_fmaf(hi[t][mm][j], hi[t][mm][j], sq);
                    }
                sq = reduce_q(sq);
                if (q == 0 && row_ok) lpo[row] = -0.5f * sq - (float)D * 0.91893853320467274178f - ld_tot;
            }
            if (sldo && q == 0 && row_ok) sldo[row] = ld_tot;
            if (zo && row_ok) {
                float* zr = zo + row * D + 4 * q;
#pragma unroll
                for (int mm = 0; mm < HT; ++mm) {
                    *reinterpret_cast<f4*>(zr + 16 * mm) = lo[t][mm];
                    *reinterpret_cast<f4*>(zr + H + 16 * mm) = hi[t][mm];
                }
            }
        }
        if (!has_next) break;
        grp = nxt;
    }
}

template <int H, int L>
static void launch_images16(const float* params, float* images, int64_t Mp, int S, int U, int64_t pstride,
                            int64_t slot, hipStream_t st) {
    hipLaunchKernelGGL((flow_images_f16_kernel<H, L>), grid_xm(2 * S, Mp), dim3(64), 0, st,
                       params, images, S, U, pstride, slot, Mp);
}

int launch_flow_images_f16(const float* params, float* images, int64_t Mp, int D, int S, int L, int U,
                           int64_t pstride, hipStream_t st) {
    const int64_t slot = mfma_image_floats(D, 3);
    if (D == 64) {
        if (L == 1) launch_images16<32, 1>(params, images, Mp, S, U, pstride, slot, st);
        else if (L == 2) launch_images16<32, 2>(params, images, Mp, S, U, pstride, slot, st);
        else launch_images16<32, 3>(params, images, Mp, S, U, pstride, slot, st);
    } else {
        if (L == 1) launch_images16<16, 1>(params, images, Mp, S, U, pstride, slot, st);
        else if (L == 2) launch_images16<16, 2>(params, images, Mp, S, U, pstride, slot, st);
        else launch_images16<16, 3>(params, images, Mp, S, U, pstride, slot, st);
    }
    return check_launch("flow_images_f16");
}

template <int H, int L, bool INV, int NT, int NW, int SS = 0>
static int launch16_t(const FlowF16Args& a, int64_t M, hipStream_t st) {
    const size_t smem = (size_t)2 * a.S * (F16Image<H, L>::FLOATS + 2 * 2 * H) * sizeof(float) + 16 + NW * sizeof(float) +
                        7 * 2 * H * sizeof(float);
    auto kern = flow_fused_f16_kernel<H, L, INV, NT, NW, SS>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return fail(TNF_ELAUNCH, "flow_fused_f16: cannot reserve %zu B of LDS", smem);
    const int64_t ngroups = (a.N + 16 * NT - 1) / (16 * NT);
    int64_t bx = (ngroups + NW - 1) / NW;
    int64_t cap = (256 + M - 1) / M;
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(kern, grid_xm(bx, M), dim3(NW * 64), smem, st, a);
    return TNF_OK;
}

template <int H, int L, bool INV>
static int launch16_v(const FlowF16Args& a, int64_t M, int variant, hipStream_t st) {
    if (L == 2) {
        if (variant == 11) return launch16_t<H, L, INV, 1, 8>(a, M, st);
        if (variant == 12) return launch16_t<H, L, INV, 1, 16>(a, M, st);
        if (variant == 13) return launch16_t<H, L, INV, 2, 16>(a, M, st);
        if (variant == 14) return launch16_t<H, L, INV, 2, 8>(a, M, st);  // run-time stage loop
    }
    // the reference's usual depth (num_stages = 4): layer loop fully unrolled, 3 % faster at D = 64
    if (a.S == 4) return launch16_t<H, L, INV, 2, 8, 4>(a, M, st);
    return launch16_t<H, L, INV, 2, 8>(a, M, st);
}

template <int H>
static int launch16_h(const FlowF16Args& a, int L, int inverse, int64_t M, int variant, hipStream_t st) {
    switch (L) {
        case 1: return inverse ? launch16_v<H, 1, true>(a, M, variant, st) : launch16_v<H, 1, false>(a, M, variant, st);
        case 2: return inverse ? launch16_v<H, 2, true>(a, M, variant, st) : launch16_v<H, 2, false>(a, M, variant, st);
        default: return inverse ? launch16_v<H, 3, true>(a, M, variant, st) : launch16_v<H, 3, false>(a, M, variant, st);
    }
}

int launch_flow_fused_f16(const float* z, const float* images, const float* fold, const float* ldc,
                          float* z_out, float* sum_log_det, float* log_prob, int64_t Mz, int64_t Mp,
                          int64_t N, int D, int S, int L, int U, int inverse, int variant, hipStream_t st,
                          const float* params, int64_t pstride, const float* bn_mean, const float* bn_alpha,
                          const float* interval_consts) {
    const int64_t M = Mz > Mp ? Mz : Mp;
    if (N <= 0) return TNF_OK;
    if ((!fold || !images) && (!params || !bn_mean || !bn_alpha))
        return fail(TNF_EINVAL, "flow_fused_f16: nothing to build the operands from");
    const FlowLayout fl = flow_layout(D, S, L, U);
    FlowF16Args a{z, images, fold, ldc, z_out, sum_log_det, log_prob, Mz, Mp, N, mfma_image_floats(D, 3), S,
                  params, bn_mean, bn_alpha, pstride, fl.stage, fl.p_up + fl.p_low, fl.p_up, U, interval_consts};
    int rc = (D == 64) ? launch16_h<32>(a, L, inverse, M, variant, st) : launch16_h<16>(a, L, inverse, M, variant, st);
    if (rc != TNF_OK) return rc;
    return check_launch("flow_fused_f16");
}

}  // namespace tnf
