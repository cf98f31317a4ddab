// MAF on the matrix pipe (float32): the transposed, accumulator-chained fp32-MFMA formulation of
// mfma_tile.h / coupling_wide.hip applied to the masked autoregressive twin MLPs of
// MAF (bijectors.py:597-806).  Differences to a coupling layer:
//   * the nets see ALL D features (DT = ceil(D/16) input tiles) and emit mu / alpha for all of them,
//     so the output accumulators of one pass are, lane for lane, the B operands of the next pass:
//     the D-1 sequential passes of the sampling direction (bijectors.py:752-754) chain in registers;
//   * weights are W * mask (masks shared by both nets, bijectors.py:698-740), there are no biases --
//     the activation folding of mfma_tile.h still yields accumulator initial values (column sums);
//   * D need not be a multiple of 4: feature loads / stores are per element when it is not.
// One workgroup builds the folded operand image of its parameter row in LDS (per-context rows for
// conditional flows), each wave walks 16-sample tiles.  Optional per-feature FMAs before (`pre`: the
// folded Affine^-1 . BatchNorm^-1 of NormFlow('AR').log_prob) and after (`post`: BatchNorm . Affine of the
// frozen forward), and the base-density epilogue, make NormFlow('AR') one kernel per call.
#include "maf_tile.h"
#include "support_math.h"

namespace tnf {

template <int DT, int UT, bool INV, bool VEC>
__global__ void __launch_bounds__(256)
maf_mfma_kernel(MafArgs a, MafLayout wl) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = a.D, U = a.U;
    float* cfold = lds;           // pre A|B, post A|B (4 * 16 * DT floats)
    float* ivc = lds + 4 * 16 * DT;  // ToInterval constants, 7 rows of 16 * DT (flags 0 = identity when absent)
    float* img = ivc + 7 * 16 * DT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const int64_t m = grid_m();
    if (m >= (a.Mz > a.Mp ? a.Mz : a.Mp)) return;
    const int64_t mz = a.Mz == 1 ? 0 : m, mp = a.Mp == 1 ? 0 : m;
    const bool has_pre = a.pre != nullptr, has_post = a.post != nullptr;
    constexpr int DP = 16 * DT;
    for (int i = threadIdx.x; i < 2 * DP; i += 256) {
        const int half = i / DP, f = i - half * DP;  // half 0: A (default 1), half 1: B (default 0)
        const bool ok = f < D;
        cfold[i] = (has_pre && ok) ? a.pre[mp * a.fold_stride + half * D + f] : (half == 0 ? 1.f : 0.f);
        cfold[2 * DP + i] = (has_post && ok) ? a.post[mp * a.fold_stride + half * D + f] : (half == 0 ? 1.f : 0.f);
    }
    for (int i = threadIdx.x; i < 7 * DP; i += 256) {
        const int row = i / DP, f = i - row * DP;
        ivc[i] = (a.iv && f < D) ? a.iv[row * D + f] : 0.f;
    }
    const bool has_iv = a.iv != nullptr;
    build_maf_image(img, a.params + mp * a.pstride, a.masks, wl, D, U, lane, wave, 4, a.bf16);
    __syncthreads();

    const float* wsrc = img + lane * 4;
    const float* bsrc = img + wl.NWG() * 256 + q * 4;
    auto wgrp = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(wsrc + g * 256); };
    auto bgrp = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(bsrc + g * 16); };
    const f4 zero = {0.f, 0.f, 0.f, 0.f};

    const float* zb = a.z + mz * a.N * D;
    float* zo = a.z_out ? a.z_out + m * a.N * D : nullptr;
    float* ldo = a.ld_out ? a.ld_out + m * a.N : nullptr;
    float* lpo = a.log_prob ? a.log_prob + m * a.N : nullptr;
    const float ldc = a.ldc ? a.ldc[mp] : 0.f;

    // twin nets on the tile's current iterate x -> mu (plain), al2 (alpha * log2 e)
    auto nets = [&](const f4 (&xin)[DT], f4 (&mu)[DT], f4 (&al2)[DT]) {
        f4 rt[UT], rs[UT], x[DT];
#pragma unroll
        for (int mm = 0; mm < DT; ++mm) x[mm] = rbf16_4(xin[mm], a.bf16);
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            f4 at = zero, as = zero;
#pragma unroll
            for (int mm = 0; mm < DT; ++mm) {
                const f4 wt = wgrp(wl.g_w0(0, ut, mm)), ws = wgrp(wl.g_w0(1, ut, mm));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    at = mfma4(wt[j], x[mm][j], at);
                    as = mfma4(ws[j], x[mm][j], as);
                }
            }
            rt[ut] = rbf16_4(sig2_4(at), a.bf16);
            rs[ut] = rbf16_4(sig2_4(as), a.bf16);
        }
        for (int l = 0; l < wl.L - 1; ++l) {
            f4 nt[UT], ns[UT];
#pragma unroll
            for (int uo = 0; uo < UT; ++uo) {
                f4 at = bgrp(wl.b_bh(l, 0, uo)), as = bgrp(wl.b_bh(l, 1, uo));
#pragma unroll
                for (int ui = 0; ui < UT; ++ui) {
                    const f4 wt = wgrp(wl.g_wh(l, 0, uo, ui)), ws = wgrp(wl.g_wh(l, 1, uo, ui));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        at = mfma4(wt[j], rt[ui][j], at);
                        as = mfma4(ws[j], rs[ui][j], as);
                    }
                }
                nt[uo] = rbf16_4(sig2_4(at), a.bf16);
                ns[uo] = rbf16_4(sig2_4(as), a.bf16);
            }
#pragma unroll
            for (int u = 0; u < UT; ++u) {
                rt[u] = nt[u];
                rs[u] = ns[u];
            }
        }
#pragma unroll
        for (int mo = 0; mo < DT; ++mo) {
            f4 tt = bgrp(wl.b_b2(0, mo)), sv = bgrp(wl.b_b2(1, mo));
#pragma unroll
            for (int ui = 0; ui < UT; ++ui) {
                const f4 wt = wgrp(wl.g_w2(0, mo, ui)), ws = wgrp(wl.g_w2(1, mo, ui));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tt = mfma4(wt[j], rt[ui][j], tt);
                    sv = mfma4(ws[j], rs[ui][j], sv);
                }
            }
            mu[mo] = tt;
            al2[mo] = sv;
        }
    };

    const int64_t ntiles = (a.N + 15) >> 4;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row = tile * 16 + s;
        const bool row_ok = row < a.N;
        const float* zr = zb + (row_ok ? row : a.N - 1) * D;
        float sup_ld = 0.f;  // log-det of the fused support layer (this lane's features)
        f4 x[DT];
#pragma unroll
        for (int mm = 0; mm < DT; ++mm) {
            const int f0 = 16 * mm + 4 * q;
            if (VEC) {
                x[mm] = f0 < D ? *reinterpret_cast<const f4*>(zr + f0) : zero;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) x[mm][j] = ld_sel(zr, f0 + j, f0 + j < D);
            }
            if (INV && has_iv) {  // support layer first in the inverse pass: ToInterval^-1 on the raw input
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float o, l;
                    interval_fast<true>(x[mm][j], ivc, DP, f0 + j, o, l);
                    x[mm][j] = o;
                    sup_ld += l;
                }
            }
            const f4 A = *reinterpret_cast<const f4*>(cfold + f0), B = *reinterpret_cast<const f4*>(cfold + DP + f0);
            x[mm] = x[mm] * A + B;  // padded features: 0 * 1 + 0
        }
        asm volatile("" ::: "memory");  // operand reads stay inside the tile loop

        f4 mu[DT], al2[DT], y[DT];
        if (INV) {  // one pass: z' = (z - mu) / exp(alpha)      (bijectors.py:758-764)
            nets(x, mu, al2);
#pragma unroll
            for (int mm = 0; mm < DT; ++mm)
#pragma unroll
                for (int j = 0; j < 4; ++j) y[mm][j] = (x[mm][j] - mu[mm][j]) * __builtin_amdgcn_exp2f(-al2[mm][j]);
        } else {  // D - 1 passes: z <- u exp(alpha(z)) + mu(z), z_0 = u      (bijectors.py:742-756)
#pragma unroll
            for (int mm = 0; mm < DT; ++mm) y[mm] = x[mm];
            for (int it = 0; it < D - 1; ++it) {
                nets(y, mu, al2);
#pragma unroll
                for (int mm = 0; mm < DT; ++mm)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        y[mm][j] = __builtin_fmaf(x[mm][j], __builtin_amdgcn_exp2f(al2[mm][j]), mu[mm][j]);
            }
            if (D == 1) {
#pragma unroll
                for (int mm = 0; mm < DT; ++mm) al2[mm] = zero;
            }
        }
        float ssum = 0.f;  // padded features carry alpha = 0 (zero weights, zero column sums)
#pragma unroll
        for (int mm = 0; mm < DT; ++mm) ssum += (al2[mm][0] + al2[mm][1]) + (al2[mm][2] + al2[mm][3]);
#pragma unroll
        for (int mm = 0; mm < DT; ++mm) {
            const int f0 = 16 * mm + 4 * q;
            const f4 A = *reinterpret_cast<const f4*>(cfold + 2 * DP + f0), B = *reinterpret_cast<const f4*>(cfold + 3 * DP + f0);
            y[mm] = y[mm] * A + B;
            if (!INV && has_iv) {  // support layer last in the forward pass
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float o, l;
                    interval_fast<false>(y[mm][j], ivc, DP, f0 + j, o, l);
                    y[mm][j] = o;
                    sup_ld += l;
                }
            }
        }
        float ld_tot = a.ld_sign * reduce_q(ssum) * kLn2 + reduce_q(sup_ld);
        if (a.add_ldc) ld_tot += ldc;
        if (lpo) {
            float sq = 0.f;
#pragma unroll
            for (int mm = 0; mm < DT; ++mm)
#pragma unroll
                for (int j = 0; j < 4; ++j) sq = __builtin_fmaf(y[mm][j], y[mm][j], sq);  // padded features are 0
            sq = reduce_q(sq);
            if (q == 0 && row_ok) lpo[row] = -0.5f * sq - (float)D * 0.91893853320467274178f - ld_tot;
        }
        if (ldo && q == 0 && row_ok) ldo[row] = ld_tot;
        if (zo && row_ok) {
            float* zw = zo + row * D;
#pragma unroll
            for (int mm = 0; mm < DT; ++mm) {
                const int f0 = 16 * mm + 4 * q;
                if (VEC) {
                    if (f0 < D) *reinterpret_cast<f4*>(zw + f0) = y[mm];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (f0 + j < D) zw[f0 + j] = y[mm][j];
                }
            }
        }
    }
}

// NormFlow('AR') = [MAF, BatchNorm, Affine] (density_estimator.py:271-274); parameter row [MAF | alpha (D) | shift (D)].
// Folds the two parameter-only bijectors into one per-feature FMA and the constant log-det:
//   inverse chain (log_prob):  x = ((z - shift) / e^a) alpha_bn + mean_bn = z A + B      -> `pre` of the MAF kernel
//   forward chain (frozen):    z = e^a (y - mean_bn) / alpha_bn + shift   = y A + B      -> `post`
//   ldc = sum a - sum log alpha_bn  (both directions: the reference's log-dets are the forward ones)
__global__ void __launch_bounds__(64)
ar_fold_kernel(const float* __restrict__ params, int64_t pstride, int64_t p_maf, const float* __restrict__ bn_mean,
               const float* __restrict__ bn_alpha, float* __restrict__ fold, float* __restrict__ ldc, int D, int inverse) {
    const int64_t m = blockIdx.x;
    const float* ap = params + m * pstride + p_maf;
    float acc = 0.f;
    for (int d = threadIdx.x; d < D; d += 64) {
        const float al = bn_alpha[d], mu = bn_mean[d], av = ap[d], sh = ap[D + d];
        acc += av - logf(al);
        const float ea = expf(av);
        float A, B;
        if (inverse) {
            A = al / ea;
            B = mu - sh * A;
        } else {
            A = ea / al;
            B = sh - mu * A;
        }
        fold[m * 2 * D + d] = A;
        fold[m * 2 * D + D + d] = B;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (threadIdx.x == 0) ldc[m] = acc;
}

int launch_ar_fold(const float* params, int64_t pstride, int64_t p_maf, const float* bn_mean, const float* bn_alpha,
                   float* fold, float* ldc, int64_t Mp, int D, int inverse, hipStream_t st) {
    hipLaunchKernelGGL(ar_fold_kernel, dim3((unsigned)Mp), dim3(64), 0, st, params, pstride, p_maf, bn_mean, bn_alpha,
                       fold, ldc, D, inverse);
    return check_launch("ar_fold");
}

static MafLayout maf_layout(int D, int L, int U) {
    MafLayout wl;
    wl.UT = (U + 15) / 16;
    wl.DT = (D + 15) / 16;
    wl.L = L;
    return wl;
}

bool maf_mfma_supported(int D, int L, int U) {
    if (D < 1 || D > 64 || L < 1 || L > 5 || U < 1 || U > 64) return false;
    const MafLayout wl = maf_layout(D, L, U);
    return (size_t)(11 * 16 * wl.DT + wl.floats()) * sizeof(float) <= 150 * 1024;
}

template <int DT, int UT>
static int launch_maf_du(const MafArgs& a, const MafLayout& wl, dim3 grid, size_t smem, hipStream_t st) {
    const bool vec = (a.D % 4) == 0;
#define TNF_MAF_GO(INV, VEC)                                                                                     \
    do {                                                                                                         \
        auto k = maf_mfma_kernel<DT, UT, INV, VEC>;                                                              \
        if (smem > 64 * 1024 &&                                                                                  \
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) \
            return fail(TNF_ELAUNCH, "maf_mfma: cannot reserve %zu B of LDS", smem);                             \
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, a, wl);                                                 \
    } while (0)
    if (a.inverse) {
        if (vec) TNF_MAF_GO(true, true); else TNF_MAF_GO(true, false);
    } else {
        if (vec) TNF_MAF_GO(false, true); else TNF_MAF_GO(false, false);
    }
#undef TNF_MAF_GO
    return TNF_OK;
}

template <int DT>
static int launch_maf_d(const MafArgs& a, const MafLayout& wl, dim3 grid, size_t smem, hipStream_t st) {
    switch (wl.UT) {
        case 1: return launch_maf_du<DT, 1>(a, wl, grid, smem, st);
        case 2: return launch_maf_du<DT, 2>(a, wl, grid, smem, st);
        case 3: return launch_maf_du<DT, 3>(a, wl, grid, smem, st);
        default: return launch_maf_du<DT, 4>(a, wl, grid, smem, st);
    }
}

int launch_maf_mfma(const MafArgs& a, hipStream_t st) {
    if (!maf_mfma_supported(a.D, a.L, a.U))
        return fail(TNF_EUNSUPPORTED, "maf_mfma: no kernel for D=%d L=%d U=%d", a.D, a.L, a.U);
    const int64_t M = a.Mz > a.Mp ? a.Mz : a.Mp;
    if (a.N <= 0) return TNF_OK;
    const MafLayout wl = maf_layout(a.D, a.L, a.U);
    const size_t smem = (size_t)(11 * 16 * wl.DT + wl.floats()) * sizeof(float);
    const int64_t ntiles = (a.N + 15) / 16;
    int64_t bx = (ntiles + 3) / 4;
    int64_t cap = 2048 / M;
    if (cap < 1) cap = 1;
    if (bx > cap) bx = cap;
    const dim3 grid = grid_xm(bx, M);
    int rc;
    switch (wl.DT) {
        case 1: rc = launch_maf_d<1>(a, wl, grid, smem, st); break;
        case 2: rc = launch_maf_d<2>(a, wl, grid, smem, st); break;
        case 3: rc = launch_maf_d<3>(a, wl, grid, smem, st); break;
        default: rc = launch_maf_d<4>(a, wl, grid, smem, st); break;
    }
    if (rc != TNF_OK) return rc;
    return check_launch("maf_mfma");
}

}  // namespace tnf
