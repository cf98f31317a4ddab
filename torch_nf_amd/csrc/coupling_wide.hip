// "Wide" per-layer MFMA kernel: the same transposed, accumulator-chained fp32-MFMA formulation
// as mfma_tile.h / coupling_mfma.hip, generalised to
//     D = 8..128 with D % 8 == 0     (half width H = D/2, HT = ceil(H/16) feature tiles, masked)
//     hidden width U <= 64           (UT = ceil(U/16) unit tiles)
//     num_layers L = 1..5            (run-time loop over the hidden layers)
// Templates fix only the tile counts (HT, UT); H, U and L are run-time and everything that does
// not fill a 16-wide tile is zero-padded in the operand image (weights) or masked (loads and
// stores).  The layer's operands live in LDS as a lane-ordered image
//     [group g][lane][4]   weights: layer 0 (net, ut, m), hidden (l, net, uo, ui), out (net, mo, ui)
//     [group g][q][4]      biases (accumulator initial values)
// built either by the workgroup itself (bijector-level calls) or by wide_images_kernel once per
// flow-level call.  Activation folding is identical to mfma_tile.h.
// One 16-sample tile per wave iteration; per tile and layer 4*(HT*UT + (L-1)*UT*UT + HT*UT) * 2
// MFMAs.  This is the coverage path for shapes without a narrow specialisation: HBM-bound for
// small U, fp32-pipe-bound for U >= 32.
#include "wide_tile.h"

namespace tnf {

__global__ void __launch_bounds__(64)
wide_images_kernel(const float* __restrict__ params, float* __restrict__ images, WideLayout wl, int D, int S,
                   int U, int64_t pstride, int64_t image_floats, int64_t Mp) {
    const int c = blockIdx.x;
    const int64_t m = grid_m();
    if (m >= Mp) return;
    const int64_t pc = coupling_num_params(D, wl.L, U, 1);
    const int64_t stage = 2 * pc + 2 * (int64_t)D;
    build_wide_image(images + (m * 2 * S + c) * image_floats, params + m * pstride + (c >> 1) * stage + (c & 1) * pc,
                     wl, D / 2, U, threadIdx.x);
}

template <int HT, int UT, bool INV>
__global__ void __launch_bounds__(256)
coupling_wide_kernel(MfmaLayerArgs a, WideLayout wl) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = a.D, H = D / 2, U = a.U;
    float* cfold = lds;            // pre A|B, post A|B: 4*D floats (D <= 128)
    float* img = lds + 4 * 128;    // operand image

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const int64_t m = grid_m();
    if (m >= (a.Mz > a.Mp ? a.Mz : a.Mp)) return;
    const int64_t mz = a.Mz == 1 ? 0 : m, mp = a.Mp == 1 ? 0 : m;

    const bool has_pre = a.pre != nullptr, has_post = a.post != nullptr;
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        cfold[i] = has_pre ? a.pre[mp * a.fold_stride + i] : 0.f;
        cfold[2 * D + i] = has_post ? a.post[mp * a.fold_stride + i] : 0.f;
    }
    if (a.image) {
        const f4* isrc = reinterpret_cast<const f4*>(a.image + mp * a.image_stride);
        f4* idst = reinterpret_cast<f4*>(img);
        for (int i = threadIdx.x; i < wl.floats() / 4; i += 256) idst[i] = isrc[i];
    } else if (wave == 0) {
        build_wide_image(img, a.params + mp * a.pstride, wl, H, U, lane);
    }
    __syncthreads();

    const float* wsrc = img + lane * 4;
    const float* bsrc = img + wl.NWG() * 256 + q * 4;
    auto wgrp = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(wsrc + g * 256); };
    auto bgrp = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(bsrc + g * 16); };

    const int c_off = a.upper ? 0 : H;
    const int t_off = a.upper ? H : 0;
    const float* zb = a.z + mz * a.N * D;
    float* zo = a.z_out ? a.z_out + m * a.N * D : nullptr;
    const float* ldi = a.ld_in ? a.ld_in + m * a.N : nullptr;
    float* ldo = a.ld_out ? a.ld_out + m * a.N : nullptr;
    float* lpo = a.log_prob ? a.log_prob + m * a.N : nullptr;
    const float ldc = a.ldc ? a.ldc[mp] : 0.f;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    bool fok[HT];  // does this lane's float4 of feature tile mm exist (H % 4 == 0)?
#pragma unroll
    for (int mm = 0; mm < HT; ++mm) fok[mm] = 16 * mm + 4 * q < H;

    const int64_t ntiles = (a.N + 15) >> 4;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row = tile * 16 + s;
        const bool row_ok = row < a.N;
        const float* zr = zb + (row_ok ? row : a.N - 1) * D + 4 * q;
        f4 x[HT], y[HT];
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            x[mm] = fok[mm] ? *reinterpret_cast<const f4*>(zr + c_off + 16 * mm) : zero;
            y[mm] = fok[mm] ? *reinterpret_cast<const f4*>(zr + t_off + 16 * mm) : zero;
        }
        const float ld_prev = (ldi && q == 0 && row_ok) ? ldi[row] : 0.f;
        if (has_pre) {
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                if (!fok[mm]) continue;
                const f4 ax = *reinterpret_cast<const f4*>(&cfold[c_off + 16 * mm + 4 * q]);
                const f4 bx = *reinterpret_cast<const f4*>(&cfold[D + c_off + 16 * mm + 4 * q]);
                const f4 ay = *reinterpret_cast<const f4*>(&cfold[t_off + 16 * mm + 4 * q]);
                const f4 by = *reinterpret_cast<const f4*>(&cfold[D + t_off + 16 * mm + 4 * q]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[mm][j] = __builtin_fmaf(x[mm][j], ax[j], bx[j]);
                    y[mm][j] = __builtin_fmaf(y[mm][j], ay[j], by[j]);
                }
            }
        }
        asm volatile("" ::: "memory");  // operand reads stay inside the tile loop

        // ---- layer 0 ----
        f4 rt[UT], rs[UT];
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            f4 at = bgrp(wl.b_b0(0, ut)), as = bgrp(wl.b_b0(1, ut));
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                const f4 wt = wgrp(wl.g_w0(0, ut, mm)), ws = wgrp(wl.g_w0(1, ut, mm));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    at = mfma4(wt[j], x[mm][j], at);
                    as = mfma4(ws[j], x[mm][j], as);
                }
            }
            rt[ut] = sig2_4(at);
            rs[ut] = sig2_4(as);
        }
        // ---- hidden layers ----
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
                nt[uo] = sig2_4(at);
                ns[uo] = sig2_4(as);
            }
#pragma unroll
            for (int u = 0; u < UT; ++u) {
                rt[u] = nt[u];
                rs[u] = ns[u];
            }
        }
        // ---- output layer + scale-shift ----
        float ssum = 0.f;
#pragma unroll
        for (int mo = 0; mo < HT; ++mo) {
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
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float s2 = sv[j];  // 0 for padded features (zero weights and bias)
                ssum += s2;
                if (INV) y[mo][j] = (y[mo][j] - tt[j]) * __builtin_amdgcn_exp2f(-s2);
                else y[mo][j] = __builtin_fmaf(y[mo][j], __builtin_amdgcn_exp2f(s2), tt[j]);
            }
        }
        if (has_post) {
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                if (!fok[mm]) continue;
                const f4 ax = *reinterpret_cast<const f4*>(&cfold[2 * D + c_off + 16 * mm + 4 * q]);
                const f4 bx = *reinterpret_cast<const f4*>(&cfold[3 * D + c_off + 16 * mm + 4 * q]);
                const f4 ay = *reinterpret_cast<const f4*>(&cfold[2 * D + t_off + 16 * mm + 4 * q]);
                const f4 by = *reinterpret_cast<const f4*>(&cfold[3 * D + t_off + 16 * mm + 4 * q]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[mm][j] = __builtin_fmaf(x[mm][j], ax[j], bx[j]);
                    y[mm][j] = __builtin_fmaf(y[mm][j], ay[j], by[j]);
                }
            }
        }
        const float sred = reduce_q(ssum) * kLn2;
        float ld_tot = __builtin_fmaf(a.ld_sign, sred, ld_prev);
        if (a.add_ldc) ld_tot += ldc;
        if (lpo) {
            float sq = 0.f;
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                if (!fok[mm]) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sq = __builtin_fmaf(x[mm][j], x[mm][j], sq);
                    sq = __builtin_fmaf(y[mm][j], y[mm][j], sq);
                }
            }
            sq = reduce_q(sq);
            if (q == 0 && row_ok) lpo[row] = -0.5f * sq - (float)D * 0.91893853320467274178f - ld_tot;
        }
        if (ldo && q == 0 && row_ok) ldo[row] = ld_tot;
        if (zo && row_ok) {
            float* zw = zo + row * D + 4 * q;
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                if (!fok[mm]) continue;
                *reinterpret_cast<f4*>(zw + c_off + 16 * mm) = x[mm];
                *reinterpret_cast<f4*>(zw + t_off + 16 * mm) = y[mm];
            }
        }
    }
}

int64_t wide_image_floats(int D, int L, int U) { return wide_layout(D, L, U).floats(); }

bool wide_supported(int D, int L, int U) {
    if (D < 8 || D > 128 || (D % 8) != 0) return false;
    if (L < 1 || L > 5 || U < 1 || U > 64) return false;
    return (4 * 128 + wide_layout(D, L, U).floats()) * sizeof(float) <= 150 * 1024;
}

int launch_wide_images(const float* params, float* images, int64_t Mp, int D, int S, int L, int U,
                       int64_t pstride, hipStream_t st) {
    const WideLayout wl = wide_layout(D, L, U);
    hipLaunchKernelGGL(wide_images_kernel, grid_xm(2 * S, Mp), dim3(64), 0, st, params, images, wl, D, S, U, pstride,
                       (int64_t)wl.floats(), Mp);
    return check_launch("wide_images");
}

template <int HT, int UT>
static int launch_wide_hu(const MfmaLayerArgs& a, const WideLayout& wl, dim3 grid, size_t smem, hipStream_t st) {
    if (a.inverse) {
        auto k = coupling_wide_kernel<HT, UT, true>;
        if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return fail(TNF_ELAUNCH, "coupling_wide: cannot reserve %zu B of LDS", smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, a, wl);
    } else {
        auto k = coupling_wide_kernel<HT, UT, false>;
        if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return fail(TNF_ELAUNCH, "coupling_wide: cannot reserve %zu B of LDS", smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, a, wl);
    }
    return TNF_OK;
}

template <int HT>
static int launch_wide_h(const MfmaLayerArgs& a, const WideLayout& wl, dim3 grid, size_t smem, hipStream_t st) {
    switch (wl.UT) {
        case 1: return launch_wide_hu<HT, 1>(a, wl, grid, smem, st);
        case 2: return launch_wide_hu<HT, 2>(a, wl, grid, smem, st);
        case 3: return launch_wide_hu<HT, 3>(a, wl, grid, smem, st);
        default: return launch_wide_hu<HT, 4>(a, wl, grid, smem, st);
    }
}

int launch_coupling_wide(const MfmaLayerArgs& a, hipStream_t st) {
    if (!wide_supported(a.D, a.L, a.U))
        return fail(TNF_EUNSUPPORTED, "coupling_wide: no kernel for D=%d L=%d U=%d", a.D, a.L, a.U);
    const int64_t M = a.Mz > a.Mp ? a.Mz : a.Mp;
    if (a.N <= 0) return TNF_OK;
    const WideLayout wl = wide_layout(a.D, a.L, a.U);
    const size_t smem = (size_t)(4 * 128 + wl.floats()) * sizeof(float);
    const int64_t ntiles = (a.N + 15) / 16;
    int64_t bx = (ntiles + 3) / 4;
    int64_t cap = 1024 / M;
    if (cap < 1) cap = 1;
    if (bx > cap) bx = cap;
    const dim3 grid = grid_xm(bx, M);
    int rc;
    switch (wl.HT) {
        case 1: rc = launch_wide_h<1>(a, wl, grid, smem, st); break;
        case 2: rc = launch_wide_h<2>(a, wl, grid, smem, st); break;
        case 3: rc = launch_wide_h<3>(a, wl, grid, smem, st); break;
        default: rc = launch_wide_h<4>(a, wl, grid, smem, st); break;
    }
    if (rc != TNF_OK) return rc;
    return check_launch("coupling_wide");
}

}  // namespace tnf
