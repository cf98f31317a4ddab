// One fused kernel per RealNVP coupling layer (the k = 2S design): float32 MFMA twin
// MLP + in-register scale-shift + wavefront-reduced log|det J|, with the adjacent
// BatchNorm / Affine bijectors folded in as per-feature FMAs and, on the last layer
// of a log_prob chain, the base Gaussian density.  HBM traffic per sample and launch:
// read D floats + write D floats (+ 8 B of running log-det), i.e. 520 B at D = 64.
#include <cstring>
#include "mfma_tile.h"
#include "tnf_common.h"

#ifndef TNF_LAYER_NTSTORE
#define TNF_LAYER_NTSTORE 0
#endif

namespace tnf {

// ---------------------------------------------------------------------------
// Fold the parameter-only bijectors between coupling layers into one FMA per feature.
// Forward order per stage (density_estimator.py:260-270): C_up, BN, C_low, BN, Affine.
// inverse chain: layer c is preceded by  [Affine^-1 (c odd)] then BN^-1[c]:
//      v -> ((v - shift)/e^a) * alpha + mu = v*(alpha/e^a) + (mu - shift*alpha/e^a)
// forward chain: layer c is followed by BN[c] then [Affine (c odd)]:
//      v -> ((v - mu)/alpha) * e^a + shift = v*(e^a/alpha) + (shift - mu*e^a/alpha)
// ldc[m] = sum of all parameter-only log-dets: sum_s sum(a_s) - sum_c sum(log alpha_c)
// (bijectors.py:293, 417).   fold layout: (Mp, 2S, 2, D) floats.
// ---------------------------------------------------------------------------
//
// chain = 1 (the in-place inverse chain of tnf_flow_log_prob_f32, one kernel per layer, walked c = 2S-1 .. 0): from the
// second kernel on a kernel stores only the half it transforms; the conditioner half stays in memory as it was BEFORE
// this kernel's fold, and that fold is owed to it.  It is paid by the next kernel, which transforms exactly that half:
// for c <= 2S-3 the constants of the transformed half of layer c are  fold_c o fold_{c+1}.  (520 -> 392 B per sample and
// launch at D = 64.)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void fold_consts(const float* __restrict__ p, const float* __restrict__ bn_mean,
                                            const float* __restrict__ bn_alpha, const FlowLayout& fl, int D, int c, int d,
                                            int inverse, float& A, float& B, float& ld) {
    const float alpha = bn_alpha[c * D + d], mu = bn_mean[c * D + d];
    ld = -logf(alpha);
    float ea = 1.f, shift = 0.f;
    if (c & 1) {
        const float* ap = p + (c >> 1) * fl.stage + fl.p_up + fl.p_low;
        const float a = ap[d];
        ld += a;
        ea = expf(a);
        shift = ap[D + d];
    }
    if (inverse) {
        A = alpha / ea;
        B = mu - shift * A;
    } else {
        A = ea / alpha;
        B = shift - mu * A;
    }
}

__global__ void __launch_bounds__(256)
flow_fold_kernel(const float* __restrict__ params, const float* __restrict__ bn_mean,
                 const float* __restrict__ bn_alpha, float* __restrict__ fold,
                 float* __restrict__ ldc, int D, int S, int L, int U, int64_t pstride, int inverse, int chain,
                 const int* __restrict__ gate) {
    if (gate && *gate == 0) return;  // a conditionally needed launch (tnf_set_launch_gate): nothing to do
    __shared__ float red[256];
    const int m = blockIdx.x;
    const float* p = params + (int64_t)m * pstride;
    const FlowLayout fl = flow_layout(D, S, L, U);
    float acc = 0.f;
    for (int idx = threadIdx.x; idx < 2 * S * D; idx += 256) {
        const int c = idx / D, d = idx - c * D;
        float A, B, ld;
        fold_consts(p, bn_mean, bn_alpha, fl, D, c, d, inverse, A, B, ld);
        acc += ld;
        // layer c transforms the upper half when c is even (density_estimator.py:260-270: RealNVP(upper) first)
        if (chain && c <= 2 * S - 3 && ((d >= D / 2) == ((c & 1) == 0))) {
            float A1, B1, ld1;
            fold_consts(p, bn_mean, bn_alpha, fl, D, c + 1, d, inverse, A1, B1, ld1);
            B = __builtin_fmaf(A, B1, B);
            A = A * A1;
        }
        float* f = fold + (((int64_t)m * 2 * S + c) * 2) * D;
        f[d] = A;
        f[D + d] = B;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) ldc[m] = red[0];
}

// ---------------------------------------------------------------------------
// Backward of the fold (inverse chain): from the accumulated gradients of the fold constants
//   A = alpha_bn / e^a,  B = mu_bn - shift * A        (layers with an Affine in front: c odd)
// to the Affine parameters:  g_a = -A dA + shift A dB,  g_shift = -A dB,
// plus the constant log-det term of log_prob = ... - ldc, ldc = sum_f a_f - sum log alpha_bn:
//   g_a -= sum over the samples of g_log_prob.
// g_fold: (Mp, 2S, 2, D).  One workgroup per parameter row.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
flow_fold_backward_kernel(const float* __restrict__ params, const float* __restrict__ bn_alpha,
                          const float* __restrict__ g_fold, const float* __restrict__ glp_sum,
                          float* __restrict__ g_params, int D, int S, int L, int U, int64_t pstride,
                          int64_t gpstride, const int* __restrict__ gate) {
    if (gate && *gate == 0) return;  // a conditionally needed launch (tnf_set_launch_gate): nothing to do
    const int64_t mp = blockIdx.x;
    const float sum_glp = glp_sum[mp];  // sum of g_log_prob over the samples that use this row
    const FlowLayout fl = flow_layout(D, S, L, U);
    const float* p = params + mp * pstride;
    float* gp = g_params + mp * gpstride;
    for (int idx = threadIdx.x; idx < S * D; idx += 256) {
        const int st = idx / D, d = idx - st * D;
        const int c = 2 * st + 1;
        const int64_t off = st * fl.stage + fl.p_up + fl.p_low;
        const float a = p[off + d], shift = p[off + D + d];
        const float A = bn_alpha[c * D + d] / expf(a);
        const float* gf = g_fold + ((mp * 2 * S + c) * 2) * D;
        const float dA = gf[d], dB = gf[D + d];
        atomicAdd(gp + off + d, -A * dA + shift * A * dB - sum_glp);
        atomicAdd(gp + off + D + d, -A * dB);
    }
}

int launch_flow_fold_backward(const float* params, const float* bn_alpha, const float* g_fold, const float* glp_sum,
                              float* g_params, int64_t Mp, int D, int S, int L, int U, int64_t pstride,
                              int64_t gpstride, hipStream_t st) {
    hipLaunchKernelGGL(flow_fold_backward_kernel, dim3((unsigned)Mp), dim3(256), 0, st, params, bn_alpha, g_fold,
                       glp_sum, g_params, D, S, L, U, pstride, gpstride, g_launch_gate);
    return check_launch("flow_fold_backward");
}

// ---------------------------------------------------------------------------
// Build every layer's MFMA operand image once per call (one wave per (layer, context)):
// the gather from the packed parameter row, the activation folding and the column sums
// happen here, so the hot kernels start from coalesced float4 loads of a lane-ordered
// image ((4HT + 2(L-1)) x 1 KB of weights + bias groups) instead of ~90 scattered loads.
// images: (Mp, 2S, image_floats) floats.
// ---------------------------------------------------------------------------
template <int H, int L>
__global__ void __launch_bounds__(64)
flow_images_kernel(const float* __restrict__ params, float* __restrict__ images, int S, int U,
                   int64_t pstride, int64_t image_floats, int64_t Mp, const int* __restrict__ gate) {
    if (gate && *gate == 0) return;  // a conditionally needed launch (tnf_set_launch_gate): nothing to do
    constexpr int D = 2 * H;
    const int c = blockIdx.x;
    const int64_t m = grid_m();
    if (m >= Mp) return;
    const int64_t pc = coupling_num_params(D, L, U, 1);
    const int64_t stage = 2 * pc + 2 * D;
    LayerW<H, L> w;
    load_layer_w<H, L>(w, params + m * pstride + (c >> 1) * stage + (c & 1) * pc, U, threadIdx.x);
    store_layer_image<H, L>(images + (m * 2 * S + c) * image_floats, w, threadIdx.x);
}

int64_t mfma_image_floats(int D, int L) {
    const int HT = (D / 2 + 15) / 16;
    return (int64_t)(4 * HT + 2 * (L - 1)) * 256 + (2 + 2 * (L - 1) + 2 * HT) * 16;
}

int launch_flow_images(const float* params, float* images, int64_t Mp, int D, int S, int L, int U, int64_t pstride,
                       hipStream_t st) {
    const dim3 grid = grid_xm(2 * S, Mp);
    const int64_t fl = mfma_image_floats(D, L);
#define TNF_IMG(HH, LL) \
    hipLaunchKernelGGL((flow_images_kernel<HH, LL>), grid, dim3(64), 0, st, params, images, S, U, pstride, fl, Mp, g_launch_gate)
    if (D == 64) {
        if (L == 1) TNF_IMG(32, 1); else if (L == 2) TNF_IMG(32, 2); else TNF_IMG(32, 3);
    } else {
        if (L == 1) TNF_IMG(16, 1); else if (L == 2) TNF_IMG(16, 2); else TNF_IMG(16, 3);
    }
#undef TNF_IMG
    return check_launch("flow_images");
}

int launch_flow_prep(const float* params, const float* bn_mean, const float* bn_alpha, float* fold,
                     float* ldc, float* images, int64_t Mp, int D, int S, int L, int U,
                     int64_t pstride, int inverse, hipStream_t st, int chain) {
    hipLaunchKernelGGL(flow_fold_kernel, dim3((unsigned)Mp), dim3(256), 0, st, params, bn_mean,
                       bn_alpha, fold, ldc, D, S, L, U, pstride, inverse, chain, g_launch_gate);
    if (!images) return check_launch("flow_prep");  // wide shapes build their own images
    return launch_flow_images(params, images, Mp, D, S, L, U, pstride, st);
}

// ---------------------------------------------------------------------------
// The per-layer kernel.  256 threads = 4 waves; each wave keeps the layer's MFMA
// operands in registers and walks 16-sample tiles grid-stride, with the next tile's
// loads in flight during the current tile's arithmetic.
// ---------------------------------------------------------------------------
template <int H, int L, bool INV, int NT, bool LDSOP>
__global__ void __launch_bounds__(256)
coupling_mfma_kernel(MfmaLayerArgs a) {
    if (a.gate && *a.gate == 0) return;  // a conditionally needed launch (tnf_set_launch_gate): nothing to do
    constexpr int D = 2 * H;
    constexpr int HT = (H + 15) / 16;
    typedef LdsLayerImage<H, L> Img;
    __shared__ __attribute__((aligned(16))) float cfold[4 * D + (LDSOP ? Img::FLOATS : 0)];  // pre A|B, post A|B, [image]

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    // workgroup per context (default) or, for many contexts with a handful of samples each
    // (SNPE-style N = 1 calls), one context per wave: every wave gathers its own operands anyway
    const int64_t Mtot = a.Mz > a.Mp ? a.Mz : a.Mp;
    int64_t m = a.wave_m ? (int64_t)blockIdx.x * 4 + wave : grid_m();
    const bool m_ok = m < Mtot;
    if (!m_ok) {
        if (!a.wave_m) return;  // whole workgroup out of range
        m = Mtot - 1;           // keep the wave alive for the barrier below; it gets no samples
    }
    const int64_t Nv = m_ok ? a.N : 0;
    const int64_t mz = a.Mz == 1 ? 0 : m, mp = a.Mp == 1 ? 0 : m;

    const bool has_pre = a.pre != nullptr, has_post = a.post != nullptr;
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        cfold[i] = has_pre ? a.pre[mp * a.fold_stride + i] : 0.f;
        cfold[2 * D + i] = has_post ? a.post[mp * a.fold_stride + i] : 0.f;
    }

    // Operands: registers (LayerW, 72 VGPRs, no LDS traffic in the loop) or an LDS copy of the
    // prepared image (frees the registers for a third / fourth wave per SIMD).
    LayerW<H, L> w;
    if constexpr (LDSOP) {
        const f4* isrc = reinterpret_cast<const f4*>(a.image + mp * a.image_stride);
        f4* idst = reinterpret_cast<f4*>(cfold + 4 * D);
        for (int i = threadIdx.x; i < Img::FLOATS / 4; i += 256) idst[i] = isrc[i];
    } else if (a.image) {  // flow-level chains: lane-ordered image prepared by flow_images_kernel
        const LdsOperands<H, L> src(a.image + mp * a.image_stride, lane);
#pragma unroll
        for (int net = 0; net < 2; ++net) {
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                const f4 v0 = src.w0(net, mm), v2 = src.w2(net, mm);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    w.w0[net][mm * 4 + j] = v0[j];
                    w.w2[net][mm][j] = v2[j];
                }
                w.b2[net][mm] = src.b2(net, mm);
            }
#pragma unroll
            for (int l = 0; l < L - 1; ++l) {
                const f4 vh = src.wh(l, net);
#pragma unroll
                for (int j = 0; j < 4; ++j) w.wh[l][net][j] = vh[j];
                w.bh[l][net] = src.bh(l, net);
            }
            w.b0[net] = src.b0(net);
        }
    } else {  // bijector-level call: gather straight from the packed parameter row
        load_layer_w<H, L>(w, a.params + mp * a.pstride, a.U, lane);
    }
    const RegOperands<H, L> rop{w};
    const LdsOperands<H, L> lop(cfold + 4 * D, lane);
    __syncthreads();

    const int c_off = a.upper ? 0 : H;
    const int t_off = a.upper ? H : 0;
    const float* zb = a.z + mz * a.N * D;
    float* zo = a.z_out ? a.z_out + m * a.N * D : nullptr;
    const float* ldi = a.ld_in ? a.ld_in + m * a.N : nullptr;
    float* ldo = a.ld_out ? a.ld_out + m * a.N : nullptr;
    float* lpo = a.log_prob ? a.log_prob + m * a.N : nullptr;
    const float ldc = a.ldc ? a.ldc[mp] : 0.f;

    // a "group" = NT consecutive 16-sample tiles handled by one wave per iteration
    const int64_t ngroups = (Nv + 16 * NT - 1) / (16 * NT);
    const int64_t gstride = a.wave_m ? 1 : (int64_t)gridDim.x * 4;
    int64_t grp = a.wave_m ? 0 : (int64_t)blockIdx.x * 4 + wave;
    if (grp >= ngroups) return;

    f4 nx[NT][HT], ny[NT][HT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int64_t row = (grp * NT + t) * 16 + s;
        if (row >= a.N) row = a.N - 1;
        const float* zr = zb + row * D + 4 * q;
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            nx[t][mm] = *reinterpret_cast<const f4*>(zr + c_off + 16 * mm);
            ny[t][mm] = *reinterpret_cast<const f4*>(zr + t_off + 16 * mm);
        }
    }

    for (; grp < ngroups; grp += gstride) {
        f4 x[NT][HT], y[NT][HT];
        float ld_prev[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                x[t][mm] = nx[t][mm];
                y[t][mm] = ny[t][mm];
            }
            const int64_t row = (grp * NT + t) * 16 + s;
            ld_prev[t] = (ldi && q == 0 && row < a.N) ? ldi[row] : 0.f;
        }
        // next group's loads stay in flight during this group's arithmetic
        if (grp + gstride < ngroups) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                int64_t nrow = ((grp + gstride) * NT + t) * 16 + s;
                if (nrow >= a.N) nrow = a.N - 1;
                const float* zr = zb + nrow * D + 4 * q;
#pragma unroll
                for (int mm = 0; mm < HT; ++mm) {
                    nx[t][mm] = *reinterpret_cast<const f4*>(zr + c_off + 16 * mm);
                    ny[t][mm] = *reinterpret_cast<const f4*>(zr + t_off + 16 * mm);
                }
            }
        }
        if (has_pre) {
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                const f4 ax = *reinterpret_cast<const f4*>(&cfold[c_off + 16 * mm + 4 * q]);
                const f4 bx = *reinterpret_cast<const f4*>(&cfold[D + c_off + 16 * mm + 4 * q]);
                const f4 ay = *reinterpret_cast<const f4*>(&cfold[t_off + 16 * mm + 4 * q]);
                const f4 by = *reinterpret_cast<const f4*>(&cfold[D + t_off + 16 * mm + 4 * q]);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        x[t][mm][j] = __builtin_fmaf(x[t][mm][j], ax[j], bx[j]);
                        y[t][mm][j] = __builtin_fmaf(y[t][mm][j], ay[j], by[j]);
                    }
            }
        }
        float ssum[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) ssum[t] = 0.f;
        if constexpr (LDSOP) {
            asm volatile("" ::: "memory");  // keep the operand reads in the loop (no hoisting back into VGPRs)
            coupling_tile<H, L, INV, NT>(lop, x, y, ssum);
        }
        else
            coupling_tile<H, L, INV, NT>(rop, x, y, ssum);
        if (has_post) {
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                const f4 ax = *reinterpret_cast<const f4*>(&cfold[2 * D + c_off + 16 * mm + 4 * q]);
                const f4 bx = *reinterpret_cast<const f4*>(&cfold[3 * D + c_off + 16 * mm + 4 * q]);
                const f4 ay = *reinterpret_cast<const f4*>(&cfold[2 * D + t_off + 16 * mm + 4 * q]);
                const f4 by = *reinterpret_cast<const f4*>(&cfold[3 * D + t_off + 16 * mm + 4 * q]);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        x[t][mm][j] = __builtin_fmaf(x[t][mm][j], ax[j], bx[j]);
                        y[t][mm][j] = __builtin_fmaf(y[t][mm][j], ay[j], by[j]);
                    }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t row = (grp * NT + t) * 16 + s;
            const bool row_ok = row < a.N;
            const float sred = reduce_q(ssum[t]) * kLn2;  // the tile code sums s*log2(e)
            float ld_tot = __builtin_fmaf(a.ld_sign, sred, ld_prev[t]);
            if (a.add_ldc) ld_tot += ldc;
            if (lpo) {
                float sq = 0.f;
#pragma unroll
                for (int mm = 0; mm < HT; ++mm)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sq = __builtin_fmaf(x[t][mm][j], x[t][mm][j], sq);
                        sq = __builtin_fmaf(y[t][mm][j], y[t][mm][j], sq);
                    }
                sq = reduce_q(sq);
                // density_estimator.py:413-416
                if (q == 0 && row_ok)
                    lpo[row] = -0.5f * sq - (float)D * 0.91893853320467274178f - ld_tot;
            }
            if (ldo && q == 0 && row_ok) ldo[row] = ld_tot;
            if (zo && row_ok) {
                float* zr = zo + row * D + 4 * q;
#pragma unroll
                for (int mm = 0; mm < HT; ++mm) {
#if TNF_LAYER_NTSTORE  // around the caches: see TNF2_RANGE_NTMEM in flow_fused2.hip
                    if (!a.skip_cond_store) __builtin_nontemporal_store(x[t][mm], reinterpret_cast<f4*>(zr + c_off + 16 * mm));
                    __builtin_nontemporal_store(y[t][mm], reinterpret_cast<f4*>(zr + t_off + 16 * mm));
#else
                    if (!a.skip_cond_store) *reinterpret_cast<f4*>(zr + c_off + 16 * mm) = x[t][mm];
                    *reinterpret_cast<f4*>(zr + t_off + 16 * mm) = y[t][mm];
#endif
                }
            }
        }
    }
}

bool mfma_supported(int D, int L, int U) {
    if (D != 32 && D != 64) return false;
    if (L < 1 || L > 3) return false;
    return U >= 1 && U <= 16;
}

// Per-layer chain variants (g_layer_variant, TNF_OPT_LAYER_VARIANT):
//   10 (default): split-f16 tile code of the whole-flow kernel, one coupling layer per launch (flow_fused2.hip);
//   10 + n: n layers per launch;   0..3: this file's fp32-MFMA kernel --
//   0: operands in registers, 2 tiles per wave iteration     1: operands in LDS, 2 tiles
//   2: operands in LDS, 1 tile                               3: operands in registers, 1 tile
thread_local int g_layer_variant = 10;
thread_local int g_train_bwd_fp32 = 0;

template <int H, int L, bool INV, int NT, bool LDSOP>
static void launch_k(const MfmaLayerArgs& a, int64_t M, hipStream_t st) {
    const int64_t ngroups = (a.N + 16 * NT - 1) / (16 * NT);
    if (!LDSOP && !a.pre && !a.post && !a.image && M >= 8 && ngroups <= 2) {
        MfmaLayerArgs b = a;
        b.wave_m = 1;
        hipLaunchKernelGGL((coupling_mfma_kernel<H, L, INV, NT, LDSOP>), dim3((unsigned)((M + 3) / 4)), dim3(256),
                           0, st, b);
        return;
    }
    int64_t bx = (ngroups + 3) / 4;
    int64_t cap = 2048 / M;
    if (cap < 1) cap = 1;
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL((coupling_mfma_kernel<H, L, INV, NT, LDSOP>), grid_xm(bx, M), dim3(256),
                       0, st, a);
}

template <int H, int L, bool INV>
static void launch_v(const MfmaLayerArgs& a, int64_t M, hipStream_t st) {
    const int v = (a.image && g_layer_variant < 10) ? g_layer_variant : 0;  // the LDS variants need the prepared image
    if (L == 2 && v == 1) launch_k<H, L, INV, 2, true>(a, M, st);
    else if (L == 2 && v == 2) launch_k<H, L, INV, 1, true>(a, M, st);
    else if (L == 2 && v == 3) launch_k<H, L, INV, 1, false>(a, M, st);
    else launch_k<H, L, INV, 2, false>(a, M, st);
}

template <int H>
static void launch_h(const MfmaLayerArgs& a, int64_t M, hipStream_t st) {
    switch (a.L) {
        case 1: a.inverse ? launch_v<H, 1, true>(a, M, st) : launch_v<H, 1, false>(a, M, st); break;
        case 2: a.inverse ? launch_v<H, 2, true>(a, M, st) : launch_v<H, 2, false>(a, M, st); break;
        default: a.inverse ? launch_v<H, 3, true>(a, M, st) : launch_v<H, 3, false>(a, M, st); break;
    }
}

int launch_coupling_mfma(const MfmaLayerArgs& a, hipStream_t st) {
    if (!mfma_supported(a.D, a.L, a.U))
        return fail(TNF_EUNSUPPORTED, "coupling_mfma: no kernel for D=%d L=%d U=%d", a.D, a.L, a.U);
    const int64_t M = a.Mz > a.Mp ? a.Mz : a.Mp;
    if (a.N <= 0) return TNF_OK;
    if (a.D == 64) launch_h<32>(a, M, st);
    else launch_h<16>(a, M, st);
    return check_launch("coupling_mfma");
}

// ---------------------------------------------------------------------------
// NormFlow.forward with freeze_bn=False and no autograd (the reference's default sampling call,
// density_estimator.py:374-388 with bijectors.py:401-415): every BatchNorm normalises with the statistics of the
// batch in front of it, so a layer's output must be complete before the next one starts.  The chain keeps that
// order but folds each BatchNorm (+ the Affine behind the second one of a stage) into the NEXT coupling
// kernel's load stage:  per layer  coupling kernel (pre-fold) -> bn_stats -> bn_finalize -> fold constants,
// and one elementwise pass at the end for the last fold.  The statistics land in bn_mean_out / bn_alpha_out
// (2S, D) exactly as BatchNorm.forward(use_last=False) would cache them.
// ---------------------------------------------------------------------------
// fold after layer c (forward order): z -> (z - mean)/alpha [-> e^a z + shift when c is odd]; ldc[m] += its log-det
__global__ void __launch_bounds__(256)
flow_batch_fold_kernel(const float* __restrict__ params, int64_t pstride, int64_t affine_off, const float* __restrict__ mean,
                       const float* __restrict__ rstd, const float* __restrict__ ld_bn, float* __restrict__ fold,
                       float* __restrict__ ldc, int D, int has_affine, int first) {
    const int64_t m = blockIdx.x;
    const float* ap = params + m * pstride + affine_off;
    float acc = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float rs = rstd[d], mu = mean[d];
        float A = rs, B = -mu * rs;
        if (has_affine) {
            const float av = ap[d], ea = expf(av);
            acc += av;
            A = ea * rs;
            B = ap[D + d] - mu * A;
        }
        fold[m * 2 * D + d] = A;
        fold[m * 2 * D + D + d] = B;
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) ldc[m] = (first ? 0.f : ldc[m]) + red[0] + *ld_bn;
}

// The same with the BatchNorm finalisation in front (bn_finalize_kernel's arithmetic): mean / alpha of the layer's batch
// statistics from moments = [sum (D) | sum of squares (D) | row count] doubles -- every workgroup derives them itself
// (D values), workgroup 0 hands them out.  One launch instead of two on a chain whose small kernels are launch-bound.
// (Tried and dropped: the statistics as per-wave column sums inside the coupling kernel.  Its 182 VGPRs + 64 AGPRs leave
// no room for accumulators at two waves per SIMD: with doubles the kernel falls to one wave (103 us instead of 55), with
// floats and one tile per wave it takes 71 us, and reducing 8,192 partial rows costs more than the 31 us pass it saves.)
__global__ void __launch_bounds__(256)
flow_batch_finalize_fold_kernel(const float* __restrict__ params, int64_t pstride, int64_t affine_off,
                                const double* __restrict__ moments, float eps, float* __restrict__ mean_out,
                                float* __restrict__ alpha_out, float* __restrict__ fold, float* __restrict__ ldc, int D,
                                int has_affine, int first) {
    const int64_t m = blockIdx.x;
    const float* ap = params + m * pstride + affine_off;
    const double rows = moments[2 * D];
    float acc = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const double mud = moments[d] / rows;
        double var_b = moments[D + d] / rows - mud * mud;
        if (var_b < 0.0) var_b = 0.0;
        const double ad = sqrt(var_b + (double)eps);
        const float mu = (float)mud, rs = (float)(1.0 / ad);
        if (m == 0) {
            mean_out[d] = mu;
            alpha_out[d] = (float)ad;
        }
        acc -= logf((float)ad);  // BatchNorm's log-det
        float A = rs, B = -mu * rs;
        if (has_affine) {
            const float av = ap[d], ea = expf(av);
            acc += av;
            A = ea * rs;
            B = ap[D + d] - mu * A;
        }
        fold[m * 2 * D + d] = A;
        fold[m * 2 * D + D + d] = B;
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) ldc[m] = (first ? 0.f : ldc[m]) + red[0];
}

// z_out <- z A[m] + B[m] (z_out may be z), sum_log_det[m][n] += ldc[m]
__global__ void __launch_bounds__(256)
flow_fold_apply_kernel(const float* z, float* z_out, float* __restrict__ sld, const float* __restrict__ fold,
                       const float* __restrict__ ldc, int64_t Mp, int64_t N, int D) {
    const int64_t m = grid_m();
    const int64_t mp = Mp == 1 ? 0 : m;
    const float* A = fold + mp * 2 * D;
    const float* B = A + D;
    const float* zr = z + m * N * D;
    float* zo = z_out + m * N * D;
    const int64_t total = N * D;
    const int64_t step = (int64_t)gridDim.x * 1024;
    for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < total; i0 += 4 * step) {
        f4 v[4];  // four 16-byte accesses in flight per lane; D % 4 == 0: the four values of one share a row
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * step;
            if (i < total) v[u] = *reinterpret_cast<const f4*>(zr + i);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * step;
            if (i < total) {
                const int d = (int)(i % D);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[u][j] = __builtin_fmaf(v[u][j], A[d + j], B[d + j]);
                *reinterpret_cast<f4*>(zo + i) = v[u];
            }
        }
    }
    const float c = ldc[mp];
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) sld[m * N + n] += c;
}

static int64_t fb_head_bytes(int64_t Mp, int D) { return (((Mp * 2 * D + Mp + D + 1) * 4 + 15) / 16) * 16; }
int64_t flow_forward_batch_workspace(int64_t Mp, int D, int S, int L) {
    // fold (Mp, 2, D) | ldc (Mp) | rstd (D) | ld_bn (1) | moments (2 D + 2 doubles) | operand images (Mp, 2S, image)
    return fb_head_bytes(Mp, D) + (2 * (int64_t)D + 2) * 8 + Mp * 2 * S * mfma_image_floats(D, L) * 4;
}

// The chain in steps, so that a sample-sharded caller can put a collective between a layer's local moments and the
// statistics derived from them (one all-reduce of [sum, sum of squares, count] per BatchNorm layer):
//   begin                    operand images of every layer (workspace)
//   layer c                  coupling layer c with the fold behind layer c-1 in its load stage; then the LOCAL moments
//                            of its output: moments = [sum (D) | sum of squares (D) | row count] doubles, overwritten
//   (caller: sum `moments` over the ranks that share the batch)
//   fold c                   statistics from the (global) moments -> bn_mean_out / bn_alpha_out row c, fold constants
//                            for the next layer's load stage, constant log-det
//   end                      the last fold as an elementwise pass; sum_log_det += constant log-dets
// tnf_flow_forward_batch_f32 is exactly begin, (layer, fold) x 2S, end with the local moments.
struct FbWs {
    float *fold, *ldc, *rstd, *ld_bn, *images;
    double* moments;
};
static FbWs fb_ws(void* ws, int64_t Mp, int D) {
    FbWs w;
    w.fold = reinterpret_cast<float*>(ws);
    w.ldc = w.fold + Mp * 2 * D;
    w.rstd = w.ldc + Mp;
    w.ld_bn = w.rstd + D;
    w.moments = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + fb_head_bytes(Mp, D));
    w.images = reinterpret_cast<float*>(w.moments + 2 * D + 2);
    return w;
}

int flow_forward_batch_begin(const float* params, int64_t Mp, int D, int S, int L, int U, int64_t pstride, void* ws,
                             hipStream_t st) {
    if (!mfma_supported(D, L, U)) return fail(TNF_EUNSUPPORTED, "flow_forward_batch: D=%d L=%d U=%d", D, L, U);
    return launch_flow_images(params, fb_ws(ws, Mp, D).images, Mp, D, S, L, U, pstride, st);
}

int flow_forward_batch_layer(int c, const float* z_in, const float* params, float* z_out, float* sum_log_det,
                             double* moments, int64_t M, int64_t Mp, int64_t N, int D, int S, int L, int U, int64_t pstride,
                             void* ws, hipStream_t st) {
    const FbWs w = fb_ws(ws, Mp, D);
    const FlowLayout fl = flow_layout(D, S, L, U);
    const int nl = 2 * S;
    const int64_t img_floats = mfma_image_floats(D, L);
    if (!moments) moments = w.moments;
    if (N > 0) {
        MfmaLayerArgs a = {};
        a.z = z_in;
        a.z_out = z_out;
        a.params = params + (c >> 1) * fl.stage + ((c & 1) ? fl.p_up : 0);
        a.pstride = pstride;
        a.image = w.images + (int64_t)c * img_floats;
        a.image_stride = (int64_t)nl * img_floats;
        a.pre = c == 0 ? nullptr : w.fold;
        a.fold_stride = 2 * (int64_t)D;
        a.ld_in = c == 0 ? nullptr : sum_log_det;
        a.ld_out = sum_log_det;
        a.ld_sign = 1.f;
        a.Mz = M; a.Mp = Mp; a.N = N;
        a.D = D; a.L = L; a.U = U; a.upper = (c & 1) ? 0 : 1; a.inverse = 0;
        int rc = launch_coupling_mfma(a, st);
        if (rc) return rc;
    }
    return launch_bn_moments(z_out, moments, M * N, D, st);  // an empty shard contributes zeros
}

int flow_forward_batch_fold(int c, const float* params, const double* moments, float* bn_mean_out, float* bn_alpha_out,
                            int64_t Mp, int D, int S, int L, int U, int64_t pstride, float eps, void* ws, hipStream_t st) {
    const FbWs w = fb_ws(ws, Mp, D);
    const FlowLayout fl = flow_layout(D, S, L, U);
    if (!moments) moments = w.moments;
    hipLaunchKernelGGL(flow_batch_finalize_fold_kernel, dim3((unsigned)Mp), dim3(256), 0, st, params, pstride,
                       (int64_t)(c >> 1) * fl.stage + fl.p_up + fl.p_low, moments, eps, bn_mean_out + (int64_t)c * D,
                       bn_alpha_out + (int64_t)c * D, w.fold, w.ldc, D, c & 1, c == 0);
    return check_launch("flow_forward_batch_fold");
}

int flow_forward_batch_end(float* z_out, float* sum_log_det, int64_t M, int64_t Mp, int64_t N, int D, void* ws,
                           hipStream_t st) {
    if (N <= 0) return TNF_OK;
    const FbWs w = fb_ws(ws, Mp, D);
    int64_t nb = (N * D / 4 + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(flow_fold_apply_kernel, grid_xm(nb, M), dim3(256), 0, st, z_out, z_out, sum_log_det, w.fold, w.ldc, Mp,
                       N, D);
    return check_launch("flow_forward_batch_end");
}

int launch_flow_forward_batch(const float* omega, const float* params, float* z_out, float* sum_log_det,
                              float* bn_mean_out, float* bn_alpha_out, int64_t M, int64_t Mp, int64_t N, int D, int S,
                              int L, int U, int64_t pstride, float eps, void* ws, hipStream_t st) {
    if (!mfma_supported(D, L, U)) return fail(TNF_EUNSUPPORTED, "flow_forward_batch: D=%d L=%d U=%d", D, L, U);
    if (N <= 0) return TNF_OK;
    int rc = flow_forward_batch_begin(params, Mp, D, S, L, U, pstride, ws, st);
    if (rc) return rc;
    for (int c = 0; c < 2 * S; ++c) {
        rc = flow_forward_batch_layer(c, c == 0 ? omega : z_out, params, z_out, sum_log_det, nullptr, M, Mp, N, D, S, L, U,
                                      pstride, ws, st);
        if (rc) return rc;
        rc = flow_forward_batch_fold(c, params, nullptr, bn_mean_out, bn_alpha_out, Mp, D, S, L, U, pstride, eps, ws, st);
        if (rc) return rc;
    }
    return flow_forward_batch_end(z_out, sum_log_det, M, Mp, N, D, ws, st);
}

// ---------------------------------------------------------------------------
// The same stack WITH autograd (sampling-based objectives: loss on z = nf(N) and its log-density with fresh
// batch statistics).  Forward: the chain above run out of place -- states[c] = output of coupling layer c before
// the fold behind it, folds[c] = that fold's constants.  Backward: per layer one coupling_bwd_mfma launch (its
// fold gradients dA = sum g x_saved, dB = sum g per context are exactly the sums the batch-statistics backward
// needs), then the fold's own backward:
//   x = (v - mu) r e + sh,  mu = mean(v), r = 1/sqrt(var_b(v) + eps) over all R = M N rows, e = exp(a), per feature
//   g_a = e r (P - mu Q) + S_m,  g_sh = Q                       (P = sum_n g_x v, Q = sum_n g_x, S_m = sum_n g_sld)
//   g_r = sum_m e (P - mu Q) + (sum_m S_m)/r,  g_mu = -r sum_m e Q      (sum_log_det carries -log alpha = log r)
//   g_v = g_x r e + k0 + k1 v,  k1 = -g_r r^3 / R,  k0 = g_mu / R - k1 mu
// ---------------------------------------------------------------------------
// per-context sums over the samples: PQ[mp] += [sum g v (D) | sum g (D)];  S[mp] += sum g_sld (when sld given).
// D % 4 == 0: 16-byte loads, D/4 lanes per row, four rows in flight per lane.
__global__ void __launch_bounds__(256)
fold_sums_kernel(const float* __restrict__ g, const float* __restrict__ v, const float* __restrict__ g_sld,
                 float* __restrict__ PQ, float* __restrict__ Ssum, int64_t Mp, int64_t N, int D, int64_t rows_per_block) {
    extern __shared__ float fsred[];  // [rpi][2][D]
    const int64_t m = grid_m();
    const int64_t mp = Mp == 1 ? 0 : m;
    const int tid = threadIdx.x;
    const int lanes = D >> 2, rpi = 256 / lanes;
    const int r = tid / lanes, q = tid - r * lanes;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    const float* gb = g + m * N * D;
    const float* vb = v + m * N * D;
    f4 p = {0.f, 0.f, 0.f, 0.f}, s = {0.f, 0.f, 0.f, 0.f};
    if (r < rpi) {
        for (int64_t row = r0 + r; row < r1; row += 4 * (int64_t)rpi) {
            f4 gv[4], vv[4];  // eight 16-byte loads in flight per lane
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t rw = row + u * (int64_t)rpi;
                const bool ok = rw < r1;
                gv[u] = ok ? *reinterpret_cast<const f4*>(gb + rw * D + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
                vv[u] = ok ? *reinterpret_cast<const f4*>(vb + rw * D + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    p[j] = __builtin_fmaf(gv[u][j], vv[u][j], p[j]);
                    s[j] += gv[u][j];
                }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fsred[(r * 2 + 0) * D + 4 * q + j] = p[j];
            fsred[(r * 2 + 1) * D + 4 * q + j] = s[j];
        }
    }
    __syncthreads();
    for (int d = tid; d < 2 * D; d += 256) {
        float a = 0.f;
        for (int rr = 0; rr < rpi; ++rr) a += fsred[rr * 2 * D + d];
        atomicAdd(PQ + mp * 2 * D + d, a);
    }
    if (g_sld) {
        __syncthreads();
        float sacc = 0.f;
        for (int64_t row = r0 + tid; row < r1; row += 256) sacc += g_sld[m * N + row];
        for (int off = 32; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off);
        if ((tid & 63) == 0) atomicAdd(Ssum + mp, sacc);
    }
}

// the backward of the fold behind layer c, in two steps so that thousands of contexts do not queue up in one block:
// (1) per (context, feature): the Affine gradients, and the contributions to g_r, g_mu and the sum of S_m into
//     racc [g_r part (D) | g_mu part (D) | sum S (1)] (zeroed by the caller);  (2) one block: kk = [k0 (D) | k1 (D)]
__global__ void __launch_bounds__(256)
fold_backward_ctx_kernel(const float* __restrict__ params, int64_t pstride, int64_t affine_off, const float* __restrict__ mean,
                         const float* __restrict__ alpha, float* __restrict__ PQ, const float* __restrict__ Ssum,
                         float* __restrict__ g_params, int64_t gpstride, float* __restrict__ racc, int64_t Mp, int D,
                         int has_affine) {
    const int per = 256 / D > 0 ? 256 / D : 1;  // contexts per block (D <= 256)
    const int ml = threadIdx.x / D, d = threadIdx.x - ml * D;
    const int64_t m = (int64_t)blockIdx.x * per + ml;
    if (ml >= per || m >= Mp) return;
    const float mu = mean[d], r = 1.f / alpha[d];
    const float P = PQ[m * 2 * D + d], Q = PQ[m * 2 * D + D + d];
    PQ[m * 2 * D + d] = 0.f;  // consumed: ready for the next layer's sums (saves a memset per layer)
    PQ[m * 2 * D + D + d] = 0.f;
    float e = 1.f;
    if (has_affine) {
        e = expf(params[m * pstride + affine_off + d]);
        atomicAdd(g_params + m * gpstride + affine_off + d, e * r * (P - mu * Q) + Ssum[m]);
        atomicAdd(g_params + m * gpstride + affine_off + D + d, Q);
    }
    atomicAdd(racc + d, e * (P - mu * Q));
    atomicAdd(racc + D + d, -e * Q);
    if (d == 0) atomicAdd(racc + 2 * D, Ssum[m]);
}

__global__ void __launch_bounds__(256)
fold_backward_fin_kernel(const float* __restrict__ mean, const float* __restrict__ alpha, float* __restrict__ racc,
                         float* __restrict__ kk, double rows, int D) {
    const double s_all = (double)racc[2 * D];
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) {
        const double mu = mean[d], r = 1.0 / (double)alpha[d];
        const double g_r = (double)racc[d] + s_all / r;  // sum_log_det carries -log alpha = log r
        const double g_mu = (double)racc[D + d] * r;
        racc[d] = 0.f;  // consumed: zero for the next layer
        racc[D + d] = 0.f;
        const double k1 = -g_r * r * r * r / rows;
        kk[D + d] = (float)k1;
        kk[d] = (float)(g_mu / rows - k1 * mu);
    }
    if (threadIdx.x == 0) racc[2 * D] = 0.f;
}

// g_out = g_in A[m] + k0 + k1 v   (fold == NULL: A = 1);  g_out may alias g_in
__global__ void __launch_bounds__(256)
fold_bwd_apply_kernel(const float* g_in, const float* __restrict__ v, const float* __restrict__ fold,
                      const float* __restrict__ kk, float* g_out, int64_t Mp, int64_t N, int D) {
    const int64_t m = grid_m();
    const int64_t mp = Mp == 1 ? 0 : m;
    const float* A = fold ? fold + mp * 2 * D : nullptr;
    const int64_t total = N * D, base = m * N * D;
    const int64_t step = (int64_t)gridDim.x * 1024;
    for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < total; i0 += 4 * step) {
        f4 gv[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * step;
            if (i < total) {
                gv[u] = *reinterpret_cast<const f4*>(g_in + base + i);
                vv[u] = *reinterpret_cast<const f4*>(v + base + i);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * step;
            if (i < total) {
                const int d = (int)(i % D);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    gv[u][j] = __builtin_fmaf(gv[u][j], A ? A[d + j] : 1.f, __builtin_fmaf(kk[D + d + j], vv[u][j], kk[d + j]));
                *reinterpret_cast<f4*>(g_out + base + i) = gv[u];
            }
        }
    }
}

int64_t flow_forward_train_workspace(int64_t M, int64_t Mp, int64_t N, int D, int S, int L) {
    // forward: as flow_forward_batch.  backward: PQ (Mp, 2, D) | S (Mp) | kk (2 D) | images | 2 x g buffers (M, N, D)
    const int64_t fwd = flow_forward_batch_workspace(Mp, D, S, L);
    const int64_t bwd = (((Mp * 2 * D + Mp + 1 + 4 * D + 1) * 4 + 255) / 256) * 256 + Mp * 2 * S * mfma_image_floats(D, L) * 4 +
                        2 * (((M * N * D * 4) + 255) / 256) * 256 + 256;
    return fwd > bwd ? fwd : bwd;
}

int launch_flow_forward_train_fwd(const float* omega, const float* params, float* z_out, float* sum_log_det, float* states,
                                  float* folds, float* bn_mean_out, float* bn_alpha_out, int64_t M, int64_t Mp, int64_t N,
                                  int D, int S, int L, int U, int64_t pstride, float eps, void* ws, hipStream_t st) {
    if (!mfma_supported(D, L, U)) return fail(TNF_EUNSUPPORTED, "flow_forward_train: D=%d L=%d U=%d", D, L, U);
    if (N <= 0) return TNF_OK;
    const FbWs w = fb_ws(ws, Mp, D);
    float *ldc = w.ldc, *images = w.images;
    double* sums = w.moments;
    const FlowLayout fl = flow_layout(D, S, L, U);
    const int nl = 2 * S;
    const int64_t img_floats = mfma_image_floats(D, L), plane = M * N * D;
    int rc = launch_flow_images(params, images, Mp, D, S, L, U, pstride, st);
    if (rc) return rc;
    for (int c = 0; c < nl; ++c) {
        MfmaLayerArgs a = {};
        a.z = c == 0 ? omega : states + (int64_t)(c - 1) * plane;
        a.z_out = states + (int64_t)c * plane;
        a.params = params + (c >> 1) * fl.stage + ((c & 1) ? fl.p_up : 0);
        a.pstride = pstride;
        a.image = images + (int64_t)c * img_floats;
        a.image_stride = (int64_t)nl * img_floats;
        a.pre = c == 0 ? nullptr : folds + (int64_t)(c - 1) * Mp * 2 * D;
        a.fold_stride = 2 * (int64_t)D;
        a.ld_in = c == 0 ? nullptr : sum_log_det;
        a.ld_out = sum_log_det;
        a.ld_sign = 1.f;
        a.Mz = M; a.Mp = Mp; a.N = N;
        a.D = D; a.L = L; a.U = U; a.upper = (c & 1) ? 0 : 1; a.inverse = 0;
        rc = launch_coupling_mfma(a, st);
        if (rc) return rc;
        rc = launch_bn_moments(states + (int64_t)c * plane, sums, M * N, D, st);
        if (rc) return rc;
        hipLaunchKernelGGL(flow_batch_finalize_fold_kernel, dim3((unsigned)Mp), dim3(256), 0, st, params, pstride,
                           (int64_t)(c >> 1) * fl.stage + fl.p_up + fl.p_low, sums, eps, bn_mean_out + (int64_t)c * D,
                           bn_alpha_out + (int64_t)c * D, folds + (int64_t)c * Mp * 2 * D, ldc, D, c & 1, c == 0);
    }
    int64_t nb = (N * D / 4 + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(flow_fold_apply_kernel, grid_xm(nb, M), dim3(256), 0, st, states + (int64_t)(nl - 1) * plane, z_out,
                       sum_log_det, folds + (int64_t)(nl - 1) * Mp * 2 * D, ldc, Mp, N, D);
    return check_launch("flow_forward_train_fwd");
}

int launch_flow_forward_train_bwd(const float* omega, const float* params, const float* states, const float* folds,
                                  const float* bn_mean, const float* bn_alpha, const float* g_z, const float* g_sld,
                                  float* g_omega, float* g_params, int64_t M, int64_t Mp, int64_t N, int D, int S, int L,
                                  int U, int64_t pstride, int64_t gpstride, void* ws, hipStream_t st) {
    if (!mfma_supported(D, L, U)) return fail(TNF_EUNSUPPORTED, "flow_forward_train: D=%d L=%d U=%d", D, L, U);
    if (N <= 0) return TNF_OK;
    char* wsb = reinterpret_cast<char*>(ws);
    float* PQ = reinterpret_cast<float*>(wsb);
    float* Ssum = PQ + Mp * 2 * D;
    unsigned* gmaxw = reinterpret_cast<unsigned*>(Ssum + Mp);  // max |upstream gradient|, for the split-f16 layer kernels
    float* racc = Ssum + Mp + 1;  // [g_r part (D) | g_mu part (D) | sum S (1)]
    float* kk = racc + 2 * D + 1;
    const int64_t head = (((Mp * 2 * D + Mp + 1 + 4 * D + 1) * 4 + 255) / 256) * 256;
    float* images = reinterpret_cast<float*>(wsb + head);
    const int nl = 2 * S;
    const int64_t img_floats = mfma_image_floats(D, L), plane = M * N * D;
    const int64_t gb = ((plane * 4 + 255) / 256) * 256;
    char* gbase = wsb + head + ((Mp * nl * img_floats * 4 + 255) / 256) * 256;
    float* gbuf[2] = {reinterpret_cast<float*>(gbase), reinterpret_cast<float*>(gbase + gb)};
    const FlowLayout fl = flow_layout(D, S, L, U);
    int rc = launch_flow_images(params, images, Mp, D, S, L, U, pstride, st);
    if (rc) return rc;
    int64_t nb = (N * D / 4 + 255) / 256;
    if (nb > 2048) nb = 2048;
    const int fs_rpi = 256 / (D / 4);
    int64_t sb = (N + 4 * fs_rpi - 1) / (4 * fs_rpi);
    const int64_t sb_cap = (512 + M - 1) / M;  // few workgroups: their atomics land on the same 2 D words per context
    if (sb > sb_cap) sb = sb_cap;
    if (sb < 1) sb = 1;
    const int64_t rpb = (N + sb - 1) / sb;
    const double rows = (double)M * (double)N;
    // ---- the last fold (behind layer nl-1): sums over (g_z, v), then g_v ----
    if (hipMemsetAsync(PQ, 0, (size_t)(Mp * 2 * D + Mp + 1 + 2 * D + 1) * sizeof(float), st) != hipSuccess)  // .. racc
        return fail(TNF_ELAUNCH, "flow_forward_train_bwd: memset failed");
    rc = launch_gmax(g_z, M * N * D, gmaxw, st);
    if (rc) return rc;
    rc = launch_gmax(g_sld, M * N, gmaxw, st);
    if (rc) return rc;
    const float* v_last = states + (int64_t)(nl - 1) * plane;
    hipLaunchKernelGGL(fold_sums_kernel, grid_xm(sb, M), dim3(256), (size_t)fs_rpi * 2 * D * sizeof(float), st, g_z, v_last,
                       g_sld, PQ, Ssum, Mp, N, D, rpb);
    int cur = 0;
    for (int c = nl - 1; c >= 0; --c) {
        // fold behind layer c: PQ holds its sums (from fold_sums for the last one, else from the layer c+1 backward)
        const float* g_in = (c == nl - 1) ? g_z : gbuf[cur];
        const int64_t aff = (int64_t)(c >> 1) * fl.stage + fl.p_up + fl.p_low;
        const int per = 256 / D > 0 ? 256 / D : 1;
        hipLaunchKernelGGL(fold_backward_ctx_kernel, dim3((unsigned)((Mp + per - 1) / per)), dim3(256), 0, st, params, pstride,
                           aff, bn_mean + (int64_t)c * D, bn_alpha + (int64_t)c * D, PQ, Ssum, g_params, gpstride, racc, Mp, D,
                           c & 1);
        hipLaunchKernelGGL(fold_backward_fin_kernel, dim3(1), dim3(256), 0, st, bn_mean + (int64_t)c * D,
                           bn_alpha + (int64_t)c * D, racc, kk, rows, D);
        // g wrt v_c = g_x A + k0 + k1 v_c.  The last fold applies it in a pass of its own (A included); for the inner
        // ones the coupling backward of the layer behind already multiplied by A, and k0 + k1 v_c is added by the
        // coupling backward kernel of layer c itself in its load stage (v_c is that layer's output).
        if (c == nl - 1) {
            hipLaunchKernelGGL(fold_bwd_apply_kernel, grid_xm(nb, M), dim3(256), 0, st, g_in, states + (int64_t)c * plane,
                               folds + (int64_t)c * Mp * 2 * D, kk, gbuf[cur ^ 1], Mp, N, D);
            cur ^= 1;
        }
        // coupling layer c (PQ was zeroed by fold_backward_ctx_kernel when it consumed it)
        BwdArgs a;
        memset(&a, 0, sizeof(a));
        const int64_t poff = (c >> 1) * fl.stage + ((c & 1) ? fl.p_up : 0);
        a.z = c == 0 ? omega : states + (int64_t)(c - 1) * plane;
        a.params = params + poff;
        a.g_zout = gbuf[cur];
        a.g_ld = g_sld;
        a.ld_scale = 1.f;
        a.g_z = (c == 0 && g_omega) ? g_omega : gbuf[cur ^ 1];
        a.g_params = g_params + poff;
        a.M = M; a.Mp = Mp; a.N = N;
        a.pstride = pstride; a.gpstride = gpstride;
        a.U = U; a.upper = (c & 1) ? 0 : 1;
        a.image = images + (int64_t)c * img_floats;
        a.image_stride = (int64_t)nl * img_floats;
        a.fold = c == 0 ? nullptr : folds + (int64_t)(c - 1) * Mp * 2 * D;
        a.g_fold = c == 0 ? nullptr : PQ;
        a.fold_stride = 2 * (int64_t)D;
        a.gcorr = (c == nl - 1) ? nullptr : kk;
        a.gmax = gmaxw;
        // split-f16 layer backward (flow_bwd_f16.hip) unless asked otherwise or it would spill (L = 3 without a spare unit)
        if (g_train_bwd_fp32 || (L == 3 && U > 15)) rc = launch_coupling_backward_mfma_args(a, D, L, 0, st);
        else rc = launch_coupling_backward_f16(a, D, L, 0, st);
        if (rc) return rc;
        cur ^= 1;
    }
    return check_launch("flow_forward_train_bwd");
}

}  // namespace tnf
