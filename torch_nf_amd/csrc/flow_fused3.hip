// Whole-flow INVERSE kernel (NormFlow.log_prob / inverse_and_log_det), third formulation: the arithmetic of
// flow_fused2.hip on 32-sample groups with v_mfma_f32_32x32x16_f16 -- both conditioner nets in one MFMA, 18 matrix
// instructions per group and layer at D = 64 instead of 48 (f16_tile3.h says why that is what matters).
// One launch per call; prologue (folds, scale exponents, operand images in LDS) and work queue as in flow_fused2.hip.
#include <type_traits>

#include "f16_tile3.h"
#include "tnf_common.h"

#ifndef TNF3_NW
#define TNF3_NW 8   // waves per workgroup (one workgroup per CU: the operand images take most of the LDS)
#endif
#ifndef TNF3_UNROLL
#define TNF3_UNROLL 1  // num_stages = 4: layer loop fully unrolled (every LDS operand offset an immediate)
#endif

namespace tnf {

template <int H, int L>
__host__ __device__ constexpr int flow3_lds_floats(int nl) {
    // images | fold (nl, 2, D) | fin [A (H) | B (H)] | kappa (nl ints) pad 16 | queue head (4) | red (16) | iv (7 D)
    return nl * Img3<H, L>::FLOATS + nl * 4 * H + 2 * H + ((nl + 3) / 4) * 4 + 4 + 16 + 7 * 2 * H;
}

// all coupling layers of one 32-sample group; lo / hi: the two halves of the registers (see f16_tile3.h)
template <int H, int L, int SS, bool SLOW>
__device__ __forceinline__ void run_layers3(const float* img, int S, int lane, float (&lo)[H / 2], float (&hi)[H / 2],
                                            float& ssum) {
    typedef Img3<H, L> I;
    if constexpr (SS > 0 && !SLOW) {
#pragma unroll
        for (int i = 0; i < SS; ++i) {
            const int st = SS - 1 - i;
            coupling_tile3<H, L, false>(img + (2 * st + 1) * I::FLOATS, lane, hi, lo, ssum);
            coupling_tile3<H, L, false>(img + (2 * st) * I::FLOATS, lane, lo, hi, ssum);
        }
    } else {
        for (int st = S - 1; st >= 0; --st) {
            coupling_tile3<H, L, SLOW>(img + (2 * st + 1) * I::FLOATS, lane, hi, lo, ssum);
            coupling_tile3<H, L, SLOW>(img + (2 * st) * I::FLOATS, lane, lo, hi, ssum);
        }
    }
}

template <int H, int L, int NWAVES, int SS = 0>
__global__ void __launch_bounds__(NWAVES * 64)
flow_fused3_kernel(Flow2Args a) {
    constexpr int D = 2 * H;
    constexpr int NR = H / 2;   // registers per half and lane
    constexpr int NG = H / 8;   // float4 pieces per half and lane
    typedef Img3<H, L> I;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nl = 2 * a.S;
    float* img = lds;
    float* fold = lds + nl * I::FLOATS;          // [nl][A (D) | B (D)]
    float* fin = fold + nl * 2 * D;              // [A (H) | B (H)] pending map of the lower half after the last layer
    int* kap = reinterpret_cast<int*>(fin + 2 * H);  // [nl] kappa per layer index c
    int* qhead = kap + ((nl + 3) / 4) * 4;
    float* red = reinterpret_cast<float*>(qhead + 4);  // [16] partial log-det constants
    float* ivc = red + 16;                             // [7][D]

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int s = lane & 31, h = lane >> 5;
    const int64_t m = grid_m();
    if (m >= (a.Mz > a.Mp ? a.Mz : a.Mp)) return;
    const int64_t mz = a.Mz == 1 ? 0 : m, mp = a.Mp == 1 ? 0 : m;
    const float* prow = a.params + mp * a.pstride;

    // ---- prologue A: the maps in front of every layer (inverse chain): A = alpha_bn / e^a, B = mean_bn - shift A ----
    {
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
            const float A = alpha / ea;
            fold[c * 2 * D + d] = A;
            fold[c * 2 * D + D + d] = mu - shift * A;
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) red[wave] = acc;
        if (threadIdx.x == 0) *qhead = NWAVES;
        if (a.iv)
            for (int i = threadIdx.x; i < 7 * D; i += NWAVES * 64) ivc[i] = a.iv[i];
    }
    __syncthreads();
    // ---- prologue B: the scale exponent of every layer's conditioner input ----
    for (int c = wave; c < nl; c += NWAVES) {
        const int kc = layer_kappa<H>(prow + (c >> 1) * a.stage_stride + (c & 1) * a.low_off, a.U, lane, fold + c * 2 * D, c);
        if (lane == 0) kap[c] = kc;
    }
    __syncthreads();
    // ---- prologue C: operand images; pending map of the lower half after the last layer (c = 0) ----
    for (int c = wave; c < nl; c += NWAVES) {
        const float sc_in = pow2i(kap[c]);
        const bool first = (c == nl - 1), last = (c == 0);
        const float sc_prev = first ? 1.f : pow2i(kap[c + 1]);
        const float sig_next = last ? 1.f : pow2i(-kap[c - 1]);
        build_image3<H, L>(img + c * I::FLOATS, prow + (c >> 1) * a.stage_stride + (c & 1) * a.low_off, a.U, lane,
                           fold + c * 2 * D, first ? nullptr : fold + (c + 1) * 2 * D, c, sc_in, sc_prev, sig_next);
    }
    for (int f = threadIdx.x; f < H; f += NWAVES * 64) {
        fin[f] = fold[f] * pow2i(kap[0]);
        fin[H + f] = fold[D + f];
    }
    const bool has_iv = a.iv != nullptr;
    __syncthreads();

    const float presc = pow2i(-kap[nl - 1]);  // the first layer walked (c = nl-1, odd) conditions on the upper half
    const float* zb = a.z + mz * a.N * D;
    float* zo = a.z_out ? a.z_out + m * a.N * D : nullptr;
    float* sldo = a.sum_log_det ? a.sum_log_det + m * a.N : nullptr;
    float* lpo = a.log_prob ? a.log_prob + m * a.N : nullptr;
    float ldc = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) ldc += red[w];

    const int64_t ngroups = (a.N + 31) / 32;
    const int64_t per_block = (ngroups + gridDim.x - 1) / gridDim.x;
    const int64_t g_lo = (int64_t)blockIdx.x * per_block;
    const int64_t g_hi = (g_lo + per_block < ngroups) ? g_lo + per_block : ngroups;
    int64_t grp = g_lo + wave;
    if (grp >= g_hi) return;

    // lane (s, h) holds features 8g + 4h .. +3 (g = 0 .. NG-1) of both halves of sample s: register 4g + e
    auto load_group = [&](int64_t g, float (&dlo)[NR], float (&dhi)[NR]) {
        int64_t row = g * 32 + s;
        if (row >= a.N) row = a.N - 1;
        const float* zr = zb + row * D + 4 * h;
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) {
            const f4 vl = *reinterpret_cast<const f4*>(zr + 8 * gg);
            const f4 vh = *reinterpret_cast<const f4*>(zr + H + 8 * gg);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dlo[4 * gg + e] = vl[e];
                dhi[4 * gg + e] = vh[e];
            }
        }
    };
    // support layer (first bijector of the inverse pass) and the first layer's input scale
    auto enter = [&](float (&dlo)[NR], float (&dhi)[NR], float& ssup) {
        if (has_iv) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                float o, l;
                interval_fast<true>(dlo[i], ivc, D, r_entry(i, h), o, l);
                dlo[i] = o;
                ssup += l;
                interval_fast<true>(dhi[i], ivc, D, H + r_entry(i, h), o, l);
                dhi[i] = o;
                ssup += l;
            }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) dhi[i] *= presc;
    };
    // registers -> true values of the lower half (the upper half leaves the last layer as a true value)
    auto leave = [&](float (&dlo)[NR]) {
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) {
            const f4 fa = *reinterpret_cast<const f4*>(fin + 8 * gg + 4 * h);
            const f4 fb = *reinterpret_cast<const f4*>(fin + H + 8 * gg + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) dlo[4 * gg + e] = __builtin_fmaf(dlo[4 * gg + e], fa[e], fb[e]);
        }
    };

    float nlo[NR], nhi[NR];
    load_group(grp, nlo, nhi);

    for (;;) {
        int nxt_off = 0;
        if (lane == 0) nxt_off = atomicAdd(qhead, 1);
        const int64_t nxt = g_lo + __builtin_amdgcn_readfirstlane(nxt_off);
        const bool has_next = nxt < g_hi;
        float lo[NR], hi[NR];
        float ssum = 0.f, ssup = 0.f;  // ssup: log-det of the fused support layer (natural log, this lane's features)
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            lo[i] = nlo[i];
            hi[i] = nhi[i];
        }
        enter(lo, hi, ssup);
        if (has_next) load_group(nxt, nlo, nhi);
        run_layers3<H, L, SS, false>(img, a.S, lane, lo, hi, ssum);
        // an input beyond the f16 range of its (scaled) operand turned into NaN and reached the log-det sum:
        // re-run this group with exact first-layer contractions (also taken, harmlessly, by genuine NaN inputs)
        if (__builtin_expect(__any(ssum != ssum), 0)) {
            load_group(grp, lo, hi);
            ssum = 0.f;
            ssup = 0.f;
            enter(lo, hi, ssup);
            run_layers3<H, L, 0, true>(img, a.S, lane, lo, hi, ssum);
            if (a.slow_count && lane == 0) atomicAdd(a.slow_count, 1u);
        }
        leave(lo);
        {
            const int64_t row = grp * 32 + s;
            const bool row_ok = row < a.N;
            float part = ssum;
            part += __shfl_xor(part, 32);
            float sup = has_iv ? ssup : 0.f;
            if (has_iv) sup += __shfl_xor(sup, 32);
            const float ld_tot = __builtin_fmaf(part, kLn2, ldc) + sup;
            if (lpo) {
                float sq = 0.f;
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    sq = __builtin_fmaf(lo[i], lo[i], sq);
                    sq = __builtin_fmaf(hi[i], hi[i], sq);
                }
                sq += __shfl_xor(sq, 32);
                if (h == 0 && row_ok) lpo[row] = -0.5f * sq - (float)D * 0.91893853320467274178f - ld_tot;
            }
            if (sldo && h == 0 && row_ok) sldo[row] = ld_tot;
            if (zo && row_ok) {
                float* zr = zo + row * D + 4 * h;
#pragma unroll
                for (int gg = 0; gg < NG; ++gg) {
                    *reinterpret_cast<f4*>(zr + 8 * gg) = f4{lo[4 * gg], lo[4 * gg + 1], lo[4 * gg + 2], lo[4 * gg + 3]};
                    *reinterpret_cast<f4*>(zr + H + 8 * gg) = f4{hi[4 * gg], hi[4 * gg + 1], hi[4 * gg + 2], hi[4 * gg + 3]};
                }
            }
        }
        if (!has_next) break;
        grp = nxt;
    }
}

template <int H, int L>
static size_t flow3_lds_bytes(int S) {
    return (size_t)flow3_lds_floats<H, L>(2 * S) * sizeof(float);
}

static size_t flow3_lds_bytes_rt(int D, int S, int L) {
    if (D == 64) return L == 1 ? flow3_lds_bytes<32, 1>(S) : (L == 2 ? flow3_lds_bytes<32, 2>(S) : flow3_lds_bytes<32, 3>(S));
    return L == 1 ? flow3_lds_bytes<16, 1>(S) : (L == 2 ? flow3_lds_bytes<16, 2>(S) : flow3_lds_bytes<16, 3>(S));
}

bool flow_fused3_supported(int D, int S, int L, int U) {
    if (!mfma_supported(D, L, U) || S < 1) return false;
    return flow3_lds_bytes_rt(D, S, L) <= 160 * 1024;
}

template <int H, int L, int NW, int SS>
static int launch3_t(const Flow2Args& a, int64_t M, hipStream_t st) {
    const size_t smem = flow3_lds_bytes<H, L>(a.S);
    auto kern = flow_fused3_kernel<H, L, NW, SS>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return fail(TNF_ELAUNCH, "flow_fused3: cannot reserve %zu B of LDS", smem);
    const int64_t ngroups = (a.N + 31) / 32;
    int64_t bx = (ngroups + NW - 1) / NW;
    int64_t cap = (256 + M - 1) / M;  // one workgroup per CU (LDS-limited), persistent over its groups
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(kern, grid_xm(bx, M), dim3(NW * 64), smem, st, a);
    return TNF_OK;
}

template <int H, int L>
static int launch3_v(const Flow2Args& a, int64_t M, hipStream_t st) {
#if TNF3_UNROLL
    if (a.S == 4) return launch3_t<H, L, TNF3_NW, 4>(a, M, st);  // the reference's usual depth: layer loop unrolled
#endif
    return launch3_t<H, L, TNF3_NW, 0>(a, M, st);
}

int launch_flow_fused3(const float* z, float* z0, float* sum_log_det, float* log_prob, int64_t Mz, int64_t Mp, int64_t N,
                       int D, int S, int L, int U, const float* params, int64_t pstride, const float* bn_mean,
                       const float* bn_alpha, const float* interval_consts, unsigned* slow_count, hipStream_t st) {
    if (!flow_fused3_supported(D, S, L, U))
        return fail(TNF_EUNSUPPORTED, "flow_fused3: no kernel for D=%d S=%d L=%d U=%d", D, S, L, U);
    if (N <= 0) return TNF_OK;
    const int64_t M = Mz > Mp ? Mz : Mp;
    const FlowLayout fl = flow_layout(D, S, L, U);
    Flow2Args a{z, z0, sum_log_det, log_prob, Mz, Mp, N, S, U, params, bn_mean, bn_alpha, pstride, fl.stage,
                fl.p_up + fl.p_low, fl.p_up, interval_consts, slow_count};
    int rc;
    if (D == 64) rc = L == 1 ? launch3_v<32, 1>(a, M, st) : (L == 2 ? launch3_v<32, 2>(a, M, st) : launch3_v<32, 3>(a, M, st));
    else rc = L == 1 ? launch3_v<16, 1>(a, M, st) : (L == 2 ? launch3_v<16, 2>(a, M, st) : launch3_v<16, 3>(a, M, st));
    if (rc != TNF_OK) return rc;
    return check_launch("flow_fused3");
}

}  // namespace tnf
