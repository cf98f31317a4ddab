// The whole coupling flow in ONE kernel (k = 1): z is read once, every RealNVP layer,
// folded BatchNorm/Affine and the base density run on registers, and only the
// requested outputs are written (4 B/sample for log_prob).  All 2S layers' MFMA
// operands live in LDS in lane order (10.5 KB per layer at D=64, L=2 -> 84 KB for
// S=4 of the 160 KB per CU), fetched with ds_read_b128 per 4 K-steps and shared by
// the NT tiles a wave processes per iteration.  Compulsory HBM traffic: 4D+4 B/sample,
// so this kernel is bound by the fp32 matrix/vector pipes, not by HBM.
#ifndef TNF_STAMP
#define TNF_STAMP 0
#endif
#include "mfma_tile.h"
#include "tnf_common.h"

namespace tnf {

struct FlowFusedArgs {
    const float* z;
    const float* images;  // (Mp, 2S, Img::FLOATS) lane-ordered MFMA operands (flow_images_kernel)
    const float* fold;    // (Mp, 2S, 2, D)
    const float* ldc;   // (Mp)
    float* z_out;       // optional
    float* sum_log_det; // optional
    float* log_prob;    // optional (inverse only)
    int64_t Mz, Mp, N;
    int S, U;
};

template <int H, int NT>
__device__ __forceinline__ void apply_fold(const float* fc, int q, f4 (&lo)[NT][(H + 15) / 16],
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

template <int H, int L, bool INV, int NT, int NWAVES>
__global__ void __launch_bounds__(NWAVES * 64)
flow_fused_kernel(FlowFusedArgs a) {
    constexpr int D = 2 * H;
    constexpr int HT = (H + 15) / 16;
    typedef LdsLayerImage<H, L> Img;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nl = 2 * a.S;
    float* img = lds;                       // [nl][Img::FLOATS]
    float* fold = lds + nl * Img::FLOATS;   // [nl][2][D]

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const int64_t m = grid_m();
    if (m >= (a.Mz > a.Mp ? a.Mz : a.Mp)) return;
    const int64_t mz = a.Mz == 1 ? 0 : m, mp = a.Mp == 1 ? 0 : m;
#if TNF_STAMP
    const unsigned long long st_e0 = __builtin_amdgcn_s_memrealtime();
#endif

    {   // stage all layers' operand images and fold constants in LDS (coalesced float4 copies)
        const f4* isrc = reinterpret_cast<const f4*>(a.images + mp * (int64_t)nl * Img::FLOATS);
        f4* idst = reinterpret_cast<f4*>(img);
        for (int i = threadIdx.x; i < nl * (Img::FLOATS / 4); i += NWAVES * 64) idst[i] = isrc[i];
        const float* fsrc = a.fold + mp * (int64_t)nl * 2 * D;
        for (int i = threadIdx.x; i < nl * 2 * D; i += NWAVES * 64) fold[i] = fsrc[i];
    }
    // Work queue of this workgroup: the two waves that share a SIMD do not progress at the same
    // rate (issue arbitration favours the older one: 330 vs 505 us for equal static shares), so
    // the waves pull 32-sample groups from an LDS counter instead of owning a fixed stride.
    int* qhead = reinterpret_cast<int*>(fold + nl * 2 * D);
    if (threadIdx.x == 0) *qhead = NWAVES;
    __syncthreads();

    const float* zb = a.z + mz * a.N * D;
    float* zo = a.z_out ? a.z_out + m * a.N * D : nullptr;
    float* sldo = a.sum_log_det ? a.sum_log_det + m * a.N : nullptr;
    float* lpo = a.log_prob ? a.log_prob + m * a.N : nullptr;
    const float ldc = a.ldc[mp];

    const int64_t ngroups = (a.N + 16 * NT - 1) / (16 * NT);
    const int64_t per_block = (ngroups + gridDim.x - 1) / gridDim.x;
    const int64_t g_lo = (int64_t)blockIdx.x * per_block;
    const int64_t g_hi = (g_lo + per_block < ngroups) ? g_lo + per_block : ngroups;
    int64_t grp = g_lo + wave;
    if (grp >= g_hi) return;
#if TNF_STAMP  // diagnostic build only: in-kernel clock = d(s_memtime)/d(s_memrealtime) * 100 MHz
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

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
        float ssum[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            ssum[t] = 0.f;
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                lo[t][mm] = nlo[t][mm];
                hi[t][mm] = nhi[t][mm];
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

        if (INV) {
            // density_estimator.py:395-405: walk the stack backwards
            for (int st = a.S - 1; st >= 0; --st) {
                const int c1 = 2 * st + 1, c0 = 2 * st;
                apply_fold<H, NT>(fold + c1 * 2 * D, q, lo, hi);  // Affine^-1, BN^-1
                coupling_tile<H, L, true, NT>(LdsOperands<H, L>(img + c1 * Img::FLOATS, lane), hi, lo, ssum);
                apply_fold<H, NT>(fold + c0 * 2 * D, q, lo, hi);  // BN^-1
                coupling_tile<H, L, true, NT>(LdsOperands<H, L>(img + c0 * Img::FLOATS, lane), lo, hi, ssum);
            }
        } else {
            // density_estimator.py:375-387
            for (int st = 0; st < a.S; ++st) {
                const int c0 = 2 * st, c1 = 2 * st + 1;
                coupling_tile<H, L, false, NT>(LdsOperands<H, L>(img + c0 * Img::FLOATS, lane), lo, hi, ssum);
                apply_fold<H, NT>(fold + c0 * 2 * D, q, lo, hi);  // BN
                coupling_tile<H, L, false, NT>(LdsOperands<H, L>(img + c1 * Img::FLOATS, lane), hi, lo, ssum);
                apply_fold<H, NT>(fold + c1 * 2 * D, q, lo, hi);  // BN, Affine
            }
        }

#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t row = (grp * NT + t) * 16 + s;
            const bool row_ok = row < a.N;
            const float ld_tot = __builtin_fmaf(reduce_q(ssum[t]), kLn2, ldc);  // the tile code sums s*log2(e)
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
                if (q == 0 && row_ok)
                    lpo[row] = -0.5f * sq - (float)D * 0.91893853320467274178f - ld_tot;
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
#if TNF_STAMP
    if ((blockIdx.x % 37) == 0 && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 7)) {
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - st_t0, dr = r1 - st_r0;
        printf("stamp block %3d wave %d: entry %llu loopstart %llu end %llu (x10ns, mod 1e6) loop %.1f us clk %.3f GHz\n",
               (int)blockIdx.x, wave, st_e0 % 1000000ull, st_r0 % 1000000ull, r1 % 1000000ull, dr * 0.01,
               (double)dt / (double)dr * 0.1);
    }
#endif
}

// (tiles per wave iteration, waves per workgroup) variants; index = g_flow_variant
// (TNF_OPT_FLOW_VARIANT, a tuning hook).  One workgroup per CU (LDS-limited), so the
// waves-per-workgroup choice IS the occupancy choice: 8 -> 2 waves/SIMD ... 16 -> 4.
thread_local int g_flow_variant = 10;  // default: the split-f16 whole-flow kernel (flow_fused_f16.hip); 0..5 select the fp32-MFMA kernel

template <int H, int L>
static size_t flow_lds_bytes(int S) {
    return (size_t)2 * S * (LdsLayerImage<H, L>::FLOATS + 2 * 2 * H) * sizeof(float) + 16;  // + queue head
}

static size_t flow_lds_bytes_rt(int D, int S, int L) {
    const int H = D / 2;
    const int HT = (H + 15) / 16;
    const int floats = (4 * HT + 2 * (L - 1)) * 256 + (2 + 2 * (L - 1) + 2 * HT) * 16;
    return (size_t)2 * S * (floats + 2 * D) * sizeof(float) + 16;
}

bool flow_fused_supported(int D, int S, int L, int U) {
    if (!mfma_supported(D, L, U)) return false;
    if (S < 1) return false;
    return flow_lds_bytes_rt(D, S, L) <= 160 * 1024 - 2048;  // room for the fused support layer's constants
}

template <int H, int L, bool INV, int NT, int NW>
static int launch_t(const FlowFusedArgs& a, int64_t M, hipStream_t st) {
    const size_t smem = flow_lds_bytes<H, L>(a.S);
    auto kern = flow_fused_kernel<H, L, INV, NT, NW>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return fail(TNF_ELAUNCH, "flow_fused: cannot reserve %zu B of LDS", smem);
    const int64_t ngroups = (a.N + 16 * NT - 1) / (16 * NT);
    int64_t bx = (ngroups + NW - 1) / NW;
    int64_t cap = (256 + M - 1) / M;  // one workgroup per CU (LDS-limited), persistent over its groups
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(kern, grid_xm(bx, M), dim3(NW * 64), smem, st, a);
    return TNF_OK;
}

template <int H, int L, bool INV>
static int launch_v(const FlowFusedArgs& a, int64_t M, hipStream_t st) {
    if (L == 2) {
        switch (g_flow_variant) {
            case 1: return launch_t<H, L, INV, 1, 8>(a, M, st);
            case 2: return launch_t<H, L, INV, 1, 12>(a, M, st);
            case 3: return launch_t<H, L, INV, 1, 16>(a, M, st);
            case 4: return launch_t<H, L, INV, 2, 12>(a, M, st);
            default: break;
        }
    }
    return launch_t<H, L, INV, 2, 8>(a, M, st);
}

template <int H>
static int launch_h(const FlowFusedArgs& a, int L, int inverse, int64_t M, hipStream_t st) {
    switch (L) {
        case 1: return inverse ? launch_v<H, 1, true>(a, M, st) : launch_v<H, 1, false>(a, M, st);
        case 2: return inverse ? launch_v<H, 2, true>(a, M, st) : launch_v<H, 2, false>(a, M, st);
        default: return inverse ? launch_v<H, 3, true>(a, M, st) : launch_v<H, 3, false>(a, M, st);
    }
}

int launch_flow_fused(const float* z, const float* images, const float* fold, const float* ldc,
                      float* z_out, float* sum_log_det, float* log_prob, int64_t Mz, int64_t Mp,
                      int64_t N, int D, int S, int L, int U, int inverse, hipStream_t st) {
    if (!flow_fused_supported(D, S, L, U))
        return fail(TNF_EUNSUPPORTED, "flow_fused: no kernel for D=%d S=%d L=%d U=%d", D, S, L, U);
    const int64_t M = Mz > Mp ? Mz : Mp;
    if (N <= 0) return TNF_OK;
    FlowFusedArgs a{z, images, fold, ldc, z_out, sum_log_det, log_prob, Mz, Mp, N, S, U};
    int rc = (D == 64) ? launch_h<32>(a, L, inverse, M, st) : launch_h<16>(a, L, inverse, M, st);
    if (rc != TNF_OK) return rc;
    return check_launch("flow_fused");
}

}  // namespace tnf
