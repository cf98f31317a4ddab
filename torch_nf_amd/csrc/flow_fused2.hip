// Whole-flow INVERSE kernel (NormFlow.log_prob / inverse_and_log_det), second formulation -- see f16_tile2.h:
// pending per-feature maps instead of fold instructions, power-of-two operand normalisation, two-MFMA K = 16
// contractions, and detected (not silent) f16 range overflow with an exact re-run of the affected tiles.
// One launch per call: the workgroup's waves fold BatchNorm / Affine, pick the scales and build the layers'
// operand images in LDS, then pull 32-sample groups from an LDS work queue.
#include <type_traits>

#include "f16_tile2.h"
#include "tnf_common.h"

// launch geometry (tuning macros; the defaults are what ships)
#ifndef TNF2_NT
#define TNF2_NT 2   // 16-sample tiles per wave iteration
#endif
#ifndef TNF2_NW
#define TNF2_NW 8   // waves per workgroup (one workgroup per CU: the operand images take most of the LDS)
#endif
#ifndef TNF2_NW16
#define TNF2_NW16 12  // ... at D = 32: the tile state is half as wide, three waves per SIMD fit their registers (0.1746 -> 0.1686 ms; 16: 0.182)
#endif
#ifndef TNF2_RANGE_NT
#define TNF2_RANGE_NT 2   // the layer-range kernel's tiles per wave iteration / waves per workgroup / workgroups per CU
#endif
#ifndef TNF2_RANGE_NW
#define TNF2_RANGE_NW 8
#endif
#ifndef TNF2_RANGE_WGPC
#define TNF2_RANGE_WGPC 1
#endif
#ifndef TNF2_RANGE_ABL
#define TNF2_RANGE_ABL 0
#endif
#ifndef TNF2_RANGE_STAGE
#define TNF2_RANGE_STAGE 1  // D = 64: rows enter and leave through an LDS staging area, 1 KB per wave instruction
#endif
#ifndef TNF2_RANGE_STAGE16
#define TNF2_RANGE_STAGE16 0  // 1: stage D = 32 rows too (measured: no gain)
#endif
#ifndef TNF2_RANGE_SWZ
#define TNF2_RANGE_SWZ 1  // XOR-swizzle the staging area (conflict-free fragment reads)
#endif
#ifndef TNF2_RANGE_DMA
#define TNF2_RANGE_DMA 0  // 1: D = 64 rows enter the staging area by LDS-DMA, two slots per wave (measured: no gain)
#endif
#ifndef TNF2_RANGE_ALT
#define TNF2_RANGE_ALT 1  // odd launches of a chain sweep the rows back to front (what the launch before touched last is still on die)
#endif
#ifndef TNF2_RANGE_NTMEM
#define TNF2_RANGE_NTMEM 2  // bit 0: non-temporal loads, bit 1: non-temporal stores of the layer-range kernel's rows.
                            // Measured on the 8-launch chain at D = 64: plain 0.820 ms, nt stores 0.742, nt loads 0.909,
                            // both 1.167 -- a launch's output is the next launch's input, but 268 MB of it do not survive
                            // in the 256 MB Infinity Cache anyway; written around it, the reads of the rows still to come stay
#endif
#ifndef TNF2_FUSED_STAGE
#define TNF2_FUSED_STAGE 1  // whole-flow kernel: row outputs through LDS staging tiles + non-temporal stores
#endif
#ifndef TNF2_STAMP
#define TNF2_STAMP 0  // 1: diagnostic build that stamps s_memtime / s_memrealtime around the main loop (never shipped)
#endif
#ifndef TNF2_UNROLL
#define TNF2_UNROLL 1  // num_stages = 4: layer loop fully unrolled (every LDS operand offset an immediate)
#endif

namespace tnf {


template <int H, int L>
__host__ __device__ constexpr int flow2_lds_floats(int nl) {
    // images | fold (nl + 1, 2, D) | fin [A (D) | B (D)] | kappa (nl ints) pad 16 | queue head (4) | red (16) | iv (7 D)
    return nl * Img2<H, L>::FLOATS + (nl + 1) * 4 * H + 4 * H + ((nl + 3) / 4) * 4 + 4 + 16 + 7 * 2 * H;
}

// all coupling layers of one group of NT tiles; lo / hi: the two halves of the registers (see f16_tile2.h)
template <int H, int L, int NT, int SS, bool SLOW, bool FWD = false>
__device__ __forceinline__ void run_layers2(const float* img, int S, int lane, f4 (&lo)[NT][H / 16], f4 (&hi)[NT][H / 16],
                                            float (&ssum)[NT]) {
    typedef Img2<H, L> I;
    if constexpr (FWD) {  // sampling direction: layer 2 st conditions on the lower half, 2 st + 1 on the upper
        if constexpr (SS > 0 && !SLOW) {
#pragma unroll
            for (int st = 0; st < SS; ++st) {
                coupling_tile2<H, L, NT, false, 0, true>(img + (2 * st) * I::FLOATS, lane, lo, hi, ssum);
                coupling_tile2<H, L, NT, false, 0, true>(img + (2 * st + 1) * I::FLOATS, lane, hi, lo, ssum);
            }
        } else {
            for (int st = 0; st < S; ++st) {
                coupling_tile2<H, L, NT, SLOW, 0, true>(img + (2 * st) * I::FLOATS, lane, lo, hi, ssum);
                coupling_tile2<H, L, NT, SLOW, 0, true>(img + (2 * st + 1) * I::FLOATS, lane, hi, lo, ssum);
            }
        }
    } else if constexpr (SS > 0 && !SLOW) {
#pragma unroll
        for (int i = 0; i < SS; ++i) {
            const int st = SS - 1 - i;
            coupling_tile2<H, L, NT, false>(img + (2 * st + 1) * I::FLOATS, lane, hi, lo, ssum);
            coupling_tile2<H, L, NT, false>(img + (2 * st) * I::FLOATS, lane, lo, hi, ssum);
        }
    } else {
        for (int st = S - 1; st >= 0; --st) {
            coupling_tile2<H, L, NT, SLOW>(img + (2 * st + 1) * I::FLOATS, lane, hi, lo, ssum);
            coupling_tile2<H, L, NT, SLOW>(img + (2 * st) * I::FLOATS, lane, lo, hi, ssum);
        }
    }
}

// FWD = false: the inverse pass (z -> z0, log_prob), walking layers 2S-1 .. 0 with the folds in FRONT of the layers.
// FWD = true: the sampling pass (omega -> z, sum of log-dets), walking 0 .. 2S-1 with the folds BEHIND the layers:
// fold slot c holds the map in front of layer c in the walk -- slot 0 the identity, slot c the forward fold of layer c-1.
template <int H, int L, int NT, int NWAVES, int SS = 0, bool FWD = false>
__global__ void __launch_bounds__(NWAVES * 64)
flow_fused2_kernel(Flow2Args a) {
    constexpr int D = 2 * H;
    constexpr int HT = H / 16;
    typedef Img2<H, L> I;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nl = 2 * a.S;
    float* img = lds;
    float* fold = lds + nl * I::FLOATS;          // [nl + 1][A (D) | B (D)]
    float* fin = fold + (nl + 1) * 2 * D;        // [A (D) | B (D)] maps still pending after the last layer walked
    int* kap = reinterpret_cast<int*>(fin + 2 * D);  // [nl] kappa per layer index c
    int* qhead = kap + ((nl + 3) / 4) * 4;
    float* red = reinterpret_cast<float*>(qhead + 4);  // [16] partial log-det constants
    float* ivc = red + 16;                             // [7][D]
    float* stage = ivc + 7 * D;                        // [NWAVES][16 rows][D], only when a.stage_out (the launcher sized it)

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const int64_t m = grid_m();
    if (m >= (a.Mz > a.Mp ? a.Mz : a.Mp)) return;
    const int64_t mz = a.Mz == 1 ? 0 : m, mp = a.Mp == 1 ? 0 : m;
    const float* prow = a.params + mp * a.pstride;

    // ---- prologue A: the maps in front of every layer (flow_fold_kernel's arithmetic, inverse chain) ----
    //   A = alpha_bn / e^a, B = mean_bn - shift A  (a, shift: the Affine in front of odd layers), ldc = sum a - sum log alpha
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
            if constexpr (FWD) {  // A = e^a / alpha_bn, B = shift - mean_bn A, behind layer c: slot c + 1
                const float A = ea / alpha;
                fold[(c + 1) * 2 * D + d] = A;
                fold[(c + 1) * 2 * D + D + d] = shift - mu * A;
            } else {
                const float A = alpha / ea;
                fold[c * 2 * D + d] = A;
                fold[c * 2 * D + D + d] = mu - shift * A;
            }
        }
        if constexpr (FWD)
            for (int i = threadIdx.x; i < 2 * D; i += NWAVES * 64) fold[i] = i < D ? 1.f : 0.f;
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
    // ---- prologue C: operand images; the maps still pending after the last layer walked ----
    for (int c = wave; c < nl; c += NWAVES) {
        const float sc_in = pow2i(kap[c]);
        const float* pl = prow + (c >> 1) * a.stage_stride + (c & 1) * a.low_off;
        if constexpr (FWD) {
            // walk order 0 .. nl-1: layer c-1 was walked before c; the fold in front of layer c is slot c, the one the
            // transformed half still owes from two layers back slot c-1 (slot 0 = identity, so c = 1 is regular)
            const float sc_prev = c > 0 ? pow2i(kap[c - 1]) : 1.f;
            const float sig_next = c < nl - 1 ? pow2i(-kap[c + 1]) : 1.f;
            build_image2<H, L, 0, true>(img + c * I::FLOATS, pl, a.U, lane, fold + c * 2 * D,
                                        c > 0 ? fold + (c - 1) * 2 * D : nullptr, c, sc_in, sc_prev, sig_next);
        } else {
            const bool first = (c == nl - 1), last = (c == 0);
            const float sc_prev = first ? 1.f : pow2i(kap[c + 1]);
            const float sig_next = last ? 1.f : pow2i(-kap[c - 1]);
            build_image2<H, L>(img + c * I::FLOATS, pl, a.U, lane, fold + c * 2 * D,
                               first ? nullptr : fold + (c + 1) * 2 * D, c, sc_in, sc_prev, sig_next);
        }
    }
    if constexpr (FWD) {
        // the last layer (c = nl-1, odd) transformed the lower half (true values) and conditioned on the upper one, whose
        // registers still carry 2^kappa and the fold of slot nl-1; slot nl (behind the last layer) is owed by both
        const float* fl = fold + nl * 2 * D;
        const float* fp = fold + (nl - 1) * 2 * D;
        const float sc = pow2i(kap[nl - 1]);
        for (int f = threadIdx.x; f < D; f += NWAVES * 64) {
            const bool up = f >= H;
            fin[f] = up ? fl[f] * fp[f] * sc : fl[f];
            fin[D + f] = up ? __builtin_fmaf(fl[f], fp[D + f], fl[D + f]) : fl[D + f];
        }
    } else {
        for (int f = threadIdx.x; f < H; f += NWAVES * 64) {
            fin[f] = fold[f] * pow2i(kap[0]);
            fin[D + f] = fold[D + f];
        }
    }
    const bool has_iv = a.iv != nullptr;
    __syncthreads();

#if TNF2_PRIO
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 64 * (NWAVES / 2)) __builtin_amdgcn_s_setprio(1);
#endif
    // the first layer walked conditions on the upper half (inverse: c = nl-1) / the lower half (forward: c = 0)
    const float presc = pow2i(-kap[FWD ? 0 : nl - 1]);
    const float* zb = a.z + mz * a.N * D;
    float* zo = a.z_out ? a.z_out + m * a.N * D : nullptr;
    float* sldo = a.sum_log_det ? a.sum_log_det + m * a.N : nullptr;
    float* lpo = a.log_prob ? a.log_prob + m * a.N : nullptr;
    float ldc = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) ldc += red[w];

    const int64_t ngroups = (a.N + 16 * NT - 1) / (16 * NT);
    const int64_t per_block = (ngroups + gridDim.x - 1) / gridDim.x;
    const int64_t g_lo = (int64_t)blockIdx.x * per_block;
    const int64_t g_hi = (g_lo + per_block < ngroups) ? g_lo + per_block : ngroups;
    int64_t grp = g_lo + wave;
    if (grp >= g_hi) return;

    auto load_group = [&](int64_t g, f4 (&dlo)[NT][HT], f4 (&dhi)[NT][HT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int64_t row = (g * NT + t) * 16 + s;
            if (row >= a.N) row = a.N - 1;
            const float* zr = zb + row * D + 4 * q;
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                dlo[t][mm] = *reinterpret_cast<const f4*>(zr + 16 * mm);
                dhi[t][mm] = *reinterpret_cast<const f4*>(zr + H + 16 * mm);
            }
        }
    };
    // support layer (first bijector of the inverse pass) and the first layer's input scale
    auto support = [&](auto inv, f4 (&dlo)[NT][HT], f4 (&dhi)[NT][HT], float (&ssup)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mm = 0; mm < HT; ++mm)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float o, l;
                    interval_fast<decltype(inv)::value>(dlo[t][mm][j], ivc, D, 16 * mm + 4 * q + j, o, l);
                    dlo[t][mm][j] = o;
                    ssup[t] += l;
                    interval_fast<decltype(inv)::value>(dhi[t][mm][j], ivc, D, H + 16 * mm + 4 * q + j, o, l);
                    dhi[t][mm][j] = o;
                    ssup[t] += l;
                }
    };
    auto enter = [&](f4 (&dlo)[NT][HT], f4 (&dhi)[NT][HT], float (&ssup)[NT]) {
        if (!FWD && has_iv) support(std::true_type{}, dlo, dhi, ssup);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mm = 0; mm < HT; ++mm)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (FWD) dlo[t][mm][j] *= presc;
                    else dhi[t][mm][j] *= presc;
                }
    };
    // registers -> true values: inverse, the lower half (the upper half leaves the last layer as a true value);
    // forward, both halves (the fold behind the last layer), then the support layer
    auto leave = [&](f4 (&dlo)[NT][HT], f4 (&dhi)[NT][HT], float (&ssup)[NT]) {
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            const f4 fa = *reinterpret_cast<const f4*>(fin + 16 * mm + 4 * q);
            const f4 fb = *reinterpret_cast<const f4*>(fin + D + 16 * mm + 4 * q);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) dlo[t][mm][j] = __builtin_fmaf(dlo[t][mm][j], fa[j], fb[j]);
            if constexpr (FWD) {
                const f4 ga = *reinterpret_cast<const f4*>(fin + H + 16 * mm + 4 * q);
                const f4 gb = *reinterpret_cast<const f4*>(fin + D + H + 16 * mm + 4 * q);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dhi[t][mm][j] = __builtin_fmaf(dhi[t][mm][j], ga[j], gb[j]);
            }
        }
        if (FWD && has_iv) support(std::false_type{}, dlo, dhi, ssup);
    };

    f4 nlo[NT][HT], nhi[NT][HT];
    load_group(grp, nlo, nhi);
#if TNF2_STAMP  // diagnostic build only (tools/clock_probe.py): shader clock under this kernel's own load
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

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
        double bsq[NT];  // sampling pass with a.log_q: this lane's share of |omega|^2, float64 like the reference's base density
        if constexpr (FWD) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bsq[t] = 0.0;
                if (a.log_q) {
#pragma unroll
                    for (int mm = 0; mm < HT; ++mm)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const double vl = (double)lo[t][mm][j], vh = (double)hi[t][mm][j];
                            bsq[t] = __builtin_fma(vl, vl, bsq[t]);
                            bsq[t] = __builtin_fma(vh, vh, bsq[t]);
                        }
                }
            }
        }
        enter(lo, hi, ssup);
        if (has_next) load_group(nxt, nlo, nhi);
        run_layers2<H, L, NT, SS, false, FWD>(img, a.S, lane, lo, hi, ssum);
        // an input beyond the f16 range of its (scaled) operand turned into NaN and reached the log-det sum:
        // re-run this group with exact first-layer contractions (also taken, harmlessly, by genuine NaN inputs)
        float chk = ssum[0];
#pragma unroll
        for (int t = 1; t < NT; ++t) chk += ssum[t];
        if (__builtin_expect(__any(chk != chk), 0)) {
            load_group(grp, lo, hi);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                ssum[t] = 0.f;
                ssup[t] = 0.f;
            }
            enter(lo, hi, ssup);
            run_layers2<H, L, NT, 0, true, FWD>(img, a.S, lane, lo, hi, ssum);
            if (a.slow_count && lane == 0) atomicAdd(a.slow_count, 1u);
        }
        leave(lo, hi, ssup);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t row = (grp * NT + t) * 16 + s;
            const bool row_ok = row < a.N;
            const float ld_tot = __builtin_fmaf(reduce_q(ssum[t]), kLn2, ldc) + (has_iv ? reduce_q(ssup[t]) : 0.f);
            if (!FWD && lpo) {
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
            if constexpr (FWD) {
                if (a.log_q) {  // density_estimator.py:369-372, 387: log_q = log N(omega; 0, I) - sum of the forward log-dets
                    double b = bsq[t];
                    b += __shfl_xor(b, 16);
                    b += __shfl_xor(b, 32);
                    if (q == 0 && row_ok)
                        a.log_q[m * a.N + row] = (-0.5 * b - (double)D * 0.91893853320467274178) - (double)ld_tot;
                }
            }
            if (zo && !a.stage_out && row_ok) {
                float* zr = zo + row * D + 4 * q;
#pragma unroll
                for (int mm = 0; mm < HT; ++mm) {
                    *reinterpret_cast<f4*>(zr + 16 * mm) = lo[t][mm];
                    *reinterpret_cast<f4*>(zr + H + 16 * mm) = hi[t][mm];
                }
            }
            if (zo && a.stage_out) {
                // rows leave through a wave-private swizzled staging tile, 1 KB (whole 128-B lines) per instruction and
                // around the caches: straight from the fragment layout (64 B per lane group) the stores cost the kernel
                // 0.036 ms at D = 64 (0.227 -> 0.263), this way ... (the layer-range kernel's recipe, below)
                constexpr int CPR = D / 4, RPB = (64 / D) > 0 ? (64 / D) : 1;  // rows per 256-B bank row
                float* stg = stage + wave * 16 * D;
                auto sw = [&](int r, int ch) -> int { return r * D + ((ch ^ ((r / RPB) & (CPR - 1))) << 2); };
#pragma unroll
                for (int mm = 0; mm < HT; ++mm) {
                    *reinterpret_cast<f4*>(stg + sw(s, 4 * mm + q)) = lo[t][mm];
                    *reinterpret_cast<f4*>(stg + sw(s, H / 4 + 4 * mm + q)) = hi[t][mm];
                }
                const int64_t row0 = (grp * NT + t) * 16;
#pragma unroll
                for (int k = 0; k < 16 * D / 256; ++k) {
                    const int off = k * 256 + lane * 4;
                    const int r = off / D;
                    const f4 v = *reinterpret_cast<const f4*>(stg + sw(r, (off % D) >> 2));
                    if (row0 + r < a.N) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(zo + (row0 + r) * D + (off % D)));
                }
            }
        }
        if (!has_next) break;
        grp = nxt;
    }
#if TNF2_STAMP
    if (threadIdx.x == 0 && a.slow_count) {  // [block][cycles, 100 MHz ticks]; the buffer is nobody's output
        a.slow_count[2 * blockIdx.x] = (unsigned)(__builtin_amdgcn_s_memtime() - st_c0);
        a.slow_count[2 * blockIdx.x + 1] = (unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);
    }
#endif
}

// ---------------------------------------------------------------------------
// The same tile code over a RANGE of coupling layers c_hi .. c_lo (walked downwards) per launch: the k = 2S design of
// north_star (one fused kernel per coupling layer, z round-trips through HBM between them) and anything in between.
// What crosses a launch boundary, per sample: the half the range's last layer transformed (true values), the running
// log-det, and -- only when it is not in the buffer already -- that layer's conditioner half AS IT WAS BEFORE ITS FOLD:
// the fold is owed to it and paid by the next launch, whose first layer transforms exactly that half (foldprev of
// build_image2).  So an in-place middle launch reads 2 halves and writes 1: 392 B per sample instead of 520 at D = 64.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void2;

struct Range2Args {
    Flow2Args f;
    int c_hi, c_lo;       // layer range, 2S-1 >= c_hi >= c_lo >= 0
    const float* ld_in;   // running log-det of the launches before (M, N), or NULL
    int store_cond;       // non-final launches: also store the conditioner half of layer c_lo (before its fold)
    // Prepared prologues.  Every launch of a chain spends ~8 us folding constants and building its layers' operand images
    // before it touches a sample -- 8 % of a 2^20-sample launch, with the memory system idle.  So the chain runs ONE
    // preparation launch first (prep_out != NULL, grid.x = number of launches: workgroup x does the prologue of launch x
    // and dumps the LDS region [images .. support constants] to prep_out), and the streaming launches (prep != NULL) just
    // copy their region back.  per_launch: layers per launch of the chain (the preparation launch derives its ranges
    // from it); slot: floats between the regions of consecutive launches.
    const float* prep;
    float* prep_out;
    int per_launch;
    int64_t prep_slot;
    // D = 64 rows enter the staging area by LDS-DMA, two slots per wave (set by launch_range_t when the LDS has the room):
    // two groups in flight per wave instead of one that also occupies 32 registers on its way
    int dma = 0;
};

// floats of the LDS region a prepared prologue consists of (everything behind the staging area)
template <int H, int L>
__host__ __device__ constexpr int range2_region_floats(int nr) {
    return nr * Img2<H, L>::FLOATS + (nr + 1) * 4 * H + 2 * H + ((nr + 3) / 4) * 4 + 4 + 16 + 7 * 2 * H;
}

template <int H, int L>
__host__ __device__ constexpr int range2_lds_floats(int nr) {
    // staging (waves x NT x 16 rows x D) | images (nr) | fold (nr + 1, 2, D) | fin (2 H) | kappa (nr ints, padded) |
    // queue head (4) | red (16) | iv (7 D)
    return TNF2_RANGE_NW * TNF2_RANGE_NT * 32 * H + nr * Img2<H, L>::FLOATS + (nr + 1) * 4 * H + 2 * H + ((nr + 3) / 4) * 4 +
           4 + 16 + 7 * 2 * H;
}

__device__ __forceinline__ void fold_inverse(const Flow2Args& a, const float* prow, int D, int c, int d, float& A, float& B,
                                             float& ld) {
    const float alpha = a.bn_alpha[c * D + d], mu = a.bn_mean[c * D + d];
    ld = -logf(alpha);
    float ea = 1.f, shift = 0.f;
    if (c & 1) {
        const float* ap = prow + (c >> 1) * a.stage_stride + a.affine_off;
        const float av = ap[d];
        ld += av;
        ea = expf(av);
        shift = ap[D + d];
    }
    A = alpha / ea;
    B = mu - shift * A;
}

// the same map in the sampling direction (BatchNorm, then the Affine behind odd layers): v -> A v + B, the arithmetic
// of flow_fused2_kernel<.., FWD>'s prologue
__device__ __forceinline__ void fold_forward(const Flow2Args& a, const float* prow, int D, int c, int d, float& A, float& B) {
    const float alpha = a.bn_alpha[c * D + d], mu = a.bn_mean[c * D + d];
    float ea = 1.f, shift = 0.f;
    if (c & 1) {
        const float* ap = prow + (c >> 1) * a.stage_stride + a.affine_off;
        ea = expf(ap[d]);
        shift = ap[D + d];
    }
    A = ea / alpha;
    B = shift - mu * A;
}

// HI_FIRST / HI_LAST: layer c_hi / c_lo conditions on the upper half (c odd).  Compile-time, so that the two register
// halves are never selected by a run-time index (that would put them in scratch memory).
//
// FWD = true: ONE coupling layer of the SAMPLING direction per launch (c_hi == c_lo == c, the k = 2S design for
// NormFlow.forward with frozen statistics, density_estimator.py:374-388): the walk goes c = 0 .. 2S-1 and the BatchNorm /
// Affine behind a layer (F_c) is applied by nobody -- a half sits in the buffer as the layer that last transformed it
// emitted it, owing every fold since.  At launch c the conditioner half (emitted by layer c-1) owes F_{c-1}, absorbed by
// the layer-0 weights; the half this launch transforms (emitted by layer c-2) owes F_{c-2} then F_{c-1}, composed into its
// Ay, By -- exactly one step of the whole-flow kernel's forward walk (build_image2<.., FWD>: foldc = F_{c-1}, foldprev =
// F_{c-2}), with the register scales reset at the launch boundary.  Middle launches store only the half they transformed;
// the last one turns both halves into true values (one fma per feature).
template <int H, int L, int NT, int NWAVES, bool HI_FIRST, bool HI_LAST, int PREC, bool FWD = false>
__global__ void __launch_bounds__(NWAVES * 64)
flow_range2_kernel(Range2Args ra) {
    static_assert(!FWD || (HI_FIRST == HI_LAST && PREC == 0), "the sampling direction runs one layer per launch, split-f16");
    constexpr int D = 2 * H;
    constexpr int HT = H / 16;
    typedef Img2<H, L> I;
    const Flow2Args& a = ra.f;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nl = 2 * a.S;
    int c_hi_ = ra.c_hi, c_lo_ = ra.c_lo;
    if (ra.prep_out) {  // preparation launch: workgroup x stands in for launch x of the chain
        if constexpr (FWD) {
            c_hi_ = c_lo_ = (int)blockIdx.x;
        } else {
            c_hi_ = nl - 1 - (int)blockIdx.x * ra.per_launch;
            c_lo_ = c_hi_ - ra.per_launch + 1 > 0 ? c_hi_ - ra.per_launch + 1 : 0;
        }
    }
    const int c_hi = c_hi_, c_lo = c_lo_, nr = c_hi - c_lo + 1;
    const int c_top = c_hi < nl - 1 ? c_hi + 1 : c_hi;  // folds are needed for c_lo .. c_top
    const bool final_ = FWD ? c_hi == nl - 1 : c_lo == 0;
    const bool dma = ra.dma != 0;                      // (uniform) two staging slots per wave, filled by LDS-DMA
    float* stage = lds;                                // [NWAVES][1 or 2 slots][NT * 16 rows][D]: wave-private staging of the rows in flight
    float* img = lds + (dma ? NWAVES * 2 * (NT * 16 * D + 64) : NWAVES * NT * 16 * D);  // [nr] image of layer c at index c - c_lo
    float* fold = img + nr * I::FLOATS;                // [nr + 1][A (D) | B (D)], layer c at index c - c_lo
    float* fin = fold + (nr + 1) * 2 * D;              // pending map of layer c_lo's conditioner half: [A (H) | B (H)]
    int* kap = reinterpret_cast<int*>(fin + 2 * H);    // [nr]
    int* qhead = kap + ((nr + 3) / 4) * 4;
    float* red = reinterpret_cast<float*>(qhead + 4);  // [16]
    float* ivc = red + 16;                             // [7][D]

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const int64_t m = grid_m();
    if (m >= (a.Mz > a.Mp ? a.Mz : a.Mp)) return;
    const int64_t mz = a.Mz == 1 ? 0 : m, mp = a.Mp == 1 ? 0 : m;
    const float* prow = a.params + mp * a.pstride;
    const bool has_iv = !FWD && a.iv != nullptr && c_hi == nl - 1;
    const int launch_ix = FWD ? c_lo : (nl - 1 - c_hi) / (ra.per_launch > 0 ? ra.per_launch : 1);
    const int nlaunch = ra.per_launch > 0 ? (nl + ra.per_launch - 1) / ra.per_launch : 1;
    const int region = range2_region_floats<H, L>(nr);  // img .. ivc, contiguous

    if (ra.prep) {  // prepared prologue: copy the region back
        const f4* src = reinterpret_cast<const f4*>(ra.prep + ((int64_t)mp * nlaunch + launch_ix) * ra.prep_slot);
        f4* dst = reinterpret_cast<f4*>(img);
        for (int i = threadIdx.x; i < region / 4; i += NWAVES * 64) dst[i] = src[i];
        __syncthreads();
    } else if constexpr (FWD) {
        const int c = c_lo;
        const float* pl = prow + (c >> 1) * a.stage_stride + (c & 1) * a.low_off;
        // fold[0] = the map in front of layer c (F_{c-1}; identity for c = 0), fold[1] = the one before (F_{c-2})
        for (int i = threadIdx.x; i < 2 * D; i += NWAVES * 64) {
            const int idx = i / D, d = i - idx * D, src = c - 1 - idx;
            float A = 1.f, B = 0.f;
            if (src >= 0) fold_forward(a, prow, D, src, d, A, B);
            fold[idx * 2 * D + d] = A;
            fold[idx * 2 * D + D + d] = B;
        }
        float acc = 0.f;
        if (final_)
            for (int i = threadIdx.x; i < nl * D; i += NWAVES * 64) {
                float A, B, ld;
                fold_inverse(a, prow, D, i / D, i % D, A, B, ld);
                acc += ld;
            }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) red[wave] = acc;
        if (threadIdx.x == 0) *qhead = NWAVES;
        __syncthreads();
        if (wave == 0) {
            const int kc = layer_kappa<H>(pl, a.U, lane, fold, c);
            if (lane == 0) kap[0] = kc;
        }
        __syncthreads();
        if (wave == 0)
            build_image2<H, L, 0, true>(img, pl, a.U, lane, fold, c > 0 ? fold + 2 * D : nullptr, c, pow2i(kap[0]), 1.f, 1.f);
        __syncthreads();
        // the last launch owes true values: the half it transformed F_{nl-1}, its conditioner half (registers scaled by
        // 2^kappa) F_{nl-2} then F_{nl-1}.  Kept where fold[1] was: [A (D) | B (D)] over both halves.
        if (final_) {
            const float sc = pow2i(kap[0]);
            for (int f = threadIdx.x; f < D; f += NWAVES * 64) {
                const bool cond = (f >= H) == ((c & 1) != 0);
                float Al, Bl;
                fold_forward(a, prow, D, c, f, Al, Bl);
                const float Ap = fold[f], Bp = fold[D + f];
                fold[2 * D + f] = cond ? Al * Ap * sc : Al;
                fold[3 * D + f] = cond ? __builtin_fmaf(Al, Bp, Bl) : Bl;
            }
        }
        __syncthreads();
    } else {
    {   // prologue A: folds of the range (+ the one owed from the launch before); the constant log-det on the last launch
        for (int i = threadIdx.x; i < (c_top - c_lo + 1) * D; i += NWAVES * 64) {
            const int ci = i / D, d = i - ci * D;
            float A, B, ld;
            fold_inverse(a, prow, D, c_lo + ci, d, A, B, ld);
            fold[ci * 2 * D + d] = A;
            fold[ci * 2 * D + D + d] = B;
        }
        float acc = 0.f;
        if (final_)
            for (int i = threadIdx.x; i < nl * D; i += NWAVES * 64) {
                float A, B, ld;
                fold_inverse(a, prow, D, i / D, i % D, A, B, ld);
                acc += ld;
            }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) red[wave] = acc;
        if (threadIdx.x == 0) *qhead = NWAVES;
        if (has_iv)
            for (int i = threadIdx.x; i < 7 * D; i += NWAVES * 64) ivc[i] = a.iv[i];
    }
    __syncthreads();
    for (int c = c_lo + wave; c <= c_hi; c += NWAVES) {
        const int kc = layer_kappa<H>(prow + (c >> 1) * a.stage_stride + (c & 1) * a.low_off, a.U, lane,
                                      fold + (c - c_lo) * 2 * D, c);
        if (lane == 0) kap[c - c_lo] = kc;
    }
    __syncthreads();
    for (int c = c_lo + wave; c <= c_hi; c += NWAVES) {
        const float sc_in = pow2i(kap[c - c_lo]);
        // the half layer c transforms arrives with the fold of the layer walked before it still owed (if there is one):
        // inside the range that layer also scaled its registers, across a launch boundary it did not
        const float* foldprev = c < nl - 1 ? fold + (c + 1 - c_lo) * 2 * D : nullptr;
        const float sc_prev = c < c_hi ? pow2i(kap[c + 1 - c_lo]) : 1.f;
        const float sig_next = c > c_lo ? pow2i(-kap[c - 1 - c_lo]) : 1.f;
        build_image2<H, L, PREC>(img + (c - c_lo) * I::FLOATS, prow + (c >> 1) * a.stage_stride + (c & 1) * a.low_off, a.U,
                                 lane, fold + (c - c_lo) * 2 * D, foldprev, c, sc_in, sc_prev, sig_next);
    }
    {   // conditioner half of the range's last layer: true value = fin_A * register + fin_B; register * 2^kappa = what it
        // was before that layer's fold
        const int coff = (c_lo & 1) ? H : 0;
        for (int f = threadIdx.x; f < H; f += NWAVES * 64) {
            fin[f] = fold[coff + f] * pow2i(kap[0]);
            fin[H + f] = fold[D + coff + f];
        }
    }
    __syncthreads();
    }  // in-kernel prologue
    if (ra.prep_out) {  // preparation launch: hand the region over and leave
        f4* dst = reinterpret_cast<f4*>(ra.prep_out + ((int64_t)mp * nlaunch + launch_ix) * ra.prep_slot);
        const f4* src = reinterpret_cast<const f4*>(img);
        for (int i = threadIdx.x; i < region / 4; i += NWAVES * 64) dst[i] = src[i];
        return;
    }

    const float presc = pow2i(-kap[c_hi - c_lo]);
    const float unsc = pow2i(kap[0]);
    const float* zb = a.z + mz * a.N * D;
    float* zo = a.z_out ? a.z_out + m * a.N * D : nullptr;
    const float* ldi = ra.ld_in ? ra.ld_in + m * a.N : nullptr;
    float* sldo = a.sum_log_det ? a.sum_log_det + m * a.N : nullptr;
    float* lpo = (final_ && a.log_prob) ? a.log_prob + m * a.N : nullptr;
    float ldc = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) ldc += red[w];

    const int64_t ngroups = (a.N + 16 * NT - 1) / (16 * NT);
    const int64_t per_block = (ngroups + gridDim.x - 1) / gridDim.x;
    const int64_t g_lo = (int64_t)blockIdx.x * per_block;
    const int64_t g_hi = (g_lo + per_block < ngroups) ? g_lo + per_block : ngroups;
    if (g_lo + wave >= g_hi) return;
#if TNF2_RANGE_ALT  // odd launches of a chain sweep the rows back to front: a launch starts where the one before ended
    const bool rev_ = (launch_ix & 1) != 0;
    auto gmap = [&](int64_t x) -> int64_t { return rev_ ? ngroups - 1 - x : x; };
#else
    auto gmap = [&](int64_t x) -> int64_t { return x; };
#endif
    int64_t grp = gmap(g_lo + wave);

    // ---- global <-> register traffic goes through a wave-private LDS staging area in WHOLE 1 KB pieces ----
    // The MFMA lane mapping wants lane (s, q) to hold 16 bytes of row s: loaded straight from memory that is 16 rows x
    // 64 B per wave instruction, and the layer kernel then streams at 3.8 TB/s; with 1 KB contiguous per instruction
    // (four 256-B rows) the same kernel streams at 5.1 TB/s (measured with the arithmetic removed).  So a group of
    // NT x 16 rows is loaded as it lies in memory, written to LDS as it lies (16-byte pieces XOR-swizzled by row so
    // that the fragment reads are conflict-free), read back in fragment order; results take the same way out.
    constexpr bool STAGED = (H == 32 || TNF2_RANGE_STAGE16) && TNF2_RANGE_STAGE;  // 128-B rows (D = 32) stream as fast in fragment order
    constexpr int GF = NT * 16 * D;         // floats per group
    constexpr int NI = GF / 256;            // 1 KB wave instructions per group
    constexpr int CPR = D / 4;              // 16-byte pieces per row
    constexpr int RPB = (64 / D) > 0 ? (64 / D) : 1;      // rows per 256-B bank row (64 floats)
    constexpr int SLOT = GF + 64;  // a DMA slot: the rows of a group, then their 16 NT running log-dets (padded to 256 B)
    float* const stg0 = stage + wave * (dma ? 2 * SLOT : GF);
    float* stg = stg0;  // the slot of the group at hand
    auto sw_off = [&](int r, int ch) -> int {  // float offset of piece ch of row r inside the staging area
#if TNF2_RANGE_SWZ
        return r * D + ((ch ^ ((r / RPB) & (CPR - 1))) << 2);
#else
        return r * D + (ch << 2);
#endif
    };
    auto load_raw = [&](int64_t g, f4 (&raw)[NI], float (&ldp)[NT]) {
        if constexpr (STAGED) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int off = k * 256 + lane * 4;
                int64_t row = g * (NT * 16) + off / D;
                if (row >= a.N) row = a.N - 1;
#if TNF2_RANGE_NTMEM & 1
                raw[k] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(zb + row * D + (off % D)));
#else
                raw[k] = *reinterpret_cast<const f4*>(zb + row * D + (off % D));
#endif
            }
        } else {  // fragment order straight from memory: raw[(t, half, mm)]
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                int64_t row = (g * NT + t) * 16 + s;
                if (row >= a.N) row = a.N - 1;
#pragma unroll
                for (int mm = 0; mm < HT; ++mm) {
                    raw[(t * 2 + 0) * HT + mm] = *reinterpret_cast<const f4*>(zb + row * D + 16 * mm + 4 * q);
                    raw[(t * 2 + 1) * HT + mm] = *reinterpret_cast<const f4*>(zb + row * D + H + 16 * mm + 4 * q);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int64_t row = (g * NT + t) * 16 + s;
            if (row >= a.N) row = a.N - 1;
            ldp[t] = ldi ? ldi[row] : 0.f;
        }
    };
    // The same group by LDS-DMA: lane l of instruction k fetches the 16-byte piece that belongs at LDS offset k KB + 16 l of
    // the swizzled staging image (the XOR is an involution, so the source piece is the swizzled one: still one 1-KB run
    // of global memory per wave instruction) and the copy lands in `slot` without touching a register.  The log-det
    // inputs first: the wait in front of the fragment reads allows exactly the NI copies of the NEXT group in flight.
    auto dma_load = [&](int64_t g, float* slot) {
        if (ldi) {  // (uniform) one 4-byte copy per lane: lane l -> row l mod (16 NT) of the group
            int64_t row = g * (NT * 16) + (lane & (NT * 16 - 1));
            if (row >= a.N) row = a.N - 1;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(ldi + row), (lds_void2*)(slot + GF), 4, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int off = k * 256 + lane * 4;
            const int r = off / D, pc = (off % D) >> 2;
#if TNF2_RANGE_SWZ
            const int ch = pc ^ ((r / RPB) & (CPR - 1));
#else
            const int ch = pc;
#endif
            int64_t row = g * (NT * 16) + r;
            if (row >= a.N) row = a.N - 1;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const u4*>(zb + row * D + (ch << 2)), (lds_void2*)(slot + k * 256), 16, 0,
                                             (TNF2_RANGE_NTMEM & 1) ? 2 : 0);
        }
    };
    auto unstage = [&](const f4 (&raw)[NI], f4 (&dlo)[NT][HT], f4 (&dhi)[NT][HT]) {
        if constexpr (!STAGED) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mm = 0; mm < HT; ++mm) {
                    dlo[t][mm] = raw[(t * 2 + 0) * HT + mm];
                    dhi[t][mm] = raw[(t * 2 + 1) * HT + mm];
                }
            return;
        }
        if (!dma) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int off = k * 256 + lane * 4;
                *reinterpret_cast<f4*>(stg + sw_off(off / D, (off % D) >> 2)) = raw[k];
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                dlo[t][mm] = *reinterpret_cast<const f4*>(stg + sw_off(16 * t + s, 4 * mm + q));
                dhi[t][mm] = *reinterpret_cast<const f4*>(stg + sw_off(16 * t + s, H / 4 + 4 * mm + q));
            }
    };
    auto enter = [&](f4 (&dlo)[NT][HT], f4 (&dhi)[NT][HT], float (&ssup)[NT]) {
        if (has_iv) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mm = 0; mm < HT; ++mm)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float o, l;
                        interval_fast<true>(dlo[t][mm][j], ivc, D, 16 * mm + 4 * q + j, o, l);
                        dlo[t][mm][j] = o;
                        ssup[t] += l;
                        interval_fast<true>(dhi[t][mm][j], ivc, D, H + 16 * mm + 4 * q + j, o, l);
                        dhi[t][mm][j] = o;
                        ssup[t] += l;
                    }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mm = 0; mm < HT; ++mm)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (HI_FIRST) dhi[t][mm][j] *= presc;
                    else dlo[t][mm][j] *= presc;
                }
    };
    auto layers = [&](auto slow, f4 (&dlo)[NT][HT], f4 (&dhi)[NT][HT], float (&ssum)[NT]) {
        if constexpr (FWD) {
            if constexpr (HI_FIRST) coupling_tile2<H, L, NT, decltype(slow)::value, 0, true>(img, lane, dhi, dlo, ssum);
            else coupling_tile2<H, L, NT, decltype(slow)::value, 0, true>(img, lane, dlo, dhi, ssum);
            return;
        }
        for (int c = c_hi; c >= c_lo; --c) {
            const float* im = img + (c - c_lo) * I::FLOATS;
            if (c & 1) coupling_tile2<H, L, NT, decltype(slow)::value, PREC>(im, lane, dhi, dlo, ssum);
            else coupling_tile2<H, L, NT, decltype(slow)::value, PREC>(im, lane, dlo, dhi, ssum);
        }
    };

    f4 raw[NI];
    float nld[NT];
    // vector-memory instructions a wave issues per group BEHIND the copies of the next group (stores of rows, log-dets,
    // log_prob): the wait in front of a group's fragment reads allows exactly these + the next group's copies in flight
    // (vector memory retires in order, stores included), so a wave never waits for its own stores.  Exact for every group
    // that has a successor: only the very last group of the batch can be partial, and nothing waits behind it.
    const int n_copy = NI + (ldi ? 1 : 0);
    const int n_store = (zo ? (((final_ || HI_LAST || ra.store_cond) && (final_ || !HI_LAST || ra.store_cond)) ? NI : NT * 16 / (64 / (H / 4))) : 0) +
                        (sldo ? NT : 0) + (lpo ? NT : 0);
    bool first_ = true;  // the first group has no stores of a predecessor behind its copies
    auto wait_vm = [&](int n) {  // s_waitcnt vmcnt(n), n uniform; the immediate has 6 bits
        switch (n < 63 ? n : 63) {
#define TNF2_WVM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
            TNF2_WVM(0) TNF2_WVM(1) TNF2_WVM(2) TNF2_WVM(3) TNF2_WVM(4) TNF2_WVM(5) TNF2_WVM(6) TNF2_WVM(7) TNF2_WVM(8) TNF2_WVM(9)
            TNF2_WVM(10) TNF2_WVM(11) TNF2_WVM(12) TNF2_WVM(13) TNF2_WVM(14) TNF2_WVM(15) TNF2_WVM(16) TNF2_WVM(17) TNF2_WVM(18)
            TNF2_WVM(19) TNF2_WVM(20) TNF2_WVM(21) TNF2_WVM(22) TNF2_WVM(23) TNF2_WVM(24) TNF2_WVM(25) TNF2_WVM(26) TNF2_WVM(27)
            TNF2_WVM(28) TNF2_WVM(29) TNF2_WVM(30) TNF2_WVM(31) TNF2_WVM(32)
#undef TNF2_WVM
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    if (STAGED && dma) dma_load(grp, stg);
    else load_raw(grp, raw, nld);

    for (;;) {
        int nxt_off = 0;
        if (lane == 0) nxt_off = atomicAdd(qhead, 1);
        const int64_t nxt_q = g_lo + __builtin_amdgcn_readfirstlane(nxt_off);
        const bool has_next = nxt_q < g_hi;
        const int64_t nxt = gmap(nxt_q);
        f4 lo[NT][HT], hi[NT][HT];
        float ssum[NT], ssup[NT], ldp[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            ssum[t] = 0.f;
            ssup[t] = 0.f;
            ldp[t] = nld[t];
        }
        if (STAGED && dma) {
            // the other slot was drained (its results read back into registers) before the stores of the group before
            // this one were issued: free for the group after this one, which so is in flight together with this one's tail
            if (has_next) dma_load(nxt, stg == stg0 ? stg0 + SLOT : stg0);
            wait_vm(first_ ? (has_next ? n_copy : 0) : (has_next ? n_copy + n_store : n_store));
            first_ = false;
#pragma unroll
            for (int t = 0; t < NT; ++t) ldp[t] = ldi ? stg[GF + 16 * t + s] : 0.f;
        }
        unstage(raw, lo, hi);
        enter(lo, hi, ssup);
        if (!(STAGED && dma) && has_next) load_raw(nxt, raw, nld);
#if TNF2_RANGE_ABL == 0  // (1, 2 = timing experiments: loads and stores only)
        layers(std::false_type{}, lo, hi, ssum);
#endif
        float chk = ssum[0];
#pragma unroll
        for (int t = 1; t < NT; ++t) chk += ssum[t];
        if (PREC == 0 && __builtin_expect(__any(chk != chk), 0)) {  // out-of-range input: exact first-layer contractions
            {
                f4 again[NI];  // (the prefetch of the next group stays in `raw`; with LDS-DMA this group still lies in its slot)
                if (!(STAGED && dma)) load_raw(grp, again, ldp);
                unstage(again, lo, hi);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                ssum[t] = 0.f;
                ssup[t] = 0.f;
            }
            enter(lo, hi, ssup);
            layers(std::true_type{}, lo, hi, ssum);
            if (a.slow_count && lane == 0) atomicAdd(a.slow_count, 1u);
        }
        if constexpr (FWD) {
            // last launch: both halves become true values; before: the conditioner half leaves as it came (it is stored
            // only by the first launch, which works out of place)
            const float* ff = fold + 2 * D;
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                const f4 la = *reinterpret_cast<const f4*>(ff + 16 * mm + 4 * q), lb = *reinterpret_cast<const f4*>(ff + D + 16 * mm + 4 * q);
                const f4 ha = *reinterpret_cast<const f4*>(ff + H + 16 * mm + 4 * q), hb = *reinterpret_cast<const f4*>(ff + D + H + 16 * mm + 4 * q);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (final_) {
                            lo[t][mm][j] = __builtin_fmaf(lo[t][mm][j], la[j], lb[j]);
                            hi[t][mm][j] = __builtin_fmaf(hi[t][mm][j], ha[j], hb[j]);
                        } else if constexpr (HI_LAST) {
                            hi[t][mm][j] *= unsc;
                        } else {
                            lo[t][mm][j] *= unsc;
                        }
                    }
            }
        }
        // conditioner half of the last layer: true values on the final launch, its pre-fold values otherwise
#pragma unroll
        for (int mm = 0; mm < HT && !FWD; ++mm) {
            const f4 fa = *reinterpret_cast<const f4*>(fin + 16 * mm + 4 * q);
            const f4 fb = *reinterpret_cast<const f4*>(fin + H + 16 * mm + 4 * q);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (HI_LAST) hi[t][mm][j] = final_ ? __builtin_fmaf(hi[t][mm][j], fa[j], fb[j]) : hi[t][mm][j] * unsc;
                    else lo[t][mm][j] = final_ ? __builtin_fmaf(lo[t][mm][j], fa[j], fb[j]) : lo[t][mm][j] * unsc;
                }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t row = (grp * NT + t) * 16 + s;
            const bool row_ok = row < a.N;
            float ld_tot = __builtin_fmaf(reduce_q(ssum[t]), kLn2, ldp[t]) + (has_iv ? reduce_q(ssup[t]) : 0.f);
            if (final_) ld_tot += ldc;
            if (lpo) {
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
        }
        if (zo && !STAGED) {
            const bool st_lo = final_ || HI_LAST || ra.store_cond, st_hi = final_ || !HI_LAST || ra.store_cond;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int64_t row = (grp * NT + t) * 16 + s;
                if (row < a.N) {
                    float* zr = zo + row * D + 4 * q;
#pragma unroll
                    for (int mm = 0; mm < HT; ++mm) {
#if TNF2_RANGE_NTMEM & 2
                        if (st_lo) __builtin_nontemporal_store(lo[t][mm], reinterpret_cast<f4*>(zr + 16 * mm));
                        if (st_hi) __builtin_nontemporal_store(hi[t][mm], reinterpret_cast<f4*>(zr + H + 16 * mm));
#else
                        if (st_lo) *reinterpret_cast<f4*>(zr + 16 * mm) = lo[t][mm];
                        if (st_hi) *reinterpret_cast<f4*>(zr + H + 16 * mm) = hi[t][mm];
#endif
                    }
                }
            }
        }
        if (zo && STAGED) {  // results leave through the staging area: whole 128-byte lines per row and instruction
            const bool st_lo = final_ || HI_LAST || ra.store_cond, st_hi = final_ || !HI_LAST || ra.store_cond;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mm = 0; mm < HT; ++mm) {
                    if (st_lo) *reinterpret_cast<f4*>(stg + sw_off(16 * t + s, 4 * mm + q)) = lo[t][mm];
                    if (st_hi) *reinterpret_cast<f4*>(stg + sw_off(16 * t + s, H / 4 + 4 * mm + q)) = hi[t][mm];
                }
            const int64_t row0 = grp * (NT * 16);
            if (st_lo && st_hi) {
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    const int off = k * 256 + lane * 4;
                    const int r = off / D;
                    const f4 v = *reinterpret_cast<const f4*>(stg + sw_off(r, (off % D) >> 2));
#if TNF2_RANGE_NTMEM & 2
                    if (row0 + r < a.N) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(zo + (row0 + r) * D + (off % D)));
#else
                    if (row0 + r < a.N) *reinterpret_cast<f4*>(zo + (row0 + r) * D + (off % D)) = v;
#endif
                }
            } else {
                constexpr int LPR = H / 4, RPI = 64 / LPR;  // lanes per half row, half rows per instruction
                const int hoff = st_hi ? H : 0;
#pragma unroll
                for (int k = 0; k < NT * 16 / RPI; ++k) {
                    const int r = k * RPI + lane / LPR, ch = (hoff >> 2) + lane % LPR;
                    const f4 v = *reinterpret_cast<const f4*>(stg + sw_off(r, ch));
#if TNF2_RANGE_NTMEM & 2
                    if (row0 + r < a.N) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(zo + (row0 + r) * D + (ch << 2)));
#else
                    if (row0 + r < a.N) *reinterpret_cast<f4*>(zo + (row0 + r) * D + (ch << 2)) = v;
#endif
                }
            }
        }
        if (!has_next) break;
        grp = nxt;
        if (STAGED && dma) stg = stg == stg0 ? stg0 + SLOT : stg0;
    }
}

template <int H, int L>
static size_t flow2_lds_bytes(int S) {
    return (size_t)flow2_lds_floats<H, L>(2 * S) * sizeof(float);
}

static size_t flow2_lds_bytes_rt(int D, int S, int L) {
    if (D == 64) return L == 1 ? flow2_lds_bytes<32, 1>(S) : (L == 2 ? flow2_lds_bytes<32, 2>(S) : flow2_lds_bytes<32, 3>(S));
    return L == 1 ? flow2_lds_bytes<16, 1>(S) : (L == 2 ? flow2_lds_bytes<16, 2>(S) : flow2_lds_bytes<16, 3>(S));
}

bool flow_fused2_supported(int D, int S, int L, int U) {
    if (!mfma_supported(D, L, U) || S < 1) return false;
    return flow2_lds_bytes_rt(D, S, L) <= 160 * 1024;
}

template <int H, int L, int NT, int NW, int SS, bool FWD>
static int launch2_t(const Flow2Args& a, int64_t M, hipStream_t st) {
    size_t smem = flow2_lds_bytes<H, L>(a.S);
    Flow2Args b = a;
    // row output (z0 / z): through the staging tiles when they fit beside the operand images (D = 64 only: 128-B rows gain nothing)
    const size_t stage_bytes = (size_t)NW * 16 * 2 * H * sizeof(float);
    b.stage_out = (H == 32 && a.z_out != nullptr && smem + stage_bytes <= 160 * 1024 && TNF2_FUSED_STAGE) ? 1 : 0;
    if (b.stage_out) smem += stage_bytes;
    auto kern = flow_fused2_kernel<H, L, NT, NW, SS, FWD>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return fail(TNF_ELAUNCH, "flow_fused2: cannot reserve %zu B of LDS", smem);
    const int64_t ngroups = (a.N + 16 * NT - 1) / (16 * NT);
    int64_t bx = (ngroups + NW - 1) / NW;
    int64_t cap = (256 + M - 1) / M;  // one workgroup per CU (LDS-limited), persistent over its groups
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(kern, grid_xm(bx, M), dim3(NW * 64), smem, st, b);
    return TNF_OK;
}

template <int H, int L>
static int launch2_v(const Flow2Args& a, int64_t M, int forward, hipStream_t st) {
    // the reference's usual depth (num_stages = 4): layer loop fully unrolled
    constexpr int NW = H == 16 ? TNF2_NW16 : TNF2_NW;
#if TNF2_UNROLL
    if (a.S == 4) return forward ? launch2_t<H, L, TNF2_NT, NW, 4, true>(a, M, st) : launch2_t<H, L, TNF2_NT, NW, 4, false>(a, M, st);
#endif
    return forward ? launch2_t<H, L, TNF2_NT, NW, 0, true>(a, M, st) : launch2_t<H, L, TNF2_NT, NW, 0, false>(a, M, st);
}

int launch_flow_fused2(const float* z, float* z0, float* sum_log_det, float* log_prob, int64_t Mz, int64_t Mp, int64_t N,
                       int D, int S, int L, int U, const float* params, int64_t pstride, const float* bn_mean,
                       const float* bn_alpha, const float* interval_consts, unsigned* slow_count, hipStream_t st,
                       int forward, double* log_q) {
    if (!flow_fused2_supported(D, S, L, U))
        return fail(TNF_EUNSUPPORTED, "flow_fused2: no kernel for D=%d S=%d L=%d U=%d", D, S, L, U);
    if (N <= 0) return TNF_OK;
    const int64_t M = Mz > Mp ? Mz : Mp;
    const FlowLayout fl = flow_layout(D, S, L, U);
    Flow2Args a{z, z0, sum_log_det, log_prob, Mz, Mp, N, S, U, params, bn_mean, bn_alpha, pstride, fl.stage,
                fl.p_up + fl.p_low, fl.p_up, interval_consts, slow_count};
    a.log_q = forward ? log_q : nullptr;
    int rc;
    if (D == 64) rc = L == 1 ? launch2_v<32, 1>(a, M, forward, st) : (L == 2 ? launch2_v<32, 2>(a, M, forward, st) : launch2_v<32, 3>(a, M, forward, st));
    else rc = L == 1 ? launch2_v<16, 1>(a, M, forward, st) : (L == 2 ? launch2_v<16, 2>(a, M, forward, st) : launch2_v<16, 3>(a, M, forward, st));
    if (rc != TNF_OK) return rc;
    return check_launch("flow_fused2");
}


template <int H, int L, int PREC, bool FWD = false>
static int launch_range_t(const Range2Args& ra_in, int64_t M, hipStream_t st) {
    constexpr int NT = TNF2_RANGE_NT, NW = TNF2_RANGE_NW;
    size_t smem = (size_t)range2_lds_floats<H, L>(ra_in.c_hi - ra_in.c_lo + 1) * sizeof(float);
    Range2Args ra = ra_in;
    constexpr bool staged = (H == 32 || TNF2_RANGE_STAGE16) && TNF2_RANGE_STAGE;
    const size_t slot2 = ((size_t)NW * NT * 32 * H + (size_t)NW * 2 * 64) * sizeof(float);  // a second staging slot per wave + the log-det lines
    ra.dma = (TNF2_RANGE_DMA && staged && smem + slot2 <= 160 * 1024) ? 1 : 0;
    if (ra.dma) smem += slot2;
    const bool hf = (ra.c_hi & 1) != 0, hl = (ra.c_lo & 1) != 0;
    void (*kern)(Range2Args);
    if constexpr (FWD) {
        kern = hf ? flow_range2_kernel<H, L, NT, NW, true, true, 0, true> : flow_range2_kernel<H, L, NT, NW, false, false, 0, true>;
    } else {
        kern = hf ? (hl ? flow_range2_kernel<H, L, NT, NW, true, true, PREC> : flow_range2_kernel<H, L, NT, NW, true, false, PREC>)
                  : (hl ? flow_range2_kernel<H, L, NT, NW, false, true, PREC> : flow_range2_kernel<H, L, NT, NW, false, false, PREC>);
    }
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return fail(TNF_ELAUNCH, "flow_range2: cannot reserve %zu B of LDS", smem);
    const int64_t ngroups = (ra.f.N + 16 * NT - 1) / (16 * NT);
    int64_t bx = (ngroups + NW - 1) / NW;
    int64_t cap = (256 * TNF2_RANGE_WGPC + M - 1) / M;  // registers allow one 8-wave workgroup per CU
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(kern, grid_xm(bx, M), dim3(NW * 64), smem, st, ra);
    return TNF_OK;
}

// the preparation launch of a chain (Range2Args::prep_out): one workgroup per (launch of the chain, context)
template <int H, int L, int PREC, bool FWD = false>
static int launch_range_prep_t(const Range2Args& ra, int64_t Mp, int nlaunch, hipStream_t st) {
    constexpr int NT = TNF2_RANGE_NT, NW = TNF2_RANGE_NW;
    const size_t smem = (size_t)range2_lds_floats<H, L>(ra.per_launch) * sizeof(float);
    auto kern = flow_range2_kernel<H, L, NT, NW, false, false, PREC, FWD>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return fail(TNF_ELAUNCH, "flow_range2 (preparation): cannot reserve %zu B of LDS", smem);
    hipLaunchKernelGGL(kern, grid_xm(nlaunch, Mp), dim3(NW * 64), smem, st, ra);
    return TNF_OK;
}

int64_t flow_chain2_prep_floats(int D, int S, int L, int per_launch) {
    // per context: one region per launch, each at the stride of a full range
    if (per_launch < 1) per_launch = 1;
    const int nlaunch = (2 * S + per_launch - 1) / per_launch;
    const int64_t slot = (D == 64) ? (L == 1 ? range2_region_floats<32, 1>(per_launch) : (L == 2 ? range2_region_floats<32, 2>(per_launch) : range2_region_floats<32, 3>(per_launch)))
                                   : (L == 1 ? range2_region_floats<16, 1>(per_launch) : (L == 2 ? range2_region_floats<16, 2>(per_launch) : range2_region_floats<16, 3>(per_launch)));
    return nlaunch * slot;
}

bool flow_range2_supported(int D, int L, int U, int nlayers) {
    if (!mfma_supported(D, L, U) || nlayers < 1) return false;
    const size_t b = (D == 64) ? (L == 1 ? range2_lds_floats<32, 1>(nlayers) : (L == 2 ? range2_lds_floats<32, 2>(nlayers) : range2_lds_floats<32, 3>(nlayers)))
                               : (L == 1 ? range2_lds_floats<16, 1>(nlayers) : (L == 2 ? range2_lds_floats<16, 2>(nlayers) : range2_lds_floats<16, 3>(nlayers)));
    return b * sizeof(float) <= 160 * 1024;
}

// NormFlow.log_prob / inverse_and_log_det as a CHAIN of launches, `per_launch` coupling layers each (1 = the k = 2S design).
// zbuf (M, N, D) and ldbuf (M, N): caller-owned scratch (zbuf may be z0, ldbuf may be sum_log_det).
int launch_flow_chain2(const float* z, float* zbuf, float* ldbuf, float* z0, float* sum_log_det, float* log_prob, int64_t Mz,
                       int64_t Mp, int64_t N, int D, int S, int L, int U, const float* params, int64_t pstride,
                       const float* bn_mean, const float* bn_alpha, const float* interval_consts, unsigned* slow_count,
                       int per_launch, hipStream_t st, int prec, float* prep_ws) {
    const int nl = 2 * S;
    if (per_launch < 1) per_launch = 1;
    if (!flow_range2_supported(D, L, U, per_launch))
        return fail(TNF_EUNSUPPORTED, "flow_chain2: no kernel for D=%d L=%d U=%d with %d layers per launch", D, L, U, per_launch);
    if (N <= 0) return TNF_OK;
    const int64_t M = Mz > Mp ? Mz : Mp;
    const FlowLayout fl = flow_layout(D, S, L, U);
    const int nlaunch = (nl + per_launch - 1) / per_launch;
    const int64_t prep_slot = flow_chain2_prep_floats(D, S, L, per_launch) / nlaunch;
    if (nlaunch < 2) prep_ws = nullptr;  // one launch: its own prologue is the preparation
    if (prep_ws) {  // every launch's prologue in one small launch up front (Range2Args::prep)
        Range2Args ra;
        ra.f = Flow2Args{z, nullptr, nullptr, nullptr, Mp, Mp, N, S, U, params, bn_mean, bn_alpha, pstride, fl.stage,
                         fl.p_up + fl.p_low, fl.p_up, interval_consts, nullptr};
        ra.c_hi = nl - 1;
        ra.c_lo = nl - per_launch > 0 ? nl - per_launch : 0;
        ra.ld_in = nullptr;
        ra.store_cond = 0;
        ra.prep = nullptr;
        ra.prep_out = prep_ws;
        ra.per_launch = per_launch;
        ra.prep_slot = prep_slot;
        int rc;
#define TNF_PREP(HH, LL) rc = prec == 1 ? launch_range_prep_t<HH, LL, 1>(ra, Mp, nlaunch, st) : launch_range_prep_t<HH, LL, 0>(ra, Mp, nlaunch, st)
        if (D == 64) {
            if (L == 1) TNF_PREP(32, 1); else if (L == 2) TNF_PREP(32, 2); else TNF_PREP(32, 3);
        } else {
            if (L == 1) TNF_PREP(16, 1); else if (L == 2) TNF_PREP(16, 2); else TNF_PREP(16, 3);
        }
#undef TNF_PREP
        if (rc != TNF_OK) return rc;
    }
    for (int c_hi = nl - 1; c_hi >= 0; c_hi -= per_launch) {
        const int c_lo = c_hi - per_launch + 1 > 0 ? c_hi - per_launch + 1 : 0;
        const bool first = c_hi == nl - 1, last = c_lo == 0;
        Range2Args ra;
        ra.f = Flow2Args{first ? z : zbuf, last ? z0 : zbuf, last ? sum_log_det : ldbuf, last ? log_prob : nullptr,
                         first ? Mz : M, Mp, N, S, U, params, bn_mean, bn_alpha, pstride, fl.stage, fl.p_up + fl.p_low, fl.p_up,
                         interval_consts, slow_count};
        ra.c_hi = c_hi;
        ra.c_lo = c_lo;
        ra.ld_in = first ? nullptr : ldbuf;
        // the conditioner half of the range's last layer must be written unless it is in zbuf already: a later launch of
        // ONE layer did not touch it (with more layers per launch the layer before transformed it inside the launch)
        ra.store_cond = (first || c_hi > c_lo) ? 1 : 0;
        ra.prep = prep_ws;
        ra.prep_out = nullptr;
        ra.per_launch = per_launch;
        ra.prep_slot = prep_slot;
        int rc;
#define TNF_RANGE(HH, LL) rc = prec == 1 ? launch_range_t<HH, LL, 1>(ra, M, st) : launch_range_t<HH, LL, 0>(ra, M, st)
        if (D == 64) {
            if (L == 1) TNF_RANGE(32, 1); else if (L == 2) TNF_RANGE(32, 2); else TNF_RANGE(32, 3);
        } else {
            if (L == 1) TNF_RANGE(16, 1); else if (L == 2) TNF_RANGE(16, 2); else TNF_RANGE(16, 3);
        }
#undef TNF_RANGE
        if (rc != TNF_OK) return rc;
    }
    return check_launch("flow_chain2");
}

// NormFlow.forward with frozen statistics (the sampling direction) as a chain of 2S launches, one coupling layer each:
// flow_range2_kernel<.., FWD = true>.  omega -> z (out of place on the first launch, in place afterwards; middle launches
// store only the half they transformed), sum_log_det = running log-det buffer and result.
int launch_flow_chain2_fwd(const float* omega, float* z, float* sum_log_det, int64_t Mz, int64_t Mp, int64_t N, int D, int S,
                           int L, int U, const float* params, int64_t pstride, const float* bn_mean, const float* bn_alpha,
                           unsigned* slow_count, hipStream_t st, float* prep_ws) {
    const int nl = 2 * S;
    if (!flow_range2_supported(D, L, U, 1)) return fail(TNF_EUNSUPPORTED, "flow_chain2_fwd: no kernel for D=%d L=%d U=%d", D, L, U);
    if (N <= 0) return TNF_OK;
    const int64_t M = Mz > Mp ? Mz : Mp;
    const FlowLayout fl = flow_layout(D, S, L, U);
    const int64_t prep_slot = flow_chain2_prep_floats(D, S, L, 1) / nl;
    if (nl < 2) prep_ws = nullptr;
    int rc;
#define TNF_FWD_DISPATCH(CALL)                                                              \
    if (D == 64) {                                                                          \
        if (L == 1) rc = CALL(32, 1); else if (L == 2) rc = CALL(32, 2); else rc = CALL(32, 3); \
    } else {                                                                                \
        if (L == 1) rc = CALL(16, 1); else if (L == 2) rc = CALL(16, 2); else rc = CALL(16, 3); \
    }
    if (prep_ws) {
        Range2Args ra;
        ra.f = Flow2Args{omega, nullptr, nullptr, nullptr, Mp, Mp, N, S, U, params, bn_mean, bn_alpha, pstride, fl.stage,
                         fl.p_up + fl.p_low, fl.p_up, nullptr, nullptr};
        ra.c_hi = ra.c_lo = 0;
        ra.ld_in = nullptr;
        ra.store_cond = 0;
        ra.prep = nullptr;
        ra.prep_out = prep_ws;
        ra.per_launch = 1;
        ra.prep_slot = prep_slot;
#define TNF_FWD_PREP(HH, LL) launch_range_prep_t<HH, LL, 0, true>(ra, Mp, nl, st)
        TNF_FWD_DISPATCH(TNF_FWD_PREP)
#undef TNF_FWD_PREP
        if (rc != TNF_OK) return rc;
    }
    for (int c = 0; c < nl; ++c) {
        Range2Args ra;
        ra.f = Flow2Args{c == 0 ? omega : z, z, sum_log_det, nullptr, c == 0 ? Mz : M, Mp, N, S, U, params, bn_mean, bn_alpha,
                         pstride, fl.stage, fl.p_up + fl.p_low, fl.p_up, nullptr, slow_count};
        ra.c_hi = ra.c_lo = c;
        ra.ld_in = c == 0 ? nullptr : sum_log_det;
        ra.store_cond = c == 0 ? 1 : 0;  // the first launch works out of place: the conditioner half moves too
        ra.prep = prep_ws;
        ra.prep_out = nullptr;
        ra.per_launch = 1;
        ra.prep_slot = prep_slot;
#define TNF_FWD_RANGE(HH, LL) launch_range_t<HH, LL, 0, true>(ra, M, st)
        TNF_FWD_DISPATCH(TNF_FWD_RANGE)
#undef TNF_FWD_RANGE
        if (rc != TNF_OK) return rc;
    }
#undef TNF_FWD_DISPATCH
    return check_launch("flow_chain2_fwd");
}

}  // namespace tnf
