// MFMA backward of one RealNVP coupling layer (float32, the shapes of mfma_supported()).
//
// Same lane mapping as the forward tile code (mfma_tile.h): lane (s, q), activations of a
// 16-unit tile live as acc[j] = unit 4q+j of sample s.  Three groups of v_mfma_f32_16x16x4_f32
// per 16-sample tile:
//   1. the forward pass is recomputed (40 MFMAs, folded operands from the forward image);
//   2. the deltas flow backwards through the twin MLPs with the TRANSPOSED weights as A
//      operands; the accumulator -> B-operand chaining of the forward pass works unchanged
//      (d h1 = W2 . d out, d h0 = W1 . d a1, d x = W0 . d a0: 40 MFMAs);
//   3. the weight gradients dW[k][o] = sum_s a[k,s] d[o,s] contract over the SAMPLE index,
//      which must be the K index of both MFMA operands while the registers carry samples on
//      lane&15: each 16x16 activation / delta tile is transposed through a wave-private LDS
//      scratch (4 ds_write_b32 + 4 ds_read_b32 per lane, padded rows), then 40 MFMAs
//      accumulate dW in registers across all tiles a wave processes.
// Bias gradients are per-lane partial sums reduced across the 16 sample lanes at the end.
// Each wave finally adds its dW / db to the parameter-gradient row with float atomics
// (waves in turn, plain LDS read-add-writes; then one global atomic per parameter per workgroup).
//
// tanh'(a) = 1 - h^2 = 4 r (1 - r) with r the folded sigmoid the forward pass produces.
#include "mfma_tile.h"
#include "tnf_common.h"

namespace tnf {

// A operands of the delta back-propagation, lane (r = lane&15, q = lane>>4), one f4 per group:
//   g_b2(net, mo)[j] = W2_net[k = r][o = 16mo + 4q + j]     (d h_last  = W2 . d out)
//   g_bh(l, net)[j]  = Wh_l_net[k_in = r][k_out = 4q + j]    (d h_l     = Wh_l . d a_{l+1})
//   g_b0(net, m)[j]  = W0_net[f = 16m + r][u = 4q + j]        (d x       = W0 . d a_0)
template <int H, int L>
struct BwdImage {
    static constexpr int HT = (H + 15) / 16;
    static constexpr int NWG = 4 * HT + 2 * (L - 1);
    static constexpr int FLOATS = NWG * 256;
    __device__ static constexpr int g_b2(int net, int mo) { return net * HT + mo; }
    __device__ static constexpr int g_bh(int l, int net) { return 2 * HT + 2 * l + net; }
    __device__ static constexpr int g_b0(int net, int m) { return 2 * HT + 2 * (L - 1) + net * HT + m; }
};

template <int H, int L>
__device__ __forceinline__ void build_bwd_image(float* img, const float* __restrict__ p, int U, int lane) {
    typedef BwdImage<H, L> Img;
    constexpr int HT = Img::HT;
    const int r = lane & 15, q = lane >> 4;
    float* dst = img + lane * 4;
    // layer 0: W0[f][u]
    {
        const float* wt = p;
        const float* ws = p + H * U;
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            f4 vt, vs;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = 16 * m + r, u = 4 * q + j;
                const bool ok = f < H && u < U;
                vt[j] = ld_sel(wt, f * U + u, ok);
                vs[j] = ld_sel(ws, f * U + u, ok);
            }
            *reinterpret_cast<f4*>(dst + Img::g_b0(0, m) * 256) = vt;
            *reinterpret_cast<f4*>(dst + Img::g_b0(1, m) * 256) = vs;
        }
        p += 2 * H * U + 2 * U;
    }
#pragma unroll
    for (int l = 0; l < L - 1; ++l) {
        const float* wt = p;
        const float* ws = p + U * U;
        f4 vt, vs;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ki = r, ko = 4 * q + j;
            const bool ok = ki < U && ko < U;
            vt[j] = ld_sel(wt, ki * U + ko, ok);
            vs[j] = ld_sel(ws, ki * U + ko, ok);
        }
        *reinterpret_cast<f4*>(dst + Img::g_bh(l, 0) * 256) = vt;
        *reinterpret_cast<f4*>(dst + Img::g_bh(l, 1) * 256) = vs;
        p += 2 * U * U + 2 * U;
    }
    {
        const float* wt = p;
        const float* ws = p + U * H;
#pragma unroll
        for (int mo = 0; mo < HT; ++mo) {
            f4 vt, vs;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = r, o = 16 * mo + 4 * q + j;
                const bool ok = k < U && o < H;
                vt[j] = ld_sel(wt, k * H + o, ok);
                vs[j] = ld_sel(ws, k * H + o, ok);
            }
            *reinterpret_cast<f4*>(dst + Img::g_b2(0, mo) * 256) = vt;
            *reinterpret_cast<f4*>(dst + Img::g_b2(1, mo) * 256) = vs;
        }
    }
}

__device__ __forceinline__ void lds_plus(float* p, float v) { *p += v; }

// acc layout (lane (s,q), reg j = row 4q+j, col s)  ->  operand layout with K = samples
// (lane (c = lane&15, kq = lane>>4), reg i = element [row c][sample 4i + kq]).
__device__ __forceinline__ f4 transpose_tile(f4 v, float* scr, int lane) {
    const int s = lane & 15, q = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) scr[(4 * q + j) * 17 + s] = v[j];
    f4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = scr[s * 17 + 4 * i + q];
    return o;
}

// D[rows][cols] += A[rows][K = 16 samples] . B[K][cols]   (both operands already transposed)
__device__ __forceinline__ f4 outer16(f4 a_t, f4 b_t, f4 acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = mfma4(a_t[i], b_t[i], acc);
    return acc;
}


template <int H, int L, bool INV>
__global__ void __launch_bounds__(256)
coupling_bwd_mfma_kernel(BwdArgs a) {
    if (a.gate && *a.gate == 0) return;  // a conditionally needed launch (tnf_set_launch_gate): nothing to do
    constexpr int D = 2 * H;
    constexpr int HT = (H + 15) / 16;
    constexpr int LH = (L > 1) ? (L - 1) : 1;
    typedef LdsLayerImage<H, L> FImg;
    typedef BwdImage<H, L> BImg;
    constexpr int SCR = 17 * 16;  // one padded 16x16 tile
    __shared__ __attribute__((aligned(16))) float lds[FImg::FLOATS + BImg::FLOATS + 4 * 2 * SCR + 4 * D];
    float* fimg = lds;
    float* bimg = lds + FImg::FLOATS;
    float* cf = lds + FImg::FLOATS + BImg::FLOATS + 4 * 2 * SCR;  // fold constants A | B

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const int64_t m = grid_m();
    if (m >= a.M) return;
    const int64_t mp = a.Mp == 1 ? 0 : m;
    const float* prow = a.params + mp * a.pstride;
    float* scrA = lds + FImg::FLOATS + BImg::FLOATS + wave * 2 * SCR;
    float* scrB = scrA + SCR;

    const bool has_fold = a.fold != nullptr;
    const bool finalize = a.g_lp != nullptr;
    if (a.image) {
        const f4* isrc = reinterpret_cast<const f4*>(a.image + mp * a.image_stride);
        f4* idst = reinterpret_cast<f4*>(fimg);
        for (int i = threadIdx.x; i < FImg::FLOATS / 4; i += 256) idst[i] = isrc[i];
    } else if (wave == 0) {
        LayerW<H, L> w;
        load_layer_w<H, L>(w, prow, a.U, lane);
        store_layer_image<H, L>(fimg, w, lane);
    }
    if (wave == 1) build_bwd_image<H, L>(bimg, prow, a.U, lane);
    for (int i = threadIdx.x; i < 2 * D; i += 256) cf[i] = has_fold ? a.fold[mp * a.fold_stride + i] : (i < D ? 1.f : 0.f);
    const bool has_corr = !INV && a.gcorr != nullptr;
    if (has_corr)
        for (int i = threadIdx.x; i < 2 * D; i += 256) cf[2 * D + i] = a.gcorr[i];
    __syncthreads();
    const LdsOperands<H, L> fop(fimg, lane);
    const float* bl = bimg + lane * 4;
    auto bop = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(bl + g * 256); };

    const int c_off = a.upper ? 0 : H, t_off = a.upper ? H : 0;
    const float* zb = a.z + m * a.N * D;
    const float* gzo = a.g_zout ? a.g_zout + m * a.N * D : nullptr;
    const float* gld = a.g_ld + m * a.N;
    const float* glpb = finalize ? a.g_lp + m * a.N : nullptr;
    // fold constants of this lane's features (conditioner / transformed halves)
    // (the constants are re-read from LDS where needed: keeping them in registers costs a wave per SIMD)
    const float* cfx = cf + c_off + 4 * q;
    const float* cfy = cf + t_off + 4 * q;
    f4 dAx[HT], dBx[HT], dAy[HT], dBy[HT];
#pragma unroll
    for (int mm = 0; mm < HT; ++mm) dAx[mm] = dBx[mm] = dAy[mm] = dBy[mm] = f4{0.f, 0.f, 0.f, 0.f};
    float* gzb = a.g_z + m * a.N * D;

    // gradient accumulators (persist over all tiles of this wave)
    f4 dW0[2][HT], dWh[LH][2], dW2[2][HT];
    f4 db0[2], dbh[LH][2], db2[2][HT];
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        db0[net] = zero;
#pragma unroll
        for (int t = 0; t < HT; ++t) { dW0[net][t] = zero; dW2[net][t] = zero; db2[net][t] = zero; }
#pragma unroll
        for (int l = 0; l < LH; ++l) { dWh[l][net] = zero; dbh[l][net] = zero; }
    }

    float glp_acc = 0.f;
    const int64_t ntiles = (a.N + 15) >> 4;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row = tile * 16 + s;
        const bool row_ok = row < a.N;
        const int64_t rowc = row_ok ? row : a.N - 1;
        f4 x[HT], y[HT], xs[HT], ys[HT], gx[HT], gy[HT];
        const float glp = (finalize && row_ok) ? glpb[rowc] : 0.f;
        if (q == 0) glp_acc += glp;
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            const float* zr = zb + rowc * D + 4 * q + 16 * mm;
            xs[mm] = *reinterpret_cast<const f4*>(zr + c_off);
            ys[mm] = *reinterpret_cast<const f4*>(zr + t_off);
            {
                const f4 ax = *reinterpret_cast<const f4*>(cfx + 16 * mm), bx = *reinterpret_cast<const f4*>(cfx + D + 16 * mm);
                const f4 ay = *reinterpret_cast<const f4*>(cfy + 16 * mm), by = *reinterpret_cast<const f4*>(cfy + D + 16 * mm);
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // saved input -> layer input (folded BN / Affine)
                    x[mm][j] = __builtin_fmaf(xs[mm][j], ax[j], bx[j]);
                    y[mm][j] = __builtin_fmaf(ys[mm][j], ay[j], by[j]);
                }
            }
            if (gzo) {
                const float* gr = gzo + rowc * D + 4 * q + 16 * mm;
                gx[mm] = *reinterpret_cast<const f4*>(gr + c_off);
                gy[mm] = *reinterpret_cast<const f4*>(gr + t_off);
            } else {  // finalize: d(-|out|^2/2)/d out * g_lp; the conditioner half of out is x itself
#pragma unroll
                for (int j = 0; j < 4; ++j) gx[mm][j] = -glp * x[mm][j];
                gy[mm] = zero;  // needs the transformed output: filled in below
            }
            if (has_corr) {  // the conditioner half of the output is x itself
                const f4 k0 = *reinterpret_cast<const f4*>(cfx + 2 * D + 16 * mm), k1 = *reinterpret_cast<const f4*>(cfx + 3 * D + 16 * mm);
#pragma unroll
                for (int j = 0; j < 4; ++j) gx[mm][j] += __builtin_fmaf(k1[j], x[mm][j], k0[j]);
            }
            if (!row_ok) { gx[mm] = zero; gy[mm] = zero; }  // padded rows contribute nothing
        }
        const float gl = row_ok ? a.ld_scale * gld[rowc] : 0.f;

        // ---- 1. recompute the forward pass (folded operands; r = sigmoid form of tanh) ----
        f4 r[L][2];
        {
            f4 at = fop.b0(0), as = fop.b0(1);
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                const f4 wt = fop.w0(0, mm), ws = fop.w0(1, mm);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    at = mfma4(wt[j], x[mm][j], at);
                    as = mfma4(ws[j], x[mm][j], as);
                }
            }
            r[0][0] = sig2_4(at);
            r[0][1] = sig2_4(as);
#pragma unroll
            for (int l = 0; l < L - 1; ++l) {
                const f4 wt = fop.wh(l, 0), ws = fop.wh(l, 1);
                f4 nt = fop.bh(l, 0), ns = fop.bh(l, 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    nt = mfma4(wt[j], r[l][0][j], nt);
                    ns = mfma4(ws[j], r[l][1][j], ns);
                }
                r[l + 1][0] = sig2_4(nt);
                r[l + 1][1] = sig2_4(ns);
            }
        }
        // ---- output layer, d out = (d t, d s) and the gradient of the transformed half ----
        f4 dout[2][HT];
#pragma unroll
        for (int mo = 0; mo < HT; ++mo) {
            const f4 wt = fop.w2(0, mo), ws = fop.w2(1, mo);
            f4 tt = fop.b2(0, mo), sv = fop.b2(1, mo);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tt = mfma4(wt[j], r[L - 1][0][j], tt);
                sv = mfma4(ws[j], r[L - 1][1][j], sv);
            }
            f4 dy;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float g = gy[mo][j];
                if (INV) {  // y' = (y - t) e^-s
                    const float em = __builtin_amdgcn_exp2f(-sv[j]);
                    const float yo = (y[mo][j] - tt[j]) * em;
                    if (finalize) g = -glp * yo;
                    dy[j] = g * em;
                    dout[0][mo][j] = -dy[j];
                    dout[1][mo][j] = __builtin_fmaf(-g, yo, gl);
                } else {    // y' = t + y e^s
                    const float e = __builtin_amdgcn_exp2f(sv[j]);
                    if (has_corr && row_ok)
                        g += __builtin_fmaf(cfy[3 * D + 16 * mo + j], __builtin_fmaf(y[mo][j], e, tt[j]), cfy[2 * D + 16 * mo + j]);
                    dy[j] = g * e;
                    dout[0][mo][j] = g;
                    dout[1][mo][j] = __builtin_fmaf(g * y[mo][j], e, gl);
                }
            }
            // back through the fold: g wrt the saved input, and the fold-constant gradients
            {
                const f4 ay = *reinterpret_cast<const f4*>(cfy + 16 * mo);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dAy[mo][j] = __builtin_fmaf(dy[j], ys[mo][j], dAy[mo][j]);
                    dBy[mo][j] += dy[j];
                    dy[j] *= ay[j];
                }
            }
            if (row_ok) *reinterpret_cast<f4*>(gzb + row * D + 4 * q + 16 * mo + t_off) = dy;
        }

        // ---- 2 + 3. back through the layers ----
        // output layer: dW2, db2, d h_{L-1}
        f4 dh[2];
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            f4 hcur;  // tanh output of the last hidden layer, h = 1 - 2r
#pragma unroll
            for (int j = 0; j < 4; ++j) hcur[j] = __builtin_fmaf(-2.f, r[L - 1][net][j], 1.f);
            const f4 h_t = transpose_tile(hcur, scrB, lane);
            f4 acc = zero;
#pragma unroll
            for (int mo = 0; mo < HT; ++mo) {
                db2[net][mo] += dout[net][mo];
                const f4 d_t = transpose_tile(dout[net][mo], scrA, lane);
                dW2[net][mo] = outer16(d_t, h_t, dW2[net][mo]);  // D[o][k]
                const f4 wb = bop(BImg::g_b2(net, mo));
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = mfma4(wb[j], dout[net][mo][j], acc);
            }
            dh[net] = acc;
        }
        // hidden layers, last to first
#pragma unroll
        for (int l = L - 2; l >= 0; --l) {
#pragma unroll
            for (int net = 0; net < 2; ++net) {
                f4 da, hprev;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float rr = r[l + 1][net][j];
                    da[j] = dh[net][j] * (4.f * rr * (1.f - rr));
                    hprev[j] = __builtin_fmaf(-2.f, r[l][net][j], 1.f);
                }
                dbh[l][net] += da;
                const f4 d_t = transpose_tile(da, scrA, lane);
                const f4 h_t = transpose_tile(hprev, scrB, lane);
                dWh[l][net] = outer16(d_t, h_t, dWh[l][net]);  // D[k_out][k_in]
                const f4 wb = bop(BImg::g_bh(l, net));
                f4 acc = zero;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = mfma4(wb[j], da[j], acc);
                dh[net] = acc;
            }
        }
        // first layer: dW0, db0, d x
        f4 dx[HT];
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) dx[mm] = gx[mm];
        f4 x_t[HT];
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) x_t[mm] = transpose_tile(x[mm], scrB, lane);
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            f4 da;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float rr = r[0][net][j];
                da[j] = dh[net][j] * (4.f * rr * (1.f - rr));
            }
            db0[net] += da;
            const f4 d_t = transpose_tile(da, scrA, lane);
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                dW0[net][mm] = outer16(d_t, x_t[mm], dW0[net][mm]);  // D[u][f]
                const f4 wb = bop(BImg::g_b0(net, mm));
#pragma unroll
                for (int j = 0; j < 4; ++j) dx[mm] = mfma4(wb[j], da[j], dx[mm]);
            }
        }
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            const f4 ax = *reinterpret_cast<const f4*>(cfx + 16 * mm);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dAx[mm][j] = __builtin_fmaf(dx[mm][j], xs[mm][j], dAx[mm][j]);
                dBx[mm][j] += dx[mm][j];
                dx[mm][j] *= ax[j];
            }
        }
        if (row_ok) {
#pragma unroll
            for (int mm = 0; mm < HT; ++mm)
                *reinterpret_cast<f4*>(gzb + row * D + 4 * q + 16 * mm + c_off) = dx[mm];
        }
    }

    // ---- flush: waves -> LDS (ds_add_f32) -> one global atomic per parameter per workgroup ----
    // (thousands of waves adding straight into the same 2.5 K addresses serialise at the memory
    // side; a 256-workgroup persistent grid with an LDS pre-reduction keeps it to ~0.6 M atomics)
    __syncthreads();                 // all waves are done with the operand images
    float* gacc = lds;               // reuse the image area: P floats
    const int U = a.U;
    const int P = 2 * (H * U + U) + (L - 1) * 2 * (U * U + U) + 2 * (U * H + H);
    for (int i = threadIdx.x; i < P + 2 * D; i += 256) gacc[i] = 0.f;
    __syncthreads();
    // The four waves add their register accumulators in turn with plain read-add-writes: within a wave every lane
    // owns its element, and ds_add_f32 costs ~190 cycles per wave-instruction on gfx950 (tools/lds_atomic_bench.hip;
    // 84 of them per wave made this tail tens of microseconds long).
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
            const int c = lane & 15;  // column of the dW accumulators; rows are 4q + j
            auto red16 = [&](float v) -> float {  // sum over the 16 sample lanes of a q-group
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                return v;
            };
            float* gp = gacc;
            {   // layer 0
                float* gwt = gp;
                float* gws = gp + H * U;
                float* gbt = gp + 2 * H * U;
                float* gbs = gbt + U;
        #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int u = 4 * q + j;
        #pragma unroll
                    for (int mm = 0; mm < HT; ++mm) {
                        const int f = 16 * mm + c;
                        if (u < U && f < H) {
                            lds_plus(gwt + f * U + u, dW0[0][mm][j]);
                            lds_plus(gws + f * U + u, dW0[1][mm][j]);
                        }
                    }
                    const float bt = red16(db0[0][j]), bs = red16(db0[1][j]);
                    if (s == 0 && u < U) {
                        lds_plus(gbt + u, bt);
                        lds_plus(gbs + u, bs);
                    }
                }
                gp = gbs + U;
            }
        #pragma unroll
            for (int l = 0; l < L - 1; ++l) {
                float* gwt = gp;
                float* gws = gp + U * U;
                float* gbt = gp + 2 * U * U;
                float* gbs = gbt + U;
        #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ko = 4 * q + j, ki = c;
                    if (ko < U && ki < U) {
                        lds_plus(gwt + ki * U + ko, dWh[l][0][j]);
                        lds_plus(gws + ki * U + ko, dWh[l][1][j]);
                    }
                    const float bt = red16(dbh[l][0][j]), bs = red16(dbh[l][1][j]);
                    if (s == 0 && ko < U) {
                        lds_plus(gbt + ko, bt);
                        lds_plus(gbs + ko, bs);
                    }
                }
                gp = gbs + U;
            }
            {
                float* gwt = gp;
                float* gws = gp + U * H;
                float* gbt = gp + 2 * U * H;
                float* gbs = gbt + H;
        #pragma unroll
                for (int mo = 0; mo < HT; ++mo)
        #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int o = 16 * mo + 4 * q + j, k = c;
                        if (k < U && o < H) {
                            lds_plus(gwt + k * H + o, dW2[0][mo][j]);
                            lds_plus(gws + k * H + o, dW2[1][mo][j]);
                        }
                        const float bt = red16(db2[0][mo][j]), bs = red16(db2[1][mo][j]);
                        if (s == 0 && o < H) {
                            lds_plus(gbt + o, bt);
                            lds_plus(gbs + o, bs);
                        }
                    }
            }
            if (finalize && a.glp_sum) {
                const float tot = red16(glp_acc);
                if (lane == 0) atomicAdd(a.glp_sum + mp, tot);
            }
            if (a.g_fold) {  // fold-constant gradients: reduce over the 16 sample lanes, then LDS
                float* gf = gacc + P;  // [dA (D) | dB (D)], zeroed with the rest below P? no: zero it here first
        #pragma unroll
                for (int mm = 0; mm < HT; ++mm)
        #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float ax = red16(dAx[mm][j]), bx = red16(dBx[mm][j]);
                        const float ay = red16(dAy[mm][j]), by = red16(dBy[mm][j]);
                        if (s == 0) {
                            const int fx = c_off + 16 * mm + 4 * q + j, fy = t_off + 16 * mm + 4 * q + j;
                            lds_plus(gf + fx, ax);
                            lds_plus(gf + D + fx, bx);
                            lds_plus(gf + fy, ay);
                            lds_plus(gf + D + fy, by);
                        }
                    }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    float* gout = a.g_params + mp * a.gpstride;
    for (int i = threadIdx.x; i < P; i += 256) atomicAdd(gout + i, gacc[i]);
    if (a.g_fold) {
        float* gfo = a.g_fold + mp * a.fold_stride;
        for (int i = threadIdx.x; i < 2 * D; i += 256) atomicAdd(gfo + i, gacc[P + i]);
    }
}

template <int H, int L>
static void launch_bwd_hl(const BwdArgs& a, int inverse, dim3 grid, hipStream_t st) {
    if (inverse) hipLaunchKernelGGL((coupling_bwd_mfma_kernel<H, L, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((coupling_bwd_mfma_kernel<H, L, false>), grid, dim3(256), 0, st, a);
}

int launch_coupling_backward_mfma(const float* z, const float* params, const float* g_zout,
                                  const float* g_ld, float* g_z, float* g_params, int64_t M, int64_t Mp,
                                  int64_t N, int D, int L, int U, int upper, int inverse, int64_t pstride,
                                  int64_t gpstride, hipStream_t st) {
    if (!mfma_supported(D, L, U)) return fail(TNF_EUNSUPPORTED, "coupling_backward_mfma: D=%d L=%d U=%d", D, L, U);
    BwdArgs a{z, params, g_zout, g_ld, g_z, g_params, M, Mp, N, pstride, gpstride, U, upper,
              nullptr, 0, nullptr, nullptr, 0, nullptr, 1.f, nullptr, nullptr, nullptr};
    return launch_coupling_backward_mfma_args(a, D, L, inverse, st);
}

int launch_coupling_backward_mfma_args(const BwdArgs& a, int D, int L, int inverse, hipStream_t st) {
    const int64_t M = a.M, N = a.N;
    diag_count(TNF_DIAG_BWD_LAYER_FP32);
    const int64_t ntiles = (N + 15) / 16;
    int64_t bx = (ntiles + 3) / 4;
    int64_t cap = 512 / M;  // persistent grid (2 workgroups per CU): each ends with one atomic per parameter
    if (cap < 1) cap = 1;
    if (bx > cap) bx = cap;
    const dim3 grid = grid_xm(bx, M);
    if (D == 64) {
        if (L == 1) launch_bwd_hl<32, 1>(a, inverse, grid, st);
        else if (L == 2) launch_bwd_hl<32, 2>(a, inverse, grid, st);
        else launch_bwd_hl<32, 3>(a, inverse, grid, st);
    } else {
        if (L == 1) launch_bwd_hl<16, 1>(a, inverse, grid, st);
        else if (L == 2) launch_bwd_hl<16, 2>(a, inverse, grid, st);
        else launch_bwd_hl<16, 3>(a, inverse, grid, st);
    }
    return check_launch("coupling_backward_mfma");
}

}  // namespace tnf
