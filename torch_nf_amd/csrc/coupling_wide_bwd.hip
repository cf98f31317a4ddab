// MFMA backward of one RealNVP coupling layer for the WIDE shapes (coupling_wide.hip's: D % 8 == 0 up to 128,
// num_units <= 64) -- what torch autograd derives for bijectors.py:145-242 when the hidden width exceeds 16.  Until round 3
// these shapes went through the shape-generic kernel (backward_kernels.hip): 13.7 ms for one layer at D = 64, U = 64,
// N = 2^18 (76 x its own forward); a flow with num_units = 20 trained 8 x slower than one with 15.
//
// The narrow kernels keep their weight-gradient accumulators in registers (coupling_bwd_mfma.hip) or in LDS fixed point
// (flow_bwd_f16.hip).  Neither fits here: U = 64 means 64 gradient tiles of 16 x 16 (256 registers) and 66 KB per copy.
// So the work is cut where the data changes shape, in TWO passes with the sample-contracting products as a GEMM of their own:
//
//   pass 1  coupling_wide_bwd_kernel: per 16-sample tile and wave (the forward kernel's lane mapping and folded operand
//           image, plus a transposed raw image for the way back): forward recompute, output deltas, the deltas back through
//           the twin MLPs with transposed weights as A operands (accumulator -> B-operand chaining, no LDS), g_z.  Every
//           tile the weight gradients need -- the layer inputs x / tanh outputs and the deltas in front of every tanh and at
//           the output -- is written to a RECORD in global memory as a [16 rows][16 samples] block: an accumulator register
//           j of lane (s, q) is row 4q + j, sample s, so a register is four 64-byte segments.
//   pass 2  wide_gw_kernel: dW[k][o] = sum_s in[k, s] d[o, s], one job per (MLP layer, net), the samples cut into G
//           slices.  A lane reads 16 bytes of a block -- row r, samples 4q .. 4q+3 -- which IS the fp32 MFMA operand with the
//           sample on K (any bijection between K slots and samples serves, as long as both operands use the same one): a
//           whole 1-KB block per wave instruction, no transposition anywhere.  A wave owns the gradient tiles of one delta
//           tile, accumulates them in registers inside the MFMAs over its slice, and writes them into its slice's partial
//           row; bias gradients are the row sums of the same delta blocks.
//   pass 3  backward_reduce_kernel (backward_kernels.hip) adds the G partial rows in order: no atomics, bit-reproducible.
//
// Records cost 4 (3 HT + 4 L UT) KB per 16 samples (2.4 KB per sample at D = 64, U = 64, L = 2): written once, read once,
// chunked to 2^18 samples so that the workspace stays below 1 GB whatever N is.
#include "maf_tile.h"
#include "wide_tile.h"

namespace tnf {

constexpr int64_t kWideBwdChunk = 1 << 18;  // samples per (pass 1, pass 2) round
constexpr int kWideBwdSlices = 128;         // sample slices of pass 2 per chunk

// A operands of the way back (raw weights, transposed), lane (r = lane & 15, q = lane >> 4), one f4 per group:
//   g_b2(net, ui, mo)[j] = W2_net[k = 16 ui + r][o = 16 mo + 4q + j]          (d h_last = W2 . d out)
//   g_bh(l, net, ui, uo)[j] = Wh_l_net[k_in = 16 ui + r][k_out = 16 uo + 4q + j]   (d h_l = Wh_l . d a_{l+1})
//   g_b0(net, m, ut)[j] = W0_net[f = 16 m + r][u = 16 ut + 4q + j]             (d x = W0 . d a_0)
struct WideBwdLayout {
    int UT, HT, L;
    __host__ __device__ int nB2() const { return 2 * UT * HT; }
    __host__ __device__ int nBh() const { return 2 * UT * UT; }
    __host__ __device__ int floats() const { return (2 * nB2() + (L - 1) * nBh()) * 256; }
    __host__ __device__ int g_b2(int net, int ui, int mo) const { return (net * UT + ui) * HT + mo; }
    __host__ __device__ int g_bh(int l, int net, int ui, int uo) const { return nB2() + l * nBh() + (net * UT + ui) * UT + uo; }
    __host__ __device__ int g_b0(int net, int m, int ut) const { return nB2() + (L - 1) * nBh() + (net * HT + m) * UT + ut; }
};

__device__ inline void build_wide_bwd_image(float* img, const float* __restrict__ p, WideBwdLayout bl, int H, int U, int lane) {
    const int r = lane & 15, q = lane >> 4;
    float* dst = img + lane * 4;
    {
        const float* w[2] = {p, p + H * U};
        for (int net = 0; net < 2; ++net)
            for (int m = 0; m < bl.HT; ++m)
                for (int ut = 0; ut < bl.UT; ++ut) {
                    f4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = 16 * m + r, u = 16 * ut + 4 * q + j;
                        v[j] = ld_sel(w[net], f * U + u, f < H && u < U);
                    }
                    *reinterpret_cast<f4*>(dst + bl.g_b0(net, m, ut) * 256) = v;
                }
        p += 2 * H * U + 2 * U;
    }
    for (int l = 0; l < bl.L - 1; ++l) {
        const float* w[2] = {p, p + U * U};
        for (int net = 0; net < 2; ++net)
            for (int ui = 0; ui < bl.UT; ++ui)
                for (int uo = 0; uo < bl.UT; ++uo) {
                    f4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ki = 16 * ui + r, ko = 16 * uo + 4 * q + j;
                        v[j] = ld_sel(w[net], ki * U + ko, ki < U && ko < U);
                    }
                    *reinterpret_cast<f4*>(dst + bl.g_bh(l, net, ui, uo) * 256) = v;
                }
        p += 2 * U * U + 2 * U;
    }
    {
        const float* w[2] = {p, p + U * H};
        for (int net = 0; net < 2; ++net)
            for (int ui = 0; ui < bl.UT; ++ui)
                for (int mo = 0; mo < bl.HT; ++mo) {
                    f4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 16 * ui + r, o = 16 * mo + 4 * q + j;
                        v[j] = ld_sel(w[net], k * H + o, k < U && o < H);
                    }
                    *reinterpret_cast<f4*>(dst + bl.g_b2(net, ui, mo) * 256) = v;
                }
    }
}

// blocks of one 16-sample record, in 1-KB units:
//   x (HT) | act(l, net, ut): tanh output of MLP layer l, l = 0..L-1 | da(l, net, ut): delta in front of that tanh |
//   dout(net, mo): deltas of the output layer (t net, s net)
struct WideRec {
    int UT, HT, L;
    __host__ __device__ int blocks() const { return 3 * HT + 4 * L * UT; }
    __host__ __device__ int x(int mm) const { return mm; }
    __host__ __device__ int act(int l, int net, int ut) const { return HT + (l * 2 + net) * UT + ut; }
    __host__ __device__ int da(int l, int net, int ut) const { return HT + 2 * L * UT + (l * 2 + net) * UT + ut; }
    __host__ __device__ int dout(int net, int mo) const { return HT + 4 * L * UT + net * HT + mo; }
};

struct WideBwdArgs {
    const float* z;       // (N, D) saved layer input
    const float* params;  // the one parameter row
    const float* g_zout;  // (N, D)
    const float* g_ld;    // (N)
    float* g_z;           // (N, D)
    float* rec;           // [tiles of the chunk][blocks][16][16]
    float* partials;      // [slices][P] of this chunk
    int64_t N, n0, n1;    // all samples; this chunk = [n0, n1)
    int D, U, upper, inverse;
};

// accumulator register j of lane (s, q) = row 4q + j, sample s  ->  block[row][sample]
__device__ __forceinline__ void store_block(float* blk, f4 v, int s, int q) {
#pragma unroll
    for (int j = 0; j < 4; ++j) blk[(4 * q + j) * 16 + s] = v[j];
}

template <int HT, int UT, int L>
__global__ void __launch_bounds__(256)
coupling_wide_bwd_kernel(WideBwdArgs a, WideLayout wl, WideBwdLayout bl, WideRec rc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = a.D, H = D / 2, U = a.U;
    float* img = lds;                  // forward image (folded)
    float* bimg = lds + wl.floats();   // transposed raw image
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    if (wave == 0) build_wide_image(img, a.params, wl, H, U, lane);
    if (wave == 1) build_wide_bwd_image(bimg, a.params, bl, H, U, lane);
    __syncthreads();
    const float* wsrc = img + lane * 4;
    const float* bsrc = img + wl.NWG() * 256 + q * 4;
    const float* tsrc = bimg + lane * 4;
    auto wgrp = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(wsrc + g * 256); };
    auto bgrp = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(bsrc + g * 16); };
    auto tgrp = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(tsrc + g * 256); };
    const int c_off = a.upper ? 0 : H, t_off = a.upper ? H : 0;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    bool fok[HT];
#pragma unroll
    for (int mm = 0; mm < HT; ++mm) fok[mm] = 16 * mm + 4 * q < H;

    const int64_t tile0 = a.n0 >> 4, ntiles = (a.n1 - a.n0 + 15) >> 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
        const int64_t row = (tile0 + t) * 16 + s;
        const bool row_ok = row < a.n1;
        const int64_t rowc = row_ok ? row : a.n1 - 1;
        float* rec = a.rec + t * (int64_t)rc.blocks() * 256;
        const float* zr = a.z + rowc * D + 4 * q;
        const float* gr = a.g_zout + rowc * D + 4 * q;
        f4 x[HT], y[HT], gx[HT], gy[HT];
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            const bool ok = fok[mm];
            x[mm] = ok ? *reinterpret_cast<const f4*>(zr + c_off + 16 * mm) : zero;
            y[mm] = ok ? *reinterpret_cast<const f4*>(zr + t_off + 16 * mm) : zero;
            gx[mm] = (ok && row_ok) ? *reinterpret_cast<const f4*>(gr + c_off + 16 * mm) : zero;  // padded rows: no gradient
            gy[mm] = (ok && row_ok) ? *reinterpret_cast<const f4*>(gr + t_off + 16 * mm) : zero;
            store_block(rec + rc.x(mm) * 256, x[mm], s, q);
        }
        const float gl = row_ok ? a.g_ld[rowc] : 0.f;
        asm volatile("" ::: "memory");  // operand reads stay inside the tile loop

        // ---- forward recompute: r = (1 - tanh) / 2 of every layer ----
        f4 r[L][2][UT];
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
            r[0][0][ut] = sig2_4(at);
            r[0][1][ut] = sig2_4(as);
        }
#pragma unroll
        for (int l = 0; l < L - 1; ++l)
#pragma unroll
            for (int uo = 0; uo < UT; ++uo) {
                f4 at = bgrp(wl.b_bh(l, 0, uo)), as = bgrp(wl.b_bh(l, 1, uo));
#pragma unroll
                for (int ui = 0; ui < UT; ++ui) {
                    const f4 wt = wgrp(wl.g_wh(l, 0, uo, ui)), ws = wgrp(wl.g_wh(l, 1, uo, ui));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        at = mfma4(wt[j], r[l][0][ui][j], at);
                        as = mfma4(ws[j], r[l][1][ui][j], as);
                    }
                }
                r[l + 1][0][uo] = sig2_4(at);
                r[l + 1][1][uo] = sig2_4(as);
            }
        // the layer inputs the weight gradients contract with: tanh = 1 - 2r
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int net = 0; net < 2; ++net)
#pragma unroll
                for (int ut = 0; ut < UT; ++ut) {
                    f4 h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) h[j] = __builtin_fmaf(-2.f, r[l][net][ut][j], 1.f);
                    store_block(rec + rc.act(l, net, ut) * 256, h, s, q);
                }
        // ---- output layer, output deltas, gradient of the transformed half ----
        f4 dout[2][HT];
#pragma unroll
        for (int mo = 0; mo < HT; ++mo) {
            f4 tt = bgrp(wl.b_b2(0, mo)), sv = bgrp(wl.b_b2(1, mo));
#pragma unroll
            for (int ui = 0; ui < UT; ++ui) {
                const f4 wt = wgrp(wl.g_w2(0, mo, ui)), ws = wgrp(wl.g_w2(1, mo, ui));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tt = mfma4(wt[j], r[L - 1][0][ui][j], tt);
                    sv = mfma4(ws[j], r[L - 1][1][ui][j], sv);
                }
            }
            f4 dy;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float g = gy[mo][j];
                if (a.inverse) {  // y' = (y - t) e^-s
                    const float em = __builtin_amdgcn_exp2f(-sv[j]);
                    const float yo = (y[mo][j] - tt[j]) * em;
                    dy[j] = g * em;
                    dout[0][mo][j] = -dy[j];
                    dout[1][mo][j] = __builtin_fmaf(-g, yo, gl);
                } else {          // y' = t + y e^s
                    const float e = __builtin_amdgcn_exp2f(sv[j]);
                    dy[j] = g * e;
                    dout[0][mo][j] = g;
                    dout[1][mo][j] = __builtin_fmaf(g * y[mo][j], e, gl);
                }
                if (16 * mo + 4 * q + j >= H) {  // a padded feature has no s: its delta is not the log-det gradient
                    dout[0][mo][j] = 0.f;
                    dout[1][mo][j] = 0.f;
                }
            }
            store_block(rec + rc.dout(0, mo) * 256, dout[0][mo], s, q);
            store_block(rec + rc.dout(1, mo) * 256, dout[1][mo], s, q);
            if (row_ok && fok[mo]) *reinterpret_cast<f4*>(a.g_z + row * D + 4 * q + 16 * mo + t_off) = dy;
        }
        // ---- the deltas back through the MLPs ----
        f4 dh[2][UT];
#pragma unroll
        for (int net = 0; net < 2; ++net)
#pragma unroll
            for (int ui = 0; ui < UT; ++ui) {
                f4 acc = zero;
#pragma unroll
                for (int mo = 0; mo < HT; ++mo) {
                    const f4 wb = tgrp(bl.g_b2(net, ui, mo));
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = mfma4(wb[j], dout[net][mo][j], acc);
                }
                dh[net][ui] = acc;
            }
#pragma unroll
        for (int l = L - 2; l >= 0; --l)
#pragma unroll
            for (int net = 0; net < 2; ++net) {
                f4 da[UT];
#pragma unroll
                for (int uo = 0; uo < UT; ++uo) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float rr = r[l + 1][net][uo][j];
                        da[uo][j] = dh[net][uo][j] * (4.f * rr * (1.f - rr));
                    }
                    store_block(rec + rc.da(l + 1, net, uo) * 256, da[uo], s, q);
                }
#pragma unroll
                for (int ui = 0; ui < UT; ++ui) {
                    f4 acc = zero;
#pragma unroll
                    for (int uo = 0; uo < UT; ++uo) {
                        const f4 wb = tgrp(bl.g_bh(l, net, ui, uo));
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc = mfma4(wb[j], da[uo][j], acc);
                    }
                    dh[net][ui] = acc;
                }
            }
        f4 dx[HT];
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) dx[mm] = gx[mm];
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            f4 da[UT];
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float rr = r[0][net][ut][j];
                    da[ut][j] = dh[net][ut][j] * (4.f * rr * (1.f - rr));
                }
                store_block(rec + rc.da(0, net, ut) * 256, da[ut], s, q);
            }
#pragma unroll
            for (int mm = 0; mm < HT; ++mm)
#pragma unroll
                for (int ut = 0; ut < UT; ++ut) {
                    const f4 wb = tgrp(bl.g_b0(net, mm, ut));
#pragma unroll
                    for (int j = 0; j < 4; ++j) dx[mm] = mfma4(wb[j], da[ut][j], dx[mm]);
                }
        }
        if (row_ok) {
#pragma unroll
            for (int mm = 0; mm < HT; ++mm)
                if (fok[mm]) *reinterpret_cast<f4*>(a.g_z + row * D + 4 * q + 16 * mm + c_off) = dx[mm];
        }
    }
}

// ---------------------------------------------------------------------------
// Pass 2: the sample-contracting products.  blockIdx.x = job * G + slice; job = layer * 2 + net, layer 0 .. L.
// ---------------------------------------------------------------------------
struct WideGwArgs {
    const float* rec;
    float* partials;  // [G][P]
    int64_t ntiles;   // sample tiles of the chunk
    int H, U, L, P, G;  // H: width of the first layer's input = of the last layer's output (D / 2; D for MAF)
    WideRec rc;
    const float* masks;  // MAF: the layers' binary masks (shared by both nets), gradients are mask * sum; no biases
};

template <int TB>
__device__ __forceinline__ void gw_tile_loop(const float* __restrict__ rec, int64_t t0, int64_t t1, int blocks, int a_blk,
                                             int b_blk0, int lane, f4 (&acc)[TB], float& bsum) {
    const int off = (lane & 15) * 16 + 4 * (lane >> 4);  // row r, samples 4q .. 4q+3: the MFMA operand itself
    if (t0 >= t1) return;
    f4 an = *reinterpret_cast<const f4*>(rec + (t0 * blocks + a_blk) * 256 + off), bn[TB];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) bn[tb] = *reinterpret_cast<const f4*>(rec + (t0 * blocks + b_blk0 + tb) * 256 + off);
    for (int64_t t = t0; t < t1; ++t) {
        const f4 ac = an;
        f4 bc[TB];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) bc[tb] = bn[tb];
        const int64_t tn = t + 1 < t1 ? t + 1 : t;  // the next record's blocks fly while this one's MFMAs run
        an = *reinterpret_cast<const f4*>(rec + (tn * blocks + a_blk) * 256 + off);
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) bn[tb] = *reinterpret_cast<const f4*>(rec + (tn * blocks + b_blk0 + tb) * 256 + off);
        bsum += (ac[0] + ac[1]) + (ac[2] + ac[3]);
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[tb] = mfma4(ac[i], bc[tb][i], acc[tb]);
    }
}

template <int HT, int UT>
__global__ void __launch_bounds__(256)
wide_gw_kernel(WideGwArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int job = blockIdx.x / a.G, slice = blockIdx.x - job * a.G;
    const int layer = job >> 1, net = job & 1;
    const int H = a.H, U = a.U, L = a.L;
    // delta tiles (rows = the layer's outputs) and input tiles (rows = its inputs) of this job
    const bool first = layer == 0, last = layer == L;
    const int TA = last ? HT : UT, TB = first ? HT : UT;
    const int a_blk0 = last ? a.rc.dout(net, 0) : a.rc.da(layer, net, 0);
    const int b_blk0 = first ? a.rc.x(0) : a.rc.act(layer - 1, net, 0);
    const int din = first ? H : U, dout = last ? H : U;
    // parameter offsets: RealNVP (bijectors.py:222-235) per MLP layer [W_t | W_s | b_t | b_s], W[in][out];
    // MAF (bijectors.py:698-740) per layer [W_mu | W_alpha], no biases, one mask per layer
    const bool maf = a.masks != nullptr;
    int64_t off = 0, moff = 0;
    for (int l = 0; l < layer; ++l) {
        off += 2 * ((int64_t)(l == 0 ? H : U) * U + (maf ? 0 : U));
        moff += (int64_t)(l == 0 ? H : U) * U;
    }
    const int64_t wbase = off + (int64_t)net * din * dout, bbase = off + 2 * (int64_t)din * dout + (int64_t)net * dout;
    const int64_t per = (a.ntiles + a.G - 1) / a.G;
    const int64_t t0 = slice * per, t1 = (t0 + per) < a.ntiles ? (t0 + per) : a.ntiles;
    float* part = a.partials + (int64_t)slice * a.P;

    const int ta = wave;  // one delta tile per wave (TA <= 4)
    if (ta >= TA) return;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    constexpr int TBM = HT > UT ? HT : UT;
    f4 acc[TBM];
#pragma unroll
    for (int tb = 0; tb < TBM; ++tb) acc[tb] = zero;
    float bsum = 0.f;
    if (TB == HT) {
        f4 (&acc_h)[HT] = reinterpret_cast<f4 (&)[HT]>(acc);
        gw_tile_loop<HT>(a.rec, t0, t1, a.rc.blocks(), a_blk0 + ta, b_blk0, lane, acc_h, bsum);
    } else {
        f4 (&acc_u)[UT] = reinterpret_cast<f4 (&)[UT]>(acc);
        gw_tile_loop<UT>(a.rec, t0, t1, a.rc.blocks(), a_blk0 + ta, b_blk0, lane, acc_u, bsum);
    }
    // acc[tb]: lane (c, q) register j = sum_s d[o = 16 ta + 4q + j][s] in[k = 16 tb + c][s]  ->  W[k][o]
#pragma unroll
    for (int tb = 0; tb < TBM; ++tb) {
        if (tb >= TB) break;
        const int k = 16 * tb + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = 16 * ta + 4 * q + j;
            if (k < din && o < dout)
                part[wbase + (int64_t)k * dout + o] = maf ? acc[tb][j] * a.masks[moff + (int64_t)k * dout + o] : acc[tb][j];
        }
    }
    if (maf) return;
    // bias: row sum of the delta block -- this lane holds samples 4q .. 4q+3 of row c
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (q == 0 && 16 * ta + c < dout) part[bbase + 16 * ta + c] = bsum;
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
static int wide_bwd_lmax(int UT) { return UT <= 2 ? 3 : 2; }  // the sigmoids of every layer stay in registers: L * UT <= 8

bool wide_bwd_supported(int D, int L, int U) {
    if (!wide_supported(D, L, U)) return false;
    const WideLayout wl = wide_layout(D, L, U);
    if (L > wide_bwd_lmax(wl.UT)) return false;
    const WideBwdLayout bl{wl.UT, wl.HT, L};
    return (size_t)(wl.floats() + bl.floats()) * sizeof(float) <= 156 * 1024;
}

static int64_t wide_bwd_chunks(int64_t N) { return (N + kWideBwdChunk - 1) / kWideBwdChunk; }

int64_t wide_bwd_workspace(int64_t N, int D, int L, int U) {
    const WideLayout wl = wide_layout(D, L, U);
    const WideRec rc{wl.UT, wl.HT, L};
    const int64_t chunk = N < kWideBwdChunk ? N : kWideBwdChunk;
    const int64_t rec = ((chunk + 15) / 16) * rc.blocks() * 1024;
    const int64_t P = coupling_num_params(D, L, U, 1);
    return rec + wide_bwd_chunks(N) * kWideBwdSlices * P * 4;
}

template <int HT, int UT, int L>
static int launch_wide_bwd_t(const WideBwdArgs& a, const WideGwArgs& g, const WideLayout& wl, const WideBwdLayout& bl,
                             size_t smem, hipStream_t st) {
    auto k = coupling_wide_bwd_kernel<HT, UT, L>;
    if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return fail(TNF_ELAUNCH, "coupling_wide_bwd: cannot reserve %zu B of LDS", smem);
    const int64_t ntiles = (a.n1 - a.n0 + 15) / 16;
    int64_t bx = (ntiles + 3) / 4;
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(k, dim3((unsigned)bx), dim3(256), smem, st, a, wl, bl, g.rc);
    hipLaunchKernelGGL((wide_gw_kernel<HT, UT>), dim3((unsigned)(2 * (L + 1) * g.G)), dim3(256), 0, st, g);
    return TNF_OK;
}

template <int HT, int UT>
static int launch_wide_bwd_l(const WideBwdArgs& a, const WideGwArgs& g, const WideLayout& wl, const WideBwdLayout& bl,
                             size_t smem, hipStream_t st) {
    if (wl.L == 1) return launch_wide_bwd_t<HT, UT, 1>(a, g, wl, bl, smem, st);
    if (wl.L == 2) return launch_wide_bwd_t<HT, UT, 2>(a, g, wl, bl, smem, st);
    if constexpr (UT <= 2) return launch_wide_bwd_t<HT, UT, 3>(a, g, wl, bl, smem, st);
    return fail(TNF_EUNSUPPORTED, "coupling_wide_bwd: num_layers=%d with %d unit tiles", wl.L, UT);
}

template <int HT>
static int launch_wide_bwd_u(const WideBwdArgs& a, const WideGwArgs& g, const WideLayout& wl, const WideBwdLayout& bl,
                             size_t smem, hipStream_t st) {
    switch (wl.UT) {
        case 1: return launch_wide_bwd_l<HT, 1>(a, g, wl, bl, smem, st);
        case 2: return launch_wide_bwd_l<HT, 2>(a, g, wl, bl, smem, st);
        case 3: return launch_wide_bwd_l<HT, 3>(a, g, wl, bl, smem, st);
        default: return launch_wide_bwd_l<HT, 4>(a, g, wl, bl, smem, st);
    }
}

// One shared parameter row (M_p = 1); N = all samples of the call (M batches of one row are one batch).
// g_params accumulates (the caller zeroed it).  ws: wide_bwd_workspace(N, D, L, U) bytes.
int launch_coupling_backward_wide(const float* z, const float* params, const float* g_zout, const float* g_ld, float* g_z,
                                  float* g_params, int64_t N, int D, int L, int U, int upper, int inverse,
                                  int64_t gpstride, void* ws, hipStream_t st) {
    if (!wide_bwd_supported(D, L, U)) return fail(TNF_EUNSUPPORTED, "coupling_wide_bwd: D=%d L=%d U=%d", D, L, U);
    if (N <= 0) return TNF_OK;
    diag_count(TNF_DIAG_BWD_WIDE);
    const WideLayout wl = wide_layout(D, L, U);
    const WideBwdLayout bl{wl.UT, wl.HT, L};
    const WideRec rc{wl.UT, wl.HT, L};
    const size_t smem = (size_t)(wl.floats() + bl.floats()) * sizeof(float);
    const int64_t P = coupling_num_params(D, L, U, upper);
    const int64_t chunk = N < kWideBwdChunk ? N : kWideBwdChunk;
    char* wsb = reinterpret_cast<char*>(ws);
    float* rec = reinterpret_cast<float*>(wsb);
    float* partials = reinterpret_cast<float*>(wsb + ((chunk + 15) / 16) * rc.blocks() * 1024);
    const int64_t nchunks = wide_bwd_chunks(N);
    for (int64_t ci = 0; ci < nchunks; ++ci) {
        WideBwdArgs a{z, params, g_zout, g_ld, g_z, rec, partials + ci * kWideBwdSlices * P, N, ci * kWideBwdChunk,
                      (ci + 1) * kWideBwdChunk < N ? (ci + 1) * kWideBwdChunk : N, D, U, upper, inverse};
        WideGwArgs g{rec, a.partials, (a.n1 - a.n0 + 15) / 16, D / 2, U, L, (int)P, kWideBwdSlices, rc, nullptr};
        int rcode;
        switch (wl.HT) {
            case 1: rcode = launch_wide_bwd_u<1>(a, g, wl, bl, smem, st); break;
            case 2: rcode = launch_wide_bwd_u<2>(a, g, wl, bl, smem, st); break;
            case 3: rcode = launch_wide_bwd_u<3>(a, g, wl, bl, smem, st); break;
            default: rcode = launch_wide_bwd_u<4>(a, g, wl, bl, smem, st); break;
        }
        if (rcode != TNF_OK) return rcode;
    }
    const int rcode = check_launch("coupling_wide_bwd");
    if (rcode) return rcode;
    return launch_backward_reduce(TNF_F32, partials, g_params, 1, (int)(nchunks * kWideBwdSlices), P, gpstride, st);
}

// ---------------------------------------------------------------------------
// MAF (bijectors.py:597-806), inverse direction -- what log_prob training differentiates -- for the shapes the one-kernel
// matrix-pipe backward (maf_bwd_mfma.hip, D <= 32) does not reach: D up to 64 (D % 4 == 0), num_units up to 64.  Same two
// passes: pass 1 below, pass 2 = wide_gw_kernel with the masks applied on the way out (and no biases).  The forward
// image (masked, folded: maf_tile.h) sits in LDS; the transposed masked raw weights of the way back would not fit beside it
// at D = U = 64 (2 x 96 KB), so they are an image in global memory (maf_bwd_timage_kernel, 96 KB, L2-resident) read 16 bytes
// per lane and group.
//   out = (z - mu) e^-alpha, ld = sum alpha:   dz = g e^-alpha + (nets' input gradient),  dmu = -g e^-alpha,
//   dalpha = -g out + g_ld
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
maf_bwd_timage_kernel(const float* __restrict__ p, const float* __restrict__ mk, float* __restrict__ timg, WideBwdLayout bl,
                      int D, int U) {
    // group = blockIdx.x, in the order of WideBwdLayout (HT = DT): [W2 (net, ui, mo) | Wh (l, net, ui, uo) | W0 (net, m, ut)]
    const int g = blockIdx.x, lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const int n2 = bl.nB2(), nh = bl.nBh();
    f4 v;
    if (g < n2) {  // output layer U -> D
        const int net = g / (bl.UT * bl.HT), ui = (g / bl.HT) % bl.UT, mo = g % bl.HT;
        const float* w = p + 2 * (int64_t)D * U + (int64_t)(bl.L - 1) * 2 * U * U + (int64_t)net * U * D;
        const float* m = mk + (int64_t)D * U + (int64_t)(bl.L - 1) * U * U;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 16 * ui + r, o = 16 * mo + 4 * q + j;
            const bool ok = k < U && o < D;
            v[j] = ld_sel(w, k * D + o, ok) * ld_sel(m, k * D + o, ok);
        }
    } else if (g < n2 + (bl.L - 1) * nh) {  // hidden layer l + 1: U -> U
        const int gg = g - n2, l = gg / nh, rr = gg % nh;
        const int net = rr / (bl.UT * bl.UT), ui = (rr / bl.UT) % bl.UT, uo = rr % bl.UT;
        const float* w = p + 2 * (int64_t)D * U + (int64_t)l * 2 * U * U + (int64_t)net * U * U;
        const float* m = mk + (int64_t)D * U + (int64_t)l * U * U;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ki = 16 * ui + r, ko = 16 * uo + 4 * q + j;
            const bool ok = ki < U && ko < U;
            v[j] = ld_sel(w, ki * U + ko, ok) * ld_sel(m, ki * U + ko, ok);
        }
    } else {  // layer 0: D -> U
        const int gg = g - n2 - (bl.L - 1) * nh;
        const int net = gg / (bl.HT * bl.UT), mm = (gg / bl.UT) % bl.HT, ut = gg % bl.UT;
        const float* w = p + (int64_t)net * D * U;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = 16 * mm + r, u = 16 * ut + 4 * q + j;
            const bool ok = f < D && u < U;
            v[j] = ld_sel(w, f * U + u, ok) * ld_sel(mk, f * U + u, ok);
        }
    }
    *reinterpret_cast<f4*>(timg + (int64_t)g * 256 + lane * 4) = v;
}

struct MafWideBwdArgs {
    const float* z;
    const float* params;
    const float* masks;
    const float* timg;    // transposed masked raw image (global)
    const float* g_zout;
    const float* g_ld;
    float* g_z;
    float* rec;
    int64_t n0, n1;
    int D, U;
};

template <int DT, int UT, int L>
__global__ void __launch_bounds__(256)
maf_wide_bwd_kernel(MafWideBwdArgs a, MafLayout wl, WideBwdLayout bl, WideRec rc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = a.D, U = a.U;
    float* img = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    build_maf_image(img, a.params, a.masks, wl, D, U, lane, wave, 4, 0);
    __syncthreads();
    const float* wsrc = img + lane * 4;
    const float* bsrc = img + wl.NWG() * 256 + q * 4;
    const float* tsrc = a.timg + lane * 4;
    auto wgrp = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(wsrc + g * 256); };
    auto bgrp = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(bsrc + g * 16); };
    auto tgrp = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(tsrc + (int64_t)g * 256); };
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    bool fok[DT];
#pragma unroll
    for (int mm = 0; mm < DT; ++mm) fok[mm] = 16 * mm + 4 * q < D;

    const int64_t tile0 = a.n0 >> 4, ntiles = (a.n1 - a.n0 + 15) >> 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
        const int64_t row = (tile0 + t) * 16 + s;
        const bool row_ok = row < a.n1;
        const int64_t rowc = row_ok ? row : a.n1 - 1;
        float* rec = a.rec + t * (int64_t)rc.blocks() * 256;
        const float* zr = a.z + rowc * D + 4 * q;
        const float* gr = a.g_zout + rowc * D + 4 * q;
        f4 x[DT], g[DT];
#pragma unroll
        for (int mm = 0; mm < DT; ++mm) {
            x[mm] = fok[mm] ? *reinterpret_cast<const f4*>(zr + 16 * mm) : zero;
            g[mm] = (fok[mm] && row_ok) ? *reinterpret_cast<const f4*>(gr + 16 * mm) : zero;
            store_block(rec + rc.x(mm) * 256, x[mm], s, q);
        }
        const float gl = row_ok ? a.g_ld[rowc] : 0.f;
        asm volatile("" ::: "memory");

        f4 r[L][2][UT];
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            f4 at = zero, as = zero;  // MAF has no biases: layer 0 starts from zero
#pragma unroll
            for (int mm = 0; mm < DT; ++mm) {
                const f4 wt = wgrp(wl.g_w0(0, ut, mm)), ws = wgrp(wl.g_w0(1, ut, mm));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    at = mfma4(wt[j], x[mm][j], at);
                    as = mfma4(ws[j], x[mm][j], as);
                }
            }
            r[0][0][ut] = sig2_4(at);
            r[0][1][ut] = sig2_4(as);
        }
#pragma unroll
        for (int l = 0; l < L - 1; ++l)
#pragma unroll
            for (int uo = 0; uo < UT; ++uo) {
                f4 at = bgrp(wl.b_bh(l, 0, uo)), as = bgrp(wl.b_bh(l, 1, uo));
#pragma unroll
                for (int ui = 0; ui < UT; ++ui) {
                    const f4 wt = wgrp(wl.g_wh(l, 0, uo, ui)), ws = wgrp(wl.g_wh(l, 1, uo, ui));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        at = mfma4(wt[j], r[l][0][ui][j], at);
                        as = mfma4(ws[j], r[l][1][ui][j], as);
                    }
                }
                r[l + 1][0][uo] = sig2_4(at);
                r[l + 1][1][uo] = sig2_4(as);
            }
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int net = 0; net < 2; ++net)
#pragma unroll
                for (int ut = 0; ut < UT; ++ut) {
                    f4 h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) h[j] = __builtin_fmaf(-2.f, r[l][net][ut][j], 1.f);
                    store_block(rec + rc.act(l, net, ut) * 256, h, s, q);
                }
        f4 dout[2][DT], dx[DT];
#pragma unroll
        for (int mo = 0; mo < DT; ++mo) {
            f4 mu = bgrp(wl.b_b2(0, mo)), al2 = bgrp(wl.b_b2(1, mo));
#pragma unroll
            for (int ui = 0; ui < UT; ++ui) {
                const f4 wt = wgrp(wl.g_w2(0, mo, ui)), ws = wgrp(wl.g_w2(1, mo, ui));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mu = mfma4(wt[j], r[L - 1][0][ui][j], mu);
                    al2 = mfma4(ws[j], r[L - 1][1][ui][j], al2);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float em = __builtin_amdgcn_exp2f(-al2[j]);
                const float out = (x[mo][j] - mu[j]) * em;
                const float dz = g[mo][j] * em;
                const bool real = 16 * mo + 4 * q + j < D;
                dx[mo][j] = dz;                                               // direct path; the nets' share is added below
                dout[0][mo][j] = real ? -dz : 0.f;                            // d mu
                dout[1][mo][j] = real ? __builtin_fmaf(-g[mo][j], out, gl) : 0.f;  // d alpha
            }
            store_block(rec + rc.dout(0, mo) * 256, dout[0][mo], s, q);
            store_block(rec + rc.dout(1, mo) * 256, dout[1][mo], s, q);
        }
        f4 dh[2][UT];
#pragma unroll
        for (int net = 0; net < 2; ++net)
#pragma unroll
            for (int ui = 0; ui < UT; ++ui) {
                f4 acc = zero;
#pragma unroll
                for (int mo = 0; mo < DT; ++mo) {
                    const f4 wb = tgrp(bl.g_b2(net, ui, mo));
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = mfma4(wb[j], dout[net][mo][j], acc);
                }
                dh[net][ui] = acc;
            }
#pragma unroll
        for (int l = L - 2; l >= 0; --l)
#pragma unroll
            for (int net = 0; net < 2; ++net) {
                f4 da[UT];
#pragma unroll
                for (int uo = 0; uo < UT; ++uo) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float rr = r[l + 1][net][uo][j];
                        da[uo][j] = dh[net][uo][j] * (4.f * rr * (1.f - rr));
                    }
                    store_block(rec + rc.da(l + 1, net, uo) * 256, da[uo], s, q);
                }
#pragma unroll
                for (int ui = 0; ui < UT; ++ui) {
                    f4 acc = zero;
#pragma unroll
                    for (int uo = 0; uo < UT; ++uo) {
                        const f4 wb = tgrp(bl.g_bh(l, net, ui, uo));
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc = mfma4(wb[j], da[uo][j], acc);
                    }
                    dh[net][ui] = acc;
                }
            }
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            f4 da[UT];
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float rr = r[0][net][ut][j];
                    da[ut][j] = dh[net][ut][j] * (4.f * rr * (1.f - rr));
                }
                store_block(rec + rc.da(0, net, ut) * 256, da[ut], s, q);
            }
#pragma unroll
            for (int mm = 0; mm < DT; ++mm)
#pragma unroll
                for (int ut = 0; ut < UT; ++ut) {
                    const f4 wb = tgrp(bl.g_b0(net, mm, ut));
#pragma unroll
                    for (int j = 0; j < 4; ++j) dx[mm] = mfma4(wb[j], da[ut][j], dx[mm]);
                }
        }
        if (row_ok) {
#pragma unroll
            for (int mm = 0; mm < DT; ++mm)
                if (fok[mm]) *reinterpret_cast<f4*>(a.g_z + row * D + 4 * q + 16 * mm) = dx[mm];
        }
    }
}

static MafLayout maf_wide_layout(int D, int L, int U) {
    MafLayout wl;
    wl.UT = (U + 15) / 16;
    wl.DT = (D + 15) / 16;
    wl.L = L;
    return wl;
}

bool maf_wide_bwd_supported(int D, int L, int U) {
    if (D < 4 || D > 64 || (D % 4) != 0 || U < 1 || U > 64 || L < 1) return false;
    const MafLayout wl = maf_wide_layout(D, L, U);
    if (L > wide_bwd_lmax(wl.UT)) return false;
    return (size_t)wl.floats() * sizeof(float) <= 156 * 1024;
}

static int64_t maf_num_params_(int D, int L, int U) { return 2 * (2 * (int64_t)D * U + (int64_t)(L - 1) * U * U); }

// workspace: [transposed image | records of one chunk | partial rows]
int64_t maf_wide_bwd_workspace(int64_t N, int D, int L, int U) {
    const MafLayout wl = maf_wide_layout(D, L, U);
    const WideBwdLayout bl{wl.UT, wl.DT, L};
    const WideRec rc{wl.UT, wl.DT, L};
    const int64_t chunk = N < kWideBwdChunk ? N : kWideBwdChunk;
    return (int64_t)bl.floats() * 4 + ((chunk + 15) / 16) * rc.blocks() * 1024 +
           wide_bwd_chunks(N) * kWideBwdSlices * maf_num_params_(D, L, U) * 4;
}

template <int DT, int UT, int L>
static int launch_maf_wide_t(const MafWideBwdArgs& a, const WideGwArgs& g, const MafLayout& wl, const WideBwdLayout& bl,
                             size_t smem, hipStream_t st) {
    auto k = maf_wide_bwd_kernel<DT, UT, L>;
    if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return fail(TNF_ELAUNCH, "maf_wide_bwd: cannot reserve %zu B of LDS", smem);
    const int64_t ntiles = (a.n1 - a.n0 + 15) / 16;
    int64_t bx = (ntiles + 3) / 4;
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(k, dim3((unsigned)bx), dim3(256), smem, st, a, wl, bl, g.rc);
    hipLaunchKernelGGL((wide_gw_kernel<DT, UT>), dim3((unsigned)(2 * (L + 1) * g.G)), dim3(256), 0, st, g);
    return TNF_OK;
}
template <int DT, int UT>
static int launch_maf_wide_l(const MafWideBwdArgs& a, const WideGwArgs& g, const MafLayout& wl, const WideBwdLayout& bl,
                             size_t smem, hipStream_t st) {
    if (wl.L == 1) return launch_maf_wide_t<DT, UT, 1>(a, g, wl, bl, smem, st);
    if (wl.L == 2) return launch_maf_wide_t<DT, UT, 2>(a, g, wl, bl, smem, st);
    if constexpr (UT <= 2) return launch_maf_wide_t<DT, UT, 3>(a, g, wl, bl, smem, st);
    return fail(TNF_EUNSUPPORTED, "maf_wide_bwd: num_layers=%d with %d unit tiles", wl.L, UT);
}
template <int DT>
static int launch_maf_wide_u(const MafWideBwdArgs& a, const WideGwArgs& g, const MafLayout& wl, const WideBwdLayout& bl,
                             size_t smem, hipStream_t st) {
    switch (wl.UT) {
        case 1: return launch_maf_wide_l<DT, 1>(a, g, wl, bl, smem, st);
        case 2: return launch_maf_wide_l<DT, 2>(a, g, wl, bl, smem, st);
        case 3: return launch_maf_wide_l<DT, 3>(a, g, wl, bl, smem, st);
        default: return launch_maf_wide_l<DT, 4>(a, g, wl, bl, smem, st);
    }
}

int launch_maf_backward_wide(const float* z, const float* params, const float* masks, const float* g_zout, const float* g_ld,
                             float* g_z, float* g_params, int64_t N, int D, int L, int U, int64_t gpstride, void* ws,
                             hipStream_t st) {
    if (!maf_wide_bwd_supported(D, L, U)) return fail(TNF_EUNSUPPORTED, "maf_wide_bwd: D=%d L=%d U=%d", D, L, U);
    if (N <= 0) return TNF_OK;
    diag_count(TNF_DIAG_MAF_BWD_MFMA);
    const MafLayout wl = maf_wide_layout(D, L, U);
    const WideBwdLayout bl{wl.UT, wl.DT, L};
    const WideRec rc{wl.UT, wl.DT, L};
    const size_t smem = (size_t)wl.floats() * sizeof(float);
    const int64_t P = maf_num_params_(D, L, U);
    const int64_t chunk = N < kWideBwdChunk ? N : kWideBwdChunk;
    char* wsb = reinterpret_cast<char*>(ws);
    float* timg = reinterpret_cast<float*>(wsb);
    float* rec = reinterpret_cast<float*>(wsb + (int64_t)bl.floats() * 4);
    float* partials = reinterpret_cast<float*>(wsb + (int64_t)bl.floats() * 4 + ((chunk + 15) / 16) * rc.blocks() * 1024);
    hipLaunchKernelGGL(maf_bwd_timage_kernel, dim3((unsigned)(bl.floats() / 256)), dim3(64), 0, st, params, masks, timg, bl, D, U);
    const int64_t nchunks = wide_bwd_chunks(N);
    for (int64_t ci = 0; ci < nchunks; ++ci) {
        float* part = partials + ci * kWideBwdSlices * P;
        MafWideBwdArgs a{z, params, masks, timg, g_zout, g_ld, g_z, rec, ci * kWideBwdChunk,
                         (ci + 1) * kWideBwdChunk < N ? (ci + 1) * kWideBwdChunk : N, D, U};
        WideGwArgs g{rec, part, (a.n1 - a.n0 + 15) / 16, D, U, L, (int)P, kWideBwdSlices, rc, masks};
        int rcode;
        switch (wl.DT) {
            case 1: rcode = launch_maf_wide_u<1>(a, g, wl, bl, smem, st); break;
            case 2: rcode = launch_maf_wide_u<2>(a, g, wl, bl, smem, st); break;
            case 3: rcode = launch_maf_wide_u<3>(a, g, wl, bl, smem, st); break;
            default: rcode = launch_maf_wide_u<4>(a, g, wl, bl, smem, st); break;
        }
        if (rcode != TNF_OK) return rcode;
    }
    const int rcode = check_launch("maf_wide_bwd");
    if (rcode) return rcode;
    return launch_backward_reduce(TNF_F32, partials, g_params, 1, (int)(nchunks * kWideBwdSlices), P, gpstride, st);
}

}  // namespace tnf
