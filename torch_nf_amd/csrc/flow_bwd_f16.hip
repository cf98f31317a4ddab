// Whole-flow backward of loss = f(NormFlow.log_prob(z)) in ONE kernel, split-f16 matrix path.
//
// The coupling stack is invertible, so nothing is saved by the forward except its output z0: the
// kernel starts from (z0, g_log_prob) and walks the layers in the order opposite to the inverse
// pass.  For each layer it
//   1. recomputes the twin MLP on the conditioner half (unchanged by the layer) -> r, t, s;
//   2. rebuilds the layer's input  y = y_out e^s + t  and the output deltas
//        d y = g e^-s,  d t = -d y,  d s = -g y_out - g_log_prob;
//   3. propagates the deltas through the MLPs with the transposed RAW weights as A operands
//      (the accumulator -> B-operand chaining of the forward pass works unchanged);
//   4. forms the weight gradients dW[k][o] = sum_s a[k,s] d[o,s] as MFMAs that contract over the
//      SAMPLE index: every activation / delta tile is transposed through a wave-private LDS
//      scratch, split, and multiplied; bias gradients are the same tiles times a ones operand;
//   5. undoes the folded BatchNorm/Affine in front of the layer, v = (v - B)/A, g *= A, and on
//      layers with an Affine reduces dA = sum g v, dB = sum g over the tile.
// Every contraction is three f16 MFMAs on hi/lo-split operands with fp32 accumulate (f16_tile.h).
// g_log_prob is pre-scaled by a power of two so that the deltas stay clear of the f16 subnormals;
// the scale comes out again when the gradients leave the kernel.
//
// A wave carries one 16-sample tile through all 2S layers, so the per-layer gradient accumulators
// cannot live in registers: they are fixed-point ds_add_u32 targets in LDS (AccLayout, 2S x ~11 KB) and leave with one global atomic per parameter per workgroup.  That leaves
// no room for all layers' operand images, so the waves of a workgroup step through the layers
// together: the next layer's image (forward operands | transposed operands | fold constants,
// ~21 KB, built once per call by flow_rev_images_kernel) is prefetched into registers during a
// step and committed to the other half of a two-slot LDS ring at its end, one barrier per step.
#include "f16_tile.h"
#include "tnf_common.h"

namespace tnf {

#ifndef TNF_REV_NW
#define TNF_REV_NW 12
#endif
#ifndef TNF_REV_NSCR
#define TNF_REV_NSCR 1
#endif
constexpr int kRevNW = TNF_REV_NW;
constexpr int kRevNScr = TNF_REV_NSCR;  // transposition scratch tiles per wave (1: both operands share one, LDS ops are in order)

// Transposed raw weights as split-f16 A operands, lane (r = lane&15, q = lane>>4), 16 B per lane per group:
//   W2 (d h_last[k] = sum_o W2[k][o] d out[o]):  m = k = r.  H = 32: K slot (q, i) is o = 16 (i>>2) + 4q + (i&3),
//       one group of 8 hi + one of 8 lo per net;  H = 16: K = o = 4q + j, one group [hi(4) | lo(4)] per net.
//   Wh (d h_l[k_in] = sum Wh[k_in][k_out] d a[k_out]):  m = k_in = r, K = k_out = 4q + j, [hi | lo].
//   W0 (d x[f] = sum_u W0[f][u] d a0[u]):  m = f = 16 mm + r, K = u = 4q + j, [hi | lo] per (net, mm).
template <int H, int L>
struct B16Image {
    static constexpr int HT = (H + 15) / 16;
    static constexpr int NW2 = (H == 32) ? 4 : 2;
    static constexpr int NWG = NW2 + 2 * (L - 1) + 2 * HT;
    static constexpr int FLOATS = NWG * 256;
    __device__ static constexpr int g_w2(int net, int part) { return (H == 32) ? net * 2 + part : net; }
    __device__ static constexpr int g_wh(int l, int net) { return NW2 + 2 * l + net; }
    __device__ static constexpr int g_w0(int net, int mm) { return NW2 + 2 * (L - 1) + net * HT + mm; }
};

// One layer's streamed image: [forward f16 image | transposed f16 image | A (D) | B (D) | 1/A (D) | -B/A (D)]
template <int H, int L>
struct RevImage {
    static constexpr int D = 2 * H;
    static constexpr int F_OFF = 0;
    static constexpr int B_OFF = F16Image<H, L>::FLOATS;
    static constexpr int C_OFF = B_OFF + B16Image<H, L>::FLOATS;
    static constexpr int FLOATS = C_OFF + 4 * D;
};

// hscale multiplies the transposed HIDDEN and OUTPUT weights (the operands whose product is followed by a tanh'): 1 for
// the kernels that form tanh' = 1 - h^2, 4 for the pair kernel, whose tanh' is 4 r (1 - r) with the 4 carried here.
template <int H, int L>
__device__ __forceinline__ void build_b16_image(float* img, const float* __restrict__ p, int U, int lane, float hscale = 1.f) {
    typedef B16Image<H, L> Img;
    constexpr int HT = Img::HT;
    const int r = lane & 15, q = lane >> 4;
    u4* grp = reinterpret_cast<u4*>(img) + lane;
    {   // layer 0: W0[f][u]
        const float* w[2] = {p, p + H * U};
#pragma unroll
        for (int net = 0; net < 2; ++net)
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = 16 * mm + r, u = 4 * q + j;
                    v[j] = ld_sel(w[net], f * U + u, f < H && u < U);
                }
                u4 o;
                split2(v[0], v[1], o[0], o[2]);
                split2(v[2], v[3], o[1], o[3]);
                grp[Img::g_w0(net, mm) * 64] = o;
            }
        p += 2 * H * U + 2 * U;
    }
#pragma unroll
    for (int l = 0; l < L - 1; ++l) {
        const float* w[2] = {p, p + U * U};
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ki = r, ko = 4 * q + j;
                v[j] = hscale * ld_sel(w[net], ki * U + ko, ki < U && ko < U);
            }
            u4 o;
            split2(v[0], v[1], o[0], o[2]);
            split2(v[2], v[3], o[1], o[3]);
            grp[Img::g_wh(l, net) * 64] = o;
        }
        p += 2 * U * U + 2 * U;
    }
    {   // output layer: W2[k][o]
        const float* w[2] = {p, p + U * H};
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            if constexpr (H == 32) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int o = 16 * (i >> 2) + 4 * q + (i & 3);
                    v[i] = hscale * ld_sel(w[net], r * H + o, r < U);
                }
                u4 hi, lo;
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) split2(v[2 * pp], v[2 * pp + 1], hi[pp], lo[pp]);
                grp[Img::g_w2(net, 0) * 64] = hi;
                grp[Img::g_w2(net, 1) * 64] = lo;
            } else {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = hscale * ld_sel(w[net], r * H + 4 * q + j, r < U);
                u4 o;
                split2(v[0], v[1], o[0], o[2]);
                split2(v[2], v[3], o[1], o[3]);
                grp[Img::g_w2(net, 0) * 64] = o;
            }
        }
    }
}

// grid (2S, Mp) x 64 threads: the streamed image of layer c of context row m
template <int H, int L>
__global__ void __launch_bounds__(64)
flow_rev_images_kernel(const float* __restrict__ params, const float* __restrict__ bn_mean,
                       const float* __restrict__ bn_alpha, float* __restrict__ rimg, int S, int U, int64_t pstride,
                       int64_t Mp, float hscale) {
    typedef RevImage<H, L> R;
    constexpr int D = 2 * H;
    const int c = blockIdx.x, lane = threadIdx.x;
    const int64_t m = grid_m();
    if (m >= Mp) return;
    const FlowLayout fl = flow_layout(D, S, L, U);
    const float* prow = params + m * pstride;
    const float* p = prow + (c >> 1) * fl.stage + ((c & 1) ? fl.p_up : 0);
    float* img = rimg + (m * 2 * S + c) * (int64_t)R::FLOATS;
    build_f16_image<H, L>(img + R::F_OFF, p, U, lane);
    build_b16_image<H, L>(img + R::B_OFF, p, U, lane, hscale);
    float* fc = img + R::C_OFF;
    for (int d = lane; d < D; d += 64) {  // the inverse-pass fold of flow_fold_kernel (coupling_mfma.hip)
        const float alpha = bn_alpha[c * D + d], mu = bn_mean[c * D + d];
        float ea = 1.f, shift = 0.f;
        if (c & 1) {
            const float* ap = prow + (c >> 1) * fl.stage + fl.p_up + fl.p_low;
            ea = expf(ap[d]);
            shift = ap[D + d];
        }
        const float A = alpha / ea, B = mu - shift * A;
        fc[d] = A;
        fc[D + d] = B;
        fc[2 * D + d] = 1.f / A;
        fc[3 * D + d] = -B / A;
    }
}

__global__ void __launch_bounds__(256)
flow_gmax_kernel(const float* __restrict__ g, int64_t n, unsigned* __restrict__ out, const int* __restrict__ gate) {
    if (gate && *gate == 0) return;  // a conditionally needed launch (tnf_set_launch_gate): nothing to do
    __shared__ float red[4];
    float m = 0.f;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(g) & 15) == 0 ? n >> 2 : 0;  // 16-byte part, four loads in flight
    const f4* g4 = reinterpret_cast<const f4*>(g);
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * step) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i + u * step < n4) ? g4[i + u * step] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u][0]), fabsf(v[u][1]))), fmaxf(fabsf(v[u][2]), fabsf(v[u][3])));
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) m = fmaxf(m, fabsf(g[i]));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.f) atomicMax(out, __float_as_uint(m));  // one atomic per block: they serialise on the one word
    }
}

int launch_gmax(const float* g, int64_t n, unsigned* out, hipStream_t st) {
    if (n <= 0) return TNF_OK;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(flow_gmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, n, out, g_launch_gate);
    return check_launch("gmax");
}

// acc layout (lane (s, q), reg j = row 4q + j, col s)  ->  operand layout with K = samples:
// lane (c = lane&15, kq = lane>>4) gets element [row c][samples 4 kq .. 4 kq + 3].
constexpr int kScr = 16 * 20;  // one padded tile (rows 16-byte aligned, bank-conflict-free both ways)
__device__ __forceinline__ f4 transpose16(f4 v, float* scr, int lane) {
    const int s = lane & 15, q = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) scr[(4 * q + j) * 20 + s] = v[j];
    return *reinterpret_cast<const f4*>(scr + s * 20 + 4 * q);
}

struct T16 {
    h4 hi, lo;
};
__device__ __forceinline__ T16 tsplit(f4 v, float* scr, int lane) {
    T16 o;
    split4(transpose16(v, scr, lane), o.hi, o.lo);
    return o;
}
// The same re-layout on the matrix pipe, for a tile that is already split: read as an A operand the accumulator
// layout is T^T (rows = samples, K = the tile's rows), so T^T . I comes back as [row = lane&15][samples 4q .. 4q+3].
// Each product has one non-zero term and f16 values are exact in fp32, so the round trip is exact; it costs two
// MFMAs and four conversions where the LDS path costs five LDS instructions and a fresh split.
__device__ __forceinline__ h4 pack4(f4 v) {
    const u2 w = {__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[0], v[1])),
                  __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[2], v[3]))};
    return __builtin_bit_cast(h4, w);
}
__device__ __forceinline__ T16 mtrans(h4 hi, h4 lo, h4 ident) {
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    T16 o;
    o.hi = pack4(mfma16h(hi, ident, zero));
    o.lo = pack4(mfma16h(lo, ident, zero));
    return o;
}
// The same through LDS with gfx950's transposing read: every lane stores its four f16 (rows 4q .. 4q+3 of column s)
// into a [sample][row] image with 32-byte rows, and ds_read_b64_tr_b16 hands lane (c, kq) the samples 4kq .. 4kq+3 of
// row c (cdna_hip_programming.md T10: per 16-lane group a 4-row x 16-column block, delivered column-major) -- two LDS
// instructions per half tile, no matrix or vector instruction.  scr: this wave's scratch tile (>= 1 KB); the wave's LDS
// instructions execute in order.  Needs all 64 lanes active (the tile code has no divergent branches).
// Measured (N = 2^19, steady state): backward 0.677 ms with these reads against 0.628 ms with the MFMA transposes -- the
// kernel's LDS pipe (atomics, image reads, the activation transposes) is the busier one.  Kept selectable, off.
#ifndef TNF_REV_LTRANS
#define TNF_REV_LTRANS 0
#endif
__device__ __forceinline__ h4 tr_read16(const _Float16* p) {
    typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    return __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)p));
}
__device__ __forceinline__ T16 ltrans(h4 hi, h4 lo, h4 ident, float* scr, int lane) {
#if TNF_REV_LTRANS
    (void)ident;
    _Float16* t = reinterpret_cast<_Float16*>(scr);  // [hi | lo][16 samples][16 rows]
    const int s = lane & 15, q = lane >> 4;
    *reinterpret_cast<h4*>(t + s * 16 + 4 * q) = hi;
    *reinterpret_cast<h4*>(t + 256 + s * 16 + 4 * q) = lo;
    const _Float16* a = t + (4 * q + (s >> 2)) * 16 + 4 * (s & 3);  // group q, lane s of it: image row 4q + s/4, columns 4 (s%4) ..
    T16 o;
    o.hi = tr_read16(a);
    o.lo = tr_read16(a + 256);
    return o;
#else
    (void)scr;
    (void)lane;
    return mtrans(hi, lo, ident);
#endif
}
// D[rows of a][rows of b] = sum over the 16 samples
__device__ __forceinline__ f4 outer16h(const T16& a, const T16& b) {
    f4 acc = mfma16h(a.hi, b.hi, f4{0.f, 0.f, 0.f, 0.f});
    acc = mfma16h(a.lo, b.hi, acc);
    return mfma16h(a.hi, b.lo, acc);
}
__device__ __forceinline__ f4 outer16h_acc(const T16& a, const T16& b, f4 acc) {
    acc = mfma16h(a.hi, b.hi, acc);
    acc = mfma16h(a.lo, b.hi, acc);
    return mfma16h(a.hi, b.lo, acc);
}
__device__ __forceinline__ f4 rowsum16h_acc(const T16& a, f4 acc) {
    const h4 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    acc = mfma16h(a.hi, ones, acc);
    return mfma16h(a.lo, ones, acc);
}
__device__ __forceinline__ f4 rowsum16h(const T16& a) {
    const h4 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    f4 acc = mfma16h(a.hi, ones, f4{0.f, 0.f, 0.f, 0.f});
    return mfma16h(a.lo, ones, acc);
}
__device__ __forceinline__ f4 mm3(h4 wh, h4 wl, h4 xh, h4 xl, f4 acc) {
    acc = mfma16h(wh, xh, acc);
    acc = mfma16h(wh, xl, acc);
    return mfma16h(wl, xh, acc);
}
// LDS accumulators are 32-bit FIXED POINT (value * fx): on gfx950 ds_add_f32 costs ~190 cycles per
// wave-instruction, ds_add_u32 ~4 (tools/lds_atomic_bench.hip), and a wave's 16-sample tile needs ~60 of them
// per layer.  fx is chosen on the host so that 2^13 per accumulated term cannot overflow (see launch_rev);
// amax tracks the largest term so the kernel can tell when that budget did not hold.
//
// Accumulator layout of one layer (ints), hidden width padded to 16.  A 16 x 16 weight-gradient tile is stored in the
// order its MFMA result sits in the registers: element (row r, column c) at word (c & 3) * 64 + (c >> 2) * 16 + r, i.e.
// register j of lane l = (s, q) (row s, column 4q + j) at word j * 64 + l.  Each of a tile's four ds_add_u32 instructions
// then touches 64 consecutive words: no bank conflicts, no padding, one base address and three immediate offsets.
// (Round 1 stored rows with an odd stride of 17 / H + 1 words: two-way conflicts on most banks, SQ_LDS_BANK_CONFLICT more
// than half of the LDS-active cycles in profiles/r01_g_pmc_flow_bwd.json.)
//   W0 tiles [net][mm] (rows = features 16 mm + r, columns = units) | b0 [net][16] |
//   { Wh tiles [net] (rows = units in, columns = units out) | bh [net][16] } x (L-1) |
//   W2 tiles [net][mo] (rows = units, columns = features 16 mo + c) | b2 [net][H] | fold dA [D] | fold dB [D]
template <int H, int L>
struct AccLayout {
    static constexpr int HT = H / 16, TILE = 256;
    static constexpr int o_w0 = 0;
    static constexpr int o_b0 = o_w0 + 2 * HT * TILE;
    static constexpr int o_h = o_b0 + 32;
    static constexpr int HID = 2 * TILE + 32;
    static constexpr int o_w2 = o_h + (L - 1) * HID;
    static constexpr int o_b2 = o_w2 + 2 * HT * TILE;
    static constexpr int o_fold = o_b2 + 2 * H;
    static constexpr int INTS = (o_fold + 4 * H + 3) & ~3;
    __host__ __device__ static constexpr int word(int r, int c) { return (c & 3) * 64 + (c >> 2) * 16 + r; }
    __host__ __device__ static constexpr int w0(int net, int f, int u) { return o_w0 + (net * HT + (f >> 4)) * TILE + word(f & 15, u); }
    __host__ __device__ static constexpr int wh(int l, int net, int k, int u) { return o_h + l * HID + net * TILE + word(k, u); }
    __host__ __device__ static constexpr int w2(int net, int k, int f) { return o_w2 + (net * HT + (f >> 4)) * TILE + word(k, f & 15); }
};

struct FxAcc {
    float fx;    // scale
    float amax;  // largest |term * fx| seen by this lane
};
// max(m, |a|, |b|) in one instruction.  a and b must be results of ordinary VALU instructions: the compiler does
// not see inside the asm, so it would not insert the wait states an MFMA result needs before a VALU reads it.
// Written IN PLACE ("+v") like every asm VALU result here: the inline-asm rule of f16_tile.h.
__device__ __forceinline__ float amax3(float m, float a, float b) {
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ int fx_cvt(float scaled) {
    asm("v_cvt_rpi_i32_f32 %0, %0" : "+v"(scaled));  // floor(x + 0.5), saturating; the int replaces the float's bits
    return __builtin_bit_cast(int, scaled);
}
// one tile (AccLayout): p = tile + lane; all four values of every lane are real (padded entries receive exact zeros)
__device__ __forceinline__ void lds_add4(int* p, f4 v, FxAcc& fa) {
    const float t0 = v[0] * fa.fx, t1 = v[1] * fa.fx, t2 = v[2] * fa.fx, t3 = v[3] * fa.fx;
    fa.amax = amax3(fa.amax, t0, t1);
    fa.amax = amax3(fa.amax, t2, t3);
    atomicAdd(p + 0, fx_cvt(t0));
    atomicAdd(p + 64, fx_cvt(t1));
    atomicAdd(p + 128, fx_cvt(t2));
    atomicAdd(p + 192, fx_cvt(t3));
}
// a row-sum tile holds the same four sums (rows 4q + j) in every sample lane: lanes s < 4 add row 4q + s
__device__ __forceinline__ void lds_add_rows(int* p4q, f4 v, int s, FxAcc& fa) {
    const float t = ((s & 2) ? ((s & 1) ? v[3] : v[2]) : ((s & 1) ? v[1] : v[0])) * fa.fx;
    fa.amax = fmaxf(fa.amax, fabsf(t));
    if (s < 4) atomicAdd(p4q + s, fx_cvt(t));
}

struct FlowBwdArgs {
    const float* z0;     // (M, N, D): output of the inverse pass
    const float* g_lp;   // (M, N)
    const float* rimg;   // (Mp, 2S, RevImage::FLOATS)
    const unsigned* gmax;  // bits of max |g_lp|
    float* g_z;          // (M, N, D) or NULL
    float* g_params;     // (Mp, gpstride), accumulated
    float* g_fold;       // (Mp, 2S, 2, D), accumulated (odd layers only)
    float* glp_sum;      // (Mp), accumulated
    int64_t M, Mp, N, gpstride, stage, low_off;
    int S, U;
    float fx;            // fixed-point scale of the LDS accumulators
    // deterministic cross-workgroup reduction: every workgroup stores its fixed-point accumulators (plain stores) as
    // row `block` of partials (nblocks, prow) and its sum of g_log_prob into glp_part[block]; flow_bwd_reduce_kernel
    // adds the rows in block order.  NULL: a workgroup owns its gradient row alone and writes it directly.
    int* partials;
    float* glp_part;
    int* overflow;       // set to 1 when a fixed-point accumulator may have wrapped (the gradient rows are NaN then)
};

// One coupling layer backwards on one tile.  x: conditioner half (= layer input and output);
// y: in = transformed half of the OUTPUT, out = of the INPUT;  gx, gy: in = gradients wrt the layer's
// outputs, out = wrt its inputs;  gl = d loss / d (sum of s) for this sample (natural log units).
// index inside AccLayout of element k of the layer's parameter block (bijectors.py:222-235: per MLP layer
// [W_t | W_s | b_t | b_s], W[in][out]); with SPARE the biases behind a tanh live in row 15 of the weight tiles
template <int H, int L, bool SPARE>
__device__ __forceinline__ int acc_src(int kk, int U) {
    typedef AccLayout<H, L> A_;
    if (kk < 2 * H * U + 2 * U) {
        if (kk < 2 * H * U) return A_::w0(kk / (H * U), (kk / U) % H, kk % U);
        return A_::o_b0 + ((kk - 2 * H * U) / U) * 16 + (kk - 2 * H * U) % U;
    }
    kk -= 2 * H * U + 2 * U;
    const int hs = 2 * U * U + 2 * U;
    if (kk < (L - 1) * hs) {
        const int l = kk / hs, r = kk - l * hs;
        if (r < 2 * U * U) {
            const int net = r / (U * U), rr = r - net * U * U;
            return A_::wh(l, net, rr / U, rr % U);
        }
        if (SPARE) return A_::wh(l, (r - 2 * U * U) / U, 15, (r - 2 * U * U) % U);
        return A_::o_h + l * A_::HID + 2 * A_::TILE + ((r - 2 * U * U) / U) * 16 + (r - 2 * U * U) % U;
    }
    kk -= (L - 1) * hs;
    if (kk < 2 * U * H) {
        const int net = kk / (U * H), rr = kk - net * U * H;
        return A_::w2(net, rr / H, rr % H);
    }
    if (SPARE) return A_::w2((kk - 2 * U * H) / H, 15, (kk - 2 * U * H) % H);
    return A_::o_b2 + (kk - 2 * U * H);
}

// Where a tile's weight-gradient contributions go.  LdsFxAccum: the whole-flow kernel (a wave meets every layer, so
// the accumulators are shared fixed-point LDS words).  RegAccum: a kernel that stays on one layer keeps them in
// registers and lets the MFMAs accumulate.
template <int H, int L>
struct LdsFxAccum {
    typedef AccLayout<H, L> A_;
    int* acc;
    FxAcc& fa;
    int s, q;
    __device__ __forceinline__ void w2(int net, int mo, const T16& d_t, const T16& h_t) {
        lds_add4(acc + A_::o_w2 + (net * A_::HT + mo) * A_::TILE + 16 * q + s, outer16h(d_t, h_t), fa);
    }
    __device__ __forceinline__ void b2(int net, int mo, const T16& d_t) {
        lds_add_rows(acc + A_::o_b2 + net * H + 16 * mo + 4 * q, rowsum16h(d_t), s, fa);
    }
    __device__ __forceinline__ void wh(int l, int net, const T16& d_t, const T16& h_t) {
        lds_add4(acc + A_::o_h + l * A_::HID + net * A_::TILE + 16 * q + s, outer16h(d_t, h_t), fa);
    }
    __device__ __forceinline__ void bh(int l, int net, const T16& d_t) {
        lds_add_rows(acc + A_::o_h + l * A_::HID + 2 * A_::TILE + net * 16 + 4 * q, rowsum16h(d_t), s, fa);
    }
    __device__ __forceinline__ void b0(int net, const T16& d_t) {
        lds_add_rows(acc + A_::o_b0 + net * 16 + 4 * q, rowsum16h(d_t), s, fa);
    }
    __device__ __forceinline__ void w0(int net, int mm, const T16& d_t, const T16& x_t) {
        lds_add4(acc + A_::o_w0 + (net * A_::HT + mm) * A_::TILE + 16 * q + s, outer16h(d_t, x_t), fa);
    }
};

template <int H, int L>
struct RegAccum {
    static constexpr int HT = (H + 15) / 16;
    static constexpr int LH = (L > 1) ? (L - 1) : 1;
    f4 W0[2][HT], Wh[LH][2], W2[2][HT], B0[2], Bh[LH][2], B2[2][HT];
    __device__ __forceinline__ void clear() {
        const f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            B0[net] = z;
#pragma unroll
            for (int t = 0; t < HT; ++t) W0[net][t] = W2[net][t] = B2[net][t] = z;
#pragma unroll
            for (int l = 0; l < LH; ++l) Wh[l][net] = Bh[l][net] = z;
        }
    }
    __device__ __forceinline__ void w2(int net, int mo, const T16& d_t, const T16& h_t) { W2[net][mo] = outer16h_acc(d_t, h_t, W2[net][mo]); }
    __device__ __forceinline__ void b2(int net, int mo, const T16& d_t) { B2[net][mo] = rowsum16h_acc(d_t, B2[net][mo]); }
    __device__ __forceinline__ void wh(int l, int net, const T16& d_t, const T16& h_t) { Wh[l][net] = outer16h_acc(d_t, h_t, Wh[l][net]); }
    __device__ __forceinline__ void bh(int l, int net, const T16& d_t) { Bh[l][net] = rowsum16h_acc(d_t, Bh[l][net]); }
    __device__ __forceinline__ void b0(int net, const T16& d_t) { B0[net] = rowsum16h_acc(d_t, B0[net]); }
    __device__ __forceinline__ void w0(int net, int mm, const T16& d_t, const T16& x_t) { W0[net][mm] = outer16h_acc(d_t, x_t, W0[net][mm]); }
};

// SPARE (num_units <= 15): hidden unit 15 is padding, so the transposed activation operand carries a row of
// ones there and the bias gradients of the layers behind a tanh arrive as column 15 of the weight-gradient tiles.
// MODE 0: the inverse-pass layer of the whole-flow kernel (y comes in as the layer's OUTPUT and leaves as its
// input).  MODE 1: a forward-direction layer y' = t + y e^s with its INPUT saved (y stays); kc, when not NULL,
// points at this lane's [k0 | k1] constants (k1 at kc + D) of the batch-moment correction for the transformed
// half: the upstream gradient is g + k0 + k1 y'.  MODE 2: an inverse-direction layer y' = (y - t) e^-s with its
// INPUT saved; kmask != 0 with kc == NULL means "finalize": the upstream gradient of the transformed half is
// gl_fin * y' (the base density's -g_log_prob y'), passed in kmask.
template <int H, int L, bool SPARE, int MODE, class ACCP>
__device__ __forceinline__ void layer_bwd16(const float* img, ACCP& accp, float* scrA, float* scrB, int lane, int U,
                                            const f4 (&x)[(H + 15) / 16], f4 (&y)[(H + 15) / 16],
                                            f4 (&gx)[(H + 15) / 16], f4 (&gy)[(H + 15) / 16], float gl,
                                            const float* kc = nullptr, float kmask = 1.f) {
    typedef F16Image<H, L> FImg;
    typedef B16Image<H, L> BImg;
    typedef RevImage<H, L> R;
    constexpr int HT = FImg::HT;
    const int s = lane & 15, q = lane >> 4;
    const u4* fg = reinterpret_cast<const u4*>(img + R::F_OFF) + lane;
    const u4* bg = reinterpret_cast<const u4*>(img + R::B_OFF) + lane;
    const float* bl = img + R::F_OFF + FImg::NWG * 256 + q * 4;
    auto bias = [&](int g) -> f4 { return *reinterpret_cast<const f4*>(bl + g * 16); };
    auto hl = [&](const u4* base, int g, h4& hi, h4& lo) {
        const u4 wv = base[g * 64];
        hi = __builtin_bit_cast(h4, u2{wv[0], wv[1]});
        lo = __builtin_bit_cast(h4, u2{wv[2], wv[3]});
    };
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    h4 ident;  // B operand of the identity: lane (n = s, q) holds K = 4q + i
#pragma unroll
    for (int i = 0; i < 4; ++i) ident[i] = (4 * q + i == s) ? (_Float16)1.f : (_Float16)0.f;

    // ---- 1. forward recompute (h = tanh = 1 - 2r is what the backward needs: tanh' = 1 - h^2) ----
    f4 r[L][2], h[L][2];
    h4 xs_hi[HT], xs_lo[HT];  // the split conditioner input, reused for its transposed form
    if constexpr (H == 32) {
        u4 a_, b_;
        split2(x[0][0], x[0][1], a_[0], b_[0]);
        split2(x[0][2], x[0][3], a_[1], b_[1]);
        split2(x[1][0], x[1][1], a_[2], b_[2]);
        split2(x[1][2], x[1][3], a_[3], b_[3]);
        const h8 xh = __builtin_bit_cast(h8, a_), xl = __builtin_bit_cast(h8, b_);
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            xs_hi[mm] = __builtin_bit_cast(h4, u2{a_[2 * mm], a_[2 * mm + 1]});
            xs_lo[mm] = __builtin_bit_cast(h4, u2{b_[2 * mm], b_[2 * mm + 1]});
        }
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            const h8 wh = __builtin_bit_cast(h8, fg[FImg::g_w0(net, 0) * 64]);
            const h8 wl = __builtin_bit_cast(h8, fg[FImg::g_w0(net, 1) * 64]);
            f4 a0 = mfma32h(wh, xh, bias(FImg::b_b0(net)));
            a0 = mfma32h(wh, xl, a0);
            a0 = mfma32h(wl, xh, a0);
            r[0][net] = sig2_4(a0);
        }
    } else {
        h4 xh, xl;
        split4(x[0], xh, xl);
        xs_hi[0] = xh;
        xs_lo[0] = xl;
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            h4 wh, wl;
            hl(fg, FImg::g_w0(net, 0), wh, wl);
            r[0][net] = sig2_4(mm3(wh, wl, xh, xl, bias(FImg::b_b0(net))));
        }
    }
#pragma unroll
    for (int l = 0; l < L - 1; ++l)
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            h4 rh, rl, wh, wl;
            split4(r[l][net], rh, rl);
            hl(fg, FImg::g_wh(l, net), wh, wl);
            r[l + 1][net] = sig2_4(mm3(wh, wl, rh, rl, bias(FImg::b_bh(l, net))));
        }
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int net = 0; net < 2; ++net)
#pragma unroll
            for (int j = 0; j < 4; ++j) h[l][net][j] = __builtin_fmaf(-2.f, r[l][net][j], 1.f);
    auto with_ones = [&](f4 v) -> f4 {  // unit 15 lives in lane group q = 3, register 3
        if (SPARE) v[3] = (q == 3) ? 1.f : v[3];
        return v;
    };
    f4 dout[2][HT];
    {
        h4 rh[2], rl[2];
        split4(r[L - 1][0], rh[0], rl[0]);
        split4(r[L - 1][1], rh[1], rl[1]);
#pragma unroll
        for (int mo = 0; mo < HT; ++mo) {
            h4 wh, wl;
            hl(fg, FImg::g_w2(0, mo), wh, wl);
            const f4 tt = mm3(wh, wl, rh[0], rl[0], bias(FImg::b_b2(0, mo)));
            hl(fg, FImg::g_w2(1, mo), wh, wl);
            const f4 sv = mm3(wh, wl, rh[1], rl[1], bias(FImg::b_b2(1, mo)));
            // ---- 2. rebuild the input, output deltas ----
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = __builtin_amdgcn_exp2f(sv[j]);
                if (MODE == 2) {
                    const float em = __builtin_amdgcn_exp2f(-sv[j]);
                    const float yo = (y[mo][j] - tt[j]) * em;
                    const float g = gy[mo][j] + kmask * yo;
                    const float dy = g * em;
                    dout[0][mo][j] = -dy;
                    dout[1][mo][j] = __builtin_fmaf(-g, yo, gl);
                    gy[mo][j] = dy;
                } else if (MODE == 1) {
                    float g = gy[mo][j];
                    if (kc) g += kmask * __builtin_fmaf(kc[2 * H + 16 * mo + j], __builtin_fmaf(y[mo][j], e, tt[j]), kc[16 * mo + j]);
                    const float dy = g * e;
                    dout[0][mo][j] = g;
                    dout[1][mo][j] = __builtin_fmaf(g * y[mo][j], e, gl);
                    gy[mo][j] = dy;
                } else {
                    const float em = __builtin_amdgcn_exp2f(-sv[j]);
                    const float g = gy[mo][j], yo = y[mo][j];
                    const float dy = g * em;
                    dout[0][mo][j] = -dy;
                    dout[1][mo][j] = __builtin_fmaf(-g, yo, gl);
                    y[mo][j] = __builtin_fmaf(yo, e, tt[j]);
                    gy[mo][j] = dy;
                }
            }
        }
    }

    // ---- 3. output layer: dW2, db2, d h_{L-1} ----
    f4 dh[2];
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        const T16 h_t = tsplit(with_ones(h[L - 1][net]), scrB, lane);
        h4 dsh[HT], dsl[HT];
#pragma unroll
        for (int mo = 0; mo < HT; ++mo) {
            split4(dout[net][mo], dsh[mo], dsl[mo]);
            const T16 d_t = ltrans(dsh[mo], dsl[mo], ident, scrA, lane);
            accp.w2(net, mo, d_t, h_t);  // [o = 16 mo + 4q + j][k = s]
            if (!SPARE) accp.b2(net, mo, d_t);
        }
        if constexpr (H == 32) {
            const u2 h0_ = __builtin_bit_cast(u2, dsh[0]), h1_ = __builtin_bit_cast(u2, dsh[1]);
            const u2 l0_ = __builtin_bit_cast(u2, dsl[0]), l1_ = __builtin_bit_cast(u2, dsl[1]);
            const h8 dhi = __builtin_bit_cast(h8, u4{h0_[0], h0_[1], h1_[0], h1_[1]});
            const h8 dlo = __builtin_bit_cast(h8, u4{l0_[0], l0_[1], l1_[0], l1_[1]});
            const h8 wh = __builtin_bit_cast(h8, bg[BImg::g_w2(net, 0) * 64]);
            const h8 wl = __builtin_bit_cast(h8, bg[BImg::g_w2(net, 1) * 64]);
            f4 a0 = mfma32h(wh, dhi, zero);
            a0 = mfma32h(wh, dlo, a0);
            dh[net] = mfma32h(wl, dhi, a0);
        } else {
            h4 wh, wl;
            hl(bg, BImg::g_w2(net, 0), wh, wl);
            dh[net] = mm3(wh, wl, dsh[0], dsl[0], zero);
        }
    }
    // ---- 4. hidden layers, last to first ----
#pragma unroll
    for (int l = L - 2; l >= 0; --l)
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            f4 da;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float hh = h[l + 1][net][j];
                da[j] = dh[net][j] * __builtin_fmaf(-hh, hh, 1.f);
            }
            h4 dhi, dlo;
            split4(da, dhi, dlo);
            const T16 d_t = ltrans(dhi, dlo, ident, scrA, lane);
            const T16 h_t = tsplit(with_ones(h[l][net]), scrB, lane);
            accp.wh(l, net, d_t, h_t);  // [k_out = 4q + j][k_in = s]
            if (!SPARE) accp.bh(l, net, d_t);
            h4 wh, wl;
            hl(bg, BImg::g_wh(l, net), wh, wl);
            dh[net] = mm3(wh, wl, dhi, dlo, zero);
        }
    // ---- 5. first layer: dW0, db0, d x ----
    T16 x_t[HT];
#pragma unroll
    for (int mm = 0; mm < HT; ++mm) x_t[mm] = ltrans(xs_hi[mm], xs_lo[mm], ident, scrA, lane);
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        f4 da;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float hh = h[0][net][j];
            da[j] = dh[net][j] * __builtin_fmaf(-hh, hh, 1.f);
        }
        h4 dhi, dlo;
        split4(da, dhi, dlo);
        const T16 d_t = ltrans(dhi, dlo, ident, scrA, lane);
        accp.b0(net, d_t);
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            accp.w0(net, mm, d_t, x_t[mm]);  // [u = 4q + j][f = 16 mm + s]
            h4 wh, wl;
            hl(bg, BImg::g_w0(net, mm), wh, wl);
            gx[mm] = mm3(wh, wl, dhi, dlo, gx[mm]);
        }
    }
}

// Undo the fold in front of a layer on one half (features f0 + 16 mm + 4q + j): v <- (v - B)/A, g <- g A;
// with AFFINE the tile's contributions to dA = sum g v, dB = sum g go to gf: the two tiles are transposed
// through the scratch so that a lane holds four samples of one feature, summed in the lane, and the four
// lanes that share a feature add to the same accumulator word.
template <int H, bool AFFINE>
__device__ __forceinline__ void unfold_half(const float* fc, int* gf, FxAcc& fa, float* scrA, float* scrB, int lane,
                                            int f0, f4 (&v)[(H + 15) / 16], f4 (&g)[(H + 15) / 16]) {
    constexpr int D = 2 * H;
    constexpr int HT = (H + 15) / 16;
    const int s = lane & 15, q = lane >> 4;
#pragma unroll
    for (int mm = 0; mm < HT; ++mm) {
        const int f = f0 + 16 * mm + 4 * q;
        const f4 A = *reinterpret_cast<const f4*>(fc + f);
        const f4 iA = *reinterpret_cast<const f4*>(fc + 2 * D + f);
        const f4 C = *reinterpret_cast<const f4*>(fc + 3 * D + f);
        f4 vp, gv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vp[j] = __builtin_fmaf(v[mm][j], iA[j], C[j]);
            gv[j] = g[mm][j] * vp[j];
        }
        if (AFFINE) {
            const f4 ta = transpose16(gv, scrA, lane), tb = transpose16(g[mm], scrB, lane);
            const float da = ((ta[0] + ta[1]) + (ta[2] + ta[3])) * fa.fx, db = ((tb[0] + tb[1]) + (tb[2] + tb[3])) * fa.fx;
            fa.amax = amax3(fa.amax, da, db);
            atomicAdd(gf + f0 + 16 * mm + s, fx_cvt(da));       // feature = row s of the transposed tile
            atomicAdd(gf + D + f0 + 16 * mm + s, fx_cvt(db));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g[mm][j] *= A[j];
            v[mm][j] = vp[j];
        }
    }
}

// Second pass of the deterministic reduction: gradient row mp = sum over its nred workgroups' fixed-point rows, in
// block order, in 64-bit integers (exact, so the order would not even matter), scaled back once.  A wrapped
// accumulator anywhere (*overflow) poisons the rows with NaN instead: never a wrong finite gradient.
__global__ void __launch_bounds__(256)
flow_bwd_reduce_kernel(const int* __restrict__ partials, const float* __restrict__ glp_part, const unsigned* __restrict__ gmax,
                       const int* __restrict__ overflow, float* __restrict__ g_params, float* __restrict__ g_fold,
                       float* __restrict__ glp_sum, int64_t nred, int nl, int P, int D, int64_t gpstride, int64_t stage,
                       int64_t low_off, float fx, int64_t Mp, int tanh_pair, int U, int L) {
    const int64_t mp = grid_m();  // parameter rows ride on grid y and z (grid_xm): any Mp
    if (mp >= Mp) return;
    const int64_t prow = (int64_t)nl * (P + 2 * D);
    float isc = 1.f;
    {
        const float gm = __uint_as_float(*gmax);
        if (gm > 0.f && gm < 3.0e38f) {
            int e;
            (void)frexpf(gm, &e);
            int k = 1 - e;
            k = k > 120 ? 120 : (k < -120 ? -120 : k);
            isc = ldexpf(1.f, -k);
        }
    }
    const float unfx = isc / fx;
    const float poison = *overflow ? __builtin_nanf("") : 0.f;
    const int* src = partials + mp * nred * prow;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < prow; i += (int64_t)gridDim.x * 256) {
        long long acc = 0;
        int64_t b = 0;
        for (; b + 8 <= nred; b += 8) {  // eight rows in flight
            int v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(b + u) * prow + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (long long)v[u];
        }
        for (; b < nred; ++b) acc += (long long)src[b * prow + i];
        const int c = (int)(i / (P + 2 * D));
        const int k = (int)(i - (int64_t)c * (P + 2 * D));
        if (tanh_pair && k < P) {
            // the rows of flow_bwd_pair_kernel: a weight behind a tanh holds G = sum_s r_k d_o; its gradient is
            // db_o - 2 G with db_o the bias entry of the same MLP layer and net (flow_bwd_pair.h)
            const int H = D / 2;
            int kk = k - (2 * H * U + 2 * U), bias = -1;
            if (kk >= 0) {
                const int hs = 2 * U * U + 2 * U;
                if (kk < (L - 1) * hs) {
                    const int l = kk / hs, r = kk - l * hs;
                    if (r < 2 * U * U) bias = 2 * H * U + 2 * U + l * hs + 2 * U * U + (r / (U * U)) * U + r % U;
                } else {
                    kk -= (L - 1) * hs;
                    if (kk < 2 * U * H) bias = 2 * H * U + 2 * U + (L - 1) * hs + 2 * U * H + (kk / (U * H)) * H + kk % H;
                }
            }
            if (bias >= 0) {
                long long db = 0;
                const int* bsrc = src + (int64_t)c * (P + 2 * D) + bias;
                int64_t bb = 0;
                for (; bb + 8 <= nred; bb += 8) {
                    int v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = bsrc[(bb + u) * prow];
#pragma unroll
                    for (int u = 0; u < 8; ++u) db += (long long)v[u];
                }
                for (; bb < nred; ++bb) db += (long long)bsrc[bb * prow];
                acc = db - 2 * acc;
            }
        }
        const float v = (float)acc * unfx + poison;
        if (k < P) g_params[mp * gpstride + (c >> 1) * stage + ((c & 1) ? low_off : 0) + k] += v;
        else if (c & 1) g_fold[(mp * nl + c) * 2 * D + (k - P)] += v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float t = 0.f;
        for (int64_t bb = 0; bb < nred; ++bb) t += glp_part[mp * nred + bb];
        glp_sum[mp] += t;
    }
}

template <int H, int L, int NW, bool SPARE>
__global__ void __launch_bounds__(NW * 64)
flow_bwd_f16_kernel(FlowBwdArgs a) {
    typedef RevImage<H, L> R;
    constexpr int D = 2 * H;
    constexpr int HT = (H + 15) / 16;
    constexpr int RU4 = R::FLOATS / 4;
    constexpr int SLOT = (R::FLOATS + 255) & ~255;  // ring slot: whole 1-KB LDS-DMA pieces
    constexpr int NPIECE = SLOT / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nl = 2 * a.S;
    const int U = a.U;
    typedef AccLayout<H, L> A_;
    constexpr int ACC = A_::INTS;
    float* ring = lds;                          // [2][SLOT]
    int* accb = reinterpret_cast<int*>(lds + 2 * SLOT);  // [nl][ACC] fixed point
    float* scr = lds + 2 * SLOT + nl * ACC;              // [NW][kRevNScr][kScr]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const int64_t m = grid_m();
    if (m >= a.M) return;
    const int64_t mp = a.Mp == 1 ? 0 : m;
    float* scrA = scr + wave * kRevNScr * kScr;
    float* scrB = scrA + (kRevNScr - 1) * kScr;
    const u4* isrc = reinterpret_cast<const u4*>(a.rimg + mp * (int64_t)nl * R::FLOATS);

    // layer image c -> ring slot by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave-instruction, no staging registers --
    // round 2 prefetched the next image through 28 VGPRs per lane); a piece's tail beyond the image re-reads its last
    // 16 bytes into the slot's padding
    typedef __attribute__((address_space(3))) void lds_void_;
    auto fetch = [&](int c, float* slot) {
        const u4* src = isrc + (int64_t)c * RU4;
        for (int i = wave; i < NPIECE; i += NW) {
            const int idx = i * 64 + lane;
            __builtin_amdgcn_global_load_lds(src + (idx < RU4 ? idx : RU4 - 1), (lds_void_*)(slot + i * 256), 16, 0, 0);
        }
    };
    fetch(0, ring);
    for (int i = threadIdx.x; i < nl * ACC; i += NW * 64) accb[i] = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // power-of-two scale that brings max |g_lp| into [1, 2)
    float sc = 1.f, isc = 1.f;
    {
        const float gm = __uint_as_float(*a.gmax);
        if (gm > 0.f && gm < 3.0e38f) {
            int e;
            (void)frexpf(gm, &e);  // gm = f 2^e, f in [0.5, 1)
            int k = 1 - e;
            k = k > 120 ? 120 : (k < -120 ? -120 : k);
            sc = ldexpf(1.f, k);
            isc = ldexpf(1.f, -k);
        }
    }
    __syncthreads();

    const int64_t ntiles = (a.N + 15) >> 4;
    const int64_t stride = (int64_t)gridDim.x * NW;
    const int64_t iters = (ntiles + stride - 1) / stride;
    const float* zb = a.z0 + m * a.N * D;
    const float* glb = a.g_lp + m * a.N;
    float* gzb = a.g_z ? a.g_z + m * a.N * D : nullptr;
    float glp_acc = 0.f;
    FxAcc fa{a.fx, 0.f};
    int step = 0;
    const int64_t nsteps = iters * nl;

    for (int64_t it = 0; it < iters; ++it) {
        const int64_t tile = (it * gridDim.x + blockIdx.x) * NW + wave;
        const int64_t row = tile * 16 + s;
        const bool row_ok = row < a.N;
        const int64_t rowc = row_ok ? row : a.N - 1;
        f4 lo[HT], hi[HT], glo[HT], ghi[HT];
        const float glp = row_ok ? sc * glb[rowc] : 0.f;
        if (q == 0) glp_acc += glp;
        {
            const float* zr = zb + rowc * D + 4 * q;
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                lo[mm] = *reinterpret_cast<const f4*>(zr + 16 * mm);
                hi[mm] = *reinterpret_cast<const f4*>(zr + H + 16 * mm);
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // d (-|z0|^2 / 2) / d z0, times g_log_prob
                    glo[mm][j] = -glp * lo[mm][j];
                    ghi[mm][j] = -glp * hi[mm][j];
                }
            }
        }
        const float gl = -glp;  // log_prob = base - sum of the layers' log-dets

        for (int c = 0; c < nl; ++c, ++step) {
            const float* img = ring + (step & 1) * SLOT;
            const bool more = (int64_t)step + 1 < nsteps;
            if (more) fetch((c + 1 == nl) ? 0 : c + 1, ring + ((step + 1) & 1) * SLOT);
            int* acc = accb + c * ACC;
            const float* fc = img + R::C_OFF;
            if ((c & 1) == 0) {  // RealNVP(upper): conditioner = low half
                LdsFxAccum<H, L> ap{acc, fa, s, q};
                layer_bwd16<H, L, SPARE, 0>(img, ap, scrA, scrB, lane, U, lo, hi, glo, ghi, gl);
                unfold_half<H, false>(fc, acc + A_::o_fold, fa, scrA, scrB, lane, 0, lo, glo);
                unfold_half<H, false>(fc, acc + A_::o_fold, fa, scrA, scrB, lane, H, hi, ghi);
            } else {             // RealNVP(lower) behind BatchNorm + Affine
                LdsFxAccum<H, L> ap{acc, fa, s, q};
                layer_bwd16<H, L, SPARE, 0>(img, ap, scrA, scrB, lane, U, hi, lo, ghi, glo, gl);
                unfold_half<H, true>(fc, acc + A_::o_fold, fa, scrA, scrB, lane, 0, lo, glo);
                unfold_half<H, true>(fc, acc + A_::o_fold, fa, scrA, scrB, lane, H, hi, ghi);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next image have landed
            __syncthreads();
        }
        if (gzb && row_ok) {
            float* gr = gzb + row * D + 4 * q;
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                f4 a0, a1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a0[j] = glo[mm][j] * isc;
                    a1[j] = ghi[mm][j] * isc;
                }
                *reinterpret_cast<f4*>(gr + 16 * mm) = a0;
                *reinterpret_cast<f4*>(gr + H + 16 * mm) = a1;
            }
        }
    }

    // ---- flush ----
    // a term above the fixed-point budget may have wrapped an accumulator: flag it (the reduction then poisons the
    // result instead of returning a wrong gradient; the budget is 2^13 per term in units where max |g_log_prob| is 1..2)
    float amax = fa.amax;
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    float* red = scr;  // the transposition scratch is free now
    __syncthreads();
    if (lane == 0) red[wave] = amax;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) amax = fmaxf(amax, red[w]);
    const bool wrapped = !(amax * (float)(iters * NW) < 2147483648.f);  // (also true for NaN / inf terms)
    if (wrapped && threadIdx.x == 0) atomicOr(a.overflow, 1);
    const int P = 2 * (H * U + U) + (L - 1) * 2 * (U * U + U) + 2 * (U * H + H);
    float tot = glp_acc;
    tot += __shfl_xor(tot, 1);
    tot += __shfl_xor(tot, 2);
    tot += __shfl_xor(tot, 4);
    tot += __shfl_xor(tot, 8);  // lanes 0..15 (q = 0) carried the terms
    __syncthreads();
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    {
        // this workgroup's fixed-point contribution, in the order of the gradient row: layer c -> [parameter block (P) |
        // fold (2 D)]; flow_bwd_reduce_kernel adds the rows up
        const int64_t nred = (a.Mp == 1 ? a.M : 1) * gridDim.x;
        const int64_t blk = (a.Mp == 1 ? m : 0) * gridDim.x + blockIdx.x;
        const int64_t prow = (int64_t)nl * (P + 2 * D);
        int* dst = a.partials + (mp * nred + blk) * prow;
        for (int i = threadIdx.x; i < nl * (P + 2 * D); i += NW * 64) {
            const int c = i / (P + 2 * D);
            const int k = i - c * (P + 2 * D);
            const int* acc = accb + c * ACC;
            dst[i] = (k >= P) ? ((c & 1) ? acc[A_::o_fold + (k - P)] : 0) : acc[acc_src<H, L, SPARE>(k, U)];
        }
        if (threadIdx.x == 0) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[w];
            a.glp_part[mp * nred + blk] = t * isc;
        }
    }
}

}  // namespace tnf
#include "flow_bwd_pair.h"
namespace tnf {

// ---------------------------------------------------------------------------
// One forward-direction coupling layer backwards with its INPUT saved (the training-mode chain of coupling_mfma.hip:
// sampling with fresh batch statistics under autograd).  Same tile code as above, split-f16 contractions; a wave
// stays on the layer, so the weight gradients accumulate in registers inside the MFMAs and leave once.
//   BwdArgs as for coupling_bwd_mfma_kernel: fold (A | B applied to the saved input), g_fold (dA | dB sums),
//   gcorr (k0 | k1: upstream gradient g + k0 + k1 * output), ld_scale * g_ld on the sum of s.
// ---------------------------------------------------------------------------
constexpr int kLayerNW = 8;

template <int H, int L, bool SPARE, bool INV>
__global__ void __launch_bounds__(kLayerNW * 64)
coupling_bwd_f16_kernel(BwdArgs a) {
    if (a.gate && *a.gate == 0) return;  // a conditionally needed launch (tnf_set_launch_gate): nothing to do
    typedef RevImage<H, L> R;
    typedef AccLayout<H, L> A_;
    constexpr int D = 2 * H;
    constexpr int HT = (H + 15) / 16;
    constexpr int NW = kLayerNW;
    static_assert(R::FLOATS >= A_::INTS, "the flush reuses the image area");
    __shared__ __attribute__((aligned(16))) float lds[R::FLOATS + NW * kScr];
    float* img = lds;
    float* cst = lds + R::C_OFF;  // A | B | k0 | k1
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const int64_t m = grid_m();
    if (m >= a.M) return;
    const int64_t mp = a.Mp == 1 ? 0 : m;
    const int U = a.U;
    const float* prow = a.params + mp * a.pstride;
    float* scrA = lds + R::FLOATS + wave * kScr;
    if (wave == 0) build_f16_image<H, L>(img + R::F_OFF, prow, U, lane);
    if (wave == 1) build_b16_image<H, L>(img + R::B_OFF, prow, U, lane);
    const bool has_corr = !INV && a.gcorr != nullptr;
    const bool finalize = INV && a.g_lp != nullptr;  // last layer of a log_prob chain: seeds from the base density
    // the deltas are split into f16 halves: keep them clear of the f16 subnormals whatever the loss scale
    float sc = 1.f, isc = 1.f;
    if (a.gmax) {
        const float gm = __uint_as_float(*a.gmax);
        if (gm > 0.f && gm < 3.0e38f) {
            int e;
            (void)frexpf(gm, &e);
            int k = 1 - e;
            k = k > 120 ? 120 : (k < -120 ? -120 : k);
            sc = ldexpf(1.f, k);
            isc = ldexpf(1.f, -k);
        }
    }
    for (int i = threadIdx.x; i < 2 * D; i += NW * 64) {
        cst[i] = a.fold ? a.fold[mp * a.fold_stride + i] : (i < D ? 1.f : 0.f);
        cst[2 * D + i] = has_corr ? sc * a.gcorr[i] : 0.f;
    }
    __syncthreads();

    const int c_off = a.upper ? 0 : H, t_off = a.upper ? H : 0;
    const float* zb = a.z + m * a.N * D;
    const float* gzo = a.g_zout ? a.g_zout + m * a.N * D : nullptr;
    const float* gld = a.g_ld + m * a.N;
    const float* glpb = finalize ? a.g_lp + m * a.N : nullptr;
    float glp_acc = 0.f;
    float* gzb = a.g_z + m * a.N * D;
    const float* cx = cst + c_off + 4 * q;  // this lane's conditioner / transformed features
    const float* cy = cst + t_off + 4 * q;
    RegAccum<H, L> ra;
    ra.clear();
    f4 dAx[HT], dBx[HT], dAy[HT], dBy[HT];
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mm = 0; mm < HT; ++mm) dAx[mm] = dBx[mm] = dAy[mm] = dBy[mm] = zero;

    const int64_t ntiles = (a.N + 15) >> 4;
    const int64_t tstep = (int64_t)gridDim.x * NW;
    f4 nxs[HT], nys[HT];  // the next tile's saved input, in flight while this tile computes (prefetching g as well
    {                     // gained nothing: its loads already overlap the forward recompute)
        int64_t r0 = ((int64_t)blockIdx.x * NW + wave) * 16 + s;
        if (r0 >= a.N) r0 = a.N - 1;
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            nxs[mm] = *reinterpret_cast<const f4*>(zb + r0 * D + 4 * q + 16 * mm + c_off);
            nys[mm] = *reinterpret_cast<const f4*>(zb + r0 * D + 4 * q + 16 * mm + t_off);
        }
    }
    for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < ntiles; tile += tstep) {
        const int64_t row = tile * 16 + s;
        const bool row_ok = row < a.N;
        const int64_t rowc = row_ok ? row : a.N - 1;
        int64_t nrow = (tile + tstep) * 16 + s;
        if (nrow >= a.N) nrow = a.N - 1;
        f4 xs[HT], ys[HT], x[HT], y[HT], gx[HT], gy[HT];
        const float glp = (finalize && row_ok) ? sc * glpb[rowc] : 0.f;
        if (q == 0) glp_acc += glp;
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            xs[mm] = nxs[mm];
            ys[mm] = nys[mm];
            nxs[mm] = *reinterpret_cast<const f4*>(zb + nrow * D + 4 * q + 16 * mm + c_off);
            nys[mm] = *reinterpret_cast<const f4*>(zb + nrow * D + 4 * q + 16 * mm + t_off);
            gx[mm] = zero;
            gy[mm] = zero;
            if (gzo && row_ok) {
                const float* gr = gzo + rowc * D + 4 * q + 16 * mm;
                gx[mm] = sc * *reinterpret_cast<const f4*>(gr + c_off);
                gy[mm] = sc * *reinterpret_cast<const f4*>(gr + t_off);
            }
            const f4 ax = *reinterpret_cast<const f4*>(cx + 16 * mm), bx = *reinterpret_cast<const f4*>(cx + D + 16 * mm);
            const f4 ay = *reinterpret_cast<const f4*>(cy + 16 * mm), by = *reinterpret_cast<const f4*>(cy + D + 16 * mm);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[mm][j] = __builtin_fmaf(xs[mm][j], ax[j], bx[j]);
                y[mm][j] = __builtin_fmaf(ys[mm][j], ay[j], by[j]);
            }
        }
        const float gl = row_ok ? sc * a.ld_scale * gld[rowc] : 0.f;
        if (INV) {
            if (finalize) {  // d(-|out|^2 / 2)/d out . g_lp: the conditioner half of out is x itself
#pragma unroll
                for (int mm = 0; mm < HT; ++mm)
#pragma unroll
                    for (int j = 0; j < 4; ++j) gx[mm][j] = -glp * x[mm][j];
            }
            layer_bwd16<H, L, SPARE, 2>(img, ra, scrA, scrA, lane, U, x, y, gx, gy, gl, nullptr, finalize ? -glp : 0.f);
        } else {
            layer_bwd16<H, L, SPARE, 1>(img, ra, scrA, scrA, lane, U, x, y, gx, gy, gl, has_corr ? cy + 2 * D : nullptr,
                                        row_ok ? 1.f : 0.f);
        }
        if (has_corr && row_ok) {  // the conditioner half of the output is x itself; added here (the layer code is linear
                                   // in gx) so that the g loads stay in flight during the forward recompute
#pragma unroll
            for (int mm = 0; mm < HT; ++mm) {
                const f4 k0 = *reinterpret_cast<const f4*>(cx + 2 * D + 16 * mm), k1 = *reinterpret_cast<const f4*>(cx + 3 * D + 16 * mm);
#pragma unroll
                for (int j = 0; j < 4; ++j) gx[mm][j] += __builtin_fmaf(k1[j], x[mm][j], k0[j]);
            }
        }
        // back through the fold: g w.r.t. the saved input, and the fold-constant sums
#pragma unroll
        for (int mm = 0; mm < HT; ++mm) {
            const f4 ax = *reinterpret_cast<const f4*>(cx + 16 * mm), ay = *reinterpret_cast<const f4*>(cy + 16 * mm);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dAx[mm][j] = __builtin_fmaf(gx[mm][j], xs[mm][j], dAx[mm][j]);
                dBx[mm][j] += gx[mm][j];
                dAy[mm][j] = __builtin_fmaf(gy[mm][j], ys[mm][j], dAy[mm][j]);
                dBy[mm][j] += gy[mm][j];
                gx[mm][j] *= ax[j] * isc;
                gy[mm][j] *= ay[j] * isc;
            }
            if (row_ok) {
                float* gr = gzb + row * D + 4 * q + 16 * mm;
                *reinterpret_cast<f4*>(gr + c_off) = gx[mm];
                *reinterpret_cast<f4*>(gr + t_off) = gy[mm];
            }
        }
    }

    // ---- flush: register accumulators -> LDS (waves in turn, plain adds) -> one global atomic per parameter ----
    __syncthreads();
    float* gacc = lds;  // AccLayout, floats
    for (int i = threadIdx.x; i < A_::INTS; i += NW * 64) gacc[i] = 0.f;
    __syncthreads();
    auto add4 = [&](float* p, f4 v) {
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] += v[j];
    };
    auto add_tile = [&](float* tile, f4 v) {  // AccLayout tile: register j of this lane at word j * 64 + lane
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[j * 64 + lane] += v[j];
    };
    auto red16 = [&](float v) -> float {
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        return v;
    };
    for (int turn = 0; turn < NW; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int net = 0; net < 2; ++net) {
#pragma unroll
                for (int mm = 0; mm < HT; ++mm) {
                    add_tile(gacc + A_::o_w0 + (net * HT + mm) * A_::TILE, ra.W0[net][mm]);
                    add_tile(gacc + A_::o_w2 + (net * HT + mm) * A_::TILE, ra.W2[net][mm]);
                    if (!SPARE && s == 0) add4(gacc + A_::o_b2 + net * H + 16 * mm + 4 * q, ra.B2[net][mm]);
                }
#pragma unroll
                for (int l = 0; l < L - 1; ++l) {
                    add_tile(gacc + A_::o_h + l * A_::HID + net * A_::TILE, ra.Wh[l][net]);
                    if (!SPARE && s == 0) add4(gacc + A_::o_h + l * A_::HID + 2 * A_::TILE + net * 16 + 4 * q, ra.Bh[l][net]);
                }
                if (s == 0) add4(gacc + A_::o_b0 + net * 16 + 4 * q, ra.B0[net]);
            }
            if (a.g_fold) {
#pragma unroll
                for (int mm = 0; mm < HT; ++mm)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float ax = red16(dAx[mm][j]), bx = red16(dBx[mm][j]);
                        const float ay = red16(dAy[mm][j]), by = red16(dBy[mm][j]);
                        if (s == 0) {
                            const int fx = c_off + 16 * mm + 4 * q + j, fy = t_off + 16 * mm + 4 * q + j;
                            gacc[A_::o_fold + fx] += ax;
                            gacc[A_::o_fold + D + fx] += bx;
                            gacc[A_::o_fold + fy] += ay;
                            gacc[A_::o_fold + D + fy] += by;
                        }
                    }
            }
        }
        __syncthreads();
    }
    if (finalize && a.glp_sum) {
        const float tot = red16(glp_acc);
        if (lane == 0) atomicAdd(a.glp_sum + mp, tot * isc);
    }
    const int P = 2 * (H * U + U) + (L - 1) * 2 * (U * U + U) + 2 * (U * H + H);
    float* gout = a.g_params + mp * a.gpstride;
    for (int k = threadIdx.x; k < P; k += NW * 64) atomicAdd(gout + k, gacc[acc_src<H, L, SPARE>(k, U)] * isc);
    if (a.g_fold) {
        float* gfo = a.g_fold + mp * a.fold_stride;
        for (int i = threadIdx.x; i < 2 * D; i += NW * 64) atomicAdd(gfo + i, gacc[A_::o_fold + i] * isc);
    }
}

template <int H, int L>
static void launch_layer_f16(const BwdArgs& a, int inverse, dim3 grid, hipStream_t st) {
    const dim3 blk(kLayerNW * 64);
    if (inverse) {
        if (a.U <= 15) hipLaunchKernelGGL((coupling_bwd_f16_kernel<H, L, true, true>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((coupling_bwd_f16_kernel<H, L, false, true>), grid, blk, 0, st, a);
    } else {
        if (a.U <= 15) hipLaunchKernelGGL((coupling_bwd_f16_kernel<H, L, true, false>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((coupling_bwd_f16_kernel<H, L, false, false>), grid, blk, 0, st, a);
    }
}

// shapes of mfma_supported(); g_ld and g_z required; g_z_out may be NULL only when g_lp seeds the layer (finalize)
int launch_coupling_backward_f16(const BwdArgs& a, int D, int L, int inverse, hipStream_t st) {
    if (!(D == 64 || D == 32) || L < 1 || L > 3 || a.U < 1 || a.U > 16)
        return fail(TNF_EUNSUPPORTED, "coupling_backward_f16: D=%d L=%d U=%d", D, L, a.U);
    if (a.N <= 0) return TNF_OK;
    diag_count(TNF_DIAG_BWD_LAYER_F16);
    const int64_t ntiles = (a.N + 15) / 16;
    int64_t bx = (ntiles + kLayerNW - 1) / kLayerNW;
    int64_t cap = (256 + a.M - 1) / a.M;
    if (bx > cap) bx = cap;
    const dim3 grid = grid_xm(bx, a.M);
    if (D == 64) {
        if (L == 1) launch_layer_f16<32, 1>(a, inverse, grid, st);
        else if (L == 2) launch_layer_f16<32, 2>(a, inverse, grid, st);
        else launch_layer_f16<32, 3>(a, inverse, grid, st);
    } else {
        if (L == 1) launch_layer_f16<16, 1>(a, inverse, grid, st);
        else if (L == 2) launch_layer_f16<16, 2>(a, inverse, grid, st);
        else launch_layer_f16<16, 3>(a, inverse, grid, st);
    }
    return check_launch("coupling_backward_f16");
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int64_t rev_image_floats(int D, int L) {
    const int H = D / 2, HT = (H + 15) / 16;
    const int64_t f = (int64_t)(((H == 32) ? 4 : 2) + 2 * (L - 1) + 2 * HT) * 256 + (2 + 2 * (L - 1) + 2 * HT) * 16;
    const int64_t b = (int64_t)(((H == 32) ? 4 : 2) + 2 * (L - 1) + 2 * HT) * 256;
    return f + b + 4 * D;
}



static int64_t rev_acc_ints(int D, int L) {
    const int H = D / 2;
    const int64_t HT = H / 16;
    return (2 * HT * 256 + 32 + (int64_t)(L - 1) * (2 * 256 + 32) + 2 * HT * 256 + 2 * H + 4 * H + 3) & ~3LL;
}
static int64_t rev_lds_bytes(int D, int S, int L, int U) {  // the round-2 kernel (one tile per wave)
    (void)U;
    const int64_t slot = (rev_image_floats(D, L) + 255) & ~255LL;  // whole 1-KB LDS-DMA pieces
    return (2 * slot + 2 * S * rev_acc_ints(D, L) + (int64_t)kRevNW * kRevNScr * kScr) * 4;
}
static int64_t pair_lds_bytes(int D, int S, int L) {  // the pair kernel (PairLds<H, L>::floats)
    const int64_t slot = (rev_image_floats(D, L) + 255) & ~255LL;
    return (2 * slot + 2 * S * rev_acc_ints(D, L) + (int64_t)kPairNW * kPairNT * kScr) * 4;
}

// TNF_OPT_REV_VARIANT: 0 = flow_bwd_f16_kernel (default), 1 = the magic-number form of flow_bwd_pair.h (measured
// slower or equal on every shape tried; kept selectable as the evidence)
thread_local int g_rev_variant = 0;

int flow_train_rev_supported(int D, int S, int L, int U) {
    if (!(D == 64 || D == 32) || L < 1 || L > 3 || U < 1 || U > 16 || S < 1) return 0;
    return rev_lds_bytes(D, S, L, U) <= 160 * 1024 ? 1 : 0;
}
static bool rev_use_pair(int D, int S, int L, int U) {
    (void)U;
    return g_rev_variant == 1 && pair_lds_bytes(D, S, L) <= 160 * 1024;
}

// launch geometry of the backward kernels (shared with the workspace size: the partial rows depend on it)
static int64_t rev_blocks_x(int64_t M, int64_t N, bool pair) {
    const int64_t units = pair ? (N + 16 * kPairNT - 1) / (16 * kPairNT) : (N + 15) / 16;  // groups of kPairNT tiles / tiles
    const int nw = pair ? kPairNW : kRevNW;
    int64_t bx = (units + nw - 1) / nw;
    const int64_t cap = (256 + M - 1) / M;
    return bx > cap ? cap : (bx < 1 ? 1 : bx);
}
// workspace: [rimg | g_fold (Mp, 2S, 2, D) + glp_sum (Mp) + gmax (1) + overflow (1) | glp partials | partial rows];
// g_fold .. overflow are zeroed by the backward
struct RevWs {
    int64_t rimg, gfold, glpp, part, total, nred;
};
static RevWs rev_ws(int64_t M, int64_t Mp, int64_t N, int D, int S, int L, int U) {
    RevWs w;
    const int H = D / 2;
    const int64_t P = 2 * (H * U + U) + (int64_t)(L - 1) * 2 * (U * U + U) + 2 * (U * H + H);
    const int64_t bx0 = rev_blocks_x(M, N > 0 ? N : 1, false), bx1 = rev_blocks_x(M, N > 0 ? N : 1, true);
    w.nred = (Mp == 1 ? M : 1) * (bx0 > bx1 ? bx0 : bx1);  // workgroups that add into one gradient row (either kernel)
    w.rimg = 0;
    w.gfold = ((Mp * 2 * S * rev_image_floats(D, L) * 4 + 255) / 256) * 256;
    w.glpp = w.gfold + ((Mp * 2 * S * 2 * D + Mp + 2) * 4 + 255) / 256 * 256;
    w.part = w.glpp + ((Mp * w.nred * 4 + 255) / 256) * 256;
    w.total = w.part + ((Mp * w.nred * 2 * S * (P + 2 * D) * 4 + 255) / 256) * 256;
    return w;
}
int64_t flow_train_rev_workspace(int64_t M, int64_t Mp, int64_t N, int D, int S, int L, int U) {
    // one shared parameter row: the M * N samples are one batch (launch_flow_bwd_rev), so the partial rows do not grow with M
    if (Mp == 1 && M > 1) return rev_ws(1, 1, M * N, D, S, L, U).total;
    return rev_ws(M, Mp, N, D, S, L, U).total;
}

template <int H, int L>
static int launch_rev(const float* z0, const float* params, const float* bn_mean, const float* bn_alpha,
                      const float* g_lp, float* g_z, float* g_params, int64_t M, int64_t Mp, int64_t N, int S, int U,
                      int64_t pstride, int64_t gpstride, char* ws, int* overflow_out, hipStream_t st) {
    constexpr int D = 2 * H;
    typedef RevImage<H, L> R;
    static_assert(R::FLOATS % 4 == 0, "image is copied in 16-byte units");
    diag_count(TNF_DIAG_BWD_FLOW_REV);
    if (rev_image_floats(D, L) != R::FLOATS) return fail(TNF_ELAUNCH, "flow_bwd_f16: image size mismatch");
    if (rev_lds_bytes(D, S, L, U) != (2 * (int64_t)PairLds<H, L>::SLOT + 2 * S * (int64_t)AccLayout<H, L>::INTS + (int64_t)kRevNW * kRevNScr * kScr) * 4 ||
        pair_lds_bytes(D, S, L) != PairLds<H, L>::floats(2 * S) * 4)
        return fail(TNF_ELAUNCH, "flow_bwd_f16: LDS size mismatch");
    const bool pair = rev_use_pair(D, S, L, U);
    const RevWs w = rev_ws(M, Mp, N, D, S, L, U);
    float* rimg = reinterpret_cast<float*>(ws + w.rimg);
    float* gfold = reinterpret_cast<float*>(ws + w.gfold);
    float* glp_sum = gfold + Mp * 2 * S * 2 * D;
    unsigned* gmax = reinterpret_cast<unsigned*>(glp_sum + Mp);
    int* overflow = reinterpret_cast<int*>(gmax + 1);
    if (hipMemsetAsync(gfold, 0, (size_t)(Mp * 2 * S * 2 * D + Mp + 2) * sizeof(float), st) != hipSuccess)
        return fail(TNF_ELAUNCH, "flow_bwd_f16: memset failed");
    hipLaunchKernelGGL((flow_rev_images_kernel<H, L>), grid_xm(2 * S, Mp), dim3(64), 0, st, params, bn_mean, bn_alpha,
                       rimg, S, U, pstride, Mp, pair ? 4.f : 1.f);
    {
        const int64_t n = M * N;
        int64_t blocks = (n + 255) / 256;
        if (blocks > 256) blocks = 256;
        hipLaunchKernelGGL(flow_gmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g_lp, n, gmax, (const int*)nullptr);
    }
    const FlowLayout fl = flow_layout(D, S, L, U);
    const int64_t bx = rev_blocks_x(M, N, pair);
    const int nw = pair ? kPairNW : kRevNW;
    const int64_t units = pair ? (N + 16 * kPairNT - 1) / (16 * kPairNT) : (N + 15) / 16;
    // an accumulator receives iters * NW terms per workgroup; allow 2^13 per term inside the int32 range (2^11 in the
    // round-3 kernel, whose magic-number sums hold 22 bits per term and round three times per term instead of once)
    const int64_t adds = ((units + bx * nw - 1) / (bx * nw)) * nw;
    int fbits = 31 - 13;
    for (int64_t v = 1; v < adds; v <<= 1) --fbits;
    if (pair && fbits > kPairMaxFbits) fbits = kPairMaxFbits;
    if (fbits < 0) fbits = 0;
    const float fx = ldexpf(1.f, fbits);
    FlowBwdArgs a{z0, g_lp, rimg, gmax, g_z, g_params, gfold, glp_sum, M, Mp, N, gpstride, fl.stage, fl.p_up, S, U, fx,
                  reinterpret_cast<int*>(ws + w.part), reinterpret_cast<float*>(ws + w.glpp), overflow};
    // the partial rows are indexed by this launch's own grid; the reduction reads as many
    const int64_t nred = (Mp == 1 ? M : 1) * bx;
    if (pair) {
        const size_t smem = (size_t)pair_lds_bytes(D, S, L);
        auto kern = U <= 15 ? flow_bwd_pair_kernel<H, L, true> : flow_bwd_pair_kernel<H, L, false>;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return fail(TNF_ELAUNCH, "flow_bwd_pair: cannot reserve %zu B of LDS", smem);
        hipLaunchKernelGGL(kern, grid_xm(bx, M), dim3(kPairNW * 64), smem, st, a);
    } else {
        const size_t smem = (size_t)rev_lds_bytes(D, S, L, U);
        auto kern = U <= 15 ? flow_bwd_f16_kernel<H, L, kRevNW, true> : flow_bwd_f16_kernel<H, L, kRevNW, false>;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return fail(TNF_ELAUNCH, "flow_bwd_f16: cannot reserve %zu B of LDS", smem);
        hipLaunchKernelGGL(kern, grid_xm(bx, M), dim3(kRevNW * 64), smem, st, a);
    }
    int rc = check_launch("flow_bwd_f16");
    if (rc) return rc;
    {
        const int H_ = D / 2;
        const int P = 2 * (H_ * U + U) + (L - 1) * 2 * (U * U + U) + 2 * (U * H_ + H_);
        const int64_t prow = (int64_t)2 * S * (P + 2 * D);
        hipLaunchKernelGGL(flow_bwd_reduce_kernel, grid_xm((prow + 255) / 256, Mp), dim3(256), 0, st,
                           reinterpret_cast<const int*>(ws + w.part), reinterpret_cast<const float*>(ws + w.glpp), gmax, overflow,
                           g_params, gfold, glp_sum, nred, 2 * S, P, D, gpstride, fl.stage, fl.p_up, fx, Mp, pair ? 1 : 0, U, L);
        rc = check_launch("flow_bwd_reduce");
        if (rc) return rc;
    }
    if (overflow_out && hipMemcpyAsync(overflow_out, overflow, sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(TNF_ELAUNCH, "flow_bwd_f16: copy of the overflow flag failed");
    return launch_flow_fold_backward(params, bn_alpha, gfold, glp_sum, g_params, Mp, D, S, L, U, pstride, gpstride, st);
}

int launch_flow_bwd_rev(const float* z0, const float* params, const float* bn_mean, const float* bn_alpha,
                        const float* g_lp, float* g_z, float* g_params, int64_t M, int64_t Mp, int64_t N, int D, int S,
                        int L, int U, int64_t pstride, int64_t gpstride, void* ws, int* overflow_out, hipStream_t st) {
    if (!flow_train_rev_supported(D, S, L, U))
        return fail(TNF_EUNSUPPORTED, "flow_bwd_f16: D=%d S=%d L=%d U=%d", D, S, L, U);
    if (N <= 0) return TNF_OK;
    if (Mp == 1 && M > 1) {  // z0 (M, N, D) and g_lp (M, N) are contiguous: M rows of one parameter row = one batch of M * N
        N *= M;
        M = 1;
    }
    char* wsb = reinterpret_cast<char*>(ws);
#define TNF_REV(HH, LL) \
    return launch_rev<HH, LL>(z0, params, bn_mean, bn_alpha, g_lp, g_z, g_params, M, Mp, N, S, U, pstride, gpstride, wsb, \
                              overflow_out, st)
    if (D == 64) {
        if (L == 1) TNF_REV(32, 1);
        if (L == 2) TNF_REV(32, 2);
        TNF_REV(32, 3);
    }
    if (L == 1) TNF_REV(16, 1);
    if (L == 2) TNF_REV(16, 2);
    TNF_REV(16, 3);
#undef TNF_REV
}

}  // namespace tnf
