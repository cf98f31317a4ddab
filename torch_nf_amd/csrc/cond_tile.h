// Shared pieces of the conditional-flow kernels (cond_flow.hip: forward, cond_flow_bwd.hip: backward):
// split-f16 helpers, the tile program, the LDS tile stream and the per-tile MFMA.
#pragma once
#include "mfma_tile.h"
#include "tnf_common.h"

namespace tnf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 cmfma32h(h8 a, h8 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// v = hi + lo, hi = rtz_f16(v), lo = rtz_f16(v - hi)   (pairs packed into one dword each)
__device__ __forceinline__ void csplit2(float v0, float v1, unsigned& hi, unsigned& lo) {
    const auto h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(v1));
    hi = hb;
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
__device__ __forceinline__ void csplit8(f4 v0, f4 v1, h8& hi, h8& lo) {
    unsigned a0, a1, a2, a3, b0, b1, b2, b3;
    csplit2(v0[0], v0[1], a0, b0);
    csplit2(v0[2], v0[3], a1, b1);
    csplit2(v1[0], v1[1], a2, b2);
    csplit2(v1[2], v1[3], a3, b3);
    hi = __builtin_bit_cast(h8, (u4){a0, a1, a2, a3});
    lo = __builtin_bit_cast(h8, (u4){b0, b1, b2, b3});
}

// ---------------------------------------------------------------------------
// The tile program: which 16 parameter-row entries tile t holds, in consumption order.
// ---------------------------------------------------------------------------
struct CondCfg {
    int D, S, L, U, H;  // H: width of the conditioner's last hidden layer (multiple of 32)
    int DT, HT;         // 16-feature tiles of z and of one coupling half
    int64_t TC, TS, T;  // tiles per coupling layer, per stage, in total
    FlowLayout fl;
};
__host__ __device__ inline CondCfg cond_cfg(int D, int S, int L, int U, int H) {
    CondCfg c;
    c.D = D; c.S = S; c.L = L; c.U = U; c.H = H;
    c.DT = D / 16;
    c.HT = D / 32;
    c.TC = 2 * (int64_t)(D / 2) + 2 + (int64_t)(L - 1) * (2 * U + 2) + 2 * (int64_t)U * c.HT + 2 * c.HT;
    c.TS = 2 * c.DT + 2 * c.TC;
    c.T = c.TS * S;
    c.fl = flow_layout(D, S, L, U);
    return c;
}
// (first parameter index, number of valid rows) of tile r_f of one coupling layer whose parameters start
// at `off`; r_f counts in FORWARD order: layer 0 weights (k-major, [t, s] per k), its two bias tiles,
// the hidden layers likewise, the output layer's weights (k-major, then o-tile, then [t, s]), its biases.
__host__ __device__ inline void cond_coupling_desc(const CondCfg& c, int64_t off, int64_t r, int64_t& base,
                                                   int& count) {
    const int Hd = c.D / 2, U = c.U;
    const int64_t n0 = 2 * (int64_t)Hd + 2;
    if (r < n0) {  // layer 0: Hd -> U
        count = U;
        if (r < 2 * Hd) base = off + (r & 1) * (int64_t)Hd * U + (r >> 1) * U;
        else base = off + 2 * (int64_t)Hd * U + (r - 2 * Hd) * U;
        return;
    }
    r -= n0;
    off += 2 * (int64_t)Hd * U + 2 * U;
    const int64_t nh = 2 * (int64_t)U + 2;
    if (r < (c.L - 1) * nh) {  // hidden layers: U -> U
        off += (r / nh) * (2 * (int64_t)U * U + 2 * U);
        r %= nh;
        count = U;
        if (r < 2 * U) base = off + (r & 1) * (int64_t)U * U + (r >> 1) * U;
        else base = off + 2 * (int64_t)U * U + (r - 2 * U) * U;
        return;
    }
    r -= (c.L - 1) * nh;
    off += (c.L - 1) * (2 * (int64_t)U * U + 2 * U);
    count = 16;  // output layer: U -> Hd, Hd a multiple of 16
    if (r < 2 * (int64_t)U * c.HT) {
        const int64_t k = r / (2 * c.HT), rem = r % (2 * c.HT);
        base = off + (rem & 1) * (int64_t)U * Hd + k * Hd + 16 * (rem >> 1);
    } else {
        r -= 2 * (int64_t)U * c.HT;
        base = off + 2 * (int64_t)U * Hd + (r & 1) * Hd + 16 * (r >> 1);
    }
}

// Tile t of the FORWARD program (the flow's inverse pass): stages S-1..0, per stage
// [Affine: (alpha, shift) per 16 features | RealNVP(lower) | RealNVP(upper)].
__host__ __device__ inline void cond_tile_desc(const CondCfg& c, int64_t t, int64_t& base, int& count) {
    const int stage = c.S - 1 - (int)(t / c.TS);
    int64_t r = t % c.TS;
    const int64_t so = (int64_t)stage * c.fl.stage;
    if (r < 2 * c.DT) {
        base = so + c.fl.p_up + c.fl.p_low + (r & 1) * c.D + 16 * (r >> 1);
        count = 16;
        return;
    }
    r -= 2 * c.DT;
    if (r < c.TC) cond_coupling_desc(c, so + c.fl.p_up, r, base, count);
    else cond_coupling_desc(c, so, r - c.TC, base, count);
}

// Tile t of the BACKWARD program: stages 0..S-1, per stage [RealNVP(upper) | RealNVP(lower) | Affine],
// and inside a coupling layer the MLP back to front: output layer (weights, biases), hidden layers
// L-1..1, layer 0.  Tiles stay paired [t net, s net] ([alpha, shift] for the Affine).
__host__ __device__ inline void cond_tile_desc_bwd(const CondCfg& c, int64_t t, int64_t& base, int& count) {
    const int stage = (int)(t / c.TS);
    int64_t r = t % c.TS;
    const int64_t so = (int64_t)stage * c.fl.stage;
    if (r >= 2 * c.TC) {
        r -= 2 * c.TC;
        base = so + c.fl.p_up + c.fl.p_low + (r & 1) * c.D + 16 * (r >> 1);
        count = 16;
        return;
    }
    int64_t off = so;
    if (r >= c.TC) {
        r -= c.TC;
        off = so + c.fl.p_up;
    }
    const int64_t n0 = 2 * (int64_t)(c.D / 2) + 2, nh = 2 * (int64_t)c.U + 2;
    const int64_t nlast = 2 * (int64_t)c.U * c.HT + 2 * c.HT;
    int64_t rf;
    if (r < nlast) rf = n0 + (c.L - 1) * nh + r;
    else if (r - nlast < (c.L - 1) * nh) {
        const int64_t rr = r - nlast;
        const int64_t l = c.L - 1 - rr / nh;  // hidden layer index 1..L-1
        rf = n0 + (l - 1) * nh + rr % nh;
    } else rf = r - nlast - (c.L - 1) * nh;
    cond_coupling_desc(c, off, rf, base, count);
}

// power-of-two scale that brings the largest magnitude just below 2^14 (f16 max is 65504)
__device__ __forceinline__ float cond_scale(unsigned maxbits) {
    const float mx = __uint_as_float(maxbits);
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
    return ldexpf(1.f, 13 - ilogbf(mx));
}

// the workgroup's view of the image: chunks of G tiles, double-buffered in LDS
template <int TILE_U4_, int G, int NTHREADS>
struct TileStream {
    static constexpr int TILE_U4 = TILE_U4_;
    static constexpr int CHUNK_U4 = G * TILE_U4;
    static constexpr int PF = (CHUNK_U4 + NTHREADS - 1) / NTHREADS;
    const u4* img;
    u4* stg;
    int64_t total_u4;
    int t;
    u4 pf[PF];

    __device__ __forceinline__ void fetch(int chunk) {
        const int64_t base = (int64_t)chunk * CHUNK_U4;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            int64_t g = base + threadIdx.x + i * NTHREADS;
            g = g < total_u4 ? g : total_u4 - 1;  // clamped, never predicated (see ld_sel)
            pf[i] = img[g];
        }
    }
    __device__ __forceinline__ void commit(int chunk) {
        u4* dst = stg + (chunk & 1) * CHUNK_U4;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int idx = threadIdx.x + i * NTHREADS;
            if (idx < CHUNK_U4) dst[idx] = pf[i];
        }
    }
    __device__ __forceinline__ void init(const u4* image, u4* stage, int64_t tiles) {
        img = image;
        stg = stage;
        total_u4 = tiles * TILE_U4;
        t = 0;
        fetch(0);
        commit(0);
        __syncthreads();
        fetch(1);
    }
    // every wave of the workgroup calls next() the same number of times, in the same order
    __device__ __forceinline__ const u4* next() {
        const int chunk = t / G, in = t - chunk * G;
        if (in == 0 && t > 0) {
            commit(chunk);   // safe: all waves left chunk-2 (same buffer) before the previous barrier
            __syncthreads();
            fetch(chunk + 1);
        }
        ++t;
        return stg + (chunk & 1) * CHUNK_U4 + in * TILE_U4;
    }
};

template <int KS, int BT>
__device__ __forceinline__ void tile_gemm(const u4* tp, int lane, const h8 (&bh)[BT][KS], const h8 (&bl)[BT][KS],
                                          f4 (&P)[BT]) {
    const f4 c0 = *reinterpret_cast<const f4*>(tp + KS * 128 + (lane >> 4));
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) P[bt] = c0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const h8 ah = __builtin_bit_cast(h8, tp[(ks * 2 + 0) * 64 + lane]);
        const h8 al = __builtin_bit_cast(h8, tp[(ks * 2 + 1) * 64 + lane]);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) P[bt] = cmfma32h(ah, bh[bt][ks], P[bt]);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) P[bt] = cmfma32h(ah, bl[bt][ks], P[bt]);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) P[bt] = cmfma32h(al, bh[bt][ks], P[bt]);
    }
}

template <int KS> constexpr int kCondG = 8 / KS;  // tiles per LDS chunk (~16.5 KB)

int launch_cond_image(const float* W, const float* b, int64_t ldw, const CondCfg& cfg, void* ws, void* image,
                      int backward_order, hipStream_t st);


}  // namespace tnf
