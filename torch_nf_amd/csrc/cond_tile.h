// Shared pieces of the conditional-flow kernels (cond_flow.hip: forward, cond_flow_bwd.hip: backward):
// split-f16 helpers, the tile program, the LDS tile stream and the per-tile MFMA.
#pragma once
#ifndef TNF_COND_ABLATE
#define TNF_COND_ABLATE 0
#endif
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
    // in place ("+v"), never into a fresh register: the inline-asm rule of f16_tile.h
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(v0) : "v"(hb));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v1) : "v"(hb));
    hi = hb;
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0, v1));
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

// Tile t of the SAMPLING program (the flow's forward pass, density_estimator.py:374-388): stages 0..S-1, per stage
// [RealNVP(upper) | RealNVP(lower) | Affine], every coupling layer's MLP front to back like the inverse program.
__host__ __device__ inline void cond_tile_desc_fwd(const CondCfg& c, int64_t t, int64_t& base, int& count) {
    const int stage = (int)(t / c.TS);
    int64_t r = t % c.TS;
    const int64_t so = (int64_t)stage * c.fl.stage;
    if (r < c.TC) cond_coupling_desc(c, so, r, base, count);
    else if (r < 2 * c.TC) cond_coupling_desc(c, so + c.fl.p_up, r - c.TC, base, count);
    else {
        r -= 2 * c.TC;
        base = so + c.fl.p_up + c.fl.p_low + (r & 1) * c.D + 16 * (r >> 1);
        count = 16;
    }
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

// The workgroup's view of an operand image: chunks of G tiles in a ring of NS LDS slots, filled by LDS-DMA
// (global_load_lds_dwordx4: 1 KB per wave-instruction, no staging registers, lane-linear destination =
// a plain contiguous copy).  On entering chunk c -- after the barrier that proves every wave is done
// with chunk c-1 -- the copy of chunk c+NS-1 is requested into the slot just vacated, so NS-1 chunks are
// in flight: the image streams from L2 / Infinity Cache with ~2 us latency under load, several chunks of
// MFMA work.  The wait before the barrier is a COUNTED vmcnt (the NS-2 newer copies stay in flight; a
// plain __syncthreads() would drain them with vmcnt(0)) followed by a raw s_barrier.
typedef __attribute__((address_space(3))) void lds_void;

template <int TILE_U4_, int G, int NW, int NS>
struct TileStream {
    static constexpr int TILE_U4 = TILE_U4_;
    static constexpr int CHUNK_U4 = G * TILE_U4;
    static constexpr int NI = (CHUNK_U4 + 63) / 64;  // wave-instructions per chunk
    static constexpr int SLOT_U4 = NI * 64;          // slot size: the tail instruction runs into padding
    static constexpr int LDS_U4 = NS * SLOT_U4;
    static constexpr int KEEP = (NS - 2) * (NI / NW);  // copies every wave may leave in flight at a barrier
    const u4* img;
    u4* stg;
    int64_t total_u4;
    int in, chunk, rslot;

    __device__ __forceinline__ void copy(int c, int slot) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        u4* dst = stg + slot * SLOT_U4;
        const int64_t base = (int64_t)c * CHUNK_U4;
        for (int i = wave; i < NI; i += NW) {
            int64_t g = base + i * 64 + lane;
            g = g < total_u4 ? g : total_u4 - 1;  // clamped: past the end the data is never used
            __builtin_amdgcn_global_load_lds(img + g, (lds_void*)(dst + i * 64), 16, 0, 0);
        }
    }
    __device__ __forceinline__ void arrive() {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(KEEP) : "memory");
    }
    __device__ __forceinline__ void init(const u4* image, u4* stage, int64_t tiles) {
        img = image;
        stg = stage;
        total_u4 = tiles * TILE_U4;
        in = 0;
        chunk = 0;
        rslot = 0;
#pragma unroll
        for (int c = 0; c < NS - 1; ++c) copy(c, c);
        arrive();
        copy(NS - 1, NS - 1);
    }
    // every wave of the workgroup calls next() the same number of times, in the same order
    __device__ __forceinline__ const u4* next() {
        if (in == G) {
            in = 0;
#if TNF_COND_ABLATE != 2  // timing experiment 2: no ring advance (no barrier, no copies)
            ++chunk;
            const int vacated = rslot;
            rslot = rslot + 1 == NS ? 0 : rslot + 1;
            arrive();
            copy(chunk + NS - 1, vacated);
#endif
        }
        const u4* p = stg + rslot * SLOT_U4 + in * TILE_U4;
        ++in;
        return p;
    }
};

// fast transcendental forms (v_exp_f32 / v_rcp_f32 / v_log_f32, ~1 ulp): the precise library versions
// cost ~50 VALU instructions each and, with two waves per SIMD, that is time the matrix pipe idles
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * sig2(kTwoLog2e * x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(kLog2e * x); }
__device__ __forceinline__ float fast_log(float x) { return kLn2 * __builtin_amdgcn_logf(x); }

// MFMA A operands of one tile (split-f16 halves per K step) + the accumulator's initial value
template <int KS>
struct TileOps {
    u4 a[KS][2];
    f4 c0;
};
template <int KS>
__device__ __forceinline__ void load_ops(TileOps<KS>& o, const u4* tp, int lane) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        o.a[ks][0] = tp[(ks * 2 + 0) * 64 + lane];
        o.a[ks][1] = tp[(ks * 2 + 1) * 64 + lane];
    }
    o.c0 = *reinterpret_cast<const f4*>(tp + KS * 128 + (lane >> 4));
}
template <int KS, int BT>
__device__ __forceinline__ void gemm_ops(const TileOps<KS>& o, const h8 (&bh)[BT][KS], const h8 (&bl)[BT][KS],
                                         f4 (&P)[BT]) {
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) P[bt] = o.c0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const h8 ah = __builtin_bit_cast(h8, o.a[ks][0]);
        const h8 al = __builtin_bit_cast(h8, o.a[ks][1]);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) P[bt] = cmfma32h(ah, bh[bt][ks], P[bt]);
#if TNF_COND_ABLATE != 1  // timing experiment 1: one MFMA per K step instead of three
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) P[bt] = cmfma32h(ah, bl[bt][ks], P[bt]);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) P[bt] = cmfma32h(al, bh[bt][ks], P[bt]);
#endif
    }
}

// Two-slot operand pipe over a TileStream: tiles alternate between the slots (every tile sequence of the
// programs is made of [t net, s net] pairs), and a slot is refilled from LDS right after its MFMAs were
// issued, one tile-time before it is needed again.
template <int KS, int BT, typename Stream>
struct TilePipe {
    Stream ts;
    TileOps<KS> op0, op1;
    __device__ __forceinline__ void init(const u4* image, u4* stage, int64_t tiles, int lane) {
        ts.init(image, stage, tiles);
        load_ops<KS>(op0, ts.next(), lane);
        load_ops<KS>(op1, ts.next(), lane);
    }
    __device__ __forceinline__ void gemm0(int lane, const h8 (&bh)[BT][KS], const h8 (&bl)[BT][KS], f4 (&P)[BT]) {
        gemm_ops<KS, BT>(op0, bh, bl, P);
        load_ops<KS>(op0, ts.next(), lane);
    }
    // slot 1 is refilled by the caller AFTER it consumed the pair's results (refill1): LDS returns in order
    // and hipcc waits lgkmcnt(0) for the scalar inputs of that consumption, so a refill issued before it
    // would be waited for on the spot; issued after, it has the next pair's slot-0 MFMAs to land
    __device__ __forceinline__ void gemm1(int lane, const h8 (&bh)[BT][KS], const h8 (&bl)[BT][KS], f4 (&P)[BT]) {
        gemm_ops<KS, BT>(op1, bh, bl, P);
    }
    __device__ __forceinline__ void refill1(int lane) { load_ops<KS>(op1, ts.next(), lane); }
    __device__ __forceinline__ void skip1(int lane) { load_ops<KS>(op1, ts.next(), lane); }  // = refill1
    __device__ __forceinline__ void skip_pair(int lane) {
        load_ops<KS>(op0, ts.next(), lane);
        load_ops<KS>(op1, ts.next(), lane);
    }
};

template <int KS> constexpr int kCondG = 8 / KS;  // tiles per LDS chunk (~16.5 KB)
// ring depth of the forward stream: 3 slots when the 160 KB of LDS allow it next to the waves' state
template <int KS, int BT, int NW> constexpr int kCondNS = (NW * BT <= 16) ? 3 : 2;

int launch_cond_image(const float* W, const float* b, int64_t ldw, const CondCfg& cfg, void* ws, void* image,
                      int backward_order, hipStream_t st);


}  // namespace tnf
