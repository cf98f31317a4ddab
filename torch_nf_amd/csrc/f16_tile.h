// Split-f16 operand helpers shared by the whole-flow kernels (flow_fused_f16.hip, flow_bwd_f16.hip):
// v = hi + lo with hi = rtz_f16(v), lo = rtz_f16(v - hi); an fp32-accurate contraction is three f16
// MFMAs (hi.hi + lo.hi + hi.lo) with fp32 accumulate.  See flow_fused_f16.hip for the measurements.
#pragma once
#include "mfma_tile.h"

namespace tnf {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// Inline-asm rule of this code base (measured the hard way, flow_fused2.hip round 2): hipcc neither sees the registers
// an asm VALU instruction READS as results of an in-flight MFMA, nor pads the WAR / WAW hazards of the registers it
// WRITES against MFMAs still reading (SrcC, up to 7 wait states for an 8-pass MFMA) or writing them.  A fresh "=v"
// output may land in exactly such a register -- results then change with the schedule and from run to run.  So an asm
// VALU instruction here only ever (a) reads results of ordinary VALU instructions and (b) writes IN PLACE ("+v") over a
// value an ordinary VALU instruction produced after the MFMAs in question: the compiler resolved every MFMA hazard of
// that register when it scheduled the producer, and it copies the value first (v_mov, visible) if it is still live.
// two floats -> packed (hi, hi) and (lo, lo) f16 pairs
struct HiLo {
    unsigned hi, lo;
};
__device__ __forceinline__ HiLo split2v(float v0, float v1) {
    const auto h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
#if TNF_ABLATE == 2  // timing experiment only: no remainder
    return HiLo{__builtin_bit_cast(unsigned, h), __builtin_bit_cast(unsigned, h)};
#endif
    // v - (float)hi as ONE mixed-precision FMA reading the f16 half directly (hipcc does not
    // select v_fma_mix_f32 for this pattern; it emits v_cvt_f32_f16 + v_sub_f32).  Exact: the
    // difference of v and its rtz-f16 truncation is representable in fp32.
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    // the remainder replaces the value IN PLACE ("+v"): see the inline-asm rule above
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(v0) : "v"(hb));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v1) : "v"(hb));
    const auto l = __builtin_amdgcn_cvt_pkrtz(v0, v1);
    return HiLo{hb, __builtin_bit_cast(unsigned, l)};
}
// (vector elements cannot bind to references, hence the macro)
#define split2(V0, V1, HI, LO)              \
    do {                                    \
        const HiLo hl_ = split2v((V0), (V1)); \
        (HI) = hl_.hi;                      \
        (LO) = hl_.lo;                      \
    } while (0)

__device__ __forceinline__ void split4(f4 v, h4& hi, h4& lo) {
    u2 a, b;
    split2(v[0], v[1], a[0], b[0]);
    split2(v[2], v[3], a[1], b[1]);
    hi = __builtin_bit_cast(h4, a);
    lo = __builtin_bit_cast(h4, b);
}

__device__ __forceinline__ f4 mfma16h(h4 a, h4 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f4 mfma32h(h8 a, h8 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// LDS / global image of one layer's split-f16 operands: 16-byte groups [g][lane], then the
// fp32 bias groups [g][q][4] exactly as in LdsLayerImage.
template <int H, int L>
struct F16Image {
    static constexpr int HT = (H + 15) / 16;
    static constexpr int NW0 = (H == 32) ? 4 : 2;            // layer-0 groups
    static constexpr int NWG = NW0 + 2 * (L - 1) + 2 * HT;   // 16-byte-per-lane groups
    static constexpr int NBG = 2 + 2 * (L - 1) + 2 * HT;
    static constexpr int FLOATS = NWG * 256 + NBG * 16;      // size in 4-byte units
    __device__ static constexpr int g_w0(int net, int part) { return (H == 32) ? net * 2 + part : net; }
    __device__ static constexpr int g_wh(int l, int net) { return NW0 + 2 * l + net; }
    __device__ static constexpr int g_w2(int net, int mo) { return NW0 + 2 * (L - 1) + net * HT + mo; }
    __device__ static constexpr int b_b0(int net) { return net; }
    __device__ static constexpr int b_bh(int l, int net) { return 2 + 2 * l + net; }
    __device__ static constexpr int b_b2(int net, int mo) { return 2 + 2 * (L - 1) + net * HT + mo; }
};

static_assert(F16Image<32, 3>::FLOATS <= LdsLayerImage<32, 3>::FLOATS, "f16 image must fit the fp32 image slot");
static_assert(F16Image<16, 3>::FLOATS <= LdsLayerImage<16, 3>::FLOATS, "f16 image must fit the fp32 image slot");

// ---------------------------------------------------------------------------
// Prep: one wave per (layer, context) folds + gathers the fp32 operands (load_layer_w),
// splits them and writes the f16 image.
// ---------------------------------------------------------------------------
// one wave: gather + fold + split the operands of the coupling layer whose parameters start at `p` and
// write its image to `img` (global memory in the prep kernel, LDS when the flow kernel builds its own)
template <int H, int L>
__device__ __forceinline__ void build_f16_image(float* img, const float* __restrict__ p, int U, int lane) {
    typedef F16Image<H, L> Img;
    constexpr int HT = Img::HT;
    LayerW<H, L> w;
    load_layer_w<H, L>(w, p, U, lane);
    u4* grp = reinterpret_cast<u4*>(img) + lane;  // group g at grp[g * 64]
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        if constexpr (H == 32) {
            u4 hi, lo;
#pragma unroll
            for (int p = 0; p < 4; ++p) split2(w.w0[net][2 * p], w.w0[net][2 * p + 1], hi[p], lo[p]);
            grp[Img::g_w0(net, 0) * 64] = hi;
            grp[Img::g_w0(net, 1) * 64] = lo;
        } else {
            u4 v;
            split2(w.w0[net][0], w.w0[net][1], v[0], v[2]);
            split2(w.w0[net][2], w.w0[net][3], v[1], v[3]);
            grp[Img::g_w0(net, 0) * 64] = v;  // [hi(4) | lo(4)]
        }
#pragma unroll
        for (int l = 0; l < L - 1; ++l) {
            u4 v;
            split2(w.wh[l][net][0], w.wh[l][net][1], v[0], v[2]);
            split2(w.wh[l][net][2], w.wh[l][net][3], v[1], v[3]);
            grp[Img::g_wh(l, net) * 64] = v;
        }
#pragma unroll
        for (int mo = 0; mo < HT; ++mo) {
            u4 v;
            split2(w.w2[net][mo][0], w.w2[net][mo][1], v[0], v[2]);
            split2(w.w2[net][mo][2], w.w2[net][mo][3], v[1], v[3]);
            grp[Img::g_w2(net, mo) * 64] = v;
        }
    }
    if ((lane & 15) == 0) {
        float* bl = img + Img::NWG * 256 + (lane >> 4) * 4;
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            *reinterpret_cast<f4*>(bl + Img::b_b0(net) * 16) = w.b0[net];
#pragma unroll
            for (int l = 0; l < L - 1; ++l) *reinterpret_cast<f4*>(bl + Img::b_bh(l, net) * 16) = w.bh[l][net];
#pragma unroll
            for (int mo = 0; mo < HT; ++mo) *reinterpret_cast<f4*>(bl + Img::b_b2(net, mo) * 16) = w.b2[net][mo];
        }
    }
}

}  // namespace tnf
