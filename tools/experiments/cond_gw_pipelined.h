// NOT COMPILED INTO THE LIBRARY -- the hand-ordered g_W kernel measured in DESIGN.md 3.11.10 (round 3), kept for the record.
// It replaces cond_gw_kernel in torch_nf_amd/csrc/cond_flow_bwd.hip (same GwArgs / GwJob / gw_job, TNF_GW_CH 64): weight item
// slots per wave at compile time, every instruction of the contraction `asm volatile` in a fixed order (operand i + 1 built
// three vector instructions at a time behind the twelve MFMAs of item i), MFMAs in place, packed staging maps, no spill in
// the loop.  3.13 ms at 2^18 contexts against 3.03-3.19 ms for the kernel that ships: the SIMD's vector issue port, which the
// MFMAs share (8 of their 16 cycles), is what binds, and re-ordering does not unload it.  Known loose end: the H = 32
// (two column tiles) instantiation failed test_cond_flow_training_gradients[32-1-1-15-4-hidden0-64-1-False].
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<I + 1, N>(f);
    }
}

// acc += a . b in place: the builtin lets the register allocator give D and C different registers, and under this kernel's
// pressure it did (accumulators rotating through twice their number).  Hazards hipcc no longer sees: the accumulators are
// read by vector instructions only in the epilogue, far behind the last MFMA; a VALU result feeding SrcA / SrcB and an MFMA
// chained on SrcC == vDst are interlocked by the hardware.
#ifndef TNF_GW_ASM_MFMA
#define TNF_GW_ASM_MFMA 1
#endif
__device__ __forceinline__ void gw_mfma(h8 a, h8 b, f4& acc) {
#if TNF_GW_ASM_MFMA
    asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#else
    acc = cmfma32h(a, b, acc);
#endif
}
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ constexpr h2 kOnes2 = {(_Float16)1.f, (_Float16)1.f};

template <int OFF>
__device__ __forceinline__ h8 lds_read16_blind(unsigned base) {
    h8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF));
    return v;
}

template <int DT, int KS, int NW>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2)))
cond_gw_kernel(GwArgs a) {
    constexpr int JT = 2 * KS, NT = 64 * NW;
    constexpr int XR = 32;                        // staged x rows: one 32-wide layer-0 input or 2 x 16 hidden units
    constexpr int IPW = XR / NW;                  // weight items (one input unit k of one segment) per wave at most
    constexpr int CH = gw_ch<KS>();               // contexts staged per step
    constexpr int CS = CH + 4;                    // padded row stride of the transposed staging buffers
    constexpr int NX = CH * (XR / 4) / NT, ND = CH * 8 / NT;  // float4 a thread stages per step: x (8 per context), deltas (2 x 4)
    static_assert(NX * NT == CH * (XR / 4) && ND * NT == CH * 8 && IPW * NW == XR, "staging maps assume these divide");
    constexpr int HB_U4 = (CH / 32) * JT * 2 * 64;  // one step's h operands (32 KB at H = 64)
    // Two staging buffers each: a wave writes the next step's rows right behind its own MFMAs of this step, and ONE
    // barrier per step publishes them.  Rows: x of segment sg, input k at sg * xrs + k; deltas of segment sg, output o at
    // 16 sg + o; four dump rows at the end take the stores of the lanes that have nothing to stage.
    constexpr int XTF = (XR + 4) * CS, DTF = (32 + 4) * CS;
    __shared__ __attribute__((aligned(16))) float xT2[2 * XTF];
    __shared__ __attribute__((aligned(16))) float dT2[2 * DTF];
    // h as ready MFMA B operands [sub-step][jt][hi/lo][lane], copied from the pre-split image by LDS-DMA (contiguous).
    // Two OBJECTS and the step loop unrolled by two, so that every access names its slot at compile time; the reads are
    // issued by hand (lds_read16_blind): hipcc orders a visible LDS read behind every LDS-DMA copy in flight that it cannot
    // prove disjoint -- s_waitcnt vmcnt(0) in front of the first operand read, i.e. the whole fetch exposed once per step.
    __shared__ __attribute__((aligned(16))) u4 hB0[HB_U4];
    __shared__ __attribute__((aligned(16))) u4 hB1[HB_U4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    // XCD-aware placement: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own
    // L2.  All jobs of one context slice are given ids of the same residue mod 8, so the slice's h image
    // (a.spp steps, ~1 MB) is fetched into ONE L2 and re-read there by the other jobs.
    const int xcd = blockIdx.x & 7, tq = blockIdx.x >> 3;
    const int split = __builtin_amdgcn_readfirstlane(xcd + 8 * (tq / a.jobs));
    __shared__ GwJob job;
    if (tid == 0) job = gw_job(a, tq % a.jobs);
    __syncthreads();
    const int d_in = __builtin_amdgcn_readfirstlane(job.d_in), nseg = __builtin_amdgcn_readfirstlane(job.nseg);
    const int xs = __builtin_amdgcn_readfirstlane((int)job.xs), dstr = __builtin_amdgcn_readfirstlane((int)job.dstr);
    const int xq = (d_in + 3) >> 2, xrs = 4 * xq;  // float4 per x row of a segment (rows are padded to 4 in the records)
    const int xrow4 = nseg * xq;                  // float4 per context over all segments
    // weight item SLOTS of the job: slot = x row = sg * xrs + k (the k >= d_in ones are padding: computed, never written out),
    // so that a wave's slots wave, wave + NW, .. are rows a compile-time stride apart and every wave has as many
    const int nslot = nseg * xrs;
    const int ni = nslot > wave ? (nslot - wave + NW - 1) / NW : 0;
    const int nbu = nseg * JT;                    // bias units: (segment, 16-column tile of h), 3 MFMAs each
    const f4 zero = {0.f, 0.f, 0.f, 0.f};

    // ---- staging maps (the same every step).  Lanes 2n, 2n+1 take float4 k, k+1 of context n: the four transposing
    // ds_write_b32 of a lane pair hit banks 16 apart and a 32-lane group covers all 32 (with the float4 index fastest
    // the stores were 4-way conflicts).
    typedef const __attribute__((address_space(1))) f4* gf4p;
    const float* xbase = job.seg[0].x ? job.seg[0].x : a.h;
    const float* dbase = job.seg[0].d;
    const int segdx = nseg > 1 && job.seg[0].x ? (int)(job.seg[1].x - job.seg[0].x) : 0;
    const int segdd = nseg > 1 ? (int)(job.seg[1].d - job.seg[0].d) : 0;
    const int cnt0 = job.seg[0].count, cnt1 = nseg > 1 ? job.seg[1].count : 0;
    // per staged float4 pair (x: i < NX, deltas: NX == ND) two words, loop-invariant and kept small -- the kernel runs at the
    // register limit: [x element offset : 12 | delta element offset : 12 | valid deltas : 3] and [x store : 16 | delta store : 16]
    static_assert(NX == ND && (XR + 4) * CS * 4 < 65536, "packed staging maps");
    unsigned mlanes[NX], mdst[NX];
    const int sctx = (tid >> 1) & (CH - 1);  // the context of a step this thread stages (NT is a multiple of 2 CH or divides it)
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int idx = tid + i * NT;
        const int hi = idx / (2 * CH);
        const int k4g = (idx & 1) + 2 * hi;
        const bool okx = k4g < xrow4;
        const int sgx = okx ? k4g / xq : 0, k4 = okx ? k4g - sgx * xq : 0;
        const int xl = (sgx ? segdx : 0) + 4 * k4;
        const int xd = ((okx ? sgx * xrs + 4 * k4 : XR) * CS + sctx) * 4;  // byte offset in a buffer
        const int o4 = (idx & 1) + 2 * (hi & 1), sgd = hi >> 1;
        const bool okd = sgd < nseg;
        const int dl = (okd && sgd ? segdd : 0) + 4 * o4;
        const int dd = ((okd ? sgd * 16 + 4 * o4 : 32) * CS + sctx) * 4;
        const int left = (sgd ? cnt1 : cnt0) - 4 * o4;
        const int ne = !okd || left < 0 ? 0 : (left < 4 ? left : 4);
        mlanes[i] = (unsigned)xl | ((unsigned)dl << 12) | ((unsigned)ne << 24);
        mdst[i] = (unsigned)xd | ((unsigned)dd << 16);
    }
    // Nothing touches the loaded registers before commit(): an instruction on them in between would make the wave wait for
    // the loads in front of its MFMAs instead of behind them.  Contexts past M: the address is clamped to row M - 1 (finite
    // values) and the deltas are stored as zeros.
    f4 px[NX], pd[ND];
    const unsigned mlast = (unsigned)(a.M - 1);
    auto fetch = [&](int64_t mbase) __attribute__((always_inline)) {
        const unsigned mb = (unsigned)mbase;
        unsigned m = mb + sctx;
        m = m < mlast ? m : mlast;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            px[i] = *(gf4p)(xbase + ((uint64_t)(m * (unsigned)xs + (mlanes[i] & 0xfffu))));
            pd[i] = *(gf4p)(dbase + ((uint64_t)(m * (unsigned)dstr + ((mlanes[i] >> 12) & 0xfffu))));
        }
    };
    auto commit = [&](int buf, int64_t mbase) __attribute__((always_inline)) {
        char* xT = reinterpret_cast<char*>(xT2 + buf * XTF);
        char* dT = reinterpret_cast<char*>(dT2 + buf * DTF);
        const unsigned mb = (unsigned)mbase;
        const bool in = mb + sctx <= mlast;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<float*>(xT + (mdst[i] & 0xffffu) + e * CS * 4) = px[i][e];
            const int ne = in ? (int)(mlanes[i] >> 24) : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<float*>(dT + (mdst[i] >> 16) + e * CS * 4) = e < ne ? pd[i][e] : 0.f;
        }
    };
    const int64_t himg_u4 = ((a.M + 31) / 32) * JT * 2 * 64;
    auto fetch_h = [&](int64_t step, u4* hdst) __attribute__((always_inline)) {  // asynchronous: complete behind the issuing wave's vmcnt(0) + a barrier
        const int64_t base = step * HB_U4;
        for (int i = wave; i < HB_U4 / 64; i += NW) {
            int64_t g = base + i * 64 + lane;
            g = g < himg_u4 ? g : himg_u4 - 1;  // past the last group: never used (the deltas are zero there)
            __builtin_amdgcn_global_load_lds(a.himg + g, (lds_void*)(hdst + i * 64), 16, 0, 0);
        }
    };

    const int nsteps = (int)((a.M + CH - 1) / CH);
    const int st0 = __builtin_amdgcn_readfirstlane(split * a.spp);
    const int st_end = __builtin_amdgcn_readfirstlane((st0 + a.spp) < nsteps ? (st0 + a.spp) : nsteps);
    if (st0 < st_end) {  // step 0 into buffer 0
        fetch((int64_t)st0 * CH);
        fetch_h(st0, hB0);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        commit(0, (int64_t)st0 * CH);
    }
    __syncthreads();

    // ---- the contraction, with the number of weight items of this wave at compile time: no branch inside a step, so that
    // item i + 1's operand (8 products, their f16 halves: ~30 vector instructions) is built in the shadow of item i's
    // twelve MFMAs -- one MFMA, then two or three of those instructions, pinned by sched_barrier.  As separate per-item
    // blocks (hipcc guarded each with the exec mask) a wave did one after the other and the matrix pipe idled 3/4 of the time.
    auto run = [&](auto ni_c) __attribute__((always_inline)) {
        constexpr int NI = decltype(ni_c)::value;
        f4 acc[NI > 0 ? NI : 1][JT], accb[(2 * JT + NW - 1) / NW];
        float gb[NI > 0 ? NI : 1], gbb[(2 * JT + NW - 1) / NW];
        int dseg[NI > 0 ? NI : 1];  // byte offset of the item's delta tile (uniform)
#pragma unroll
        for (int i = 0; i < (NI > 0 ? NI : 1); ++i) {
            gb[i] = 0.f;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) acc[i][jt] = zero;
            dseg[i] = xrs ? ((wave + i * NW) / xrs) * 16 * CS * 4 : 0;
        }
#pragma unroll
        for (int u = 0; u < (2 * JT + NW - 1) / NW; ++u) {
            accb[u] = zero;
            gbb[u] = 0.f;
        }

        const int arow0 = (wave * CS + 8 * q) * 4, drow0 = (r * CS + 8 * q) * 4;
        const unsigned ones2 = 0x3c003c00u;  // (1, 1) in f16
        auto step = [&](auto slot, int st) __attribute__((always_inline)) {
            constexpr int cur = decltype(slot)::value;
            // here buffer `cur` and h slot `cur` are complete and visible, and nobody reads the other ones any more
            if (st + 1 < st_end) {  // both in flight during the MFMAs below
                fetch((int64_t)(st + 1) * CH);
                fetch_h(st + 1, cur ? hB0 : hB1);
            }
            const char* xT = reinterpret_cast<const char*>(xT2 + cur * XTF);
            const char* dT = reinterpret_cast<const char*>(dT2 + cur * DTF);
            const unsigned hbase = (unsigned)(uintptr_t)(lds_void*)((cur ? hB1 : hB0) + lane);
            static_for<0, CH / 32>([&](auto sub_c) __attribute__((always_inline)) {
                constexpr int sub = decltype(sub_c)::value;
                h8 Bh[JT], Bl[JT];
                static_for<0, JT>([&](auto jt_c) __attribute__((always_inline)) {
                    constexpr int jt = decltype(jt_c)::value;
                    Bh[jt] = lds_read16_blind<((sub * JT + jt) * 2 + 0) * 1024>(hbase);
                    Bl[jt] = lds_read16_blind<((sub * JT + jt) * 2 + 1) * 1024>(hbase);
                });
                // ---- the items of this sub-step, issued in a fixed order (every instruction below is `asm volatile`; hipcc
                // keeps those in program order).  Left to itself it hoisted all four operand builds and the next sub-step's B
                // reads in front of the MFMAs -- 32 + 32 more live registers, spilled, and every scratch reload waits
                // (vmcnt) for the rows prefetched from global memory.
                // Operand of item i: 8 products x_k[m] delta_o[m] -> hi / lo f16 halves + their sum for the bias gradient:
                // 32 vector micro-operations, dealt out GW_OPC at a time behind each MFMA of item i - 1.
                f4 rx__[2], rd__[2];            // x row (the same for the 16 lanes of a q) and delta row r, contexts 8q .. 8q+7
                float pp__[8];                  // products, then their residuals
                unsigned ph__[2][4], pl__[2][4];  // two operand sets: in use by the MFMAs / under construction
                f4 (&rx)[2] = rx__, (&rd)[2] = rd__;
                unsigned (&ph)[2][4] = ph__, (&pl)[2][4] = pl__;
                const unsigned ones2__ = ones2;
                const unsigned xa = (unsigned)(uintptr_t)(lds_void*)(xT + arow0 + sub * 128);
                const unsigned da = (unsigned)(uintptr_t)(lds_void*)(dT + drow0 + sub * 128);
                auto raw = [&](auto i_c) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_c)::value;
                    f4 (&rx_)[2] = rx, (&rd_)[2] = rd;  // (an asm operand alone does not capture)
                    const unsigned xa_ = xa, da_ = da;
                    unsigned d2;  // formed here and now: hoisted out of the loop it was spilled, and a scratch reload waits (vmcnt)
                    const int ds_ = dseg[i];
                    asm volatile("v_add_u32 %0, %1, %2" : "=v"(d2) : "s"(ds_), "v"(da_));  // for the rows in flight from HBM
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rx_[0]) : "v"(xa_), "n"(i * (NW * CS * 4)));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rx_[1]) : "v"(xa_), "n"(i * (NW * CS * 4) + 16));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(rd_[0]) : "v"(d2));
                    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(rd_[1]) : "v"(d2));
                };
                auto wait_raw = [&]() __attribute__((always_inline)) {
                    f4 (&rx_)[2] = rx, (&rd_)[2] = rd;
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rx_[0]), "+v"(rx_[1]), "+v"(rd_[0]), "+v"(rd_[1]));
                };
                // micro-operation K of the operand built into set B; GBV: the bias-gradient accumulator of that item
#define GW_OP(B, K, GBV)                                                                                                        \
    {                                                                                                                           \
        f4 (&rx)[2] = rx__, (&rd)[2] = rd__;                                                                                     \
        float (&pp)[8] = pp__;                                                                                                  \
        unsigned (&ph)[2][4] = ph__, (&pl)[2][4] = pl__;                                                                         \
        const unsigned ones2 = ones2__;                                                                                         \
        float& gbv_ = GBV;                                                                                                      \
        if constexpr ((K) < 8)                                                                                                  \
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(pp[K]) : "v"(rx[(K) >> 2][(K) & 3]), "v"(rd[(K) >> 2][(K) & 3]));         \
        else if constexpr ((K) < 24) {                                                                                          \
            constexpr int j_ = ((K) - 8) >> 2, o_ = ((K) - 8) & 3;                                                               \
            if constexpr (o_ == 0)                                                                                              \
                asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(ph[B][j_]) : "v"(pp[2 * j_]), "v"(pp[2 * j_ + 1]));         \
            else if constexpr (o_ == 1)                                                                                         \
                asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(pp[2 * j_]) : "v"(ph[B][j_])); \
            else if constexpr (o_ == 2)                                                                                         \
                asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(pp[2 * j_ + 1]) : "v"(ph[B][j_])); \
            else                                                                                                                \
                asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(pl[B][j_]) : "v"(pp[2 * j_]), "v"(pp[2 * j_ + 1]));         \
        } else if constexpr ((K) < 32) {                                                                                        \
            constexpr int j_ = ((K) - 24) >> 1;                                                                                  \
            if constexpr ((((K) - 24) & 1) == 0)                                                                                 \
                asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(gbv_) : "v"(ph[B][j_]), "v"(ones2));                             \
            else                                                                                                                \
                asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(gbv_) : "v"(pl[B][j_]), "v"(ones2));                             \
        }                                                                                                                       \
    }
                constexpr int NCH = 3 * JT;                       // MFMAs of one item
                constexpr int GW_OPC = (32 + NCH - 1) / NCH;      // micro-operations behind each of them
                if constexpr (NI > 0) {
                    raw(std::integral_constant<int, 0>());
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rx[0]), "+v"(rx[1]), "+v"(rd[0]), "+v"(rd[1]), "+v"(Bh[0]), "+v"(Bl[0]));
                    static_for<0, 8>([&](auto k_c) __attribute__((always_inline)) { GW_OP(0, decltype(k_c)::value, gb[0]) });
                    if constexpr (NI > 1) raw(std::integral_constant<int, 1>());
                    static_for<8, 32>([&](auto k_c) __attribute__((always_inline)) { GW_OP(0, decltype(k_c)::value, gb[0]) });
                }
                if constexpr (JT == 4)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bh[0]), "+v"(Bh[1]), "+v"(Bh[2]), "+v"(Bh[3]), "+v"(Bl[0]),
                                 "+v"(Bl[1]), "+v"(Bl[2]), "+v"(Bl[3]), "+v"(rx[0]), "+v"(rx[1]), "+v"(rd[0]), "+v"(rd[1]));
                else {
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bh[jt]), "+v"(Bl[jt]));
                    wait_raw();
                }
                static_for<0, NI>([&](auto i_c) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_c)::value, b = i & 1, nb = b ^ 1;
                    constexpr int gi = i + 1 < NI ? i + 1 : 0;
                    const h8 ah = __builtin_bit_cast(h8, (u4){ph[b][0], ph[b][1], ph[b][2], ph[b][3]});
                    const h8 al = __builtin_bit_cast(h8, (u4){pl[b][0], pl[b][1], pl[b][2], pl[b][3]});
                    if constexpr (i > 0 && i + 1 < NI) wait_raw();  // rows of item i + 1, requested one item ago
                    static_for<0, NCH>([&](auto c_c) __attribute__((always_inline)) {
                        constexpr int c = decltype(c_c)::value, term = c / JT, jt = c % JT;
                        f4& acc_ = acc[i][jt];
                        const h8 a_ = term == 1 ? al : ah, b_ = term == 2 ? Bl[jt] : Bh[jt];
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc_) : "v"(a_), "v"(b_));
                        if constexpr (i + 1 < NI) {
                            static_for<c * GW_OPC, (c + 1) * GW_OPC>([&](auto k_c) __attribute__((always_inline)) {
                                constexpr int k = decltype(k_c)::value;
                                GW_OP(nb, k, gb[gi])
                                if constexpr (k == 7 && i + 2 < NI) raw(std::integral_constant<int, (i + 2 < NI ? i + 2 : 0)>());
                            });
                        }
                    });
                    if constexpr (i + 1 < NI && NCH * GW_OPC < 32)
                        static_for<NCH * GW_OPC, 32>([&](auto k_c) __attribute__((always_inline)) {
                            constexpr int k = decltype(k_c)::value;
                            GW_OP(nb, k, gb[gi])
                        });
                });
                asm volatile("" ::: "memory");
                // bias units: operand = the delta tile itself (x = 1), one 16-column tile of h each
#pragma unroll
                for (int u = 0; u < (2 * JT + NW - 1) / NW; ++u) {
                    const int bu = wave + u * NW;  // uniform
                    if (bu < nbu) {
                        const int sg = bu / JT, jtu = bu - sg * JT;
                        const char* dr = dT + ((sg * 16 + r) * CS + 8 * q) * 4 + sub * 128;
                        const f4 d0 = *reinterpret_cast<const f4*>(dr), d1 = *reinterpret_cast<const f4*>(dr + 16);
                        h8 ah, al;
                        csplit8(d0, d1, ah, al);
                        h8 bh = Bh[0], bl = Bl[0];
#pragma unroll
                        for (int jt = 1; jt < JT; ++jt) {
                            bh = jtu == jt ? Bh[jt] : bh;
                            bl = jtu == jt ? Bl[jt] : bl;
                        }
                        gw_mfma(ah, bh, accb[u]);
                        gw_mfma(al, bh, accb[u]);
                        gw_mfma(ah, bl, accb[u]);
                        if (jtu == 0) gbb[u] += (d0[0] + d0[1]) + (d0[2] + d0[3]) + (d1[0] + d1[1]) + (d1[2] + d1[3]);
                    }
                }
            });
            if (st + 1 < st_end) {
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the rows fetched above, and this wave's share of the h copy
                commit(cur ^ 1, (int64_t)(st + 1) * CH);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        int st = st0;
        while (st < st_end) {
            step(std::integral_constant<int, 0>(), st);
            if (++st >= st_end) break;
            step(std::integral_constant<int, 1>(), st);
            ++st;
        }

        const float ig = 1.f / cond_gscale(*a.gmaxbits);  // the deltas were computed on scaled upstream gradients
        // ---- merge: rows of the accumulators = outputs 4q + jj, columns = hidden unit 16 jt + r ----
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int it = wave + i * NW, sg = it / xrs, k = it - sg * xrs;
            if (k >= d_in) continue;  // a padding slot
            const GwSeg& g = job.seg[sg];
            const int64_t pb = g.wbase + (int64_t)k * g.wk;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int o = 4 * q + jj;
                if (o < g.count) {
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) atomicAdd(a.g_W + (pb + o) * a.ldgw + 16 * jt + r, acc[i][jt][jj] * ig);
                }
            }
            const float sres = reduce_q(gb[i]);
            if (q == 0 && r < g.count) atomicAdd(a.g_b + pb + r, sres * ig);
        }
#pragma unroll
        for (int u = 0; u < (2 * JT + NW - 1) / NW; ++u) {
            const int bu = wave + u * NW;
            if (bu < nbu) {
                const int sg = bu / JT, jtu = bu - sg * JT;
                const GwSeg& g = job.seg[sg];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int o = 4 * q + jj;
                    if (o < g.count) atomicAdd(a.g_W + (g.bbase + o) * a.ldgw + 16 * jtu + r, accb[u][jj] * ig);
                }
                if (jtu == 0) {
                    const float sres = reduce_q(gbb[u]);
                    if (q == 0 && r < g.count) atomicAdd(a.g_b + g.bbase + r, sres * ig);
                }
            }
        }
    };
    switch (ni) {
        case 0: run(std::integral_constant<int, 0>()); break;
        case 1: run(std::integral_constant<int, 1>()); break;
        case 2: run(std::integral_constant<int, 2>()); break;
        case 3: run(std::integral_constant<int, 3>()); break;
        default: run(std::integral_constant<int, IPW>()); break;
    }
}

#undef GW_OP

