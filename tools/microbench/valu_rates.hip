// Per-SIMD issue cost of the vector instructions the whole-flow kernels are made of (gfx950).
// Each wave runs REPS x 32 independent instructions of one kind between two s_memtime stamps; W waves per SIMD
// (blocks of 256 * W threads, one block per CU) run the same loop concurrently.  Reported: cycles per
// wave-instruction seen by ONE wave, and cycles of SIMD time per wave-instruction (= the former / W).
//   hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REPS 256
#define UNROLL 32

#define BODY_asm(INSTR)                                               \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) asm volatile(INSTR : "+v"(x[u]) : "v"(c0), "v"(c1));

template <int KIND>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* cyc, float c0, float c1) {
    float x[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) x[u] = c0 * (threadIdx.x + u);
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < REPS; ++r) {
        if constexpr (KIND == 0) { BODY_asm("v_fma_f32 %0, %0, %1, %2") }
        if constexpr (KIND == 1) { BODY_asm("v_exp_f32 %0, %0") }
        if constexpr (KIND == 2) { BODY_asm("v_rcp_f32 %0, %0") }
        if constexpr (KIND == 3) { BODY_asm("v_cvt_pk_f16_f32 %0, %0, %1") }
        if constexpr (KIND == 4) { BODY_asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]") }
        if constexpr (KIND == 5) { BODY_asm("v_fma_mixlo_f16 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]") }
        if constexpr (KIND == 6) { BODY_asm("v_mul_f32 %0, %0, %1") }
        if constexpr (KIND == 7) { BODY_asm("v_cvt_pkrtz_f16_f32 %0, %0, %1") }
        if constexpr (KIND == 8) {  // the sigmoid + split mix of one 4-value group: 4 exp, 4 fma, 4 rcp, 2 cvt, 4 mix, 2 cvt
#pragma unroll
            for (int u = 0; u < UNROLL; u += 4) {
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                             "v_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %4, %4\n\t"
                             "v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3"
                             : "+v"(x[u]), "+v"(x[u + 1]), "+v"(x[u + 2]), "+v"(x[u + 3]) : "v"(c0));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) s += x[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_instr) {
    for (int W : {1, 2, 4}) {
        const int threads = 256 * W, blocks = 256;
        float* out;
        unsigned long long* cyc;
        hipMalloc(&out, sizeof(float) * threads * blocks);
        hipMalloc(&cyc, sizeof(unsigned long long) * blocks * threads / 64);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        const double per = med / ((double)REPS * UNROLL * per_instr / UNROLL);
        printf("%-34s W=%d waves/SIMD: %7.2f cycles per wave-instruction (one wave), %6.2f of SIMD time\n", name, W,
               per / 1.0, per / W);
        hipFree(out);
        hipFree(cyc);
    }
}

int main() {
    run<0>("v_fma_f32", UNROLL);
    run<6>("v_mul_f32", UNROLL);
    run<1>("v_exp_f32", UNROLL);
    run<2>("v_rcp_f32", UNROLL);
    run<3>("v_cvt_pk_f16_f32", UNROLL);
    run<7>("v_cvt_pkrtz_f16_f32", UNROLL);
    run<4>("v_fma_mix_f32", UNROLL);
    run<5>("v_fma_mixlo_f16", UNROLL);
    run<8>("sigmoid group (4 exp 4 fma 4 rcp)/12", UNROLL * 3);
    return 0;
}
