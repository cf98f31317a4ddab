// the per-layer chain's exact traffic: launch 0 reads zin (256 B rows) and writes zbuf (full rows); launches 1..6 read zbuf rows and
// write one half in place; launch 7 reads zbuf and writes 4 B per row.  + 4 B/row log-det read and write per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT_ST, bool NT_LD>
__global__ void __launch_bounds__(512) k(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ ld, long nrows, int mode, int half, int reverse) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long ngroups = nrows / 32;
    const long per_block = (ngroups + gridDim.x - 1) / gridDim.x;
    long g = blockIdx.x * per_block + wave;
    const long gstep = nw, gend = (blockIdx.x + 1) * per_block < ngroups ? (blockIdx.x + 1) * per_block : ngroups;
    f4 buf[8]; float l0 = 0.f;
    auto ldg = [&](long gg) {
        const long ga = reverse ? ngroups - 1 - gg : gg;
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            const f4* p = reinterpret_cast<const f4*>(src + ga * 2048 + k2 * 256 + lane * 4);
            buf[k2] = NT_LD ? __builtin_nontemporal_load(p) : *p;
        }
        l0 = ld[ga * 32 + (lane & 31)];
    };
    if (g < gend) ldg(g);
    for (; g < gend; g += gstep) {
        f4 v[8];
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) v[k2] = buf[k2];
        const float lv = l0;
        if (g + gstep < gend) ldg(g + gstep);
        const long ga = reverse ? ngroups - 1 - g : g;
        if (mode == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                float* p = dst + ga * 2048 + k2 * 256 + lane * 4;
                f4 o = v[k2] * 0.999f;
                if (NT_ST) __builtin_nontemporal_store(o, reinterpret_cast<f4*>(p)); else *reinterpret_cast<f4*>(p) = o;
            }
        } else if (mode == 1) {
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                const int r = k2 * 8 + lane / 8, c = (lane % 8) * 4 + half * 32;
                f4 o = v[2 * k2] * 0.999f + v[2 * k2 + 1] * 0.001f;
                float* p = dst + (ga * 32 + r) * 64 + c;
                if (NT_ST) __builtin_nontemporal_store(o, reinterpret_cast<f4*>(p)); else *reinterpret_cast<f4*>(p) = o;
            }
        }
        float acc = lv;
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) acc += v[k2][0];
        if (lane < 32) ld[ga * 32 + lane] = acc;
    }
}
template <bool NT_ST, bool NT_LD>
static void run(const char* name, float* zin, float* zbuf, float* ld, long nrows, int alt) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto chain = [&]() {
        for (int l = 0; l < 8; ++l) {
            const int rev = alt ? (l & 1) : 0;
            if (l == 0) hipLaunchKernelGGL((k<NT_ST, NT_LD>), dim3(256), dim3(512), 0, 0, zin, zbuf, ld, nrows, 0, 0, rev);
            else if (l < 7) hipLaunchKernelGGL((k<NT_ST, NT_LD>), dim3(256), dim3(512), 0, 0, zbuf, zbuf, ld, nrows, 1, l & 1, rev);
            else hipLaunchKernelGGL((k<NT_ST, NT_LD>), dim3(256), dim3(512), 0, 0, zbuf, zbuf, ld, nrows, 2, 0, rev);
        }
    };
    for (int i = 0; i < 20; ++i) chain();
    (void)hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) chain();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= reps;
    printf("%-22s alternate %d: chain %.3f ms  %.1f us/launch  algorithmic(3900 B/row) %.2f TB/s = %.3f of 8\n", name, alt, ms, ms * 125, nrows * 3900.0 / ms / 1e9, nrows * 3900.0 / ms / 8e9);
}
int main(int argc, char** argv) {
    const long nrows = argc > 1 ? atol(argv[1]) : (1L << 20);
    float *zin, *zbuf, *ld;
    (void)hipMalloc(&zin, nrows * 256); (void)hipMalloc(&zbuf, nrows * 256); (void)hipMalloc(&ld, nrows * 4);
    (void)hipMemset(zin, 0, nrows * 256); (void)hipMemset(zbuf, 0, nrows * 256); (void)hipMemset(ld, 0, nrows * 4);
    for (int alt = 0; alt < 2; ++alt) {
        run<true, false>("nt stores", zin, zbuf, ld, nrows, alt);
        run<false, false>("plain stores", zin, zbuf, ld, nrows, alt);
        run<false, true>("plain st, nt loads", zin, zbuf, ld, nrows, alt);
        run<true, true>("nt st, nt loads", zin, zbuf, ld, nrows, alt);
    }
    return 0;
}
