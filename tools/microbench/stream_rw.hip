// streaming microbenchmark: read full 256-B rows, write 128-B half rows (like one middle launch of the per-layer chain)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int DEPTH, bool NT_ST>
__global__ void __launch_bounds__(512) k_stream(const float* __restrict__ in, float* __restrict__ out, long nrows, int interleave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long ngroups = nrows / 32;  // 32 rows of 256 B = 8 KB per group
    const long per_block = (ngroups + gridDim.x - 1) / gridDim.x;
    long g, gstep, gend;
    if (interleave) { g = (long)blockIdx.x * nw + wave; gstep = (long)gridDim.x * nw; gend = ngroups; }
    else { g = blockIdx.x * per_block + wave; gstep = nw; gend = (blockIdx.x + 1) * per_block < ngroups ? (blockIdx.x + 1) * per_block : ngroups; }
    f4 buf[DEPTH][8];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const long gg = g + d * gstep;
        if (gg < gend)
#pragma unroll
            for (int k = 0; k < 8; ++k) buf[d][k] = *reinterpret_cast<const f4*>(in + gg * 2048 + k * 256 + lane * 4);
    }
    for (; g < gend; g += DEPTH * gstep) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long gg = g + d * gstep;
            if (gg >= gend) break;
            f4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = buf[d][k];
            const long gn = gg + DEPTH * gstep;
            if (gn < gend)
#pragma unroll
                for (int k = 0; k < 8; ++k) buf[d][k] = *reinterpret_cast<const f4*>(in + gn * 2048 + k * 256 + lane * 4);
            // write the lower half of every row: 4 instructions of 8 half rows (128 B each)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = k * 8 + lane / 8, c = (lane % 8) * 4;
                // take data from the matching loaded registers (any: keep the loads alive)
                f4 o = v[2 * k] + v[2 * k + 1];
                float* p = out + (gg * 32 + r) * 64 + c;
                if (NT_ST) __builtin_nontemporal_store(o, reinterpret_cast<f4*>(p));
                else *reinterpret_cast<f4*>(p) = o;
            }
        }
    }
}
template <int DEPTH, bool NT_ST>
static void run(const char* name, const float* in, float* out, long nrows, int blocks, int threads, int interleave) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_stream<DEPTH, NT_ST>), dim3(blocks), dim3(threads), 0, 0, in, out, nrows, interleave);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<DEPTH, NT_ST>), dim3(blocks), dim3(threads), 0, 0, in, out, nrows, interleave);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    printf("%-28s blocks %5d x %4d thr  interleave %d: %.3f ms  %.2f TB/s actual (read 256 + write 128 B/row)\n", name, blocks, threads, interleave, ms, nrows * 384.0 / ms / 1e9);
}
int main(int argc, char** argv) {
    const long nrows = argc > 1 ? atol(argv[1]) : (1L << 22);
    float *in, *out;
    hipMalloc(&in, nrows * 256); hipMalloc(&out, nrows * 256);
    hipMemset(in, 0, nrows * 256); hipMemset(out, 0, nrows * 256);
    for (int il = 0; il < 2; ++il) {
        run<1, true>("depth1 nt", in, out, nrows, 256, 512, il);
        run<2, true>("depth2 nt", in, out, nrows, 256, 512, il);
        run<1, false>("depth1 plain", in, out, nrows, 256, 512, il);
        run<2, false>("depth2 plain", in, out, nrows, 256, 512, il);
        run<1, true>("depth1 nt 2 WG/CU", in, out, nrows, 512, 512, il);
        run<1, true>("depth1 nt 4 WG/CU", in, out, nrows, 1024, 512, il);
        run<1, false>("depth1 plain 4 WG/CU", in, out, nrows, 1024, 512, il);
        run<2, false>("depth2 plain 4 WG/CU", in, out, nrows, 1024, 512, il);
        run<1, true>("depth1 nt 4 WG/CU x256thr", in, out, nrows, 2048, 256, il);
    }
    return 0;
}
