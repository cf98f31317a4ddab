// microbenchmark: cost of LDS accumulation forms, 8 waves per workgroup, one workgroup per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE, bool PRIVATE>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int reps) {
    __shared__ float acc[8 * 2560];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8 * 2560; i += 512) acc[i] = 0.f;
    __syncthreads();
    float* base = acc + (PRIVATE ? wave * 2560 : 0);
    const int s = lane & 15, q = lane >> 4;
    float v = 1.0f + lane * 1e-3f;
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            float* p = base + t * 240 + s * 15 + 4 * q;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE == 0) atomicAdd(p + j, v);
                else if (MODE == 1) atomicAdd(reinterpret_cast<int*>(p) + j, (int)(v * 1024.f));
                else if (MODE == 2) p[j] += v;            // plain read-modify-write (needs ownership)
                else if (MODE == 3) { }
            }
            if (MODE == 4) {  // 16-byte read-modify-write on a lane-linear image
                float4* p4 = reinterpret_cast<float4*>(base + t * 256) + lane;
                float4 a = *p4; a.x += v; a.y += v; a.z += v; a.w += v; *p4 = a;
            }
            v += 1e-6f;
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = acc[threadIdx.x] + v;
}
template <int MODE, bool PRIVATE>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int reps = 200;
    k<MODE, PRIVATE><<<256, 512>>>(out, cyc, reps);
    k<MODE, PRIVATE><<<256, 512>>>(out, cyc, reps);
    hipDeviceSynchronize();
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= 256;
    // clock64 = s_memtime at 100 MHz; report per wave-instruction in ns
    printf("%-28s %8.1f ticks per 40-op group per wave-set  (%.2f ticks per LDS op per wave)\n", name, avg / reps, avg / reps / 40);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, false>("ds_add_f32 shared");
    run<0, true>("ds_add_f32 private");
    run<1, false>("ds_add_u32 shared");
    run<1, true>("ds_add_u32 private");
    run<2, true>("plain rmw b32 private");
    run<4, true>("plain rmw b128 private");
    run<3, true>("empty");
    return 0;
}
