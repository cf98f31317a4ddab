#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
// pattern 0: linear float4 grid-stride copy. pattern 1: MFMA tile pattern (lane (s,q): row s, 16B at q*16 + {0,64,128,192}), NT tiles/wave iter
__global__ void __launch_bounds__(256) copy_lin(const f4* __restrict__ in, f4* __restrict__ out, long n4) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) out[i] = in[i];
}
template <int NT>
__global__ void __launch_bounds__(256) copy_tile(const float* __restrict__ in, float* __restrict__ out, long N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane & 15, q = lane >> 4;
  const long ngroups = N / (16 * NT);
  for (long g = blockIdx.x * 4L + wave; g < ngroups; g += gridDim.x * 4L) {
    f4 v[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float* r = in + ((g * NT + t) * 16 + s) * 64 + 4 * q;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[t][c] = *(const f4*)(r + 16 * c);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float* r = out + ((g * NT + t) * 16 + s) * 64 + 4 * q;
#pragma unroll
      for (int c = 0; c < 4; ++c) *(f4*)(r + 16 * c) = v[t][c];
    }
  }
}
int main() {
  const long N = 1 << 20; const long bytes = N * 64 * 4;
  float *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {1024, 2048, 4096, 8192}) {
    for (int pat = 0; pat < 3; ++pat) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (pat == 0) copy_lin<<<grid, 256>>>((const f4*)a, (f4*)b, bytes / 16);
        else if (pat == 1) copy_tile<1><<<grid, 256>>>(a, b, N);
        else copy_tile<2><<<grid, 256>>>(a, b, N);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("grid %5d pattern %d: %.1f us  %.2f TB/s (read+write)\n", grid, pat, best * 1e3, 2.0 * bytes / best / 1e9);
    }
  }
  return 0;
}
