#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
// mode bits: 1 = waves 0-3 run f32 MFMA loop, 2 = waves 4-7 run VALU fma loop, 4 = waves 4-7 run v_exp loop, 8 = waves 0-3 run bf16 MFMA
__global__ void __launch_bounds__(512) k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  f4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
  float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a+4, v5=a+5, v6=a+6, v7=a+7;
  if (wave < 4) {
    if (mode & 1) {
      for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
      }
    } else if (mode & 8) {
      s8 x = {1,2,3,4,5,6,7,8};
      for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, c3, 0, 0, 0);
      }
    }
  } else {
    if (mode & 2) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v0 = __builtin_fmaf(v0, b, a); v1 = __builtin_fmaf(v1, b, a); v2 = __builtin_fmaf(v2, b, a); v3 = __builtin_fmaf(v3, b, a);
          v4 = __builtin_fmaf(v4, b, a); v5 = __builtin_fmaf(v5, b, a); v6 = __builtin_fmaf(v6, b, a); v7 = __builtin_fmaf(v7, b, a);
        }
      }
    } else if (mode & 4) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          v0 = __builtin_amdgcn_exp2f(v0); v1 = __builtin_amdgcn_exp2f(v1); v2 = __builtin_amdgcn_exp2f(v2); v3 = __builtin_amdgcn_exp2f(v3);
          v4 = __builtin_amdgcn_exp2f(v4); v5 = __builtin_amdgcn_exp2f(v5); v6 = __builtin_amdgcn_exp2f(v6); v7 = __builtin_amdgcn_exp2f(v7);
        }
      }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  const char* names[] = {"f32 MFMA only (4/iter)", "VALU fma only (32/iter)", "f32 MFMA + VALU fma", "v_exp only (16/iter)", "f32 MFMA + v_exp", "bf16 MFMA only", "bf16 MFMA + VALU fma", "bf16 MFMA + v_exp"};
  const int modes[] = {1, 2, 3, 4, 5, 8, 10, 12};
  for (int t = 0; t < 8; ++t) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, 512>>>(out, 100, modes[t]);
    hipEventRecord(e0);
    k<<<256, 512>>>(out, iters, modes[t]);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  (%.1f ns/iter)\n", names[t], ms, ms * 1e6 / iters);
  }
  return 0;
}
