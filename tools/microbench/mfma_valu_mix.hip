#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
// KIND: 0 = f32 16x16x4, 1 = f16 16x16x16, 2 = f16 16x16x32. Loop body: NM mfma, NV fma, NT exp (independent streams)
template <int KIND, int NM, int NV, int NT>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  h4 x4 = {(_Float16)1.f, (_Float16)2.f, (_Float16)3.f, (_Float16)4.f};
  h8 x8 = {(_Float16)1.f, (_Float16)2.f, (_Float16)3.f, (_Float16)4.f, (_Float16)1.f, (_Float16)2.f, (_Float16)3.f, (_Float16)4.f};
  f4 c[4]; for (int i = 0; i < 4; ++i) c[i] = f4{0, 0, 0, 0};
  float v[8], e[8];
  for (int i = 0; i < 8; ++i) { v[i] = a + i; e[i] = a * 0.1f + i; }
  for (int it = 0; it < iters; ++it) {
    constexpr int G = NM > 0 ? NM : 1;
#pragma unroll
    for (int i = 0; i < G; ++i) {
      if (NM > 0) {
        if (KIND == 0) c[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i & 3], 0, 0, 0);
        else if (KIND == 1) c[i & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(x4, x4, c[i & 3], 0, 0, 0);
        else c[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, x8, c[i & 3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NV / G; ++j) v[(i * (NV / G) + j) & 7] = __builtin_fmaf(v[(i * (NV / G) + j) & 7], b, a);
#pragma unroll
      for (int j = 0; j < NT / G; ++j) e[(i * (NT / G) + j) & 7] = __builtin_amdgcn_exp2f(e[(i * (NT / G) + j) & 7]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0; for (int i = 0; i < 4; ++i) r += c[i][0] + c[i][3]; for (int i = 0; i < 8; ++i) r += v[i] + e[i];
  out[blockIdx.x * 1024 + threadIdx.x] = r;
}
__global__ void denorm_test(float* out) {
  h4 a = {(_Float16)9.5367431640625e-07f, 0, 0, 0};  // 2^-20: f16 subnormal
  h4 b = {(_Float16)1.f, 0, 0, 0};
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
template <int KIND, int NM, int NV, int NT>
void run(const char* name, float* out, int threads) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND, NM, NV, NT><<<256, threads>>>(out, 100);
  hipEventRecord(e0);
  k<KIND, NM, NV, NT><<<256, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s thr %3d: %7.1f cycles/iter @2.4GHz\n", name, threads, ms * 1e-3 / iters * 2.4e9);
}
int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4 + 64);
  denorm_test<<<1, 64>>>(out + 256 * 1024); float d; hipMemcpy(&d, out + 256 * 1024, 4, hipMemcpyDeviceToHost);
  printf("f16 subnormal input through MFMA: %g (expect 9.53674e-07 if kept)\n", d);
  for (int thr : {256, 512, 1024}) {
    run<0, 4, 0, 0>("f32 16x16x4: 4 mfma", out, thr);
    run<1, 4, 0, 0>("f16 16x16x16: 4 mfma", out, thr);
    run<2, 4, 0, 0>("f16 16x16x32: 4 mfma", out, thr);
    run<0, 0, 16, 0>("16 fma", out, thr);
    run<0, 0, 0, 8>("8 exp", out, thr);
    run<0, 4, 16, 0>("f32: 4 mfma + 16 fma", out, thr);
    run<0, 4, 0, 8>("f32: 4 mfma + 8 exp", out, thr);
    run<2, 4, 16, 0>("f16x32: 4 mfma + 16 fma", out, thr);
    run<2, 4, 0, 8>("f16x32: 4 mfma + 8 exp", out, thr);
    run<2, 4, 16, 8>("f16x32: 4 mfma + 16 fma + 8 exp", out, thr);
    run<1, 4, 16, 8>("f16x16: 4 mfma + 16 fma + 8 exp", out, thr);
    run<2, 4, 8, 4>("f16x32: 4 mfma + 8 fma + 4 exp", out, thr);
    run<2, 4, 24, 8>("f16x32: 4 mfma + 24 fma + 8 exp", out, thr);
    run<0, 0, 24, 8>("24 fma + 8 exp", out, thr);
    run<0, 0, 16, 8>("16 fma + 8 exp", out, thr);
    run<0, 4, 16, 8>("f32: 4 mfma + 16 fma + 8 exp", out, thr);
  }
  return 0;
}
