// Shape- and dtype-generic kernels (any D, U, L; float32 / float64; M broadcast).
// These are the coverage path: every shape the reference accepts runs here when
// no MFMA specialisation exists (coupling_mfma.hip / flow_fused.hip hold those).
#include "tnf_common.h"

namespace tnf {

// ---------------------------------------------------------------------------
// RealNVP coupling layer, generic.
// One workgroup = one parameter row m and a tile of TS samples.  Activations of
// the twin t/s MLP ping-pong through LDS; weights stream from global (L2).
// Reference: bijectors.py:145-242.
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T tnf_tanh(T x);
template <>
__device__ __forceinline__ float tnf_tanh<float>(float x) { return tanhf(x); }
template <>
__device__ __forceinline__ double tnf_tanh<double>(double x) { return tanh(x); }
template <typename T>
__device__ __forceinline__ T tnf_exp(T x);
template <>
__device__ __forceinline__ float tnf_exp<float>(float x) { return expf(x); }
template <>
__device__ __forceinline__ double tnf_exp<double>(double x) { return exp(x); }

template <typename T>
__global__ void __launch_bounds__(256)
coupling_generic_kernel(const T* __restrict__ z, const T* __restrict__ params, T* __restrict__ z_out,
                        T* __restrict__ log_det, int64_t Mz, int64_t Mp, int64_t N, int D, int L,
                        int U, int upper, int inverse, int64_t pstride, int ld_mode, int TS, int W) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* buf = reinterpret_cast<T*>(smem_raw);  // [net][pingpong][TS][W]
    const int tid = threadIdx.x;
    const int64_t m = grid_m();
    if (m >= (Mz > Mp ? Mz : Mp)) return;
    const int64_t n0 = (int64_t)blockIdx.x * TS;
    const int ts = (int)((N - n0) < (int64_t)TS ? (N - n0) : (int64_t)TS);
    const int h = D / 2;
    const CouplingDims cd = coupling_dims(D, upper);
    const int c_off = upper ? 0 : h;  // conditioner half z1 (bijectors.py:157-160)
    const int t_off = upper ? h : 0;  // transformed half z2
    const T* zt = z + ((Mz == 1 ? 0 : m) * N + n0) * D;
    T* zo = z_out + (m * N + n0) * D;
    const T* p = params + (Mp == 1 ? 0 : m) * pstride;
    const int64_t plane = (int64_t)TS * W;
    T* bt[2] = {buf, buf + plane};
    T* bs[2] = {buf + 2 * plane, buf + 3 * plane};

    for (int idx = tid; idx < ts * cd.d_in; idx += 256) {
        const int i = idx / cd.d_in, k = idx - i * cd.d_in;
        const T v = zt[(int64_t)i * D + c_off + k];
        bt[0][i * W + k] = v;
        zo[(int64_t)i * D + c_off + k] = v;  // pass-through half, bit-identical
    }
    __syncthreads();

    int cur = 0;
    for (int l = 0; l <= L; ++l) {
        const int din = (l == 0) ? cd.d_in : U;
        const int dout = (l == L) ? cd.d_out : U;
        const int64_t nw = (int64_t)din * dout;
        const T* wt = p;
        const T* ws = p + nw;
        const T* bias_t = p + 2 * nw;
        const T* bias_s = bias_t + dout;
        p = bias_s + dout;
        const T* xt_base = bt[cur];
        const T* xs_base = (l == 0) ? bt[cur] : bs[cur];  // both nets read z1 at layer 0 (:168)
        for (int idx = tid; idx < ts * dout; idx += 256) {
            const int i = idx / dout, o = idx - i * dout;
            const T* xt = xt_base + i * W;
            const T* xs = xs_base + i * W;
            T acc_t = 0, acc_s = 0;
            for (int k = 0; k < din; ++k) {
                acc_t += xt[k] * wt[(int64_t)k * dout + o];
                acc_s += xs[k] * ws[(int64_t)k * dout + o];
            }
            acc_t += bias_t[o];
            acc_s += bias_s[o];
            if (l < L) {
                acc_t = tnf_tanh<T>(acc_t);
                acc_s = tnf_tanh<T>(acc_s);
            }
            bt[cur ^ 1][i * W + o] = acc_t;
            bs[cur ^ 1][i * W + o] = acc_s;
        }
        __syncthreads();
        cur ^= 1;
    }

    for (int idx = tid; idx < ts * cd.d_out; idx += 256) {
        const int i = idx / cd.d_out, o = idx - i * cd.d_out;
        const T t = bt[cur][i * W + o];
        const T s = bs[cur][i * W + o];
        const T z2 = zt[(int64_t)i * D + t_off + o];
        const T e = tnf_exp<T>(s);
        zo[(int64_t)i * D + t_off + o] = inverse ? (z2 - t) / e : t + z2 * e;
    }
    if (tid < ts) {
        T acc = 0;
        for (int o = 0; o < cd.d_out; ++o) acc += bs[cur][tid * W + o];
        T* ld = log_det + m * N + n0 + tid;
        if (ld_mode == TNF_LD_STORE) *ld = acc;
        else if (ld_mode == TNF_LD_ADD) *ld += acc;
        else *ld -= acc;
    }
}

int launch_coupling_generic(int dtype, const void* z, const void* params, void* z_out,
                            void* log_det, int64_t Mz, int64_t Mp, int64_t N, int D, int L, int U,
                            int upper, int inverse, int64_t pstride, int ld_mode, hipStream_t st) {
    const int64_t M = Mz > Mp ? Mz : Mp;
    const CouplingDims cd = coupling_dims(D, upper);
    int W = cd.d_in > cd.d_out ? cd.d_in : cd.d_out;
    if (U > W) W = U;
    const size_t esz = dtype == TNF_F64 ? 8 : 4;
    int64_t TS = (int64_t)(64 * 1024) / (int64_t)(4 * (size_t)W * esz);
    if (TS > 64) TS = 64;
    if (TS > N) TS = N;
    if (TS < 1) TS = 1;
    const size_t smem = (size_t)4 * TS * W * esz;
    if (smem > 160 * 1024) return fail(TNF_EUNSUPPORTED, "coupling: layer width %d needs %zu B of LDS", W, smem);
    const int64_t tiles = (N + TS - 1) / TS;
    if (tiles > 0x7fffffff)
        return fail(TNF_EUNSUPPORTED, "coupling: grid too large (tiles=%lld, M=%lld)", (long long)tiles, (long long)M);
    const dim3 grid = grid_xm(tiles, M);
    if (dtype == TNF_F32) {
        if (smem > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)coupling_generic_kernel<float>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(coupling_generic_kernel<float>, grid, dim3(256), smem, st,
                           (const float*)z, (const float*)params, (float*)z_out, (float*)log_det, Mz,
                           Mp, N, D, L, U, upper, inverse, pstride, ld_mode, (int)TS, W);
    } else {
        if (smem > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)coupling_generic_kernel<double>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(coupling_generic_kernel<double>, grid, dim3(256), smem, st,
                           (const double*)z, (const double*)params, (double*)z_out, (double*)log_det,
                           Mz, Mp, N, D, L, U, upper, inverse, pstride, ld_mode, (int)TS, W);
    }
    return check_launch("coupling_generic");
}

// ---------------------------------------------------------------------------
// Affine (bijectors.py:277-315): elementwise + per-row sum(alpha).
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
affine_kernel(const T* __restrict__ z, const T* __restrict__ params, T* __restrict__ z_out,
              int64_t Mz, int64_t Mp, int64_t N, int D, int inverse, int64_t pstride, int64_t total) {
    const int64_t ND = N * D;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * 256) {
        const int64_t m = idx / ND;
        const int64_t rem = idx - m * ND;
        const int d = (int)(rem % D);
        const T* p = params + (Mp == 1 ? 0 : m) * pstride;
        const T scale = tnf_exp<T>(p[d]);
        const T shift = p[D + d];
        const T v = z[(Mz == 1 ? 0 : m) * ND + rem];
        z_out[idx] = inverse ? (v - shift) / scale : scale * v + shift;
    }
}

template <typename T>
__global__ void __launch_bounds__(64)
affine_logdet_kernel(const T* __restrict__ params, T* __restrict__ log_det, int D, int64_t pstride) {
    const T* p = params + (int64_t)blockIdx.x * pstride;
    T acc = 0;
    for (int d = threadIdx.x; d < D; d += 64) acc += p[d];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (threadIdx.x == 0) log_det[blockIdx.x] = acc;
}

int launch_affine(int dtype, const void* z, const void* params, void* z_out, void* log_det,
                  int64_t Mz, int64_t Mp, int64_t N, int D, int inverse, int64_t pstride,
                  hipStream_t st) {
    const int64_t M = Mz > Mp ? Mz : Mp;
    const int64_t total = M * N * D;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (dtype == TNF_F32) {
        if (total > 0)
            hipLaunchKernelGGL(affine_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st,
                               (const float*)z, (const float*)params, (float*)z_out, Mz, Mp, N, D,
                               inverse, pstride, total);
        hipLaunchKernelGGL(affine_logdet_kernel<float>, dim3((unsigned)Mp), dim3(64), 0, st,
                           (const float*)params, (float*)log_det, D, pstride);
    } else {
        if (total > 0)
            hipLaunchKernelGGL(affine_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st,
                               (const double*)z, (const double*)params, (double*)z_out, Mz, Mp, N, D,
                               inverse, pstride, total);
        hipLaunchKernelGGL(affine_logdet_kernel<double>, dim3((unsigned)Mp), dim3(64), 0, st,
                           (const double*)params, (double*)log_det, D, pstride);
    }
    return check_launch("affine");
}

// ---------------------------------------------------------------------------
// BatchNorm with cached statistics (bijectors.py:397-399, 420-426).
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
bn_apply_kernel(const T* __restrict__ z, const float* __restrict__ mean,
                const float* __restrict__ alpha, T* __restrict__ z_out, int D, int inverse,
                int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * 256) {
        const int d = (int)(idx % D);
        const T a = (T)alpha[d];
        const T mu = (T)mean[d];
        const T v = z[idx];
        z_out[idx] = inverse ? v * a + mu : (v - mu) / a;
    }
}

__global__ void __launch_bounds__(64)
bn_logdet_kernel(const float* __restrict__ alpha, float* __restrict__ log_det, int D) {
    float acc = 0.f;
    for (int d = threadIdx.x; d < D; d += 64) acc += logf(alpha[d]);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (threadIdx.x == 0) *log_det = -acc;
}

int launch_bn_apply(int dtype, const void* z, const float* mean, const float* alpha, void* z_out,
                    float* log_det, int64_t rows, int D, int inverse, hipStream_t st) {
    const int64_t total = rows * D;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (total > 0) {
        if (dtype == TNF_F32)
            hipLaunchKernelGGL(bn_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st,
                               (const float*)z, mean, alpha, (float*)z_out, D, inverse, total);
        else
            hipLaunchKernelGGL(bn_apply_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st,
                               (const double*)z, mean, alpha, (double*)z_out, D, inverse, total);
    }
    hipLaunchKernelGGL(bn_logdet_kernel, dim3(1), dim3(64), 0, st, alpha, log_det, D);
    return check_launch("bn_apply");
}

// ---------------------------------------------------------------------------
// BatchNorm with batch statistics (bijectors.py:401-417), float32 data.
// Pass 1: per-feature sum / sum-of-squares in float64 (block partials, one
//         double atomic per feature per block).
// Pass 2: one block turns the sums into mean / alpha / 1/alpha / log_det.
// Pass 3: elementwise normalise.
// In exact arithmetic the reference's cached statistics reduce to
//   alpha = sqrt(var_biased + eps),  mean = mu
// (alpha = sqrt(var_u(z)) / sqrt(var_u(z)/(var_b+eps)); mean(z - z_norm*alpha) = mu);
// the kernels compute those directly.
// workspace (doubles): [sum (D) | sumsq (D)] then floats [rstd (D)].
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
bn_stats_kernel(const float* __restrict__ z, double* __restrict__ sums, int64_t rows, int D,
                int64_t rows_per_block) {
    __shared__ double red1[256];
    __shared__ double red2[256];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (int dc = 0; dc < D; dc += 256) {
        const int Dc = (D - dc) < 256 ? (D - dc) : 256;
        const int rpi = 256 / Dc;  // rows handled per iteration
        const int r = tid / Dc, d = tid - r * Dc;
        double s1 = 0.0, s2 = 0.0;
        if (r < rpi) {
            for (int64_t row = r0 + r; row < r1; row += rpi) {
                const double v = (double)z[row * D + dc + d];
                s1 += v;
                s2 += v * v;
            }
        }
        red1[tid] = s1;
        red2[tid] = s2;
        __syncthreads();
        if (tid < Dc) {
            double a = 0.0, b = 0.0;
            for (int rr = 0; rr < rpi; ++rr) {
                a += red1[rr * Dc + tid];
                b += red2[rr * Dc + tid];
            }
            atomicAdd(&sums[dc + tid], a);
            atomicAdd(&sums[D + dc + tid], b);
        }
        __syncthreads();
    }
}

// D % 4 == 0 and D <= 1024: 16-byte loads, D/4 lanes per row, float partial sums over short runs of rows
// (64 values) folded into double accumulators -- same sums as bn_stats_kernel to ~1e-7 relative.
__global__ void __launch_bounds__(256)
bn_stats_vec_kernel(const float* __restrict__ z, double* __restrict__ sums, int64_t rows, int D,
                    int64_t rows_per_block, int write_count) {
    extern __shared__ double red[];  // [rpi][2][D]
    const int tid = threadIdx.x;
    const int lanes = D >> 2;            // threads per row
    const int rpi = 256 / lanes;         // rows per iteration
    const int r = tid / lanes, q = tid - r * lanes;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    if (r < rpi) {
        int64_t row = r0 + r;
        while (row < r1) {
            float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < 8 && row < r1; ++k, row += 8 * (int64_t)rpi) {
                float4 v[8];  // eight rows in flight per lane
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t rw = row + u * (int64_t)rpi;
                    v[u] = rw < r1 ? *reinterpret_cast<const float4*>(z + rw * D + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a1[0] += v[u].x; a1[1] += v[u].y; a1[2] += v[u].z; a1[3] += v[u].w;
                    a2[0] = fmaf(v[u].x, v[u].x, a2[0]); a2[1] = fmaf(v[u].y, v[u].y, a2[1]);
                    a2[2] = fmaf(v[u].z, v[u].z, a2[2]); a2[3] = fmaf(v[u].w, v[u].w, a2[3]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s1[j] += (double)a1[j];
                s2[j] += (double)a2[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[(r * 2 + 0) * D + 4 * q + j] = s1[j];
            red[(r * 2 + 1) * D + 4 * q + j] = s2[j];
        }
    }
    __syncthreads();
    for (int d = tid; d < 2 * D; d += 256) {
        double a = 0.0;
        for (int rr = 0; rr < rpi; ++rr) a += red[rr * 2 * D + d];
        atomicAdd(&sums[d], a);
    }
    if (write_count && blockIdx.x == 0 && tid == 0) sums[2 * D] = (double)rows;  // moments = [sums | sums of squares | count]
}

__global__ void __launch_bounds__(256)
bn_finalize_kernel(const double* __restrict__ sums, float* __restrict__ mean_out,
                   float* __restrict__ alpha_out, float* __restrict__ rstd, float* __restrict__ log_det,
                   int64_t rows_arg, int D, float eps) {
    __shared__ float red[256];
    float acc = 0.f;
    // rows_arg < 0: the row count rides behind the sums (moments = [sum (D) | sum of squares (D) | count]) -- the
    // count of the GLOBAL batch when the moments were summed over the ranks of a sample-sharded forward
    const double rows = rows_arg < 0 ? sums[2 * D] : (double)rows_arg;
    for (int d = threadIdx.x; d < D; d += 256) {
        const double mu = sums[d] / rows;
        double var_b = sums[D + d] / rows - mu * mu;
        if (var_b < 0.0) var_b = 0.0;
        const double a = sqrt(var_b + (double)eps);
        mean_out[d] = (float)mu;
        alpha_out[d] = (float)a;
        rstd[d] = (float)(1.0 / a);
        acc += logf((float)a);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *log_det = -red[0];
}

__global__ void __launch_bounds__(256)
bn_normalize_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                    const float* __restrict__ rstd, float* __restrict__ z_out, int D, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * 256) {
        const int d = (int)(idx % D);
        z_out[idx] = (z[idx] - mean[d]) * rstd[d];
    }
}

// returns true when the kernel also wrote the row count behind the sums (write_count asked for and the vector kernel ran)
static bool launch_bn_sums(const float* z, double* sums, int64_t rows, int D, int64_t blocks, int64_t rpb, hipStream_t st,
                           int write_count = 0) {
    const bool vec = (D % 4) == 0 && D <= 1024 && (reinterpret_cast<uintptr_t>(z) & 15) == 0;
    if (vec) {
        const int rpi = 256 / (D / 4);
        // a streaming read: eight 16-byte loads per lane in flight, and few enough workgroups (two per CU) that
        // their 2 D double atomics per workgroup do not queue up on the 2 D result words (4096 workgroups: 126 us)
        blocks = (rows + 8 * rpi - 1) / (8 * rpi);
        if (blocks > 512) blocks = 512;
        if (blocks < 1) blocks = 1;
        rpb = (rows + blocks - 1) / blocks;
        hipLaunchKernelGGL(bn_stats_vec_kernel, dim3((unsigned)blocks), dim3(256), (size_t)rpi * 2 * D * sizeof(double), st, z,
                           sums, rows, D, rpb, write_count);
        return write_count != 0;
    }
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, st, z, sums, rows, D, rpb);
    return false;
}

// local moments of z (rows, D) for a batch-statistics BatchNorm whose normalisation is folded into the next kernel by
// the caller: moments = [sum (D) | sum of squares (D) | row count] doubles, overwritten
__global__ void bn_count_kernel(double* __restrict__ moments, int D, double rows) { moments[2 * D] = rows; }

int launch_bn_moments(const float* z, double* moments, int64_t rows, int D, hipStream_t st) {
    if (hipMemsetAsync(moments, 0, sizeof(double) * (2 * (size_t)D + 1), st) != hipSuccess)  // an empty shard: count 0
        return fail(TNF_ELAUNCH, "bn_moments: memset failed");
    bool counted = rows <= 0;
    if (rows > 0) {
        int64_t blocks = (rows + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        const int64_t rpb = (rows + blocks - 1) / blocks;
        counted = launch_bn_sums(z, moments, rows, D, blocks, rpb, st, 1);
    }
    if (!counted) hipLaunchKernelGGL(bn_count_kernel, dim3(1), dim3(1), 0, st, moments, D, (double)rows);
    return check_launch("bn_moments");
}

// mean / alpha / 1/alpha / log-det from moments (the count is read from moments[2 D])
int launch_bn_finalize(const double* moments, float* mean_out, float* alpha_out, float* rstd, float* log_det, int D,
                       float eps, hipStream_t st) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), 0, st, moments, mean_out, alpha_out, rstd, log_det,
                       (int64_t)-1, D, eps);
    return check_launch("bn_finalize");
}

// second half of a sample-sharded batch-statistics forward: statistics from the (all-reduced) moments -- the row count is
// read from moments[2 D] -- then this rank's rows normalised with them.  rstd: D floats of scratch.
int launch_bn_normalize_from_moments(const float* z, const double* moments, float* z_out, float* mean_out,
                                     float* alpha_out, float* log_det, float* rstd, int64_t rows, int D, float eps,
                                     hipStream_t st) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), 0, st, moments, mean_out, alpha_out, rstd, log_det,
                       (int64_t)-1, D, eps);
    const int64_t total = rows * D;
    if (total > 0) {
        int64_t nb = (total + 255) / 256;
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(bn_normalize_kernel, dim3((unsigned)nb), dim3(256), 0, st, z, mean_out, rstd, z_out, D, total);
    }
    return check_launch("bn_normalize_from_moments");
}

int launch_bn_batch_forward(const float* z, float* z_out, float* mean_out, float* alpha_out,
                            float* log_det, int64_t rows, int D, float eps, void* ws, hipStream_t st) {
    double* sums = reinterpret_cast<double*>(ws);
    float* rstd = reinterpret_cast<float*>(sums + 2 * (size_t)D);
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)D, st) != hipSuccess)
        return fail(TNF_ELAUNCH, "bn_batch_forward: memset failed");
    int64_t blocks = (rows + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    const int64_t rpb = (rows + blocks - 1) / blocks;
    launch_bn_sums(z, sums, rows, D, blocks, rpb, st);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), 0, st, sums, mean_out, alpha_out, rstd,
                       log_det, rows, D, eps);
    const int64_t total = rows * D;
    int64_t nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(bn_normalize_kernel, dim3((unsigned)nb), dim3(256), 0, st, z, mean_out, rstd,
                       z_out, D, total);
    return check_launch("bn_batch_forward");
}

// ---------------------------------------------------------------------------
// Base Gaussian log-density in float64 (density_estimator.py:369-372).
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
base_log_density_kernel(const T* __restrict__ omega, double* __restrict__ out, int64_t rows, int D) {
    // 4 lanes per row: coalesced enough for the 256-B rows of D = 64 and exact in float64
    const int sub = threadIdx.x & 3;
    for (int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2; r < rows; r += ((int64_t)gridDim.x * 256) >> 2) {
        const T* w = omega + r * D;
        double acc = 0.0;
        for (int d = sub; d < D; d += 4) acc += (double)w[d] * (double)w[d];
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (sub == 0) out[r] = -0.5 * acc - (double)D * 0.91893853320467274178;
    }
}

int launch_base_log_density(int dtype, const void* omega, double* out, int64_t rows, int D, hipStream_t st) {
    int64_t blocks = (rows * 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (rows > 0) {
        if (dtype == TNF_F32)
            hipLaunchKernelGGL(base_log_density_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st,
                               (const float*)omega, out, rows, D);
        else
            hipLaunchKernelGGL(base_log_density_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st,
                               (const double*)omega, out, rows, D);
    }
    return check_launch("base_log_density");
}

}  // namespace tnf
