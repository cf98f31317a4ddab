// Support layers: parameter-free bijectors appended to a NormFlow to constrain its support
// (NormFlow(..., support_layer=...), density_estimator.py:278-282).
//   ToInterval (bijectors.py:429-557): per feature tanh (two-sided bounds), softplus (one-sided) or identity
//   ToSimplex  (bijectors.py:560-594): (rows, D_in) -> (rows, D_in + 1) on the simplex; forward only
// Both are HBM-bound elementwise maps with a per-row log-det: a workgroup stages R rows through LDS so
// that every global access is coalesced (thread <-> element) and the row reduction runs thread <-> row.
// Algorithmic bytes per row: ToInterval 2*D*sizeof(T) + sizeof(T); ToSimplex (2*D_in + 2)*sizeof(T).
#include "tnf_common.h"

namespace tnf {

template <typename T> struct Mth;
template <> struct Mth<float> {
    static __device__ __forceinline__ float tanh(float x) { return tanhf(x); }
    static __device__ __forceinline__ float exp(float x) { return expf(x); }
    static __device__ __forceinline__ float log(float x) { return logf(x); }
    static __device__ __forceinline__ float log1p(float x) { return log1pf(x); }
    static __device__ __forceinline__ float abs(float x) { return fabsf(x); }
};
template <> struct Mth<double> {
    static __device__ __forceinline__ double tanh(double x) { return ::tanh(x); }
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double log1p(double x) { return ::log1p(x); }
    static __device__ __forceinline__ double abs(double x) { return fabs(x); }
};

// torch.nn.functional.softplus (beta 1, threshold 20) and logsigmoid as torch evaluates them
template <typename T>
__device__ __forceinline__ T softplus_t(T x) { return x > (T)20 ? x : Mth<T>::log1p(Mth<T>::exp(x)); }
template <typename T>
__device__ __forceinline__ T logsigmoid_t(T x) { return (x < 0 ? x : (T)0) - Mth<T>::log1p(Mth<T>::exp(-Mth<T>::abs(x))); }
template <typename T>
__device__ __forceinline__ T sigmoid_t(T x) { return (T)1 / ((T)1 + Mth<T>::exp(-x)); }

// consts rows (float, D each): 0 tanh_flg, 1 softplus_flg, 2 tanh_m, 3 tanh_c, 4 softplus_m, 5 softplus_c,
// 6 log(tanh_m) as the reference rounds it (float32 log of the float32 tanh_m)
enum { IV_TF = 0, IV_SF, IV_TM, IV_TC, IV_SM, IV_SC, IV_LTM, IV_ROWS };
#define TNF_IV_EPS 1e-12

// value and log-det term of one element; when GRAD also d out/d in and d ld/d in
template <typename T, bool INV, bool GRAD>
__device__ __forceinline__ void interval_elem(T x, const float* __restrict__ c, int D, int d, T& out, T& ld, T& dout,
                                              T& dld) {
    const T eps = (T)TNF_IV_EPS;
    out = x;
    ld = 0;
    dout = 1;
    dld = 0;
    if (c[IV_TF * D + d] != 0.0f) {
        const T tm = (T)c[IV_TM * D + d], tc = (T)c[IV_TC * D + d], ltm = (T)c[IV_LTM * D + d];
        T zi = x, dzi = 1;
        if (INV) {  // torch_atanh, bijectors.py:555-557
            const T u = (x - tc) / tm;
            zi = (T)0.5 * (Mth<T>::log((T)1 + u + eps) - Mth<T>::log((T)1 - u + eps));
            if (GRAD) dzi = (T)0.5 * ((T)1 / ((T)1 + u + eps) + (T)1 / ((T)1 - u + eps)) / tm;
        }
        const T t = Mth<T>::tanh(zi);
        const T omt = (T)1 - t * t;
        ld = ltm + Mth<T>::log(omt + eps);
        out = INV ? zi : tm * t + tc;
        if (GRAD) {
            const T dl = (T)-2 * t * omt / (omt + eps);
            dout = INV ? dzi : tm * omt;
            dld = dl * dzi;
        }
    } else if (c[IV_SF * D + d] != 0.0f) {
        const T sm = (T)c[IV_SM * D + d], sc = (T)c[IV_SC * D + d];
        if (INV) {
            const T e = Mth<T>::exp((x - sc) / sm);
            const T zi = Mth<T>::log(e - (T)1 + eps);
            out = zi;
            ld = logsigmoid_t<T>(zi);
            if (GRAD) {
                dout = e / (e - (T)1 + eps) / sm;
                dld = ((T)1 - sigmoid_t<T>(zi)) * dout;
            }
        } else {
            out = sm * softplus_t<T>(x) + sc;
            ld = logsigmoid_t<T>(x);
            if (GRAD) {
                const T s = sigmoid_t<T>(x);
                dout = sm * (x > (T)20 ? (T)1 : s);
                dld = (T)1 - s;
            }
        }
    }
}

template <typename T, bool INV>
__global__ void __launch_bounds__(256)
to_interval_kernel(const T* __restrict__ z, const float* __restrict__ consts, T* __restrict__ z_out,
                   T* __restrict__ log_det, int64_t rows, int D, int R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);  // [R][D + 1] log-det terms
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * R;
    const int nr = (int)((rows - r0) < (int64_t)R ? (rows - r0) : (int64_t)R);
    const T* zt = z + r0 * D;
    T* zo = z_out + r0 * D;
    for (int idx = tid; idx < nr * D; idx += 256) {
        const int i = idx / D, d = idx - i * D;
        T out, ld, a, b;
        interval_elem<T, INV, false>(zt[idx], consts, D, d, out, ld, a, b);
        zo[idx] = out;
        lds[i * (D + 1) + d] = ld;
    }
    __syncthreads();
    for (int i = tid; i < nr; i += 256) {
        T acc = 0;
        for (int d = 0; d < D; ++d) acc += lds[i * (D + 1) + d];
        log_det[r0 + i] = acc;
    }
}

// g_z = g_zout * d out/d z + g_ld[row] * d ld/d z, recomputed from the layer's input
template <typename T, bool INV>
__global__ void __launch_bounds__(256)
to_interval_backward_kernel(const T* __restrict__ z, const float* __restrict__ consts, const T* __restrict__ g_zout,
                            const T* __restrict__ g_ld, T* __restrict__ g_z, int64_t total, int D) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int64_t row = idx / D;
    const int d = (int)(idx - row * D);
    T out, ld, dout, dld;
    interval_elem<T, INV, true>(z[idx], consts, D, d, out, ld, dout, dld);
    g_z[idx] = g_zout[idx] * dout + g_ld[row] * dld;
}

static int rows_per_block(int width, size_t esz, int planes) {
    int64_t R = (int64_t)(32 * 1024) / (int64_t)((size_t)planes * (size_t)(width + 1) * esz);
    if (R > 256) R = 256;
    if (R < 1) R = 1;
    return (int)R;
}

int launch_to_interval(int dtype, const void* z, const float* consts, void* z_out, void* log_det, int64_t rows, int D,
                       int inverse, hipStream_t st) {
    if (rows == 0) return 0;
    const size_t esz = dtype == TNF_F64 ? 8 : 4;
    const int R = rows_per_block(D, esz, 1);
    const size_t smem = (size_t)R * (D + 1) * esz;
    if (smem > 64 * 1024) return fail(TNF_EUNSUPPORTED, "to_interval: D=%d needs %zu B of LDS per row", D, smem);
    const int64_t blocks = (rows + R - 1) / R;
    if (blocks > 0x7fffffff) return fail(TNF_EUNSUPPORTED, "to_interval: grid too large");
#define TNF_IV_LAUNCH(T, INV)                                                                                   \
    hipLaunchKernelGGL((to_interval_kernel<T, INV>), dim3((unsigned)blocks), dim3(256), smem, st, (const T*)z, \
                       consts, (T*)z_out, (T*)log_det, rows, D, R)
    if (dtype == TNF_F32) {
        if (inverse) TNF_IV_LAUNCH(float, true); else TNF_IV_LAUNCH(float, false);
    } else {
        if (inverse) TNF_IV_LAUNCH(double, true); else TNF_IV_LAUNCH(double, false);
    }
#undef TNF_IV_LAUNCH
    return check_launch("to_interval");
}

int launch_to_interval_backward(int dtype, const void* z, const float* consts, const void* g_zout, const void* g_ld,
                                void* g_z, int64_t rows, int D, int inverse, hipStream_t st) {
    const int64_t total = rows * D;
    if (total == 0) return 0;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffff) return fail(TNF_EUNSUPPORTED, "to_interval_backward: grid too large");
#define TNF_IV_LAUNCH(T, INV)                                                                                  \
    hipLaunchKernelGGL((to_interval_backward_kernel<T, INV>), dim3((unsigned)blocks), dim3(256), 0, st,       \
                       (const T*)z, consts, (const T*)g_zout, (const T*)g_ld, (T*)g_z, total, D)
    if (dtype == TNF_F32) {
        if (inverse) TNF_IV_LAUNCH(float, true); else TNF_IV_LAUNCH(float, false);
    } else {
        if (inverse) TNF_IV_LAUNCH(double, true); else TNF_IV_LAUNCH(double, false);
    }
#undef TNF_IV_LAUNCH
    return check_launch("to_interval_backward");
}

// ---------------------------------------------------------------------------
// ToSimplex.forward_and_log_det (bijectors.py:574-591), literal:
//   ex = exp(z); S = sum ex; den = S + 1
//   log_det = log(1 - S/den + 1e-10) - Dc*log(den) + sum z       (Dc = the bijector's own D attribute)
//   out = [ex/den, 1/den]
// ---------------------------------------------------------------------------
#define TNF_SX_EPS 1e-10

template <typename T>
__global__ void __launch_bounds__(256)
to_simplex_kernel(const T* __restrict__ z, T* __restrict__ z_out, T* __restrict__ log_det, int64_t rows, int Din,
                  int Dc, int R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* ex = reinterpret_cast<T*>(smem_raw);  // [R][Din + 1]; column Din holds 1/den
    const int tid = threadIdx.x;
    const int W = Din + 1;
    const int64_t r0 = (int64_t)blockIdx.x * R;
    const int nr = (int)((rows - r0) < (int64_t)R ? (rows - r0) : (int64_t)R);
    const T* zt = z + r0 * Din;
    for (int idx = tid; idx < nr * Din; idx += 256) {
        const int i = idx / Din, d = idx - i * Din;
        ex[i * W + d] = zt[idx];
    }
    __syncthreads();
    for (int i = tid; i < nr; i += 256) {
        T S = 0, sz = 0;
        for (int d = 0; d < Din; ++d) {
            const T v = ex[i * W + d];
            const T e = Mth<T>::exp(v);
            ex[i * W + d] = e;
            S += e;
            sz += v;
        }
        const T den = S + (T)1;
        log_det[r0 + i] = Mth<T>::log((T)1 - S / den + (T)TNF_SX_EPS) - (T)Dc * Mth<T>::log(den) + sz;
        ex[i * W + Din] = den;
    }
    __syncthreads();
    T* zo = z_out + r0 * W;
    for (int idx = tid; idx < nr * W; idx += 256) {
        const int i = idx / W, d = idx - i * W;
        const T den = ex[i * W + Din];
        zo[idx] = (d < Din ? ex[idx] : (T)1) / den;
    }
}

// g_z_j = o_j (g_j - sum_i g_i o_i) + g_ld (1 - Dc o_j - o_j (1-r)/(1-r+eps)),  o = ex/den, r = S/den,
// the sum running over all Din+1 outputs (o_last = 1/den)
template <typename T>
__global__ void __launch_bounds__(256)
to_simplex_backward_kernel(const T* __restrict__ z, const T* __restrict__ g_zout, const T* __restrict__ g_ld,
                           T* __restrict__ g_z, int64_t rows, int Din, int Dc, int R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int W = Din + 1;
    T* ex = reinterpret_cast<T*>(smem_raw);      // [R][W]: exp(z); column Din: den
    T* gg = ex + (int64_t)R * W;                 // [R][W]: g_zout; afterwards column Din: sum_i g_i o_i
    T* rr = gg + (int64_t)R * W;                 // [R]: r = S/den
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * R;
    const int nr = (int)((rows - r0) < (int64_t)R ? (rows - r0) : (int64_t)R);
    const T* zt = z + r0 * Din;
    const T* gt = g_zout + r0 * W;
    for (int idx = tid; idx < nr * Din; idx += 256) {
        const int i = idx / Din, d = idx - i * Din;
        ex[i * W + d] = Mth<T>::exp(zt[idx]);
    }
    for (int idx = tid; idx < nr * W; idx += 256) gg[idx] = gt[idx];
    __syncthreads();
    for (int i = tid; i < nr; i += 256) {
        T S = 0, dot = 0;
        for (int d = 0; d < Din; ++d) {
            S += ex[i * W + d];
            dot += gg[i * W + d] * ex[i * W + d];
        }
        const T den = S + (T)1;
        dot = (dot + gg[i * W + Din]) / den;
        ex[i * W + Din] = den;
        gg[i * W + Din] = dot;
        rr[i] = S / den;
    }
    __syncthreads();
    T* gz = g_z + r0 * Din;
    for (int idx = tid; idx < nr * Din; idx += 256) {
        const int i = idx / Din, d = idx - i * Din;
        const T o = ex[i * W + d] / ex[i * W + Din];
        const T omr = (T)1 - rr[i];
        gz[idx] = o * (gg[i * W + d] - gg[i * W + Din]) +
                  g_ld[r0 + i] * ((T)1 - (T)Dc * o - o * omr / (omr + (T)TNF_SX_EPS));
    }
}

int launch_to_simplex(int dtype, const void* z, void* z_out, void* log_det, int64_t rows, int Din, int Dc,
                      hipStream_t st) {
    if (rows == 0) return 0;
    const size_t esz = dtype == TNF_F64 ? 8 : 4;
    const int R = rows_per_block(Din, esz, 1);
    const size_t smem = (size_t)R * (Din + 1) * esz;
    if (smem > 64 * 1024) return fail(TNF_EUNSUPPORTED, "to_simplex: D=%d needs %zu B of LDS per row", Din, smem);
    const int64_t blocks = (rows + R - 1) / R;
    if (blocks > 0x7fffffff) return fail(TNF_EUNSUPPORTED, "to_simplex: grid too large");
    if (dtype == TNF_F32)
        hipLaunchKernelGGL(to_simplex_kernel<float>, dim3((unsigned)blocks), dim3(256), smem, st, (const float*)z,
                           (float*)z_out, (float*)log_det, rows, Din, Dc, R);
    else
        hipLaunchKernelGGL(to_simplex_kernel<double>, dim3((unsigned)blocks), dim3(256), smem, st, (const double*)z,
                           (double*)z_out, (double*)log_det, rows, Din, Dc, R);
    return check_launch("to_simplex");
}

int launch_to_simplex_backward(int dtype, const void* z, const void* g_zout, const void* g_ld, void* g_z, int64_t rows,
                               int Din, int Dc, hipStream_t st) {
    if (rows == 0) return 0;
    const size_t esz = dtype == TNF_F64 ? 8 : 4;
    const int R = rows_per_block(Din, esz, 3);
    const size_t smem = ((size_t)2 * R * (Din + 1) + R) * esz;
    if (smem > 64 * 1024)
        return fail(TNF_EUNSUPPORTED, "to_simplex_backward: D=%d needs %zu B of LDS per row", Din, smem);
    const int64_t blocks = (rows + R - 1) / R;
    if (blocks > 0x7fffffff) return fail(TNF_EUNSUPPORTED, "to_simplex_backward: grid too large");
    if (dtype == TNF_F32)
        hipLaunchKernelGGL(to_simplex_backward_kernel<float>, dim3((unsigned)blocks), dim3(256), smem, st,
                           (const float*)z, (const float*)g_zout, (const float*)g_ld, (float*)g_z, rows, Din, Dc, R);
    else
        hipLaunchKernelGGL(to_simplex_backward_kernel<double>, dim3((unsigned)blocks), dim3(256), smem, st,
                           (const double*)z, (const double*)g_zout, (const double*)g_ld, (double*)g_z, rows, Din, Dc,
                           R);
    return check_launch("to_simplex_backward");
}

}  // namespace tnf
