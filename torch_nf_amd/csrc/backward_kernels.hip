// Backward passes of the bijector kernels (autograd of the drop-in classes).
// The reference relies on torch autograd over every aten op of bijectors.py:145-318;
// here each bijector has ONE hand-written backward kernel that recomputes the layer's
// activations from the saved input (nothing but z and the parameters is kept from the
// forward pass) and reduces the parameter gradient over the samples with float atomics
// (one atomic per parameter per workgroup; summation order, hence the last bits, can
// differ from run to run).  Shape- and dtype-generic like generic_kernels.hip.
#include "tnf_common.h"

namespace tnf {

template <typename T>
__device__ __forceinline__ T bw_tanh(T x);
template <>
__device__ __forceinline__ float bw_tanh<float>(float x) { return tanhf(x); }
template <>
__device__ __forceinline__ double bw_tanh<double>(double x) { return tanh(x); }
template <typename T>
__device__ __forceinline__ T bw_exp(T x);
template <>
__device__ __forceinline__ float bw_exp<float>(float x) { return expf(x); }
template <>
__device__ __forceinline__ double bw_exp<double>(double x) { return exp(x); }

// ---------------------------------------------------------------------------
// RealNVP.  Forward (bijectors.py:168-179 / 194-205):
//   t, s = MLP_t(x), MLP_s(x);  fwd: y' = t + y e^s;  inv: y' = (y - t) e^-s;  ld = sum(s)
// Given g_x' (conditioner half of grad z_out), g_y' and g_ld:
//   fwd: dy = g_y' e^s,   dt =  g_y',        ds = g_y' y e^s + g_ld
//   inv: dy = g_y' e^-s,  dt = -g_y' e^-s,   ds = -g_y' y'   + g_ld
// then back through the twin MLPs (tanh' = 1 - h^2), dx = g_x' + W0_t d_t0 + W0_s d_s0.
// One workgroup = one parameter row m and a tile of TS samples.
// LDS: act[net][l][TS][W] (inputs of layer l; l = 0 is x, shared), out[net][TS][W],
//      delta[net][2][TS][W].
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
coupling_backward_kernel(const T* __restrict__ z, const T* __restrict__ params,
                         const T* __restrict__ g_zout, const T* __restrict__ g_ld,
                         T* __restrict__ g_z, T* __restrict__ g_params, int64_t M, int64_t Mp, int64_t N,
                         int D, int L, int U, int upper, int inverse, int64_t pstride, int64_t gpstride,
                         int TS, int W, T* __restrict__ partials, int64_t prow) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x;
    const int64_t m = grid_m();
    if (m >= M) return;
    const int h = D / 2;
    const CouplingDims cd = coupling_dims(D, upper);
    const int c_off = upper ? 0 : h, t_off = upper ? h : 0;
    const int64_t mp = Mp == 1 ? 0 : m;
    const T* p0 = params + mp * pstride;
    // Where this workgroup's parameter-gradient contributions go.  Deterministic mode (`partials`): its own row of the
    // partial buffer -- every element is owned by one thread (the loops below map an index to the same thread for every
    // tile), stored on the workgroup's first tile and read-modify-written afterwards; backward_reduce_kernel adds the
    // rows in workgroup order.  With one workgroup per parameter row the row of g_params itself is that buffer.
    // Legacy mode (no workspace): one atomic per parameter per tile, summation order left to the hardware.
    const bool det = partials != nullptr || prow < 0;
    T* gp0 = partials ? partials + (m * gridDim.x + blockIdx.x) * prow : g_params + mp * gpstride;
    const bool rmw_always = partials == nullptr;  // g_params accumulates (the caller zeroed it)
    const int64_t ntiles = (N + TS - 1) / TS;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool first = tile == blockIdx.x && !rmw_always;
    const int64_t n0 = tile * TS;
    const int ts = (int)((N - n0) < (int64_t)TS ? (N - n0) : (int64_t)TS);
    const T* zt = z + (m * N + n0) * D;
    const T* gzo = g_zout + (m * N + n0) * D;
    T* gz = g_z + (m * N + n0) * D;
    __syncthreads();  // the previous tile's readers are done with the activation planes
    const int64_t plane = (int64_t)TS * W;
    // act(net, l): l = 0..L ; act(*, 0) is x for both nets
    T* actx = smem;                                   // [TS][W]
    T* acts = smem + plane;                           // [2][L][TS][W]  (l = 1..L)
    T* outb = acts + 2 * (int64_t)L * plane;          // [2][TS][W]   t, s
    T* dlt = outb + 2 * plane;                        // [2][2][TS][W]
    auto act = [&](int net, int l) -> T* { return l == 0 ? actx : acts + ((int64_t)net * L + (l - 1)) * plane; };

    for (int idx = tid; idx < ts * cd.d_in; idx += 256) {
        const int i = idx / cd.d_in, k = idx - i * cd.d_in;
        actx[i * W + k] = zt[(int64_t)i * D + c_off + k];
    }
    __syncthreads();

    // ---- recompute the forward pass, keeping every activation ----
    {
        const T* p = p0;
        for (int l = 0; l <= L; ++l) {
            const int din = (l == 0) ? cd.d_in : U;
            const int dout = (l == L) ? cd.d_out : U;
            const int64_t nw = (int64_t)din * dout;
            const T* wt = p;
            const T* ws = p + nw;
            const T* bt = p + 2 * nw;
            const T* bs = bt + dout;
            p = bs + dout;
            const T* xt = act(0, l);
            const T* xs = act(1, l);
            T* ot = (l == L) ? outb : act(0, l + 1);
            T* os = (l == L) ? outb + plane : act(1, l + 1);
            for (int idx = tid; idx < ts * dout; idx += 256) {
                const int i = idx / dout, o = idx - i * dout;
                T a_t = 0, a_s = 0;
                for (int k = 0; k < din; ++k) {
                    a_t += xt[i * W + k] * wt[(int64_t)k * dout + o];
                    a_s += xs[i * W + k] * ws[(int64_t)k * dout + o];
                }
                a_t += bt[o];
                a_s += bs[o];
                if (l < L) {
                    a_t = bw_tanh<T>(a_t);
                    a_s = bw_tanh<T>(a_s);
                }
                ot[i * W + o] = a_t;
                os[i * W + o] = a_s;
            }
            __syncthreads();
        }
    }

    // ---- output stage: gradients w.r.t. y, t, s ----
    int cur = 0;
    T* dt_ = dlt;                 // delta of the t-net, buffer `cur`
    T* ds_ = dlt + 2 * plane;     // delta of the s-net
    for (int idx = tid; idx < ts * cd.d_out; idx += 256) {
        const int i = idx / cd.d_out, o = idx - i * cd.d_out;
        const T t = outb[i * W + o], s = outb[plane + i * W + o];
        const T y = zt[(int64_t)i * D + t_off + o];
        const T gy = gzo[(int64_t)i * D + t_off + o];
        const T gl = g_ld[m * N + n0 + i];
        T dy, dt, dsv;
        if (inverse) {
            const T em = bw_exp<T>(-s);
            dy = gy * em;
            dt = -dy;
            dsv = -gy * ((y - t) * em) + gl;
        } else {
            const T e = bw_exp<T>(s);
            dy = gy * e;
            dt = gy;
            dsv = gy * y * e + gl;
        }
        gz[(int64_t)i * D + t_off + o] = dy;
        dt_[i * W + o] = dt;
        ds_[i * W + o] = dsv;
    }
    __syncthreads();

    // ---- back through the layers ----
    // parameter offsets of layer l inside the row
    for (int l = L; l >= 0; --l) {
        const int din = (l == 0) ? cd.d_in : U;
        const int dout = (l == L) ? cd.d_out : U;
        int64_t off = 0;
        for (int ll = 0; ll < l; ++ll) {
            const int di = (ll == 0) ? cd.d_in : U;
            off += 2 * ((int64_t)di * U + U);
        }
        const int64_t nw = (int64_t)din * dout;
        const T* wt = p0 + off;
        const T* ws = wt + nw;
        T* gwt = gp0 + off;
        T* gws = gwt + nw;
        T* gbt = gwt + 2 * nw;
        T* gbs = gbt + dout;
        const T* dtc = dlt + (int64_t)cur * plane;
        const T* dsc = dlt + (2 + (int64_t)cur) * plane;
        const T* xt = act(0, l);
        const T* xs = act(1, l);
        // weight and bias gradients of this tile
        for (int64_t idx = tid; idx < nw; idx += 256) {
            const int k = (int)(idx / dout), o = (int)(idx - (int64_t)k * dout);
            T a_t = 0, a_s = 0;
            for (int i = 0; i < ts; ++i) {
                a_t += xt[i * W + k] * dtc[i * W + o];
                a_s += xs[i * W + k] * dsc[i * W + o];
            }
            if (det) {
                gwt[idx] = first ? a_t : gwt[idx] + a_t;
                gws[idx] = first ? a_s : gws[idx] + a_s;
            } else {
                atomicAdd(gwt + idx, a_t);
                atomicAdd(gws + idx, a_s);
            }
        }
        for (int o = tid; o < dout; o += 256) {
            T a_t = 0, a_s = 0;
            for (int i = 0; i < ts; ++i) {
                a_t += dtc[i * W + o];
                a_s += dsc[i * W + o];
            }
            if (det) {
                gbt[o] = first ? a_t : gbt[o] + a_t;
                gbs[o] = first ? a_s : gbs[o] + a_s;
            } else {
                atomicAdd(gbt + o, a_t);
                atomicAdd(gbs + o, a_s);
            }
        }
        // delta of the previous layer (or of x)
        if (l > 0) {
            T* dtn = dlt + (int64_t)(cur ^ 1) * plane;
            T* dsn = dlt + (2 + (int64_t)(cur ^ 1)) * plane;
            for (int idx = tid; idx < ts * din; idx += 256) {
                const int i = idx / din, k = idx - i * din;
                T a_t = 0, a_s = 0;
                for (int o = 0; o < dout; ++o) {
                    a_t += wt[(int64_t)k * dout + o] * dtc[i * W + o];
                    a_s += ws[(int64_t)k * dout + o] * dsc[i * W + o];
                }
                const T ht = xt[i * W + k], hs = xs[i * W + k];
                dtn[i * W + k] = a_t * (1 - ht * ht);
                dsn[i * W + k] = a_s * (1 - hs * hs);
            }
        } else {
            for (int idx = tid; idx < ts * din; idx += 256) {
                const int i = idx / din, k = idx - i * din;
                T a = 0;
                for (int o = 0; o < dout; ++o)
                    a += wt[(int64_t)k * dout + o] * dtc[i * W + o] + ws[(int64_t)k * dout + o] * dsc[i * W + o];
                gz[(int64_t)i * D + c_off + k] = gzo[(int64_t)i * D + c_off + k] + a;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    }  // tiles
}

// g_params[mp][i] += sum over the G partial rows of parameter row mp, in workgroup order (deterministic)
template <typename T>
__global__ void __launch_bounds__(256)
backward_reduce_kernel(const T* __restrict__ partials, T* __restrict__ g_params, int64_t rows, int G, int64_t P,
                       int64_t gpstride) {
    const int64_t mp = grid_m();
    if (mp >= rows) return;
    const T* src = partials + mp * G * P;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256) {
        T acc = 0;
        for (int b = 0; b < G; ++b) acc += src[(int64_t)b * P + i];
        g_params[mp * gpstride + i] += acc;
    }
}
int launch_backward_reduce(int dtype, const void* partials, void* g_params, int64_t rows, int G, int64_t P,
                           int64_t gpstride, hipStream_t st) {
    const dim3 grid = grid_xm((P + 255) / 256, rows);
    if (dtype == TNF_F32)
        hipLaunchKernelGGL(backward_reduce_kernel<float>, grid, dim3(256), 0, st, (const float*)partials, (float*)g_params, rows, G, P, gpstride);
    else
        hipLaunchKernelGGL(backward_reduce_kernel<double>, grid, dim3(256), 0, st, (const double*)partials, (double*)g_params, rows, G, P, gpstride);
    return check_launch("backward_reduce");
}

// Deterministic geometry of the shape-generic backward kernels: G persistent workgroups per parameter row.  One shared
// row: the M sample batches are one batch (the tensors are contiguous); per-context rows: G = 1 once there are enough
// contexts to fill the chip, and a row's only workgroup then writes g_params itself.
void backward_det_geometry(int64_t M, int64_t Mp, int64_t tiles_per_m, int* G, int64_t* rows) {
    const int64_t target = 512;
    int64_t g = Mp == 1 ? target : (target + M - 1) / M;
    const int64_t tiles = Mp == 1 ? tiles_per_m * M : tiles_per_m;
    if (g > tiles) g = tiles;
    if (g < 1) g = 1;
    *G = (int)g;
    *rows = Mp == 1 ? 1 : M;
}

int64_t coupling_backward_workspace(int dtype, int64_t M, int64_t Mp, int64_t N, int D, int L, int U, int upper) {
    const size_t esz = dtype == TNF_F64 ? 8 : 4;
    const CouplingDims cd = coupling_dims(D, upper);
    int W = cd.d_in > cd.d_out ? cd.d_in : cd.d_out;
    if (U > W) W = U;
    int64_t TS = (int64_t)(96 * 1024) / (int64_t)((size_t)(1 + 2 * L + 2 + 4) * W * esz);
    if (TS > 64) TS = 64;
    if (TS < 1) TS = 1;
    int G;
    int64_t rows;
    backward_det_geometry(M, Mp, ((N > 0 ? N : 1) + TS - 1) / TS, &G, &rows);  // (an upper bound: TS shrinks with N)
    return G > 1 ? rows * G * coupling_num_params(D, L, U, upper) * (int64_t)esz : 0;
}

int launch_coupling_backward(int dtype, const void* z, const void* params, const void* g_zout,
                             const void* g_ld, void* g_z, void* g_params, int64_t M, int64_t Mp,
                             int64_t N, int D, int L, int U, int upper, int inverse, int64_t pstride,
                             int64_t gpstride, hipStream_t st, void* ws, int64_t ws_bytes) {
    const CouplingDims cd = coupling_dims(D, upper);
    diag_count(TNF_DIAG_BWD_GENERIC);
    if (ws_bytes >= 0 && Mp == 1 && M > 1) {  // deterministic mode, one shared row: one batch of M * N samples
        N *= M;
        M = 1;
    }
    int W = cd.d_in > cd.d_out ? cd.d_in : cd.d_out;
    if (U > W) W = U;
    const size_t esz = dtype == TNF_F64 ? 8 : 4;
    const int planes = 1 + 2 * L + 2 + 4;
    int64_t TS = (int64_t)(96 * 1024) / (int64_t)((size_t)planes * W * esz);
    if (TS > 64) TS = 64;
    if (TS > N) TS = N;
    if (TS < 1) TS = 1;
    const size_t smem = (size_t)planes * TS * W * esz;
    if (smem > 160 * 1024)
        return fail(TNF_EUNSUPPORTED, "coupling_backward: layer width %d needs %zu B of LDS", W, smem);
    int64_t tiles = (N + TS - 1) / TS;
    if (tiles > 0x7fffffff) return fail(TNF_EUNSUPPORTED, "coupling_backward: grid too large");
    // ws_bytes < 0: legacy mode (atomics).  Otherwise deterministic: G workgroups per parameter row, partial rows in ws
    void* partials = nullptr;
    int64_t prow = 0;
    int G = 1;
    int64_t rows = 0;
    if (ws_bytes >= 0) {
        backward_det_geometry(M, Mp, tiles, &G, &rows);
        prow = coupling_num_params(D, L, U, upper);
        if (G > 1) {
            if (!ws || ws_bytes < rows * G * prow * (int64_t)esz)
                return fail(TNF_EWORKSPACE, "coupling_backward: workspace %lld < %lld", (long long)ws_bytes,
                            (long long)(rows * G * prow * (int64_t)esz));
            partials = ws;
        } else {
            prow = -1;  // the row's only workgroup accumulates in g_params itself
        }
        tiles = G;
    }
    const dim3 grid = grid_xm(tiles, M);
    if (dtype == TNF_F32) {
        auto k = coupling_backward_kernel<float>;
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, (const float*)z, (const float*)params,
                           (const float*)g_zout, (const float*)g_ld, (float*)g_z, (float*)g_params, M, Mp, N, D,
                           L, U, upper, inverse, pstride, gpstride, (int)TS, W, (float*)partials, prow);
    } else {
        auto k = coupling_backward_kernel<double>;
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, (const double*)z, (const double*)params,
                           (const double*)g_zout, (const double*)g_ld, (double*)g_z, (double*)g_params, M, Mp,
                           N, D, L, U, upper, inverse, pstride, gpstride, (int)TS, W, (double*)partials, prow);
    }
    const int rc = check_launch("coupling_backward");
    if (rc || !partials) return rc;
    return launch_backward_reduce(dtype, partials, g_params, rows, G, prow, gpstride, st);
}

// ---------------------------------------------------------------------------
// Affine (bijectors.py:277-315): z' = e^a z + b (fwd) or (z - b) e^-a (inv), ld = sum(a).
//   fwd: dz = g e^a,  da = sum_n g z e^a + g_ld,          db = sum_n g
//   inv: dz = g e^-a, da = -sum_n g (z - b) e^-a + g_ld,  db = -sum_n g e^-a
// Workgroup = (row chunk, m): per-feature partial sums in registers -> LDS -> atomics.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
affine_backward_kernel(const T* __restrict__ z, const T* __restrict__ params,
                       const T* __restrict__ g_zout, const T* __restrict__ g_ld, T* __restrict__ g_z,
                       T* __restrict__ g_params, int64_t M, int64_t Mp, int64_t N, int D, int inverse,
                       int64_t pstride, int64_t gpstride, int64_t rows_per_block) {
    __shared__ double red_a[256];
    __shared__ double red_b[256];
    const int tid = threadIdx.x;
    const int64_t m = grid_m();
    if (m >= M) return;
    const int64_t mp = Mp == 1 ? 0 : m;
    const T* p = params + mp * pstride;
    T* gp = g_params + mp * gpstride;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    for (int dc = 0; dc < D; dc += 256) {
        const int Dc = (D - dc) < 256 ? (D - dc) : 256;
        const int rpi = 256 / Dc;
        const int r = tid / Dc, d = tid - r * Dc;
        double sa = 0.0, sb = 0.0;
        if (r < rpi) {
            const T a = p[dc + d], b = p[D + dc + d];
            const T e = bw_exp<T>(inverse ? -a : a);
            for (int64_t row = r0 + r; row < r1; row += rpi) {
                const int64_t at = (m * N + row) * D + dc + d;
                const T g = g_zout[at], zv = z[at];
                g_z[at] = g * e;
                if (inverse) {
                    sa -= (double)(g * (zv - b) * e);
                    sb -= (double)(g * e);
                } else {
                    sa += (double)(g * zv * e);
                    sb += (double)g;
                }
            }
        }
        red_a[tid] = sa;
        red_b[tid] = sb;
        __syncthreads();
        if (tid < Dc) {
            double a = 0.0, b = 0.0;
            for (int rr = 0; rr < rpi; ++rr) {
                a += red_a[rr * Dc + tid];
                b += red_b[rr * Dc + tid];
            }
            atomicAdd(gp + dc + tid, (T)a);
            atomicAdd(gp + D + dc + tid, (T)b);
        }
        __syncthreads();
    }
}

// g_alpha += g_ld (the log-det is sum(alpha) per parameter row, shape (Mp, 1))
template <typename T>
__global__ void __launch_bounds__(256)
affine_backward_ld_kernel(const T* __restrict__ g_ld, T* __restrict__ g_params, int D, int64_t gpstride) {
    const int64_t mp = blockIdx.x;
    const T g = g_ld[mp];
    for (int d = threadIdx.x; d < D; d += 256) atomicAdd(g_params + mp * gpstride + d, g);
}

int launch_affine_backward(int dtype, const void* z, const void* params, const void* g_zout,
                           const void* g_ld, void* g_z, void* g_params, int64_t M, int64_t Mp, int64_t N,
                           int D, int inverse, int64_t pstride, int64_t gpstride, hipStream_t st) {
    int64_t blocks = (N + 255) / 256;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    const int64_t rpb = (N + blocks - 1) / blocks;
    const dim3 grid = grid_xm(blocks, M);
    if (dtype == TNF_F32) {
        if (N > 0)
            hipLaunchKernelGGL(affine_backward_kernel<float>, grid, dim3(256), 0, st, (const float*)z,
                               (const float*)params, (const float*)g_zout, (const float*)g_ld, (float*)g_z,
                               (float*)g_params, M, Mp, N, D, inverse, pstride, gpstride, rpb);
        hipLaunchKernelGGL(affine_backward_ld_kernel<float>, dim3((unsigned)Mp), dim3(256), 0, st,
                           (const float*)g_ld, (float*)g_params, D, gpstride);
    } else {
        if (N > 0)
            hipLaunchKernelGGL(affine_backward_kernel<double>, grid, dim3(256), 0, st, (const double*)z,
                               (const double*)params, (const double*)g_zout, (const double*)g_ld,
                               (double*)g_z, (double*)g_params, M, Mp, N, D, inverse, pstride, gpstride, rpb);
        hipLaunchKernelGGL(affine_backward_ld_kernel<double>, dim3((unsigned)Mp), dim3(256), 0, st,
                           (const double*)g_ld, (double*)g_params, D, gpstride);
    }
    return check_launch("affine_backward");
}

// ---------------------------------------------------------------------------
// BatchNorm with cached statistics: dz = g * alpha (inverse) or g / alpha (frozen forward).
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
bn_apply_backward_kernel(const T* __restrict__ g_zout, const float* __restrict__ alpha,
                         T* __restrict__ g_z, int D, int inverse, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const T a = (T)alpha[idx % D];
        g_z[idx] = inverse ? g_zout[idx] * a : g_zout[idx] / a;
    }
}

int launch_bn_apply_backward(int dtype, const void* g_zout, const float* alpha, void* g_z, int64_t rows,
                             int D, int inverse, hipStream_t st) {
    const int64_t total = rows * D;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (total > 0) {
        if (dtype == TNF_F32)
            hipLaunchKernelGGL(bn_apply_backward_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st,
                               (const float*)g_zout, alpha, (float*)g_z, D, inverse, total);
        else
            hipLaunchKernelGGL(bn_apply_backward_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st,
                               (const double*)g_zout, alpha, (double*)g_z, D, inverse, total);
    }
    return check_launch("bn_apply_backward");
}

// ---------------------------------------------------------------------------
// BatchNorm with batch statistics (bijectors.py:401-417), backward.  With x^ = (z - mu)/alpha,
// alpha = sqrt(var_b + eps), log_det = -sum_d log alpha_d and n rows:
//   dz = (1/alpha) [ g - mean(g) - x^ (mean(g x^) + g_ld / n) ]
// (the usual batch-norm backward plus d log_det/dz = -x^/(n alpha)).  Two passes: per-feature
// sums of g and g*x^ in float64 (block partials -> double atomics), then elementwise.
// workspace: 2*D doubles.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
bn_batch_bwd_sums_kernel(const float* __restrict__ zn, const float* __restrict__ g, double* __restrict__ sums,
                         int64_t rows, int D, int64_t rows_per_block) {
    __shared__ double red1[256];
    __shared__ double red2[256];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (int dc = 0; dc < D; dc += 256) {
        const int Dc = (D - dc) < 256 ? (D - dc) : 256;
        const int rpi = 256 / Dc;
        const int r = tid / Dc, d = tid - r * Dc;
        double s1 = 0.0, s2 = 0.0;
        if (r < rpi) {
            for (int64_t row = r0 + r; row < r1; row += rpi) {
                const double gv = (double)g[row * D + dc + d];
                s1 += gv;
                s2 += gv * (double)zn[row * D + dc + d];
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

__global__ void __launch_bounds__(256)
bn_batch_bwd_apply_kernel(const float* __restrict__ zn, const float* __restrict__ g,
                          const float* __restrict__ g_ld, const float* __restrict__ alpha,
                          const double* __restrict__ sums, float* __restrict__ g_z, int64_t rows, int D,
                          int64_t total, const double* __restrict__ count) {
    const double inv_n = 1.0 / (count ? *count : (double)rows);  // count: the GLOBAL row count of a sample-sharded batch
    const double gl = g_ld ? (double)*g_ld : 0.0;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int d = (int)(idx % D);
        const float mg = (float)(sums[d] * inv_n);
        const float mgx = (float)((sums[D + d] + gl) * inv_n);
        g_z[idx] = (g[idx] - mg - zn[idx] * mgx) / alpha[d];
    }
}

// the two halves of the backward, separately callable: a sample-sharded batch sums `sums` over the ranks in between
int launch_bn_batch_backward_sums(const float* zn, const float* g, double* sums, int64_t rows, int D, hipStream_t st) {
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * (size_t)D, st) != hipSuccess)
        return fail(TNF_ELAUNCH, "bn_batch_backward: memset failed");
    if (rows <= 0) return TNF_OK;  // an empty shard contributes zeros
    int64_t blocks = (rows + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    const int64_t rpb = (rows + blocks - 1) / blocks;
    hipLaunchKernelGGL(bn_batch_bwd_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, st, zn, g, sums, rows, D, rpb);
    return check_launch("bn_batch_backward_sums");
}
int launch_bn_batch_backward_apply(const float* zn, const float* g, const float* g_ld, const float* alpha,
                                   const double* sums, const double* count, float* g_z, int64_t rows, int D,
                                   hipStream_t st) {
    const int64_t total = rows * D;
    if (total <= 0) return TNF_OK;
    int64_t nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(bn_batch_bwd_apply_kernel, dim3((unsigned)nb), dim3(256), 0, st, zn, g, g_ld, alpha, sums, g_z,
                       rows, D, total, count);
    return check_launch("bn_batch_backward_apply");
}
int launch_bn_batch_backward(const float* zn, const float* g, const float* g_ld, const float* alpha, float* g_z,
                             int64_t rows, int D, void* ws, hipStream_t st) {
    double* sums = reinterpret_cast<double*>(ws);
    const int rc = launch_bn_batch_backward_sums(zn, g, sums, rows, D, st);
    if (rc) return rc;
    return launch_bn_batch_backward_apply(zn, g, g_ld, alpha, sums, nullptr, g_z, rows, D, st);
}

}  // namespace tnf
