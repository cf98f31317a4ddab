// MAF (masked autoregressive flow) bijector -- the reference's NormFlow default arch_type "AR"
// (bijectors.py:597-806).  Twin masked MLPs (f_mu, f_alpha) without biases over ALL D inputs:
//   inverse (density evaluation, ONE pass):  z' = (z - f_mu(z)) / exp(f_alpha(z)),  ld = sum f_alpha
//   forward (sampling, D-1 sequential passes): z_{i+1} = u * exp(f_alpha(z_i)) + f_mu(z_i), z_0 = u
// Shape- and dtype-generic like generic_kernels.hip: one workgroup = one parameter row m and a
// tile of TS samples, activations ping-pong through LDS, masked weights stream from L2.
// Packed parameters (bijectors.py:698-740): [W_mu0 | W_alpha0 | ... | W_mu_last | W_alpha_last],
// W row-major [in][out]; `masks` holds one binary matrix per layer in the same order/shape
// (D x U, (U x U) x (L-1), U x D), shared by both nets and by all parameter rows.
#include "tnf_common.h"

namespace tnf {

template <typename T>
__device__ __forceinline__ T maf_tanh(T x);
template <>
__device__ __forceinline__ float maf_tanh<float>(float x) { return tanhf(x); }
template <>
__device__ __forceinline__ double maf_tanh<double>(double x) { return tanh(x); }
template <typename T>
__device__ __forceinline__ T maf_exp(T x);
template <>
__device__ __forceinline__ float maf_exp<float>(float x) { return expf(x); }
template <>
__device__ __forceinline__ double maf_exp<double>(double x) { return exp(x); }

// one evaluation of the twin masked nets on the tile: input zin [TS][W] -> mu, alpha in bm[cur], ba[cur]
template <typename T>
__device__ __forceinline__ int maf_net(const T* __restrict__ params, const T* __restrict__ masks, const T* zin,
                                       T* bm0, T* bm1, T* ba0, T* ba1, int ts, int D, int L, int U, int W,
                                       int tid) {
    T* bm[2] = {bm0, bm1};
    T* ba[2] = {ba0, ba1};
    const T* p = params;
    const T* mk = masks;
    int cur = 0;
    for (int l = 0; l <= L; ++l) {
        const int din = (l == 0) ? D : U;
        const int dout = (l == L) ? D : U;
        const int64_t nw = (int64_t)din * dout;
        const T* wm = p;
        const T* wa = p + nw;
        p += 2 * nw;
        const T* xm = (l == 0) ? zin : bm[cur];
        const T* xa = (l == 0) ? zin : ba[cur];
        T* om = bm[cur ^ 1];
        T* oa = ba[cur ^ 1];
        for (int idx = tid; idx < ts * dout; idx += 256) {
            const int i = idx / dout, o = idx - i * dout;
            T am = 0, aa = 0;
            for (int k = 0; k < din; ++k) {
                const T mv = mk[(int64_t)k * dout + o];
                am += xm[i * W + k] * (mv * wm[(int64_t)k * dout + o]);
                aa += xa[i * W + k] * (mv * wa[(int64_t)k * dout + o]);
            }
            if (l < L) {
                am = maf_tanh<T>(am);
                aa = maf_tanh<T>(aa);
            }
            om[i * W + o] = am;
            oa[i * W + o] = aa;
        }
        mk += nw;
        __syncthreads();
        cur ^= 1;
    }
    return cur;
}

template <typename T>
__global__ void __launch_bounds__(256)
maf_kernel(const T* __restrict__ z, const T* __restrict__ params, const T* __restrict__ masks,
           T* __restrict__ z_out, T* __restrict__ log_det, T* __restrict__ alpha_out, int64_t Mz, int64_t Mp, int64_t N,
           int D, int L, int U, int inverse, int64_t pstride, int TS, int W) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x;
    const int64_t m = grid_m();
    if (m >= (Mz > Mp ? Mz : Mp)) return;
    const int64_t n0 = (int64_t)blockIdx.x * TS;
    const int ts = (int)((N - n0) < (int64_t)TS ? (N - n0) : (int64_t)TS);
    const int64_t plane = (int64_t)TS * W;
    T* bm[2] = {smem, smem + plane};
    T* ba[2] = {smem + 2 * plane, smem + 3 * plane};
    T* zc = smem + 4 * plane;   // current iterate [TS][W]
    const T* zt = z + ((Mz == 1 ? 0 : m) * N + n0) * D;
    T* zo = z_out + (m * N + n0) * D;
    const T* p = params + (Mp == 1 ? 0 : m) * pstride;

    for (int idx = tid; idx < ts * D; idx += 256) {
        const int i = idx / D, d = idx - i * D;
        zc[i * W + d] = zt[(int64_t)i * D + d];
    }
    __syncthreads();
    if (inverse) {
        const int cur = maf_net<T>(p, masks, zc, bm[0], bm[1], ba[0], ba[1], ts, D, L, U, W, tid);
        for (int idx = tid; idx < ts * D; idx += 256) {
            const int i = idx / D, d = idx - i * D;
            zo[(int64_t)i * D + d] = (zc[i * W + d] - bm[cur][i * W + d]) / maf_exp<T>(ba[cur][i * W + d]);
            if (alpha_out) alpha_out[(m * N + n0 + i) * D + d] = ba[cur][i * W + d];  // per-dimension f_alpha(z)
        }
        if (tid < ts) {
            T acc = 0;
            for (int d = 0; d < D; ++d) acc += ba[cur][tid * W + d];
            log_det[m * N + n0 + tid] = acc;
        }
    } else {
        int cur = 0;
        for (int it = 0; it < D - 1; ++it) {  // bijectors.py:752-754
            cur = maf_net<T>(p, masks, zc, bm[0], bm[1], ba[0], ba[1], ts, D, L, U, W, tid);
            for (int idx = tid; idx < ts * D; idx += 256) {
                const int i = idx / D, d = idx - i * D;
                zc[i * W + d] = zt[(int64_t)i * D + d] * maf_exp<T>(ba[cur][i * W + d]) + bm[cur][i * W + d];
            }
            __syncthreads();
        }
        for (int idx = tid; idx < ts * D; idx += 256) {
            const int i = idx / D, d = idx - i * D;
            zo[(int64_t)i * D + d] = zc[i * W + d];
        }
        if (tid < ts) {
            T acc = 0;
            for (int d = 0; d < D; ++d) acc += ba[cur][tid * W + d];
            log_det[m * N + n0 + tid] = acc;
        }
    }
}

static int maf_tile(int D, int U, int L, size_t esz, int planes, int64_t N, int* W_out, size_t* smem_out) {
    int W = D > U ? D : U;
    (void)L;
    int64_t TS = (int64_t)(64 * 1024) / (int64_t)((size_t)planes * W * esz);
    if (TS > 64) TS = 64;
    if (TS > N) TS = N;
    if (TS < 1) TS = 1;
    *W_out = W;
    *smem_out = (size_t)planes * TS * W * esz;
    return (int)TS;
}

int launch_maf(int dtype, const void* z, const void* params, const void* masks, void* z_out, void* log_det,
               int64_t Mz, int64_t Mp, int64_t N, int D, int L, int U, int inverse, int64_t pstride, hipStream_t st,
               void* alpha_out) {
    const int64_t M = Mz > Mp ? Mz : Mp;
    const size_t esz = dtype == TNF_F64 ? 8 : 4;
    int W;
    size_t smem;
    const int TS = maf_tile(D, U, L, esz, 5, N, &W, &smem);
    if (smem > 160 * 1024) return fail(TNF_EUNSUPPORTED, "maf: layer width %d needs %zu B of LDS", W, smem);
    const int64_t tiles = (N + TS - 1) / TS;
    if (tiles > 0x7fffffff) return fail(TNF_EUNSUPPORTED, "maf: grid too large");
    const dim3 grid = grid_xm(tiles, M);
    if (dtype == TNF_F32) {
        auto k = maf_kernel<float>;
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, (const float*)z, (const float*)params, (const float*)masks,
                           (float*)z_out, (float*)log_det, (float*)alpha_out, Mz, Mp, N, D, L, U, inverse, pstride, TS, W);
    } else {
        auto k = maf_kernel<double>;
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, (const double*)z, (const double*)params,
                           (const double*)masks, (double*)z_out, (double*)log_det, (double*)alpha_out, Mz, Mp, N, D, L, U,
                           inverse, pstride, TS, W);
    }
    return check_launch("maf");
}

// ---------------------------------------------------------------------------
// Backward of the inverse direction (what log_prob training differentiates):
//   out = (z - mu) e^-alpha, ld = sum alpha
//   dz = g e^-alpha + (nets' input gradient),  dmu = -g e^-alpha,  dalpha = -g out + g_ld
// LDS: actx [TS][W] (= z), act[net][l = 1..L][TS][W], out[net][TS][W], delta[net][2][TS][W], and -- when it
// fits (`lacc`) -- one accumulator per parameter: a workgroup then walks its tiles grid-stride, every
// weight index is owned by one thread, and the parameter gradient leaves the workgroup once at the end
// (plain stores when the workgroup owns its context's row, else one atomic per parameter per workgroup).
// Without it every tile sends one atomic per parameter.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
maf_backward_kernel(const T* __restrict__ z, const T* __restrict__ params, const T* __restrict__ masks,
                    const T* __restrict__ g_zout, const T* __restrict__ g_ld, T* __restrict__ g_z,
                    T* __restrict__ g_params, int64_t M, int64_t Mp, int64_t N, int D, int L, int U,
                    int64_t pstride, int64_t gpstride, int TS, int W, int lacc, int lw, int P,
                    T* __restrict__ partials, int det) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x;
    const int64_t m = grid_m();
    if (m >= M) return;
    const int64_t mp = Mp == 1 ? 0 : m;
    const T* p0 = params + mp * pstride;
    // deterministic mode (det): contributions go to this workgroup's row of `partials` (backward_reduce_kernel adds the rows
    // in workgroup order) or, with one workgroup per parameter row (partials == NULL), straight into g_params -- every
    // element owned by one thread, no atomics (see coupling_backward_kernel)
    T* gp0 = partials ? partials + (m * gridDim.x + blockIdx.x) * (int64_t)P : g_params + mp * gpstride;
    const int64_t plane = (int64_t)TS * W;
    T* actx = smem;
    T* acts = smem + plane;                       // [2][L][TS][W]
    T* outb = acts + 2 * (int64_t)L * plane;      // [2][TS][W]  mu, alpha
    T* dlt = outb + 2 * plane;                    // [2][2][TS][W]
    T* gacc = dlt + 4 * plane;                    // [P] when lacc
    T* wl = gacc + (lacc ? P : 0);                // [P] when lw: the masked weights W * M of this parameter row
    auto act = [&](int net, int l) -> T* { return l == 0 ? actx : acts + ((int64_t)net * L + (l - 1)) * plane; };
    if (lacc)
        for (int i = tid; i < P; i += 256) gacc[i] = 0;
    if (lw) {  // parameter index -> mask index: per layer [W_mu | W_alpha] share one mask
        int64_t off = 0, moff = 0;
        for (int l = 0; l <= L; ++l) {
            const int64_t nw = (int64_t)((l == 0) ? D : U) * ((l == L) ? D : U);
            for (int64_t i = tid; i < nw; i += 256) {
                const T mv = masks[moff + i];
                wl[off + i] = mv * p0[off + i];
                wl[off + nw + i] = mv * p0[off + nw + i];
            }
            off += 2 * nw;
            moff += nw;
        }
    }
    const T* pw = lw ? wl : p0;  // weights as the loops below read them (masked already when lw)
    const int64_t ntiles = (N + TS - 1) / TS;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool first = det && partials != nullptr && tile == blockIdx.x;  // partial rows start from this tile's value
    const int64_t n0 = tile * TS;
    const int ts = (int)((N - n0) < (int64_t)TS ? (N - n0) : (int64_t)TS);
    const T* zt = z + (m * N + n0) * D;
    const T* gzo = g_zout + (m * N + n0) * D;
    T* gz = g_z + (m * N + n0) * D;
    __syncthreads();  // the previous tile's readers are done with the activation planes

    for (int idx = tid; idx < ts * D; idx += 256) {
        const int i = idx / D, d = idx - i * D;
        actx[i * W + d] = zt[(int64_t)i * D + d];
    }
    __syncthreads();
    {   // forward recompute, keeping every activation
        const T* p = pw;
        const T* mk = masks;
        for (int l = 0; l <= L; ++l) {
            const int din = (l == 0) ? D : U, dout = (l == L) ? D : U;
            const int64_t nw = (int64_t)din * dout;
            const T* wm = p;
            const T* wa = p + nw;
            p += 2 * nw;
            const T* xm = act(0, l);
            const T* xa = act(1, l);
            T* om = (l == L) ? outb : act(0, l + 1);
            T* oa = (l == L) ? outb + plane : act(1, l + 1);
            for (int idx = tid; idx < ts * dout; idx += 256) {
                const int i = idx / dout, o = idx - i * dout;
                T am = 0, aa = 0;
                if (lw) {
                    for (int k = 0; k < din; ++k) {
                        am += xm[i * W + k] * wm[k * dout + o];
                        aa += xa[i * W + k] * wa[k * dout + o];
                    }
                } else
                for (int k = 0; k < din; ++k) {
                    const T mv = mk[(int64_t)k * dout + o];
                    am += xm[i * W + k] * (mv * wm[(int64_t)k * dout + o]);
                    aa += xa[i * W + k] * (mv * wa[(int64_t)k * dout + o]);
                }
                if (l < L) {
                    am = maf_tanh<T>(am);
                    aa = maf_tanh<T>(aa);
                }
                om[i * W + o] = am;
                oa[i * W + o] = aa;
            }
            mk += nw;
            __syncthreads();
        }
    }
    int cur = 0;
    for (int idx = tid; idx < ts * D; idx += 256) {
        const int i = idx / D, d = idx - i * D;
        const T mu = outb[i * W + d], al = outb[plane + i * W + d];
        const T em = maf_exp<T>(-al);
        const T g = gzo[(int64_t)i * D + d];
        const T gl = g_ld[m * N + n0 + i];
        const T dz = g * em;
        gz[(int64_t)i * D + d] = dz;                              // direct path; the nets' share is added below
        dlt[i * W + d] = -dz;                                     // d mu
        dlt[2 * plane + i * W + d] = -g * ((actx[i * W + d] - mu) * em) + gl;  // d alpha
    }
    __syncthreads();
    for (int l = L; l >= 0; --l) {
        const int din = (l == 0) ? D : U, dout = (l == L) ? D : U;
        int64_t off = 0, moff = 0;
        for (int ll = 0; ll < l; ++ll) {
            const int di = (ll == 0) ? D : U;
            off += 2 * (int64_t)di * U;
            moff += (int64_t)di * U;
        }
        const int64_t nw = (int64_t)din * dout;
        const T* wm = pw + off;
        const T* wa = wm + nw;
        const T* mk = masks + moff;
        T* gwm = gp0 + off;
        T* gwa = gwm + nw;
        const T* dmc = dlt + (int64_t)cur * plane;
        const T* dac = dlt + (2 + (int64_t)cur) * plane;
        const T* xm = act(0, l);
        const T* xa = act(1, l);
        for (int64_t idx = tid; idx < nw; idx += 256) {
            const int k = (int)(idx / dout), o = (int)(idx - (int64_t)k * dout);
            const T mv = mk[idx];
            if (mv != (T)0) {  // masked weights get exactly zero gradient, like autograd through Ms * W
                T am = 0, aa = 0;
                for (int i = 0; i < ts; ++i) {
                    am += xm[i * W + k] * dmc[i * W + o];
                    aa += xa[i * W + k] * dac[i * W + o];
                }
                if (lacc) {  // idx is owned by this thread for every tile of the workgroup
                    gacc[off + idx] += mv * am;
                    gacc[off + nw + idx] += mv * aa;
                } else if (det) {
                    gwm[idx] = first ? mv * am : gwm[idx] + mv * am;
                    gwa[idx] = first ? mv * aa : gwa[idx] + mv * aa;
                } else {
                    atomicAdd(gwm + idx, mv * am);
                    atomicAdd(gwa + idx, mv * aa);
                }
            } else if (first && !lacc) {
                gwm[idx] = 0;
                gwa[idx] = 0;
            }
        }
        if (l > 0) {
            T* dmn = dlt + (int64_t)(cur ^ 1) * plane;
            T* dan = dlt + (2 + (int64_t)(cur ^ 1)) * plane;
            for (int idx = tid; idx < ts * din; idx += 256) {
                const int i = idx / din, k = idx - i * din;
                T am = 0, aa = 0;
                if (lw) {
                    for (int o = 0; o < dout; ++o) {
                        am += wm[k * dout + o] * dmc[i * W + o];
                        aa += wa[k * dout + o] * dac[i * W + o];
                    }
                } else
                for (int o = 0; o < dout; ++o) {
                    const T mv = mk[(int64_t)k * dout + o];
                    am += mv * wm[(int64_t)k * dout + o] * dmc[i * W + o];
                    aa += mv * wa[(int64_t)k * dout + o] * dac[i * W + o];
                }
                const T hm = xm[i * W + k], ha = xa[i * W + k];
                dmn[i * W + k] = am * (1 - hm * hm);
                dan[i * W + k] = aa * (1 - ha * ha);
            }
        } else {
            for (int idx = tid; idx < ts * din; idx += 256) {
                const int i = idx / din, k = idx - i * din;
                T a = 0;
                if (lw) {
                    for (int o = 0; o < dout; ++o) a += wm[k * dout + o] * dmc[i * W + o] + wa[k * dout + o] * dac[i * W + o];
                } else
                for (int o = 0; o < dout; ++o) {
                    const T mv = mk[(int64_t)k * dout + o];
                    a += mv * (wm[(int64_t)k * dout + o] * dmc[i * W + o] + wa[(int64_t)k * dout + o] * dac[i * W + o]);
                }
                gz[(int64_t)i * D + k] += a;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    }  // tiles
    if (lacc) {
        __syncthreads();
        const bool own = Mp > 1 && gridDim.x == 1;  // this workgroup is the only writer of its context's row
        for (int i = tid; i < P; i += 256) {
            const T v = gacc[i];
            if (own || (det && partials)) gp0[i] = v;
            else if (det) gp0[i] += v;  // one workgroup per row, g_params accumulates
            else if (v != (T)0) atomicAdd(gp0 + i, v);
        }
    }
}

int64_t maf_backward_workspace(int dtype, int64_t M, int64_t Mp, int64_t N, int D, int L, int U) {
    const size_t esz = dtype == TNF_F64 ? 8 : 4;
    int W;
    size_t smem;
    const int TS = maf_tile(D, U, L, esz, 1 + 2 * L + 2 + 4, N > 0 ? N : 1, &W, &smem);
    int G;
    int64_t rows;
    backward_det_geometry(M, Mp, ((N > 0 ? N : 1) + TS - 1) / TS, &G, &rows);
    const int64_t P = 2 * (2 * (int64_t)D * U + (int64_t)(L - 1) * U * U);
    return G > 1 ? rows * G * P * (int64_t)esz : 0;
}

int launch_maf_backward(int dtype, const void* z, const void* params, const void* masks, const void* g_zout,
                        const void* g_ld, void* g_z, void* g_params, int64_t M, int64_t Mp, int64_t N, int D, int L,
                        int U, int64_t pstride, int64_t gpstride, hipStream_t st, void* ws, int64_t ws_bytes) {
    const size_t esz = dtype == TNF_F64 ? 8 : 4;
    diag_count(TNF_DIAG_MAF_BWD_GENERIC);
    const int det = ws_bytes >= 0;
    if (det && Mp == 1 && M > 1) {  // one shared row: one batch of M * N samples
        N *= M;
        M = 1;
    }
    int W;
    size_t smem;
    const int TS = maf_tile(D, U, L, esz, 1 + 2 * L + 2 + 4, N, &W, &smem);
    if (smem > 160 * 1024) return fail(TNF_EUNSUPPORTED, "maf_backward: layer width %d needs %zu B of LDS", W, smem);
    int64_t tiles = (N + TS - 1) / TS;
    if (tiles > 0x7fffffff) return fail(TNF_EUNSUPPORTED, "maf_backward: grid too large");
    // per-parameter accumulators in LDS when they fit next to the activation planes
    const int64_t P = 2 * (2 * (int64_t)D * U + (int64_t)(L - 1) * U * U);
    const int lw = smem + (size_t)P * esz <= 150 * 1024;  // the masked weights in LDS (first: it speeds every loop)
    if (lw) smem += (size_t)P * esz;
    const int lacc = smem + (size_t)P * esz <= 150 * 1024;
    if (lacc) {
        smem += (size_t)P * esz;
        if (Mp > 1) tiles = 1;  // one workgroup per context owns the row: plain stores, no atomics
        else {
            int64_t cap = 1024 / M;
            if (cap < 1) cap = 1;
            if (tiles > cap) tiles = cap;
        }
    }
    void* partials = nullptr;
    int G = 1;
    int64_t rows = 0;
    if (det) {  // G persistent workgroups per parameter row, their contributions added in workgroup order
        backward_det_geometry(M, Mp, tiles, &G, &rows);
        if (G > 1) {
            if (!ws || ws_bytes < rows * G * P * (int64_t)esz)
                return fail(TNF_EWORKSPACE, "maf_backward: workspace %lld < %lld", (long long)ws_bytes,
                            (long long)(rows * G * P * (int64_t)esz));
            partials = ws;
        }
        tiles = G;
    }
    const dim3 grid = grid_xm(tiles, M);
    if (dtype == TNF_F32) {
        auto k = maf_backward_kernel<float>;
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, (const float*)z, (const float*)params, (const float*)masks,
                           (const float*)g_zout, (const float*)g_ld, (float*)g_z, (float*)g_params, M, Mp, N, D, L,
                           U, pstride, gpstride, TS, W, lacc, lw, (int)P, (float*)partials, det);
    } else {
        auto k = maf_backward_kernel<double>;
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, (const double*)z, (const double*)params,
                           (const double*)masks, (const double*)g_zout, (const double*)g_ld, (double*)g_z,
                           (double*)g_params, M, Mp, N, D, L, U, pstride, gpstride, TS, W, lacc, lw, (int)P,
                           (double*)partials, det);
    }
    const int rc = check_launch("maf_backward");
    if (rc || !partials) return rc;
    return launch_backward_reduce(dtype, partials, g_params, rows, G, P, gpstride, st);
}

}  // namespace tnf
