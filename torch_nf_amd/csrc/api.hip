// C-ABI entry points of libtnf_hip.so (declared in include/tnf.h): argument checks,
// kernel selection and the launch sequences of the flow-level chains.
#include <string.h>

#include <atomic>

#include "tnf_common.h"

namespace tnf {

thread_local int g_force_generic = 0;
thread_local int g_operand_prec = 0;
thread_local const int* g_launch_gate = nullptr;

static std::atomic<long long> g_diag_launches[TNF_DIAG_FAMILIES];
void diag_count(int family) {
    if (family >= 0 && family < TNF_DIAG_FAMILIES) g_diag_launches[family].fetch_add(1, std::memory_order_relaxed);
}

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TNF_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return TNF_OK;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int check_mnd(const char* fn, int64_t Mz, int64_t Mp, int64_t N, int D) {
    if (Mz < 1 || Mp < 1 || N < 0) return fail(TNF_EINVAL, "%s: bad batch sizes M_z=%lld M_p=%lld N=%lld", fn, (long long)Mz, (long long)Mp, (long long)N);
    if (Mz != Mp && Mz != 1 && Mp != 1)
        return fail(TNF_EINVAL, "%s: M_z=%lld and M_p=%lld do not broadcast", fn, (long long)Mz, (long long)Mp);
    if (D < 1) return fail(TNF_EINVAL, "%s: D=%d must be positive", fn, D);
    return TNF_OK;
}

static int64_t round16(int64_t b) { return (b + 15) & ~(int64_t)15; }

struct FlowWs {
    int64_t fold, ldc, images, zbuf, ldbuf, total;
};
static int64_t flow_image_slot(int D, int L, int U) {
    // narrow shapes: fp32 and split-f16 images share one slot size (the L = 3 fp32 image);
    // wide shapes: the wide image of exactly this (D, L, U)
    if (!mfma_supported(D, L, U)) return wide_image_floats(D, L, U);
    // narrow shapes: fp32 and split-f16 images share one slot size (the L = 3 fp32 image); flow_ws takes the larger of
    // this and flow_prep_slot (the layer-range chain's prepared prologues)
    const int64_t a = mfma_image_floats(D, 3);
    return a;
}
static int64_t flow_prep_slot(int D, int S, int L, int U) {
    // per layer, so that 2S slots hold a chain's prepared prologues -- for EVERY number of layers per launch the chain
    // accepts (TNF_OPT_LAYER_VARIANT 10 + n): a launch's region grows with n faster than the number of launches falls
    if (!mfma_supported(D, L, U)) return 0;
    int64_t most = 0;
    for (int n = 1; n <= 2 * S; ++n) {
        if (!flow_range2_supported(D, L, U, n)) break;
        const int64_t need = flow_chain2_prep_floats(D, S, L, n);
        if (need > most) most = need;
    }
    return (most + 2 * S - 1) / (2 * S);
}
static FlowWs flow_ws(int64_t M, int64_t N, int D, int S, int L, int U) {
    FlowWs w;
    w.fold = 0;
    w.ldc = round16(M * 2 * S * 2 * D * (int64_t)sizeof(float));
    w.images = w.ldc + round16(M * (int64_t)sizeof(float));
    const int64_t slot = flow_image_slot(D, L, U) > flow_prep_slot(D, S, L, U) ? flow_image_slot(D, L, U) : flow_prep_slot(D, S, L, U);
    w.zbuf = w.images + round16(M * 2 * S * slot * (int64_t)sizeof(float));
    w.ldbuf = w.zbuf + round16(M * N * D * (int64_t)sizeof(float));
    w.total = w.ldbuf + round16(M * N * (int64_t)sizeof(float));
    return w;
}

}  // namespace tnf

using namespace tnf;

extern "C" {

int tnf_version(void) { return TNF_VERSION; }

const char* tnf_last_error(void) { return err_buf(); }

int tnf_set_launch_gate(const int32_t* flag) {
    g_launch_gate = reinterpret_cast<const int*>(flag);
    return TNF_OK;
}

__global__ void __launch_bounds__(256)
gated_copy_kernel(const int* __restrict__ gate, float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    if (*gate == 0) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

int tnf_gated_copy_f32(const int32_t* flag, float* dst, const float* src, int64_t n, void* stream) {
    if (!flag || !dst || !src || n < 0) return fail(TNF_EINVAL, "tnf_gated_copy_f32: NULL pointer or n=%lld", (long long)n);
    if (n == 0) return TNF_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gated_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const int*>(flag), dst, src, n);
    return check_launch("gated_copy");
}

int tnf_set_option(int32_t key, int32_t value) {
    if (key == TNF_OPT_FORCE_GENERIC) {
        g_force_generic = value;
        return TNF_OK;
    }
    if (key == TNF_OPT_OPERAND_PREC) {
        if (value != 0 && value != 1) return fail(TNF_EINVAL, "tnf_set_option: operand precision %d", value);
        g_operand_prec = value;
        return TNF_OK;
    }
    if (key == TNF_OPT_FLOW_VARIANT) {
        g_flow_variant = value;
        return TNF_OK;
    }
    if (key == TNF_OPT_TRAIN_BWD_FP32) {
        g_train_bwd_fp32 = value;
        return TNF_OK;
    }
    if (key == TNF_OPT_LAYER_VARIANT) {
        g_layer_variant = value;
        return TNF_OK;
    }
    if (key == TNF_OPT_COND_VARIANT) {
        g_cond_variant = value;
        return TNF_OK;
    }
    if (key == TNF_OPT_REV_VARIANT) {
        if (value != 0 && value != 1) return fail(TNF_EINVAL, "tnf_set_option: reversible-backward variant %d", value);
        g_rev_variant = value;
        return TNF_OK;
    }
    return fail(TNF_EINVAL, "tnf_set_option: unknown key %d", key);
}

int64_t tnf_diag_launch_count(int32_t family) {
    if (family < 0 || family >= TNF_DIAG_FAMILIES) return fail(TNF_EINVAL, "tnf_diag_launch_count: family %d", family);
    return g_diag_launches[family].load(std::memory_order_relaxed);
}

int tnf_get_option(int32_t key, int32_t* value) {
    if (!value) return fail(TNF_EINVAL, "tnf_get_option: value is NULL");
    switch (key) {
        case TNF_OPT_FORCE_GENERIC: *value = g_force_generic; return TNF_OK;
        case TNF_OPT_FLOW_VARIANT: *value = g_flow_variant; return TNF_OK;
        case TNF_OPT_LAYER_VARIANT: *value = g_layer_variant; return TNF_OK;
        case TNF_OPT_COND_VARIANT: *value = g_cond_variant; return TNF_OK;
        case TNF_OPT_TRAIN_BWD_FP32: *value = g_train_bwd_fp32; return TNF_OK;
        case TNF_OPT_OPERAND_PREC: *value = g_operand_prec; return TNF_OK;
        case TNF_OPT_REV_VARIANT: *value = g_rev_variant; return TNF_OK;
    }
    return fail(TNF_EINVAL, "tnf_get_option: unknown key %d", key);
}

int64_t tnf_coupling_num_params(int32_t D, int32_t L, int32_t U, int32_t upper) {
    if (D < 1 || L < 1 || U < 1) return fail(TNF_EINVAL, "tnf_coupling_num_params: D=%d L=%d U=%d", D, L, U);
    return coupling_num_params(D, L, U, upper);
}

int64_t tnf_flow_num_params(int32_t D, int32_t S, int32_t L, int32_t U) {
    if (D < 1 || S < 1 || L < 1 || U < 1)
        return fail(TNF_EINVAL, "tnf_flow_num_params: D=%d S=%d L=%d U=%d", D, S, L, U);
    return flow_layout(D, S, L, U).total;
}

int tnf_has_fast_path(int32_t D, int32_t L, int32_t U) {
    return (mfma_supported(D, L, U) || wide_supported(D, L, U)) ? 1 : 0;
}

int tnf_coupling(int32_t dtype, const void* z, const void* params, void* z_out, void* log_det,
                 int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t L, int32_t U,
                 int32_t upper, int32_t inverse, int64_t pstride, int32_t ld_mode, void* stream) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_coupling: dtype %d", dtype);
    int rc = check_mnd("tnf_coupling", M_z, M_p, N, D);
    if (rc) return rc;
    if (D < 2) return fail(TNF_EINVAL, "tnf_coupling: D=%d needs both halves non-empty", D);
    if (L < 1 || U < 1) return fail(TNF_EINVAL, "tnf_coupling: num_layers=%d num_units=%d", L, U);
    if (ld_mode != TNF_LD_STORE && ld_mode != TNF_LD_ADD && ld_mode != TNF_LD_SUB)
        return fail(TNF_EINVAL, "tnf_coupling: ld_mode %d", ld_mode);
    if (pstride < coupling_num_params(D, L, U, upper))
        return fail(TNF_EINVAL, "tnf_coupling: params row has %lld elements, layer needs %lld",
                    (long long)pstride, (long long)coupling_num_params(D, L, U, upper));
    if (!z || !params || !z_out || !log_det) return fail(TNF_EINVAL, "tnf_coupling: NULL pointer");
    if (z == z_out) return fail(TNF_EINVAL, "tnf_coupling: z_out must not alias z");
    if (N == 0) return TNF_OK;
    hipStream_t st = as_stream(stream);
    const int64_t Mmax = M_z > M_p ? M_z : M_p;
    if (dtype == TNF_F32 && !g_force_generic && mfma_supported(D, L, U) && (N >= 16 || Mmax >= 8) &&
        aligned16(z) && aligned16(z_out)) {
        MfmaLayerArgs a;
        memset(&a, 0, sizeof(a));
        a.z = (const float*)z;
        a.z_out = (float*)z_out;
        a.params = (const float*)params;
        a.pstride = pstride;
        a.ld_in = ld_mode == TNF_LD_STORE ? nullptr : (const float*)log_det;
        a.ld_out = (float*)log_det;
        a.ld_sign = ld_mode == TNF_LD_SUB ? -1.f : 1.f;
        a.Mz = M_z; a.Mp = M_p; a.N = N;
        a.D = D; a.L = L; a.U = U; a.upper = upper; a.inverse = inverse;
        return launch_coupling_mfma(a, st);
    }
    if (dtype == TNF_F32 && !g_force_generic && wide_supported(D, L, U) && N >= 16 && aligned16(z) && aligned16(z_out)) {
        MfmaLayerArgs a;
        memset(&a, 0, sizeof(a));
        a.z = (const float*)z;
        a.z_out = (float*)z_out;
        a.params = (const float*)params;
        a.pstride = pstride;
        a.ld_in = ld_mode == TNF_LD_STORE ? nullptr : (const float*)log_det;
        a.ld_out = (float*)log_det;
        a.ld_sign = ld_mode == TNF_LD_SUB ? -1.f : 1.f;
        a.Mz = M_z; a.Mp = M_p; a.N = N;
        a.D = D; a.L = L; a.U = U; a.upper = upper; a.inverse = inverse;
        return launch_coupling_wide(a, st);
    }
    return launch_coupling_generic(dtype, z, params, z_out, log_det, M_z, M_p, N, D, L, U, upper,
                                   inverse, pstride, ld_mode, st);
}

int tnf_affine(int32_t dtype, const void* z, const void* params, void* z_out, void* log_det,
               int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t inverse, int64_t pstride,
               void* stream) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_affine: dtype %d", dtype);
    int rc = check_mnd("tnf_affine", M_z, M_p, N, D);
    if (rc) return rc;
    if (pstride < 2 * (int64_t)D)
        return fail(TNF_EINVAL, "tnf_affine: params row has %lld elements, needs %d", (long long)pstride, 2 * D);
    if (!z || !params || !z_out || !log_det) return fail(TNF_EINVAL, "tnf_affine: NULL pointer");
    return launch_affine(dtype, z, params, z_out, log_det, M_z, M_p, N, D, inverse, pstride,
                         as_stream(stream));
}

int tnf_bn_apply(int32_t dtype, const void* z, const float* mean, const float* alpha, void* z_out,
                 float* log_det, int64_t rows, int32_t D, int32_t inverse, void* stream) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_bn_apply: dtype %d", dtype);
    if (rows < 0 || D < 1) return fail(TNF_EINVAL, "tnf_bn_apply: rows=%lld D=%d", (long long)rows, D);
    if (!z || !mean || !alpha || !z_out || !log_det) return fail(TNF_EINVAL, "tnf_bn_apply: NULL pointer");
    return launch_bn_apply(dtype, z, mean, alpha, z_out, log_det, rows, D, inverse, as_stream(stream));
}

int64_t tnf_bn_batch_workspace_bytes(int32_t D) {
    if (D < 1) return fail(TNF_EINVAL, "tnf_bn_batch_workspace_bytes: D=%d", D);
    return round16(2 * (int64_t)D * (int64_t)sizeof(double) + (int64_t)D * (int64_t)sizeof(float));
}

int tnf_bn_batch_forward_f32(const float* z, float* z_out, float* mean_out, float* alpha_out,
                             float* log_det, int64_t rows, int32_t D, float eps, void* workspace,
                             int64_t workspace_bytes, void* stream) {
    if (rows < 2 || D < 1)
        return fail(TNF_EINVAL, "tnf_bn_batch_forward_f32: rows=%lld (need > 1 value per feature) D=%d", (long long)rows, D);
    if (!z || !z_out || !mean_out || !alpha_out || !log_det || !workspace)
        return fail(TNF_EINVAL, "tnf_bn_batch_forward_f32: NULL pointer");
    if (workspace_bytes < tnf_bn_batch_workspace_bytes(D))
        return fail(TNF_EWORKSPACE, "tnf_bn_batch_forward_f32: workspace %lld < %lld", (long long)workspace_bytes,
                    (long long)tnf_bn_batch_workspace_bytes(D));
    return launch_bn_batch_forward(z, z_out, mean_out, alpha_out, log_det, rows, D, eps, workspace,
                                   as_stream(stream));
}

static int coupling_backward_impl(int32_t dtype, const void* z, const void* params, const void* g_z_out,
                          const void* g_log_det, void* g_z, void* g_params, int64_t M, int64_t M_p,
                          int64_t N, int32_t D, int32_t L, int32_t U, int32_t upper, int32_t inverse,
                          int64_t pstride, int64_t gpstride, void* stream, void* ws, int64_t ws_bytes) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_coupling_backward: dtype %d", dtype);
    if (M < 1 || N < 0 || D < 2 || L < 1 || U < 1 || (M_p != 1 && M_p != M))
        return fail(TNF_EINVAL, "tnf_coupling_backward: M=%lld M_p=%lld N=%lld D=%d L=%d U=%d", (long long)M,
                    (long long)M_p, (long long)N, D, L, U);
    const int64_t need = coupling_num_params(D, L, U, upper);
    if (pstride < need || gpstride < need)
        return fail(TNF_EINVAL, "tnf_coupling_backward: parameter rows (%lld / %lld) shorter than %lld",
                    (long long)pstride, (long long)gpstride, (long long)need);
    if (!z || !params || !g_z_out || !g_log_det || !g_z || !g_params)
        return fail(TNF_EINVAL, "tnf_coupling_backward: NULL pointer");
    if (N == 0) return TNF_OK;
    if (dtype == TNF_F32 && !g_force_generic && mfma_supported(D, L, U) && N >= 16 && aligned16(z) &&
        aligned16(g_z_out) && aligned16(g_z))
        return launch_coupling_backward_mfma((const float*)z, (const float*)params, (const float*)g_z_out,
                                             (const float*)g_log_det, (float*)g_z, (float*)g_params, M, M_p, N,
                                             D, L, U, upper, inverse, pstride, gpstride, as_stream(stream));
    // the wide shapes (num_units up to 64) with one shared parameter row: two-pass MFMA backward (coupling_wide_bwd.hip)
    if (dtype == TNF_F32 && !g_force_generic && M_p == 1 && ws && wide_bwd_supported(D, L, U) && aligned16(z) &&
        aligned16(g_z_out) && aligned16(g_z) && aligned16(ws) && ws_bytes >= wide_bwd_workspace(M * N, D, L, U))
        return launch_coupling_backward_wide((const float*)z, (const float*)params, (const float*)g_z_out,
                                             (const float*)g_log_det, (float*)g_z, (float*)g_params, M * N, D, L, U, upper,
                                             inverse, gpstride, ws, as_stream(stream));
    // (a caller that sized the workspace for the MFMA kernel -- 0 bytes -- but lands here, e.g. through unaligned
    // pointers, gets the legacy atomic reduction rather than an error)
    if (ws_bytes == 0 && coupling_backward_workspace(dtype, M, M_p, N, D, L, U, upper) > 0) ws_bytes = -1;
    return launch_coupling_backward(dtype, z, params, g_z_out, g_log_det, g_z, g_params, M, M_p, N, D, L, U,
                                    upper, inverse, pstride, gpstride, as_stream(stream), ws, ws_bytes);
}

int tnf_coupling_backward(int32_t dtype, const void* z, const void* params, const void* g_z_out,
                          const void* g_log_det, void* g_z, void* g_params, int64_t M, int64_t M_p,
                          int64_t N, int32_t D, int32_t L, int32_t U, int32_t upper, int32_t inverse,
                          int64_t pstride, int64_t gpstride, void* stream) {
    return coupling_backward_impl(dtype, z, params, g_z_out, g_log_det, g_z, g_params, M, M_p, N, D, L, U, upper, inverse,
                                  pstride, gpstride, stream, nullptr, -1);
}

int64_t tnf_coupling_backward_workspace_bytes(int32_t dtype, int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t L,
                                              int32_t U, int32_t upper) {
    if ((dtype != TNF_F32 && dtype != TNF_F64) || M < 1 || N < 0 || D < 2 || L < 1 || U < 1 || (M_p != 1 && M_p != M))
        return fail(TNF_EINVAL, "tnf_coupling_backward_workspace_bytes: dtype=%d M=%lld M_p=%lld N=%lld D=%d L=%d U=%d", dtype,
                    (long long)M, (long long)M_p, (long long)N, D, L, U);
    if (dtype == TNF_F32 && !g_force_generic && mfma_supported(D, L, U) && N >= 16) return 0;  // the MFMA kernel takes none
    const int64_t gen = coupling_backward_workspace(dtype, M, M_p, N, D, L, U, upper);
    if (dtype == TNF_F32 && !g_force_generic && M_p == 1 && wide_bwd_supported(D, L, U)) {
        const int64_t wide = wide_bwd_workspace(M * N, D, L, U);  // records of the two-pass MFMA backward + partial rows
        return wide > gen ? wide : gen;
    }
    return gen;
}

int tnf_coupling_backward_ws(int32_t dtype, const void* z, const void* params, const void* g_z_out,
                             const void* g_log_det, void* g_z, void* g_params, int64_t M, int64_t M_p,
                             int64_t N, int32_t D, int32_t L, int32_t U, int32_t upper, int32_t inverse,
                             int64_t pstride, int64_t gpstride, void* workspace, int64_t workspace_bytes, void* stream) {
    if (workspace_bytes < 0 || (workspace_bytes > 0 && !workspace))
        return fail(TNF_EINVAL, "tnf_coupling_backward_ws: workspace %p of %lld bytes", workspace, (long long)workspace_bytes);
    return coupling_backward_impl(dtype, z, params, g_z_out, g_log_det, g_z, g_params, M, M_p, N, D, L, U, upper, inverse,
                                  pstride, gpstride, stream, workspace, workspace_bytes);
}

int tnf_affine_backward(int32_t dtype, const void* z, const void* params, const void* g_z_out,
                        const void* g_log_det, void* g_z, void* g_params, int64_t M, int64_t M_p,
                        int64_t N, int32_t D, int32_t inverse, int64_t pstride, int64_t gpstride,
                        void* stream) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_affine_backward: dtype %d", dtype);
    if (M < 1 || N < 0 || D < 1 || (M_p != 1 && M_p != M))
        return fail(TNF_EINVAL, "tnf_affine_backward: M=%lld M_p=%lld N=%lld D=%d", (long long)M, (long long)M_p,
                    (long long)N, D);
    if (pstride < 2 * (int64_t)D || gpstride < 2 * (int64_t)D)
        return fail(TNF_EINVAL, "tnf_affine_backward: parameter rows shorter than %d", 2 * D);
    if (!params || !g_log_det || !g_params || (N > 0 && (!z || !g_z_out || !g_z)))
        return fail(TNF_EINVAL, "tnf_affine_backward: NULL pointer");
    return launch_affine_backward(dtype, z, params, g_z_out, g_log_det, g_z, g_params, M, M_p, N, D, inverse,
                                  pstride, gpstride, as_stream(stream));
}

int tnf_bn_apply_backward(int32_t dtype, const void* g_z_out, const float* alpha, void* g_z, int64_t rows,
                          int32_t D, int32_t inverse, void* stream) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_bn_apply_backward: dtype %d", dtype);
    if (rows < 0 || D < 1) return fail(TNF_EINVAL, "tnf_bn_apply_backward: rows=%lld D=%d", (long long)rows, D);
    if (!alpha || (rows > 0 && (!g_z_out || !g_z))) return fail(TNF_EINVAL, "tnf_bn_apply_backward: NULL pointer");
    return launch_bn_apply_backward(dtype, g_z_out, alpha, g_z, rows, D, inverse, as_stream(stream));
}

int tnf_bn_batch_backward_f32(const float* z_norm, const float* g_z_out, const float* g_log_det,
                              const float* alpha, float* g_z, int64_t rows, int32_t D, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    if (rows < 2 || D < 1) return fail(TNF_EINVAL, "tnf_bn_batch_backward_f32: rows=%lld D=%d", (long long)rows, D);
    if (!z_norm || !g_z_out || !alpha || !g_z || !workspace)
        return fail(TNF_EINVAL, "tnf_bn_batch_backward_f32: NULL pointer");
    if (workspace_bytes < tnf_bn_batch_workspace_bytes(D))
        return fail(TNF_EWORKSPACE, "tnf_bn_batch_backward_f32: workspace %lld < %lld", (long long)workspace_bytes,
                    (long long)tnf_bn_batch_workspace_bytes(D));
    return launch_bn_batch_backward(z_norm, g_z_out, g_log_det, alpha, g_z, rows, D, workspace, as_stream(stream));
}

int tnf_bn_batch_moments_f32(const float* z, double* moments, int64_t rows, int32_t D, void* stream) {
    if (rows < 0 || D < 1 || !moments || (rows > 0 && !z))
        return fail(TNF_EINVAL, "tnf_bn_batch_moments_f32: rows=%lld D=%d or NULL pointer", (long long)rows, D);
    return launch_bn_moments(z, moments, rows, D, as_stream(stream));
}

int tnf_bn_batch_normalize_f32(const float* z, const double* moments, float* z_out, float* mean_out, float* alpha_out,
                               float* log_det, int64_t rows, int32_t D, float eps, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    if (rows < 0 || D < 1) return fail(TNF_EINVAL, "tnf_bn_batch_normalize_f32: rows=%lld D=%d", (long long)rows, D);
    if (!moments || !mean_out || !alpha_out || !log_det || !workspace || (rows > 0 && (!z || !z_out)))
        return fail(TNF_EINVAL, "tnf_bn_batch_normalize_f32: NULL pointer");
    if (workspace_bytes < (int64_t)D * (int64_t)sizeof(float))
        return fail(TNF_EWORKSPACE, "tnf_bn_batch_normalize_f32: workspace %lld < %lld", (long long)workspace_bytes,
                    (long long)D * (long long)sizeof(float));
    return launch_bn_normalize_from_moments(z, moments, z_out, mean_out, alpha_out, log_det,
                                            reinterpret_cast<float*>(workspace), rows, D, eps, as_stream(stream));
}

int tnf_bn_batch_backward_sums_f32(const float* z_norm, const float* g_z_out, double* sums, int64_t rows, int32_t D,
                                   void* stream) {
    if (rows < 0 || D < 1 || !sums || (rows > 0 && (!z_norm || !g_z_out)))
        return fail(TNF_EINVAL, "tnf_bn_batch_backward_sums_f32: rows=%lld D=%d or NULL pointer", (long long)rows, D);
    return launch_bn_batch_backward_sums(z_norm, g_z_out, sums, rows, D, as_stream(stream));
}

int tnf_bn_batch_backward_apply_f32(const float* z_norm, const float* g_z_out, const float* g_log_det,
                                    const float* alpha, const double* sums, const double* count, float* g_z,
                                    int64_t rows, int32_t D, void* stream) {
    if (rows < 0 || D < 1 || !alpha || !sums || !count || (rows > 0 && (!z_norm || !g_z_out || !g_z)))
        return fail(TNF_EINVAL, "tnf_bn_batch_backward_apply_f32: rows=%lld D=%d or NULL pointer", (long long)rows, D);
    return launch_bn_batch_backward_apply(z_norm, g_z_out, g_log_det, alpha, sums, count, g_z, rows, D, as_stream(stream));
}

int64_t tnf_maf_num_params(int32_t D, int32_t L, int32_t U) {
    if (D < 1 || L < 1 || U < 1) return fail(TNF_EINVAL, "tnf_maf_num_params: D=%d L=%d U=%d", D, L, U);
    return 2 * (2 * (int64_t)D * U + (int64_t)(L - 1) * U * U);
}

int tnf_maf(int32_t dtype, const void* z, const void* params, const void* masks, void* z_out, void* log_det,
            int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t L, int32_t U, int32_t inverse, int64_t pstride,
            void* stream) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_maf: dtype %d", dtype);
    int rc = check_mnd("tnf_maf", M_z, M_p, N, D);
    if (rc) return rc;
    if (L < 1 || U < 1) return fail(TNF_EINVAL, "tnf_maf: num_layers=%d num_units=%d", L, U);
    if (pstride < tnf_maf_num_params(D, L, U))
        return fail(TNF_EINVAL, "tnf_maf: params row has %lld elements, layer needs %lld", (long long)pstride,
                    (long long)tnf_maf_num_params(D, L, U));
    if (!z || !params || !masks || !z_out || !log_det) return fail(TNF_EINVAL, "tnf_maf: NULL pointer");
    if (N == 0) return TNF_OK;
    if (dtype == TNF_F32 && !g_force_generic && maf_mfma_supported(D, L, U)) {
        MafArgs a = {};
        a.z = (const float*)z; a.z_out = (float*)z_out; a.params = (const float*)params; a.masks = (const float*)masks;
        a.pstride = pstride; a.ld_out = (float*)log_det; a.ld_sign = 1.f;
        a.Mz = M_z; a.Mp = M_p; a.N = N; a.D = D; a.L = L; a.U = U; a.inverse = inverse;
        return launch_maf_mfma(a, as_stream(stream));
    }
    return launch_maf(dtype, z, params, masks, z_out, log_det, M_z, M_p, N, D, L, U, inverse, pstride, as_stream(stream));
}

int tnf_ar_flow_supported(int32_t D, int32_t L, int32_t U) { return maf_mfma_supported(D, L, U) ? 1 : 0; }

int64_t tnf_ar_flow_workspace_bytes(int64_t M_p, int32_t D) {
    if (M_p < 1 || D < 1) return fail(TNF_EINVAL, "tnf_ar_flow_workspace_bytes: M_p=%lld D=%d", (long long)M_p, D);
    return round16(M_p * (2 * (int64_t)D + 1) * (int64_t)sizeof(float));
}

static int ar_flow_run(const char* fn, int inverse, const float* z, const float* params, const float* masks,
                       const float* bn_mean, const float* bn_alpha, const float* interval_consts, float* z_out,
                       float* sum_log_det, float* log_prob,
                       int64_t M_z, int64_t M_p, int64_t N, int D, int L, int U, int64_t pstride, void* workspace,
                       int64_t workspace_bytes, void* stream) {
    int rc = check_mnd(fn, M_z, M_p, N, D);
    if (rc) return rc;
    if (!maf_mfma_supported(D, L, U)) return fail(TNF_EUNSUPPORTED, "%s: no kernel for D=%d L=%d U=%d", fn, D, L, U);
    const int64_t p_maf = tnf_maf_num_params(D, L, U);
    if (pstride < p_maf + 2 * (int64_t)D)
        return fail(TNF_EINVAL, "%s: params row has %lld elements, the flow needs %lld", fn, (long long)pstride,
                    (long long)(p_maf + 2 * D));
    if (!z || !params || !masks || !bn_mean || !bn_alpha || !workspace) return fail(TNF_EINVAL, "%s: NULL pointer", fn);
    if (workspace_bytes < tnf_ar_flow_workspace_bytes(M_p, D))
        return fail(TNF_EWORKSPACE, "%s: workspace %lld < %lld", fn, (long long)workspace_bytes,
                    (long long)tnf_ar_flow_workspace_bytes(M_p, D));
    if (N == 0) return TNF_OK;
    float* fold = (float*)workspace;
    float* ldc = fold + M_p * 2 * (int64_t)D;
    rc = launch_ar_fold(params, pstride, p_maf, bn_mean, bn_alpha, fold, ldc, M_p, D, inverse, as_stream(stream));
    if (rc) return rc;
    MafArgs a = {};
    a.z = z; a.z_out = z_out; a.params = params; a.masks = masks; a.pstride = pstride;
    if (inverse) a.pre = fold; else a.post = fold;
    a.fold_stride = 2 * (int64_t)D;
    a.ld_out = sum_log_det; a.ld_sign = 1.f; a.ldc = ldc; a.add_ldc = 1; a.log_prob = log_prob;
    a.iv = interval_consts;
    a.Mz = M_z; a.Mp = M_p; a.N = N; a.D = D; a.L = L; a.U = U; a.inverse = inverse;
    a.bf16 = g_operand_prec == 1;
    return launch_maf_mfma(a, as_stream(stream));
}

int tnf_ar_flow_log_prob_f32(const float* z, const float* params, const float* masks, const float* bn_mean,
                             const float* bn_alpha, const float* interval_consts, float* log_prob, float* z0,
                             float* sum_log_det, int64_t M_z,
                             int64_t M_p, int64_t N, int32_t D, int32_t L, int32_t U, int64_t pstride, void* workspace,
                             int64_t workspace_bytes, void* stream) {
    if (!log_prob && !z0 && !sum_log_det) return fail(TNF_EINVAL, "tnf_ar_flow_log_prob_f32: no output requested");
    return ar_flow_run("tnf_ar_flow_log_prob_f32", 1, z, params, masks, bn_mean, bn_alpha, interval_consts, z0, sum_log_det,
                       log_prob, M_z,
                       M_p, N, D, L, U, pstride, workspace, workspace_bytes, stream);
}

int tnf_ar_flow_forward_f32(const float* omega, const float* params, const float* masks, const float* bn_mean,
                            const float* bn_alpha, const float* interval_consts, float* z_out, float* sum_log_det,
                            int64_t M_z, int64_t M_p, int64_t N,
                            int32_t D, int32_t L, int32_t U, int64_t pstride, void* workspace, int64_t workspace_bytes,
                            void* stream) {
    if (!z_out || !sum_log_det) return fail(TNF_EINVAL, "tnf_ar_flow_forward_f32: NULL pointer");
    return ar_flow_run("tnf_ar_flow_forward_f32", 0, omega, params, masks, bn_mean, bn_alpha, interval_consts, z_out,
                       sum_log_det, nullptr,
                       M_z, M_p, N, D, L, U, pstride, workspace, workspace_bytes, stream);
}

// ---- training through NormFlow('AR').log_prob: parameter gradients of the whole stack in one backward kernel ----
int tnf_ar_flow_train_supported(int32_t D, int32_t L, int32_t U) {
    return (maf_mfma_supported(D, L, U) && maf_bwd_mfma_supported(D, L, U)) ? 1 : 0;
}

int64_t tnf_ar_flow_bwd_workspace_bytes(int64_t M_p, int32_t D) {
    if (M_p < 1 || D < 1) return fail(TNF_EINVAL, "tnf_ar_flow_bwd_workspace_bytes: M_p=%lld D=%d", (long long)M_p, D);
    return round16((M_p * (4 * (int64_t)D + 2) + 1) * (int64_t)sizeof(float));
}

int tnf_ar_flow_log_prob_bwd_f32(const float* z, const float* params, const float* masks, const float* bn_mean,
                                 const float* bn_alpha, const float* interval_consts, const float* g_log_prob,
                                 float* g_params, int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t L, int32_t U,
                                 int64_t pstride, int64_t gpstride, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
    const char* fn = "tnf_ar_flow_log_prob_bwd_f32";
    if (M < 1 || N < 0 || (M_p != 1 && M_p != M))
        return fail(TNF_EINVAL, "%s: M=%lld M_p=%lld N=%lld", fn, (long long)M, (long long)M_p, (long long)N);
    if (!tnf_ar_flow_train_supported(D, L, U)) return fail(TNF_EUNSUPPORTED, "%s: no kernel for D=%d L=%d U=%d", fn, D, L, U);
    const int64_t p_maf = tnf_maf_num_params(D, L, U);
    if (pstride < p_maf + 2 * (int64_t)D || gpstride < p_maf + 2 * (int64_t)D)
        return fail(TNF_EINVAL, "%s: parameter rows shorter than %lld", fn, (long long)(p_maf + 2 * D));
    if (!z || !params || !masks || !bn_mean || !bn_alpha || !g_log_prob || !g_params || !workspace)
        return fail(TNF_EINVAL, "%s: NULL pointer", fn);
    if (workspace_bytes < tnf_ar_flow_bwd_workspace_bytes(M_p, D))
        return fail(TNF_EWORKSPACE, "%s: workspace %lld < %lld", fn, (long long)workspace_bytes,
                    (long long)tnf_ar_flow_bwd_workspace_bytes(M_p, D));
    if (N == 0) return TNF_OK;
    float* fold = (float*)workspace;
    float* ldc = fold + M_p * 2 * (int64_t)D;
    float* g_fold = ldc + M_p;
    float* glp_sum = g_fold + M_p * 2 * (int64_t)D;
    int rc = launch_ar_fold(params, pstride, p_maf, bn_mean, bn_alpha, fold, ldc, M_p, D, 1, as_stream(stream));
    if (rc) return rc;
    return launch_ar_flow_backward(z, params, masks, fold, interval_consts, g_log_prob, g_params, g_fold, glp_sum, M, M_p,
                                   N, D, L, U, pstride, gpstride, as_stream(stream));
}

/* MAF.inverse_and_log_det with the per-dimension f_alpha(z) (M,N,D) as an extra output (generic kernel, float32 /
 * float64): the diagonal of the inverse map's Jacobian is e^-alpha, which the backward of the SAMPLING direction
 * needs (ops._MafFn: implicit differentiation of G(x, theta) = omega through the inverse-direction backward). */
int tnf_maf_inverse_alpha(int32_t dtype, const void* z, const void* params, const void* masks, void* z_out, void* log_det,
                          void* alpha_out, int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t L, int32_t U,
                          int64_t pstride, void* stream) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_maf_inverse_alpha: dtype %d", dtype);
    int rc = check_mnd("tnf_maf_inverse_alpha", M_z, M_p, N, D);
    if (rc) return rc;
    if (L < 1 || U < 1) return fail(TNF_EINVAL, "tnf_maf_inverse_alpha: L=%d U=%d", L, U);
    if (pstride < tnf_maf_num_params(D, L, U))
        return fail(TNF_EINVAL, "tnf_maf_inverse_alpha: params row has %lld elements, layer needs %lld", (long long)pstride,
                    (long long)tnf_maf_num_params(D, L, U));
    if (!z || !params || !masks || !z_out || !log_det || !alpha_out) return fail(TNF_EINVAL, "tnf_maf_inverse_alpha: NULL pointer");
    if (N == 0) return TNF_OK;
    return launch_maf(dtype, z, params, masks, z_out, log_det, M_z, M_p, N, D, L, U, 1, pstride, as_stream(stream), alpha_out);
}

static int maf_backward_impl(int32_t dtype, const void* z, const void* params, const void* masks, const void* g_z_out,
                     const void* g_log_det, void* g_z, void* g_params, int64_t M, int64_t M_p, int64_t N, int32_t D,
                     int32_t L, int32_t U, int64_t pstride, int64_t gpstride, void* stream, void* ws, int64_t ws_bytes) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_maf_backward: dtype %d", dtype);
    if (M < 1 || N < 0 || D < 1 || L < 1 || U < 1 || (M_p != 1 && M_p != M))
        return fail(TNF_EINVAL, "tnf_maf_backward: M=%lld M_p=%lld N=%lld D=%d L=%d U=%d", (long long)M,
                    (long long)M_p, (long long)N, D, L, U);
    const int64_t need = tnf_maf_num_params(D, L, U);
    if (pstride < need || gpstride < need) return fail(TNF_EINVAL, "tnf_maf_backward: parameter rows shorter than %lld", (long long)need);
    if (!z || !params || !masks || !g_z_out || !g_log_det || !g_z || !g_params)
        return fail(TNF_EINVAL, "tnf_maf_backward: NULL pointer");
    if (N == 0) return TNF_OK;
    if (dtype == TNF_F32 && !g_force_generic && maf_bwd_mfma_supported(D, L, U))
        return launch_maf_backward_mfma((const float*)z, (const float*)params, (const float*)masks, (const float*)g_z_out,
                                        (const float*)g_log_det, (float*)g_z, (float*)g_params, M, M_p, N, D, L, U,
                                        pstride, gpstride, as_stream(stream));
    // D > 32 (the one-kernel matrix-pipe backward stops there) with one shared parameter row: the two-pass MFMA backward
    if (dtype == TNF_F32 && !g_force_generic && M_p == 1 && ws && maf_wide_bwd_supported(D, L, U) && aligned16(z) &&
        aligned16(g_z_out) && aligned16(g_z) && aligned16(ws) && ws_bytes >= maf_wide_bwd_workspace(M * N, D, L, U))
        return launch_maf_backward_wide((const float*)z, (const float*)params, (const float*)masks, (const float*)g_z_out,
                                        (const float*)g_log_det, (float*)g_z, (float*)g_params, M * N, D, L, U, gpstride, ws,
                                        as_stream(stream));
    if (ws_bytes == 0 && maf_backward_workspace(dtype, M, M_p, N, D, L, U) > 0) ws_bytes = -1;
    return launch_maf_backward(dtype, z, params, masks, g_z_out, g_log_det, g_z, g_params, M, M_p, N, D, L, U,
                               pstride, gpstride, as_stream(stream), ws, ws_bytes);
}

int tnf_maf_backward(int32_t dtype, const void* z, const void* params, const void* masks, const void* g_z_out,
                     const void* g_log_det, void* g_z, void* g_params, int64_t M, int64_t M_p, int64_t N, int32_t D,
                     int32_t L, int32_t U, int64_t pstride, int64_t gpstride, void* stream) {
    return maf_backward_impl(dtype, z, params, masks, g_z_out, g_log_det, g_z, g_params, M, M_p, N, D, L, U, pstride,
                             gpstride, stream, nullptr, -1);
}

int64_t tnf_maf_backward_workspace_bytes(int32_t dtype, int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t L, int32_t U) {
    if ((dtype != TNF_F32 && dtype != TNF_F64) || M < 1 || N < 0 || D < 1 || L < 1 || U < 1 || (M_p != 1 && M_p != M))
        return fail(TNF_EINVAL, "tnf_maf_backward_workspace_bytes: dtype=%d M=%lld M_p=%lld N=%lld D=%d L=%d U=%d", dtype,
                    (long long)M, (long long)M_p, (long long)N, D, L, U);
    if (dtype == TNF_F32 && !g_force_generic && maf_bwd_mfma_supported(D, L, U)) return 0;  // the MFMA kernel takes none
    const int64_t gen = maf_backward_workspace(dtype, M, M_p, N, D, L, U);
    if (dtype == TNF_F32 && !g_force_generic && M_p == 1 && maf_wide_bwd_supported(D, L, U)) {
        const int64_t wide = maf_wide_bwd_workspace(M * N, D, L, U);
        return wide > gen ? wide : gen;
    }
    return gen;
}

int tnf_maf_backward_ws(int32_t dtype, const void* z, const void* params, const void* masks, const void* g_z_out,
                        const void* g_log_det, void* g_z, void* g_params, int64_t M, int64_t M_p, int64_t N, int32_t D,
                        int32_t L, int32_t U, int64_t pstride, int64_t gpstride, void* workspace, int64_t workspace_bytes,
                        void* stream) {
    if (workspace_bytes < 0 || (workspace_bytes > 0 && !workspace))
        return fail(TNF_EINVAL, "tnf_maf_backward_ws: workspace %p of %lld bytes", workspace, (long long)workspace_bytes);
    return maf_backward_impl(dtype, z, params, masks, g_z_out, g_log_det, g_z, g_params, M, M_p, N, D, L, U, pstride,
                             gpstride, stream, workspace, workspace_bytes);
}

int tnf_cond_flow_supported(int32_t D, int32_t S, int32_t L, int32_t U, int32_t H) {
    return cond_flow_supported(D, S, L, U, H) ? 1 : 0;
}

int64_t tnf_cond_flow_workspace_bytes(int32_t D, int32_t S, int32_t L, int32_t U, int32_t H) {
    if (!cond_flow_supported(D, S, L, U, H))
        return fail(TNF_EUNSUPPORTED, "tnf_cond_flow_workspace_bytes: no kernel for D=%d S=%d L=%d U=%d H=%d", D, S, L, U, H);
    return cond_flow_workspace(D, S, L, U, H);
}

int tnf_cond_flow_log_prob_f32(const float* z, const float* h, const float* W, const float* b,
                               const float* bn_mean, const float* bn_alpha, float* log_prob, float* z0,
                               float* sum_log_det, int64_t M, int32_t D, int32_t S, int32_t L, int32_t U, int32_t H,
                               int64_t ldh, int64_t ldw, void* workspace, int64_t workspace_bytes, void* stream) {
    if (M < 0) return fail(TNF_EINVAL, "tnf_cond_flow_log_prob_f32: M=%lld", (long long)M);
    if (!cond_flow_supported(D, S, L, U, H))
        return fail(TNF_EUNSUPPORTED, "tnf_cond_flow_log_prob_f32: no kernel for D=%d S=%d L=%d U=%d H=%d", D, S, L, U, H);
    if (ldh < H || ldw < H || (ldh & 3) || (ldw & 3))
        return fail(TNF_EINVAL, "tnf_cond_flow_log_prob_f32: ldh=%lld ldw=%lld must be multiples of 4 and >= H=%d",
                    (long long)ldh, (long long)ldw, H);
    if (M == 0) return TNF_OK;
    if (!z || !h || !W || !b || !bn_mean || !bn_alpha || !log_prob || !workspace)
        return fail(TNF_EINVAL, "tnf_cond_flow_log_prob_f32: NULL pointer");
    if (((uintptr_t)h & 15) || ((uintptr_t)W & 15) || ((uintptr_t)z & 15) || ((uintptr_t)workspace & 255))
        return fail(TNF_EINVAL, "tnf_cond_flow_log_prob_f32: z, h, W must be 16-byte and the workspace 256-byte aligned");
    const int64_t need = cond_flow_workspace(D, S, L, U, H);
    if (workspace_bytes < need)
        return fail(TNF_EWORKSPACE, "tnf_cond_flow_log_prob_f32: workspace %lld < %lld", (long long)workspace_bytes,
                    (long long)need);
    return launch_cond_flow_log_prob(z, h, W, b, bn_mean, bn_alpha, log_prob, z0, sum_log_det, nullptr, M, D, S, L, U,
                                     H, ldh, ldw, workspace, as_stream(stream));
}

int tnf_cond_flow_forward_f32(const float* omega, const float* h, const float* W, const float* b,
                              const float* bn_mean, const float* bn_alpha, float* z_out, float* sum_log_det, int64_t M,
                              int32_t D, int32_t S, int32_t L, int32_t U, int32_t H, int64_t ldh, int64_t ldw,
                              void* workspace, int64_t workspace_bytes, void* stream) {
    if (M < 0) return fail(TNF_EINVAL, "tnf_cond_flow_forward_f32: M=%lld", (long long)M);
    if (!cond_flow_supported(D, S, L, U, H))
        return fail(TNF_EUNSUPPORTED, "tnf_cond_flow_forward_f32: no kernel for D=%d S=%d L=%d U=%d H=%d", D, S, L, U, H);
    if (ldh < H || ldw < H || (ldh & 3) || (ldw & 3))
        return fail(TNF_EINVAL, "tnf_cond_flow_forward_f32: ldh=%lld ldw=%lld must be multiples of 4 and >= H=%d",
                    (long long)ldh, (long long)ldw, H);
    if (M == 0) return TNF_OK;
    if (!omega || !h || !W || !b || !bn_mean || !bn_alpha || !z_out || !sum_log_det || !workspace)
        return fail(TNF_EINVAL, "tnf_cond_flow_forward_f32: NULL pointer");
    if (((uintptr_t)h & 15) || ((uintptr_t)W & 15) || ((uintptr_t)omega & 15) || ((uintptr_t)z_out & 15) ||
        ((uintptr_t)workspace & 255))
        return fail(TNF_EINVAL, "tnf_cond_flow_forward_f32: omega, z_out, h, W must be 16-byte and the workspace 256-byte aligned");
    const int64_t need = cond_flow_workspace(D, S, L, U, H);
    if (workspace_bytes < need)
        return fail(TNF_EWORKSPACE, "tnf_cond_flow_forward_f32: workspace %lld < %lld", (long long)workspace_bytes,
                    (long long)need);
    return launch_cond_flow_forward(omega, h, W, b, bn_mean, bn_alpha, z_out, sum_log_det, M, D, S, L, U, H, ldh, ldw,
                                    workspace, as_stream(stream));
}

int64_t tnf_cond_flow_acts_floats(int64_t M, int32_t D, int32_t S, int32_t L) {
    if (M < 0 || D < 2 || S < 1 || L < 1) return fail(TNF_EINVAL, "tnf_cond_flow_acts_floats: M=%lld D=%d S=%d L=%d", (long long)M, D, S, L);
    return cond_acts_floats(M, D, S, L);
}

int64_t tnf_cond_flow_deltas_floats(int64_t M, int32_t D, int32_t S, int32_t L, int32_t H) {
    if (M < 0 || D < 2 || S < 1 || L < 1 || H < 1)
        return fail(TNF_EINVAL, "tnf_cond_flow_deltas_floats: M=%lld D=%d S=%d L=%d H=%d", (long long)M, D, S, L, H);
    return cond_deltas_floats(M, D, S, L, H);
}

int64_t tnf_cond_flow_bwd_workspace_bytes(int32_t D, int32_t S, int32_t L, int32_t U, int32_t H) {
    if (!cond_flow_supported(D, S, L, U, H))
        return fail(TNF_EUNSUPPORTED, "tnf_cond_flow_bwd_workspace_bytes: no kernel for D=%d S=%d L=%d U=%d H=%d", D, S, L, U, H);
    return cond_flow_bwd_workspace(D, S, L, U, H);
}

static int cond_train_checks(const char* fn, int64_t M, int D, int S, int L, int U, int H, int64_t ldh, int64_t ldw,
                             const void* h, const void* W, const void* ws) {
    if (M < 0) return fail(TNF_EINVAL, "%s: M=%lld", fn, (long long)M);
    if (!cond_flow_supported(D, S, L, U, H))
        return fail(TNF_EUNSUPPORTED, "%s: no kernel for D=%d S=%d L=%d U=%d H=%d", fn, D, S, L, U, H);
    if (ldh < H || ldw < H || (ldh & 3) || (ldw & 3))
        return fail(TNF_EINVAL, "%s: ldh=%lld ldw=%lld must be multiples of 4 and >= H=%d", fn, (long long)ldh, (long long)ldw, H);
    if (((uintptr_t)h & 15) || ((uintptr_t)W & 15) || ((uintptr_t)ws & 255))
        return fail(TNF_EINVAL, "%s: h, W must be 16-byte and the workspace 256-byte aligned", fn);
    return TNF_OK;
}

int tnf_cond_flow_log_prob_fwd_f32(const float* z, const float* h, const float* W, const float* b,
                                   const float* bn_mean, const float* bn_alpha, float* log_prob, float* acts,
                                   int64_t M, int32_t D, int32_t S, int32_t L, int32_t U, int32_t H, int64_t ldh,
                                   int64_t ldw, void* workspace, int64_t workspace_bytes, void* stream) {
    int rc = cond_train_checks("tnf_cond_flow_log_prob_fwd_f32", M, D, S, L, U, H, ldh, ldw, h, W, workspace);
    if (rc) return rc;
    if (M == 0) return TNF_OK;
    if (!z || !h || !W || !b || !bn_mean || !bn_alpha || !log_prob || !acts || !workspace)
        return fail(TNF_EINVAL, "tnf_cond_flow_log_prob_fwd_f32: NULL pointer");
    if (((uintptr_t)z & 15) || ((uintptr_t)acts & 15))
        return fail(TNF_EINVAL, "tnf_cond_flow_log_prob_fwd_f32: z and acts must be 16-byte aligned");
    if (workspace_bytes < cond_flow_workspace(D, S, L, U, H))
        return fail(TNF_EWORKSPACE, "tnf_cond_flow_log_prob_fwd_f32: workspace %lld < %lld", (long long)workspace_bytes,
                    (long long)cond_flow_workspace(D, S, L, U, H));
    return launch_cond_flow_log_prob(z, h, W, b, bn_mean, bn_alpha, log_prob, nullptr, nullptr, acts, M, D, S, L, U, H,
                                     ldh, ldw, workspace, as_stream(stream));
}

int tnf_cond_flow_log_prob_bwd_f32(const float* g_log_prob, const float* h, const float* W, const float* b,
                                   const float* bn_mean, const float* bn_alpha, const float* acts, float* deltas,
                                   float* g_h, float* g_W, float* g_b, float* g_z, int64_t M, int32_t D, int32_t S,
                                   int32_t L, int32_t U, int32_t H, int64_t ldh, int64_t ldw, int64_t ldgh,
                                   int64_t ldgw, void* workspace, int64_t workspace_bytes, void* stream) {
    int rc = cond_train_checks("tnf_cond_flow_log_prob_bwd_f32", M, D, S, L, U, H, ldh, ldw, h, W, workspace);
    if (rc) return rc;
    if (ldgh < H || ldgw < H || (ldgh & 3))
        return fail(TNF_EINVAL, "tnf_cond_flow_log_prob_bwd_f32: ldgh=%lld ldgw=%lld", (long long)ldgh, (long long)ldgw);
    if (!g_W || !g_b) return fail(TNF_EINVAL, "tnf_cond_flow_log_prob_bwd_f32: NULL pointer");
    if (M == 0) {
        const int64_t P = flow_layout(D, S, L, U).total;
        if (hipMemsetAsync(g_W, 0, (size_t)P * ldgw * sizeof(float), as_stream(stream)) != hipSuccess ||
            hipMemsetAsync(g_b, 0, (size_t)P * sizeof(float), as_stream(stream)) != hipSuccess)
            return fail(TNF_ELAUNCH, "tnf_cond_flow_log_prob_bwd_f32: memset failed");
        return TNF_OK;
    }
    if (!g_log_prob || !h || !W || !b || !bn_mean || !bn_alpha || !acts || !deltas || !g_h || !workspace)
        return fail(TNF_EINVAL, "tnf_cond_flow_log_prob_bwd_f32: NULL pointer");
    if (((uintptr_t)acts & 15) || ((uintptr_t)deltas & 15) || ((uintptr_t)g_h & 15) || ((uintptr_t)g_z & 15))
        return fail(TNF_EINVAL, "tnf_cond_flow_log_prob_bwd_f32: acts, deltas, g_h, g_z must be 16-byte aligned");
    if (workspace_bytes < cond_flow_bwd_workspace(D, S, L, U, H))
        return fail(TNF_EWORKSPACE, "tnf_cond_flow_log_prob_bwd_f32: workspace %lld < %lld", (long long)workspace_bytes,
                    (long long)cond_flow_bwd_workspace(D, S, L, U, H));
    return launch_cond_flow_backward(g_log_prob, h, W, b, bn_mean, bn_alpha, acts, deltas, g_h, g_W, g_b, g_z, M, D, S, L,
                                     U, H, ldh, ldw, ldgh, ldgw, workspace, as_stream(stream));
}

int tnf_to_interval(int32_t dtype, const void* z, const float* consts, void* z_out, void* log_det, int64_t rows,
                    int32_t D, int32_t inverse, void* stream) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_to_interval: dtype %d", dtype);
    if (rows < 0 || D < 1) return fail(TNF_EINVAL, "tnf_to_interval: rows=%lld D=%d", (long long)rows, D);
    if (rows == 0) return TNF_OK;
    if (!z || !consts || !z_out || !log_det) return fail(TNF_EINVAL, "tnf_to_interval: NULL pointer");
    return launch_to_interval(dtype, z, consts, z_out, log_det, rows, D, inverse, as_stream(stream));
}

int tnf_to_interval_backward(int32_t dtype, const void* z, const float* consts, const void* g_z_out,
                             const void* g_log_det, void* g_z, int64_t rows, int32_t D, int32_t inverse,
                             void* stream) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_to_interval_backward: dtype %d", dtype);
    if (rows < 0 || D < 1) return fail(TNF_EINVAL, "tnf_to_interval_backward: rows=%lld D=%d", (long long)rows, D);
    if (rows == 0) return TNF_OK;
    if (!z || !consts || !g_z_out || !g_log_det || !g_z)
        return fail(TNF_EINVAL, "tnf_to_interval_backward: NULL pointer");
    return launch_to_interval_backward(dtype, z, consts, g_z_out, g_log_det, g_z, rows, D, inverse, as_stream(stream));
}

int tnf_to_simplex(int32_t dtype, const void* z, void* z_out, void* log_det, int64_t rows, int32_t D_in,
                   int32_t D_attr, void* stream) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_to_simplex: dtype %d", dtype);
    if (rows < 0 || D_in < 1 || D_attr < 1)
        return fail(TNF_EINVAL, "tnf_to_simplex: rows=%lld D_in=%d D_attr=%d", (long long)rows, D_in, D_attr);
    if (rows == 0) return TNF_OK;
    if (!z || !z_out || !log_det) return fail(TNF_EINVAL, "tnf_to_simplex: NULL pointer");
    return launch_to_simplex(dtype, z, z_out, log_det, rows, D_in, D_attr, as_stream(stream));
}

int tnf_to_simplex_backward(int32_t dtype, const void* z, const void* g_z_out, const void* g_log_det, void* g_z,
                            int64_t rows, int32_t D_in, int32_t D_attr, void* stream) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_to_simplex_backward: dtype %d", dtype);
    if (rows < 0 || D_in < 1 || D_attr < 1)
        return fail(TNF_EINVAL, "tnf_to_simplex_backward: rows=%lld D_in=%d D_attr=%d", (long long)rows, D_in, D_attr);
    if (rows == 0) return TNF_OK;
    if (!z || !g_z_out || !g_log_det || !g_z) return fail(TNF_EINVAL, "tnf_to_simplex_backward: NULL pointer");
    return launch_to_simplex_backward(dtype, z, g_z_out, g_log_det, g_z, rows, D_in, D_attr, as_stream(stream));
}

int tnf_base_log_density_f64(int32_t dtype, const void* omega, double* out, int64_t rows, int32_t D,
                             void* stream) {
    if (dtype != TNF_F32 && dtype != TNF_F64) return fail(TNF_EINVAL, "tnf_base_log_density_f64: dtype %d", dtype);
    if (rows < 0 || D < 1) return fail(TNF_EINVAL, "tnf_base_log_density_f64: rows=%lld D=%d", (long long)rows, D);
    if (!omega || !out) return fail(TNF_EINVAL, "tnf_base_log_density_f64: NULL pointer");
    return launch_base_log_density(dtype, omega, out, rows, D, as_stream(stream));
}

int tnf_flow_fused_supported(int32_t D, int32_t S, int32_t L, int32_t U) {
    return flow_fused_supported(D, S, L, U) ? 1 : 0;
}

int64_t tnf_flow_workspace_bytes(int64_t M, int64_t N, int32_t D, int32_t S, int32_t L, int32_t U,
                                 int32_t fusion) {
    if (M < 1 || N < 0 || D < 1 || S < 1 || L < 1 || U < 1)
        return fail(TNF_EINVAL, "tnf_flow_workspace_bytes: M=%lld N=%lld D=%d S=%d L=%d U=%d", (long long)M,
                    (long long)N, D, S, L, U);
    if (!mfma_supported(D, L, U) && !wide_supported(D, L, U))
        return fail(TNF_EUNSUPPORTED, "tnf_flow_workspace_bytes: no fused kernel for D=%d L=%d U=%d", D, L, U);
    const FlowWs w = flow_ws(M, N, D, S, L, U);
    return fusion == TNF_FUSE_FLOW ? w.zbuf : w.total;  // the whole-flow kernel needs no z / log-det scratch
}

static int flow_common_checks(const char* fn, int64_t M_z, int64_t M_p, int64_t N, int D, int S,
                              int L, int U, int64_t pstride, int fusion, const void* ws,
                              int64_t ws_bytes, int* use_fused) {
    int rc = check_mnd(fn, M_z, M_p, N, D);
    if (rc) return rc;
    if (S < 1 || L < 1 || U < 1) return fail(TNF_EINVAL, "%s: S=%d L=%d U=%d", fn, S, L, U);
    if (pstride < flow_layout(D, S, L, U).total)
        return fail(TNF_EINVAL, "%s: params row has %lld elements, flow needs %lld", fn,
                    (long long)pstride, (long long)flow_layout(D, S, L, U).total);
    if (!mfma_supported(D, L, U) && !wide_supported(D, L, U))
        return fail(TNF_EUNSUPPORTED, "%s: no fused kernel for D=%d L=%d U=%d (compose bijector-level calls)", fn, D, L, U);
    if (fusion == TNF_FUSE_AUTO) *use_fused = flow_fused_supported(D, S, L, U) ? 1 : 0;
    else if (fusion == TNF_FUSE_FLOW) {
        if (!flow_fused_supported(D, S, L, U))
            return fail(TNF_EUNSUPPORTED, "%s: whole-flow kernel unavailable for D=%d S=%d L=%d U=%d", fn, D, S, L, U);
        *use_fused = 1;
    } else if (fusion == TNF_FUSE_LAYER) *use_fused = 0;
    else return fail(TNF_EINVAL, "%s: fusion %d", fn, fusion);
    const int64_t M = M_z > M_p ? M_z : M_p;
    const FlowWs w = flow_ws(M, N, D, S, L, U);
    const int64_t need = *use_fused ? w.zbuf : w.total;
    if (!ws || ws_bytes < need)
        return fail(TNF_EWORKSPACE, "%s: workspace %lld < %lld", fn, (long long)ws_bytes, (long long)need);
    return TNF_OK;
}

static int flow_log_prob_impl(const float* z, const float* params, const float* bn_mean,
                          const float* bn_alpha, const float* interval_consts, float* log_prob, float* z0,
                          float* sum_log_det,
                          int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t S, int32_t L,
                          int32_t U, int64_t pstride, int32_t fusion, void* workspace,
                          int64_t workspace_bytes, void* stream, uint32_t* exact_reruns) {
    int use_fused = 0;
    int rc = flow_common_checks("tnf_flow_log_prob_f32", M_z, M_p, N, D, S, L, U, pstride, fusion,
                                workspace, workspace_bytes, &use_fused);
    if (rc) return rc;
    if (!z || !params || !bn_mean || !bn_alpha) return fail(TNF_EINVAL, "tnf_flow_log_prob_f32: NULL pointer");
    if (!log_prob && !z0 && !sum_log_det) return fail(TNF_EINVAL, "tnf_flow_log_prob_f32: no output requested");
    if (!aligned16(z) || (z0 && !aligned16(z0)))
        return fail(TNF_EINVAL, "tnf_flow_log_prob_f32: z / z0 must be 16-byte aligned");
    if (z0 == z) return fail(TNF_EINVAL, "tnf_flow_log_prob_f32: z0 must not alias z");
    if (N == 0) return TNF_OK;
    hipStream_t st = as_stream(stream);
    const int64_t M = M_z > M_p ? M_z : M_p;
    const FlowWs w = flow_ws(M, N, D, S, L, U);
    char* wsb = reinterpret_cast<char*>(workspace);
    float* fold = reinterpret_cast<float*>(wsb + w.fold);
    float* ldc = reinterpret_cast<float*>(wsb + w.ldc);
    float* images = reinterpret_cast<float*>(wsb + w.images);
    const bool narrow = mfma_supported(D, L, U);
    const int64_t img_floats = narrow ? mfma_image_floats(D, L) : wide_image_floats(D, L, U);
    const bool f16 = use_fused && g_flow_variant >= 10;  // builds its own (split-f16) images
    if (g_operand_prec == 1) {
        // the bf16 experiment runs on the layer-range kernel: all 2S layers in one launch, or one layer per launch
        if (!narrow || !flow_range2_supported(D, L, U, use_fused ? 2 * S : 1))
            return fail(TNF_EUNSUPPORTED, "tnf_flow_log_prob_f32: no bf16-operand kernel for D=%d S=%d L=%d U=%d", D, S, L, U);
        float* zb = z0 ? z0 : reinterpret_cast<float*>(wsb + w.zbuf);
        float* lb = sum_log_det ? sum_log_det : reinterpret_cast<float*>(wsb + w.ldbuf);
        return launch_flow_chain2(z, use_fused ? nullptr : zb, use_fused ? nullptr : lb, z0, sum_log_det, log_prob, M_z, M_p, N, D,
                                  S, L, U, params, pstride, bn_mean, bn_alpha, interval_consts, nullptr,
                                  use_fused ? 2 * S : 1, st, 1);
    }
    const bool chain2 = !use_fused && narrow && g_layer_variant >= 10 && flow_range2_supported(D, L, U, 1);
    if (interval_consts && !f16 && !chain2)
        return fail(TNF_EUNSUPPORTED, "tnf_flow_log_prob_f32: a fused support layer needs the whole-flow kernel");
    if (f16 && g_flow_variant == 20 && flow_fused3_supported(D, S, L, U))  // 32-sample groups, 32x32x16 MFMAs (f16_tile3.h)
        return launch_flow_fused3(z, z0, sum_log_det, log_prob, M_z, M_p, N, D, S, L, U, params, pstride, bn_mean, bn_alpha,
                                  interval_consts, exact_reruns, st);
    if (f16 && (g_flow_variant == 10 || g_flow_variant == 20) && flow_fused2_supported(D, S, L, U))  // f16_tile2.h
        return launch_flow_fused2(z, z0, sum_log_det, log_prob, M_z, M_p, N, D, S, L, U, params, pstride, bn_mean, bn_alpha,
                                  interval_consts, exact_reruns, st);
    if (f16)  // ONE launch: the flow kernel builds its split-f16 operands and folds BN / Affine in its prologue
        return launch_flow_fused_f16(z, nullptr, nullptr, nullptr, z0, sum_log_det, log_prob, M_z, M_p, N, D, S, L, U,
                                     1, g_flow_variant, st, params, pstride, bn_mean, bn_alpha, interval_consts);
    if (!use_fused && narrow && g_layer_variant >= 10 && flow_range2_supported(D, L, U, 1)) {
        // default per-layer chain: the whole-flow kernel's tile code, ONE coupling layer per launch (flow_fused2.hip)
        float* zb = z0 ? z0 : reinterpret_cast<float*>(wsb + w.zbuf);
        float* lb = sum_log_det ? sum_log_det : reinterpret_cast<float*>(wsb + w.ldbuf);
        return launch_flow_chain2(z, zb, lb, z0, sum_log_det, log_prob, M_z, M_p, N, D, S, L, U, params, pstride, bn_mean,
                                  bn_alpha, interval_consts, exact_reruns, g_layer_variant >= 11 ? g_layer_variant - 10 : 1, st,
                                  0, images);  // `images`: the workspace region that holds the prepared prologues
    }
    // fp32-MFMA per-layer chain on the narrow shapes: in place from the second kernel on, each kernel storing only the
    // half it transforms (the folds are composed accordingly, flow_fold_kernel chain = 1)
    const int chain = (!use_fused && narrow) ? 1 : 0;
    rc = launch_flow_prep(params, bn_mean, bn_alpha, fold, ldc, narrow ? images : nullptr, M_p, D, S, L, U, pstride, 1, st,
                          chain);
    if (rc) return rc;
    if (!narrow) {
        rc = launch_wide_images(params, images, M_p, D, S, L, U, pstride, st);
        if (rc) return rc;
    }
    if (use_fused)
        return launch_flow_fused(z, images, fold, ldc, z0, sum_log_det, log_prob, M_z, M_p, N, D, S,
                                 L, U, 1, st);
    // one launch per coupling layer, last forward layer first
    float* zbuf = z0 ? z0 : reinterpret_cast<float*>(wsb + w.zbuf);
    float* ldbuf = sum_log_det ? sum_log_det : reinterpret_cast<float*>(wsb + w.ldbuf);
    const FlowLayout fl = flow_layout(D, S, L, U);
    const int nl = 2 * S;
    for (int c = nl - 1; c >= 0; --c) {
        MfmaLayerArgs a;
        memset(&a, 0, sizeof(a));
        const bool first = (c == nl - 1), last = (c == 0);
        a.z = first ? z : zbuf;
        a.z_out = (last && !z0) ? nullptr : zbuf;
        a.params = params + (c >> 1) * fl.stage + ((c & 1) ? fl.p_up : 0);
        a.pstride = pstride;
        a.image = images + (int64_t)c * img_floats;
        a.image_stride = (int64_t)nl * img_floats;
        a.pre = fold + (int64_t)c * 2 * D;
        a.fold_stride = (int64_t)nl * 2 * D;
        a.ld_in = first ? nullptr : ldbuf;
        a.ld_out = (last && !sum_log_det) ? nullptr : ldbuf;
        a.ld_sign = 1.f;
        a.ldc = ldc;
        a.add_ldc = last ? 1 : 0;
        a.log_prob = last ? log_prob : nullptr;
        a.Mz = first ? M_z : M; a.Mp = M_p; a.N = N;
        a.D = D; a.L = L; a.U = U; a.upper = (c & 1) ? 0 : 1; a.inverse = 1;
        a.skip_cond_store = (chain && !first && !(last && z0)) ? 1 : 0;
        rc = narrow ? launch_coupling_mfma(a, st) : launch_coupling_wide(a, st);
        if (rc) return rc;
    }
    return TNF_OK;
}

int tnf_flow_log_prob_f32(const float* z, const float* params, const float* bn_mean,
                          const float* bn_alpha, const float* interval_consts, float* log_prob, float* z0,
                          float* sum_log_det,
                          int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t S, int32_t L,
                          int32_t U, int64_t pstride, int32_t fusion, void* workspace,
                          int64_t workspace_bytes, void* stream) {
    return flow_log_prob_impl(z, params, bn_mean, bn_alpha, interval_consts, log_prob, z0, sum_log_det, M_z, M_p, N, D, S, L,
                              U, pstride, fusion, workspace, workspace_bytes, stream, nullptr);
}

/* Same call with a diagnostic: *exact_reruns (device word, caller zeroes it) += the number of 32-sample groups the
 * whole-flow kernel re-ran with exact fp32 first-layer contractions because a conditioner input left the f16
 * operand range (f16_tile2.h, point 3).  Results do not depend on it. */
int tnf_flow_log_prob_diag_f32(const float* z, const float* params, const float* bn_mean,
                               const float* bn_alpha, const float* interval_consts, float* log_prob, float* z0,
                               float* sum_log_det, int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t S, int32_t L,
                               int32_t U, int64_t pstride, int32_t fusion, void* workspace, int64_t workspace_bytes,
                               void* stream, uint32_t* exact_reruns) {
    return flow_log_prob_impl(z, params, bn_mean, bn_alpha, interval_consts, log_prob, z0, sum_log_det, M_z, M_p, N, D, S, L,
                              U, pstride, fusion, workspace, workspace_bytes, stream, exact_reruns);
}

// ---- training pair: log_prob with saved per-layer inputs, and its backward ----
struct TrainWs {
    int64_t fold, ldc, images, gfold, ldbuf, gbuf, total;
};
static TrainWs train_ws(int64_t M, int64_t Mp, int64_t N, int D, int S, int L) {
    TrainWs w;
    w.fold = 0;
    w.ldc = round16(Mp * 2 * S * 2 * D * (int64_t)sizeof(float));
    w.images = w.ldc + round16(Mp * (int64_t)sizeof(float));
    w.gfold = w.images + round16(Mp * 2 * S * mfma_image_floats(D, L) * (int64_t)sizeof(float));
    // g_fold (Mp, 2S, 2, D) followed by the per-row sums of g_log_prob (Mp): zeroed together
    w.ldbuf = w.gfold + round16((Mp * 2 * S * 2 * D + Mp + 1) * (int64_t)sizeof(float));  // + max |g_log_prob|
    w.gbuf = w.ldbuf + round16(M * N * (int64_t)sizeof(float));
    w.total = w.gbuf + 2 * round16(M * N * D * (int64_t)sizeof(float));
    return w;
}

int64_t tnf_flow_train_workspace_bytes(int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t S, int32_t L, int32_t U) {
    if (M < 1 || (M_p != 1 && M_p != M) || N < 0 || S < 1)
        return fail(TNF_EINVAL, "tnf_flow_train_workspace_bytes: M=%lld M_p=%lld N=%lld S=%d", (long long)M,
                    (long long)M_p, (long long)N, S);
    if (!mfma_supported(D, L, U))
        return fail(TNF_EUNSUPPORTED, "tnf_flow_train_workspace_bytes: no training kernels for D=%d L=%d U=%d", D, L, U);
    return train_ws(M, M_p, N, D, S, L).total;
}

static int train_checks(const char* fn, int64_t M, int64_t M_p, int64_t N, int D, int S, int L, int U,
                        int64_t pstride, const void* ws, int64_t ws_bytes) {
    if (M < 1 || (M_p != 1 && M_p != M) || N < 0 || S < 1)
        return fail(TNF_EINVAL, "%s: M=%lld M_p=%lld N=%lld S=%d", fn, (long long)M, (long long)M_p, (long long)N, S);
    if (!mfma_supported(D, L, U))
        return fail(TNF_EUNSUPPORTED, "%s: no training kernels for D=%d L=%d U=%d", fn, D, L, U);
    if (pstride < flow_layout(D, S, L, U).total)
        return fail(TNF_EINVAL, "%s: params row has %lld elements, flow needs %lld", fn, (long long)pstride,
                    (long long)flow_layout(D, S, L, U).total);
    if (!ws || ws_bytes < train_ws(M, M_p, N, D, S, L).total)
        return fail(TNF_EWORKSPACE, "%s: workspace %lld < %lld", fn, (long long)ws_bytes,
                    (long long)train_ws(M, M_p, N, D, S, L).total);
    return TNF_OK;
}

int tnf_flow_log_prob_fwd_f32(const float* z, const float* params, const float* bn_mean, const float* bn_alpha,
                              float* log_prob, float* states, int64_t M, int64_t M_p, int64_t N, int32_t D,
                              int32_t S, int32_t L, int32_t U, int64_t pstride, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    int rc = train_checks("tnf_flow_log_prob_fwd_f32", M, M_p, N, D, S, L, U, pstride, workspace, workspace_bytes);
    if (rc) return rc;
    if (!z || !params || !bn_mean || !bn_alpha || !log_prob || !states)
        return fail(TNF_EINVAL, "tnf_flow_log_prob_fwd_f32: NULL pointer");
    if (!aligned16(z) || !aligned16(states)) return fail(TNF_EINVAL, "tnf_flow_log_prob_fwd_f32: z / states must be 16-byte aligned");
    if (N == 0) return TNF_OK;
    hipStream_t st = as_stream(stream);
    const TrainWs w = train_ws(M, M_p, N, D, S, L);
    char* wsb = reinterpret_cast<char*>(workspace);
    float* fold = reinterpret_cast<float*>(wsb + w.fold);
    float* ldc = reinterpret_cast<float*>(wsb + w.ldc);
    float* images = reinterpret_cast<float*>(wsb + w.images);
    float* ldbuf = reinterpret_cast<float*>(wsb + w.ldbuf);
    rc = launch_flow_prep(params, bn_mean, bn_alpha, fold, ldc, images, M_p, D, S, L, U, pstride, 1, st);
    if (rc) return rc;
    const FlowLayout fl = flow_layout(D, S, L, U);
    const int64_t img_floats = mfma_image_floats(D, L);
    const int nl = 2 * S;
    const int64_t plane = M * N * D;
    for (int c = nl - 1; c >= 0; --c) {
        MfmaLayerArgs a;
        memset(&a, 0, sizeof(a));
        const bool first = (c == nl - 1), last = (c == 0);
        a.z = first ? z : states + (int64_t)c * plane;       // states[c] = input of layer kernel c
        a.z_out = last ? nullptr : states + (int64_t)(c - 1) * plane;
        a.params = params + (c >> 1) * fl.stage + ((c & 1) ? fl.p_up : 0);
        a.pstride = pstride;
        a.image = images + (int64_t)c * img_floats;
        a.image_stride = (int64_t)nl * img_floats;
        a.pre = fold + (int64_t)c * 2 * D;
        a.fold_stride = (int64_t)nl * 2 * D;
        a.ld_in = first ? nullptr : ldbuf;
        a.ld_out = last ? nullptr : ldbuf;
        a.ld_sign = 1.f;
        a.ldc = ldc;
        a.add_ldc = last ? 1 : 0;
        a.log_prob = last ? log_prob : nullptr;
        a.Mz = M; a.Mp = M_p; a.N = N;
        a.D = D; a.L = L; a.U = U; a.upper = (c & 1) ? 0 : 1; a.inverse = 1;
        a.gate = g_launch_gate;
        rc = launch_coupling_mfma(a, st);
        if (rc) return rc;
    }
    return TNF_OK;
}

int tnf_flow_log_prob_bwd_f32(const float* z, const float* states, const float* params, const float* bn_mean,
                              const float* bn_alpha, const float* g_log_prob, float* g_z, float* g_params,
                              int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t S, int32_t L, int32_t U,
                              int64_t pstride, int64_t gpstride, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    int rc = train_checks("tnf_flow_log_prob_bwd_f32", M, M_p, N, D, S, L, U, pstride, workspace, workspace_bytes);
    if (rc) return rc;
    if (!z || !states || !params || !bn_mean || !bn_alpha || !g_log_prob || !g_z || !g_params)
        return fail(TNF_EINVAL, "tnf_flow_log_prob_bwd_f32: NULL pointer");
    if (gpstride < flow_layout(D, S, L, U).total) return fail(TNF_EINVAL, "tnf_flow_log_prob_bwd_f32: g_params row too short");
    if (N == 0) return TNF_OK;
    hipStream_t st = as_stream(stream);
    const TrainWs w = train_ws(M, M_p, N, D, S, L);
    char* wsb = reinterpret_cast<char*>(workspace);
    float* fold = reinterpret_cast<float*>(wsb + w.fold);
    float* ldc = reinterpret_cast<float*>(wsb + w.ldc);
    float* images = reinterpret_cast<float*>(wsb + w.images);
    float* gfold = reinterpret_cast<float*>(wsb + w.gfold);
    float* gbuf[2] = {reinterpret_cast<float*>(wsb + w.gbuf),
                      reinterpret_cast<float*>(wsb + w.gbuf + round16(M * N * D * (int64_t)sizeof(float)))};
    rc = launch_flow_prep(params, bn_mean, bn_alpha, fold, ldc, images, M_p, D, S, L, U, pstride, 1, st);
    if (rc) return rc;
    float* glp_sum = gfold + M_p * 2 * S * 2 * D;
    unsigned* gmaxw = reinterpret_cast<unsigned*>(glp_sum + M_p);
    if (hipMemsetAsync(gfold, 0, (size_t)(M_p * 2 * S * 2 * D + M_p + 1) * sizeof(float), st) != hipSuccess)
        return fail(TNF_ELAUNCH, "tnf_flow_log_prob_bwd_f32: memset failed");
    rc = launch_gmax(g_log_prob, M * N, gmaxw, st);
    if (rc) return rc;
    const FlowLayout fl = flow_layout(D, S, L, U);
    const int64_t img_floats = mfma_image_floats(D, L);
    const int nl = 2 * S;
    const int64_t plane = M * N * D;
    for (int c = 0; c < nl; ++c) {
        BwdArgs a;
        memset(&a, 0, sizeof(a));
        const int64_t poff = (c >> 1) * fl.stage + ((c & 1) ? fl.p_up : 0);
        a.z = (c == nl - 1) ? z : states + (int64_t)c * plane;
        a.params = params + poff;
        a.g_zout = (c == 0) ? nullptr : gbuf[(c - 1) & 1];
        a.g_ld = g_log_prob;
        a.ld_scale = -1.f;  // log_prob = base - sum of the layers' log-dets
        a.g_z = (c == nl - 1) ? g_z : gbuf[c & 1];
        a.g_params = g_params + poff;
        a.M = M; a.Mp = M_p; a.N = N;
        a.pstride = pstride; a.gpstride = gpstride;
        a.U = U; a.upper = (c & 1) ? 0 : 1;
        a.image = images + (int64_t)c * img_floats;
        a.image_stride = (int64_t)nl * img_floats;
        a.fold = fold + (int64_t)c * 2 * D;
        a.g_fold = gfold + (int64_t)c * 2 * D;
        a.fold_stride = (int64_t)nl * 2 * D;
        a.g_lp = (c == 0) ? g_log_prob : nullptr;
        a.glp_sum = glp_sum;
        a.gmax = gmaxw;
        a.gate = g_launch_gate;
        // split-f16 layer backward unless asked otherwise or it would spill (L = 3 without a spare unit)
        if (g_train_bwd_fp32 || (L == 3 && U > 15)) rc = launch_coupling_backward_mfma_args(a, D, L, 1, st);
        else rc = launch_coupling_backward_f16(a, D, L, 1, st);
        if (rc) return rc;
    }
    return launch_flow_fold_backward(params, bn_alpha, gfold, glp_sum, g_params, M_p, D, S, L, U, pstride, gpstride,
                                     st);
}

// ---- reversible training pair: whole-flow forward that keeps only z0, one-kernel backward ----
int tnf_flow_train_rev_supported(int32_t D, int32_t S, int32_t L, int32_t U) {
    return (flow_fused_supported(D, S, L, U) && flow_train_rev_supported(D, S, L, U)) ? 1 : 0;
}

int64_t tnf_flow_train_rev_workspace_bytes(int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t S, int32_t L, int32_t U) {
    if (M < 1 || (M_p != 1 && M_p != M) || N < 0)
        return fail(TNF_EINVAL, "tnf_flow_train_rev_workspace_bytes: M=%lld M_p=%lld N=%lld", (long long)M, (long long)M_p,
                    (long long)N);
    if (!tnf_flow_train_rev_supported(D, S, L, U))
        return fail(TNF_EUNSUPPORTED, "tnf_flow_train_rev_workspace_bytes: D=%d S=%d L=%d U=%d", D, S, L, U);
    return flow_train_rev_workspace(M, M_p, N, D, S, L, U);
}

static int rev_checks(const char* fn, int64_t M, int64_t M_p, int64_t N, int D, int S, int L, int U, int64_t pstride) {
    if (M < 1 || (M_p != 1 && M_p != M) || N < 0)
        return fail(TNF_EINVAL, "%s: M=%lld M_p=%lld N=%lld", fn, (long long)M, (long long)M_p, (long long)N);
    if (!tnf_flow_train_rev_supported(D, S, L, U))
        return fail(TNF_EUNSUPPORTED, "%s: no reversible training kernels for D=%d S=%d L=%d U=%d", fn, D, S, L, U);
    if (pstride < flow_layout(D, S, L, U).total)
        return fail(TNF_EINVAL, "%s: params row has %lld elements, flow needs %lld", fn, (long long)pstride,
                    (long long)flow_layout(D, S, L, U).total);
    return TNF_OK;
}

int tnf_flow_log_prob_fwd_rev_f32(const float* z, const float* params, const float* bn_mean, const float* bn_alpha,
                                  float* log_prob, float* z0, int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t S,
                                  int32_t L, int32_t U, int64_t pstride, void* stream) {
    int rc = rev_checks("tnf_flow_log_prob_fwd_rev_f32", M, M_p, N, D, S, L, U, pstride);
    if (rc) return rc;
    if (N == 0) return TNF_OK;
    if (!z || !params || !bn_mean || !bn_alpha || !log_prob || !z0)
        return fail(TNF_EINVAL, "tnf_flow_log_prob_fwd_rev_f32: NULL pointer");
    if (!aligned16(z) || !aligned16(z0)) return fail(TNF_EINVAL, "tnf_flow_log_prob_fwd_rev_f32: z / z0 must be 16-byte aligned");
    if (g_operand_prec == 1) {  // the bf16 experiment: forward activations from bf16 operands (the backward stays split-f16)
        if (!flow_range2_supported(D, L, U, 2 * S))
            return fail(TNF_EUNSUPPORTED, "tnf_flow_log_prob_fwd_rev_f32: no bf16-operand kernel for D=%d S=%d L=%d U=%d", D, S, L, U);
        return launch_flow_chain2(z, nullptr, nullptr, z0, nullptr, log_prob, M, M_p, N, D, S, L, U, params, pstride, bn_mean,
                                  bn_alpha, nullptr, nullptr, 2 * S, as_stream(stream), 1);
    }
    if (g_flow_variant == 20 && flow_fused3_supported(D, S, L, U))
        return launch_flow_fused3(z, z0, nullptr, log_prob, M, M_p, N, D, S, L, U, params, pstride, bn_mean, bn_alpha, nullptr,
                                  nullptr, as_stream(stream));
    if ((g_flow_variant == 10 || g_flow_variant == 20) && flow_fused2_supported(D, S, L, U))
        return launch_flow_fused2(z, z0, nullptr, log_prob, M, M_p, N, D, S, L, U, params, pstride, bn_mean, bn_alpha, nullptr,
                                  nullptr, as_stream(stream));
    return launch_flow_fused_f16(z, nullptr, nullptr, nullptr, z0, nullptr, log_prob, M, M_p, N, D, S, L, U, 1,
                                 g_flow_variant >= 10 ? g_flow_variant : 10, as_stream(stream), params, pstride, bn_mean,
                                 bn_alpha, nullptr);
}

int tnf_flow_log_prob_bwd_rev_f32(const float* z0, const float* params, const float* bn_mean, const float* bn_alpha,
                                  const float* g_log_prob, float* g_z, float* g_params, int64_t M, int64_t M_p,
                                  int64_t N, int32_t D, int32_t S, int32_t L, int32_t U, int64_t pstride,
                                  int64_t gpstride, void* workspace, int64_t workspace_bytes, int32_t* overflow,
                                  void* stream) {
    int rc = rev_checks("tnf_flow_log_prob_bwd_rev_f32", M, M_p, N, D, S, L, U, pstride);
    if (rc) return rc;
    if (N == 0) return TNF_OK;
    if (!z0 || !params || !bn_mean || !bn_alpha || !g_log_prob || !g_params)
        return fail(TNF_EINVAL, "tnf_flow_log_prob_bwd_rev_f32: NULL pointer");
    if (!aligned16(z0) || (g_z && !aligned16(g_z)))
        return fail(TNF_EINVAL, "tnf_flow_log_prob_bwd_rev_f32: z0 / g_z must be 16-byte aligned");
    if (gpstride < flow_layout(D, S, L, U).total)
        return fail(TNF_EINVAL, "tnf_flow_log_prob_bwd_rev_f32: g_params row too short");
    if (!workspace || workspace_bytes < flow_train_rev_workspace(M, M_p, N, D, S, L, U))
        return fail(TNF_EWORKSPACE, "tnf_flow_log_prob_bwd_rev_f32: workspace %lld < %lld", (long long)workspace_bytes,
                    (long long)flow_train_rev_workspace(M, M_p, N, D, S, L, U));
    return launch_flow_bwd_rev(z0, params, bn_mean, bn_alpha, g_log_prob, g_z, g_params, M, M_p, N, D, S, L, U, pstride,
                               gpstride, workspace, overflow, as_stream(stream));
}

// NormFlow.forward with batch-statistics BatchNorm (freeze_bn=False), no autograd: one C call for the whole stack
int64_t tnf_flow_forward_batch_workspace_bytes(int64_t M_p, int32_t D, int32_t S, int32_t L) {
    if (M_p < 1 || D < 2 || S < 1 || L < 1)
        return fail(TNF_EINVAL, "tnf_flow_forward_batch_workspace_bytes: M_p=%lld D=%d S=%d L=%d", (long long)M_p, D, S, L);
    return flow_forward_batch_workspace(M_p, D, S, L);
}

int tnf_flow_forward_batch_f32(const float* omega, const float* params, float* z_out, float* sum_log_det,
                               float* bn_mean_out, float* bn_alpha_out, int64_t M, int64_t M_p, int64_t N, int32_t D,
                               int32_t S, int32_t L, int32_t U, int64_t pstride, float eps, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    const char* fn = "tnf_flow_forward_batch_f32";
    if (M < 1 || (M_p != 1 && M_p != M) || N < 0 || S < 1)
        return fail(TNF_EINVAL, "%s: M=%lld M_p=%lld N=%lld S=%d", fn, (long long)M, (long long)M_p, (long long)N, S);
    if (!mfma_supported(D, L, U)) return fail(TNF_EUNSUPPORTED, "%s: no kernel for D=%d L=%d U=%d", fn, D, L, U);
    if (M * N < 2) return fail(TNF_EINVAL, "%s: batch statistics need more than one row", fn);
    if (pstride < flow_layout(D, S, L, U).total)
        return fail(TNF_EINVAL, "%s: params row has %lld elements, flow needs %lld", fn, (long long)pstride,
                    (long long)flow_layout(D, S, L, U).total);
    if (!omega || !params || !z_out || !sum_log_det || !bn_mean_out || !bn_alpha_out || !workspace)
        return fail(TNF_EINVAL, "%s: NULL pointer", fn);
    if (!aligned16(omega) || !aligned16(z_out)) return fail(TNF_EINVAL, "%s: omega / z_out must be 16-byte aligned", fn);
    if (z_out == omega) return fail(TNF_EINVAL, "%s: z_out must not alias omega", fn);
    if (workspace_bytes < flow_forward_batch_workspace(M_p, D, S, L))
        return fail(TNF_EWORKSPACE, "%s: workspace %lld < %lld", fn, (long long)workspace_bytes,
                    (long long)flow_forward_batch_workspace(M_p, D, S, L));
    return launch_flow_forward_batch(omega, params, z_out, sum_log_det, bn_mean_out, bn_alpha_out, M, M_p, N, D, S, L, U,
                                     pstride, eps, workspace, as_stream(stream));
}

// The same chain in steps (see coupling_mfma.hip): begin, then per coupling layer c = 0 .. 2S-1 `layer` (kernel + LOCAL
// moments of its output) and `fold` (statistics from the moments the caller may have summed over ranks), then end.
static int fb_step_checks(const char* fn, int64_t M_p, int D, int S, int L, int U, int64_t pstride, const void* ws,
                          int64_t ws_bytes) {
    if (M_p < 1 || S < 1) return fail(TNF_EINVAL, "%s: M_p=%lld S=%d", fn, (long long)M_p, S);
    if (!mfma_supported(D, L, U)) return fail(TNF_EUNSUPPORTED, "%s: no kernel for D=%d L=%d U=%d", fn, D, L, U);
    if (pstride < flow_layout(D, S, L, U).total)
        return fail(TNF_EINVAL, "%s: params row has %lld elements, flow needs %lld", fn, (long long)pstride,
                    (long long)flow_layout(D, S, L, U).total);
    if (!ws || ws_bytes < flow_forward_batch_workspace(M_p, D, S, L))
        return fail(TNF_EWORKSPACE, "%s: workspace %lld < %lld", fn, (long long)ws_bytes,
                    (long long)flow_forward_batch_workspace(M_p, D, S, L));
    return TNF_OK;
}

int tnf_flow_forward_batch_begin_f32(const float* params, int64_t M_p, int32_t D, int32_t S, int32_t L, int32_t U,
                                     int64_t pstride, void* workspace, int64_t workspace_bytes, void* stream) {
    const char* fn = "tnf_flow_forward_batch_begin_f32";
    int rc = fb_step_checks(fn, M_p, D, S, L, U, pstride, workspace, workspace_bytes);
    if (rc) return rc;
    if (!params) return fail(TNF_EINVAL, "%s: NULL pointer", fn);
    return flow_forward_batch_begin(params, M_p, D, S, L, U, pstride, workspace, as_stream(stream));
}

int tnf_flow_forward_batch_layer_f32(int32_t layer, const float* z_in, const float* params, float* z_out,
                                     float* sum_log_det, double* moments, int64_t M, int64_t M_p, int64_t N, int32_t D,
                                     int32_t S, int32_t L, int32_t U, int64_t pstride, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    const char* fn = "tnf_flow_forward_batch_layer_f32";
    int rc = fb_step_checks(fn, M_p, D, S, L, U, pstride, workspace, workspace_bytes);
    if (rc) return rc;
    if (M < 1 || (M_p != 1 && M_p != M) || N < 0) return fail(TNF_EINVAL, "%s: M=%lld M_p=%lld N=%lld", fn, (long long)M, (long long)M_p, (long long)N);
    if (layer < 0 || layer >= 2 * S) return fail(TNF_EINVAL, "%s: layer %d of %d", fn, layer, 2 * S);
    if (!params || !moments || (N > 0 && (!z_in || !z_out || !sum_log_det))) return fail(TNF_EINVAL, "%s: NULL pointer", fn);
    if (!aligned16(z_in) || !aligned16(z_out) || (reinterpret_cast<uintptr_t>(moments) & 7))
        return fail(TNF_EINVAL, "%s: z_in / z_out must be 16-byte and moments 8-byte aligned", fn);
    if (layer == 0 && z_in == z_out) return fail(TNF_EINVAL, "%s: z_out must not alias the base draw", fn);
    return flow_forward_batch_layer(layer, z_in, params, z_out, sum_log_det, moments, M, M_p, N, D, S, L, U, pstride,
                                    workspace, as_stream(stream));
}

int tnf_flow_forward_batch_fold_f32(int32_t layer, const float* params, const double* moments, float* bn_mean_out,
                                    float* bn_alpha_out, int64_t M_p, int32_t D, int32_t S, int32_t L, int32_t U,
                                    int64_t pstride, float eps, void* workspace, int64_t workspace_bytes, void* stream) {
    const char* fn = "tnf_flow_forward_batch_fold_f32";
    int rc = fb_step_checks(fn, M_p, D, S, L, U, pstride, workspace, workspace_bytes);
    if (rc) return rc;
    if (layer < 0 || layer >= 2 * S) return fail(TNF_EINVAL, "%s: layer %d of %d", fn, layer, 2 * S);
    if (!params || !moments || !bn_mean_out || !bn_alpha_out) return fail(TNF_EINVAL, "%s: NULL pointer", fn);
    return flow_forward_batch_fold(layer, params, moments, bn_mean_out, bn_alpha_out, M_p, D, S, L, U, pstride, eps, workspace,
                                   as_stream(stream));
}

int tnf_flow_forward_batch_end_f32(float* z_out, float* sum_log_det, int64_t M, int64_t M_p, int64_t N, int32_t D,
                                   int32_t S, int32_t L, void* workspace, int64_t workspace_bytes, void* stream) {
    const char* fn = "tnf_flow_forward_batch_end_f32";
    if (M < 1 || (M_p != 1 && M_p != M) || N < 0 || D < 2 || S < 1 || L < 1)
        return fail(TNF_EINVAL, "%s: M=%lld M_p=%lld N=%lld D=%d", fn, (long long)M, (long long)M_p, (long long)N, D);
    if (!workspace || workspace_bytes < flow_forward_batch_workspace(M_p, D, S, L))
        return fail(TNF_EWORKSPACE, "%s: workspace too small", fn);
    if (N > 0 && (!z_out || !sum_log_det)) return fail(TNF_EINVAL, "%s: NULL pointer", fn);
    return flow_forward_batch_end(z_out, sum_log_det, M, M_p, N, D, workspace, as_stream(stream));
}

// ... and the same stack under autograd (sampling-based objectives): forward keeps every coupling layer's output
int64_t tnf_flow_forward_train_workspace_bytes(int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t S, int32_t L) {
    if (M < 1 || (M_p != 1 && M_p != M) || N < 0 || D < 2 || S < 1 || L < 1)
        return fail(TNF_EINVAL, "tnf_flow_forward_train_workspace_bytes: M=%lld M_p=%lld N=%lld D=%d S=%d L=%d",
                    (long long)M, (long long)M_p, (long long)N, D, S, L);
    return flow_forward_train_workspace(M, M_p, N, D, S, L);
}

static int fwd_train_checks(const char* fn, int64_t M, int64_t M_p, int64_t N, int D, int S, int L, int U, int64_t pstride,
                            const void* ws, int64_t ws_bytes) {
    if (M < 1 || (M_p != 1 && M_p != M) || N < 0 || S < 1)
        return fail(TNF_EINVAL, "%s: M=%lld M_p=%lld N=%lld S=%d", fn, (long long)M, (long long)M_p, (long long)N, S);
    if (!mfma_supported(D, L, U)) return fail(TNF_EUNSUPPORTED, "%s: no kernel for D=%d L=%d U=%d", fn, D, L, U);
    if (M * N < 2) return fail(TNF_EINVAL, "%s: batch statistics need more than one row", fn);
    if (pstride < flow_layout(D, S, L, U).total)
        return fail(TNF_EINVAL, "%s: params row has %lld elements, flow needs %lld", fn, (long long)pstride,
                    (long long)flow_layout(D, S, L, U).total);
    if (!ws || ws_bytes < flow_forward_train_workspace(M, M_p, N, D, S, L))
        return fail(TNF_EWORKSPACE, "%s: workspace %lld < %lld", fn, (long long)ws_bytes,
                    (long long)flow_forward_train_workspace(M, M_p, N, D, S, L));
    return TNF_OK;
}

int tnf_flow_forward_train_fwd_f32(const float* omega, const float* params, float* z_out, float* sum_log_det,
                                   float* states, float* folds, float* bn_mean_out, float* bn_alpha_out, int64_t M,
                                   int64_t M_p, int64_t N, int32_t D, int32_t S, int32_t L, int32_t U, int64_t pstride,
                                   float eps, void* workspace, int64_t workspace_bytes, void* stream) {
    const char* fn = "tnf_flow_forward_train_fwd_f32";
    int rc = fwd_train_checks(fn, M, M_p, N, D, S, L, U, pstride, workspace, workspace_bytes);
    if (rc) return rc;
    if (!omega || !params || !z_out || !sum_log_det || !states || !folds || !bn_mean_out || !bn_alpha_out)
        return fail(TNF_EINVAL, "%s: NULL pointer", fn);
    if (!aligned16(omega) || !aligned16(z_out) || !aligned16(states))
        return fail(TNF_EINVAL, "%s: omega / z_out / states must be 16-byte aligned", fn);
    return launch_flow_forward_train_fwd(omega, params, z_out, sum_log_det, states, folds, bn_mean_out, bn_alpha_out, M, M_p,
                                         N, D, S, L, U, pstride, eps, workspace, as_stream(stream));
}

int tnf_flow_forward_train_bwd_f32(const float* omega, const float* params, const float* states, const float* folds,
                                   const float* bn_mean, const float* bn_alpha, const float* g_z, const float* g_sum_log_det,
                                   float* g_omega, float* g_params, int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t S,
                                   int32_t L, int32_t U, int64_t pstride, int64_t gpstride, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
    const char* fn = "tnf_flow_forward_train_bwd_f32";
    int rc = fwd_train_checks(fn, M, M_p, N, D, S, L, U, pstride, workspace, workspace_bytes);
    if (rc) return rc;
    if (!omega || !params || !states || !folds || !bn_mean || !bn_alpha || !g_z || !g_sum_log_det || !g_params)
        return fail(TNF_EINVAL, "%s: NULL pointer", fn);
    if (gpstride < flow_layout(D, S, L, U).total) return fail(TNF_EINVAL, "%s: g_params row too short", fn);
    if (!aligned16(omega) || !aligned16(states) || !aligned16(g_z) || (g_omega && !aligned16(g_omega)) ||
        (reinterpret_cast<uintptr_t>(workspace) & 255))
        return fail(TNF_EINVAL, "%s: omega / states / g_z / g_omega must be 16-byte and the workspace 256-byte aligned", fn);
    return launch_flow_forward_train_bwd(omega, params, states, folds, bn_mean, bn_alpha, g_z, g_sum_log_det, g_omega,
                                         g_params, M, M_p, N, D, S, L, U, pstride, gpstride, workspace, as_stream(stream));
}

static int flow_forward_impl(const float* omega, const float* params, const float* bn_mean,
                         const float* bn_alpha, const float* interval_consts, float* z_out, float* sum_log_det,
                         int64_t M_z,
                         int64_t M_p, int64_t N, int32_t D, int32_t S, int32_t L, int32_t U,
                         int64_t pstride, int32_t fusion, void* workspace, int64_t workspace_bytes,
                         void* stream, double* log_q) {
    int use_fused = 0;
    int rc = flow_common_checks("tnf_flow_forward_f32", M_z, M_p, N, D, S, L, U, pstride, fusion,
                                workspace, workspace_bytes, &use_fused);
    if (rc) return rc;
    if (!omega || !params || !bn_mean || !bn_alpha || !z_out || !sum_log_det)
        return fail(TNF_EINVAL, "tnf_flow_forward_f32: NULL pointer");
    if (!aligned16(omega) || !aligned16(z_out))
        return fail(TNF_EINVAL, "tnf_flow_forward_f32: omega / z_out must be 16-byte aligned");
    if (z_out == omega) return fail(TNF_EINVAL, "tnf_flow_forward_f32: z_out must not alias omega");
    if (N == 0) return TNF_OK;
    hipStream_t st = as_stream(stream);
    const int64_t M = M_z > M_p ? M_z : M_p;
    const FlowWs w = flow_ws(M, N, D, S, L, U);
    char* wsb = reinterpret_cast<char*>(workspace);
    float* fold = reinterpret_cast<float*>(wsb + w.fold);
    float* ldc = reinterpret_cast<float*>(wsb + w.ldc);
    float* images = reinterpret_cast<float*>(wsb + w.images);
    const bool narrow = mfma_supported(D, L, U);
    const int64_t img_floats = narrow ? mfma_image_floats(D, L) : wide_image_floats(D, L, U);
    const bool f16 = use_fused && g_flow_variant >= 10;
    if (interval_consts && !f16)
        return fail(TNF_EUNSUPPORTED, "tnf_flow_forward_f32: a fused support layer needs the whole-flow kernel");
    if (f16 && (g_flow_variant == 10 || g_flow_variant == 20) && flow_fused2_supported(D, S, L, U))  // f16_tile2.h, FWD
        return launch_flow_fused2(omega, z_out, sum_log_det, nullptr, M_z, M_p, N, D, S, L, U, params, pstride, bn_mean, bn_alpha,
                                  interval_consts, nullptr, st, 1, log_q);
    if (log_q)
        return fail(TNF_EUNSUPPORTED, "tnf_flow_forward_logq_f32: only the default whole-flow kernel writes log_q "
                                      "(D=%d S=%d L=%d U=%d, fusion %d)", D, S, L, U, fusion);
    if (f16)
        return launch_flow_fused_f16(omega, nullptr, nullptr, nullptr, z_out, sum_log_det, nullptr, M_z, M_p, N, D, S, L,
                                     U, 0, g_flow_variant, st, params, pstride, bn_mean, bn_alpha, interval_consts);
    if (!use_fused && narrow && g_layer_variant >= 10 && flow_range2_supported(D, L, U, 1))
        // default per-layer chain of the sampling direction: the whole-flow kernel's tile code, ONE coupling layer per
        // launch, staged 1-KB loads, half-row stores, prepared prologues (flow_range2_kernel<.., FWD>, flow_fused2.hip)
        return launch_flow_chain2_fwd(omega, z_out, sum_log_det, M_z, M_p, N, D, S, L, U, params, pstride, bn_mean, bn_alpha,
                                      nullptr, st, images);
    rc = launch_flow_prep(params, bn_mean, bn_alpha, fold, ldc, narrow ? images : nullptr, M_p, D, S, L, U, pstride, 0, st);
    if (rc) return rc;
    if (!narrow) {
        rc = launch_wide_images(params, images, M_p, D, S, L, U, pstride, st);
        if (rc) return rc;
    }
    if (use_fused)
        return launch_flow_fused(omega, images, fold, ldc, z_out, sum_log_det, nullptr, M_z, M_p, N,
                                 D, S, L, U, 0, st);
    const FlowLayout fl = flow_layout(D, S, L, U);
    const int nl = 2 * S;
    for (int c = 0; c < nl; ++c) {
        MfmaLayerArgs a;
        memset(&a, 0, sizeof(a));
        const bool first = (c == 0), last = (c == nl - 1);
        a.z = first ? omega : z_out;
        a.z_out = z_out;
        a.params = params + (c >> 1) * fl.stage + ((c & 1) ? fl.p_up : 0);
        a.pstride = pstride;
        a.image = images + (int64_t)c * img_floats;
        a.image_stride = (int64_t)nl * img_floats;
        a.post = fold + (int64_t)c * 2 * D;
        a.fold_stride = (int64_t)nl * 2 * D;
        a.ld_in = first ? nullptr : sum_log_det;
        a.ld_out = sum_log_det;
        a.ld_sign = 1.f;
        a.ldc = ldc;
        a.add_ldc = last ? 1 : 0;
        a.Mz = first ? M_z : M; a.Mp = M_p; a.N = N;
        a.D = D; a.L = L; a.U = U; a.upper = (c & 1) ? 0 : 1; a.inverse = 0;
        rc = narrow ? launch_coupling_mfma(a, st) : launch_coupling_wide(a, st);
        if (rc) return rc;
    }
    return TNF_OK;
}

int tnf_flow_forward_f32(const float* omega, const float* params, const float* bn_mean,
                         const float* bn_alpha, const float* interval_consts, float* z_out, float* sum_log_det,
                         int64_t M_z,
                         int64_t M_p, int64_t N, int32_t D, int32_t S, int32_t L, int32_t U,
                         int64_t pstride, int32_t fusion, void* workspace, int64_t workspace_bytes,
                         void* stream) {
    return flow_forward_impl(omega, params, bn_mean, bn_alpha, interval_consts, z_out, sum_log_det, M_z, M_p, N, D, S, L, U,
                             pstride, fusion, workspace, workspace_bytes, stream, nullptr);
}

int tnf_flow_forward_logq_f32(const float* omega, const float* params, const float* bn_mean,
                              const float* bn_alpha, const float* interval_consts, float* z_out, float* sum_log_det,
                              double* log_q, int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t S, int32_t L, int32_t U,
                              int64_t pstride, int32_t fusion, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!log_q) return fail(TNF_EINVAL, "tnf_flow_forward_logq_f32: NULL log_q");
    return flow_forward_impl(omega, params, bn_mean, bn_alpha, interval_consts, z_out, sum_log_det, M_z, M_p, N, D, S, L, U,
                             pstride, fusion, workspace, workspace_bytes, stream, log_q);
}

}  // extern "C"
