// Shared host/device helpers for libtnf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tnf.h"

namespace tnf {

// ---- error reporting (thread-local message behind tnf_last_error) ---------
char* err_buf();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

void diag_count(int family);  // api.hip: process-wide launch counters behind tnf_diag_launch_count

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// The batch index m (parameter row / context) rides on grid dimensions y and z so that any M
// works (gridDim.y alone stops at 65,535 -- SNPE-style calls have 10^5..10^6 contexts).
inline dim3 grid_xm(int64_t bx, int64_t M) {
    const int64_t my = M < 32768 ? M : 32768;
    const int64_t mz = (M + my - 1) / my;
    return dim3((unsigned)bx, (unsigned)my, (unsigned)mz);
}
#if defined(__HIPCC__)
__device__ __forceinline__ int64_t grid_m() { return (int64_t)blockIdx.y + (int64_t)gridDim.y * (int64_t)blockIdx.z; }
#endif

// ---- packed parameter layout of one RealNVP layer (bijectors.py:222-242) ---
// [W_t (d_in*d_out, row-major [in][out]) | W_s | b_t (d_out) | b_s (d_out)] per MLP layer,
// layers: d_in0 -> U, (U -> U) x (L-1), U -> d_out.
struct CouplingDims {
    int d_in, d_out;
};
__host__ __device__ inline CouplingDims coupling_dims(int D, int upper) {
    CouplingDims c;
    c.d_in = D / 2;
    c.d_out = D / 2;
    if (D & 1) {
        c.d_in += upper ? 0 : 1;
        c.d_out += upper ? 1 : 0;
    }
    return c;
}
__host__ __device__ inline int64_t coupling_num_params(int D, int L, int U, int upper) {
    CouplingDims c = coupling_dims(D, upper);
    return 2 * ((int64_t)c.d_in * U + (int64_t)c.d_out * U + c.d_out + U +
                (int64_t)(L - 1) * (U + 1) * U);
}
// Offsets inside a parameter row of NormFlow(arch_type="coupling")
// (density_estimator.py:260-270, 379-384): per stage [RealNVP(up) | RealNVP(low) | Affine(2D)].
struct FlowLayout {
    int64_t p_up, p_low, stage, total;
};
__host__ __device__ inline FlowLayout flow_layout(int D, int S, int L, int U) {
    FlowLayout f;
    f.p_up = coupling_num_params(D, L, U, 1);
    f.p_low = coupling_num_params(D, L, U, 0);
    f.stage = f.p_up + f.p_low + 2 * (int64_t)D;
    f.total = f.stage * S;
    return f;
}

// ---- runtime options (testing hooks; per calling thread, like the error string: tnf_set_option changes the
// kernel selection of the thread that called it and of no other) ---------------------------------------------
extern thread_local int g_force_generic;
extern thread_local int g_flow_variant;   // flow_fused.hip
extern thread_local int g_layer_variant;  // coupling_mfma.hip
extern thread_local int g_train_bwd_fp32; // coupling_mfma.hip
extern thread_local int g_cond_variant;   // cond_flow.hip
extern thread_local const int* g_launch_gate;  // api.hip: tnf_set_launch_gate
extern thread_local int g_rev_variant;    // flow_bwd_f16.hip
extern thread_local int g_operand_prec;   // api.hip: 0 = fp32-accurate split-f16 operands, 1 = bf16 operands (experiment)

// ---- kernels implemented in the .hip files ----------------------------------
int launch_coupling_generic(int dtype, const void* z, const void* params, void* z_out,
                            void* log_det, int64_t Mz, int64_t Mp, int64_t N, int D, int L, int U,
                            int upper, int inverse, int64_t pstride, int ld_mode, hipStream_t st);

bool mfma_supported(int D, int L, int U);
// Fast per-layer kernel.  pre/post: folded per-feature constants [A(D)|B(D)] per m (stride
// fold_stride floats) or NULL.  ld_in may be NULL (=0).  ld_out = ld_in + ld_sign*sum(s).
// If log_prob != NULL the kernel finalises: log_prob = -0.5*|z'|^2 - D*log(sqrt(2pi)) - (ld + ldc[m]).
struct MfmaLayerArgs {
    const float* z;
    float* z_out;  // may be NULL when finalising
    const float* params;
    int64_t pstride;
    const float* image;  // optional prepared operand image (per m: image_stride floats); else gather from params
    int64_t image_stride;
    const float* pre;
    const float* post;
    int64_t fold_stride;
    const float* ld_in;
    float* ld_out;  // may be NULL
    float ld_sign;
    const float* ldc;  // per-m log-det constant, added when writing ld_out_total / log_prob
    float* log_prob;
    int add_ldc;  // add ldc[m] into ld_out as well
    int64_t Mz, Mp, N;
    int D, L, U, upper, inverse;
    int wave_m;  // set by the launcher: one context (m) per wave instead of per workgroup (many contexts, few samples)
    int skip_cond_store;  // in-place chains (z_out == z): leave the conditioner half in memory as it is -- the fold this
                          // kernel applied to it is composed into the next kernel's constants (flow_fold_kernel, chain = 1)
    const int* gate;      // optional device flag: the kernel returns at once while *gate == 0 (tnf_set_launch_gate)
};
int launch_coupling_mfma(const MfmaLayerArgs& a, hipStream_t st);

// wide per-layer kernel (coupling_wide.hip): D % 8 == 0, D <= 128, U <= 64, L <= 5
bool wide_supported(int D, int L, int U);
int64_t wide_image_floats(int D, int L, int U);
int launch_coupling_wide(const MfmaLayerArgs& a, hipStream_t st);
// MFMA backward of the wide shapes (coupling_wide_bwd.hip): one shared parameter row, N = all samples of the call
bool wide_bwd_supported(int D, int L, int U);
int64_t wide_bwd_workspace(int64_t N, int D, int L, int U);
bool maf_wide_bwd_supported(int D, int L, int U);
int64_t maf_wide_bwd_workspace(int64_t N, int D, int L, int U);
int launch_maf_backward_wide(const float* z, const float* params, const float* masks, const float* g_zout, const float* g_ld,
                             float* g_z, float* g_params, int64_t N, int D, int L, int U, int64_t gpstride, void* ws,
                             hipStream_t st);
int launch_coupling_backward_wide(const float* z, const float* params, const float* g_zout, const float* g_ld, float* g_z,
                                  float* g_params, int64_t N, int D, int L, int U, int upper, int inverse,
                                  int64_t gpstride, void* ws, hipStream_t st);
int launch_wide_images(const float* params, float* images, int64_t Mp, int D, int S, int L, int U,
                       int64_t pstride, hipStream_t st);

// Per-call preparation for the flow-level chains: fold BN/Affine constants (fold: (Mp, 2S, 2, D),
// ldc: (Mp)) and build the lane-ordered MFMA operand images (Mp, 2S, mfma_image_floats(D, L)).
int64_t mfma_image_floats(int D, int L);
int launch_flow_fold_backward(const float* params, const float* bn_alpha, const float* g_fold, const float* glp_sum,
                              float* g_params, int64_t Mp, int D, int S, int L, int U, int64_t pstride,
                              int64_t gpstride, hipStream_t st);
int launch_flow_prep(const float* params, const float* bn_mean, const float* bn_alpha, float* fold,
                     float* ldc, float* images, int64_t Mp, int D, int S, int L, int U,
                     int64_t pstride, int inverse, hipStream_t st, int chain = 0);

int launch_flow_fused(const float* z, const float* images, const float* fold, const float* ldc,
                      float* z_out, float* sum_log_det, float* log_prob, int64_t Mz, int64_t Mp,
                      int64_t N, int D, int S, int L, int U, int inverse, hipStream_t st);
bool flow_fused_supported(int D, int S, int L, int U);
// whole-flow inverse kernel, second formulation (flow_fused2.hip / f16_tile2.h)
// arguments of the whole-flow inverse kernels (flow_fused2.hip, flow_fused3.hip)
struct Flow2Args {
    const float* z;
    float* z_out;        // z0 (optional)
    float* sum_log_det;  // optional
    float* log_prob;     // optional
    int64_t Mz, Mp, N;
    int S, U;
    const float* params;
    const float* bn_mean;
    const float* bn_alpha;
    int64_t pstride, stage_stride, affine_off, low_off;
    const float* iv;     // (7, D) constants of a fused ToInterval support layer, or NULL
    unsigned* slow_count;  // optional: += number of groups re-run through the exact path (testing / diagnostics)
    int stage_out;         // set by the launcher: row outputs leave through LDS staging tiles (flow_fused2.hip)
    double* log_q;         // sampling pass, optional: log N(omega; 0, I) - (sum of log-dets), (M, N) float64 -- the
                           // log-density NormFlow.forward returns beside the samples (density_estimator.py:369-388)
};

bool flow_fused2_supported(int D, int S, int L, int U);
// forward = 0: inverse pass (z -> z0, sum of log-dets, log_prob); 1: sampling pass (omega -> z in z0, sum of log-dets)
int launch_flow_fused2(const float* z, float* z0, float* sum_log_det, float* log_prob, int64_t Mz, int64_t Mp, int64_t N,
                       int D, int S, int L, int U, const float* params, int64_t pstride, const float* bn_mean,
                       const float* bn_alpha, const float* interval_consts, unsigned* slow_count, hipStream_t st,
                       int forward = 0, double* log_q = nullptr);
// the same tile code as a chain of launches with `per_launch` coupling layers each (1 = one kernel per coupling layer)
bool flow_fused3_supported(int D, int S, int L, int U);  // flow_fused3.hip: the same on 32-sample groups (32x32x16 MFMAs)
int launch_flow_fused3(const float* z, float* z0, float* sum_log_det, float* log_prob, int64_t Mz, int64_t Mp, int64_t N,
                       int D, int S, int L, int U, const float* params, int64_t pstride, const float* bn_mean,
                       const float* bn_alpha, const float* interval_consts, unsigned* slow_count, hipStream_t st);
bool flow_range2_supported(int D, int L, int U, int nlayers);
int launch_flow_chain2(const float* z, float* zbuf, float* ldbuf, float* z0, float* sum_log_det, float* log_prob, int64_t Mz,
                       int64_t Mp, int64_t N, int D, int S, int L, int U, const float* params, int64_t pstride,
                       const float* bn_mean, const float* bn_alpha, const float* interval_consts, unsigned* slow_count,
                       int per_launch, hipStream_t st, int prec = 0,  // prec 1: the bf16 experiment (f16_tile2.h)
                       float* prep_ws = nullptr);  // flow_chain2_prep_floats(...) * Mp floats: prologues prepared by one launch
int launch_flow_chain2_fwd(const float* omega, float* z, float* sum_log_det, int64_t Mz, int64_t Mp, int64_t N, int D, int S,
                           int L, int U, const float* params, int64_t pstride, const float* bn_mean, const float* bn_alpha,
                           unsigned* slow_count, hipStream_t st, float* prep_ws);
int64_t flow_chain2_prep_floats(int D, int S, int L, int per_launch);
// split-f16 variant of the whole-flow kernel (flow_fused_f16.hip); images in slots of mfma_image_floats(D, 3)
int launch_flow_images_f16(const float* params, float* images, int64_t Mp, int D, int S, int L, int U,
                           int64_t pstride, hipStream_t st);
int launch_flow_fused_f16(const float* z, const float* images, const float* fold, const float* ldc,
                          float* z_out, float* sum_log_det, float* log_prob, int64_t Mz, int64_t Mp,
                          int64_t N, int D, int S, int L, int U, int inverse, int variant, hipStream_t st,
                          const float* params = nullptr, int64_t pstride = 0, const float* bn_mean = nullptr,
                          const float* bn_alpha = nullptr,  // fold == NULL: folded inside the kernel from these
                          const float* interval_consts = nullptr);  // fused ToInterval support layer (7, D)

int launch_bn_moments(const float* z, double* moments, int64_t rows, int D, hipStream_t st);
int launch_bn_finalize(const double* moments, float* mean_out, float* alpha_out, float* rstd, float* log_det, int D,
                       float eps, hipStream_t st);
int64_t flow_forward_batch_workspace(int64_t Mp, int D, int S, int L);
int flow_forward_batch_begin(const float* params, int64_t Mp, int D, int S, int L, int U, int64_t pstride, void* ws,
                             hipStream_t st);
int flow_forward_batch_layer(int c, const float* z_in, const float* params, float* z_out, float* sum_log_det,
                             double* moments, int64_t M, int64_t Mp, int64_t N, int D, int S, int L, int U, int64_t pstride,
                             void* ws, hipStream_t st);
int flow_forward_batch_fold(int c, const float* params, const double* moments, float* bn_mean_out, float* bn_alpha_out,
                            int64_t Mp, int D, int S, int L, int U, int64_t pstride, float eps, void* ws, hipStream_t st);
int flow_forward_batch_end(float* z_out, float* sum_log_det, int64_t M, int64_t Mp, int64_t N, int D, void* ws,
                           hipStream_t st);
int launch_flow_forward_batch(const float* omega, const float* params, float* z_out, float* sum_log_det,
                              float* bn_mean_out, float* bn_alpha_out, int64_t M, int64_t Mp, int64_t N, int D, int S,
                              int L, int U, int64_t pstride, float eps, void* ws, hipStream_t st);
int64_t flow_forward_train_workspace(int64_t M, int64_t Mp, int64_t N, int D, int S, int L);
int launch_flow_forward_train_fwd(const float* omega, const float* params, float* z_out, float* sum_log_det, float* states,
                                  float* folds, float* bn_mean_out, float* bn_alpha_out, int64_t M, int64_t Mp, int64_t N,
                                  int D, int S, int L, int U, int64_t pstride, float eps, void* ws, hipStream_t st);
int launch_flow_forward_train_bwd(const float* omega, const float* params, const float* states, const float* folds,
                                  const float* bn_mean, const float* bn_alpha, const float* g_z, const float* g_sld,
                                  float* g_omega, float* g_params, int64_t M, int64_t Mp, int64_t N, int D, int S, int L,
                                  int U, int64_t pstride, int64_t gpstride, void* ws, hipStream_t st);
// reversible whole-flow training backward (flow_bwd_f16.hip)
int flow_train_rev_supported(int D, int S, int L, int U);
int64_t flow_train_rev_workspace(int64_t M, int64_t Mp, int64_t N, int D, int S, int L, int U);
int launch_flow_bwd_rev(const float* z0, const float* params, const float* bn_mean, const float* bn_alpha,
                        const float* g_lp, float* g_z, float* g_params, int64_t M, int64_t Mp, int64_t N, int D, int S,
                        int L, int U, int64_t pstride, int64_t gpstride, void* ws, int* overflow_out, hipStream_t st);

int launch_affine(int dtype, const void* z, const void* params, void* z_out, void* log_det,
                  int64_t Mz, int64_t Mp, int64_t N, int D, int inverse, int64_t pstride,
                  hipStream_t st);
int launch_bn_apply(int dtype, const void* z, const float* mean, const float* alpha, void* z_out,
                    float* log_det, int64_t rows, int D, int inverse, hipStream_t st);
int launch_bn_batch_forward(const float* z, float* z_out, float* mean_out, float* alpha_out,
                            float* log_det, int64_t rows, int D, float eps, void* ws, hipStream_t st);

int launch_coupling_backward(int dtype, const void* z, const void* params, const void* g_zout,
                             const void* g_ld, void* g_z, void* g_params, int64_t M, int64_t Mp,
                             int64_t N, int D, int L, int U, int upper, int inverse, int64_t pstride,
                             int64_t gpstride, hipStream_t st, void* ws = nullptr, int64_t ws_bytes = -1);
// ws_bytes >= 0: deterministic reduction of the parameter gradient (partial rows in ws, added in workgroup order)
int64_t coupling_backward_workspace(int dtype, int64_t M, int64_t Mp, int64_t N, int D, int L, int U, int upper);
void backward_det_geometry(int64_t M, int64_t Mp, int64_t tiles_per_m, int* G, int64_t* rows);
int launch_backward_reduce(int dtype, const void* partials, void* g_params, int64_t rows, int G, int64_t P,
                           int64_t gpstride, hipStream_t st);
int64_t maf_backward_workspace(int dtype, int64_t M, int64_t Mp, int64_t N, int D, int L, int U);
// arguments of the MFMA backward kernel (coupling_bwd_mfma.hip)
struct BwdArgs {
    const float* z;
    const float* params;
    const float* g_zout;
    const float* g_ld;
    float* g_z;
    float* g_params;
    int64_t M, Mp, N, pstride, gpstride;
    int U, upper;
    // flow-level extensions (all optional / neutral by default):
    const float* image;   // prepared forward operand image of this layer (per mp: image_stride floats)
    int64_t image_stride;
    const float* fold;    // [A (D) | B (D)] applied to the saved input before the layer (per mp: fold_stride)
    float* g_fold;        // accumulates [dA (D) | dB (D)] (per mp: fold_stride)
    int64_t fold_stride;
    const float* g_lp;    // finalize: upstream gradient is that of log_prob = -|out|^2/2 - ... (M,N)
    float ld_scale;       // gradient w.r.t. sum(s) = ld_scale * g_ld[row]
    float* glp_sum;       // finalize: accumulates sum over the samples of g_lp (per mp), for the constant log-det
    const unsigned* gmax; // split-f16 layer kernel only: float bits of max |upstream gradient| of the whole chain (or NULL):
                          // the kernel works on gradients scaled by the power of two that brings it into [1, 2)
    const float* gcorr;   // [k0 (D) | k1 (D)] or NULL: the upstream gradient is g_zout + k0 + k1 * (this layer's output)
                          // (batch-statistics backward of the fold behind the layer, forward direction only)
    const int* gate;      // optional device flag: the kernel returns at once while *gate == 0 (tnf_set_launch_gate)
};
int launch_coupling_backward_mfma_args(const BwdArgs& a, int D, int L, int inverse, hipStream_t st);
int launch_coupling_backward_f16(const BwdArgs& a, int D, int L, int inverse, hipStream_t st);  // split-f16 (flow_bwd_f16.hip)
int launch_gmax(const float* g, int64_t n, unsigned* out, hipStream_t st);  // atomicMax of |g| (float bits) into *out
int launch_coupling_backward_mfma(const float* z, const float* params, const float* g_zout,
                                  const float* g_ld, float* g_z, float* g_params, int64_t M, int64_t Mp,
                                  int64_t N, int D, int L, int U, int upper, int inverse, int64_t pstride,
                                  int64_t gpstride, hipStream_t st);
int launch_affine_backward(int dtype, const void* z, const void* params, const void* g_zout,
                           const void* g_ld, void* g_z, void* g_params, int64_t M, int64_t Mp, int64_t N,
                           int D, int inverse, int64_t pstride, int64_t gpstride, hipStream_t st);
int launch_bn_apply_backward(int dtype, const void* g_zout, const float* alpha, void* g_z, int64_t rows,
                             int D, int inverse, hipStream_t st);

int launch_cond_flow_forward(const float* omega, const float* h, const float* W, const float* b, const float* bn_mean,
                             const float* bn_alpha, float* z_out, float* sum_log_det, int64_t M, int D, int S, int L, int U,
                             int H, int64_t ldh, int64_t ldw, void* ws, hipStream_t st);
int launch_bn_normalize_from_moments(const float* z, const double* moments, float* z_out, float* mean_out,
                                     float* alpha_out, float* log_det, float* rstd, int64_t rows, int D, float eps,
                                     hipStream_t st);
int launch_bn_batch_backward_sums(const float* zn, const float* g, double* sums, int64_t rows, int D, hipStream_t st);
int launch_bn_batch_backward_apply(const float* zn, const float* g, const float* g_ld, const float* alpha,
                                   const double* sums, const double* count, float* g_z, int64_t rows, int D,
                                   hipStream_t st);
int launch_bn_batch_backward(const float* zn, const float* g, const float* g_ld, const float* alpha, float* g_z,
                             int64_t rows, int D, void* ws, hipStream_t st);
bool cond_flow_supported(int D, int S, int L, int U, int H);
int64_t cond_flow_workspace(int D, int S, int L, int U, int H);
int launch_cond_flow_log_prob(const float* z, const float* h, const float* W, const float* b, const float* bn_mean,
                              const float* bn_alpha, float* log_prob, float* z0, float* sum_log_det, float* acts,
                              int64_t M, int D, int S, int L, int U, int H, int64_t ldh, int64_t ldw, void* ws,
                              hipStream_t st);
int64_t cond_acts_floats(int64_t M, int D, int S, int L);
int64_t cond_deltas_floats(int64_t M, int D, int S, int L, int H);
int64_t cond_flow_bwd_workspace(int D, int S, int L, int U, int H);
int launch_cond_flow_backward(const float* g_lp, const float* h, const float* W, const float* b, const float* bn_mean,
                              const float* bn_alpha, const float* acts, float* deltas, float* g_h, float* g_W,
                              float* g_b, float* g_z, int64_t M, int D, int S, int L, int U, int H, int64_t ldh,
                              int64_t ldw, int64_t ldgh, int64_t ldgw, void* ws, hipStream_t st);
int launch_to_interval(int dtype, const void* z, const float* consts, void* z_out, void* log_det, int64_t rows, int D,
                       int inverse, hipStream_t st);
int launch_to_interval_backward(int dtype, const void* z, const float* consts, const void* g_zout, const void* g_ld,
                                void* g_z, int64_t rows, int D, int inverse, hipStream_t st);
int launch_to_simplex(int dtype, const void* z, void* z_out, void* log_det, int64_t rows, int Din, int Dc,
                      hipStream_t st);
int launch_to_simplex_backward(int dtype, const void* z, const void* g_zout, const void* g_ld, void* g_z, int64_t rows,
                               int Din, int Dc, hipStream_t st);
// MAF on the matrix pipe (maf_mfma.hip): float32, D <= 64, U <= 64, L <= 5
struct MafArgs {
    const float* z;
    float* z_out;        // may be NULL when only log_prob is wanted
    const float* params;
    const float* masks;
    int64_t pstride;
    const float* pre;    // [A (D) | B (D)] per parameter row, or NULL
    const float* post;
    int64_t fold_stride;
    float* ld_out;       // (M, N) or NULL: ld_sign * sum(alpha) (+ ldc[m] when add_ldc)
    float ld_sign;
    const float* ldc;
    int add_ldc;
    float* log_prob;     // (M, N) or NULL: -|z'|^2/2 - D log sqrt(2 pi) - (ld_sign * sum(alpha) + ldc[m])
    const float* iv;     // ToInterval constants (7, D) of a fused support layer, or NULL
    int64_t Mz, Mp, N;
    int D, L, U, inverse;
    int bf16;            // != 0: the bf16 experiment (operands rounded to bf16, mfma_tile.h rbf16)
};

bool maf_mfma_supported(int D, int L, int U);
int launch_maf_mfma(const MafArgs& a, hipStream_t st);
int launch_ar_fold(const float* params, int64_t pstride, int64_t p_maf, const float* bn_mean, const float* bn_alpha,
                   float* fold, float* ldc, int64_t Mp, int D, int inverse, hipStream_t st);
bool maf_bwd_mfma_supported(int D, int L, int U);
int launch_ar_flow_backward(const float* z, const float* params, const float* masks, const float* fold,
                            const float* interval_consts, const float* g_lp, float* g_params, float* g_fold,
                            float* glp_sum, int64_t M, int64_t Mp, int64_t N, int D, int L, int U, int64_t pstride,
                            int64_t gpstride, hipStream_t st);
int launch_maf_backward_mfma(const float* z, const float* params, const float* masks, const float* g_zout,
                             const float* g_ld, float* g_z, float* g_params, int64_t M, int64_t Mp, int64_t N, int D,
                             int L, int U, int64_t pstride, int64_t gpstride, hipStream_t st);
int launch_maf(int dtype, const void* z, const void* params, const void* masks, void* z_out, void* log_det,
               int64_t Mz, int64_t Mp, int64_t N, int D, int L, int U, int inverse, int64_t pstride, hipStream_t st, void* alpha_out = nullptr);
int launch_maf_backward(int dtype, const void* z, const void* params, const void* masks, const void* g_zout,
                        const void* g_ld, void* g_z, void* g_params, int64_t M, int64_t Mp, int64_t N, int D, int L,
                        int U, int64_t pstride, int64_t gpstride, hipStream_t st, void* ws = nullptr, int64_t ws_bytes = -1);
int launch_base_log_density(int dtype, const void* omega, double* out, int64_t rows, int D, hipStream_t st);

}  // namespace tnf
