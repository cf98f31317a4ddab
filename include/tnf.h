/*
 * tnf.h -- C ABI of libtnf_hip.so: MI355X (gfx950) kernels for the torch_nf
 * coupling-flow hot path (RealNVP / BatchNorm / Affine forward, inverse, log_prob).
 *
 * The reference (srbittner/torch_nf) is pure Python on PyTorch-CPU and has no
 * FFI of its own; every entry point below therefore replaces a Python method of
 * the reference, cited as file:line relative to the upstream repo root.  The
 * binding a maintainer would add on the reference side is the ctypes stub shown
 * in INTEGRATION.md (and shipped as torch_nf_amd/_lib.py).
 *
 * Conventions
 *  - Plain C linkage, plain pointers and sizes; no torch types, no exceptions.
 *  - All data pointers are DEVICE pointers on the current HIP device, owned by
 *    the caller, contiguous row-major:
 *        z, z_out   (M, N, D)      sample-major: one sample = D contiguous values
 *        params     (M_p, >=|theta|) with `params_row_stride` elements per row,
 *                   packed exactly like the reference packs them
 *                   (bijectors.py:222-235, 281-287; density_estimator.py:379-402)
 *        log_det    (M, N)
 *    M = max(M_z, M_p); M_z and M_p must be equal or one of them 1 (broadcast,
 *    as torch.matmul does at bijectors.py:237-238).
 *  - `dtype`: TNF_F32 or TNF_F64 is the type of z / params / outputs.  BatchNorm
 *    statistics are always float32 (bijectors.py:345-346).
 *  - Kernels are enqueued asynchronously on `stream` (a hipStream_t passed as
 *    void*; NULL = the default stream) and never synchronise.
 *  - Return value: 0 on success, a negative TNF_E* code otherwise;
 *    tnf_last_error() then returns a thread-local message.  Nothing is launched
 *    when an argument check fails.
 */
#ifndef TNF_H
#define TNF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNF_VERSION 100 /* 0.1.0 */

enum { TNF_F32 = 0, TNF_F64 = 1 };

enum {
    TNF_OK = 0,
    TNF_EINVAL = -1,       /* bad argument (shape, NULL pointer, unsupported combination) */
    TNF_EUNSUPPORTED = -2, /* valid request that this build has no kernel for */
    TNF_ELAUNCH = -3,      /* HIP reported an error at launch */
    TNF_EWORKSPACE = -4    /* workspace too small */
};

/* log-det output modes of tnf_coupling */
enum { TNF_LD_STORE = 0, TNF_LD_ADD = 1, TNF_LD_SUB = -1 };

/* fusion granularity of the flow-level entry points */
enum {
    TNF_FUSE_AUTO = 0,  /* fastest available for the shape */
    TNF_FUSE_LAYER = 1, /* one fused kernel per coupling layer (k = 2*S launches) */
    TNF_FUSE_FLOW = 2   /* the whole flow in one kernel (k = 1) */
};

int tnf_version(void);
const char* tnf_last_error(void);

/* Testing hooks.  TNF_OPT_FORCE_GENERIC != 0 makes tnf_coupling skip the MFMA
 * specialisations so the tests can compare both kernels on the same inputs. */
enum {
    TNF_OPT_FORCE_GENERIC = 1,
    TNF_OPT_FLOW_VARIANT = 2,  /* tuning: (tiles per wave, waves per workgroup) of the whole-flow kernel */
    TNF_OPT_LAYER_VARIANT = 3, /* tuning: launch geometry of the per-layer kernel */
    TNF_OPT_COND_VARIANT = 4,  /* tuning: contexts per wave / waves per workgroup of the conditional-flow kernel */
    TNF_OPT_TRAIN_BWD_FP32 = 5, /* != 0: tnf_flow_forward_train_bwd_f32 uses the fp32-MFMA layer backward kernel */
    TNF_OPT_REV_VARIANT = 7,    /* whole-flow training backward: 0 (default) flow_bwd_f16_kernel, 1 the magic-number
                                 * form of flow_bwd_pair.h (an experiment that did not pay: DESIGN.md 3.11.1) */
    TNF_OPT_OPERAND_PREC = 6    /* 0 (default): fp32-accurate operands.  1: tnf_flow_log_prob_f32 and
                                 * tnf_flow_log_prob_fwd_rev_f32 round every conditioner operand to bf16 (one MFMA per
                                 * contraction) -- the precision experiment of BASELINE configs[4], not a parity path */
};
int tnf_set_option(int32_t key, int32_t value);
/* The calling thread's value of a key (options are per thread).  A caller that hands work to another thread -- the
 * Python binding's autograd Functions run their backward on autograd's device thread -- reads the options where the
 * forward ran and re-enters them where the backward runs. */
int tnf_get_option(int32_t key, int32_t* value);

/* Diagnostics: how many launches of a backward-kernel family this PROCESS has enqueued so far (any thread, any
 * stream).  Tests that select a kernel through an option read the counters around a step to prove the intended kernel
 * ran (a backward runs on autograd's thread, where a thread-local option set elsewhere would be silently absent). */
enum {
    TNF_DIAG_BWD_LAYER_FP32 = 0,  /* coupling_bwd_mfma_kernel: fp32-MFMA layer backward */
    TNF_DIAG_BWD_LAYER_F16 = 1,   /* coupling_bwd_f16_kernel: split-f16 layer backward */
    TNF_DIAG_BWD_GENERIC = 2,     /* coupling_backward_kernel<T>: shape-generic layer backward */
    TNF_DIAG_BWD_FLOW_REV = 3,    /* flow_bwd_f16_kernel: one-kernel reversible backward */
    TNF_DIAG_MAF_BWD_MFMA = 4,    /* maf_bwd_mfma_kernel */
    TNF_DIAG_MAF_BWD_GENERIC = 5, /* maf_backward_kernel<T> */
    TNF_DIAG_BWD_WIDE = 6,        /* coupling_wide_bwd_kernel: MFMA backward of the wide coupling shapes */
    TNF_DIAG_FAMILIES = 8
};
int64_t tnf_diag_launch_count(int32_t family);

/* Conditionally needed launches.  While a gate is set (per thread; NULL clears it), the layer kernels launched by
 * tnf_flow_log_prob_fwd_f32 / tnf_flow_log_prob_bwd_f32 read *flag on the device and return at once while it is 0:
 * the calls can be enqueued unconditionally behind a kernel that may or may not raise the flag, with no host round trip.
 * Used for the fp32 recomputation of a training backward whose fixed-point accumulators overflowed
 * (tnf_flow_log_prob_bwd_rev_f32's `overflow`): torch_nf_amd/ops.py:_FlowLogProbRevFn.  The small preparation kernels of
 * those calls run regardless (they write private workspace only).
 * tnf_gated_copy_f32: dst[0..n) = src[0..n) if *flag != 0, nothing otherwise. */
int tnf_set_launch_gate(const int32_t* flag);
int tnf_gated_copy_f32(const int32_t* flag, float* dst, const float* src, int64_t n, void* stream);

/* Number of packed parameters of one RealNVP layer / of the whole coupling flow.
 * Replaces RealNVP.count_num_params (bijectors.py:244-262) and
 * NormFlow.count_num_params (density_estimator.py:418-421) for arch_type
 * "coupling".  Negative on invalid arguments. */
int64_t tnf_coupling_num_params(int32_t D, int32_t num_layers, int32_t num_units,
                                int32_t transform_upper);
int64_t tnf_flow_num_params(int32_t D, int32_t num_stages, int32_t num_layers,
                            int32_t num_units);

/* Returns 1 if (D, L, U) has an MFMA kernel in this build (the specialised D = 32 / 64,
 * U <= 16, L <= 3 kernels, or the wide per-layer kernel: D % 8 == 0, D <= 128, U <= 64,
 * L <= 5), else 0 (the generic kernels then run it). */
int tnf_has_fast_path(int32_t D, int32_t num_layers, int32_t num_units);

/* ---- bijector level -------------------------------------------------- */

/* RealNVP.forward_and_log_det (bijectors.py:145-181) when inverse == 0,
 * RealNVP.inverse_and_log_det (bijectors.py:183-206) when inverse != 0;
 * the twin t/s MLP is RealNVP._t_s_layer (bijectors.py:208-242).
 * log_det receives sum(s) (the FORWARD log-det in both directions) according
 * to ld_mode.  The conditioner half of z is copied to z_out bit-identically.
 * z_out must not alias z. */
int tnf_coupling(int32_t dtype, const void* z, const void* params, void* z_out, void* log_det,
                 int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t num_layers,
                 int32_t num_units, int32_t transform_upper, int32_t inverse,
                 int64_t params_row_stride, int32_t ld_mode, void* stream);

/* Affine.forward_and_log_det (bijectors.py:277-295) / inverse_and_log_det
 * (bijectors.py:297-315).  params = [alpha (D) | shift (D)].
 * log_det (M_p, 1) receives sum(alpha). */
int tnf_affine(int32_t dtype, const void* z, const void* params, void* z_out, void* log_det,
               int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t inverse,
               int64_t params_row_stride, void* stream);

/* BatchNorm with cached statistics: inverse != 0 -> z*alpha + mean
 * (bijectors.py:420-426); inverse == 0 -> (z - mean)/alpha, the use_last=True
 * branch (bijectors.py:397-399).  log_det (one float32, like the statistics) receives
 * -sum(log(alpha)).  rows = M*N. */
int tnf_bn_apply(int32_t dtype, const void* z, const float* mean, const float* alpha, void* z_out,
                 float* log_det, int64_t rows, int32_t D, int32_t inverse, void* stream);

/* BatchNorm with batch statistics over all rows (use_last=False branch,
 * bijectors.py:401-417), float32: z_out = (z - mu)/sqrt(var_biased + eps);
 * mean_out / alpha_out (D) receive the statistics the reference caches
 * (alpha = sqrt(var_unbiased(z)) / sqrt(var_unbiased(z_out)), mean = mean(z - z_out*alpha));
 * log_det (1) = -sum(log(alpha)).  workspace: tnf_bn_batch_workspace_bytes(D) bytes. */
int64_t tnf_bn_batch_workspace_bytes(int32_t D);
int tnf_bn_batch_forward_f32(const float* z, float* z_out, float* mean_out, float* alpha_out,
                             float* log_det, int64_t rows, int32_t D, float eps, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* ---- backward passes (what torch autograd derives for the reference) ----------------
 * Each recomputes the layer from its saved INPUT z; nothing else is kept from the forward
 * pass.  g_z_out / g_log_det are the gradients w.r.t. the forward outputs; g_z receives the
 * gradient w.r.t. z (M,N,D) and g_params ACCUMULATES (float atomics; zero it first) the
 * gradient w.r.t. the packed parameter rows (M_p rows of g_params_row_stride elements).
 * z must already be broadcast to M rows (M_p is 1 or M). */
int tnf_coupling_backward(int32_t dtype, const void* z, const void* params, const void* g_z_out,
                          const void* g_log_det, void* g_z, void* g_params, int64_t M, int64_t M_p,
                          int64_t N, int32_t D, int32_t num_layers, int32_t num_units,
                          int32_t transform_upper, int32_t inverse, int64_t params_row_stride,
                          int64_t g_params_row_stride, void* stream);
/* The same with a caller-owned workspace (tnf_coupling_backward_workspace_bytes): the shape-generic kernel -- every
 * shape the MFMA backward does not cover: num_units > 16, odd D, float64 ... -- then reduces the parameter gradient
 * DETERMINISTICALLY: a fixed number of persistent workgroups per parameter row, each summing its tiles in order into its
 * own partial row, the rows added in workgroup order (bit-reproducible gradients; no float atomics).  The narrow MFMA
 * shapes take no workspace (0 bytes) and keep their atomic tail. */
int64_t tnf_coupling_backward_workspace_bytes(int32_t dtype, int64_t M, int64_t M_p, int64_t N, int32_t D,
                                              int32_t num_layers, int32_t num_units, int32_t transform_upper);
int tnf_coupling_backward_ws(int32_t dtype, const void* z, const void* params, const void* g_z_out,
                             const void* g_log_det, void* g_z, void* g_params, int64_t M, int64_t M_p,
                             int64_t N, int32_t D, int32_t num_layers, int32_t num_units,
                             int32_t transform_upper, int32_t inverse, int64_t params_row_stride,
                             int64_t g_params_row_stride, void* workspace, int64_t workspace_bytes, void* stream);
/* g_log_det is (M_p, 1) like the forward log_det. */
int tnf_affine_backward(int32_t dtype, const void* z, const void* params, const void* g_z_out,
                        const void* g_log_det, void* g_z, void* g_params, int64_t M, int64_t M_p,
                        int64_t N, int32_t D, int32_t inverse, int64_t params_row_stride,
                        int64_t g_params_row_stride, void* stream);
/* Cached-statistics BatchNorm: g_z = g_z_out * alpha (inverse) or / alpha (frozen forward);
 * the statistics are constants of the graph. */
int tnf_bn_apply_backward(int32_t dtype, const void* g_z_out, const float* alpha, void* g_z, int64_t rows,
                          int32_t D, int32_t inverse, void* stream);

/* Backward of tnf_bn_batch_forward_f32: with x^ = z_out of the forward call and n = rows,
 *   g_z = (g - mean(g) - x^ (mean(g x^) + g_log_det/n)) / alpha
 * g_log_det: device pointer to ONE float (gradient w.r.t. the 0-dim log_det), may be NULL (= 0).
 * workspace: tnf_bn_batch_workspace_bytes(D) bytes. */
int tnf_bn_batch_backward_f32(const float* z_norm, const float* g_z_out, const float* g_log_det,
                              const float* alpha, float* g_z, int64_t rows, int32_t D, void* workspace,
                              int64_t workspace_bytes, void* stream);

/* Sample-sharded batch statistics WITH gradients (one process per GPU, each holding `rows` rows of the batch): the two
 * calls above cut at their exchange step.  The reference computes the statistics of the whole batch
 * (bijectors.py:401-410) and differentiates through them (:414-415); sharded, the per-feature sums have to cross ranks
 * once in each direction -- torch_nf_amd/distributed.py all-reduces the small buffers between the halves:
 *   forward   tnf_bn_batch_moments_f32   moments (2 D + 1 doubles) = [sum | sum of squares | rows] of THIS rank's rows
 *             -- all-reduce(moments, SUM) --
 *             tnf_bn_batch_normalize_f32 statistics from the reduced moments (count read from moments[2 D]),
 *                                        z_out = this rank's rows normalised; workspace: D floats
 *   backward  tnf_bn_batch_backward_sums_f32   sums (2 D doubles) = [sum g | sum g x^] of this rank's rows
 *             -- all-reduce(sums, SUM) --
 *             tnf_bn_batch_backward_apply_f32  g_z = (g - sum g / n - x^ (sum g x^ + g_log_det) / n) / alpha with
 *                                              n = *count (device pointer to the reduced moments[2 D])
 * rows may be 0 (an empty shard still takes part in the reductions). */
int tnf_bn_batch_moments_f32(const float* z, double* moments, int64_t rows, int32_t D, void* stream);
int tnf_bn_batch_normalize_f32(const float* z, const double* moments, float* z_out, float* mean_out, float* alpha_out,
                               float* log_det, int64_t rows, int32_t D, float eps, void* workspace,
                               int64_t workspace_bytes, void* stream);
int tnf_bn_batch_backward_sums_f32(const float* z_norm, const float* g_z_out, double* sums, int64_t rows, int32_t D,
                                   void* stream);
int tnf_bn_batch_backward_apply_f32(const float* z_norm, const float* g_z_out, const float* g_log_det,
                                    const float* alpha, const double* sums, const double* count, float* g_z,
                                    int64_t rows, int32_t D, void* stream);

/* ---- MAF, the bijector of NormFlow's default arch_type "AR" (bijectors.py:597-806) ----------
 * Twin masked MLPs without biases; params = [W_mu0 | W_alpha0 | ... | W_mu_last | W_alpha_last]
 * (bijectors.py:698-740), W row-major [in][out]; `masks` = the binary matrices Ms of
 * MAF._get_masks (bijectors.py:663-696) concatenated in layer order (D*U, (L-1) x U*U, U*D
 * elements of `dtype`), shared by all parameter rows.
 * inverse != 0: MAF.inverse_and_log_det (one pass, :758-764); inverse == 0:
 * MAF.forward_and_log_det (D-1 sequential passes, :742-756).  log_det (M,N) = sum(f_alpha).
 * tnf_maf_backward differentiates the INVERSE direction (what log_prob training needs). */
int64_t tnf_maf_num_params(int32_t D, int32_t num_layers, int32_t num_units);
int tnf_maf(int32_t dtype, const void* z, const void* params, const void* masks, void* z_out, void* log_det,
            int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t num_layers, int32_t num_units,
            int32_t inverse, int64_t params_row_stride, void* stream);
/* MAF.inverse_and_log_det (bijectors.py:758-764) with the per-dimension f_alpha(z) (M,N,D) as an extra output.
 * The backward of the sampling direction (autograd through MAF.forward_and_log_det, bijectors.py:752-754) is built on
 * it: x solves G(x, theta) = omega with G the inverse map, whose Jacobian is triangular with diagonal e^-alpha. */
int tnf_maf_inverse_alpha(int32_t dtype, const void* z, const void* params, const void* masks, void* z_out, void* log_det,
                          void* alpha_out, int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t num_layers,
                          int32_t num_units, int64_t params_row_stride, void* stream);

int tnf_maf_backward(int32_t dtype, const void* z, const void* params, const void* masks, const void* g_z_out,
                     const void* g_log_det, void* g_z, void* g_params, int64_t M, int64_t M_p, int64_t N,
                     int32_t D, int32_t num_layers, int32_t num_units, int64_t params_row_stride,
                     int64_t g_params_row_stride, void* stream);
/* ... with a caller-owned workspace: deterministic reduction in the shape-generic kernel (D > 32 and the other shapes the
 * MFMA backward does not cover), exactly as tnf_coupling_backward_ws. */
int64_t tnf_maf_backward_workspace_bytes(int32_t dtype, int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t num_layers,
                                         int32_t num_units);
int tnf_maf_backward_ws(int32_t dtype, const void* z, const void* params, const void* masks, const void* g_z_out,
                        const void* g_log_det, void* g_z, void* g_params, int64_t M, int64_t M_p, int64_t N,
                        int32_t D, int32_t num_layers, int32_t num_units, int64_t params_row_stride,
                        int64_t g_params_row_stride, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- conditional flow, one sample per context (SNPE layout) ------------- */
/* ConditionalDensityEstimator.log_prob(z[:, None, :], x) (conditional_density_estimator.py:101-104)
 * with the last Linear of param_net fused into the flow: params[m] = W . h[m] + b is generated
 * tile by tile on the matrix cores and consumed immediately, the (M, D_params) tensor never exists.
 *   z (M, D) float32; h (M, ldh) float32, the first H columns = output of param_net's last activation;
 *   W (D_params, ldw) / b (D_params) = weight / bias of param_net's last Linear (torch layout);
 *   bn_mean / bn_alpha (2*S, D); log_prob (M); z0 (M, D) and sum_log_det (M) optional (NULL).
 * Supported (tnf_cond_flow_supported): arch_type "coupling", D in {32, 64}, num_units <= 16,
 * num_layers <= 5, H in {32, 64, 128} (pad h and W with zero columns for other widths).
 * Rows of h and W must be 16-byte aligned (ldh, ldw multiples of 4). */
int tnf_cond_flow_supported(int32_t D, int32_t num_stages, int32_t num_layers, int32_t num_units, int32_t H);
int64_t tnf_cond_flow_workspace_bytes(int32_t D, int32_t num_stages, int32_t num_layers, int32_t num_units,
                                      int32_t H);
int tnf_cond_flow_log_prob_f32(const float* z, const float* h, const float* W, const float* b,
                               const float* bn_mean, const float* bn_alpha, float* log_prob, float* z0,
                               float* sum_log_det, int64_t M, int32_t D, int32_t num_stages, int32_t num_layers,
                               int32_t num_units, int32_t H, int64_t ldh, int64_t ldw, void* workspace,
                               int64_t workspace_bytes, void* stream);

/* The SAMPLING direction of the same path: ConditionalDensityEstimator.__call__(x, N = 1) with frozen statistics
 * (conditional_density_estimator.py:93-99 over density_estimator.py:374-388): omega (M, D) base draws, one per
 * context, pushed forwards through the flow whose parameters are generated from h on the fly; z_out (M, D),
 * sum_log_det (M) = the forward log-dets (log q(z) = log N(omega; 0, I) - sum_log_det).  Same shapes, alignment and
 * workspace (tnf_cond_flow_workspace_bytes) as tnf_cond_flow_log_prob_f32. */
int tnf_cond_flow_forward_f32(const float* omega, const float* h, const float* W, const float* b,
                              const float* bn_mean, const float* bn_alpha, float* z_out, float* sum_log_det, int64_t M,
                              int32_t D, int32_t num_stages, int32_t num_layers, int32_t num_units, int32_t H,
                              int64_t ldh, int64_t ldw, void* workspace, int64_t workspace_bytes, void* stream);

/* Training pair of the same path.  tnf_cond_flow_log_prob_fwd_f32 = tnf_cond_flow_log_prob_f32 that also
 * saves, in `acts` (tnf_cond_flow_acts_floats(M, D, S, L) floats), the activations the backward needs.
 * tnf_cond_flow_log_prob_bwd_f32: from g_log_prob (M) to the gradients of param_net's last Linear --
 * g_W (D_params, ldgw), g_b (D_params), both overwritten -- and of its input, g_h (M, ldgh); g_z (M, D)
 * optional (NULL).  `deltas` is scratch of tnf_cond_flow_deltas_floats(M, D, S, L, H) floats; the workspace
 * needs tnf_cond_flow_bwd_workspace_bytes.  Neither params (M, D_params) nor their gradient is ever
 * materialised (torch autograd through the reference would hold both: 2 x 82 KB per context at D = 64). */
int64_t tnf_cond_flow_acts_floats(int64_t M, int32_t D, int32_t num_stages, int32_t num_layers);
int64_t tnf_cond_flow_deltas_floats(int64_t M, int32_t D, int32_t num_stages, int32_t num_layers, int32_t H);
int64_t tnf_cond_flow_bwd_workspace_bytes(int32_t D, int32_t num_stages, int32_t num_layers, int32_t num_units,
                                          int32_t H);
int tnf_cond_flow_log_prob_fwd_f32(const float* z, const float* h, const float* W, const float* b,
                                   const float* bn_mean, const float* bn_alpha, float* log_prob, float* acts,
                                   int64_t M, int32_t D, int32_t num_stages, int32_t num_layers, int32_t num_units,
                                   int32_t H, int64_t ldh, int64_t ldw, void* workspace, int64_t workspace_bytes,
                                   void* stream);
int tnf_cond_flow_log_prob_bwd_f32(const float* g_log_prob, const float* h, const float* W, const float* b,
                                   const float* bn_mean, const float* bn_alpha, const float* acts, float* deltas,
                                   float* g_h, float* g_W, float* g_b, float* g_z, int64_t M, int32_t D,
                                   int32_t num_stages, int32_t num_layers, int32_t num_units, int32_t H, int64_t ldh,
                                   int64_t ldw, int64_t ldgh, int64_t ldgw, void* workspace, int64_t workspace_bytes,
                                   void* stream);

/* ---- support layers (parameter-free bijectors appended by NormFlow(..., support_layer=)) ---- */
/* ToInterval.forward_and_log_det / inverse_and_log_det (bijectors.py:509-557).  z, z_out (rows, D) and
 * log_det (rows) of `dtype`; consts is a (7, D) float32 device array holding the bijector's constant rows
 * tanh_flg, softplus_flg, tanh_m, tanh_c, softplus_m, softplus_c (bijectors.py:475-480) and
 * log(tanh_m) evaluated in float32 like the reference does.  The inverse returns the forward
 * log-det at the recovered point, like the reference.  eps = 1e-12 (:445, :556).
 * tnf_to_interval_backward: g_z = g_z_out * d z_out/d z + g_log_det[row] * d log_det/d z for the
 * direction `inverse`, recomputed from the layer's input z. */
int tnf_to_interval(int32_t dtype, const void* z, const float* consts, void* z_out, void* log_det, int64_t rows,
                    int32_t D, int32_t inverse, void* stream);
int tnf_to_interval_backward(int32_t dtype, const void* z, const float* consts, const void* g_z_out,
                             const void* g_log_det, void* g_z, int64_t rows, int32_t D, int32_t inverse,
                             void* stream);
/* ToSimplex.forward_and_log_det (bijectors.py:574-591): z (rows, D_in) -> z_out (rows, D_in + 1) on the
 * simplex, log_det (rows) = log(1 - S/(S+1) + 1e-10) - D_attr*log(S+1) + sum z with S = sum exp z and
 * D_attr the bijector's own D attribute (the reference uses self.D whatever the input width).  The
 * reference defines no inverse.  tnf_to_simplex_backward: gradient w.r.t. z given g_z_out (rows, D_in+1)
 * and g_log_det (rows). */
int tnf_to_simplex(int32_t dtype, const void* z, void* z_out, void* log_det, int64_t rows, int32_t D_in,
                   int32_t D_attr, void* stream);
int tnf_to_simplex_backward(int32_t dtype, const void* z, const void* g_z_out, const void* g_log_det, void* g_z,
                            int64_t rows, int32_t D_in, int32_t D_attr, void* stream);

/* ---- flow level, arch_type "AR": [MAF, BatchNorm, Affine] (density_estimator.py:271-274), float32 ---- */
/* One kernel per call: the Affine and the cached-statistics BatchNorm fold into a per-feature FMA in front of
 * (log_prob: Affine^-1, BN^-1, MAF^-1, base density; :390-416) or behind (frozen forward: MAF, BN, Affine;
 * :374-388 with use_last=True) the MAF kernel.  params rows: [MAF | alpha (D) | shift (D)]; bn_mean /
 * bn_alpha (D).  tnf_ar_flow_log_prob_f32: any of log_prob, z0, sum_log_det may be NULL (not all).
 * interval_consts: NULL, or the (7, D) constant block of a ToInterval support layer (see tnf_to_interval)
 * appended to the stack (density_estimator.py:278-282): its inverse then runs in the kernel's load stage
 * (log_prob) / its forward map in the store stage (forward), with hardware transcendentals, and its
 * log-det is included in sum_log_det / log_prob.
 * Supported (tnf_ar_flow_supported): D <= 64, num_units <= 64, num_layers <= 5. */
int tnf_ar_flow_supported(int32_t D, int32_t num_layers, int32_t num_units);
int64_t tnf_ar_flow_workspace_bytes(int64_t M_p, int32_t D);
int tnf_ar_flow_log_prob_f32(const float* z, const float* params, const float* masks, const float* bn_mean,
                             const float* bn_alpha, const float* interval_consts, float* log_prob, float* z0,
                             float* sum_log_det, int64_t M_z,
                             int64_t M_p, int64_t N, int32_t D, int32_t num_layers, int32_t num_units,
                             int64_t params_row_stride, void* workspace, int64_t workspace_bytes, void* stream);
int tnf_ar_flow_forward_f32(const float* omega, const float* params, const float* masks, const float* bn_mean,
                            const float* bn_alpha, const float* interval_consts, float* z_out, float* sum_log_det,
                            int64_t M_z, int64_t M_p,
                            int64_t N, int32_t D, int32_t num_layers, int32_t num_units, int64_t params_row_stride,
                            void* workspace, int64_t workspace_bytes, void* stream);

/* Training through NormFlow(arch_type="AR").log_prob (density_estimator.py:390-416 under autograd, the inner
 * loop of scripts/lfi_mat.py:23-57): gradient of sum(g_log_prob * log_prob) w.r.t. the parameter rows
 * [MAF | Affine alpha | Affine shift] in ONE backward kernel -- it applies ToInterval^-1 (interval_consts,
 * may be NULL) and the folded Affine^-1 . BatchNorm^-1 itself, recomputes the MAF, seeds the backward from
 * g_log_prob (base density and log-dets) and reduces the MAF and Affine gradients.  z (M,N,D) is a constant of
 * the graph (no g_z); the forward is tnf_ar_flow_log_prob_f32.  g_params: M_p rows, accumulated, zero it first.
 * BatchNorm statistics are constants.  Shapes: tnf_ar_flow_train_supported() (D <= 32, num_layers <= 3). */
int tnf_ar_flow_train_supported(int32_t D, int32_t num_layers, int32_t num_units);
int64_t tnf_ar_flow_bwd_workspace_bytes(int64_t M_p, int32_t D);
int tnf_ar_flow_log_prob_bwd_f32(const float* z, const float* params, const float* masks, const float* bn_mean,
                                 const float* bn_alpha, const float* interval_consts, const float* g_log_prob,
                                 float* g_params, int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t num_layers,
                                 int32_t num_units, int64_t params_row_stride, int64_t g_params_row_stride,
                                 void* workspace, int64_t workspace_bytes, void* stream);

/* Base density of NormFlow.forward, float64 like the reference's numpy expression
 * log(prod_d exp(-w_d^2/2)/sqrt(2 pi)) (density_estimator.py:369-372), evaluated as
 * sum_d(-w_d^2/2) - D*log(sqrt(2 pi)) in float64.  omega (rows, D) of `dtype` (TNF_F64 for
 * the reference's host draw, TNF_F32 for a device-side draw) -> out (rows) float64. */
int tnf_base_log_density_f64(int32_t dtype, const void* omega, double* out, int64_t rows, int32_t D,
                             void* stream);

/* ---- flow level (arch_type "coupling", float32) ------------------------ */
/* Stack per stage: RealNVP(upper), BN, RealNVP(lower), BN, Affine
 * (density_estimator.py:260-270); bn_mean / bn_alpha are (2*S, D) float32 in
 * forward order.  Workspace size for either entry point: */
int64_t tnf_flow_workspace_bytes(int64_t M, int64_t N, int32_t D, int32_t num_stages, int32_t num_layers,
                                 int32_t num_units, int32_t fusion);

/* Returns 1 if the whole-flow kernel (TNF_FUSE_FLOW) exists for this configuration
 * (MFMA fast path and all 2*S layers' operands fit the 160 KB of LDS), else 0. */
int tnf_flow_fused_supported(int32_t D, int32_t num_stages, int32_t num_layers, int32_t num_units);

/* NormFlow.log_prob (density_estimator.py:408-416) = inverse_and_log_det
 * (density_estimator.py:390-406) + base Gaussian.  Outputs, each optional (NULL):
 *   log_prob (M,N), z0 (M,N,D) the base-space point, sum_log_det (M,N).
 * interval_consts (here and in tnf_flow_forward_f32): NULL, or the (7, D) constants of a ToInterval
 * support layer appended to the stack (see tnf_to_interval), evaluated in the whole-flow kernel's
 * load stage (log_prob) / store stage (forward); only with the whole-flow kernel (else TNF_EUNSUPPORTED). */
int tnf_flow_log_prob_f32(const float* z, const float* params, const float* bn_mean,
                          const float* bn_alpha, const float* interval_consts, float* log_prob, float* z0,
                          float* sum_log_det, int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t num_stages,
                          int32_t num_layers, int32_t num_units, int64_t params_row_stride,
                          int32_t fusion, void* workspace, int64_t workspace_bytes, void* stream);

/* tnf_flow_log_prob_f32 with a diagnostic output: *exact_reruns (one uint32 on the device, zeroed by the caller) is
 * incremented once per 32-sample group that the whole-flow kernel had to re-run with exact fp32 first-layer
 * contractions because a conditioner input left the range of its split-f16 operand (|x| 2^-kappa >= 65520; see
 * torch_nf_amd/csrc/f16_tile2.h).  The results are the same with or without the counter; the reference
 * (bijectors.py:237-241, plain fp32 matmul) has no such range, and neither has the result of this call. */
int tnf_flow_log_prob_diag_f32(const float* z, const float* params, const float* bn_mean, const float* bn_alpha,
                               const float* interval_consts, float* log_prob, float* z0, float* sum_log_det, int64_t M_z,
                               int64_t M_p, int64_t N, int32_t D, int32_t num_stages, int32_t num_layers,
                               int32_t num_units, int64_t params_row_stride, int32_t fusion, void* workspace,
                               int64_t workspace_bytes, void* stream, uint32_t* exact_reruns);

/* Training pair for loss = f(NormFlow.log_prob(z)) (the reference differentiates
 * density_estimator.py:390-416 with torch autograd).  The forward runs one fused kernel per
 * coupling layer and keeps each kernel's INPUT in `states` (2*S - 1, M, N, D) -- states[c] is the
 * input of layer kernel c; the last-applied layer's input is z itself.  The backward walks the
 * layers the other way with one MFMA backward kernel each (recompute, folded BatchNorm/Affine,
 * base density on the first), accumulating into g_params (M_p rows of g_params_row_stride floats,
 * zero it first; float atomics) and writing g_z (M,N,D).  M_p is 1 or M; z has M rows.
 * Available for the shapes of tnf_flow_fused_supported(); BatchNorm statistics are constants. */
int64_t tnf_flow_train_workspace_bytes(int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t num_stages,
                                       int32_t num_layers, int32_t num_units);
int tnf_flow_log_prob_fwd_f32(const float* z, const float* params, const float* bn_mean, const float* bn_alpha,
                              float* log_prob, float* states, int64_t M, int64_t M_p, int64_t N, int32_t D,
                              int32_t num_stages, int32_t num_layers, int32_t num_units,
                              int64_t params_row_stride, void* workspace, int64_t workspace_bytes, void* stream);
int tnf_flow_log_prob_bwd_f32(const float* z, const float* states, const float* params, const float* bn_mean,
                              const float* bn_alpha, const float* g_log_prob, float* g_z, float* g_params,
                              int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t num_stages,
                              int32_t num_layers, int32_t num_units, int64_t params_row_stride,
                              int64_t g_params_row_stride, void* workspace, int64_t workspace_bytes, void* stream);

/* NormFlow.forward with freeze_bn=False and no autograd (the default sampling call: density_estimator.py:374-388
 * with the batch-statistics BatchNorm of bijectors.py:401-415) as ONE call: per coupling layer a fused kernel
 * whose load stage applies the BatchNorm (and Affine) in front of it, then the batch statistics of its output.
 * Outputs: z_out (M,N,D), sum_log_det (M,N) = sum of the forward log-dets, and the statistics every BatchNorm
 * layer caches, bn_mean_out / bn_alpha_out (2*num_stages, D), forward order.  Statistics are over all M*N rows. */
int64_t tnf_flow_forward_batch_workspace_bytes(int64_t M_p, int32_t D, int32_t num_stages, int32_t num_layers);
int tnf_flow_forward_batch_f32(const float* omega, const float* params, float* z_out, float* sum_log_det,
                               float* bn_mean_out, float* bn_alpha_out, int64_t M, int64_t M_p, int64_t N, int32_t D,
                               int32_t num_stages, int32_t num_layers, int32_t num_units, int64_t params_row_stride,
                               float eps, void* workspace, int64_t workspace_bytes, void* stream);

/* The same chain in steps, for a caller that shards the samples over ranks ("one all-reduce of
 * [sum z, sum z^2] per BatchNorm layer ... to reproduce the reference's full-batch statistics", bijectors.py:401-410).
 * All steps share one workspace of tnf_flow_forward_batch_workspace_bytes() that must stay untouched in between:
 *   begin                 builds the layers' operand images;
 *   layer c (0..2S-1)     coupling layer c of this rank's rows (z_in = omega for c = 0, else z_out in place), then the
 *                         LOCAL moments of its output: moments = [sum (D) | sum of squares (D) | row count], 2D+1
 *                         doubles on the device, overwritten;
 *   -- the caller sums `moments` over the ranks (RCCL all-reduce on the same stream); single rank: nothing --
 *   fold c                statistics of the (global) batch -> rows c of bn_mean_out / bn_alpha_out (2S, D), and the
 *                         BatchNorm (+ Affine) fold that the next layer's load stage applies;
 *   end                   the last fold as an elementwise pass over z_out; sum_log_det += the constant log-dets.
 * tnf_flow_forward_batch_f32 is this sequence with the local moments. */
int tnf_flow_forward_batch_begin_f32(const float* params, int64_t M_p, int32_t D, int32_t num_stages, int32_t num_layers,
                                     int32_t num_units, int64_t params_row_stride, void* workspace,
                                     int64_t workspace_bytes, void* stream);
int tnf_flow_forward_batch_layer_f32(int32_t layer, const float* z_in, const float* params, float* z_out,
                                     float* sum_log_det, double* moments, int64_t M, int64_t M_p, int64_t N, int32_t D,
                                     int32_t num_stages, int32_t num_layers, int32_t num_units,
                                     int64_t params_row_stride, void* workspace, int64_t workspace_bytes, void* stream);
int tnf_flow_forward_batch_fold_f32(int32_t layer, const float* params, const double* moments, float* bn_mean_out,
                                    float* bn_alpha_out, int64_t M_p, int32_t D, int32_t num_stages, int32_t num_layers,
                                    int32_t num_units, int64_t params_row_stride, float eps, void* workspace,
                                    int64_t workspace_bytes, void* stream);
int tnf_flow_forward_batch_end_f32(float* z_out, float* sum_log_det, int64_t M, int64_t M_p, int64_t N, int32_t D,
                                   int32_t num_stages, int32_t num_layers, void* workspace, int64_t workspace_bytes,
                                   void* stream);

/* The same call under autograd (objectives on samples z = nf(N) and their log-density, e.g. the reference's
 * train_efn loop, with fresh batch statistics).  Forward: as above, out of place -- states (2*num_stages, M,N,D)
 * keeps every coupling layer's output before the fold behind it, folds (2*num_stages, M_p, 2, D) that fold's
 * constants.  Backward: given g_z (M,N,D) and g_sum_log_det (M,N), one coupling backward kernel per layer (its
 * per-context fold sums are the sums the batch-statistics backward needs) plus the fold's own backward -- gradients
 * through the batch mean and variance included; g_params (M_p rows, accumulated, zero it first), g_omega optional. */
int64_t tnf_flow_forward_train_workspace_bytes(int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t num_stages,
                                               int32_t num_layers);
int tnf_flow_forward_train_fwd_f32(const float* omega, const float* params, float* z_out, float* sum_log_det,
                                   float* states, float* folds, float* bn_mean_out, float* bn_alpha_out, int64_t M,
                                   int64_t M_p, int64_t N, int32_t D, int32_t num_stages, int32_t num_layers,
                                   int32_t num_units, int64_t params_row_stride, float eps, void* workspace,
                                   int64_t workspace_bytes, void* stream);
int tnf_flow_forward_train_bwd_f32(const float* omega, const float* params, const float* states, const float* folds,
                                   const float* bn_mean, const float* bn_alpha, const float* g_z, const float* g_sum_log_det,
                                   float* g_omega, float* g_params, int64_t M, int64_t M_p, int64_t N, int32_t D,
                                   int32_t num_stages, int32_t num_layers, int32_t num_units, int64_t params_row_stride,
                                   int64_t g_params_row_stride, void* workspace, int64_t workspace_bytes, void* stream);

/* Reversible training pair for the same loss (density_estimator.py:390-416 under autograd).  The
 * coupling stack is invertible, so the forward is the whole-flow kernel of tnf_flow_log_prob_f32
 * and keeps only its output z0 (M,N,D); the backward is ONE kernel that walks the layers from z0,
 * rebuilding each layer's input from its output while it back-propagates (split-f16 MFMA, fp32
 * accumulate).  g_z may be NULL when the gradient w.r.t. z is not wanted; g_params as above
 * (accumulated, zero it first).  BatchNorm statistics are constants.  Available when
 * tnf_flow_train_rev_supported() is 1 (D in {32, 64}, num_units <= 16, the layers' gradient
 * accumulators fit the 160 KB LDS).
 * The parameter gradient is REPRODUCIBLE bit for bit (like the reference's CPU loop,
 * notebooks/LFI_learning_rules.ipynb:295-304): workgroups accumulate in 32-bit fixed point (integer adds commute)
 * and store their partial rows; a second kernel adds the rows in block order in 64-bit integers.
 * `overflow` (device int32, may be NULL): set to 1 when a gradient term exceeded the fixed-point budget (2^13 in units
 * where max |g_log_prob| is in [1, 2): a heavy-tailed sample whose deltas also approach the f16 range of the
 * split operands) -- the returned g_params rows are then NaN, never a wrapped sum, and the caller re-runs the step
 * through tnf_flow_log_prob_fwd_f32 / _bwd_f32 with fp32 layer kernels (TNF_OPT_TRAIN_BWD_FP32), which have no such
 * budget; torch_nf_amd.ops does exactly that. */
int tnf_flow_train_rev_supported(int32_t D, int32_t num_stages, int32_t num_layers, int32_t num_units);
int64_t tnf_flow_train_rev_workspace_bytes(int64_t M, int64_t M_p, int64_t N, int32_t D, int32_t num_stages,
                                           int32_t num_layers, int32_t num_units);
int tnf_flow_log_prob_fwd_rev_f32(const float* z, const float* params, const float* bn_mean, const float* bn_alpha,
                                  float* log_prob, float* z0, int64_t M, int64_t M_p, int64_t N, int32_t D,
                                  int32_t num_stages, int32_t num_layers, int32_t num_units,
                                  int64_t params_row_stride, void* stream);
int tnf_flow_log_prob_bwd_rev_f32(const float* z0, const float* params, const float* bn_mean, const float* bn_alpha,
                                  const float* g_log_prob, float* g_z, float* g_params, int64_t M, int64_t M_p,
                                  int64_t N, int32_t D, int32_t num_stages, int32_t num_layers, int32_t num_units,
                                  int64_t params_row_stride, int64_t g_params_row_stride, void* workspace,
                                  int64_t workspace_bytes, int32_t* overflow, void* stream);

/* The deterministic part of NormFlow.forward with freeze_bn=True
 * (density_estimator.py:374-388): pushes base samples `omega` (M,N,D) through
 * the stack.  Outputs: z_out (M,N,D); sum_log_det (M,N) = sum of the forward
 * log-dets (the caller subtracts it from the base log-density, :387). */
int tnf_flow_forward_f32(const float* omega, const float* params, const float* bn_mean,
                         const float* bn_alpha, const float* interval_consts, float* z_out, float* sum_log_det,
                         int64_t M_z, int64_t M_p, int64_t N, int32_t D, int32_t num_stages, int32_t num_layers,
                         int32_t num_units, int64_t params_row_stride, int32_t fusion,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* The same plus the log-density NormFlow.forward returns beside the samples (density_estimator.py:369-372, 387):
 * log_q[m][n] = log N(omega[m][n]; 0, I) - sum_log_det[m][n], float64 (the base density summed in float64 from the
 * float32 draw), written by the whole-flow kernel itself -- no separate pass over omega.  TNF_EUNSUPPORTED unless the
 * default whole-flow kernel runs the call (fusion AUTO / FLOW, tnf_flow_fused_supported shapes). */
int tnf_flow_forward_logq_f32(const float* omega, const float* params, const float* bn_mean, const float* bn_alpha,
                              const float* interval_consts, float* z_out, float* sum_log_det, double* log_q, int64_t M_z,
                              int64_t M_p, int64_t N, int32_t D, int32_t num_stages, int32_t num_layers,
                              int32_t num_units, int64_t params_row_stride, int32_t fusion, void* workspace,
                              int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TNF_H */
