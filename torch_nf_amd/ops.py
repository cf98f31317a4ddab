"""Tensor-level wrappers around the C ABI (torch tensors in, torch tensors out).

Everything numerical happens in libtnf_hip.so; torch is used for device memory,
the current stream, and autograd bookkeeping.  Inputs that live on the host are
staged to the HIP device and results are handed back on the input's device, so
code written against the reference (which is CPU-only) keeps working -- but the
arithmetic always runs on the GPU; there is no CPU implementation here.
"""
import contextlib
import functools
import threading

import torch

from . import _lib
from ._lib import lib, check

_DTYPES = {torch.float32: _lib.F32, torch.float64: _lib.F64}


def _records_options(fn):
    """Function.forward: remember the calling thread's tnf_set_option values on the ctx ..."""
    @functools.wraps(fn)
    def forward(ctx, *args):
        ctx.tnf_options = _lib.options_snapshot()
        return fn(ctx, *args)
    return staticmethod(forward)


def _reenters_options(fn):
    """... Function.backward: run under them.  autograd calls backward on its own device thread, and every option of the
    library is thread-local: without this a kernel variant chosen around a training step (TNF_OPT_TRAIN_BWD_FP32,
    TNF_OPT_FORCE_GENERIC, the operand precision) would govern the forward half of the step only."""
    @functools.wraps(fn)
    def backward(ctx, *grads):
        with _lib.options_reentered(ctx.tnf_options):
            return fn(ctx, *grads)
    return staticmethod(backward)


def _dtype_code(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError("torch_nf_amd kernels take float32 or float64 tensors, not %s" % t.dtype)


def _stage(t, dev):
    """Contiguous copy/view of `t` on the compute device."""
    if t.device != dev:
        t = t.to(dev)
    return t.contiguous()


def _stats(t, dev):
    """BatchNorm statistics as contiguous float32 on the device (no-op when they already are)."""
    if t.device == dev and t.dtype == torch.float32 and t.is_contiguous() and not t.requires_grad:
        return t
    return _stage(t.detach().float(), dev)


def _rows(params, dev):
    """(M_p, P) parameter rows with unit inner stride -> (tensor, row_stride)."""
    if params.dim() != 2:
        raise ValueError("params must be (M, D_params), got shape %s" % (tuple(params.shape),))
    if params.device != dev:
        params = params.to(dev)
    if params.stride(1) != 1 or (params.shape[0] > 1 and params.stride(0) < params.shape[1]):
        params = params.contiguous()
    stride = params.stride(0) if params.shape[0] > 1 else max(params.stride(0), params.shape[1])
    return params, stride


def _bcast_M(Mz, Mp):
    if Mz != Mp and Mz != 1 and Mp != 1:
        raise RuntimeError("batch dimensions of z (%d) and params (%d) do not broadcast" % (Mz, Mp))
    return max(Mz, Mp)


def _check3(z):
    if z.dim() != 3:
        raise ValueError("z must be (M, N, D), got shape %s" % (tuple(z.shape),))


# ---------------------------------------------------------------------------
# RealNVP coupling layer
# ---------------------------------------------------------------------------
def coupling_raw(z, params, D, L, U, upper, inverse):
    """No-autograd call of tnf_coupling.  Returns (z_out (M,N,D), log_det (M,N))."""
    _check3(z)
    dev = _lib.require_device()
    home = z.device
    code = _dtype_code(z)
    if params.dtype != z.dtype:
        raise TypeError("z (%s) and params (%s) must have the same dtype" % (z.dtype, params.dtype))
    zc = _stage(z, dev)
    pc, pstride = _rows(params, dev)
    Mz, N = zc.shape[0], zc.shape[1]
    Mp = pc.shape[0]
    M = _bcast_M(Mz, Mp)
    if zc.shape[2] != D:
        raise ValueError("last dimension of z (%d) must equal D (%d)" % (zc.shape[2], D))
    z_out = torch.empty((M, N, D), dtype=z.dtype, device=dev)
    log_det = torch.empty((M, N), dtype=z.dtype, device=dev)
    if N == 0:
        return z_out.to(home), log_det.to(home)
    check(lib.tnf_coupling(code, zc.data_ptr(), pc.data_ptr(), z_out.data_ptr(), log_det.data_ptr(),
                           Mz, Mp, N, D, L, U, int(upper), int(inverse), pstride, _lib.LD_STORE,
                           _lib.stream_ptr()))
    if home != dev:
        z_out, log_det = z_out.to(home), log_det.to(home)
    return z_out, log_det


class _CouplingFn(torch.autograd.Function):
    @_records_options
    def forward(ctx, z, params, D, L, U, upper, inverse):
        z_out, log_det = coupling_raw(z, params, D, L, U, upper, inverse)
        ctx.save_for_backward(z, params, z_out)
        ctx.cfg = (D, L, U, upper, inverse)
        return z_out, log_det

    @_reenters_options
    def backward(ctx, g_z, g_ld):
        from . import grad  # backward kernels live behind the same C ABI

        z, params, z_out = ctx.saved_tensors
        gz, gp = grad.coupling_backward(z, params, z_out, g_z, g_ld, *ctx.cfg)
        return gz, gp, None, None, None, None, None


def coupling(z, params, D, L, U, upper, inverse):
    if torch.is_grad_enabled() and (z.requires_grad or params.requires_grad):
        return _CouplingFn.apply(z, params, D, L, U, upper, inverse)
    return coupling_raw(z, params, D, L, U, upper, inverse)


# ---------------------------------------------------------------------------
# Affine
# ---------------------------------------------------------------------------
def affine_raw(z, params, D, inverse):
    _check3(z)
    dev = _lib.require_device()
    home = z.device
    code = _dtype_code(z)
    if params.dtype != z.dtype:
        raise TypeError("z (%s) and params (%s) must have the same dtype" % (z.dtype, params.dtype))
    zc = _stage(z, dev)
    pc, pstride = _rows(params, dev)
    Mz, N = zc.shape[0], zc.shape[1]
    Mp = pc.shape[0]
    M = _bcast_M(Mz, Mp)
    if zc.shape[2] != D:
        raise ValueError("last dimension of z (%d) must equal D (%d)" % (zc.shape[2], D))
    z_out = torch.empty((M, N, D), dtype=z.dtype, device=dev)
    log_det = torch.empty((Mp, 1), dtype=z.dtype, device=dev)
    check(lib.tnf_affine(code, zc.data_ptr(), pc.data_ptr(), z_out.data_ptr(), log_det.data_ptr(),
                         Mz, Mp, N, D, int(inverse), pstride, _lib.stream_ptr()))
    if home != dev:
        z_out, log_det = z_out.to(home), log_det.to(home)
    return z_out, log_det


class _AffineFn(torch.autograd.Function):
    @_records_options
    def forward(ctx, z, params, D, inverse):
        z_out, log_det = affine_raw(z, params, D, inverse)
        ctx.save_for_backward(z, params, z_out)
        ctx.cfg = (D, inverse)
        return z_out, log_det

    @_reenters_options
    def backward(ctx, g_z, g_ld):
        from . import grad

        z, params, z_out = ctx.saved_tensors
        gz, gp = grad.affine_backward(z, params, z_out, g_z, g_ld, *ctx.cfg)
        return gz, gp, None, None


def affine(z, params, D, inverse):
    if torch.is_grad_enabled() and (z.requires_grad or params.requires_grad):
        return _AffineFn.apply(z, params, D, inverse)
    return affine_raw(z, params, D, inverse)


# ---------------------------------------------------------------------------
# BatchNorm
# ---------------------------------------------------------------------------
def bn_apply_raw(z, mean, alpha, inverse):
    _check3(z)
    dev = _lib.require_device()
    home = z.device
    code = _dtype_code(z)
    zc = _stage(z, dev)
    D = zc.shape[2]
    mean_c = _stage(mean.detach().float(), dev)
    alpha_c = _stage(alpha.detach().float(), dev)
    z_out = torch.empty_like(zc)
    log_det = torch.empty((), dtype=torch.float32, device=dev)
    check(lib.tnf_bn_apply(code, zc.data_ptr(), mean_c.data_ptr(), alpha_c.data_ptr(), z_out.data_ptr(),
                           log_det.data_ptr(), zc.shape[0] * zc.shape[1], D, int(inverse),
                           _lib.stream_ptr()))
    if home != dev:
        z_out, log_det = z_out.to(home), log_det.to(home)
    return z_out, log_det


class _BnApplyFn(torch.autograd.Function):
    """Cached-statistics BatchNorm.  Gradients flow to z and -- when the cached statistics still carry the graph of the
    batch-mode forward that produced them, as the reference's do (bijectors.py:414-415, no detach) -- to mean and alpha:
      inverse  out = z alpha + mean        g_alpha = sum g_out z      - g_ld / alpha,  g_mean = sum g_out
      forward  out = (z - mean) / alpha    g_alpha = -sum g_out out / alpha - g_ld / alpha,  g_mean = -sum g_out / alpha
    (log_det = -sum log alpha in both).  The per-feature sums are a handful of torch reductions on the device."""

    @_records_options
    def forward(ctx, z, mean, alpha, inverse):
        z_out, log_det = bn_apply_raw(z, mean, alpha, inverse)
        stats_grad = mean.requires_grad or alpha.requires_grad
        ctx.save_for_backward(alpha, *((z if inverse else z_out,) if stats_grad else ()))
        ctx.inverse = inverse
        ctx.stats_grad = stats_grad
        return z_out, log_det

    @_reenters_options
    def backward(ctx, g_z, g_ld):
        from . import grad

        alpha = ctx.saved_tensors[0]
        gz = grad.bn_apply_backward(g_z, alpha, ctx.inverse) if ctx.needs_input_grad[0] else None
        g_mean = g_alpha = None
        if ctx.stats_grad:
            v = ctx.saved_tensors[1]
            dev = v.device
            a = alpha.detach().to(dev).to(v.dtype)
            D = v.shape[-1]
            g_alpha = torch.zeros(D, dtype=v.dtype, device=dev)
            g_mean = torch.zeros(D, dtype=v.dtype, device=dev)
            if g_z is not None:
                go = g_z.to(dev).reshape(-1, D)
                if ctx.inverse:
                    g_alpha = (go * v.reshape(-1, D)).sum(0)
                    g_mean = go.sum(0)
                else:
                    g_alpha = -(go * v.reshape(-1, D)).sum(0) / a
                    g_mean = -go.sum(0) / a
            if g_ld is not None:
                g_alpha = g_alpha - g_ld.to(dev).to(v.dtype) / a
            g_alpha, g_mean = g_alpha.to(alpha.dtype).to(alpha.device), g_mean.to(alpha.dtype).to(alpha.device)
        return gz, g_mean, g_alpha, None


def bn_apply(z, mean, alpha, inverse):
    if torch.is_grad_enabled() and (z.requires_grad or mean.requires_grad or alpha.requires_grad):
        return _BnApplyFn.apply(z, mean, alpha, inverse)
    return bn_apply_raw(z, mean, alpha, inverse)


class _BnBatchFn(torch.autograd.Function):
    """Batch-statistics BatchNorm with gradients to z through the normalisation, the batch moments and the log-det --
    and through the RETURNED statistics: mean and alpha are differentiable outputs, so a later use of the cached
    statistics in the same graph (nf(N) then nf.log_prob(z), the reference's pattern: bijectors.py:414-415 caches them
    without detach) back-propagates into this batch:  mean = sum z / R,  alpha = sqrt(var_b + eps)  give
    g_z += g_mean / R + g_alpha z_norm / R  (z_norm = (z - mean) / alpha)."""

    @_records_options
    def forward(ctx, z, eps):
        z_norm, log_det, mean, alpha = _bn_batch_forward_raw(z, eps)
        ctx.save_for_backward(z_norm, alpha)
        return z_norm, log_det, mean, alpha

    @_reenters_options
    def backward(ctx, g_zn, g_ld, g_mean, g_alpha):
        from . import grad

        z_norm, alpha = ctx.saved_tensors
        gz = grad.bn_batch_backward(z_norm, alpha, g_zn, g_ld)
        if g_mean is not None or g_alpha is not None:
            rows = z_norm.numel() // z_norm.shape[-1]
            extra = 0.0
            if g_mean is not None:
                extra = g_mean.to(gz.device).to(gz.dtype) / rows
            if g_alpha is not None:
                extra = extra + z_norm.to(gz.device) * (g_alpha.to(gz.device).to(gz.dtype) / rows)
            gz = gz + extra
        return gz, None


def bn_batch_forward(z, eps, reduce=None):
    """Batch-statistics BatchNorm forward (float32).  Returns (z_norm, log_det, mean, alpha).
    reduce: the exchange step of a SAMPLE-SHARDED batch (distributed.moment_reducer(group): sums a small tensor in
    place over the ranks that hold the other rows) -- statistics and their gradients are then those of the whole batch."""
    if reduce is not None:
        return bn_batch_forward_sharded(z, eps, reduce)
    if torch.is_grad_enabled() and z.requires_grad:
        return _BnBatchFn.apply(z, eps)
    return _bn_batch_forward_raw(z, eps)


class HipBnShardKernels(object):
    """The four halves of the sample-sharded batch-statistics BatchNorm on the HIP device (tnf_bn_batch_moments_f32,
    _normalize_f32, _backward_sums_f32, _backward_apply_f32).  bn_batch_forward_sharded takes the kernels as an object so
    that the CPU tests can drive the same exchange logic with stand-ins (tests/test_distributed_gloo.py)."""

    @staticmethod
    def moments(z):
        D = z.shape[-1]
        mom = torch.empty(2 * D + 1, dtype=torch.float64, device=z.device)
        check(lib.tnf_bn_batch_moments_f32(z.data_ptr(), mom.data_ptr(), z.numel() // D, D, _lib.stream_ptr()))
        return mom

    @staticmethod
    def normalize(z, moments, eps):
        D = z.shape[-1]
        dev = z.device
        z_out = torch.empty_like(z)
        mean = torch.empty(D, dtype=torch.float32, device=dev)
        alpha = torch.empty(D, dtype=torch.float32, device=dev)
        log_det = torch.empty((), dtype=torch.float32, device=dev)
        ws = torch.empty(D, dtype=torch.float32, device=dev)
        check(lib.tnf_bn_batch_normalize_f32(z.data_ptr(), moments.data_ptr(), z_out.data_ptr(), mean.data_ptr(),
                                             alpha.data_ptr(), log_det.data_ptr(), z.numel() // D, D, float(eps),
                                             ws.data_ptr(), 4 * D, _lib.stream_ptr()))
        return z_out, log_det, mean, alpha

    @staticmethod
    def backward_sums(z_norm, g):
        D = z_norm.shape[-1]
        sums = torch.empty(2 * D, dtype=torch.float64, device=z_norm.device)
        check(lib.tnf_bn_batch_backward_sums_f32(z_norm.data_ptr(), g.data_ptr(), sums.data_ptr(), z_norm.numel() // D, D,
                                                 _lib.stream_ptr()))
        return sums

    @staticmethod
    def backward_apply(z_norm, g, g_ld, alpha, sums, count):
        D = z_norm.shape[-1]
        out = torch.empty_like(z_norm)
        check(lib.tnf_bn_batch_backward_apply_f32(z_norm.data_ptr(), g.data_ptr(), None if g_ld is None else g_ld.data_ptr(),
                                                  alpha.data_ptr(), sums.data_ptr(), count.data_ptr(), out.data_ptr(),
                                                  z_norm.numel() // D, D, _lib.stream_ptr()))
        return out


class _BnBatchShardedFn(torch.autograd.Function):
    """_BnBatchFn for a batch whose rows are spread over ranks ("an all-reduce of [sum z, sum z^2] per BN
    layer ... required for parity with the single-device answer", here WITH gradients -- the reference differentiates
    through the batch moments, bijectors.py:401-415):
      forward   local moments -> reduce -> statistics of the whole batch -> this rank's rows normalised
      backward  local [sum g | sum g x^] -> reduce -> g_z of this rank's rows with the global sums and row count;
                gradients arriving at the RETURNED statistics (a later use of the cached mean / alpha in the same graph)
                are summed over the ranks too, since every rank holds a replica of them.
    Every rank must take part in every reduction, also with an empty shard: the collectives are matched by call order."""

    @staticmethod
    def forward(ctx, z, eps, reduce, kernels):
        mom = reduce(kernels.moments(z))
        z_norm, log_det, mean, alpha = kernels.normalize(z, mom, eps)
        ctx.set_materialize_grads(False)  # unused outputs arrive as None: no exchange for statistics nobody used again
        ctx.save_for_backward(z_norm, alpha, mom)
        ctx.reduce, ctx.kernels = reduce, kernels
        return z_norm, log_det, mean, alpha

    @staticmethod
    def backward(ctx, g_zn, g_ld, g_mean, g_alpha):
        z_norm, alpha, mom = ctx.saved_tensors
        reduce, kernels = ctx.reduce, ctx.kernels
        D = z_norm.shape[-1]
        g = torch.zeros_like(z_norm) if g_zn is None else g_zn.contiguous().to(z_norm.dtype)
        sums = kernels.backward_sums(z_norm, g)
        if g_ld is not None:  # log_det is replicated like the statistics: its gradient rides in the same reduction
            sums[D:] += g_ld.to(sums.device).double()  # (it meets the sum of g x^ in the formula, tnf.h)
        sums = reduce(sums)
        gz = kernels.backward_apply(z_norm, g, None, alpha, sums, mom[2 * D:])
        # replicas of mean / alpha: their gradient is the sum over the ranks' later uses.  (Structural: present on every
        # rank or on none -- the ranks run the same graph.)
        if g_mean is not None or g_alpha is not None:
            gs = torch.zeros(2 * D, dtype=torch.float64, device=z_norm.device)
            if g_mean is not None:
                gs[:D] = g_mean.to(gs.device).double()
            if g_alpha is not None:
                gs[D:] = g_alpha.to(gs.device).double()
            gs = reduce(gs) / mom[2 * D]
            gz = gz + (gs[:D] + z_norm * gs[D:]).to(gz.dtype)
        return gz, None, None, None


def bn_batch_forward_sharded(z, eps, reduce, kernels=None):
    """The sample-sharded batch-statistics forward (see _BnBatchShardedFn).  z: this rank's rows (M, N_local, D) float32."""
    _check3(z)
    if z.dtype != torch.float32:
        raise TypeError("BatchNorm batch statistics are implemented for float32 (got %s)" % z.dtype)
    home = z.device
    if kernels is None:
        kernels = HipBnShardKernels
        dev = _lib.require_device()
        zc = z if z.device == dev and z.is_contiguous() else z.to(dev).contiguous()
    else:
        zc = z.contiguous()
    if torch.is_grad_enabled() and zc.requires_grad:
        out = _BnBatchShardedFn.apply(zc, eps, reduce, kernels)
    else:
        out = kernels.normalize(zc, reduce(kernels.moments(zc)), eps)
    if home != zc.device:
        out = tuple(t.to(home) for t in out)
    return out


def _bn_batch_forward_raw(z, eps):
    _check3(z)
    if z.dtype != torch.float32:
        raise TypeError("BatchNorm batch statistics are implemented for float32 (got %s)" % z.dtype)
    dev = _lib.require_device()
    home = z.device
    zc = _stage(z.detach(), dev)
    D = zc.shape[2]
    rows = zc.shape[0] * zc.shape[1]
    if rows < 2:
        raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                         % (torch.Size([rows, D]),))
    z_out = torch.empty_like(zc)
    mean = torch.empty(D, dtype=torch.float32, device=dev)
    alpha = torch.empty(D, dtype=torch.float32, device=dev)
    log_det = torch.empty((), dtype=torch.float32, device=dev)
    ws_bytes = lib.tnf_bn_batch_workspace_bytes(D)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.tnf_bn_batch_forward_f32(zc.data_ptr(), z_out.data_ptr(), mean.data_ptr(), alpha.data_ptr(),
                                       log_det.data_ptr(), rows, D, float(eps), ws.data_ptr(), ws_bytes,
                                       _lib.stream_ptr()))
    if home != dev:
        z_out, log_det, mean, alpha = z_out.to(home), log_det.to(home), mean.to(home), alpha.to(home)
    return z_out, log_det, mean, alpha


# ---------------------------------------------------------------------------
# base density (float64, like the reference's numpy expression)
# ---------------------------------------------------------------------------
def base_log_density_f64(omega):
    """omega (M,N,D) float64 or float32 tensor -> (M,N) float64 log N(omega; 0, I)."""
    dev = _lib.require_device()
    home = omega.device
    oc = _stage(omega, dev)
    code = _dtype_code(oc)
    M, N, D = oc.shape
    out = torch.empty((M, N), dtype=torch.float64, device=dev)
    check(lib.tnf_base_log_density_f64(code, oc.data_ptr(), out.data_ptr(), M * N, D, _lib.stream_ptr()))
    return out if home == dev else out.to(home)


# ---------------------------------------------------------------------------
# flow level
# ---------------------------------------------------------------------------
_PREC_CODES = {"fp32": 0, "bf16": 1}
_prec_state = threading.local()


def current_operand_precision():
    return getattr(_prec_state, "prec", "fp32")


@contextlib.contextmanager
def operand_precision(prec):
    """The fp32-vs-bf16 experiment (BASELINE.json configs[4]): inside `operand_precision("bf16")` the RealNVP log_prob
    kernels (flow_log_prob_raw and the forward leg of flow_log_prob_train) and the autoregressive-flow kernels (forward,
    inverse and backward) round every conditioner operand to bf16.  Per thread, like every tnf_set_option key; autograd
    runs backward on its own thread, so the Functions below record the precision their forward ran in and re-enter it.
    Not a parity path: tools/bf16_sweep.py reports its error and its speed."""
    code = _PREC_CODES[prec]
    before = current_operand_precision()
    check(lib.tnf_set_option(_lib.OPT_OPERAND_PREC, code))
    _prec_state.prec = prec
    try:
        yield
    finally:
        check(lib.tnf_set_option(_lib.OPT_OPERAND_PREC, _PREC_CODES[before]))
        _prec_state.prec = before


def has_fast_path(D, L, U):
    return bool(lib.tnf_has_fast_path(D, L, U))


def resolve_fusion(D, S, L, U, fusion):
    """AUTO -> the whole-flow kernel when it exists for the shape, else one kernel per layer."""
    if fusion == _lib.FUSE_AUTO:
        return _lib.FUSE_FLOW if lib.tnf_flow_fused_supported(D, S, L, U) else _lib.FUSE_LAYER
    return fusion


_ws_cache = {}


def _workspace(nbytes, dev):
    """Grow-only scratch buffer per (device, stream); the kernels of one call are
    stream-ordered, so reuse on the same stream is safe."""
    key = (dev.index, _lib.stream_ptr())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _ws_cache[key] = buf
    return buf


def flow_log_prob_raw(z, params, bn_mean, bn_alpha, D, S, L, U, fusion=_lib.FUSE_AUTO,
                      want_z0=False, want_sld=False, want_lp=True, interval_consts=None, count_reruns=False):
    """tnf_flow_log_prob_f32.  Returns (log_prob | None, z0 | None, sum_log_det | None).
    interval_consts: (7, D) device constants of a ToInterval support layer fused into the whole-flow kernel.
    count_reruns: also return a one-element int32 device tensor = the number of 32-sample groups the whole-flow kernel
    re-ran with exact fp32 first-layer contractions (inputs outside the split-f16 operand range; tnf_flow_log_prob_diag_f32)."""
    _check3(z)
    dev = _lib.require_device()
    home = z.device
    if z.dtype != torch.float32 or params.dtype != torch.float32:
        raise TypeError("the fused flow kernels are float32")
    zc = _stage(z, dev)
    pc, pstride = _rows(params, dev)
    mean_c = _stats(bn_mean, dev)
    alpha_c = _stats(bn_alpha, dev)
    Mz, N = zc.shape[0], zc.shape[1]
    Mp = pc.shape[0]
    M = _bcast_M(Mz, Mp)
    fusion = resolve_fusion(D, S, L, U, fusion)
    lp = torch.empty((M, N), dtype=torch.float32, device=dev) if want_lp else None
    z0 = torch.empty((M, N, D), dtype=torch.float32, device=dev) if want_z0 else None
    sld = torch.empty((M, N), dtype=torch.float32, device=dev) if want_sld else None
    if N == 0:
        return tuple(t.to(home) if t is not None else None for t in (lp, z0, sld))
    ws_bytes = check(lib.tnf_flow_workspace_bytes(M, N, D, S, L, U, fusion))
    ws = _workspace(ws_bytes, dev)
    args = (zc.data_ptr(), pc.data_ptr(), mean_c.data_ptr(), alpha_c.data_ptr(),
            None if interval_consts is None else interval_consts.data_ptr(),
            lp.data_ptr() if want_lp else None, z0.data_ptr() if want_z0 else None,
            sld.data_ptr() if want_sld else None, Mz, Mp, N, D, S, L, U, pstride, fusion, ws.data_ptr(),
            ws.numel(), _lib.stream_ptr())
    reruns = None
    if count_reruns:
        reruns = torch.zeros(1, dtype=torch.int32, device=dev)
        check(lib.tnf_flow_log_prob_diag_f32(*args, reruns.data_ptr()))
    else:
        check(lib.tnf_flow_log_prob_f32(*args))
    if home != dev:
        lp = lp.to(home) if want_lp else None
        z0 = z0.to(home) if want_z0 else None
        sld = sld.to(home) if want_sld else None
    return (lp, z0, sld, reruns) if count_reruns else (lp, z0, sld)


def flow_forward_raw(omega, params, bn_mean, bn_alpha, D, S, L, U, fusion=_lib.FUSE_AUTO, interval_consts=None,
                     want_log_q=False):
    """tnf_flow_forward_f32 (frozen BatchNorm).  Returns (z (M,N,D), sum_log_det (M,N)).
    want_log_q: a third value, log_q (M,N) float64 = log N(omega; 0, I) - sum_log_det written by the whole-flow kernel
    itself (tnf_flow_forward_logq_f32), or None when the call does not run on that kernel."""
    _check3(omega)
    dev = _lib.require_device()
    home = omega.device
    if omega.dtype != torch.float32 or params.dtype != torch.float32:
        raise TypeError("the fused flow kernels are float32")
    oc = _stage(omega, dev)
    pc, pstride = _rows(params, dev)
    mean_c = _stats(bn_mean, dev)
    alpha_c = _stats(bn_alpha, dev)
    Mz, N = oc.shape[0], oc.shape[1]
    Mp = pc.shape[0]
    M = _bcast_M(Mz, Mp)
    fusion = resolve_fusion(D, S, L, U, fusion)
    z_out = torch.empty((M, N, D), dtype=torch.float32, device=dev)
    sld = torch.empty((M, N), dtype=torch.float32, device=dev)
    if N == 0:
        out = (z_out.to(home), sld.to(home))
        return out + (torch.empty((M, N), dtype=torch.float64, device=home),) if want_log_q else out
    ws_bytes = check(lib.tnf_flow_workspace_bytes(M, N, D, S, L, U, fusion))
    ws = _workspace(ws_bytes, dev)
    log_q = None
    if want_log_q and fusion == _lib.FUSE_FLOW:
        log_q = torch.empty((M, N), dtype=torch.float64, device=dev)
        rc = lib.tnf_flow_forward_logq_f32(oc.data_ptr(), pc.data_ptr(), mean_c.data_ptr(), alpha_c.data_ptr(),
                                           None if interval_consts is None else interval_consts.data_ptr(),
                                           z_out.data_ptr(), sld.data_ptr(), log_q.data_ptr(), Mz, Mp, N, D, S, L, U,
                                           pstride, fusion, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        if rc == _lib.EUNSUPPORTED:
            log_q = None  # a selectable variant without that output: the plain entry below, the caller adds the density
        else:
            check(rc)
    if log_q is None:
        check(lib.tnf_flow_forward_f32(oc.data_ptr(), pc.data_ptr(), mean_c.data_ptr(), alpha_c.data_ptr(),
                                       None if interval_consts is None else interval_consts.data_ptr(),
                                       z_out.data_ptr(), sld.data_ptr(), Mz, Mp, N, D, S, L, U, pstride,
                                       fusion, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    if home != dev:
        z_out, sld = z_out.to(home), sld.to(home)
        log_q = None if log_q is None else log_q.to(home)
    return (z_out, sld, log_q) if want_log_q else (z_out, sld)


# ---------------------------------------------------------------------------
# flow level, with autograd: log_prob through one fused kernel per layer, backward through one
# MFMA backward kernel per layer (tnf_flow_log_prob_fwd_f32 / _bwd_f32)
# ---------------------------------------------------------------------------
def flow_forward_batch_raw(omega, params, D, S, L, U, eps, reduce_moments=None):
    """tnf_flow_forward_batch_f32: NormFlow.forward with batch-statistics BatchNorm and no autograd in one call
    -> (z, sum_log_det, bn_mean (2S, D), bn_alpha (2S, D)), all on the compute device.

    reduce_moments: optional callable applied IN PLACE to the (2D + 1,) float64 device tensor [sum | sum of squares |
    row count] of every BatchNorm layer's input before its statistics are formed -- for a sample-sharded forward this
    is the all-reduce over the ranks that share the batch (torch_nf_amd.distributed.moment_reducer), after which every
    rank normalises with the statistics of the GLOBAL batch exactly as the single-device call would
    (bijectors.py:401-415).  The chain then runs in steps (tnf_flow_forward_batch_begin / _layer / _fold / _end_f32)
    with the collective on the same stream between a layer and its fold."""
    dev = _lib.require_device()
    oc = _stage(omega.detach(), dev)
    pc, pstride = _rows(params.detach(), dev)
    if reduce_moments is not None:
        return run_batch_steps(FlowForwardBatchSteps(oc, pc, D, S, L, U, eps), reduce_moments)
    M, N = oc.shape[0], oc.shape[1]
    Mp = pc.shape[0]
    z = torch.empty_like(oc)
    sld = torch.empty((M, N), dtype=torch.float32, device=dev)
    mean = torch.empty((2 * S, D), dtype=torch.float32, device=dev)
    alpha = torch.empty((2 * S, D), dtype=torch.float32, device=dev)
    nbytes = check(lib.tnf_flow_forward_batch_workspace_bytes(Mp, D, S, L))
    ws = _workspace(nbytes, dev)
    check(lib.tnf_flow_forward_batch_f32(oc.data_ptr(), pc.data_ptr(), z.data_ptr(), sld.data_ptr(), mean.data_ptr(),
                                         alpha.data_ptr(), M, Mp, N, D, S, L, U, pstride, float(eps), ws.data_ptr(),
                                         nbytes, _lib.stream_ptr()))
    return z, sld, mean, alpha


class FlowForwardBatchSteps:
    """One rank's share of a batch-statistics forward, step by step (tnf_flow_forward_batch_begin / _layer / _fold /
    _end_f32): `layer(c)` returns this rank's moments of layer c's output, `fold(c)` consumes the moments AS THEY ARE
    THEN (summed over the ranks by the caller) -- see run_batch_steps."""

    def __init__(self, omega, params, D, S, L, U, eps):
        dev = _lib.require_device()
        self.omega = _stage(omega.detach(), dev)
        self.params, self.pstride = _rows(params.detach(), dev)
        self.cfg = (D, S, L, U)
        self.eps = float(eps)
        self.n_layers = 2 * S
        M, N = self.omega.shape[0], self.omega.shape[1]
        self.M, self.N, self.Mp = M, N, self.params.shape[0]
        self.z = torch.empty_like(self.omega)
        self.sld = torch.empty((M, N), dtype=torch.float32, device=dev)
        self.mean = torch.empty((2 * S, D), dtype=torch.float32, device=dev)
        self.alpha = torch.empty((2 * S, D), dtype=torch.float32, device=dev)
        self.nbytes = check(lib.tnf_flow_forward_batch_workspace_bytes(self.Mp, D, S, L))
        # a workspace of its own: it must survive the collectives between the steps
        self.ws = torch.empty(self.nbytes, dtype=torch.uint8, device=dev)
        self.moments = torch.empty(2 * D + 1, dtype=torch.float64, device=dev)

    def begin(self):
        D, S, L, U = self.cfg
        check(lib.tnf_flow_forward_batch_begin_f32(self.params.data_ptr(), self.Mp, D, S, L, U, self.pstride,
                                                   self.ws.data_ptr(), self.nbytes, _lib.stream_ptr()))

    def layer(self, c):
        D, S, L, U = self.cfg
        src = self.omega if c == 0 else self.z
        check(lib.tnf_flow_forward_batch_layer_f32(c, src.data_ptr(), self.params.data_ptr(), self.z.data_ptr(),
                                                   self.sld.data_ptr(), self.moments.data_ptr(), self.M, self.Mp, self.N,
                                                   D, S, L, U, self.pstride, self.ws.data_ptr(), self.nbytes,
                                                   _lib.stream_ptr()))
        return self.moments

    def fold(self, c):
        D, S, L, U = self.cfg
        check(lib.tnf_flow_forward_batch_fold_f32(c, self.params.data_ptr(), self.moments.data_ptr(), self.mean.data_ptr(),
                                                  self.alpha.data_ptr(), self.Mp, D, S, L, U, self.pstride, self.eps,
                                                  self.ws.data_ptr(), self.nbytes, _lib.stream_ptr()))

    def end(self):
        D, S, L, U = self.cfg
        check(lib.tnf_flow_forward_batch_end_f32(self.z.data_ptr(), self.sld.data_ptr(), self.M, self.Mp, self.N, D, S, L,
                                                 self.ws.data_ptr(), self.nbytes, _lib.stream_ptr()))
        return self.z, self.sld, self.mean, self.alpha


def run_batch_steps(steps, reduce_moments=None):
    """The sequence of a (sample-sharded) batch-statistics forward over any object with the FlowForwardBatchSteps
    protocol: per coupling layer, the rank-local kernel + moments, then the exchange (`reduce_moments`: in-place sum
    over the ranks that share the batch; None on a single rank), then the fold built from the global moments."""
    steps.begin()
    for c in range(steps.n_layers):
        moments = steps.layer(c)
        if reduce_moments is not None:
            reduce_moments(moments)
        steps.fold(c)
    return steps.end()


class _FlowForwardTrainFn(torch.autograd.Function):
    """NormFlow.forward with fresh batch statistics under autograd (tnf_flow_forward_train_fwd_f32 / _bwd_f32): one
    node for the whole stack.  Returns (z, sum_log_det, bn_mean, bn_alpha); the statistics are not differentiable
    outputs (the BatchNorm layers cache them detached, as everywhere in this package)."""

    @_records_options
    def forward(ctx, omega, params, D, S, L, U, eps):
        dev = _lib.require_device()
        oc = _stage(omega.detach(), dev)
        pc, pstride = _rows(params.detach(), dev)
        M, N = oc.shape[0], oc.shape[1]
        Mp = pc.shape[0]
        z = torch.empty_like(oc)
        sld = torch.empty((M, N), dtype=torch.float32, device=dev)
        states = torch.empty((2 * S, M, N, D), dtype=torch.float32, device=dev)
        folds = torch.empty((2 * S, Mp, 2, D), dtype=torch.float32, device=dev)
        mean = torch.empty((2 * S, D), dtype=torch.float32, device=dev)
        alpha = torch.empty((2 * S, D), dtype=torch.float32, device=dev)
        nbytes = check(lib.tnf_flow_forward_train_workspace_bytes(M, Mp, N, D, S, L))
        ws = _workspace(nbytes, dev)
        check(lib.tnf_flow_forward_train_fwd_f32(oc.data_ptr(), pc.data_ptr(), z.data_ptr(), sld.data_ptr(),
                                                 states.data_ptr(), folds.data_ptr(), mean.data_ptr(), alpha.data_ptr(),
                                                 M, Mp, N, D, S, L, U, pstride, float(eps), ws.data_ptr(), nbytes,
                                                 _lib.stream_ptr()))
        ctx.save_for_backward(oc, pc, states, folds, mean, alpha)
        ctx.cfg = (D, S, L, U, pstride, omega.device, params.device, tuple(params.shape))
        ctx.mark_non_differentiable(mean, alpha)
        return z, sld, mean, alpha

    @_reenters_options
    def backward(ctx, g_z, g_sld, _gm, _ga):
        oc, pc, states, folds, mean, alpha = ctx.saved_tensors
        D, S, L, U, pstride, o_home, p_home, p_shape = ctx.cfg
        dev = oc.device
        M, N = oc.shape[0], oc.shape[1]
        Mp = pc.shape[0]
        gz = _stage(_grad_or_zeros(g_z, oc.shape, torch.float32, dev).float(), dev)
        gs = _stage(_grad_or_zeros(g_sld, (M, N), torch.float32, dev).float(), dev)
        go = torch.empty_like(oc) if ctx.needs_input_grad[0] else None
        gp = torch.zeros(p_shape, dtype=torch.float32, device=dev)
        nbytes = check(lib.tnf_flow_forward_train_workspace_bytes(M, Mp, N, D, S, L))
        ws = _workspace(nbytes, dev)
        check(lib.tnf_flow_forward_train_bwd_f32(oc.data_ptr(), pc.data_ptr(), states.data_ptr(), folds.data_ptr(),
                                                 mean.data_ptr(), alpha.data_ptr(), gz.data_ptr(), gs.data_ptr(),
                                                 go.data_ptr() if go is not None else None, gp.data_ptr(), M, Mp, N, D, S,
                                                 L, U, pstride, gp.shape[1], ws.data_ptr(), nbytes, _lib.stream_ptr()))
        if go is not None and o_home != dev:
            go = go.to(o_home)
        return go, (gp if p_home == dev else gp.to(p_home)), None, None, None, None, None


def flow_forward_train(omega, params, D, S, L, U, eps):
    return _FlowForwardTrainFn.apply(omega, params, D, S, L, U, eps)


def flow_train_supported(M, Mp, N, D, S, L, U):
    return Mp in (1, M) and N >= 32 and lib.tnf_flow_train_workspace_bytes(M, Mp, max(N, 1), D, S, L, U) >= 0


class _FlowLogProbFn(torch.autograd.Function):
    @_records_options
    def forward(ctx, z, params, bn_mean, bn_alpha, D, S, L, U):
        dev = _lib.require_device()
        zc = _stage(z.detach(), dev)
        pc, pstride = _rows(params.detach(), dev)
        mean_c, alpha_c = _stats(bn_mean, dev), _stats(bn_alpha, dev)
        M, N = zc.shape[0], zc.shape[1]
        Mp = pc.shape[0]
        lp = torch.empty((M, N), dtype=torch.float32, device=dev)
        states = torch.empty((2 * S - 1, M, N, D), dtype=torch.float32, device=dev)
        ws_bytes = check(lib.tnf_flow_train_workspace_bytes(M, Mp, N, D, S, L, U))
        ws = _workspace(ws_bytes, dev)
        check(lib.tnf_flow_log_prob_fwd_f32(zc.data_ptr(), pc.data_ptr(), mean_c.data_ptr(), alpha_c.data_ptr(),
                                            lp.data_ptr(), states.data_ptr(), M, Mp, N, D, S, L, U, pstride,
                                            ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        ctx.save_for_backward(zc, pc, mean_c, alpha_c, states)
        ctx.cfg = (D, S, L, U, pstride, z.device, params.device, tuple(params.shape))
        return lp if z.device == dev else lp.to(z.device)

    @_reenters_options
    def backward(ctx, g_lp):
        zc, pc, mean_c, alpha_c, states = ctx.saved_tensors
        D, S, L, U, pstride, z_home, p_home, p_shape = ctx.cfg
        dev = zc.device
        M, N = zc.shape[0], zc.shape[1]
        Mp = pc.shape[0]
        g = _stage(g_lp.float(), dev)
        gz = torch.empty_like(zc)
        gp = torch.zeros(p_shape, dtype=torch.float32, device=dev)
        ws_bytes = check(lib.tnf_flow_train_workspace_bytes(M, Mp, N, D, S, L, U))
        ws = _workspace(ws_bytes, dev)
        check(lib.tnf_flow_log_prob_bwd_f32(zc.data_ptr(), states.data_ptr(), pc.data_ptr(), mean_c.data_ptr(),
                                            alpha_c.data_ptr(), g.data_ptr(), gz.data_ptr(), gp.data_ptr(), M, Mp,
                                            N, D, S, L, U, pstride, gp.shape[1], ws.data_ptr(), ws.numel(),
                                            _lib.stream_ptr()))
        return (gz if z_home == dev else gz.to(z_home)), (gp if p_home == dev else gp.to(p_home)), None, None, \
            None, None, None, None


def flow_train_rev_supported(M, Mp, N, D, S, L, U):
    """The reversible pair: whole-flow forward that keeps only z0, one-kernel backward."""
    # per-context parameter rows with a handful of samples each (the SNPE layout) are not this kernel's case:
    # every context would build ~170 KB of operand images for one tile (conditional_density_estimator fuses those)
    return Mp in (1, M) and N >= 1 and (Mp == 1 or N >= 32) and lib.tnf_flow_train_rev_supported(D, S, L, U) == 1


class _FlowLogProbRevFn(torch.autograd.Function):
    """log_prob through the whole-flow kernel (tnf_flow_log_prob_fwd_rev_f32); the backward rebuilds every
    layer's input from z0 inside one kernel (tnf_flow_log_prob_bwd_rev_f32), so no activations are kept.

    The backward's gradient accumulators are 32-bit fixed point and its cross-workgroup sum runs in block order (what
    makes the parameter gradient reproducible bit for bit).  A gradient term beyond their budget -- a heavy-tailed
    sample, whose deltas also approach the f16 range of the split operands -- is flagged by the kernel (its own result
    is NaN then, never a wrapped sum), and the step is recomputed through the per-layer pair with fp32 layer kernels
    (tnf_flow_log_prob_fwd_f32 / _bwd_f32, no such budget).  `overflow_recovery` says how:
      "device" (default)  the recomputation is enqueued behind the backward unconditionally, gated on the flag ON THE
                          DEVICE (tnf_set_launch_gate): its layer kernels return at once while the flag is clear, a
                          gated copy moves its gradients over the poisoned ones when it is set.  No host round trip, so
                          it also works inside a HIP-graph capture; costs ~20 empty launches and the fallback's buffers.
      "host"              the flag is read back (one synchronisation per backward) and the recomputation launched only
                          when needed; under a capture the flag cannot be read and the NaN poison stands.
      "off"               no recovery: an overflowing step yields a NaN gradient.
    `check_overflow = False` is the old spelling of "off"."""

    check_overflow = True
    overflow_recovery = "device"
    overflow_fallbacks = 0  # "host" mode: how often the fp32 pair had to take over (diagnostic)
    last_overflow_flag = None  # "device" mode: the flag tensor of the latest backward (diagnostic, read it after a sync)

    @_records_options
    def forward(ctx, z, params, bn_mean, bn_alpha, D, S, L, U):
        dev = _lib.require_device()
        zc = _stage(z.detach(), dev)
        pc, pstride = _rows(params.detach(), dev)
        mean_c, alpha_c = _stats(bn_mean, dev), _stats(bn_alpha, dev)
        M, N = zc.shape[0], zc.shape[1]
        Mp = pc.shape[0]
        lp = torch.empty((M, N), dtype=torch.float32, device=dev)
        z0 = torch.empty_like(zc)
        check(lib.tnf_flow_log_prob_fwd_rev_f32(zc.data_ptr(), pc.data_ptr(), mean_c.data_ptr(), alpha_c.data_ptr(),
                                                lp.data_ptr(), z0.data_ptr(), M, Mp, N, D, S, L, U, pstride,
                                                _lib.stream_ptr()))
        ctx.save_for_backward(z0, pc, mean_c, alpha_c, zc)  # zc: the caller's tensor (no copy), for the fallback only
        ctx.cfg = (D, S, L, U, pstride, z.device, params.device, tuple(params.shape))
        return lp if z.device == dev else lp.to(z.device)

    @_reenters_options
    def backward(ctx, g_lp):
        z0, pc, mean_c, alpha_c, zc = ctx.saved_tensors
        D, S, L, U, pstride, z_home, p_home, p_shape = ctx.cfg
        dev = z0.device
        M, N = z0.shape[0], z0.shape[1]
        Mp = pc.shape[0]
        g = _stage(g_lp.float(), dev)
        gz = torch.empty_like(z0) if ctx.needs_input_grad[0] else None
        gp = torch.zeros(p_shape, dtype=torch.float32, device=dev)
        ws_bytes = check(lib.tnf_flow_train_rev_workspace_bytes(M, Mp, N, D, S, L, U))
        ws = _workspace(ws_bytes, dev)
        mode = _FlowLogProbRevFn.overflow_recovery if _FlowLogProbRevFn.check_overflow else "off"
        if mode != "off" and not flow_train_supported(M, Mp, N, D, S, L, U):
            mode = "off"  # no per-layer pair for this shape: the poison is all there is
        if mode == "host" and torch.cuda.is_current_stream_capturing():
            mode = "off"
        flag = torch.zeros(1, dtype=torch.int32, device=dev) if mode != "off" else None
        check(lib.tnf_flow_log_prob_bwd_rev_f32(z0.data_ptr(), pc.data_ptr(), mean_c.data_ptr(), alpha_c.data_ptr(),
                                                g.data_ptr(), gz.data_ptr() if gz is not None else None,
                                                gp.data_ptr(), M, Mp, N, D, S, L, U, pstride, gp.shape[1],
                                                ws.data_ptr(), ws.numel(), None if flag is None else flag.data_ptr(),
                                                _lib.stream_ptr()))
        if mode == "host" and int(flag.item()) != 0:
            # a term left the fixed-point budget: the same step through the per-layer pair, fp32 layer kernels
            _FlowLogProbRevFn.overflow_fallbacks += 1
            gz, gp = _flow_log_prob_grad_fp32(zc, pc, pstride, mean_c, alpha_c, g, D, S, L, U, p_shape,
                                              ctx.needs_input_grad[0])
        elif mode == "device":
            # the same recomputation, enqueued now and decided on the device: every layer kernel of the pair exits at
            # once while the flag is clear; when it is set, gated copies put its gradients in place of the poison
            _FlowLogProbRevFn.last_overflow_flag = flag
            check(lib.tnf_set_launch_gate(flag.data_ptr()))
            try:
                gz2, gp2 = _flow_log_prob_grad_fp32(zc, pc, pstride, mean_c, alpha_c, g, D, S, L, U, p_shape,
                                                    ctx.needs_input_grad[0])
            finally:
                check(lib.tnf_set_launch_gate(None))
            check(lib.tnf_gated_copy_f32(flag.data_ptr(), gp.data_ptr(), gp2.data_ptr(), gp.numel(), _lib.stream_ptr()))
            if gz is not None:
                check(lib.tnf_gated_copy_f32(flag.data_ptr(), gz.data_ptr(), gz2.data_ptr(), gz.numel(), _lib.stream_ptr()))
        if gz is not None and z_home != dev:
            gz = gz.to(z_home)
        return gz, (gp if p_home == dev else gp.to(p_home)), None, None, None, None, None, None


def _flow_log_prob_grad_fp32(zc, pc, pstride, mean_c, alpha_c, g, D, S, L, U, p_shape, want_gz):
    """Gradient of log_prob w.r.t. (z, params) through the per-layer training pair with the fp32-MFMA layer backward."""
    dev = zc.device
    M, N = zc.shape[0], zc.shape[1]
    Mp = pc.shape[0]
    lp = torch.empty((M, N), dtype=torch.float32, device=dev)
    states = torch.empty((2 * S - 1, M, N, D), dtype=torch.float32, device=dev)
    ws_bytes = check(lib.tnf_flow_train_workspace_bytes(M, Mp, N, D, S, L, U))
    ws = _workspace(ws_bytes, dev)
    check(lib.tnf_flow_log_prob_fwd_f32(zc.data_ptr(), pc.data_ptr(), mean_c.data_ptr(), alpha_c.data_ptr(), lp.data_ptr(),
                                        states.data_ptr(), M, Mp, N, D, S, L, U, pstride, ws.data_ptr(), ws.numel(),
                                        _lib.stream_ptr()))
    gz = torch.empty_like(zc)
    gp = torch.zeros(p_shape, dtype=torch.float32, device=dev)
    before = _lib.options_snapshot()
    check(lib.tnf_set_option(_lib.OPT_TRAIN_BWD_FP32, 1))
    try:
        check(lib.tnf_flow_log_prob_bwd_f32(zc.data_ptr(), states.data_ptr(), pc.data_ptr(), mean_c.data_ptr(),
                                            alpha_c.data_ptr(), g.data_ptr(), gz.data_ptr(), gp.data_ptr(), M, Mp, N, D, S,
                                            L, U, pstride, gp.shape[1], ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    finally:
        check(lib.tnf_set_option(_lib.OPT_TRAIN_BWD_FP32, before[_lib._OPTION_KEYS.index(_lib.OPT_TRAIN_BWD_FP32)]))
    return (gz if want_gz else None), gp


def flow_log_prob_train(z, params, bn_mean, bn_alpha, D, S, L, U, reversible=True):
    M, Mp, N = z.shape[0], (params.shape[0] if params.dim() > 1 else 1), z.shape[1]
    if reversible and flow_train_rev_supported(M, Mp, N, D, S, L, U):
        return _FlowLogProbRevFn.apply(z, params, bn_mean, bn_alpha, D, S, L, U)
    return _FlowLogProbFn.apply(z, params, bn_mean, bn_alpha, D, S, L, U)


# ---------------------------------------------------------------------------
# MAF (arch_type "AR")
# ---------------------------------------------------------------------------
def maf_raw(z, params, masks, D, L, U, inverse):
    """tnf_maf.  masks: 1-D tensor (the layer masks concatenated).  Returns (z_out, log_det (M,N))."""
    _check3(z)
    dev = _lib.require_device()
    home = z.device
    code = _dtype_code(z)
    if params.dtype != z.dtype:
        raise TypeError("z (%s) and params (%s) must have the same dtype" % (z.dtype, params.dtype))
    zc = _stage(z, dev)
    pc, pstride = _rows(params, dev)
    mk = masks if (masks.device == dev and masks.dtype == z.dtype) else _stage(masks.to(z.dtype), dev)
    Mz, N = zc.shape[0], zc.shape[1]
    Mp = pc.shape[0]
    M = _bcast_M(Mz, Mp)
    if zc.shape[2] != D:
        raise ValueError("last dimension of z (%d) must equal D (%d)" % (zc.shape[2], D))
    z_out = torch.empty((M, N, D), dtype=z.dtype, device=dev)
    log_det = torch.empty((M, N), dtype=z.dtype, device=dev)
    if N > 0:
        check(lib.tnf_maf(code, zc.data_ptr(), pc.data_ptr(), mk.data_ptr(), z_out.data_ptr(), log_det.data_ptr(),
                          Mz, Mp, N, D, L, U, int(inverse), pstride, _lib.stream_ptr()))
    if home != dev:
        z_out, log_det = z_out.to(home), log_det.to(home)
    return z_out, log_det


def maf_inverse_alpha_raw(x, params, masks, D, L, U):
    """tnf_maf_inverse_alpha: per-dimension f_alpha(x) (M, N, D) of MAF.inverse_and_log_det on the compute device."""
    dev = _lib.require_device()
    xc = _stage(x, dev)
    pc, pstride = _rows(params, dev)
    mk = masks if (masks.device == dev and masks.dtype == x.dtype) else _stage(masks.to(x.dtype), dev)
    Mz, N = xc.shape[0], xc.shape[1]
    Mp = pc.shape[0]
    M = _bcast_M(Mz, Mp)
    out = torch.empty((M, N, D), dtype=x.dtype, device=dev)
    ld = torch.empty((M, N), dtype=x.dtype, device=dev)
    alpha = torch.empty((M, N, D), dtype=x.dtype, device=dev)
    if N > 0:
        check(lib.tnf_maf_inverse_alpha(_dtype_code(x), xc.data_ptr(), pc.data_ptr(), mk.data_ptr(), out.data_ptr(),
                                        ld.data_ptr(), alpha.data_ptr(), Mz, Mp, N, D, L, U, pstride, _lib.stream_ptr()))
    return alpha


class _MafFn(torch.autograd.Function):
    """MAF in either direction.  The backward of the sampling direction (the reference differentiates the D - 1
    passes of bijectors.py:752-754) uses the implicit function theorem instead: the sample x solves
    G(x, theta) = omega with G the one-pass inverse map, whose Jacobian G_x is triangular in the autoregressive order
    with diagonal e^-alpha.  With K(g_out, g_ld) the inverse-direction backward kernel (K_z = G_x^T g_out + A_x^T g_ld,
    K_theta likewise; A = sum alpha), the total gradient w.r.t. x is w = g_x + A_x^T g_ld, v = G_x^-T w is reached
    exactly by D sweeps of  v <- v + e^alpha (g_x - K_z(v, -g_ld))  (the error moves strictly along the
    autoregressive order), and then  g_omega = v,  g_theta = -K_theta(v, -g_ld)."""

    @_records_options
    def forward(ctx, z, params, masks, D, L, U, inverse):
        z_out, log_det = maf_raw(z, params, masks, D, L, U, inverse)
        if inverse:
            ctx.save_for_backward(z, params, masks)
        else:
            ctx.save_for_backward(z_out, params, masks)  # the sample: everything is evaluated there
        ctx.cfg = (D, L, U, inverse, z.shape[0])
        ctx.prec = current_operand_precision()
        return z_out, log_det

    @_reenters_options
    def backward(ctx, g_z, g_ld):
        with operand_precision(ctx.prec):  # autograd's thread: re-enter the forward's operand precision
            return _MafFn._backward(ctx, g_z, g_ld)

    @staticmethod
    def _backward(ctx, g_z, g_ld):
        from . import grad

        z, params, masks = ctx.saved_tensors
        D, L, U, inverse, Mz = ctx.cfg
        if inverse:
            gz, gp = grad.maf_backward(z, params, masks, g_z, g_ld, D, L, U)
            return gz, gp, None, None, None, None, None
        x = z.detach()
        dev = _lib.require_device()
        gx = torch.zeros_like(x) if g_z is None else g_z.to(x.dtype)
        gl = torch.zeros(x.shape[:2], dtype=x.dtype, device=x.device) if g_ld is None else g_ld.to(x.dtype)
        ea = torch.exp(maf_inverse_alpha_raw(x, params.detach(), masks, D, L, U)).to(x.device)
        v = ea * gx
        for _ in range(D):  # a strictly triangular D x D coupling is nilpotent of index <= D: exact after D sweeps
            kz, _gp = grad.maf_backward(x, params, masks, v, -gl, D, L, U)
            v = v + ea * (gx - kz)
        _kz, gp = grad.maf_backward(x, params, masks, v, -gl, D, L, U)
        g_omega = v
        if Mz != v.shape[0]:
            g_omega = v.sum(0, keepdim=True)
        return g_omega, -gp, None, None, None, None, None


def maf(z, params, masks, D, L, U, inverse):
    if torch.is_grad_enabled() and (z.requires_grad or params.requires_grad):
        return _MafFn.apply(z, params, masks, D, L, U, inverse)
    return maf_raw(z, params, masks, D, L, U, inverse)


# ---------------------------------------------------------------------------
# Support layers (ToInterval / ToSimplex): parameter-free, elementwise + per-row log-det
# ---------------------------------------------------------------------------
def _grad_or_zeros(g, shape, dtype, dev):
    if g is None:
        return torch.zeros(shape, dtype=dtype, device=dev)
    return _stage(g, dev)


def to_interval_raw(z, consts, inverse):
    """tnf_to_interval.  consts: (7, D) float32 rows (see include/tnf.h).  Returns (z_out, log_det (M,N))."""
    _check3(z)
    dev = _lib.require_device()
    home = z.device
    code = _dtype_code(z)
    zc = _stage(z, dev)
    cc = _stats(consts, dev)
    M, N, D = zc.shape
    if cc.shape != (7, D):
        raise ValueError("last dimension of z (%d) must equal the bijector's D (%d)" % (D, cc.shape[1]))
    z_out = torch.empty_like(zc)
    log_det = torch.empty((M, N), dtype=z.dtype, device=dev)
    check(lib.tnf_to_interval(code, zc.data_ptr(), cc.data_ptr(), z_out.data_ptr(), log_det.data_ptr(), M * N, D,
                              int(inverse), _lib.stream_ptr()))
    if home != dev:
        z_out, log_det = z_out.to(home), log_det.to(home)
    return z_out, log_det


class _ToIntervalFn(torch.autograd.Function):
    @_records_options
    def forward(ctx, z, consts, inverse):
        out = to_interval_raw(z, consts, inverse)
        ctx.save_for_backward(z, consts)
        ctx.inverse = inverse
        return out

    @_reenters_options
    def backward(ctx, g_z, g_ld):
        z, consts = ctx.saved_tensors
        dev = _lib.require_device()
        zc = _stage(z.detach(), dev)
        M, N, D = zc.shape
        g_zo = _grad_or_zeros(g_z, (M, N, D), z.dtype, dev)
        g_l = _grad_or_zeros(g_ld, (M, N), z.dtype, dev)
        gz = torch.empty_like(zc)
        check(lib.tnf_to_interval_backward(_dtype_code(z), zc.data_ptr(), _stats(consts, dev).data_ptr(),
                                           g_zo.data_ptr(), g_l.data_ptr(), gz.data_ptr(), M * N, D,
                                           int(ctx.inverse), _lib.stream_ptr()))
        return (gz if z.device == dev else gz.to(z.device)), None, None


def to_interval(z, consts, inverse):
    if torch.is_grad_enabled() and z.requires_grad:
        return _ToIntervalFn.apply(z, consts, inverse)
    return to_interval_raw(z, consts, inverse)


def to_simplex_raw(z, D_attr):
    """tnf_to_simplex: (M, N, D_in) -> ((M, N, D_in + 1), log_det (M, N))."""
    _check3(z)
    dev = _lib.require_device()
    home = z.device
    code = _dtype_code(z)
    zc = _stage(z, dev)
    M, N, Din = zc.shape
    z_out = torch.empty((M, N, Din + 1), dtype=z.dtype, device=dev)
    log_det = torch.empty((M, N), dtype=z.dtype, device=dev)
    check(lib.tnf_to_simplex(code, zc.data_ptr(), z_out.data_ptr(), log_det.data_ptr(), M * N, Din, D_attr,
                             _lib.stream_ptr()))
    if home != dev:
        z_out, log_det = z_out.to(home), log_det.to(home)
    return z_out, log_det


class _ToSimplexFn(torch.autograd.Function):
    @_records_options
    def forward(ctx, z, D_attr):
        out = to_simplex_raw(z, D_attr)
        ctx.save_for_backward(z)
        ctx.D_attr = D_attr
        return out

    @_reenters_options
    def backward(ctx, g_z, g_ld):
        (z,) = ctx.saved_tensors
        dev = _lib.require_device()
        zc = _stage(z.detach(), dev)
        M, N, Din = zc.shape
        g_zo = _grad_or_zeros(g_z, (M, N, Din + 1), z.dtype, dev)
        g_l = _grad_or_zeros(g_ld, (M, N), z.dtype, dev)
        gz = torch.empty_like(zc)
        check(lib.tnf_to_simplex_backward(_dtype_code(z), zc.data_ptr(), g_zo.data_ptr(), g_l.data_ptr(),
                                          gz.data_ptr(), M * N, Din, ctx.D_attr, _lib.stream_ptr()))
        return (gz if z.device == dev else gz.to(z.device)), None


def to_simplex(z, D_attr):
    if torch.is_grad_enabled() and z.requires_grad:
        return _ToSimplexFn.apply(z, D_attr)
    return to_simplex_raw(z, D_attr)


# ---------------------------------------------------------------------------
# Conditional flow with one sample per context: param_net's last Linear fused into the flow
# ---------------------------------------------------------------------------
def _cond_width(H):
    for w in (32, 64, 128):
        if H <= w:
            return w
    return 0


def cond_flow_supported(D, S, L, U, H):
    Hp = _cond_width(H)
    return bool(Hp) and bool(lib.tnf_cond_flow_supported(D, S, L, U, Hp))


def _pad_cols(t, width):
    t = t.detach().float()
    if t.shape[1] != width:
        t = torch.nn.functional.pad(t, (0, width - t.shape[1]))
    return t.contiguous()


def cond_flow_log_prob_raw(z, h, weight, bias, bn_mean, bn_alpha, D, S, L, U, want_z0=False, want_sld=False):
    """tnf_cond_flow_log_prob_f32: z (M, D), h (M, H) = input of param_net's last Linear, weight
    (D_params, H) / bias (D_params) of that Linear.  Returns (log_prob (M), z0 | None, sum_log_det | None)."""
    dev = _lib.require_device()
    M, H = h.shape
    Hp = _cond_width(H)
    if not Hp:
        raise ValueError("conditioner width %d not supported by the fused kernel (max 128)" % H)
    zc = _stage(z.detach().float(), dev)
    if zc.shape != (M, D):
        raise ValueError("z must be (M, D) = (%d, %d), got %s" % (M, D, tuple(zc.shape)))
    hc = _pad_cols(_stage(h, dev), Hp)
    wc = _pad_cols(_stage(weight, dev), Hp)
    bc = _stage(bias.detach().float(), dev)
    mean = _stats(bn_mean, dev)
    alpha = _stats(bn_alpha, dev)
    lp = torch.empty((M,), dtype=torch.float32, device=dev)
    z0 = torch.empty((M, D), dtype=torch.float32, device=dev) if want_z0 else None
    sld = torch.empty((M,), dtype=torch.float32, device=dev) if want_sld else None
    nbytes = check(lib.tnf_cond_flow_workspace_bytes(D, S, L, U, Hp))
    ws = _workspace(nbytes, dev)
    if M > 0:
        check(lib.tnf_cond_flow_log_prob_f32(zc.data_ptr(), hc.data_ptr(), wc.data_ptr(), bc.data_ptr(),
                                             mean.data_ptr(), alpha.data_ptr(), lp.data_ptr(),
                                             z0.data_ptr() if want_z0 else None,
                                             sld.data_ptr() if want_sld else None, M, D, S, L, U, Hp,
                                             hc.stride(0), wc.stride(0), ws.data_ptr(), nbytes, _lib.stream_ptr()))
    return lp, z0, sld


def cond_flow_forward_raw(omega, h, weight, bias, bn_mean, bn_alpha, D, S, L, U):
    """tnf_cond_flow_forward_f32: omega (M, D) base draws, h (M, H) = input of param_net's last Linear, weight
    (D_params, H) / bias (D_params) of that Linear.  Returns (z (M, D), sum_log_det (M)); params never exist."""
    dev = _lib.require_device()
    M, H = h.shape
    Hp = _cond_width(H)
    if not Hp:
        raise ValueError("conditioner width %d not supported by the fused kernel (max 128)" % H)
    oc = _stage(omega.detach().float(), dev)
    if oc.shape != (M, D):
        raise ValueError("omega must be (M, D) = (%d, %d), got %s" % (M, D, tuple(oc.shape)))
    hc = _pad_cols(_stage(h.detach(), dev), Hp)
    wc = _pad_cols(_stage(weight.detach(), dev), Hp)
    bc = _stage(bias.detach().float(), dev)
    mean = _stats(bn_mean, dev)
    alpha = _stats(bn_alpha, dev)
    z = torch.empty((M, D), dtype=torch.float32, device=dev)
    sld = torch.empty((M,), dtype=torch.float32, device=dev)
    nbytes = check(lib.tnf_cond_flow_workspace_bytes(D, S, L, U, Hp))
    ws = _workspace(nbytes, dev)
    if M > 0:
        check(lib.tnf_cond_flow_forward_f32(oc.data_ptr(), hc.data_ptr(), wc.data_ptr(), bc.data_ptr(), mean.data_ptr(),
                                            alpha.data_ptr(), z.data_ptr(), sld.data_ptr(), M, D, S, L, U, Hp,
                                            hc.stride(0), wc.stride(0), ws.data_ptr(), nbytes, _lib.stream_ptr()))
    return z, sld


class _CondFlowLogProbFn(torch.autograd.Function):
    """Training pair of the fused conditioner + flow: tnf_cond_flow_log_prob_fwd_f32 keeps per-layer
    activations (6 KB per context at D = 64), tnf_cond_flow_log_prob_bwd_f32 returns the gradients of
    param_net's last Linear and of its input; torch autograd carries on through the rest of param_net."""

    @_records_options
    def forward(ctx, z, h, weight, bias, bn_mean, bn_alpha, D, S, L, U):
        dev = _lib.require_device()
        M, H = h.shape
        Hp = _cond_width(H)
        zc = _stage(z.detach().float(), dev)
        hc = _pad_cols(_stage(h, dev), Hp)
        wc = _pad_cols(_stage(weight, dev), Hp)
        bc = _stage(bias.detach().float(), dev)
        mean, alpha = _stats(bn_mean, dev), _stats(bn_alpha, dev)
        lp = torch.empty((M,), dtype=torch.float32, device=dev)
        acts = torch.empty((max(1, check(lib.tnf_cond_flow_acts_floats(M, D, S, L))),), dtype=torch.float32, device=dev)
        nbytes = check(lib.tnf_cond_flow_workspace_bytes(D, S, L, U, Hp))
        ws = _workspace(nbytes, dev)
        if M > 0:
            check(lib.tnf_cond_flow_log_prob_fwd_f32(zc.data_ptr(), hc.data_ptr(), wc.data_ptr(), bc.data_ptr(),
                                                     mean.data_ptr(), alpha.data_ptr(), lp.data_ptr(),
                                                     acts.data_ptr(), M, D, S, L, U, Hp, hc.stride(0), wc.stride(0),
                                                     ws.data_ptr(), nbytes, _lib.stream_ptr()))
        ctx.save_for_backward(hc, wc, bc, mean, alpha, acts)
        ctx.cfg = (M, H, Hp, D, S, L, U)
        ctx.homes = (z.device, h.device, weight.device, bias.device)
        return lp

    @_reenters_options
    def backward(ctx, g_lp):
        hc, wc, bc, mean, alpha, acts = ctx.saved_tensors
        M, H, Hp, D, S, L, U = ctx.cfg
        dev = hc.device
        P = wc.shape[0]
        need_z = ctx.needs_input_grad[0]
        g = _stage(g_lp.detach().float(), dev)
        g_h = torch.zeros((M, Hp), dtype=torch.float32, device=dev)
        g_w = torch.empty((P, Hp), dtype=torch.float32, device=dev)
        g_b = torch.empty((P,), dtype=torch.float32, device=dev)
        g_z = torch.zeros((M, D), dtype=torch.float32, device=dev) if need_z else None
        deltas = torch.empty((max(1, check(lib.tnf_cond_flow_deltas_floats(M, D, S, L, Hp))),), dtype=torch.float32,
                             device=dev)
        nbytes = check(lib.tnf_cond_flow_bwd_workspace_bytes(D, S, L, U, Hp))
        ws = _workspace(nbytes, dev)
        check(lib.tnf_cond_flow_log_prob_bwd_f32(g.data_ptr(), hc.data_ptr(), wc.data_ptr(), bc.data_ptr(),
                                                 mean.data_ptr(), alpha.data_ptr(), acts.data_ptr(),
                                                 deltas.data_ptr(), g_h.data_ptr(), g_w.data_ptr(), g_b.data_ptr(),
                                                 g_z.data_ptr() if need_z else None, M, D, S, L, U, Hp,
                                                 hc.stride(0), wc.stride(0), g_h.stride(0), g_w.stride(0),
                                                 ws.data_ptr(), nbytes, _lib.stream_ptr()))
        hz, hh, hw, hb = ctx.homes
        out_z = g_z.to(hz) if need_z else None
        return (out_z, g_h[:, :H].to(hh), g_w[:, :H].to(hw), g_b.to(hb), None, None, None, None, None, None)


def cond_flow_log_prob_train(z, h, weight, bias, bn_mean, bn_alpha, D, S, L, U):
    return _CondFlowLogProbFn.apply(z, h, weight, bias, bn_mean, bn_alpha, D, S, L, U)


# ---------------------------------------------------------------------------
# NormFlow(arch_type="AR") = [MAF, BatchNorm, Affine] in one kernel per call (float32, no autograd)
# ---------------------------------------------------------------------------
def ar_flow_supported(D, L, U):
    return bool(lib.tnf_ar_flow_supported(D, L, U))


def _ar_common(z, params, masks, bn_mean, bn_alpha, D):
    _check3(z)
    dev = _lib.require_device()
    zc = _stage(z.detach(), dev)
    pc, pstride = _rows(params.detach(), dev)
    mk = masks if (masks.device == dev and masks.dtype == torch.float32) else _stage(masks.float(), dev)
    mean, alpha = _stats(bn_mean.reshape(-1), dev), _stats(bn_alpha.reshape(-1), dev)
    Mz, N = zc.shape[0], zc.shape[1]
    Mp = pc.shape[0]
    M = _bcast_M(Mz, Mp)
    if zc.shape[2] != D:
        raise ValueError("last dimension of z (%d) must equal D (%d)" % (zc.shape[2], D))
    nbytes = check(lib.tnf_ar_flow_workspace_bytes(Mp, D))
    return dev, zc, pc, pstride, mk, mean, alpha, Mz, Mp, M, N, _workspace(nbytes, dev), nbytes


def ar_flow_log_prob_raw(z, params, masks, bn_mean, bn_alpha, D, L, U, want_lp=True, want_z0=False, want_sld=False,
                         interval_consts=None):
    """tnf_ar_flow_log_prob_f32 -> (log_prob | None, z0 | None, sum_log_det | None) on z's device.
    interval_consts: the (7, D) device constants of a ToInterval support layer fused into the kernel."""
    home = z.device
    dev, zc, pc, pstride, mk, mean, alpha, Mz, Mp, M, N, ws, nbytes = _ar_common(z, params, masks, bn_mean, bn_alpha, D)
    lp = torch.empty((M, N), dtype=torch.float32, device=dev) if want_lp else None
    z0 = torch.empty((M, N, D), dtype=torch.float32, device=dev) if want_z0 else None
    sld = torch.empty((M, N), dtype=torch.float32, device=dev) if want_sld else None
    check(lib.tnf_ar_flow_log_prob_f32(zc.data_ptr(), pc.data_ptr(), mk.data_ptr(), mean.data_ptr(), alpha.data_ptr(),
                                       None if interval_consts is None else interval_consts.data_ptr(),
                                       lp.data_ptr() if want_lp else None, z0.data_ptr() if want_z0 else None,
                                       sld.data_ptr() if want_sld else None, Mz, Mp, N, D, L, U, pstride,
                                       ws.data_ptr(), nbytes, _lib.stream_ptr()))
    if home != dev:
        lp, z0, sld = (t if t is None else t.to(home) for t in (lp, z0, sld))
    return lp, z0, sld


def ar_flow_train_supported(M, Mp, D, L, U):
    return Mp in (1, M) and lib.tnf_ar_flow_train_supported(D, L, U) == 1


class _ArFlowLogProbFn(torch.autograd.Function):
    """NormFlow('AR').log_prob with z a constant of the graph (the LFI scripts' training loop): the forward is the
    one-kernel inference path (tnf_ar_flow_log_prob_f32), the backward ONE kernel for the whole stack
    (tnf_ar_flow_log_prob_bwd_f32: ToInterval^-1, the folded Affine / BatchNorm, MAF recompute and backward,
    base density) writing the gradient of every parameter slice into one buffer."""

    @_records_options
    def forward(ctx, z, params, masks, bn_mean, bn_alpha, interval_consts, D, L, U):
        lp, _, _ = ar_flow_log_prob_raw(z, params, masks, bn_mean, bn_alpha, D, L, U, interval_consts=interval_consts)
        dev = _lib.require_device()
        pc, pstride = _rows(params.detach(), dev)
        ctx.save_for_backward(_stage(z.detach(), dev), pc,
                              masks if (masks.device == dev and masks.dtype == torch.float32) else _stage(masks.float(), dev),
                              _stats(bn_mean.reshape(-1), dev), _stats(bn_alpha.reshape(-1), dev))
        ctx.consts = interval_consts
        ctx.cfg = (D, L, U, pstride, params.device, tuple(params.shape))
        ctx.prec = current_operand_precision()
        return lp

    @_reenters_options
    def backward(ctx, g_lp):
        zc, pc, mk, mean, alpha = ctx.saved_tensors
        D, L, U, pstride, p_home, p_shape = ctx.cfg
        dev = zc.device
        M, N = zc.shape[0], zc.shape[1]
        Mp = pc.shape[0]
        g = _stage(g_lp.float(), dev)
        gp = torch.zeros(p_shape, dtype=torch.float32, device=dev)
        nbytes = check(lib.tnf_ar_flow_bwd_workspace_bytes(Mp, D))
        ws = _workspace(nbytes, dev)
        with operand_precision(ctx.prec):
            check(lib.tnf_ar_flow_log_prob_bwd_f32(zc.data_ptr(), pc.data_ptr(), mk.data_ptr(), mean.data_ptr(),
                                                   alpha.data_ptr(),
                                                   None if ctx.consts is None else ctx.consts.data_ptr(), g.data_ptr(),
                                                   gp.data_ptr(), M, Mp, N, D, L, U, pstride, gp.shape[1], ws.data_ptr(),
                                                   nbytes, _lib.stream_ptr()))
        return None, (gp if p_home == dev else gp.to(p_home)), None, None, None, None, None, None, None


def ar_flow_log_prob_train(z, params, masks, bn_mean, bn_alpha, interval_consts, D, L, U):
    return _ArFlowLogProbFn.apply(z, params, masks, bn_mean, bn_alpha, interval_consts, D, L, U)


def ar_flow_forward_raw(omega, params, masks, bn_mean, bn_alpha, D, L, U, interval_consts=None):
    """tnf_ar_flow_forward_f32 (cached BatchNorm statistics) -> (z, sum_log_det) on the compute device."""
    dev, zc, pc, pstride, mk, mean, alpha, Mz, Mp, M, N, ws, nbytes = _ar_common(omega, params, masks, bn_mean,
                                                                               bn_alpha, D)
    z = torch.empty((M, N, D), dtype=torch.float32, device=dev)
    sld = torch.empty((M, N), dtype=torch.float32, device=dev)
    check(lib.tnf_ar_flow_forward_f32(zc.data_ptr(), pc.data_ptr(), mk.data_ptr(), mean.data_ptr(), alpha.data_ptr(),
                                      None if interval_consts is None else interval_consts.data_ptr(),
                                      z.data_ptr(), sld.data_ptr(), Mz, Mp, N, D, L, U, pstride, ws.data_ptr(), nbytes,
                                      _lib.stream_ptr()))
    return z, sld
