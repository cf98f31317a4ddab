"""ctypes binding of libtnf_hip.so (the C ABI declared in include/tnf.h).

This is the only place the package touches native code.  The library is built
in-tree by `__graft_entry__.build()` / `make -C torch_nf_amd/csrc`; if it is
missing the import fails loudly -- there is no CPU or PyTorch fallback for the
flow arithmetic anywhere in this package.
"""
import ctypes
import os

import torch  # imported first so the HIP runtime torch ships is the one the library binds to

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TNF_LIB_PATH") or os.path.join(_HERE, "lib", "libtnf_hip.so")  # override: kernel A/B builds

F32, F64 = 0, 1
LD_STORE, LD_ADD, LD_SUB = 0, 1, -1
FUSE_AUTO, FUSE_LAYER, FUSE_FLOW = 0, 1, 2
OPT_FORCE_GENERIC, OPT_FLOW_VARIANT, OPT_LAYER_VARIANT, OPT_COND_VARIANT, OPT_TRAIN_BWD_FP32 = 1, 2, 3, 4, 5
OPT_OPERAND_PREC = 6
OPT_REV_VARIANT = 7
EUNSUPPORTED = -2
DIAG_BWD_LAYER_FP32, DIAG_BWD_LAYER_F16, DIAG_BWD_GENERIC, DIAG_BWD_FLOW_REV = 0, 1, 2, 3
DIAG_MAF_BWD_MFMA, DIAG_MAF_BWD_GENERIC, DIAG_BWD_WIDE = 4, 5, 6

_vp, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float

# name -> (restype, argtypes); kept in step with include/tnf.h (tests/test_cabi.py checks it)
SIGNATURES = {
    "tnf_version": (ctypes.c_int, []),
    "tnf_last_error": (ctypes.c_char_p, []),
    "tnf_set_option": (ctypes.c_int, [_i32, _i32]),
    "tnf_get_option": (ctypes.c_int, [_i32, _vp]),
    "tnf_diag_launch_count": (_i64, [_i32]),
    "tnf_set_launch_gate": (ctypes.c_int, [_vp]),
    "tnf_gated_copy_f32": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "tnf_coupling_num_params": (_i64, [_i32, _i32, _i32, _i32]),
    "tnf_flow_num_params": (_i64, [_i32, _i32, _i32, _i32]),
    "tnf_has_fast_path": (ctypes.c_int, [_i32, _i32, _i32]),
    "tnf_coupling": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32,
                                    _i32, _i64, _i32, _vp]),
    "tnf_affine": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i64, _vp]),
    "tnf_bn_apply": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "tnf_bn_batch_workspace_bytes": (_i64, [_i32]),
    "tnf_bn_batch_forward_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp, _i64, _vp]),
    "tnf_bn_batch_moments_f32": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "tnf_bn_batch_normalize_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp, _i64, _vp]),
    "tnf_bn_batch_backward_sums_f32": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "tnf_bn_batch_backward_apply_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "tnf_coupling_backward": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                             _i32, _i32, _i32, _i64, _i64, _vp]),
    "tnf_coupling_backward_workspace_bytes": (_i64, [_i32, _i64, _i64, _i64, _i32, _i32, _i32, _i32]),
    "tnf_coupling_backward_ws": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                                _i32, _i32, _i32, _i64, _i64, _vp, _i64, _vp]),
    "tnf_maf_backward_workspace_bytes": (_i64, [_i32, _i64, _i64, _i64, _i32, _i32, _i32]),
    "tnf_maf_backward_ws": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                           _i32, _i64, _i64, _vp, _i64, _vp]),
    "tnf_affine_backward": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                           _i64, _i64, _vp]),
    "tnf_bn_apply_backward": (ctypes.c_int, [_i32, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "tnf_bn_batch_backward_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    "tnf_cond_flow_supported": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "tnf_cond_flow_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "tnf_cond_flow_log_prob_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32,
                                                   _i32, _i32, _i64, _i64, _vp, _i64, _vp]),
    "tnf_cond_flow_forward_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32,
                                                  _i32, _i32, _i64, _i64, _vp, _i64, _vp]),
    "tnf_cond_flow_acts_floats": (_i64, [_i64, _i32, _i32, _i32]),
    "tnf_cond_flow_deltas_floats": (_i64, [_i64, _i32, _i32, _i32, _i32]),
    "tnf_cond_flow_bwd_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "tnf_cond_flow_log_prob_fwd_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32,
                                                       _i32, _i32, _i64, _i64, _vp, _i64, _vp]),
    "tnf_cond_flow_log_prob_bwd_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                       _i64, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _vp,
                                                       _i64, _vp]),
    "tnf_ar_flow_supported": (ctypes.c_int, [_i32, _i32, _i32]),
    "tnf_ar_flow_workspace_bytes": (_i64, [_i64, _i32]),
    "tnf_ar_flow_log_prob_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                                 _i32, _i64, _vp, _i64, _vp]),
    "tnf_ar_flow_forward_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32,
                                                _i64, _vp, _i64, _vp]),
    "tnf_to_interval": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "tnf_to_interval_backward": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "tnf_to_simplex": (ctypes.c_int, [_i32, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "tnf_to_simplex_backward": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "tnf_maf_num_params": (_i64, [_i32, _i32, _i32]),
    "tnf_maf": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i64, _vp]),
    "tnf_maf_backward": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                        _i32, _i64, _i64, _vp]),
    "tnf_base_log_density_f64": (ctypes.c_int, [_i32, _vp, _vp, _i64, _i32, _vp]),
    "tnf_flow_workspace_bytes": (_i64, [_i64, _i64, _i32, _i32, _i32, _i32, _i32]),
    "tnf_flow_fused_supported": (ctypes.c_int, [_i32, _i32, _i32, _i32]),
    "tnf_flow_log_prob_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32,
                                             _i32, _i32, _i32, _i64, _i32, _vp, _i64, _vp]),
    "tnf_flow_log_prob_diag_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32,
                                                  _i32, _i32, _i32, _i64, _i32, _vp, _i64, _vp, _vp]),
    "tnf_flow_train_workspace_bytes": (_i64, [_i64, _i64, _i64, _i32, _i32, _i32, _i32]),
    "tnf_flow_log_prob_fwd_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32,
                                                 _i32, _i64, _vp, _i64, _vp]),
    "tnf_flow_log_prob_bwd_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32,
                                                 _i32, _i32, _i32, _i64, _i64, _vp, _i64, _vp]),
    "tnf_flow_forward_batch_workspace_bytes": (_i64, [_i64, _i32, _i32, _i32]),
    "tnf_flow_forward_batch_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32,
                                                  _i32, _i64, ctypes.c_float, _vp, _i64, _vp]),
    "tnf_flow_forward_batch_begin_f32": (ctypes.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _i64, _vp, _i64, _vp]),
    "tnf_flow_forward_batch_layer_f32": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32,
                                                        _i32, _i64, _vp, _i64, _vp]),
    "tnf_flow_forward_batch_fold_f32": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i64,
                                                       ctypes.c_float, _vp, _i64, _vp]),
    "tnf_flow_forward_batch_end_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "tnf_flow_forward_train_workspace_bytes": (_i64, [_i64, _i64, _i64, _i32, _i32, _i32]),
    "tnf_flow_forward_train_fwd_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                                      _i32, _i32, _i64, ctypes.c_float, _vp, _i64, _vp]),
    "tnf_flow_forward_train_bwd_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                                      _i32, _i32, _i32, _i32, _i64, _i64, _vp, _i64, _vp]),
    "tnf_maf_inverse_alpha": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i64,
                                             _vp]),
    "tnf_ar_flow_train_supported": (ctypes.c_int, [_i32, _i32, _i32]),
    "tnf_ar_flow_bwd_workspace_bytes": (_i64, [_i64, _i32]),
    "tnf_ar_flow_log_prob_bwd_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32,
                                                    _i32, _i32, _i64, _i64, _vp, _i64, _vp]),
    "tnf_flow_train_rev_supported": (ctypes.c_int, [_i32, _i32, _i32, _i32]),
    "tnf_flow_train_rev_workspace_bytes": (_i64, [_i64, _i64, _i64, _i32, _i32, _i32, _i32]),
    "tnf_flow_log_prob_fwd_rev_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                                     _i32, _i32, _i64, _vp]),
    "tnf_flow_log_prob_bwd_rev_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32,
                                                     _i32, _i32, _i32, _i64, _i64, _vp, _i64, _vp, _vp]),
    "tnf_flow_forward_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                            _i32, _i32, _i64, _i32, _vp, _i64, _vp]),
    "tnf_flow_forward_logq_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                                 _i32, _i32, _i64, _i32, _vp, _i64, _vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "torch_nf_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C torch_nf_amd/csrc` (hipcc, --offload-arch=gfx950). "
            "There is no non-HIP fallback." % LIB_PATH
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header / library out of step
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class TnfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libtnf_hip error %d: %s" % (code, msg))
        self.code = code


def check(rc):
    if rc < 0:
        raise TnfError(rc, lib.tnf_last_error().decode("utf-8", "replace"))
    return rc


_OPTION_KEYS = (OPT_FORCE_GENERIC, OPT_FLOW_VARIANT, OPT_LAYER_VARIANT, OPT_COND_VARIANT, OPT_TRAIN_BWD_FP32,
                OPT_OPERAND_PREC, OPT_REV_VARIANT)


def options_snapshot():
    """The calling thread's tnf_set_option values (options are thread-local in the library)."""
    out = []
    val = ctypes.c_int32(0)
    for key in _OPTION_KEYS:
        check(lib.tnf_get_option(key, ctypes.addressof(val)))
        out.append(val.value)
    return tuple(out)


class options_reentered(object):
    """`with options_reentered(snapshot)`: run a block under the options another thread had.  autograd calls
    Function.backward on its own device thread, where every option is still at its default: the Functions in ops.py
    record options_snapshot() in forward and re-enter it in backward, so a kernel variant chosen for a training step
    (TNF_OPT_TRAIN_BWD_FP32, TNF_OPT_FORCE_GENERIC, the operand precision) governs both of its halves."""

    def __init__(self, snapshot):
        self.snapshot = snapshot

    def __enter__(self):
        self.before = options_snapshot()
        if self.before != self.snapshot:
            for key, value in zip(_OPTION_KEYS, self.snapshot):
                check(lib.tnf_set_option(key, value))

    def __exit__(self, *exc):
        if self.before != self.snapshot:
            for key, value in zip(_OPTION_KEYS, self.before):
                lib.tnf_set_option(key, value)
        return False


def require_device():
    """The product path is HIP only: fail loudly when no GPU is visible."""
    if not torch.cuda.is_available():
        raise RuntimeError(
            "torch_nf_amd needs a HIP device (MI355X / gfx950): torch.cuda.is_available() is False "
            "and there is deliberately no CPU implementation of the flow kernels."
        )
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream
