"""The C-ABI library loads on a machine without a GPU, exports every symbol include/tnf.h
declares, and its host-only entry points / argument checks behave (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "tnf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tnf_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    from torch_nf_amd import _lib

    names = declared_functions()
    assert len(names) >= 15 and "tnf_flow_log_prob_f32" in names and "tnf_coupling" in names
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libtnf_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "torch_nf_amd/_lib.py does not bind %s" % n
    assert sorted(_lib.SIGNATURES) == names, "binding table and header are out of step"


def test_version_and_counts(oracle):
    from torch_nf_amd._lib import lib

    assert lib.tnf_version() == 100
    for D in (2, 4, 5, 9, 32, 64):
        for L in (1, 2, 5):
            for U in (15, 16, 100):
                for up in (0, 1):
                    assert lib.tnf_coupling_num_params(D, L, U, up) == oracle.coupling_num_params(D, L, U, bool(up))
                assert lib.tnf_flow_num_params(D, 3, L, U) == oracle.flow_num_params(D, 3, L, U)
    assert lib.tnf_flow_num_params(64, 4, 2, 15) == 20464
    assert lib.tnf_has_fast_path(64, 2, 15) == 1 and lib.tnf_has_fast_path(32, 2, 15) == 1
    assert lib.tnf_has_fast_path(5, 2, 15) == 0 and lib.tnf_has_fast_path(64, 2, 65) == 0
    assert lib.tnf_has_fast_path(64, 2, 64) == 1 and lib.tnf_has_fast_path(8, 5, 20) == 1  # wide per-layer kernel
    assert lib.tnf_flow_fused_supported(64, 4, 2, 15) == 1
    assert lib.tnf_flow_fused_supported(64, 40, 2, 15) == 0  # 80 layers of operands do not fit 160 KB of LDS


def test_argument_checks_return_codes_without_launching():
    from torch_nf_amd import _lib

    lib = _lib.lib
    dummy = ctypes.c_void_p(16)  # never dereferenced: every call below fails validation first
    rc = lib.tnf_coupling(_lib.F32, dummy, dummy, dummy, dummy, 2, 3, 4, 8, 2, 15, 1, 0, 1000, 0, None)
    assert rc == -1 and b"do not broadcast" in lib.tnf_last_error()
    rc = lib.tnf_coupling(_lib.F32, dummy, dummy, dummy, dummy, 1, 1, 4, 8, 2, 15, 1, 0, 10, 0, None)
    assert rc == -1 and b"params row has 10 elements" in lib.tnf_last_error()
    rc = lib.tnf_coupling(7, dummy, dummy, dummy, dummy, 1, 1, 4, 8, 2, 15, 1, 0, 1000, 0, None)
    assert rc == -1 and b"dtype" in lib.tnf_last_error()
    rc = lib.tnf_coupling(_lib.F32, dummy, dummy, None, dummy, 1, 1, 4, 8, 2, 15, 1, 0, 1000, 0, None)
    assert rc == -1 and b"NULL" in lib.tnf_last_error()
    rc = lib.tnf_flow_log_prob_f32(dummy, dummy, dummy, dummy, None, dummy, None, None, 1, 1, 4, 6, 1, 2, 15, 10000,
                                   0, dummy, 1 << 30, None)
    assert rc == _lib.EUNSUPPORTED and b"no fused kernel" in lib.tnf_last_error()
    rc = lib.tnf_flow_log_prob_f32(dummy, dummy, dummy, dummy, None, dummy, None, None, 1, 1, 4, 64, 4, 2, 15, 20464,
                                   0, dummy, 16, None)
    assert rc == -4 and b"workspace" in lib.tnf_last_error()
    assert lib.tnf_set_option(99, 1) == -1
    try:
        _lib.check(lib.tnf_coupling_num_params(0, 1, 1, 1))
        assert False
    except _lib.TnfError as e:
        assert e.code == -1
    assert lib.tnf_flow_workspace_bytes(1, 1 << 20, 64, 4, 2, 15, _lib.FUSE_FLOW) < 1 << 18  # constants + operand images only
    assert lib.tnf_flow_workspace_bytes(1, 1 << 20, 64, 4, 2, 15, _lib.FUSE_LAYER) > (1 << 20) * 64 * 4


def test_widened_entries_host_side(oracle):
    """Host-only behaviour of the entries added for SURVEY 8f: counts, support predicates, sizes and the
    argument checks that fail before any launch."""
    from torch_nf_amd import _lib

    lib = _lib.lib
    dummy = ctypes.c_void_p(256)
    # MAF
    for D, L, U in [(4, 2, 20), (64, 2, 64), (5, 1, 15)]:
        assert lib.tnf_maf_num_params(D, L, U) == oracle.maf_num_params(D, L, U)
    assert lib.tnf_ar_flow_supported(4, 2, 20) == 1 and lib.tnf_ar_flow_supported(64, 2, 64) == 1
    assert lib.tnf_ar_flow_supported(64, 5, 64) == 0  # 192 KB of operands do not fit the LDS
    assert lib.tnf_ar_flow_supported(65, 2, 20) == 0 and lib.tnf_ar_flow_supported(8, 2, 65) == 0
    assert lib.tnf_ar_flow_workspace_bytes(3, 8) >= 3 * 17 * 4
    rc = lib.tnf_ar_flow_log_prob_f32(dummy, dummy, dummy, dummy, dummy, None, dummy, None, None, 1, 1, 4, 4, 2, 20, 10,
                                      dummy, 1 << 20, None)
    assert rc == -1 and b"params row has 10 elements" in lib.tnf_last_error()
    rc = lib.tnf_ar_flow_log_prob_f32(dummy, dummy, dummy, dummy, dummy, None, None, None, None, 1, 1, 4, 4, 2, 20, 10000,
                                      dummy, 1 << 20, None)
    assert rc == -1 and b"no output requested" in lib.tnf_last_error()
    rc = lib.tnf_maf(7, dummy, dummy, dummy, dummy, dummy, 1, 1, 4, 4, 2, 20, 1, 10000, None)
    assert rc == -1 and b"dtype" in lib.tnf_last_error()
    # support layers
    assert lib.tnf_to_interval(_lib.F32, dummy, dummy, dummy, dummy, 0, 4, 0, None) == 0  # empty input: no-op
    assert lib.tnf_to_interval(_lib.F32, None, dummy, dummy, dummy, 5, 4, 0, None) == -1
    assert lib.tnf_to_simplex(_lib.F64, dummy, dummy, dummy, 3, 0, 4, None) == -1
    # conditional flow
    assert lib.tnf_cond_flow_supported(64, 4, 2, 15, 64) == 1 and lib.tnf_cond_flow_supported(32, 1, 5, 16, 128) == 1
    assert lib.tnf_cond_flow_supported(64, 4, 2, 17, 64) == 0 and lib.tnf_cond_flow_supported(64, 4, 2, 15, 50) == 0
    assert lib.tnf_cond_flow_supported(16, 4, 2, 15, 64) == 0
    ws = lib.tnf_cond_flow_workspace_bytes(64, 4, 2, 15, 64)
    assert 5 << 20 < ws < 6 << 20                      # the 1,328-tile operand image (4,160 B per tile)
    assert lib.tnf_cond_flow_bwd_workspace_bytes(64, 4, 2, 15, 64) > 2 * ws - (1 << 20)
    assert lib.tnf_cond_flow_workspace_bytes(64, 4, 2, 15, 50) == _lib.EUNSUPPORTED
    M = 1000
    assert lib.tnf_cond_flow_acts_floats(M, 64, 4, 2) == 4 * M * 64 + 8 * M * (3 * 32 + 64)
    assert lib.tnf_cond_flow_deltas_floats(M, 64, 4, 2, 64) == 8 * M * 64 + 8 * M * (64 + 64) + 1024 * 64
    rc = lib.tnf_cond_flow_log_prob_f32(dummy, dummy, dummy, dummy, dummy, dummy, dummy, None, None, 10, 64, 4, 2, 15,
                                        64, 62, 64, dummy, ws, None)
    assert rc == -1 and b"multiples of 4" in lib.tnf_last_error()
    rc = lib.tnf_cond_flow_log_prob_f32(dummy, dummy, dummy, dummy, dummy, dummy, dummy, None, None, 10, 64, 4, 2, 15,
                                        64, 64, 64, dummy, 1024, None)
    assert rc == -4 and b"workspace" in lib.tnf_last_error()
    rc = lib.tnf_cond_flow_log_prob_f32(dummy, dummy, dummy, dummy, dummy, dummy, dummy, None, None, 10, 64, 4, 2, 15,
                                        64, 64, 64, ctypes.c_void_p(264), ws, None)
    assert rc == -1 and b"aligned" in lib.tnf_last_error()
    assert lib.tnf_set_option(_lib.OPT_COND_VARIANT, 0) == 0


def test_training_entries_host_side():
    """Support queries, workspace sizes and argument checks of the training entry points added after the metric
    path (reversible pair, AR one-kernel backward, batch-statistics chains): host-only, no launches."""
    from torch_nf_amd._lib import lib

    # reversible pair: the 2S layers' gradient accumulators must fit the 160 KB LDS
    assert lib.tnf_flow_train_rev_supported(64, 4, 2, 15) == 1 and lib.tnf_flow_train_rev_supported(32, 4, 3, 16) == 1
    assert lib.tnf_flow_train_rev_supported(64, 5, 2, 15) == 0 and lib.tnf_flow_train_rev_supported(64, 4, 3, 15) == 0
    assert lib.tnf_flow_train_rev_supported(8, 2, 2, 15) == 0 and lib.tnf_flow_train_rev_supported(64, 4, 2, 17) == 0
    w1 = lib.tnf_flow_train_rev_workspace_bytes(1, 1, 4096, 64, 4, 2, 15)
    w4 = lib.tnf_flow_train_rev_workspace_bytes(4, 4, 4096, 64, 4, 2, 15)
    assert 0 < w1 < w4 and lib.tnf_flow_train_rev_workspace_bytes(1, 1, 4096, 64, 5, 2, 15) < 0
    # the deterministic reduction keeps one partial gradient row per workgroup: 256 of them at the benchmark size
    assert lib.tnf_flow_train_rev_workspace_bytes(1, 1, 1 << 19, 64, 4, 2, 15) >= 256 * 8 * (2494 + 128) * 4
    # AR one-kernel backward: the MFMA MAF backward's shapes
    assert lib.tnf_ar_flow_train_supported(6, 2, 15) == 1 and lib.tnf_ar_flow_train_supported(21, 2, 42) == 1
    assert lib.tnf_ar_flow_train_supported(33, 2, 15) == 0 and lib.tnf_ar_flow_train_supported(6, 4, 15) == 0
    assert lib.tnf_ar_flow_bwd_workspace_bytes(2000, 6) >= 2000 * (4 * 6 + 2) * 4
    # batch-statistics chains
    assert 0 < lib.tnf_flow_forward_batch_workspace_bytes(1, 64, 4, 2) < lib.tnf_flow_forward_batch_workspace_bytes(8, 64, 4, 2)
    a = lib.tnf_flow_forward_train_workspace_bytes(1, 1, 1 << 16, 64, 4, 2)
    assert a >= 2 * (1 << 16) * 64 * 4 and lib.tnf_flow_forward_train_workspace_bytes(2, 3, 10, 64, 4, 2) < 0
    # argument checks come before any launch (NULL pointers, no GPU needed)
    EINVAL, EUNSUP = -1, -2
    n = None
    assert lib.tnf_flow_log_prob_fwd_rev_f32(n, n, n, n, n, n, 1, 1, 16, 8, 2, 2, 15, 100000, n) == EUNSUP
    assert lib.tnf_flow_log_prob_bwd_rev_f32(n, n, n, n, n, n, n, 2, 3, 16, 64, 4, 2, 15, 100000, 100000, n, 0, n, n) == EINVAL
    assert lib.tnf_ar_flow_log_prob_bwd_f32(n, n, n, n, n, n, n, n, 1, 1, 16, 40, 2, 15, 100000, 100000, n, 0, n) == EUNSUP
    assert lib.tnf_flow_forward_batch_f32(n, n, n, n, n, n, 1, 1, 1, 64, 4, 2, 15, 100000, 1e-5, n, 0, n) == EINVAL  # one row
    assert lib.tnf_flow_forward_train_bwd_f32(n, n, n, n, n, n, n, n, n, n, 1, 1, 64, 8, 2, 2, 15, 100000, 100000, n, 0, n) == EUNSUP
    assert lib.tnf_maf_inverse_alpha(7, n, n, n, n, n, n, 1, 1, 4, 3, 2, 15, 1000, n) == EINVAL  # dtype
    assert b"dtype" in lib.tnf_last_error()
