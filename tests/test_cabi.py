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
    rc = lib.tnf_flow_log_prob_f32(dummy, dummy, dummy, dummy, dummy, None, None, 1, 1, 4, 6, 1, 2, 15, 10000,
                                   0, dummy, 1 << 30, None)
    assert rc == _lib.EUNSUPPORTED and b"no fused kernel" in lib.tnf_last_error()
    rc = lib.tnf_flow_log_prob_f32(dummy, dummy, dummy, dummy, dummy, None, None, 1, 1, 4, 64, 4, 2, 15, 20464,
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
