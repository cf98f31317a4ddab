"""Host-side behaviour of the drop-in classes (no GPU needed): constructor validation,
clamps with the reference's printed warnings, parameter counting, stack layout.
Mirrors the reference's tests/test_bijectors.py:17-68,300-330, tests/test_density_estimators.py:147-245,
tests/test_conditional_density_estimators.py:15-50 and tests/test_error_formatters.py."""
import numpy as np
import pytest
import torch
from pytest import raises

import torch_nf_amd as tnf
from torch_nf_amd.error_formatters import dbg_check, format_type_err_msg


def test_format_type_err_msg():
    x, s1, s2 = 20, "foo", "bar"
    d = {"x": x}
    assert format_type_err_msg(x, s1, s2, int) == "int argument foo must be int not str."
    assert format_type_err_msg(d, s2, x, str) == "dict argument bar must be str not int."
    assert format_type_err_msg(s1, s2, x, dict) == "str argument bar must be dict not int."
    with raises(ValueError):
        format_type_err_msg(d, s1, s2, str)
    with raises(ValueError):
        format_type_err_msg(d, s1, x, int)


def test_dbg_check(capsys):
    y = torch.normal(0.0, 1.0, (20, 50, 4))
    assert not dbg_check(y, "y")
    y[0, 5, 2] = np.nan
    assert dbg_check(y, "y")
    y[3, 1, 2] = np.inf
    assert dbg_check(y, "y")
    assert "infs 1/4000" in capsys.readouterr().out


def test_bijector_protocol():
    b = tnf.Bijector(4)
    assert b.D == 4 and b.count_num_params() == 0
    with raises(TypeError, match="Bijector argument D must be int not str."):
        tnf.Bijector("foo")
    with raises(TypeError):
        tnf.Bijector(4.0)
    with raises(TypeError):
        tnf.Bijector(True)  # exact-type check: bool is not int
    with raises(ValueError, match="Bijector dimensionality must be positive."):
        tnf.Bijector(-1)
    z, p = np.zeros((3, 4)), np.zeros((5, 6))
    for call in (b, b.forward_and_log_det, b.inverse_and_log_det):
        with raises(NotImplementedError):
            call(z, p)


def test_realnvp_validation(capsys, oracle):
    r = tnf.RealNVP(4, 2, 15, transform_upper=False)
    assert (r.name, r.D, r.num_layers, r.num_units, r.transform_upper) == ("RealNVP", 4, 2, 15, False)
    r = tnf.RealNVP(4, 6, 2000)
    assert r.num_layers == 5 and r.num_units == 1000
    out = capsys.readouterr().out
    assert "Warning: RealNVP.num_layers set to maximum of 5 (received 6)." in out
    assert "Warning: num_units set to maximum of 1,000 (received 2000)." in out
    assert tnf.RealNVP(4, 3, 10).num_units == 15
    assert "Warning: num_units set to minimum of 15 (received 10)." in capsys.readouterr().out
    with raises(TypeError, match="RealNVP argument num_layers must be int not str."):
        tnf.RealNVP(4, "foo", 10)
    with raises(ValueError, match="RealNVP.num_layers must be positive."):
        tnf.RealNVP(4, -1, 10)
    with raises(TypeError):
        tnf.RealNVP(4, 2, "foo")
    with raises(TypeError, match="RealNVP argument transform_upper must be bool not str."):
        tnf.RealNVP(2, 2, 20, "foo")
    for D in (2, 4, 5, 7, 8, 32, 64):
        for L in (1, 2, 5):
            for U in (15, 16, 64):
                for up in (True, False):
                    assert tnf.RealNVP(D, L, U, up).count_num_params() == oracle.coupling_num_params(D, L, U, up)
    assert tnf.RealNVP(64, 2, 15).count_num_params() == 2494


def test_affine_and_batchnorm_validation(capsys):
    a = tnf.Affine(4)
    assert (a.name, a.D, a.count_num_params()) == ("Affine", 4, 8)
    bn = tnf.BatchNorm(4, 0.05, 1e-7)
    assert (bn.name, bn.D, bn.momentum, bn.eps, bn.count_num_params()) == ("BatchNorm", 4, 0.05, 1e-7, 0)
    assert np.isclose(bn.get_last_mean(), np.zeros(4)).all() and bn.get_last_mean().dtype == torch.float32
    assert np.isclose(bn.get_last_alpha(), np.ones(4)).all()
    assert tnf.BatchNorm(4, 1.01).momentum == 1.0
    assert "set to maximum of 1.0" in capsys.readouterr().out
    with raises(TypeError):
        tnf.BatchNorm(4, "foo")
    with raises(ValueError, match="BatchNorm.momentum cannot be negative."):
        tnf.BatchNorm(4, -1.0)
    with raises(TypeError):
        tnf.BatchNorm(4, 0.5, "foo")
    with raises(ValueError, match="BatchNorm.eps cannot be negative."):
        tnf.BatchNorm(4, 0.5, -1.0)


def test_normflow_validation_and_layout(capsys, oracle):
    nf = tnf.NormFlow(4, False, "coupling", 1, 2, 30, None)
    assert (nf.arch_type, nf.num_stages, nf.num_layers, nf.num_units, nf.support_layer) == ("coupling", 1, 2, 30, None)
    assert tnf.NormFlow(4, False, "coupling", 1, 2, 10).num_units == 15
    assert "Warning: NormFlow.num_layers set to minimum of 15 (received 10)." in capsys.readouterr().out
    bad = [
        (TypeError, ("foo", False, "coupling", 1, 2, 20, None)), (ValueError, (-1, False, "coupling", 1, 2, 20, None)),
        (ValueError, (1, False, "coupling", 1, 2, 20, None)),
        (TypeError, (4, False, 1, 1, 2, 20, None)), (ValueError, (4, False, "foo", 1, 2, 20, None)),
        (TypeError, (4, 1, "coupling", 1, 2, 20, None)),
        (TypeError, (4, False, "coupling", "foo", 2, 20, None)), (ValueError, (4, False, "coupling", -1, 2, 20, None)),
        (TypeError, (4, False, "coupling", 1, "foo", 20, None)), (ValueError, (4, False, "coupling", 1, -1, 20, None)),
        (TypeError, (4, False, "coupling", 1, 2, "foo", None)), (ValueError, (4, False, "coupling", 1, 2, -1, None)),
        (TypeError, (4, False, "coupling", 1, 2, 20, "foo")),
    ]
    for exc, args in bad:
        with raises(exc):
            tnf.NormFlow(*args)
    # stack order (reference tests/test_density_estimators.py:213-224)
    nf = tnf.NormFlow(4, True, "coupling", 2, 2, 20)
    kinds = [type(b).__name__ for b in nf.bijectors]
    assert kinds == ["RealNVP", "BatchNorm", "RealNVP", "BatchNorm", "Affine"] * 2
    assert [b.transform_upper for b in nf.bijectors if b.name == "RealNVP"] == [True, False, True, False]
    assert not hasattr(nf, "params")  # conditioner=True flows own no parameters
    # parameter counts
    for (D, S), want in {(2, 1): 1148, (32, 4): 12272, (64, 4): 20464}.items():
        nf = tnf.NormFlow(D, False, "coupling", S, 2, 15, device="cpu")
        assert nf.D_params == want == oracle.flow_num_params(D, S, 2, 15)
        assert nf.params.shape == (1, want) and nf.params.requires_grad and nf.params.is_leaf
    assert type(tnf.NormFlow(4, False, "affine").bijectors[0]).__name__ == "Affine"
    # the reference's default architecture: MAF, BatchNorm, Affine (density_estimator.py:271-274)
    nf = tnf.NormFlow(4, False, num_layers=2, num_units=20, device="cpu")
    assert nf.arch_type == "AR" and [type(b).__name__ for b in nf.bijectors] == ["MAF", "BatchNorm", "Affine"]
    assert nf.D_params == oracle.maf_num_params(4, 2, 20) + 8
    # support layer: appended after the parameterised stack (density_estimator.py:278-282;
    # reference tests/test_density_estimators.py:166-171, 213-224)
    nf = tnf.NormFlow(4, True, "coupling", 2, 2, 20, tnf.ToSimplex(4))
    assert len(nf.bijectors) == 11 and type(nf.bijectors[10]).__name__ == "ToSimplex"
    assert issubclass(type(nf.support_layer), tnf.Bijector)
    assert nf.D_params == oracle.flow_num_params(4, 2, 2, 20)
    with raises(TypeError, match="Support layer not Bijector."):
        tnf.NormFlow(4, True, "coupling", 2, 2, 20, "foo")


def test_support_layer_validation(oracle):
    """ToInterval / ToSimplex constructors (reference tests/test_bijectors.py:203-266, 349-353)."""
    D = 4
    lb, ub = -0.5 * np.array([1.0, np.inf, 1, np.inf]), 0.5 * np.array([1.0, 1.0, np.inf, np.inf])
    iv = tnf.ToInterval(D, lb, ub)
    assert iv.name == "ToInterval" and iv.D == D and iv.count_num_params() == 0 and iv._eps == 1e-12
    ref = oracle.interval_consts(lb, ub)
    for a, b in zip((iv.tanh_flg, iv.softplus_flg, iv.tanh_m, iv.tanh_c, iv.softplus_m, iv.softplus_c), ref):
        assert a.dtype == torch.float32 and a.shape == (1, 1, D) and torch.equal(a, b)
    assert iv._consts.shape == (7, D) and torch.equal(iv._consts[6], torch.log(iv._consts[2]))
    with raises(ValueError, match="Lower and upper bounds must be same length."):
        tnf.ToInterval(D, -np.ones((D,)), np.ones((D + 1,)))
    ub_bad = np.ones((D,))
    ub_bad[3] = -2
    with raises(ValueError, match="Lower bound"):
        tnf.ToInterval(D, -np.ones((D,)), ub_bad)
    with raises(TypeError):
        tnf.ToInterval(D, "[-1,-1,-1,-1]", np.ones((D,)))
    with raises(TypeError):
        tnf.ToInterval(D, -np.ones((D,)), "[1,1,1,1]")
    assert isinstance(tnf.ToInterval(D, [-1, -1, -1, -1], [1, 1, 1, 1]).lb, np.ndarray)
    sx = tnf.ToSimplex(D)
    assert sx.name == "ToSimplex" and sx.D == D and sx.count_num_params() == 0
    with raises(TypeError):  # no inverse in the reference: the base method wants params
        sx.inverse_and_log_det(torch.zeros(1, 1, 3))


def test_maf_validation_and_masks(capsys, oracle):
    """MAF constructor (reference tests/test_bijectors.py:125-160) and mask RNG parity: the same
    np.random state yields the degree vectors / masks of MAF._get_masks (bijectors.py:663-696)."""
    maf = tnf.MAF(4, 2, 20)
    assert (maf.name, maf.D, maf.num_layers, maf.num_units, maf.fwd_fac) == ("MAF", 4, 2, 20, True)
    assert maf.count_num_params() == oracle.maf_num_params(4, 2, 20) == 2 * (2 * 4 * 20 + 400)
    m2 = tnf.MAF(4, 6, 2000)
    assert m2.num_layers == 5 and m2.num_units == 1000
    assert tnf.MAF(4, 1, 3).num_units == 5
    out = capsys.readouterr().out
    assert "Warning: MAF.num_layers set to maximum of 5 (received 6)." in out
    assert "Warning: num_units set to minimum of 15 (received 3)." in out  # the reference's wording
    with raises(TypeError):
        tnf.MAF(4, "foo", 10)
    with raises(ValueError):
        tnf.MAF(4, -1, 10)
    with raises(TypeError):
        tnf.MAF(4, 2, "foo")
    with raises(TypeError, match="MAF argument fwd_fac must be bool not str."):
        tnf.MAF(4, 2, 10, "foo")
    for D, L, U, fwd in [(4, 2, 15, True), (7, 3, 20, True), (5, 1, 9, False)]:
        np.random.seed(7)
        maf = tnf.MAF(D, L, U, fwd_fac=fwd)
        np.random.seed(7)
        ms, Ms = oracle.maf_masks(D, L, U, fwd)
        assert len(maf.ms) == L + 1 and all(np.array_equal(a, b) for a, b in zip(maf.ms, ms))
        assert all(np.array_equal(a[0].numpy(), b) for a, b in zip(maf.Ms, Ms))
        assert [tuple(M.shape) for M in maf.Ms] == [(1, D, U)] + [(1, U, U)] * (L - 1) + [(1, U, D)]
    # autoregressive property of the composed masks: output d depends only on inputs with degree < d
    np.random.seed(1)
    maf = tnf.MAF(6, 2, 30)
    conn = maf.Ms[0][0]
    for M in maf.Ms[1:]:
        conn = conn @ M[0]
    assert torch.equal(conn > 0, torch.triu(conn > 0, diagonal=1)), "strictly upper triangular connectivity"


def test_param_init_matches_reference_rng():
    """Same torch seed -> same xavier_normal_ draw as the reference's _param_init
    (density_estimator.py:352-356): zeros(1, D_params) filled by xavier_normal_ on the host."""
    torch.manual_seed(0)
    nf = tnf.NormFlow(4, False, "coupling", 1, 2, 15, device="cpu")
    torch.manual_seed(0)
    want = torch.nn.init.xavier_normal_(torch.zeros(1, nf.D_params))
    assert torch.equal(nf.params.detach(), want)


def test_cde_validation():
    nf = tnf.NormFlow(4, True, "coupling", 1, 2, 20, None, device="cpu")
    cde = tnf.ConditionalDensityEstimator(nf, 10, [50, 100])
    assert isinstance(cde, torch.nn.Module) and cde.D_params == nf.D_params
    assert list(dict(cde.param_net.named_children())) == ["linear1", "tanh1", "linear2", "relu2", "linear3"]
    assert cde.param_net.linear3.out_features == nf.D_params
    names = list(dict(tnf.ConditionalDensityEstimator(nf, 10, [50, 100], dropout=True).param_net.named_children()))
    assert names == ["linear1", "tanh1", "dropout1", "linear2", "relu2", "dropout2", "linear3"]
    with raises(TypeError):
        tnf.ConditionalDensityEstimator("foo", 10, [50])
    with raises(TypeError):
        tnf.ConditionalDensityEstimator(nf, "foo", [50])
    with raises(ValueError):
        tnf.ConditionalDensityEstimator(nf, 0, [50])
    with raises(TypeError):
        tnf.ConditionalDensityEstimator(nf, 10, "foo")
    with raises(TypeError):
        tnf.ConditionalDensityEstimator(nf, 10, [20, "foo"])
    with raises(ValueError):
        tnf.ConditionalDensityEstimator(nf, 10, [20, -4])
    nf.D_params = 4.0
    with raises(TypeError):
        tnf.ConditionalDensityEstimator(nf, 10, [50])
    nf.D_params = 0
    with raises(ValueError):
        tnf.ConditionalDensityEstimator(nf, 10, [50])

    class Sub(tnf.NormFlow):
        pass

    with raises(TypeError):  # exact type, subclasses rejected (conditional_density_estimator.py:48-49)
        tnf.ConditionalDensityEstimator(Sub(4, True, "coupling", 1, 2, 20, device="cpu"), 10, [50])


def test_no_cpu_fallback():
    """Without a HIP device the compute entry points refuse loudly."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    layer = tnf.RealNVP(4, 2, 15)
    z = torch.zeros(1, 3, 4)
    p = torch.zeros(1, layer.count_num_params())
    with raises(RuntimeError, match="needs a HIP device"):
        layer(z, p)
    with raises(RuntimeError, match="needs a HIP device"):
        tnf.NormFlow(4, False, "coupling", 1, 2, 15).log_prob(z)


def test_install_as_torch_nf():
    import sys

    saved = {k: v for k, v in sys.modules.items() if k == "torch_nf" or k.startswith("torch_nf.")}
    try:
        tnf.install_as_torch_nf()
        import torch_nf.bijectors as rb
        from torch_nf.density_estimator import NormFlow

        assert rb.RealNVP is tnf.RealNVP and NormFlow is tnf.NormFlow
    finally:
        for k in [k for k in sys.modules if k == "torch_nf" or k.startswith("torch_nf.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_shard_bounds():
    from torch_nf_amd.distributed import shard_bounds

    for n in (0, 1, 7, 8, 1 << 20, (1 << 20) + 3):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with raises(ValueError):
        shard_bounds(10, 2, 2)


def test_lfi_host_pieces():
    """CPU-only parts of the LFI stand-ins (torch_nf_amd/systems.py, lfi.py): the Mat simulator, its uniform prior
    and the atom index sampler of the APT loss."""
    from torch_nf_amd.lfi import _atom_indices
    from torch_nf_amd.systems import Mat

    mat = Mat(3)
    assert mat.D == 6 and mat.D_x == 2 and mat.lb.shape == (6,) and (mat.lb < mat.ub).all()
    z = np.array([[1.0, 0.5, 0.0, 2.0, 0.25, 3.0]])  # upper triangle, row-wise
    A = mat.matrices(z)[0]
    assert np.allclose(A, A.T) and np.allclose(np.diag(A), [1.0, 2.0, 3.0]) and A[0, 1] == 0.5 and A[1, 2] == 0.25
    x = mat.simulate(z)
    assert x.shape == (1, 2) and np.isclose(x[0, 0], np.linalg.det(A)) and np.isclose(x[0, 1], 6.0)
    np.random.seed(0)
    zs = mat.sample_prior(100)
    assert zs.shape == (100, 6) and (zs >= mat.lb).all() and (zs <= mat.ub).all()
    lp = mat.log_prior(zs)
    assert np.allclose(lp, -6 * np.log(4.0))
    assert np.isneginf(mat.log_prior(np.full((1, 6), 5.0)))[0]
    lpt = mat.log_prior(torch.tensor(zs, dtype=torch.float32).reshape(10, 10, 6))
    assert lpt.shape == (10, 10) and torch.allclose(lpt, torch.full((10, 10), float(-6 * np.log(4.0))))
    with raises(ValueError):
        Mat(0)
    torch.manual_seed(0)
    atoms = _atom_indices(40, 8, torch.device("cpu"))
    assert atoms.shape == (40, 8) and bool((atoms[:, 0] == torch.arange(40)).all())
    assert all(len(set(r.tolist())) == 8 for r in atoms)
    assert _atom_indices(5, 100, torch.device("cpu")).shape == (5, 5)  # never more atoms than batch rows
