"""The CPU oracle (oracle/flow_oracle.py) against the golden vectors produced by RUNNING
the reference (oracle/gen_golden.py).  Exact equality was asserted when the vectors were
generated; here the comparison allows a few ulp so it survives a different host CPU /
MKL code path (1e-6 for float32, 1e-13 for float64)."""
import numpy as np
import torch

from conftest import load_golden


def T(a):
    return torch.from_numpy(np.array(a))  # np.array keeps 0-dim shapes


def close(a, b):
    a, b = T(a) if isinstance(a, np.ndarray) else a, T(b) if isinstance(b, np.ndarray) else b
    assert a.dtype == b.dtype and a.shape == b.shape, (a.dtype, b.dtype, a.shape, b.shape)
    tol = 1e-13 if a.dtype == torch.float64 else 2e-6
    torch.testing.assert_close(a, b, rtol=tol, atol=tol)


def test_coupling_cases(oracle):
    g = load_golden("coupling")
    for ci, (D, L, U, upper, Mz, Mp, N, dt, extra) in enumerate(g["meta"].tolist()):
        k = "c%02d_" % ci
        z, p = T(g[k + "z"]), T(g[k + "params"])
        assert p.shape[1] == oracle.coupling_num_params(D, L, U, bool(upper)) + extra
        zf, ldf = oracle.coupling(z, p, D, L, U, bool(upper), False)
        zi, ldi = oracle.coupling(z, p, D, L, U, bool(upper), True)
        close(zf, g[k + "z_fwd"]); close(ldf, g[k + "ld_fwd"])
        close(zi, g[k + "z_inv"]); close(ldi, g[k + "ld_inv"])
        # pass-through half is bit-identical (reference tests/test_bijectors.py:82-83)
        h = D // 2
        sl = slice(0, h) if upper else slice(h, D)
        assert torch.equal(zf[:, :, sl], z[:, :, sl].expand(zf.shape[0], -1, -1))


def test_affine_and_bn_cases(oracle):
    g = load_golden("affine_bn")
    for ci, (D, Mz, Mp, N, dt, extra) in enumerate(g["affine_meta"].tolist()):
        k = "a%02d_" % ci
        z, p = T(g[k + "z"]), T(g[k + "params"])
        zf, ldf = oracle.affine(z, p, D, False)
        zi, ldi = oracle.affine(z, p, D, True)
        close(zf, g[k + "z_fwd"]); close(ldf, g[k + "ld_fwd"])
        close(zi, g[k + "z_inv"]); close(ldi, g[k + "ld_inv"])
        # the reference's only known-answer test (tests/test_bijectors.py:286-295)
        ref = z * torch.exp(p[:, None, :D]) + p[:, None, D:2 * D]
        assert float(((zf - ref) ** 2).sum()) < 1e-10
    for ci, (D, M, N) in enumerate(g["bn_meta"].tolist()):
        k = "b%02d_" % ci
        zb, ldb, mean, alpha = oracle.bn_forward_batch(T(g[k + "z"]))
        torch.testing.assert_close(zb, T(g[k + "z_batch"]), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(mean, T(g[k + "mean"]), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(alpha, T(g[k + "alpha"]), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(ldb, T(g[k + "ld_batch"]), rtol=1e-5, atol=1e-5)
        m, a = T(g[k + "mean"]), T(g[k + "alpha"])
        zf, ldf = oracle.bn_forward_frozen(T(g[k + "z2"]), m, a)
        zi, ldi = oracle.bn_inverse(T(g[k + "z2"]), m, a)
        close(zf, g[k + "z_frozen"]); close(ldf, g[k + "ld_frozen"])
        close(zi, g[k + "z_inv"]); close(ldi, g[k + "ld_inv"])


def test_flow_cases(oracle):
    g = load_golden("flow")
    for ci, (D, S, L, U, N) in enumerate(g["meta"].tolist()):
        k = "f%02d_" % ci
        p = T(g[k + "params"])
        assert p.shape[1] == oracle.flow_num_params(D, S, L, U)
        stats = [(T(m), T(a)) for m, a in zip(g[k + "bn_mean"], g[k + "bn_alpha"])]
        z0, sld = oracle.flow_inverse(T(g[k + "z_test"]), p, D, S, L, U, stats)
        lp = oracle.flow_log_prob(T(g[k + "z_test"]), p, D, S, L, U, stats)
        torch.testing.assert_close(z0, T(g[k + "z0"]), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(sld, T(g[k + "sum_log_det"]), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(lp, T(g[k + "log_prob"]), rtol=1e-6, atol=1e-5)
        z, lq, _ = oracle.flow_forward(g[k + "omega_fz"], p, D, S, L, U, stats)
        torch.testing.assert_close(z, T(g[k + "z_fz"]), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(lq, T(g[k + "logq_fz"]), rtol=1e-6, atol=1e-5)
        assert lq.dtype == torch.float64 and z.dtype == torch.float32  # density_estimator.py:367-372
        z, lq, st2 = oracle.flow_forward(g[k + "omega"], p, D, S, L, U, None)
        torch.testing.assert_close(z, T(g[k + "z_fwd"]), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(lq, T(g[k + "logq_fwd"]), rtol=1e-5, atol=1e-4)


def test_flow_gradients(oracle):
    g = load_golden("flow")
    metas = g["meta"].tolist()
    ci = [i for i, m in enumerate(metas) if "f%02d_grad_params" % i in g][0]
    D, S, L, U, N = metas[ci]
    k = "f%02d_" % ci
    stats = [(T(m), T(a)) for m, a in zip(g[k + "bn_mean"], g[k + "bn_alpha"])]
    p = T(g[k + "params"]).clone().requires_grad_()
    z = T(g[k + "z_test"]).clone().requires_grad_()
    loss = -torch.mean(oracle.flow_log_prob(z, p, D, S, L, U, stats))
    loss.backward()
    torch.testing.assert_close(loss.detach(), T(g[k + "loss"]), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(p.grad, T(g[k + "grad_params"]), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(z.grad, T(g[k + "grad_z"]), rtol=1e-4, atol=1e-7)


def test_maf_cases(oracle):
    g = load_golden("maf")
    for ci, (D, L, U, fwd, Mz, Mp, N, dt) in enumerate(g["meta"].tolist()):
        k = "m%02d_" % ci
        ms = [g[k + "ms%d" % i] for i in range(L + 1)]
        # masks follow from the stored degree vectors exactly as in MAF._get_masks
        m_prev = np.arange(1, D + 1) if fwd else np.arange(D, -1, -1)
        Ms, k_prev = [], D
        for m in ms[:-1]:
            Ms.append((m_prev[:k_prev, None] <= m[None, :]).astype(np.float32))
            k_prev, m_prev = len(m), m
        Ms.append((m_prev[:k_prev, None] < ms[-1][None, :D]).astype(np.float32))
        z, p = T(g[k + "z"]), T(g[k + "params"])
        Msd = [M.astype(np.float64) for M in Ms] if dt else Ms
        zf, ldf = oracle.maf(z, p, D, L, U, Msd, False)
        zi, ldi = oracle.maf(z, p, D, L, U, Msd, True)
        close(zf, g[k + "z_fwd"]); close(ldf, g[k + "ld_fwd"])
        close(zi, g[k + "z_inv"]); close(ldi, g[k + "ld_inv"])
    for ci, (D, L, U, N) in enumerate(g["flow_meta"].tolist()):
        k = "n%02d_" % ci
        ms = [g[k + "ms%d" % i] for i in range(L + 1)]
        Ms, k_prev, m_prev = [], D, np.arange(1, D + 1)
        for m in ms[:-1]:
            Ms.append((m_prev[:k_prev, None] <= m[None, :]).astype(np.float32))
            k_prev, m_prev = len(m), m
        Ms.append((m_prev[:k_prev, None] < ms[-1][None, :D]).astype(np.float32))
        stat = (T(g[k + "bn_mean"]), T(g[k + "bn_alpha"]))
        lp = oracle.ar_flow_log_prob(T(g[k + "z_test"]), T(g[k + "params"]), D, L, U, Ms, stat)
        torch.testing.assert_close(lp, T(g[k + "log_prob"]), rtol=1e-6, atol=1e-5)


def test_support_cases(oracle):
    g = load_golden("support")
    for ci, (D, M, N, dt) in enumerate(g["interval_meta"].tolist()):
        k = "i%02d_" % ci
        consts = oracle.interval_consts(g[k + "lb"], g[k + "ub"])
        zf, ldf = oracle.to_interval(T(g[k + "z"]), consts, False)
        zi, ldi = oracle.to_interval(T(g[k + "z_fwd"]), consts, True)
        close(zf, g[k + "z_fwd"]); close(ldf, g[k + "ld_fwd"])
        close(zi, g[k + "z_inv"]); close(ldi, g[k + "ld_inv"])
    for ci, (Din, Dattr, M, N, dt) in enumerate(g["simplex_meta"].tolist()):
        k = "s%02d_" % ci
        zf, ldf = oracle.to_simplex(T(g[k + "z"]), Dattr)
        close(zf, g[k + "z_fwd"]); close(ldf, g[k + "ld_fwd"])


def test_cde_cases(oracle):
    g = load_golden("cde")
    for ci, row in enumerate(g["meta"].tolist()):
        D, S, L, U, D_x, nh, M, N = row[:8]
        k = "d%02d_" % ci
        p = T(g[k + "params"])
        stats = [(T(m), T(a)) for m, a in zip(g[k + "bn_mean"], g[k + "bn_alpha"])]
        lp = oracle.flow_log_prob(T(g[k + "z_test"]), p, D, S, L, U, stats)
        torch.testing.assert_close(lp, T(g[k + "log_prob"]), rtol=1e-6, atol=1e-5)
