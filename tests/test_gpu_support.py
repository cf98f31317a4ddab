"""ToInterval / ToSimplex kernels and NormFlow(support_layer=...) on the GPU against the reference's
outputs (tests/golden/support.npz, written by oracle/gen_golden.py:gen_support)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tnf():
    import torch_nf_amd

    assert torch.cuda.is_available()
    return torch_nf_amd


def T(a, dev="cuda"):
    return torch.from_numpy(np.array(a)).to(dev)


def tol(dt):
    # float32: tanhf/expf/log1pf differ from the host's vector math by a few ulp, and the log-det term
    # log(1 - tanh(z)^2 + eps) amplifies that where tanh saturates; float64 is tight
    return dict(rtol=1e-11, atol=1e-11) if dt else dict(rtol=3e-5, atol=3e-5)


def test_golden_to_interval(tnf):
    g = load_golden("support")
    for ci, (D, M, N, dt) in enumerate(g["interval_meta"].tolist()):
        k = "i%02d_" % ci
        layer = tnf.ToInterval(D, g[k + "lb"], g[k + "ub"])
        z = T(g[k + "z"])
        zf, ldf = layer(z)
        zi, ldi = layer.inverse_and_log_det(T(g[k + "z_fwd"]))
        assert zf.dtype == z.dtype and ldf.shape == (M, N)
        torch.testing.assert_close(zf.cpu(), T(g[k + "z_fwd"], "cpu"), **tol(dt))
        torch.testing.assert_close(ldf.cpu(), T(g[k + "ld_fwd"], "cpu"), **tol(dt))
        # the inverse recovers z up to the conditioning of atanh / log(exp - 1) at the data's distance to
        # the bounds; compare with the reference's own inverse loosely in float32 and with z itself
        itol = dict(rtol=1e-9, atol=1e-9) if dt else dict(rtol=2e-3, atol=2e-3)
        torch.testing.assert_close(zi.cpu(), T(g[k + "z_inv"], "cpu"), **itol)
        torch.testing.assert_close(ldi.cpu(), T(g[k + "ld_inv"], "cpu"), **itol)
        # gradients, both directions
        wz, wl = T(g[k + "wz"]), T(g[k + "wl"])
        for inv, x, key in ((False, z, "g_fwd"), (True, T(g[k + "z_fwd"]), "g_inv")):
            xr = x.clone().requires_grad_()
            o, l = layer.inverse_and_log_det(xr) if inv else layer(xr)
            ((o * wz).sum() + (l * wl).sum()).backward()
            gt = dict(rtol=1e-8, atol=1e-8) if dt else dict(rtol=5e-3, atol=5e-3) if inv else dict(rtol=1e-4, atol=1e-4)
            torch.testing.assert_close(xr.grad.cpu(), T(g[k + key], "cpu"), **gt)


def test_to_interval_reference_test(tnf):
    """The reference's own test (tests/test_bijectors.py:203-233): float64 round trips."""
    D, M, N = 4, 20, 50
    rng = np.random.RandomState(0)
    iv = tnf.ToInterval(D, float("-inf") * np.ones((D,)), float("inf") * np.ones((D,)))
    z_in = torch.tensor(rng.normal(0.0, 1.0, (M, N, D))).cuda()
    z, log_det = iv(z_in)
    z_inv, log_det_inv = iv.inverse_and_log_det(z)
    assert float(((z_in - z) ** 2).sum()) < 1e-10 and float(((z_in - z_inv) ** 2).sum()) < 1e-10
    assert float(((log_det - log_det_inv) ** 2).sum()) < 1e-10
    b = 0.5
    iv = tnf.ToInterval(D, -b * np.array([1.0, np.inf, 1, np.inf]), b * np.array([1.0, 1.0, np.inf, np.inf]))
    z_in = torch.tensor(rng.normal(0.0, 2.0, (M, N, D))).cuda()
    z, log_det = iv(z_in)
    assert (z[:, :, 0] > -1).all() and (z[:, :, 0] < 1).all() and (z[:, :, 1] < 1).all() and (z[:, :, 2] > -1).all()
    z_inv, log_det_inv = iv.inverse_and_log_det(z)
    assert float(((z_in - z_inv) ** 2).sum()) < 1e-4
    assert float(((log_det - log_det_inv) ** 2).sum()) < 1e-4


def test_golden_to_simplex(tnf):
    g = load_golden("support")
    for ci, (Din, Dattr, M, N, dt) in enumerate(g["simplex_meta"].tolist()):
        k = "s%02d_" % ci
        layer = tnf.ToSimplex(Dattr)
        zr = T(g[k + "z"]).requires_grad_()
        zf, ldf = layer(zr)
        assert zf.shape == (M, N, Din + 1)
        torch.testing.assert_close(zf.detach().cpu(), T(g[k + "z_fwd"], "cpu"), **tol(dt))
        torch.testing.assert_close(ldf.detach().cpu(), T(g[k + "ld_fwd"], "cpu"), **tol(dt))
        torch.testing.assert_close(zf.detach().sum(2).cpu(), torch.ones(M, N, dtype=zf.dtype), rtol=1e-5, atol=1e-5)
        ((zf * T(g[k + "wz"])).sum() + (ldf * T(g[k + "wl"])).sum()).backward()
        gt = dict(rtol=1e-9, atol=1e-9) if dt else dict(rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(zr.grad.cpu(), T(g[k + "g_fwd"], "cpu"), **gt)


def test_golden_flow_with_support(tnf):
    g = load_golden("support")
    for ci, (D, S, L, U, N, kind) in enumerate(g["flow_meta"].tolist()):
        k = "f%02d_" % ci
        sup = tnf.ToSimplex(D) if kind else tnf.ToInterval(D, g[k + "lb"], g[k + "ub"])
        nf = tnf.NormFlow(D, False, "coupling", S, L, U, sup)
        nf.params = T(g[k + "params"])
        with torch.no_grad():
            z, lq = nf._forward_from(g[k + "omega"], nf.params, freeze_bn=False)
        torch.testing.assert_close(z.cpu(), T(g[k + "z_fwd"], "cpu"), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(lq.cpu(), T(g[k + "logq_fwd"], "cpu"), rtol=1e-5, atol=2e-4)
        for b, m, a in zip(nf._bn_layers(), g[k + "bn_mean"], g[k + "bn_alpha"]):
            b.set_last_stats(T(m), T(a))
        with torch.no_grad():
            z2, lq2 = nf._forward_from(g[k + "omega_frozen"], nf.params, freeze_bn=True)
        torch.testing.assert_close(z2.cpu(), T(g[k + "z_frozen"], "cpu"), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(lq2.cpu(), T(g[k + "logq_frozen"], "cpu"), rtol=1e-5, atol=2e-4)
        if kind:
            with pytest.raises(TypeError):  # ToSimplex has no inverse (as in the reference)
                nf.log_prob(z)
            continue
        zt = T(g[k + "z_fwd"])
        with torch.no_grad():
            lp = nf.log_prob(zt)
        torch.testing.assert_close(lp.cpu(), T(g[k + "log_prob"], "cpu"), rtol=1e-4, atol=2e-3)
        nf.params = T(g[k + "params"]).requires_grad_()
        loss = -torch.mean(nf.log_prob(zt))
        loss.backward()
        torch.testing.assert_close(loss.detach().cpu(), T(g[k + "loss"], "cpu"), rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(nf.params.grad.cpu(), T(g[k + "grad_params"], "cpu"), rtol=2e-3, atol=2e-4)


def test_support_edge_cases(tnf):
    iv = tnf.ToInterval(3, [-1.0, -np.inf, 0.0], [1.0, np.inf, np.inf])
    z, ld = iv(torch.zeros(2, 0, 3, device="cuda"))
    assert z.shape == (2, 0, 3) and ld.shape == (2, 0)
    # CPU tensors are staged and come back on the CPU
    z, ld = iv(torch.zeros(1, 5, 3))
    assert z.device.type == "cpu" and torch.allclose(z[0, 0], torch.tensor([0.0, 0.0, float(np.log(2.0))]))
    with pytest.raises(ValueError):
        iv(torch.zeros(1, 5, 4, device="cuda"))
    # softplus threshold branch (z > 20) and saturated tanh stay finite
    z, ld = iv(torch.tensor([[[30.0, 30.0, 30.0], [-30.0, -30.0, -30.0]]], device="cuda"))
    assert torch.isfinite(z).all() and torch.isfinite(ld).all()
    assert float(z[0, 0, 2]) == 30.0


def test_fused_support_ar(tnf):
    """NormFlow('AR', support_layer=ToInterval): log_prob and the frozen forward as ONE kernel (ToInterval in the
    kernel's load / store stage, hardware transcendentals) against the composition of the standalone kernels
    (libm) of the same package."""
    for D, L, U, M, N in [(6, 2, 15, 1, 700), (21, 2, 42, 3, 130), (4, 1, 20, 2, 64)]:
        np.random.seed(D)
        torch.manual_seed(D)
        lb = np.where(np.arange(D) % 3 == 0, -np.inf, -2.0 - np.arange(D) / 8.0)
        ub = np.where(np.arange(D) % 3 == 1, np.inf, 2.5 + np.arange(D) / 4.0)
        nf = tnf.NormFlow(D, True, "AR", 1, L, U, tnf.ToInterval(D, lb, ub))
        g = torch.Generator().manual_seed(D)
        nf.bijectors[1].set_last_stats(torch.randn(D, generator=g) * 0.1, torch.rand(D, generator=g) * 0.5 + 0.75)
        # moderate magnitudes: where tanh saturates in float32 (|z| > 8) log(1 - tanh^2 + eps) is -16 or -28
        # depending on the last ulp of tanh, in the reference as much as here
        params = (torch.randn(M, nf.D_params, generator=g) * 0.1).cuda()
        omega = torch.randn(M, N, D, generator=g).clamp_(-2.5, 2.5).cuda()
        with torch.no_grad():
            z, lq = nf._forward_from(omega, params, freeze_bn=True)          # fused: MAF, BN, Affine, ToInterval
            lp = nf.log_prob(z, params)                                      # fused: ToInterval^-1, ..., base density
        p2 = params.clone().requires_grad_()                                 # autograd mode -> per-bijector kernels
        z2, lq2 = nf._forward_from(omega, p2, freeze_bn=True)
        lp2 = nf.log_prob(z2.detach(), p2)
        tol = 2e-4 * max(1, D // 8)
        torch.testing.assert_close(z, z2.detach(), rtol=tol, atol=tol)
        torch.testing.assert_close(lq, lq2.detach(), rtol=1e-5, atol=10 * tol)
        torch.testing.assert_close(lp, lp2.detach(), rtol=1e-4, atol=20 * tol)
        # forward / log_prob consistency through the support layer
        assert float((lq.float() - lp).abs().max()) < 5e-2


def test_fused_support_coupling(tnf):
    """NormFlow('coupling', support_layer=ToInterval) with the whole-flow kernel: ToInterval runs in the kernel's
    load (log_prob) / store (frozen forward) stage; compared with the standalone-kernel composition."""
    for D, S, M, N in [(64, 4, 1, 3000), (32, 2, 2, 257)]:
        np.random.seed(D)
        torch.manual_seed(D)
        lb = np.where(np.arange(D) % 3 == 0, -np.inf, -3.0 - np.arange(D) / 16.0)
        ub = np.where(np.arange(D) % 3 == 1, np.inf, 3.5 + np.arange(D) / 8.0)
        nf = tnf.NormFlow(D, True, "coupling", S, 2, 15, tnf.ToInterval(D, lb, ub))
        g = torch.Generator().manual_seed(D)
        for b in nf._bn_layers():
            b.set_last_stats(torch.randn(D, generator=g) * 0.05, torch.rand(D, generator=g) * 0.2 + 0.9)
        params = (torch.randn(M, nf.D_params, generator=g) * 0.05).cuda()
        omega = torch.randn(M, N, D, generator=g).clamp_(-2.5, 2.5).cuda()
        with torch.no_grad():
            assert nf._whole_flow() and nf._fused_support() is not None
            z, lq = nf._forward_from(omega, params, freeze_bn=True)
            lp = nf.log_prob(z, params)
            nf.fusion = tnf._lib.FUSE_LAYER  # per-layer chain: the support layer runs as its own kernel
            z2, lq2 = nf._forward_from(omega, params, freeze_bn=True)
            lp2 = nf.log_prob(z2, params)
            nf.fusion = tnf._lib.FUSE_AUTO
        torch.testing.assert_close(z, z2, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(lq, lq2, rtol=1e-5, atol=2e-3)
        torch.testing.assert_close(lp, lp2, rtol=1e-5, atol=5e-3)
        assert float((lq.float() - lp).abs().max()) < 5e-2
