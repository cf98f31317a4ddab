"""MAF / NormFlow(arch_type="AR") on the GPU against the reference's golden vectors (tests/golden/maf.npz:
masks from the reference's own np.random stream, outputs of MAF.forward/inverse_and_log_det, NormFlow('AR')
forward / log_prob and the gradients of -mean(log_prob))."""
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


def test_golden_maf(tnf):
    g = load_golden("maf")
    for ci, (D, L, U, fwd, Mz, Mp, N, dt) in enumerate(g["meta"].tolist()):
        k = "m%02d_" % ci
        np.random.seed(0)
        layer = tnf.MAF(D, L, U, fwd_fac=bool(fwd))
        layer.set_masks([g[k + "ms%d" % i] for i in range(L + 1)])
        z, p = T(g[k + "z"]), T(g[k + "params"])
        zf, ldf = layer.forward_and_log_det(z, p)
        zi, ldi = layer.inverse_and_log_det(z, p)
        tol = dict(rtol=1e-11, atol=1e-11) if dt else dict(rtol=2e-5, atol=5e-6)
        for got, name in ((zf, "z_fwd"), (ldf, "ld_fwd"), (zi, "z_inv"), (ldi, "ld_inv")):
            torch.testing.assert_close(got.cpu(), T(g[k + name], "cpu"), **tol)
        # forward then inverse returns the input (autoregressive inverse is exact after D-1 passes)
        zr, ldr = layer.inverse_and_log_det(zf, p)
        torch.testing.assert_close(zr.cpu(), z.cpu().expand_as(zr), rtol=1e-4, atol=1e-4)


def test_golden_ar_flow(tnf):
    g = load_golden("maf")
    for ci, (D, L, U, N) in enumerate(g["flow_meta"].tolist()):
        k = "n%02d_" % ci
        np.random.seed(0)
        nf = tnf.NormFlow(D, False, "AR", 1, L, U)
        nf.bijectors[0].set_masks([g[k + "ms%d" % i] for i in range(L + 1)])
        nf.params = T(g[k + "params"])
        with torch.no_grad():
            z, lq = nf._forward_from(g[k + "omega"], nf.params, freeze_bn=False)
        torch.testing.assert_close(z.cpu(), T(g[k + "z_fwd"], "cpu"), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(lq.cpu(), T(g[k + "logq_fwd"], "cpu"), rtol=1e-5, atol=1e-4)
        nf.bijectors[1].set_last_stats(T(g[k + "bn_mean"]), T(g[k + "bn_alpha"]))
        with torch.no_grad():
            lp = nf.log_prob(T(g[k + "z_test"]))
        torch.testing.assert_close(lp.cpu(), T(g[k + "log_prob"], "cpu"), rtol=1e-5, atol=1e-5)
        # gradients of -mean(log_prob) w.r.t. params and z (hand-written MAF backward kernel)
        nf.params = T(g[k + "params"]).requires_grad_()
        zt = T(g[k + "z_test"]).requires_grad_()
        loss = -torch.mean(nf.log_prob(zt))
        loss.backward()
        torch.testing.assert_close(loss.detach().cpu(), T(g[k + "loss"], "cpu"), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(nf.params.grad.cpu(), T(g[k + "grad_params"], "cpu"), rtol=2e-4, atol=2e-6)
        torch.testing.assert_close(zt.grad.cpu(), T(g[k + "grad_z"], "cpu"), rtol=2e-4, atol=1e-7)


def test_maf_grad_fp64_and_broadcast(tnf, oracle):
    rng = np.random.RandomState(2)
    for D, L, U, Mz, Mp, N in [(5, 2, 9, 2, 2, 6), (4, 1, 7, 3, 1, 5), (6, 3, 8, 1, 1, 20)]:
        np.random.seed(D)
        layer = tnf.MAF(D, L, U)
        Ms = [M[0].numpy().astype(np.float64) for M in layer.Ms]
        p0 = torch.tensor(rng.normal(0, 0.4, (Mp, layer.count_num_params())))
        z0 = torch.tensor(rng.normal(0, 1, (Mz, N, D)))
        M = max(Mz, Mp)
        wz, wl = torch.tensor(rng.normal(0, 1, (M, N, D))), torch.tensor(rng.normal(0, 1, (M, N)))
        pr, zr = p0.clone().requires_grad_(), z0.clone().requires_grad_()
        zo, ld = oracle.maf(zr, pr, D, L, U, Ms, True)
        ((zo * wz).sum() + (ld * wl).sum()).backward()
        p, z = p0.cuda().requires_grad_(), z0.cuda().requires_grad_()
        zo, ld = layer.inverse_and_log_det(z, p)
        ((zo * wz.cuda()).sum() + (ld * wl.cuda()).sum()).backward()
        torch.testing.assert_close(z.grad.cpu(), zr.grad, rtol=1e-9, atol=1e-9)
        torch.testing.assert_close(p.grad.cpu(), pr.grad, rtol=1e-9, atol=1e-9)
    with pytest.raises(NotImplementedError):  # sampling-direction autograd is not built: loud
        zo, ld = layer.forward_and_log_det(z0.cuda().requires_grad_(), p0.cuda())
        zo.sum().backward()


def test_reference_self_consistency_ar(tnf):
    """The reference's own AR test (tests/test_density_estimators.py:233-238): forward's log_q vs log_prob(z)."""
    np.random.seed(0)
    torch.manual_seed(0)
    nf = tnf.NormFlow(4, False, "AR", num_layers=2, num_units=20)
    assert type(nf.bijectors[0]).__name__ == "MAF"
    with torch.no_grad():
        z, log_q = nf(10)
        log_q_inv = nf.log_prob(z)
    assert z.shape == (1, 10, 4) and log_q.shape == (1, 10)
    assert float(((log_q.cpu().double() - log_q_inv.cpu().double()) ** 2).sum()) < 1e-2
