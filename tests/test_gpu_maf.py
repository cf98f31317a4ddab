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
    # (the sampling direction differentiates too: test_maf_sampling_direction_backward)


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


@pytest.mark.parametrize("D,L,U,Mz,Mp,N", [
    (4, 2, 20, 1, 1, 1000), (5, 1, 15, 2, 2, 77), (7, 3, 33, 1, 3, 50), (16, 2, 32, 1, 1, 257),
    (2, 2, 15, 3, 1, 19), (64, 2, 64, 1, 1, 130), (33, 2, 40, 2, 2, 31), (12, 5, 17, 1, 1, 64),
])
def test_maf_mfma_vs_oracle(tnf, oracle, D, L, U, Mz, Mp, N):
    """The matrix-pipe MAF kernel (float32, any D <= 64 incl. D % 4 != 0, per-context parameter rows) against the
    CPU oracle in both directions, and against the shape-generic kernel (TNF_OPT_FORCE_GENERIC)."""
    from torch_nf_amd import _lib

    np.random.seed(D * 100 + U)
    layer = tnf.MAF(D, L, U)
    L, U = layer.num_layers, layer.num_units
    rng = np.random.RandomState(3)
    # weight scale ~ 1/sqrt(fan-in) keeps the pre-activations O(1): z' = (z - mu) e^-alpha is ill-conditioned otherwise
    params = torch.tensor(rng.normal(0, 0.4 / np.sqrt(max(1.0, U / 16.0)), (Mp, layer.count_num_params())), dtype=torch.float32)
    z = torch.tensor(rng.normal(0, 1, (Mz, N, D)), dtype=torch.float32)
    Ms = [M[0].numpy() for M in layer.Ms]
    assert bool(_lib.lib.tnf_ar_flow_supported(D, L, U))
    for inverse in (True, False):
        ref_z, ref_ld = oracle.maf(z, params, D, L, U, Ms, inverse)
        fn = layer.inverse_and_log_det if inverse else layer.forward_and_log_det
        got_z, got_ld = fn(z.cuda(), params.cuda())
        _lib.lib.tnf_set_option(_lib.OPT_FORCE_GENERIC, 1)
        try:
            gen_z, gen_ld = fn(z.cuda(), params.cuda())
        finally:
            _lib.lib.tnf_set_option(_lib.OPT_FORCE_GENERIC, 0)
        # the sampling direction iterates the nets D-1 times: errors compound with D
        tol = dict(rtol=2e-5, atol=2e-5) if inverse else dict(rtol=1e-4 * max(1, D // 8), atol=1e-4 * max(1, D // 8))
        torch.testing.assert_close(got_z.cpu(), ref_z, **tol)
        torch.testing.assert_close(got_ld.cpu(), ref_ld, **tol)
        torch.testing.assert_close(got_z, gen_z, **tol)
        torch.testing.assert_close(got_ld, gen_ld, **tol)


def test_ar_flow_fused_paths(tnf, oracle):
    """NormFlow('AR') as ONE kernel (tnf_ar_flow_log_prob_f32 / tnf_ar_flow_forward_f32) vs the oracle and vs the
    per-bijector composition; per-context parameter rows; N = 2^18 consistency of forward and log_prob."""
    for D, L, U, M, N in [(4, 2, 20, 1, 500), (10, 2, 32, 3, 65), (64, 1, 64, 1, 100)]:
        np.random.seed(D)
        torch.manual_seed(D)
        nf = tnf.NormFlow(D, True, "AR", 1, L, U)
        g = torch.Generator().manual_seed(D)
        nf.bijectors[1].set_last_stats(torch.randn(D, generator=g) * 0.1, torch.rand(D, generator=g) * 0.5 + 0.75)
        params = (torch.randn(M, nf.D_params, generator=g) * 0.3).cuda()
        z = torch.randn(M, N, D, generator=g).cuda()
        Ms = [Mk[0].numpy() for Mk in nf.bijectors[0].Ms]
        stat = (nf.bijectors[1].get_last_mean().cpu().float(), nf.bijectors[1].get_last_alpha().cpu().float())
        with torch.no_grad():
            assert nf._ar_fused_ok(z, params)
            lp = nf.log_prob(z, params)
            z0, sld = nf.inverse_and_log_det(z, params)
            zf, lq = nf._forward_from(z, params, freeze_bn=True)
        ref_lp = oracle.ar_flow_log_prob(z.cpu(), params.cpu(), D, L, U, Ms, stat)
        torch.testing.assert_close(lp.cpu(), ref_lp, rtol=1e-5, atol=2e-5)
        ref = -0.5 * (z0 ** 2).sum(2) - D * np.log(np.sqrt(2 * np.pi)) - sld
        torch.testing.assert_close(lp, ref, rtol=1e-5, atol=2e-5)
        # frozen forward vs the per-bijector composition of the same package (autograd mode switches paths)
        p2 = params.clone().requires_grad_()
        zf2, lq2 = nf._forward_from(z, p2, freeze_bn=True)
        tol = 1e-4 * max(1, D // 8)
        torch.testing.assert_close(zf, zf2.detach(), rtol=tol, atol=tol)
        torch.testing.assert_close(lq, lq2.detach(), rtol=1e-5, atol=tol)
    # forward -> log_prob consistency at N = 2^18 (the reference's own AR check, size-independent)
    np.random.seed(0)
    nf = tnf.NormFlow(8, False, "AR", 1, 2, 32)
    with torch.no_grad():
        nf(64)  # sets BatchNorm statistics
        z, log_q = nf.sample(1 << 18)
        lp = nf.log_prob(z)
    assert float((log_q.float() - lp).abs().max()) < 5e-3


@pytest.mark.parametrize("D,L,U,Mz,Mp,N", [
    (4, 2, 20, 1, 1, 300), (6, 2, 15, 5, 5, 100), (7, 3, 33, 3, 3, 37), (21, 2, 42, 4, 4, 50),
    (16, 1, 16, 1, 1, 64), (3, 2, 64, 2, 2, 19), (32, 2, 32, 1, 1, 130), (5, 3, 17, 3, 1, 21),
])
def test_maf_backward_mfma(tnf, oracle, D, L, U, Mz, Mp, N):
    """Matrix-pipe backward of MAF.inverse_and_log_det (float32; shared and per-context parameter rows, D % 4 != 0,
    1..3 layers) against torch autograd through the CPU oracle and against the shape-generic backward kernel."""
    from torch_nf_amd import _lib

    np.random.seed(D * 10 + U)
    layer = tnf.MAF(D, L, U)
    L, U = layer.num_layers, layer.num_units
    rng = np.random.RandomState(4)
    p0 = torch.tensor(rng.normal(0, 0.4 / np.sqrt(max(1.0, U / 16.0)), (Mp, layer.count_num_params())), dtype=torch.float32)
    z0 = torch.tensor(rng.normal(0, 1, (Mz, N, D)), dtype=torch.float32)
    M = max(Mz, Mp)
    wz = torch.tensor(rng.normal(0, 1, (M, N, D)), dtype=torch.float32)
    wl = torch.tensor(rng.normal(0, 1, (M, N)), dtype=torch.float32)
    Ms = [Mk[0].numpy() for Mk in layer.Ms]
    pr, zr = p0.clone().requires_grad_(), z0.clone().requires_grad_()
    zo, ld = oracle.maf(zr, pr, D, L, U, Ms, True)
    ((zo * wz).sum() + (ld * wl).sum()).backward()
    grads = []
    for generic in (0, 1):
        before = [_lib.lib.tnf_diag_launch_count(f) for f in (_lib.DIAG_MAF_BWD_MFMA, _lib.DIAG_MAF_BWD_GENERIC)]
        _lib.lib.tnf_set_option(_lib.OPT_FORCE_GENERIC, generic)
        try:
            p, z = p0.cuda().requires_grad_(), z0.cuda().requires_grad_()
            zo, ld = layer.inverse_and_log_det(z, p)
            ((zo * wz.cuda()).sum() + (ld * wl.cuda()).sum()).backward()
            grads.append((z.grad.cpu(), p.grad.cpu()))
        finally:
            _lib.lib.tnf_set_option(_lib.OPT_FORCE_GENERIC, 0)
        # backward runs on autograd's thread; the Function re-enters the forward's options there -- prove it
        ran = [_lib.lib.tnf_diag_launch_count(f) - b for f, b in
               zip((_lib.DIAG_MAF_BWD_MFMA, _lib.DIAG_MAF_BWD_GENERIC), before)]
        assert ran[generic] == 1 and ran[1 - generic] == 0, (generic, ran)

    def close(a, b, tol):
        scale = float(b.abs().max().clamp_min(1e-30))
        assert float((a - b).abs().max()) <= tol * scale, (float((a - b).abs().max()), scale)

    for gz, gp in grads:
        close(gz, zr.grad, 3e-5)
        close(gp, pr.grad, 3e-5)
    # masked weights get exactly zero gradient (autograd through Ms * W)
    mflat = torch.cat([torch.tensor(Mk).reshape(-1) for Mk in Ms])
    idx, off = [], 0
    for Mk in Ms:
        n = Mk.size
        idx += [off + i for i in range(2 * n) if Mk.reshape(-1)[i % n] == 0]
        off += 2 * n
    assert float(grads[0][1][:, idx].abs().max()) == 0.0 if idx else True


@pytest.mark.parametrize("D,L,U,M,Mp,N,support", [(6, 2, 15, 40, 40, 100, True), (6, 2, 15, 3, 1, 257, True),
                                                   (21, 2, 42, 8, 8, 50, True), (16, 3, 32, 2, 2, 64, False),
                                                   (5, 1, 20, 1, 1, 33, False)])
def test_ar_flow_training_one_kernel_backward(tnf, oracle, D, L, U, M, Mp, N, support):
    """NormFlow('AR').log_prob with z constant and params requiring grad: forward = the one-kernel inference path,
    backward = tnf_ar_flow_log_prob_bwd_f32 (ToInterval^-1, folded Affine / BatchNorm, MAF recompute + backward,
    base density in one kernel).  Gradients w.r.t. the parameter rows against (a) this package's per-bijector
    autograd path and (b) torch autograd over the oracle (= the reference's training path)."""
    rng = np.random.RandomState(D + M + N)
    torch.manual_seed(D + M)
    lb, ub = -2.0 * np.ones(D), 2.0 * np.ones(D)
    lb[::2] = -np.inf
    sup = tnf.ToInterval(D, lb, ub) if support else None
    nf = tnf.NormFlow(D, True, "AR", 1, L, U, sup)
    nf.bijectors[1].set_last_stats(torch.tensor(rng.normal(0, 0.3, D)).float(),
                                   torch.tensor(np.exp(rng.normal(0, 0.2, D))).float())
    p0 = torch.tensor(rng.normal(0, 0.2, (Mp, nf.D_params))).float()
    z = torch.tensor(rng.uniform(-1.5, 1.5, (M, N, D))).float().cuda()
    w = torch.tensor(rng.uniform(0.2, 1.0, (M, N))).float().cuda()
    res = {}
    for fused in (True, False):
        nf.fused_ar_training = fused
        p = p0.clone().cuda().requires_grad_()
        assert nf._ar_train_ok(z, p) == fused
        loss = -(nf.log_prob(z, p) * w).sum() / N
        loss.backward()
        res[fused] = (loss.detach().cpu(), p.grad.cpu())
    # (b) oracle
    Ms = [Mk[0].numpy() for Mk in nf.bijectors[0].Ms]
    stat = (nf.bijectors[1].get_last_mean().cpu().float(), nf.bijectors[1].get_last_alpha().cpu().float())
    pr = p0.clone().requires_grad_()
    zc = z.cpu()
    if support:
        zi, ld = oracle.to_interval(zc, oracle.interval_consts(lb, ub), True)
    else:
        zi, ld = zc, 0.0
    lp = oracle.ar_flow_log_prob(zi, pr, D, nf.num_layers, nf.num_units, Ms, stat) - ld
    loss_o = -(lp * w.cpu()).sum() / N
    loss_o.backward()
    scale = float(pr.grad.abs().max())
    torch.testing.assert_close(res[True][0], loss_o.detach(), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(res[True][1], res[False][1], rtol=2e-4, atol=2e-5 * scale)
    torch.testing.assert_close(res[True][1], pr.grad, rtol=5e-4, atol=5e-5 * scale)


@pytest.mark.parametrize("D,L,U,M,Mp,N,dt", [(6, 2, 15, 3, 3, 40, torch.float32), (5, 1, 20, 2, 1, 33, torch.float32),
                                              (8, 3, 16, 1, 1, 64, torch.float64), (16, 2, 32, 2, 2, 50, torch.float32)])
def test_maf_sampling_direction_backward(tnf, oracle, D, L, U, M, Mp, N, dt):
    """Autograd through MAF.forward_and_log_det (the reference differentiates its D - 1 passes,
    bijectors.py:752-754): here the implicit-function backward of ops._MafFn -- D + 1 launches of the
    inverse-direction backward kernel around the per-dimension alpha of tnf_maf_inverse_alpha.  Gradients w.r.t.
    the parameter rows and the base draw against torch autograd through the oracle's iterated forward."""
    rng = np.random.RandomState(D * 10 + L)
    np.random.seed(D + L)
    maf = tnf.MAF(D, L, U)
    Ms = [Mk[0].numpy() for Mk in maf.Ms]
    n = maf.count_num_params()
    p0 = torch.tensor(rng.normal(0, 0.3, (Mp, n))).to(dt)
    om0 = torch.tensor(rng.normal(0, 1, (M, N, D))).to(dt)
    wz = torch.tensor(rng.normal(0, 1, (M, N, D))).to(dt)
    wl = torch.tensor(rng.normal(0, 1, (M, N))).to(dt)
    p, om = p0.clone().cuda().requires_grad_(), om0.clone().cuda().requires_grad_()
    z, ld = maf(om, p)
    ((z * wz.cuda()).sum() + (ld * wl.cuda()).sum()).backward()
    pr, omr = p0.clone().requires_grad_(), om0.clone().requires_grad_()
    zr, ldr = oracle.maf(omr, pr, D, maf.num_layers, maf.num_units, Ms, False)
    ((zr * wz).sum() + (ldr * wl).sum()).backward()
    tol = dict(rtol=1e-9, atol=1e-9) if dt == torch.float64 else dict(rtol=2e-3, atol=2e-4 * float(pr.grad.abs().max()))
    torch.testing.assert_close(z.detach().cpu(), zr.detach(), rtol=1e-4 if dt == torch.float32 else 1e-10,
                               atol=1e-4 if dt == torch.float32 else 1e-10)
    torch.testing.assert_close(p.grad.cpu(), pr.grad, **tol)
    tol_o = dict(rtol=1e-9, atol=1e-9) if dt == torch.float64 else dict(rtol=2e-3, atol=2e-4 * float(omr.grad.abs().max()))
    torch.testing.assert_close(om.grad.cpu(), omr.grad, **tol_o)


def test_ar_flow_sampling_with_gradients(tnf, oracle):
    """NormFlow(arch_type='AR') sampling with fresh batch statistics under autograd (an EFN-style objective on the
    reference's default architecture): MAF sampling-direction backward + batch-statistics BatchNorm + Affine, against
    torch autograd through the oracle."""
    D, L, U, M, N = 6, 2, 15, 2, 300
    np.random.seed(11)
    rng = np.random.RandomState(11)
    nf = tnf.NormFlow(D, True, "AR", 1, L, U)
    Ms = [Mk[0].numpy() for Mk in nf.bijectors[0].Ms]
    p0 = torch.tensor(rng.normal(0, 0.2, (M, nf.D_params))).float()
    omega = rng.normal(0, 1, (M, N, D))
    p = p0.clone().cuda().requires_grad_()
    z, lq = nf._forward_from(omega, p, freeze_bn=False)
    loss = lq.mean() + (z ** 2).mean()
    loss.backward()
    pr = p0.clone().requires_grad_()
    zr, lqr, _ = oracle.ar_flow_forward(omega, pr, D, nf.num_layers, nf.num_units, Ms, None)
    loss_r = lqr.mean() + (zr ** 2).mean()
    loss_r.backward()
    torch.testing.assert_close(loss.detach().cpu().double(), loss_r.detach().double(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(p.grad.cpu(), pr.grad, rtol=5e-3, atol=2e-4 * float(pr.grad.abs().max()))
