"""Fused conditioner + flow kernel (tnf_cond_flow_log_prob_f32): ConditionalDensityEstimator.log_prob with one
sample per context (reference conditional_density_estimator.py:101-104 called as cde.log_prob(z[:, None, :], x))
against the CPU oracle (param_net's last Linear in torch, then oracle.flow_log_prob) and against the
materialised path of this package (hipBLASLt Linear + per-bijector kernels)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tnf():
    import torch_nf_amd

    assert torch.cuda.is_available()
    return torch_nf_amd


def _make(tnf, D, S, L, U, Dx, hidden, seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    nf = tnf.NormFlow(D, True, "coupling", S, L, U)
    cde = tnf.ConditionalDensityEstimator(nf, Dx, hidden)
    g = torch.Generator().manual_seed(seed)
    for b in nf._bn_layers():
        b.set_last_stats(torch.randn(D, generator=g) * 0.1, torch.rand(D, generator=g) * 0.5 + 0.75)
    with torch.no_grad():
        for p in cde.param_net.parameters():
            p.mul_(0.5)
    return nf, cde


def _oracle_lp(oracle, nf, cde, z, x):
    net = cde.param_net.cpu().double()
    params = net(x.cpu().double()).float()
    stats = [(b.get_last_mean().cpu().float(), b.get_last_alpha().cpu().float()) for b in nf._bn_layers()]
    lp = oracle.flow_log_prob(z.cpu(), params, nf.D, nf.num_stages, nf.num_layers, nf.num_units, stats)
    cde.param_net.float().cuda()
    return lp


@pytest.mark.parametrize("D,S,L,U,Dx,hidden,M,variant", [
    (64, 4, 2, 15, 32, [64, 64], 1000, 0),
    (64, 4, 2, 15, 32, [64, 64], 257, 3),
    (64, 1, 1, 16, 8, [32], 77, 2),
    (32, 2, 3, 15, 5, [50], 300, 1),      # hidden width padded 50 -> 64
    (32, 2, 2, 15, 5, [100, 100], 33, 3),  # padded 100 -> 128
    (64, 2, 2, 15, 5, [128], 530, 0),
    (32, 1, 2, 15, 6, [], 100, 0),         # no hidden layer: h = x, width padded 6 -> 32
])
def test_cond_flow_log_prob(tnf, oracle, D, S, L, U, Dx, hidden, M, variant):
    from torch_nf_amd import _lib

    nf, cde = _make(tnf, D, S, L, U, Dx, hidden, 11 + M)
    x = torch.randn(M, Dx, device="cuda")
    z = torch.randn(M, 1, D, device="cuda")
    _lib.lib.tnf_set_option(_lib.OPT_COND_VARIANT, variant)
    try:
        with torch.no_grad():
            assert cde._fused_conditioner_ok(z, x)
            lp_f = cde.log_prob(z, x)
            cde.fuse_conditioner = False
            lp_m = cde.log_prob(z, x)
    finally:
        _lib.lib.tnf_set_option(_lib.OPT_COND_VARIANT, 0)
    assert lp_f.shape == (M, 1) and lp_f.dtype == torch.float32
    lp_o = _oracle_lp(oracle, nf, cde, z, x)
    # north_star tolerance: log_prob rtol <= 1e-5 against the reference's CPU path
    torch.testing.assert_close(lp_f.cpu(), lp_o, rtol=1e-5, atol=1e-5)
    # the two GPU paths are each within 1e-5 of the oracle, hence within 2e-5 of one another
    torch.testing.assert_close(lp_f, lp_m, rtol=2e-5, atol=2e-5)


def test_cond_flow_outputs_and_selection(tnf, oracle):
    from torch_nf_amd import ops

    nf, cde = _make(tnf, 32, 2, 2, 15, 4, [64], 5)
    M = 200
    x = torch.randn(M, 4, device="cuda")
    z = torch.randn(M, 1, 32, device="cuda")
    with torch.no_grad():
        h = cde.param_net[:-1](x)
        last = cde.param_net[-1]
        mean, alpha = nf._bn_stats(torch.device("cuda", torch.cuda.current_device()))
        lp, z0, sld = ops.cond_flow_log_prob_raw(z[:, 0, :], h, last.weight, last.bias, mean, alpha, 32, 2, 2, 15,
                                                 want_z0=True, want_sld=True)
        z0_m, sld_m = nf.inverse_and_log_det(z, cde.param_net(x))
    torch.testing.assert_close(z0, z0_m[:, 0, :], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(sld, sld_m[:, 0], rtol=1e-5, atol=1e-5)
    ref = -0.5 * (z0 ** 2).sum(1) - 32 * np.log(np.sqrt(2 * np.pi)) - sld
    torch.testing.assert_close(lp, ref, rtol=1e-5, atol=1e-5)
    # selection: more than one sample per context, float64 or few contexts -> materialised path
    assert not cde._fused_conditioner_ok(torch.randn(M, 2, 32, device="cuda"), x)
    assert not cde._fused_conditioner_ok(z[:8], x[:8])
    assert cde._fused_conditioner_ok(z, x)  # grad mode: the fused training pair
    with torch.no_grad():
        assert cde._fused_conditioner_ok(z, x)
        assert not cde._fused_conditioner_ok(z.double(), x)
        # CPU inputs are staged and the result comes back on the CPU
        lp_cpu = cde.log_prob(z.cpu(), x.cpu())
    assert lp_cpu.device.type == "cpu"
    torch.testing.assert_close(lp_cpu[:, 0], lp.cpu(), rtol=1e-6, atol=1e-6)


def test_cond_flow_weight_scaling(tnf, oracle):
    """Operand scaling: tiny and large last-layer weights keep fp32-level accuracy (the f16 halves are
    scaled by a power of two chosen from max|W|, |b|)."""
    for scale in (1e-4, 4.0):
        nf, cde = _make(tnf, 32, 1, 2, 15, 4, [32], 3)
        with torch.no_grad():
            last = cde.param_net[-1]
            last.weight.mul_(scale)
            last.bias.mul_(min(scale, 1.0))
            x = torch.randn(64, 4, device="cuda") * (0.05 if scale > 1 else 1.0)
            z = torch.randn(64, 1, 32, device="cuda")
            lp_f = cde.log_prob(z, x)
        lp_o = _oracle_lp(oracle, nf, cde, z, x)
        ok = torch.isfinite(lp_o[:, 0]) & (lp_o[:, 0].abs() < 1e4)  # large weights: some contexts blow up
        assert ok.sum() > 32
        torch.testing.assert_close(lp_f.cpu()[ok], lp_o[ok], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("D,S,L,U,Dx,hidden,M,variant,weighted", [
    (32, 1, 1, 15, 4, [32], 64, 1, False),
    (32, 2, 2, 15, 4, [32], 100, 0, True),
    (64, 4, 2, 15, 32, [64, 64], 300, 3, True),
    (32, 2, 3, 15, 5, [50], 130, 3, False),
    (64, 2, 2, 16, 5, [128], 77, 1, True),
    (64, 4, 2, 15, 32, [64, 64], 300, 5, True),   # variant 5: walk without g_h + the separate g_h kernel
    (32, 2, 3, 15, 5, [50], 130, 5, False),
])
def test_cond_flow_training_gradients(tnf, oracle, D, S, L, U, Dx, hidden, M, variant, weighted):
    """-(w * log_prob).mean() through the fused training pair (tnf_cond_flow_log_prob_fwd/bwd_f32): gradients of
    every param_net parameter and of z against (a) this package's materialised autograd path and (b) torch
    autograd through the CPU oracle (= the reference's own training path, params materialised)."""
    from torch_nf_amd import _lib

    nf, cde = _make(tnf, D, S, L, U, Dx, hidden, 21 + M)
    x = torch.randn(M, Dx, device="cuda")
    z = torch.randn(M, 1, D, device="cuda", requires_grad=True)
    w = (torch.rand(M, 1, device="cuda") + 0.1) if weighted else torch.ones(M, 1, device="cuda")
    _lib.lib.tnf_set_option(_lib.OPT_COND_VARIANT, variant)
    res = []
    try:
        for fuse in (True, False):
            cde.fuse_conditioner = fuse
            cde.zero_grad()
            z.grad = None
            assert cde._fused_conditioner_ok(z, x) == fuse
            loss = -(cde.log_prob(z, x) * w).mean()
            loss.backward()
            res.append((loss.detach().cpu(), [p.grad.detach().cpu().clone() for p in cde.param_net.parameters()],
                        z.grad.detach().cpu().clone()))
    finally:
        _lib.lib.tnf_set_option(_lib.OPT_COND_VARIANT, 0)
    # (b) the oracle: double-precision param_net, float32 flow like the reference
    net = cde.param_net.cpu()
    for p in net.parameters():
        p.grad = None
    zc = z.detach().cpu().clone().requires_grad_()
    stats = [(b.get_last_mean().cpu().float(), b.get_last_alpha().cpu().float()) for b in nf._bn_layers()]
    lp = oracle.flow_log_prob(zc, net(x.cpu()), nf.D, nf.num_stages, nf.num_layers, nf.num_units, stats)
    loss_o = -(lp * w.cpu()).mean()
    loss_o.backward()
    ref = (loss_o.detach(), [p.grad.clone() for p in net.parameters()], zc.grad.clone())
    cde.param_net.cuda()

    def close(a, b, tol):
        scale = float(b.abs().max().clamp_min(1e-30))
        assert float((a - b).abs().max()) <= tol * scale, (float((a - b).abs().max()), scale)

    for other, tol in ((res[1], 2e-5), (ref, 5e-5)):
        torch.testing.assert_close(res[0][0], other[0], rtol=1e-5, atol=1e-5)
        for a, b in zip(res[0][1], other[1]):
            close(a, b, tol)
        close(res[0][2], other[2], tol)
    from conftest import grad_err

    for a, b in zip(res[0][1], ref[1]):
        grad_err("fused conditioner + flow training: d param_net", a, b, 1.1e-5)  # 4 x the 2.6e-6 measured in round 3
    grad_err("fused conditioner + flow training: d z", res[0][2], ref[2], 4e-6)   # ... 9.6e-7


def test_cond_flow_training_tiny_upstream_gradient(tnf):
    """Upstream gradients of 1e-9 (huge batches, loss scaling): the backward rescales them by a power of two
    before the f16 split, so nothing underflows."""
    nf, cde = _make(tnf, 32, 1, 2, 15, 4, [32], 9)
    x = torch.randn(200, 4, device="cuda")
    z = torch.randn(200, 1, 32, device="cuda")
    grads = []
    for scale in (1.0, 1e-9):
        cde.zero_grad()
        (cde.log_prob(z, x).sum() * scale).backward()
        grads.append([p.grad.clone() / scale for p in cde.param_net.parameters()])
    for a, b in zip(*grads):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())


@pytest.mark.parametrize("D,S,L,U,Dx,hidden,M", [
    (64, 4, 2, 15, 32, [64, 64], 300), (32, 2, 3, 16, 10, [50, 100], 17), (64, 1, 1, 15, 8, [32], 70000),
])
def test_cond_flow_sampling_direction(tnf, oracle, D, S, L, U, Dx, hidden, M):
    """cde(x, N = 1, freeze_bn = True) through the fused conditioner + flow kernel in its SAMPLING direction
    (tnf_cond_flow_forward_f32; conditional_density_estimator.py:93-99 over density_estimator.py:374-388): samples and
    log-density against the CPU oracle's flow_forward on the same host draw (np.random.seed) and frozen statistics, against
    the materialised path of this package, and log_prob(z, x) == log_q on the samples just drawn."""
    nf, cde = _make(tnf, D, S, L, U, Dx, hidden, seed=M)
    x = torch.randn(M, Dx, generator=torch.Generator().manual_seed(1))
    xd = x.cuda()
    with torch.no_grad():
        assert cde._fused_sampling_ok(xd)
        np.random.seed(7)
        z, lq = cde(xd, N=1, freeze_bn=True)
        np.random.seed(7)
        cde.fuse_conditioner = False
        z_m, lq_m = cde(xd, N=1, freeze_bn=True)
        cde.fuse_conditioner = True
        lp = cde.log_prob(z, xd)
    assert z.shape == (M, 1, D) and lq.shape == (M, 1) and lq.dtype == torch.float64
    torch.testing.assert_close(z, z_m, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(lq, lq_m, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(lp.double(), lq, rtol=1e-5, atol=2e-4)
    if M <= 1000:  # the oracle on the same draw
        np.random.seed(7)
        omega = np.random.normal(0.0, 1.0, (M, 1, D))
        net = cde.param_net.cpu().double()
        params = net(x.double()).float()
        cde.param_net.float().cuda()
        stats = [(b.get_last_mean().cpu().float(), b.get_last_alpha().cpu().float()) for b in nf._bn_layers()]
        z_r, lq_r, _ = oracle.flow_forward(omega, params, D, S, L, U, stats)
        torch.testing.assert_close(z.cpu(), z_r.float(), rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(lq.cpu(), lq_r.double(), rtol=1e-5, atol=1e-4)
    # the device-draw variant takes the same kernel; under autograd with a trainable context network it stands back
    with torch.no_grad():
        z_s, lq_s = cde.sample(xd, N=1, generator=torch.Generator(device="cuda").manual_seed(3))
        lp_s = cde.log_prob(z_s, xd)
    torch.testing.assert_close(lp_s.double(), lq_s, rtol=1e-5, atol=2e-4)
    assert not cde._fused_sampling_ok(xd)  # grad mode, parameters require grad


def test_cond_flow_trunk_split_k_weight_gradients(tnf):
    """From 32,768 contexts on, the hidden Linears of param_net get their weight gradients as batched products over row
    blocks (`_SplitKLinear`: the stock g^T x is one K = M GEMM on a handful of workgroups).  Same gradients as the stock
    modules (fuse off -> materialised path through nn.Linear's own backward), ragged row count, dropout-free."""
    D, S, L, U, Dx, hidden, M = 32, 1, 2, 15, 6, [64, 32], 32768 + 77
    nf, cde = _make(tnf, D, S, L, U, Dx, hidden, 5)
    x = torch.randn(M, Dx, device="cuda")
    z = torch.randn(M, 1, D, device="cuda")
    grads = []
    for fuse in (True, False):
        cde.fuse_conditioner = fuse
        cde.zero_grad()
        (-cde.log_prob(z, x).mean()).backward()
        grads.append([p.grad.clone() for p in cde.param_net.parameters()])
    cde.fuse_conditioner = True
    for a, b in zip(*grads):
        scale = float(b.abs().max().clamp_min(1e-30))
        assert float((a - b).abs().max()) <= 2e-5 * scale, (float((a - b).abs().max()), scale)
    # and the helper on its own against the stock module, exactly the same rows
    lin = torch.nn.Linear(48, 64).cuda()
    inp = torch.randn(40000 + 3, 48, device="cuda", requires_grad=True)
    gout = torch.randn(40000 + 3, 64, device="cuda")
    from torch_nf_amd.conditional_density_estimator import _SplitKLinear
    y0 = lin(inp)
    g0 = torch.autograd.grad(y0, [inp, lin.weight, lin.bias], gout)
    y1 = _SplitKLinear.apply(inp, lin.weight, lin.bias)
    g1 = torch.autograd.grad(y1, [inp, lin.weight, lin.bias], gout)
    torch.testing.assert_close(y1, y0, rtol=1e-5, atol=1e-5)
    for a, b in zip(g1, g0):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-3)
