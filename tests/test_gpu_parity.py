"""Parity of the HIP path (through the C ABI) with the reference, on a real MI355X.

Three kinds of evidence, all `-m gpu`:
  1. the committed golden vectors (outputs of the reference itself, tests/golden/*.npz);
  2. the CPU oracle on fresh seeded inputs at sizes it finishes in seconds;
  3. size-independent properties at the benchmark size N = 2^20 (round trips,
     forward/log_prob consistency, shard invariance, pass-through bit identity).

Tolerances (float32): z and log-dets rtol 1e-5 / atol 2e-6 per layer (looser after an
8-layer chain, stated where used); log_prob rtol 1e-5 -- the figure BASELINE.json's
north_star states.  float64 cases: 1e-12.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

LOGP_RTOL = 1e-5  # north_star: "log_prob matching reference to rtol 1e-5"


def T(a, dev="cuda"):
    return torch.from_numpy(np.array(a)).to(dev)


def tol(dtype):
    return dict(rtol=1e-12, atol=1e-12) if dtype == torch.float64 else dict(rtol=1e-5, atol=2e-6)


@pytest.fixture(scope="module")
def tnf():
    import torch_nf_amd

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch_nf_amd


def _force_generic(tnf, on):
    tnf._lib.check(tnf._lib.lib.tnf_set_option(tnf._lib.OPT_FORCE_GENERIC, int(on)))


# 10 = the default whole-flow kernels (inverse: flow_fused2.hip, forward: flow_fused_f16.hip); 15 = the first split-f16
# formulation in both directions (flow_fused_f16.hip); 0 = the fp32-MFMA whole-flow kernel (flow_fused.hip)
DEFAULT_FLOW_VARIANT = 10


@pytest.fixture(params=[10, 20, 15, 0], ids=["wholeflow_f16split_16x16", "wholeflow_f16split_32x32", "wholeflow_f16split_v1", "wholeflow_f32mfma"])
def flow_variant(request, tnf):
    """Run a test once per whole-flow kernel implementation (both sit behind TNF_FUSE_FLOW)."""
    tnf._lib.check(tnf._lib.lib.tnf_set_option(tnf._lib.OPT_FLOW_VARIANT, request.param))
    yield request.param
    tnf._lib.check(tnf._lib.lib.tnf_set_option(tnf._lib.OPT_FLOW_VARIANT, DEFAULT_FLOW_VARIANT))


@pytest.fixture(params=[10, 12, 0], ids=["chain_f16split", "chain_f16split_2layers_per_launch", "chain_f32mfma"])
def layer_variant(request, tnf):
    """Run a test once per implementation of the per-layer chain (TNF_FUSE_LAYER): the whole-flow kernel's tile code
    with one (10) or two (12) coupling layers per launch, and the fp32-MFMA layer kernel with half-row stores (0)."""
    tnf._lib.check(tnf._lib.lib.tnf_set_option(tnf._lib.OPT_LAYER_VARIANT, request.param))
    yield request.param
    tnf._lib.check(tnf._lib.lib.tnf_set_option(tnf._lib.OPT_LAYER_VARIANT, 10))


# --------------------------------------------------------------------------
# 1. golden vectors
# --------------------------------------------------------------------------
@pytest.mark.parametrize("generic", [False, True])
def test_golden_coupling(tnf, generic):
    g = load_golden("coupling")
    _force_generic(tnf, generic)
    try:
        for ci, (D, L, U, upper, Mz, Mp, N, dt, extra) in enumerate(g["meta"].tolist()):
            k = "c%02d_" % ci
            z, p = T(g[k + "z"]), T(g[k + "params"])
            layer = tnf.RealNVP(D, L, U, transform_upper=bool(upper))
            assert p.shape[1] == layer.count_num_params() + extra  # trailing params are ignored
            zf, ldf = layer.forward_and_log_det(z, p)
            zi, ldi = layer.inverse_and_log_det(z, p)
            torch.cuda.synchronize()
            t = tol(z.dtype)
            torch.testing.assert_close(zf.cpu(), T(g[k + "z_fwd"], "cpu"), **t)
            torch.testing.assert_close(ldf.cpu(), T(g[k + "ld_fwd"], "cpu"), **t)
            torch.testing.assert_close(zi.cpu(), T(g[k + "z_inv"], "cpu"), **t)
            torch.testing.assert_close(ldi.cpu(), T(g[k + "ld_inv"], "cpu"), **t)
            h = D // 2
            sl = slice(0, h) if upper else slice(h, D)
            assert torch.equal(zf[:, :, sl], z[:, :, sl].expand(zf.shape[0], -1, -1)), "pass-through half"
            assert torch.equal(zi[:, :, sl], z[:, :, sl].expand(zf.shape[0], -1, -1)), "pass-through half"
    finally:
        _force_generic(tnf, False)


def test_golden_affine_bn(tnf):
    g = load_golden("affine_bn")
    for ci, (D, Mz, Mp, N, dt, extra) in enumerate(g["affine_meta"].tolist()):
        k = "a%02d_" % ci
        z, p = T(g[k + "z"]), T(g[k + "params"])
        layer = tnf.Affine(D)
        zf, ldf = layer.forward_and_log_det(z, p)
        zi, ldi = layer.inverse_and_log_det(z, p)
        t = tol(z.dtype)
        for got, name in ((zf, "z_fwd"), (ldf, "ld_fwd"), (zi, "z_inv"), (ldi, "ld_inv")):
            torch.testing.assert_close(got.cpu(), T(g[k + name], "cpu"), **t)
    for ci, (D, M, N) in enumerate(g["bn_meta"].tolist()):
        k = "b%02d_" % ci
        bn = tnf.BatchNorm(D)
        zb, ldb = bn(T(g[k + "z"]))
        torch.testing.assert_close(zb.cpu(), T(g[k + "z_batch"], "cpu"), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(bn.get_last_mean().cpu(), T(g[k + "mean"], "cpu"), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(bn.get_last_alpha().cpu(), T(g[k + "alpha"], "cpu"), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(ldb.cpu(), T(g[k + "ld_batch"], "cpu"), rtol=1e-5, atol=1e-5)
        assert ldb.dim() == 0
        bn.set_last_stats(T(g[k + "mean"]), T(g[k + "alpha"]))
        zf, ldf = bn(T(g[k + "z2"]), use_last=True)
        zi, ldi = bn.inverse_and_log_det(T(g[k + "z2"]))
        torch.testing.assert_close(zf.cpu(), T(g[k + "z_frozen"], "cpu"), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(zi.cpu(), T(g[k + "z_inv"], "cpu"), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(ldf.cpu(), T(g[k + "ld_frozen"], "cpu"), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(ldi.cpu(), T(g[k + "ld_inv"], "cpu"), rtol=1e-5, atol=1e-5)


def _flow_from_golden(tnf, g, ci, conditioner=False):
    D, S, L, U, N = g["meta"].tolist()[ci][:5]
    nf = tnf.NormFlow(D, conditioner, "coupling", S, L, U)
    return nf, D, S, L, U, N


def _install_stats(nf, mean, alpha):
    for b, m, a in zip(nf._bn_layers(), mean, alpha):
        b.set_last_stats(T(m), T(a))


@pytest.mark.parametrize("fusion", ["auto", "layer", "flow", "bijectors"])
def test_golden_flow(tnf, fusion, flow_variant, layer_variant):
    g = load_golden("flow")
    L_ = tnf._lib
    for ci in range(len(g["meta"])):
        nf, D, S, L, U, N = _flow_from_golden(tnf, g, ci)
        k = "f%02d_" % ci
        fast = tnf.ops.has_fast_path(D, L, U)
        if fusion in ("layer", "flow") and not fast:
            continue
        if fusion == "flow" and not tnf._lib.lib.tnf_flow_fused_supported(D, S, L, U):
            continue
        nf.params = T(g[k + "params"])
        _install_stats(nf, g[k + "bn_mean"], g[k + "bn_alpha"])
        nf.fusion = {"auto": L_.FUSE_AUTO, "layer": L_.FUSE_LAYER, "flow": L_.FUSE_FLOW,
                     "bijectors": L_.FUSE_AUTO}[fusion]
        if fusion == "bijectors":
            nf._fused_ok = lambda z, p: False  # the reference's per-bijector loop, one kernel each
        z_test = T(g[k + "z_test"])
        with torch.no_grad():
            lp = nf.log_prob(z_test)
            z0, sld = nf.inverse_and_log_det(z_test, nf.params)
            z_fz, lq_fz = nf._forward_from(g[k + "omega_fz"], nf.params, freeze_bn=True)
        torch.testing.assert_close(lp.cpu(), T(g[k + "log_prob"], "cpu"), rtol=LOGP_RTOL, atol=1e-5)
        torch.testing.assert_close(z0.cpu(), T(g[k + "z0"], "cpu"), rtol=2e-5, atol=1e-5)
        torch.testing.assert_close(sld.cpu(), T(g[k + "sum_log_det"], "cpu"), rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(z_fz.cpu(), T(g[k + "z_fz"], "cpu"), rtol=2e-5, atol=1e-5)
        assert lq_fz.dtype == torch.float64 and z_fz.dtype == torch.float32
        torch.testing.assert_close(lq_fz.cpu(), T(g[k + "logq_fz"], "cpu"), rtol=LOGP_RTOL, atol=2e-5)


def test_golden_flow_batch_stats_forward(tnf):
    """forward with freeze_bn=False: batch statistics computed on the GPU, cached stats match."""
    g = load_golden("flow")
    for ci in range(len(g["meta"])):
        nf, D, S, L, U, N = _flow_from_golden(tnf, g, ci)
        k = "f%02d_" % ci
        nf.params = T(g[k + "params"])
        with torch.no_grad():
            z, lq = nf._forward_from(g[k + "omega"], nf.params, freeze_bn=False)
        torch.testing.assert_close(z.cpu(), T(g[k + "z_fwd"], "cpu"), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(lq.cpu(), T(g[k + "logq_fwd"], "cpu"), rtol=LOGP_RTOL, atol=1e-4)
        for b, m, a in zip(nf._bn_layers(), g[k + "bn_mean"], g[k + "bn_alpha"]):
            torch.testing.assert_close(b.get_last_mean().cpu(), T(m, "cpu"), rtol=1e-4, atol=1e-4)
            torch.testing.assert_close(b.get_last_alpha().cpu(), T(a, "cpu"), rtol=1e-4, atol=1e-5)


def test_golden_cde(tnf):
    g = load_golden("cde")
    for ci, row in enumerate(g["meta"].tolist()):
        D, S, L, U, D_x, nh, M, N = row[:8]
        hidden = row[8:8 + nh]
        k = "d%02d_" % ci
        nf = tnf.NormFlow(D, True, "coupling", S, L, U)
        cde = tnf.ConditionalDensityEstimator(nf, D_x, hidden)
        _install_stats(nf, g[k + "bn_mean"], g[k + "bn_alpha"])
        params = T(g[k + "params"])
        with torch.no_grad():
            lp = nf.log_prob(T(g[k + "z_test"]), params)
        torch.testing.assert_close(lp.cpu(), T(g[k + "log_prob"], "cpu"), rtol=LOGP_RTOL, atol=1e-5)
        sd = {n[len(k) + 3:]: T(g[n]) for n in g if n.startswith(k + "sd_")}
        if sd:  # whole module: param_net (torch.nn on the GPU) -> flow kernels
            cde.load_state_dict(sd)
            with torch.no_grad():
                lp2 = cde.log_prob(T(g[k + "z_test"]), T(g[k + "x"]))
            torch.testing.assert_close(lp2.cpu(), T(g[k + "log_prob"], "cpu"), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("fused", [True, False], ids=["one_call_chain", "per_bijector"])
def test_golden_cde_forward(tnf, fused):
    """BASELINE configs[2], forward leg: `cde(x, N)` of the reference (conditional_density_estimator.py:93-99 ->
    density_estimator.py:364-388, per-context parameter rows, fresh batch statistics) -- samples, float64
    log-densities and the statistics each BatchNorm layer caches, against the reference's own outputs."""
    g = load_golden("cde")
    for ci, row in enumerate(g["meta"].tolist()):
        D, S, L, U, D_x, nh, M, N = row[:8]
        k = "d%02d_" % ci
        nf = tnf.NormFlow(D, True, "coupling", S, L, U)
        nf.fused_batch_forward = fused
        params = T(g[k + "params"])
        with torch.no_grad():
            z, lq = nf._forward_from(g[k + "omega"], params, freeze_bn=False)
        assert z.dtype == torch.float32 and lq.dtype == torch.float64 and z.shape == (M, N, D)
        torch.testing.assert_close(z.cpu(), T(g[k + "z_fwd"], "cpu"), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(lq.cpu(), T(g[k + "logq_fwd"], "cpu"), rtol=LOGP_RTOL, atol=2e-4)
        for b, m, a in zip(nf._bn_layers(), g[k + "bn_mean"], g[k + "bn_alpha"]):
            torch.testing.assert_close(b.get_last_mean().cpu(), T(m, "cpu"), rtol=1e-4, atol=1e-4)
            torch.testing.assert_close(b.get_last_alpha().cpu(), T(a, "cpu"), rtol=1e-4, atol=1e-5)
        sd = {n[len(k) + 3:]: T(g[n]) for n in g if n.startswith(k + "sd_")}
        if sd:  # the whole module: param_net -> flow, host draw reproduced through np.random.seed-free injection
            hidden = row[8:8 + nh]
            cde = tnf.ConditionalDensityEstimator(nf, D_x, hidden)
            cde.load_state_dict(sd)
            with torch.no_grad():
                p2 = cde._params_for(T(g[k + "x"]))
                z2, lq2 = nf._forward_from(g[k + "omega"], p2, freeze_bn=False)
            torch.testing.assert_close(z2.cpu(), T(g[k + "z_fwd"], "cpu"), rtol=2e-4, atol=2e-4)
            torch.testing.assert_close(lq2.cpu(), T(g[k + "logq_fwd"], "cpu"), rtol=2e-5, atol=5e-4)


@pytest.mark.parametrize("fusion", ["layer", "flow", "bijectors"])
def test_golden_cde_frozen_forward(tnf, oracle, fusion, flow_variant):
    """`cde(x, N, freeze_bn=True)` with the reference's cached statistics: the golden inputs through the per-layer
    chain, the whole-flow kernel and the per-bijector loop, against the oracle's frozen forward on the same draw
    (the oracle equals the reference bit for bit on these inputs, oracle/gen_golden.py)."""
    g = load_golden("cde")
    L_ = tnf._lib
    for ci, row in enumerate(g["meta"].tolist()):
        D, S, L, U, D_x, nh, M, N = row[:8]
        k = "d%02d_" % ci
        if fusion in ("layer", "flow") and not tnf.ops.has_fast_path(D, L, U):
            continue
        if fusion == "flow" and not L_.lib.tnf_flow_fused_supported(D, S, L, U):
            continue
        nf = tnf.NormFlow(D, True, "coupling", S, L, U)
        _install_stats(nf, g[k + "bn_mean"], g[k + "bn_alpha"])
        nf.fusion = {"layer": L_.FUSE_LAYER, "flow": L_.FUSE_FLOW, "bijectors": L_.FUSE_AUTO}[fusion]
        if fusion == "bijectors":
            nf._fused_ok = lambda z, p: False
        params = torch.from_numpy(g[k + "params"])
        stats = [(torch.from_numpy(m), torch.from_numpy(a)) for m, a in zip(g[k + "bn_mean"], g[k + "bn_alpha"])]
        z_want, lq_want, _ = oracle.flow_forward(g[k + "omega"], params, D, S, L, U, stats)
        with torch.no_grad():
            z, lq = nf._forward_from(g[k + "omega"], params.cuda(), freeze_bn=True)
            lp = nf.log_prob(z, params.cuda())
        torch.testing.assert_close(z.cpu(), z_want, rtol=2e-5, atol=1e-5)
        torch.testing.assert_close(lq.cpu(), lq_want, rtol=LOGP_RTOL, atol=2e-5)
        torch.testing.assert_close(lp.cpu().double(), lq_want, rtol=1e-4, atol=1e-3)  # forward <-> log_prob


# --------------------------------------------------------------------------
# 2. oracle on fresh inputs
# --------------------------------------------------------------------------
def _rand_flow(tnf, D, S, L, U, seed, sigma=0.1, M=1):
    rng = np.random.RandomState(seed)
    nf = tnf.NormFlow(D, M > 1, "coupling", S, L, U)
    params = torch.tensor(rng.normal(0.0, sigma, (M, nf.D_params))).float()
    mean = rng.normal(0.0, 0.3, (2 * S, D)).astype(np.float32)
    alpha = np.exp(rng.normal(0.0, 0.2, (2 * S, D))).astype(np.float32)
    _install_stats(nf, mean, alpha)
    if M == 1:
        nf.params = params.cuda()  # conditioner=False flows use their own row
    stats = [(torch.from_numpy(m), torch.from_numpy(a)) for m, a in zip(mean, alpha)]
    return nf, params, stats


@pytest.mark.parametrize("D,S,L,U,N", [(64, 4, 2, 15, 16384), (32, 4, 2, 15, 16384), (64, 1, 1, 16, 1000),
                                       (32, 2, 3, 15, 4097), (64, 4, 2, 15, 31), (2, 1, 2, 15, 1024),
                                       (6, 2, 2, 20, 513),
                                       # shapes of the wide per-layer MFMA kernel (D % 8 == 0, U <= 64, L <= 5)
                                       (64, 2, 2, 20, 2049), (64, 2, 2, 64, 1000), (16, 3, 2, 15, 777),
                                       (8, 2, 3, 20, 333), (128, 1, 2, 32, 500), (24, 2, 5, 17, 450),
                                       (40, 2, 1, 50, 129), (64, 4, 4, 15, 64)])
def test_oracle_log_prob(tnf, oracle, flow_variant, layer_variant, D, S, L, U, N):
    nf, params, stats = _rand_flow(tnf, D, S, L, U, seed=N + D)
    z = torch.randn(1, N, D, generator=torch.Generator().manual_seed(1))
    want = oracle.flow_log_prob(z, params, D, S, L, U, stats)
    fusions = [tnf._lib.FUSE_AUTO]
    if tnf.ops.has_fast_path(D, L, U):
        fusions = [tnf._lib.FUSE_LAYER] + ([tnf._lib.FUSE_FLOW] if tnf._lib.lib.tnf_flow_fused_supported(D, S, L, U) else [])
    for fusion in fusions:
        nf.fusion = fusion
        with torch.no_grad():
            got = nf.log_prob(z.cuda(), params.cuda())
        torch.testing.assert_close(got.cpu(), want, rtol=LOGP_RTOL, atol=1e-5)


def test_oracle_many_contexts(tnf, oracle, flow_variant, layer_variant):
    """M_p = M_z > 1 (per-context weights, cfg 3 shape family) and M_p = 1 broadcast."""
    D, S, L, U = 64, 4, 2, 15
    for M, N in [(16, 512), (3, 40), (64, 1), (5, 17)]:
        nf, params, stats = _rand_flow(tnf, D, S, L, U, seed=M, M=M)
        z = torch.randn(M, N, D, generator=torch.Generator().manual_seed(2))
        want = oracle.flow_log_prob(z, params, D, S, L, U, stats)
        for fusion in (tnf._lib.FUSE_LAYER, tnf._lib.FUSE_FLOW):
            nf.fusion = fusion
            with torch.no_grad():
                got = nf.log_prob(z.cuda(), params.cuda())
            torch.testing.assert_close(got.cpu(), want, rtol=LOGP_RTOL, atol=1e-5)
        # one parameter row shared by all M sample batches
        want1 = oracle.flow_log_prob(z, params[:1], D, S, L, U, stats)
        with torch.no_grad():
            got1 = nf.log_prob(z.cuda(), params[:1].cuda())
        torch.testing.assert_close(got1.cpu(), want1, rtol=LOGP_RTOL, atol=1e-5)


def test_oracle_forward_many_contexts(tnf, oracle, flow_variant):
    """BASELINE configs[2] shape family, forward leg at D = 64: per-context parameter rows (M_p = M > 1) through the
    per-layer chain and the whole-flow kernel, frozen statistics, against the oracle's forward on the same draw; plus
    fresh batch statistics (the reference's default `cde(x, N)`) through the one-call chain."""
    D, S, L, U = 64, 4, 2, 15
    for M, N in [(16, 512), (3, 40), (5, 17), (2, 33)]:
        nf, params, stats = _rand_flow(tnf, D, S, L, U, seed=100 + M, M=M)
        omega = np.random.RandomState(M).normal(0, 1, (M, N, D))
        z_want, lq_want, _ = oracle.flow_forward(omega, params, D, S, L, U, stats)
        fusions = [tnf._lib.FUSE_LAYER, tnf._lib.FUSE_FLOW] if N >= 32 else [tnf._lib.FUSE_AUTO]
        for fusion in fusions:
            nf.fusion = fusion
            with torch.no_grad():
                z, lq = nf._forward_from(omega, params.cuda(), freeze_bn=True)
            torch.testing.assert_close(z.cpu(), z_want, rtol=2e-5, atol=1e-5)
            torch.testing.assert_close(lq.cpu(), lq_want, rtol=LOGP_RTOL, atol=2e-5)
        nf.fusion = tnf._lib.FUSE_AUTO
        zb_want, lqb_want, stb = oracle.flow_forward(omega, params, D, S, L, U, None)
        with torch.no_grad():
            zb, lqb = nf._forward_from(omega, params.cuda(), freeze_bn=False)
        torch.testing.assert_close(zb.cpu(), zb_want, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(lqb.cpu(), lqb_want, rtol=LOGP_RTOL, atol=5e-4)
        for b, (m, a) in zip(nf._bn_layers(), stb):
            torch.testing.assert_close(b.get_last_mean().cpu(), m, rtol=1e-4, atol=1e-4)
            torch.testing.assert_close(b.get_last_alpha().cpu(), a, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("M,N", [(16, 1 << 16), (2048, 512)])
def test_config2_full_size_properties(tnf, oracle, M, N):
    """BASELINE configs[2] at its full size (SURVEY 8d cfg 3: ConditionalDensityEstimator over NormFlow(64, coupling,
    4 stages), D_x = 32, hidden [64, 64], M*N = 2^20): `cde(x, N, freeze_bn=True)` then `cde.log_prob(z, x)`.
    Size-independent properties where the oracle is too slow: forward <-> log_prob consistency, inverse(forward) round
    trip, whole-flow kernel == per-layer chain == per-bijector loop, plus the oracle on a few contexts."""
    D, S, L, U, D_x = 64, 4, 2, 15, 32
    torch.manual_seed(5)
    np.random.seed(5)
    nf = tnf.NormFlow(D, True, "coupling", S, L, U)
    cde = tnf.ConditionalDensityEstimator(nf, D_x, [64, 64])
    rng = np.random.RandomState(M)
    mean = rng.normal(0.0, 0.3, (2 * S, D)).astype(np.float32)
    alpha = np.exp(rng.normal(0.0, 0.2, (2 * S, D))).astype(np.float32)
    _install_stats(nf, mean, alpha)
    stats = [(torch.from_numpy(m), torch.from_numpy(a)) for m, a in zip(mean, alpha)]
    x = torch.randn(M, D_x, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    omega = torch.randn(M, N, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    L_ = tnf._lib
    with torch.no_grad():
        params = cde._params_for(x)
        assert params.shape == (M, nf.D_params)
        out = {}
        for name, fusion in (("flow", L_.FUSE_FLOW), ("layer", L_.FUSE_LAYER)):
            nf.fusion = fusion
            z, lq = nf._forward_from(omega, params, freeze_bn=True)
            lp = cde.log_prob(z, x)  # the public call: param_net again, then the flow kernels
            z0, sld = nf.inverse_and_log_det(z, params)
            out[name] = (z, lq, lp, z0, sld)
        nf.fusion = L_.FUSE_AUTO
    z, lq, lp, z0, sld = out["flow"]
    assert z.shape == (M, N, D) and lq.dtype == torch.float64 and bool(torch.isfinite(lq).all())
    # forward <-> log_prob (the reference's own check, tests/test_conditional_density_estimators.py:55-69)
    torch.testing.assert_close(lp.double(), lq, rtol=1e-4, atol=2e-3)
    # inverse(forward(omega)) == omega
    torch.testing.assert_close(z0, omega, rtol=1e-4, atol=2e-4)
    # whole-flow kernel (split-f16 contractions) == per-layer chain (fp32 MFMA).  z tolerance: the SAMPLING direction
    # through 8 layers with the weights of an untrained param_net amplifies the two kernels' ~1e-7 rounding
    # differences ~100x on a few hundred of the 6.7e7 values (observed 3e-5); the densities carry the 1e-5 bar
    torch.testing.assert_close(out["layer"][0], z, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out["layer"][1], lq, rtol=LOGP_RTOL, atol=1e-4)
    torch.testing.assert_close(out["layer"][2], lp, rtol=LOGP_RTOL, atol=1e-4)
    # ... == the per-bijector loop and == the oracle, on a few contexts
    sel = torch.tensor([0, M // 2, M - 1], device="cuda")
    nsub = min(N, 4096)
    om_s, p_s = omega[sel][:, :nsub].contiguous(), params[sel].contiguous()
    nf2 = tnf.NormFlow(D, True, "coupling", S, L, U)
    _install_stats(nf2, mean, alpha)
    nf2._fused_ok = lambda z_, p_: False
    with torch.no_grad():
        zb, lqb = nf2._forward_from(om_s, p_s, freeze_bn=True)
    torch.testing.assert_close(zb, z[sel][:, :nsub], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(lqb, lq[sel][:, :nsub], rtol=LOGP_RTOL, atol=1e-4)
    z_want, lq_want, _ = oracle.flow_forward(om_s.double().cpu().numpy(), p_s.cpu(), D, S, L, U, stats)
    torch.testing.assert_close(z[sel][:, :nsub].cpu(), z_want, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(lq[sel][:, :nsub].cpu(), lq_want, rtol=LOGP_RTOL, atol=1e-4)
    lp_want = oracle.flow_log_prob(z[sel][:, :nsub].cpu(), p_s.cpu(), D, S, L, U, stats)
    torch.testing.assert_close(lp[sel][:, :nsub].cpu(), lp_want, rtol=LOGP_RTOL, atol=1e-5)


def test_config0_plumbing(tnf, oracle):
    """BASELINE configs[0]: 2-D Gaussian, 2-layer RealNVP (1 stage), batch = 1024 contexts,
    N = 1 (the LFI_gauss layout) -- the reference's own CPU-runnable case, here HIP vs oracle."""
    D, S, L, U, M = 2, 1, 2, 15, 1024
    nf, params, stats = _rand_flow(tnf, D, S, L, U, seed=7, M=M)
    omega = np.random.RandomState(3).normal(0, 1, (M, 1, D))
    z_want, lq_want, _ = oracle.flow_forward(omega, params, D, S, L, U, stats)
    with torch.no_grad():
        z, lq = nf._forward_from(omega, params.cuda(), freeze_bn=True)
        lp = nf.log_prob(z, params.cuda())
    torch.testing.assert_close(z.cpu(), z_want, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lq.cpu(), lq_want, rtol=LOGP_RTOL, atol=1e-5)
    torch.testing.assert_close(lp.cpu().double(), lq_want, rtol=1e-4, atol=1e-4)


def test_host_tensors_are_staged(tnf, oracle):
    """CPU tensors in -> computed on the GPU -> CPU tensors out (drop-in for CPU-only callers)."""
    D, L, U = 8, 2, 15
    layer = tnf.RealNVP(D, L, U)
    rng = np.random.RandomState(0)
    p = torch.tensor(rng.normal(0, 0.1, (3, layer.count_num_params())))
    z = torch.tensor(rng.normal(0, 1, (3, 7, D)))
    got, ld = layer(z, p)
    want, ld_want = oracle.coupling(z, p, D, L, U, True, False)
    assert got.device.type == "cpu" and got.dtype == torch.float64
    torch.testing.assert_close(got, want, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(ld, ld_want, rtol=1e-12, atol=1e-12)


# --------------------------------------------------------------------------
# 3. properties at the benchmark size
# --------------------------------------------------------------------------
@pytest.mark.parametrize("D", [64, 32])
def test_full_size_properties(tnf, oracle, flow_variant, D):
    S, L, U, N = 4, 2, 15, 1 << 20
    nf, params, stats = _rand_flow(tnf, D, S, L, U, seed=11)
    params = params.cuda()
    gen = torch.Generator(device="cuda").manual_seed(1)
    omega = torch.randn(1, N, D, device="cuda", generator=gen)
    mean, alpha = nf._bn_stats(torch.device("cuda"))
    ops, L_ = tnf.ops, tnf._lib
    with torch.no_grad():
        z, sld_f = ops.flow_forward_raw(omega, params, mean, alpha, D, S, L, U, L_.FUSE_FLOW)
        z_l, sld_fl = ops.flow_forward_raw(omega, params, mean, alpha, D, S, L, U, L_.FUSE_LAYER)
        lp, z0, sld_i = ops.flow_log_prob_raw(z, params, mean, alpha, D, S, L, U, L_.FUSE_FLOW,
                                              want_z0=True, want_sld=True)
        lp_l, z0_l, sld_il = ops.flow_log_prob_raw(z, params, mean, alpha, D, S, L, U, L_.FUSE_LAYER,
                                                   want_z0=True, want_sld=True)
    # whole-flow kernel == per-layer chain
    torch.testing.assert_close(z, z_l, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lp, lp_l, rtol=LOGP_RTOL, atol=1e-5)
    # inverse(forward(omega)) == omega, and the two log-det sums agree
    torch.testing.assert_close(z0, omega, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(z0_l, omega, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(sld_i, sld_f, rtol=1e-4, atol=1e-4)
    # log_prob(z) == log N(omega) - sum_log_det  (density_estimator.py:387 vs :413-416)
    base = -0.5 * (omega.double() ** 2).sum(2) - D * np.log(np.sqrt(2 * np.pi))
    torch.testing.assert_close(lp.double(), base - sld_f.double(), rtol=1e-4, atol=1e-3)
    # shard invariance: evaluating a slice == slicing the evaluation (what multi-GPU relies on)
    lo, hi = 123457, 654321
    with torch.no_grad():
        lp_s, _, _ = ops.flow_log_prob_raw(z[:, lo:hi].contiguous(), params, mean, alpha, D, S, L, U, L_.FUSE_FLOW)
    assert torch.equal(lp_s, lp[:, lo:hi])
    # a 2^16-sample slice against the CPU oracle (parity on the same run)
    sl = slice(1 << 19, (1 << 19) + (1 << 16))
    want = oracle.flow_log_prob(z[:, sl].cpu(), params.cpu(), D, S, L, U, stats)
    torch.testing.assert_close(lp[:, sl].cpu(), want, rtol=LOGP_RTOL, atol=1e-5)
    assert torch.isfinite(lp).all()


def test_full_size_layer_roundtrip(tnf):
    """One RealNVP layer at N = 2^20: inverse(forward(z)) == z, log-dets equal, pass-through exact."""
    D, L, U, N = 64, 2, 15, 1 << 20
    layer = tnf.RealNVP(D, L, U, transform_upper=False)
    rng = np.random.RandomState(5)
    p = torch.tensor(rng.normal(0, 0.1, (1, layer.count_num_params()))).float().cuda()
    z = torch.randn(1, N, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))
    with torch.no_grad():
        zf, ldf = layer.forward_and_log_det(z, p)
        zi, ldi = layer.inverse_and_log_det(zf, p)
    assert torch.equal(zf[:, :, D // 2:], z[:, :, D // 2:])
    torch.testing.assert_close(zi, z, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ldi, ldf, rtol=1e-5, atol=1e-6)


def test_edge_shapes(tnf, oracle, flow_variant, layer_variant):
    """Ragged / tiny inputs: N below one MFMA tile, N = 1, tails that are not a multiple of 16 or 32."""
    D, S, L, U = 64, 4, 2, 15
    nf, params, stats = _rand_flow(tnf, D, S, L, U, seed=21)
    for N in (1, 2, 15, 16, 17, 33, 63, 257):
        z = torch.randn(1, N, D, generator=torch.Generator().manual_seed(N))
        want = oracle.flow_log_prob(z, params, D, S, L, U, stats)
        for fusion in (tnf._lib.FUSE_LAYER, tnf._lib.FUSE_FLOW):
            nf.fusion = fusion
            with torch.no_grad():
                got = nf.log_prob(z.cuda(), params.cuda())
            torch.testing.assert_close(got.cpu(), want, rtol=LOGP_RTOL, atol=1e-5)
    empty = torch.zeros(1, 0, D, device="cuda")
    with torch.no_grad():
        assert nf.log_prob(empty, params.cuda()).shape == (1, 0)


def test_errors_are_loud(tnf):
    layer = tnf.RealNVP(8, 2, 15)
    z = torch.zeros(2, 4, 8, device="cuda")
    with pytest.raises(tnf._lib.TnfError):  # parameter row too short
        layer(z, torch.zeros(2, 10, device="cuda"))
    with pytest.raises(RuntimeError):  # M_z = 2 vs M_p = 3 do not broadcast
        layer(z, torch.zeros(3, layer.count_num_params(), device="cuda"))
    with pytest.raises(TypeError):
        layer(z.half(), torch.zeros(2, layer.count_num_params(), device="cuda").half())


def test_many_contexts_beyond_grid_y(tnf, oracle):
    """M > 65,535 parameter rows (gridDim.y alone would overflow) with N = 1: the SNPE layout
    cde.log_prob(z[:, None, :], x) of the reference's training loops (notebooks/LFI_learning_rules.ipynb:304)."""
    D, S, L, U, M = 4, 1, 2, 15, 70001
    rng = np.random.RandomState(5)
    nf = tnf.NormFlow(D, True, "coupling", S, L, U)
    params = torch.tensor(rng.normal(0, 0.1, (M, nf.D_params))).float()
    mean = rng.normal(0, 0.3, (2 * S, D)).astype(np.float32)
    alpha = np.exp(rng.normal(0, 0.2, (2 * S, D))).astype(np.float32)
    _install_stats(nf, mean, alpha)
    stats = [(torch.from_numpy(m), torch.from_numpy(a)) for m, a in zip(mean, alpha)]
    z = torch.tensor(rng.normal(0, 1, (M, 1, D))).float()
    with torch.no_grad():
        got = nf.log_prob(z.cuda(), params.cuda()).cpu()
    sel = torch.cat([torch.arange(0, 300), torch.arange(65400, 65700), torch.arange(M - 300, M)])
    want = oracle.flow_log_prob(z[sel], params[sel], D, S, L, U, stats)
    torch.testing.assert_close(got[sel], want, rtol=LOGP_RTOL, atol=1e-5)
    # and the MFMA shapes: D = 64 with per-context weights, few samples each (per-bijector path)
    D, S = 64, 4
    M = 300
    nf = tnf.NormFlow(D, True, "coupling", S, L, U)
    params = torch.tensor(rng.normal(0, 0.1, (M, nf.D_params))).float()
    mean = rng.normal(0, 0.3, (2 * S, D)).astype(np.float32)
    alpha = np.exp(rng.normal(0, 0.2, (2 * S, D))).astype(np.float32)
    _install_stats(nf, mean, alpha)
    stats = [(torch.from_numpy(m), torch.from_numpy(a)) for m, a in zip(mean, alpha)]
    for N in (1, 20):
        z = torch.tensor(rng.normal(0, 1, (M, N, D))).float()
        with torch.no_grad():
            got = nf.log_prob(z.cuda(), params.cuda()).cpu()
        torch.testing.assert_close(got, oracle.flow_log_prob(z, params, D, S, L, U, stats), rtol=LOGP_RTOL, atol=1e-5)


def test_sample_uses_device_rng_and_matches_log_prob(tnf):
    """NormFlow.sample (extension): device-side base draw; its log_q equals log_prob of its z."""
    for D in (64, 6):
        nf, params, stats = _rand_flow(tnf, D, 2, 2, 15, seed=31)
        gen = torch.Generator(device="cuda").manual_seed(3)
        with torch.no_grad():
            z, lq = nf.sample(5000, generator=gen)
            lp = nf.log_prob(z)
        assert z.shape == (1, 5000, D) and z.dtype == torch.float32 and lq.dtype == torch.float64
        assert z.is_cuda and torch.isfinite(lq).all()
        torch.testing.assert_close(lp.double(), lq, rtol=1e-4, atol=1e-3)
        with torch.no_grad():
            z2, _ = nf.sample(5000, generator=torch.Generator(device="cuda").manual_seed(3))
        assert torch.equal(z, z2)  # same generator state -> same draw, same kernels -> same bits


def test_forward_log_q_from_the_sampling_kernel(tnf):
    """tnf_flow_forward_logq_f32: the whole-flow sampling kernel writes log_q = log N(omega; 0, I) - sum_log_det itself
    (density_estimator.py:369-372, 387) -- the same float64 value as the separate base-density kernel minus the kernel's
    own float32 log-det, with per-context rows and with a ToInterval support layer; variants without that output say so
    (None) and NormFlow falls back to the separate kernel."""
    ops, L_ = tnf.ops, tnf._lib
    D, S, L, U = 64, 4, 2, 15
    for M, N in ((1, 5000), (3, 77)):
        nf, params, stats = _rand_flow(tnf, D, S, L, U, seed=50 + M, M=M)
        _install_stats(nf, [m.numpy() for m, _ in stats], [a.numpy() for _, a in stats])
        mean, alpha = nf._bn_stats(torch.device("cuda"))
        om = torch.randn(M, N, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(M))
        with torch.no_grad():
            z, sld, lq = ops.flow_forward_raw(om, params.cuda(), mean, alpha, D, S, L, U, L_.FUSE_FLOW, want_log_q=True)
            z2, sld2 = ops.flow_forward_raw(om, params.cuda(), mean, alpha, D, S, L, U, L_.FUSE_FLOW)
            want = ops.base_log_density_f64(om) - sld2
        assert lq is not None and lq.dtype == torch.float64
        assert torch.equal(z, z2) and torch.equal(sld, sld2)
        torch.testing.assert_close(lq, want, rtol=1e-13, atol=1e-10)
    L_.check(L_.lib.tnf_set_option(L_.OPT_FLOW_VARIANT, 15))
    try:
        with torch.no_grad():
            _, _, none = ops.flow_forward_raw(om, params.cuda(), mean, alpha, D, S, L, U, L_.FUSE_FLOW, want_log_q=True)
        assert none is None
    finally:
        L_.check(L_.lib.tnf_set_option(L_.OPT_FLOW_VARIANT, DEFAULT_FLOW_VARIANT))


def test_hip_graph_capture(tnf, oracle):
    """The C-ABI calls enqueue on torch's current stream and never synchronise or allocate, so a
    log_prob call can be captured into a HIP graph and replayed on new data."""
    D, S, L, U, N = 64, 4, 2, 15, 4096
    nf, params, stats = _rand_flow(tnf, D, S, L, U, seed=41)
    z_static = torch.randn(1, N, D, device="cuda")
    with torch.no_grad():
        nf.log_prob(z_static)  # warm-up: workspace allocation, BN-stat cache
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out_static = nf.log_prob(z_static)
        z_new = torch.randn(1, N, D, generator=torch.Generator().manual_seed(9))
        z_static.copy_(z_new)
        g.replay()
        torch.cuda.synchronize()
    want = oracle.flow_log_prob(z_new, params, D, S, L, U, stats)
    torch.testing.assert_close(out_static.cpu(), want, rtol=LOGP_RTOL, atol=1e-5)


@pytest.mark.parametrize("D,S,L,M,N", [(64, 4, 2, 1, 5000), (32, 2, 3, 3, 700), (64, 1, 1, 2, 33),
                                       # fewer than 32 rows: partial tiles in every kernel of the chain (one parameter row;
                                       # with several rows and N < 32 the flow composes per bijector anyway)
                                       (64, 2, 2, 1, 2), (32, 1, 2, 1, 5), (64, 1, 2, 1, 15), (64, 4, 2, 1, 31), (32, 2, 1, 4, 31)])
def test_batch_stats_forward_one_call_vs_per_bijector(tnf, D, S, L, M, N):
    """NormFlow.forward(freeze_bn=False) without autograd: the one-call chain (tnf_flow_forward_batch_f32: BatchNorm
    and Affine folded into the next coupling kernel, statistics over all M*N rows) against the per-bijector
    composition -- samples, log-densities and the statistics every BatchNorm layer caches."""
    U = 15
    rng = np.random.RandomState(D + N)
    nf = tnf.NormFlow(D, True, "coupling", S, L, U)
    params = torch.tensor(rng.normal(0, 0.1, (M, nf.D_params))).float().cuda()
    omega = rng.normal(0, 1, (M, N, D))
    out = {}
    for fused in (True, False):
        nf.fused_batch_forward = fused
        with torch.no_grad():
            z, lq = nf._forward_from(omega, params, freeze_bn=False)
        out[fused] = (z, lq, [b.get_last_mean().clone() for b in nf._bn_layers()],
                      [b.get_last_alpha().clone() for b in nf._bn_layers()])
    torch.testing.assert_close(out[True][0], out[False][0], rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(out[True][1], out[False][1], rtol=1e-6, atol=2e-4)
    for a, b in zip(out[True][2] + out[True][3], out[False][2] + out[False][3]):
        torch.testing.assert_close(a.cpu(), b.cpu(), rtol=2e-5, atol=2e-5)
    # the cached statistics serve the frozen paths afterwards
    with torch.no_grad():
        lp = nf.log_prob(out[True][0], params)
    torch.testing.assert_close(lp.double(), out[True][1], rtol=1e-5, atol=2e-3)


def test_batch_stats_forward_full_size_round_trip(tnf):
    """Sampling with fresh batch statistics at 2^19 samples (D=64, 8 coupling layers), where the oracle is too slow:
    the density the one-call chain reports for its samples is the density log_prob assigns to them afterwards (the
    reference's own forward -> log_prob check, tests/test_density_estimators.py:147-245), the cached statistics are
    those of a unit-variance, zero-mean batch behind every BatchNorm, and the one-node autograd pair returns the same
    samples."""
    D, S, L, U, N = 64, 4, 2, 15, 1 << 19
    rng = np.random.RandomState(3)
    nf = tnf.NormFlow(D, False, "coupling", S, L, U)
    nf.params = torch.tensor(rng.normal(0.0, 0.1, (1, nf.D_params))).float().cuda()
    omega = torch.randn(1, N, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    with torch.no_grad():
        z, lq = nf._forward_from(omega, nf.params, freeze_bn=False)
        lp = nf.log_prob(z)
    assert bool(torch.isfinite(z).all()) and lq.dtype == torch.float64
    torch.testing.assert_close(lp.double(), lq, rtol=1e-5, atol=5e-3)
    alphas = torch.stack([b.get_last_alpha() for b in nf._bn_layers()])
    assert bool((alphas > 0).all()) and alphas.shape == (2 * S, D)
    p = nf.params.clone().requires_grad_()
    z2, lq2 = nf._forward_from(omega, p, freeze_bn=False)
    torch.testing.assert_close(z2.detach(), z, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lq2.detach(), lq, rtol=1e-6, atol=1e-3)


@pytest.mark.parametrize("D,S,L,M,N,cut", [(64, 4, 2, 1, 5000, 1777), (32, 2, 3, 1, 700, 699), (64, 2, 2, 3, 333, 100)])
def test_batch_stats_forward_sharded_steps(tnf, D, S, L, M, N, cut):
    """SURVEY 8(e) bullet 3: a sample-sharded `nf(N, freeze_bn=False)` reproduces the single-device statistics when the
    per-layer moments [sum | sum of squares | count] are summed over the shards between a layer and its fold.  Two
    shards of one batch are driven in lockstep through the stepwise C ABI on one GPU (the all-reduce stands between
    `layer` and `fold`; here it is a plain sum of the two moment vectors) and compared with the one-call chain over
    the whole batch: samples, log-det sums and the cached statistics."""
    U = 15
    rng = np.random.RandomState(D + N)
    nf = tnf.NormFlow(D, True, "coupling", S, L, U)
    params = torch.tensor(rng.normal(0, 0.1, (M, nf.D_params))).float().cuda()
    omega = torch.tensor(rng.normal(0, 1, (M, N, D))).float().cuda()
    ops = tnf.ops
    with torch.no_grad():
        z_ref, sld_ref, mean_ref, alpha_ref = ops.flow_forward_batch_raw(omega, params, D, S, L, U, 1e-5)
        shards = [ops.FlowForwardBatchSteps(omega[:, :cut].contiguous(), params, D, S, L, U, 1e-5),
                  ops.FlowForwardBatchSteps(omega[:, cut:].contiguous(), params, D, S, L, U, 1e-5)]
        for sh in shards:
            sh.begin()
        for c in range(2 * S):
            moms = [sh.layer(c) for sh in shards]
            total = moms[0] + moms[1]
            assert float(total[2 * D].item()) == M * N
            for mo in moms:
                mo.copy_(total)
            for sh in shards:
                sh.fold(c)
        outs = [sh.end() for sh in shards]
    z = torch.cat([o[0] for o in outs], dim=1)
    sld = torch.cat([o[1] for o in outs], dim=1)
    torch.testing.assert_close(z, z_ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(sld, sld_ref, rtol=1e-5, atol=1e-4)
    for o in outs:  # every shard ends up with the statistics of the whole batch
        torch.testing.assert_close(o[2], mean_ref, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(o[3], alpha_ref, rtol=1e-5, atol=1e-6)
    # the reducer hook of NormFlow: with an identity "all-reduce" (one rank) the stepwise path equals the one call
    calls = []
    nf.batch_stats_reduce = lambda mo: calls.append(mo.shape) or mo
    with torch.no_grad():
        z1, lq1 = nf._forward_from(omega, params, freeze_bn=False)
    assert len(calls) == 2 * S and calls[0] == (2 * D + 1,)
    torch.testing.assert_close(z1, z_ref, rtol=0, atol=0)
    nf.batch_stats_reduce = None
    # ... and under autograd (round 3) the per-bijector composition honours it too: BatchNorm layers cut at their
    # exchange steps, one exchange per layer forward (and one backward: tests/test_gpu_grad.py)
    calls.clear()
    nf.batch_stats_reduce = lambda mo: calls.append(mo.shape) or mo
    z2, _ = nf._forward_from(omega, params.clone().requires_grad_(), freeze_bn=False)
    assert len(calls) == 2 * S and calls[0] == (2 * D + 1,)
    torch.testing.assert_close(z2.detach(), z_ref, rtol=1e-4, atol=1e-4)
    nf.batch_stats_reduce = None


def _rel_err(got, want):
    return ((got.double() - want.double()).abs() / want.double().abs().clamp_min(1e-3)).max().item()


@pytest.mark.parametrize("D", [64, 32])
def test_operand_range(tnf, oracle, D):
    """The reference is plain fp32 (bijectors.py:172,198,237-241) and has no operand range; the default path computes
    its contractions on split-f16 operands, which do.  These inputs leave the comfortable range on purpose: samples
    scaled by 1e5, BatchNorm means of 1e5, first-layer weights of 1e-6 (with large inputs, so that their products are
    O(1)) and of 1e3, everything scaled down by 1e-5.  Truth = the oracle's arithmetic in float64; the bar is rtol
    1e-5 on log_prob, or four times the float32 oracle's own deviation from float64 where fp32 itself cannot do better.
    The power-of-two operand normalisation keeps the small / large weight cases on the fast path; inputs whose scaled
    magnitude passes 65520 are detected and those groups re-run exactly (counted by the diagnostic entry point)."""
    S, L, U, N = 4, 2, 15, 3000
    ops, L_ = tnf.ops, tnf._lib
    h = D // 2

    def scale_layer0(params, fac):
        """multiply the first-layer weights (W_t, W_s of the h -> U layer) of every coupling layer by fac"""
        p = params.clone()
        lay = oracle.flow_layout(D, S, L, U)
        off = 0
        for kind, n, up in lay:
            if kind == "coupling":
                p[:, off:off + 2 * h * U] *= fac
            off += n
        return p

    cases = {}
    nf, params, stats = _rand_flow(tnf, D, S, L, U, seed=77)
    z = torch.randn(1, N, D, generator=torch.Generator().manual_seed(3))
    cases["plain"] = (z, params, stats, False)
    cases["z_1e5"] = (z * 1e5, params, stats, True)
    # None: whether any group leaves the fast path depends on the draw -- only the result is checked
    cases["z_3e4_partly"] = (torch.where(torch.rand(1, N, 1, generator=torch.Generator().manual_seed(4)) < 0.01, z * 3e4, z),
                             params, stats, None)
    cases["bn_mean_1e5"] = (z, params, [(m + 1e5, a) for m, a in stats], None)
    cases["w0_1e-6_z_1e5"] = (z * 1e5, scale_layer0(params, 1e-5), stats, False)
    cases["w0_1e3"] = (z, scale_layer0(params, 1e4), stats, None)
    cases["all_params_1e-5"] = (z, params * 1e-5, stats, False)
    for name, (zz, pp, st, expect_reruns) in cases.items():
        want64 = oracle.flow_log_prob(zz.double(), pp.double(), D, S, L, U, [(m.double(), a.double()) for m, a in st])
        want32 = oracle.flow_log_prob(zz, pp, D, S, L, U, st)
        assert torch.isfinite(want64).all(), name
        bar = max(LOGP_RTOL, 4.0 * _rel_err(want32, want64))
        _install_stats(nf, [m.numpy() for m, _ in st], [a.numpy() for _, a in st])
        mean, alpha = nf._bn_stats(torch.device("cuda"))
        with torch.no_grad():
            lp, z0, sld, reruns = ops.flow_log_prob_raw(zz.cuda(), pp.cuda(), mean, alpha, D, S, L, U, L_.FUSE_FLOW,
                                                        want_z0=True, want_sld=True, count_reruns=True)
            lp_l, _, _ = ops.flow_log_prob_raw(zz.cuda(), pp.cuda(), mean, alpha, D, S, L, U, L_.FUSE_LAYER)
        err = _rel_err(lp.cpu(), want64)
        assert err <= bar, "%s: whole-flow kernel rel err %.3g > %.3g (float32 oracle: %.3g)" % (
            name, err, bar, _rel_err(want32, want64))
        assert _rel_err(lp_l.cpu(), want64) <= bar, name + " (per-layer chain)"
        n_re = int(reruns.item())
        if expect_reruns is True:
            assert n_re > 0, name + ": out-of-range inputs must take the exact path"
        elif expect_reruns is False:
            assert n_re == 0, name + ": %d groups left the fast path" % n_re
        st64 = [(m.double(), a.double()) for m, a in st]
        z0_want, _ = oracle.flow_inverse(zz.double(), pp.double(), D, S, L, U, st64)
        z0_f32, _ = oracle.flow_inverse(zz, pp, D, S, L, U, st)
        scale = z0_want.abs().amax(dim=2, keepdim=True).clamp_min(1.0)  # per sample: its largest coordinate
        zerr = ((z0.cpu().double() - z0_want).abs() / scale).max().item()
        zbar = max(2e-5, 4.0 * ((z0_f32.double() - z0_want).abs() / scale).max().item())
        assert zerr <= zbar, "%s: z0 err %.3g > %.3g" % (name, zerr, zbar)


@pytest.mark.parametrize("D", [64, 32])
def test_operand_range_forward(tnf, oracle, D):
    """test_operand_range for the SAMPLING direction (NormFlow.forward with frozen statistics, bijectors.py:172): base
    draws scaled by 1e4 and by 1e5, BatchNorm means of 1e5, first-layer weights of 1e-5 with large draws and of 1e4,
    everything scaled down.  Truth = the oracle's forward in float64; z is compared per sample relative to its largest
    coordinate, the summed log-det absolutely relative to its size -- each with the float32 oracle's own deviation as
    the yardstick where fp32 itself cannot do better."""
    S, L, U, N = 4, 2, 15, 2000
    ops, L_ = tnf.ops, tnf._lib
    h = D // 2

    def scale_layer0(params, fac):
        p = params.clone()
        off = 0
        for kind, n, up in oracle.flow_layout(D, S, L, U):
            if kind == "coupling":
                p[:, off:off + 2 * h * U] *= fac
            off += n
        return p

    def fwd(omega, params, st):
        """the bijector loop of density_estimator.py:374-388 in the dtype of its inputs: (z, sum of log-dets)"""
        z, sld, idx, bn_i = omega, 0.0, 0, 0
        for kind, n, upper in oracle.flow_layout(D, S, L, U):
            if kind == "coupling":
                z, ld = oracle.coupling(z, params[:, idx:idx + n], D, L, U, upper, False)
                idx += n
            elif kind == "affine":
                z, ld = oracle.affine(z, params[:, idx:idx + n], D, False)
                idx += n
            else:
                z, ld = oracle.bn_forward_frozen(z, st[bn_i][0], st[bn_i][1])
                bn_i += 1
            sld = sld + ld
        return z, sld.expand(z.shape[0], z.shape[1]) if sld.dim() == 2 else sld

    nf, params, stats = _rand_flow(tnf, D, S, L, U, seed=78)
    om = torch.randn(1, N, D, generator=torch.Generator().manual_seed(5))
    cases = {
        "plain": (om, params, stats),
        "omega_1e4": (om * 1e4, params, stats),
        "omega_1e5": (om * 1e5, params, stats),
        "bn_mean_1e5": (om, params, [(m + 1e5, a) for m, a in stats]),
        "w0_1e-5_omega_1e5": (om * 1e5, scale_layer0(params, 1e-5), stats),
        "w0_1e4": (om, scale_layer0(params, 1e4), stats),
        "all_params_1e-5": (om, params * 1e-5, stats),
    }
    for name, (oo, pp, st) in cases.items():
        st64 = [(m.double(), a.double()) for m, a in st]
        z64, sld64 = fwd(oo.double(), pp.double(), st64)
        z32, sld32 = fwd(oo, pp, st)
        assert torch.isfinite(z64).all() and torch.isfinite(sld64).all(), name
        _install_stats(nf, [m.numpy() for m, _ in st], [a.numpy() for _, a in st])
        mean, alpha = nf._bn_stats(torch.device("cuda"))
        scale = z64.abs().amax(dim=2, keepdim=True).clamp_min(1.0)
        # 8 x for z: the kernels' fp32 arithmetic is not the oracle's op for op (exp2 with folded constants, FMA folds),
        # and at draws of 1e5 the forward's cancellations amplify every rounding the same way they amplify the oracle's
        zbar = max(2e-5, 8.0 * ((z32.double() - z64).abs() / scale).max().item())
        lbar = max(LOGP_RTOL, 8.0 * _rel_err(sld32, sld64))
        for fusion in (L_.FUSE_FLOW, L_.FUSE_LAYER):
            with torch.no_grad():
                z, sld = ops.flow_forward_raw(oo.cuda(), pp.cuda(), mean, alpha, D, S, L, U, fusion)
            zerr = ((z.cpu().double() - z64).abs() / scale).max().item()
            assert zerr <= zbar, "%s fusion %d: z err %.3g > %.3g" % (name, fusion, zerr, zbar)
            lerr = _rel_err(sld.cpu(), sld64)
            assert lerr <= lbar, "%s fusion %d: sum log-det err %.3g > %.3g" % (name, fusion, lerr, lbar)


def test_bf16_operand_experiment_is_scoped_and_close(tnf):
    """TNF_OPT_OPERAND_PREC (the fp32-vs-bf16 sweep of BASELINE configs[4], tools/bf16_sweep.py): inside
    ops.operand_precision("bf16") the coupling and autoregressive log_prob kernels use bf16 conditioner operands
    -- results move, but stay within bf16's 2^-8 operand rounding of the fp32 path -- and the default path is
    bit-identical before and after the block.  Parity unpinned: the reference has no reduced-precision mode."""
    ops, L_ = tnf.ops, tnf._lib
    D, S, L, U, N = 64, 4, 2, 15, 4096
    nf, params, stats = _rand_flow(tnf, D, S, L, U, seed=5)
    _install_stats(nf, [m.numpy() for m, _ in stats], [a.numpy() for _, a in stats])
    mean, alpha = nf._bn_stats(torch.device("cuda"))
    z = torch.randn(1, N, D, generator=torch.Generator().manual_seed(6)).cuda()
    pc = params.cuda()
    for fusion in (L_.FUSE_FLOW, L_.FUSE_LAYER):
        ref = ops.flow_log_prob_raw(z, pc, mean, alpha, D, S, L, U, fusion)[0]
        with ops.operand_precision("bf16"):
            low = ops.flow_log_prob_raw(z, pc, mean, alpha, D, S, L, U, fusion)[0]
        again = ops.flow_log_prob_raw(z, pc, mean, alpha, D, S, L, U, fusion)[0]
        assert torch.equal(ref, again)
        err = _rel_err(low.cpu(), ref.cpu().double())
        assert 1e-6 < err < 3e-2, err
    # autoregressive flow (the LFI step's kernels)
    D, L, U = 6, 2, 12
    torch.manual_seed(8)
    np.random.seed(8)
    nf = tnf.NormFlow(D, False, "AR", 1, L, U)
    with torch.no_grad():
        nf(256)
        za = torch.randn(1, 2048, D).cuda()
        ref = nf.log_prob(za)
        with ops.operand_precision("bf16"):
            low = nf.log_prob(za)
        assert torch.equal(ref, nf.log_prob(za))
    err = _rel_err(low.cpu(), ref.cpu().double())
    assert 1e-6 < err < 3e-2, err
