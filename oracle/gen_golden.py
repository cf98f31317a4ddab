"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

Usage (cwd must be outside the repo so nothing shadows the reference's
namespace package):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python3 /root/repo/oracle/gen_golden.py

Every fixture holds the inputs AND the reference's outputs, so the GPU box
(which never sees /root/reference) needs nothing but the .npz files.  While
generating, each case is also replayed through oracle/flow_oracle.py and must
match the reference exactly (same aten ops, same order) -- that is what pins
the oracle.  The reference is only imported and called here; no source text
of it is stored anywhere.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

import torch_nf.bijectors as rb  # the reference (PYTHONPATH=/root/reference)
import torch_nf.density_estimator as rde
from torch_nf.conditional_density_estimator import ConditionalDensityEstimator as RefCDE

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
spec = importlib.util.spec_from_file_location("flow_oracle", os.path.join(HERE, "flow_oracle.py"))
orc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(orc)

assert "/root/reference" in rb.__file__, rb.__file__
torch.set_num_threads(1)  # deterministic reduction order for the fixtures


def same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and bool(torch.equal(a, b))


def tdtype(name):
    return torch.float32 if name == "f32" else torch.float64


def npy(t):
    return t.detach().numpy().copy()


# ---------------------------------------------------------------------------
# 1. RealNVP coupling layers
# ---------------------------------------------------------------------------
COUPLING_CASES = [
    # D, L, U, upper, Mz, Mp, N, dtype, extra, sigma
    (4, 2, 15, True, 3, 3, 5, "f64", 0, 0.1),
    (4, 2, 15, False, 3, 3, 5, "f64", 10, 0.1),
    (5, 1, 15, False, 3, 3, 5, "f64", 0, 0.1),
    (5, 1, 15, True, 3, 3, 5, "f32", 0, 0.1),
    (5, 2, 15, False, 2, 2, 9, "f32", 3, 0.1),
    (8, 1, 15, False, 3, 3, 5, "f64", 0, 0.1),
    (2, 2, 15, True, 1, 1, 257, "f32", 0, 0.1),
    (2, 2, 15, False, 1, 1, 257, "f32", 0, 0.1),
    (32, 2, 15, True, 1, 1, 257, "f32", 0, 0.1),
    (32, 2, 15, False, 1, 1, 257, "f32", 0, 0.1),
    (64, 2, 15, True, 1, 1, 257, "f32", 0, 0.1),
    (64, 2, 15, False, 1, 1, 257, "f32", 0, 0.1),
    (64, 2, 15, True, 1, 1, 257, "f32", 7, 0.02),
    (64, 1, 16, True, 1, 1, 64, "f32", 0, 0.1),
    (64, 3, 15, False, 1, 1, 64, "f32", 0, 0.1),
    (64, 5, 15, True, 1, 1, 33, "f32", 0, 0.1),
    (64, 2, 64, True, 1, 1, 33, "f32", 0, 0.05),
    (32, 3, 64, False, 3, 1, 5, "f32", 0, 0.05),
    (64, 2, 15, True, 3, 3, 40, "f32", 0, 0.1),
    (64, 2, 15, False, 3, 1, 40, "f32", 0, 0.1),
    (32, 2, 15, True, 5, 5, 1, "f32", 0, 0.1),
    (16, 2, 20, True, 2, 2, 19, "f32", 0, 0.1),
    (64, 2, 15, False, 1, 1, 100, "f64", 0, 0.1),
]


def gen_coupling():
    rng = np.random.RandomState(1234)
    out = {}
    meta = []
    for ci, (D, L, U, upper, Mz, Mp, N, dt, extra, sigma) in enumerate(COUPLING_CASES):
        layer = rb.RealNVP(D, L, U, transform_upper=upper)
        n_par = layer.count_num_params()
        assert n_par == orc.coupling_num_params(D, L, U, upper)
        params = torch.tensor(rng.normal(0.0, sigma, (Mp, n_par + extra))).to(tdtype(dt))
        z = torch.tensor(rng.normal(0.0, 1.0, (Mz, N, D))).to(tdtype(dt))
        zf, ldf = layer.forward_and_log_det(z, params)
        zi, ldi = layer.inverse_and_log_det(z, params)
        ozf, oldf = orc.coupling(z, params, D, L, U, upper, False)
        ozi, oldi = orc.coupling(z, params, D, L, U, upper, True)
        assert same(zf, ozf) and same(ldf, oldf) and same(zi, ozi) and same(ldi, oldi), ci
        k = "c%02d_" % ci
        out[k + "z"], out[k + "params"] = npy(z), npy(params)
        out[k + "z_fwd"], out[k + "ld_fwd"] = npy(zf), npy(ldf)
        out[k + "z_inv"], out[k + "ld_inv"] = npy(zi), npy(ldi)
        meta.append([D, L, U, int(upper), Mz, Mp, N, 0 if dt == "f32" else 1, extra])
    out["meta"] = np.array(meta, dtype=np.int64)
    np.savez(os.path.join(OUT, "coupling.npz"), **out)
    print("coupling: %d cases" % len(meta))


# ---------------------------------------------------------------------------
# 2. Affine and BatchNorm
# ---------------------------------------------------------------------------
def gen_affine_bn():
    rng = np.random.RandomState(99)
    out = {}
    aff_meta = []
    for ci, (D, Mz, Mp, N, dt, extra) in enumerate(
        [(4, 5, 5, 7, "f32", 0), (64, 1, 1, 33, "f32", 0), (64, 3, 1, 9, "f32", 2), (5, 2, 2, 3, "f64", 0)]
    ):
        layer = rb.Affine(D)
        params = torch.tensor(rng.normal(0.0, 1.0, (Mp, 2 * D + extra))).to(tdtype(dt))
        z = torch.tensor(rng.normal(0.0, 1.0, (Mz, N, D))).to(tdtype(dt))
        zf, ldf = layer.forward_and_log_det(z, params)
        zi, ldi = layer.inverse_and_log_det(z, params)
        ozf, oldf = orc.affine(z, params, D, False)
        ozi, oldi = orc.affine(z, params, D, True)
        assert same(zf, ozf) and same(ldf, oldf) and same(zi, ozi) and same(ldi, oldi)
        k = "a%02d_" % ci
        out[k + "z"], out[k + "params"] = npy(z), npy(params)
        out[k + "z_fwd"], out[k + "ld_fwd"] = npy(zf), npy(ldf)
        out[k + "z_inv"], out[k + "ld_inv"] = npy(zi), npy(ldi)
        aff_meta.append([D, Mz, Mp, N, 0 if dt == "f32" else 1, extra])
    out["affine_meta"] = np.array(aff_meta, dtype=np.int64)

    bn_meta = []
    for ci, (D, M, N, loc) in enumerate([(4, 4, 50, 10.0), (64, 1, 257, 0.5), (5, 3, 11, -3.0), (32, 2, 300, 2.0)]):
        layer = rb.BatchNorm(D)
        z = torch.tensor(rng.normal(loc, 1.0 + 0.5 * ci, (M, N, D))).float()
        zb, ldb = layer(z)  # batch statistics
        mean, alpha = layer.get_last_mean(), layer.get_last_alpha()
        ozb, oldb, omean, oalpha = orc.bn_forward_batch(z)
        assert same(zb, ozb) and same(ldb, oldb) and same(mean, omean) and same(alpha, oalpha)
        z2 = torch.tensor(rng.normal(loc, 1.0, (M, N, D))).float()
        zfz, ldfz = layer(z2, use_last=True)
        zinv, ldinv = layer.inverse_and_log_det(z2)
        a, b = orc.bn_forward_frozen(z2, mean, alpha)
        c, d = orc.bn_inverse(z2, mean, alpha)
        assert same(zfz, a) and same(ldfz, b) and same(zinv, c) and same(ldinv, d)
        k = "b%02d_" % ci
        out[k + "z"], out[k + "z_batch"], out[k + "ld_batch"] = npy(z), npy(zb), npy(ldb)
        out[k + "mean"], out[k + "alpha"] = npy(mean), npy(alpha)
        out[k + "z2"], out[k + "z_frozen"], out[k + "ld_frozen"] = npy(z2), npy(zfz), npy(ldfz)
        out[k + "z_inv"], out[k + "ld_inv"] = npy(zinv), npy(ldinv)
        bn_meta.append([D, M, N])
    out["bn_meta"] = np.array(bn_meta, dtype=np.int64)
    np.savez(os.path.join(OUT, "affine_bn.npz"), **out)
    print("affine: %d cases, bn: %d cases" % (len(aff_meta), len(bn_meta)))


# ---------------------------------------------------------------------------
# 3. NormFlow('coupling')
# ---------------------------------------------------------------------------
FLOW_CASES = [
    # D, S, L, U, N, init
    (2, 1, 2, 15, 257, "xavier"),
    (4, 1, 2, 15, 257, "n01"),
    (4, 2, 2, 20, 64, "n01"),
    (5, 1, 2, 15, 100, "n01"),
    (32, 4, 2, 15, 257, "xavier"),
    (32, 4, 2, 15, 257, "n01"),
    (64, 4, 2, 15, 257, "xavier"),
    (64, 4, 2, 15, 521, "n01"),
    (64, 2, 3, 16, 130, "n01"),
]


def flow_stats(nf):
    return [(b.get_last_mean(), b.get_last_alpha()) for b in nf.bijectors if b.name == "BatchNorm"]


def gen_flow():
    out = {}
    meta = []
    for ci, (D, S, L, U, N, init) in enumerate(FLOW_CASES):
        np.random.seed(ci)
        torch.manual_seed(ci)
        nf = rde.NormFlow(D, False, "coupling", S, L, U)
        assert nf.D_params == orc.flow_num_params(D, S, L, U)
        if init == "n01":
            nf.params = torch.tensor(np.random.normal(0.0, 0.1, (1, nf.D_params))).float().requires_grad_()
        params = nf.params.detach()
        # forward with batch statistics: replay the host draw to capture omega
        st = np.random.get_state()
        omega = np.random.normal(0.0, 1.0, (1, N, D))
        np.random.set_state(st)
        z, log_q = nf(N)
        stats = flow_stats(nf)
        oz, olog_q, ostats = orc.flow_forward(omega, params, D, S, L, U, None)
        assert same(z.detach(), oz) and same(log_q.detach(), olog_q)
        for (m, a), (om, oa) in zip(stats, ostats):
            assert same(m.detach(), om) and same(a.detach(), oa)
        # forward with frozen statistics on a fresh draw
        st = np.random.get_state()
        omega_fz = np.random.normal(0.0, 1.0, (1, N, D))
        np.random.set_state(st)
        z_fz, log_q_fz = nf(N, freeze_bn=True)
        dstats = [(m.detach(), a.detach()) for m, a in stats]
        oz_fz, olog_q_fz, _ = orc.flow_forward(omega_fz, params, D, S, L, U, dstats)
        assert same(z_fz.detach(), oz_fz) and same(log_q_fz.detach(), olog_q_fz)
        # density of fresh points
        z_test = torch.tensor(np.random.normal(0.0, 1.0, (1, N, D))).float()
        z0, sld = nf.inverse_and_log_det(z_test, nf.params)
        lp = nf.log_prob(z_test)
        oz0, osld = orc.flow_inverse(z_test, params, D, S, L, U, dstats)
        olp = orc.flow_log_prob(z_test, params, D, S, L, U, dstats)
        assert same(z0.detach(), oz0) and same(sld.detach(), osld) and same(lp.detach(), olp)
        # self-consistency of the flow: log_prob(forward sample) vs forward's log_q
        lp_self = nf.log_prob(z_fz.detach())
        k = "f%02d_" % ci
        out[k + "params"] = npy(params)
        out[k + "omega"], out[k + "z_fwd"], out[k + "logq_fwd"] = omega, npy(z), npy(log_q)
        out[k + "bn_mean"] = np.stack([npy(m) for m, _ in stats])
        out[k + "bn_alpha"] = np.stack([npy(a) for _, a in stats])
        out[k + "omega_fz"], out[k + "z_fz"], out[k + "logq_fz"] = omega_fz, npy(z_fz), npy(log_q_fz)
        out[k + "z_test"], out[k + "z0"], out[k + "sum_log_det"], out[k + "log_prob"] = (
            npy(z_test), npy(z0), npy(sld), npy(lp))
        out[k + "log_prob_self"] = npy(lp_self)
        meta.append([D, S, L, U, N])
        # gradients of the training loss for the metric-shaped model
        if (D, S, N) == (64, 4, 257):
            zt = z_test.clone().requires_grad_()
            p = params.clone().requires_grad_()
            nf.params = p
            loss = -torch.mean(nf.log_prob(zt))
            loss.backward()
            out[k + "loss"] = npy(loss)
            out[k + "grad_params"], out[k + "grad_z"] = npy(p.grad), npy(zt.grad)
            p2 = params.clone().requires_grad_()
            z2 = z_test.clone().requires_grad_()
            oloss = -torch.mean(orc.flow_log_prob(z2, p2, D, S, L, U, dstats))
            oloss.backward()
            assert same(p.grad, p2.grad) and same(zt.grad, z2.grad)
    out["meta"] = np.array(meta, dtype=np.int64)
    np.savez(os.path.join(OUT, "flow.npz"), **out)
    print("flow: %d cases" % len(meta))


# ---------------------------------------------------------------------------
# 4. ConditionalDensityEstimator
# ---------------------------------------------------------------------------
def gen_cde():
    out = {}
    meta = []
    cases = [(4, 1, 2, 20, 10, [50, 100], 20, 50), (4, 1, 2, 20, 10, [50, 100], 1, 50),
             (64, 4, 2, 15, 10, [50, 100], 4, 50), (2, 1, 2, 15, 2, [100], 64, 1)]
    for ci, (D, S, L, U, D_x, hidden, M, N) in enumerate(cases):
        np.random.seed(100 + ci)
        torch.manual_seed(100 + ci)
        nf = rde.NormFlow(D, True, "coupling", S, L, U)
        cde = RefCDE(nf, D_x, hidden)
        x = torch.tensor(np.random.normal(0.0, 1.0, (M, D_x))).float()
        with torch.no_grad():
            params = cde.param_net(x)
        st = np.random.get_state()
        omega = np.random.normal(0.0, 1.0, (M, N, D))
        np.random.set_state(st)
        z, log_q = cde(x, N=N)
        stats = [(m.detach(), a.detach()) for m, a in flow_stats(nf)]
        oz, olog_q, _ = orc.flow_forward(omega, params, D, S, L, U, None)
        assert same(z.detach(), oz) and same(log_q.detach(), olog_q)
        z_test = torch.tensor(np.random.normal(0.0, 1.0, (M, N, D))).float()
        lp = cde.log_prob(z_test, x)
        olp = orc.flow_log_prob(z_test, params, D, S, L, U, stats)
        assert same(lp.detach(), olp)
        k = "d%02d_" % ci
        out[k + "x"], out[k + "params"] = npy(x), npy(params)
        out[k + "omega"], out[k + "z_fwd"], out[k + "logq_fwd"] = omega, npy(z), npy(log_q)
        out[k + "bn_mean"] = np.stack([npy(m) for m, _ in stats])
        out[k + "bn_alpha"] = np.stack([npy(a) for _, a in stats])
        out[k + "z_test"], out[k + "log_prob"] = npy(z_test), npy(lp)
        if D <= 4:  # the param_net weights are small enough to ship
            for name, t in cde.state_dict().items():
                out[k + "sd_" + name] = npy(t)
        meta.append([D, S, L, U, D_x, len(hidden), M, N] + hidden + [0] * (2 - len(hidden)))
    out["meta"] = np.array(meta, dtype=np.int64)
    np.savez(os.path.join(OUT, "cde.npz"), **out)
    print("cde: %d cases" % len(meta))


# ---------------------------------------------------------------------------
# 5. MAF and NormFlow('AR')
# ---------------------------------------------------------------------------
def gen_maf():
    out = {}
    meta = []
    cases = [(4, 2, 15, True, 3, 3, 7, "f64"), (5, 1, 20, True, 2, 2, 9, "f64"), (4, 3, 15, False, 2, 2, 5, "f64"),
             (8, 2, 20, True, 1, 1, 65, "f32"), (6, 2, 15, True, 3, 1, 11, "f32"), (16, 2, 32, True, 1, 1, 33, "f32")]
    for ci, (D, L, U, fwd_fac, Mz, Mp, N, dt) in enumerate(cases):
        np.random.seed(200 + ci)
        st = np.random.get_state()
        layer = rb.MAF(D, L, U, fwd_fac=fwd_fac)
        np.random.set_state(st)
        ms, Ms = orc.maf_masks(D, L, U, fwd_fac)  # same RNG stream -> same masks
        assert all(np.array_equal(a, b) for a, b in zip(ms, layer.ms))
        assert all(np.array_equal(a, b[0].numpy()) for a, b in zip(Ms, layer.Ms))
        n_par = layer.count_num_params()
        assert n_par == orc.maf_num_params(D, L, U)
        rng = np.random.RandomState(300 + ci)
        params = torch.tensor(rng.normal(0.0, 0.3, (Mp, n_par))).to(tdtype(dt))
        z = torch.tensor(rng.normal(0.0, 1.0, (Mz, N, D))).to(tdtype(dt))
        if dt == "f64":
            layer.Ms = [m.double() for m in layer.Ms]  # the reference's float32 masks promote anyway
        zf, ldf = layer.forward_and_log_det(z, params)
        zi, ldi = layer.inverse_and_log_det(z, params)
        ozf, oldf = orc.maf(z, params, D, L, U, Ms, False)
        ozi, oldi = orc.maf(z, params, D, L, U, Ms, True)
        assert same(zf, ozf) and same(ldf, oldf) and same(zi, ozi) and same(ldi, oldi), ci
        k = "m%02d_" % ci
        out[k + "z"], out[k + "params"] = npy(z), npy(params)
        for i, m in enumerate(ms):
            out[k + "ms%d" % i] = np.asarray(m)
        out[k + "z_fwd"], out[k + "ld_fwd"] = npy(zf), npy(ldf)
        out[k + "z_inv"], out[k + "ld_inv"] = npy(zi), npy(ldi)
        meta.append([D, L, U, int(fwd_fac), Mz, Mp, N, 0 if dt == "f32" else 1])
    out["meta"] = np.array(meta, dtype=np.int64)

    # NormFlow(arch_type="AR"): forward (batch stats), frozen forward, log_prob, gradients
    fmeta = []
    for ci, (D, L, U, N) in enumerate([(4, 2, 20, 50), (8, 2, 15, 129), (16, 1, 32, 64)]):
        np.random.seed(400 + ci)
        torch.manual_seed(400 + ci)
        st = np.random.get_state()
        nf = rde.NormFlow(D, False, "AR", 1, L, U)
        np.random.set_state(st)
        ms, Ms = orc.maf_masks(D, L, U, True)
        nf.params = torch.tensor(np.random.normal(0.0, 0.2, (1, nf.D_params))).float().requires_grad_()
        params = nf.params.detach()
        st = np.random.get_state()
        omega = np.random.normal(0.0, 1.0, (1, N, D))
        np.random.set_state(st)
        z, log_q = nf(N)
        bn = nf.bijectors[1]
        stat = (bn.get_last_mean().detach(), bn.get_last_alpha().detach())
        oz, olq, ostat = orc.ar_flow_forward(omega, params, D, L, U, Ms, None)
        assert same(z.detach(), oz) and same(log_q.detach(), olq) and same(stat[0], ostat[0].detach())
        z_test = torch.tensor(np.random.normal(0.0, 1.0, (1, N, D))).float()
        lp = nf.log_prob(z_test)
        olp = orc.ar_flow_log_prob(z_test, params, D, L, U, Ms, stat)
        assert same(lp.detach(), olp)
        zt = z_test.clone().requires_grad_()
        p = params.clone().requires_grad_()
        nf.params = p
        loss = -torch.mean(nf.log_prob(zt))
        loss.backward()
        k = "n%02d_" % ci
        out[k + "params"], out[k + "omega"] = npy(params), omega
        for i, m in enumerate(ms):
            out[k + "ms%d" % i] = np.asarray(m)
        out[k + "z_fwd"], out[k + "logq_fwd"] = npy(z), npy(log_q)
        out[k + "bn_mean"], out[k + "bn_alpha"] = npy(stat[0]), npy(stat[1])
        out[k + "z_test"], out[k + "log_prob"] = npy(z_test), npy(lp)
        out[k + "loss"], out[k + "grad_params"], out[k + "grad_z"] = npy(loss), npy(p.grad), npy(zt.grad)
        fmeta.append([D, L, U, N])
    out["flow_meta"] = np.array(fmeta, dtype=np.int64)
    np.savez(os.path.join(OUT, "maf.npz"), **out)
    print("maf: %d bijector cases, %d AR flows" % (len(meta), len(fmeta)))


def gen_support():
    """ToInterval / ToSimplex alone and as NormFlow support layers (forward, log_prob, gradients)."""
    out = {}
    inf = np.inf
    bounds = [
        (4, [-inf] * 4, [inf] * 4, "f64", 1.0),
        (4, [-0.5, -inf, -0.5, -inf], [0.5, 0.5, inf, inf], "f64", 2.0),       # the reference test's mix
        (4, [-0.5, -inf, -0.5, -inf], [0.5, 0.5, inf, inf], "f32", 1.0),
        (6, [-1.0, 0.0, -inf, 2.0, -3.0, -inf], [1.0, inf, 0.25, 7.5, inf, inf], "f32", 1.0),
        (64, list(np.where(np.arange(64) % 3 == 0, -inf, -1.0 - np.arange(64) / 16.0)),
         list(np.where(np.arange(64) % 4 == 1, inf, 1.0 + np.arange(64) / 8.0)), "f32", 1.0),
    ]
    meta = []
    for ci, (D, lb, ub, dt, sig) in enumerate(bounds):
        layer = rb.ToInterval(D, np.array(lb, dtype=np.float64), np.array(ub, dtype=np.float64))
        consts = orc.interval_consts(lb, ub)
        for a, b in zip(consts, (layer.tanh_flg, layer.softplus_flg, layer.tanh_m, layer.tanh_c, layer.softplus_m,
                                 layer.softplus_c)):
            assert same(a, b)
        rng = np.random.RandomState(500 + ci)
        M, N = (3, 17) if D < 64 else (2, 130)
        z = torch.tensor(rng.normal(0.0, sig, (M, N, D))).to(tdtype(dt))
        zf, ldf = layer(z)
        zi, ldi = layer.inverse_and_log_det(zf)
        ozf, oldf = orc.to_interval(z, consts, False)
        ozi, oldi = orc.to_interval(zf, consts, True)
        assert same(zf, ozf) and same(ldf, oldf) and same(zi, ozi) and same(ldi, oldi), ci
        # gradients of a fixed linear functional of (out, log_det), both directions
        wz, wl = torch.tensor(rng.normal(0, 1, (M, N, D))).to(z.dtype), torch.tensor(rng.normal(0, 1, (M, N))).to(z.dtype)
        g = []
        for inv, x in ((False, z), (True, zf.detach())):
            xr = x.clone().requires_grad_()
            o, l = layer.inverse_and_log_det(xr) if inv else layer(xr)
            ((o * wz).sum() + (l * wl).sum()).backward()
            g.append(xr.grad)
        k = "i%02d_" % ci
        out[k + "lb"], out[k + "ub"] = np.array(lb, dtype=np.float64), np.array(ub, dtype=np.float64)
        out[k + "z"], out[k + "z_fwd"], out[k + "ld_fwd"] = npy(z), npy(zf), npy(ldf)
        out[k + "z_inv"], out[k + "ld_inv"] = npy(zi), npy(ldi)
        out[k + "wz"], out[k + "wl"], out[k + "g_fwd"], out[k + "g_inv"] = npy(wz), npy(wl), npy(g[0]), npy(g[1])
        meta.append([D, M, N, 0 if dt == "f32" else 1])
    out["interval_meta"] = np.array(meta, dtype=np.int64)

    smeta = []
    for ci, (Din, Dattr, M, N, dt) in enumerate([(3, 4, 20, 50, "f32"), (4, 4, 3, 9, "f64"), (64, 64, 2, 70, "f32"),
                                                 (7, 7, 1, 33, "f32")]):
        layer = rb.ToSimplex(Dattr)
        rng = np.random.RandomState(600 + ci)
        z = torch.tensor(rng.normal(0.0, 1.0, (M, N, Din))).to(tdtype(dt))
        zr = z.clone().requires_grad_()
        zf, ldf = layer(zr)
        ozf, oldf = orc.to_simplex(z, Dattr)
        assert same(zf.detach(), ozf) and same(ldf.detach(), oldf), ci
        wz = torch.tensor(rng.normal(0, 1, (M, N, Din + 1))).to(z.dtype)
        wl = torch.tensor(rng.normal(0, 1, (M, N))).to(z.dtype)
        ((zf * wz).sum() + (ldf * wl).sum()).backward()
        k = "s%02d_" % ci
        out[k + "z"], out[k + "z_fwd"], out[k + "ld_fwd"] = npy(z), npy(zf), npy(ldf)
        out[k + "wz"], out[k + "wl"], out[k + "g_fwd"] = npy(wz), npy(wl), npy(zr.grad)
        smeta.append([Din, Dattr, M, N, 0 if dt == "f32" else 1])
    out["simplex_meta"] = np.array(smeta, dtype=np.int64)

    # NormFlow with a support layer: coupling + ToInterval (the LFI scripts' configuration,
    # scripts/lfi_mat.py:37-40) and coupling + ToSimplex (tests/test_density_estimators.py:213)
    fmeta = []
    for ci, (D, S, L, U, N, kind) in enumerate([(6, 1, 2, 15, 40, "interval"), (32, 2, 2, 15, 64, "interval"),
                                                (4, 2, 2, 20, 30, "simplex")]):
        np.random.seed(700 + ci)
        torch.manual_seed(700 + ci)
        if kind == "interval":
            lb = np.where(np.arange(D) % 3 == 0, -inf, -2.0 - np.arange(D) / 8.0)
            ub = np.where(np.arange(D) % 3 == 1, inf, 2.5 + np.arange(D) / 4.0)
            sup = rb.ToInterval(D, lb, ub)
        else:
            lb = ub = np.zeros(0)
            sup = rb.ToSimplex(D)
        nf = rde.NormFlow(D, False, "coupling", S, L, U, sup)
        nf.params = torch.tensor(np.random.normal(0.0, 0.15, (1, nf.D_params))).float().requires_grad_()
        params = nf.params.detach()
        st = np.random.get_state()
        omega = np.random.normal(0.0, 1.0, (1, N, D))
        np.random.set_state(st)
        z, log_q = nf(N)
        bns = [b for b in nf.bijectors if b.name == "BatchNorm"]
        k = "f%02d_" % ci
        out[k + "lb"], out[k + "ub"] = lb, ub
        out[k + "params"], out[k + "omega"] = npy(params), omega
        out[k + "z_fwd"], out[k + "logq_fwd"] = npy(z), npy(log_q)
        out[k + "bn_mean"] = np.stack([npy(b.get_last_mean()) for b in bns])
        out[k + "bn_alpha"] = np.stack([npy(b.get_last_alpha()) for b in bns])
        st = np.random.get_state()
        omega2 = np.random.normal(0.0, 1.0, (1, N, D))
        np.random.set_state(st)
        with torch.no_grad():
            z2, log_q2 = nf(N, freeze_bn=True)   # frozen statistics (new base draw)
        out[k + "omega_frozen"], out[k + "z_frozen"], out[k + "logq_frozen"] = omega2, npy(z2), npy(log_q2)
        if kind == "interval":
            z_test = z.detach().clone()          # points inside the support
            lp = nf.log_prob(z_test)
            p = params.clone().requires_grad_()
            nf.params = p
            loss = -torch.mean(nf.log_prob(z_test))
            loss.backward()
            out[k + "log_prob"], out[k + "loss"], out[k + "grad_params"] = npy(lp), npy(loss), npy(p.grad)
        fmeta.append([D, S, L, U, N, 0 if kind == "interval" else 1])
    out["flow_meta"] = np.array(fmeta, dtype=np.int64)
    np.savez(os.path.join(OUT, "support.npz"), **out)
    print("support: %d interval, %d simplex, %d flows" % (len(meta), len(smeta), len(fmeta)))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = os.environ.get("GOLDEN_ONLY")  # e.g. GOLDEN_ONLY=maf regenerates one fixture file
    for name, fn in (("coupling", gen_coupling), ("affine_bn", gen_affine_bn), ("flow", gen_flow), ("cde", gen_cde),
                     ("maf", gen_maf), ("support", gen_support)):
        if only is None or only == name:
            fn()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("total fixture bytes: %d" % tot)
