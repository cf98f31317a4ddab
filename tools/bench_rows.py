"""Rows of bench.py's JSON line that sit outside the timed headline region: BASELINE configs[2] and the single-GPU
extras for the SURVEY 8f rows.  (The CPU baseline -- the only leg that may touch oracle/ -- lives in bench.py itself.)"""
import time

import numpy as np
import torch


SETTLE_MS = 100.0  # back-to-back calls before timing: the GPU's clock needs ~50 ms of load to settle (bench.py `settle`)


def timeit(fn, reps, warm=2):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < SETTLE_MS:
        fn()
        torch.cuda.synchronize()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def config2_row(tnf, dev):
    """BASELINE configs[2] (SURVEY 8d cfg 3): ConditionalDensityEstimator(NormFlow(64, True, 'coupling', 4, 2, 15),
    D_x = 32, [64, 64]), (M, N) = (16, 2^16): `cde(x, N, freeze_bn=True)` then `cde.log_prob(z, x)`.  The base draw is
    resident in HBM like the headline's input (the reference-compatible host draw, np.random.normal of 6.7e7 doubles +
    PCIe, is timed separately as `host_rng_forward_ms`)."""
    np.random.seed(0)
    torch.manual_seed(0)
    D, S, L, U, M, N = 64, 4, 2, 15, 16, 1 << 16
    nf = tnf.NormFlow(D, True, "coupling", S, L, U, device=dev)
    cde = tnf.ConditionalDensityEstimator(nf, 32, [64, 64])
    x = torch.randn(M, 32, device=dev)
    omega = torch.randn(M, N, D, device=dev)
    with torch.no_grad():
        cde(x, 256)  # populate the BatchNorm statistics (batch mode)
        z, _ = nf._forward_from(omega, cde._params_for(x), freeze_bn=True)

        def fwd():
            nf._forward_from(omega, cde._params_for(x), freeze_bn=True)

        def lpf():
            cde.log_prob(z, x)

        t_f = timeit(fwd, 10)
        t_l = timeit(lpf, 10)
        t0 = time.perf_counter()
        cde(x, N, freeze_bn=True)
        torch.cuda.synchronize()
        t_host = time.perf_counter() - t0
    tot = M * N
    return {"what": "ConditionalDensityEstimator(NormFlow(64,True,'coupling',4,2,15), D_x=32, [64,64]), (M, N) = (16, 2^16): "
                    "forward = param_net + whole-flow kernel with per-context rows (base draw resident in HBM, frozen "
                    "statistics, float64 log_q); log_prob = cde.log_prob(z, x)",
            "samples": tot, "forward_ms": round(t_f * 1e3, 4), "log_prob_ms": round(t_l * 1e3, 4),
            "value": round(tot / (t_f + t_l) / 1e6, 1), "unit": "M samples/s (forward + log_prob)",
            "forward_value": round(tot / t_f / 1e6, 1), "log_prob_value": round(tot / t_l / 1e6, 1),
            "host_rng_forward_ms": round(t_host * 1e3, 1)}


def widened_rows(tnf):
    """Single-GPU extras for the SURVEY 8f rows built after the metric path: not the
    metric, a few seconds in total."""
    res = {}
    np.random.seed(0)
    torch.manual_seed(0)
    nf = tnf.NormFlow(64, True, "coupling", 4, 2, 15)
    cde = tnf.ConditionalDensityEstimator(nf, 32, [64, 64])
    M = 1 << 18
    x = torch.randn(M, 32, device="cuda")
    z = torch.randn(M, 1, 64, device="cuda")

    def infer():
        with torch.no_grad():
            cde.log_prob(z, x)

    def train():
        cde.zero_grad()
        (-cde.log_prob(z, x).mean()).backward()

    t = timeit(infer, 5)
    res["cond_flow_log_prob"] = {"contexts": M, "ms": round(t * 1e3, 3), "value": round(M / t / 1e6, 1),
                                 "unit": "M contexts/s", "what": "cde.log_prob(z[:, None, :], x), D=64 S=4, param_net "
                                 "[64,64]: last Linear fused into the flow kernel, params (M, 20464) never materialised"}
    t = timeit(train, 3)
    res["cond_flow_train_step"] = {"contexts": M, "ms": round(t * 1e3, 3), "value": round(M / t / 1e6, 2),
                                   "unit": "M contexts/s", "what": "forward + backward through param_net (fused pair; hidden weight "
                                   "gradients of param_net as split-K products)"}
    del x, z, cde, nf
    nf = tnf.NormFlow(16, False, "AR", 1, 2, 32)
    z = torch.randn(1, 1 << 20, 16, device="cuda")
    with torch.no_grad():
        nf(64)
        t = timeit(lambda: nf.log_prob(z), 10)
    res["ar_log_prob"] = {"samples": 1 << 20, "ms": round(t * 1e3, 3), "value": round((1 << 20) / t / 1e6, 1),
                          "unit": "M samples/s", "what": "NormFlow(16, arch_type='AR', num_layers=2, num_units=32).log_prob, "
                          "one matrix-pipe MAF kernel"}
    del z, nf
    # the LFI scripts' inner step (scripts/lfi_mat.py:23-57): AR flow + ToInterval through param_net [64,64]
    D_l, M_l, N_l = 6, 2000, 100
    lb, ub = -2.0 * np.ones(D_l), 2.0 * np.ones(D_l)
    lb[::2] = -np.inf
    nfl = tnf.NormFlow(D_l, True, "AR", 1, 2, 2 * D_l, tnf.ToInterval(D_l, lb, ub))
    cdel = tnf.ConditionalDensityEstimator(nfl, 3, [64, 64])
    xl = torch.randn(M_l, 3, device="cuda")
    zl = torch.rand(M_l, N_l, D_l, device="cuda") * 3.0 - 1.5
    optl = torch.optim.Adam(cdel.parameters(), lr=1e-3, capturable=True)

    def lfi_step():
        optl.zero_grad(set_to_none=True)
        loss = -cdel.log_prob(zl, xl).mean()
        loss.backward()
        optl.step()
        return loss.detach()

    te = timeit(lfi_step, 10)
    gs = tnf.graphs.GraphedStep(lfi_step, warmup=3)
    tg = timeit(gs, 20)
    res["lfi_train_step"] = {"samples": M_l * N_l, "ms": round(tg * 1e3, 3), "eager_ms": round(te * 1e3, 3),
                             "value": round(M_l * N_l / tg / 1e6, 1), "unit": "M samples/s",
                             "what": "AR flow (D=6) + ToInterval conditioned through param_net [64,64], 2000 contexts x 100 "
                             "samples: loss, one-kernel AR backward, Adam; replayed as one HIP graph (eager_ms: eagerly)"}
    del cdel, nfl, xl, zl, optl, gs
    # sampling with fresh batch statistics under autograd (the reference's train_efn objective shape)
    nfe = tnf.NormFlow(64, False, "coupling", 4, 2, 15)
    nfe.params = (torch.randn(1, nfe.D_params, device="cuda") * 0.1).requires_grad_()
    om = torch.randn(1, 1 << 19, 64, device="cuda")

    def efn():
        nfe.params.grad = None
        ze, lqe = nfe._forward_from(om, nfe.params, freeze_bn=False)
        (lqe.mean() + (ze ** 2).mean()).backward()

    t = timeit(efn, 5)
    res["forward_train_step"] = {"samples": 1 << 19, "ms": round(t * 1e3, 3), "value": round((1 << 19) / t / 1e6, 1),
                                 "unit": "M samples/s", "what": "z, log_q = nf(N) with fresh batch statistics, D=64 S=4; "
                                 "backward through the batch moments (one autograd node, tnf_flow_forward_train_*)"}
    del nfe, om
    # flows wider than the one-kernel shapes (num_units = 64; MAF at D = 64): per-layer kernels, two-pass MFMA backward
    for key, arch, stages in (("wide_flow_train_step", "coupling", 4), ("wide_ar_flow_train_step", "AR", 2)):
        np.random.seed(0)
        nfw = tnf.NormFlow(64, False, arch, stages, 2, 64)
        zw = torch.randn(1, 1 << 18, 64, device="cuda")
        with torch.no_grad():
            nfw(4096)

        def wide():
            nfw.params.grad = None
            (-nfw.log_prob(zw).mean()).backward()

        t = timeit(wide, 3)
        res[key] = {"samples": 1 << 18, "ms": round(t * 1e3, 3), "value": round((1 << 18) / t / 1e6, 1), "unit": "M samples/s",
                    "what": "NormFlow(64, False, %r, %d, 2, num_units=64): -mean(log_prob) forward + backward; wide "
                            "fp32-MFMA layer kernels, backward = records -> sample-contracting GEMM -> ordered reduce "
                            "(coupling_wide_bwd.hip)" % (arch, stages)}
        del nfw, zw
    return res
