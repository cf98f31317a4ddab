#!/usr/bin/env python3
"""BASELINE configs[2]: ConditionalDensityEstimator(NormFlow(64, True, 'coupling', 4, 2, 15), D_x=32, [64, 64]):
cde(x, N, freeze_bn=True) then cde.log_prob(z, x) for (M, N) with M*N = 2^20 (SURVEY 8d cfg 3)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_nf_amd as tnf  # noqa: E402

D, S, L, U, D_x = 64, 4, 2, 15, 32
torch.manual_seed(0)
np.random.seed(0)
nf = tnf.NormFlow(D, True, "coupling", S, L, U)
cde = tnf.ConditionalDensityEstimator(nf, D_x, [64, 64])
with torch.no_grad():
    for p in cde.param_net.parameters():
        p.mul_(0.3)
    for (M, N) in [(16, 1 << 16), (1, 1 << 20), (2048, 512), (1 << 14, 64), (1 << 17, 8), (1 << 20, 1)]:
        x = torch.randn(M, D_x, device="cuda")
        params = cde.param_net(x)
        omega = torch.randn(M, N, D, device="cuda")
        mean, alpha = nf._bn_stats(torch.device("cuda"))
        def fwd():
            return nf._forward_from(omega, params, freeze_bn=True)
        def lp(z):
            return nf.log_prob(z, params)
        z, sld = fwd()
        out = lp(z)
        torch.cuda.synchronize()
        res = []
        for fn in (fwd, lambda: lp(z), lambda: cde.param_net(x)):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 10)
        print("M=%7d N=%7d: forward %8.3f ms  log_prob %8.3f ms  (param_net %6.3f ms)  -> log_prob %7.1f M samples/s"
              % (M, N, res[0] * 1e3, res[1] * 1e3, res[2] * 1e3, M * N / res[1] / 1e6))
