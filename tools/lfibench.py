#!/usr/bin/env python3
"""The inner step of the reference's LFI scripts (scripts/lfi_mat.py:23-57: autoregressive flow with a
ToInterval support layer, conditioned through param_net [64, 64], M = 2000 contexts x M_atom = 100
samples): loss = -mean(cde.log_prob(z, x)); loss.backward().  Times this package on the GPU and, with
--cpu, the same step through the CPU oracle (= the reference's arithmetic under torch autograd).
The APT driver and the Mat simulator themselves are not part of the snapshot."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_nf_amd as tnf  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--d", type=int, default=3)       # Mat(d): D = d (d + 1) / 2
ap.add_argument("--M", type=int, default=2000)
ap.add_argument("--atoms", type=int, default=100)
ap.add_argument("--cpu", action="store_true")
args = ap.parse_args()

D = args.d * (args.d + 1) // 2
D_x = args.d
np.random.seed(1)
torch.manual_seed(1)
lb, ub = -2.0 * np.ones(D), 2.0 * np.ones(D)
lb[::2] = -np.inf
nf = tnf.NormFlow(D, True, "AR", 1, 2, 2 * D, tnf.ToInterval(D, lb, ub))
cde = tnf.ConditionalDensityEstimator(nf, D_x, [64, 64])
M, N = args.M, args.atoms
x = torch.randn(M, D_x, device="cuda")
z = (torch.rand(M, N, D, device="cuda") * 3.0 - 1.5)
print("D=%d D_params=%d M=%d N=%d" % (D, nf.D_params, M, N))


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def infer():
    with torch.no_grad():
        cde.log_prob(z, x)


def train():
    cde.zero_grad()
    (-cde.log_prob(z, x).mean()).backward()


ti, tt = timeit(infer, 10), timeit(train, 5)
print("GPU: log_prob %.3f ms (%.1f M samples/s)   train step %.3f ms (%.1f M samples/s)"
      % (ti * 1e3, M * N / ti / 1e6, tt * 1e3, M * N / tt / 1e6))
with torch.no_grad():
    zs, lq = cde(x[:1], N=M * N, freeze_bn=True)
ts = timeit(lambda: cde(x[:1], N=M * N, freeze_bn=True), 5)
print("GPU: posterior sampling cde(x0, N=%d) %.3f ms (%.1f M samples/s)" % (M * N, ts * 1e3, M * N / ts / 1e6))

if args.cpu:
    import importlib.util
    spec = importlib.util.spec_from_file_location("flow_oracle", os.path.join(ROOT, "oracle", "flow_oracle.py"))
    orc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(orc)
    net = torch.nn.Sequential(*[m for m in cde.param_net]).cpu()
    Ms = [Mk[0].numpy() for Mk in nf.bijectors[0].Ms]
    consts = orc.interval_consts(lb, ub)
    stat = (nf.bijectors[1].get_last_mean().cpu().float(), nf.bijectors[1].get_last_alpha().cpu().float())
    xc, zc = x.cpu(), z.cpu()

    def cpu_step():
        for p in net.parameters():
            p.grad = None
        zi, ld = orc.to_interval(zc, consts, True)
        lp = orc.ar_flow_log_prob(zi, net(xc), D, nf.num_layers, nf.num_units, Ms, stat) - ld
        (-lp.mean()).backward()

    cpu_step()
    t0 = time.perf_counter()
    for _ in range(3):
        cpu_step()
    tc = (time.perf_counter() - t0) / 3
    print("CPU oracle (%d threads): train step %.1f ms (%.2f M samples/s)  -> GPU %.0fx" %
          (torch.get_num_threads(), tc * 1e3, M * N / tc / 1e6, tc / tt))
    cde.param_net.cuda()
