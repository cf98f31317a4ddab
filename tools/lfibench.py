#!/usr/bin/env python3
"""The inner step of the reference's LFI scripts (scripts/lfi_mat.py:23-57: autoregressive flow with a
ToInterval support layer, conditioned through param_net [64, 64], M = 2000 contexts x M_atom = 100
samples): loss = -mean(cde.log_prob(z, x)); loss.backward().  Times this package on the GPU;
tests/lfi_cpu_baseline.py times the same step through the CPU oracle beside it.
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


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def run(d=3, M=2000, atoms=100):
    """Build the lfi_mat model, time log_prob / training step / posterior sampling on the GPU and
    return everything a caller needs to repeat the step elsewhere."""
    D = d * (d + 1) // 2
    D_x = d
    np.random.seed(1)
    torch.manual_seed(1)
    lb, ub = -2.0 * np.ones(D), 2.0 * np.ones(D)
    lb[::2] = -np.inf
    nf = tnf.NormFlow(D, True, "AR", 1, 2, 2 * D, tnf.ToInterval(D, lb, ub))
    cde = tnf.ConditionalDensityEstimator(nf, D_x, [64, 64])
    N = atoms
    x = torch.randn(M, D_x, device="cuda")
    z = (torch.rand(M, N, D, device="cuda") * 3.0 - 1.5)
    print("D=%d D_params=%d M=%d N=%d" % (D, nf.D_params, M, N))

    def infer():
        with torch.no_grad():
            cde.log_prob(z, x)

    def train():
        cde.zero_grad()
        (-cde.log_prob(z, x).mean()).backward()

    ti, tt = timeit(infer, 10), timeit(train, 5)
    print("GPU: log_prob %.3f ms (%.1f M samples/s)   train step %.3f ms (%.1f M samples/s)"
          % (ti * 1e3, M * N / ti / 1e6, tt * 1e3, M * N / tt / 1e6))
    # the same step with Adam, eagerly and replayed as one HIP graph (graphs.GraphedStep, what lfi.train_APT does)
    opt = torch.optim.Adam(cde.parameters(), lr=1e-3, capturable=True)

    def full_step():
        opt.zero_grad(set_to_none=True)
        loss = -cde.log_prob(z, x).mean()
        loss.backward()
        opt.step()
        return loss.detach()

    te = timeit(full_step, 10)
    gs = tnf.graphs.GraphedStep(full_step, warmup=3)
    tg = timeit(gs, 20)
    print("GPU: step + Adam eager %.3f ms, as one HIP graph %.3f ms (%.1f M samples/s)"
          % (te * 1e3, tg * 1e3, M * N / tg / 1e6))
    with torch.no_grad():
        cde(x[:1], N=M * N, freeze_bn=True)
    ts = timeit(lambda: cde(x[:1], N=M * N, freeze_bn=True), 5)
    print("GPU: posterior sampling cde(x0, N=%d) %.3f ms (%.1f M samples/s)" % (M * N, ts * 1e3, M * N / ts / 1e6))
    td = timeit(lambda: cde.sample(x[:1], N=M * N), 5)
    print("GPU: posterior sampling cde.sample(x0, N=%d), device-side draw: %.3f ms (%.1f M samples/s)"
          % (M * N, td * 1e3, M * N / td / 1e6))
    return dict(D=D, M=M, N=N, lb=lb, ub=ub, nf=nf, cde=cde, x=x, z=z, train_s=tt, infer_s=ti)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=3)       # Mat(d): D = d (d + 1) / 2
    ap.add_argument("--M", type=int, default=2000)
    ap.add_argument("--atoms", type=int, default=100)
    args = ap.parse_args()
    run(args.d, args.M, args.atoms)
