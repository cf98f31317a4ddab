#!/usr/bin/env python3
"""BASELINE configs[2]: ConditionalDensityEstimator(NormFlow(64, True, 'coupling', 4, 2, 15), D_x=32, [64, 64]).

Part 1: cde(x, N, freeze_bn=True) then cde.log_prob(z, x) for (M, N) with M*N = 2^20 (SURVEY 8d cfg 3),
        params materialised by param_net.
Part 2: the SNPE layout, one sample per context (N = 1): fused conditioner + flow kernel vs the
        materialised path, inference (no_grad) and one training step (forward + backward through
        param_net).  --max-materialised limits the context count of the materialised runs (82 KB of
        params per context, twice that under autograd).
"""
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
ap.add_argument("--max-materialised", type=int, default=1 << 16)
ap.add_argument("--skip-shapes", action="store_true")
args = ap.parse_args()

D, S, L, U, D_x = 64, 4, 2, 15, 32
torch.manual_seed(0)
np.random.seed(0)
nf = tnf.NormFlow(D, True, "coupling", S, L, U)
cde = tnf.ConditionalDensityEstimator(nf, D_x, [64, 64])
with torch.no_grad():
    for p in cde.param_net.parameters():
        p.mul_(0.3)


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


if not args.skip_shapes:
    cde.fuse_conditioner = False
    with torch.no_grad():
        for (M, N) in [(16, 1 << 16), (1, 1 << 20), (2048, 512), (1 << 14, 64), (1 << 17, 8), (1 << 20, 1)]:
            x = torch.randn(M, D_x, device="cuda")
            params = cde.param_net(x)
            omega = torch.randn(M, N, D, device="cuda")
            z, _ = nf._forward_from(omega, params, freeze_bn=True)
            res = [timeit(lambda: nf._forward_from(omega, params, freeze_bn=True), 10),
                   timeit(lambda: nf.log_prob(z, params), 10), timeit(lambda: cde.param_net(x), 10)]
            print("M=%7d N=%7d: forward %8.3f ms  log_prob %8.3f ms  (param_net %6.3f ms)  -> log_prob %7.1f M samples/s"
                  % (M, N, res[0] * 1e3, res[1] * 1e3, res[2] * 1e3, M * N / res[1] / 1e6))
            del params, omega, z

print("one sample per context (N = 1): cde.log_prob(z[:, None, :], x)")
for M in (1 << 12, 1 << 16, 1 << 18, 1 << 20):
    x = torch.randn(M, D_x, device="cuda")
    z = torch.randn(M, 1, D, device="cuda")

    def infer():
        with torch.no_grad():
            cde.log_prob(z, x)

    def train():
        cde.zero_grad()
        (-cde.log_prob(z, x).mean()).backward()

    row = []
    for fused in (True, False):
        if not fused and M > args.max_materialised:
            row += [float("nan")] * 2
            continue
        cde.fuse_conditioner = fused
        row += [timeit(infer, 5), timeit(train, 3)]
    print("M=%8d  fused: log_prob %8.3f ms (%7.1f M ctx/s)  train step %8.3f ms (%6.2f M ctx/s) | materialised: "
          "log_prob %8.3f ms  train step %8.3f ms" % (M, row[0] * 1e3, M / row[0] / 1e6, row[1] * 1e3, M / row[1] / 1e6,
                                                      row[2] * 1e3, row[3] * 1e3))

print("one sample per context (N = 1), SAMPLING: cde.sample(x, N=1) (device draw, frozen statistics)")
for M in (1 << 12, 1 << 16, 1 << 18, 1 << 20):
    x = torch.randn(M, D_x, device="cuda")

    def draw():
        with torch.no_grad():
            cde.sample(x, N=1)

    row = []
    for fused in (True, False):
        if not fused and M > args.max_materialised:
            row.append(float("nan"))
            continue
        cde.fuse_conditioner = fused
        row.append(timeit(draw, 5))
    cde.fuse_conditioner = True
    print("M=%8d  fused: %8.3f ms (%7.1f M samples/s) | materialised: %8.3f ms" % (M, row[0] * 1e3, M / row[0] / 1e6, row[1] * 1e3))
