#!/usr/bin/env python3
"""Sampling with fresh batch statistics (NormFlow.forward with freeze_bn=False: the reference's default sampling
call and, under autograd, its train_efn loop, notebooks/two_network_arch.ipynb:84-92) at D=64, 8 coupling layers:
the one-call chains (tnf_flow_forward_batch_f32, tnf_flow_forward_train_fwd/bwd_f32) against the per-bijector
composition, and an EFN-style step with Adam eagerly and as one HIP graph.  Usage: python tools/efnbench.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_nf_amd as tnf  # noqa: E402


def bench(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


D, S, L, U = 64, 4, 2, 15
nf = tnf.NormFlow(D, False, "coupling", S, L, U)
nf.params = (torch.randn(1, nf.D_params) * 0.1).cuda().requires_grad_()
for N in (1 << 16, 1 << 19):
    omega = torch.randn(1, N, D, device="cuda")

    def sample():
        with torch.no_grad():
            nf._forward_from(omega, nf.params, freeze_bn=False)

    def train():
        nf.params.grad = None
        z, lq = nf._forward_from(omega, nf.params, freeze_bn=False)
        (lq.mean() + (z ** 2).mean()).backward()

    def frozen():
        with torch.no_grad():
            nf._forward_from(omega, nf.params, freeze_bn=True)

    row = []
    for fused in (True, False):
        nf.fused_batch_forward = fused
        row.append((bench(sample), bench(train)))
    nf.fused_batch_forward = True
    print("N=%7d  fresh statistics, no autograd: %.3f ms (per bijector %.3f)   with backward: %.3f ms (per bijector %.3f)"
          "   frozen statistics: %.3f ms" % (N, row[0][0], row[1][0], row[0][1], row[1][1], bench(frozen)))

opt = torch.optim.Adam([nf.params], lr=1e-4, capturable=True)
for N in (1 << 14, 1 << 16):
    def step():
        opt.zero_grad(set_to_none=True)
        z, lq = nf.sample(N, freeze_bn=False)
        loss = lq.mean() + (z ** 2).mean()
        loss.backward()
        opt.step()
        return loss.detach()

    te = bench(step)
    gs = tnf.graphs.GraphedStep(step, warmup=3)
    print("N=%7d  EFN-style step (device draw, fresh statistics, backward, Adam): eager %.3f ms, one HIP graph %.3f ms"
          % (N, te, bench(gs)))
