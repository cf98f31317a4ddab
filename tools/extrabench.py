#!/usr/bin/env python3
"""Timing of the widened rows: ToInterval / ToSimplex support layers, MAF, NormFlow('AR'),
NormFlow('coupling', support_layer=ToInterval).  Prints one line per case with the achieved
algorithmic bandwidth (bytes per row as in DESIGN.md) next to the time."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_nf_amd as tnf  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def line(name, rows, t, bytes_per_row):
    print("%-44s rows %8d  %8.3f ms  %8.1f M rows/s  %7.1f GB/s alg" %
          (name, rows, t * 1e3, rows / t / 1e6, rows * bytes_per_row / t / 1e9))


torch.manual_seed(0)
np.random.seed(0)
with torch.no_grad():
    for D in (8, 64):
        N = 1 << 20
        lb = np.where(np.arange(D) % 3 == 0, -np.inf, -2.0)
        ub = np.where(np.arange(D) % 3 == 1, np.inf, 2.5)
        iv = tnf.ToInterval(D, lb, ub)
        z = torch.randn(1, N, D, device="cuda")
        y, _ = iv(z)
        line("ToInterval forward D=%d" % D, N, timeit(lambda: iv(z)), 8 * D + 4)
        line("ToInterval inverse D=%d" % D, N, timeit(lambda: iv.inverse_and_log_det(y)), 8 * D + 4)
        sx = tnf.ToSimplex(D)
        line("ToSimplex forward D=%d" % D, N, timeit(lambda: sx(z)), 8 * D + 8)
    from torch_nf_amd import _lib
    for (D, L, U, N) in [(4, 2, 20, 1 << 20), (16, 2, 32, 1 << 20), (64, 2, 64, 1 << 18)]:
        np.random.seed(0)
        maf = tnf.MAF(D, L, U)
        p = torch.randn(1, maf.count_num_params(), device="cuda") * 0.2
        z = torch.randn(1, N, D, device="cuda")
        for generic in (0, 1):  # matrix-pipe kernel, then the shape-generic one
            _lib.lib.tnf_set_option(_lib.OPT_FORCE_GENERIC, generic)
            tag = "generic" if generic else "mfma"
            line("MAF inverse D=%d L=%d U=%d [%s]" % (D, L, U, tag), N, timeit(lambda: maf.inverse_and_log_det(z, p), 5), 8 * D + 4)
            line("MAF forward (D-1 passes) D=%d [%s]" % (D, tag), N, timeit(lambda: maf(z, p), 3), 8 * D + 4)
        _lib.lib.tnf_set_option(_lib.OPT_FORCE_GENERIC, 0)
        nf = tnf.NormFlow(D, False, "AR", 1, L, U)
        line("NormFlow('AR').log_prob D=%d (one kernel)" % D, N, timeit(lambda: nf.log_prob(z), 5), 4 * D + 4)
        line("NormFlow('AR').sample D=%d (one kernel)" % D, N, timeit(lambda: nf.sample(N), 3), 4 * D + 12)
    D, N = 64, 1 << 20
    lb = np.where(np.arange(D) % 3 == 0, -np.inf, -6.0)
    ub = np.where(np.arange(D) % 3 == 1, np.inf, 6.0)
    for sup in (None, tnf.ToInterval(D, lb, ub)):
        nf = tnf.NormFlow(D, False, "coupling", 4, 2, 15, sup)
        z, _ = nf.sample(N)
        line("coupling S=4 log_prob, support=%s" % (sup.name if sup else None), N, timeit(lambda: nf.log_prob(z)), 4 * D + 4)
        line("coupling S=4 sample(frozen), support=%s" % (sup.name if sup else None), N, timeit(lambda: nf.sample(N)), 4 * D + 12)
