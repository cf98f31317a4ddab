#!/usr/bin/env python3
"""The reference's scripts/lfi_mat.py (BASELINE configs[4]) on this package: APT on the matrix det/trace
simulator with an autoregressive flow + ToInterval support layer, conditioned through param_net [64, 64].
`torch_nf.systems` / `torch_nf.lfi` are not in the reference snapshot; torch_nf_amd.systems / .lfi are
from-scratch stand-ins (parity unpinned).  Usage: python examples/lfi_mat.py --d 2 --num-iters 500"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_nf_amd as tnf  # noqa: E402
from torch_nf_amd.lfi import train_APT  # noqa: E402
from torch_nf_amd.systems import Mat  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--d", type=int, default=2)
ap.add_argument("--rs", type=int, default=1)
ap.add_argument("--M", type=int, default=2000)
ap.add_argument("--M-atom", type=int, default=100)
ap.add_argument("--num-iters", type=int, default=1000)
ap.add_argument("--R", type=int, default=4)
args = ap.parse_args()

np.random.seed(args.rs)
torch.manual_seed(args.rs)
mat = Mat(args.d, noise=0.05)
x0 = np.array([[0.0, args.d / 2]])  # det = 0, trace = d / 2  (scripts/lfi_mat.py:32)
support_layer = tnf.ToInterval(mat.D, mat.lb, mat.ub)
nf = tnf.NormFlow(mat.D, True, "AR", num_stages=1, num_layers=2, num_units=2 * mat.D, support_layer=support_layer)
print("# params ", nf.D_params)
cde = tnf.ConditionalDensityEstimator(nf, x0.shape[1], [64, 64], dropout=False)
cde, losses, zs, log_probs, it_time = train_APT(cde, mat, x0, M=args.M, M_atom=args.M_atom, R=args.R,
                                               num_iters=args.num_iters, verbose=True)
t0 = time.time()
z, lq = cde(torch.tensor(x0).float(), args.M)
time_per_sample = (time.time() - t0) / args.M
T_x = mat.simulate(zs[-1])
print("%.2f ms per training iteration (%d contexts x %d atoms), %.2e s per posterior sample"
      % (it_time * 1e3, args.M, args.M_atom, time_per_sample))
print("posterior predictive statistics: mean", T_x.mean(0), " std", T_x.std(0), " target", x0[0])
