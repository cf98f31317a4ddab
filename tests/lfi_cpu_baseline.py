#!/usr/bin/env python3
"""CPU leg of the LFI inner-step comparison (not a pytest module): runs tools/lfibench.py's GPU
timing, then the same loss + backward through the CPU oracle (= the reference's arithmetic under
torch autograd, scripts/lfi_mat.py:23-57) on the host cores.  Lives under tests/ because it calls
the oracle.  Usage: python tests/lfi_cpu_baseline.py [--d 3] [--M 2000] [--atoms 100]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import flow_oracle as orc  # noqa: E402
import lfibench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--d", type=int, default=3)
ap.add_argument("--M", type=int, default=2000)
ap.add_argument("--atoms", type=int, default=100)
args = ap.parse_args()

r = lfibench.run(args.d, args.M, args.atoms)
nf, cde = r["nf"], r["cde"]
net = torch.nn.Sequential(*[m for m in cde.param_net]).cpu()
Ms = [Mk[0].numpy() for Mk in nf.bijectors[0].Ms]
consts = orc.interval_consts(r["lb"], r["ub"])
stat = (nf.bijectors[1].get_last_mean().cpu().float(), nf.bijectors[1].get_last_alpha().cpu().float())
xc, zc = r["x"].cpu(), r["z"].cpu()


def cpu_step():
    for p in net.parameters():
        p.grad = None
    zi, ld = orc.to_interval(zc, consts, True)
    lp = orc.ar_flow_log_prob(zi, net(xc), r["D"], nf.num_layers, nf.num_units, Ms, stat) - ld
    (-lp.mean()).backward()


cpu_step()
t0 = time.perf_counter()
for _ in range(3):
    cpu_step()
tc = (time.perf_counter() - t0) / 3
print("CPU oracle (%d threads): train step %.1f ms (%.2f M samples/s)  -> GPU %.0fx" %
      (torch.get_num_threads(), tc * 1e3, r["M"] * r["N"] / tc / 1e6, tc / r["train_s"]))
