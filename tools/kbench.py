#!/usr/bin/env python3
"""Quick kernel timing on the GPU box: whole-flow kernel and per-layer chain (HIP events),
plus the largest difference between the two paths.  Usage: python tools/kbench.py [D] [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_nf_amd as tnf  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300  # >= 200: an idle GPU needs ~50 ms of load to reach its clock
S, L, U, N = 4, 2, 15, 1 << 20
rng = np.random.RandomState(0)
nf = tnf.NormFlow(D, False, "coupling", S, L, U)
params = torch.tensor(rng.normal(0.0, 0.1, (1, nf.D_params))).float()
nf.params = params.cuda()
mean = rng.normal(0.0, 0.3, (2 * S, D)).astype(np.float32)
alpha = np.exp(rng.normal(0.0, 0.2, (2 * S, D))).astype(np.float32)
for b, m, a in zip(nf._bn_layers(), mean, alpha):
    b.set_last_stats(torch.from_numpy(m).cuda(), torch.from_numpy(a).cuda())
fv = int(os.environ.get("TNF_FLOW_VARIANT", "-1"))
lv = int(os.environ.get("TNF_LAYER_VARIANT", "-1"))
if fv >= 0:
    tnf._lib.check(tnf._lib.lib.tnf_set_option(tnf._lib.OPT_FLOW_VARIANT, fv))
if lv >= 0:
    tnf._lib.check(tnf._lib.lib.tnf_set_option(tnf._lib.OPT_LAYER_VARIANT, lv))
z = torch.randn(1, N, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
want = None
for name, fusion in (("flow ", tnf._lib.FUSE_FLOW), ("layer", tnf._lib.FUSE_LAYER)):
    nf.fusion = fusion
    with torch.no_grad():
        for _ in range(3 if steps < 100 else 250):  # short runs: cold clocks on purpose (PMC passes); long runs: settled
            lp = nf.log_prob(z)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in ev:
            a.record()
            lp = nf.log_prob(z)
            b.record()
        torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    want = lp.clone() if want is None else want
    rel = ((lp - want).abs() / want.abs().clamp_min(1e-3)).max().item()
    print("fv=%d lv=%d " % (fv, lv), end="")
    print("D=%d %s: mean %.4f ms  min %.4f ms  -> %.0f M samples/s (mean)   max rel diff vs flow path %.2e"
          % (D, name, ms.mean(), ms.min(), N / ms.mean() / 1e3, rel))
