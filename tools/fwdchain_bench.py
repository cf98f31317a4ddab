#!/usr/bin/env python3
"""The SAMPLING direction with frozen statistics (NormFlow.forward / sample, density_estimator.py:374-388) at D = 64 / 32,
N = 2^20: whole-flow kernel against the per-layer chains -- TNF_OPT_LAYER_VARIANT 10 (default: flow_range2_kernel<.., FWD>,
one launch per coupling layer, half-row stores) and 0 (the round-1 fp32-MFMA coupling_mfma_kernel chain, full rows).
Roofline of a chain: SURVEY 8(d)'s B_alg(k = 2S) bytes per sample over the time of the 2S launches, against 8 TB/s.
Usage: python tools/fwdchain_bench.py [D] [steps] [settle]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_nf_amd as tnf  # noqa: E402
from torch_nf_amd import _lib, ops  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
settle = int(sys.argv[3]) if len(sys.argv) > 3 else 300
S, L, U, N = 4, 2, 15, 1 << 20
rng = np.random.RandomState(0)
nf = tnf.NormFlow(D, False, "coupling", S, L, U)
params = torch.tensor(rng.normal(0.0, 0.1, (1, nf.D_params))).float().cuda()
mean = torch.tensor(rng.normal(0.0, 0.3, (2 * S, D))).float().cuda()
alpha = torch.tensor(np.exp(rng.normal(0.0, 0.2, (2 * S, D)))).float().cuda()
omega = torch.randn(1, N, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
k = 2 * S
b_alg = 4 * D * (2 * k - 1) + 8 * (k - 1) + 4
out = {}
for name, fusion, variant in (("whole-flow kernel", _lib.FUSE_FLOW, 10), ("layer chain (default)", _lib.FUSE_LAYER, 10),
                              ("layer chain (fp32 MFMA)", _lib.FUSE_LAYER, 0)):
    _lib.check(_lib.lib.tnf_set_option(_lib.OPT_LAYER_VARIANT, variant))
    with torch.no_grad():
        for _ in range(settle):
            z, sld = ops.flow_forward_raw(omega, params, mean, alpha, D, S, L, U, fusion)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in ev:
            a.record()
            z, sld = ops.flow_forward_raw(omega, params, mean, alpha, D, S, L, U, fusion)
            b.record()
        torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    out[name] = (z.clone(), sld.clone())
    roof = "" if fusion == _lib.FUSE_FLOW else "  %.2f TB/s of the %d B/sample = %.3f of 8 TB/s" % (
        N * b_alg / ms / 1e9, b_alg, N * b_alg / ms / 1e9 / 8.0)
    print("D=%d %-24s %.4f ms  %.0f M samples/s%s" % (D, name, ms, N / ms / 1e3, roof), flush=True)
_lib.lib.tnf_set_option(_lib.OPT_LAYER_VARIANT, 10)
zr, sr = out["whole-flow kernel"]
for name, (z, sld) in out.items():
    print("%-24s max |z - z_flow| %.2e   max |sld - sld_flow| / max |sld| %.2e" % (
        name, float((z - zr).abs().max()), float((sld - sr).abs().max() / sr.abs().max())))
