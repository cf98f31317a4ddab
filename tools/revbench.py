#!/usr/bin/env python3
"""A/B of the whole-flow training backward kernels (BASELINE cfg 4: D=64, 8 coupling layers, 2^19 samples per GPU):
flow_bwd_f16_kernel (TNF_OPT_REV_VARIANT 0, default) against the magic-number form of flow_bwd_pair.h (1), gradients of
-mean(log_prob) against each other and against the per-layer pair with fp32-MFMA layer kernels.  Backward time = HIP
events around loss.backward() with overflow recovery off (the kernel itself + its reduction + the fold backward), after
`settle` untimed steps.  Usage: python tools/revbench.py [N] [steps] [settle] [D]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_nf_amd as tnf  # noqa: E402
from torch_nf_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
settle = int(sys.argv[3]) if len(sys.argv) > 3 else 100
D = int(sys.argv[4]) if len(sys.argv) > 4 else 64
S, L, U = 4, 2, 15
rng = np.random.RandomState(0)
nf = tnf.NormFlow(D, False, "coupling", S, L, U)
p0 = torch.tensor(rng.normal(0.0, 0.1, (1, nf.D_params))).float().cuda()
mean = rng.normal(0.0, 0.3, (2 * S, D)).astype(np.float32)
alpha = np.exp(rng.normal(0.0, 0.2, (2 * S, D))).astype(np.float32)
for b, m, a in zip(nf._bn_layers(), mean, alpha):
    b.set_last_stats(torch.from_numpy(m).cuda(), torch.from_numpy(a).cuda())
z = torch.randn(1, N, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
tnf.ops._FlowLogProbRevFn.overflow_recovery = "off"
grads = {}


def run(name, variant, reversible, fp32=0):
    _lib.check(_lib.lib.tnf_set_option(_lib.OPT_REV_VARIANT, variant))
    _lib.check(_lib.lib.tnf_set_option(_lib.OPT_TRAIN_BWD_FP32, fp32))
    nf.reversible_training = reversible
    nf.params = p0.clone().requires_grad_()
    for _ in range(settle):
        nf.params.grad = None
        (-nf.log_prob(z).mean()).backward()
    ev = []
    for _ in range(steps):
        nf.params.grad = None
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        loss = -nf.log_prob(z).mean()
        e[1].record()
        loss.backward()
        e[2].record()
        ev.append(e)
    torch.cuda.synchronize()
    tf = np.median([e[0].elapsed_time(e[1]) for e in ev])
    tb = np.median([e[1].elapsed_time(e[2]) for e in ev])
    grads[name] = nf.params.grad.clone()
    print("%-22s N=%d D=%d: forward %.3f ms  backward %.3f ms" % (name, N, D, tf, tb), flush=True)
    _lib.lib.tnf_set_option(_lib.OPT_REV_VARIANT, 0)
    _lib.lib.tnf_set_option(_lib.OPT_TRAIN_BWD_FP32, 0)


run("magic form (1)", 1, True)
run("default kernel (0)", 0, True)
run("magic form again", 1, True)
if N <= 1 << 19:
    run("per-layer fp32", 0, False, 1)
ref = grads.get("per-layer fp32", grads["default kernel (0)"])
top = ref.abs().max().item()
for k, g in grads.items():
    print("%-22s max |g - ref| / max |ref| = %.3e   finite %s" % (k, (g - ref).abs().max().item() / top, bool(torch.isfinite(g).all())))
print("magic form reproducible: %s" % bool(torch.equal(grads["magic form (1)"], grads["magic form again"])))
