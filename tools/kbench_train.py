#!/usr/bin/env python3
"""Training-step kernels on the GPU box (BASELINE cfg 4: D=64, 8 coupling layers, 2^19 samples per GPU):
loss = -mean(NormFlow.log_prob(z)); backward.  Times the reversible pair (whole-flow forward, one-kernel
backward) and the per-layer pair with HIP events, after `settle` untimed steps back to back (an idle GPU needs ~50 ms
of load to reach its clock).  Usage: python tools/kbench_train.py [N] [steps] [settle]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_nf_amd as tnf  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
settle = int(sys.argv[3]) if len(sys.argv) > 3 else 100
D, S, L, U = 64, 4, 2, 15
rng = np.random.RandomState(0)
nf = tnf.NormFlow(D, False, "coupling", S, L, U)
nf.params = torch.tensor(rng.normal(0.0, 0.1, (1, nf.D_params))).float().cuda().requires_grad_()
mean = rng.normal(0.0, 0.3, (2 * S, D)).astype(np.float32)
alpha = np.exp(rng.normal(0.0, 0.2, (2 * S, D))).astype(np.float32)
for b, m, a in zip(nf._bn_layers(), mean, alpha):
    b.set_last_stats(torch.from_numpy(m).cuda(), torch.from_numpy(a).cuda())
z = torch.randn(1, N, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
grads = {}
for name, rev in (("reversible", True), ("layers", False)):
    nf.reversible_training = rev
    tf, tb = [], []
    for _ in range(settle):  # no synchronisation in here: the GPU stays busy until the timed steps start
        nf.params.grad = None
        (-nf.log_prob(z).mean()).backward()
    for i in range(steps + 2):
        nf.params.grad = None
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        loss = -nf.log_prob(z).mean()
        e[1].record()
        loss.backward()
        e[2].record()
        if i >= 2:
            tf.append(e)
    torch.cuda.synchronize()
    tf, tb = [e[0].elapsed_time(e[1]) for e in tf], [e[1].elapsed_time(e[2]) for e in tf]
    grads[name] = nf.params.grad.clone()
    print("%-10s N=%d: forward %.3f ms  backward %.3f ms  -> %.1f M samples/s (fwd+bwd)"
          % (name, N, np.mean(tf), np.mean(tb), N / (np.mean(tf) + np.mean(tb)) / 1e3))
d = (grads["reversible"] - grads["layers"]).abs().max().item() / grads["layers"].abs().max().item()
print("max |g_rev - g_layers| / max |g| = %.2e" % d)
