#!/usr/bin/env python3
"""Training step of the fused conditional flow at 2^18 contexts, per variant library (TNF_LIB_PATH), in subprocesses."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1].split(",")
M = sys.argv[2] if len(sys.argv) > 2 else str(1 << 18)
code = r'''
import os, sys, time, numpy as np, torch
sys.path.insert(0, %r)
import torch_nf_amd as tnf
torch.manual_seed(0); np.random.seed(0)
M = int(%r)
nf = tnf.NormFlow(64, True, "coupling", 4, 2, 15)
cde = tnf.ConditionalDensityEstimator(nf, 32, [64, 64])
x = torch.randn(M, 32, device="cuda"); z = torch.randn(M, 1, 64, device="cuda")
def train():
    cde.zero_grad()
    (-cde.log_prob(z, x).mean()).backward()
def infer():
    with torch.no_grad(): cde.log_prob(z, x)
out = []
for fn in (infer, train):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(7):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    out.append(np.median(ts) * 1e3)
g = sum(float(p.grad.double().abs().sum()) for p in cde.parameters())
print("RES infer %%.3f ms  train %%.3f ms  gradsum %%.6e" %% (out[0], out[1], g))
''' % (ROOT, M)
for n in names:
    env = dict(os.environ)
    if n != "base":
        env["TNF_LIB_PATH"] = os.path.join(ROOT, "scratch", "abl2", "lib%s.so" % n)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("RES")]
    print("%-8s %s" % (n, line[0] if line else "FAILED " + out.stderr[-400:]), flush=True)
