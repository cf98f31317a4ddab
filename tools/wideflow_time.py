import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_nf_amd as tnf
for arch, D, U, N in (("coupling", 64, 64, 1 << 18), ("coupling", 64, 20, 1 << 18), ("AR", 64, 64, 1 << 18)):
    np.random.seed(0); torch.manual_seed(0)
    nf = tnf.NormFlow(D, False, arch, 4 if arch == "coupling" else 2, 2, U)
    z = torch.randn(1, N, D, device="cuda")
    with torch.no_grad():
        nf(4096)
    lib = tnf._lib.lib
    def infer():
        with torch.no_grad(): nf.log_prob(z)
    def train():
        nf.params.grad = None
        (-nf.log_prob(z).mean()).backward()
    for fn, name in ((infer, "log_prob"), (train, "train")):
        c0 = [lib.tnf_diag_launch_count(i) for i in range(7)]
        fn(); torch.cuda.synchronize()
        c1 = [lib.tnf_diag_launch_count(i) for i in range(7)]
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("%s D=%d U=%d N=%d %-8s %8.3f ms %8.1f M samples/s  launches %s" % (arch, D, U, N, name, dt * 1e3, N / dt / 1e6, [b - a for a, b in zip(c0, c1)]), flush=True)
