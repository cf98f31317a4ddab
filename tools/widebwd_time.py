import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_nf_amd as tnf
D, L = 64, 2
for U, N in ((20, 1 << 18), (64, 1 << 18), (15, 1 << 18)):
    layer = tnf.RealNVP(D, L, U)
    rng = np.random.RandomState(0)
    p = torch.tensor(rng.normal(0, 0.1, (1, layer.count_num_params()))).float().cuda().requires_grad_()
    z = torch.randn(1, N, D, device="cuda")
    def fwd():
        with torch.no_grad():
            layer.inverse_and_log_det(z, p)
    def step():
        p.grad = None
        zo, ld = layer.inverse_and_log_det(z, p)
        (zo.sum() + ld.sum()).backward()
    for fn, name in ((fwd, "forward"), (step, "fwd+bwd")):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("RealNVP D=%d U=%d N=%d %-8s %8.3f ms  %8.1f M samples/s" % (D, U, N, name, dt * 1e3, N / dt / 1e6), flush=True)
