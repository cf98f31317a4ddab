import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_nf_amd as tnf
for D in (32, 64):
    for lg in (18, 19, 20, 21, 22):
        N = 1 << lg
        S, L, U = 4, 2, 15
        rng = np.random.RandomState(0)
        nf = tnf.NormFlow(D, False, "coupling", S, L, U)
        nf.params = torch.tensor(rng.normal(0, 0.1, (1, nf.D_params))).float().cuda()
        mean = rng.normal(0, 0.3, (2 * S, D)).astype(np.float32); alpha = np.exp(rng.normal(0, 0.2, (2 * S, D))).astype(np.float32)
        for b, m, a in zip(nf._bn_layers(), mean, alpha): b.set_last_stats(torch.from_numpy(m).cuda(), torch.from_numpy(a).cuda())
        z = torch.randn(1, N, D, device="cuda")
        nf.fusion = tnf._lib.FUSE_LAYER
        with torch.no_grad():
            for _ in range(200): lp = nf.log_prob(z)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
            for a, b in ev:
                a.record(); lp = nf.log_prob(z); b.record()
            torch.cuda.synchronize()
        ms = np.median([a.elapsed_time(b) for a, b in ev])
        byt = N * (4 * D * 15 + 8 * 7 + 4)
        print("D=%d N=2^%d chain %.4f ms  %.1f us/launch  algorithmic %.2f TB/s (frac %.3f)" % (D, lg, ms, ms * 1e3 / 8, byt / ms / 1e9, byt / ms / 1e9 / 8), flush=True)
        del z, nf
