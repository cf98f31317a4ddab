"""Random-shape sweep: each new fused training path against the composition it replaces."""
import sys, itertools, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch_nf_amd as tnf
rng = np.random.RandomState(123)
bad = 0
def relerr(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
# 1. reversible log_prob training vs per-layer pair
for it in range(14):
    D = int(rng.choice([32, 64])); S = int(rng.randint(1, 5)); L = int(rng.randint(1, 4)); U = int(rng.choice([15, 16, 15, 8, 12]))
    U = max(U, 15)
    M = int(rng.choice([1, 1, 2, 3])); Mp = int(rng.choice([1, M])); N = int(rng.choice([1, 7, 32, 33, 100, 1000, 4097]))
    if Mp > 1 and N < 32: N = 64
    nf = tnf.NormFlow(D, True, "coupling", S, L, U)
    if not tnf.ops.flow_train_rev_supported(M, Mp, N, D, S, L, U) or not tnf.ops.flow_train_supported(M, Mp, N, D, S, L, U):
        continue
    for b in nf._bn_layers():
        b.set_last_stats(torch.tensor(rng.normal(0, 0.3, D)).float(), torch.tensor(np.exp(rng.normal(0, 0.2, D))).float())
    p0 = torch.tensor(rng.normal(0, 0.1, (Mp, nf.D_params))).float().cuda()
    z0 = torch.tensor(rng.normal(0, 1, (M, N, D))).float().cuda()
    w = torch.tensor(rng.uniform(0.1, 1, (M, N))).float().cuda()
    out = {}
    for rev in (True, False):
        nf.reversible_training = rev
        p = p0.clone().requires_grad_(); z = z0.clone().requires_grad_()
        (nf.log_prob(z, p) * w).sum().backward()
        out[rev] = (p.grad.clone(), z.grad.clone())
    e = (relerr(out[True][0], out[False][0]), relerr(out[True][1], out[False][1]))
    ok = e[0] < 2e-4 and e[1] < 2e-4
    bad += not ok
    print("rev  D=%d S=%d L=%d U=%d M=%d Mp=%d N=%d  gp %.1e gz %.1e %s" % (D, S, L, U, M, Mp, N, e[0], e[1], "" if ok else "<<<<"))
# 2. AR fused training vs per-bijector
for it in range(14):
    D = int(rng.choice([2, 3, 5, 6, 8, 13, 16, 21, 32])); L = int(rng.randint(1, 4)); U = int(rng.choice([15, 20, 32, 42, 64]))
    M = int(rng.choice([1, 2, 5, 40])); Mp = int(rng.choice([1, M])); N = int(rng.choice([1, 5, 16, 17, 100, 257]))
    sup = bool(rng.randint(0, 2))
    lb, ub = -2.0 * np.ones(D), 2.0 * np.ones(D); lb[::2] = -np.inf
    nf = tnf.NormFlow(D, True, "AR", 1, L, U, tnf.ToInterval(D, lb, ub) if sup else None)
    if not tnf.ops.ar_flow_train_supported(M, Mp, D, nf.num_layers, nf.num_units) or not tnf.ops.ar_flow_supported(D, nf.num_layers, nf.num_units):
        continue
    nf.bijectors[1].set_last_stats(torch.tensor(rng.normal(0, 0.3, D)).float(), torch.tensor(np.exp(rng.normal(0, 0.2, D))).float())
    p0 = torch.tensor(rng.normal(0, 0.2, (Mp, nf.D_params))).float().cuda()
    z = torch.tensor(rng.uniform(-1.5, 1.5, (M, N, D))).float().cuda()
    w = torch.tensor(rng.uniform(0.1, 1, (M, N))).float().cuda()
    out = {}
    for fused in (True, False):
        nf.fused_ar_training = fused
        p = p0.clone().requires_grad_()
        (nf.log_prob(z, p) * w).sum().backward()
        out[fused] = p.grad.clone()
    e = relerr(out[True], out[False])
    ok = e < 3e-4
    bad += not ok
    print("ar   D=%d L=%d U=%d M=%d Mp=%d N=%d sup=%d  gp %.1e %s" % (D, nf.num_layers, nf.num_units, M, Mp, N, sup, e, "" if ok else "<<<<"))
# 3. batch-statistics forward chain (no grad and with grad) vs per-bijector
for it in range(12):
    D = int(rng.choice([32, 64])); S = int(rng.randint(1, 5)); L = int(rng.randint(1, 4)); U = 15
    M = int(rng.choice([1, 2, 4])); N = int(rng.choice([8, 33, 100, 1000, 5000]))
    nf = tnf.NormFlow(D, True, "coupling", S, L, U)
    p0 = torch.tensor(rng.normal(0, 0.1, (M, nf.D_params))).float().cuda()
    om = torch.tensor(rng.normal(0, 1, (M, N, D))).float().cuda()
    w = torch.tensor(rng.uniform(0.5, 1.5, (M, N))).float().cuda()
    out = {}
    for fused in (True, False):
        nf.fused_batch_forward = fused
        with torch.no_grad():
            z, lq = nf._forward_from(om, p0, freeze_bn=False)
        p = p0.clone().requires_grad_()
        z2, lq2 = nf._forward_from(om, p, freeze_bn=False)
        ((lq2 * w).mean() + (z2 ** 2).mean()).backward()
        out[fused] = (z, lq, z2.detach(), p.grad.clone())
    e = (relerr(out[True][0], out[False][0]), relerr(out[True][1].float(), out[False][1].float()), relerr(out[True][2], out[False][2]), relerr(out[True][3], out[False][3]))
    ok = e[0] < 1e-4 and e[1] < 1e-5 and e[2] < 1e-4 and e[3] < 2e-3
    bad += not ok
    print("bfwd D=%d S=%d L=%d M=%d N=%d  z %.1e lq %.1e z(grad mode) %.1e gp %.1e %s" % (D, S, L, M, N, e[0], e[1], e[2], e[3], "" if ok else "<<<<"))
print("BAD:", bad)
