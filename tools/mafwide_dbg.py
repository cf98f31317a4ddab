import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import torch_nf_amd as tnf
import flow_oracle as oracle
D, L, U, M, N = 64, 3, 32, 2, 50
rng = np.random.RandomState(D + U + N)
np.random.seed(D + L)
layer = tnf.MAF(D, L, U)
Ms = [Mk[0].numpy() for Mk in layer.Ms]
p0 = torch.tensor(rng.normal(0, 0.05, (1, layer.count_num_params()))).float()
z0 = torch.tensor(rng.normal(0, 1, (M, N, D))).float()
wz = torch.tensor(rng.normal(0, 1, (M, N, D))).float()
wl = torch.tensor(rng.normal(0, 1, (M, N))).float()
res = []
for generic in (0, 1):
    tnf._lib.lib.tnf_set_option(tnf._lib.OPT_FORCE_GENERIC, generic)
    p, z = p0.cuda().requires_grad_(), z0.cuda().requires_grad_()
    zo, ld = layer.inverse_and_log_det(z, p)
    ((zo * wz.cuda()).sum() + (ld * wl.cuda()).sum()).backward()
    res.append(p.grad.cpu().double()[0])
tnf._lib.lib.tnf_set_option(tnf._lib.OPT_FORCE_GENERIC, 0)
pr, zr = p0.double().clone().requires_grad_(), z0.double().clone().requires_grad_()
zo, ld = oracle.maf(zr, pr, D, L, U, Ms, True)
((zo * wz.double()).sum() + (ld * wl.double()).sum()).backward()
ref = pr.grad[0]
print("ref dtype", ref.dtype, "max", float(ref.abs().max()))
off = 0
dims = [(D, U)] + [(U, U)] * (L - 1) + [(U, D)]
for l, (a, b) in enumerate(dims):
    for net in range(2):
        sl = slice(off, off + a * b); off += a * b
        for name, r in (("mfma", res[0]), ("generic", res[1])):
            e = (r[sl] - ref[sl]).abs()
            print("layer %d net %d %-8s max|ref| %.3e  max err %.3e  (rel to block max %.2e) argmax %d" % (
                l, net, name, float(ref[sl].abs().max()), float(e.max()), float(e.max() / ref[sl].abs().max()), int(e.argmax())))
