#!/usr/bin/env python3
"""Interleaved A/B of whole-flow kernel builds: each variant library in its own subprocess round-robin, R rounds."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1].split(",")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
D = sys.argv[3] if len(sys.argv) > 3 else "64"
res = {n: [] for n in names}
code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import torch_nf_amd as tnf
D=int(%r); S,L,U,N=4,2,15,1<<20
rng=np.random.RandomState(0)
nf=tnf.NormFlow(D,False,"coupling",S,L,U)
nf.params=torch.tensor(rng.normal(0,0.1,(1,nf.D_params))).float().cuda()
mean=rng.normal(0,0.3,(2*S,D)).astype(np.float32); alpha=np.exp(rng.normal(0,0.2,(2*S,D))).astype(np.float32)
for b,m,a in zip(nf._bn_layers(),mean,alpha): b.set_last_stats(torch.from_numpy(m).cuda(),torch.from_numpy(a).cuda())
z=torch.randn(1,N,D,device="cuda",generator=torch.Generator(device="cuda").manual_seed(1))
nf.fusion=int(os.environ.get("TNF_FUSION", tnf._lib.FUSE_FLOW))
lv=int(os.environ.get("TNF_LAYER_VARIANT","-1"))
if lv>=0: tnf._lib.check(tnf._lib.lib.tnf_set_option(tnf._lib.OPT_LAYER_VARIANT, lv))
fv=int(os.environ.get("TNF_FLOW_VARIANT","-1"))
if fv>=0: tnf._lib.check(tnf._lib.lib.tnf_set_option(tnf._lib.OPT_FLOW_VARIANT, fv))
with torch.no_grad():
    for _ in range(400): lp=nf.log_prob(z)
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a,b in ev:
        a.record(); lp=nf.log_prob(z); b.record()
    torch.cuda.synchronize()
ms=np.array([a.elapsed_time(b) for a,b in ev])
print("RES %%.5f %%.5f %%.6e" %% (np.median(ms), ms.min(), float(lp.double().sum())))
''' % (ROOT, D)
for r in range(rounds):
    for n in names:
        env = dict(os.environ)
        if n == "base":
            pass
        elif n.startswith("fv"):
            env["TNF_FLOW_VARIANT"] = n[2:]
        else:
            env["TNF_LIB_PATH"] = os.path.join(ROOT, "scratch", "abl2", "lib%s.so" % n)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RES")]
        if not line:
            print(n, "FAILED", out.stderr[-500:]); continue
        med, mn, chk = line[0].split()[1:]
        res[n].append((float(med), float(mn), chk))
for n in names:
    if res[n]:
        print("%-4s median-of-medians %.4f ms  best-min %.4f ms  (%s)  checksum %s" % (n, sorted(m for m,_,_ in res[n])[len(res[n])//2], min(m for _,m,_ in res[n]), " ".join("%.4f" % m for m,_,_ in res[n]), res[n][0][2]))
