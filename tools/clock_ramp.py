#!/usr/bin/env python3
"""How long an idle MI355X takes to reach the clock it holds under load: 3,000 NormFlow.log_prob calls (D = 64,
8 coupling layers, N = 2^20) back to back after 0.5 s of idleness, HIP events around each launch; then a 0.2 s idle gap
and 60 more.  profiles/r02_clock_ramp.txt; DESIGN.md 3.10.0."""
import os, sys, numpy as np, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_nf_amd as tnf
D,S,L,U,N=64,4,2,15,1<<20
torch.manual_seed(0); np.random.seed(0)
nf=tnf.NormFlow(D,False,"coupling",S,L,U)
with torch.no_grad(): nf(4096)
z=torch.randn(1,N,D,device="cuda")
torch.cuda.synchronize(); time.sleep(0.5)
K=3000
ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(K)]
with torch.no_grad():
    for a,b in ev:
        a.record(); lp=nf.log_prob(z); b.record()
torch.cuda.synchronize()
ms=np.array([a.elapsed_time(b) for a,b in ev])
t=np.cumsum(ms)
for i in [0,1,2,5,10,20,30,50,75,100,150,200,300,500,1000,2000,2999]:
    print("launch %5d  t=%8.1f ms  this %.4f ms   mean of next 20: %.4f" % (i, t[i], ms[i], ms[i:i+20].mean()))
# idle gap then again
time.sleep(0.2)
ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(60)]
with torch.no_grad():
    for a,b in ev:
        a.record(); lp=nf.log_prob(z); b.record()
torch.cuda.synchronize()
ms2=np.array([a.elapsed_time(b) for a,b in ev])
print("after a 0.2 s idle gap: first 20 mean %.4f, launches 40-60 mean %.4f" % (ms2[:20].mean(), ms2[40:].mean()))
