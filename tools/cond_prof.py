#!/usr/bin/env python3
"""Workload for rocprofv3 / tools/pmc.sh: 3 fused conditional-flow training steps and 3 fused log_prob calls
(D=64, S=4, param_net [64,64]) at argv[1] contexts (default 2^20)."""
import sys
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_nf_amd as tnf
torch.manual_seed(0); np.random.seed(0)
D, S, L, U = 64, 4, 2, 15
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
nf = tnf.NormFlow(D, True, "coupling", S, L, U)
cde = tnf.ConditionalDensityEstimator(nf, 32, [64, 64])
x = torch.randn(M, 32, device="cuda"); z = torch.randn(M, 1, D, device="cuda")
for _ in range(3):
    cde.zero_grad()
    (-cde.log_prob(z, x).mean()).backward()
with torch.no_grad():
    for _ in range(3):
        cde.log_prob(z, x)
torch.cuda.synchronize()
