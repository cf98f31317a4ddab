#!/usr/bin/env python3
"""Shader clock held under the whole-flow kernel's own load (MI355X_MICROARCH.md, 'DVFS give-back' item 6).
Needs a DIAGNOSTIC build of libtnf_hip.so with -DTNF2_STAMP=1 (flow_fused2.hip: s_memtime / s_memrealtime stamps around
the main loop, written to the diagnostic counter buffer; never shipped):
  make -C torch_nf_amd/csrc stamp && TNF_LIB_PATH=torch_nf_amd/lib/libtnf_hip_stamp.so python tools/clock_probe.py
Runs the kernel back to back for ~3 s on random data, then reports the median over workgroups of
delta(s_memtime) / delta(s_memrealtime) x 100 MHz for the last launch, next to the same figure for a memory-bound launch
(the per-layer chain kernel cannot be stamped this way; its clock is read from GRBM_GUI_ACTIVE in profiles/r02_pmc.json)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_nf_amd as tnf  # noqa: E402
from torch_nf_amd import _lib  # noqa: E402

D, S, L, U, N = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 4, 2, 15, 1 << 20
torch.manual_seed(0)
np.random.seed(0)
nf = tnf.NormFlow(D, False, "coupling", S, L, U)
with torch.no_grad():
    nf(4096)
mean, alpha = nf._bn_stats(torch.device("cuda"))
z = torch.randn(1, N, D, device="cuda")
lp = torch.empty(1, N, device="cuda")
stamps = torch.zeros(2 * 256, dtype=torch.int32, device="cuda")
lib = _lib.lib
ws = torch.empty(_lib.check(lib.tnf_flow_workspace_bytes(1, N, D, S, L, U, _lib.FUSE_FLOW)), dtype=torch.uint8, device="cuda")
p = nf.params.detach()


def launch():
    _lib.check(lib.tnf_flow_log_prob_diag_f32(z.data_ptr(), p.data_ptr(), mean.data_ptr(), alpha.data_ptr(), None,
                                              lp.data_ptr(), None, None, 1, 1, N, D, S, L, U, p.shape[1], _lib.FUSE_FLOW,
                                              ws.data_ptr(), ws.numel(), _lib.stream_ptr(), stamps.data_ptr()))


t0 = time.time()
n = 0
while time.time() - t0 < 3.0:
    for _ in range(200):
        launch()
    torch.cuda.synchronize()
    n += 200
st = stamps.cpu().numpy().astype(np.int64).reshape(256, 2)
st = st[st[:, 1] > 0]
clk = st[:, 0] / st[:, 1] * 0.1  # cycles per 100 MHz tick -> GHz
print("whole-flow kernel, D = %d, after %d back-to-back launches: main loop %.1f us (median over %d workgroups), "
      "shader clock %.3f GHz (min %.3f, max %.3f)" % (D, n, np.median(st[:, 1]) / 100.0, len(st), np.median(clk), clk.min(), clk.max()))
