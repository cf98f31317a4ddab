#!/usr/bin/env python3
"""fp32 vs bf16 tolerance sweep (BASELINE.json configs[4]) -- PARITY UNPINNED: the reference has no
reduced-precision path and its LFI driver is not in the snapshot, so nothing here is compared with the reference;
the yardstick is this package's own fp32-accurate path.

`ops.operand_precision("bf16")` (TNF_OPT_OPERAND_PREC = 1) rounds every conditioner-MLP operand to bf16:
  * RealNVP log_prob (flow_fused2.hip layer-range kernel): ONE v_mfma_f32_16x16x{32,16}_bf16 per contraction instead of
    the three split-f16 products -- a real bf16 kernel, so its time is reported too;
  * autoregressive flow (maf_mfma.hip, maf_bwd_mfma.hip: the LFI step): operands rounded to bf16 and fed to the fp32 MFMA,
    which reproduces a bf16 MFMA with fp32 accumulation up to summation order -- accuracy only, no speed claim.
The context network (torch Linear layers) and everything outside the conditioner contractions stay fp32.

Sections: (1) log_prob error + time on the BASELINE shapes, per operand scale; (2) the LFI step of scripts/lfi_mat.py:23-57:
log_prob / gradient error of one step; (3) APT training twice from the same seed, fp32 and bf16: loss trajectories, with a
second fp32 seed as the run-to-run yardstick.   Usage: python tools/bf16_sweep.py [--iters 300] [--out FILE]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_nf_amd as tnf  # noqa: E402
from torch_nf_amd import _lib, ops  # noqa: E402
from torch_nf_amd.lfi import train_APT  # noqa: E402
from torch_nf_amd.systems import Mat  # noqa: E402

OUT = []


def say(s=""):
    print(s, flush=True)
    OUT.append(s)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def err_stats(a, b):
    """|a - b| relative to |b| (b = fp32 path), over finite entries."""
    a, b = a.double().flatten(), b.double().flatten()
    rel = (a - b).abs() / b.abs().clamp_min(1e-30)
    return rel.max().item(), rel.median().item(), (a - b).abs().max().item()


def coupling_section(N):
    say("== 1. RealNVP log_prob, N = 2^%d samples, 8 coupling layers (num_stages 4, L 2, U 15) ==" % int(np.log2(N)))
    say("%-4s %-12s %-7s | %-11s %-11s %-11s | %-9s %-9s %-6s" % ("D", "weights", "fusion", "max rel", "median rel", "max abs",
                                                                  "fp32 ms", "bf16 ms", "ratio"))
    for D in (32, 64):
        S, L, U = 4, 2, 15
        for wscale, label in ((1.0, "init"), (3.0, "init x 3"), (10.0, "init x 10")):
            torch.manual_seed(0)
            np.random.seed(0)
            nf = tnf.NormFlow(D, False, "coupling", S, L, U)
            with torch.no_grad():
                nf(4096)  # batch statistics for the BatchNorm layers
                nf.params.mul_(wscale)
                z, _ = nf.sample(N) if wscale == 1.0 else (torch.randn(1, N, D, device="cuda"), None)
            z = z.contiguous()
            mean, alpha = nf._bn_stats(z.device)
            for fusion, fl in ((_lib.FUSE_FLOW, "flow"), (_lib.FUSE_LAYER, "layer")):
                def run():
                    return ops.flow_log_prob_raw(z, nf.params, mean, alpha, D, S, L, U, fusion=fusion)[0]
                ref = run()
                t32 = timeit(run)
                with ops.operand_precision("bf16"):
                    got = run()
                    t16 = timeit(run)
                mx, med, ab = err_stats(got, ref)
                say("%-4d %-12s %-7s | %-11.3e %-11.3e %-11.3e | %-9.4f %-9.4f %-6.2f" % (D, label, fl, mx, med, ab, t32 * 1e3,
                                                                                         t16 * 1e3, t32 / t16))
    say("(rel = |lp_bf16 - lp_fp32| / |lp_fp32|; the fp32 column is the default split-f16 path, itself <= 1e-5 of the oracle)")
    say()


def build_lfi(d, seed):
    np.random.seed(seed)
    torch.manual_seed(seed)
    mat = Mat(d, noise=0.05)
    x0 = np.array([[0.0, d / 2]])
    nf = tnf.NormFlow(mat.D, True, "AR", num_stages=1, num_layers=2, num_units=2 * mat.D,
                      support_layer=tnf.ToInterval(mat.D, mat.lb, mat.ub))
    cde = tnf.ConditionalDensityEstimator(nf, x0.shape[1], [64, 64], dropout=False)
    return mat, x0, cde


def lfi_step_section(d, M, atoms):
    say("== 2. one LFI step (scripts/lfi_mat.py:23-57): AR flow + ToInterval, d = %d, %d contexts x %d atoms ==" % (d, M, atoms))
    mat, x0, cde = build_lfi(d, 1)
    torch.manual_seed(2)
    x = torch.randn(M, x0.shape[1], device="cuda")
    z = torch.rand(M, atoms, mat.D, device="cuda") * 3.0 - 1.5
    res = {}
    for prec in ("fp32", "bf16"):
        with ops.operand_precision(prec):
            cde.zero_grad()
            lp = cde.log_prob(z, x)
            loss = -lp.mean()
            loss.backward()
            grads = torch.cat([p.grad.flatten() for p in cde.param_net.parameters()]).clone()
        res[prec] = (lp.detach().clone(), loss.item(), grads)
    mx, med, ab = err_stats(res["bf16"][0], res["fp32"][0])
    g32, g16 = res["fp32"][2].double(), res["bf16"][2].double()
    say("log_prob: max rel %.3e  median rel %.3e  max abs %.3e" % (mx, med, ab))
    say("loss: fp32 %.6f  bf16 %.6f  (diff %.3e)" % (res["fp32"][1], res["bf16"][1], abs(res["fp32"][1] - res["bf16"][1])))
    say("context-net gradient: |g_bf16 - g_fp32| / |g_fp32| = %.3e (2-norms), cosine %.8f"
        % ((g16 - g32).norm().item() / g32.norm().item(), torch.dot(g16, g32).item() / (g16.norm() * g32.norm()).item()))
    say()


def lfi_train_section(d, iters, R, M, atoms):
    say("== 3. APT training, d = %d, R = %d rounds x %d iterations, %d contexts x %d atoms ==" % (d, R, iters, M, atoms))
    mat, x0, cde = build_lfi(d, 0)
    train_APT(cde, mat, x0, M=M, M_atom=atoms, R=1, num_iters=8)  # warm-up: allocator, graph capture, lazy init
    runs = {}
    for name, prec, seed in (("fp32", "fp32", 1), ("bf16", "bf16", 1), ("fp32 seed 2", "fp32", 2)):
        mat, x0, cde = build_lfi(d, seed)
        with ops.operand_precision(prec):
            cde, losses, zs, lps, it_time = train_APT(cde, mat, x0, M=M, M_atom=atoms, R=R, num_iters=iters)
        T = mat.simulate(zs[-1])
        runs[name] = (np.asarray(losses, dtype=np.float64), T.mean(0), T.std(0), it_time)
    w = max(1, iters // 10)
    ref = runs["fp32"][0]
    say("loss, mean over windows of %d iterations (same seed => same simulations and batches until rounding separates them):" % w)
    say("%-10s %-12s %-12s %-12s | %-12s %-12s" % ("iter", "fp32", "bf16", "fp32 seed 2", "|bf16-fp32|", "|seed2-fp32|"))
    for s in range(0, len(ref), w):
        a, b, c = (runs[k][0][s:s + w].mean() for k in ("fp32", "bf16", "fp32 seed 2"))
        say("%-10d %-12.5f %-12.5f %-12.5f | %-12.3e %-12.3e" % (s, a, b, c, abs(b - a), abs(c - a)))
    dv = np.abs(runs["bf16"][0] - ref)
    ds = np.abs(runs["fp32 seed 2"][0] - ref)
    first = int(np.argmax(dv > 1e-3)) if (dv > 1e-3).any() else -1
    say("per-iteration |loss_bf16 - loss_fp32|: first 10 its max %.3e, overall max %.3e, mean %.3e; first iteration above 1e-3: %d"
        % (dv[:10].max(), dv.max(), dv.mean(), first))
    say("yardstick, |loss_fp32(seed 2) - loss_fp32(seed 1)|: max %.3e, mean %.3e" % (ds.max(), ds.mean()))
    for k in ("fp32", "bf16", "fp32 seed 2"):
        say("%-12s posterior predictive T(x): mean %s  std %s  (target %s);  %.3f ms / iteration"
            % (k, np.round(runs[k][1], 4), np.round(runs[k][2], 4), [0.0, d / 2], runs[k][3] * 1e3))
    say("(the AR kernels emulate bf16 operands through the fp32 MFMA: iteration times are NOT a bf16 speed measurement)")
    say()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--d", type=int, default=3)
    ap.add_argument("--logn", type=int, default=20)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    say("# fp32 vs bf16 tolerance sweep -- parity unpinned (no reference counterpart); device: %s" % torch.cuda.get_device_name(0))
    say()
    coupling_section(1 << a.logn)
    lfi_step_section(a.d, 2000, 100)
    lfi_train_section(a.d, a.iters, a.rounds, 2000, 100)
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(OUT) + "\n")


if __name__ == "__main__":
    main()
