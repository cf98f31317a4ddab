#!/usr/bin/env python3
"""Headline benchmark: M samples/s of NormFlow.log_prob, D=64, 8 RealNVP coupling layers
(num_stages=4, L=2, U=15), N=2^20 samples per GPU, float32, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one NormFlow.log_prob call over one (1, 2^20, 64) batch through the C ABI
(tnf_flow_log_prob_f32).  The path shards over samples with no data-path collective, so
with N GPUs every rank evaluates its own 2^20-sample batch (weak scaling) and `value` is
the samples all ranks processed per second of the slowest rank.

One JSON line is printed by rank 0.  Besides the contract's keys it carries
  roofline      -- the dominant kernel of the timed path, from HIP events recorded on the
                   stream the kernels run on, inside the timed region;
  cpu_baseline  -- the PyTorch-CPU oracle (oracle/flow_oracle.py, verified equal to the
                   reference in the build container) timed on this box's host cores;
  layer_chain   -- the same call with one fused kernel per coupling layer (k = 8, the
                   HBM-bound design), measured after the timed region, with its own roofline;
  parity        -- max relative error of the GPU log_prob vs the oracle on 2^16 samples;
  train_step    -- BASELINE configs[3] per GPU: -mean(log_prob) on 2^19 samples, backward (reversible pair), Adam
                   (with N GPUs: + the RCCL all-reduce of the flat gradient), outside the timed region;
  widened       -- (1 GPU) the SURVEY 8f rows: fused conditional flow, AR log_prob, the LFI step as a HIP graph,
                   sampling with fresh batch statistics under autograd.
--no-extras skips train_step and widened (a clean per-kernel average under rocprofv3 --stats).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
F16_MFMA_TFLOPS = 2500.0  # dense f16 MFMA peak (MI355X_MICROARCH.md)
F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32 MFMA = f32 vector peak (spec)

D, S, L, U = 64, 4, 2, 15
N_PER_GPU = 1 << 20


def flop_per_sample(D, S, L, U):
    """F_alg = L_c*[4*(h*U + (L-1)*U^2 + U*h) + 2*(L*U + h) + 6h] + 6*D*S + 3*D."""
    h, Lc = D // 2, 2 * S
    return Lc * (4 * (h * U + (L - 1) * U * U + U * h) + 2 * (L * U + h) + 6 * h) + 6 * D * S + 3 * D


def bytes_per_sample_chain(D, k):
    """B_alg(k) = 4D(2k-1) + 8(k-1) + 4 for k round trips of z through HBM."""
    return 4 * D * (2 * k - 1) + 8 * (k - 1) + 4


def build_model(tnf, dev):
    """NormFlow(64, False, 'coupling', 4, 2, 15) with xavier_normal_ params (seed 0) and BatchNorm
    statistics populated by one 4,096-sample batch-mode forward."""
    torch.manual_seed(0)
    np.random.seed(0)
    nf = tnf.NormFlow(D, False, "coupling", S, L, U, device=dev)
    with torch.no_grad():
        nf(4096)  # batch-statistics forward on the GPU; caches mean/alpha in every BatchNorm
    return nf


def cpu_baseline(nf, threads):
    """Time the oracle on the host: same model, same synthetic z (generator seed 1)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import flow_oracle as orc

    torch.set_num_threads(threads)
    params = nf.params.detach().cpu()
    stats = [(b.get_last_mean().cpu(), b.get_last_alpha().cpu()) for b in nf._bn_layers()]
    n = N_PER_GPU
    z = torch.randn(1, n, D, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        t0 = time.perf_counter()
        orc.flow_log_prob(z, params, D, S, L, U, stats)  # warm-up
        warm = time.perf_counter() - t0
        if warm > 20.0:  # keep the default run within minutes
            n = 1 << 18
            z = z[:, :n].contiguous()
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            orc.flow_log_prob(z, params, D, S, L, U, stats)
            best = min(best, time.perf_counter() - t0)
    return {
        "value": round(n / best / 1e6, 4),
        "unit": "M samples/s",
        "cores": threads,
        "kind": "port",
        "sample": "oracle/flow_oracle.py flow_log_prob (PyTorch CPU, fp32, no_grad) on z (1, %d, %d), "
                  "1 warm-up + best of 3, %d torch threads" % (n, D, threads),
    }, orc, params, stats


def widened_rows(tnf):
    """Single-GPU extras for the SURVEY 8f rows built after the metric path: not the
    metric, a few seconds in total."""
    def timeit(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    res = {}
    np.random.seed(0)
    torch.manual_seed(0)
    nf = tnf.NormFlow(64, True, "coupling", 4, 2, 15)
    cde = tnf.ConditionalDensityEstimator(nf, 32, [64, 64])
    M = 1 << 18
    x = torch.randn(M, 32, device="cuda")
    z = torch.randn(M, 1, 64, device="cuda")

    def infer():
        with torch.no_grad():
            cde.log_prob(z, x)

    def train():
        cde.zero_grad()
        (-cde.log_prob(z, x).mean()).backward()

    t = timeit(infer, 5)
    res["cond_flow_log_prob"] = {"contexts": M, "ms": round(t * 1e3, 3), "value": round(M / t / 1e6, 1),
                                 "unit": "M contexts/s", "what": "cde.log_prob(z[:, None, :], x), D=64 S=4, param_net "
                                 "[64,64]: last Linear fused into the flow kernel, params (M, 20464) never materialised"}
    t = timeit(train, 3)
    res["cond_flow_train_step"] = {"contexts": M, "ms": round(t * 1e3, 3), "value": round(M / t / 1e6, 2),
                                   "unit": "M contexts/s", "what": "forward + backward through param_net (fused pair)"}
    del x, z, cde, nf
    nf = tnf.NormFlow(16, False, "AR", 1, 2, 32)
    z = torch.randn(1, 1 << 20, 16, device="cuda")
    with torch.no_grad():
        nf(64)
        t = timeit(lambda: nf.log_prob(z), 10)
    res["ar_log_prob"] = {"samples": 1 << 20, "ms": round(t * 1e3, 3), "value": round((1 << 20) / t / 1e6, 1),
                          "unit": "M samples/s", "what": "NormFlow(16, arch_type='AR', num_layers=2, num_units=32).log_prob, "
                          "one matrix-pipe MAF kernel"}
    del z, nf
    # the LFI scripts' inner step (scripts/lfi_mat.py:23-57): AR flow + ToInterval through param_net [64,64]
    D_l, M_l, N_l = 6, 2000, 100
    lb, ub = -2.0 * np.ones(D_l), 2.0 * np.ones(D_l)
    lb[::2] = -np.inf
    nfl = tnf.NormFlow(D_l, True, "AR", 1, 2, 2 * D_l, tnf.ToInterval(D_l, lb, ub))
    cdel = tnf.ConditionalDensityEstimator(nfl, 3, [64, 64])
    xl = torch.randn(M_l, 3, device="cuda")
    zl = torch.rand(M_l, N_l, D_l, device="cuda") * 3.0 - 1.5
    optl = torch.optim.Adam(cdel.parameters(), lr=1e-3, capturable=True)

    def lfi_step():
        optl.zero_grad(set_to_none=True)
        loss = -cdel.log_prob(zl, xl).mean()
        loss.backward()
        optl.step()
        return loss.detach()

    te = timeit(lfi_step, 10)
    gs = tnf.graphs.GraphedStep(lfi_step, warmup=3)
    tg = timeit(gs, 20)
    res["lfi_train_step"] = {"samples": M_l * N_l, "ms": round(tg * 1e3, 3), "eager_ms": round(te * 1e3, 3),
                             "value": round(M_l * N_l / tg / 1e6, 1), "unit": "M samples/s",
                             "what": "AR flow (D=6) + ToInterval conditioned through param_net [64,64], 2000 contexts x 100 "
                             "samples: loss, one-kernel AR backward, Adam; replayed as one HIP graph (eager_ms: eagerly)"}
    del cdel, nfl, xl, zl, optl, gs
    # sampling with fresh batch statistics under autograd (the reference's train_efn objective shape)
    nfe = tnf.NormFlow(64, False, "coupling", 4, 2, 15)
    nfe.params = (torch.randn(1, nfe.D_params, device="cuda") * 0.1).requires_grad_()
    om = torch.randn(1, 1 << 19, 64, device="cuda")

    def efn():
        nfe.params.grad = None
        ze, lqe = nfe._forward_from(om, nfe.params, freeze_bn=False)
        (lqe.mean() + (ze ** 2).mean()).backward()

    t = timeit(efn, 5)
    res["forward_train_step"] = {"samples": 1 << 19, "ms": round(t * 1e3, 3), "value": round((1 << 19) / t / 1e6, 1),
                                 "unit": "M samples/s", "what": "z, log_q = nf(N) with fresh batch statistics, D=64 S=4; "
                                 "backward through the batch moments (one autograd node, tnf_flow_forward_train_*)"}
    return res


def event_ms(pairs):
    return [a.elapsed_time(b) for a, b in pairs]


def read_traffic(kernel_key):
    """HBM bytes per launch from the committed PMC summary (profiles/*_pmc.json), if any."""
    path = os.path.join(ROOT, "profiles", "r01_pmc.json")
    try:
        with open(path) as f:
            return json.load(f)[kernel_key]["hbm_bytes_per_launch"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline path only: skip the training step and the widened rows (keeps a rocprofv3 "
                         "--stats average of the headline kernel free of their smaller launches)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(os.cpu_count(), 16)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import torch_nf_amd as tnf

    L_ = tnf._lib
    nf = build_model(tnf, dev)
    if world > 1:  # identical model everywhere: rank 0's parameters and BatchNorm statistics
        dist.broadcast(nf.params.data, 0)
        for b in nf._bn_layers():
            m, a = b.get_last_mean().clone(), b.get_last_alpha().clone()
            dist.broadcast(m, 0)
            dist.broadcast(a, 0)
            b.set_last_stats(m, a)

    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    z = torch.randn(1, N_PER_GPU, D, device=dev, generator=gen)  # resident in HBM before timing

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(fusion, steps, warmup):
        nf.fusion = fusion
        with torch.no_grad():
            for _ in range(warmup):
                nf.log_prob(z)
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                     for _ in range(steps)]
            barrier()
            t0 = time.perf_counter()
            for a, b in pairs:
                a.record()
                out = nf.log_prob(z)
                b.record()
            barrier()
            wall = time.perf_counter() - t0
        return wall, event_ms(pairs), out

    fused = bool(L_.lib.tnf_flow_fused_supported(D, S, L, U))
    main_fusion = L_.FUSE_FLOW if fused else L_.FUSE_LAYER
    wall, ev, lp = run(main_fusion, args.steps, args.warmup)
    wall_t = torch.tensor([wall], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall_max = float(wall_t.item())

    # the per-layer chain (k = 8), outside the timed region
    wall_l, ev_l, lp_l = run(L_.FUSE_LAYER, max(5, args.steps // 2), 2)

    # BASELINE configs[3], per-GPU share: one training step (loss = -mean log_prob, backward, Adam)
    # on 2^19 samples; outside the timed region.  With N GPUs the only collective is the gradient.
    train_ms = None
    if (rank == 0 or world > 1) and not args.no_extras:
        n_tr = 1 << 19
        p_train = nf.params.detach().clone().requires_grad_()
        nf_params_saved, nf.params = nf.params, p_train
        opt = torch.optim.Adam([p_train], lr=1e-4)
        z_tr = z[:, :n_tr].contiguous()

        def train_step():
            opt.zero_grad(set_to_none=True)
            loss = -nf.log_prob(z_tr).mean()
            loss.backward()
            if world > 1:
                from torch_nf_amd.distributed import allreduce_gradients
                allreduce_gradients([p_train], average=True)
            opt.step()

        for _ in range(2):
            train_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            train_step()
        torch.cuda.synchronize()
        train_ms = (time.perf_counter() - t0) / 5 * 1e3
        nf.params = nf_params_saved

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_samples = N_PER_GPU * world * args.steps
    value = total_samples / wall_max / 1e6
    flops = flop_per_sample(D, S, L, U)
    ev_mean = float(np.mean(ev))  # ms per call on the launch stream = the one kernel of the call (it builds its operands in its prologue)
    if fused:
        achieved = N_PER_GPU * flops / (ev_mean * 1e-3) / 1e12
        # Matrix-pipe roof of THIS formulation: every fp32-accurate product is three f16 MFMA products
        # (hi*hi + lo*hi + hi*lo, fp32 accumulate), so the ceiling for algorithmic fp32 flops is the dense
        # f16 MFMA peak / 3.  (The kernel is not bound by it: sigmoids/exps and the operand splitting on the
        # vector pipe are -- DESIGN.md 3.4; the dense fp32 MFMA peak, 157.3 TFLOP/s, is already exceeded.)
        split_peak = F16_MFMA_TFLOPS / 3.0
        roofline = {"bound": "mfma", "kernel": "flow_fused_f16_kernel<32,2,inverse,2,8,4>", "achieved": round(achieved, 3),
                    "peak": round(split_peak, 1), "unit": "TFLOP/s", "frac": round(achieved / split_peak, 4),
                    "traffic": read_traffic("flow_fused_f16_kernel"),
                    "note": "achieved = algorithmic fp32 flops (42,176 per sample) / launch time; peak = dense f16 MFMA "
                            "peak (2,500 TFLOP/s) / 3, because each fp32-accurate contraction is issued as 3 split-f16 "
                            "MFMAs with fp32 accumulate; the binding unit is the vector pipe (sigmoids/exps + operand "
                            "splitting), not the matrix pipe or HBM",
                    "algorithmic_flop_per_sample": flops, "launch_ms": round(ev_mean, 4),
                    # what the matrix pipe itself sees: 24 f16 MFMAs per 16 samples and layer
                    # (6 x 16x16x32 + 18 x 16x16x16) = 15,360 flop/sample/layer, against 2.5 PFLOP/s dense f16
                    "issued_f16_mfma_tflops": round(N_PER_GPU * 15360 * 2 * S / (ev_mean * 1e-3) / 1e12, 1),
                    "f16_mfma_peak_tflops": F16_MFMA_TFLOPS,
                    "fp32_mfma_peak_tflops": F32_MFMA_TFLOPS,
                    "frac_of_fp32_mfma_peak": round(achieved / F32_MFMA_TFLOPS, 4),
                    "hbm_compulsory_frac": round(N_PER_GPU * bytes_per_sample_chain(D, 1) / (ev_mean * 1e-3) / 1e9
                                                 / HBM_PEAK_GBS, 4)}
    else:
        roofline = None
    k = 2 * S
    evl_mean = float(np.mean(ev_l))
    chain_bytes = N_PER_GPU * bytes_per_sample_chain(D, k)
    chain_gbs = chain_bytes / (evl_mean * 1e-3) / 1e9
    layer_chain = {
        "value": round(N_PER_GPU / (evl_mean * 1e-3) / 1e6, 2), "unit": "M samples/s", "launches": k,
        "ms_per_step": round(evl_mean, 4),
        "roofline": {"bound": "hbm", "kernel": "coupling_mfma_kernel<32,2,inverse>",
                     "achieved": round(chain_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(chain_gbs / HBM_PEAK_GBS, 4),
                     "traffic": read_traffic("coupling_mfma_kernel"),
                     "algorithmic_bytes_per_sample_all_launches": bytes_per_sample_chain(D, k)},
    }
    if roofline is None:
        roofline = layer_chain["roofline"]

    out = {
        "metric": "M samples/sec NormFlow.log_prob, D=64 RealNVPx8, N=2^20",
        "value": round(value, 2),
        "unit": "M samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall_max / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "NormFlow(64,False,'coupling',num_stages=4,num_layers=2,num_units=15).log_prob, "
                               "z (1, 2^20, 64) fp32 per GPU resident in HBM, xavier params seed 0, "
                               "BN stats from one 4096-sample forward",
                   "samples_per_gpu": N_PER_GPU, "D": D, "coupling_layers": 2 * S,
                   "fusion": "whole-flow kernel (k=1), split-f16 MFMA" if fused else "one kernel per coupling layer (k=8)",
                   "arithmetic": "fp32 I/O, fp32 accumulate and VALU; matrix operands split hi+lo into f16",
                   "sharding": "samples, no collective in the timed region"},
        "roofline": roofline,
        "layer_chain": layer_chain,
        "train_step": None if train_ms is None else {
            "ms": round(train_ms, 3), "samples_per_gpu": 1 << 19,
            "value": round((1 << 19) * world / (train_ms * 1e-3) / 1e6, 1), "unit": "M samples/s",
            "what": "loss = -mean(log_prob): whole-flow forward keeping z0, one-kernel reversible backward (split-f16 MFMA); "
                    + ("RCCL all-reduce of the flat gradient; " if world > 1 else "") + "Adam step"},
    }

    if world == 1 and not args.no_extras:
        out["widened"] = widened_rows(tnf)

    if not args.no_cpu_baseline and world == 1:
        threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
        base, orc, params, stats = cpu_baseline(nf, threads)
        out["cpu_baseline"] = base
        sl = slice(1 << 19, (1 << 19) + (1 << 16))
        with torch.no_grad():
            want = orc.flow_log_prob(z[:, sl].cpu(), params, D, S, L, U, stats)
        rel = ((lp[:, sl].cpu() - want).abs() / want.abs().clamp_min(1e-3)).max().item()
        rel_l = ((lp_l[:, sl].cpu() - want).abs() / want.abs().clamp_min(1e-3)).max().item()
        out["parity"] = {"max_rel_err_vs_oracle": float("%.3g" % rel), "layer_chain_max_rel_err": float("%.3g" % rel_l),
                         "samples": 1 << 16, "tolerance": 1e-5}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
