#!/usr/bin/env python3
"""Headline benchmark: M samples/s of NormFlow.log_prob, D=64, 8 RealNVP coupling layers
(num_stages=4, L=2, U=15), N=2^20 samples per GPU, float32, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU over RCCL (torch.distributed backend "nccl").  Either launch it under
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment, as the driver does) or just call
`python bench.py --gpus N`: a parent that never touches the GPU then starts the N ranks itself through
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a CHILD process
(nothing is ever re-exec'ed) and exits with its code.

A "step" is one NormFlow.log_prob call over one (1, 2^20, 64) batch through the C ABI
(tnf_flow_log_prob_f32).  The path shards over samples with no data-path collective, so
with N GPUs every rank evaluates its own 2^20-sample batch (weak scaling) and `value` is
the samples all ranks processed per second of the slowest rank.

Timing protocol: every timed row is  settle (--settle-ms of the same call back to back: an MI355X that has been idle
needs ~50 ms of load to reach the clock it holds)  ->  W warm-up steps  ->  barrier + synchronize ->
K timed steps  ->  barrier + synchronize; `value` comes from that.  The contract's literal W + K steps on the idle GPU
run first and are reported as `cold_start`.

One JSON line is printed by rank 0.  Besides the contract's keys it carries
  steady_state, cold_start -- the protocol above and the cold figure;
  roofline      -- the dominant kernel of the timed path, from HIP events recorded on the
                   stream the kernels run on, inside the timed region;
  cpu_baseline  -- the PyTorch-CPU oracle (oracle/flow_oracle.py, verified equal to the
                   reference in the build container) timed on this box's host cores (os.cpu_count() threads;
                   `threads16` = the same at 16 threads);
  layer_chain   -- the same call with one fused kernel per coupling layer (k = 8, the
                   HBM-bound design of north_star), measured after the timed region, with its own roofline;
  sampling_layer_chain -- (1 GPU) the sampling direction, frozen statistics, one fused kernel per coupling layer;
  parity        -- max relative error of the GPU log_prob vs the oracle on 2^16 samples (ENFORCED: exit code 1
                   when it exceeds the tolerance);
  rccl_ranks    -- sum over ranks of 1 through an RCCL all-reduce (proves N ranks took part);
  strong_scaling-- the same call with N = 2^20 samples IN TOTAL split over the ranks (north_star quotes both);
  train_step    -- BASELINE configs[3] per GPU: -mean(log_prob) on 2^19 samples, backward (reversible pair), Adam
                   (with N GPUs: + the RCCL all-reduce of the flat gradient), outside the timed region; also with the
                   backward's overflow recovery off / by host read-back, and replayed as one HIP graph;
  configs       -- (1 GPU) BASELINE configs[1] (D = 32, own roofline) and configs[2] (ConditionalDensityEstimator,
                   (M, N) = (16, 2^16): frozen forward + log_prob);
  widened       -- (1 GPU) the SURVEY 8f rows: fused conditional flow, AR log_prob, the LFI step as a HIP graph,
                   sampling with fresh batch statistics under autograd.
--no-extras skips train_step, configs and widened (a clean per-kernel average under rocprofv3 --stats).
"""
import argparse
import gc
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
F16_MFMA_TFLOPS = 2500.0  # dense f16 MFMA peak (MI355X_MICROARCH.md)
F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32 MFMA = f32 vector peak (spec)

D, S, L, U = 64, 4, 2, 15
N_PER_GPU = 1 << 20
PARITY_TOL = 1e-5


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="run the measured call back to back for this long before the W warm-up steps, so that the GPU has "
                         "reached the clock it holds under sustained load (it needs ~50 ms after an idle gap); 0 = off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline path only: skip the training step, the configs rows and the widened rows (keeps a "
                         "rocprofv3 --stats average of the headline kernel free of their smaller launches)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = os.cpu_count()")
    ap.add_argument("--dry-run-launcher", action="store_true",
                    help="print the command the self-launcher would run for --gpus N and exit (no GPU, no children)")
    return ap.parse_args()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(args, port=None):
    """The child command `python bench.py --gpus N` starts when it is not already one rank of N."""
    port = port or _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_extras:
        cmd.append("--no-extras")
    if args.cpu_threads:
        cmd += ["--cpu-threads", str(args.cpu_threads)]
    return cmd


def self_launch(args):
    """Parent of a plain `python bench.py --gpus N` (N > 1): it has made no GPU call (torch is not even imported
    yet) and starts the ranks as a child process group; rank 0's JSON line reaches stdout through the child."""
    cmd = launcher_command(args)
    if args.dry_run_launcher:
        print(json.dumps({"launcher": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this host driver)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    return subprocess.run(cmd, env=env).returncode


def flop_per_sample(D, S, L, U):
    """F_alg = L_c*[4*(h*U + (L-1)*U^2 + U*h) + 2*(L*U + h) + 6h] + 6*D*S + 3*D."""
    h, Lc = D // 2, 2 * S
    return Lc * (4 * (h * U + (L - 1) * U * U + U * h) + 2 * (L * U + h) + 6 * h) + 6 * D * S + 3 * D


def bytes_per_sample_chain(D, k):
    """B_alg(k) = 4D(2k-1) + 8(k-1) + 4 for k round trips of z through HBM."""
    return 4 * D * (2 * k - 1) + 8 * (k - 1) + 4


def kernel_source_hash():
    """Hash of the kernel sources: a PMC summary under profiles/ is only quoted for the code it was collected on."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "torch_nf_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    return h.hexdigest()[:16]


def read_traffic(kernel_key):
    """HBM bytes per launch from the newest committed PMC summary whose `kernel_source_sha16` equals the hash of the
    kernel sources in this tree; None (and the reason) otherwise -- a stale counter is not a measurement."""
    prof = os.path.join(ROOT, "profiles")
    cur = kernel_source_hash()
    cands = sorted((f for f in os.listdir(prof) if f.endswith("pmc.json")), reverse=True) if os.path.isdir(prof) else []
    for name in cands:
        try:
            with open(os.path.join(prof, name)) as f:
                js = json.load(f)
        except Exception:
            continue
        if js.get("kernel_source_sha16") != cur or kernel_key not in js:
            continue
        return js[kernel_key].get("hbm_bytes_per_launch"), {"file": "profiles/" + name, "kernel_source_sha16": cur}
    return None, {"file": None, "kernel_source_sha16": cur,
                  "why_null": "no profiles/*pmc.json was collected on these kernel sources"}


def _time_oracle(torch, orc, z, params, D, S, L, U, stats, threads, reps):
    torch.set_num_threads(threads)
    with torch.no_grad():
        orc.flow_log_prob(z, params, D, S, L, U, stats)  # warm-up
        best = float("inf")
        for _ in range(reps):
            t0 = time.perf_counter()
            orc.flow_log_prob(z, params, D, S, L, U, stats)
            best = min(best, time.perf_counter() - t0)
    return best


def cpu_baseline(nf, threads, D, S, L, U):
    """Time the oracle on the host: same model, same kind of synthetic z (generator seed 1), bounded samples (~20 s of
    CPU work in all).  `threads` = os.cpu_count() (or --cpu-threads); a one-GPU box exposes all 256 hardware threads of
    its host but schedules only a 16-CPU share, so that count can be slower than 16 threads by an order of magnitude:
    both are timed, the FASTER one is `value` / `cores`, the other is reported beside it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import flow_oracle as orc
    import torch

    params = nf.params.detach().cpu()
    stats = [(b.get_last_mean().cpu(), b.get_last_alpha().cpu()) for b in nf._bn_layers()]
    rows = []
    for thr, n, reps in ([(min(16, threads), 1 << 19, 2)] + ([(threads, 1 << 14, 1)] if threads > 16 else [])):
        z = torch.randn(1, n, D, generator=torch.Generator().manual_seed(1))
        best = _time_oracle(torch, orc, z, params, D, S, L, U, stats, thr, reps)
        rows.append({"value": round(n / best / 1e6, 4), "cores": thr, "samples": n, "reps": "1 warm-up + best of %d" % reps})
    torch.set_num_threads(min(16, threads))
    best = max(rows, key=lambda r: r["value"])
    row = {
        "value": best["value"], "unit": "M samples/s", "cores": best["cores"], "kind": "port",
        "sample": "oracle/flow_oracle.py flow_log_prob (PyTorch CPU, fp32, no_grad) on z (1, %d, %d), %s, %d torch threads "
                  "(os.cpu_count() = %d)" % (best["samples"], D, best["reps"], best["cores"], os.cpu_count() or 1),
        "all_thread_counts": rows,
    }
    return row, orc, params, stats


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_run_launcher or (world == 1 and args.gpus > 1):
        sys.exit(self_launch(args))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 control flow on a one-GPU box: TNF_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses
    # gloo (RCCL refuses two ranks on one device); the line then says so and is not a scaling measurement
    rehearse = os.environ.get("TNF_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())

    import torch_nf_amd as tnf

    L_ = tnf._lib

    def build_model(D_):
        """NormFlow(D, False, 'coupling', 4, 2, 15) with xavier_normal_ params (seed 0) and BatchNorm
        statistics populated by one 4,096-sample batch-mode forward."""
        torch.manual_seed(0)
        np.random.seed(0)
        nf_ = tnf.NormFlow(D_, False, "coupling", S, L, U, device=dev)
        with torch.no_grad():
            nf_(4096)  # batch-statistics forward on the GPU; caches mean/alpha in every BatchNorm
        return nf_

    nf = build_model(D)
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

    def max_over_ranks(x):
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def settle(call, ms):
        """Back-to-back launches of `call` for `ms` of wall time (synchronising every 25): after an idle gap an MI355X
        takes ~50 ms of sustained load to reach its steady clock -- the whole-flow kernel runs 0.278 ms per launch at
        first and 0.220 ms from the 200th launch on.  W = 5 warm-up steps end inside that ramp."""
        n = 0
        if ms > 0:
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < ms:
                for _ in range(25):
                    call()
                torch.cuda.synchronize()
                n += 25
        return n

    def run(model, zz, fusion, steps, warmup, settle_ms=None):
        model.fusion = fusion
        with torch.no_grad():
            # events first: HIP creates an event at its first record, and now and then that costs the host tens of ms --
            # neither inside the timed region nor between the settle phase and it (an idle GPU drops its clock again)
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                     for _ in range(steps)]
            for a, b in pairs:
                a.record()
                b.record()
            gc.collect()
            gc.disable()
            settle(lambda: model.log_prob(zz), args.settle_ms if settle_ms is None else settle_ms)
            for _ in range(warmup):
                model.log_prob(zz)
            barrier()
            t0 = time.perf_counter()
            for a, b in pairs:
                a.record()
                out = model.log_prob(zz)
                b.record()
            barrier()
            wall = time.perf_counter() - t0
            gc.enable()
        model.fusion = L_.FUSE_AUTO
        return wall, [a.elapsed_time(b) for a, b in pairs], out

    fused = bool(L_.lib.tnf_flow_fused_supported(D, S, L, U))
    main_fusion = L_.FUSE_FLOW if fused else L_.FUSE_LAYER
    # the contract's literal W + K steps on a GPU that has just been idle (the model set-up above), for the record ...
    wall_c, ev_c, _ = run(nf, z, main_fusion, args.steps, args.warmup, settle_ms=0.0)
    wall_c = max_over_ranks(wall_c)
    # ... and the measurement proper: the same W + K steps once the clock has settled
    wall, ev, lp = run(nf, z, main_fusion, args.steps, args.warmup)
    wall_max = max_over_ranks(wall)

    # the per-layer chain (k = 8), outside the timed region
    wall_l, ev_l, lp_l = run(nf, z, L_.FUSE_LAYER, max(5, args.steps // 2), 2)

    # strong scaling: 2^20 samples in total, split over the ranks (north_star: "N=2^20 ... at 1, 2, 4 and 8 GPUs")
    strong = None
    if world > 1:
        from torch_nf_amd.distributed import shard_bounds

        lo, hi = shard_bounds(N_PER_GPU, world, rank)
        z_s = z[:, :hi - lo].contiguous()
        wall_s, _, _ = run(nf, z_s, main_fusion, args.steps, args.warmup)
        wall_s = max_over_ranks(wall_s)
        strong = {"samples_total": N_PER_GPU, "samples_per_gpu": hi - lo, "ms_per_step": round(wall_s / args.steps * 1e3, 4),
                  "value": round(N_PER_GPU * args.steps / wall_s / 1e6, 2), "unit": "M samples/s", "scaling": "strong"}

    # BASELINE configs[3], per-GPU share: one training step (loss = -mean log_prob, backward, Adam)
    # on 2^19 samples; outside the timed region.  With N GPUs the only collective is the gradient.
    train_ms = None
    if not args.no_extras:
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

        def time_train(reps=10):
            # settle by COUNT here, not by wall time: with N > 1 the step contains a collective, so every rank must
            # run the same number of them
            for _ in range(int(args.settle_ms / 0.75) if args.settle_ms > 0 else 0):
                train_step()
            for _ in range(2):
                train_step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                train_step()
            barrier()
            return max_over_ranks(time.perf_counter() - t0) / reps * 1e3

        train_ms = time_train()
        # the same step with the other overflow-recovery modes of the backward (ops._FlowLogProbRevFn.overflow_recovery):
        # default "device" = the fp32 recomputation is always enqueued, its kernels gated on the device-side flag;
        # "host" = read the flag back (a host synchronisation per step); "off" = no recovery (NaN poison stands);
        # and -- single GPU -- the default step replayed as ONE HIP graph (no host work at all)
        from torch_nf_amd import ops as _ops
        _ops._FlowLogProbRevFn.overflow_recovery = "off"
        train_ms_off = time_train()
        _ops._FlowLogProbRevFn.overflow_recovery = "host"
        train_ms_host = time_train()
        _ops._FlowLogProbRevFn.overflow_recovery = "device"
        train_ms_graph = None
        if world == 1:
            try:
                opt = torch.optim.Adam([p_train], lr=1e-4, capturable=True)
                gs = tnf.graphs.GraphedStep(train_step, warmup=3)
                for _ in range(2):
                    gs()
                barrier()
                t0 = time.perf_counter()
                for _ in range(10):
                    gs()
                barrier()
                train_ms_graph = (time.perf_counter() - t0) / 10 * 1e3
            except RuntimeError as e:  # capture unavailable: the eager numbers stand
                train_ms_graph = "unavailable: " + str(e).splitlines()[0][:120]
        nf.params = nf_params_saved

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    def fused_roofline(D_, ev_mean, n, kernel):
        flops = flop_per_sample(D_, S, L, U)
        achieved = n * flops / (ev_mean * 1e-3) / 1e12
        # Matrix-pipe roof of THIS formulation: every fp32-accurate product is three f16 MFMA products
        # (hi*hi + lo*hi + hi*lo, fp32 accumulate), so the ceiling for algorithmic fp32 flops is the dense
        # f16 MFMA peak / 3.  (The kernel is not bound by it: sigmoids/exps and the operand splitting on the
        # vector pipe are -- DESIGN.md 3.4; the dense fp32 MFMA peak, 157.3 TFLOP/s, is already exceeded.)
        split_peak = F16_MFMA_TFLOPS / 3.0
        traffic, stamp = read_traffic("flow_fused2_kernel") if D_ == 64 else (None, None)
        r = {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 3), "peak": round(split_peak, 1),
             "unit": "TFLOP/s", "frac": round(achieved / split_peak, 4), "traffic": traffic,
             "algorithmic_flop_per_sample": flops, "launch_ms": round(ev_mean, 4),
             "fp32_mfma_peak_tflops": F32_MFMA_TFLOPS, "frac_of_fp32_mfma_peak": round(achieved / F32_MFMA_TFLOPS, 4),
             "algorithmic_bytes_per_launch": int(n * bytes_per_sample_chain(D_, 1)),
             "hbm_compulsory_frac": round(n * bytes_per_sample_chain(D_, 1) / (ev_mean * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if stamp is not None:
            r["traffic_source"] = stamp
        return r

    def chain_roofline(D_, evl_mean, n, kernel):
        k = 2 * S
        gbs = n * bytes_per_sample_chain(D_, k) / (evl_mean * 1e-3) / 1e9
        traffic, stamp = read_traffic("flow_range2_kernel") if D_ == 64 else (None, None)
        r = {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
             "traffic_is": "HBM bytes per dispatch of the kernel, mean over the chain's 8 streaming launches AND its one "
                           "preparation launch (x 9/8 per streaming launch; PMC: 2 x FETCH_SIZE + WRITE_SIZE); it is BELOW "
                           "the SURVEY 8(d) figure because an in-place launch stores only the half it transformed",
             "algorithmic_bytes_per_launch_mean": int(n * bytes_per_sample_chain(D_, k) / k),
             "algorithmic_bytes_per_sample_all_launches": bytes_per_sample_chain(D_, k)}
        if stamp is not None:
            r["traffic_source"] = stamp
        return r

    total_samples = N_PER_GPU * world * args.steps
    value = total_samples / wall_max / 1e6
    ev_mean = float(np.mean(ev))  # ms per call on the launch stream = the one kernel of the call (it builds its operands in its prologue)
    roofline = None
    if fused:
        roofline = fused_roofline(D, ev_mean, N_PER_GPU, "flow_fused2_kernel<32,2,2,8,4>")
        roofline["note"] = ("achieved = algorithmic fp32 flops (42,176 per sample) / launch time; peak = dense f16 MFMA "
                            "peak (2,500 TFLOP/s) / 3, because each fp32-accurate contraction is issued as 3 split-f16 "
                            "MFMAs with fp32 accumulate; the binding unit is the vector pipe (sigmoids/exps + operand "
                            "splitting), not the matrix pipe or HBM")
        # what the matrix pipe itself sees: 24 f16 MFMAs per 16 samples and layer
        # (6 x 16x16x32 + 18 x 16x16x16) = 15,360 flop/sample/layer, against 2.5 PFLOP/s dense f16
        roofline["issued_f16_mfma_tflops"] = round(N_PER_GPU * 15360 * 2 * S / (ev_mean * 1e-3) / 1e12, 1)
        roofline["f16_mfma_peak_tflops"] = F16_MFMA_TFLOPS
    evl_mean = float(np.mean(ev_l))
    layer_chain = {
        "value": round(N_PER_GPU / (evl_mean * 1e-3) / 1e6, 2), "unit": "M samples/s", "launches": 2 * S,
        "ms_per_step": round(evl_mean, 4),
        "target_60pct_of_hbm_roof": round(0.6 * HBM_PEAK_GBS * 1e9 / bytes_per_sample_chain(D, 2 * S) / 1e6, 1),
        "roofline": chain_roofline(D, evl_mean, N_PER_GPU, "flow_range2_kernel<32,2,2,8,*,*,0> (one coupling layer per launch)"),
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
        "steady_state": {"settle_ms": args.settle_ms,
                         "what": "every timed row is preceded by settle_ms of the same call back to back, then its W warm-up "
                                 "steps: after an idle gap the GPU needs ~50 ms of sustained load to reach the clock it holds "
                                 "(whole-flow kernel: 0.278 ms per launch at first, 0.220 ms from the 200th on)"},
        "cold_start": {"value": round(N_PER_GPU * world * args.steps / wall_c / 1e6, 2), "unit": "M samples/s",
                       "ms_per_step": round(wall_c / args.steps * 1e3, 4), "launch_ms": round(float(np.mean(ev_c)), 4),
                       "what": "the same W warm-up + K timed steps without the settle phase, on a GPU idle since set-up"},
        "rccl_ranks": rccl_ranks,
        "collective_backend": None if world == 1 else ("gloo, all ranks on cuda:0 (REHEARSAL, not a measurement)" if rehearse
                                                       else "nccl (RCCL)"),
        "roofline": roofline,
        "layer_chain": layer_chain,
        "strong_scaling": strong,
        "train_step": None if train_ms is None else {
            "ms": round(train_ms, 3), "samples_per_gpu": 1 << 19,
            "overflow_recovery": "device-gated fp32 recomputation (default)",
            "ms_recovery_off": round(train_ms_off, 3), "ms_recovery_by_host_readback": round(train_ms_host, 3),
            "ms_as_one_hip_graph": train_ms_graph if not isinstance(train_ms_graph, float) else round(train_ms_graph, 3),
            "value": round((1 << 19) * world / (train_ms * 1e-3) / 1e6, 1), "unit": "M samples/s",
            "what": "loss = -mean(log_prob): whole-flow forward keeping z0, one-kernel reversible backward (split-f16 MFMA); "
                    + ("RCCL all-reduce of the flat gradient (81,856 B, one bucket); " if world > 1 else "") + "Adam step"},
    }

    if world == 1 and not args.no_extras:
        from tools import bench_rows

        cfg = {}
        # BASELINE configs[1]: D = 32, same depth
        nf32 = build_model(32)
        z32 = torch.randn(1, N_PER_GPU, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        _, e32, _ = run(nf32, z32, L_.FUSE_FLOW, 20, 3)
        _, e32l, _ = run(nf32, z32, L_.FUSE_LAYER, 10, 2)
        m32, m32l = float(np.mean(e32)), float(np.mean(e32l))
        cfg["configs[1]"] = {
            "what": "NormFlow(32,False,'coupling',4,2,15).log_prob, z (1, 2^20, 32)",
            "value": round(N_PER_GPU / (m32 * 1e-3) / 1e6, 1), "unit": "M samples/s", "ms_per_step": round(m32, 4),
            "roofline": fused_roofline(32, m32, N_PER_GPU, "flow_fused2_kernel<16,2,2,8,4>"),
            "layer_chain": {"value": round(N_PER_GPU / (m32l * 1e-3) / 1e6, 1), "ms_per_step": round(m32l, 4),
                            "roofline": chain_roofline(32, m32l, N_PER_GPU, "flow_range2_kernel<16,2,2,8,*,*,0> (one coupling layer per launch)")}}
        del nf32, z32
        cfg["configs[2]"] = bench_rows.config2_row(tnf, dev)
        out["configs"] = cfg
        # the SAMPLING direction with frozen statistics (density_estimator.py:374-388) as the k = 2S chain: one
        # flow_range2_kernel<.., FWD> launch per coupling layer (round 3; the fp32-MFMA chain it replaces ran at 0.55-0.57)
        bn_m, bn_a = nf._bn_stats(dev)
        omega = torch.randn(1, N_PER_GPU, D, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
        with torch.no_grad():
            settle(lambda: tnf.ops.flow_forward_raw(omega, nf.params, bn_m, bn_a, D, S, L, U, L_.FUSE_LAYER), args.settle_ms)
            prs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for a_, b_ in prs:
                a_.record()
                tnf.ops.flow_forward_raw(omega, nf.params, bn_m, bn_a, D, S, L, U, L_.FUSE_LAYER)
                b_.record()
            torch.cuda.synchronize()
        ms_s = float(np.mean([a_.elapsed_time(b_) for a_, b_ in prs]))
        out["sampling_layer_chain"] = {
            "what": "NormFlow.forward with frozen statistics (sampling direction), one fused kernel per coupling layer",
            "value": round(N_PER_GPU / (ms_s * 1e-3) / 1e6, 1), "unit": "M samples/s", "ms_per_step": round(ms_s, 4),
            "launches": 2 * S,
            "roofline": {"bound": "hbm", "kernel": "flow_range2_kernel<32,2,2,8,*,*,0,true> (one coupling layer per launch)",
                         "achieved": round(N_PER_GPU * bytes_per_sample_chain(D, 2 * S) / (ms_s * 1e-3) / 1e9, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(N_PER_GPU * bytes_per_sample_chain(D, 2 * S) / (ms_s * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": None,
                         "algorithmic_bytes_per_sample_all_launches": bytes_per_sample_chain(D, 2 * S)}}
        del omega
        out["widened"] = bench_rows.widened_rows(tnf)

    parity_ok = True
    if not args.no_cpu_baseline and world == 1:
        threads = args.cpu_threads or (os.cpu_count() or 1)
        base, orc, params, stats = cpu_baseline(nf, threads, D, S, L, U)
        out["cpu_baseline"] = base
        sl = slice(1 << 19, (1 << 19) + (1 << 16))
        with torch.no_grad():
            want = orc.flow_log_prob(z[:, sl].cpu(), params, D, S, L, U, stats)
        rel = ((lp[:, sl].cpu() - want).abs() / want.abs().clamp_min(1e-3)).max().item()
        rel_l = ((lp_l[:, sl].cpu() - want).abs() / want.abs().clamp_min(1e-3)).max().item()
        parity_ok = rel <= PARITY_TOL and rel_l <= PARITY_TOL
        out["parity"] = {"max_rel_err_vs_oracle": float("%.3g" % rel), "layer_chain_max_rel_err": float("%.3g" % rel_l),
                         "samples": 1 << 16, "tolerance": PARITY_TOL, "ok": parity_ok}
    print(json.dumps(out))
    sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()
    if not parity_ok:
        sys.exit("bench.py: GPU log_prob differs from the oracle by more than %g -- the number above is INVALID" % PARITY_TOL)


if __name__ == "__main__":
    main()
