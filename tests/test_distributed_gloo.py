"""The N > 1 path on CPU: world_size-2 gloo processes exercise the sample sharding, the
ragged all-gather, the one-bucket gradient all-reduce and the BatchNorm moment exchange of
torch_nf_amd/distributed.py.  The per-shard arithmetic is injected (the CPU oracle stands in
for the HIP kernels, which need a GPU); what is under test is the distributed plumbing."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

D, S, L, U, N = 8, 1, 2, 15, 37  # odd N -> ragged shards


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(seed=0):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import flow_oracle as orc

    rng = np.random.RandomState(seed)
    params = torch.tensor(rng.normal(0, 0.1, (1, orc.flow_num_params(D, S, L, U)))).float()
    stats = [(torch.tensor(rng.normal(0, 0.3, D)).float(), torch.tensor(np.exp(rng.normal(0, 0.2, D))).float())
             for _ in range(2 * S)]
    z = torch.tensor(rng.normal(0, 1, (1, N, D))).float()
    return orc, params, stats, z


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_nf_amd import distributed as tdist

        orc, params, stats, z = _setup()
        torch.set_num_threads(1)
        z_local = tdist.shard_samples(z)
        lo, hi = tdist.shard_bounds(N, world, rank)
        assert z_local.shape[1] == hi - lo

        # 1. density evaluation shards with no collective; gather only to compare
        fn = lambda zz: orc.flow_log_prob(zz, params, D, S, L, U, stats)
        lp_all = tdist.sharded_log_prob(fn, z_local, gather=True)
        full = fn(z)
        assert lp_all.shape == full.shape
        torch.testing.assert_close(lp_all, full, rtol=1e-6, atol=1e-6)

        # 2. training step: per-rank loss on the local shard, ONE all-reduce of the flat gradient
        p = params.clone().requires_grad_()
        loss_local = -orc.flow_log_prob(z_local, p, D, S, L, U, stats).sum() / N
        loss_local.backward()
        tdist.allreduce_gradients([p])
        p_full = params.clone().requires_grad_()
        (-orc.flow_log_prob(z, p_full, D, S, L, U, stats).mean()).backward()
        torch.testing.assert_close(p.grad, p_full.grad, rtol=1e-5, atol=1e-7)

        # 3. batch-statistics BatchNorm across shards: global moments from per-rank sums
        zl = z_local.reshape(-1, D).double()
        n, mean, var_b = tdist.allreduce_moments(torch.tensor(float(zl.shape[0])), zl.sum(0), (zl * zl).sum(0))
        zf = z.reshape(-1, D).double()
        assert int(n.item()) == N
        torch.testing.assert_close(mean, zf.mean(0), rtol=1e-12, atol=1e-12)
        torch.testing.assert_close(var_b, zf.var(0, unbiased=False), rtol=1e-10, atol=1e-12)
        out.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


# ---------------------------------------------------------------------------
# The exchange sequence of a sample-sharded batch-statistics forward (SURVEY 8e bullet 3): ops.run_batch_steps drives
# begin / layer / all-reduce of the moments / fold / end.  On the GPU the steps are the C ABI
# (ops.FlowForwardBatchSteps, tests/test_gpu_parity.py::test_batch_stats_forward_sharded_steps); here the same driver
# and the same reducer (distributed.moment_reducer over gloo) run over an oracle-backed stand-in for the kernels, and
# the two ranks' shards must reproduce the single-process forward with full-batch statistics.
# ---------------------------------------------------------------------------
class _OracleSteps:
    """The FlowForwardBatchSteps protocol over the CPU oracle (one coupling layer per step; the BatchNorm (+ Affine)
    behind it is applied in the next layer's load stage / in end(), from the moments as they are at fold time)."""

    def __init__(self, orc, omega, params, eps=1e-5):
        self.orc, self.params, self.eps = orc, params, eps
        self.n_layers = 2 * S
        self.z = torch.tensor(omega).float()
        self.log_q = torch.tensor(orc.base_log_density_f64(omega))
        self.layout = [(k, n, up) for k, n, up in orc.flow_layout(D, S, L, U)]
        self.pending = None  # (mean, alpha) of the BatchNorm behind the last layer
        self.stats = []

    def begin(self):
        self.offsets, off = [], 0
        for kind, n, up in self.layout:
            self.offsets.append(off)
            off += n

    def _apply_pending(self, c_prev):
        mean, alpha = self.pending
        self.z, ld = self.orc.bn_forward_frozen(self.z, mean, alpha)
        self.log_q = self.log_q - ld
        if c_prev & 1:  # the Affine that closes a stage
            i = [j for j, (k, _, _) in enumerate(self.layout) if k == "affine"][c_prev >> 1]
            n = self.layout[i][1]
            self.z, ld = self.orc.affine(self.z, self.params[:, self.offsets[i]:self.offsets[i] + n], D, False)
            self.log_q = self.log_q - ld

    def layer(self, c):
        if c > 0:
            self._apply_pending(c - 1)
        i = [j for j, (k, _, _) in enumerate(self.layout) if k == "coupling"][c]
        kind, n, up = self.layout[i]
        self.z, ld = self.orc.coupling(self.z, self.params[:, self.offsets[i]:self.offsets[i] + n], D, L, U, up, False)
        self.log_q = self.log_q - ld
        rows = self.z.reshape(-1, D).double()
        self.moments = torch.cat([rows.sum(0), (rows * rows).sum(0), torch.tensor([float(rows.shape[0])]).double()])
        return self.moments

    def fold(self, c):
        n = self.moments[2 * D]
        mean = self.moments[:D] / n
        var_b = (self.moments[D:2 * D] / n - mean * mean).clamp_min(0.0)
        self.pending = (mean.float(), torch.sqrt(var_b + self.eps).float())
        self.stats.append(self.pending)

    def end(self):
        self._apply_pending(self.n_layers - 1)
        return self.z, self.log_q, self.stats


def _worker_steps(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_nf_amd import distributed as tdist
        from torch_nf_amd import ops

        orc, params, _, _ = _setup(3)
        torch.set_num_threads(1)
        omega = np.random.RandomState(11).normal(0, 1, (1, N, D))
        lo, hi = tdist.shard_bounds(N, world, rank)
        z_l, lq_l, stats_l = ops.run_batch_steps(_OracleSteps(orc, omega[:, lo:hi], params), tdist.moment_reducer())
        z_full, lq_full, stats_full = orc.flow_forward(omega, params, D, S, L, U, None)
        torch.testing.assert_close(z_l, z_full[:, lo:hi], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(lq_l, lq_full[:, lo:hi], rtol=1e-6, atol=1e-4)
        for (m, a), (mf, af) in zip(stats_l, stats_full):
            torch.testing.assert_close(m, mf, rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(a, af, rtol=1e-5, atol=1e-6)
        # without the exchange the shard's own statistics differ (the test would be vacuous otherwise)
        z_loc, _, _ = ops.run_batch_steps(_OracleSteps(orc, omega[:, lo:hi], params), None)
        assert (z_loc - z_full[:, lo:hi]).abs().max() > 1e-3
        out.put((rank, "ok"))
    except Exception as e:
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_batch_statistics():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_steps, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_bench_launcher_dry_run():
    """`python bench.py --gpus N` (not under torch.distributed.run) starts its N ranks itself as a CHILD process
    group; the dry run prints that command without touching a GPU or starting anything."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "7", "--warmup", "2",
                          "--dry-run-launcher"], capture_output=True, text=True, env=env, timeout=120)
    assert res.returncode == 0, res.stderr
    cmd = json.loads(res.stdout.strip().splitlines()[-1])["launcher"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")):]
    assert tail[1:] == ["--gpus", "8", "--steps", "7", "--warmup", "2"]
