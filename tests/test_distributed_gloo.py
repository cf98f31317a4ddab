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
