"""Multi-GPU execution of the flow path: one process per GPU, samples sharded.

Every sample's forward / inverse / log_prob is independent given the parameters and
the (frozen) BatchNorm statistics, so density evaluation shards along N with NO
data-path collective: each rank runs the same HIP kernels on its contiguous slice.
Collectives (RCCL over xGMI via torch.distributed backend "nccl"; "gloo" in the CPU
tests) appear only where the path has a real exchange step:
  * training: one all-reduce (sum) of the flat parameter gradient per step
    (D_params floats: 81,856 B for the D=64 / 4-stage model -- latency-bound, so it
    is sent as ONE bucket);
  * batch-statistics BatchNorm in a sharded `forward`: an all-reduce of the
    per-feature [count, sum, sum-of-squares] triple per BatchNorm layer.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world_size, rank):
    """Contiguous, balanced [lo, hi) slice of range(n) for `rank` (first n % world ranks get one extra)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank %d / world_size %d" % (rank, world_size))
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_samples(z, world_size=None, rank=None, dim=1):
    """This rank's slice of z along the sample axis (dim 1 of (M, N, D))."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    lo, hi = shard_bounds(z.shape[dim], world_size, rank)
    return z.narrow(dim, lo, hi - lo)


def sharded_log_prob(log_prob_fn, z_local, gather=False, group=None):
    """Evaluate `log_prob_fn` on this rank's samples; optionally all-gather the (M, N_local)
    results into (M, N) (ragged shards are padded to the longest and trimmed)."""
    lp = log_prob_fn(z_local)
    if not gather:
        return lp
    world = dist.get_world_size(group)
    n_local = torch.tensor([lp.shape[1]], device=lp.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(sizes)
    padded = lp if lp.shape[1] == n_max else torch.nn.functional.pad(lp, (0, n_max - lp.shape[1]))
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded.contiguous(), group=group)
    return torch.cat([p[:, :s] for p, s in zip(parts, sizes)], dim=1)


def allreduce_gradients(params, group=None, average=False):
    """Sum (or average) the .grad of every tensor in `params` across ranks with ONE
    flattened all-reduce (the gradient of the flow path is tiny; see module docstring)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    # RCCL reduces and averages in one collective; gloo (the CPU tests) has no AVG
    avg_op = average and dist.get_backend(group) == "nccl"
    op = dist.ReduceOp.AVG if avg_op else dist.ReduceOp.SUM
    if len(grads) == 1 and grads[0].is_contiguous():
        # the flow's own parameter row (NormFlow.params, 81,856 B at D=64): reduced IN PLACE -- one latency-bound
        # collective and no staging kernels around it
        dist.all_reduce(grads[0], op=op, group=group)
        if average and not avg_op:
            grads[0] /= dist.get_world_size(group)
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=op, group=group)
    if average and not avg_op:
        flat /= dist.get_world_size(group)
    torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)])


def moment_reducer(group=None):
    """The exchange step of a sample-sharded batch-statistics forward: a callable that sums, in place and across the
    ranks of `group`, the float64 [sum (D) | sum of squares (D) | row count] moments a BatchNorm layer's input has on
    this rank.  Install it as `NormFlow.batch_stats_reduce`: `nf(N, freeze_bn=False)` on every rank's shard then
    normalises with (and caches) the statistics of the global batch (bijectors.py:401-415 over all ranks' rows) --
    the 2*num_stages small all-reduces SURVEY 8(e) prescribes, enqueued between a layer kernel and its fold."""
    def reduce_(moments):
        dist.all_reduce(moments, op=dist.ReduceOp.SUM, group=group)
        return moments
    return reduce_


def allreduce_moments(count, total, total_sq, group=None):
    """Combine per-rank [count, sum, sum_sq] (float64) into global mean / biased variance --
    the exchange a sharded batch-statistics BatchNorm needs (bijectors.py:401-410 semantics
    over the GLOBAL batch)."""
    packed = torch.cat([count.reshape(1).double(), total.double().reshape(-1), total_sq.double().reshape(-1)])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    d = total.numel()
    n = packed[0]
    mean = packed[1:1 + d] / n
    var_b = packed[1 + d:1 + 2 * d] / n - mean * mean
    return n, mean, var_b.clamp_min(0.0)
