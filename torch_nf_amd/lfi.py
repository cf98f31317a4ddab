"""Sequential neural posterior estimation drivers (stand-in for the reference's `torch_nf.lfi`, which is NOT
part of the snapshot: only the call `train_APT(cnf, mat, x0, M=, M_atom=, R=, num_iters=, verbose=)` and its
five return values survive -- scripts/lfi_mat.py:44-57, notebooks/LFI_mat_det_trace.ipynb cell 3).  Written
from the published algorithm: APT / SNPE-C with atomic proposals (Greenberg, Nonnenmacher & Macke, "Automatic
Posterior Transformation for Likelihood-Free Inference", ICML 2019, eq. 7-8): in round r parameters are drawn
from the current posterior estimate q(. | x0) (the prior in round 0), simulated, and the conditional density
estimator is trained on all rounds' pairs with the contrastive loss

    L = - mean_i log [ q(z_i | x_i) / p(z_i)  /  sum_{j in atoms(i)} q(z_j | x_i) / p(z_j) ],

atoms(i) = {i} + (M_atom - 1) other parameters of the batch.  PARITY UNPINNED (no reference outputs exist).

The density evaluations are one `cde.log_prob(z (M, M_atom, D), x (M, D_x))` call per step -- the (contexts x
atoms) layout of this package's kernels (per-context flow parameters, M_atom samples each).

Data-parallel mode (BASELINE configs[4]: the LFI script on the GPUs of one node, one process per GPU): pass `group`
(or just initialise torch.distributed: the default group is used).  Every round's simulations and every step's batch
of contexts are split over the ranks -- a rank simulates and stores its slice of the round's parameters, draws its
M / world contexts from its own pairs, takes its atoms among them -- and the gradient of the context network is
averaged with ONE flattened all-reduce per step (distributed.allreduce_gradients: RCCL over xGMI with backend "nccl").
The ranks start from the same context network (broadcast from rank 0) and apply the same averaged gradient, so their
models stay bit-identical; no other collective is needed.  Steps run eagerly in this mode (a captured collective is
not something this package relies on).
"""
import time

import numpy as np
import torch
import torch.distributed as dist


def _atom_indices(M, M_atom, device, generator=None):
    """(M, M_atom) int64: column 0 is the row's own index, the others are M_atom - 1 distinct other rows:
    row i takes (i + o_k) mod M for one random set of distinct offsets o_k in [1, M - 1] drawn per call (a sort
    of M - 1 uniforms instead of a top-k over an M x M score matrix, which cost more than the training step)."""
    M_atom = min(M_atom, M)
    own = torch.arange(M, device=device)[:, None]
    if M_atom <= 1:
        return own
    offsets = torch.rand(M - 1, device=device, generator=generator).argsort()[:M_atom - 1] + 1
    return torch.cat((own, (own + offsets[None, :]) % M), 1)


def apt_loss(cde, z_b, x_b, logp_b, atoms):
    """The contrastive APT loss of one batch: z_b (Mb, D) parameters, x_b (Mb, D_x) their simulations, logp_b (Mb)
    their prior log-density, atoms (Mb, M_atom) int64 rows of the batch with column 0 the row's own index."""
    lp = cde.log_prob(z_b[atoms], x_b) - logp_b[atoms]            # (Mb, M_atom)
    return -(lp[:, 0] - torch.logsumexp(lp, dim=1)).mean()


def _dp(group):
    """(world size, rank, group) of the data-parallel run, (1, 0, None) without torch.distributed."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group), group
    return 1, 0, None


def _sync_model(cde, group):
    """Every rank starts from rank 0's context network."""
    for t in list(cde.param_net.parameters()) + list(cde.param_net.buffers()):
        dist.broadcast(t.data, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)


def train_APT(cde, system, x0, M=1000, M_atom=100, R=4, num_iters=1000, lr=1e-3, num_sims=None, verbose=False,
              use_graph=None, group=None):
    """Train `cde` (a ConditionalDensityEstimator) towards p(z | x0) for the simulator `system`.

    :param system: object with sample_prior(N), log_prior(z), simulate(z) (see systems.Mat).
    :param x0: (1, D_x) numpy observation.
    :param M: batch size (contexts per step, over all ranks).  :param M_atom: atoms per context.
    :param R: rounds.  :param num_iters: optimisation steps per round.
    :param num_sims: simulations per round, over all ranks (default 10 * M).
    :param use_graph: replay each round's optimisation step as one HIP graph (graphs.GraphedStep); default: on a
        HIP device, single process.  A round whose step cannot be captured runs eagerly.
    :param group: torch.distributed process group of the data-parallel run (default group when None and
        torch.distributed is initialised); see the module docstring.
    :return: (cde, losses (R * num_iters), zs [R x (n, D)] posterior samples after each round,
              log_probs [R x (n,)], it_time seconds per iteration) -- the five values of the reference's call site.
    """
    from .distributed import allreduce_gradients, shard_bounds

    world, rank, group = _dp(group)
    dev = next(cde.param_net.parameters()).device
    if use_graph is None:
        use_graph = dev.type == "cuda" and world == 1
    if world > 1:
        use_graph = False
        _sync_model(cde, group)
    num_sims = num_sims or 10 * M
    x0_t = torch.as_tensor(np.asarray(x0), dtype=torch.float32, device=dev).reshape(1, -1)
    opt = torch.optim.Adam(cde.param_net.parameters(), lr=lr, capturable=bool(use_graph))
    net_params = [p for p in cde.param_net.parameters()]
    Z = torch.empty((0, system.D), dtype=torch.float32, device=dev)
    X = torch.empty((0, x0_t.shape[1]), dtype=torch.float32, device=dev)
    losses, zs, log_probs = [], [], []
    t_train, n_it = 0.0, 0
    for r in range(R):
        if r == 0:
            z_new = system.sample_prior(num_sims)
        else:
            with torch.no_grad():  # proposals from the current posterior: device-side draw on a HIP device
                z_s, _ = cde.sample(x0_t, N=num_sims) if dev.type == "cuda" else cde(x0_t, N=num_sims, freeze_bn=True)
            z_new = z_s[0].detach().cpu().numpy().astype(np.float64)
        if world > 1:  # this rank's slice of the round (same RNG seed on every rank: disjoint slices of one draw;
            lo, hi = shard_bounds(z_new.shape[0], world, rank)  # different seeds: independent draws -- both are fine)
            z_new = z_new[lo:hi]
        x_new = system.simulate(z_new)
        ok = np.isfinite(x_new).all(1) & np.isfinite(system.log_prior(z_new))
        Z = torch.cat((Z, torch.as_tensor(z_new[ok], dtype=torch.float32, device=dev)))
        X = torch.cat((X, torch.as_tensor(x_new[ok], dtype=torch.float32, device=dev)))
        n = Z.shape[0]
        Mb = max(1, min(M // world, n))
        Z_r, X_r = Z, X  # this round's pairs: fixed addresses for the captured step
        logp_r = system.log_prior(Z_r)  # the prior density of every stored parameter, once per round

        def step():
            idx = torch.randint(0, n, (Mb,), device=dev)
            loss = apt_loss(cde, Z_r[idx], X_r[idx], logp_r[idx], _atom_indices(Mb, M_atom, dev))
            opt.zero_grad(set_to_none=True)
            loss.backward()
            if world > 1:
                allreduce_gradients(net_params, group=group, average=True)
            opt.step()
            return loss.detach()

        torch.cuda.synchronize() if dev.type == "cuda" else None
        t0 = time.perf_counter()
        done = 0
        if use_graph and num_iters > 4:
            try:
                from .graphs import GraphedStep

                gs = GraphedStep(step, warmup=3)
                losses.extend(gs.warmup_outputs)
                done = 3
                for _ in range(num_iters - done):
                    losses.append(gs().clone())
                done = num_iters
            except RuntimeError as e:  # a step this device / build cannot capture: finish the round eagerly
                if verbose:
                    print("round %d: graph capture unavailable (%s), running eagerly" % (r, str(e).splitlines()[0]))
                torch.cuda.synchronize()
        for _ in range(num_iters - done):
            losses.append(step())
        torch.cuda.synchronize() if dev.type == "cuda" else None
        t_train += time.perf_counter() - t0
        n_it += num_iters
        with torch.no_grad():
            z_s, lq = cde(x0_t, N=M)
        zs.append(z_s[0].detach().cpu().numpy())
        log_probs.append(lq[0].detach().cpu().numpy())
        if verbose and rank == 0:
            print("round %d: %d pairs%s, loss %.4f" % (r, n, " on this rank" if world > 1 else "",
                                                       float(torch.stack(losses[-max(1, num_iters // 10):]).mean())))
    losses = torch.stack(losses).cpu().numpy() if losses else np.zeros(0)
    return cde, losses, zs, log_probs, t_train / max(1, n_it)


def train_SNPE(cde, system, x0, M=1000, R=4, num_iters=1000, lr=1e-3, num_sims=None, verbose=False, group=None):
    """Sequential neural posterior estimation WITHOUT atoms: the call `losses = train_SNPE(cnf, gauss, x0, M=M, R=R,
    num_iters=num_iters)` of notebooks/LFI_gauss.ipynb (its traceback shows `train_SNPE(cnf, system, x0, M, R,
    num_iters, verbose)` and a `loss.item()` / `zero_grad()` / `backward()` loop; nothing else of it survives).
    Written from the published algorithm -- SNPE-B (Lueckmann et al., "Flexible statistical inference for mechanistic
    models of neural dynamics", NeurIPS 2017): round r draws parameters from the current posterior estimate q(. | x0)
    (the prior in round 0) and minimises the importance-weighted negative log-likelihood

        L = - sum_i w_i log q(z_i | x_i),   w_i ~ p(z_i) / p~_r(z_i)  (self-normalised over the batch),

    p~_r the proposal z_i was drawn from (its log-density is stored with the pair).  One `cde.log_prob(z (M, 1, D), x)`
    call per step: the one-sample-per-context layout of the fused conditioner + flow kernels.  Data-parallel like
    train_APT.  Returns the losses (R * num_iters), like the reference's call site.  PARITY UNPINNED."""
    from .distributed import allreduce_gradients, shard_bounds

    world, rank, group = _dp(group)
    dev = next(cde.param_net.parameters()).device
    if world > 1:
        _sync_model(cde, group)
    num_sims = num_sims or 10 * M
    x0_t = torch.as_tensor(np.asarray(x0), dtype=torch.float32, device=dev).reshape(1, -1)
    opt = torch.optim.Adam(cde.param_net.parameters(), lr=lr)
    net_params = [p for p in cde.param_net.parameters()]
    Z = torch.empty((0, system.D), dtype=torch.float32, device=dev)
    X = torch.empty((0, x0_t.shape[1]), dtype=torch.float32, device=dev)
    LW = torch.empty((0,), dtype=torch.float32, device=dev)  # log p(z) - log p~(z) of every stored pair
    losses = []
    for r in range(R):
        if r == 0:
            z_new = system.sample_prior(num_sims)
            lq_new = np.asarray(system.log_prior(z_new), dtype=np.float64)
        else:
            with torch.no_grad():
                z_s, lq = cde.sample(x0_t, N=num_sims) if dev.type == "cuda" else cde(x0_t, N=num_sims, freeze_bn=True)
            z_new = z_s[0].detach().cpu().numpy().astype(np.float64)
            lq_new = lq[0].detach().cpu().numpy().astype(np.float64)
        if world > 1:
            lo, hi = shard_bounds(z_new.shape[0], world, rank)
            z_new, lq_new = z_new[lo:hi], lq_new[lo:hi]
        x_new = system.simulate(z_new)
        lp_new = np.asarray(system.log_prior(z_new), dtype=np.float64)
        ok = np.isfinite(x_new).all(1) & np.isfinite(lp_new) & np.isfinite(lq_new)
        Z = torch.cat((Z, torch.as_tensor(z_new[ok], dtype=torch.float32, device=dev)))
        X = torch.cat((X, torch.as_tensor(x_new[ok], dtype=torch.float32, device=dev)))
        LW = torch.cat((LW, torch.as_tensor((lp_new - lq_new)[ok], dtype=torch.float32, device=dev)))
        n = Z.shape[0]
        Mb = max(1, min(M // world, n))
        for _ in range(num_iters):
            idx = torch.randint(0, n, (Mb,), device=dev)
            w = torch.softmax(LW[idx], dim=0)
            loss = -(w * cde.log_prob(Z[idx][:, None, :], X[idx])[:, 0]).sum()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            if world > 1:
                allreduce_gradients(net_params, group=group, average=True)
            opt.step()
            losses.append(loss.detach())
        if verbose and rank == 0:
            print("round %d: %d pairs, loss %.4f" % (r, n, float(torch.stack(losses[-max(1, num_iters // 10):]).mean())))
    return torch.stack(losses).cpu().numpy() if losses else np.zeros(0)
