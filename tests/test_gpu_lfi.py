"""End-to-end smoke of the LFI driver (parity unpinned: no reference outputs exist): APT on the Mat
simulator with an AR flow + ToInterval, a few dozen steps -- finite losses that decrease and a posterior whose
simulated statistics moved towards the observation (after round 0: 0.4-0.6 vs 1.65 for the prior)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_apt_smoke():
    import torch_nf_amd as tnf
    from torch_nf_amd.lfi import _atom_indices, train_APT
    from torch_nf_amd.systems import Mat

    np.random.seed(3)
    torch.manual_seed(3)
    mat = Mat(2, noise=0.05)
    assert mat.D == 3 and mat.simulate(np.array([[1.0, 0.5, 2.0]])).shape == (1, 2)
    np.testing.assert_allclose(Mat(2).simulate(np.array([[1.0, 0.5, 2.0]])), [[1.75, 3.0]])  # det, trace of [[1,.5],[.5,2]]
    atoms = _atom_indices(50, 10, torch.device("cuda"))
    assert atoms.shape == (50, 10) and bool((atoms[:, 0] == torch.arange(50, device="cuda")).all())
    assert all(len(set(r.tolist())) == 10 for r in atoms)
    x0 = np.array([[0.0, 1.0]])
    nf = tnf.NormFlow(mat.D, True, "AR", 1, 2, 15, tnf.ToInterval(mat.D, mat.lb, mat.ub))
    cde = tnf.ConditionalDensityEstimator(nf, 2, [32, 32])
    cde, losses, zs, log_probs, it_time = train_APT(cde, mat, x0, M=256, M_atom=16, R=2, num_iters=150, lr=2e-3)
    assert losses.shape == (300,) and np.isfinite(losses).all() and it_time > 0
    # round 0 (prior proposal): the contrastive loss goes down; later rounds train on a shifted proposal and
    # are only required to stay finite (their loss is not comparable across rounds)
    assert losses[120:150].mean() < losses[:30].mean() - 0.5, (losses[:30].mean(), losses[120:150].mean())
    assert len(zs) == 2 and zs[-1].shape == (256, 3) and log_probs[-1].shape == (256,)
    for z in zs:
        assert (z > mat.lb - 1e-4).all() and (z < mat.ub + 1e-4).all()  # ToInterval keeps the support
    prior_err = np.abs(mat.simulate(mat.sample_prior(4000)) - x0).mean()
    post_err = np.abs(mat.simulate(zs[0]) - x0).mean()
    assert post_err < 0.7 * prior_err, (post_err, prior_err)


def test_train_apt_graph_and_eager_agree_in_outcome():
    """The same short APT run with each round's optimisation step replayed as one HIP graph (the default on a
    HIP device) and eagerly: both must capture nothing stale -- finite, decreasing round-0 losses of the same
    size and posteriors equally close to the observation.  (The random draws differ between the two modes, so
    the loss sequences are compared statistically, not element-wise.)"""
    import torch_nf_amd as tnf
    from torch_nf_amd.lfi import train_APT
    from torch_nf_amd.systems import Mat

    out = {}
    for mode in (True, False):
        np.random.seed(5)
        torch.manual_seed(5)
        mat = Mat(2, noise=0.05)
        x0 = np.array([[0.0, 1.0]])
        nf = tnf.NormFlow(mat.D, True, "AR", 1, 2, 15, tnf.ToInterval(mat.D, mat.lb, mat.ub))
        cde = tnf.ConditionalDensityEstimator(nf, 2, [32, 32])
        msgs = []
        import builtins
        real_print = builtins.print
        builtins.print = lambda *a, **k: msgs.append(" ".join(str(x) for x in a))
        try:
            cde, losses, zs, _, it_time = train_APT(cde, mat, x0, M=256, M_atom=16, R=1, num_iters=200, lr=2e-3,
                                                    use_graph=mode, verbose=True)
        finally:
            builtins.print = real_print
        assert not any("graph capture unavailable" in m for m in msgs), msgs  # the graphed round really replayed
        assert losses.shape == (200,) and np.isfinite(losses).all()
        out[mode] = (losses, np.abs(mat.simulate(zs[0]) - x0).mean(), it_time)
    lg, le = out[True][0], out[False][0]
    assert lg[150:].mean() < lg[:30].mean() - 0.5 and le[150:].mean() < le[:30].mean() - 0.5
    assert abs(lg[150:].mean() - le[150:].mean()) < 0.35, (lg[150:].mean(), le[150:].mean())
    assert abs(out[True][1] - out[False][1]) < 0.25, (out[True][1], out[False][1])


def test_graphed_step_matches_eager_updates():
    """graphs.GraphedStep on a deterministic step (fixed z, Adam): k replays move the parameters exactly as k eager
    steps do, up to the order of the float atomics in the gradient reduction."""
    import torch_nf_amd as tnf
    from torch_nf_amd.graphs import GraphedStep

    D, S, L, U, N = 32, 2, 2, 15, 4096
    rng = np.random.RandomState(0)
    p0 = torch.tensor(rng.normal(0, 0.1, (1, tnf.NormFlow(D, False, "coupling", S, L, U).D_params))).float()
    z = torch.tensor(rng.normal(0, 1, (1, N, D))).float().cuda()
    res = {}
    for mode in ("eager", "graph"):
        nf = tnf.NormFlow(D, False, "coupling", S, L, U)
        nf.params = p0.clone().cuda().requires_grad_()
        opt = torch.optim.Adam([nf.params], lr=1e-3, capturable=True)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = -nf.log_prob(z).mean()
            loss.backward()
            opt.step()
            return loss.detach()

        if mode == "eager":
            ls = [step().item() for _ in range(8)]
        else:
            gs = GraphedStep(step, warmup=3)
            ls = [w.item() for w in gs.warmup_outputs] + [gs().item() for _ in range(5)]
        res[mode] = (np.array(ls), nf.params.detach().cpu())
    np.testing.assert_allclose(res["graph"][0], res["eager"][0], rtol=1e-5)
    torch.testing.assert_close(res["graph"][1], res["eager"][1], rtol=1e-4, atol=1e-6)


def test_cde_sample_device_rng_is_consistent_with_log_prob():
    """ConditionalDensityEstimator.sample (device RNG extension): the returned log-density is the density of the
    returned samples (the reference's own forward -> log_prob consistency check, on the AR + ToInterval stack)."""
    import torch_nf_amd as tnf

    torch.manual_seed(2)
    np.random.seed(2)
    D = 6
    lb, ub = -2.0 * np.ones(D), 2.0 * np.ones(D)
    lb[::2] = -np.inf
    nf = tnf.NormFlow(D, True, "AR", 1, 2, 15, tnf.ToInterval(D, lb, ub))
    cde = tnf.ConditionalDensityEstimator(nf, 3, [32, 32])
    x = torch.randn(5, 3, device="cuda")
    with torch.no_grad():
        z, lq = cde.sample(x, N=4000)
        lp = cde.log_prob(z, x)
    assert z.shape == (5, 4000, D) and z.is_cuda and bool(torch.isfinite(z).all())
    assert bool((z[..., 1::2] > -2.0 - 1e-4).all()) and bool((z[..., 1::2] < 2.0 + 1e-4).all())  # the bounded features
    torch.testing.assert_close(lp.double(), lq.double(), rtol=1e-4, atol=2e-3)
