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
