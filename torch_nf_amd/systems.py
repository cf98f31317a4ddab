"""Simulators for the likelihood-free-inference drivers (stand-in for the reference's `torch_nf.systems`,
which is NOT part of the snapshot: only call sites survive -- scripts/lfi_mat.py:23, 37 and
notebooks/LFI_mat_det_trace.ipynb cells 2, 8).  What those call sites fix: `Mat(d)` has `.D = d (d + 1) / 2`
(the notebook prints samples of shape (1, 100, 6) at d = 3), bounds `.lb` / `.ub` handed to `ToInterval`, and
`simulate(z (N, D)) -> (N, 2)` summary statistics named "det / trace".  Everything else here (the entry
ordering, the bounds, the order of the two statistics, the optional observation noise) is this package's
choice: PARITY UNPINNED.
"""
import numpy as np
import torch


class Mat(object):
    """Symmetric d x d matrix A(z) filled row-wise from its D = d (d + 1) / 2 free entries;
    statistics x = (det A, trace A).  Uniform prior on [lb, ub]^D."""

    def __init__(self, d, bound=2.0, noise=0.0):
        if type(d) is not int or d < 1:
            raise ValueError("Mat dimension d must be a positive int.")
        self.d = d
        self.D = d * (d + 1) // 2
        self.D_x = 2
        self.lb = -bound * np.ones(self.D)
        self.ub = bound * np.ones(self.D)
        self.noise = float(noise)
        self._iu = np.triu_indices(d)

    def sample_prior(self, N):
        return np.random.uniform(self.lb, self.ub, (N, self.D))

    def log_prior(self, z):
        """log density of the uniform prior, -inf outside the box; z (..., D) numpy or torch."""
        vol = float(np.sum(np.log(self.ub - self.lb)))
        if torch.is_tensor(z):
            key = (z.dtype, z.device)  # device copies of the bounds are made once (and keep the step capturable)
            if getattr(self, "_bounds_key", None) != key:
                self._bounds = (torch.as_tensor(self.lb, dtype=z.dtype, device=z.device),
                                torch.as_tensor(self.ub, dtype=z.dtype, device=z.device))
                self._bounds_key = key
            lb, ub = self._bounds
            inside = ((z >= lb) & (z <= ub)).all(-1)
            return torch.where(inside, torch.full(inside.shape, -vol, dtype=z.dtype, device=z.device),
                               torch.full(inside.shape, -float("inf"), dtype=z.dtype, device=z.device))
        inside = np.all((z >= self.lb) & (z <= self.ub), axis=-1)
        return np.where(inside, -vol, -np.inf)

    def matrices(self, z):
        z = np.asarray(z, dtype=np.float64)
        A = np.zeros(z.shape[:-1] + (self.d, self.d))
        A[..., self._iu[0], self._iu[1]] = z
        A[..., self._iu[1], self._iu[0]] = z
        return A

    def simulate(self, z):
        """z (N, D) -> x (N, 2) = (det A, trace A) (+ N(0, noise^2) when noise > 0)."""
        A = self.matrices(z)
        x = np.stack((np.linalg.det(A), np.trace(A, axis1=-2, axis2=-1)), axis=-1)
        if self.noise > 0.0:
            x = x + np.random.normal(0.0, self.noise, x.shape)
        return x
