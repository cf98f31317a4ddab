"""Bijectors of the coupling-flow hot path, with the reference's Python interface.

Drop-in for the hot-path classes of the reference's torch_nf/bijectors.py:
  Bijector  (bijectors.py:7-71)    protocol: .name, .D, count_num_params(),
                                   __call__ / forward_and_log_det / inverse_and_log_det
  RealNVP   (bijectors.py:74-262)  coupling layer, twin t/s MLP from packed params
  Affine    (bijectors.py:265-318) per-dimension exp(alpha)*z + shift
  BatchNorm (bijectors.py:321-426) normalisation with log-det and cached statistics
Same constructor arguments, validation (exact-type checks -> TypeError with the
reference's message, range checks -> ValueError, clamps with a printed warning),
return conventions ((z_out, log_det) with log_det (M,N) / (M,1) / 0-dim) and
packed-parameter layout.  The arithmetic runs in the HIP kernels of
libtnf_hip.so (include/tnf.h); nothing here computes on the CPU.
"""
import numpy as np
import torch

from . import ops
from .error_formatters import format_type_err_msg


class _Checked:
    """Attribute with the reference's validation protocol: the value must be of
    exactly `typ` (TypeError otherwise), then `rule(value)` either returns the
    value to store (possibly clamped, after printing a warning) or raises ValueError."""

    def __init__(self, label, typ, rule=None):
        self.label, self.typ, self.rule = label, typ, rule

    def __set_name__(self, owner, attr):
        self.slot = "_checked_" + attr

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        return getattr(obj, self.slot)

    def __set__(self, obj, val):
        if type(val) is not self.typ:
            raise TypeError(format_type_err_msg(obj, self.label, val, self.typ))
        setattr(obj, self.slot, val if self.rule is None else self.rule(val))


def _positive_dim(val):
    if val < 1:
        raise ValueError("Bijector dimensionality must be positive.")
    return val


class Bijector(object):
    """Base class of the bijectors composed into normalizing flows (bijectors.py:7-71).

    :param D: Dimensionality of the bijection.
    :type D: int
    """

    D = _Checked("D", int, _positive_dim)

    def __init__(self, D):
        super().__init__()
        self.D = D

    def __call__(self, z, params):
        return self.forward_and_log_det(z, params)

    def forward_and_log_det(self, z, params):
        """z (M, N, D), params (M, >=|theta|) -> (z', log|det J|)."""
        raise NotImplementedError()

    def inverse_and_log_det(self, z, params):
        """Inverse map; returns the FORWARD log-det like the reference does."""
        raise NotImplementedError()

    def count_num_params(self):
        return 0


def _clamp_layers(val):
    if val < 1:
        raise ValueError("RealNVP.num_layers must be positive.")
    if val > 5:
        print("Warning: RealNVP.num_layers set to maximum of 5 (received %d)." % val)
        return 5
    return val


def _clamp_units(val):
    if val < 15:
        print("Warning: num_units set to minimum of 15 (received %d)." % val)
        return 15
    if val > 1000:
        print("Warning: num_units set to maximum of 1,000 (received %d)." % val)
        return 1000
    return val


class RealNVP(Bijector):
    """RealNVP coupling bijector (bijectors.py:74-262).

    Two independent fully connected nets (shift t, log-scale s), both fed the
    conditioner half z1, transform the other half: z2' = t + z2*exp(s).  With
    transform_upper=True z1 = z[..., :D//2] and z2 = z[..., D//2:], otherwise the roles
    swap.  One call = one fused HIP kernel (tnf_coupling): MLP, scale-shift, and the
    wavefront-reduced log-det.

    :param D: Dimensionality of the bijection.
    :param num_layers: Number of hidden layers of each net (1..5, clamped).
    :param num_units: Hidden width (15..1000, clamped).
    :param transform_upper: transform the upper half conditioned on the lower half.
    """

    num_layers = _Checked("num_layers", int, _clamp_layers)
    num_units = _Checked("num_units", int, _clamp_units)
    transform_upper = _Checked("transform_upper", bool)

    def __init__(self, D, num_layers, num_units, transform_upper=True):
        super().__init__(D)
        self.name = "RealNVP"
        self.num_layers = num_layers
        self.num_units = num_units
        self.transform_upper = transform_upper

    def _run(self, z, params, inverse):
        return ops.coupling(z, params, self.D, self.num_layers, self.num_units,
                            self.transform_upper, inverse)

    def forward_and_log_det(self, z, params):
        """bijectors.py:145-181: returns (cat(z1, t + z2*exp(s)), sum(s))."""
        return self._run(z, params, False)

    def inverse_and_log_det(self, z, params):
        """bijectors.py:183-206: returns (cat(z1, (z2 - t)/exp(s)), sum(s))."""
        return self._run(z, params, True)

    def count_num_params(self):
        """bijectors.py:244-262 (the C ABI exposes the same count as tnf_coupling_num_params)."""
        h = self.D // 2
        d_in = h + (self.D % 2) * (not self.transform_upper)
        d_out = h + (self.D % 2) * self.transform_upper
        U, L = self.num_units, self.num_layers
        return int(2 * (d_in * U + d_out * U + d_out + U + (L - 1) * (U + 1) * U))


class Affine(Bijector):
    """Per-dimension scale and shift (bijectors.py:265-318); params = [alpha | shift]."""

    def __init__(self, D):
        super().__init__(D)
        self.name = "Affine"

    def forward_and_log_det(self, z, params):
        """exp(alpha)*z + shift, log_det = sum(alpha) of shape (M, 1)."""
        return ops.affine(z, params, self.D, False)

    def inverse_and_log_det(self, z, params):
        """(z - shift)/exp(alpha), same (forward) log_det."""
        return ops.affine(z, params, self.D, True)

    def count_num_params(self):
        return 2 * self.D


def _clamp_momentum(val):
    if val < 0.0:
        raise ValueError("BatchNorm.momentum cannot be negative.")
    if val > 1.0:
        print("Warning: BathNorm.momentum  set to maximum of 1.0 (received %.2E)." % val)
        return 1.0
    return val


def _nonneg_eps(val):
    if val < 0.0:
        raise ValueError("BatchNorm.eps cannot be negative.")
    return val


class BatchNorm(Bijector):
    """Batch-norm bijector that propagates its log-det and remembers the statistics
    of its last batch-mode forward (bijectors.py:321-426).

    forward(use_last=False): statistics over all M*N rows (HIP reduction in float64),
        z_norm = (z - mu)/sqrt(var + eps); caches mean and alpha -- WITH their graph when autograd is
        recording, like the reference (bijectors.py:414-415: no detach), so that `nf(N)` followed by
        `nf.log_prob(z)` or the inverse in one graph back-propagates through the batch moments.
    forward(use_last=True):  (z - last_mean)/last_alpha.
    inverse:                 z*last_alpha + last_mean.
    log_det = -sum(log(alpha)) (0-dim) in every case.

    `momentum` is accepted and validated for interface parity; like in the reference
    (which builds an nn.BatchNorm1d but never evaluates it in eval mode) running
    averages never influence any output.
    """

    momentum = _Checked("momentum", float, _clamp_momentum)
    eps = _Checked("eps", float, _nonneg_eps)

    def __init__(self, D, momentum=0.1, eps=1e-5):
        super().__init__(D)
        self.name = "BatchNorm"
        self.momentum = momentum
        self.eps = eps
        self._last_mean = torch.tensor(np.zeros(D)).float()
        self._last_alpha = torch.tensor(np.ones(D)).float()
        self._version = 0  # bumped whenever the cached statistics change (NormFlow caches the stacked copy)
        # sample-sharded batches (one process per GPU): a callable that sums a small tensor in place over the ranks
        # holding the other rows (distributed.moment_reducer(group)); batch-mode forward and its backward then use the
        # statistics of the WHOLE batch.  None: this process holds the whole batch.
        self.stats_reduce = None

    def get_last_mean(self):
        return self._last_mean

    def get_last_alpha(self):
        return self._last_alpha

    def set_last_stats(self, mean, alpha):
        """Install cached statistics (e.g. restored from a checkpoint or all-reduced)."""
        self._last_mean = mean.detach().float()
        self._last_alpha = alpha.detach().float()
        self._version += 1

    def __call__(self, z, use_last=False):
        return self.forward_and_log_det(z, use_last=use_last)

    def _stats_for(self, z):
        """The cached statistics where the kernels will read them: one copy per (version, device) instead of a
        host-to-device transfer per call (which also cannot be captured into a HIP graph)."""
        if not torch.cuda.is_available():
            return self._last_mean, self._last_alpha  # ops.bn_apply raises the "needs a HIP device" error
        dev = torch.device("cuda", torch.cuda.current_device())
        if self._last_mean.device == dev:
            return self._last_mean, self._last_alpha
        if torch.is_grad_enabled() and (self._last_mean.requires_grad or self._last_alpha.requires_grad):
            # host-resident statistics that still carry the graph of their batch-mode forward (a CPU caller, like every
            # caller of the reference): a differentiable copy, never the detached cache (bijectors.py:414-415)
            return self._last_mean.float().to(dev), self._last_alpha.float().to(dev)
        key = (self._version, dev)
        if getattr(self, "_dev_key", None) != key:
            self._dev_stats = (self._last_mean.detach().float().to(dev), self._last_alpha.detach().float().to(dev))
            self._dev_key = key
        return self._dev_stats

    def forward_and_log_det(self, z, use_last=False):
        if use_last:
            return ops.bn_apply(z, *self._stats_for(z), False)
        z_norm, log_det, mean, alpha = ops.bn_batch_forward(z, self.eps, reduce=self.stats_reduce)
        self._last_mean, self._last_alpha = mean, alpha
        self._version += 1
        return z_norm, log_det

    def inverse_and_log_det(self, z):
        return ops.bn_apply(z, *self._stats_for(z), True)


def _clamp_maf_layers(val):
    if val < 1:
        raise ValueError("MAF.num_layers must be positive.")
    if val > 5:
        print("Warning: MAF.num_layers set to maximum of 5 (received %d)." % val)
        return 5
    return val


def _clamp_maf_units(val):
    if val < 5:
        print("Warning: num_units set to minimum of 15 (received %d)." % val)  # sic: the reference stores 5
        return 5
    if val > 1000:
        print("Warning: num_units set to maximum of 1,000 (received %d)." % val)
        return 1000
    return val


class MAF(Bijector):
    """Masked autoregressive flow bijector (bijectors.py:597-806), the bijector of NormFlow's default
    arch_type "AR".

    Twin masked MLPs (f_mu, f_alpha), no biases, tanh on the hidden layers.  The degree vectors are
    drawn at construction from the host numpy RNG exactly like the reference (np.random.randint per
    hidden layer, bijectors.py:673), so np.random.seed reproduces the reference's masks; `ms` / `Ms`
    are kept as attributes like in the reference.  inverse_and_log_det is one kernel pass,
    forward_and_log_det runs the reference's D-1 sequential passes inside one kernel.
    """

    num_layers = _Checked("num_layers", int, _clamp_maf_layers)
    num_units = _Checked("num_units", int, _clamp_maf_units)
    fwd_fac = _Checked("fwd_fac", bool)

    def __init__(self, D, num_layers, num_units, fwd_fac=True):
        super().__init__(D)
        self.name = "MAF"
        self.num_layers = num_layers
        self.num_units = num_units
        self.fwd_fac = fwd_fac
        self._get_masks()

    def _get_masks(self):
        """bijectors.py:663-696 (the odd arange(D, -1, -1) of fwd_fac=False included)."""
        D, K = self.D, self.num_units
        ms, Ms = [], []
        k_prev = D
        m_prev = np.arange(1, D + 1) if self.fwd_fac else np.arange(D, -1, -1)
        for _ in range(self.num_layers):
            m = np.random.randint(1, D, (K,))
            Ms.append((m_prev[:k_prev, None] <= m[None, :]).astype(np.float32))
            ms.append(m)
            k_prev, m_prev = K, m
        m = np.arange(1, D + 1) if self.fwd_fac else np.arange(D, -1, -1)
        Ms.append((m_prev[:k_prev, None] < m[None, :D]).astype(np.float32))
        ms.append(m)
        self.set_masks(ms, Ms)
        return None

    def set_masks(self, ms, Ms=None):
        """Install degree vectors (e.g. restored from a checkpoint); the masks follow from them."""
        if Ms is None:
            D = self.D
            Ms, k_prev = [], D
            m_prev = np.arange(1, D + 1) if self.fwd_fac else np.arange(D, -1, -1)
            for m in ms[:-1]:
                Ms.append((np.asarray(m_prev)[:k_prev, None] <= np.asarray(m)[None, :]).astype(np.float32))
                k_prev, m_prev = len(m), m
            Ms.append((np.asarray(m_prev)[:k_prev, None] < np.asarray(ms[-1])[None, :D]).astype(np.float32))
        self.ms = [np.asarray(m) for m in ms]
        self.Ms = [torch.tensor(M[None, :, :]).float() for M in Ms]
        self._masks_flat = torch.cat([torch.tensor(M).float().reshape(-1) for M in Ms])

    def _masks_for(self, dtype):
        """The concatenated layer masks on the compute device in `dtype` (uploaded once, then cached)."""
        from . import _lib

        key = (_lib.require_device(), dtype)
        cache = self.__dict__.setdefault("_masks_dev", {})
        if cache.get("src") is not self._masks_flat:  # set_masks() installed new masks
            cache.clear()
            cache["src"] = self._masks_flat
        if key not in cache:
            cache[key] = self._masks_flat.to(device=key[0], dtype=dtype).contiguous()
        return cache[key]

    def forward_and_log_det(self, z, params):
        return ops.maf(z, params, self._masks_for(z.dtype), self.D, self.num_layers, self.num_units, False)

    def inverse_and_log_det(self, z, params):
        return ops.maf(z, params, self._masks_for(z.dtype), self.D, self.num_layers, self.num_units, True)

    def count_num_params(self):
        return int(2 * (2 * self.D * self.num_units + (self.num_layers - 1) * (self.num_units ** 2)))


class ToInterval(Bijector):
    """Maps each feature to an interval (bijectors.py:429-553): tanh where both bounds are finite,
    +-softplus where one is, identity where neither.  Parameter-free; called as `layer(z)`.

    :param D: Dimensionality of the bijection.
    :type D: int
    :param lb: Lower bound of interval.
    :type lb: np.ndarray or list
    :param ub: Upper bound of interval.
    :type ub: np.ndarray or list
    """

    def __init__(self, D, lb, ub):
        super().__init__(D)
        self.name = "ToInterval"
        self.lb = lb
        self.ub = ub
        self._eps = 1e-12

        if self.lb.shape[0] != self.ub.shape[0]:
            raise ValueError("Lower and upper bounds must be same length.")
        for lb_i, ub_i in zip(self.lb, self.ub):
            if lb_i > ub_i:
                raise ValueError("Lower bound %.2E > upper bound %.2E." % (lb_i, ub_i))

        # per-feature constants, float64 on the host then rounded to float32 like the reference (:454-480)
        rows = np.zeros((6, self.D))
        rows[2] = rows[4] = 1.0
        for i in range(self.D):
            lb_i, ub_i = self.lb[i], self.ub[i]
            has_lb, has_ub = not np.isneginf(lb_i), not np.isposinf(ub_i)
            if has_lb and has_ub:
                rows[0, i], rows[2, i], rows[3, i] = 1, (ub_i - lb_i) / 2.0, (ub_i + lb_i) / 2.0
            elif has_lb:
                rows[1, i], rows[4, i], rows[5, i] = 1, 1.0, lb_i
            elif has_ub:
                rows[1, i], rows[4, i], rows[5, i] = 1, -1.0, ub_i
        c = torch.tensor(rows).float()
        (self.tanh_flg, self.softplus_flg, self.tanh_m, self.tanh_c, self.softplus_m,
         self.softplus_c) = (c[i][None, None, :] for i in range(6))
        # the kernels' constant block: the six rows + log(tanh_m) in float32 (torch.log(self.tanh_m), :515)
        self._consts = torch.cat((c, torch.log(c[2:3])), 0).contiguous()
        self._consts_dev = {}

    def _check_bound(self, label, val):
        if type(val) not in [list, np.ndarray]:
            raise TypeError(format_type_err_msg(self, label, val, np.ndarray))
        return np.array(val) if type(val) is list else val

    @property
    def lb(self):
        return self._lb

    @lb.setter
    def lb(self, val):
        self._lb = self._check_bound("lb", val)

    @property
    def ub(self):
        return self._ub

    @ub.setter
    def ub(self, val):
        self._ub = self._check_bound("ub", val)

    def _device_consts(self):
        from . import _lib

        dev = _lib.require_device()
        if dev not in self._consts_dev:
            self._consts_dev[dev] = self._consts.to(dev)
        return self._consts_dev[dev]

    def __call__(self, z):
        return self.forward_and_log_det(z)

    def forward_and_log_det(self, z):
        """bijectors.py:509-527 -> tnf_to_interval(inverse=0)."""
        return ops.to_interval(z, self._device_consts(), False)

    def inverse_and_log_det(self, z):
        """bijectors.py:529-553 -> tnf_to_interval(inverse=1); like the reference, the log-det
        returned is the forward one at the recovered point."""
        return ops.to_interval(z, self._device_consts(), True)


class ToSimplex(Bijector):
    """Maps (M, N, D_in) to the simplex in D_in + 1 dimensions (bijectors.py:560-594).  Forward only:
    the reference defines no inverse, so `inverse_and_log_det(z)` fails exactly as it does there
    (the base class method wants `params`).

    :param D: Dimensionality of the bijection.
    :type D: int
    """

    def __init__(self, D):
        super().__init__(D)
        self.name = "ToSimplex"

    def __call__(self, z):
        return self.forward_and_log_det(z)

    def forward_and_log_det(self, z):
        """bijectors.py:574-591 -> tnf_to_simplex."""
        return ops.to_simplex(z, self.D)

    def count_num_params(self):
        return 0
