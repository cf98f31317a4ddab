"""NormFlow with the reference's interface, executing on MI355X.

Drop-in for DensityEstimator / NormFlow of the reference's
torch_nf/density_estimator.py (:11-55, :240-421): same constructor, validation,
bijector stack, flat-parameter slicing, `__call__(N, params, freeze_bn)`,
`forward`, `inverse_and_log_det`, `log_prob`, `count_num_params`, `D_params`,
`params`, `bijectors`.

Execution:
  * arch_type="coupling", float32, no autograd, a shape with an MFMA kernel
    (tnf_has_fast_path): ONE call of tnf_flow_log_prob_f32 / tnf_flow_forward_f32
    (whole flow in one kernel, or one kernel per coupling layer).
  * anything else: the reference's bijector loop, each bijector one HIP kernel.
arch_type "AR" (MAF) composes per bijector.  A support layer (ToInterval / ToSimplex) is one
extra elementwise kernel after the stack in `forward` and before it in `log_prob`.
"""
import numpy as np
import torch

from . import _lib, ops
from .bijectors import MAF, Affine, BatchNorm, Bijector, RealNVP, _Checked


def _min_two(val):
    if val < 2:
        raise ValueError("DensityEstimator D %d must be greater than 1." % val)
    return val


class DensityEstimator(object):
    """Abstract base (density_estimator.py:11-55)."""

    D = _Checked("D", int, _min_two)
    conditioner = _Checked("conditioner", bool)

    def __init__(self, D, conditioner=False):
        super().__init__()
        self.D = D
        self.conditioner = conditioner

    def __call__(self, N=100, params=None):
        if not self.conditioner:
            return self.forward(self.params, N)
        return self.forward(params, N)

    def forward(self, params, N=100, freeze_bn=False):
        raise NotImplementedError()

    def log_prob(self, z, params=None):
        raise NotImplementedError()

    def count_num_params(self):
        raise NotImplementedError()

    def _param_init(self):
        raise NotImplementedError()


def _arch(val):
    if val not in ("coupling", "AR", "affine"):
        raise ValueError('NormalizingFlow arch_type must be "coupling", "AR", or "affine".')
    return val


def _stages(val):
    if val < 1:
        raise ValueError("NormalizingFlow num_stages %d must be greater than 0." % val)
    return val


def _layers(val):
    if val < 1:
        raise ValueError("NormalizingFlow num_layers arg %d must be greater than 0." % val)
    return val


def _units(val):
    if val < 1:
        raise ValueError("NormalizingFlow num_units %d must be greater than 0." % val)
    if val < 15:
        print("Warning: NormFlow.num_layers set to minimum of 15 (received %d)." % val)
        return 15
    return val


class NormFlow(DensityEstimator):
    """Normalizing flow q(z) = N(omega; 0, I) pushed through a bijector stack
    (density_estimator.py:240-421).

    arch_type="coupling": num_stages x [RealNVP(upper), BatchNorm, RealNVP(lower),
    BatchNorm, Affine] (:260-270).  All bijector parameters live in one flat row
    `params` (1, D_params) -- or are supplied per context as (M, D_params) when
    conditioner=True -- and are consumed front-to-back by `forward` (:379-384) and
    back-to-front by `inverse_and_log_det` (:399-402).

    Extra (not in the reference): `device` -- where the flow's own `params` live
    (default: the current HIP device when there is one).
    """

    arch_type = _Checked("arch_type", str, _arch)
    num_stages = _Checked("num_stages", int, _stages)
    num_layers = _Checked("num_layers", int, _layers)
    num_units = _Checked("num_units", int, _units)

    def __init__(self, D, conditioner=False, arch_type="AR", num_stages=1, num_layers=2,
                 num_units=15, support_layer=None, device=None):
        super().__init__(D, conditioner)
        self.arch_type = arch_type
        self.num_stages = num_stages
        self.num_layers = num_layers
        self.num_units = num_units
        self.support_layer = support_layer
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() \
                else torch.device("cpu")
        self.device = torch.device(device)
        self.fusion = _lib.FUSE_AUTO
        # sample-sharded sampling with fresh statistics (one process per GPU): a callable that sums a BatchNorm layer's
        # [sum | sum of squares | count] moments over the ranks sharing the batch (distributed.moment_reducer(group))
        self.batch_stats_reduce = None

        self.bijectors = []
        if self.arch_type == "coupling":
            for _ in range(self.num_stages):
                self.bijectors.append(RealNVP(D, self.num_layers, self.num_units, transform_upper=True))
                self.bijectors.append(BatchNorm(D))
                self.bijectors.append(RealNVP(D, self.num_layers, self.num_units, transform_upper=False))
                self.bijectors.append(BatchNorm(D))
                self.bijectors.append(Affine(D))
        elif self.arch_type == "AR":  # density_estimator.py:271-274
            self.bijectors.append(MAF(D, self.num_layers, self.num_units, fwd_fac=True))
            self.bijectors.append(BatchNorm(D))
            self.bijectors.append(Affine(D))
        else:
            self.bijectors.append(Affine(D))

        self._n_core = len(self.bijectors)  # the parameterised stack; a support layer comes after it
        if support_layer is not None:
            if issubclass(type(support_layer), Bijector):
                self.bijectors.append(support_layer)  # density_estimator.py:278-282
            else:
                raise TypeError("Support layer not Bijector.")

        self.count_num_params()
        if not self.conditioner:
            self._param_init()

    # -- parameters ---------------------------------------------------------
    def count_num_params(self):
        """density_estimator.py:418-421."""
        self.D_params = 0
        for bijector in self.bijectors:
            self.D_params += bijector.count_num_params()

    def _param_init(self):
        """xavier_normal_ on a (1, D_params) row (density_estimator.py:352-356); drawn on the
        host so torch.manual_seed reproduces the reference's initialisation, then moved."""
        init = torch.nn.init.xavier_normal_(torch.zeros(1, self.D_params))
        self.params = init.to(self.device).requires_grad_(True)
        return None

    # -- helpers ------------------------------------------------------------
    def _bn_layers(self):
        return [b for b in self.bijectors if b.name == "BatchNorm"]

    def _bn_stats(self, dev):
        """(2S, D) stacks of the cached BatchNorm statistics on `dev`; rebuilt only when a
        BatchNorm layer's statistics changed (keeps the per-call host work off the hot path)."""
        bns = self._bn_layers()
        key = (dev, tuple(b._version for b in bns))
        cached = self.__dict__.get("_bn_cache")
        if cached is None or cached[0] != key:
            mean = torch.stack([b.get_last_mean().detach().float().to(dev) for b in bns])
            alpha = torch.stack([b.get_last_alpha().detach().float().to(dev) for b in bns])
            cached = (key, mean, alpha)
            self.__dict__["_bn_cache"] = cached
        return cached[1], cached[2]

    def _ar_fused_ok(self, z, params):
        """[MAF, BatchNorm, Affine] as one kernel: float32, no autograd, shape covered by the MFMA MAF kernel."""
        if self.arch_type != "AR" or z.dtype != torch.float32 or params.dtype != torch.float32:
            return False
        if torch.is_grad_enabled() and (z.requires_grad or params.requires_grad or self._stats_in_graph()):
            return False
        return z.dim() == 3 and ops.ar_flow_supported(self.D, self.num_layers, self.num_units)

    def _ar_args(self):
        maf, bn = self.bijectors[0], self.bijectors[1]
        mean, alpha = bn._stats_for(None)  # device copies, made once per version of the statistics
        return (maf._masks_for(torch.float32), mean.detach(), alpha.detach(), self.D, self.num_layers, self.num_units)

    def _whole_flow(self):
        """Does the fused coupling path run as ONE kernel (the only one with a fused support stage)?"""
        return ops.resolve_fusion(self.D, self.num_stages, self.num_layers, self.num_units, self.fusion) == _lib.FUSE_FLOW

    def _fused_support(self):
        """The (7, D) device constants of a ToInterval support layer that the one-kernel paths evaluate in
        their load / store stage, or None (no support layer).  Other support layers are not fused."""
        if self._n_core == len(self.bijectors):
            return None
        sup = self.bijectors[-1]
        return sup._device_consts() if sup.name == "ToInterval" else False

    def _batch_chain_ok(self, z, params):
        """Shapes of the one-call chains with fresh batch statistics (the narrow MFMA layer kernels)."""
        return (getattr(self, "fused_batch_forward", True) and z.dim() == 3 and z.size(0) == params.size(0)
                and ops.flow_train_supported(z.size(0), params.size(0), 32, self.D, self.num_stages, self.num_layers,
                                             self.num_units))

    def _stats_in_graph(self):
        """Do the cached BatchNorm statistics still carry the graph of the batch-mode forward that produced them
        (per-bijector forward under autograd -- the reference's behaviour, bijectors.py:414-415)?  Then every later
        use of them in the same graph must differentiate through them: only the per-bijector composition does."""
        return torch.is_grad_enabled() and any(b.get_last_mean().requires_grad or b.get_last_alpha().requires_grad
                                               for b in self._bn_layers())

    def _fused_ok(self, z, params):
        """One-call fused path: coupling stack, float32, no autograd, MFMA-covered shape."""
        if self.arch_type != "coupling" or self._stats_in_graph():
            return False
        if z.dtype != torch.float32 or params.dtype != torch.float32:
            return False
        if torch.is_grad_enabled() and (z.requires_grad or params.requires_grad):
            return False
        # per-context weights with very few samples each (the SNPE layout, N = 1): the prepared
        # operand images (~100 KB per context) would outweigh the samples; compose per bijector
        if params.size(0) > 1 and z.size(1) < 32:
            return False
        return ops.has_fast_path(self.D, self.num_layers, self.num_units)

    # -- sampling -----------------------------------------------------------
    def __call__(self, N=100, params=None, freeze_bn=False):
        if not self.conditioner:
            return self.forward(self.params, N, freeze_bn=freeze_bn)
        return self.forward(params, N, freeze_bn=freeze_bn)

    def forward(self, params, N=100, freeze_bn=False):
        """Draw N samples per parameter row and their log-density
        (density_estimator.py:364-388).  The base draw is host numpy float64 exactly like
        the reference (so np.random.seed reproduces it); returns z float32 and log_q
        float64 on the parameters' device."""
        M = params.size(0)
        omega = np.random.normal(0.0, 1.0, (M, N, self.D))
        return self._forward_from(omega, params, freeze_bn)

    def sample(self, N=100, params=None, freeze_bn=True, generator=None):
        """Extension (not in the reference): like `forward`, but the base draw comes from the
        device RNG (`torch.randn` on the flow's device, optional `generator`), so no host RNG,
        no float64 staging and no PCIe copy.  Not reproducible against np.random.seed."""
        if not self.conditioner:
            params = self.params
        dev = _lib.require_device()
        omega = torch.randn((params.size(0), N, self.D), device=dev, dtype=torch.float32, generator=generator)
        return self._forward_from(omega, params, freeze_bn)

    def _forward_from(self, omega, params, freeze_bn=False):
        """`forward` with the base draw injected: numpy float64 (M,N,D) like the reference's host
        draw, or a torch tensor (e.g. a device-side torch.randn draw, which skips the host RNG and
        the 8 B/value PCIe copy)."""
        home = params.device
        dev = _lib.require_device()
        p_dev = params if params.device == dev else params.to(dev)
        log_q = None
        if torch.is_tensor(omega) and omega.dtype == torch.float32:
            z = omega.detach().to(dev)  # device-side draw: no float64 round trip
            # the base density of a float32 draw can come out of the whole-flow sampling kernel itself (below);
            # every other route evaluates it here
            if not (freeze_bn and self._fused_ok(z, p_dev) and self._whole_flow()):
                log_q = ops.base_log_density_f64(z)
        else:
            if torch.is_tensor(omega):
                omega64 = omega.detach().to(device=dev, dtype=torch.float64)
            else:
                omega64 = torch.as_tensor(np.ascontiguousarray(omega), dtype=torch.float64).to(dev)
            z = omega64.float()
            log_q = ops.base_log_density_f64(omega64)

        sup = self._fused_support()
        support_done = False
        # sample-sharded batch statistics (self.batch_stats_reduce): the stepwise no-autograd chain exchanges the moments
        # between its launches; every other case -- autograd, other shapes, arch_type "AR" -- runs the per-bijector
        # composition with BatchNorm layers that exchange their moments forward and their gradient sums backward
        # (ops._BnBatchShardedFn).  The one-node training chain computes local moments inside one C call: not offered.
        for b in self._bn_layers():
            b.stats_reduce = None if freeze_bn else self.batch_stats_reduce
        if freeze_bn and self._ar_fused_ok(z, p_dev) and sup is not False:
            z, sld = ops.ar_flow_forward_raw(z, p_dev, *self._ar_args(), interval_consts=sup)
            log_q = log_q - sld
            support_done = True
        elif freeze_bn and self._fused_ok(z, p_dev):
            mean, alpha = self._bn_stats(dev)
            fuse_sup = (sup is not None and sup is not False and self._whole_flow())
            if log_q is None:  # float32 device draw on the whole-flow kernel: it writes log N(omega) - sld itself
                omega_dev = z
                z, sld, log_q = ops.flow_forward_raw(z, p_dev, mean, alpha, self.D, self.num_stages, self.num_layers,
                                                     self.num_units, self.fusion,
                                                     interval_consts=sup if fuse_sup else None, want_log_q=True)
                if log_q is None:
                    log_q = ops.base_log_density_f64(omega_dev) - sld
            else:
                z, sld = ops.flow_forward_raw(z, p_dev, mean, alpha, self.D, self.num_stages,
                                              self.num_layers, self.num_units, self.fusion,
                                              interval_consts=sup if fuse_sup else None)
                log_q = log_q - sld
            support_done = fuse_sup
        elif (not freeze_bn and self._fused_ok(z, p_dev) and self._batch_chain_ok(z, p_dev)
              and (z.size(0) * z.size(1) > 1 or self.batch_stats_reduce is not None)):
            # fresh batch statistics, no autograd: one call for the whole stack; every BatchNorm layer ends up with
            # the statistics its own forward(use_last=False) would have cached
            bns = self._bn_layers()
            z, sld, means, alphas = ops.flow_forward_batch_raw(z, p_dev, self.D, self.num_stages, self.num_layers,
                                                               self.num_units, bns[0].eps,
                                                               reduce_moments=self.batch_stats_reduce)
            for i, b in enumerate(bns):
                b.set_last_stats(means[i], alphas[i])
            log_q = log_q - sld
        elif (not freeze_bn and self.arch_type == "coupling" and z.dtype == torch.float32 and self.batch_stats_reduce is None
              and p_dev.dtype == torch.float32 and self._batch_chain_ok(z, p_dev) and z.size(0) * z.size(1) >= 32):
            # fresh batch statistics under autograd: one node for the whole stack (gradients through the batch
            # moments included)
            bns = self._bn_layers()
            z, sld, means, alphas = ops.flow_forward_train(z, p_dev, self.D, self.num_stages, self.num_layers,
                                                           self.num_units, bns[0].eps)
            for i, b in enumerate(bns):
                b.set_last_stats(means[i], alphas[i])
            log_q = log_q - sld
        else:
            idx = 0
            for bijector in self.bijectors[:self._n_core]:
                if bijector.name == "BatchNorm":
                    z, log_det = bijector(z, use_last=freeze_bn)
                else:
                    n = bijector.count_num_params()
                    z, log_det = bijector(z, p_dev[:, idx:idx + n])
                    idx += n
                log_q = log_q - log_det
        for bijector in ([] if support_done else self.bijectors[self._n_core:]):  # parameter-free support layer (:385-386)
            z, log_det = bijector(z)
            log_q = log_q - log_det
        if home != dev:
            z, log_q = z.to(home), log_q.to(home)
        return z, log_q

    # -- density ------------------------------------------------------------
    def inverse_and_log_det(self, z, params):
        """Map z back to the base space, accumulating the forward log-dets
        (density_estimator.py:390-406).  Returns (z0, sum_log_det float32 (M,N))."""
        if self._n_core == len(self.bijectors) and self._ar_fused_ok(z, params):
            _, z0, sld = ops.ar_flow_log_prob_raw(z, params, *self._ar_args(), want_lp=False, want_z0=True, want_sld=True)
            return z0, sld
        if self._n_core == len(self.bijectors) and self._fused_ok(z, params):
            dev = _lib.require_device()
            mean, alpha = self._bn_stats(dev)
            _, z0, sld = ops.flow_log_prob_raw(z, params, mean, alpha, self.D, self.num_stages,
                                               self.num_layers, self.num_units, self.fusion,
                                               want_z0=True, want_sld=True, want_lp=False)
            return z0, sld
        return self._core_inverse(z, params, self.bijectors)

    def _core_inverse(self, z, params, bijectors=None):
        """The reference's loop, one HIP kernel per bijector (the whole stack, or only its
        parameterised core when `bijectors` is None)."""
        if bijectors is None:
            bijectors = self.bijectors[:self._n_core]
        home = z.device
        dev = _lib.require_device()
        if home != dev:
            z = z.to(dev)
        if params.device != dev:
            params = params.to(dev)
        M = max(z.size(0), params.size(0))
        idx = self.D_params
        sum_log_det = torch.zeros((M, z.size(1)), device=dev)
        for bijector in reversed(bijectors):
            n = bijector.count_num_params()
            if n > 0:
                z, log_det = bijector.inverse_and_log_det(z, params[:, idx - n:idx])
                idx -= n
            else:
                z, log_det = bijector.inverse_and_log_det(z)
            sum_log_det = sum_log_det + log_det
        sum_log_det = sum_log_det.float()  # the reference accumulates into a float32 buffer (:394)
        if home != dev:
            z, sum_log_det = z.to(home), sum_log_det.to(home)
        return z, sum_log_det

    def log_prob(self, z, params=None):
        """log q(z) (density_estimator.py:408-416)."""
        if not self.conditioner:
            params = self.params
        if self._n_core < len(self.bijectors):
            sup = self._fused_support()
            if sup is not False and self._ar_fused_ok(z, params):
                # ToInterval^-1 in the load stage of the one-kernel AR path
                return ops.ar_flow_log_prob_raw(z, params, *self._ar_args(), interval_consts=sup)[0]
            if sup is not False and self._ar_train_ok(z, params):
                masks, mean, alpha, D, L, U = self._ar_args()
                return ops.ar_flow_log_prob_train(z, params, masks, mean, alpha, sup, D, L, U)
            if sup is not False and self._fused_ok(z, params) and self._whole_flow():
                mean, alpha = self._bn_stats(_lib.require_device())  # ... or of the whole-flow coupling kernel
                return ops.flow_log_prob_raw(z, params, mean, alpha, self.D, self.num_stages, self.num_layers,
                                             self.num_units, self.fusion, interval_consts=sup)[0]
            # support layer first (it is the last bijector of the stack), then the core's density:
            # log q(z) = log q_core(s^-1(z)) - log|det ds| -- same sum as density_estimator.py:395-416
            zc, ld_support = self.bijectors[-1].inverse_and_log_det(z)
            return self._core_log_prob(zc, params) - ld_support
        return self._core_log_prob(z, params)

    def _train_path(self, z, params):
        if self._stats_in_graph():
            return None  # the fused training pairs treat the statistics as constants
        shape = (z.size(0), params.size(0), z.size(1), self.D, self.num_stages, self.num_layers, self.num_units)
        if getattr(self, "reversible_training", True) and ops.flow_train_rev_supported(*shape):
            return "reversible"
        if ops.flow_train_supported(*shape):
            return "layers"
        return None

    def _ar_train_ok(self, z, params):
        """Training through the AR stack with z a constant: one forward kernel, one backward kernel."""
        return (self.arch_type == "AR" and getattr(self, "fused_ar_training", True) and torch.is_grad_enabled()
                and not self._stats_in_graph()
                and params.requires_grad and not z.requires_grad and z.dim() == 3
                and z.dtype == torch.float32 and params.dtype == torch.float32
                and z.size(0) == max(z.size(0), params.size(0))
                and ops.ar_flow_supported(self.D, self.num_layers, self.num_units)
                and ops.ar_flow_train_supported(z.size(0), params.size(0), self.D, self.num_layers, self.num_units))

    def _core_log_prob(self, z, params):
        if self._ar_fused_ok(z, params):
            return ops.ar_flow_log_prob_raw(z, params, *self._ar_args())[0]
        if self._ar_train_ok(z, params):
            masks, mean, alpha, D, L, U = self._ar_args()
            return ops.ar_flow_log_prob_train(z, params, masks, mean, alpha, None, D, L, U)
        if self._fused_ok(z, params):
            dev = _lib.require_device()
            mean, alpha = self._bn_stats(dev)
            lp, _, _ = ops.flow_log_prob_raw(z, params, mean, alpha, self.D, self.num_stages,
                                             self.num_layers, self.num_units, self.fusion)
            return lp
        if (self.arch_type == "coupling" and z.dtype == torch.float32 and params.dtype == torch.float32
                and z.dim() == 3 and z.size(0) == max(z.size(0), params.size(0))
                and self._train_path(z, params) is not None):
            # training (BatchNorm stats constant): the reversible pair -- whole-flow forward, one-kernel
            # backward from z0 -- or, for the shapes it does not cover, one fused kernel per layer each way
            mean, alpha = self._bn_stats(_lib.require_device())
            return ops.flow_log_prob_train(z, params, mean, alpha, self.D, self.num_stages, self.num_layers,
                                           self.num_units, reversible=self._train_path(z, params) == "reversible")
        z0, sum_log_det = self._core_inverse(z, params)
        log_q = torch.sum(-(z0 ** 2), axis=2) / 2.0 - self.D * np.log(np.sqrt(2.0 * np.pi))
        return log_q - sum_log_det
