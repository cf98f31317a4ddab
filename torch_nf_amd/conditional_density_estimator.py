"""ConditionalDensityEstimator with the reference's interface.

Drop-in for torch_nf/conditional_density_estimator.py:10-104 of the reference: an
nn.Module whose `param_net` (stock torch.nn Linear/Tanh[/Dropout] stack -- it runs on
hipBLASLt when the module lives on the GPU and is NOT re-implemented here) maps a
context x (M, D_x) to one flat flow-parameter row per context (M, D_params); sampling
and density evaluation are delegated to the wrapped NormFlow, i.e. to the HIP kernels
with per-context weights (M_p = M).
"""
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, ops
from . import density_estimator as de
from .bijectors import _Checked


class _SplitKLinear(torch.autograd.Function):
    """y = x W^T + b for a hidden Linear of param_net with MANY rows (contexts), with the weight gradient g^T x computed as
    a batched product over row blocks + a sum: the stock backward is one GEMM with K = M rows and a 64 x 64 output, which
    hipBLASLt runs on a handful of workgroups (0.6 ms at M = 2^18 for 2 GFLOP; 2 x that per SNPE step).  Same arithmetic up
    to the order of the fp32 sum over rows; the module stays a stock nn.Linear (state_dict, optimizer unchanged)."""

    BLOCKS = 256

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        M = x.size(0)
        B = _SplitKLinear.BLOCKS
        rows = (M // B) * B
        g_w = torch.bmm(g[:rows].view(B, M // B, -1).transpose(1, 2), x[:rows].reshape(B, M // B, -1)).sum(0)
        if rows < M:
            g_w = g_w + g[rows:].t() @ x[rows:]
        g_x = g @ weight if ctx.needs_input_grad[0] else None
        return g_x, g_w, g.sum(0)


def _pos_dx(val):
    if val < 1:
        raise ValueError("D_x %d must be greater than 0." % val)
    return val


def _pos_dparams(val):
    if val < 1:
        raise ValueError("D_params %d must be greater than 0." % val)
    return val


class ConditionalDensityEstimator(torch.nn.Module):
    """:param density_estimator: a NormFlow built with conditioner=True (exact type, as in
    the reference :48-49).  :param D_x: context width.  :param hidden_layers: list of
    hidden widths of param_net.  :param dropout: add nn.Dropout after every activation."""

    D_x = _Checked("D_x", int, _pos_dx)
    D_params = _Checked("D_params", int, _pos_dparams)

    def __init__(self, density_estimator, D_x, hidden_layers, dropout=False):
        super().__init__()
        self.density_estimator = density_estimator
        self.D_x = D_x
        self.D_params = density_estimator.D_params
        self.hidden_layers = hidden_layers
        self.dropout = dropout

        # conditional_density_estimator.py:19-40 (module names are part of the state_dict)
        widths = [D_x] + list(self.hidden_layers)
        layers = []
        for i in range(1, len(widths)):
            layers.append(("linear%d" % i, torch.nn.Linear(widths[i - 1], widths[i])))
            layers.append((("tanh%d" if i == 1 else "relu%d") % i, torch.nn.Tanh()))
            if self.dropout:
                layers.append(("dropout%d" % i, torch.nn.Dropout()))
        layers.append(("linear%d" % len(widths), torch.nn.Linear(widths[-1], self.D_params)))
        self.param_net = torch.nn.Sequential(OrderedDict(layers))
        # extension (not in the reference): fuse param_net's last Linear into the flow kernel for
        # one-sample-per-context calls with at least this many contexts
        self.fuse_conditioner = True
        self.fuse_min_contexts = 16
        if density_estimator.device.type != "cpu":
            self.param_net.to(density_estimator.device)

    @property
    def density_estimator(self):
        return self.__dict__["_cde_flow"]

    @density_estimator.setter
    def density_estimator(self, val):
        if type(val) not in [de.NormFlow]:
            from .error_formatters import format_type_err_msg
            raise TypeError(format_type_err_msg(self, "density_estimator", val, de.DensityEstimator))
        self.__dict__["_cde_flow"] = val

    @property
    def hidden_layers(self):
        return self.__dict__["_cde_hidden"]

    @hidden_layers.setter
    def hidden_layers(self, val):
        from .error_formatters import format_type_err_msg
        if type(val) is not list:
            raise TypeError(format_type_err_msg(self, "hidden_layers", val, list))
        for i, width in enumerate(val):
            if type(width) is not int:
                raise TypeError(format_type_err_msg(self, "hidden_layers[%d]" % i, val, int))
            if width < 1:
                raise ValueError("Hidden unit counts must be positive.")
        self.__dict__["_cde_hidden"] = val

    def _trunk(self, x):
        """param_net without its last Linear (what the fused kernels take as `h`)."""
        mods = self.param_net[:-1]
        if not (torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and x.size(0) >= 32768 and x.dtype == torch.float32):
            return mods(x) if len(mods) else x
        h = x
        for mod in mods:  # rows >= 32768: the hidden Linears' weight gradients as split-K products (_SplitKLinear)
            if type(mod) is torch.nn.Linear and mod.bias is not None and mod.weight.dtype == torch.float32:
                h = _SplitKLinear.apply(h, mod.weight, mod.bias)
            else:
                h = mod(h)
        return h

    def _params_for(self, x):
        weight = next(self.param_net.parameters())
        if x.device != weight.device:
            x = x.to(weight.device)
        return self.param_net(x)

    def __call__(self, x, N=100, freeze_bn=False):
        """conditional_density_estimator.py:93-99: (z (M,N,D), log_q (M,N)).  One sample per context with frozen
        statistics runs the fused conditioner + flow kernel in its sampling direction (tnf_cond_flow_forward_f32): like
        `log_prob`'s fused path, the (M, D_params) parameter tensor is never materialised."""
        if N == 1 and freeze_bn and self._fused_sampling_ok(x):
            nf = self.density_estimator
            omega = np.random.normal(0.0, 1.0, (x.size(0), 1, nf.D))  # the reference's host draw (density_estimator.py:366)
            o64 = torch.as_tensor(omega, dtype=torch.float64).to(_lib.require_device())
            z, sld = self._fused_sampling(x, o64.float())
            return self._home(x, z, ops.base_log_density_f64(o64) - sld)
        params = self._params_for(x)
        return self.density_estimator(N=N, params=params, freeze_bn=freeze_bn)

    def sample(self, x, N=100, freeze_bn=True, generator=None):
        """Extension (not in the reference): like `__call__`, but the base draw comes from the device RNG
        (`NormFlow.sample`), so posterior sampling is not bound by `np.random.normal` and the PCIe copy
        (2*10^5 draws at D=6: 13 ms through `cde(x0, N)`, 0.3 ms here).  Not reproducible against np.random.seed."""
        if N == 1 and freeze_bn and self._fused_sampling_ok(x):
            omega = torch.randn((x.size(0), 1, self.density_estimator.D), device=_lib.require_device(),
                                dtype=torch.float32, generator=generator)
            z, sld = self._fused_sampling(x, omega)
            return self._home(x, z, ops.base_log_density_f64(omega) - sld)
        params = self._params_for(x)
        return self.density_estimator.sample(N, params, freeze_bn=freeze_bn, generator=generator)

    def _home(self, x, z, log_q):
        home = next(self.param_net.parameters()).device  # where NormFlow would return them: the parameters' device
        return (z if z.device == home else z.to(home)), (log_q if log_q.device == home else log_q.to(home))

    def _fused_sampling_ok(self, x):
        """cde(x, N = 1) with frozen statistics on a coupling flow, outside autograd (the samples' gradient with respect
        to the context network goes through the materialised path)."""
        nf = self.density_estimator
        last = self.param_net[-1]
        if not (self.fuse_conditioner and x.dim() == 2 and x.size(0) >= self.fuse_min_contexts):
            return False
        if nf.arch_type != "coupling" or nf.support_layer is not None or nf._stats_in_graph():
            return False
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.param_net.parameters())):
            return False
        if x.dtype != torch.float32 or last.weight.dtype != torch.float32:
            return False
        return ops.cond_flow_supported(nf.D, nf.num_stages, nf.num_layers, nf.num_units, last.in_features)

    def _fused_sampling(self, x, omega):
        """omega (M, 1, D) float32 on the device -> (z (M, 1, D), sum_log_det (M, 1))."""
        nf = self.density_estimator
        last = self.param_net[-1]
        weight = last.weight
        xd = x if x.device == weight.device else x.to(weight.device)
        with torch.no_grad():
            h = self.param_net[:-1](xd) if len(self.param_net) > 1 else xd
        mean, alpha = nf._bn_stats(_lib.require_device())
        z, sld = ops.cond_flow_forward_raw(omega[:, 0, :], h, weight, last.bias, mean, alpha, nf.D, nf.num_stages,
                                           nf.num_layers, nf.num_units)
        return z[:, None, :], sld[:, None]

    def _fused_conditioner_ok(self, z, x):
        """One sample per context (the SNPE layout z[:, None, :]) on a coupling flow: the last Linear
        of param_net runs inside the flow kernel (tnf_cond_flow_log_prob_f32) and the (M, D_params)
        parameter tensor is never materialised."""
        nf = self.density_estimator
        last = self.param_net[-1]
        if not (self.fuse_conditioner and z.dim() == 3 and z.size(1) == 1 and x.dim() == 2
                and z.size(0) == x.size(0) and z.size(0) >= self.fuse_min_contexts):
            return False
        if nf.arch_type != "coupling" or nf.support_layer is not None or z.size(2) != nf.D:
            return False
        if nf._stats_in_graph():
            return False  # the fused kernels treat the statistics as constants (bijectors.py:414-415 keeps their graph)
        if z.dtype != torch.float32 or x.dtype != torch.float32 or last.weight.dtype != torch.float32:
            return False
        return ops.cond_flow_supported(nf.D, nf.num_stages, nf.num_layers, nf.num_units, last.in_features)

    def log_prob(self, z, x):
        """conditional_density_estimator.py:101-104."""
        if self._fused_conditioner_ok(z, x):
            nf = self.density_estimator
            last = self.param_net[-1]
            weight = last.weight
            home = z.device
            xd = x if x.device == weight.device else x.to(weight.device)
            h = self._trunk(xd)
            mean, alpha = nf._bn_stats(_lib.require_device())
            if torch.is_grad_enabled() and (z.requires_grad or h.requires_grad or weight.requires_grad
                                            or last.bias.requires_grad):
                # training: forward keeps activations, hand-written backward (BatchNorm stats constant)
                lp = ops.cond_flow_log_prob_train(z[:, 0, :], h, weight, last.bias, mean, alpha, nf.D,
                                                  nf.num_stages, nf.num_layers, nf.num_units)
            else:
                lp, _, _ = ops.cond_flow_log_prob_raw(z[:, 0, :], h, weight, last.bias, mean, alpha, nf.D,
                                                      nf.num_stages, nf.num_layers, nf.num_units)
            lp = lp[:, None]
            return lp if lp.device == home else lp.to(home)
        params = self._params_for(x)
        return self.density_estimator.log_prob(z, params)
