"""CPU oracle for the coupling-flow hot path (TEST INFRASTRUCTURE ONLY).

This module is a functional, pure-PyTorch-CPU restatement of the reference's
`forward` / `inverse` / `log_prob` arithmetic for RealNVP + BatchNorm + Affine
coupling flows.  It exists so that `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` have something to check the HIP kernels
against (and to time on the host cores).  Nothing in `torch_nf_amd/` may import
it: the product path is the HIP extension and fails loudly without it.

Pinned: every function here is checked against outputs of the reference itself
(imported from /root/reference in the build container by
`oracle/gen_golden.py`; vectors committed under `tests/golden/`), to exact
equality in fp32 and fp64 -- it issues the same aten ops in the same order.

Reference citations (relative to the upstream repo root):
  * twin-MLP layer ............ torch_nf/bijectors.py:208-242  (RealNVP._t_s_layer)
  * coupling forward .......... torch_nf/bijectors.py:145-181
  * coupling inverse .......... torch_nf/bijectors.py:183-206
  * parameter count ........... torch_nf/bijectors.py:244-262
  * Affine .................... torch_nf/bijectors.py:277-318
  * BatchNorm ................. torch_nf/bijectors.py:389-426
  * stack layout .............. torch_nf/density_estimator.py:260-270
  * flow forward .............. torch_nf/density_estimator.py:364-388
  * flow inverse .............. torch_nf/density_estimator.py:390-406
  * log_prob .................. torch_nf/density_estimator.py:408-416
  * MAF ....................... torch_nf/bijectors.py:597-806
  * ToInterval / torch_atanh .. torch_nf/bijectors.py:429-557
  * ToSimplex ................. torch_nf/bijectors.py:560-594
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "coupling_dims",
    "coupling_num_params",
    "coupling",
    "affine",
    "bn_inverse",
    "bn_forward_frozen",
    "bn_forward_batch",
    "flow_layout",
    "flow_num_params",
    "flow_inverse",
    "flow_log_prob",
    "flow_forward",
    "base_log_density_f64",
    "maf_masks",
    "maf_num_params",
    "maf",
    "ar_flow_log_prob",
    "ar_flow_forward",
    "interval_consts",
    "to_interval",
    "to_simplex",
]


# --------------------------------------------------------------------------
# RealNVP coupling layer
# --------------------------------------------------------------------------
def coupling_dims(D, transform_upper):
    """(conditioner width, transformed width) -- bijectors.py:155-165."""
    h = D // 2
    d_in, d_out = h, h
    if D % 2 == 1:
        d_in += int(not transform_upper)
        d_out += int(transform_upper)
    return d_in, d_out


def coupling_num_params(D, num_layers, num_units, transform_upper):
    """bijectors.py:244-262."""
    d_in, d_out = coupling_dims(D, transform_upper)
    U, L = num_units, num_layers
    return 2 * (d_in * U + d_out * U + d_out + U + (L - 1) * (U + 1) * U)


def _twin_layer(x_t, x_s, params, off, d_in, d_out, squash):
    """One layer of the shift (t) and log-scale (s) nets: packed params are
    [W_t | W_s | b_t | b_s], W row-major [in][out] -- bijectors.py:222-241."""
    n_w = d_in * d_out
    w_t = params[:, off:off + n_w].view(-1, d_in, d_out)
    off += n_w
    w_s = params[:, off:off + n_w].view(-1, d_in, d_out)
    off += n_w
    b_t = params[:, off:off + d_out].view(-1, 1, d_out)
    off += d_out
    b_s = params[:, off:off + d_out].view(-1, 1, d_out)
    off += d_out
    t = torch.matmul(x_t, w_t) + b_t
    s = torch.matmul(x_s, w_s) + b_s
    if squash:
        t = torch.tanh(t)
        s = torch.tanh(s)
    return t, s, off


def coupling(z, params, D, num_layers, num_units, transform_upper, inverse):
    """RealNVP layer.  z (M,N,D), params (M, >=|theta|) -> (z', log_det (M,N)).

    forward : z2' = t + z2*exp(s)      (bijectors.py:172)
    inverse : z2' = (z2 - t)/exp(s)    (bijectors.py:198)
    log_det is the FORWARD log-det sum(s) in both directions (bijectors.py:179,205).
    """
    h = D // 2
    d_in, d_out = coupling_dims(D, transform_upper)
    if transform_upper:
        z1, z2 = z[:, :, :h], z[:, :, h:]
    else:
        z2, z1 = z[:, :, :h], z[:, :, h:]
    t, s, off = _twin_layer(z1, z1, params, 0, d_in, num_units, True)
    for _ in range(num_layers - 1):
        t, s, off = _twin_layer(t, s, params, off, num_units, num_units, True)
    t, s, off = _twin_layer(t, s, params, off, num_units, d_out, False)
    if inverse:
        z2 = (z2 - t) / torch.exp(s)
    else:
        z2 = t + z2 * torch.exp(s)
    if transform_upper:
        out = torch.cat([z1, z2], dim=2)
    else:
        out = torch.cat([z2, z1], dim=2)
    return out, torch.sum(s, dim=2)


# --------------------------------------------------------------------------
# Affine
# --------------------------------------------------------------------------
def affine(z, params, D, inverse):
    """params = [alpha (D) | shift (D)]; log_det = sum(alpha), shape (M,1)
    in both directions -- bijectors.py:277-315."""
    alpha = params[:, :D]
    scale = torch.exp(alpha)[:, None, :]
    shift = params[:, D:2 * D][:, None, :]
    if inverse:
        out = (z - shift) / scale
    else:
        out = scale * z + shift
    return out, torch.sum(alpha, axis=1, keepdim=True)


# --------------------------------------------------------------------------
# BatchNorm bijector
# --------------------------------------------------------------------------
def bn_inverse(z, mean, alpha):
    """bijectors.py:420-426 (0-dim log_det)."""
    out = z * alpha
    out = out + mean
    return out, -torch.sum(torch.log(alpha))


def bn_forward_frozen(z, mean, alpha):
    """use_last=True branch -- bijectors.py:397-399,417."""
    return (z - mean) / alpha, -torch.sum(torch.log(alpha))


def bn_forward_batch(z, eps=1e-5, momentum=0.1):
    """use_last=False branch -- bijectors.py:401-417.
    Returns (z_norm, log_det, mean, alpha); the caller caches mean/alpha."""
    D = z.shape[-1]
    z_vec = z.view(-1, D)
    z_var = torch.var(z_vec, dim=0)
    z_norm = F.batch_norm(z_vec, None, None, None, None, True, momentum, eps)
    z_norm_var = torch.var(z_norm, dim=0)
    alpha = torch.sqrt(z_var) / torch.sqrt(z_norm_var)
    zn_alpha = z_norm * alpha[None, :]
    mean = torch.mean(z_vec - zn_alpha, dim=0)
    z_norm = z_norm.view(z.shape[0], z.shape[1], D)
    return z_norm, -torch.sum(torch.log(alpha)), mean, alpha


# --------------------------------------------------------------------------
# the 'coupling' NormFlow stack
# --------------------------------------------------------------------------
def flow_layout(D, num_stages, num_layers, num_units):
    """[(kind, n_params, transform_upper)] in forward order:
    per stage RealNVP(up), BN, RealNVP(low), BN, Affine -- density_estimator.py:260-270."""
    out = []
    for _ in range(num_stages):
        out.append(("coupling", coupling_num_params(D, num_layers, num_units, True), True))
        out.append(("bn", 0, None))
        out.append(("coupling", coupling_num_params(D, num_layers, num_units, False), False))
        out.append(("bn", 0, None))
        out.append(("affine", 2 * D, None))
    return out


def flow_num_params(D, num_stages, num_layers, num_units):
    return sum(n for _, n, _ in flow_layout(D, num_stages, num_layers, num_units))


def flow_inverse(z, params, D, num_stages, num_layers, num_units, bn_stats):
    """density_estimator.py:390-406.  bn_stats: list of (mean, alpha), one per BN
    layer in forward order.  sum_log_det is allocated fp32 like the reference."""
    layout = flow_layout(D, num_stages, num_layers, num_units)
    idx = sum(n for _, n, _ in layout)
    bn_i = sum(1 for k, _, _ in layout if k == "bn")
    sum_log_det = torch.zeros((z.shape[0], z.shape[1]))
    for kind, n, upper in reversed(layout):
        if kind == "coupling":
            z, ld = coupling(z, params[:, idx - n:idx], D, num_layers, num_units, upper, True)
            idx -= n
        elif kind == "affine":
            z, ld = affine(z, params[:, idx - n:idx], D, True)
            idx -= n
        else:
            bn_i -= 1
            z, ld = bn_inverse(z, bn_stats[bn_i][0], bn_stats[bn_i][1])
        sum_log_det += ld
    return z, sum_log_det


def flow_log_prob(z, params, D, num_stages, num_layers, num_units, bn_stats):
    """density_estimator.py:408-416."""
    z0, sum_log_det = flow_inverse(z, params, D, num_stages, num_layers, num_units, bn_stats)
    log_q = torch.sum(-(z0 ** 2), axis=2) / 2.0 - D * np.log(np.sqrt(2.0 * np.pi))
    return log_q - sum_log_det


def base_log_density_f64(omega):
    """log N(omega; 0, I) the way the reference does it, in float64 numpy
    -- density_estimator.py:369-372."""
    return np.log(np.prod(np.exp((-np.square(omega)) / 2.0) / np.sqrt(2.0 * np.pi), axis=2))


def flow_forward(omega, params, D, num_stages, num_layers, num_units, bn_stats=None):
    """density_estimator.py:364-388 with the host draw `omega` (float64 numpy,
    shape (M,N,D)) injected.  bn_stats=None -> batch statistics (freeze_bn=False),
    else the frozen (mean, alpha) pairs.  Returns (z fp32, log_q fp64, bn_stats)."""
    z = torch.tensor(omega).float()
    log_q = torch.tensor(base_log_density_f64(omega))
    idx = 0
    bn_i = 0
    new_stats = []
    for kind, n, upper in flow_layout(D, num_stages, num_layers, num_units):
        if kind == "coupling":
            z, ld = coupling(z, params[:, idx:idx + n], D, num_layers, num_units, upper, False)
            idx += n
        elif kind == "affine":
            z, ld = affine(z, params[:, idx:idx + n], D, False)
            idx += n
        else:
            if bn_stats is None:
                z, ld, mean, alpha = bn_forward_batch(z)
            else:
                mean, alpha = bn_stats[bn_i]
                z, ld = bn_forward_frozen(z, mean, alpha)
            new_stats.append((mean, alpha))
            bn_i += 1
        log_q = log_q - ld
    return z, log_q, new_stats


# --------------------------------------------------------------------------
# MAF (masked autoregressive flow) -- bijectors.py:597-806, arch_type "AR"
# --------------------------------------------------------------------------
def maf_masks(D, num_layers, num_units, fwd_fac=True, rng=np.random):
    """Degree vectors `ms` and binary masks `Ms` drawn exactly like MAF._get_masks
    (bijectors.py:663-696): hidden degrees from rng.randint(1, D, (K,)), layer masks
    M[k_prev, k] = m_prev[k_prev] <= m[k], final mask strict (<).  fwd_fac=False keeps the
    reference's arange(D, -1, -1) (D+1 entries, the last one unused)."""
    ms, Ms = [], []
    k_prev = D
    m_prev = np.arange(1, D + 1) if fwd_fac else np.arange(D, -1, -1)
    for _ in range(num_layers):
        m = rng.randint(1, D, (num_units,))
        Ms.append((m_prev[:k_prev, None] <= m[None, :]).astype(np.float32))
        ms.append(m)
        k_prev, m_prev = num_units, m
    m = np.arange(1, D + 1) if fwd_fac else np.arange(D, -1, -1)
    Ms.append((m_prev[:k_prev, None] < m[None, :D]).astype(np.float32))
    ms.append(m)
    return ms, Ms


def maf_num_params(D, num_layers, num_units):
    """bijectors.py:796-806."""
    return 2 * (2 * D * num_units + (num_layers - 1) * num_units ** 2)


def _maf_net(z, params, D, L, U, Ms):
    """(f_mu, f_alpha): twin masked MLPs without biases, tanh on the hidden layers
    (bijectors.py:702-790).  params = [W_mu0 | W_alpha0 | ... | W_mu_last | W_alpha_last]."""
    dims = [D] + [U] * L + [D]
    off = 0
    f_mu, f_alpha = z, z
    for i in range(L + 1):
        d_in, d_out = dims[i], dims[i + 1]
        n = d_in * d_out
        mask = torch.as_tensor(Ms[i])[None, :, :]
        w_mu = mask * params[:, off:off + n].view(-1, d_in, d_out)
        off += n
        w_alpha = mask * params[:, off:off + n].view(-1, d_in, d_out)
        off += n
        f_mu = torch.matmul(f_mu, w_mu)
        f_alpha = torch.matmul(f_alpha, w_alpha)
        if i < L:
            f_mu = torch.tanh(f_mu)
            f_alpha = torch.tanh(f_alpha)
    return f_mu, f_alpha


def maf(z, params, D, num_layers, num_units, Ms, inverse):
    """MAF.forward_and_log_det (D-1 passes, bijectors.py:742-756) / inverse_and_log_det
    (one pass, :758-764)."""
    if inverse:
        f_mu, f_alpha = _maf_net(z, params, D, num_layers, num_units, Ms)
        return (z - f_mu) / torch.exp(f_alpha), torch.sum(f_alpha, dim=2)
    u = z
    for _ in range(D - 1):
        f_mu, f_alpha = _maf_net(z, params, D, num_layers, num_units, Ms)
        z = u * torch.exp(f_alpha) + f_mu
    return z, torch.sum(f_alpha, axis=2)


def ar_flow_log_prob(z, params, D, num_layers, num_units, Ms, bn_stat):
    """NormFlow(arch_type="AR").log_prob: stack [MAF, BatchNorm, Affine]
    (density_estimator.py:271-274, 390-416)."""
    n_maf = maf_num_params(D, num_layers, num_units)
    sum_log_det = torch.zeros((z.shape[0], z.shape[1]))
    z, ld = affine(z, params[:, n_maf:n_maf + 2 * D], D, True)
    sum_log_det += ld
    z, ld = bn_inverse(z, bn_stat[0], bn_stat[1])
    sum_log_det += ld
    z, ld = maf(z, params[:, :n_maf], D, num_layers, num_units, Ms, True)
    sum_log_det += ld
    log_q = torch.sum(-(z ** 2), axis=2) / 2.0 - D * np.log(np.sqrt(2.0 * np.pi))
    return log_q - sum_log_det


def ar_flow_forward(omega, params, D, num_layers, num_units, Ms, bn_stat=None):
    """NormFlow(arch_type="AR").forward with the host draw injected."""
    n_maf = maf_num_params(D, num_layers, num_units)
    z = torch.tensor(omega).float()
    log_q = torch.tensor(base_log_density_f64(omega))
    z, ld = maf(z, params[:, :n_maf], D, num_layers, num_units, Ms, False)
    log_q = log_q - ld
    if bn_stat is None:
        z, ld, mean, alpha = bn_forward_batch(z)
    else:
        mean, alpha = bn_stat
        z, ld = bn_forward_frozen(z, mean, alpha)
    log_q = log_q - ld
    z, ld = affine(z, params[:, n_maf:n_maf + 2 * D], D, False)
    log_q = log_q - ld
    return z, log_q, (mean, alpha)


# --------------------------------------------------------------------------
# Support layers
# --------------------------------------------------------------------------
def interval_consts(lb, ub):
    """Per-feature constants of ToInterval.__init__ (bijectors.py:454-480): float32 (1,1,D) tensors
    (tanh_flg, softplus_flg, tanh_m, tanh_c, softplus_m, softplus_c)."""
    lb, ub = np.asarray(lb, dtype=np.float64), np.asarray(ub, dtype=np.float64)
    D = lb.shape[0]
    tf, sf = np.zeros(D), np.zeros(D)
    tm, tc, sm, sc = np.ones(D), np.zeros(D), np.ones(D), np.zeros(D)
    for i in range(D):
        has_lb, has_ub = not np.isneginf(lb[i]), not np.isposinf(ub[i])
        if has_lb and has_ub:
            tf[i], tm[i], tc[i] = 1, (ub[i] - lb[i]) / 2.0, (ub[i] + lb[i]) / 2.0
        elif has_lb:
            sf[i], sm[i], sc[i] = 1, 1.0, lb[i]
        elif has_ub:
            sf[i], sm[i], sc[i] = 1, -1.0, ub[i]
    return tuple(torch.tensor(a).float()[None, None, :] for a in (tf, sf, tm, tc, sm, sc))


def _interval_tanh_ldj(z, tf, tm, eps):
    tanh_z = torch.tanh(z)
    return tanh_z, torch.sum(tf * (torch.log(tm) + torch.log(1.0 - (tanh_z ** 2) + eps)), axis=2)


def to_interval(z, consts, inverse, eps=1e-12):
    """ToInterval.forward_and_log_det (bijectors.py:509-527) / inverse_and_log_det (:529-553)."""
    tf, sf, tm, tc, sm, sc = consts
    if not inverse:
        tanh_z, tanh_ldj = _interval_tanh_ldj(z, tf, tm, eps)
        z = tf * (tm * tanh_z + tc) + (1 - tf) * z
        out = sm * F.softplus(z) + sc
        softplus_ldj = torch.sum(sf * F.logsigmoid(z), axis=2)
        z = sf * out + (1 - sf) * z
        return z, tanh_ldj + softplus_ldj
    softplus_inv = torch.log(torch.exp(sf * (z - sc) / sm) - 1 + eps)
    z = sf * softplus_inv + (1 - sf) * z
    softplus_ldj = torch.sum(sf * F.logsigmoid(z), axis=2)
    x = tf * (z - tc) / tm
    tanh_inv = 0.5 * (torch.log(1 + x + eps) - torch.log(1 - x + eps))  # torch_atanh, :555-557
    z = tf * tanh_inv + (1 - tf) * z
    _, tanh_ldj = _interval_tanh_ldj(z, tf, tm, eps)
    return z, tanh_ldj + softplus_ldj


def to_simplex(z, D_attr):
    """ToSimplex.forward_and_log_det (bijectors.py:574-591); D_attr is the bijector's D attribute."""
    ex = torch.exp(z)
    sum_ex = torch.sum(ex, dim=2)
    den = sum_ex + 1.0
    log_det = torch.log(1.0 - (sum_ex / den) + 1e-10) - D_attr * torch.log(den) + torch.sum(z, axis=2)
    return torch.cat((ex / den[:, :, None], 1.0 / den[:, :, None]), axis=2), log_det
