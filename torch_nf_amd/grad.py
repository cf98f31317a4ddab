"""Backward passes of the bijector kernels (autograd hooks of ops.py)."""


def _todo(what):
    raise NotImplementedError(
        "torch_nf_amd: the HIP backward kernel for %s is not built yet; run under torch.no_grad() "
        "(there is deliberately no PyTorch fallback)." % what)


def coupling_backward(z, params, z_out, g_z, g_ld, D, L, U, upper, inverse):
    _todo("RealNVP")


def affine_backward(z, params, z_out, g_z, g_ld, D, inverse):
    _todo("Affine")


def bn_apply_backward(g_z, alpha, inverse):
    _todo("BatchNorm")
