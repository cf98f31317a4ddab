"""Backward passes of the bijector kernels (the autograd hooks of ops.py).

Each function calls one HIP backward kernel through the C ABI (tnf_*_backward); the
activations are recomputed inside the kernel from the saved layer input, so autograd keeps
only z and the parameters alive.  Parameter gradients are reduced over the samples inside the kernels: the
shape-generic coupling / MAF kernels deterministically (fixed workgroups, partial rows added in order, a workspace
from tnf_*_backward_workspace_bytes), the narrow MFMA layer kernels with one float atomic per parameter per workgroup.
"""
import torch

from . import _lib
from ._lib import lib, check


class _OpsProxy(object):  # ops imports this module lazily; the workspace cache lives there
    def __getattr__(self, name):
        from . import ops as _ops

        return getattr(_ops, name)


ops = _OpsProxy()


def _dev(t, dev):
    return t if t.device == dev else t.to(dev)


def coupling_backward(z, params, z_out, g_z, g_ld, D, L, U, upper, inverse):
    dev = _lib.require_device()
    home_z, home_p = z.device, params.device
    code = _lib.F32 if z.dtype == torch.float32 else _lib.F64
    zc = _dev(z.detach(), dev).contiguous()
    pc = _dev(params.detach(), dev)
    if pc.stride(1) != 1:
        pc = pc.contiguous()
    Mz, N = zc.shape[0], zc.shape[1]
    Mp = pc.shape[0]
    M = max(Mz, Mp)
    if Mz != M:  # broadcast z explicitly; its gradient is summed back over m below
        zc = zc.expand(M, N, D).contiguous()
    g_zo = torch.zeros((M, N, D), dtype=z.dtype, device=dev) if g_z is None else _dev(g_z, dev).contiguous()
    g_l = torch.zeros((M, N), dtype=z.dtype, device=dev) if g_ld is None else _dev(g_ld, dev).contiguous()
    gz = torch.empty((M, N, D), dtype=z.dtype, device=dev)
    gp = torch.zeros(tuple(params.shape), dtype=params.dtype, device=dev)
    pstride = pc.stride(0) if Mp > 1 else max(pc.stride(0), pc.shape[1])
    if N > 0:
        # with a workspace the shape-generic kernel (num_units > 16, odd D, float64 ...) reduces deterministically
        nbytes = check(lib.tnf_coupling_backward_workspace_bytes(code, M, Mp, N, D, L, U, int(upper)))
        ws = ops._workspace(nbytes, dev) if nbytes else None
        check(lib.tnf_coupling_backward_ws(code, zc.data_ptr(), pc.data_ptr(), g_zo.data_ptr(), g_l.data_ptr(),
                                           gz.data_ptr(), gp.data_ptr(), M, Mp, N, D, L, U, int(upper), int(inverse),
                                           pstride, gp.shape[1], ws.data_ptr() if nbytes else None, nbytes,
                                           _lib.stream_ptr()))
    if Mz != M:
        gz = gz.sum(0, keepdim=True)
    return (gz if home_z == dev else gz.to(home_z)), (gp if home_p == dev else gp.to(home_p))


def affine_backward(z, params, z_out, g_z, g_ld, D, inverse):
    dev = _lib.require_device()
    home_z, home_p = z.device, params.device
    code = _lib.F32 if z.dtype == torch.float32 else _lib.F64
    zc = _dev(z.detach(), dev).contiguous()
    pc = _dev(params.detach(), dev)
    if pc.stride(1) != 1:
        pc = pc.contiguous()
    Mz, N = zc.shape[0], zc.shape[1]
    Mp = pc.shape[0]
    M = max(Mz, Mp)
    if Mz != M:
        zc = zc.expand(M, N, D).contiguous()
    g_zo = torch.zeros((M, N, D), dtype=z.dtype, device=dev) if g_z is None else _dev(g_z, dev).contiguous()
    g_l = torch.zeros((Mp, 1), dtype=z.dtype, device=dev) if g_ld is None else _dev(g_ld, dev).contiguous()
    gz = torch.empty((M, N, D), dtype=z.dtype, device=dev)
    gp = torch.zeros(tuple(params.shape), dtype=params.dtype, device=dev)
    pstride = pc.stride(0) if Mp > 1 else max(pc.stride(0), pc.shape[1])
    check(lib.tnf_affine_backward(code, zc.data_ptr(), pc.data_ptr(), g_zo.data_ptr(), g_l.data_ptr(),
                                  gz.data_ptr(), gp.data_ptr(), M, Mp, N, D, int(inverse), pstride, gp.shape[1],
                                  _lib.stream_ptr()))
    if Mz != M:
        gz = gz.sum(0, keepdim=True)
    return (gz if home_z == dev else gz.to(home_z)), (gp if home_p == dev else gp.to(home_p))


def bn_apply_backward(g_z, alpha, inverse):
    if g_z is None:
        return None
    dev = _lib.require_device()
    home = g_z.device
    code = _lib.F32 if g_z.dtype == torch.float32 else _lib.F64
    gc = _dev(g_z, dev).contiguous()
    ac = _dev(alpha.detach().float(), dev).contiguous()
    out = torch.empty_like(gc)
    D = gc.shape[-1]
    check(lib.tnf_bn_apply_backward(code, gc.data_ptr(), ac.data_ptr(), out.data_ptr(), gc.numel() // D, D,
                                    int(inverse), _lib.stream_ptr()))
    return out if home == dev else out.to(home)


def bn_batch_backward(z_norm, alpha, g_zn, g_ld):
    dev = _lib.require_device()
    home = z_norm.device
    zc = _dev(z_norm.detach(), dev).contiguous()
    D = zc.shape[-1]
    rows = zc.numel() // D
    gc = torch.zeros_like(zc) if g_zn is None else _dev(g_zn, dev).contiguous().float()
    gl = None if g_ld is None else _dev(g_ld, dev).reshape(1).float().contiguous()
    ac = _dev(alpha.detach().float(), dev).contiguous()
    out = torch.empty_like(zc)
    ws_bytes = lib.tnf_bn_batch_workspace_bytes(D)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.tnf_bn_batch_backward_f32(zc.data_ptr(), gc.data_ptr(), None if gl is None else gl.data_ptr(),
                                        ac.data_ptr(), out.data_ptr(), rows, D, ws.data_ptr(), ws_bytes,
                                        _lib.stream_ptr()))
    return out if home == dev else out.to(home)


def maf_backward(z, params, masks, g_z, g_ld, D, L, U):
    dev = _lib.require_device()
    home_z, home_p = z.device, params.device
    code = _lib.F32 if z.dtype == torch.float32 else _lib.F64
    zc = _dev(z.detach(), dev).contiguous()
    pc = _dev(params.detach(), dev)
    if pc.stride(1) != 1:
        pc = pc.contiguous()
    mk = _dev(masks.to(z.dtype), dev).contiguous()
    Mz, N = zc.shape[0], zc.shape[1]
    Mp = pc.shape[0]
    M = max(Mz, Mp)
    if Mz != M:
        zc = zc.expand(M, N, D).contiguous()
    g_zo = torch.zeros((M, N, D), dtype=z.dtype, device=dev) if g_z is None else _dev(g_z, dev).contiguous()
    g_l = torch.zeros((M, N), dtype=z.dtype, device=dev) if g_ld is None else _dev(g_ld, dev).contiguous()
    gz = torch.empty((M, N, D), dtype=z.dtype, device=dev)
    gp = torch.zeros(tuple(params.shape), dtype=params.dtype, device=dev)
    pstride = pc.stride(0) if Mp > 1 else max(pc.stride(0), pc.shape[1])
    if N > 0:
        nbytes = check(lib.tnf_maf_backward_workspace_bytes(code, M, Mp, N, D, L, U))
        ws = ops._workspace(nbytes, dev) if nbytes else None
        check(lib.tnf_maf_backward_ws(code, zc.data_ptr(), pc.data_ptr(), mk.data_ptr(), g_zo.data_ptr(), g_l.data_ptr(),
                                      gz.data_ptr(), gp.data_ptr(), M, Mp, N, D, L, U, pstride, gp.shape[1],
                                      ws.data_ptr() if nbytes else None, nbytes, _lib.stream_ptr()))
    if Mz != M:
        gz = gz.sum(0, keepdim=True)
    return (gz if home_z == dev else gz.to(home_z)), (gp if home_p == dev else gp.to(home_p))
