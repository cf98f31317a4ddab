"""Analysis script (not collected by pytest; lives under tests/ because it imports the oracle): emulate the 3-MFMA
f16-split coupling MLP in numpy and compare log_prob with the oracle.  Run from the repo root: python tests/split_f16_emulation.py"""
import sys, numpy as np, torch
sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import flow_oracle as orc
from conftest import load_golden

def split(v, scale=1.0):
    v = (v * scale).astype(np.float32)
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi, lo

def mm3(x, w, wscale):
    """x (N,K) f32, w (K,O) f32 -> f32 accumulate of xh*wh + xl*wh + xh*wl, weights scaled by wscale"""
    xh, xl = split(x)
    wh, wl = split(w, wscale)
    f = lambda a, b: a.astype(np.float32) @ b.astype(np.float32)   # products of f16 values are exact in f32
    return (f(xh, wh) + f(xl, wh) + f(xh, wl)) / np.float32(wscale)

def coupling(z, p, D, L, U, upper, inverse, wscale):
    h = D // 2
    z1, z2 = (z[:, :h], z[:, h:]) if upper else (z[:, h:], z[:, :h])
    off = 0
    def layer(xt, xs, din, dout, act):
        nonlocal off
        wt = p[off:off + din * dout].reshape(din, dout); off += din * dout
        ws = p[off:off + din * dout].reshape(din, dout); off += din * dout
        bt = p[off:off + dout]; off += dout
        bs = p[off:off + dout]; off += dout
        t = mm3(xt, wt, wscale) + bt
        s = mm3(xs, ws, wscale) + bs
        return (np.tanh(t), np.tanh(s)) if act else (t, s)
    t, s = layer(z1, z1, h, U, True)
    for _ in range(L - 1):
        t, s = layer(t, s, U, U, True)
    t, s = layer(t, s, U, h, False)
    z2 = (z2 - t) / np.exp(s) if inverse else t + z2 * np.exp(s)
    out = np.concatenate([z1, z2], 1) if upper else np.concatenate([z2, z1], 1)
    return out.astype(np.float32), s.sum(1).astype(np.float32)

def flow_log_prob(z, params, D, S, L, U, stats, wscale):
    lay = orc.flow_layout(D, S, L, U)
    idx = sum(n for _, n, _ in lay); bi = 2 * S
    sld = np.zeros(z.shape[0], np.float32)
    for kind, n, upper in reversed(lay):
        if kind == "coupling":
            z, ld = coupling(z, params[idx - n:idx], D, L, U, upper, True, wscale); idx -= n
        elif kind == "affine":
            a, sh = params[idx - n:idx - n + D], params[idx - n + D:idx]; idx -= n
            z = (z - sh) / np.exp(a); ld = a.sum()
        else:
            bi -= 1; m, al = stats[bi]; z = z * al + m; ld = -np.log(al).sum()
        sld = sld + ld
    return (-(z ** 2)).sum(1) / 2 - D * np.log(np.sqrt(2 * np.pi)) - sld

g = load_golden("flow")
for ci, (D, S, L, U, N) in enumerate(g["meta"].tolist()):
    if D not in (32, 64): continue
    k = "f%02d_" % ci
    stats = list(zip(g[k + "bn_mean"], g[k + "bn_alpha"]))
    for ws in (1.0, 256.0):
        lp = flow_log_prob(g[k + "z_test"][0], g[k + "params"][0], D, S, L, U, stats, ws)
        ref = g[k + "log_prob"][0]
        rel = np.abs(lp - ref) / np.maximum(np.abs(ref), 1e-3)
        print("case %d D=%d S=%d L=%d wscale=%g: max rel err %.2e  (mean %.2e)" % (ci, D, S, L, ws, rel.max(), rel.mean()))
