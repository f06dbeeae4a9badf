"""The reverse Gru scan with and without its dL/dx product inside (csrc/gru_bwd16.hip, DX), in one process: time per launch, da / rh
against the plain pass (must be the same bits), dx against da . iW in float64.
    tools/build_bwd16_variants.sh "" "-DGW_DXPOS=0" "-DGW_DXPOS=1"; python tools/bwd16_dx_ab.py"""
import ctypes
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "libbwd16_variants.so"))
T, B, n = 800, int(os.environ.get("B", "1024")), int(os.environ.get("N", "96"))
isz = int(os.environ.get("I", str(n)))
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda")
g.manual_seed(1)
z = torch.sigmoid(torch.randn(T * B, 2 * n, device="cuda", generator=g))
h = torch.tanh(torch.randn((T + 1) * B, n, device="cuda", generator=g)) * 0.5
dy = torch.randn(T * B, n, device="cuda", generator=g) * 1e-3
dy[:, :] *= torch.exp(torch.randn(1, B, 1, device="cuda", generator=g) * 3.0).expand(T, B, 1).reshape(T * B, 1)    # chunks orders of magnitude apart
sW = torch.randn(2 * n, n, device="cuda", generator=g) / np.sqrt(2 * n)
sW2 = torch.randn(n, n, device="cuda", generator=g) / np.sqrt(2 * n)
iW = torch.randn(3 * n, isz, device="cuda", generator=g) / np.sqrt(n + isz)
hout, hprev = h[B:], h[:-B]
vp, l, i = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
plain = V.slk_gw_v0
plain.restype, plain.argtypes = i, [vp, l, vp, l, vp, vp, l, vp, vp, vp, vp, i, i, i, i, i, i, vp]


def run(k, rev):
    da = torch.zeros(T * B, 3 * n, device="cuda")
    rh = torch.zeros(T * B, n, device="cuda")
    dx = torch.full((T * B, isz), float("nan"), device="cuda")
    if k == 0:
        call = lambda: plain(dy.data_ptr(), n, hprev.data_ptr(), n, z.data_ptr(), hout.data_ptr(), n, sW.data_ptr(), sW2.data_ptr(),
                             da.data_ptr(), rh.data_ptr(), T, B, n, rev, 1, 2, st)
    else:
        f = getattr(V, "slk_gwdx_v%d" % k)
        f.restype, f.argtypes = i, [vp, l, vp, l, vp, vp, l, vp, vp, vp, vp, vp, vp, l, i, i, i, i, i, i, i, vp, l, i, vp]
        call = lambda: f(dy.data_ptr(), n, hprev.data_ptr(), n, z.data_ptr(), hout.data_ptr(), n, sW.data_ptr(), sW2.data_ptr(), iW.data_ptr(),
                         da.data_ptr(), rh.data_ptr(), dx.data_ptr(), isz, T, B, n, isz, rev, 1, 2, None, 0, 0, st)
    rc = call()
    assert rc == 0, rc
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    return da, rh, dx, ts


nv = int(os.environ.get("NV", "3"))
for rev in (0, 1):
    base = run(0, rev)
    print("reverse=%d plain: %s ms (%.0f cycles per step at 2.4 GHz)" % (rev, ["%.3f" % t for t in base[3]], min(base[3]) * 1e6 / T * 2.4))
    for k in range(1, nv):
        da, rh, dx, ts = run(k, rev)
        same = torch.equal(da, base[0]) and torch.equal(rh, base[1])
        ref = (da.double() @ iW.double())
        # per chunk: the error against the chunk's largest |dx| (gradients of different chunks are orders of magnitude apart)
        err = (dx.double() - ref).abs().view(T, B, isz).amax(dim=(0, 2)) / ref.abs().view(T, B, isz).amax(dim=(0, 2))
        print("   variant %d: %s ms (%.0f cycles per step)  da, rh same bits: %s   dx: worst chunk %.3g of its largest entry, nan %d" % (
            k, ["%.3f" % t for t in ts], min(ts) * 1e6 / T * 2.4, same, float(err.max()), int(torch.isnan(dx).sum())), flush=True)
