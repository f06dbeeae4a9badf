"""Differential fuzz of the Gru reverse scan on the fp16-split plan (csrc/gru_bwd16.hip) against the float32 kernels: random width,
T, B, direction, gradient scale (1e-12 .. 1e+4, with chunks orders of magnitude apart inside a batch), weight scale, saturated gates.
Agreement is asked per chunk, relative to that chunk's largest gradient.     python tools/fuzz_gru_bwd16.py [cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sloika_amd import _lib  # noqa: E402

L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.RandomState(2029)
g = torch.Generator(device="cuda")
g.manual_seed(11)
bad = 0
worst = 0.0
for case in range(ncase):
    n = int(rs.choice([16, 32, 48, 64, 96, 112, 128]))
    T = int(rs.randint(1, 80))
    B = int(rs.choice([rs.randint(1, 40), rs.randint(40, 300), rs.randint(1000, 1100)]))
    rev = int(rs.randint(2))
    scale = float(10.0 ** rs.uniform(-12, 4))
    wscale = float(rs.choice([1.0, 2.0, 4.0]))
    z = torch.sigmoid(torch.randn(T, B, n, device="cuda", generator=g) * 2)
    r = torch.sigmoid(torch.randn(T, B, n, device="cuda", generator=g) * 2)
    c = torch.tanh(torch.randn(T, B, n, device="cuda", generator=g) * 1.5)
    z[torch.rand(T, B, n, device="cuda", generator=g) < 0.02] = 1.0
    h = torch.zeros(T + 1, B, n, device="cuda")
    for t in range(T):
        h[t + 1] = z[t] * h[t] + (1 - z[t]) * c[t]
    dy = torch.randn(T, B, n, device="cuda", generator=g) * scale * 10.0 ** (-3 * torch.rand(T, B, 1, device="cuda", generator=g))
    dy[torch.rand(T, B, n, device="cuda", generator=g) < 0.3] = 0.0
    dy[:, torch.rand(B, device="cuda", generator=g) < 0.2] *= 1e-6
    zr = torch.cat([z, r], dim=2)
    hout, hprev = h[1:], h[:-1]
    if rev:                                              # the scan walks its own order backwards: reversed layers see time reversed
        dy, zr, hout, hprev = [torch.flip(a, dims=[0]) for a in (dy, zr, hout, hprev)]
    dy, zr, hout, hprev = [a.contiguous() for a in (dy, zr, hout, hprev)]
    sW = wscale * torch.randn(2 * n, n, device="cuda", generator=g) / np.sqrt(2 * n)
    sW2 = wscale * torch.randn(n, n, device="cuda", generator=g) / np.sqrt(2 * n)
    outs = []
    for name in ("slk_gru_backward_f32", "slk_gru_backward16_f32", "slk_gru_backward16_f32"):
        da = torch.full((T * B, 3 * n), float("nan"), device="cuda")
        rh = torch.full((T * B, n), float("nan"), device="cuda")
        rc = getattr(L, name)(dy.data_ptr(), n, hprev.data_ptr(), n, zr.data_ptr(), hout.data_ptr(), n, sW.data_ptr(), sW2.data_ptr(),
                              da.data_ptr(), rh.data_ptr(), T, B, n, rev, 1, 2, st)
        assert rc == 0, (name, rc)
        outs.append((da.reshape(T, B, 3 * n), rh))
    want, got, again = outs
    top = want[0].abs().amax(dim=(0, 2), keepdim=True).clamp_min(1e-35)
    d = ((got[0] - want[0]).abs() / top).max().item()
    worst = max(worst, d)
    ok = bool(torch.isfinite(got[0]).all()) and d < 1e-4 and torch.equal(got[0], again[0]) and torch.equal(got[1], again[1]) and \
        (got[1] - want[1]).abs().max().item() < 1e-6
    if not ok:
        bad += 1
        print("MISMATCH n=%d T=%d B=%d rev=%d scale=%.3g wscale=%g: %.3g" % (n, T, B, rev, scale, wscale, d), flush=True)
print("cases %d, mismatches %d, largest difference relative to a chunk's largest gradient %.3g" % (ncase, bad, worst))
