"""Differential fuzz of the reverse Gru scan WITH its dL/dx inside (csrc/gru_bwd16.hip, DX / DA: slk_gru_backward16_dx_f32): random width,
input width, T, B, direction, gradient scale (1e-12 .. 1e+4, chunks orders of magnitude apart inside a batch), weight scale, saturated
gates, with and without the layer below's activation derivative.  Asked: da and rh bit for bit those of the plain pass
(slk_gru_backward16_f32), dx within 2e-6 of a chunk's largest |dx| of the float64 product da . iW (times fun'(yref)), nothing written
outside a row's insize floats, the same bits on a second launch.            python tools/fuzz_gru_bwd16_dx.py [cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sloika_amd import _lib  # noqa: E402

L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.RandomState(4242)
g = torch.Generator(device="cuda")
g.manual_seed(23)
ACTS = {None: 0, "tanh": 1, "sigmoid": 2, "relu": 3, "elu": 12}
from sloika_amd import activation  # noqa: E402
ACT_ID = {k: (activation.act_id(getattr(activation, k)) if k else 0) for k in ACTS}
bad, worst = 0, 0.0
for case in range(ncase):
    n = int(rs.choice([16, 32, 48, 64, 80, 96]))
    width = 64 if n <= 64 else 96
    isz = int(rs.choice([16, 32, 48, 64, 80, 96, rs.randint(1, 97)]))
    isz = min(isz, width)
    T = int(rs.randint(1, 90))
    B = int(rs.choice([rs.randint(1, 40), rs.randint(40, 300), rs.randint(1000, 1100)]))
    rev = int(rs.randint(2))
    scale = float(10.0 ** rs.uniform(-12, 4))
    wscale = float(rs.choice([1.0, 2.0, 4.0]))
    dact = [None, "tanh", "sigmoid", "relu", "elu"][int(rs.randint(5))]
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
    if rev:
        dy, zr, hout, hprev = [torch.flip(a, dims=[0]) for a in (dy, zr, hout, hprev)]
    dy, zr, hout, hprev = [a.contiguous() for a in (dy, zr, hout, hprev)]
    sW = wscale * torch.randn(2 * n, n, device="cuda", generator=g) / np.sqrt(2 * n)
    sW2 = wscale * torch.randn(n, n, device="cuda", generator=g) / np.sqrt(2 * n)
    iW = wscale * torch.randn(3 * n, isz, device="cuda", generator=g) / np.sqrt(n + isz)
    yb = torch.tanh(torch.randn(T * B, isz, device="cuda", generator=g)) if dact != "sigmoid" else \
        torch.sigmoid(torch.randn(T * B, isz, device="cuda", generator=g))
    da0 = torch.full((T * B, 3 * n), float("nan"), device="cuda")
    rh0 = torch.full((T * B, n), float("nan"), device="cuda")
    rc = L.slk_gru_backward16_f32(dy.data_ptr(), n, hprev.data_ptr(), n, zr.data_ptr(), hout.data_ptr(), n, sW.data_ptr(), sW2.data_ptr(),
                                  da0.data_ptr(), rh0.data_ptr(), T, B, n, rev, 1, 2, st)
    assert rc == 0
    ldx = isz + int(rs.randint(0, 5))
    outs = []
    for rep in range(2):
        da = torch.full((T * B, 3 * n), float("nan"), device="cuda")
        rh = torch.full((T * B, n), float("nan"), device="cuda")
        dx = torch.full((T * B, ldx), float("nan"), device="cuda")
        rc = L.slk_gru_backward16_dx_f32(dy.data_ptr(), n, hprev.data_ptr(), n, zr.data_ptr(), hout.data_ptr(), n, sW.data_ptr(),
                                         sW2.data_ptr(), iW.data_ptr(), da.data_ptr(), rh.data_ptr(), dx.data_ptr(), ldx, T, B, n, isz, rev,
                                         1, 2, yb.data_ptr() if dact else None, isz, ACT_ID[dact], st)
        assert rc == 0, rc
        outs.append((da, rh, dx))
    da, rh, dx = outs[0]
    ref = da0.double() @ iW.double()
    if dact:
        y = yb.double()
        ref = ref * {"tanh": 1 - y * y, "sigmoid": y * (1 - y), "relu": (y > 0).double(), "elu": torch.where(y > 0, torch.ones_like(y), y + 1)}[dact]
    top = ref.abs().view(T, B, isz).amax(dim=(0, 2), keepdim=True).clamp_min(1e-300)
    d = ((dx[:, :isz].double() - ref).abs().view(T, B, isz) / top).max().item()
    worst = max(worst, d)
    ok = (torch.equal(da, da0) and torch.equal(rh, rh0) and bool(torch.isfinite(dx[:, :isz]).all()) and d < 2e-6
          and bool(torch.isnan(dx[:, isz:]).all()) and all(torch.equal(a, b) or (torch.isnan(a) == torch.isnan(b)).all() and
                                                           torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)) for a, b in zip(outs[0], outs[1])))
    if not ok:
        bad += 1
        print("MISMATCH n=%d insize=%d T=%d B=%d rev=%d scale=%.3g wscale=%g dact=%s: %.3g (da same %s, rh same %s)" % (
            n, isz, T, B, rev, scale, wscale, dact, d, torch.equal(da, da0), torch.equal(rh, rh0)), flush=True)
print("cases %d, mismatches %d, largest dx difference relative to a chunk's largest |dx| %.3g" % (ncase, bad, worst))
