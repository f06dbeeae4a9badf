"""Differential fuzz of the two scan plans for wide Gru layers (csrc/gru_scan16.hip: two tiles per wave; csrc/gru_scan1t.hip: one):
random n in {112, 128} (and narrower multiples of 16 padded inside), T, B, direction, ragged lengths, output stride.  The plan is
chosen per process (SLOIKA_AMD_SCAN1T), so the script runs the cases and writes one line per case -- two random projections and
the largest magnitude of the output -- and is run twice:
    SLOIKA_AMD_SCAN1T=0 python tools/fuzz_gru_scan_plans.py > a.txt; python tools/fuzz_gru_scan_plans.py > b.txt
    python tools/fuzz_gru_scan_plans.py --compare a.txt b.txt"""
import os
import sys

import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--compare":
    a = [l.split() for l in open(sys.argv[2]) if l.startswith("case")]
    b = [l.split() for l in open(sys.argv[3]) if l.startswith("case")]
    assert len(a) == len(b) and len(a) > 0
    bad = 0
    worst = 0.0
    for x, y in zip(a, b):
        assert x[:8] == y[:8], (x, y)
        va, vb = np.array(x[8:], dtype=np.float64), np.array(y[8:], dtype=np.float64)
        d = np.abs(va - vb).max() / max(1.0, np.abs(va).max())
        worst = max(worst, d)
        if not d < 1e-4:
            bad += 1
            print("MISMATCH", " ".join(x[:8]), va, vb)
    print("cases %d, mismatches %d, largest relative difference of the projections %.3g" % (len(a), bad, worst))
    sys.exit(1 if bad else 0)

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sloika_amd import _lib  # noqa: E402

L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(2028)
g = torch.Generator(device="cuda")
g.manual_seed(9)
for case in range(ncase):
    n = int(rs.choice([112, 128]))
    T = int(rs.randint(1, 70))
    B = int(rs.choice([rs.randint(1, 40), rs.randint(40, 300), rs.randint(1000, 1100)]))
    rev = int(rs.randint(2))
    ragged = bool(rs.rand() < 0.5)
    ldy = n + 16 * int(rs.randint(0, 2))
    vI = torch.randn(T * B, 3 * n, device="cuda", generator=g)
    sW = 2 * torch.randn(2 * n, n, device="cuda", generator=g) / np.sqrt(2 * n)
    sW2 = 2 * torch.randn(n, n, device="cuda", generator=g) / np.sqrt(2 * n)
    lens = torch.randint(1, T + 1, (B,), device="cuda", dtype=torch.int32, generator=g) if ragged else None
    y = torch.zeros(T, B, ldy, device="cuda")
    rc = L.slk_gru_scan16_f32(vI.data_ptr(), 3 * n, sW.data_ptr(), sW2.data_ptr(), y.data_ptr(), ldy, T, B, n, rev, 1, 2,
                              None if lens is None else lens.data_ptr(), st)
    assert rc == 0, rc
    if lens is not None:                                 # rows past a chunk's end are unspecified
        mask = (torch.arange(T, device="cuda")[:, None] < lens[None, :]).float()[:, :, None]
        y = y * mask
    p1 = torch.randn(y.numel(), device="cuda", generator=g)
    p2 = torch.randn(y.numel(), device="cuda", generator=g)
    yf = y.reshape(-1).double()
    print("case %d n=%d T=%d B=%d rev=%d ragged=%d ldy=%d %.9e %.9e %.9e" % (case, n, T, B, rev, ragged, ldy, float(yf @ p1.double()) / np.sqrt(y.numel()),
                                                                        float(yf @ p2.double()) / np.sqrt(y.numel()), float(yf.abs().max())), flush=True)
