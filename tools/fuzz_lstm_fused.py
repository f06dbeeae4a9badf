"""Differential fuzz of the fused Lstm layer kernel (csrc/lstm_fused16.hip) against projection GEMM + csrc/lstm_scan16.hip: random
sizes, input widths, T, B, direction, ragged lengths, strides, with and without bias / peepholes.  The two paths share nothing but
the arithmetic scheme (fp16-split products), so they must agree to float32 rounding of the products (2e-5 on outputs in [-1, 1]);
every case also repeats bit for bit.     python tools/fuzz_lstm_fused.py [cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sloika_amd import _lib  # noqa: E402

L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.RandomState(2027)
g = torch.Generator(device="cuda")
g.manual_seed(7)
bad = 0
worst = 0.0
for case in range(ncase):
    n = 16 * int(rs.randint(1, 5))
    I = 4 * int(rs.randint(1, 17))
    T = int(rs.randint(1, 70))
    B = int(rs.choice([rs.randint(1, 40), rs.randint(40, 300), rs.randint(1000, 1100)]))
    rev = int(rs.randint(2))
    ragged = rs.rand() < 0.5
    bias = rs.rand() < 0.7
    peep = rs.rand() < 0.7
    ldx = I + 4 * int(rs.randint(0, 3))
    ldy = n + 16 * int(rs.randint(0, 2))
    iW = torch.randn(4 * n, I, device="cuda", generator=g) / np.sqrt(I + n)
    sW = 2 * torch.randn(4 * n, n, device="cuda", generator=g) / np.sqrt(2 * n)
    b = torch.randn(4 * n, device="cuda", generator=g) if bias else None
    p = torch.randn(3, n, device="cuda", generator=g) / np.sqrt(n) if peep else None
    xw = torch.randn(T, B, ldx, device="cuda", generator=g) * float(10.0 ** rs.uniform(-2, 1))
    lens = torch.randint(1, T + 1, (B,), device="cuda", dtype=torch.int32, generator=g) if ragged else None
    outs = []
    for rep in range(2):
        y = torch.full((T, B, ldy), float("nan"), device="cuda")
        rc = L.slk_lstm_fused16_f32(xw.data_ptr(), ldx, iW.data_ptr(), sW.data_ptr(), None if b is None else b.data_ptr(),
                                    None if p is None else p.data_ptr(), y.data_ptr(), ldy, T, B, I, n, rev, 1, 2,
                                    None if lens is None else lens.data_ptr(), st)
        assert rc == 0, rc
        outs.append(torch.nan_to_num(y, nan=9.0))
    # the two-kernel path: vW = x iW^T + b (fp32 MFMA), then the scan
    vW = torch.empty(T * B, 4 * n, device="cuda")
    zb = torch.zeros(4 * n, device="cuda") if b is None else b
    assert L.slk_gemm_bias_act_f32(xw.data_ptr(), ldx, iW.data_ptr(), zb.data_ptr(), vW.data_ptr(), 4 * n, T * B, I, 4 * n, 0, st) == 0
    y2 = torch.full((T, B, ldy), float("nan"), device="cuda")
    assert L.slk_lstm_scan16_f32(vW.data_ptr(), sW.data_ptr(), None if p is None else p.data_ptr(), y2.data_ptr(), ldy, T, B, n, rev, 1, 2,
                                 None if lens is None else lens.data_ptr(), st) == 0
    y2 = torch.nan_to_num(y2, nan=9.0)
    d = (outs[0] - y2).abs().max().item()
    worst = max(worst, d)
    if not torch.equal(outs[0], outs[1]) or not d < 2e-5:
        bad += 1
        print("MISMATCH n=%d I=%d T=%d B=%d rev=%d ragged=%s bias=%s peep=%s ldx=%d ldy=%d: repeat %s, diff %.3g" % (
            n, I, T, B, rev, ragged, bias, peep, ldx, ldy, torch.equal(outs[0], outs[1]), d), flush=True)
print("cases %d, mismatches %d, largest difference %.3g" % (ncase, bad, worst))
