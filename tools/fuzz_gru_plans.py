"""Differential fuzz of the three Gru plans (four / eight / sixteen chunks per workgroup): random T, B, direction, ragged lengths,
saved gates and input strides for every fused size -- four and eight chunks must agree bit for bit (the same two-MFMA products in the
same order), sixteen (three-term products) within float32 rounding."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.RandomState(2026)
g = torch.Generator(device='cuda'); g.manual_seed(5)
shapes = [(96, 96), (64, 64), (32, 96), (128, 96), (64, 96), (48, 32), (16, 64)]
W = {}
for I, n in shapes:
    W[(I, n)] = (torch.randn(3 * n, I, device='cuda', generator=g) / np.sqrt(I + n), torch.randn(3 * n, device='cuda', generator=g),
                 2 * torch.randn(2 * n, n, device='cuda', generator=g) / np.sqrt(2 * n), 2 * torch.randn(n, n, device='cuda', generator=g) / np.sqrt(2 * n))
bad = 0
for case in range(ncase):
    I, n = shapes[rs.randint(len(shapes))]
    iW, bb, sW, sW2 = W[(I, n)]
    T = int(rs.randint(1, 70)); B = int(rs.choice([rs.randint(1, 40), rs.randint(40, 300), rs.randint(1000, 1100)]))
    rev = int(rs.randint(2)); ragged = rs.rand() < 0.5; save = rs.rand() < 0.5
    ldx = I + 4 * int(rs.randint(0, 3)); ldy = n + 16 * int(rs.randint(0, 2))
    xw = torch.randn(T, B, ldx, device='cuda', generator=g)
    lens = torch.randint(1, T + 1, (B,), device='cuda', dtype=torch.int32, generator=g) if ragged else None
    outs = []
    for plan in (1, 2, 3):
        y = torch.full((T, B, ldy), float('nan'), device='cuda')
        zr = torch.full((T * B, 2 * n), float('nan'), device='cuda') if save else None
        rc = L.slk_gru_bar16_f32(xw.data_ptr(), ldx, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), bb.data_ptr(), y.data_ptr(), ldy, T, B, I, n,
                                 rev | (plan << 8), 1, 2, None if lens is None else lens.data_ptr(), None if zr is None else zr.data_ptr(), st)
        assert rc == 0, rc
        outs.append((torch.nan_to_num(y, nan=9.0), None if zr is None else torch.nan_to_num(zr, nan=9.0)))
    for k in (1, 2):
        if k == 1:
            same = torch.equal(outs[0][0], outs[k][0]) and (outs[0][1] is None or torch.equal(outs[0][1], outs[k][1]))
        else:
            same = (outs[0][0] - outs[k][0]).abs().max().item() < 3e-6 and \
                (outs[0][1] is None or (outs[0][1] - outs[k][1]).abs().max().item() < 3e-6)
        if not same:
            bad += 1
            print("MISMATCH plan", k + 1, "I=%d n=%d T=%d B=%d rev=%d ragged=%s save=%s ldx=%d ldy=%d" % (I, n, T, B, rev, ragged, save, ldx, ldy), flush=True)
print("cases %d, mismatches %d" % (ncase, bad))
