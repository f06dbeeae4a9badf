"""Race screen for the kernels added at the end of round 2 (csrc/gru_bar16d.hip, csrc/gru_bar16q.hip, csrc/gru_scan16.hip): many launches of the same
inputs -- full-size batches, ragged lengths, both directions, saved gates -- must all reproduce the first one bit for bit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
g = torch.Generator(device='cuda'); g.manual_seed(11)
for I, n, plan in [(96, 96, 2), (64, 64, 2), (128, 96, 2), (16, 64, 2), (48, 32, 2), (96, 96, 3), (64, 64, 3), (32, 96, 3), (64, 96, 3)]:
    iW = torch.randn(3 * n, I, device='cuda', generator=g) / np.sqrt(I + n)
    bb = torch.randn(3 * n, device='cuda', generator=g)
    sW = 2 * torch.randn(2 * n, n, device='cuda', generator=g) / np.sqrt(2 * n)
    sW2 = 2 * torch.randn(n, n, device='cuda', generator=g) / np.sqrt(2 * n)
    for T, B in [(800, 2048), (333, 2051), (57, 4099)] if plan == 2 else [(800, 4096), (333, 4099), (57, 2051)]:
        x = torch.randn(T, B, I, device='cuda', generator=g)
        lens = torch.randint(1, T + 1, (B,), device='cuda', dtype=torch.int32, generator=g)
        for rev in (0, 1):
            for lp in (None, lens):
                first = None
                for rep in range(reps if T == 800 else max(4, reps // 6)):
                    y = torch.full((T, B, n), float('nan'), device='cuda')
                    zr = torch.full((T * B, 2 * n), float('nan'), device='cuda') if rep % 2 else None
                    assert L.slk_gru_bar16_f32(x.data_ptr(), I, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), bb.data_ptr(), y.data_ptr(), n, T, B, I, n,
                                               rev | (plan << 8), 1, 2, None if lp is None else lp.data_ptr(), None if zr is None else zr.data_ptr(), st) == 0
                    yy = torch.nan_to_num(y, nan=9.0)
                    if first is None: first = yy
                    elif not torch.equal(first, yy):
                        bad += 1; print("plan", plan, "MISMATCH", I, n, T, B, rev, lp is not None, rep, flush=True); break
        print("plan %d: %d->%d T=%d B=%d ok" % (plan, I, n, T, B), flush=True) if not bad else None
for n in (112, 128):
    sW = 2 * torch.randn(2 * n, n, device='cuda', generator=g) / np.sqrt(2 * n)
    sW2 = 2 * torch.randn(n, n, device='cuda', generator=g) / np.sqrt(2 * n)
    for T, B in [(800, 1024), (333, 1021), (57, 2050)]:
        vI = torch.randn(T * B, 3 * n, device='cuda', generator=g)
        lens = torch.randint(1, T + 1, (B,), device='cuda', dtype=torch.int32, generator=g)
        for rev in (0, 1):
            for lp in (None, lens):
                first = None
                for rep in range(reps if (T, B) == (800, 1024) else max(4, reps // 6)):
                    y = torch.full((T, B, n), float('nan'), device='cuda')
                    assert L.slk_gru_scan16_f32(vI.data_ptr(), 3 * n, sW.data_ptr(), sW2.data_ptr(), y.data_ptr(), n, T, B, n, rev, 1, 2,
                                                None if lp is None else lp.data_ptr(), st) == 0
                    yy = torch.nan_to_num(y, nan=9.0)
                    if first is None: first = yy
                    elif not torch.equal(first, yy):
                        bad += 1; print("scan16 MISMATCH", n, T, B, rev, lp is not None, rep, flush=True); break
        print("scan16 n=%d T=%d B=%d ok" % (n, T, B), flush=True) if not bad else None
print("mismatches:", bad)
