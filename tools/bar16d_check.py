"""csrc/gru_bar16d.hip (eight chunks per workgroup; with --quad csrc/gru_bar16q.hip, sixteen) against gru_bar16.hip (four): identical
arithmetic per (neuron, chunk), so the outputs must be bit-identical; then the timing at batches beyond one workgroup per CU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
PLAN = 3 if '--quad' in sys.argv else 2

def run(plan, x, iW, sW, sW2, b, T, B, I, n, rev, lens=None, zr=None, y=None):
    if y is None: y = torch.full((T, B, n), float('nan'), device='cuda')
    rc = L.slk_gru_bar16_f32(x.data_ptr(), I, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), b.data_ptr(), y.data_ptr(), n, T, B, I, n, rev | (plan << 8), 1, 2,
                             None if lens is None else lens.data_ptr(), None if zr is None else zr.data_ptr(), st)
    return rc, y

shapes = [(96, 96), (64, 64), (32, 96), (128, 96), (64, 96), (48, 32), (16, 64)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:] if 'x' in a]
bad = 0
for I, n in shapes:
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    iW = torch.randn(3 * n, I, device='cuda', generator=g) / np.sqrt(I + n)
    bb = torch.randn(3 * n, device='cuda', generator=g)
    sW = torch.randn(2 * n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
    sW2 = torch.randn(n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
    for T, B, rev in [(1, 1, 0), (3, 2, 0), (4, 8, 1), (5, 7, 0), (8, 9, 1), (9, 16, 0), (17, 13, 1), (2, 40, 0), (7, 35, 1), (23, 9, 0), (41, 5, 1), (100, 33, 0), (333, 1021, 1)]:
        x = torch.randn(T, B, I, device='cuda', generator=g)
        for ragged in (False, True, None):
            lens = torch.randint(1, T + 1, (B,), device='cuda', dtype=torch.int32) if ragged else None
            zr_a = torch.full((T * B, 2 * n), float('nan'), device='cuda'); zr_b = zr_a.clone()
            if ragged is None: zr_a = zr_b = None         # the instantiation that does not save the gates
            rc_a, ya = run(1, x, iW, sW, sW2, bb, T, B, I, n, rev, lens, zr_a)
            rc_b, yb = run(PLAN, x, iW, sW, sW2, bb, T, B, I, n, rev, lens, zr_b)
            torch.cuda.synchronize()
            assert rc_a == 0 and rc_b == 0, (rc_a, rc_b)
            same = torch.equal(torch.nan_to_num(ya, nan=7.0), torch.nan_to_num(yb, nan=7.0))
            samez = True if zr_a is None else torch.equal(torch.nan_to_num(zr_a, nan=7.0), torch.nan_to_num(zr_b, nan=7.0))
            if ragged is None and torch.isnan(yb).any(): same = False
            d = (torch.nan_to_num(ya, nan=7.0) - torch.nan_to_num(yb, nan=7.0)).abs().max().item()
            if not (same and samez):
                bad += 1
                print("I=%d n=%d T=%d B=%d rev=%d ragged=%s: y identical %s (max diff %.3g), gates identical %s" % (I, n, T, B, rev, ragged, same, d, samez), flush=True)
    print("I=%d n=%d: compared" % (I, n), flush=True)
    for T, B in [(800, 2048), (800, 1024), (800, 4096)]:
        x = torch.randn(T, B, I, device='cuda', generator=g)
        y = torch.empty(T, B, n, device='cuda')
        def timeit(plan, reps=5):
            f = lambda: run(plan, x, iW, sW, sW2, bb, T, B, I, n, 0, y=y)[0]
            assert f() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): f()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        for rnd in range(2):
            a, b = timeit(1), timeit(PLAN)
            print("I=%d n=%d T=%d B=%d: four-chunk plan %.3f ms   %s-chunk plan %.3f ms" % (I, n, T, B, a, "sixteen" if PLAN == 3 else "eight", b), flush=True)
print("mismatching cases:", bad)
