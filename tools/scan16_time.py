"""Launch time of csrc/gru_scan16.hip (and of the fp32 recurrence kernel it replaces) for n = 112 / 128 at T = 800."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
T = 800
for n in (128, 112):
    for B in (1024, 256):
        g = torch.Generator(device='cuda'); g.manual_seed(1)
        sW = torch.randn(2 * n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
        sW2 = torch.randn(n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
        vI = torch.randn(T * B, 3 * n, device='cuda', generator=g)
        y = torch.empty(T, B, n, device='cuda')
        fa = lambda: L.slk_gru_scan16_f32(vI.data_ptr(), 3 * n, sW.data_ptr(), sW2.data_ptr(), y.data_ptr(), n, T, B, n, 0, 1, 2, None, st)
        fb = lambda: L.slk_gru_recurrent_f32(vI.data_ptr(), sW.data_ptr(), sW2.data_ptr(), y.data_ptr(), n, T, B, n, 0, 1, 2, st)
        res = {}
        for name, f in (("scan16", fa), ("fp32 recurrence", fb)):
            assert f() == 0
            torch.cuda.synchronize()
            ts = []
            for rnd in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): f()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 5)
            res[name] = min(ts)
        print("n=%d B=%d: scan16 %.3f ms (%.0f cycles/step at 2.35 GHz)   fp32 recurrence %.3f ms" % (n, B, res["scan16"], res["scan16"] * 1e6 / T * 2.35, res["fp32 recurrence"]), flush=True)
