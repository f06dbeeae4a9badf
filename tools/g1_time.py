"""Scan-only time of the one-tile-per-wave Gru scan (csrc/gru_scan1t.hip) at n = 96 against the whole-layer kernel gru_bar16_kernel<96,96>
(projection inside): is a layer kernel with six one-tile chain waves worth building?   python tools/g1_time.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
G = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libg1.so"))
vp = C.c_void_p
G.slk_gru_scan1t_launch.argtypes = [vp, C.c_long, vp, vp, vp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
T = 800
for n in (96, 112):
    for B in (1024,):
        g = torch.Generator(device='cuda'); g.manual_seed(1)
        sW = torch.randn(2 * n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
        sW2 = torch.randn(n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
        iW = torch.randn(3 * n, n, device='cuda', generator=g) / np.sqrt(2 * n)
        x = torch.tanh(torch.randn(T, B, n, device='cuda', generator=g))
        vI = torch.randn(T * B, 3 * n, device='cuda', generator=g)
        y = torch.empty(T, B, n, device='cuda')
        fa = lambda: G.slk_gru_scan1t_launch(vI.data_ptr(), 3 * n, sW.data_ptr(), sW2.data_ptr(), y.data_ptr(), n, T, B, n, 0, None, st)
        fs = {"scan1t (scan only)": fa}
        if n == 96:
            fs["bar16 (whole layer)"] = lambda: L.slk_gru_bar16_f32(x.data_ptr(), n, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), None, y.data_ptr(), n, T, B, n, n, 0, 1, 2, None, None, st)
        for name, f in fs.items():
            rc = f(); assert rc == 0, rc
            torch.cuda.synchronize()
            ts = []
            for rnd in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): f()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 5)
            print("n=%d B=%d %-22s %.3f ms" % (n, B, name, min(ts)), flush=True)
