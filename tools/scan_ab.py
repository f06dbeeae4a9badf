"""A/B of two library builds on ONE device for the wide Gru scans (slk_gru_scan16_f32, B = 1024, T' = 800, n = 112 / 128 / 144):
    python tools/scan_ab.py tools/_build/libref_<rev>.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
_lib.require_gpu()
libs = [C.CDLL(sys.argv[1]), C.CDLL(os.environ.get("AB_LIB", _lib.LIB_PATH))]
vp = C.c_void_p
T, B = 800, 1024
for n in (112, 128, 144):
    vI = torch.randn((T, B, 3 * n), device="cuda") * 0.5
    sW = torch.randn((2 * n, n), device="cuda") * 0.1
    sW2 = torch.randn((n, n), device="cuda") * 0.1
    outs, calls = [], []
    for lib in libs:
        f = lib.slk_gru_scan16_f32
        f.argtypes = [vp, C.c_long, vp, vp, vp, C.c_long] + [C.c_int] * 6 + [vp, vp]
        y = torch.empty((T, B, n), device="cuda")
        outs.append(y)
        def call(f=f, y=y):
            assert f(vI.data_ptr(), 3 * n, sW.data_ptr(), sW2.data_ptr(), y.data_ptr(), n, T, B, n, 0, 1, 2, None, None) == 0
        calls.append(call)
    for c in calls: c()
    torch.cuda.synchronize()
    diff = (outs[0] - outs[1]).abs().max().item()
    res = [[], []]
    for rnd in range(7):
        for k, c in enumerate(calls):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): c()
            e1.record(); torch.cuda.synchronize(); res[k].append(e0.elapsed_time(e1) / 5 * 1e3)
    a, b = float(np.median(res[0])), float(np.median(res[1]))
    print("scan n = %d: %.1f -> %.1f us (%+.1f %%), largest difference %.2e" % (n, a, b, (b / a - 1) * 100, diff), flush=True)
