"""A/B of two library builds on ONE device for the first Convolution of the raw models (slk_conv1d_f32: 1024 chunks x 4000 samples ->
96 features, window 11, stride 5):   python tools/conv_ab.py tools/_build/libref_<rev>.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
_lib.require_gpu()
libs = [C.CDLL(sys.argv[1]), C.CDLL(os.environ.get("AB_LIB", _lib.LIB_PATH))]
vp = C.c_void_p
B, L, F, WL, ST = int(os.environ.get("AB_B", "1024")), 4000, int(os.environ.get("AB_F", "96")), 11, 5
x = torch.randn((B, L), device="cuda")
W = torch.randn((F, 1, WL), device="cuda") * 0.3
bias = torch.randn(F, device="cuda") * 0.1
for act in (int(a) for a in os.environ.get("AB_ACT", "3,1").split(",")):
    outs, calls = [], []
    for lib in libs:
        lib.slk_conv1d_out_len.argtypes = [C.c_int] * 5
        To = lib.slk_conv1d_out_len(L, WL, ST, WL // 2, (WL - 1) // 2)
        f = lib.slk_conv1d_f32
        f.argtypes = [vp, C.c_long, C.c_long, vp, vp, vp] + [C.c_int] * 9 + [vp]
        y = torch.empty((To, B, F), device="cuda")
        outs.append(y)
        def call(f=f, y=y):
            assert f(x.data_ptr(), 1, L, W.data_ptr(), bias.data_ptr(), y.data_ptr(), L, B, 1, F, WL, ST, WL // 2, (WL - 1) // 2, act, None) == 0
        calls.append(call)
    for c in calls: c()
    torch.cuda.synchronize()
    same = torch.equal(outs[0], outs[1])
    res = [[], []]
    for rnd in range(9):
        for k, c in enumerate(calls):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): c()
            e1.record(); torch.cuda.synchronize(); res[k].append(e0.elapsed_time(e1) / 10 * 1e3)
    a, b = float(np.median(res[0])), float(np.median(res[1]))
    print("conv %d x %d -> %d features, activation %d: %.1f -> %.1f us (%+.1f %%), same results: %s" % (B, L, F, act, a, b, (b / a - 1) * 100, same), flush=True)
