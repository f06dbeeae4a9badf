"""A/B of two library builds on ONE device for the Gru projection products of the pretrained architecture through
slk_gemm_bias_act_f16x3 ([819200 x K] . [N x K]^T, (K, N) = (128, 336), (112, 432), (144, 336)):
    python tools/gemm_ab.py tools/_build/libref_<rev>.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
_lib.require_gpu()
libs = [C.CDLL(sys.argv[1]), C.CDLL(os.environ.get("AB_LIB", _lib.LIB_PATH))]
vp = C.c_void_p
M = int(os.environ.get("AB_M", "819200"))
for K, N in ((128, 336), (112, 432), (144, 336), (96, 288)):
    x = torch.randn((M, K), device="cuda")
    W = torch.randn((N, K), device="cuda") * 0.2
    b = torch.randn(N, device="cuda")
    KP = (K + 15) // 16 * 16
    outs, calls = [], []
    for lib in libs:
        hi = torch.empty((N, KP), dtype=torch.float16, device="cuda"); lo = torch.empty_like(hi); inv = torch.empty(N, device="cuda")
        lib.slk_split_f16x2_f32.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp]
        assert lib.slk_split_f16x2_f32(W.data_ptr(), N, K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), None) == 0
        f = lib.slk_gemm_bias_act_f16x3
        f.argtypes = [vp, C.c_long, vp, vp, vp, vp, vp, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, vp]
        y = torch.empty((M, N), device="cuda")
        outs.append(y)
        def call(f=f, y=y, hi=hi, lo=lo, inv=inv):
            assert f(x.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, K, N, 0, None) == 0
        calls.append(call)
    for c in calls: c()
    torch.cuda.synchronize()
    same = torch.equal(outs[0], outs[1])
    res = [[], []]
    for rnd in range(7):
        for k, c in enumerate(calls):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): c()
            e1.record(); torch.cuda.synchronize(); res[k].append(e0.elapsed_time(e1) / 5 * 1e3)
    a, bb = float(np.median(res[0])), float(np.median(res[1]))
    gb = (M * K * 4 + M * N * 4) / 1e9
    print("gemm %d x %d . %d: %.1f -> %.1f us (%+.1f %%), %.2f TB/s, same results: %s" % (M, K, N, a, bb, (bb / a - 1) * 100, gb / bb * 1e3, same), flush=True)
    del outs, calls, x, W
