"""Time the variants of csrc/gru_bwd16.hip built by tools/build_bwd16_variants.sh in one process (T' = 800; B, N from the environment)
and compare each one's results with variant 0's.     python tools/bwd16_variants.py <number of variants>"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "libbwd16_variants.so"))
nv = int(sys.argv[1])
T, B, n = 800, int(os.environ.get("B", "1024")), int(os.environ.get("N", "96"))
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda")
g.manual_seed(1)
z = torch.sigmoid(torch.randn(T * B, 2 * n, device="cuda", generator=g))
h = torch.tanh(torch.randn((T + 1) * B, n, device="cuda", generator=g)) * 0.5
dy = torch.randn(T * B, n, device="cuda", generator=g) * 1e-3
sW = torch.randn(2 * n, n, device="cuda", generator=g) / np.sqrt(2 * n)
sW2 = torch.randn(n, n, device="cuda", generator=g) / np.sqrt(2 * n)
hout, hprev = h[B:], h[:-B]
vp, l, i = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
outs, times = [], {}
for rnd in range(6):
    for k in range(nv):
        f = getattr(V, "slk_gw_v%d" % k)
        f.restype, f.argtypes = i, [vp, l, vp, l, vp, vp, l, vp, vp, vp, vp, i, i, i, i, i, i, vp]
        da = torch.zeros(T * B, 3 * n, device="cuda")
        rh = torch.zeros(T * B, n, device="cuda")
        call = lambda: f(dy.data_ptr(), n, hprev.data_ptr(), n, z.data_ptr(), hout.data_ptr(), n, sW.data_ptr(), sW2.data_ptr(), da.data_ptr(),
                         rh.data_ptr(), T, B, n, 0, 1, 2, st)
        assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        times.setdefault(k, []).append(e0.elapsed_time(e1) / 20)
        if rnd == 0:
            outs.append((da, rh))
for k in range(nv):
    d = max(float((outs[k][0] - outs[0][0]).abs().max()), float((outs[k][1] - outs[0][1]).abs().max()))
    print("variant %d: %s ms  (%.0f cycles per step at 2.4 GHz)  max |difference to variant 0| %.3g  (max |da| %.3g)" % (
        k, ["%.3f" % t for t in times[k]], min(times[k]) * 1e6 / T * 2.4, d, float(outs[0][0].abs().max())), flush=True)
