"""Time the Gru reverse scans (csrc/gru_bwd16.hip against the fp32 kernels) at T' = 800.
    python tools/gru_bwd_time.py [n ...]        (B from the environment, default 1024)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sloika_amd import _lib  # noqa: E402

L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
T, B = 800, int(os.environ.get("B", "1024"))
for n in [int(a) for a in sys.argv[1:]] or [96, 64, 128]:
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    z = torch.sigmoid(torch.randn(T * B, 2 * n, device="cuda", generator=g))
    h = torch.tanh(torch.randn((T + 1) * B, n, device="cuda", generator=g)) * 0.5
    dy = torch.randn(T * B, n, device="cuda", generator=g) * 1e-3
    sW = torch.randn(2 * n, n, device="cuda", generator=g) / np.sqrt(2 * n)
    sW2 = torch.randn(n, n, device="cuda", generator=g) / np.sqrt(2 * n)
    da = torch.empty(T * B, 3 * n, device="cuda")
    rh = torch.empty(T * B, n, device="cuda")
    hout, hprev = h[B:], h[:-B]
    res = {}
    for rnd in range(2):
        for name in ("slk_gru_backward16_f32", "slk_gru_backward_f32"):
            f = lambda: getattr(L, name)(dy.data_ptr(), n, hprev.data_ptr(), n, z.data_ptr(), hout.data_ptr(), n, sW.data_ptr(), sW2.data_ptr(),
                                         da.data_ptr(), rh.data_ptr(), T, B, n, 0, 1, 2, st)
            assert f() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 5)
    print("n=%d T=%d B=%d: fp16-split %s ms (%.0f cycles per step at 2.4 GHz)   fp32 %s ms" % (
        n, T, B, ["%.3f" % v for v in res["slk_gru_backward16_f32"]], min(res["slk_gru_backward16_f32"]) * 1e6 / T * 2.4,
        ["%.3f" % v for v in res["slk_gru_backward_f32"]]), flush=True)
