"""Cycles per scan step of the default Gru kernel for every instantiated size (and batch sizes that do not fill the chip)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
entry = ([a for a in sys.argv[1:] if a.startswith("slk_")] or ["slk_gru_bar16_f32"])[0]
shapes = [(96, 96, 800, b) for b in (256, 1024, 512, 768, 1024, 128, 1020, 256)] if "--batch-sweep" in sys.argv else [(96, 96, 800, 1024), (96, 96, 800, 256), (64, 64, 2000, 256), (64, 64, 2000, 1024), (32, 96, 2000, 1024), (128, 96, 2000, 1024),
                   (64, 96, 800, 1024), (48, 32, 800, 1024), (16, 64, 800, 1024)]
for I, n, T, B in shapes:
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    iW = torch.randn(3 * n, I, device='cuda', generator=g) / np.sqrt(I + n)
    bb = torch.randn(3 * n, device='cuda', generator=g)
    sW = torch.randn(2 * n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
    sW2 = torch.randn(n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
    x = torch.randn(T, B, I, device='cuda', generator=g)
    y = torch.empty(T, B, n, device='cuda')
    f = lambda: getattr(L, entry)(x.data_ptr(), I, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), bb.data_ptr(), y.data_ptr(), n, T, B, I, n, 0, 1, 2, None, None, st)
    assert f() == 0
    torch.cuda.synchronize()
    ts = []
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    t = min(ts)
    print("I=%3d n=%3d T=%4d B=%4d: %.3f ms  %.0f cycles/step at 2.4 GHz" % (I, n, T, B, t, t * 1e6 / T * 2.4), flush=True)
