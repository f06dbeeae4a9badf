"""In-process timing of the variants built by tools/build_scan16_variants.sh (n = 128, T = 800, B = 1024 and 256)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libscan16_variants.so"))
st = torch.cuda.current_stream().cuda_stream
NV = int(sys.argv[1]); n = 128; T = 800
vp, i_, l_ = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
g = torch.Generator(device='cuda'); g.manual_seed(1)
sW = torch.randn(2 * n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
sW2 = torch.randn(n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
for B in (1024, 256):
    vI = torch.randn(T * B, 3 * n, device='cuda', generator=g)
    y = torch.empty(T, B, n, device='cuda')
    fs = []
    for v in range(NV):
        f = getattr(lib, "slk_s16_v%d" % v); f.argtypes = [vp, l_, vp, vp, vp, l_, i_, i_, i_, i_, i_, i_, vp, vp]; f.restype = i_
        fs.append(f)
    run = lambda f: f(vI.data_ptr(), 3 * n, sW.data_ptr(), sW2.data_ptr(), y.data_ptr(), n, T, B, n, 0, 1, 2, None, st)
    res = [[] for _ in range(NV)]
    for v in range(NV): assert run(fs[v]) == 0
    torch.cuda.synchronize()
    for rnd in range(5):
        for v in range(NV):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); [run(fs[v]) for _ in range(5)]; e1.record(); torch.cuda.synchronize(); res[v].append(e0.elapsed_time(e1) / 5)
    for v in range(NV):
        t = float(np.median(res[v]))
        print("B=%d v%d: %.3f ms (%.0f cycles/step at 2.35 GHz)" % (B, v, t, t * 1e6 / T * 2.35))
