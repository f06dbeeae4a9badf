"""A/B timing of gru_fused16 variants built from git history (tools/_build/variants), interleaved rounds in ONE process."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libgf16_variants.so"))
names = sys.argv[1:]
T, B, n, I = 800, 1024, 96, 96
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(T, B, I, device="cuda"); iW = torch.randn(3 * n, I, device="cuda") / np.sqrt(I + n)
bb = torch.randn(3 * n, device="cuda"); sW = torch.randn(2 * n, n, device="cuda") / np.sqrt(2 * n)
sW2 = torch.randn(n, n, device="cuda") / np.sqrt(2 * n); y = torch.empty(T, B, n, device="cuda")
vp, i_, l_ = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
fns = {}
for v in names:
    f = getattr(lib, "slk_gf16_" + v)
    f.argtypes = [vp, l_, vp, vp, vp, vp, vp, l_, i_, i_, i_, i_, i_, i_, i_, vp, vp, vp]
    f.restype = i_
    fns[v] = f
def run(v):
    return fns[v](x.data_ptr(), I, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), bb.data_ptr(), y.data_ptr(), n, T, B, I, n, 0, 1, 2, None, None, st)
res = {v: [] for v in names}
for v in names:
    assert run(v) == 0
torch.cuda.synchronize()
for rnd in range(6):
    for v in names:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(v)
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 10)
for v in names:
    print("%-10s median %.4f  min %.4f ms" % (v, float(np.median(res[v])), min(res[v])))
