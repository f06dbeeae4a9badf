"""In-process timing of the variants built by tools/build_bar16q_variants.sh (I = n = 96, T = 800).  usage: bar16d_variants.py NV [B]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libbar16q_variants.so"))
st = torch.cuda.current_stream().cuda_stream
NV = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
I = n = 96; T = 800
g = torch.Generator(device='cuda'); g.manual_seed(1)
iW = torch.randn(3 * n, I, device='cuda', generator=g) / np.sqrt(I + n)
bb = torch.randn(3 * n, device='cuda', generator=g)
sW = torch.randn(2 * n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
sW2 = torch.randn(n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
x = torch.randn(T, B, I, device='cuda', generator=g)
y = torch.empty(T, B, n, device='cuda')
vp, i_, l_ = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
fs = []
for v in range(NV):
    f = getattr(lib, "slk_q_v%d" % v); f.argtypes = [vp, l_, vp, vp, vp, vp, vp, l_, i_, i_, i_, i_, i_, vp, vp, vp]; f.restype = i_
    fs.append(f)
def run(f): assert f(x.data_ptr(), I, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), bb.data_ptr(), y.data_ptr(), n, T, B, I, n, 0, None, None, st) == 0
ref = None
for v in range(NV):
    run(fs[v]); torch.cuda.synchronize()
    if v == 0: ref = y.clone()
    else: print("v%d identical to v0: %s" % (v, torch.equal(ref, y)))
res = [[] for _ in range(NV)]
for rnd in range(5):
    for v in range(NV):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [run(fs[v]) for _ in range(5)]; e1.record(); torch.cuda.synchronize(); res[v].append(e0.elapsed_time(e1) / 5)
for v in range(NV):
    print("v%d: median %.3f ms  min %.3f ms  (%.0f cycles/step at 2.35 GHz)" % (v, float(np.median(res[v])), min(res[v]), float(np.median(res[v])) * 1e6 / T * 2.35))


