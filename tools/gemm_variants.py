"""A/B of the softmax projection's store path in ONE process (tools/_build/libgemm_variants.so: v0 direct stores, v1 through LDS)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libgemm_variants.so"))
st = torch.cuda.current_stream().cuda_stream
M, K, N, ld = 819200, 96, 1025, 1056
x = torch.tanh(torch.randn(M, K, device="cuda")); W = torch.randn(N, K, device="cuda") * 0.5; b = torch.randn(N, device="cuda")
hi = torch.empty(N, K, dtype=torch.float16, device="cuda"); lo = torch.empty_like(hi); inv = torch.empty(N, device="cuda")
y = torch.empty(M, ld, device="cuda"); stats = torch.empty(M, 2, device="cuda")
vp, i_, l_ = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
res = {}
NV = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for v in range(NV):
    sp = getattr(lib, "slk_sp_v%d" % v); sp.argtypes = [vp, i_, i_, vp, vp, vp, vp]; sp.restype = i_
    assert sp(W.data_ptr(), N, K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), st) == 0
    f = getattr(lib, "slk_lr_v%d" % v); f.argtypes = [vp, l_, vp, vp, vp, vp, vp, l_, l_, i_, i_, vp, vp]; f.restype = i_
    res[v] = (f, [])
def run(f): assert f(x.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), b.data_ptr(), y.data_ptr(), ld, M, K, N, stats.data_ptr(), st) == 0
for v in range(NV): run(res[v][0])
torch.cuda.synchronize()
for rnd in range(6):
    for v in range(NV):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [run(res[v][0]) for _ in range(5)]; e1.record(); torch.cuda.synchronize(); res[v][1].append(e0.elapsed_time(e1) / 5)
for v in range(NV):
    print("v%d: median %.3f ms  min %.3f ms" % (v, float(np.median(res[v][1])), min(res[v][1])))
