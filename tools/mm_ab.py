"""A/B of two library builds on ONE device for the chunk normalisation (slk_med_mad_normalise_f32, 1024 chunks x 4000 samples):
    python tools/mm_ab.py tools/_build/libref_<rev>.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
_lib.require_gpu()
libs = [C.CDLL(sys.argv[1]), C.CDLL(os.environ.get("MM_LIB", _lib.LIB_PATH))]
vp = C.c_void_p
B, L = int(os.environ.get("MM_B", "1024")), int(os.environ.get("MM_L", "4000"))
x = (torch.randn((B, L), device="cuda") * 12.0 + 90.0).round_() + torch.randn((B, L), device="cuda") * 0.01   # many near-ties, like a quantised signal
outs, calls = [], []
for lib in libs:
    f = lib.slk_med_mad_normalise_f32
    f.argtypes = [vp, C.c_int, C.c_int, vp, C.c_long, C.c_long, vp, vp, vp]
    o = torch.empty_like(x); med = torch.empty(B, device="cuda"); mad = torch.empty(B, device="cuda")
    outs.append((o, med, mad))
    def call(f=f, o=o, med=med, mad=mad):
        assert f(x.data_ptr(), B, L, o.data_ptr(), L, 1, med.data_ptr(), mad.data_ptr(), None) == 0
    calls.append(call)
for c in calls: c()
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
res = [[], []]
for rnd in range(9):
    for k, c in enumerate(calls):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): c()
        e1.record(); torch.cuda.synchronize(); res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
a, b = float(np.median(res[0])), float(np.median(res[1]))
print("normalise %d x %d: %.1f -> %.1f us (%+.1f %%), same results: %s" % (B, L, a, b, (b / a - 1) * 100, same), flush=True)
