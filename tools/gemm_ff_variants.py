"""A/B of the row GEMM's workgroup height for FeedForward shapes (few output columns: start-up per workgroup dominates) in one
process (tools/build_gemm_variants.sh "" "-DGH_NWM=2" ...).   usage: gemm_ff_variants.py NV"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libgemm_variants.so"))
st = torch.cuda.current_stream().cuda_stream
NV = int(sys.argv[1]) if len(sys.argv) > 1 else 2
vp, i_, l_ = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
SHAPES = [(2000 * 1024, 192, 128), (2000 * 256, 128, 64), (800 * 1024, 288, 96)]
if os.environ.get("SHAPES"):          # e.g. SHAPES=819200:128:336,819200:112:432  (the Gru projections of models/pretrained.pkl)
    SHAPES = [tuple(int(v) for v in t.split(":")) for t in os.environ["SHAPES"].split(",")]
for M, K, N in SHAPES:
    x = torch.tanh(torch.randn(M, K, device="cuda")); W = torch.randn(N, K, device="cuda") * 0.5; b = torch.randn(N, device="cuda")
    KP = (K + 15) // 16 * 16
    hi = torch.empty(N, KP, dtype=torch.float16, device="cuda"); lo = torch.empty_like(hi); inv = torch.empty(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    res = {}
    ref = None
    for v in range(NV):
        sp = getattr(lib, "slk_sp_v%d" % v); sp.argtypes = [vp, i_, i_, vp, vp, vp, vp]; sp.restype = i_
        assert sp(W.data_ptr(), N, K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), st) == 0
        f = getattr(lib, "slk_gb_v%d" % v); f.argtypes = [vp, l_, vp, vp, vp, vp, vp, l_, l_, i_, i_, i_, vp]; f.restype = i_
        rc = f(x.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, K, N, 1, st)
        torch.cuda.synchronize()
        if rc != 0: print("v%d: rc %d for K=%d N=%d" % (v, rc, K, N)); res[v] = None; continue
        if ref is None: ref = y.clone()
        else: print("v%d identical: %s" % (v, torch.equal(ref, y)))
        res[v] = (f, [])
    for rnd in range(5):
        for v in range(NV):
            if res[v] is None: continue
            f = res[v][0]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f(x.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, K, N, 1, st)
            e1.record(); torch.cuda.synchronize(); res[v][1].append(e0.elapsed_time(e1) / 5)
    gb = (M * K * 4 + M * N * 4) / 1e9
    for v in range(NV):
        if res[v] is None: continue
        t = float(np.median(res[v][1]))
        print("M=%d K=%d N=%d v%d: %.3f ms  (%.2f TB/s of x read + y written)" % (M, K, N, v, t, gb / t))
