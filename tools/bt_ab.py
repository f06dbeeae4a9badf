"""A/B of two library builds on ONE device for the fused decoder's entry point on the two kinds of paths of tools/bt_time.py.
    python tools/bt_ab.py tools/_build/libref_<rev>.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
_lib.require_gpu()
libs = [C.CDLL(sys.argv[1]), C.CDLL(os.environ.get("BT_LIB", _lib.LIB_PATH))]
vp = C.c_void_p
T, B, K, S = int(os.environ.get("BT_T", "800")), int(os.environ.get("BT_B", "1024")), 96, 1025
rs = np.random.RandomState(3)
x = torch.tanh(torch.randn((T, B, K), device="cuda"))
for name, blank_bias in (("random weights (bench)", 0.0), ("blank-dominated (trained-like)", 6.0)):
    W = torch.from_numpy((rs.normal(size=(S, K)) * 0.5).astype(np.float32)).cuda()
    b = torch.from_numpy(rs.normal(size=S).astype(np.float32)); b[0] += blank_bias; b = b.cuda()
    calls, outs = [], []
    for L in libs:
        L.slk_viterbi_kmer_workspace_bytes.restype = C.c_size_t; L.slk_viterbi_kmer_workspace_bytes.argtypes = [C.c_int] * 4
        nws = L.slk_viterbi_kmer_workspace_bytes(T, B, 4, 5)
        ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
        L.slk_softmax_viterbi_pack_bytes.restype = C.c_size_t; L.slk_softmax_viterbi_pack_bytes.argtypes = [C.c_int] * 3
        pack = torch.empty(L.slk_softmax_viterbi_pack_bytes(K, 4, 5), dtype=torch.uint8, device="cuda")
        L.slk_softmax_viterbi_pack_f32.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
        assert L.slk_softmax_viterbi_pack_f32(W.data_ptr(), b.data_ptr(), K, 4, 5, pack.data_ptr(), None) == 0
        f = L.slk_softmax_viterbi_f32
        f.argtypes = [vp, C.c_long, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp, C.c_int, vp, C.c_size_t, vp, vp, vp, vp, vp]
        sc = torch.empty(B, dtype=torch.float32, device="cuda"); pa = torch.full((B, T), -1, dtype=torch.int32, device="cuda"); le = torch.empty(B, dtype=torch.int32, device="cuda")
        outs.append((sc, pa, le))
        def call(f=f, pack=pack, ws=ws, nws=nws, sc=sc, pa=pa, le=le):
            assert f(x.data_ptr(), K, pack.data_ptr(), K, T, B, 4, 5, 0.0, 1e-5, None, 0, ws.data_ptr(), nws, sc.data_ptr(), pa.data_ptr(), le.data_ptr(), None, None) == 0
        calls.append(call)
    for c in calls: c()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b_) for a, b_ in zip(outs[0], outs[1]))
    res = [[], []]
    for rnd in range(7):
        for k, c in enumerate(calls):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): c()
            e1.record(); torch.cuda.synchronize(); res[k].append(e0.elapsed_time(e1) / 5)
    a, b2 = float(np.median(res[0])), float(np.median(res[1]))
    print("%-32s %.3f -> %.3f ms (%+.1f %%), same results: %s" % (name, a, b2, (b2 / a - 1) * 100, same), flush=True)
