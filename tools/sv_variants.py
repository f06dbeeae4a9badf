"""Time the variants built by tools/build_sv_variants.sh against each other in ONE process (devices of the pool differ):
    python tools/sv_variants.py name0 name1 ...        (one label per variant, in build order)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from sloika_amd import _lib
    _lib.require_gpu()
    names = sys.argv[1:]
    K = int(os.environ.get("SV_KS", "6")) * 16
    T, B, S = int(os.environ.get("SV_T", "800")), int(os.environ.get("SV_B", "1024")), 1025
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libsv_variants.so"))
    rs = np.random.RandomState(1)
    x = torch.tanh(torch.randn((T, B, K), device="cuda"))
    W = torch.from_numpy((rs.normal(size=(S, K)) * 0.5).astype(np.float32)).cuda()
    b = torch.from_numpy(rs.normal(size=S).astype(np.float32)).cuda()
    nws = _lib.lib().slk_viterbi_kmer_workspace_bytes(T, B, 4, 5)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    sc = torch.empty(B, dtype=torch.float32, device="cuda")
    pa = torch.empty((B, T), dtype=torch.int32, device="cuda")
    le = torch.empty(B, dtype=torch.int32, device="cuda")
    vp = C.c_void_p
    fns = []
    for i, name in enumerate(names):
        pb = getattr(lib, "slk_svpb_v%d" % i)
        pb.restype = C.c_size_t
        n = pb(K, 4, 5)
        pack = torch.empty(n, dtype=torch.uint8, device="cuda")
        pk = getattr(lib, "slk_svp_v%d" % i)
        pk.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
        assert pk(W.data_ptr(), b.data_ptr(), K, 4, 5, pack.data_ptr(), None) == 0
        f = getattr(lib, "slk_sv_v%d" % i)
        f.argtypes = [vp, C.c_long, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp, vp, C.c_size_t,
                      vp, vp, vp, vp, vp]
        fns.append((name, f, pack))
    torch.cuda.synchronize()
    res = {n: [] for n in names}
    for rnd in range(5):
        for name, f, pack in fns:
            def call():
                rc = f(x.data_ptr(), K, pack.data_ptr(), K, T, B, 4, 5, 0.0, 1e-5, None, ws.data_ptr(), nws, sc.data_ptr(),
                       pa.data_ptr(), le.data_ptr(), None, None)
                assert rc == 0, rc
            call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 5)
    for name in names:
        v = sorted(res[name])
        print("%-28s median %.3f ms  (min %.3f, max %.3f)" % (name, v[len(v) // 2], v[0], v[-1]))


if __name__ == "__main__":
    main()
