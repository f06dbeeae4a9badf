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
    PLAN = int(os.environ.get("SV_PLAN", "0"))
    BS = 8 if (PLAN == 4 or (PLAN == 0 and B >= 1024)) else 16
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
    plans = {}
    for i, name in enumerate(names):
        plans[name] = int(name.split("@")[1]) if "@" in name else PLAN          # "label@4": this variant runs plan 4
        pb = getattr(lib, "slk_svpb_v%d" % i)
        pb.restype = C.c_size_t
        n = pb(K, 4, 5)
        pack = torch.empty(n, dtype=torch.uint8, device="cuda")
        pk = getattr(lib, "slk_svp_v%d" % i)
        pk.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
        assert pk(W.data_ptr(), b.data_ptr(), K, 4, 5, pack.data_ptr(), None) == 0
        f = getattr(lib, "slk_sv_v%d" % i)
        f.argtypes = [vp, C.c_long, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp, C.c_int, vp,
                      C.c_size_t, vp, vp, vp, vp, vp]
        fns.append((name, f, pack))
    torch.cuda.synchronize()
    res = {n: [] for n in names}
    for rnd in range(5):
        for name, f, pack in fns:
            def call():
                rc = f(x.data_ptr(), K, pack.data_ptr(), K, T, B, 4, 5, 0.0, 1e-5, None, plans[name], ws.data_ptr(), nws,
                       sc.data_ptr(), pa.data_ptr(), le.data_ptr(), None, None)
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
    if os.environ.get("SV_DIAG"):
        # variants built with -DSV_DIAG: per-step shader-clock stamps of workgroup 0, wave 0
        dbg = torch.zeros(8192, dtype=torch.int64, device="cuda")
        for name, f, pack in fns:
            if "diag" not in name:          # only builds with -DSV_DIAG treat the dump pointer as a stamp table
                continue
            dbg.zero_()
            for _ in range(3):
                rc = f(x.data_ptr(), K, pack.data_ptr(), K, T, B, 4, 5, 0.0, 1e-5, None, plans[name], ws.data_ptr(), nws,
                       sc.data_ptr(), pa.data_ptr(), le.data_ptr(), dbg.data_ptr(), None)
                assert rc == 0
            torch.cuda.synchronize()
            st = dbg.cpu().numpy().reshape(-1, 16)
            bs = 8 if plans[name] == 4 else 16
            nper = (T + bs - 1) // bs + 1
            flat = st[:nper].reshape(-1).astype(np.int64)
            d = np.diff(flat)
            per = d[16 * 5:16 * (nper - 3)].reshape(-1, 16)      # main-loop periods; column k = duration of step k+1 (k = 15: next step 0)
            med = np.median(per, axis=0)
            print("%-20s steps 1..15,0: %s  | period %d cycles" % (name, " ".join("%4d" % v for v in med), int(med.sum())))
            full = dbg.cpu().numpy()
            s1 = full[:nper * 16].reshape(-1, 16)[6:nper - 3]
            s2 = full[4096:4096 + nper * 16].reshape(-1, 16)[6:nper - 3]
            if os.environ.get("SV_SUB"):
                # round 5: sub-step stamps of steps 14 and 15 (wave 0): top of the step (LDS requests issued), after the vector side
                # work, after the programme's chunks, at the end (J, tail MFMAs), then the barrier
                def md(a):
                    return int(np.median(a))
                for st, base in ((14, 0), (15, 4)):
                    print("    step %d: barrier->top %d, side work %d, chunks D1-D8 %d, J + tail %d, to the barrier stamp %d" % (
                        st, md(s2[:, base] - s1[:, st - 1]), md(s2[:, base + 1] - s2[:, base]), md(s2[:, base + 2] - s2[:, base + 1]),
                        md(s2[:, base + 3] - s2[:, base + 2]), md(s1[:, st] - s2[:, base + 3])))
            elif s2[:, :9].min() > 0:
                def md(a):
                    return int(np.median(a))
                print("    after step 11 -> before flush %d, flush %d, to end of step 12 %d (exp_sum alone %d, from step start %d)" % (
                    md(s2[:, 0] - s1[:, 11]), md(s2[:, 1] - s2[:, 0]), md(s1[:, 12] - s2[:, 1]), md(s2[:, 8] - s2[:, 7]), md(s2[:, 7] - s2[:, 1])))
                print("    after step 15 -> tail start %d, tail %d; step 0: to prepare %d, prepare %d, wload %d, rest of step 0 %d" % (
                    md(s2[:-1, 2] - s1[:-1, 15]), md(s2[:-1, 3] - s2[:-1, 2]), md(s2[1:, 4] - s2[:-1, 3]), md(s2[1:, 5] - s2[1:, 4]),
                    md(s2[1:, 6] - s2[1:, 5]), md(s1[1:, 0] - s2[1:, 6])))
    for name in names:
        v = sorted(res[name])
        print("%-28s median %.3f ms  (min %.3f, max %.3f)" % (name, v[len(v) // 2], v[0], v[-1]))


if __name__ == "__main__":
    main()
