"""A/B of two library builds on ONE device, one process: per-launch time of slk_gru_bar16_f32 (whole Gru layer, T' = 800) for the
four- / eight- / sixteen-chunk plans, interleaved rounds, and the largest difference between the two builds' results.
    python tools/gru_ab.py tools/_build/libref_<rev>.so [sloika_amd/_build/libsloika_amd.so] [IxN ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from sloika_amd import _lib
    _lib.require_gpu()
    paths = [a for a in sys.argv[1:] if a.endswith(".so")]
    if len(paths) == 1:
        paths.append(_lib.LIB_PATH)
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if "x" in a and not a.endswith(".so")] or [(96, 96)]
    libs = [C.CDLL(p) for p in paths]
    vp = C.c_void_p
    for L in libs:
        L.slk_gru_bar16_f32.argtypes = [vp, C.c_long, vp, vp, vp, vp, vp, C.c_long] + [C.c_int] * 7 + [vp, vp, vp]
        L.slk_gru_bar16_f32.restype = C.c_int
    st = torch.cuda.current_stream().cuda_stream
    T = 800
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    for I, n in shapes:
        iW = torch.randn(3 * n, I, device="cuda", generator=g) / np.sqrt(I + n)
        bb = torch.randn(3 * n, device="cuda", generator=g)
        sW = 2 * torch.randn(2 * n, n, device="cuda", generator=g) / np.sqrt(2 * n)
        sW2 = 2 * torch.randn(n, n, device="cuda", generator=g) / np.sqrt(2 * n)
        for plan, B in ((1, 1024), (2, 2048), (3, 4096), (2, 1024), (3, 1024)):
            x = torch.randn(T, B, I, device="cuda", generator=g)
            ys = [torch.empty(T, B, n, device="cuda") for _ in libs]

            def run(k, rev=0):
                rc = libs[k].slk_gru_bar16_f32(x.data_ptr(), I, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), bb.data_ptr(), ys[k].data_ptr(),
                                               n, T, B, I, n, rev | (plan << 8), 1, 2, None, None, st)
                assert rc == 0, rc
            for k in range(len(libs)):
                run(k)
            torch.cuda.synchronize()
            diff = (ys[0] - ys[1]).abs().max().item()
            res = [[] for _ in libs]
            for rnd in range(5):
                for k in range(len(libs)):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for r in range(6):
                        run(k, r & 1)
                    e1.record()
                    torch.cuda.synchronize()
                    res[k].append(e0.elapsed_time(e1) / 6)
            a, b = float(np.median(res[0])), float(np.median(res[1]))
            print("%d->%d plan %d (%2d chunks per workgroup) B=%4d: %.3f -> %.3f ms per launch (%+.1f %%), largest difference %.3g"
                  % (I, n, plan, 4 << (plan - 1), B, a, b, (b / a - 1) * 100, diff), flush=True)


if __name__ == "__main__":
    main()
