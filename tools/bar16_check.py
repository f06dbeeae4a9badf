"""csrc/gru_bar16.hip: in-process timing, the ablation launches and the per-section stamps of the diagnostic build
    tools/build_diag_lib.sh && SLOIKA_AMD_LIB=$PWD/tools/_build/libsloika_amd_diag.so python tools/bar16_check.py [IxN ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np, torch
from sloika_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream

def run(entry, x, iW, sW, sW2, b, T, B, I, n, rev, lens=None, zr=None):
    y = torch.full((T, B, n), float('nan'), device='cuda')
    rc = getattr(L, entry)(x.data_ptr(), I, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), b.data_ptr(), y.data_ptr(), n, T, B, I, n, rev, 1, 2,
                           None if lens is None else lens.data_ptr(), None if zr is None else zr.data_ptr(), st)
    torch.cuda.synchronize()
    return rc, y

shapes = [(96, 96)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for I, n in shapes:
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    iW = torch.randn(3 * n, I, device='cuda', generator=g) / np.sqrt(I + n)
    bb = torch.randn(3 * n, device='cuda', generator=g)
    sW = torch.randn(2 * n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
    sW2 = torch.randn(n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
    T, B = 800, 1024
    x = torch.randn(T, B, I, device='cuda', generator=g)
    y = torch.empty(T, B, n, device='cuda')
    def timeit(entry, reps=10):
        f = lambda: getattr(L, entry)(x.data_ptr(), I, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), bb.data_ptr(), y.data_ptr(), n, T, B, I, n, 0, 1, 2, None, None, st)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    for rnd in range(3):
        b = timeit('slk_gru_bar16_f32')
        print("I=%d n=%d T=800 B=1024: bar16 %.3f ms  (%.0f cycles/step at 2.4 GHz)" % (I, n, b, b * 1e6 / T * 2.4), flush=True)

    def timeit_code(code, reps=10):
        f = lambda: L.slk_gru_bar16_f32(x.data_ptr(), I, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), bb.data_ptr(), y.data_ptr(), n, T, B, I, n, 2 * code, 1, 2, None, None, st)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    base = timeit_code(0)
    print("ablations (results garbage), cycles/step at 2.4 GHz; production %.0f" % (base * 1e6 / T * 2.4))
    for code, nm in ((2, "no s_barrier"), (3, "chain: no MFMAs"), (4, "cheap activations"), (5, "service waves idle"), (6, "no h_out stores"),
                     (7, "no s_barrier + service idle"), (8, "no s_barrier + no chain MFMAs"), (9, "no barrier, no chain MFMAs, service idle"), (10, "everything off")):
        ms = timeit_code(code)
        print("   %-44s %6.0f" % (nm, ms * 1e6 / T * 2.4))
    import ctypes
    L.slk_debug_read_bar16.argtypes = [ctypes.c_void_p]
    L.slk_gru_bar16_f32(x.data_ptr(), I, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), bb.data_ptr(), y.data_ptr(), n, T, B, I, n, 2, 1, 2, None, None, st)
    torch.cuda.synchronize()
    stp = (ctypes.c_ulonglong * 64)()
    L.slk_debug_read_bar16(stp)
    names = ["barrier A", "reads + projection MFMAs + wait h", "r MFMAs", "z MFMAs + r epilogue + write", "barrier B", "reads + last z block + wait r*h",
             "c MFMAs + z epilogue", "tanh, blend, split, writes, stores"]
    for wv in range(4):
        v = [stp[wv * 16 + i] / T for i in range(16)]
        if wv < n // 32:
            print("chain wave %d: cycles per step, total %.0f" % (wv, sum(v[:8])))
            for i, nm in enumerate(names):
                print("   %-44s %7.0f" % (nm, v[i]))
        else:
            print("service wave %d: cycles per GROUP: work per interval %s   barrier wait per interval %s" % (
                wv, " ".join("%5.0f" % (4 * a) for a in v[:8]), " ".join("%5.0f" % (4 * a) for a in v[8:])))
