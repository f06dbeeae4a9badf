"""Where each workgroup of the Gru kernel ran and how long it took (diagnostic launch 11 of csrc/gru_bar16.hip): is a full-chip
launch (B = 1024: 256 workgroups on 256 CUs) slower because the clock drops, or because some CUs are slower?"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
I = n = 96; T = 800
g = torch.Generator(device='cuda'); g.manual_seed(1)
iW = torch.randn(3 * n, I, device='cuda', generator=g) / np.sqrt(I + n)
bb = torch.randn(3 * n, device='cuda', generator=g)
sW = torch.randn(2 * n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
sW2 = torch.randn(n, n, device='cuda', generator=g) / np.sqrt(2 * n) * 2
L.slk_debug_read_bar16_wg.argtypes = [ctypes.c_void_p]
codes = {11: "production", 12: "chain waves without MFMAs", 13: "service waves idle", 14: "no MFMAs at all but the chain's projection share",
         15: "cheap activations", 16: "state in column group 0 only, zeros in the other three (ZC experiment)", 18: "update-gate weights in accumulation registers (ZACC experiment)", 19: "workgroups started out of phase"}
runs = [(1024, c) for c in (11, 16, 11, 16, 11, 16)] + [(768, 11), (256, 11)] + [(1024, c) for c in (12, 13, 14, 15, 11, 16)]
WARM = 20
abc = int(sys.argv[sys.argv.index('--ab') + 1]) if '--ab' in sys.argv and len(sys.argv) > sys.argv.index('--ab') + 1 else 16
if '--ab' in sys.argv: runs = [(1024, c) for c in (11, abc) * 8]
for B, code in runs:
    x = torch.randn(T, B, I, device='cuda', generator=g)
    y = torch.empty(T, B, n, device='cuda')
    f = lambda code: L.slk_gru_bar16_f32(x.data_ptr(), I, iW.data_ptr(), sW.data_ptr(), sW2.data_ptr(), bb.data_ptr(), y.data_ptr(), n, T, B, I, n, 2 * code, 1, 2, None, None, st)
    for _ in range(WARM): assert f(code) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(code); e1.record(); torch.cuda.synchronize()
    out = np.zeros((1024, 4), dtype=np.uint64)
    assert L.slk_debug_read_bar16_wg(out.ctypes.data) == 0
    nwg = (B + 3) // 4
    o = out[:nwg]
    cyc, real = o[:, 0].astype(np.float64), o[:, 1].astype(np.float64)
    hw = o[:, 2]; xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf; hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    start = o[:, 3].astype(np.float64); start -= start.min()
    ms = real / 100e3                                     # 100 MHz wall clock
    mhz = cyc / real * 100.0
    if '--ab' in sys.argv:
        print("%-12s event %.3f ms, clock median %.0f MHz, cycles/step %.0f" % ("production" if code == 11 else codes.get(code, str(code))[:28], e0.elapsed_time(e1), np.median(mhz), np.median(cyc) / T), flush=True)
        continue
    print("[%s]" % codes[code])
    print("B=%d: event %.3f ms; workgroups: %.3f..%.3f ms (median %.3f), shader clock %.0f..%.0f MHz (median %.0f), start spread %.1f us"
          % (B, e0.elapsed_time(e1), ms.min(), ms.max(), np.median(ms), mhz.min(), mhz.max(), np.median(mhz), start.max() / 100.0))
    print("   cycles/step: min %.0f median %.0f max %.0f" % (cyc.min() / T, np.median(cyc) / T, cyc.max() / T))
    for xc in range(8):
        m = xcc == xc
        if m.any(): print("   xcc %d: %3d workgroups, %.3f..%.3f ms, clock %.0f MHz, distinct (se,sh,cu) %d" % (xc, m.sum(), ms[m].min(), ms[m].max(), np.median(mhz[m]), len(set(zip(se[m], sh[m], cu[m])))))
    slow = np.argsort(-ms)[:8]
    print("   slowest:", [(int(i), int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), round(float(ms[i]), 3)) for i in slow])
