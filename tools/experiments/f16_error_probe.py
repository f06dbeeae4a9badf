"""Error of the Gru kernels against the float64 numpy restatement (diagnostic; run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as orc, oracle_np
from sloika_amd import _lib
orc.build()
L = _lib.lib()
try:
    L.slk_gru_fused16_f32
except AttributeError:
    import sys
    sys.exit('slk_gru_fused16_f32 left libsloika_amd.so in round 4 (csrc/gru_fused16.hip -> tools/experiments/): build a library from there (tools/experiments/build_gf16_variants.sh) and load it instead')

s = torch.cuda.current_stream().cuda_stream
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
for (I, n, T, B, scale, seed) in [(96, 96, 23, 9, 2.0, 215), (96, 96, 200, 16, 2.0, 1), (96, 96, 800, 16, 1.0, 2), (96, 96, 800, 16, 3.0, 3), (64, 64, 200, 8, 2.0, 4)]:
    rs = np.random.RandomState(seed)
    iW = (rs.normal(size=(3 * n, I)) / np.sqrt(I + n)).astype(np.float32)
    sW = (scale * rs.normal(size=(2 * n, n)) / np.sqrt(2 * n)).astype(np.float32)
    sW2 = (scale * rs.normal(size=(n, n)) / np.sqrt(2 * n)).astype(np.float32)
    b = rs.normal(size=3 * n).astype(np.float32)
    x = rs.normal(size=(T, B, I)).astype(np.float32)
    ref = oracle_np.gru(x, iW, sW, sW2, b)
    c32 = orc.gru(x, iW, sW, sW2, b)
    xd, iWd, sWd, sW2d, bd = dev(x), dev(iW), dev(sW), dev(sW2), dev(b)
    out = {}
    for name in ("fused_f32rec", "fused16"):
        y = torch.full((T, B, n), float("nan"), device="cuda")
        if name == "fused16":
            rc = L.slk_gru_fused16_f32(xd.data_ptr(), I, iWd.data_ptr(), sWd.data_ptr(), sW2d.data_ptr(), bd.data_ptr(), y.data_ptr(), n, T, B, I, n, 0, 1, 2, None, None, s)
        else:
            rc = L.slk_gru_fused_f32(xd.data_ptr(), I, iWd.data_ptr(), sWd.data_ptr(), sW2d.data_ptr(), bd.data_ptr(), y.data_ptr(), n, T, B, I, n, 0, 1, 2, s)
        assert rc == 0
        out[name] = y.cpu().numpy()
    def e(a): 
        d = np.abs(a - ref); return "max %.2e  p99.9 %.2e  mean %.2e" % (d.max(), np.quantile(d, 0.999), d.mean())
    print("I=%d n=%d T=%d B=%d scale=%g" % (I, n, T, B, scale))
    print("   C oracle f32  :", e(c32))
    for k, v in out.items(): print("   %-14s:" % k, e(v))
