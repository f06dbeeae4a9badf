import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle_np
from sloika_amd import _lib
L = _lib.lib()
try:
    L.slk_gru_fused16_f32
except AttributeError:
    import sys
    sys.exit('slk_gru_fused16_f32 left libsloika_amd.so in round 4 (csrc/gru_fused16.hip -> tools/experiments/): build a library from there (tools/experiments/build_gf16_variants.sh) and load it instead')

s = torch.cuda.current_stream().cuda_stream
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
I, n, T, B, scale, seed = 96, 96, 200, 16, 2.0, 1
rs = np.random.RandomState(seed)
iW0 = (rs.normal(size=(3 * n, I)) / np.sqrt(I + n)).astype(np.float32)
sW0 = (scale * rs.normal(size=(2 * n, n)) / np.sqrt(2 * n)).astype(np.float32)
sW20 = (scale * rs.normal(size=(n, n)) / np.sqrt(2 * n)).astype(np.float32)
b = rs.normal(size=3 * n).astype(np.float32)
x = rs.normal(size=(T, B, I)).astype(np.float32)
for name, iW, sW, sW2 in (("projection only (sW=sW2=0)", iW0, sW0 * 0, sW20 * 0), ("recurrence only (iW=0)", iW0 * 0, sW0, sW20),
                          ("z/r only (sW2=0)", iW0, sW0, sW20 * 0), ("candidate only (sW=0)", iW0, sW0 * 0, sW20)):
    ref = oracle_np.gru(x, iW, sW, sW2, b)
    xd, iWd, sWd, sW2d, bd = dev(x), dev(iW), dev(sW), dev(sW2), dev(b)
    y = torch.full((T, B, n), float("nan"), device="cuda")
    assert L.slk_gru_fused16_f32(xd.data_ptr(), I, iWd.data_ptr(), sWd.data_ptr(), sW2d.data_ptr(), bd.data_ptr(), y.data_ptr(), n, T, B, I, n, 0, 1, 2, None, None, s) == 0
    d = np.abs(y.cpu().numpy() - ref)
    print("%-32s max %.2e p99.9 %.2e mean %.2e  n>5e-6: %d" % (name, d.max(), np.quantile(d, 0.999), d.mean(), (d > 5e-6).sum()))
