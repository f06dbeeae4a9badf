import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
iW = (rs.normal(size=(3 * n, I)) / np.sqrt(I + n)).astype(np.float32)
sW = (scale * rs.normal(size=(2 * n, n)) / np.sqrt(2 * n)).astype(np.float32) * 0
sW2 = (scale * rs.normal(size=(n, n)) / np.sqrt(2 * n)).astype(np.float32)
b = rs.normal(size=3 * n).astype(np.float32)
x = rs.normal(size=(T, B, I)).astype(np.float32)
xd, iWd, sWd, sW2d, bd = dev(x), dev(iW), dev(sW), dev(sW2), dev(b)
y = torch.full((T, B, n), float("nan"), device="cuda")
assert L.slk_gru_fused16_f32(xd.data_ptr(), I, iWd.data_ptr(), sWd.data_ptr(), sW2d.data_ptr(), bd.data_ptr(), y.data_ptr(), n, T, B, I, n, 0, 1, 2, None, None, s) == 0
y = y.cpu().numpy().astype(np.float64)
vI = x.astype(np.float64) @ iW.astype(np.float64).T + b
sig = lambda v: 1 / (1 + np.exp(-v))
W2 = sW2.astype(np.float64)
nev = 0
for t in range(1, T):
    hp = y[t - 1]                                   # the GPU's own previous state: one-step residuals only
    z = sig(vI[t, :, :n]); r = sig(vI[t, :, n:2 * n])
    rh = r * hp
    pre = vI[t, :, 2 * n:] + rh @ W2.T
    hn = z * hp + (1 - z) * np.tanh(pre)
    d = y[t] - hn                                    # [B, n]
    for bb in range(B):
        if np.abs(d[bb]).max() > 3e-6:
            # residual in pre-activation space
            dpre = d[bb] / ((1 - z[bb]) * (1 - np.tanh(pre[bb]) ** 2))
            # which single k explains it: dpre ~ W2[:, k] * delta
            best = None
            for k in range(n):
                col = W2[:, k]
                delta = (col @ dpre) / (col @ col)
                res = np.linalg.norm(dpre - col * delta) / np.linalg.norm(dpre)
                if best is None or res < best[0]:
                    best = (res, k, delta)
            res, k, delta = best
            v = rh[bb, k]
            hi = np.float64(np.float16(v)); lo = np.float64(np.float16(v - hi))
            print("t=%d chunk=%d max|d|=%.2e  best k=%d (unexplained %.2f) delta=%.3e  rh_k=%.6e hi=%.6e lo=%.3e  delta/rh=%.3e delta/lo=%.3f" % (
                t, bb, np.abs(d[bb]).max(), k, res, delta, v, hi, lo, delta / v, delta / lo if lo else np.nan))
            nev += 1
            if nev > 25: sys.exit()
