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
iW = (rs.normal(size=(3 * n, I)) / np.sqrt(I + n)).astype(np.float32)
sW = (scale * rs.normal(size=(2 * n, n)) / np.sqrt(2 * n)).astype(np.float32)
sW2 = (scale * rs.normal(size=(n, n)) / np.sqrt(2 * n)).astype(np.float32)
b = rs.normal(size=3 * n).astype(np.float32)
x = rs.normal(size=(T, B, I)).astype(np.float32)
ref = oracle_np.gru(x, iW, sW, sW2, b)
xd, iWd, sWd, sW2d, bd = dev(x), dev(iW), dev(sW), dev(sW2), dev(b)
outs = []
for rep in range(3):
    y = torch.full((T, B, n), float("nan"), device="cuda")
    assert L.slk_gru_fused16_f32(xd.data_ptr(), I, iWd.data_ptr(), sWd.data_ptr(), sW2d.data_ptr(), bd.data_ptr(), y.data_ptr(), n, T, B, I, n, 0, 1, 2, None, None, s) == 0
    outs.append(y.cpu().numpy())
print("deterministic:", np.array_equal(outs[0], outs[1]), np.array_equal(outs[0], outs[2]))
d = np.abs(outs[0] - ref)
bad = np.argwhere(d > 5e-6)
print("n bad", len(bad))
seen = set()
for t, bb, j in bad:
    if bb in seen: continue
    seen.add(bb)
    row = d[t, bb]
    print("chunk %d first bad step %d: neurons>5e-6: %s  max %.2e ; prev step max %.2e" % (bb, t, np.flatnonzero(row > 5e-6)[:12], row.max(), d[t-1, bb].max() if t else 0))
    # state magnitudes at t-1 for this chunk
    hp = ref[t-1, bb] if t else np.zeros(n)
    print("    min|h_prev| %.2e  count |h_prev|<1e-3: %d" % (np.abs(hp).min(), (np.abs(hp) < 1e-3).sum()))
