"""Time slk_gru_backward_f32 alone (diagnostic builds: hipcc -DGBWD_DIAG=n)."""
import ctypes, sys, os, time
import numpy as np, torch
lib = ctypes.CDLL(sys.argv[1])
B = int(sys.argv[2]); T, n, I = 800, 96, 96
M = T * B
g = lambda *s: torch.rand(*s, device="cuda")
dy, xh, zr, h, sW, sW2 = g(M, n), g(M, I + n), g(M, 2 * n), g(M, n), g(2 * n, n) * 0.1, g(n, n) * 0.1
da = torch.empty(M, 3 * n, device="cuda")
rh = torch.empty(M, n, device="cuda")
vp = ctypes.c_void_p
f = lib.slk_gru_backward_f32
f.argtypes = [vp, ctypes.c_long, vp, ctypes.c_long, vp, vp, ctypes.c_long, vp, vp, vp, vp] + [ctypes.c_int] * 6 + [vp]
def run():
    return f(dy.data_ptr(), n, xh.data_ptr() + 4 * I, I + n, zr.data_ptr(), h.data_ptr(), n, sW.data_ptr(), sW2.data_ptr(), da.data_ptr(), rh.data_ptr(), T, B, n, 0, 1, 2, None)
assert run() == 0
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10): run()
torch.cuda.synchronize()
print("%s B=%d: %.3f ms" % (os.path.basename(sys.argv[1]), B, (time.time() - t0) * 100))
