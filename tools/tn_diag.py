"""Time slk_gemm_tn_f32 alone (diagnostic builds: hipcc -DTN_UNROLL=.. -DTN_WPE=..)."""
import ctypes, sys, os, time
import torch
lib = ctypes.CDLL(sys.argv[1])
M = 819200
vp, L, I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
lib.slk_gemm_tn_workspace_bytes.restype = ctypes.c_size_t
lib.slk_gemm_tn_workspace_bytes.argtypes = [L, I, I]
lib.slk_gemm_tn_f32.argtypes = [vp, L, vp, L, vp, L, L, I, I, vp, vp, ctypes.c_size_t, vp]
for n1, n2 in ((288, 96), (96, 96), (1025, 96)):
    lda = 1056 if n1 == 1025 else n1
    A = torch.rand(M, lda, device="cuda"); B = torch.rand(M, n2, device="cuda"); C = torch.empty(n1, n2, device="cuda")
    nb = lib.slk_gemm_tn_workspace_bytes(M, n1, n2); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    run = lambda: lib.slk_gemm_tn_f32(A.data_ptr(), lda, B.data_ptr(), n2, C.data_ptr(), n2, M, n1, n2, None, ws.data_ptr(), nb, None)
    assert run() == 0
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    print("%s %dx%d: %.3f ms  %.1f TFLOP/s" % (os.path.basename(sys.argv[1]), n1, n2, dt * 1e3, 2.0 * M * n1 * n2 / dt / 1e12))
