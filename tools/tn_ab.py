"""Time slk_gemm_tn_bf16x6_f32 of two builds of the library in one process at the training step's shapes, and compare results.
    python tools/tn_ab.py <lib A> [<lib B>]      (default A: sloika_amd/_build/libsloika_amd.so)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = sys.argv[1:] or [os.path.join(ROOT, "sloika_amd", "_build", "libsloika_amd.so")]
libs = [ctypes.CDLL(p) for p in paths]
vp, L, I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
for lib in libs:
    lib.slk_gemm_tn_workspace_bytes.restype = ctypes.c_size_t
    lib.slk_gemm_tn_workspace_bytes.argtypes = [L, I, I]
    lib.slk_gemm_tn_bf16x6_f32.argtypes = [vp, L, vp, L, vp, L, L, I, I, vp, vp, ctypes.c_size_t, vp]
M = int(os.environ.get("M", "819200"))
st = torch.cuda.current_stream().cuda_stream
for n1, n2, lda, cs in ((288, 96, 288, True), (192, 96, 288, False), (96, 96, 288, False), (1025, 96, 1056, True), (96, 11, 96, True)):
    A = torch.randn(M, lda, device="cuda") * 1e-3
    B = torch.randn(M, n2, device="cuda")
    out = []
    for k, lib in enumerate(libs):
        C = torch.zeros(n1, n2, device="cuda")
        col = torch.zeros(n1, device="cuda")
        nb = lib.slk_gemm_tn_workspace_bytes(M, n1, n2)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        run = lambda: lib.slk_gemm_tn_bf16x6_f32(A.data_ptr(), lda, B.data_ptr(), n2, C.data_ptr(), n2, M, n1, n2,
                                                 col.data_ptr() if cs else None, ws.data_ptr(), nb, st)
        assert run() == 0
        torch.cuda.synchronize()
        ts = []
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        out.append((min(ts), C.clone(), col.clone()))
    ref = (A[:, :n1].double().t() @ B.double())
    msg = "%4d x %3d (lda %4d%s):" % (n1, n2, lda, ", column sums" if cs else "")
    for k, (t, C, col) in enumerate(out):
        err = float((C.double() - ref).abs().max() / ref.abs().max())
        msg += "   %s %.3f ms (%.0f TFLOP/s, error %.1e of the largest entry)" % (os.path.basename(paths[k])[:16], t, 2.0 * M * n1 * n2 / t / 1e9, err)
    if len(out) == 2:
        msg += "   colsum diff %.2e" % float((out[0][2] - out[1][2]).abs().max())
    print(msg, flush=True)

# the three weight gradients of a Gru layer in one launch (slk_gemm_tn_multi_bf16x6_f32), where the builds have it
n = 96
da = torch.randn(M, 3 * n, device="cuda") * 1e-3
xs = [torch.randn(M, n, device="cuda") for _ in range(3)]
shapes = [(3 * n, n, 0), (2 * n, n, 0), (n, n, 2 * n)]
msg = "Gru layer, three problems in one launch:"
for k, lib in enumerate(libs):
    if not hasattr(lib, "slk_gemm_tn_multi_bf16x6_f32"):
        continue
    vps, longs, ints = ctypes.c_void_p * 3, ctypes.c_long * 3, ctypes.c_int * 3
    n1s, n2s = ints(*[s_[0] for s_ in shapes]), ints(*[s_[1] for s_ in shapes])
    lib.slk_gemm_tn_multi_workspace_bytes.restype = ctypes.c_size_t
    lib.slk_gemm_tn_multi_workspace_bytes.argtypes = [L, I, vp, vp]
    lib.slk_gemm_tn_multi_bf16x6_f32.argtypes = [I, vp, vp, vp, vp, vp, vp, L, vp, vp, vp, vp, ctypes.c_size_t, vp]
    nb = lib.slk_gemm_tn_multi_workspace_bytes(M, 3, n1s, n2s)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    Cs = [torch.zeros(a, b, device="cuda") for a, b, _ in shapes]
    col = torch.zeros(3 * n, device="cuda")
    args = (3, vps(*[da.data_ptr() + 4 * s_[2] for s_ in shapes]), longs(3 * n, 3 * n, 3 * n), vps(*[x.data_ptr() for x in xs]), longs(n, n, n),
            vps(*[c.data_ptr() for c in Cs]), longs(n, n, n), M, n1s, n2s, vps(col.data_ptr(), None, None), ws.data_ptr(), nb, st)
    assert lib.slk_gemm_tn_multi_bf16x6_f32(*args) == 0
    torch.cuda.synchronize()
    ts = []
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.slk_gemm_tn_multi_bf16x6_f32(*args)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    msg += "   %s %.3f ms" % (os.path.basename(paths[k])[:16], min(ts))
print(msg, flush=True)
