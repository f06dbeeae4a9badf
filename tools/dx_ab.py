"""Time slk_gemm_bias_act_bf16x6 of two builds of the library in one process at the training step's dL/dx shapes.
    python tools/dx_ab.py <lib A> [<lib B>]"""
import ctypes
import os
import sys

import numpy as np
import torch

paths = sys.argv[1:]
libs = [ctypes.CDLL(p) for p in paths]
vp, L, I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
for lib in libs:
    lib.slk_pack_bf16x3_bytes.restype = ctypes.c_size_t
    lib.slk_pack_bf16x3_bytes.argtypes = [I, I]
    lib.slk_pack_bf16x3_f32.argtypes = [vp, I, I, vp, vp]
    lib.slk_gemm_bias_act_bf16x6.argtypes = [vp, L, vp, vp, vp, L, L, I, I, I, vp]
st = torch.cuda.current_stream().cuda_stream
for M, K, N in ((819200, 1056, 96), (819200, 288, 96), (819200, 256, 64), (4000, 1056, 96)):
    x = torch.randn(M, K, device="cuda") * 1e-3
    W = torch.randn(N, K, device="cuda") / np.sqrt(K)
    outs, msg = [], "M=%d K=%d N=%d:" % (M, K, N)
    for k, lib in enumerate(libs):
        y = torch.empty(M, N, device="cuda")
        packed = torch.empty(lib.slk_pack_bf16x3_bytes(N, K), dtype=torch.uint8, device="cuda")
        assert lib.slk_pack_bf16x3_f32(W.data_ptr(), N, K, packed.data_ptr(), st) == 0
        run = lambda: lib.slk_gemm_bias_act_bf16x6(x.data_ptr(), K, packed.data_ptr(), None, y.data_ptr(), N, M, K, N, 0, st)
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
        outs.append(y)
        msg += "   %s %.3f ms (%.0f TFLOP/s)" % (os.path.basename(paths[k])[:16], min(ts), 2.0 * M * K * N / min(ts) / 1e9)
    if len(outs) == 2:
        msg += "   largest difference %.3g (largest entry %.3g)" % (float((outs[0] - outs[1]).abs().max()), float(outs[0].abs().max()))
    print(msg, flush=True)
