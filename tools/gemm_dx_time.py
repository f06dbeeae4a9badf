"""Time the long-row GEMMs of the training step (dL/dx products): csrc/gemm_bf16x6.hip against the fp32 MFMA kernel.
    python tools/gemm_dx_time.py [M:K:N ...]      (default: the shapes of raw_0.98_rgrgr at batch 1024)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sloika_amd import _lib  # noqa: E402

L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(819200, 1040, 96), (819200, 288, 96), (819200, 256, 64)]
for M, K, N in shapes:
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / np.sqrt(K)
    y = torch.empty(M, N, device="cuda")
    y2 = torch.empty(M, N, device="cuda")
    packed = torch.empty(L.slk_pack_bf16x3_bytes(N, K), dtype=torch.uint8, device="cuda")

    def new():
        assert L.slk_pack_bf16x3_f32(W.data_ptr(), N, K, packed.data_ptr(), st) == 0
        assert L.slk_gemm_bias_act_bf16x6(x.data_ptr(), K, packed.data_ptr(), None, y.data_ptr(), N, M, K, N, 0, st) == 0

    def old():
        assert L.slk_gemm_bias_act_f32(x.data_ptr(), K, W.data_ptr(), None, y2.data_ptr(), N, M, K, N, 0, st) == 0

    res = {}
    for name, f in (("bf16x6", new), ("fp32", old), ("bf16x6", new), ("fp32", old)):
        f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) / 5)
    gb = 4.0 * M * (K + N) / 1e9
    ref = x[:2048].double() @ W.double().t()
    print("M=%d K=%d N=%d: bf16x6 %s ms (%.2f TB/s)  fp32 %s ms   max |diff| vs float64 on 2048 rows: %.3g / %.3g" % (
        M, K, N, ["%.3f" % v for v in res["bf16x6"]], gb / min(res["bf16x6"]), ["%.3f" % v for v in res["fp32"]],
        (y[:2048].double() - ref).abs().max().item(), (y2[:2048].double() - ref).abs().max().item()), flush=True)
