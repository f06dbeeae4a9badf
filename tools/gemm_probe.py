"""Softmax projection kernel alone: time and HBM-write rate (B=1024, T=800 rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
M, K, N, ld = 819200, 96, 1025, 1056
x = torch.tanh(torch.randn(M, K, device="cuda")); W = torch.randn(N, K, device="cuda") * 0.5; b = torch.randn(N, device="cuda")
hi = torch.empty(N, K, dtype=torch.float16, device="cuda"); lo = torch.empty_like(hi); inv = torch.empty(N, device="cuda")
assert L.slk_split_f16x2_f32(W.data_ptr(), N, K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), st) == 0
y = torch.empty(M, ld, device="cuda"); stats = torch.empty(M, 2, device="cuda")
def run(): assert L.slk_linear_rowstats_f16x3(x.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), b.data_ptr(), y.data_ptr(), ld, M, K, N, stats.data_ptr(), st) == 0
run(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [run() for _ in range(5)]; e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
print("gemm_rows_f16x3 M=%d: median %.3f ms min %.3f ms (%.2f TB/s written)" % (M, float(np.median(ts)), min(ts), M * N * 4 / min(ts) / 1e9))
ref = (x[:256].double() @ W.double().T + b.double()).float()
print("max err rows 0..255:", float((y[:256, :N] - ref).abs().max()))
