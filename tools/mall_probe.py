"""How much do softmax projection + Viterbi forward gain when the logits of a T-block stay in the Infinity Cache?
(estimate before building the blocked pipeline: GEMM and decoder called per 32-step block on one reused buffer)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
T, B, K, N, ld = 800, 1024, 96, 1025, 1056
x = torch.tanh(torch.randn(T * B, K, device="cuda"))
W = torch.randn(N, K, device="cuda") * 0.5
b = torch.randn(N, device="cuda")
hi = torch.empty(N, K, dtype=torch.float16, device="cuda"); lo = torch.empty_like(hi); inv = torch.empty(N, device="cuda")
assert L.slk_split_f16x2_f32(W.data_ptr(), N, K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), st) == 0
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for TB in (800, 64, 32, 16):
    nblk = T // TB
    rows = TB * B
    y = torch.empty(rows, ld, device="cuda"); stats = torch.empty(rows, 2, device="cuda")
    nb = L.slk_viterbi_kmer_workspace_bytes(TB, B, 4, 5)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    sc = torch.empty(B, device="cuda"); path = torch.empty(B, TB, dtype=torch.int32, device="cuda"); ln = torch.empty(B, dtype=torch.int32, device="cuda")
    def gemm(i):
        assert L.slk_linear_rowstats_f16x3(x.data_ptr() + i * rows * K * 4, K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), b.data_ptr(), y.data_ptr(), ld, rows, K, N, stats.data_ptr(), st) == 0
    def vit():
        assert L.slk_viterbi_kmer_logits_f32(y.data_ptr(), ld, stats.data_ptr(), TB, B, 4, 5, 0.0, 1e-5, ws.data_ptr(), nb, sc.data_ptr(), path.data_ptr(), ln.data_ptr(), st) == 0
    tg = timeit(lambda: [gemm(i) for i in range(nblk)])
    tv = timeit(lambda: [vit() for i in range(nblk)])
    tb = timeit(lambda: [(gemm(i), vit()) for i in range(nblk)])
    print("T-block %3d (%4.0f MB of logits): gemm %.3f ms  viterbi(+backtrace) %.3f ms  alternating %.3f ms" % (TB, rows * ld * 4 / 1e6, tg, tv, tb))
