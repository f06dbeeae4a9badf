"""Time of the fused decoder's backtrace (csrc/decode.hip, viterbi_backtrace_kernel FMT 2) on two kinds of paths: the bench's random
weights (a move at almost every step) and a blank-dominated posterior as trained models produce (long runs of stays).
    python tools/bt_time.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib
_lib.require_gpu()
L = _lib.lib()
T, B, K, S = 800, 1024, 96, 1025
rs = np.random.RandomState(3)
x = torch.tanh(torch.randn((T, B, K), device="cuda"))
nws = L.slk_viterbi_kmer_workspace_bytes(T, B, 4, 5)
ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
sc = torch.empty(B, dtype=torch.float32, device="cuda"); pa = torch.empty((B, T), dtype=torch.int32, device="cuda"); le = torch.empty(B, dtype=torch.int32, device="cuda")
for name, blank_bias in (("random weights (bench)", 0.0), ("blank-dominated (trained-like)", 6.0)):
    W = torch.from_numpy((rs.normal(size=(S, K)) * 0.5).astype(np.float32)).cuda()
    b = torch.from_numpy(rs.normal(size=S).astype(np.float32)); b[0] += blank_bias; b = b.cuda()
    pack = torch.empty(L.slk_softmax_viterbi_pack_bytes(K, 4, 5), dtype=torch.uint8, device="cuda")
    assert L.slk_softmax_viterbi_pack_f32(W.data_ptr(), b.data_ptr(), K, 4, 5, pack.data_ptr(), None) == 0
    def call():
        rc = L.slk_softmax_viterbi_f32(x.data_ptr(), K, pack.data_ptr(), K, T, B, 4, 5, 0.0, 1e-5, None, 0, ws.data_ptr(), nws, sc.data_ptr(), pa.data_ptr(), le.data_ptr(), None, None)
        assert rc == 0, rc
    call(); torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
    print("%-32s forward + backtrace %.3f ms, mean path length %.0f of %d steps" % (name, min(ts), le.float().mean().item(), T), flush=True)
