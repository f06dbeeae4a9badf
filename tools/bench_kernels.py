#!/usr/bin/env python3
"""Micro-benchmarks of single kernels through the C ABI (development aid, not the driver's bench)."""
import argparse
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sloika_amd import _lib


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=800)
    ap.add_argument("--B", type=int, default=1024)
    ap.add_argument("--n", type=int, default=96)
    ap.add_argument("--what", default="gru,gemm,softmax,viterbi")
    a = ap.parse_args()
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    T, B, n = a.T, a.B, a.n
    what = a.what.split(",")
    if "gru" in what:
        vI = torch.randn(T, B, 3 * n, device="cuda") * 0.5
        sW = torch.randn(2 * n, n, device="cuda") / np.sqrt(2 * n)
        sW2 = torch.randn(n, n, device="cuda") / np.sqrt(2 * n)
        y = torch.empty(T, B, n, device="cuda")
        for generic in (0,):
            ms = timeit(lambda: L.slk_gru_recurrent_f32_ex(vI.data_ptr(), sW.data_ptr(), sW2.data_ptr(), y.data_ptr(), n, T, B,
                                                           n, 0, 1, 2, generic, st))
            fl = 6.0 * T * B * n * n
            print("gru_recurrent n=%d B=%d T=%d generic=%d: %.3f ms  %.1f TF  %.0f ns/step" % (n, B, T, generic, ms, fl / ms / 1e9, ms * 1e6 / T))
    # (the "gruf16" / "gruf" sections timed csrc/gru_fused16.hip / gru_fused.hip, which left the library in round 4: tools/experiments/)
    if "gemm" in what:
        for (K, N) in ((n, 3 * n), (n, 1025)):
            M = T * B
            x = torch.randn(M, K, device="cuda")
            W = torch.randn(N, K, device="cuda")
            b = torch.randn(N, device="cuda")
            y = torch.empty(M, N, device="cuda")
            ms = timeit(lambda: L.slk_gemm_bias_act_f32(x.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, K, N, 0, st), reps=5)
            print("gemm M=%d K=%d N=%d: %.3f ms  %.1f TF" % (M, K, N, ms, 2.0 * M * K * N / ms / 1e9))
            del x, W, y
    if "conv" in what:
        Tin, Cout = 4000, n
        xs = torch.randn(B, Tin, device="cuda")
        Wc = torch.randn(Cout, 1, 11, device="cuda") * 0.3
        bc = torch.randn(Cout, device="cuda")
        Tout = L.slk_conv1d_out_len(Tin, 11, 5, 5, 5)
        yc = torch.empty(Tout, B, Cout, device="cuda")
        for act in (0, 1, 3):
            ms = timeit(lambda: L.slk_conv1d_f32(xs.data_ptr(), 1, Tin, Wc.data_ptr(), bc.data_ptr(), yc.data_ptr(), Tin, B, 1, Cout, 11, 5, 5, 5, act, st), reps=5)
            print("conv1d chunk-major B=%d Cout=%d act=%d: %.3f ms  %.0f GB/s written" % (B, Cout, act, ms, 4.0 * Tout * B * Cout / ms / 1e6))
        xt = xs.t().contiguous()
        ms = timeit(lambda: L.slk_conv1d_f32(xt.data_ptr(), B, 1, Wc.data_ptr(), bc.data_ptr(), yc.data_ptr(), Tin, B, 1, Cout, 11, 5, 5, 5, 3, st), reps=5)
        print("conv1d [T,B,1] layout act=3: %.3f ms" % ms)
    if "gemmrows" in what:
        M, K, N = T * B, n, 1025
        x = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda") * 0.3
        b = torch.randn(N, device="cuda")
        for ld in (1025, 1056):
            y = torch.empty(M, ld, device="cuda")
            stats = torch.empty(M, 2, device="cuda")
            for rnd in range(2):
                ms0 = timeit(lambda: L.slk_linear_rowstats_f32(x.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr(), ld, M, K, N, None, st), reps=5)
                ms1 = timeit(lambda: L.slk_linear_rowstats_f32(x.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr(), ld, M, K, N, stats.data_ptr(), st), reps=5)
                print("gemm_rows M=%d K=%d N=%d ld=%d: no-stats %.3f ms (%.1f TF)  with-stats %.3f ms (%.1f TF)" % (M, K, N, ld, ms0, 2.0 * M * K * N / ms0 / 1e9, ms1, 2.0 * M * K * N / ms1 / 1e9))
            del y
    if "f16x3" in what:
        M, K, N = T * B, n, 1025
        x = torch.tanh(torch.randn(M, K, device="cuda"))
        W = torch.randn(N, K, device="cuda") * 0.5
        b = torch.randn(N, device="cuda")
        KP = (K + 15) // 16 * 16
        hi = torch.empty(N, KP, dtype=torch.float16, device="cuda")
        lo = torch.empty(N, KP, dtype=torch.float16, device="cuda")
        inv = torch.empty((N,), dtype=torch.float32, device="cuda")
        assert L.slk_split_f16x2_f32(W.data_ptr(), N, K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), st) == 0
        ld = 1056
        y = torch.empty(M, ld, device="cuda")
        y2 = torch.empty(M, ld, device="cuda")
        stats = torch.empty(M, 2, device="cuda")
        stats2 = torch.empty(M, 2, device="cuda")
        for rnd in range(2):
            ms0 = timeit(lambda: L.slk_linear_rowstats_f16x3(x.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), b.data_ptr(), y.data_ptr(), ld, M, K, N, None, st), reps=5)
            ms1 = timeit(lambda: L.slk_linear_rowstats_f16x3(x.data_ptr(), K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), b.data_ptr(), y.data_ptr(), ld, M, K, N, stats.data_ptr(), st), reps=5)
            print("gemm_rows_f16x3 M=%d K=%d N=%d: no-stats %.3f ms (%.1f TF-equiv)  with-stats %.3f ms (%.1f TF-equiv, %.0f GB/s written)" % (M, K, N, ms0, 2.0 * M * K * N / ms0 / 1e9, ms1, 2.0 * M * K * N / ms1 / 1e9, 4.0 * M * N / ms1 / 1e6))
        L.slk_linear_rowstats_f32(x.data_ptr(), K, W.data_ptr(), b.data_ptr(), y2.data_ptr(), ld, M, K, N, stats2.data_ptr(), st)
        torch.cuda.synchronize()
        sub = slice(0, 20000)
        ref = x[sub].double() @ W.double().t() + b.double()
        e16 = (y[sub, :N].double() - ref).abs().max().item()
        e32 = (y2[sub, :N].double() - ref).abs().max().item()
        print("max |logit error| vs float64: f16x3 %.3e   fp32 MFMA %.3e   (|logit| max %.1f)" % (e16, e32, ref.abs().max().item()))
        print("stats max rel diff (1/sum): %.3e" % ((stats[:, 1] - stats2[:, 1]).abs() / stats2[:, 1]).max().item())
    if "softmax" in what:
        M, N = T * B, 1025
        y = torch.randn(M, N, device="cuda")
        ms = timeit(lambda: L.slk_softmax_rows_f32(y.data_ptr(), M, N, st), reps=5)
        print("softmax_rows M=%d N=%d: %.3f ms  %.0f GB/s" % (M, N, ms, 8.0 * M * N / ms / 1e6))
        del y
    if "viterbi" in what:
        post = torch.rand(T, B, 1025, device="cuda")
        post /= post.sum(dim=2, keepdim=True)
        nb = L.slk_viterbi_kmer_workspace_bytes(T, B, 4, 5)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        sc = torch.empty(B, device="cuda")
        pa = torch.empty(B, T, dtype=torch.int32, device="cuda")
        le = torch.empty(B, dtype=torch.int32, device="cuda")
        ms = timeit(lambda: L.slk_viterbi_kmer_f32(post.data_ptr(), T, B, 4, 5, 0.0, 0, 1e-5, ws.data_ptr(), nb, sc.data_ptr(),
                                                   pa.data_ptr(), le.data_ptr(), st), reps=5)
        print("viterbi T=%d B=%d: %.3f ms (%.0f ns/step/chunk-wave)" % (T, B, ms, ms * 1e6 / T))


if __name__ == "__main__":
    main()
