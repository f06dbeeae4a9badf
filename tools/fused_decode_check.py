"""GPU check of csrc/softmax_viterbi.hip (projection + softmax + prepare_post + log + Viterbi in one kernel):
   (1) the dumped log-posteriors against a float64 numpy evaluation,
   (2) paths / scores against the ORACLE decoder run on those dumped log-posteriors (bit for bit),
   (3) agreement with the two-kernel path (projection kernel + decoder on the logits),
   (4) timing of both at the bench size.
    python tools/fused_decode_check.py [--time] [--K 96]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one_case(rs, T, B, K, skip, ragged, plan):
    import torch
    from sloika_amd import decode, layers
    from oracle import oracle as orc
    S = 1025
    x = np.tanh(rs.normal(size=(T, B, K))).astype(np.float32)
    x[min(3, T - 1)] = 0.0
    W = (rs.normal(size=(S, K)) * 0.5).astype(np.float32)
    b = rs.normal(size=S).astype(np.float32)
    b[0] += 3.0
    sm = layers.Softmax(K, S, has_bias=True)
    sm.W.set_value(W)
    sm.b.set_value(b)
    pack = sm.viterbi_pack(4, 5)
    assert pack is not None
    xd = torch.from_numpy(x).cuda()
    lens, ln = None, np.full(B, T, dtype=np.int32)
    if ragged:
        ln = rs.randint(1, T + 1, size=B).astype(np.int32)
        ln[0] = T
        lens = torch.from_numpy(ln).cuda()
    dump = torch.full((T, B, S), float("nan"), dtype=torch.float32, device="cuda")
    sc, pa, le = decode.viterbi_fused_batch(xd, pack, 5, skip_pen=skip, lengths=lens, lp_dump=dump, plan=plan)
    torch.cuda.synchronize()
    lp = dump.cpu().numpy()
    # (1) float64 evaluation
    l64 = x.astype(np.float64) @ W.astype(np.float64).T + b
    l64 -= l64.max(axis=2, keepdims=True)
    p64 = np.exp(l64)
    p64 /= p64.sum(axis=2, keepdims=True)
    ref = np.log(1e-5 + (1 - 1e-5) * p64 + 1e-10)
    for bb in range(B):
        lp[int(ln[bb]):, bb] = ref[int(ln[bb]):, bb]          # rows past a chunk's end are not part of the contract
    err = np.abs(lp - ref).max()
    perr = np.abs(np.exp(lp) - np.exp(ref)).max()
    # (2) oracle on the dumped log-posteriors
    ok = True
    scn, pan, len_ = sc.cpu().numpy(), pa.cpu().numpy(), le.cpu().numpy()
    for bb in range(B):
        Tb = int(ln[bb])
        o_s, o_p, o_l = orc.viterbi_batch(np.ascontiguousarray(lp[:Tb, bb:bb + 1]), 5, skip_pen=skip)
        if not (o_l[0] == len_[bb] and np.array_equal(o_p[0, :o_l[0]], pan[bb, :len_[bb]]) and o_s[0] == scn[bb]
                and (pan[bb, len_[bb]:] == -1).all()):
            ok = False
            print("  chunk", bb, "oracle len", o_l[0], "got", len_[bb], "score", o_s[0], scn[bb])
    # (3) the two-kernel path
    logits, stats, ld = sm.logits_and_stats(xd)
    s2, p2, l2 = decode.viterbi_logits_batch(logits, stats, 5, T, B, ld=ld, skip_pen=skip, lengths=lens)
    same = bool(torch.equal(pa, p2) and torch.equal(le, l2))
    print("plan %d T=%d B=%d K=%d skip=%g ragged=%d: lp err %.2e, posterior err %.2e, oracle-on-dump %s, equals two-kernel path %s, "
          "score diff %.2e" % (plan, T, B, K, skip, ragged, err, perr, "OK" if ok else "MISMATCH", same,
                                float((sc - s2).abs().max())))
    return ok and err < 2e-4 and perr < 2e-5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--K", type=int, default=96)
    ap.add_argument("--B", type=int, default=1024)
    ap.add_argument("--T", type=int, default=800)
    a = ap.parse_args()
    import torch
    from sloika_amd import _lib, decode, layers
    _lib.require_gpu()
    rs = np.random.RandomState(3)
    S = 1025
    bad = 0
    # plan 4 exists only in builds with -DSV_WITH_NCH4
    PLANS = (2, 4) if os.environ.get("SV_WITH_NCH4") else (2,)
    for (T, B, K, skip, ragged) in ((50, 5, a.K, 0.0, False), (37, 2, a.K, 4.0, False), (1, 3, a.K, 0.0, False),
                                    (16, 1, 64, 0.0, False), (33, 4, 128, 0.0, False), (70, 7, 112, 0.0, True),
                                    (200, 6, 96, 0.0, True), (9, 9, 96, 0.0, True)):
        for plan in PLANS:
            if not one_case(rs, T, B, K, skip, ragged, plan):
                bad += 1
    print("FAILED cases:", bad)
    if a.time:
        T, B, K = a.T, a.B, a.K
        x = torch.tanh(torch.randn((T, B, K), device="cuda"))
        sm = layers.Softmax(K, S, has_bias=True)
        sm.W.set_value((rs.normal(size=(S, K)) * 0.5).astype(np.float32))
        sm.b.set_value(rs.normal(size=S).astype(np.float32))
        pack = sm.viterbi_pack(4, 5)
        ws = decode.ViterbiWorkspace()

        def fused4():
            return decode.viterbi_fused_batch(x, pack, 5, workspace=ws, plan=4)

        def fused2():
            return decode.viterbi_fused_batch(x, pack, 5, workspace=ws, plan=2)

        def pair():
            logits, stats, ld = sm.logits_and_stats(x)
            return decode.viterbi_logits_batch(logits, stats, 5, T, B, ld=ld, workspace=ws)

        legs = (("fused 2", fused2), ("two-kernel", pair))
        if os.environ.get("SV_WITH_NCH4"):
            legs = (("fused 4", fused4),) + legs
        for name, fn in legs * 2:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 10
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print("%-11s T=%d B=%d K=%d: %.3f ms per call" % (name, T, B, K, e0.elapsed_time(e1) / n))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
