"""Basecall the example reads of tests/golden/reads.npz with the trained pretrained.pkl weights and compare with the basecall
ONT's software stored in the fast5 files (sequence identity from the edit distance)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def edit_distance(a, b):
    a, b = np.frombuffer(a.encode(), dtype=np.uint8), np.frombuffer(b.encode(), dtype=np.uint8)
    idx = np.arange(len(b) + 1)
    prev = idx.copy()
    for i, ca in enumerate(a, 1):
        cur = np.minimum(prev[:-1] + (b != ca), prev[1:] + 1)
        cur = np.concatenate(([i], cur))
        prev = np.minimum.accumulate(cur - idx) + idx
    return int(prev[-1])


def main():
    from sloika_amd import basecall, bio, models
    g = np.load(os.path.join(ROOT, "tests", "golden", "reads.npz"))
    net = models.from_weights_npz(os.path.join(ROOT, "tests", "golden", "pretrained_weights.npz"))
    calc_post = net.compile()
    kmers = bio.all_kmers(5)
    from sloika_amd import layers
    for n in (5, 3):
        dig, off, rng, rate = g["meta_%d" % n]
        sig0 = (g["adc_%d" % n].astype(np.float64) + off) * (rng / dig)
        both = []
        for split in (True, False):                       # fp16x3 projections vs plain fp32 MFMA everywhere
            layers.SPLIT_F16 = split
            layers.Softmax.split_f16 = split
            _, sc, call, _ = basecall.raw_read_worker(calc_post, sig0, kmer_len=5, skip=5.0, name="read%d" % n)
            both.append((sc, bio.kmers_to_sequence([kmers[i] for i in call], always_move=True)))
        layers.SPLIT_F16 = layers.Softmax.split_f16 = True
        print("read%d: fp16x3 vs all-fp32 arithmetic: scores %.4f / %.4f, sequences %s (edit distance %d of %d)" % (
            n, both[0][0], both[1][0], "identical" if both[0][1] == both[1][1] else "differ",
            edit_distance(both[0][1], both[1][1]), len(both[0][1])))
        signal = (g["adc_%d" % n].astype(np.float64) + off) * (rng / dig)
        stored = g["called_%d" % n].tobytes().decode()
        for skip in (0.0, 5.0):
            t0 = time.perf_counter()
            name, score, call, nsamp = basecall.raw_read_worker(calc_post, signal, kmer_len=5, skip=skip, name="read%d" % n)
            dt = time.perf_counter() - t0
            seq = bio.kmers_to_sequence([kmers[i] for i in call], always_move=True)
            d = edit_distance(seq, stored)
            print("read%d skip=%.0f: %d samples -> %d bases (stored %d) in %.2f s; edit distance %d, identity %.3f; %s..." % (
                n, skip, nsamp, len(seq), len(stored), dt, d, 1.0 - d / max(len(seq), len(stored)), seq[:50]))


def ragged_throughput():
    """Whole reads one at a time (the reference's mode) vs the same reads as one ragged batch."""
    import torch
    from sloika_amd import basecall, models, pipeline
    g = np.load(os.path.join(ROOT, "tests", "golden", "reads.npz"))
    net = models.from_weights_npz(os.path.join(ROOT, "tests", "golden", "pretrained_weights.npz"))
    calc_post = net.compile()
    sig = {}
    for n in (5, 3):
        dig, off, rng, rate = g["meta_%d" % n]
        sig[n] = ((g["adc_%d" % n].astype(np.float64) + off) * (rng / dig)).astype(np.float32)
    rs = np.random.RandomState(1)
    reads = []
    for i in range(64):                                   # 64 reads of 8k..51k samples cut from the two example reads
        src = sig[3] if i % 2 else sig[5]
        n = int(rs.randint(8000, len(src)))
        reads.append(src[:n])
    total = sum(len(r) - len(r) % 100 for r in reads)
    bc = pipeline.Basecaller(net, kmer_len=5, min_prob=1e-5, skip=5.0)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for r in reads:
            basecall.raw_read_worker(calc_post, r, trim=(0, 0), kmer_len=5, skip=5.0)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out = bc.call_reads(reads)
        out[1].cpu()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print("64 whole reads, %d samples: one by one %.3f s (%.2f M samples/s), one ragged batch %.3f s (%.2f M samples/s)" % (
        total, t1 - t0, total / (t1 - t0) / 1e6, t2 - t1, total / (t2 - t1) / 1e6))


if __name__ == "__main__":
    main()
    ragged_throughput()
