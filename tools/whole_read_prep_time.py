"""Where the host-side preparation of the whole-read mode spends its time (pipeline.Basecaller.prepare_read_batches), per stage.
    python tools/whole_read_prep_time.py [nreads]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bench
    from sloika_amd import _lib, batch, models, pipeline
    _lib.require_gpu()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    reads = bench.synthetic_reads(n)
    net = models.randomise_zero_layers(models.build_model("raw_0.98_rgrgr", klen=5, sd=0.5, seed=11))
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev, off, lens = batch.upload_reads_windowed(reads)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        bounds = batch.open_pore_bounds_many(dev, off, lens, 0.0)
        t2 = time.perf_counter()
        nsamp = [hi - lo for lo, hi in bounds]
        buckets = pipeline.Basecaller.length_buckets(nsamp, 256, 0.08)
        t3 = time.perf_counter()
        batches, nsamp2 = pipeline.Basecaller.prepare_read_batches(net, reads, max_batch=256, max_waste=0.08, kmer_len=5, skip=0.0)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print("reads %d (%.0f M samples): pack + upload %.3f s, bounds %.3f s, bucketing %.3f s | whole prepare_read_batches %.3f s (%d batches)"
              % (n, sum(lens) / 1e6, t1 - t0, t2 - t1, t3 - t2, t4 - t3, len(batches)), flush=True)


if __name__ == "__main__":
    main()
