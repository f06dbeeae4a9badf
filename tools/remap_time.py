#!/usr/bin/env python3
"""Time the remap DP (slk_map_to_sequence_batch_f32: transducer.map_to_sequence, sloika/transducer.py:14-73) on synthetic reads.

    python tools/remap_time.py [--reads 64] [--events 1800] [--positions 600] [--reps 5]

Posteriors follow a monotone walk through the reference's states plus Dirichlet noise (the generator of
tests/golden/make_remap_goldens.py).  Prints ms per launch, reads/s and events*positions/s; with --check the first read is
compared with the CPU oracle.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=64)
    ap.add_argument("--events", type=int, default=1800)
    ap.add_argument("--positions", type=int, default=600)
    ap.add_argument("--slip", type=float, default=5.0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    import torch
    import make_remap_goldens as mrg
    from sloika_amd import _lib, device as D
    rs = np.random.RandomState(7)
    base_states = [int(v) for v in rs.randint(1, 1025, size=a.positions)]
    posts = [np.log(np.maximum(mrg.plausible_posterior(np.random.RandomState(100 + r), base_states, a.events), 1e-5)).astype(np.float32)
             for r in range(min(a.reads, 4))]
    dev = D.device()
    nread = a.reads
    lt = torch.cat([torch.from_numpy(posts[r % len(posts)]) for r in range(nread)]).to(dev)
    ev_off = torch.arange(nread + 1, dtype=torch.int64) * a.events
    pos_off = torch.arange(nread + 1, dtype=torch.int64) * a.positions
    ws_off = torch.arange(nread, dtype=torch.int64) * a.events * a.positions
    seq = torch.tensor(base_states * nread, dtype=torch.int32)
    ev_off, pos_off, ws_off, seq = (t.to(dev) for t in (ev_off, pos_off, ws_off, seq))
    ws = torch.empty(nread * a.events * a.positions, dtype=torch.int32, device=dev)
    score = torch.empty(nread, dtype=torch.float32, device=dev)
    path = torch.empty(nread * a.events, dtype=torch.int32, device=dev)
    L = _lib.lib()

    def launch():
        _lib.check(L.slk_map_to_sequence_batch_f32(lt.data_ptr(), 1025, ev_off.data_ptr(), seq.data_ptr(), pos_off.data_ptr(), nread,
                                                   a.positions, a.slip, None, None, ws.data_ptr(), ws_off.data_ptr(),
                                                   score.data_ptr(), path.data_ptr(), D.stream_ptr()), "remap")
    launch()
    torch.cuda.synchronize()
    times = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        launch()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = min(times)
    print("reads %d  events %d  positions %d  slip %g:  %.3f ms per launch  (%.1f us per event step of the slowest read), "
          "%.0f reads/s, %.2f G cells/s" % (nread, a.events, a.positions, a.slip, t * 1e3, t * 1e6 / a.events, nread / t,
                                            nread * a.events * a.positions / t / 1e9))
    if a.check:
        from oracle import oracle
        oracle.build()
        sc, pa = oracle.map_to_sequence(posts[0], base_states, slip=a.slip, log=True)
        got = path[: a.events].cpu().numpy()
        print("check vs oracle: score %s path %s" % (np.float32(sc) == score[0].item(), np.array_equal(got, pa)))


if __name__ == "__main__":
    main()
