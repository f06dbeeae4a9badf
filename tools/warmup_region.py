#!/usr/bin/env python3
"""bench.py's main region by itself in a fresh process: W warm-up steps, synchronise, K steps on the wall clock -- with the device
time of every step beside it (events on the launch stream), to see where the region's time goes.   [W] [K]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch  # noqa: E402
import bench  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
run = bench.Runner(torch, "raw_0.98_rgrgr", 1024, 4000, 1)
for i in range(W):
    run.step(i)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
host = []
t0 = time.perf_counter()
ev[0].record()
for i in range(K):
    run.step(i)
    ev[i + 1].record()
    host.append((time.perf_counter() - t0) * 1e3)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
print("wall %.3f ms per step (%.2f ms total, last issue at %.2f, drain %.2f); device per step: %s" % (
    (t2 - t0) / K * 1e3, (t2 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, " ".join("%.2f" % v for v in ms)))
print("host issue done at: %s" % " ".join("%.1f" % h for h in host))
