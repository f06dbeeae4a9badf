"""Timing of the remap DP (transducer.map_to_sequence_batch, csrc/transducer.hip): 256 reads of
800 x 1025 posteriors against 400 states each, the shape BASELINE.md times the reference on (87 ms per read, one core)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import transducer
nread, nev, nst, npos = 256, 800, 1025, 400
rs = np.random.RandomState(0)
lp = torch.log_softmax(torch.randn(nread, nev, nst, device="cuda"), dim=2)
trans = [lp[i] for i in range(nread)]
seqs = [rs.randint(1, nst, size=npos) for _ in range(nread)]
transducer.map_to_sequence_batch(trans[:4], seqs[:4], 5.0)
torch.cuda.synchronize()
for n in (1, 16, 256):
    t0 = time.perf_counter()
    for _ in range(3):
        scores, paths = transducer.map_to_sequence_batch(trans[:n], seqs[:n], 5.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("map_to_sequence_batch: %3d reads of %d x %d vs %d states: %.2f ms per call, %.3f ms per read (host wrapper included)" % (n, nev, nst, npos, dt * 1e3, dt * 1e3 / n))
