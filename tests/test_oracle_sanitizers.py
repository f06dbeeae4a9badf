"""the CPU restatement under AddressSanitizer + UndefinedBehaviourSanitizer (oracle/Makefile `asan`).  The oracle is the
checker of every GPU parity test: an out-of-bounds read in IT would make those tests compare against garbage.  Its entry points run here
on small inputs in a child process with libasan preloaded (the Python interpreter itself is not instrumented)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from oracle import oracle as orc
from sloika_amd import models
rs = np.random.RandomState(3)
# decode.viterbi / prepare_post (decode.py:21-93), klen 3 and 5, with and without a skip penalty
for klen, T in ((3, 17), (5, 40)):
    ns = 4 ** klen + 1
    post = rs.dirichlet(np.ones(ns), size=T).astype(np.float32)
    for skip in (0.0, 3.0):
        score, path = orc.viterbi(post, klen, skip_pen=skip)
        assert len(path) <= T and np.isfinite(score)
    orc.prepare_post(post[:, None, :], 1e-5)
lp = np.log(rs.dirichlet(np.ones(1025), size=(30, 3)).astype(np.float32))
orc.viterbi_batch(lp, 5, skip_pen=0.0)
# slip_update / map_to_sequence (viterbi_helpers.pyx:12-35, transducer.py:14-73), the smallest legal input included
for n in (3, 10, 257):
    orc.slip_update(rs.normal(size=n).astype(np.float32), 5.0)
trans = np.log(rs.dirichlet(np.ones(65), size=50).astype(np.float32))
orc.map_to_sequence(trans, rs.randint(1, 65, size=12).astype(np.int32), slip=5.0)
# normalisation and whole networks (conv, Gru both directions, Lstm, FeedForward, Softmax, Window), ragged shapes
orc.med_mad_normalise(rs.normal(size=(5, 333)).astype(np.float32))
for name, L, B in (("raw_0.98_rgrgr", 203, 3), ("baseline_raw_gru", 101, 2), ("tiny_gru", 23, 2), ("baseline_lstm", 19, 3)):
    net = models.randomise_zero_layers(models.build_model(name, klen=5 if "raw" in name else 3, sd=0.5, seed=5))
    x = rs.normal(size=(L, B, net.insize)).astype(np.float32)
    y = orc.run_network(net.spec(), x)
    assert np.isfinite(y).all()
print("SANITIZED_OK")
'''


def _libasan():
    try:
        out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, timeout=30).stdout.strip()
    except (OSError, subprocess.SubprocessError):
        return None
    return out if out and os.path.sep in out and os.path.exists(out) else None


def test_oracle_entry_points_under_asan_and_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip("no libasan beside gcc")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    lib = os.path.join(ROOT, "oracle", "_build", "liboracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan, SLOIKA_ORACLE_LIB=lib, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SANITIZED_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
