"""`bench.py --gpus N` is the multi-GPU entry point (one process per GPU, reference fan-out: sloika/iterators.py:343-351,
bin/basecall_network.py:100-101).  On this CPU-only box it must really create N ranks, and each must fail loudly at
require_gpu() -- never fall back to one rank or to a CPU path."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0",
                           "--cpu-chunks", "0"] + list(extra), env=env, capture_output=True, text=True, timeout=600)


def test_gpus_2_starts_two_ranks_that_refuse_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-box check of the launcher; on a GPU box bench.py itself is the test")
    r = _run("--gpus", "2")
    assert r.returncode != 0
    assert "rank 0 of 2 starting" in r.stderr and "rank 1 of 2 starting" in r.stderr
    assert '"metric"' not in r.stdout                      # no result line without a GPU
    assert "GPU" in r.stderr or "gpu" in r.stderr            # require_gpu()'s message, not a silent fallback


def test_gpus_must_match_the_launcher_world_size():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr
