"""`bench.py --gpus N` is the multi-GPU entry point (one process per GPU, reference fan-out: sloika/iterators.py:343-351,
bin/basecall_network.py:100-101).  On this CPU-only box it must really create N ranks, and each must fail loudly at
require_gpu() -- never fall back to one rank or to a CPU path."""
import json
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


def _stub(*extra, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-device", "--steps", "5", "--warmup", "1"] + list(extra),
                          env=env, capture_output=True, text=True, timeout=600)


def test_eight_ranks_rank_plumbing_on_gloo():
    """The N > 1 code path of bench.py end to end without a GPU: `--gpus 8` starts eight ranks under torch.distributed.run, every
    rank binds the device LOCAL_RANK names, the barriers and the max over ranks run on a real (gloo) process group, rank 0 alone
    prints ONE line whose time is the slowest rank's, and `per_rank_ms` shows every rank's own."""
    r = _stub("--gpus", "8")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 5 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["value"] is None and "stub" in d["data"]                   # no throughput is claimed from a stub
    assert len(d["per_rank_ms"]) == 8
    assert d["per_rank_ms"][7] > d["per_rank_ms"][0] * 1.5              # StubRunner sleeps 1 + r / 4 units: the straggler is visible
    assert d["ms_per_step"] >= max(d["per_rank_ms"]) * 0.9              # ... and the line's time is the maximum over ranks
    assert d["config"]["global_batch"] == 8 * d["config"]["batch_per_gpu"]
    for rank in range(8):
        assert "rank %d of 8 starting" % rank in r.stderr
        if rank:
            assert "rank %d bound to device %d" % (rank, rank) in r.stderr
    # what the driver's record needs on the day the 8-GPU run is not skipped: who the ranks were, how many the collective saw
    assert d["devices"] == ["stub:%d" % r_ for r_ in range(8)] and d["rccl_ranks_seen"] == 8
    # the record is the LAST line of stdout and short enough for the driver's capture (round 5's 22 KB line was cut: parsed = null)
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert last == lines[0] and len(last) < 4096
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    with open(os.path.join(ROOT, d["detail"])) as fh:
        full = json.load(fh)
    assert full["n_gpus"] == 8 and len(full["per_rank_ms"]) == 8


def test_compact_line_of_a_full_run_stays_under_four_kilobytes():
    """Round 5's own 22 KB line (profiles/r05f_bench.json) through compact_line: the contract's fields, `roofline` with traffic and
    MfmaUtil, `cpu_baseline` and the legs' scalars survive, in under 4 KB."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r05f_bench.json")) as fh:
        full = json.loads(fh.read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 15000
    full["devices"], full["rccl_ranks_seen"], full["detail"] = ["AMD Instinct MI355X@0000:05:00"] * 8, 8, "bench_detail.json"
    full["per_rank_ms"] = [3.7512345678] * 8
    s = bench.compact_line(full)
    assert len(s) < 4096, len(s)
    d = json.loads(s)
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-3 and abs(d["ms_per_step"] - full["ms_per_step"]) < 1e-2
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "mfma_util", "hbm_gbs", "stale"):
        assert k in d["roofline"], k
    assert d["roofline"]["frac"] > 0 and d["roofline"]["traffic"] > 0
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert d["cpu_baseline"][k] is not None
    for k in ("sustained_one", "four_in_flight_det", "train_ms", "pretrained", "b256_rgrgr_one", "whole_reads_from_host"):
        assert d.get(k), k
    assert d["config"]["workload"] and d["dtype"] and d["vs_baseline"] is None


def test_a_failing_rank_fails_the_launch():
    r = _stub("--gpus", "4", SLOIKA_AMD_STUB_FAIL_RANK="2")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_stub_single_process_and_ranks_wrap_onto_devices():
    """One process, no launcher: no process group, the same line.  With more ranks than devices the binding wraps (LOCAL_RANK % devices)."""
    r = _stub()
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["per_rank_ms"] and len(d["per_rank_ms"]) == 1
    r = _stub("--gpus", "4", SLOIKA_AMD_STUB_DEVICES="2")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rank 3 bound to device 1" in r.stderr and "rank 2 bound to device 0" in r.stderr
