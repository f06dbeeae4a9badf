"""GPU parity: slip_update and map_to_sequence through the C ABI, bit exact against reference goldens."""
import hashlib

import numpy as np
import pytest

from tests.gpu_util import need_gpu
from tests.test_oracle_transducer import _map_input

pytestmark = pytest.mark.gpu


def test_slip_update_goldens(golden_transducer):
    need_gpu()
    from sloika_amd import viterbi_helpers
    g = golden_transducer
    for n in (3, 4, 10, 400):
        for slip in (0.0, 5.0):
            fs, fp = viterbi_helpers.slip_update(g["slip_x_%d" % n], slip)
            assert fs.dtype == np.float32 and fp.dtype == np.int64
            assert np.array_equal(fs, g["slip_fs_%d_%g" % (n, slip)]) and np.array_equal(fp, g["slip_fp_%d_%g" % (n, slip)])
    fs, fp = viterbi_helpers.slip_update(g["slip_x_tie"], 0.0)
    assert np.array_equal(fs, g["slip_fs_tie"]) and np.array_equal(fp, g["slip_fp_tie"])
    with pytest.raises(ValueError):
        viterbi_helpers.slip_update(np.zeros(2, dtype=np.float32), 1.0)


def test_map_to_sequence_goldens(golden_cases, golden_transducer):
    need_gpu()
    from sloika_amd import transducer
    g = golden_transducer
    for case in golden_cases["map_cases"]:
        trans = _map_input(case, g)
        name = case["name"]
        if not case["log"]:
            trans = np.log(trans)          # feed the same float32 logs the reference computed (transducer.py:30)
        pi = g["map_pi_" + name] if case["has_pi"] else None
        pf = g["map_pf_" + name] if case["has_pf"] else None
        score, path = transducer.map_to_sequence(trans, g["map_seq_" + name], slip=case["slip"], prior_initial=pi,
                                                 prior_final=pf, log=True)
        assert np.array_equal(path, g["map_path_" + name]), name
        assert float(score) == float.fromhex(case["score_hex"]), name


def test_map_to_sequence_device_log_close(golden_cases, golden_transducer):
    need_gpu()
    from sloika_amd import transducer
    g = golden_transducer
    case = [c for c in golden_cases["map_cases"] if c["name"] == "m100_post"][0]
    score, path = transducer.map_to_sequence(g["map_trans_m100_post"], g["map_seq_m100_post"], slip=5.0, log=False)
    assert float(score) == pytest.approx(float.fromhex(case["score_hex"]), rel=1e-5)
    assert (path == g["map_path_m100_post"]).mean() > 0.95
    with pytest.raises(ValueError):
        transducer.map_to_sequence(g["map_trans_m100_post"], g["map_seq_m100_post"], slip=None)
