"""pytest configuration.

Markers
  gpu : needs a real MI355X (run by the driver on the GPU box with `-m gpu`); everything else must
        pass on a CPU-only machine with `-m "not gpu"`.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real AMD GPU (MI355X)")


def regen_post(gen, stored):
    """Re-create an input that tests/golden/make_goldens.py did not store, from its recipe."""
    kind = gen.get("kind")
    if kind == "stored" or (kind is None and "key" in gen):
        return stored[gen["key"]]
    if kind == "dirichlet":
        return np.random.RandomState(gen["seed"]).dirichlet(
            np.ones(gen["nst"]) * gen["alpha"], size=gen["nev"]).astype(np.float32)
    if kind == "tie":
        rs = np.random.RandomState(gen["seed"])
        return np.asarray(gen["levels"], dtype=np.float32)[rs.randint(0, len(gen["levels"]), size=(gen["nev"], gen["nst"]))]
    raise ValueError(gen)


@pytest.fixture(scope="session")
def golden_cases():
    with open(os.path.join(GOLDEN, "cases.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_decode():
    return dict(np.load(os.path.join(GOLDEN, "decode.npz")))


@pytest.fixture(scope="session")
def golden_transducer():
    return dict(np.load(os.path.join(GOLDEN, "transducer.npz")))


@pytest.fixture(scope="session")
def golden_signal():
    return dict(np.load(os.path.join(GOLDEN, "signal.npz")))


@pytest.fixture(scope="session")
def golden_prepare_post():
    return dict(np.load(os.path.join(GOLDEN, "prepare_post.npz")))


@pytest.fixture(scope="session")
def golden_bio():
    with open(os.path.join(GOLDEN, "bio.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


def decode_case_input(case, golden_decode):
    import hashlib
    key = "post_" + case["name"]
    post = golden_decode[key] if key in golden_decode else regen_post(case["gen"], golden_decode)
    assert hashlib.sha256(np.ascontiguousarray(post).tobytes()).hexdigest() == case["sha256"], \
        "regenerated golden input differs from the one the reference saw"
    return post
