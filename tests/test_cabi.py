"""The C-ABI library loads on a CPU-only box and exports exactly what include/sloika_amd.h declares
(no compute calls here: there is no GPU and no CPU fallback)."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT


def _declared():
    with open(os.path.join(ROOT, "include", "sloika_amd.h")) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slk_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    from sloika_amd import build
    return build.build()


def test_header_symbols_exported(built):
    names = _declared()
    assert len(names) >= 25
    lib = ctypes.CDLL(built)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/sloika_amd.h but not exported: %s" % missing


def test_library_exports_nothing_but_the_header(built):
    """-fvisibility=hidden + SLK_API: the dynamic symbol table holds the declared entry points and no other slk_* name."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", built], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\b(slk_[A-Za-z0-9_]+)\b", out)))
    assert exported == _declared()
    others = [ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[1] in ("T", "t", "D", "B")
              and not ln.split()[-1].startswith(("slk_", "_Z", "__hip", "_fini", "_init", "__"))]
    assert not others, others


def test_python_prototypes_cover_header(built):
    from sloika_amd import _lib
    assert sorted(_lib.PROTOTYPES) == _declared()


def test_host_only_entry_points(built):
    from sloika_amd import _lib
    L = _lib.lib()
    assert L.slk_abi_version() == 1
    assert _lib.error_string(0) == "ok" and "invalid" in _lib.error_string(-1)
    assert L.slk_conv1d_out_len(4000, 11, 5, 5, 5) == 800          # T' of the rgrgr front end
    assert L.slk_conv1d_out_len(4000, 11, 2, 5, 5) == 2000
    assert L.slk_conv1d_out_len(3, 11, 5, 0, 0) == 0
    assert L.slk_gru_workspace_bytes(800, 1024, 96) == 800 * 1024 * 288 * 4
    assert L.slk_viterbi_kmer_workspace_bytes(800, 1024, 4, 5) >= 800 * 1024 * 1024
    assert L.slk_viterbi_kmer_workspace_bytes(800, 4, 4, 2) == 0    # klen < 3: decode.py:50


def test_fails_loudly_without_gpu(built):
    import torch
    from sloika_amd import _lib, layers
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.SloikaAmdError):
        _lib.require_gpu()
    import numpy as np
    net = layers.FeedForward(3, 4)
    with pytest.raises(_lib.SloikaAmdError):
        net.compile()(np.zeros((2, 1, 3), dtype=np.float32))


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under sloika_amd/ may import, load or name it."""
    pkg = os.path.join(ROOT, "sloika_amd")
    for dirpath, _, files in os.walk(pkg):
        if "_build" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                with open(os.path.join(dirpath, f)) as fh:
                    text = fh.read()
                assert "liboracle" not in text and "import oracle" not in text and "from oracle" not in text, f
