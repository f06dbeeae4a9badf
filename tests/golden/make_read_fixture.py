#!/usr/bin/env python3
"""Extract two of the reference's example reads (data/reads/*.fast5, the files its own tests open: test_fast5.py:98-110)
into a small fixture for the GPU box, where neither /root/reference nor an HDF5 library exists:

    tests/golden/reads.npz
        adc_<n>      int16 ADC samples of Raw/Reads/Read_*/Signal
        meta_<n>     float64 [digitisation, offset, range, sampling_rate]   (UniqueGlobalKey/channel_id)
        called_<n>   the 1D template basecall ONT's own software stored in the file (Analyses/Basecall_1D_000), as bytes
        sha_<n>      sha256 of the fast5 file the arrays came from

Data only (inputs and a third-party expected output); read with sloika_amd/fast5.py, this repository's HDF5 subset reader.
Run from the repository root in the build container:  python tests/golden/make_read_fixture.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sloika_amd import fast5  # noqa: E402

READS = os.path.join("/root/reference", "data", "reads")


def main():
    out = {}
    for n in (5, 3):
        path = os.path.join(READS, "read%d.fast5" % n)
        f = fast5.Fast5(path)
        out["adc_%d" % n] = f.get_read(scale=False).astype(np.int16)
        m = f.channel_meta
        out["meta_%d" % n] = np.array([m["digitisation"], m["offset"], m["range"], m["sampling_rate"]], dtype=np.float64)
        out["called_%d" % n] = np.frombuffer(f.stored_basecall()[1].encode("ascii"), dtype=np.uint8)
        out["sha_%d" % n] = np.frombuffer(hashlib.sha256(open(path, "rb").read()).hexdigest().encode("ascii"), dtype=np.uint8)
    dst = os.path.join(ROOT, "tests", "golden", "reads.npz")
    np.savez_compressed(dst, **out)
    print(dst, os.path.getsize(dst), "bytes", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
