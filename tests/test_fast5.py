"""The HDF5-subset reader (sloika_amd/fast5.py) against the reference's example reads, and the fixture cut from them.

On the GPU box only the fixture exists; in the build container the reader is also run on the original files (the ones the
reference's own test/unit/test_fast5.py opens)."""
import hashlib
import os

import numpy as np
import pytest

REF_READS = os.path.join("/root/reference", "data", "reads")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
have_ref = os.path.isdir(REF_READS)


def test_read_fixture_is_well_formed():
    g = np.load(os.path.join(GOLDEN, "reads.npz"))
    for n in (5, 3):
        adc, meta, called = g["adc_%d" % n], g["meta_%d" % n], g["called_%d" % n].tobytes().decode()
        assert adc.dtype == np.int16 and adc.ndim == 1 and len(adc) > 30000
        assert meta.shape == (4,) and meta[0] == 8192.0 and meta[3] > 3000.0           # digitisation, sampling rate
        assert set(called) <= set("ACGT") and len(called) > 3000
        pa = (adc.astype(np.float64) + meta[1]) * (meta[2] / meta[0])
        assert 50.0 < np.median(pa) < 150.0                                             # picoamperes of a DNA strand


@pytest.mark.skipif(not have_ref, reason="reference checkout not present (GPU box)")
def test_reader_on_reference_reads_and_fixture_provenance():
    from sloika_amd import fast5
    g = np.load(os.path.join(GOLDEN, "reads.npz"))
    lengths = {}
    for n in range(1, 9):
        path = os.path.join(REF_READS, "read%d.fast5" % n)
        f = fast5.Fast5(path)
        assert {"Raw", "UniqueGlobalKey"} <= set(f.h5.root.keys())        # read8 carries no Analyses group
        adc = f.get_read(scale=False)
        assert adc.dtype == np.int16 and adc.ndim == 1
        lengths[n] = len(adc)
        assert int(f.read_attrs()["duration"]) == len(adc)                # the reader saw every chunk of the dataset
        sig = f.get_read()
        m = f.channel_meta
        np.testing.assert_allclose(sig, (adc.astype(np.float64) + m["offset"]) * m["range"] / m["digitisation"])
        assert 3000.0 <= f.sample_rate <= 6024.0                         # 3012, 4000 and 6024 Hz runs among the eight
        if n in (5, 3):
            assert np.array_equal(adc, g["adc_%d" % n])
            assert f.stored_basecall()[1] == g["called_%d" % n].tobytes().decode()
            assert hashlib.sha256(open(path, "rb").read()).hexdigest() == g["sha_%d" % n].tobytes().decode()
    assert lengths[1] == 114400                                           # test/unit/test_fast5.py:103
    assert fast5.Fast5(os.path.join(REF_READS, "read8.fast5")).stored_basecall() is None
    with pytest.raises(fast5.Fast5Error):
        fast5.HDF5File(os.path.join("/root/reference", "data", "strands.txt"))


def test_reader_rejects_non_hdf5(tmp_path):
    from sloika_amd import fast5
    p = tmp_path / "x.fast5"
    p.write_bytes(b"not an hdf5 file at all")
    with pytest.raises(fast5.Fast5Error):
        fast5.HDF5File(str(p))
