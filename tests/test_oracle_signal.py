"""Pin the oracle's chunk front end (median/MAD normalisation) against reference-generated goldens.

Reference: sloika/tools/chunkify_raw.py:172-185, sloika/maths.py:4-27, test/unit/test_maths.py:37-60.
"""
import numpy as np


def test_med_mad_kat(oracle):
    # test/unit/test_maths.py:53-60 with the default factor folded out
    x = np.array([[0.5, 0.5, 0.5, 0.5], [0.5, 0.5, 1.0, 1.0], [0.0, 0.5, 0.5, 1.0]], dtype=np.float32)
    _, med, mad = oracle.med_mad_normalise(x, return_stats=True)
    assert np.allclose(med, [0.5, 0.75, 0.5])
    assert np.allclose(mad / np.float32(1.4826), [0, 0.25, 0.25])


def test_per_chunk_normalisation_bit_exact(oracle, golden_signal):
    g = golden_signal
    chunks = g["chunks_none"]
    assert chunks.shape == (5, 4000)
    assert np.array_equal(chunks.reshape(-1), g["signal"][:20000])      # raw_chunkify reshape, :172-176
    out = oracle.med_mad_normalise(chunks)
    assert np.array_equal(out, g["chunks_per_chunk"])


def test_per_read_normalisation_bit_exact(oracle, golden_signal):
    g = golden_signal
    sig = g["signal"]
    out, med, mad = oracle.med_mad_normalise(sig[None, :], return_stats=True)
    assert med[0] == g["med_mad_read"][0] and mad[0] == g["med_mad_read"][1]
    assert np.array_equal(out[0], g["read_norm"])                        # basecall.py:117-118
    # chunkify_raw.py:182-183: median/mad over the trimmed 5x4000 block
    out2 = oracle.med_mad_normalise(sig[None, :20000])
    assert np.array_equal(out2.reshape(5, 4000), g["chunks_per_read"])
