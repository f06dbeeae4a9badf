"""sloika_amd -- MI355X (gfx950) native implementation of ONE hot path of nanoporetech/sloika:

    chunkify/normalise -> conv front end -> stacked GRU/LSTM -> softmax -> k-mer Viterbi decode

behind the reference's own layer / decode / worker API (see DESIGN.md and INTEGRATION.md).  The arithmetic runs in
hand-written HIP kernels (sloika_amd/csrc) loaded through a C ABI (include/sloika_amd.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
