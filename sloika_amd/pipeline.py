"""The hot path end to end, resident on one GPU:

    raw chunks [B, chunk_len] (HBM)  --normalise-->  --conv-->  --GRU/LSTM stack-->  --softmax-->
    --prepare_post + log + k-mer Viterbi + backtrace-->  (scores[B], paths[B,T'], lens[B])

This is the batched restatement of bin/basecall_network.py `raw` (sloika/basecall.py:88-121, 26-51) that
BASELINE.json's metric is quoted on.  All arithmetic happens in the HIP kernels behind include/sloika_amd.h;
torch only owns the buffers and the stream.
"""
import os

import numpy as np

from . import _lib, batch, decode, layers

#: decode straight from the Softmax layer's input (csrc/softmax_viterbi.hip: projection, softmax, prepare_post, log and the
#: Viterbi forward pass in one kernel, the logits never written) where that kernel applies; SLOIKA_AMD_FUSED_DECODE=0 keeps the
#: projection kernel + decoder pair
FUSED_DECODE = os.environ.get("SLOIKA_AMD_FUSED_DECODE", "1") != "0"


class Basecaller(object):
    def __init__(self, network, kmer_len=5, nbase=4, min_prob=1e-5, skip=0.0, normalisation='per-chunk', in_flight=1,
                 fused_decode=None):
        """skip default 0.0 is the CLI default (bin/basecall_network.py:38).

        in_flight: how many batches the caller keeps in flight at a time, each on a HIP stream of its own (one Basecaller
        per stream).  With two or more, a Gru layer runs eight chunks per workgroup (csrc/gru_bar16d.hip) whenever that lets
        the layers of all the batches share the chip -- batch 1024, two in flight: 2 x 128 workgroups on 256 CUs -- instead of
        one workgroup per four chunks each taking the whole device in turn."""
        if not isinstance(network, layers.Layer):
            raise TypeError("network must be a sloika_amd.layers.Layer")
        self.network = network
        self.kmer_len, self.nbase, self.min_prob, self.skip = kmer_len, nbase, min_prob, skip
        self.normalisation = normalisation
        self.in_flight = max(1, int(in_flight))
        if self.in_flight > 2:
            from . import device as D
            D.want_hw_queues(4 * self.in_flight)
        self.fused_decode = FUSED_DECODE if fused_decode is None else bool(fused_decode)
        self._ws = decode.ViterbiWorkspace()
        _lib.lib()

    def _hidden(self, chunks, upto):
        """Run the network on [B, chunk_len] device signal up to (not including) layer index `upto`."""
        from . import device as D
        cd = D.to_dev(chunks)
        net = self.network
        seq = net.layers if isinstance(net, layers.Serial) else [net]
        first = seq[0]
        if (self.normalisation == 'per-chunk' and isinstance(first, layers.Convolution) and first.insize == 1
                and len(seq) > 1):
            # the conv front end reads the chunk-major normalised signal directly: no [T,B,1] transpose
            norm = batch.normalise_chunks(cd, 'per-chunk', out_layout='chunk')
            B, T = norm.shape
            x = first.run_strided(norm.data_ptr(), T, B, 1, T, norm.device)
            rest = seq[1:upto]
        else:
            x = batch.normalise_chunks(cd, self.normalisation, out_layout='network')
            rest = seq[:upto]
        keep = layers._HINTS.in_flight                   # (thread local: one forward pass per host thread at a time)
        layers._HINTS.in_flight = self.in_flight
        try:
            for layer in rest:
                x = layer._forward(x, None, False)
        finally:
            layers._HINTS.in_flight = keep
        return x

    def posteriors(self, chunks):
        """[B, chunk_len] device signal -> [T', B, nstate] posteriors (network layout)."""
        net = self.network
        n = len(net.layers) if isinstance(net, layers.Serial) else 1
        return self._hidden(chunks, n)

    def _fused_pack(self, last, hid):
        """The Softmax layer's weights packed for csrc/softmax_viterbi.hip, or None when that kernel does not apply."""
        if not self.fused_decode or hid.stride(1) % 4 or hid.data_ptr() % 16:
            return None
        if hid.stride(2) != 1 or hid.stride(0) != hid.shape[1] * hid.stride(1):
            return None
        return last.viterbi_pack(self.nbase, self.kmer_len)

    def call_chunks(self, chunks, lp_dump=None):
        """-> device tensors (scores float32 [B], paths int32 [B, T'] (-1 padded), lens int32 [B]).

        When the network ends in a Softmax layer whose shape csrc/softmax_viterbi.hip covers, the decoder starts from that
        layer's INPUT and neither the logits nor the posterior (3.4 GB each at B=1024) are ever written; `lp_dump`, a float32
        device tensor [T', B, nstate], then receives the log-posteriors the dynamic programme consumed (tests).  Otherwise
        (and with fused_decode=False) the decoder consumes the layer's logits + row statistics, bit-identical to decoding
        `posteriors()`."""
        net = self.network
        last = net.layers[-1] if isinstance(net, layers.Serial) else None
        if type(last) is layers.Softmax and len(net.layers) > 1:
            hid = self._hidden(chunks, len(net.layers) - 1)
            pack = self._fused_pack(last, hid)
            if pack is not None:
                return decode.viterbi_fused_batch(hid, pack, self.kmer_len, skip_pen=self.skip, nbase=self.nbase,
                                                  min_prob=self.min_prob, workspace=self._ws, lp_dump=lp_dump)
            if lp_dump is not None:
                raise ValueError("lp_dump needs the fused decoder (csrc/softmax_viterbi.hip does not cover this network)")
            logits, stats, ld = last.logits_and_stats(hid)
            T, B = hid.shape[0], hid.shape[1]
            return decode.viterbi_logits_batch(logits, stats, self.kmer_len, T, B, ld=ld, skip_pen=self.skip,
                                               nbase=self.nbase, min_prob=self.min_prob, workspace=self._ws)
        post = self.posteriors(chunks)
        return decode.viterbi_batch(post, self.kmer_len, skip_pen=self.skip, nbase=self.nbase,
                                    min_prob=self.min_prob, workspace=self._ws)

    def call_bases(self, chunks, alphabet='ACGT'):
        """call_chunks + states -> bases on the device (what basecall.SeqPrinter.write does per read, basecall.py:157-163,
        with always_move as for a transducer model): -> (scores device [B], list of B base strings)."""
        from . import bio
        scores, paths, lens = self.call_chunks(chunks)
        return scores, bio.paths_to_bases(paths, lens, self.kmer_len, alphabet, always_move=True)

    def call_reads(self, signals, trim=(0, 0), open_pore_fraction=0.0):
        """Whole reads of different lengths in ONE batch (the reference calls them one at a time, basecall.py:88-121):
        `signals` is a list of 1-D float arrays (already scaled, e.g. fast5.Fast5.get_read()); each is trimmed as raw_worker does, median/MAD
        normalised over its own length, zero-padded to the longest, and the network + decoder run on the padded batch
        with per-read lengths (layers.ragged), so every read gets exactly what a batch-1 call would give.
        -> device tensors (scores [B], paths [B, T'max] (-1 padded), lens [B]) and the per-read sample counts."""
        import torch
        from . import device as D, util
        net = self.network
        if not isinstance(net, layers.Serial) or type(net.layers[-1]) is not layers.Softmax:
            raise ValueError("call_reads needs a Serial network ending in a Softmax layer")
        # basecall.py:111-112: trim_open_pore (which also cuts the read to whole 100-sample windows), then trim_array
        sigs = [util.trim_array(np.asarray(batch.trim_open_pore(np.asarray(s, dtype=np.float32), open_pore_fraction)), *trim)
                for s in signals]
        nsamp = [len(s) for s in sigs]
        if min(nsamp) < 1:
            raise ValueError("empty read after trimming")
        B, lmax = len(sigs), max(nsamp)
        x = torch.zeros((lmax, B, 1), dtype=torch.float32, device=D.device())
        for b, sig in enumerate(sigs):                       # per-read normalisation (basecall.py:117-118)
            x[:nsamp[b], b:b + 1, :] = batch.normalise_chunks(D.to_dev(sig).reshape(1, -1), 'per-chunk', out_layout='network')
        with layers.ragged(nsamp) as ctx:
            hid = x
            for layer in net.layers[:-1]:
                hid = layer._forward(hid, None, False)
            lengths = layers.ragged.current
            pack = self._fused_pack(net.layers[-1], hid)
            if pack is None:
                logits, stats, ld = net.layers[-1].logits_and_stats(hid)
        T = hid.shape[0]
        if pack is not None:
            scores, paths, lens = decode.viterbi_fused_batch(hid, pack, self.kmer_len, skip_pen=self.skip, nbase=self.nbase,
                                                             min_prob=self.min_prob, workspace=self._ws,
                                                             lengths=lengths.contiguous())
        else:
            scores, paths, lens = decode.viterbi_logits_batch(logits, stats, self.kmer_len, T, B, ld=ld, skip_pen=self.skip,
                                                              nbase=self.nbase, min_prob=self.min_prob, workspace=self._ws,
                                                              lengths=lengths.contiguous())
        return scores, paths, lens, nsamp

    def call_chunks_host(self, chunks):
        scores, paths, lens = self.call_chunks(chunks)
        scores, paths, lens = scores.cpu().numpy(), paths.cpu().numpy(), lens.cpu().numpy()
        return scores, [paths[i, : lens[i]].tolist() for i in range(len(lens))]


def synthetic_chunks(nchunk, chunk_len=4000, seed=0xdeadbeef, dwell=10.0, noise=0.15, first_chunk=0):
    """Synthetic raw signal: per chunk, piecewise-constant levels ~N(0,1) with geometric dwell
    (mean `dwell` samples) plus N(0, noise^2), RandomState(seed + chunk_id), scaled to a pA-like range so that
    normalisation does real work.  float32 [nchunk, chunk_len]."""
    out = np.empty((nchunk, chunk_len), dtype=np.float32)
    for c in range(nchunk):
        rs = np.random.RandomState((seed + first_chunk + c) % (2 ** 32))
        nseg = int(chunk_len / dwell * 2) + 16
        d = rs.geometric(1.0 / dwell, size=nseg)
        while d.sum() < chunk_len:
            d = np.concatenate([d, rs.geometric(1.0 / dwell, size=nseg)])
        levels = rs.normal(size=len(d))
        sig = np.repeat(levels, d)[:chunk_len]
        sig = sig + rs.normal(scale=noise, size=chunk_len)
        out[c] = (sig * 12.0 + 90.0).astype(np.float32)
    return out
