"""Worker-level API of the basecalling path (sloika/basecall.py), batched over chunks.

    decode_post(post, kmer_len, transducer, bad, min_prob, skip, ...)   basecall.py:26-51
    raw_chunk_worker(calc_post, chunks, ...)                            batched counterpart of raw_worker :88-121
    SeqPrinter                                                          basecall.py:124-163
"""
import sys

import numpy as np

from . import bio, decode
from .variables import DEFAULT_ALPHABET, nstate


def decode_post(post, kmer_len, transducer=True, bad=True, min_prob=1e-5, skip=5.0, trans=None, nbase=4,
                eta=1e-10):
    """Decode Viterbi state sequence from a [T,1,nstate] posterior (basecall.py:26-51): (score, path)."""
    if not transducer:
        raise NotImplementedError("the non-transducer decoder (sloika/olddecode.py) is outside the accelerated "
                                  "path; bin/basecall_network.py defaults to --transducer")
    if post.shape[2] != nstate(kmer_len, transducer=transducer, bad_state=bad, nbase=nbase):
        raise ValueError("posterior does not have nstate(kmer_len) states")        # basecall.py:43
    if post.shape[1] != 1:
        raise ValueError("decode_post takes one read: [time, 1, state] (np.squeeze(axis=1), decode.py:30)")
    scores, paths, lens = decode.viterbi_batch(post, kmer_len, skip_pen=skip, nbase=nbase, min_prob=min_prob)
    n = int(lens[0].item())
    return np.float32(scores[0].item()), [int(v) for v in paths[0, :n].cpu().numpy()]


def decode_post_batch(post, kmer_len, min_prob=1e-5, skip=5.0, nbase=4, workspace=None):
    """Batched decode_post over the batch axis of [T,B,nstate]: device (scores[B], paths[B,T], lens[B])."""
    return decode.viterbi_batch(post, kmer_len, skip_pen=skip, nbase=nbase, min_prob=min_prob, workspace=workspace)


def raw_chunk_worker(calc_post, chunks, kmer_len, min_prob=1e-5, skip=5.0, nbase=4, normalisation='per-chunk',
                     names=None):
    """Batched counterpart of raw_worker (basecall.py:88-121) for equal-length chunks.

    chunks: [nchunk, chunk_len] raw signal.  Returns a list of (name, score, call, n_samples) tuples, one per
    chunk, in order -- the tuple the reference's worker returns per read.
    """
    from . import batch
    inmat = batch.normalise_chunks(chunks, normalisation, out_layout='network')     # basecall.py:117-118
    post = calc_post(inmat)                                                          # basecall.py:119
    scores, paths, lens = decode_post_batch(post, kmer_len, min_prob=min_prob, skip=skip, nbase=nbase)
    scores, paths, lens = scores.cpu().numpy(), paths.cpu().numpy(), lens.cpu().numpy()
    res = []
    for i in range(paths.shape[0]):
        name = names[i] if names is not None else "chunk_%d" % i
        res.append((name, scores[i], [int(v) for v in paths[i, : lens[i]]], int(np.shape(chunks)[1])))
    return res


def raw_read_worker(calc_post, signal, trim=(200, 10), open_pore_fraction=0.0, kmer_len=5, min_prob=1e-5, skip=5.0,
                    nbase=4, name="read"):
    """The array part of raw_worker (basecall.py:110-121) for ONE whole read held in memory (the reference reads it
    from a fast5 file first): trim_open_pore, trim_array, per-read median/MAD normalisation, calc_post on [T,1,1],
    decode_post.  Returns (name, score, call, n_samples) or None for an empty read, like the reference."""
    from . import batch, util
    signal = batch.trim_open_pore(signal, open_pore_fraction)              # basecall.py:111
    signal = util.trim_array(signal, *trim)                                 # basecall.py:112
    if len(signal) == 0:
        sys.stderr.write("Read too short in {}\n".format(name))
        return None
    inmat = batch.normalise_chunks(signal.reshape(1, -1), 'per-chunk', out_layout='network')   # basecall.py:117-118
    post = calc_post(inmat)
    score, call = decode_post(post, kmer_len, True, True, min_prob, skip=skip, nbase=nbase)
    return name, score, call, int(inmat.shape[0])


def raw_worker(fast5_file_name, trim, open_pore_fraction, kmer_len, transducer, bad, min_prob, alphabet=DEFAULT_ALPHABET,
               skip=5.0, trans=None, calc_post=None):
    """Worker for basecalling one single-read fast5 file from raw data (basecall.py:88-121), same arguments and return
    value `(name, score, call, n_samples)` / `None`.  The file is read with sloika_amd.fast5 (no h5py/libhdf5 needed);
    `calc_post` is the compiled model (the reference keeps it in a process global set by init_worker, :12-23)."""
    import os
    from . import fast5
    if calc_post is None:
        calc_post = globals().get("calc_post")
    if calc_post is None:
        raise ValueError("raw_worker needs a compiled model: pass calc_post= or call init_worker first")
    if trans is not None or not transducer or not bad:
        raise NotImplementedError("only the transducer decode with a bad-state column is on the GPU path")
    try:
        signal = fast5.Fast5(fast5_file_name).get_read(raw=True)
        sn = os.path.splitext(os.path.basename(fast5_file_name))[0]
    except Exception as e:                                           # basecall.py:105-107
        sys.stderr.write("Error getting raw data for file {}\n{!r}\n".format(fast5_file_name, e))
        return None
    return raw_read_worker(calc_post, signal, trim=trim, open_pore_fraction=open_pore_fraction, kmer_len=kmer_len,
                           min_prob=min_prob, skip=skip, nbase=len(alphabet), name=sn)


def init_worker(model):
    """Set the process-global `calc_post` (basecall.py:12-23): `model` is a model file name or a Layer."""
    global calc_post
    from . import helpers, layers
    net = model if isinstance(model, layers.Layer) else helpers.load_model(model)
    calc_post = net.compile()


class SeqPrinter(object):
    """Formats fasta strings and writes them to stdout or file (basecall.py:124-163).

    :param kmer_len: length of kmer to use for converting states to kmers
    :param datatype: collective noun for data used as model input, e.g. "events" or "samples"
    :param transducer: if True then transitions from a kmer back to itself are not allowed when converting
        kmers to a sequence
    :param fname: name of output file or None to use sys.stdout
    :param alphabet: str alphabet (bin/basecall_network.py:93-94 passes a str; the bytes default of the
        reference raises TypeError inside kmers_to_sequence, so bytes are decoded here)
    """

    def __init__(self, kmer_len, datatype="events", transducer=False, fname=None, alphabet=DEFAULT_ALPHABET):
        if isinstance(alphabet, bytes):
            alphabet = alphabet.decode('utf-8')
        self.kmer_len, self.alphabet = kmer_len, alphabet
        self.transducer = transducer
        self.datatype = datatype
        if fname is None:
            self.fh = sys.stdout
            self.close_fh = False
        else:
            self.fh = open(fname, 'w')
            self.close_fh = True

    def __del__(self):
        if getattr(self, "close_fh", False):
            self.fh.close()

    def write(self, read_name, score, call, nev):
        seq = bio.states_to_sequence(call, self.kmer_len, self.alphabet, always_move=self.transducer)
        self.fh.write(">{} score {:.0f}, {} {} to {} bases\n".format(read_name, score, nev, self.datatype, len(seq)))
        self.fh.write(seq + '\n')
        return len(seq)
