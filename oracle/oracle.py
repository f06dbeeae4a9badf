"""ctypes front-end of the CPU oracle (oracle/sloika_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, by `__graft_entry__.smoke()` and by the
`cpu_baseline` leg of bench.py -- never by the product package `sloika_amd`.

Python-level functions mirror the reference signatures they restate:

    viterbi(post, klen, skip_pen, log, nbase)        sloika/decode.py:39-93
    prepare_post(post, min_prob)                     sloika/decode.py:21-36
    slip_update(x, slip)                             sloika/viterbi_helpers.pyx:12-35
    map_to_sequence(trans, sequence, slip, ...)      sloika/transducer.py:14-73
    med_mad_normalise(chunks)                        sloika/tools/chunkify_raw.py:178-181
    run_network(spec, x)                             sloika/layers.py (Serial/Parallel/Reverse/...)

`spec` is a neutral nested-dict description of a layer graph (see `run_network`).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
#: SLOIKA_ORACLE_LIB: another build of the same source (oracle/Makefile: `asan`, `native`)
_LIB_PATH = os.environ.get("SLOIKA_ORACLE_LIB") or os.path.join(_HERE, "_build", "liboracle.so")

ACTIVATIONS = ["linear", "tanh", "sigmoid", "elu", "relu", "relu_smooth", "softplus", "exp", "erf", "L1mL2",
               "fair", "retu", "tanh_pm", "sigmoid_pm", "bounded_linear", "sin", "cauchy", "geman_mcclure",
               "welsh"]
ACT_ID = {name: i for i, name in enumerate(ACTIVATIONS)}


def build(force=False):
    """Compile oracle/_build/liboracle.so with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "sloika_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


def build_native():
    """oracle/_build/liboracle_native.so: the same source compiled -O3 -march=native ON THIS MACHINE (always rebuilt: the file of another
    machine may not run here).  For the CPU baseline of bench.py only; None when the compiler is missing."""
    try:
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "native"])
    except (OSError, subprocess.CalledProcessError):
        return None
    return os.path.join(_HERE, "_build", "liboracle_native.so")


def use_library(path=None):
    """Switch the build of the oracle this process calls (None: the default portable build)."""
    global _lib, _LIB_PATH
    _LIB_PATH = path or os.path.join(_HERE, "_build", "liboracle.so")
    _lib = None


_lib = None
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_activation_f32.restype = C.c_float
        _lib.orc_activation_f32.argtypes = [C.c_int, C.c_float]
        _lib.orc_conv1d_out_len.restype = C.c_int
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(C.c_int(int(n)))


def activation(name, x):
    x = np.asarray(x, dtype=np.float32)
    out = np.empty_like(x)
    l = lib()
    flat_in, flat_out = x.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = l.orc_activation_f32(ACT_ID[name], float(flat_in[i]))
    return out


# ---------------------------------------------------------------------------------------------
# layers
# ---------------------------------------------------------------------------------------------
def conv1d(x, W, b, stride, padding, act):
    x, W, b = _f32(x), _f32(W), _f32(b)
    T, B, Cin = x.shape
    Cout, Cin2, winlen = W.shape
    assert Cin == Cin2
    Tout = lib().orc_conv1d_out_len(T, winlen, stride, padding[0], padding[1])
    y = np.empty((Tout, B, Cout), dtype=np.float32)
    lib().orc_conv1d_f32(_p(x, _f32p), T, B, Cin, _p(W, _f32p), _p(b, _f32p), Cout, winlen, stride,
                         padding[0], padding[1], ACT_ID[act], _p(y, _f32p))
    return y


def window(x, w):
    x = _f32(x)
    T, B, F = x.shape
    y = np.empty((T, B, w * F), dtype=np.float32)
    lib().orc_window_f32(_p(x, _f32p), T, B, F, w, _p(y, _f32p))
    return y


def feedforward(x, W, b, act):
    x, W, b = _f32(x), _f32(W), _f32(b)
    T, B, I = x.shape
    N = W.shape[0]
    y = np.empty((T, B, N), dtype=np.float32)
    lib().orc_feedforward_f32(_p(x, _f32p), C.c_size_t(T * B), I, _p(W, _f32p), _p(b, _f32p), N, ACT_ID[act],
                              _p(y, _f32p))
    return y


def softmax(x, W, b):
    x, W, b = _f32(x), _f32(W), _f32(b)
    T, B, I = x.shape
    N = W.shape[0]
    y = np.empty((T, B, N), dtype=np.float32)
    lib().orc_softmax_f32(_p(x, _f32p), C.c_size_t(T * B), I, _p(W, _f32p), _p(b, _f32p), N, _p(y, _f32p))
    return y


def gru(x, iW, sW, sW2, b, act="tanh", gate="sigmoid", reverse=False):
    x, iW, sW, sW2, b = _f32(x), _f32(iW), _f32(sW), _f32(sW2), _f32(b)
    T, B, I = x.shape
    n = sW2.shape[0]
    assert iW.shape == (3 * n, I) and sW.shape == (2 * n, n)
    y = np.empty((T, B, n), dtype=np.float32)
    lib().orc_gru_f32(_p(x, _f32p), T, B, I, _p(iW, _f32p), _p(sW, _f32p), _p(sW2, _f32p), _p(b, _f32p), n,
                      int(reverse), ACT_ID[act], ACT_ID[gate], _p(y, _f32p))
    return y


def lstm(x, iW, sW, b, p, act="tanh", gate="sigmoid", reverse=False):
    x, iW, sW, b, p = _f32(x), _f32(iW), _f32(sW), _f32(b), _f32(p)
    T, B, I = x.shape
    n = sW.shape[1]
    assert iW.shape == (4 * n, I) and sW.shape == (4 * n, n)
    y = np.empty((T, B, n), dtype=np.float32)
    lib().orc_lstm_f32(_p(x, _f32p), T, B, I, _p(iW, _f32p), _p(sW, _f32p), _p(b, _f32p), _p(p, _f32p), n,
                       int(reverse), ACT_ID[act], ACT_ID[gate], _p(y, _f32p))
    return y


def run_network(spec, x, reverse=False):
    """Evaluate a layer graph.  `spec` is a dict with key 'type' in
    serial / parallel / reverse / convolution / window / feed-forward / softmax / GRU / LSTM
    (the type strings of the reference's Layer.json(), layers.py), weights as numpy arrays in the
    layout the reference's `step`/`run` code reads them.
    """
    t = spec["type"]
    if t == "serial":                        # layers.py:1556-1560
        assert not reverse
        for sub in spec["sublayers"]:
            x = run_network(sub, x)
        return x
    if t == "parallel":                      # layers.py:1486-1487
        outs = [run_network(sub, x, reverse) for sub in spec["sublayers"]]
        return np.concatenate(outs, axis=2)
    if t == "reverse":                       # layers.py:1449-1450
        return run_network(spec["sublayer"], x, not reverse)
    if t == "GRU":
        return gru(x, spec["iW"], spec["sW"], spec["sW2"], spec.get("b"), spec["activation"], spec["gate"], reverse)
    if t == "LSTM":
        return lstm(x, spec["iW"], spec["sW"], spec.get("b"), spec.get("p"), spec["activation"], spec["gate"],
                    reverse)
    # time-local layers: reversal of input and output cancels exactly for padding-symmetric ops,
    # but for generality follow the definition literally.
    if reverse:
        return run_network(spec, x[::-1])[::-1]
    if t == "convolution":
        return conv1d(x, spec["W"], spec.get("b"), spec["stride"], tuple(spec["padding"]), spec["activation"])
    if t == "window":
        return window(x, spec["w"])
    if t == "feed-forward":
        return feedforward(x, spec["W"], spec.get("b"), spec["activation"])
    if t == "softmax":
        return softmax(x, spec["W"], spec.get("b"))
    raise ValueError("oracle: unsupported layer type %r" % t)


# ---------------------------------------------------------------------------------------------
# chunk front end
# ---------------------------------------------------------------------------------------------
def med_mad_normalise(chunks, return_stats=False):
    """chunks: float32 [nchunk, chunk_len] -> (x - median) / (1.4826 * MAD), per chunk."""
    chunks = _f32(chunks)
    nchunk, chunk_len = chunks.shape
    out = np.empty_like(chunks)
    med = np.empty(nchunk, dtype=np.float32)
    mad = np.empty(nchunk, dtype=np.float32)
    lib().orc_med_mad_normalise_f32(_p(chunks, _f32p), nchunk, chunk_len, _p(out, _f32p), _p(med, _f32p),
                                    _p(mad, _f32p))
    return (out, med, mad) if return_stats else out


# ---------------------------------------------------------------------------------------------
# decode
# ---------------------------------------------------------------------------------------------
def prepare_post(post, min_prob=1e-5):
    post = _f32(post)
    assert post.ndim == 3 and post.shape[1] == 1     # np.squeeze(post, axis=1), decode.py:30
    post = post[:, 0, :]
    out = np.empty_like(post)
    lib().orc_prepare_post_f32(_p(post, _f32p), C.c_size_t(post.size), C.c_double(min_prob), _p(out, _f32p))
    return out


_ETA = 1e-10


def viterbi(post, klen, skip_pen=0.0, log=False, nbase=4):
    """decode.viterbi (decode.py:39-93).  dtype follows the input (float32 or float64), as numpy
    does in the reference.  Returns (score, list_of_states)."""
    post = np.asarray(post)
    if post.dtype not in (np.float32, np.float64):
        post = post.astype(np.float64)
    nev, nst = post.shape
    assert klen >= 3, "Kmer not long enough to apply Viterbi with skips"
    assert nbase ** klen + 1 == nst
    lpost = np.ascontiguousarray(np.log(post + _ETA) if not log else post)   # decode.py:56
    path = np.empty(nev, dtype=np.int32)
    n = C.c_int32(0)
    if lpost.dtype == np.float32:
        score = C.c_float(0)
        rc = lib().orc_viterbi_kmer_f32(_p(lpost, _f32p), nev, nbase, klen, C.c_double(skip_pen),
                                        C.byref(score), _p(path, _i32p), C.byref(n), None)
        sc = np.float32(score.value)
    else:
        score = C.c_double(0)
        rc = lib().orc_viterbi_kmer_f64(_p(lpost, _f64p), nev, nbase, klen, C.c_double(skip_pen),
                                        C.byref(score), _p(path, _i32p), C.byref(n), None)
        sc = np.float64(score.value)
    assert rc == 0
    return sc, [int(v) for v in path[: n.value]]


def viterbi_batch(lpost, klen, skip_pen=0.0, nbase=4):
    """lpost: float32 [T, B, nstate] log-posteriors -> (scores[B], paths[B,T], lens[B])."""
    lpost = _f32(lpost)
    T, B, nst = lpost.shape
    assert nbase ** klen + 1 == nst
    scores = np.empty(B, dtype=np.float32)
    paths = np.full((B, T), -1, dtype=np.int32)
    lens = np.empty(B, dtype=np.int32)
    rc = lib().orc_viterbi_kmer_batch_f32(_p(lpost, _f32p), T, B, nbase, klen, C.c_double(skip_pen),
                                          _p(scores, _f32p), _p(paths, _i32p), _p(lens, _i32p))
    assert rc == 0
    for b in range(B):
        paths[b, lens[b]:] = -1
    return scores, paths, lens


def slip_update(x, slip):
    x = _f32(x)
    n = x.shape[0]
    fs = np.empty(n, dtype=np.float32)
    fp = np.empty(n, dtype=np.int64)
    rc = lib().orc_slip_update_f32(_p(x, _f32p), n, C.c_float(slip), _p(fs, _f32p), _p(fp, _i64p))
    assert rc == 0, "slip_update needs len(x) >= 3"
    return fs, fp


def map_to_sequence(trans, sequence, slip=None, prior_initial=None, prior_final=None, log=True):
    assert slip is None or slip >= 0.0, 'Slip penalty should be non-negative'
    with np.errstate(all="ignore"):
        slip32 = np.float32(np.nan if slip is None else slip)          # transducer.py:27
    trans = _f32(trans)
    ltrans = np.ascontiguousarray(trans if log else np.log(trans))     # transducer.py:30
    nev, nst = ltrans.shape
    seq = np.ascontiguousarray(sequence, dtype=np.int32)
    npos = seq.shape[0]
    pi = None if prior_initial is None else np.ascontiguousarray(prior_initial, dtype=np.float64)
    pf = None if prior_final is None else np.ascontiguousarray(prior_final, dtype=np.float64)
    path = np.empty(nev, dtype=np.int32)
    score = C.c_float(0)
    rc = lib().orc_map_to_sequence_f32(_p(ltrans, _f32p), nev, nst, _p(seq, _i32p), npos, C.c_float(slip32),
                                       _p(pi, _f64p), _p(pf, _f64p), C.byref(score), _p(path, _i32p), None)
    assert rc == 0
    return np.float32(score.value), path
