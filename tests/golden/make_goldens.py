#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE in the build container.

Run (only where /root/reference exists; never on the GPU box, never from tests):

    python tests/golden/make_goldens.py

What it does
  * puts /root/reference on sys.path together with throw-away stub modules for the packages the
    reference imports but that are absent here (theano -> only `config.floatX`, h5py, Bio,
    fast5_research); none of the stubbed functionality is exercised;
  * builds the reference's only native unit, sloika/viterbi_helpers.pyx, in a temp dir from a copy
    with the two numpy-2 dtype tokens substituted (`np.int` -> `np.int64`, `np.int_t` -> `np.int64_t`),
    and appends that dir to `sloika.__path__`.  Nothing of it is written into this repository;
  * `np.int = int` is restored for the lifetime of this process (numpy >= 1.24 removed the alias the
    reference still uses in tools/chunkify_raw.py);
  * calls the reference functions on seeded inputs and stores inputs + outputs as .npz/.json data.

Only data (arrays, numbers, strings produced by the reference) is committed; no reference source.
"""
import hashlib
import json
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np

REF = os.environ.get("SLOIKA_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _setup_reference():
    tmp = tempfile.mkdtemp(prefix="sloika_ref_stub_")
    for name, body in {
        "theano/__init__.py": "class _C:\n    floatX = 'float32'\nconfig = _C()\n",
        "h5py/__init__.py": "",
        "Bio/__init__.py": "from . import SeqIO\n",
        "Bio/SeqIO.py": "",
        "fast5_research/__init__.py": "class Fast5:\n    pass\ndef iterate_fast5(*a, **k):\n    return []\n",
    }.items():
        path = os.path.join(tmp, name)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            fh.write(body)
    vh = os.path.join(tmp, "vh")
    os.makedirs(vh)
    with open(os.path.join(REF, "sloika", "viterbi_helpers.pyx")) as fh:
        src = fh.read()
    src = src.replace("np.int_t", "np.int64_t").replace("ITYPE = np.int\n", "ITYPE = np.int64\n")
    with open(os.path.join(vh, "viterbi_helpers.pyx"), "w") as fh:
        fh.write(src)
    with open(os.path.join(vh, "setup.py"), "w") as fh:
        fh.write("from setuptools import setup, Extension\nfrom Cython.Build import cythonize\nimport numpy\n"
                 "setup(ext_modules=cythonize([Extension('viterbi_helpers', ['viterbi_helpers.pyx'],"
                 " include_dirs=[numpy.get_include()])]))\n")
    subprocess.check_call([sys.executable, "setup.py", "-q", "build_ext", "--inplace"], cwd=vh,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "test", "unit"))
    sys.path.insert(0, tmp)
    if not hasattr(np, "int"):
        np.int = int
    import sloika
    sloika.__path__.append(vh)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def dirichlet_post(seed, nev, nst, alpha=0.05, dtype=np.float32):
    """Seeded posterior matrix; legacy RandomState streams are frozen by numpy's compatibility policy."""
    return np.random.RandomState(seed).dirichlet(np.ones(nst) * alpha, size=nev).astype(dtype)


def tie_post(seed, nev, nst, levels):
    """Unnormalised 'posterior' drawn from a few exactly representable levels -> many exact ties."""
    rs = np.random.RandomState(seed)
    return np.asarray(levels, dtype=np.float32)[rs.randint(0, len(levels), size=(nev, nst))]


def synth_signal(seed, n, dwell=10.0, noise=0.15):
    """Piecewise-constant levels ~N(0,1) with geometric dwell plus N(0, noise^2): SURVEY.md 8(d)."""
    rs = np.random.RandomState(seed)
    out = np.empty(n, dtype=np.float32)
    i = 0
    while i < n:
        d = rs.geometric(1.0 / dwell)
        out[i:i + d] = rs.normal()
        i += d
    out += rs.normal(scale=noise, size=n).astype(np.float32)
    return (out * 12.0 + 90.0).astype(np.float32)       # pA-like offset/scale so normalisation is non-trivial


def main():
    _setup_reference()
    from sloika import decode, transducer, viterbi_helpers, batch, basecall, bio, maths, util
    from sloika.tools import chunkify_raw
    import test_decode

    # ------------------------------------------------------------------ decode.viterbi
    test_decode.TestDecode.setUpClass()
    test_decode.TestDecodeModifiedBases.setUpClass()
    T = test_decode.TestDecode
    M = test_decode.TestDecodeModifiedBases
    dec = {"kat_post": T.post, "kat_post3": T.post3, "kat_labels": T.labels, "kat_bases": T.bases,
           "kat_mod_post": M.post, "kat_mod_seq": np.asarray(M.seq)}
    cases = []

    def add_case(name, post, klen, skip, nbase=4, log=False, store=True, gen=None):
        score, path = decode.viterbi(post, klen, skip_pen=skip, log=log, nbase=nbase)
        c = {"name": name, "klen": klen, "skip_pen": skip, "nbase": nbase, "log": log,
             "dtype": str(post.dtype), "shape": list(post.shape), "sha256": sha(post),
             "score_repr": repr(float(score)), "score_hex": float(score).hex(), "gen": gen}
        dec["path_" + name] = np.asarray(path, dtype=np.int32)
        if store:
            dec["post_" + name] = post
        cases.append(c)

    add_case("kat3_skip0", T.post3, 3, 0.0)
    add_case("kat3_skip3", T.post3, 3, 3.0)
    add_case("kat_mod5", M.post, 3, 5.0, nbase=5)
    p50 = dirichlet_post(1, 50, 1025)
    dec["post_d50"] = p50
    for skip in (0.0, 3.0, 5.0):
        add_case("d50_skip%g" % skip, p50, 5, skip, store=False, gen={"kind": "stored", "key": "post_d50"})
    p800 = dirichlet_post(2, 800, 1025)
    for skip in (0.0, 5.0):
        add_case("d800_skip%g" % skip, p800, 5, skip, store=False,
                 gen={"kind": "dirichlet", "seed": 2, "nev": 800, "nst": 1025, "alpha": 0.05})
    p2000 = dirichlet_post(3, 2000, 1025, alpha=0.5)
    add_case("d2000_skip0", p2000, 5, 0.0, store=False,
             gen={"kind": "dirichlet", "seed": 3, "nev": 2000, "nst": 1025, "alpha": 0.5})
    add_case("tie_k3", tie_post(4, 60, 65, [0.5, 0.25, 0.125]), 3, 0.0)
    add_case("tie_k3_skip2", tie_post(5, 60, 65, [0.5, 0.25, 0.125]), 3, 2.0)
    add_case("tie_k5", tie_post(6, 40, 1025, [0.5, 0.25]), 5, 0.0, store=False,
             gen={"kind": "tie", "seed": 6, "nev": 40, "nst": 1025, "levels": [0.5, 0.25]})
    add_case("const_k3", np.full((12, 65), 0.125, dtype=np.float32), 3, 0.0)      # every comparison ties
    add_case("k4_f32", dirichlet_post(7, 120, 257, alpha=0.3), 4, 1.5)
    add_case("k3_f64", dirichlet_post(8, 30, 65, alpha=0.3, dtype=np.float64), 3, 0.5)
    add_case("k3_log", np.log(dirichlet_post(9, 25, 65, alpha=0.5) + 1e-3), 3, 4.0, log=True)
    add_case("k3_T1", dirichlet_post(10, 1, 65, alpha=0.5), 3, 0.0)
    add_case("k3_T2", dirichlet_post(11, 2, 65, alpha=0.5), 3, 0.0)
    add_case("nb5_k3", dirichlet_post(12, 40, 126, alpha=0.2), 3, 1.0, nbase=5)
    dec["argmax_kat"] = decode.argmax(T.post.copy(), zero_is_blank=False)
    np.savez_compressed(os.path.join(OUT, "decode.npz"), **dec)

    # ------------------------------------------------------------------ prepare_post / decode_post
    pp_in = dirichlet_post(20, 20, 65, alpha=0.5)[:, None, :]
    pp = {"pp_in": pp_in, "pp_out": decode.prepare_post(pp_in, min_prob=1e-5),
          "pp_out_1e3": decode.prepare_post(pp_in, min_prob=1e-3)}
    dp_in = dirichlet_post(21, 60, 1025)[:, None, :]
    for skip in (0.0, 5.0):
        score, call = basecall.decode_post(dp_in, 5, True, True, 1e-5, skip=skip)
        pp["dp_call_skip%g" % skip] = np.asarray(call, dtype=np.int32)
        pp["dp_score_skip%g" % skip] = np.float64(score)
    pp["dp_in"] = dp_in
    np.savez_compressed(os.path.join(OUT, "prepare_post.npz"), **pp)

    # ------------------------------------------------------------------ slip_update / map_to_sequence
    tr = {}
    for n in (3, 4, 10, 400):
        x = np.random.RandomState(0xdeadbeef % (2 ** 31) + n).normal(size=n).astype(np.float32)
        for slip in (0.0, 5.0):
            fs, fp = viterbi_helpers.slip_update(x, slip)
            tr["slip_x_%d" % n] = x
            tr["slip_fs_%d_%g" % (n, slip)] = fs
            tr["slip_fp_%d_%g" % (n, slip)] = fp.astype(np.int64)
    # exact-tie input for the `>=` rule (viterbi_helpers.pyx:27)
    xt = np.asarray([1, 1, 1, 2, 2, 1, 3, 3, 3, 0], dtype=np.float32)
    fs, fp = viterbi_helpers.slip_update(xt, 0.0)
    tr["slip_x_tie"], tr["slip_fs_tie"], tr["slip_fp_tie"] = xt, fs, fp.astype(np.int64)

    mcases = []

    def add_map(name, trans, seq, slip, pi, pf, log, store=True, gen=None):
        score, path = transducer.map_to_sequence(trans, seq, slip=slip, prior_initial=pi, prior_final=pf, log=log)
        tr["map_path_" + name] = np.asarray(path, dtype=np.int32)
        tr["map_seq_" + name] = np.asarray(seq, dtype=np.int32)
        if store:
            tr["map_trans_" + name] = trans
        if pi is not None:
            tr["map_pi_" + name] = pi
        if pf is not None:
            tr["map_pf_" + name] = pf
        mcases.append({"name": name, "slip": slip, "log": log, "score_hex": float(score).hex(),
                       "shape": list(trans.shape), "sha256": sha(trans), "gen": gen,
                       "has_pi": pi is not None, "has_pf": pf is not None})

    rs = np.random.RandomState(30)
    post100 = dirichlet_post(31, 100, 65, alpha=0.5)
    seq30 = rs.randint(1, 65, size=30)
    add_map("m100_log", np.log(post100), seq30, 5.0, None, None, True)
    add_map("m100_post", post100, seq30, 5.0, None, None, False)
    add_map("m100_slip0", post100, seq30, 0.0, None, None, False, store=False, gen={"key": "map_trans_m100_post"})
    pi = util.geometric_prior(30, 2.0)
    pf = util.geometric_prior(30, 2.0, rev=True)
    add_map("m100_priors", post100, seq30, 3.0, pi, pf, False, store=False, gen={"key": "map_trans_m100_post"})
    add_map("m100_pi_only", post100, seq30, 3.0, pi, None, False, store=False, gen={"key": "map_trans_m100_post"})
    seq3 = rs.randint(1, 65, size=3)
    add_map("m100_npos3", post100, seq3, 1.0, None, None, False, store=False, gen={"key": "map_trans_m100_post"})
    post800 = dirichlet_post(32, 800, 1025)
    seq400 = rs.randint(1, 1025, size=400)
    add_map("m800", decode.prepare_post(post800[:, None, :]), seq400, 5.0, util.geometric_prior(400, 50.0),
            util.geometric_prior(400, 50.0, rev=True), False, store=False,
            gen={"kind": "prepare_post(dirichlet)", "seed": 32, "nev": 800, "nst": 1025, "alpha": 0.05})
    tie_tr = np.log(tie_post(33, 50, 65, [0.5, 0.25, 0.125]))
    add_map("mtie", tie_tr, rs.randint(1, 65, size=20), 0.0, None, None, True)
    tr["geometric_prior_30_2"] = pi
    tr["geometric_prior_30_2_rev"] = pf
    np.savez_compressed(os.path.join(OUT, "transducer.npz"), **tr)

    # ------------------------------------------------------------------ signal front end
    sig = {}
    signal = synth_signal(40, 5 * 4000 + 123)
    sig["signal"] = signal
    for frac in (0.0, 0.3):
        trimmed = batch.trim_open_pore(signal, frac)
        start = int(np.flatnonzero(signal == trimmed[0])[0]) if trimmed.size else 0
        # locate the slice exactly (trim_open_pore returns a view)
        start = (trimmed.__array_interface__["data"][0] - signal.__array_interface__["data"][0]) // 4
        sig["trim_open_pore_%g" % frac] = np.asarray([start, start + len(trimmed)], dtype=np.int64)
        trimmed = batch.trim_open_pore(signal, frac, var_method='std')
        start = (trimmed.__array_interface__["data"][0] - signal.__array_interface__["data"][0]) // 4
        sig["trim_open_pore_std_%g" % frac] = np.asarray([start, start + len(trimmed)], dtype=np.int64)
    med, mad = maths.med_mad(signal)
    sig["med_mad_read"] = np.asarray([med, mad], dtype=np.float32)
    # basecall.py:117-118 maths on the whole read
    inmat = (signal - np.median(signal)) / maths.mad(signal)
    sig["read_norm"] = inmat.astype(np.float32)
    # raw_chunkify through a hand-built, registered mapping table (labels are not used here)
    chunk_len, klen = 4000, 5
    batch.init_chunk_identity_worker(klen, b"ACGT")
    nblock = len(signal) // 50
    mt = np.zeros(nblock, dtype=[("start", "<i8"), ("length", "<i8"), ("move", "<i8"), ("kmer", "S5")])
    mt["start"] = np.arange(nblock) * 50
    mt["length"] = 50
    mt["length"][-1] = len(signal) - mt["start"][-1]
    kmers = bio.all_kmers(klen, b"ACGT")
    ks = np.random.RandomState(41).randint(0, 1024, size=nblock)
    mt["kmer"] = [kmers[i] for i in ks]
    mt["move"] = np.random.RandomState(42).randint(0, 2, size=nblock)
    mt["move"][0] = 1
    for norm in ("per-chunk", "per-read", "none"):
        inMat, labels, bad = chunkify_raw.raw_chunkify(signal, mt, chunk_len, klen, norm, 5, False)
        sig["chunks_" + norm.replace("-", "_")] = np.ascontiguousarray(inMat[:, :, 0]).astype(np.float32)
        sig["chunks_dtype_" + norm.replace("-", "_")] = np.asarray(str(inMat.dtype))
    sig["trim_array_200_10"] = np.asarray([200, len(signal) - 10], dtype=np.int64)   # util.trim_array(signal, 200, 10)
    assert np.array_equal(util.trim_array(signal, 200, 10), signal[200:-10])
    np.savez_compressed(os.path.join(OUT, "signal.npz"), **sig)

    # ------------------------------------------------------------------ bio / SeqPrinter
    bj = {"all_kmers_3_head": bio.all_kmers(3)[:8], "cases": []}
    for name in ("d50_skip0", "d50_skip5", "d800_skip0"):
        path = dec["path_" + name]
        kmers_s = bio.all_kmers(5, "ACGT")
        kp = [kmers_s[i] for i in path]
        bj["cases"].append({"name": name, "seq_always_move": bio.kmers_to_sequence(kp, always_move=True),
                            "seq_allow_stay": bio.kmers_to_sequence(kp, always_move=False)})
    bj["kat"] = {"kmers": ["AAC", "ACT", "ACT", "CTG", "GGA"],          # no overlap at the end -> full append
                 "always_move": bio.kmers_to_sequence(["AAC", "ACT", "ACT", "CTG", "GGA"], always_move=True),
                 "allow_stay": bio.kmers_to_sequence(["AAC", "ACT", "ACT", "CTG", "GGA"], always_move=False)}
    import io
    buf = io.StringIO()
    sp = basecall.SeqPrinter(5, datatype="samples", transducer=True, alphabet="ACGT")
    sp.fh = buf
    nb = sp.write("read_x", -206.52707, [int(v) for v in dec["path_d50_skip0"]], 250)
    bj["seqprinter"] = {"text": buf.getvalue(), "nbases": nb, "read_name": "read_x", "score": -206.52707,
                        "nev": 250, "path_key": "path_d50_skip0"}
    with open(os.path.join(OUT, "bio.json"), "w") as fh:
        json.dump(bj, fh, indent=1)

    # ------------------------------------------------------------------ pretrained.pkl weights
    class Holder:
        def __init__(self, *a, **k):
            pass

        def __setstate__(self, state):
            self.__dict__.update(state if isinstance(state, dict) else {"state": state})

    class U(pickle.Unpickler):
        def find_class(self, module, name):
            if module.startswith("theano") or module.startswith("sloika"):
                return type(name, (Holder,), {"_mod": module})
            return super().find_class(module, name)

    with open(os.path.join(REF, "models", "pretrained.pkl"), "rb") as fh:
        net = U(fh, encoding="latin1").load()

    def value(shared):
        return np.asarray(shared.container.storage[0], dtype=np.float32)

    w = {}
    desc = []
    for i, layer in enumerate(net.layers):
        lname = type(layer).__name__
        inner = layer
        rev = False
        if lname == "Reverse":
            inner, rev = layer.layer, True
        iname = type(inner).__name__
        d = {"index": i, "type": iname, "reverse": rev}
        for attr in ("W", "b", "iW", "sW", "sW2", "p"):
            if hasattr(inner, attr):
                w["l%d_%s" % (i, attr)] = value(getattr(inner, attr))
        for attr in ("_insize", "_size", "winlen", "stride", "padding", "padding_mode", "has_bias"):
            if hasattr(inner, attr):
                v = getattr(inner, attr)
                if isinstance(v, (tuple, list)):
                    v = [int(q) for q in v]
                elif isinstance(v, (bool, np.bool_)):
                    v = bool(v)
                elif isinstance(v, (int, np.integer)):
                    v = int(v)
                d[attr] = v
        for attr in ("fun", "gatefun"):
            if hasattr(inner, attr):
                f = getattr(inner, attr)
                d[attr] = getattr(f, "__name__", type(f).__name__)
        desc.append(d)
    w["description_json"] = np.asarray(json.dumps(desc))
    np.savez_compressed(os.path.join(OUT, "pretrained_weights.npz"), **w)

    meta = {"decode_cases": cases, "map_cases": mcases, "numpy": np.__version__,
            "reference": "nanoporetech/sloika @ /root/reference"}
    with open(os.path.join(OUT, "cases.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    for f in sorted(os.listdir(OUT)):
        print("%-28s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
