"""Layer / model-file API of the basecalling path: the classes of sloika/layers.py that shipped models use,
with the same constructor signatures, attribute names and methods, executing on gfx950 HIP kernels through
the C ABI (include/sloika_amd.h) instead of compiling Theano graphs.

    reference                                   here
    ---------                                   ----
    Layer.compile() -> th.function              Layer.compile() -> callable(ndarray|tensor [T,B,F]) -> same kind
    Layer.run(symbolic)                         Layer.run(device tensor [T,B,F]) -> device tensor [T',B,size]
    th.shared leaves (.get_value/.set_value)    Shared leaves (same two methods, numpy storage + device mirror)
    RNN.run = th.scan(step)                     persistent recurrent kernels (csrc/recurrent.hip)
    Reverse(layer).run = layer.run(x[::-1])[::-1]   time-order flag on the recurrent kernels (no copies)
    Parallel.run = concatenate(...)             sub-layers write straight into their slice (row stride `ld`)

Convention (sloika/layers.py:13-14): tensors are row major (time, batch, feature), float32.
There is no CPU fallback: without the HIP library / a GPU, `run` raises.
"""
import abc
import os
from collections import OrderedDict
from functools import reduce

import numpy as np

from . import _lib, activation, conv, profiler
from .config import sloika_dtype
from .variables import DEFAULT_NBASE, nkmer

_FORGET_BIAS = 2.0        # layers.py:17


def zeros(size):
    return np.zeros(size, dtype=sloika_dtype)          # layers.py:21-22


class Shared(object):
    """Stand-in for a Theano shared variable: numpy storage, lazily mirrored to HBM.  While a training step owns the
    parameter (sloika_amd.train.TrainingStep) the device copy is the master and the numpy side is refreshed from it on
    demand, so `get_value()` and pickling see what the optimiser wrote, as with Theano's updates."""

    def __init__(self, value):
        self._value = np.ascontiguousarray(value, dtype=sloika_dtype)
        self._dev = None
        self._device_is_master = False
        self._version = 0                    # bumped by every set_value: caches derived from the value key on it

    def _pull(self):
        if getattr(self, "_device_is_master", False) and self._dev is not None:
            self._value = np.ascontiguousarray(self._dev.detach().cpu().numpy(), dtype=sloika_dtype)

    def get_value(self, borrow=False):
        self._pull()
        return self._value if borrow else self._value.copy()

    def set_value(self, value, borrow=False):
        value = np.ascontiguousarray(value, dtype=sloika_dtype)
        self._version = getattr(self, "_version", 0) + 1
        if getattr(self, "_device_is_master", False) and self._dev is not None and value.shape == tuple(self._dev.shape):
            import torch
            self._dev.copy_(torch.from_numpy(value))         # keep the optimiser's flat buffer as the storage
            self._value = value
            return
        self._value = value
        self._dev = None

    def dev(self):
        if self._dev is None:
            from . import device as D
            self._dev = D.to_dev(self._value)
        return self._dev

    @property
    def shape(self):
        return self._value.shape

    def __getstate__(self):
        self._pull()
        return {"_value": self._value}

    def __setstate__(self, state):
        self._value = np.ascontiguousarray(state["_value"], dtype=sloika_dtype)
        self._dev = None
        self._device_is_master = False
        self._version = 0


def shared(value):
    return Shared(value)


def _extract(x, shape=None):
    xv = x.get_value()
    if shape is not None:
        xv = xv.reshape(shape)
    return xv.tolist()


_CU_COUNT = {}


def _cu_count(dev):
    """Compute units of `dev`, asked once per device (torch's first get_device_properties initialises amdsmi: 26 ms on the host, which
    used to sit in the first forward pass; pipeline.Basecaller asks at construction)."""
    import torch
    key = (dev.type, dev.index)
    n = _CU_COUNT.get(key)
    if n is None:
        n = _CU_COUNT[key] = torch.cuda.get_device_properties(dev).multi_processor_count
    return n


def _scratch(shape, dtype, dev):
    """Uninitialised device memory for the forward pass in progress (device.scratch: out of the caller's Arena when one is active)."""
    from . import device as D
    return D.scratch(shape, dtype, dev)


def _stream():
    from . import device as D
    return D.stream_ptr()


#: The package has two switches (set in the shell, read once at import):
#:   SLOIKA_AMD_EXACT_F32=1    every product in plain float32 MFMA, no fp16 / bf16 splits anywhere: the correctness fallback
#:                             (two-kernel Gru / Lstm, fp32 softmax projection + decoder on the logits)
#:   SLOIKA_AMD_DEBUG=a,b,...  comparison runs of one plan against another, never needed for results:
#:                             recurrent_f32    split projections, but the recurrences in float32 MFMA (csrc/recurrent.hip, lstm_mfma.hip)
#:                             no_lstm_fused    Lstm as projection GEMM + csrc/lstm_scan16.hip instead of csrc/lstm_fused16.hip
#:                             no_gru64_share   64-wide Gru layers never two workgroups per CU
#:                             no_fused_decode  Softmax writes the logits, csrc/decode.hip reads them (pipeline.FUSED_DECODE)
#:                             bf16_ff_max=N    widest FeedForward output on csrc/gemm_bf16x6.hip
#:                             xent_in_place    training: logits written, then the loss gradient in place over them (train.hip)
#:                                              instead of the two passes that never store logits (train.XENT_TWO_PASS)
_DEBUG = dict((t.partition("=")[0], t.partition("=")[2]) for t in os.environ.get("SLOIKA_AMD_DEBUG", "").split(",") if t)
#: Time-parallel projections (softmax, and the input projections of recurrent layers that have no fused kernel) run on the
#: FP16 matrix pipe as a 3-term split of every float32 operand (csrc/gemm_rows_f16x3.hip: float32 accumulation, error a few
#: float32 ulps, ~5x the fp32-MFMA rate) unless SLOIKA_AMD_EXACT_F32=1.
SPLIT_F16 = os.environ.get("SLOIKA_AMD_EXACT_F32", "0") != "1"
#: ... and so do the recurrent products (csrc/gru_bar16*.hip, gru_scan16.hip, lstm_fused16.hip, lstm_scan16.hip).
RECURRENT_F16 = "recurrent_f32" not in _DEBUG
#: widest FeedForward output that takes csrc/gemm_bf16x6.hip (128 -> 64: 0.78 against 1.10 ms, 192 -> 96: 0.29 against 0.39; 192 -> 128
#: as one 128-column block: 1.69 against 1.72 ms, i.e. no gain, so it stays with the row kernel)
BF16_FF_MAX = int(_DEBUG.get("bf16_ff_max") or 96)
#: Gru layers up to 64 wide run their four-chunk workgroups two per CU where one per CU does not hold what is meant to run together (the
#: directions of a birnn at B = 1024: `baseline_gru` 16.9 -> 14.9 ms per step against the eight-chunk plan on half the chip each)
GRU64_SHARE = "no_gru64_share" not in _DEBUG
#: An Lstm layer of up to 64 units and 64 inputs runs as ONE kernel that computes its input projection inside the scan
#: (csrc/lstm_fused16.hip)
LSTM_FUSED = "no_lstm_fused" not in _DEBUG


class _PlanHints(__import__("threading").local):
    """Execution-plan hints of the forward pass that is running ON THIS HOST THREAD (a Basecaller per thread may run with its
    own `in_flight`: two threads no longer race on a module global).  With `deterministic` set results never depend on them, only speed;
    without it the sixteen-chunk Gru plan may be chosen, whose states agree with the other plans' to float32 rounding only.

    gru_plan_bits  bits 8-9 of the `reverse` argument of slk_gru_bar16_f32 (0 = plan by batch size, 2 / 3 = eight / sixteen
                   chunks per workgroup): set by Parallel while it runs the directions of a birnn side by side at a batch where
                   only the eight-chunk plan lets them share the chip
    in_flight      batches the caller keeps in flight on streams of their own (pipeline.Basecaller(in_flight=N) sets it around
                   a forward pass): a recurrent layer then counts N times its workgroups when it decides whether they fit the
                   device's CUs together"""
    gru_plan_bits = 0
    in_flight = 1
    #: pipeline.Basecaller(deterministic=True), the default: never the sixteen-chunk Gru plan (csrc/gru_bar16q.hip), the only plan whose
    #: hidden states differ from the others' (three-term recurrent products where they take two: agreement to float32 rounding, up to 2 %
    #: of chunks called differently) -- so that what a chunk is called does not depend on batch size or batches in flight
    deterministic = False


_HINTS = _PlanHints()


def _gru_plan_for(B, share, ncu, per_cu=1):
    """Plan bits for `share` Gru launches of batch B that are meant to run at the same time: 0 when their four-chunk
    workgroups fit the CUs together (or nothing helps), 2 when only the eight-chunk ones do, 3 when only the sixteen-chunk ones.
    per_cu = 2 for layers up to 64 wide: their eight- and sixteen-chunk kernels (csrc/gru_bar16d.hip, gru_bar16q.hip: < 256
    registers, < 80 KB of LDS) run two workgroups per CU, and two eight-chunk workgroups on a CU beat one of sixteen
    (baseline_raw_gru, eight batches of 256 in flight: 520 -> 545 M samples/s).  The four-chunk kernel keeps a CU to itself."""
    if ((B + 3) // 4) * share <= ncu:
        return 0
    if per_cu == 2 and GRU64_SHARE and ((B + 3) // 4) * share <= 2 * ncu:
        return 5                   # four chunks per workgroup, two workgroups per CU (bit 2 = bit 10 of `reverse`: may share a CU)
    if ((B + 7) // 8) * share <= ncu * per_cu:
        return 2
    return 3 if ((B + 15) // 16) * share <= ncu * per_cu else 0


#: (insize, size) of the Gru layers that run as ONE kernel, projection included (the instantiations of csrc/gru_bar16.hip,
#: gru_bar16d.hip, gru_bar16q.hip: every Gru of models/ up to 96 wide)
GRU_LAYER_SHAPES = frozenset([(96, 96), (64, 64), (32, 96), (128, 96), (64, 96), (48, 32), (16, 64)])
#: sizes of the fp16-split scan behind a projection GEMM (csrc/gru_scan1t.hip: 112, 128; gru_scan16.hip: 144)
GRU_SCAN16_SIZES = frozenset([112, 128, 144])


def gru_plan(insize, size, fun, gatefun):
    """THE plan table of a Gru layer: which kernels run it.  (How many chunks a workgroup of the "layer" plan takes is a matter
    of batch size and batches in flight: _gru_plan_for.)

      "layer"   one persistent kernel for the whole layer           slk_gru_bar16_f32
      "scan16"  projection GEMM + scan on the fp16 split             slk_gru_scan16_f32
      "scan"    projection GEMM + float32-MFMA / portable scan       slk_gru_recurrent_f32, slk_gru_recurrent_ragged_f32
    """
    tanh_sigmoid = fun is activation.tanh and gatefun is activation.sigmoid
    if SPLIT_F16 and RECURRENT_F16 and tanh_sigmoid:
        if (insize, size) in GRU_LAYER_SHAPES:
            return "layer"
        if size in GRU_SCAN16_SIZES:
            return "scan16"
    return "scan"


def gru_pad_shape(insize, size, fun, gatefun):
    """(insize, size) of the zero-padded twin a Gru layer runs as, or its own shape when it runs as it is.  The reference's Gru takes
    any (insize, size) (layers.py:952-977, every model factory has a `size=` argument); the kernels exist for a table of shapes.  A
    padding neuron has zero weights and zero bias: its state stays exactly 0 (h(-1) = 0, candidate fun(0) = 0 for tanh / linear) and
    padded input columns meet zero weights, so the leading `size` outputs are those of the unpadded layer.
      * up to 96 wide with up to 128 inputs: the next shape of the one-kernel plan (GRU_LAYER_SHAPES: smallest width first, then the
        fewest inputs) -- Gru(80, 80) runs as 96 -> 96, Gru(40, 72) as 64 -> 96, Gru(20, 20) as 48 -> 32;
      * up to 144 wide: the next multiple of 16 (csrc/gru_scan1t.hip / gru_scan16.hip: 112, 128, 144);
      * sizes that are not multiples of 16 otherwise: the next multiple (the MFMA scan of csrc/recurrent.hip)."""
    if activation.act_name(fun) not in ("tanh", "linear"):
        return insize, size
    tanh_sigmoid = fun is activation.tanh and gatefun is activation.sigmoid
    if SPLIT_F16 and RECURRENT_F16 and tanh_sigmoid:
        fits = sorted((n, i) for (i, n) in GRU_LAYER_SHAPES if n >= size and i >= insize)
        if fits:
            return fits[0][1], fits[0][0]
    if size <= 144 and ((size % 16) or (insize % 16 and size > 16)):
        return (insize + 15) // 16 * 16, (size + 15) // 16 * 16
    return insize, size


#: widest Lstm on the fp16-split scan (csrc/lstm_scan16.hip: every multiple of 16 up to this, zero-padded to the next instantiation)
LSTM_SCAN16_MAX = 128


def lstm_plan(insize, size, fun, gatefun):
    """The plan table of an Lstm layer, as gru_plan: "layer" = slk_lstm_fused16_f32 (size and insize up to 64: every Lstm of
    models/), "scan16" = projection GEMM + slk_lstm_scan16_f32, "scan" = projection GEMM + slk_lstm_recurrent*_f32."""
    tanh_sigmoid = fun is activation.tanh and gatefun is activation.sigmoid
    if SPLIT_F16 and RECURRENT_F16 and tanh_sigmoid and size % 16 == 0:
        if LSTM_FUSED and size <= 64 and insize <= 64 and insize % 4 == 0:
            return "layer"
        if size <= LSTM_SCAN16_MAX:
            return "scan16"
    return "scan"


def gru_f16_entry():
    """The C-ABI entry of the whole-layer Gru kernel (projection and recurrence as fp16 splits: csrc/gru_bar16.hip)."""
    return _lib.lib().slk_gru_bar16_f32


def _derived_cache(owner, attr, params, build):
    """Device tensors derived from the Shared leaves `params` (fp16 splits, re-laid-out weights, zero-padded twins), kept on
    `owner` under `attr` and re-made by `build()` whenever a leaf's device buffer OR its version (Shared.set_value) changes.
    The kernels that build them run on the stream of the first caller: an event recorded behind them is kept with the cache
    and a caller on any other stream waits for it before its kernels read the tensors (several Basecallers, one per stream,
    share one network)."""
    import torch
    key = tuple((p.dev(), getattr(p, "_version", 0)) for p in params)
    cache = owner.__dict__.get(attr)
    stale = cache is None or len(cache[0]) != len(key) or any(a[0] is not b[0] or a[1] != b[1] for a, b in zip(cache[0], key))
    cur = torch.cuda.current_stream()
    if stale:
        tensors = build()
        ev = torch.cuda.Event()
        ev.record(cur)
        # The generation this one replaces stays referenced until the NEXT replacement: kernels queued on other streams may still be
        # reading it, and torch's per-stream allocator would hand its memory to the next allocation on the building stream the
        # moment the last reference goes (weights replaced while inference is in flight; one extra copy of the derived tensors).
        owner.__dict__[attr] = cache = (key, tensors, ev, cur, None if cache is None else cache[1])
    elif cache[3] != cur and not cache[2].query():
        cur.wait_event(cache[2])
    return cache[1]


def _split_f16_cached(owner, attr, param, rows, k):
    """fp16 hi/lo parts of the [rows][k] weight `param` (rows scaled by powers of two) and the inverse row scales, on the
    device, re-made whenever the parameter changes."""
    import torch

    def build():
        wd = param.dev()
        kp = (k + 15) // 16 * 16
        hi = torch.empty((rows, kp), dtype=torch.float16, device=wd.device)
        lo = torch.empty((rows, kp), dtype=torch.float16, device=wd.device)
        inv = torch.empty((rows,), dtype=torch.float32, device=wd.device)
        _lib.check(_lib.lib().slk_split_f16x2_f32(wd.data_ptr(), rows, k, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(),
                                                  _stream()), "split_f16")
        return hi, lo, inv

    return _derived_cache(owner, attr, (param,), build)


def _pack_bf16_cached(owner, attr, param, rows, k):
    """The [rows][k] weight `param` cut into three bf16 pieces (csrc/gemm_bf16x6.hip), on the device, re-made whenever the
    parameter changes."""
    import torch

    def build():
        wd = param.dev()
        packed = torch.empty(_lib.lib().slk_pack_bf16x3_bytes(rows, k), dtype=torch.uint8, device=wd.device)
        _lib.check(_lib.lib().slk_pack_bf16x3_f32(wd.data_ptr(), rows, k, packed.data_ptr(), _stream()), "pack_bf16")
        return (packed,)

    return _derived_cache(owner, attr, (param,), build)[0]


def _projection(owner, x, W, b, ws_ptr, rows, k, n_out, stage, name):
    """ws[rows][n_out] = x.W^T + b for a recurrent layer's input projection: fp16x3 kernel where it applies
    (k <= 192, n_out <= 2048), else the fp32 MFMA GEMM."""
    L = _lib.lib()
    use_f16 = SPLIT_F16 and k <= 192 and n_out <= 2048
    with profiler.region(stage, 2.0 * rows * k * n_out, 4.0 * rows * (k + n_out),
                         f16x3_flops=2.0 * rows * k * n_out if use_f16 else 0.0):
        rc = _lib.SLK_ERR_UNSUPPORTED
        if use_f16:
            hi, lo, inv = _split_f16_cached(owner, "_iw16", W, n_out, k)
            rc = L.slk_linear_rowstats_f16x3(x.data_ptr(), _row_stride(x), hi.data_ptr(), lo.data_ptr(), inv.data_ptr(),
                                             b.dev().data_ptr(), ws_ptr, n_out, rows, k, n_out, None, _stream())
        if rc == _lib.SLK_ERR_UNSUPPORTED:
            rc = L.slk_gemm_bias_act_f32(x.data_ptr(), _row_stride(x), W.dev().data_ptr(), b.dev().data_ptr(), ws_ptr,
                                         n_out, rows, k, n_out, 0, _stream())
    _lib.check(rc, name)


class _RaggedMeta(type):
    """`ragged.current` per host thread (one forward pass per thread at a time, like _PlanHints): two threads that run ragged batches
    side by side must not see each other's lengths."""
    _tls = __import__("threading").local()

    @property
    def current(cls):
        return getattr(_RaggedMeta._tls, "current", None)

    @current.setter
    def current(cls, value):
        _RaggedMeta._tls.current = value


class ragged(object, metaclass=_RaggedMeta):
    """Context for a ragged batch: `with ragged(lengths): net.run(x)` runs a zero-padded batch [T, B, F] whose chunk b is
    only lengths[b] steps long (whole reads of different lengths).  Time-local layers are unaffected (they compute the
    padding too), Convolution maps the lengths through its stride, and recurrent layers in reverse time start every chunk
    at ITS last step -- so each chunk gets exactly what a run on the unpadded chunk alone would produce; rows beyond a
    chunk's length hold unspecified values.  `ragged.current`: the int32 device tensor [B] of the tensor currently flowing through
    the network on this host thread, or None."""

    def __init__(self, lengths):
        import torch
        from . import device as D
        if isinstance(lengths, torch.Tensor) and lengths.is_cuda:
            # lengths that were computed on the device (pipeline.Basecaller's streamed whole-read mode) stay there
            if lengths.dtype != torch.int32 or lengths.dim() != 1:
                raise ValueError("ragged lengths on the device must be a 1-D int32 tensor")
            self.lengths = lengths.contiguous()
        else:
            self.lengths = torch.as_tensor(np.ascontiguousarray(lengths, dtype=np.int32)).to(D.device())

    def __enter__(self):
        self.saved, ragged.current = ragged.current, self.lengths
        return self

    def __exit__(self, *exc):
        ragged.current = self.saved
        return False


def _check_input(x, insize):
    import torch
    if not isinstance(x, torch.Tensor) or x.dim() != 3 or x.dtype != torch.float32 or not x.is_cuda:
        raise ValueError("layer input must be a float32 device tensor [time, batch, features]")
    if x.shape[2] != insize:
        raise ValueError("layer expects %d input features, got %d" % (insize, x.shape[2]))
    if x.stride(2) != 1 or x.stride(0) != x.shape[1] * x.stride(1):
        x = x.contiguous()
    return x


def _row_stride(x):
    """Row stride (floats) of a [T,B,F] tensor whose (t,b) rows are equally spaced."""
    return x.stride(1)


def _alloc_out(x, T, B, size, out):
    """Either use the caller's (possibly strided) output slice or allocate a dense one."""
    import torch
    if out is not None:
        assert out.shape == (T, B, size) and out.stride(2) == 1 and out.stride(0) == B * out.stride(1)
        return out
    return _scratch((T, B, size), torch.float32, x.device)


class Param(object):
    """Declarative description of one weight tensor of a layer.

    view      : shape the tensor is shown in by `json(params=True)` (None = as stored)
    given     : shape `set_params` expects in its `values` dict (None = only the leading size is checked)
    store     : function(layer, array) -> array in storage layout
    flag      : name of the boolean attribute that says whether `set_params` touches it (None = always)
    """

    def __init__(self, name, view=None, given=None, store=None, flag=None):
        self.name, self.view, self.given, self.store, self.flag = name, view, given, store, flag


class Layer(metaclass=abc.ABCMeta):
    _json_type = None        # 'type' string of the reference's json() for this class
    _json_fields = ()        # ((key, attribute-or-function), ...) emitted after 'type'
    _weights = ()            # Param descriptors, in the order json() lists them

    def json(self, params=False):
        """Description of the layer in the reference's json vocabulary (same keys, same order)."""
        res = OrderedDict([('type', self._json_type)])
        for key, src in self._json_fields:
            val = src(self) if callable(src) else getattr(self, src)
            res[key] = val
        if params and self._weights:
            res['params'] = OrderedDict(
                (w.name, _extract(getattr(self, w.name), w.view(self) if w.view else None)) for w in self._weights)
        return res

    def set_params(self, values):
        """Set parameters from a dictionary of arrays given in the shapes json() shows them in."""
        for w in self._weights:
            if w.flag is not None and not getattr(self, w.flag):
                continue
            val = values[w.name]
            if w.given is not None:
                assert val.shape == w.given(self), "parameter %s has shape %r" % (w.name, val.shape)
            else:
                assert val.shape[0] == self.size
            getattr(self, w.name).set_value(w.store(self, val) if w.store else val)

    def compile(self):
        """layers.py:34-36.  Returns f(ndarray|tensor [T,B,insize]) -> same kind [T',B,size]."""
        _lib.lib()      # fail now, loudly, if the extension is missing

        def calc(x):
            from . import device as D
            xd = D.to_dev(x)
            return D.like_input(self.run(xd), x)
        calc.network = self
        return calc

    @property
    def insize(self):
        return self._insize

    @property
    def size(self):
        return self._size

    @property
    def name(self):
        return self._name

    @abc.abstractmethod
    def params(self):
        """ a list of network parameters """
        return

    def run(self, inMat):
        """ Run network layer on a device tensor """
        return self._forward(_check_input(inMat, self.insize), None, False)

    @abc.abstractmethod
    def _forward(self, x, out, reverse):
        """x: device [T,B,insize]; out: optional pre-allocated (possibly strided) [T',B,size] slice;
        reverse: evaluate as Reverse(self) would (layers.py:1449-1450)."""
        return

    @abc.abstractmethod
    def spec(self):
        """Neutral nested-dict description (type strings of `json()`, weights as numpy arrays in the layout
        the reference's run/step code reads them) -- what tests hand to the CPU oracle."""
        return


class RNN(Layer):
    pass


def _flip_run(layer, x, out):
    """Literal Reverse semantics for layers without a time-order flag."""
    import torch
    y = layer._forward(torch.flip(x, dims=[0]).contiguous(), None, False)
    y = torch.flip(y, dims=[0])
    if out is not None:
        out.copy_(y)
        return out
    return y


class Identity(Layer):
    """layers.py:91-111"""

    def __init__(self, insize, name="Identity"):
        self._insize = insize
        self._name = name

    @property
    def size(self):
        return self.insize

    def params(self):
        return []

    def json(self, params=False):
        return {'type': "identity"}

    def set_params(self, values):
        return

    def _forward(self, x, out, reverse):
        if out is not None:
            out.copy_(x)
            return out
        return x

    def spec(self):
        return {"type": "identity"}


class FeedForward(Layer):
    """  Basic feedforward layer:  out = f( inMat W + b )       (layers.py:114-158)

    :param insize: number of input features per step
    :param size: number of units (output features per step)
    :param init: callable(shape) drawing the initial weights
    :param has_bias: False leaves out the bias vector `b`
    :param fun: activation of the output (a function of sloika_amd.activation)
    :param name: label that json() and pickles carry
    """

    _json_type = "feed-forward"
    _json_fields = (('activation', lambda self: self.fun.__name__), ('size', 'size'), ('insize', 'insize'),
                    ('bias', 'has_bias'))
    _weights = (Param('W', given=lambda self: (self.size, self.insize)), Param('b', flag='has_bias'))

    def __init__(self, insize, size, init=zeros, has_bias=False,
                 fun=activation.tanh, name="Feed-forward"):
        self.has_bias = has_bias
        self.b = shared(has_bias * init(size))
        self.W = shared(init((size, insize)) / np.sqrt(size + insize))
        self._insize = insize
        self._size = size
        self._name = name
        self.fun = fun

    def params(self):
        return [self.W, self.b] if self.has_bias else [self.W]

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_w16", None)          # device caches: never pickled
        d.pop("_wbf16", None)
        return d

    def _forward(self, x, out, reverse):
        T, B, _ = x.shape                       # time-local: Reverse is the identity on it
        y = _alloc_out(x, T, B, self.size, out)
        rows = T * B
        L = _lib.lib()
        act = activation.act_id(self.fun)
        use_f16 = SPLIT_F16 and self.insize <= 192 and self.size <= 2048
        # up to 96 output columns (one column block) the LDS-staged kernel with six bf16 terms per product is the faster one:
        # 128 -> 64 at 4.1 M rows 0.78 ms against 1.10, 192 -> 96 0.29 against 0.39 (csrc/gemm_bf16x6.hip; float32-grade like the split)
        use_bf16 = SPLIT_F16 and self.size <= BF16_FF_MAX and self.insize % 4 == 0 and self.insize >= 32 and act in (0, 1, 2)
        with profiler.region("gemm_bias_act", 2.0 * rows * self.insize * self.size,
                             4.0 * rows * (self.insize + self.size),
                             f16x3_flops=2.0 * rows * self.insize * self.size if use_f16 else 0.0) as reg:
            rc = _lib.SLK_ERR_UNSUPPORTED
            if use_bf16:
                packed = _pack_bf16_cached(self, "_wbf16", self.W, self.size, self.insize)
                rc = L.slk_gemm_bias_act_bf16x6(x.data_ptr(), _row_stride(x), packed.data_ptr(), self.b.dev().data_ptr(),
                                                y.data_ptr(), _row_stride(y), rows, self.insize, self.size, act, _stream())
            if rc == _lib.SLK_ERR_UNSUPPORTED and use_f16:
                hi, lo, inv = _split_f16_cached(self, "_w16", self.W, self.size, self.insize)
                rc = L.slk_gemm_bias_act_f16x3(x.data_ptr(), _row_stride(x), hi.data_ptr(), lo.data_ptr(), inv.data_ptr(),
                                               self.b.dev().data_ptr(), y.data_ptr(), _row_stride(y), rows, self.insize,
                                               self.size, act, _stream())
            if rc == _lib.SLK_ERR_UNSUPPORTED:           # activation or size the fp16x3 kernel does not cover
                rc = L.slk_gemm_bias_act_f32(x.data_ptr(), _row_stride(x), self.W.dev().data_ptr(),
                                             self.b.dev().data_ptr(), y.data_ptr(), _row_stride(y), rows, self.insize,
                                             self.size, act, _stream())
        _lib.check(rc, "FeedForward")
        return y

    def spec(self):
        return {"type": "feed-forward", "W": self.W.get_value(), "b": self.b.get_value(),
                "activation": activation.act_name(self.fun)}


class Softmax(Layer):
    """  Softmax layer: tmp = exp( inmat W + b ); out = row_normalise( tmp )     (layers.py:268-314)

    :param insize: number of input features per step
    :param size: number of units (output features per step)
    :param init: callable(shape) drawing the initial weights
    :param has_bias: False leaves out the bias vector `b`
    :param name: label that json() and pickles carry
    """

    _json_type = "softmax_old"
    _json_fields = (('size', 'size'), ('insize', 'insize'), ('bias', 'has_bias'))
    _weights = (Param('W', given=lambda self: (self.size, self.insize)), Param('b', flag='has_bias'))

    def __init__(self, insize, size, init=zeros, has_bias=False, name="Softmax"):
        self.has_bias = has_bias
        self.b = shared(has_bias * init(size))
        self.W = shared(init((size, insize)) / np.sqrt(size + insize))
        self._insize = insize
        self._size = size
        self._name = name

    def params(self):
        return [self.W, self.b] if self.has_bias else [self.W]

    #: evaluate the projection on the FP16 matrix pipe with operands split in two halves (x.w = hi.hi + hi.lo + lo.hi,
    #: float32 accumulation; error a few float32 ulps, ~5x the fp32-MFMA rate).  Set False for plain fp32 MFMA.
    split_f16 = SPLIT_F16

    def _split_weights(self):
        """fp16 hi/lo parts of W and the inverse row scales on the device, re-made whenever W changes."""
        return _split_f16_cached(self, "_w16", self.W, self.size, self.insize)

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_w16", None)          # device caches: never pickled
        d.pop("_svpack", None)
        d.pop("_svpack_p", None)
        return d

    def viterbi_pack(self, nbase, klen, kpad=None):
        """The weights as csrc/softmax_viterbi.hip wants them (MFMA fragment order, fp16 hi/lo, column scales) for decoding
        straight from this layer's INPUT (decode.viterbi_fused_batch), or None where that kernel does not apply (state count
        other than 4^5 + 1, insize not a multiple of 16 up to 128, all-fp32 arithmetic requested).  `kpad` > insize: the input
        rows carry zero columns up to `kpad` (the output of a zero-padded Gru twin, models/raw_1.00_rGr.py: 110 -> 112); the
        weights get zero columns to match."""
        import torch
        L = _lib.lib()
        K = self.insize if kpad is None else int(kpad)
        nbytes = L.slk_softmax_viterbi_pack_bytes(K, nbase, klen) if (self.split_f16 and K >= self.insize) else 0
        if nbytes == 0 or self.size != nbase ** klen + 1:
            return None

        def build():
            wd, bd = self.W.dev(), self.b.dev()
            if K != self.insize:
                wp = torch.zeros((self.size, K), dtype=torch.float32, device=wd.device)
                wp[:, :self.insize] = wd.reshape(self.size, self.insize)
                wd = wp
            pack = torch.empty(nbytes, dtype=torch.uint8, device=wd.device)
            _lib.check(L.slk_softmax_viterbi_pack_f32(wd.data_ptr(), bd.data_ptr(), K, nbase, klen, pack.data_ptr(),
                                                      _stream()), "softmax_viterbi_pack")
            return pack

        return _derived_cache(self, "_svpack" if K == self.insize else "_svpack_p%d" % K, (self.W, self.b), build)

    def _logits(self, x, ld):
        """tmp = x.W^T + b (layers.py:310) with rows `ld` floats apart, and per-row (max, 1/sum exp) [T*B,2]."""
        import torch
        T, B, _ = x.shape
        rows, L = T * B, _lib.lib()
        y = _scratch((rows, ld), torch.float32, x.device)
        stats = _scratch((rows, 2), torch.float32, x.device)
        use_f16 = self.split_f16 and self.insize <= 128 and self.size <= 2048
        with profiler.region("softmax_gemm", 2.0 * rows * self.insize * self.size,
                             4.0 * rows * (self.insize + self.size),
                             f16x3_flops=2.0 * rows * self.insize * self.size if use_f16 else 0.0):
            if use_f16:
                hi, lo, inv = self._split_weights()
                rc = L.slk_linear_rowstats_f16x3(x.data_ptr(), _row_stride(x), hi.data_ptr(), lo.data_ptr(), inv.data_ptr(),
                                                 self.b.dev().data_ptr(), y.data_ptr(), ld, rows, self.insize,
                                                 self.size, stats.data_ptr(), _stream())
            else:
                rc = L.slk_linear_rowstats_f32(x.data_ptr(), _row_stride(x), self.W.dev().data_ptr(),
                                               self.b.dev().data_ptr(), y.data_ptr(), ld, rows, self.insize,
                                               self.size, stats.data_ptr(), _stream())
            if rc == _lib.SLK_ERR_UNSUPPORTED:
                # insize > 128: tiled GEMM, then a statistics pass over the (dense) logits
                if ld != self.size:
                    raise _lib.SloikaAmdError("internal: the statistics pass needs dense logits")
                rc = L.slk_gemm_bias_act_f32(x.data_ptr(), _row_stride(x), self.W.dev().data_ptr(),
                                             self.b.dev().data_ptr(), y.data_ptr(), ld, rows, self.insize, self.size,
                                             0, _stream())
                _lib.check(rc, "Softmax")
                rc = L.slk_softmax_rowstats_f32(y.data_ptr(), rows, self.size, stats.data_ptr(), _stream())
        _lib.check(rc, "Softmax")
        return y, stats

    def logits_and_stats(self, x):
        """What the decoder needs to rebuild the posterior on the fly (decode.viterbi_logits_batch): logits with a
        128-byte aligned row stride plus the row statistics.  The normalised posterior is never written."""
        x = _check_input(x, self.insize)
        ld = self.size if self.insize > 128 else ((self.size + 31) // 32) * 32
        y, stats = self._logits(x, ld)
        return y, stats, ld

    def _forward(self, x, out, reverse):
        T, B, _ = x.shape
        logits, stats = self._logits(x, self.size)
        y = out if (out is not None and out.stride(1) == self.size) else logits.view(T, B, self.size)
        rows = T * B
        with profiler.region("softmax_normalise", 0.0, 8.0 * rows * self.size):
            rc = _lib.lib().slk_softmax_from_stats_f32(logits.data_ptr(), self.size, stats.data_ptr(), y.data_ptr(),
                                                       self.size, rows, self.size, _stream())
        _lib.check(rc, "Softmax")
        if out is not None and y is not out:
            out.copy_(y)
            return out
        return y

    def spec(self):
        return {"type": "softmax", "W": self.W.get_value(), "b": self.b.get_value()}


class SoftmaxTheano(Softmax):
    """layers.py:222-265: same mathematics through T.nnet.softmax; `json` type differs."""

    def json(self, params=False):
        res = Softmax.json(self, params)
        res['type'] = "softmax"
        return res


class Window(Layer):
    """  Create a sliding window over input      (layers.py:317-351)

    :param w: width of the window in time steps (odd)
    :param name: label that json() and pickles carry
    """

    def __init__(self, insize, w, name="Window"):
        assert w > 0, "Window size must be positive"
        assert w % 2 == 1, 'Window size should be odd'
        self.w = w
        self._insize = insize
        self._name = name

    @property
    def size(self):
        return self.w * self.insize

    def params(self):
        return []

    def json(self, params=False):
        res = OrderedDict([('type', "window")])
        if params:
            res['params'] = OrderedDict([('w', self.w)])
        return res          # (the reference forgets this return, layers.py:338-341)

    def set_params(self, values):
        return

    def _forward(self, x, out, reverse):
        import torch
        T, B, F = x.shape                      # symmetric padding: commutes with time reversal
        x = x.contiguous()
        y = _scratch((T, B, self.size), torch.float32, x.device)
        _lib.check(_lib.lib().slk_window_f32(x.data_ptr(), y.data_ptr(), T, B, F, self.w, _stream()), "Window")
        if out is not None:
            out.copy_(y)
            return out
        return y

    def spec(self):
        return {"type": "window", "w": self.w}


class Convolution(Layer):
    """1D convolution over the first dimension       (layers.py:354-419)

    [time, batch, insize] in, [ceil((time + total padding) / stride), batch, size] out.

    :param insize: features per input step
    :param size: features per output step
    :param winlen: filter taps (input steps one output sees)
    :param stride: input steps from one output to the next
    :param init: callable(shape) drawing the initial weights
    :param has_bias: False leaves out the bias vector `b`
    :param fun: activation of the output (a function of sloika_amd.activation)
    :param padding_mode: str, int or (int, int); see conv.calculate_padding. Default: 'same'
    :param name: label that json() and pickles carry
    """

    _json_type = "convolution"
    _json_fields = (("insize", 'insize'), ("size", 'size'), ("winlen", 'winlen'), ("stride", 'stride'),
                    ("padding_mode", 'padding_mode'), ("padding", 'padding'),
                    ("activation", lambda self: self.fun.__name__))
    _weights = (Param('W', given=lambda self: (self.size, self.insize, self.winlen)), Param('b', flag='has_bias'))

    def __init__(self, insize, size, winlen, stride=1, init=zeros,
                 has_bias=False, fun=activation.tanh, padding_mode='same',
                 name="Convolution"):
        self._insize = insize
        self._size = size
        self._name = name
        self.winlen = winlen
        self.stride = stride
        self.fun = fun
        self.has_bias = has_bias
        self.padding_mode = padding_mode
        self.padding = conv.calculate_padding(padding_mode, winlen)

        fanin = insize * winlen
        fanout = (size * winlen) / float(stride)
        self.W = shared(init((size, insize, winlen)) / np.sqrt(fanin + fanout))
        self.b = shared(has_bias * init(size))

    def params(self):
        return [self.W, self.b] if self.has_bias else [self.W]

    def out_len(self, T):
        return _lib.lib().slk_conv1d_out_len(T, self.winlen, self.stride, self.padding[0], self.padding[1])

    def run_strided(self, x_ptr, T, B, x_t_stride, x_b_stride, device):
        """Convolve input addressed as x[t*x_t_stride + b*x_b_stride + c] (lets the chunk front end hand over
        chunk-major signal without a transpose)."""
        import torch
        y = _scratch((self.out_len(T), B, self.size), torch.float32, device)
        with profiler.region("conv1d", 2.0 * y.numel() * self.insize * self.winlen,
                             4.0 * (y.numel() + T * B * self.insize)):
            rc = _lib.lib().slk_conv1d_f32(x_ptr, x_t_stride, x_b_stride, self.W.dev().data_ptr(),
                                           self.b.dev().data_ptr(), y.data_ptr(), T, B, self.insize, self.size,
                                           self.winlen, self.stride, self.padding[0], self.padding[1],
                                           activation.act_id(self.fun), _stream())
        _lib.check(rc, "Convolution")
        if ragged.current is not None:
            # zero padding past a chunk's end is what the unpadded convolution would have seen, so its first
            # out_len(length) output steps are exact; map the lengths through the stride (conv.py:66-77)
            ragged.current = ((ragged.current + (self.padding[0] + self.padding[1] - self.winlen)) // self.stride + 1
                              ).clamp_(min=1).to(ragged.current.dtype)
        return y

    def _forward(self, x, out, reverse):
        if reverse:
            return _flip_run(self, x, out)
        T, B, C = x.shape
        x = x.contiguous()
        y = self.run_strided(x.data_ptr(), T, B, B * C, C, x.device)
        if out is not None:
            out.copy_(y)
            return out
        return y

    def spec(self):
        return {"type": "convolution", "W": self.W.get_value(), "b": self.b.get_value(), "stride": self.stride,
                "padding": tuple(self.padding), "activation": activation.act_name(self.fun)}


class Lstm(RNN):
    """ LSTM layer with peepholes (layers.py:599-697).  Step (:677-691):
        a = x iW^T + h sW^T + b, read per unit j as (a0, a1, a2, a3) = a[4j .. 4j+3];  c = cell state, p = peepholes
        i = gatefun(a1 + c p0)        input gate          f = gatefun(a2 + c p1)        forget gate
        c' = f c + i fun(a0)          new cell state      o = gatefun(a3 + c' p2)       output gate (peeps at the NEW state)
        h' = o fun(c')                output
    Rows of iW / sW / b are interleaved as j*4 + gate (the layout `step` reads; `json`/`set_params` keep the
    reference's block reshapes verbatim).
    """

    _json_type = "LSTM"
    _json_fields = (('activation', lambda self: self.fun.__name__), ('gate', lambda self: self.gatefun.__name__),
                    ('size', 'size'), ('insize', 'insize'), ('bias', 'has_bias'), ('peep', 'has_peep'))
    # layers.py:659-675: json()/set_params() present the weights as 4 gate blocks even though step() reads the
    # rows interleaved; kept as is so that parameter dictionaries round-trip exactly as in the reference.
    _weights = (Param('iW', view=lambda self: (4, self.size, self.insize), given=lambda self: (4, self.size, self.insize),
                      store=lambda self, v: v.reshape((self.size * 4, self.insize))),
                Param('sW', view=lambda self: (4, self.size, self.size), given=lambda self: (4, self.size, self.size),
                      store=lambda self, v: v.reshape((self.size * 4, self.size))),
                Param('b', view=lambda self: (4, self.size), given=lambda self: (4, self.size),
                      store=lambda self, v: v.transpose().reshape(-1), flag='has_bias'),
                Param('p', view=lambda self: (3, self.size), given=lambda self: (3, self.size), flag='has_peep'))

    def __init__(self, insize, size, init=zeros, has_bias=False, has_peep=False,
                 fun=activation.tanh, gatefun=activation.sigmoid, name="LSTM"):
        self._size = size
        self._insize = insize
        self._name = name
        self.has_bias = has_bias
        self.has_peep = has_peep
        self.fun = fun
        self.gatefun = gatefun

        self.b = shared(has_bias * (init(4 * size) + np.repeat([0, 0, _FORGET_BIAS, 0], size).astype(sloika_dtype)))
        self.p = shared(has_peep * init((3, size)) / np.sqrt(size))
        self.iW = shared(init((4 * size, insize)) / np.sqrt(insize + size))
        self.sW = shared(init((4 * size, size)) / np.sqrt(size + size))

    def params(self):
        params = [self.iW, self.sW]
        if self.has_bias:
            params += [self.b]
        if self.has_peep:
            params += [self.p]
        return params

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_iw16", None)         # device cache: never pickled
        return d

    def takes_fused_kernel(self):
        """Whether csrc/lstm_fused16.hip (projection inside the scan) applies to this layer's shape."""
        return lstm_plan(self.insize, self.size, self.fun, self.gatefun) == "layer"

    def _forward(self, x, out, reverse):
        import torch
        T, B, _ = x.shape
        y = _alloc_out(x, T, B, self.size, out)
        L = _lib.lib()
        n, rows = self.size, T * B
        lens = ragged.current if reverse else None       # a reversed scan starts every chunk at its own last step
        lens_p = None if lens is None else lens.data_ptr()
        act, gact = activation.act_id(self.fun), activation.act_id(self.gatefun)
        plan = lstm_plan(self.insize, n, self.fun, self.gatefun)
        if plan == "layer":
            # the whole layer in one kernel (csrc/lstm_fused16.hip): the projection is computed inside the scan and never written
            with profiler.region("lstm_fused", 8.0 * rows * n * (self.insize + n), 4.0 * rows * (self.insize + n),
                                 f16x3_flops=8.0 * rows * n * self.insize, f16x2_flops=8.0 * rows * n * n) as reg:
                rc = L.slk_lstm_fused16_f32(x.data_ptr(), _row_stride(x), self.iW.dev().data_ptr(), self.sW.dev().data_ptr(),
                                            self.b.dev().data_ptr(), self.p.dev().data_ptr(), y.data_ptr(), _row_stride(y), T, B,
                                            self.insize, n, int(reverse), act, gact, lens_p, _stream())
                if rc == _lib.SLK_ERR_UNSUPPORTED and reg is not None:
                    reg.cancel()
            if rc != _lib.SLK_ERR_UNSUPPORTED:                 # (unsupported: rows of x that are not 16-byte aligned)
                _lib.check(rc, "Lstm")
                return y
            plan = "scan16"
        nbytes = L.slk_lstm_workspace_bytes(T, B, n)
        ws = _scratch(nbytes, torch.uint8, x.device)
        # the two halves of slk_lstm_f32, timed separately: projection GEMM into the workspace, then the recurrence
        _projection(self, x, self.iW, self.b, ws.data_ptr(), rows, self.insize, 4 * n, "lstm_input_gemm", "Lstm")
        with profiler.region("lstm_recurrent", 8.0 * rows * n * n, 4.0 * rows * 5 * n):
            rc = _lib.SLK_ERR_UNSUPPORTED
            if plan == "scan16":
                # the recurrent product as an fp16 split on the barrier-stepped plan (csrc/lstm_scan16.hip; unsupported: a
                # projection of 4 GiB or more, its lanes keep 32-bit offsets)
                rc = L.slk_lstm_scan16_f32(ws.data_ptr(), self.sW.dev().data_ptr(), self.p.dev().data_ptr(), y.data_ptr(),
                                           _row_stride(y), T, B, n, int(reverse), act, gact, lens_p, _stream())
            if rc != _lib.SLK_ERR_UNSUPPORTED:
                pass
            elif lens is None:                             # float32 MFMA (csrc/lstm_mfma.hip), portable kernel beyond its sizes
                rc = L.slk_lstm_recurrent_f32(ws.data_ptr(), self.sW.dev().data_ptr(), self.p.dev().data_ptr(),
                                              y.data_ptr(), _row_stride(y), T, B, n, int(reverse), act, gact, _stream())
            else:
                rc = L.slk_lstm_recurrent_ragged_f32(ws.data_ptr(), self.sW.dev().data_ptr(), self.p.dev().data_ptr(),
                                                     y.data_ptr(), _row_stride(y), T, B, n, int(reverse), act, gact,
                                                     lens_p, _stream())
        _lib.check(rc, "Lstm")
        return y

    def spec(self):
        return {"type": "LSTM", "iW": self.iW.get_value(), "sW": self.sW.get_value(), "b": self.b.get_value(),
                "p": self.p.get_value(), "activation": activation.act_name(self.fun),
                "gate": activation.act_name(self.gatefun)}


class Gru(RNN):
    """ Gated Recurrent Unit (layers.py:952-1021).  Step (:1010-1021):
        vI = x iW^T + b ; vS = h sW^T ; z, r = gatefun(vI[:2n] + vS) ; y = (r*h) sW2^T
        hbar = fun(vI[2n:] + y) ; h = z*h + (1-z)*hbar

    :param insize: number of input features per step
    :param size: number of units (output features per step)
    :param init: callable(shape) drawing the initial weights
    :param has_bias: False leaves out the bias vector `b`
    :param fun: activation of the output (a function of sloika_amd.activation)
    :param gatefun: activation of the gates
    :param name: label that json() and pickles carry
    """

    _json_type = "GRU"
    _json_fields = (('activation', lambda self: self.fun.__name__), ('gate', lambda self: self.gatefun.__name__),
                    ('size', 'size'), ('insize', 'insize'), ('bias', 'has_bias'))
    _weights = (Param('iW', view=lambda self: (3, self.size, self.insize), given=lambda self: (3, self.size, self.insize),
                      store=lambda self, v: v.reshape((3 * self.size, self.insize))),
                Param('sW', view=lambda self: (2, self.size, self.size), given=lambda self: (2, self.size, self.size),
                      store=lambda self, v: v.reshape((2 * self.size, self.size))),
                Param('sW2', given=lambda self: (self.size, self.size)),
                Param('b', view=lambda self: (3, self.size), given=lambda self: (3, self.size),
                      store=lambda self, v: v.reshape(-1), flag='has_bias'))

    def __init__(self, insize, size, init=zeros, has_bias=False,
                 fun=activation.tanh, gatefun=activation.sigmoid, name='GRU'):
        self._size = size
        self._insize = insize
        self._name = name
        self.has_bias = has_bias
        self.fun = fun
        self.gatefun = gatefun

        self.b = shared(has_bias * init(3 * size))
        self.iW = shared(init((3 * size, insize)) / np.sqrt(insize + size))
        self.sW = shared(init((2 * size, size)) / np.sqrt(size + size))
        self.sW2 = shared(init((size, size)) / np.sqrt(size + size))

    def params(self):
        params = [self.iW, self.sW, self.sW2]
        if self.has_bias:
            params += [self.b]
        return params

    def _plan_bits(self, x, B):
        """Bits 8-9 of `reverse` for slk_gru_bar16_f32: what Parallel decided for its side-by-side sub-layers, else eight chunks
        per workgroup when that is what lets the batches in flight share the chip."""
        import torch
        ncu = _cu_count(x.device)
        if _HINTS.gru_plan_bits or _HINTS.in_flight <= 1:
            bits = _HINTS.gru_plan_bits
        else:
            bits = _gru_plan_for(B, _HINTS.in_flight, ncu, 2 if max(self.size, self.insize) <= 64 else 1)
        if _HINTS.deterministic:
            # one set of bits per chunk whatever the plan: four and eight chunks per workgroup (and two four-chunk workgroups per CU) compute
            # the same bits, sixteen do not -- where that plan would be chosen (here, or by bar16_auto_plan in the library for a batch
            # whose eight-chunk workgroups outnumber the CUs) the eight-chunk one runs, in as many rounds as it takes
            if (bits & 3) == 3 or (bits == 0 and (B + 7) // 8 > ncu):
                bits = 2
        return bits

    def _padded(self, shape=None):
        """Zero-padded copy of this layer with `shape` = (insize, size) (default: both rounded up to multiples of 16, what the MFMA
        kernels are instantiated for; gru_pad_shape names the shape a layer runs as).  Padding neurons see zero weights and zero
        bias, so their state stays exactly 0 (h0 = 0, candidate = fun(0) = 0 for tanh-like fun) and padded input columns multiply
        zero weights: the first `size` outputs are those of the unpadded layer.  Cached until a parameter changes."""
        # the cache holds the device tensors themselves (an id() of a freed tensor can be reused by its successor) and the
        # parameters' versions (set_value on a buffer the optimiser owns writes in place: same tensor, new contents)
        n, i = self.size, self.insize
        i16, n16 = shape if shape is not None else ((i + 15) // 16 * 16, (n + 15) // 16 * 16)
        key = tuple((p.dev(), getattr(p, "_version", 0)) for p in (self.iW, self.sW, self.sW2, self.b))
        cache = getattr(self, "_pad_cache", None)
        if (cache is not None and all(a[0] is b[0] and a[1] == b[1] for a, b in zip(cache[0], key))
                and (cache[1].insize, cache[1].size) == (i16, n16)):
            return cache[1]
        twin = Gru(i16, n16, has_bias=True, fun=self.fun, gatefun=self.gatefun, name=self.name)
        iW = np.zeros((3, n16, i16), dtype=sloika_dtype)
        iW[:, :n, :i] = self.iW.get_value().reshape(3, n, i)
        sW = np.zeros((2, n16, n16), dtype=sloika_dtype)
        sW[:, :n, :n] = self.sW.get_value().reshape(2, n, n)
        sW2 = np.zeros((n16, n16), dtype=sloika_dtype)
        sW2[:n, :n] = self.sW2.get_value()
        b = np.zeros((3, n16), dtype=sloika_dtype)
        b[:, :n] = self.b.get_value().reshape(3, n)
        twin.set_params({"iW": iW, "sW": sW, "sW2": sW2, "b": b})
        self._pad_cache = (key, twin, None if cache is None else cache[1])      # (the twin it replaces: kept one generation, as in _derived_cache)
        return twin

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_pad_cache", None)
        d.pop("_iw16", None)         # device caches: never pickled
        return d

    def _forward(self, x, out, reverse):
        import torch
        T, B, _ = x.shape
        n = self.size
        target = gru_pad_shape(self.insize, n, self.fun, self.gatefun)
        if target != (self.insize, n):
            # shapes without a kernel of their own (e.g. models/raw_1.00_rGr.py: 110 / 142; Gru(80, 80); size= arguments of the model
            # factories): the zero-padded twin runs on the shape that has one
            twin = self._padded(target)
            if twin.insize != self.insize:
                xp = torch.zeros((T, B, twin.insize), dtype=torch.float32, device=x.device)
                xp[:, :, :self.insize] = x
            else:
                xp = x
            yp = twin._forward(xp, None, reverse)
            y = yp[:, :, :n]
            if out is not None:
                out.copy_(y)
                return out
            y._slk_zero_padded = twin.size                # the columns behind `n` exist and are exactly zero (pipeline._fused_pack)
            return y
        y = _alloc_out(x, T, B, self.size, out)
        L = _lib.lib()
        n, rows = self.size, T * B
        # ragged batch: only a reversed scan needs to know where each chunk ends (forward scans are causal)
        lens = ragged.current if reverse else None
        if lens is not None and (lens.numel() != B or lens.device != x.device):
            raise ValueError("ragged lengths do not match the batch")
        lens_p = None if lens is None else lens.data_ptr()
        act, gact = activation.act_id(self.fun), activation.act_id(self.gatefun)
        plan = gru_plan(self.insize, n, self.fun, self.gatefun)
        if plan == "layer":
            # projection AND recurrence in one persistent kernel (csrc/gru_bar16.hip; eight / sixteen chunks per workgroup:
            # gru_bar16d.hip / gru_bar16q.hip).  Roofline bookkeeping: up to eight chunks per workgroup the recurrent products
            # take two MFMAs each, the projection three; the sixteen-chunk plan three everywhere (bar16_auto_plan)
            bits = self._plan_bits(x, B)
            ncu = _cu_count(x.device)
            two_term = bits in (1, 2, 5) or (bits == 0 and (B + 7) // 8 <= ncu)
            with profiler.region("gru_fused", 6.0 * rows * n * (n + self.insize), 4.0 * rows * (self.insize + n),
                                 f16x3_flops=6.0 * rows * n * (self.insize if two_term else n + self.insize),
                                 f16x2_flops=6.0 * rows * n * n if two_term else 0.0) as reg:
                rc = L.slk_gru_bar16_f32(x.data_ptr(), _row_stride(x), self.iW.dev().data_ptr(), self.sW.dev().data_ptr(),
                                         self.sW2.dev().data_ptr(), self.b.dev().data_ptr(), y.data_ptr(), _row_stride(y), T, B,
                                         self.insize, n, int(reverse) | (bits << 8), act, gact, lens_p, None, _stream())
                if rc == _lib.SLK_ERR_UNSUPPORTED and reg is not None:
                    reg.cancel()
            if rc != _lib.SLK_ERR_UNSUPPORTED:                 # (unsupported: rows of x that are not 16-byte aligned)
                _lib.check(rc, "Gru")
                return y
            plan = "scan"
        # projection GEMM into a workspace, then the scan
        nbytes = L.slk_gru_workspace_bytes(T, B, n)
        ws = _scratch(nbytes, torch.uint8, x.device)
        _projection(self, x, self.iW, self.b, ws.data_ptr(), rows, self.insize, 3 * n, "gru_input_gemm", "Gru")
        # (roofline bookkeeping: the fp16-split scans take two MFMAs per recurrent product, the state's halves in different column groups;
        #  the float32 scan one fp32 MFMA)
        with profiler.region("gru_recurrent", 6.0 * rows * n * n, 4.0 * rows * 4 * n,
                             f16x2_flops=6.0 * rows * n * n if plan == "scan16" else 0.0):
            rc = _lib.SLK_ERR_UNSUPPORTED
            if plan == "scan16":
                # n = 112 / 128 / 144: the barrier-stepped scan on the fp16 split (csrc/gru_scan16.hip, gru_scan1t.hip; unsupported:
                # a projection of 4 GiB or more, its lanes keep 32-bit offsets)
                rc = L.slk_gru_scan16_f32(ws.data_ptr(), 3 * n, self.sW.dev().data_ptr(), self.sW2.dev().data_ptr(), y.data_ptr(),
                                          _row_stride(y), T, B, n, int(reverse), act, gact, lens_p, _stream())
            if rc != _lib.SLK_ERR_UNSUPPORTED:
                pass
            elif lens is None:                             # float32 MFMA (csrc/recurrent.hip), portable kernel beyond n = 144
                rc = L.slk_gru_recurrent_f32(ws.data_ptr(), self.sW.dev().data_ptr(), self.sW2.dev().data_ptr(),
                                             y.data_ptr(), _row_stride(y), T, B, n, int(reverse), act, gact, _stream())
            else:
                rc = L.slk_gru_recurrent_ragged_f32(ws.data_ptr(), self.sW.dev().data_ptr(), self.sW2.dev().data_ptr(),
                                                    y.data_ptr(), _row_stride(y), T, B, n, int(reverse), act, gact,
                                                    lens_p, _stream())
        _lib.check(rc, "Gru")
        return y

    def spec(self):
        return {"type": "GRU", "iW": self.iW.get_value(), "sW": self.sW.get_value(), "sW2": self.sW2.get_value(),
                "b": self.b.get_value(), "activation": activation.act_name(self.fun),
                "gate": activation.act_name(self.gatefun)}


class Reverse(Layer):
    """  Runs a recurrent layer in reverse time (backwards)       (layers.py:1420-1450)

    :param layer: the wrapped layer; it sees the input back to front
    :param name: label that json() and pickles carry
    """

    def __init__(self, layer, name='Reverse'):
        self.layer = layer
        self._name = name

    @property
    def insize(self):
        return self.layer.insize

    @property
    def size(self):
        return self.layer.size

    def params(self):
        return self.layer.params()

    def json(self, params=False):
        return OrderedDict([('type', "reverse"),
                            ('sublayer', self.layer.json(params))])

    def set_params(self, values):
        return

    def _forward(self, x, out, reverse):
        return self.layer._forward(x, out, not reverse)

    def spec(self):
        return {"type": "reverse", "sublayer": self.layer.spec()}


class Parallel(Layer):
    """ Run multiple layers in parallel (all have same input and outputs are concatenated)   (layers.py:1453-1487)

    :param layers: layers that all read the same input; their outputs are joined along the feature axis
    :param name: label that json() and pickles carry
    """

    def __init__(self, layers, name='Parallel'):
        assert len(layers) > 0, "A Parallel layer cannot be empty"
        self.layers = layers
        self._name = name
        is_consistent = all(x.insize == self.insize for x in self.layers)
        assert is_consistent, "Parallel layer has inconsistent sizes"

    @property
    def insize(self):
        return self.layers[0].insize

    @property
    def size(self):
        return sum(x.size for x in self.layers)

    def params(self):
        return reduce(lambda x, y: x + y.params(), self.layers, [])

    def json(self, params=False):
        return OrderedDict([('type', "parallel"),
                            ('sublayers', [layer.json(params) for layer in self.layers])])

    def set_params(self, values):
        return

    def _forward(self, x, out, reverse):
        import torch
        T, B, _ = x.shape
        if all(_keeps_time(layer) for layer in self.layers):
            # every sub-layer writes its slice of the concatenated tensor directly (row stride = self.size)
            outs = out if out is not None else _scratch((T, B, self.size), torch.float32, x.device)
            streams = self._side_streams(x, B)
            off = 0
            if streams is None:
                for layer in self.layers:
                    layer._forward(x, outs[:, :, off:off + layer.size], reverse)
                    off += layer.size
                return outs
            # The sub-layers are independent (same input, disjoint output slices).  A recurrent layer occupies one
            # workgroup per 4 chunks for the whole scan, so at small batches most CUs idle: run the directions of a
            # birnn side by side on their own HIP streams (the reference runs them one after the other inside one
            # Theano function, layers.py:1486-1487).
            main = torch.cuda.current_stream(x.device)
            ready = torch.cuda.Event()
            ready.record(main)
            keep = _HINTS.gru_plan_bits
            _HINTS.gru_plan_bits = self._side_plan
            try:
                for i, layer in enumerate(self.layers):
                    st = main if i == 0 else streams[i - 1]
                    if st is not main:
                        st.wait_event(ready)
                    with torch.cuda.stream(st):
                        layer._forward(x, outs[:, :, off:off + layer.size], reverse)
                    if st is not main:
                        done = torch.cuda.Event()
                        done.record(st)
                        main.wait_event(done)
                    off += layer.size
            finally:
                _HINTS.gru_plan_bits = keep
            return outs
        cat = torch.cat([layer._forward(x, None, reverse) for layer in self.layers], dim=2)
        if out is not None:
            out.copy_(cat)
            return out
        return cat

    _streams_cache = {}

    def _side_streams(self, x, B):
        """HIP streams for sub-layers 1.. when running them concurrently pays: all sub-layers recurrent and their
        workgroups (one per 4 chunks each) fit the device's CUs together."""
        import torch
        if len(self.layers) < 2:
            return None
        for layer in self.layers:
            inner = layer.layer if isinstance(layer, Reverse) else layer
            if not isinstance(inner, (Gru, Lstm)):
                return None
        ncu = _cu_count(x.device)
        self._side_plan = 0
        share = len(self.layers) * max(1, _HINTS.in_flight)
        inners = [l.layer if isinstance(l, Reverse) else l for l in self.layers]
        if all(isinstance(l, Lstm) and l.takes_fused_kernel() for l in inners):
            ncu *= 2                   # csrc/lstm_fused16.hip fits two workgroups on a CU: the directions fill each other's waits
        if ((B + 3) // 4) * share > ncu:
            # too many four-chunk workgroups to run together; eight chunks per workgroup (csrc/gru_bar16d.hip: 1.4 x the step
            # time for twice the chunks) may still let the directions share the chip: B = 1024, two directions -> 2 x 128
            grus = all(isinstance(l.layer if isinstance(l, Reverse) else l, Gru) for l in self.layers)
            plan = _gru_plan_for(B, share, ncu, 2 if grus and all(max(l.size, l.insize) <= 64 for l in inners) else 1)
            if not (grus and SPLIT_F16 and RECURRENT_F16 and plan):
                return None
            self._side_plan = plan
        # side streams belong to the stream the caller runs on: batches in flight on different streams must not meet on one
        key = (x.device.index, len(self.layers), torch.cuda.current_stream(x.device).cuda_stream)
        if key not in Parallel._streams_cache:
            if len(Parallel._streams_cache) >= 64:                 # callers that keep creating streams: do not keep theirs for ever
                Parallel._streams_cache.clear()
            Parallel._streams_cache[key] = [torch.cuda.Stream(device=x.device) for _ in range(len(self.layers) - 1)]
        return Parallel._streams_cache[key]

    def spec(self):
        return {"type": "parallel", "sublayers": [l.spec() for l in self.layers]}


class Serial(Layer):
    """ Run multiple layers serially: output of a layer is the input for the next layer  (layers.py:1524-1560)

    :param layers: layers applied one after the other, first to last
    :param name: label that json() and pickles carry
    """

    def __init__(self, layers, name='Serial'):
        assert len(layers) > 0, "A Serial layer cannot be empty"
        self.layers = layers
        self._name = name
        is_consistent = all(x.size == y.insize for x, y in zip(layers, layers[1:]))
        assert is_consistent, "Serial layer has inconistent sizes"

    @property
    def insize(self):
        return self.layers[0].insize

    @property
    def size(self):
        return self.layers[-1].size

    def params(self):
        return reduce(lambda x, y: x + y.params(), self.layers, [])

    def json(self, params=False):
        return OrderedDict([('type', "serial"),
                            ('sublayers', [layer.json(params) for layer in self.layers])])

    def set_params(self, values):
        return

    def _forward(self, x, out, reverse):
        if reverse:
            return _flip_run(self, x, out)
        tmp = x
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            tmp = layer._forward(tmp, out if i == last else None, False)
        return tmp

    def spec(self):
        return {"type": "serial", "sublayers": [l.spec() for l in self.layers]}


def _keeps_time(layer):
    """True for layers whose output has the input's time length and that can write a strided slice."""
    if isinstance(layer, Reverse):
        return _keeps_time(layer.layer)
    return isinstance(layer, (RNN, FeedForward))


def birnn(forward, backward, name='BiRNN'):
    """  Creates a bidirectional RNN from two RNNs      (layers.py:1622-1629)

    :param forward: layer that reads the input in time order
    :param backward: layer that reads it back to front (wrapped in Reverse)
    :param name: label that json() and pickles carry
    """
    return Parallel([forward, Reverse(backward)], name=name)


def _unsupported(cls_name, where):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "sloika_amd: layer %s (sloika/layers.py:%s) is outside the accelerated basecalling path -- no shipped "
            "model uses it (see DESIGN.md, 'Out of scope')" % (cls_name, where))
    return type(cls_name, (object,), {"__init__": __init__})


# Exotic cells of the reference that no model in models/ uses: named so that `from sloika.layers import *`
# resolves, but constructing one fails loudly instead of silently running somewhere else.
Studentise = _unsupported("Studentise", "161-187")
NormaliseL1 = _unsupported("NormaliseL1", "190-219")
MaxPool = _unsupported("MaxPool", "422-465")
Recurrent = _unsupported("Recurrent", "468-520")
Scrn = _unsupported("Scrn", "523-596")
LstmCIFG = _unsupported("LstmCIFG", "700-798")
LstmO = _unsupported("LstmO", "801-883")
Forget = _unsupported("Forget", "886-949")
Mut1 = _unsupported("Mut1", "1024-1117")
Mut2 = _unsupported("Mut2", "1120-1224")
Mut3 = _unsupported("Mut3", "1227-1331")
Genmut = _unsupported("Genmut", "1334-1417")
Residual = _unsupported("Residual", "1490-1521")
Decode = _unsupported("Decode", "1563-1619")
