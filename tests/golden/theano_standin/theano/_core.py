"""Eager/lazy stand-in for the handful of Theano 0.8 primitives sloika's layer code calls.

TEST INFRASTRUCTURE, used by tests/golden/make_layer_goldens.py ONLY, in the build container ONLY.
Theano is an un-vendored dependency of the reference (requirements.txt: Theano==0.8.2) and cannot be
installed here.  This package lets the reference's `sloika/layers.py`, `conv.py`, `activation.py`,
`updates.py` and `bin/train_network.py:wrap_network` be imported and executed UNMODIFIED: symbolic
expressions become a small graph of named ops that is evaluated with torch CPU tensors when the
"compiled" function is called; `theano.grad` is torch autograd over that evaluation.

What fixtures produced under it pin: the reference's own layer code -- gate order, reshapes, padding,
scan order, initial states, slicing, loss and optimiser formulas.  What they do NOT pin: Theano's own
float32 kernels (BLAS summation order, its clipped float32 sigmoid), which were never run.
Nothing here is product code and nothing here travels to the GPU box except as the arrays it produced.
"""
import os

import numpy as np
import torch

FLOATX = os.environ.get("THEANO_STANDIN_FLOATX", "float32")
_TORCH_DTYPE = {"float32": torch.float32, "float64": torch.float64, "int32": torch.int32, "int64": torch.int64,
                "int8": torch.int8, "bool": torch.bool}


class config:                                     # `from theano import config; config.floatX`
    floatX = FLOATX


def _fx():
    return _TORCH_DTYPE[FLOATX]


# ------------------------------------------------------------------------------------------ graph
class Var:
    """A node: op name + inputs (Vars, constants, nested lists/tuples/slices of them) + static params."""
    __array_priority__ = 1000.0
    __array_ufunc__ = None                        # numpy scalars/arrays defer to our reflected operators

    def __init__(self, op, inputs=(), params=None, ndim=None, name=None):
        self.op = op
        self.inputs = tuple(inputs)
        self.params = dict(params or {})
        self._ndim = ndim
        self.name = name

    # -- structure
    @property
    def ndim(self):
        return self._ndim

    @property
    def shape(self):
        return Var("shape", (self,), ndim=None, name=None, params={"length": self._ndim})

    def __iter__(self):
        n = self.params.get("length") if self.op == "shape" else None
        if n is None:
            raise TypeError("symbolic variable of unknown length is not iterable")
        return iter([self[i] for i in range(n)])

    def __bool__(self):
        raise TypeError("truth value of a symbolic variable")

    def __getitem__(self, idx):
        nd = self._ndim
        if nd is not None:
            items = idx if isinstance(idx, tuple) else (idx,)
            nd = nd - sum(1 for i in items if not isinstance(i, slice) and i is not None) \
                + sum(1 for i in items if i is None)
        return Var("getitem", (self, idx), ndim=nd)

    # -- arithmetic
    def _bin(self, op, other, swap=False):
        a, b = (other, self) if swap else (self, other)
        nds = [v._ndim for v in (a, b) if isinstance(v, Var)]
        nd = None if any(n is None for n in nds) else max(nds + [np.ndim(v) for v in (a, b) if not isinstance(v, Var)])
        return Var(op, (a, b), ndim=nd)

    def __add__(self, o): return self._bin("add", o)
    def __radd__(self, o): return self._bin("add", o, True)
    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, True)
    def __truediv__(self, o): return self._bin("div", o)
    def __rtruediv__(self, o): return self._bin("div", o, True)
    def __pow__(self, o): return self._bin("pow", o)
    def __neg__(self): return Var("neg", (self,), ndim=self._ndim)
    def __gt__(self, o): return self._bin("gt", o)
    def __lt__(self, o): return self._bin("lt", o)
    def __ge__(self, o): return self._bin("ge", o)
    def __le__(self, o): return self._bin("le", o)
    __hash__ = object.__hash__

    # -- tensor methods the reference calls
    def transpose(self, *axes):
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        return Var("transpose", (self,), {"axes": tuple(axes)}, ndim=self._ndim)

    def dimshuffle(self, *pattern):
        if len(pattern) == 1 and isinstance(pattern[0], (tuple, list)):
            pattern = tuple(pattern[0])
        return Var("dimshuffle", (self,), {"pattern": tuple(pattern)}, ndim=len(pattern))

    def flatten(self, ndim=1):
        return Var("flatten", (self,), {"ndim": ndim}, ndim=ndim)

    def reshape(self, shape, ndim=None):
        return Var("reshape", (self, shape), ndim=len(shape) if isinstance(shape, (tuple, list)) else ndim)

    def astype(self, dtype):
        return Var("cast", (self,), {"dtype": str(dtype)}, ndim=self._ndim)

    def sum(self, axis=None, keepdims=False): return _reduce("sum", self, axis, keepdims)
    def mean(self, axis=None, keepdims=False): return _reduce("mean", self, axis, keepdims)
    def max(self, axis=None, keepdims=False): return _reduce("max", self, axis, keepdims)

    @property
    def T(self):
        return Var("transpose", (self,), {"axes": ()}, ndim=self._ndim)

    def get_scalar_constant_value(self):
        assert self.op == "constant", "not a constant"
        return np.asarray(self.params["value"])

    def eval(self, givens=None):
        env = _new_env()
        for k, v in (givens or {}).items():
            env[id(k)] = _to_tensor(v)
        return _to_numpy(_eval(self, env))


def _reduce(op, x, axis, keepdims, **extra):
    nd = x._ndim
    if nd is not None and not keepdims:
        nd = 0 if axis is None else nd - (len(axis) if isinstance(axis, (tuple, list)) else 1)
    return Var(op, (x,), dict(axis=axis, keepdims=keepdims, **extra), ndim=nd)


class SharedVariable(Var):
    """`theano.shared(value)`; also what `theano.tensor.sharedvar.TensorSharedVariable` unpickles to: the
    reference's model pickles restore it through NEWOBJ + a state dict holding `container.storage[0]`."""

    def __new__(cls, *a, **k):
        self = object.__new__(cls)
        Var.__init__(self, "shared")
        return self

    def __init__(self, value=None, name=None, **_ignored):
        Var.__init__(self, "shared", name=name)
        self.container = Container()
        self.container.storage = [np.array(value)]
        self._ndim = self.container.storage[0].ndim

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.op, self.inputs, self.params = "shared", (), {}
        self._ndim = np.ndim(self.container.storage[0])

    def __reduce_ex__(self, protocol):
        return (_rebuild_shared, (type(self), self.get_value(), self.name))

    def get_value(self, borrow=False, return_internal_type=False):
        v = self.container.storage[0]
        return v if borrow else np.array(v, copy=True)

    def set_value(self, value, borrow=False):
        old = self.container.storage[0]
        self.container.storage[0] = np.asarray(value, dtype=old.dtype).reshape(np.shape(value)).copy()
        self._ndim = self.container.storage[0].ndim


def _rebuild_shared(cls, value, name):
    return cls(value, name=name)


class Container:
    """theano.gof.link.Container: the pickles keep the array in `.storage[0]`."""
    def __init__(self, *a, **k):
        self.storage = [None]


class Inert:
    """theano.tensor.type.TensorType, theano.gof.utils.scratchpad: state holders found in model pickles."""
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)


def shared(value, name=None, **kw):
    return SharedVariable(np.asarray(value), name=name)


def placeholder(ndim, dtype=None, name=None):
    return Var("placeholder", (), {"dtype": dtype or FLOATX}, ndim=ndim, name=name)


class In:
    def __init__(self, variable, borrow=False, **kw):
        self.variable = variable


class Out:
    def __init__(self, variable, borrow=False, **kw):
        self.variable = variable


# ------------------------------------------------------------------------------------------ evaluation
def _new_env():
    return {"__leaves__": {}}


def _to_tensor(v, dtype=None):
    if isinstance(v, torch.Tensor):
        return v
    a = np.asarray(v)
    t = torch.from_numpy(np.ascontiguousarray(a)) if a.ndim else torch.tensor(a.item(), dtype=_TORCH_DTYPE.get(str(a.dtype)))
    return t.clone()


def _to_numpy(v):
    if isinstance(v, torch.Tensor):
        return v.detach().numpy().copy()
    if isinstance(v, (tuple, list)):
        return type(v)(_to_numpy(q) for q in v)
    return np.asarray(v)


def _eval(obj, env):
    if isinstance(obj, Var):
        key = id(obj)
        if key in env:
            return env[key]
        if obj.op == "placeholder":
            raise RuntimeError("unbound symbolic input %r" % (obj.name,))
        if obj.op == "shared":
            leaves = env["__leaves__"]
            if key not in leaves:
                leaves[key] = torch.from_numpy(np.array(obj.get_value())).clone().requires_grad_(
                    obj.container.storage[0].dtype.kind == "f")
            return leaves[key]
        if obj.op in _SPECIAL:
            val = _SPECIAL[obj.op](obj, env)
        else:
            args = [_eval(i, env) for i in obj.inputs]
            val = _OPS[obj.op](*args, **obj.params)
        env[key] = val
        return val
    if isinstance(obj, (list, tuple)):
        return type(obj)(_eval(i, env) for i in obj)
    if isinstance(obj, slice):
        return slice(_eval(obj.start, env), _eval(obj.stop, env), _eval(obj.step, env))
    if isinstance(obj, np.generic):                       # numpy scalars act as weakly typed Python scalars
        return obj.item()
    if isinstance(obj, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(obj)) if obj.ndim else obj.item()
    return obj


def _as_int(v):
    return int(v.item()) if isinstance(v, torch.Tensor) else (None if v is None else int(v))


def _getitem(x, idx):
    if isinstance(x, (tuple, list)):                      # a shape
        return x[idx if not isinstance(idx, torch.Tensor) else int(idx)]
    items = idx if isinstance(idx, tuple) else (idx,)
    dim = 0
    for it in items:                                      # torch has no negative-step slices: gather instead
        if isinstance(it, slice):
            start, stop, step = _as_int(it.start), _as_int(it.stop), _as_int(it.step)
            if step is not None and step < 0:
                sel = torch.arange(x.shape[dim])[slice(None)].tolist()[slice(start, stop, step)]
                x = x.index_select(dim, torch.tensor(sel, dtype=torch.long))
            else:
                x = x[(slice(None),) * dim + (slice(start, stop, step),)]
            dim += 1
        elif it is None:
            x = x.unsqueeze(dim)
            dim += 1
        elif isinstance(it, torch.Tensor) and it.ndim > 0:
            x = x.index_select(dim, it.long())
            dim += 1
        else:
            x = x.select(dim, _as_int(it))
    return x


def _bcast(a, b):
    """Python scalars stay weakly typed; everything else becomes a tensor (bools as int8 like Theano)."""
    def conv(v):
        if isinstance(v, torch.Tensor) and v.dtype == torch.bool:
            return v.to(torch.int8)
        return v
    return conv(a), conv(b)


def _axis(axis):
    return tuple(axis) if isinstance(axis, (tuple, list)) else axis


def _op_sum(x, axis=None, keepdims=False, dtype=None):
    x = x if dtype is None else x.to(_TORCH_DTYPE[dtype])
    return x.sum() if axis is None else x.sum(dim=_axis(axis), keepdim=keepdims)


def _op_mean(x, axis=None, keepdims=False, dtype=None, acc_dtype=None):
    if dtype is not None:
        x = x.to(_TORCH_DTYPE[dtype])
    elif not x.dtype.is_floating_point:
        x = x.to(_fx())
    return x.mean() if axis is None else x.mean(dim=_axis(axis), keepdim=keepdims)


def _op_var(x, axis=None, keepdims=False):
    return x.var(unbiased=False) if axis is None else x.var(dim=_axis(axis), unbiased=False, keepdim=keepdims)


def _op_max(x, axis=None, keepdims=False):
    if axis is None:
        return x.max()
    if isinstance(axis, (tuple, list)):
        return x.amax(dim=tuple(axis), keepdim=keepdims)
    return x.max(dim=axis, keepdim=keepdims)[0]


def _promote(a, b):
    """Theano upcasts mixed float32/float64 operands; torch's contractions insist on equal dtypes."""
    dt = torch.promote_types(a.dtype, b.dtype)
    return a.to(dt), b.to(dt)


def _op_tensordot(a, b, axes=2):
    if isinstance(axes, int):
        return torch.tensordot(*_promote(a, b), dims=axes)
    ax_a, ax_b = axes
    ax_a = list(ax_a) if isinstance(ax_a, (tuple, list)) else [ax_a]
    ax_b = list(ax_b) if isinstance(ax_b, (tuple, list)) else [ax_b]
    a, b = _promote(a, b)
    return torch.tensordot(a, b, dims=(ax_a, ax_b))


def _op_zeros(shape, dtype=None):
    shape = shape if isinstance(shape, (tuple, list)) else (shape,)
    return torch.zeros(tuple(_as_int(s) for s in shape), dtype=_TORCH_DTYPE[dtype or FLOATX])


def _op_repeat(x, repeats, axis=None):
    repeats = _as_int(repeats)
    if axis is None:
        return x.reshape(-1).repeat_interleave(repeats)
    if repeats == 0:
        shp = list(x.shape)
        shp[axis] = 0
        return x.new_zeros(shp)
    return x.repeat_interleave(repeats, dim=axis)


def _op_shape_pad(x, axis, n=1):
    for _ in range(n):
        x = x.unsqueeze(axis)
    return x


def _op_flatten(x, ndim=1):
    return x.reshape(tuple(x.shape[:ndim - 1]) + (-1,))


def _op_transpose(x, axes=()):
    return x.permute(*axes) if axes else x.permute(*reversed(range(x.ndim)))


def _op_dimshuffle(x, pattern):
    kept = [p for p in pattern if p != "x"]
    x = x.permute(*kept)
    for i, p in enumerate(pattern):
        if p == "x":
            x = x.unsqueeze(i)
    return x


def _op_conv2d(inp, filters, subsample=(1, 1), border_mode="valid", filter_flip=True):
    assert border_mode == "valid", "stand-in implements the mode conv.conv_1d uses"
    if filter_flip:
        filters = torch.flip(filters, dims=(2, 3))
    inp, filters = _promote(inp, filters)
    return torch.nn.functional.conv2d(inp, filters, stride=tuple(subsample))


def _op_pool2d(inp, ds, st=None, ignore_border=True, mode="max"):
    assert mode == "max" and ignore_border
    return torch.nn.functional.max_pool2d(inp, kernel_size=tuple(ds), stride=tuple(st or ds))


def _op_softmax(x):
    e = torch.exp(x - x.max(dim=1, keepdim=True)[0])
    return e / e.sum(dim=1, keepdim=True)


def _op_xent(coding, true):
    if true.dtype.is_floating_point:
        return -(true * torch.log(coding)).sum(dim=1)
    return -torch.log(coding[torch.arange(coding.shape[0]), true.long()])


def _op_switch(c, a, b):
    c = c if isinstance(c, torch.Tensor) else torch.tensor(c)
    a = a if isinstance(a, torch.Tensor) else torch.tensor(a, dtype=b.dtype if isinstance(b, torch.Tensor) else _fx())
    b = b if isinstance(b, torch.Tensor) else torch.tensor(b, dtype=a.dtype)
    return torch.where(c.bool(), a, b)


def _op_concat(tensors, axis=0):
    return torch.cat(list(tensors), dim=axis)


def _op_cast(x, dtype):
    dtype = FLOATX if dtype == "floatX" else dtype
    return x.to(_TORCH_DTYPE[dtype])


def _op_arange(start, stop=None, step=1, dtype=None):
    if stop is None:
        start, stop = 0, start
    vals = [_as_int(v) if float(v) == int(v) else float(v) for v in (start, stop, step)]
    floaty = any(isinstance(v, float) for v in (start, stop, step))
    dt = _TORCH_DTYPE[dtype] if dtype else (_fx() if floaty else torch.int64)
    return torch.arange(vals[0], vals[1], vals[2], dtype=dt)


def _elem(fn):
    def op(x):
        return fn(x if isinstance(x, torch.Tensor) else torch.tensor(x, dtype=_fx()))
    return op


def _binop(fn):
    def op(a, b):
        a, b = _bcast(a, b)
        return fn(a, b)
    return op


def _op_clip(x, lo, hi):
    return torch.clamp(x, min=lo, max=hi)


def _op_argmax(x, axis=None):
    return x.argmax() if axis is None else x.argmax(dim=axis)


def _op_relu(x, alpha=0):
    return torch.where(x > 0, x, alpha * x)


_OPS = {
    "add": _binop(lambda a, b: a + b), "sub": _binop(lambda a, b: a - b), "mul": _binop(lambda a, b: a * b),
    "div": _binop(lambda a, b: a / b), "pow": _binop(lambda a, b: a ** b), "neg": lambda a: -a,
    "gt": _binop(lambda a, b: a > b), "lt": _binop(lambda a, b: a < b), "ge": _binop(lambda a, b: a >= b),
    "le": _binop(lambda a, b: a <= b), "eq": _binop(lambda a, b: a == b),
    "maximum": _binop(lambda a, b: torch.maximum(a, b if isinstance(b, torch.Tensor) else torch.tensor(b, dtype=a.dtype))),
    "getitem": _getitem, "shape": lambda x, length=None: tuple(int(s) for s in x.shape),
    "transpose": _op_transpose, "dimshuffle": _op_dimshuffle, "flatten": _op_flatten,
    "reshape": lambda x, shape: x.reshape(tuple(_as_int(s) for s in shape)),
    "cast": _op_cast, "sum": _op_sum, "mean": _op_mean, "var": _op_var, "max": _op_max, "argmax": _op_argmax,
    "tensordot": _op_tensordot, "zeros": _op_zeros, "zeros_like": lambda x: torch.zeros_like(x),
    "repeat": _op_repeat, "shape_pad": _op_shape_pad, "concatenate": _op_concat,
    "conv2d": _op_conv2d, "pool2d": _op_pool2d, "softmax": _op_softmax, "xent": _op_xent, "switch": _op_switch,
    "clip": _op_clip, "relu": _op_relu, "arange": _op_arange,
    "constant": lambda value, dtype=None: torch.tensor(value, dtype=_TORCH_DTYPE[dtype] if dtype else None),
    "exp": _elem(torch.exp), "expm1": _elem(torch.expm1), "log": _elem(torch.log), "log1p": _elem(torch.log1p),
    "tanh": _elem(torch.tanh), "sigmoid": _elem(torch.sigmoid), "sqrt": _elem(torch.sqrt), "abs": _elem(torch.abs),
    "sqr": _elem(lambda x: x * x), "sin": _elem(torch.sin), "erf": _elem(torch.erf),
}


def _scan_eval(var, env):
    """theano.scan / theano.map: the step function was traced once on placeholders; replay it per time step."""
    seqs, init, nonseq = (_eval(i, env) for i in var.inputs)
    ph_seq, ph_state, inner = var.params["ph_seq"], var.params["ph_state"], var.params["inner"]
    n = seqs[0].shape[0]
    state = init
    outs = []
    for t in range(n):
        sub = dict(env)                                    # shares "__leaves__" (the autograd leaves) by reference
        for ph, s in zip(ph_seq, seqs):
            sub[id(ph)] = s[t]
        if ph_state is not None:
            sub[id(ph_state)] = state
        out = _eval(inner, sub)
        if ph_state is not None:
            state = out
        outs.append(out)
    return torch.stack(outs, dim=0)


def _grad_eval(var, env):
    cost = _eval(var.inputs[0], env)
    wrt = [_eval(w, env) for w in var.inputs[1]]
    grads = torch.autograd.grad(cost, wrt, retain_graph=True, allow_unused=True)
    return tuple(torch.zeros_like(w) if g is None else g for g, w in zip(grads, wrt))


_SPECIAL = {"scan": _scan_eval, "grad": _grad_eval}


def scan(fn, sequences=None, outputs_info=None, non_sequences=None, **kw):
    seqs = list(sequences) if isinstance(sequences, (list, tuple)) else [sequences]
    assert not kw.get("go_backwards"), "not used by the reference"
    ph_seq = [placeholder(None if s.ndim is None else s.ndim - 1, name="scan_seq") for s in seqs]
    ph_state = None
    args = list(ph_seq)
    if outputs_info is not None:
        assert isinstance(outputs_info, Var), "single recurrent output (all the reference uses)"
        ph_state = placeholder(outputs_info.ndim, name="scan_state")
        args.append(ph_state)
    nonseq = list(non_sequences) if isinstance(non_sequences, (list, tuple)) else ([] if non_sequences is None else [non_sequences])
    inner = fn(*(args + nonseq))
    out = Var("scan", (seqs, outputs_info, nonseq), {"ph_seq": ph_seq, "ph_state": ph_state, "inner": inner},
              ndim=None if inner.ndim is None else inner.ndim + 1)
    return out, {}


def map(fn, sequences=None, non_sequences=None, **kw):
    return scan(fn, sequences=sequences, outputs_info=None, non_sequences=non_sequences)


def grad(cost, wrt, **kw):
    single = not isinstance(wrt, (list, tuple))
    g = Var("grad", (cost, [wrt] if single else list(wrt)))
    return g[0] if single else [g[i] for i in range(len(wrt))]


class Function:
    """What `theano.function` returns: evaluates the graph on numpy inputs, applies `updates`, returns numpy."""

    def __init__(self, inputs, outputs, updates=None, **kw):
        self.inputs = [i.variable if isinstance(i, In) else i for i in inputs]
        self.single = not isinstance(outputs, (list, tuple))
        outs = [outputs] if self.single else list(outputs)
        self.outputs = [o.variable if isinstance(o, Out) else o for o in outs]
        self.updates = list(updates.items()) if hasattr(updates, "items") else list(updates or [])

    def __call__(self, *args):
        assert len(args) == len(self.inputs), "expected %d inputs" % len(self.inputs)
        env = _new_env()
        for var, val in zip(self.inputs, args):
            env[id(var)] = _to_tensor(val)
        outs = [_eval(o, env) for o in self.outputs]
        new = [(s, _eval(e, env)) for s, e in self.updates]
        for s, v in new:
            s.set_value(_to_numpy(v).astype(s.container.storage[0].dtype))
        outs = [_to_numpy(o) for o in outs]
        return outs[0] if self.single else outs


def function(inputs, outputs=None, updates=None, **kw):
    return Function(inputs, outputs, updates=updates)
