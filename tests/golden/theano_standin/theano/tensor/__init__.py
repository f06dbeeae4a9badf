"""Stand-in `theano.tensor` (see ../_core.py)."""
import builtins

import numpy as np

from .._core import Var, placeholder, _reduce, FLOATX
from . import nnet, signal, sharedvar, type as _type          # noqa: F401


def _nd(*vs):
    nds = [v.ndim if isinstance(v, Var) else np.ndim(v) for v in vs]
    return None if any(n is None for n in nds) else builtins.max(nds)


def _unary(op):
    def f(x):
        return Var(op, (x,), ndim=_nd(x))
    f.__name__ = op
    return f


exp, expm1, log, log1p, tanh, sqrt, sqr, sin, erf = (_unary(o) for o in
                                                      ("exp", "expm1", "log", "log1p", "tanh", "sqrt", "sqr", "sin", "erf"))
abs_ = _unary("abs")
square = sqr


def tensor3(name=None, dtype=None): return placeholder(3, dtype, name)
def matrix(name=None, dtype=None): return placeholder(2, dtype, name)
def vector(name=None, dtype=None): return placeholder(1, dtype, name)
def scalar(name=None, dtype=None): return placeholder(0, dtype, name)
def fmatrix(name=None): return placeholder(2, "float32", name)
def imatrix(name=None): return placeholder(2, "int32", name)
def ivector(name=None): return placeholder(1, "int32", name)


def shape(x): return x.shape
def tensordot(a, b, axes=2): return Var("tensordot", (a, b), {"axes": axes}, ndim=None if _nd(a, b) is None else a.ndim + b.ndim - 2 * (axes if isinstance(axes, int) else (len(axes[0]) if isinstance(axes[0], (tuple, list)) else 1)))
def dot(a, b): return Var("tensordot", (a, b), {"axes": 1}, ndim=None)
def concatenate(tensors, axis=0): return Var("concatenate", (list(tensors),), {"axis": axis}, ndim=_nd(*tensors))
def zeros(shape, dtype=None): return Var("zeros", (shape,), {"dtype": dtype}, ndim=len(shape) if isinstance(shape, (tuple, list)) else None)
def zeros_like(x): return Var("zeros_like", (x,), ndim=_nd(x))
def repeat(x, repeats, axis=None): return Var("repeat", (x, repeats), {"axis": axis}, ndim=_nd(x) if axis is not None else 1)
def reshape(x, shape, ndim=None): return x.reshape(shape, ndim)
def shape_padleft(x, n_ones=1): return Var("shape_pad", (x,), {"axis": 0, "n": n_ones}, ndim=None if _nd(x) is None else x.ndim + n_ones)
def shape_padright(x, n_ones=1): return Var("shape_pad", (x,), {"axis": -1, "n": n_ones}, ndim=None if _nd(x) is None else x.ndim + n_ones)
def shape_padaxis(x, axis): return Var("shape_pad", (x,), {"axis": axis, "n": 1}, ndim=None if _nd(x) is None else x.ndim + 1)
def switch(c, a, b): return Var("switch", (c, a, b), ndim=_nd(c, a, b))
def clip(x, lo, hi): return Var("clip", (x, lo, hi), ndim=_nd(x))
def maximum(a, b): return Var("maximum", (a, b), ndim=_nd(a, b))
def gt(a, b): return Var("gt", (a, b), ndim=_nd(a, b))
def eq(a, b): return Var("eq", (a, b), ndim=_nd(a, b))
def constant(value, dtype=None, name=None): return Var("constant", (), {"value": value, "dtype": dtype}, ndim=np.ndim(value))
def arange(start, stop=None, step=1, dtype=None): return Var("arange", (start, stop, step), {"dtype": dtype}, ndim=1)
def cast(x, dtype): return x.astype(dtype)


def sum(x, axis=None, keepdims=False, dtype=None, acc_dtype=None):
    return _reduce("sum", x, axis, keepdims, dtype=dtype)


def mean(x, axis=None, keepdims=False, dtype=None, acc_dtype=None):
    return _reduce("mean", x, axis, keepdims, dtype=dtype, acc_dtype=acc_dtype)


def var(x, axis=None, keepdims=False): return _reduce("var", x, axis, keepdims)
def max(x, axis=None, keepdims=False): return _reduce("max", x, axis, keepdims)
def argmax(x, axis=None): return Var("argmax", (x,), {"axis": axis}, ndim=None if _nd(x) is None else (0 if axis is None else x.ndim - 1))
