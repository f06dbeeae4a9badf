"""Activation functions of sloika/activation.py:8-115 as named handles.

In the reference these are Theano expression builders; layers store them in `fun` / `gatefun` and pickles
reference them by qualified name (`sloika.activation.tanh`).  Here each is a small callable object carrying
the id the HIP kernels switch on; the arithmetic itself lives in csrc/common.h (`slk_act`).  Calling one on an
array runs the stand-alone elementwise kernel (`slk_activation_f32`).
"""
from . import _lib


class Activation(object):
    def __init__(self, name, act_id):
        self.__name__ = name
        self.act_id = act_id

    def __call__(self, x):
        import torch
        from . import device as D
        xd = D.to_dev(x)
        y = torch.empty_like(xd)
        _lib.check(_lib.lib().slk_activation_f32(xd.data_ptr(), y.data_ptr(), xd.numel(), self.act_id,
                                                 D.stream_ptr()), "activation." + self.__name__)
        return D.like_input(y, x)

    def __repr__(self):
        return "<sloika_amd.activation.%s>" % self.__name__

    def __reduce__(self):
        # pickles by qualified name, exactly like a module-level function of the reference's activation.py (the
        # reference's pickles name sloika.activation.<name> and load here; files written here load here)
        return (_lookup, (self.__name__,))


def _lookup(name):
    if name not in _NAMES:               # a pickle's reducer: activation functions only, never another global of this module
        raise ValueError("not an activation function: %r" % (name,))
    return globals()[name]


_NAMES = ["linear", "tanh", "sigmoid", "elu", "relu", "relu_smooth", "softplus", "exp", "erf", "L1mL2", "fair",
          "retu", "tanh_pm", "sigmoid_pm", "bounded_linear", "sin", "cauchy", "geman_mcclure", "welsh"]

for _i, _n in enumerate(_NAMES):
    globals()[_n] = Activation(_n, _i)
del _i, _n

__all__ = list(_NAMES)


def act_id(fun):
    """Kernel id of an activation given as handle, name, or foreign function object with a __name__."""
    if isinstance(fun, Activation):
        return fun.act_id
    name = fun if isinstance(fun, str) else getattr(fun, "__name__", None)
    if name in _NAMES:
        return _NAMES.index(name)
    raise ValueError("unknown activation %r" % (fun,))


def act_name(fun):
    return _NAMES[act_id(fun)]
