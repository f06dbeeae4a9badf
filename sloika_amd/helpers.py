"""Model files -> runnable networks.

    load_model(path)        a pickled sloika.layers object graph (what bin/train_network.py:145-152 writes and
                            sloika/helpers.py:28-39 reads), or a models/*.py factory file
    compile_model(path)     sloika/helpers.py:54-79 boundary: model file -> callable posterior function
                            (no child process / temp file: that machinery exists only to keep Theano fork-safe)

Direction of compatibility: the reference's pickles (models/pretrained.pkl, train_network.py checkpoints) load here;
checkpoints written by sloika_amd.train.save_model name this package's classes and load here only -- the reference
could not read them anyway without rebuilding Theano shared variables.

Reference pickles hold Theano shared variables at their leaves.  Theano is not needed to read them: the
unpickler maps `theano.*` classes to inert holders, then every holder whose state carries a numpy array in
`.container.storage[0]` is replaced by a `layers.Shared`.
"""
import importlib.util
import os
import pickle

import numpy as np

from . import compat, layers


class _Holder(object):
    """Inert stand-in for any theano.* class met while unpickling."""

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state

    def get_value(self, borrow=False):
        return _shared_value(self)


def _shared_value(obj):
    cont = getattr(obj, "container", None)
    storage = getattr(cont, "storage", None)
    if isinstance(storage, list) and storage and isinstance(storage[0], np.ndarray):
        return storage[0]
    raise TypeError("object does not look like a Theano shared variable")


#: Globals a model pickle may name besides theano.* (inert holders) and sloika.* / sloika_amd.* (this package's classes):
#: what numpy needs to rebuild arrays, and the two containers layer objects hold.  Anything else -- os.system, builtins.eval
#: ... -- is refused: a model file is data, loading one must not run code (the reference's plain pickle.load would).
_ALLOWED_GLOBALS = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),      # protocol-5 array pickles
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("collections", "OrderedDict"), ("functools", "partial"),
    ("copyreg", "_reconstructor"), ("copy_reg", "_reconstructor"), ("builtins", "object"), ("__builtin__", "object"),
}


#: modules of this package (and, through compat, of `sloika`) whose classes / functions a model pickle may name
_MODEL_MODULES = ("layers", "activation")


def _model_global(module, name):
    """Resolve `module.name` for a sloika / sloika_amd global of a model pickle: only layer classes (and the Shared leaf)
    of sloika_amd.layers and the activation functions (and their `_lookup` reducer) of sloika_amd.activation.  Dotted names
    are refused outright: protocol >= 4 resolves them attribute by attribute, which would reach `sloika_amd.build.subprocess`
    or `sloika_amd._lib.os` -- any module a module of this package happens to import."""
    from . import activation
    if "." in name:
        raise pickle.UnpicklingError("model file names the dotted global %s.%s; refusing to load it" % (module, name))
    top, _, sub = module.partition(".")
    if top not in ("sloika", "sloika_amd") or sub not in _MODEL_MODULES:
        raise pickle.UnpicklingError("model file names %s.%s: only layers and activations may come from %s" % (module, name, top))
    obj = getattr(layers if sub == "layers" else activation, name, None)
    if sub == "layers":
        ok = isinstance(obj, type) and obj.__module__ == layers.__name__ and (issubclass(obj, layers.Layer) or obj is layers.Shared)
    else:
        ok = obj is not None and (obj is activation._lookup or (callable(obj) and name in activation._NAMES))
    if not ok:
        raise pickle.UnpicklingError("model file names %s.%s, which is not a layer class or an activation function; refusing "
                                     "to load it" % (module, name))
    return obj


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "theano" or module.startswith("theano."):
            if "function_module" in module or name in ("Function", "FunctionMaker"):
                raise pickle.UnpicklingError(
                    "this file holds a COMPILED Theano function (sloika/helpers.py:40-47 output); it is not "
                    "portable -- pass the model pickle (a sloika.layers object) instead")
            return type(str(name).replace(".", "_"), (_Holder,), {"__module__": module})
        if module in ("sloika", "sloika_amd") or module.startswith("sloika.") or module.startswith("sloika_amd."):
            return _model_global(module, name)
        if (module, name) in _ALLOWED_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("model file names the global %s.%s, which a sloika model does not need; refusing to "
                                     "load it" % (module, name))


def _convert_leaves(obj, seen=None):
    """Replace Theano shared-variable holders by layers.Shared, in place, through the layer graph."""
    seen = seen if seen is not None else set()
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, (list, tuple)):
        for o in obj:
            _convert_leaves(o, seen)
        return
    if not isinstance(obj, layers.Layer):
        return
    for key, val in list(vars(obj).items()):
        if isinstance(val, _Holder):
            try:
                setattr(obj, key, layers.Shared(_shared_value(val)))
            except TypeError:
                pass
        elif isinstance(val, (layers.Layer, list, tuple)):
            _convert_leaves(val, seen)
    if isinstance(obj, layers.Convolution) and isinstance(getattr(obj, "padding", None), list):
        obj.padding = tuple(obj.padding)


def load_pickle(path):
    with open(path, "rb") as fh:
        try:
            net = _Unpickler(fh).load()
        except UnicodeDecodeError:
            fh.seek(0)
            net = _Unpickler(fh, encoding="latin1").load()       # py2 pickles, helpers.py:33-39
    if not isinstance(net, layers.Layer):
        raise TypeError("%s does not contain a sloika layer object (got %r)" % (path, type(net)))
    _convert_leaves(net)
    return net


def load_factory(path, **kwargs):
    """Import a models/*.py file (bin/train_network.py:266-270 uses imp.load_source) and call network(**kwargs)."""
    compat.install()
    spec = importlib.util.spec_from_file_location("netmodule", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.network(**kwargs)


def load_model(path, **factory_kwargs):
    if os.path.splitext(path)[1] == ".py":
        return load_factory(path, **factory_kwargs)
    return load_pickle(path)


def compile_model(model_file, output_file=None):
    """Model file -> posterior function f([T,B,F]) -> [T',B,nstate] (sloika/helpers.py:54-79)."""
    net = load_model(model_file)
    return net.compile()
