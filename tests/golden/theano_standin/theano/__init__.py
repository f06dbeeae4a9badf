"""Stand-in `theano` namespace (see _core.py): test infrastructure for generating golden vectors only."""
from ._core import (config, shared, function, scan, map, grad, In, Out, Function, Var, SharedVariable)
from . import tensor, gradient, gof, compile, sandbox          # noqa: F401

__version__ = "0.8.2-standin"
