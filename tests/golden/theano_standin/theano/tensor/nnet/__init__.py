"""Stand-in `theano.tensor.nnet`.  `sigmoid` is the plain logistic: Theano's float32 C code additionally
returns exactly 0 below -88.7 and exactly 1 above 15 (a <= 3.1e-7 difference) -- that clip is not reproduced."""
from ..._core import Var


def sigmoid(x): return Var("sigmoid", (x,), ndim=x.ndim)
def relu(x, alpha=0): return Var("relu", (x,), {"alpha": alpha}, ndim=x.ndim)
def softmax(x): return Var("softmax", (x,), ndim=2)
def categorical_crossentropy(coding_dist, true_dist): return Var("xent", (coding_dist, true_dist), ndim=1)


def conv2d(input, filters, input_shape=None, filter_shape=None, border_mode="valid", subsample=(1, 1),
           filter_flip=True, **kw):
    return Var("conv2d", (input, filters), {"subsample": tuple(subsample), "border_mode": border_mode,
                                            "filter_flip": filter_flip}, ndim=4)
