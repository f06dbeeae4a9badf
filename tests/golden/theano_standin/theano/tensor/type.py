from .._core import Inert


class TensorType(Inert):
    """Class path found in the reference's model pickles; carries dtype/broadcastable only."""
