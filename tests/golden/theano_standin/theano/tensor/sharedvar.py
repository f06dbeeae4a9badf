from .._core import SharedVariable


class TensorSharedVariable(SharedVariable):
    """Class path found in the reference's model pickles (models/pretrained.pkl)."""
