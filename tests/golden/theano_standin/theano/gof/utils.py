from .._core import Inert


class scratchpad(Inert):
    pass
