from ..._core import Var


def pool_2d(input, ds, ignore_border=None, st=None, padding=(0, 0), mode="max"):
    return Var("pool2d", (input,), {"ds": tuple(ds), "st": None if st is None else tuple(st),
                                    "ignore_border": bool(ignore_border), "mode": mode}, ndim=4)
