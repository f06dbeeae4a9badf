from .._core import Function          # noqa: F401
