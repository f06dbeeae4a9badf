from .._core import Container          # noqa: F401
