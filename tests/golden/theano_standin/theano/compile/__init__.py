from . import function_module          # noqa: F401
