from . import pool          # noqa: F401
