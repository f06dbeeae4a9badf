from . import link, utils, compilelock          # noqa: F401
