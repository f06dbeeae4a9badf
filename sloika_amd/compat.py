"""Make `import sloika.module_tools as smt` (what models/*.py do, e.g. models/raw_0.98_rgrgr.py:1) and the
GLOBAL opcodes of reference model pickles (`sloika.layers Gru`, `sloika.activation tanh`, ...) resolve to this
package, so that model factories and trained .pkl files load unchanged.

`install()` registers alias entries in sys.modules; it refuses to shadow a real, importable `sloika`.
"""
import importlib
import importlib.util
import sys
import types

_SUBMODULES = ["module_tools", "layers", "activation", "variables", "config", "conv", "decode", "transducer",
               "viterbi_helpers", "bio", "util", "basecall", "batch", "helpers"]


def install(force=False):
    if "sloika" in sys.modules and not getattr(sys.modules["sloika"], "__sloika_amd_alias__", False):
        if not force:
            return sys.modules["sloika"]
    elif "sloika" not in sys.modules and not force:
        if importlib.util.find_spec("sloika") is not None:
            return importlib.import_module("sloika")      # the real package is installed: leave it alone
    pkg = types.ModuleType("sloika")
    pkg.__sloika_amd_alias__ = True
    pkg.__path__ = []
    pkg.__doc__ = "alias of sloika_amd (MI355X-native basecalling path)"
    sys.modules["sloika"] = pkg
    for name in _SUBMODULES:
        mod = importlib.import_module("sloika_amd." + name)
        sys.modules["sloika." + name] = mod
        setattr(pkg, name, mod)
    tools = types.ModuleType("sloika.tools")                   # sloika/tools/chunkify_raw.py
    tools.__path__ = []
    tools.chunkify_raw = importlib.import_module("sloika_amd.chunkify_raw")
    sys.modules["sloika.tools"] = tools
    sys.modules["sloika.tools.chunkify_raw"] = tools.chunkify_raw
    pkg.tools = tools
    return pkg
