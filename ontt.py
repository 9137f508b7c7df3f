"""Loader for the hyphen-named package directory.

`optimized-number-theoretic-transform-implementations_amd/` (the layout's fixed
name) is not a valid Python identifier, so `import` cannot spell it; this helper
loads it through importlib.  Usage:  `import ontt; lib = ontt.load()`.
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "optimized-number-theoretic-transform-implementations_amd")
_MOD = "ontt_amd"


def load():
    if _MOD in sys.modules:
        return sys.modules[_MOD]
    spec = importlib.util.spec_from_file_location(_MOD, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_MOD] = mod
    try:
        spec.loader.exec_module(mod)
    except Exception:
        del sys.modules[_MOD]
        raise
    return mod
