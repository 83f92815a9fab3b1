"""Importable alias for the package directory `pim-embedding-lookup_amd/` (a hyphen cannot appear
in an `import` statement): `import pim_embedding_lookup_amd as pel`."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("pim-embedding-lookup_amd")
sys.modules[__name__] = _pkg
