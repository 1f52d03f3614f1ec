"""Import alias: the package directory is named `genomic-resistance-mapping-grm-_amd`
(not a Python identifier), so `import grm_amd` resolves to that one package object."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("genomic-resistance-mapping-grm-_amd")
sys.modules[__name__] = _pkg
