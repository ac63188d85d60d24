"""Import alias: `import lsq_amd` == the package directory `local-search-quantization_amd/`
(whose name, fixed by the project layout, is not a valid Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("local-search-quantization_amd")
sys.modules[__name__] = _pkg
