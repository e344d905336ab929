"""Import shim: `import tspn_mi355x` -> the package in
`temporal-span-proposal-network-vidvrd_amd/` (whose directory name cannot be
written in an `import` statement).  Sub-modules are aliased too, so
`from tspn_mi355x.model import BaseModel` works and yields the same objects.
"""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_REAL = "temporal-span-proposal-network-vidvrd_amd"
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + "."):
        sys.modules[__name__ + _name[len(_REAL):]] = _mod
sys.modules[__name__] = _pkg
