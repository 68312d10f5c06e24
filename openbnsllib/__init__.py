"""``import openbnsllib`` -- the reference's module name
(reference: CMakeLists.txt:85-87, bindings/pybind_module.cpp:13) for the
MI355X-native local-score engine.  Everything here is the pybind11 module
``openbnsl_amd.openbnsllib``; the submodules are registered under this name so
that ``from openbnsllib.base import PDAG`` (as the reference's
helpers/pgmpy_bridge.py:2 does) resolves."""
import sys as _sys

import openbnsl_amd as _pkg  # noqa: F401  (brings up the one HIP runtime of the process first)
from openbnsl_amd import openbnsllib as _impl

for _name in ("base", "score", "structure_learning"):
    _sub = getattr(_impl, _name)
    globals()[_name] = _sub
    _sys.modules[__name__ + "." + _name] = _sub
for _name in dir(_impl):
    if not _name.startswith("__") and _name not in globals():
        globals()[_name] = getattr(_impl, _name)
del _name, _sub
