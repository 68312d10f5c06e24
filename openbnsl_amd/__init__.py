"""MI355X-native local-score engine for OpenBNSL.

* ``openbnsl_amd.openbnsllib`` -- drop-in for the reference's ``openbnsllib``
  Python module (local-score path only).
* ``openbnsl_amd.engine``      -- ctypes view of the C ABI (include/obnsl_engine.h).

One process must hold ONE HIP runtime.  PyTorch's ROCm wheels bundle their own
copy of libamdhip64/libhsa-runtime64; if the engine library pulled in the system
copy first, a later ``import torch`` would bring up a second runtime that finds
no GPU.  So when PyTorch is installed it is imported first: the engine library
(which asks for libamdhip64.so.7 by soname) then binds to the copy PyTorch
loaded.  Set OPENBNSL_NO_TORCH=1 to skip this (a process that never uses torch).
"""
import os as _os

if not _os.environ.get("OPENBNSL_NO_TORCH"):
    try:
        import torch as _torch  # noqa: F401
    except ImportError:  # torch is optional: the engine itself only needs HIP
        pass
