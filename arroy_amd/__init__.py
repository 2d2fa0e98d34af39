"""arroy_amd — MI355X-native implementation of arroy's distance-kernel hot path.

Layout
  csrc/            hand-written HIP (gfx950) kernels + the C ABI (include/arroy_hip.h) -> libarroy_hip.so
  _lib.py          ctypes binding of that ABI (no fallback: raises if the .so is missing)
  dataset.py       Dataset / Forest: numpy-facing handles (one method = one C-ABI call)
  distances.py     arroy::distances marker types
  index.py         host-side mirror of arroy's Writer / ArroyBuilder / Reader / QueryBuilder surface
"""
from . import distances
from ._lib import ArroyHipError, BuildCancelled, InvalidVecDimension, MissingKey, device_count, device_name
from .dataset import Dataset, Forest, Index

__all__ = ["distances", "Dataset", "Forest", "Index", "ArroyHipError", "BuildCancelled", "InvalidVecDimension", "MissingKey",
           "device_count", "device_name"]
