"""Import-time stand-in for `nptyping` (annotations only) -- used ONLY by tools/gen_golden.py."""
import numpy as _np


class _NDArrayMeta(type):
    def __getitem__(cls, item):
        return _np.ndarray


class NDArray(metaclass=_NDArrayMeta):
    pass
