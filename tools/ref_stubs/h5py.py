"""Import-time stand-in for h5py (no HDF5 I/O is exercised when generating goldens)."""


class File:  # pragma: no cover
    def __init__(self, *a, **k):
        raise RuntimeError("h5py is not installed in this container")
