"""h5py stand-in for the reference-side tests (oracle/ is test infrastructure; h5py is not installable here).

Only what gluefactory/models/cache_loader.py touches: ``File(path, "r")[name]`` -> a group whose ``keys()`` /
``__getitem__`` yield ``Dataset`` objects with ``__array__()``.  The "file" is a DIRECTORY of per-image ``<name>.npz``
archives -- the on-disk format of glue_factory_amd.cache_loader -- so the reference CacheLoader can be run on exactly
the bytes ours reads (``path`` may also name a file inside that directory, e.g. ``<dir>/features.h5``).
"""
import os

import numpy as np


class Dataset:
    def __init__(self, array):
        self._a = array

    def __array__(self, dtype=None):
        return self._a if dtype is None else self._a.astype(dtype)

    @property
    def shape(self):
        return self._a.shape


class Group:
    def __init__(self, arrays):
        self._d = {k: Dataset(v) for k, v in arrays.items()}

    def keys(self):
        return self._d.keys()

    def __getitem__(self, k):
        return self._d[k]

    def __contains__(self, k):
        return k in self._d


class File:
    def __init__(self, path, mode="r"):
        assert mode == "r", "the stand-in is read-only"
        self._root = path if os.path.isdir(path) else os.path.dirname(path)

    def __getitem__(self, name):
        with np.load(os.path.join(self._root, f"{name}.npz")) as z:
            return Group({k: z[k] for k in z.files})

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
