"""Cached local features -> batch entries, the reference's recommended way to train the matcher without
running the extractor (gluefactory/models/cache_loader.py:13-141, README "feature export"), behind the same
plugin surface: ``two_view_pipeline`` puts a CacheLoader in the ``extractor`` slot (``allow_no_extract``).

Differences in HOW: the reference stores one HDF5 group per image (h5py is not available on this target);
here a cache is a directory (or format string) of ``<name>.npz`` files, one per image, written by
``export_features``.  A ``path`` that names a FILE is read as the reference's HDF5 export (one group per image name,
nested groups as nested dicts, cache_loader.py:47-56, 98-105) when ``h5py`` is importable, so existing feature exports
keep working where that package exists; without it the error says so.  Same keys, same scaling of ``keypoints*`` / ``lines*`` by the batch's ``scales``, same
padding contract (``padding_fn`` + ``padding_length``: keypoints padded uniformly inside their bounding box,
descriptors uniformly inside their value range, scores / scales / oris / depth with zeros), same collation.
"""
import os
import string

import numpy as np
import torch

from .base_model import BaseModel


def pad_to_length(x, length, pad_dim=-2, mode="zeros", bounds=(None, None)):
    """gluefactory/models/utils/misc.py:20-60: append ``length - d`` entries along ``pad_dim``."""
    shape = list(x.shape)
    d = x.shape[pad_dim]
    assert d <= length
    if d == length:
        return x
    shape[pad_dim] = length - d
    low, high = bounds
    if mode == "zeros":
        xn = torch.zeros(*shape, device=x.device, dtype=x.dtype)
    elif mode == "ones":
        xn = torch.ones(*shape, device=x.device, dtype=x.dtype)
    elif mode == "random":
        low = low if low is not None else x.min()
        high = high if high is not None else x.max()
        xn = torch.empty(*shape, device=x.device).uniform_(float(low), float(high))
    elif mode == "random_c":        # per last-dim channel, uniform inside the existing entries' range
        xn = torch.cat([torch.empty(*shape[:-1], 1, device=x.device).uniform_(
            float(x[..., i].min()) if d > 0 else low, float(x[..., i].max()) if d > 0 else high)
            for i in range(shape[-1])], dim=-1)
    else:
        raise ValueError(mode)
    return torch.cat([x, xn.to(x.dtype)], dim=pad_dim)


_PAD_RULES = (("keypoints", -2, "random_c"), ("keypoint_scores", -1, "zeros"), ("descriptors", -2, "random"),
              ("scales", -1, "zeros"), ("oris", -1, "zeros"), ("depth_keypoints", -1, "zeros"),
              ("valid_depth_keypoints", -1, "zeros"))


def pad_local_features(pred, seq_l):
    """cache_loader.py:13-41: bring every per-keypoint entry to ``seq_l`` keypoints."""
    for key, dim, mode in _PAD_RULES:
        if key in pred:
            pred[key] = pad_to_length(pred[key], seq_l, dim, mode=mode)
    return pred


def export_features(path, name, pred):
    """Write one image's extractor outputs (tensors without the batch dimension) as ``<path>/<name>.npz``."""
    file = os.path.join(path, f"{name}.npz")
    os.makedirs(os.path.dirname(file), exist_ok=True)       # image names usually carry a scene directory
    np.savez(file, **{k: v.detach().cpu().numpy() for k, v in pred.items()
      if torch.is_tensor(v)})


def _collate(preds):
    out = {}
    for k in preds[0]:
        out[k] = torch.stack([p[k] for p in preds], 0)
    return out


def _load_hdf5_group(path, name, keys):
    """One image's group of a reference feature export (gluefactory/models/cache_loader.py:47-56 recursive_load)."""
    try:
        import h5py
    except ImportError as e:
        raise RuntimeError(f"{path} is a file, i.e. an HDF5 feature export of the reference; reading it needs the h5py package, "
                           "which is not installed here.  Point `path` at a directory of <name>.npz files written by "
                           "glue_factory_amd.cache_loader.export_features instead.") from e

    def load(grp, ks):
        return {k: torch.from_numpy(grp[k].__array__()) if isinstance(grp[k], h5py.Dataset) else load(grp[k], list(grp[k].keys()))
                for k in ks}

    with h5py.File(str(path), "r") as f:
        grp = f[name]
        return load(grp, keys if keys is not None else list(grp.keys()))


class CacheLoader(BaseModel):
    default_conf = {
        "path": "???",             # directory of .npz files (or an HDF5 file, needs h5py); may be a format string like exports/{scene}/
        "data_keys": None,         # load all keys
        "device": None,            # load to the same device as the batch
        "trainable": False,
        "add_data_path": True,     # accepted for yaml compatibility (paths are used as given)
        "collate": True,
        "scale": ["keypoints", "lines", "orig_lines"],
        "padding_fn": None,
        "padding_length": None,    # required for batching
        "numeric_type": "float32",
    }
    required_data_keys = ["name"]
    _PADDING = {"pad_local_features": pad_local_features}

    def _init(self, conf):
        self.padding_fn = None
        if conf.padding_fn is not None:
            if conf.padding_fn not in self._PADDING:
                raise ValueError(f"unknown padding_fn {conf.padding_fn!r} (known: {sorted(self._PADDING)})")
            self.padding_fn = self._PADDING[conf.padding_fn]
        self.numeric_dtype = {None: None, "float16": torch.float16, "float32": torch.float32,
                              "float64": torch.float64}[conf.numeric_type]

    def _forward(self, data):
        device = self.conf.device
        if not device:
            devices = {v.device for v in data.values() if isinstance(v, torch.Tensor)}
            assert len(devices) <= 1
            device = devices.pop() if devices else "cpu"
        var_names = [x[1] for x in string.Formatter().parse(self.conf.path) if x[1]]
        preds = []
        for i, name in enumerate(data["name"]):
            root = self.conf.path.format(**{k: data[k][i] for k in var_names})
            if os.path.isfile(root):
                pred = _load_hdf5_group(root, name, self.conf.data_keys)
            else:
                with np.load(os.path.join(root, f"{name}.npz")) as z:
                    keys = self.conf.data_keys if self.conf.data_keys is not None else list(z.keys())
                    pred = {k: torch.from_numpy(z[k]) for k in keys}
            if self.numeric_dtype is not None:
                pred = {k: v.to(self.numeric_dtype) if torch.is_floating_point(v) else v for k, v in pred.items()}
            pred = {k: v.to(device) for k, v in pred.items()}
            for k in list(pred):
                for pattern in self.conf.scale:
                    if k.startswith(pattern):
                        view_idx = k.replace(pattern, "")
                        scales = data["scales"] if len(view_idx) == 0 else data[f"view{view_idx}"]["scales"]
                        pred[k] = pred[k] * scales[i].to(pred[k])
            if self.padding_fn is not None:
                pred = self.padding_fn(pred, self.conf.padding_length)
            preds.append(pred)
        if self.conf.collate:
            return _collate(preds)
        assert len(preds) == 1
        return preds[0]

    def loss(self, pred, data):
        raise NotImplementedError
