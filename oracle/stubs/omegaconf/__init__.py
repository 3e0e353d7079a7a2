"""Minimal stand-in for the `omegaconf` package (NOT installed in this image).

TEST INFRASTRUCTURE ONLY. It exists so that `oracle/gen_golden.py` can import the
unmodified reference modules from /root/reference in the build container and
produce golden vectors. It is never imported by the product package
(`glue_factory_amd` ships its own config object, `conf.py`).
"""
import copy
from contextlib import contextmanager

from . import listconfig  # noqa: F401
from .listconfig import ListConfig


class MissingMandatoryValue(Exception):
    pass


class DictConfig(dict):
    def __init__(self, d=None):
        super().__init__()
        object.__setattr__(self, "_flags", {"struct": False, "readonly": False})
        for k in (d or {}):
            dict.__setitem__(self, k, _wrap(dict.__getitem__(d, k)))

    def items(self):                                   # (raw values: copying / merging must not trip over "???")
        return [(k, dict.__getitem__(self, k)) for k in self]

    def values(self):
        return [dict.__getitem__(self, k) for k in self]

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if isinstance(v, str) and v == "???":          # omegaconf's mandatory-value marker
            raise MissingMandatoryValue(f"Missing mandatory value: {k}")
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __setitem__(self, k, v):
        if self._flags["readonly"]:
            raise TypeError("read-only config")
        if self._flags["struct"] and k not in self:
            raise AttributeError(f"unknown key {k}")
        dict.__setitem__(self, k, _wrap(v))

    def __deepcopy__(self, memo):
        return DictConfig(copy.deepcopy(_unwrap(self)))

    def get(self, k, default=None):
        if k not in self:
            return default
        v = dict.__getitem__(self, k)
        return default if isinstance(v, str) and v == "???" else v


def _wrap(v):
    if isinstance(v, DictConfig):
        return v
    if isinstance(v, dict):
        return DictConfig(v)
    if isinstance(v, (list, tuple)) and not isinstance(v, ListConfig):
        return ListConfig([_wrap(x) for x in v])
    return v


def _unwrap(v):
    if isinstance(v, dict):
        return {k: _unwrap(dict.__getitem__(v, k)) for k in v}
    if isinstance(v, (list, tuple)):
        return [_unwrap(x) for x in v]
    return v


def _merge(a, b):
    out = DictConfig(_unwrap(a))
    for k in b:
        v = dict.__getitem__(b, k)
        if k in out and isinstance(dict.__getitem__(out, k), dict) and isinstance(v, dict):
            dict.__setitem__(out, k, _merge(dict.__getitem__(out, k), v))
        else:
            if a._flags["struct"] and k not in out:
                raise AttributeError(f"unknown key {k}")
            dict.__setitem__(out, k, _wrap(copy.deepcopy(_unwrap(v))))
    return out


def _set_flag(c, name, val):
    if isinstance(c, DictConfig):
        c._flags[name] = val
        for v in c.values():
            _set_flag(v, name, val)


class OmegaConf:
    @staticmethod
    def create(d=None):
        return DictConfig(_unwrap(d) if d is not None else {})

    @staticmethod
    def merge(*confs):
        out = DictConfig()
        for c in confs:
            if c is None:
                continue
            if not isinstance(c, DictConfig):
                c = DictConfig(c)
            struct = out._flags["struct"]
            out = _merge(out, c)
            if struct or c._flags["struct"]:
                pass
        # struct flag propagates from the first struct-flagged config
        for c in confs:
            if isinstance(c, DictConfig) and c._flags["struct"]:
                _set_flag(out, "struct", True)
                break
        return out

    @staticmethod
    def set_struct(c, v):
        _set_flag(c, "struct", v)

    @staticmethod
    def set_readonly(c, v):
        _set_flag(c, "readonly", v)

    @staticmethod
    def to_container(c, resolve=True):
        return _unwrap(c)

    @staticmethod
    def to_yaml(c):
        import yaml

        return yaml.safe_dump(_unwrap(c))

    @staticmethod
    def resolve(c):
        return None

    @staticmethod
    def load(path):
        import yaml

        with open(path) as f:
            return DictConfig(yaml.safe_load(f))

    @staticmethod
    def save(c, path):
        with open(path, "w") as f:
            f.write(OmegaConf.to_yaml(c))

    @staticmethod
    def from_cli(args=None):
        out = {}
        for a in args or []:
            k, v = a.split("=", 1)
            import yaml

            cur = out
            ks = k.split(".")
            for kk in ks[:-1]:
                cur = cur.setdefault(kk, {})
            cur[ks[-1]] = yaml.safe_load(v)
        return DictConfig(out)


@contextmanager
def read_write(c):
    old = c._flags["readonly"] if isinstance(c, DictConfig) else False
    _set_flag(c, "readonly", False)
    try:
        yield c
    finally:
        _set_flag(c, "readonly", old)


@contextmanager
def open_dict(c):
    old = c._flags["struct"] if isinstance(c, DictConfig) else False
    _set_flag(c, "struct", False)
    try:
        yield c
    finally:
        _set_flag(c, "struct", old)
