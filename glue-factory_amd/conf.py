"""Small hierarchical config object with the slice of OmegaConf semantics that the
glue-factory plugin surface relies on (gluefactory/models/base_model.py:65-90):
attribute + item access, recursive merge, struct mode (unknown keys raise), read-only mode,
conversion back to plain containers, yaml / dot-list loading.

If the real ``omegaconf`` is importable (a glue-factory environment) its DictConfig objects
are accepted everywhere a mapping is: ``Conf.create`` converts them to plain containers.
"""
import copy
from collections.abc import Mapping


class ConfError(AttributeError, KeyError):
    pass


def _plain(x):
    """Any mapping / sequence flavour (dict, Conf, omegaconf DictConfig/ListConfig) -> builtins."""
    if isinstance(x, Conf):
        return x.to_container()
    if type(x).__module__.startswith("omegaconf"):
        from omegaconf import OmegaConf  # pragma: no cover - only in a glue-factory env
        return OmegaConf.to_container(x, resolve=True)
    if isinstance(x, Mapping):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


class Conf(Mapping):
    __slots__ = ("_d", "_struct", "_ro")

    def __init__(self, data=None):
        object.__setattr__(self, "_d", {})
        object.__setattr__(self, "_struct", False)
        object.__setattr__(self, "_ro", False)
        for k, v in (_plain(data) or {}).items():
            self._d[k] = Conf(v) if isinstance(v, Mapping) else copy.deepcopy(v)

    # ---- construction helpers (OmegaConf-like static API)
    @staticmethod
    def create(data=None):
        return Conf(data)

    @staticmethod
    def merge(*confs):
        out = Conf()
        struct = False
        for c in confs:
            if c is None:
                continue
            c = c if isinstance(c, Conf) else Conf(c)
            out._merge_from(c, struct)
            struct = struct or c._struct
        if struct:
            out.set_struct(True)
        return out

    def _merge_from(self, other, struct):
        for k, v in other._d.items():
            if struct and k not in self._d:
                raise ConfError(f"Key '{k}' is not in the struct config")
            if isinstance(v, Conf) and isinstance(self._d.get(k), Conf):
                self._d[k]._merge_from(v, struct)
            else:
                self._d[k] = Conf(v) if isinstance(v, Conf) else copy.deepcopy(v)

    @staticmethod
    def from_yaml(path):
        import yaml
        with open(path) as f:
            return Conf(yaml.safe_load(f) or {})

    @staticmethod
    def from_dotlist(items):
        import yaml
        root = {}
        for it in items:
            key, val = it.split("=", 1)
            cur = root
            parts = key.split(".")
            for p in parts[:-1]:
                cur = cur.setdefault(p, {})
            cur[parts[-1]] = yaml.safe_load(val)
        return Conf(root)

    # ---- flags
    def set_struct(self, flag=True):
        object.__setattr__(self, "_struct", flag)
        for v in self._d.values():
            if isinstance(v, Conf):
                v.set_struct(flag)
        return self

    def set_readonly(self, flag=True):
        object.__setattr__(self, "_ro", flag)
        for v in self._d.values():
            if isinstance(v, Conf):
                v.set_readonly(flag)
        return self

    # ---- access
    def __getitem__(self, k):
        try:
            return self._d[k]
        except KeyError:
            raise ConfError(f"Missing key '{k}'") from None

    def __getattr__(self, k):
        if k.startswith("_"):
            raise AttributeError(k)
        return self[k]

    def __setitem__(self, k, v):
        if self._ro:
            raise ConfError("config is read-only")
        if self._struct and k not in self._d:
            raise ConfError(f"Key '{k}' is not in the struct config")
        self._d[k] = Conf(v) if isinstance(v, Mapping) else v

    def __setattr__(self, k, v):
        self[k] = v

    def __iter__(self):
        return iter(self._d)

    def __len__(self):
        return len(self._d)

    def __contains__(self, k):
        return k in self._d

    def get(self, k, default=None):
        return self._d.get(k, default)

    def pop(self, k, *default):
        if self._ro:
            raise ConfError("config is read-only")
        return self._d.pop(k, *default)

    def to_container(self):
        return {k: (v.to_container() if isinstance(v, Conf) else copy.deepcopy(v))
                for k, v in self._d.items()}

    def __deepcopy__(self, memo):
        c = Conf(self.to_container())
        c.set_struct(self._struct)
        c.set_readonly(self._ro)
        return c

    def __repr__(self):
        return f"Conf({self.to_container()!r})"


to_container = _plain
