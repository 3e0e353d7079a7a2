"""Plugin surface: the contract of gluefactory/models/base_model.py:25-157 and the model
registry of gluefactory/models/__init__.py:7-30, rebuilt on the local ``Conf`` object.

A model declares ``default_conf`` (merged down the class hierarchy), ``required_data_keys``
and implements ``_init(conf)``, ``_forward(data)`` and ``loss(pred, data)``.  When the real
``gluefactory`` package is importable, its ``BaseModel`` is a virtual parent of ours
(``gluefactory.models.get_model`` also accepts a module-level ``__main_model__``, which every
matcher module here exports, so ``model.matcher.name: glue_factory_amd.matchers.lightglue``
is the whole integration).
"""
import importlib
import importlib.util
import inspect
from copy import copy

from torch import nn

from .conf import Conf


class BaseModel(nn.Module):
    default_conf = {
        "name": None,
        "trainable": True,           # if false: parameters get requires_grad=False
        "freeze_batch_normalization": False,  # keep BN layers in eval mode while training
        "timeit": False,
    }
    required_data_keys = []
    strict_conf = False
    are_weights_initialized = False

    @classmethod
    def merged_default_conf(cls):
        """default_conf of every class on the MRO, base classes first (MetaModel semantics)."""
        total = Conf()
        for klass in reversed(cls.__mro__):
            dc = klass.__dict__.get("default_conf")
            if dc is not None:
                total = Conf.merge(total, dc)
        return total

    def __init__(self, conf=None):
        super().__init__()
        default = self.merged_default_conf()
        if self.strict_conf:
            default.set_struct(True)
        conf = Conf.create(conf or {})
        if "pad" in conf and "pad" not in default:  # backward compatibility of old yaml files
            conf["interpolation"] = {"pad": conf.pop("pad")}
        self.conf = conf = Conf.merge(default, conf)
        conf.set_readonly(True)
        conf.set_struct(True)
        self.required_data_keys = copy(self.required_data_keys)
        self._init(conf)
        if not conf.trainable:
            for p in self.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        if self.conf.freeze_batch_normalization:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def forward(self, data):
        def check(expected, given):
            for key in expected:
                assert key in given, f"Missing key {key} in data"
                if isinstance(expected, dict):
                    check(expected[key], given[key])

        check(self.required_data_keys, data)
        return self._forward(data)

    def _init(self, conf):
        raise NotImplementedError

    def _forward(self, data):
        raise NotImplementedError

    def loss(self, pred, data):
        raise NotImplementedError

    def load_state_dict(self, *args, **kwargs):
        ret = super().load_state_dict(*args, **kwargs)
        self.set_initialized()
        return ret

    def is_initialized(self):
        ok = True
        for _, child in self.named_children():
            if isinstance(child, BaseModel):
                ok = ok and child.is_initialized()
            else:
                n_params = len(list(child.parameters()))
                ok = ok and (n_params == 0 or self.are_weights_initialized)
        return ok

    def set_initialized(self, to=True):
        self.are_weights_initialized = to
        for child in self.children():
            if isinstance(child, BaseModel):
                child.set_initialized(to)


def _model_from_module(path):
    mod = importlib.import_module(path)
    classes = [c for _, c in inspect.getmembers(mod, inspect.isclass)
               if c.__module__ == path and issubclass(c, BaseModel)]
    if len(classes) == 1:
        return classes[0]
    if hasattr(mod, "__main_model__"):
        return mod.__main_model__
    raise AttributeError(f"{path}: no unique BaseModel subclass and no __main_model__")


def get_model(name):
    """Resolve a model class from a dotted name: absolute module path first, then the
    package-relative spellings glue-factory yaml files use (``matchers.lightglue``, ...)."""
    pkg = __name__.rsplit(".", 1)[0]
    short = name.split(".")[-1]
    candidates = [name, f"{pkg}.{name}", f"{pkg}.matchers.{short}", f"{pkg}.{short}"]
    tried = []
    for path in candidates:
        try:
            spec = importlib.util.find_spec(path)
        except (ModuleNotFoundError, ValueError):
            spec = None
        tried.append(path)
        if spec is None:
            continue
        try:
            return _model_from_module(path)
        except AttributeError:
            continue
    raise RuntimeError(f"Model {name} not found in any of [{' '.join(tried)}]")


class BatchedExtractionUnsupported(ValueError):
    """Raised by an extractor that cannot serve THIS batch in one call (images with different keypoint counts, no keypoint
    cap): pipeline.TwoViewPipeline falls back to one call per view on exactly this exception -- any other error of an
    extractor propagates.  (A ValueError subclass: callers that caught ValueError keep working.)"""
