"""Import shim: the package sources live in ``glue-factory_amd/`` (a directory name
Python cannot import directly); this module makes them importable as
``glue_factory_amd`` (e.g. ``model.matcher.name: glue_factory_amd.matchers.lightglue``)."""
import os as _os

_src = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                     "glue-factory_amd")
__path__.insert(0, _src)
with open(_os.path.join(_src, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_src, "__init__.py"), "exec"))
del _os, _f, _src
