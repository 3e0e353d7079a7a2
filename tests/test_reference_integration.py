"""The drop-in boundary seen FROM THE REFERENCE SIDE: the reference's own ``get_model`` and
``TwoViewPipeline`` (gluefactory/models/__init__.py:7-30, two_view_pipeline.py:21-114) resolve and construct
``glue_factory_amd.matchers.{lightglue,superglue,gluestick}`` from a yaml-style config, and a state_dict of the
reference's matcher round-trips through ours.  Build-container only: skipped where /root/reference is absent
(the GPU box).  No kernels run here (construction + state_dict only)."""
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gluefactory")),
                                reason="reference checkout not present (GPU box)")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ref_path():
    stubs = os.path.join(ROOT, "oracle", "stubs")
    added = [p for p in (stubs, REF) if p not in sys.path]
    sys.path[:0] = [stubs]
    sys.path.append(REF)
    yield
    for p in added:
        if p in sys.path:
            sys.path.remove(p)


def _ref_matcher(name):
    if name == "lightglue":
        from gluefactory.models.matchers.lightglue import LightGlue
        return LightGlue({"weights": None, "n_layers": 3, "flash": False})
    if name == "superglue":
        from gluefactory_nonfree.superglue import SuperGlue
        return SuperGlue({"weights": None, "GNN_layers": ["self", "cross"] * 2})
    from gluefactory.models.matchers.gluestick import GlueStick
    return GlueStick({"weights": None, "GNN_layers": ["self", "cross"] * 2, "num_line_iterations": 1})


CONF = {"lightglue": {"n_layers": 3}, "superglue": {"GNN_layers": ["self", "cross"] * 2, "weights": None},
        "gluestick": {"GNN_layers": ["self", "cross"] * 2, "num_line_iterations": 1, "weights": None}}


@pytest.mark.parametrize("name", ["lightglue", "superglue", "gluestick"])
def test_reference_get_model_and_pipeline_construct_the_plugin(ref_path, name):
    from omegaconf import OmegaConf
    from gluefactory.models import get_model
    from gluefactory.models.two_view_pipeline import TwoViewPipeline
    import glue_factory_amd.matchers as ours_pkg
    modname = f"glue_factory_amd.matchers.{name}"
    cls = get_model(modname)                                  # absolute module path / __main_model__ discovery
    assert cls.__module__.startswith(ours_pkg.__name__.split(".")[0]) or "glue" in cls.__module__
    pipe = TwoViewPipeline(OmegaConf.create({"matcher": {"name": modname, **CONF[name]},
                                             "extractor": {"name": None}, "allow_no_extract": True}))
    matcher = pipe.matcher
    assert type(matcher).__name__ == cls.__name__
    # the reference's state_dict loads strictly into ours and comes back identical
    ref = _ref_matcher(name)
    sd = ref.state_dict()
    res = matcher.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    back = matcher.state_dict()
    assert set(back) == set(sd)
    for k, v in sd.items():
        assert back[k].shape == v.shape and torch.equal(back[k].cpu(), v), k
    # optimizer-facing surface: same trainable parameter names
    assert [k for k, _ in matcher.named_parameters()] == [k for k, _ in ref.named_parameters()]
    # and the reverse direction: ours -> reference
    res = ref.load_state_dict(matcher.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys


def test_build_ref_closure_is_static_and_complete():
    """oracle/build_ref.py finds the reference's LightGlue module closure by READING import statements (the build imports
    and executes no reference module); the byte-compiled tree it produces is importable and complete."""
    import importlib.util
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/gluefactory"):
        pytest.skip("reference not present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import build_ref
    before = set(sys.modules)
    files = build_ref._closure(build_ref.TARGETS)
    assert not [m for m in set(sys.modules) - before if m.split(".")[0] in ("gluefactory", "gluefactory_nonfree")]
    rel = {os.path.relpath(f, "/root/reference") for f in files}
    assert "gluefactory/models/matchers/lightglue.py" in rel and "gluefactory/models/utils/losses.py" in rel
    assert not any(r.startswith("gluefactory_nonfree") for r in rel)
    # a fresh interpreter imports the reference's LightGlue from the byte-compiled tree alone
    if importlib.util.find_spec("torch") is None or not os.path.exists(os.path.join(root, "oracle", "_ref", "MANIFEST.txt")):
        pytest.skip("oracle/_ref not built")
    code = ("import sys; sys.path.insert(0, %r); from oracle import build_ref; assert build_ref.import_reference(); "
            "import gluefactory.models.matchers.lightglue as m; assert m.__file__.endswith('.pyc'); print(m.LightGlue.__name__)" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "LightGlue" in out.stdout, out.stderr[-2000:]
