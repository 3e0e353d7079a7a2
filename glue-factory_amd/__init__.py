"""MI355X-native matcher train-step hot path behind glue-factory's plugin surface.

Imported as ``glue_factory_amd`` (see the shim next to this directory).  Sub-modules:
  lib        ctypes loader of the C-ABI HIP library (include/gf_amd.h); fails loudly
  ops        torch.autograd.Function wrappers around the HIP launchers
  conf       small OmegaConf-compatible config object
  base_model BaseModel / get_model plugin surface (gluefactory/models/base_model.py)
  matchers   lightglue / superglue / gluestick
  pipeline   TwoViewPipeline-compatible composition
  gt         ground-truth assignment from homographies
  synthetic  seeded synthetic keypoint-pair generator
"""
import os as _os

# Let the stock convolution library take channels-last activations as they are (the fused extractor path keeps
# them NHWC end to end); PyTorch-ROCm reads this once, at its first convolution call.
_os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
# MIOpen solution selection for the stock convolutions of the extractor -- the same idea as the TunableOp table of the
# library GEMMs: miopen_db/ holds the user find-db / perf-db written by ONE exhaustive search on MI355X
# (torch.backends.cudnn.benchmark on SuperPoint-open, 64 x 1024^2, bf16, channels-last: tools/probe/sp_variants.py
# benchmark with MIOPEN_USER_DB_PATH set).  Replaying it (find mode 3: hybrid find, the find-db entry is taken when present) gives
# first call 0.3 s instead of 62 s and 37.9 instead of 45.8 ms per forward: the search prefers the composable-kernel
# implicit GEMM over the default ASM one, which needs a zero-filled output.  Shapes that are not in the table fall back
# to MIOpen's heuristics.  Both variables are left alone when the user set them; the table is copied to a private
# directory because the library writes to its user db.
def _miopen_db():
    """Install the shipped find-db into a private per-process-group directory; returns True when it is in place.
    The directory name carries a hash of the table (a package update never reads a stale copy) and every file is
    written to a temporary name and renamed (concurrent ranks never see a partial table)."""
    import hashlib as _hl
    import shutil as _sh
    import tempfile as _tf
    src = _os.path.join(__path__[0], "miopen_db")      # __path__[0] = the source directory (see the import shim)
    if "MIOPEN_USER_DB_PATH" in _os.environ or not _os.path.isdir(src):
        return False
    names = sorted(n for n in _os.listdir(src) if n.endswith(".txt"))
    if not names:
        return False
    try:
        h = _hl.sha1()
        for name in names:
            with open(_os.path.join(src, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
        dst = _os.path.join(_tf.gettempdir(), "gf_amd_miopen_db_%d_%s_%s"
                            % (_os.getuid(), h.hexdigest()[:12], _os.environ.get("LOCAL_RANK", "0")))
        _os.makedirs(dst, exist_ok=True)
        for name in names:
            final = _os.path.join(dst, name)
            if not _os.path.exists(final):
                tmp = "%s.%d.tmp" % (final, _os.getpid())
                _sh.copyfile(_os.path.join(src, name), tmp)
                _os.replace(tmp, final)
        _os.environ["MIOPEN_USER_DB_PATH"] = dst
        return True
    except OSError:
        return False


# find mode 3 ("hybrid": use the find-db entry when there is one, otherwise MIOpen's usual heuristics + a light search) is
# only selected together with the private table -- a user-supplied db path or a missing table leaves MIOpen's defaults.
if _miopen_db():
    _os.environ.setdefault("MIOPEN_FIND_MODE", "3")

__version__ = "0.1.0"
