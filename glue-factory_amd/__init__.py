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
# MIOpen solution selection for the stock convolutions of the extractor, unless the user already chose one: find mode
# 3 ("fast find": pick from the find-db / heuristics immediately) instead of the default hybrid search.  Measured on
# MI355X for SuperPoint-open (tools/probe/sp_variants.py, 64 x 1024^2, bf16): first call 62 s -> 0.3 s and steady state
# 45.8 -> 38.1 ms per forward (the default search settles on split-K kernels that need a zero-filled output).  Set
# here, at package import, so that it is in the environment before the library serves its first convolution.
_os.environ.setdefault("MIOPEN_FIND_MODE", "3")

__version__ = "0.1.0"
