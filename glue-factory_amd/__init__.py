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

__version__ = "0.1.0"
