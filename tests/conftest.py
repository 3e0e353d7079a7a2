import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


PROBE_SRC = os.path.join(ROOT, "tests", "csrc", "gf_test_probe.hip")
PROBE_LIB = os.path.join(ROOT, "tests", "libgf_test_probe.so")
_probe = None


def build_test_probe():
    """tests/libgf_test_probe.so: TEST-ONLY kernels (a CU holder) that have no place in the product ABI.  Built in-tree by
    __graft_entry__.build() (so it travels to the GPU box) and on demand here."""
    import subprocess
    if not os.path.exists(PROBE_LIB) or os.path.getmtime(PROBE_LIB) < os.path.getmtime(PROBE_SRC):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", PROBE_SRC,
                        "-o", PROBE_LIB], check=True, capture_output=True)
    return PROBE_LIB


def test_probe():
    global _probe
    if _probe is None:
        import ctypes
        _probe = ctypes.CDLL(build_test_probe())
        _probe.gf_test_hold_cus.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _probe.gf_test_hold_cus.restype = ctypes.c_int
        _probe.gf_test_dirty_lds.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
        _probe.gf_test_dirty_lds.restype = ctypes.c_int
    return _probe


test_probe.__test__ = False          # (a helper, not a test: keep pytest from collecting it where it is imported)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def golden_data(z, device="cpu", dtype=torch.float32):
    """Rebuild the matcher input dict from a golden file's ``data.*`` entries."""
    def t(k):
        a = torch.from_numpy(z["data." + k])
        return a.to(device=device, dtype=dtype) if a.is_floating_point() else a.to(device)
    data = {k: t(k) for k in ("keypoints0", "keypoints1", "descriptors0", "descriptors1",
                              "gt_assignment", "gt_matches0", "gt_matches1")}
    data["view0"] = {"image_size": t("image_size0")}
    data["view1"] = {"image_size": t("image_size1")}
    data["image_size0"] = data["view0"]["image_size"]
    data["image_size1"] = data["view1"]["image_size"]
    return data


@pytest.fixture(scope="session")
def golden():
    return load_golden
