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
