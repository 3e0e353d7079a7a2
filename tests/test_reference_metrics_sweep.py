"""matcher_metrics (gluefactory/models/utils/metrics.py) side by side with the reference over random predictions -- matched,
unmatched (-1) and ignored (-2) ground truth, empty rows, tied scores, the `line_` prefixes GlueStick uses (build container
only: skipped where /root/reference is absent)."""
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


@pytest.mark.parametrize("seed,batch,m,n", [(1, 3, 40, 50), (2, 1, 7, 3), (3, 4, 128, 128), (4, 2, 1, 1)])
@pytest.mark.parametrize("prefix,prefix_gt", [("", None), ("line_", None), ("line_0_", "line_")])
def test_matcher_metrics_equal_the_reference(ref_path, seed, batch, m, n, prefix, prefix_gt):
    from gluefactory.models.utils.metrics import matcher_metrics as ref_fn
    from glue_factory_amd.metrics import matcher_metrics
    g = torch.Generator().manual_seed(seed)
    gt = torch.randint(-2, n, (batch, m), generator=g)
    pred_m = torch.where(torch.rand(batch, m, generator=g) < 0.6, gt, torch.randint(-1, n, (batch, m), generator=g))
    scores = torch.rand(batch, m, generator=g)
    scores = torch.where(torch.rand(batch, m, generator=g) < 0.2, torch.full_like(scores, 0.5), scores)      # ties
    scores = torch.where(pred_m > -1, scores, torch.zeros_like(scores))
    if batch > 1:
        gt[0] = -2                     # a pair without any usable ground truth
        pred_m[-1] = -1                # a pair without any prediction
    pred = {f"{prefix}matches0": pred_m, f"{prefix}matching_scores0": scores}
    data = {f"gt_{prefix if prefix_gt is None else prefix_gt}matches0": gt}
    ref = ref_fn(pred, data, prefix=prefix, prefix_gt=prefix_gt)
    out = matcher_metrics(pred, data, prefix=prefix, prefix_gt=prefix_gt)
    assert set(out) == set(ref)
    for k in ref:
        torch.testing.assert_close(out[k], ref[k], rtol=1e-6, atol=1e-7, msg=lambda s: f"{k}: {s}")
