"""matcher_metrics (glue_factory_amd/metrics.py) against values computed by the reference's own
gluefactory/models/utils/metrics.py:4-50 (tests/golden/metrics.npz, oracle/gen_golden.py gen_metrics)."""
import numpy as np
import torch

from conftest import load_golden


def test_matcher_metrics_equal_reference_values():
    from glue_factory_amd.metrics import matcher_metrics
    z = load_golden("metrics")
    pred = {"matches0": torch.from_numpy(z["matches0"]), "matching_scores0": torch.from_numpy(z["matching_scores0"])}
    data = {"gt_matches0": torch.from_numpy(z["gt_matches0"])}
    out = matcher_metrics(pred, data)
    assert set(out) == {"match_recall", "match_precision", "accuracy", "average_precision"}
    for k, v in out.items():
        np.testing.assert_allclose(v.numpy(), z["metric." + k], rtol=1e-6, atol=1e-7, err_msg=k)
    assert (z["metric.average_precision"] > 0.1).all() and (z["metric.match_recall"] < 0.9).all()   # non-trivial case
    # prefixed form used by the GlueStick line head
    pl = {"line_matches0": pred["matches0"], "line_matching_scores0": pred["matching_scores0"]}
    dl = {"gt_line_matches0": data["gt_matches0"]}
    out_l = matcher_metrics(pl, dl, prefix="line_")
    for k, v in out.items():
        torch.testing.assert_close(out_l["line_" + k], v)
