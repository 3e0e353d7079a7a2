"""No-grad matcher metrics on the [B,N] match vectors (eval mode only).

Restates gluefactory/models/utils/metrics.py:4-50 (recall, precision, accuracy and the
ranking average precision of matches0 against gt_matches0); tiny tensors, stock torch ops."""
import torch


@torch.no_grad()
def matcher_metrics(pred, data, prefix="", prefix_gt=None):
    prefix_gt = prefix if prefix_gt is None else prefix_gt
    m = pred[f"{prefix}matches0"]
    gt = data[f"gt_{prefix_gt}matches0"]
    scores = pred[f"{prefix}matching_scores0"]
    hit = (m == gt)

    def ratio(mask):
        mask = mask.float()
        return (hit * mask).sum(1) / (1e-8 + mask.sum(1))

    has_gt = gt > -1
    labelled = gt >= -1
    predicted = (m > -1) & labelled
    order = torch.argsort(-scores)
    p_mask = predicted.float().gather(-1, order)
    r_mask = has_gt.float().gather(-1, order)
    tp = hit.gather(-1, order)
    p_pts = torch.cumsum(tp * p_mask, -1) / (1e-8 + torch.cumsum(p_mask, -1))
    r_pts = torch.cumsum(tp * r_mask, -1) / (1e-8 + r_mask.sum(-1)[:, None])
    r_diff = r_pts[..., 1:] - r_pts[..., :-1]
    ap = torch.sum(r_diff * p_pts[:, None, -1], dim=-1)
    return {
        f"{prefix}match_recall": ratio(has_gt),
        f"{prefix}match_precision": ratio(predicted),
        f"{prefix}accuracy": ratio(labelled),
        f"{prefix}average_precision": ap,
    }
