"""Two-view matching on image triplets (drop-in for gluefactory/models/triplet_pipeline.py:23-99).

When the batch carries a third view (``view2``), the extractor runs once per image and the matcher / filter / solver
run on the three pairs 0-1, 0-2, 1-2: stacked on the batch axis into ONE call (``batch_triplets``, the default -- three
times the pairs per launch suits the HIP matchers) or pair by pair; outputs come back under ``pred["0to1"]``,
``pred["0to2"]``, ``pred["1to2"]``, losses are computed on the stacked batch (or summed over the pairs) and metrics
are concatenated.  Without a third view it behaves exactly like TwoViewPipeline.

Pair selection follows gluefactory/utils/misc.py:16-46: a key that ends in the digit of the pair's left / right image
is renamed to end in 0 / 1, keys ending in ``<l>to<r>`` become ``0to1``, everything else is dropped.  Unlike the
reference helpers, nested dictionaries (``view0``: {image, image_size, ...}) are stacked / sliced recursively, so
the pipeline works on the nested batch layout of the current datasets as well as on the flat legacy one.
"""
import torch

from .pipeline import TwoViewPipeline, _with_suffix

PAIRS = ("0to1", "0to2", "1to2")


def has_triplet(data):
    return "view2" in data


def get_twoview(data, idx):
    """Entries of ``data`` that belong to the image pair ``idx`` = "<l>to<r>", renamed to the two-view names."""
    left, right = idx[0], idx[-1]
    assert idx == f"{left}to{right}"
    out = {}
    for key, value in data.items():
        if key.endswith(f"{left}to{right}"):
            out[key[:-4] + "0to1"] = value
        elif key.endswith(f"{right}to{left}"):
            out[key[:-4] + "1to0"] = value
    for key, value in data.items():
        if key[-3:-1] == "to" or not key[-1:].isdigit():
            continue
        if key[-1] == left:
            out[key[:-1] + "0"] = value
        if key[-1] == right:
            out[key[:-1] + "1"] = value
    return out


def _cat(values):
    first = values[0]
    if isinstance(first, dict):
        return {k: _cat([v[k] for v in values]) for k in first}
    if torch.is_tensor(first):
        return torch.cat(values, 0)
    if isinstance(first, (list, tuple)):
        return [x for v in values for x in v]
    return first


def _slice(value, lo, hi):
    if isinstance(value, dict):
        return {k: _slice(v, lo, hi) for k, v in value.items()}
    if torch.is_tensor(value) and value.dim() > 0:
        return value[lo:hi]
    if isinstance(value, (list, tuple)):
        return value[lo:hi]
    return value


def stack_twoviews(data, indices=PAIRS):
    """The pairs of a triplet batch stacked on the batch axis (pair 0to1 first)."""
    per_pair = [data[idx] if idx in data else get_twoview(data, idx) for idx in indices]
    keys = [k for k in per_pair[0] if all(k in d for d in per_pair)]
    return {k: _cat([d[k] for d in per_pair]) for k in keys}


def unstack_twoviews(data, batch, indices=PAIRS):
    """Per-pair views of a stacked prediction.  Private entries of a matcher (keys starting with ``_``: e.g. LightGlue's
    per-layer descriptor list and image-stacked head state for its fused loss) are NOT per-pair batched and are left
    out: the stacked prediction itself is kept for the loss (``TripletPipeline._forward``)."""
    public = {k: v for k, v in data.items() if not k.startswith("_")}
    return {idx: {k: _slice(v, i * batch, (i + 1) * batch) for k, v in public.items()} for i, idx in enumerate(indices)}


def _batch_size(data):
    for v in data.values():
        if torch.is_tensor(v) and v.dim() > 0:
            return v.shape[0]
        if isinstance(v, dict):
            b = _batch_size(v)
            if b is not None:
                return b
    return None


class TripletPipeline(TwoViewPipeline):
    default_conf = {"batch_triplets": True, **TwoViewPipeline.default_conf}

    def _stages(self, pred, data):
        for stage in ("matcher", "filter", "solver"):
            if self._has(stage):
                pred = {**pred, **getattr(self, stage)({**data, **pred})}
        return pred

    def _forward(self, data):
        if not has_triplet(data):
            return super()._forward(data)
        assert not self.conf.run_gt_in_forward, "ground truth inside forward is not defined for triplets"
        pred = {}
        for i in "012":
            pred.update(_with_suffix(self.extract_view(data, i), i))
        if self.conf.batch_triplets:
            batch = _batch_size(data["view1"])
            m_pred = self._stages(stack_twoviews(pred), stack_twoviews(data))
            # the loss runs on the stacked matcher output itself: slicing and re-stacking would tear the matcher's
            # private, not-per-pair entries apart (LightGlue: 6 of 9 layers, half of the image-stacked head state)
            pred = {**pred, **unstack_twoviews(m_pred, batch), "_stacked": m_pred}
        else:
            for idx in PAIRS:
                pred[idx] = self._stages(get_twoview(pred, idx), get_twoview(data, idx))
        return pred

    def loss(self, pred, data):
        if not has_triplet(data):
            return super().loss(pred, data)
        if self.conf.batch_triplets:
            stacked = pred["_stacked"] if "_stacked" in pred else stack_twoviews(pred)
            return super().loss(stacked, stack_twoviews(data))
        losses, metrics = {}, {}
        for idx in PAIRS:
            li, mi = super().loss(pred[idx], get_twoview(data, idx))
            for k, v in li.items():
                losses[k] = losses[k] + v if k in losses else v
            for k, v in mi.items():
                metrics[k] = torch.cat([metrics[k], v], 0) if k in metrics else v
        return losses, metrics


__main_model__ = TripletPipeline
