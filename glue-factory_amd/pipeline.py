"""Two-view pipeline: extractor -> matcher -> filter -> solver -> ground truth.

Behavioural mirror of gluefactory/models/two_view_pipeline.py:21-114 on the local BaseModel / Conf:
per-view extractor outputs get the suffix 0/1 (a view's ``cache`` dict stands in for the extractor when
``allow_no_extract`` is set: the cached-feature training mode), every later stage sees the merged
``{**data, **pred}``, the ground truth is injected under ``gt_*`` (inside ``loss`` unless
``run_gt_in_forward``), and ``loss`` adds up the ``total`` of every stage that implements one (stages
raising NotImplementedError are skipped).
Convention: ``matches0[i]`` is the index in image 1 matched to keypoint i of image 0, -1 if none.
"""
import torch
from torch import nn

from .base_model import BaseModel, BatchedExtractionUnsupported, get_model
from .conf import to_container

_STAGES = ("extractor", "matcher", "filter", "solver", "ground_truth")
_AFTER_EXTRACTION = ("matcher", "filter", "solver")


def _with_suffix(d, suffix):
    return {key + suffix: value for key, value in d.items()}


class TwoViewPipeline(BaseModel):
    default_conf = {
        **{stage: {"name": None} for stage in _STAGES},
        "extractor": {"name": None, "trainable": False},
        "allow_no_extract": False,
        "run_gt_in_forward": False,
    }
    required_data_keys = ["view0", "view1"]
    strict_conf = False
    components = list(_STAGES)

    def _init(self, conf):
        for stage in _STAGES:
            name = conf[stage].name
            if name:
                setattr(self, stage, get_model(name)(to_container(conf[stage])))

    def _has(self, stage):
        return bool(self.conf[stage].name)

    def extract_view(self, data, i):
        view = data[f"view{i}"]
        cached = view.get("cache", {})
        if not self._has("extractor"):
            return cached
        reuse_cache = bool(cached) and self.conf.allow_no_extract
        if reuse_cache:
            return cached
        return {**cached, **self.extractor(view)}          # extractor outputs win over cached entries

    def _extract_pair(self, data):
        """Both views' extractor outputs.  A FROZEN extractor (no trainable parameter, BatchNorm layers in eval mode) sees
        the two image batches as one call on their concatenation when they have the same shape -- per-image results are
        what two calls give, every kernel launch covers 2B images -- otherwise view by view like the reference.
        Every batched tensor key the two views share (`image_size`, which the non-free SuperPoint's border removal reads,
        gluefactory_nonfree/superpoint.py:236-244) rides along in the concatenated call; a view key that cannot be
        concatenated, a batch of one without a fixed keypoint count (variable-length outputs exist for b == 1 only), or
        an extractor that refuses the batch falls back to the per-view calls."""
        v0, v1 = data["view0"], data["view1"]
        ext = getattr(self, "extractor", None)
        batched = (ext is not None and getattr(ext, "batchable_views", False)
                   and not v0.get("cache") and not v1.get("cache") and "image" in v0 and "image" in v1
                   and v0["image"].shape == v1["image"].shape and v0["image"].is_cuda
                   and not any(p_.requires_grad for p_ in ext.parameters())
                   and not any(m.training for m in ext.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)))
        if batched:
            b = v0["image"].shape[0]
            conf = getattr(ext, "conf", {})
            k = ext._max_keypoints() if hasattr(ext, "_max_keypoints") else conf.get("max_num_keypoints")
            fixed_count = bool(conf.get("force_num_keypoints", False)) and (k or -1) > 0
            batched = b > 1 or fixed_count
        both = None
        if batched:
            both = {}
            for k in set(v0) | set(v1):
                a, c = v0.get(k), v1.get(k)
                if k == "cache" or (a is None and c is None):
                    continue
                if (torch.is_tensor(a) and torch.is_tensor(c) and a.dim() > 0 and a.shape == c.shape and a.shape[0] == b
                        and a.dtype == c.dtype):
                    both[k] = torch.cat([a, c], 0)
                elif not torch.is_tensor(a) and not torch.is_tensor(c):
                    try:                           # (names, scalars: ride along when identical in both views; anything whose
                        same = bool(a == c)        # comparison is not a plain truth value -- arrays, lists of tensors -- is dropped)
                    except (ValueError, RuntimeError, TypeError):
                        same = False
                    if same:
                        both[k] = a
                elif torch.is_tensor(a) or torch.is_tensor(c):
                    both = None                    # a per-view tensor that cannot ride in one call: view by view
                    break
        if both is None:
            return self.extract_view(data, "0"), self.extract_view(data, "1")
        try:
            out = self.extractor(both)
        except BatchedExtractionUnsupported:        # ("different keypoint counts" ...: what two b-sized calls can still serve;
                                                    # any other extractor error propagates)
            return self.extract_view(data, "0"), self.extract_view(data, "1")
        if not all(torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == 2 * b for t in out.values()):
            return self.extract_view(data, "0"), self.extract_view(data, "1")      # (an extractor with non-batched outputs)
        return {k: t[:b] for k, t in out.items()}, {k: t[b:] for k, t in out.items()}

    def _inject_gt(self, pred, data):
        labels = self.ground_truth({**data, **pred})
        pred.update(_with_suffix_prefix(labels))

    def _forward(self, data):
        p0, p1 = self._extract_pair(data)
        pred = {**_with_suffix(p0, "0"), **_with_suffix(p1, "1")}
        for stage in _AFTER_EXTRACTION:
            if self._has(stage):
                pred = {**pred, **getattr(self, stage)({**data, **pred})}
        if self._has("ground_truth") and self.conf.run_gt_in_forward:
            self._inject_gt(pred, data)
        return pred

    def loss(self, pred, data):
        if self._has("ground_truth") and not self.conf.run_gt_in_forward:
            self._inject_gt(pred, data)
        losses, metrics, total = {}, {}, 0
        for stage in _STAGES:
            if not self._has(stage) or not self.conf[stage].get("apply_loss", True):
                continue
            try:
                stage_losses, stage_metrics = getattr(self, stage).loss(pred, {**pred, **data})
            except NotImplementedError:
                continue
            losses.update(stage_losses)
            metrics.update(stage_metrics)
            total = stage_losses["total"] + total
        return {**losses, "total": total}, metrics


def _with_suffix_prefix(labels):
    return {f"gt_{key}": value for key, value in labels.items()}


__main_model__ = TwoViewPipeline
