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

from .base_model import BaseModel, get_model
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
        what two calls give, every kernel launch covers 2B images -- otherwise view by view like the reference."""
        v0, v1 = data["view0"], data["view1"]
        ext = getattr(self, "extractor", None)
        batched = (ext is not None and getattr(ext, "batchable_views", False)      # (its forward reads data["image"] only)
                   and not v0.get("cache") and not v1.get("cache") and "image" in v0 and "image" in v1
                   and v0["image"].shape == v1["image"].shape and v0["image"].is_cuda
                   and not any(p_.requires_grad for p_ in ext.parameters())
                   and not any(m.training for m in ext.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)))
        if not batched:
            return self.extract_view(data, "0"), self.extract_view(data, "1")
        b = v0["image"].shape[0]
        out = self.extractor({"image": torch.cat([v0["image"], v1["image"]], 0)})
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
