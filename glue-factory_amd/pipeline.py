"""Two-view pipeline: extractor -> matcher -> filter -> solver -> ground truth.

Behavioural mirror of gluefactory/models/two_view_pipeline.py:21-114 on the local BaseModel /
Conf: per-view extractor outputs get the suffix 0/1, every later component sees the merged
``{**data, **pred}``, the ground truth is injected as ``gt_*`` (in ``loss`` unless
``run_gt_in_forward``), and ``loss`` sums the ``total`` of every component that implements one
(components raising NotImplementedError are skipped).
Convention: ``matches0[i]`` is the index in image 1 matched to keypoint i of image 0, -1 if none.
"""
from .base_model import BaseModel, get_model
from .conf import to_container


class TwoViewPipeline(BaseModel):
    default_conf = {
        "extractor": {"name": None, "trainable": False},
        "matcher": {"name": None},
        "filter": {"name": None},
        "solver": {"name": None},
        "ground_truth": {"name": None},
        "allow_no_extract": False,
        "run_gt_in_forward": False,
    }
    required_data_keys = ["view0", "view1"]
    strict_conf = False
    components = ["extractor", "matcher", "filter", "solver", "ground_truth"]

    def _init(self, conf):
        for comp in self.components:
            if conf[comp].name:
                setattr(self, comp, get_model(conf[comp].name)(to_container(conf[comp])))

    def extract_view(self, data, i):
        data_i = data[f"view{i}"]
        pred_i = data_i.get("cache", {})
        skip = len(pred_i) > 0 and self.conf.allow_no_extract
        if self.conf.extractor.name and not skip:
            pred_i = {**pred_i, **self.extractor(data_i)}
        elif self.conf.extractor.name and not self.conf.allow_no_extract:
            pred_i = {**pred_i, **self.extractor({**data_i, **pred_i})}
        return pred_i

    def _forward(self, data):
        pred0, pred1 = self.extract_view(data, "0"), self.extract_view(data, "1")
        pred = {**{k + "0": v for k, v in pred0.items()}, **{k + "1": v for k, v in pred1.items()}}
        for comp in ("matcher", "filter", "solver"):
            if self.conf[comp].name:
                pred = {**pred, **getattr(self, comp)({**data, **pred})}
        if self.conf.ground_truth.name and self.conf.run_gt_in_forward:
            gt = self.ground_truth({**data, **pred})
            pred.update({f"gt_{k}": v for k, v in gt.items()})
        return pred

    def loss(self, pred, data):
        losses, metrics, total = {}, {}, 0
        if self.conf.ground_truth.name and not self.conf.run_gt_in_forward:
            gt = self.ground_truth({**data, **pred})
            pred.update({f"gt_{k}": v for k, v in gt.items()})
        for comp in self.components:
            apply = self.conf[comp].get("apply_loss", True)
            if self.conf[comp].name and apply:
                try:
                    losses_, metrics_ = getattr(self, comp).loss(pred, {**pred, **data})
                except NotImplementedError:
                    continue
                losses = {**losses, **losses_}
                metrics = {**metrics, **metrics_}
                total = losses_["total"] + total
        return {**losses, "total": total}, metrics


__main_model__ = TwoViewPipeline
