"""CPU stand-in extractor with the BaseModel plugin interface (see toy_models.py)."""
import torch

from glue_factory_amd.base_model import BaseModel


class ToyExtractor(BaseModel):
    default_conf = {"dim": 8}
    required_data_keys = ["image"]

    def _init(self, conf):
        self.lin = torch.nn.Linear(3, conf.dim)

    def _forward(self, data):
        img = data["image"]                                  # [B,3,H,W]
        feat = img.flatten(2).transpose(1, 2)[:, :16]        # 16 "keypoints"
        return {"keypoints": feat[..., :2] * 10, "descriptors": torch.tanh(self.lin(feat))}

    def loss(self, pred, data):
        raise NotImplementedError


__main_model__ = ToyExtractor
