"""CPU stand-ins with the BaseModel plugin interface (the HIP matchers need a GPU): resolved by get_model through
their absolute module path, like any external plugin."""
import torch

from glue_factory_amd.base_model import BaseModel


class ToyMatcher(BaseModel):
    default_conf = {"dim": 8}
    required_data_keys = ["descriptors0", "descriptors1"]

    def _init(self, conf):
        self.w = torch.nn.Parameter(torch.eye(conf.dim))

    def _forward(self, data):
        sim = torch.einsum("bnd,de,bme->bnm", data["descriptors0"], self.w, data["descriptors1"])
        return {"scores": sim, "matches0": sim.argmax(2)}

    def loss(self, pred, data):
        total = -pred["scores"].diagonal(dim1=1, dim2=2).mean(1) + data["view0"]["image"].mean((1, 2, 3)) * 0
        return {"total": total, "n": torch.ones_like(total)}, {"acc": (pred["matches0"] == 0).float().mean(1)}


__main_model__ = ToyMatcher
