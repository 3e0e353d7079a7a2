"""A glue-factory DATASET plugin (gluefactory/datasets/base_dataset.py:100-206 interface, resolved by the reference's own
`get_dataset(name)` through this module's absolute name) that serves seeded synthetic keypoint pairs with their ground
truth -- the cached-feature training mode without files.  Test infrastructure for tests/test_*reference_train*.py: the
reference's `training()` needs a dataset object, and its real ones need cv2 / h5py / image folders."""
import torch
from gluefactory.datasets.base_dataset import BaseDataset        # the REFERENCE's base class (oracle/_ref or /root/reference)

from glue_factory_amd.synthetic import make_pairs


class _Pairs(torch.utils.data.Dataset):
    def __init__(self, n_items, n_kpts, dim, seed, with_image):
        d = make_pairs(n_items, n_kpts, dim=dim, size=(640, 480), seed=seed)
        self.items = []
        for i in range(n_items):
            it = {k: v[i] for k, v in d.items() if torch.is_tensor(v)}
            it["view0"] = {"image_size": d["view0"]["image_size"][i]}
            it["view1"] = {"image_size": d["view1"]["image_size"][i]}
            if with_image:            # (the toy CPU model's loss reads an image)
                it["view0"]["image"] = torch.zeros(1, 8, 8)
            it["name"] = f"pair{seed}_{i}"
            it["idx"] = i
            self.items.append(it)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


class SyntheticPairs(BaseDataset):
    default_conf = {"n_train": 16, "n_val": 4, "n_kpts": 128, "dim": 256, "with_image": False}

    def _init(self, conf):
        pass

    def get_dataset(self, split):
        n = self.conf.n_train if split == "train" else self.conf.n_val
        return _Pairs(n, self.conf.n_kpts, self.conf.dim, self.conf.seed + (0 if split == "train" else 1000),
                      self.conf.with_image)
