"""Seeded synthetic keypoint pairs for parity tests and the benchmark.

Image-pair batches are generated on the CPU with an explicit torch.Generator so the
same seed gives the same tensors in the build container and on the GPU box:
  * kpts0 ~ U([0,w] x [0,h]);
  * H = fixed similarity (rotation 10 deg about the centre, scale 1.1, shift (15,-10) px);
  * the first floor(0.7 N) of kpts1 = H(kpts0) + N(0, 0.5 px), the rest uniform; kpts1 is
    then permuted;
  * descriptors: unit-norm Gaussians; a matched kpts1 descriptor is
    normalise(desc0 + 0.3 N(0, I)), unmatched ones are fresh;
  * ground truth through ``gt.gt_matches_from_homography(pos_th=3, neg_th=3)``.
"""
import math

import torch

from .gt import gt_matches_from_homography, warp_points


def similarity_homography(w, h, angle_deg=10.0, scale=1.1, shift=(15.0, -10.0)):
    a = math.radians(angle_deg)
    c, s = math.cos(a) * scale, math.sin(a) * scale
    cx, cy = w / 2.0, h / 2.0
    # x' = R (x - c) + c + t
    return torch.tensor([[c, -s, cx - c * cx + s * cy + shift[0]],
                         [s, c, cy - s * cx - c * cy + shift[1]],
                         [0.0, 0.0, 1.0]], dtype=torch.float32)


def make_pairs(batch, n0, n1=None, dim=256, size=(1024, 1024), seed=0, frac_matched=0.7,
               with_gt=True):
    """Returns a dict of CPU fp32 tensors laid out like a glue-factory batch after extraction."""
    n1 = n0 if n1 is None else n1
    g = torch.Generator().manual_seed(seed)
    w, h = size
    wh = torch.tensor([w, h], dtype=torch.float32)
    kp0 = torch.rand(batch, n0, 2, generator=g) * wh
    H = similarity_homography(w, h)[None].repeat(batch, 1, 1)
    nm = min(int(frac_matched * min(n0, n1)), n0, n1)
    kp1 = torch.rand(batch, n1, 2, generator=g) * wh
    kp1[:, :nm] = warp_points(kp0[:, :nm], H) + 0.5 * torch.randn(batch, nm, 2, generator=g)
    d0 = torch.nn.functional.normalize(torch.randn(batch, n0, dim, generator=g), dim=-1)
    d1 = torch.nn.functional.normalize(torch.randn(batch, n1, dim, generator=g), dim=-1)
    d1[:, :nm] = torch.nn.functional.normalize(
        d0[:, :nm] + 0.3 * torch.randn(batch, nm, dim, generator=g), dim=-1)
    perm = torch.stack([torch.randperm(n1, generator=g) for _ in range(batch)])
    kp1 = kp1.gather(1, perm[..., None].expand(-1, -1, 2))
    d1 = d1.gather(1, perm[..., None].expand(-1, -1, dim))
    data = {
        "keypoints0": kp0, "keypoints1": kp1,
        "descriptors0": d0, "descriptors1": d1,
        "keypoint_scores0": torch.rand(batch, n0, generator=g),
        "keypoint_scores1": torch.rand(batch, n1, generator=g),
        "H_0to1": H,
        "view0": {"image_size": wh[None].repeat(batch, 1)},
        "view1": {"image_size": wh[None].repeat(batch, 1)},
    }
    if with_gt:
        gt = gt_matches_from_homography(kp0, kp1, H, pos_th=3.0, neg_th=3.0)
        data["gt_assignment"] = gt["assignment"]
        data["gt_matches0"] = gt["matches0"]
        data["gt_matches1"] = gt["matches1"]
    return data


def to_device(data, device, non_blocking=True):
    """Recursive .to(device) (the role of gluefactory/utils/tensor.py batch_to_device)."""
    if isinstance(data, dict):
        return {k: to_device(v, device, non_blocking) for k, v in data.items()}
    if torch.is_tensor(data):
        return data.to(device, non_blocking=non_blocking)
    return data
