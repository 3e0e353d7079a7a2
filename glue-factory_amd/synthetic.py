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
        data["gt_assignment_col0"] = gt["assignment_col0"]
        data["gt_matches0"] = gt["matches0"]
        data["gt_matches1"] = gt["matches1"]
    return data


def to_device(data, device, non_blocking=None):
    """Recursive .to(device) (the role of gluefactory/utils/tensor.py batch_to_device)."""
    if isinstance(data, dict):
        return {k: to_device(v, device, non_blocking) for k, v in data.items()}
    if torch.is_tensor(data):
        # non_blocking=None: asynchronous only where that is safe -- pinned host memory or a device source.  An asynchronous
        # copy from PAGEABLE host memory on ROCm reads the source AFTER the call returned: a batch that is freed or refilled
        # right away (this module's own make_pairs loop) reaches the device corrupted now and then (round 6).
        nb = (data.device.type != "cpu" or data.is_pinned()) if non_blocking is None else non_blocking
        return data.to(device, non_blocking=nb)
    return data


def make_point_line_pairs(batch, n_kpts, n_lines, dim=256, size=(1024, 1024), seed=0, frac_matched=0.7,
                          share_junctions=0.25):
    """GlueStick-shaped batch: the point set of each image is
    [2*n_lines line-endpoint junctions ; n_kpts keypoints]; ``lines{0,1}`` [B,Nl,2,2] hold endpoint
    coordinates, ``lines_junc_idx{0,1}`` [B,Nl,2] index the junction block (a fraction of endpoints
    re-uses an earlier junction so the junction graph has shared nodes), ``line_scores`` ~ U(0,1).
    The first floor(frac_matched * Nl) lines of image 1 are the warped lines of image 0 (then
    permuted), which gives the line ground truth by construction."""
    g = torch.Generator().manual_seed(seed)
    w, h = size
    wh = torch.tensor([w, h], dtype=torch.float32)
    base = make_pairs(batch, n_kpts, dim=dim, size=size, seed=seed + 1, frac_matched=frac_matched)
    H = base["H_0to1"]
    nj = 2 * n_lines
    # image 0 lines: random segments of length >= 15 px
    p0 = torch.rand(batch, n_lines, 2, generator=g) * (wh - 60) + 30
    ang = torch.rand(batch, n_lines, generator=g) * 2 * math.pi
    length = 15 + torch.rand(batch, n_lines, generator=g) * 60
    p1 = p0 + length[..., None] * torch.stack([torch.cos(ang), torch.sin(ang)], -1)
    lines0 = torch.stack([p0, p1], 2)                                   # [B,Nl,2,2]
    nm = int(frac_matched * n_lines)
    lines1 = torch.rand(batch, n_lines, 2, 2, generator=g) * wh
    lines1[:, :nm] = warp_points(lines0[:, :nm].reshape(batch, -1, 2), H).reshape(batch, nm, 2, 2) \
        + 0.5 * torch.randn(batch, nm, 2, 2, generator=g)
    perm = torch.stack([torch.randperm(n_lines, generator=g) for _ in range(batch)])
    lines1 = lines1.gather(1, perm[:, :, None, None].expand(-1, -1, 2, 2))
    inv = torch.argsort(perm, 1)                                        # original index -> new position
    gt_l0 = torch.full((batch, n_lines), -1, dtype=torch.long)
    gt_l0[:, :nm] = inv[:, :nm]
    gt_l1 = torch.full((batch, n_lines), -1, dtype=torch.long)
    gt_l1.scatter_(1, inv[:, :nm], torch.arange(nm)[None].expand(batch, -1))
    gt_la = torch.zeros(batch, n_lines, n_lines, dtype=torch.bool)
    gt_la.scatter_(2, gt_l0.clamp(min=0)[..., None], (gt_l0 >= 0)[..., None])

    def junctions(lines):
        idx = torch.arange(nj).reshape(1, n_lines, 2).repeat(batch, 1, 1)
        share = torch.rand(batch, n_lines, 2, generator=g) < share_junctions
        prev = (torch.rand(batch, n_lines, 2, generator=g) * idx.clamp(min=1)).long()   # an earlier junction
        idx = torch.where(share & (idx > 1), prev, idx)
        coords = lines.reshape(batch, nj, 2).clone()
        return idx, coords

    idx0, jc0 = junctions(lines0)
    idx1, jc1 = junctions(lines1)
    jd0 = torch.nn.functional.normalize(torch.randn(batch, nj, dim, generator=g), dim=-1)
    jd1 = torch.nn.functional.normalize(torch.randn(batch, nj, dim, generator=g), dim=-1)
    data = dict(base)
    data["keypoints0"] = torch.cat([jc0, base["keypoints0"]], 1)
    data["keypoints1"] = torch.cat([jc1, base["keypoints1"]], 1)
    data["descriptors0"] = torch.cat([jd0, base["descriptors0"]], 1)
    data["descriptors1"] = torch.cat([jd1, base["descriptors1"]], 1)
    data["keypoint_scores0"] = torch.cat([torch.rand(batch, nj, generator=g), base["keypoint_scores0"]], 1)
    data["keypoint_scores1"] = torch.cat([torch.rand(batch, nj, generator=g), base["keypoint_scores1"]], 1)
    data.update({"lines0": lines0, "lines1": lines1, "lines_junc_idx0": idx0, "lines_junc_idx1": idx1,
                 "line_scores0": torch.rand(batch, n_lines, generator=g),
                 "line_scores1": torch.rand(batch, n_lines, generator=g)})
    gt = gt_matches_from_homography(data["keypoints0"], data["keypoints1"], H, pos_th=3.0, neg_th=3.0)
    data.update({"gt_assignment": gt["assignment"], "gt_assignment_col0": gt["assignment_col0"], "gt_matches0": gt["matches0"], "gt_matches1": gt["matches1"],
                 "gt_line_assignment": gt_la, "gt_line_assignment_col0": gt_l0.clone(),
                 "gt_line_matches0": gt_l0, "gt_line_matches1": gt_l1})
    return data
