"""Minimal pose / camera containers and the depth-reprojection helpers the depth ground truth needs.

The ground-truth code in gt.py only relies on the METHODS used by the reference
(gluefactory/geometry/wrappers.py:111-236 ``Pose``: R, t, inv, transform; :238-400 ``Camera``: image2cam,
cam2image, calibration_matrix; gluefactory/geometry/depth.py:8-71 sample_depth / project), so under
glue-factory the reference's own wrapper objects (any distortion model) are passed straight through.  The
classes here cover the stand-alone case: pinhole cameras, [..., 12] poses.
"""
import torch
import torch.nn.functional as F


class Pose:
    """SE(3) as [..., 12] = (R row-major, t), like the reference wrapper."""

    def __init__(self, data):
        assert data.shape[-1] == 12
        self._data = data

    @classmethod
    def from_Rt(cls, R, t):
        return cls(torch.cat([R.flatten(start_dim=-2), t], -1))

    @property
    def R(self):
        return self._data[..., :9].reshape(self._data.shape[:-1] + (3, 3))

    @property
    def t(self):
        return self._data[..., -3:]

    def inv(self):
        R = self.R.transpose(-1, -2)
        return Pose.from_Rt(R, -(R @ self.t.unsqueeze(-1)).squeeze(-1))

    def transform(self, p3d):
        return p3d @ self.R.transpose(-1, -2) + self.t.unsqueeze(-2)

    def to(self, *a, **k):
        return Pose(self._data.to(*a, **k))


class Camera:
    """Pinhole camera as [..., 6] = (w, h, fx, fy, cx, cy), like the reference wrapper without distortion."""
    eps = 1e-4

    def __init__(self, data):
        if data.shape[-1] != 6:
            raise NotImplementedError("stand-alone Camera is pinhole only; pass glue-factory's wrapper for "
                                      "distorted models")
        self._data = data

    size = property(lambda self: self._data[..., :2])
    f = property(lambda self: self._data[..., 2:4])
    c = property(lambda self: self._data[..., 4:6])

    def calibration_matrix(self):
        K = torch.zeros(*self._data.shape[:-1], 3, 3, device=self._data.device, dtype=self._data.dtype)
        K[..., 0, 2] = self._data[..., 4]
        K[..., 1, 2] = self._data[..., 5]
        K[..., 0, 0] = self._data[..., 2]
        K[..., 1, 1] = self._data[..., 3]
        K[..., 2, 2] = 1.0
        return K

    def image2cam(self, p2d):
        p = (p2d - self.c.unsqueeze(-2)) / self.f.unsqueeze(-2)
        return torch.cat([p, p.new_ones(p.shape[:-1] + (1,))], -1)

    def cam2image(self, p3d):
        z = p3d[..., -1]
        visible = z > self.eps
        p2d = p3d[..., :-1] / z.clamp(min=self.eps).unsqueeze(-1)
        p2d = p2d * self.f.unsqueeze(-2) + self.c.unsqueeze(-2)
        size = self.size.unsqueeze(-2)
        return p2d, visible & torch.all((p2d >= 0) & (p2d <= (size - 1)), -1)

    def to(self, *a, **k):
        return Camera(self._data.to(*a, **k))


def sample_depth(pts, depth_):
    """Bilinear depth at pixel positions, nearest where the bilinear footprint touches a hole
    (gluefactory/geometry/depth.py:8-27)."""
    depth = torch.where(depth_ > 0, depth_, depth_.new_tensor(float("nan")))[:, None]
    h, w = depth.shape[-2:]
    grid = (pts / pts.new_tensor([[w, h]]) * 2 - 1)[:, None]
    lin = F.grid_sample(depth, grid, align_corners=False, mode="bilinear")
    nn_ = F.grid_sample(depth, grid, align_corners=False, mode="nearest")
    interp = torch.where(torch.isnan(lin), nn_, lin)[:, :, 0].permute(0, 2, 1).squeeze(-1)
    return interp, (~torch.isnan(interp)) & (interp > 0)


def project(kpi, di, depthj, camera_i, camera_j, T_itoj, validi, ccth=None):
    """Reproject keypoints of view i into view j through their depth, with the optional cycle-consistency
    check (gluefactory/geometry/depth.py:40-71)."""
    kpi_3d_j = T_itoj.transform(camera_i.image2cam(kpi) * di[..., None])
    kpi_j, validj = camera_j.cam2image(kpi_3d_j)
    validi = validi & validj
    if depthj is None or ccth is None:
        return kpi_j, validi & validj
    dj, validj = sample_depth(kpi_j, depthj)
    kpi_j_3d_j = camera_j.image2cam(kpi_j) * dj[..., None]
    kpi_j_i, validj_i = camera_i.cam2image(T_itoj.inv().transform(kpi_j_3d_j))
    consistent = ((kpi - kpi_j_i) ** 2).sum(-1) < ccth
    return kpi_j, validi & consistent & validj_i & validj


def skew_symmetric(v):
    z = torch.zeros_like(v[..., 0])
    return torch.stack([z, -v[..., 2], v[..., 1], v[..., 2], z, -v[..., 0], -v[..., 1], v[..., 0], z],
                       -1).reshape(v.shape[:-1] + (3, 3))


def sym_epipolar_distance_all(p0, p1, E, eps=1e-15):
    """All-pairs symmetric epipolar distance [..., N0, N1] (gluefactory/geometry/epipolar.py:59-72)."""
    def hom(p):
        return p if p.shape[-1] == 3 else torch.cat([p, p.new_ones(p.shape[:-1] + (1,))], -1)
    p0, p1 = hom(p0), hom(p1)
    p1_E_p0 = torch.einsum("...mi,...ij,...nj->...nm", p1, E, p0).abs()
    E_p0 = torch.einsum("...ij,...nj->...ni", E, p0)
    Et_p1 = torch.einsum("...ij,...mi->...mj", E, p1)
    d0 = p1_E_p0 / (E_p0[..., None, 0] ** 2 + E_p0[..., None, 1] ** 2 + eps).sqrt()
    d1 = p1_E_p0 / (Et_p1[..., None, :, 0] ** 2 + Et_p1[..., None, :, 1] ** 2 + eps).sqrt()
    return (d0 + d1) / 2
