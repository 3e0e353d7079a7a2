"""Ground-truth assignment from a homography (no-grad), the step that feeds the loss.

Mirrors the behaviour of gluefactory/geometry/gt_generation.py:109-161
(gt_matches_from_homography) and gluefactory/geometry/homography.py:161-180
(warp_points_torch): warp both ways, symmetric squared distance, mutual nearest
neighbour within ``pos_th`` px -> positive; nearest warped neighbour farther than
``neg_th`` px -> unmatched (-1); everything else ignored (-2).

Runs on whatever device the keypoints live on (stock torch ops; the fused HIP
nearest-neighbour kernel is the "next" row of SURVEY.md §8f).
"""
import numpy as np
import torch

IGNORE_FEATURE = -2
UNMATCHED_FEATURE = -1


def inv3x3(H):
    """Inverse of [..., 3, 3] matrices by the adjugate (cross products of the rows): plain elementwise kernels, so it
    runs inside a hipGraph capture -- torch.linalg.inv goes through a solver with host-side bookkeeping and does not
    ("operation not permitted when stream is capturing")."""
    r0, r1, r2 = H[..., 0, :], H[..., 1, :], H[..., 2, :]
    c0, c1, c2 = torch.linalg.cross(r1, r2, dim=-1), torch.linalg.cross(r2, r0, dim=-1), torch.linalg.cross(r0, r1, dim=-1)
    det = (r0 * c0).sum(-1, keepdim=True)
    return torch.stack([c0, c1, c2], -1) / det[..., None]


def warp_points(points, H, inverse=False, eps=1e-5):
    """points [B,N,2], H [B,3,3] (or [3,3]) -> H (or H^-1) applied in homogeneous coords."""
    Hm = inv3x3(H) if inverse else H
    if Hm.dim() == 2:
        Hm = Hm[None]
    # w_i = H_i0 x + H_i1 y + H_i2 as broadcast multiply-adds (not a [N,3] x [3,3] library product: exact fp32 in any
    # autocast state, the same summation order as the reference's matmul of three terms, and kernel-only in a captured step)
    x, y = points[..., 0:1], points[..., 1:2]
    Hm = Hm[:, None]                                                    # [B,1,3,3]
    w = Hm[..., 0] * x + Hm[..., 1] * y + Hm[..., 2]                    # [B,N,3]
    return w[..., :2] / (w[..., 2:] + eps)


@torch.no_grad()
def gt_matches_from_homography_fused(kp0, kp1, H, pos_th=3.0, neg_th=6.0, with_reward=False):
    """Same labels through the HIP nearest-neighbour kernel (gf_gt_nn): no [B,M,N] fp32 tensor is
    built; the dense boolean ``assignment`` the plugin contract asks for is a zero-fill + scatter.
    ``reward`` (dense, unused by the matcher losses) is only produced on request (stock torch)."""
    from . import lib as _lib
    b, m = kp0.shape[:2]
    n = kp1.shape[1]
    kp0, kp1 = kp0.float().contiguous(), kp1.float().contiguous()
    # fp32 whatever the caller's autocast state: the reference calls its ground truth from inside the autocast region of
    # the train loop (train.py:470-476), where the einsum of warp_points would come back in bf16 -- 3 significant digits
    # for pixel coordinates, and a half-sized buffer for the fp32 kernel below (found as a GPU memory fault)
    with torch.autocast(device_type="cuda", enabled=False):
        kp0_1 = warp_points(kp0, H.float(), inverse=False).float().contiguous()
        kp1_0 = warp_points(kp1, H.float(), inverse=True).float().contiguous()
    dev = kp0.device
    L = _lib.load()
    st = torch.cuda.current_stream().cuda_stream

    def nn(own, own_w, oth, oth_w):
        no = own.shape[1]
        assert all(t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda for t in (own, own_w, oth, oth_w))
        arg = torch.empty((b, no), dtype=torch.int64, device=dev)
        dmin = torch.empty((b, no), dtype=torch.float32, device=dev)
        omin = torch.empty((b, no), dtype=torch.float32, device=dev)
        _lib.check(L.gf_gt_nn(own.data_ptr(), own_w.data_ptr(), oth.data_ptr(), oth_w.data_ptr(), arg.data_ptr(),
                              dmin.data_ptr(), omin.data_ptr(), b, no, oth.shape[1], st), "gf_gt_nn")
        return arg, dmin, omin

    min0, d0, own0 = nn(kp0, kp0_1, kp1, kp1_0)      # rows: dist0 = |H kp0 - kp1|^2 is the "own" distance
    min1, d1, own1 = nn(kp1, kp1_0, kp0, kp0_1)      # columns: dist1 = |kp0 - H^-1 kp1|^2
    ar0 = torch.arange(m, device=dev)[None]
    ar1 = torch.arange(n, device=dev)[None]
    pos0 = (min1.gather(1, min0) == ar0) & (d0 < pos_th ** 2)
    pos1 = (min0.gather(1, min1) == ar1) & (d1 < pos_th ** 2)
    positive = torch.zeros(b, m, n, dtype=torch.bool, device=dev)
    positive.scatter_(2, min0[..., None], pos0[..., None])
    m0 = torch.where(pos0, min0, torch.full_like(min0, IGNORE_FEATURE))
    m1 = torch.where(pos1, min1, torch.full_like(min1, IGNORE_FEATURE))
    m0 = torch.where(own0 > neg_th ** 2, torch.full_like(m0, UNMATCHED_FEATURE), m0)
    m1 = torch.where(own1 > neg_th ** 2, torch.full_like(m1, UNMATCHED_FEATURE), m1)
    out = {"assignment": positive, "matches0": m0, "matches1": m1,
           "matching_scores0": (m0 > -1).float(), "matching_scores1": (m1 > -1).float(),
           "proj_0to1": kp0_1, "proj_1to0": kp1_0,
           # private extra: the (at most one) positive column of every row of `assignment`, -1 if none.
           # Lets the sparse losses skip nonzero() on the dense matrix (a 134 MB scan + a host sync).
           "assignment_col0": torch.where(pos0, min0, torch.full_like(min0, -1))}
    if with_reward:
        dist0 = ((kp0_1[:, :, None] - kp1[:, None]) ** 2).sum(-1)
        dist1 = ((kp0[:, :, None] - kp1_0[:, None]) ** 2).sum(-1)
        dist = torch.maximum(dist0, dist1)
        out["reward"] = (dist < pos_th ** 2).float() - (dist > neg_th ** 2).float()
    return out


@torch.no_grad()
def gt_matches_from_homography(kp0, kp1, H, pos_th=3.0, neg_th=6.0):
    b, m = kp0.shape[:2]
    n = kp1.shape[1]
    if m == 0 or n == 0:
        return {
            "assignment": torch.zeros(b, m, n, dtype=torch.bool, device=kp0.device),
            "matches0": -torch.ones(b, m, dtype=torch.long, device=kp0.device),
            "matches1": -torch.ones(b, n, dtype=torch.long, device=kp0.device),
        }
    kp0_1 = warp_points(kp0, H, inverse=False)
    kp1_0 = warp_points(kp1, H, inverse=True)
    dist0 = ((kp0_1[:, :, None] - kp1[:, None]) ** 2).sum(-1)
    dist1 = ((kp0[:, :, None] - kp1_0[:, None]) ** 2).sum(-1)
    dist = torch.maximum(dist0, dist1)
    reward = (dist < pos_th ** 2).float() - (dist > neg_th ** 2).float()
    min0 = dist.argmin(-1)
    min1 = dist.argmin(-2)
    ar0 = torch.arange(m, device=kp0.device)[None]
    ar1 = torch.arange(n, device=kp0.device)[None]
    close = dist < pos_th ** 2
    # positive[b,i,j] <=> j is i's nearest, i is j's nearest, and they are close
    pos0 = (min1.gather(1, min0) == ar0) & close.gather(2, min0[..., None]).squeeze(-1)
    pos1 = (min0.gather(1, min1) == ar1) & close.gather(1, min1[:, None]).squeeze(1)
    positive = torch.zeros(b, m, n, dtype=torch.bool, device=kp0.device)
    positive.scatter_(2, min0[..., None], pos0[..., None])
    neg0 = dist0.min(-1).values > neg_th ** 2
    neg1 = dist1.min(-2).values > neg_th ** 2
    m0 = torch.where(pos0, min0, torch.full_like(min0, IGNORE_FEATURE))
    m1 = torch.where(pos1, min1, torch.full_like(min1, IGNORE_FEATURE))
    m0 = torch.where(neg0, torch.full_like(m0, UNMATCHED_FEATURE), m0)
    m1 = torch.where(neg1, torch.full_like(m1, UNMATCHED_FEATURE), m1)
    return {
        "assignment": positive,
        "assignment_col0": torch.where(pos0, min0, torch.full_like(min0, -1)),
        "reward": reward,
        "matches0": m0,
        "matches1": m1,
        "matching_scores0": (m0 > -1).float(),
        "matching_scores1": (m1 > -1).float(),
        "proj_0to1": kp0_1,
        "proj_1to0": kp1_0,
    }


# ------------------------------------------------------------------------------------------------ depth + pose
def _depth_projections(kp0, kp1, data, cc_th, kw):
    from .geometry import project, sample_depth
    camera0, camera1 = data["view0"]["camera"], data["view1"]["camera"]
    T_0to1 = data["T_0to1"]
    T_1to0 = data["T_1to0"] if "T_1to0" in data else T_0to1.inv()
    depth0, depth1 = data["view0"].get("depth"), data["view1"].get("depth")
    if "depth_keypoints0" in kw and "depth_keypoints1" in kw:
        d0, valid0 = kw["depth_keypoints0"], kw["valid_depth_keypoints0"]
        d1, valid1 = kw["depth_keypoints1"], kw["valid_depth_keypoints1"]
    else:
        assert depth0 is not None and depth1 is not None
        d0, valid0 = sample_depth(kp0, depth0)
        d1, valid1 = sample_depth(kp1, depth1)
    kp0_1, visible0 = project(kp0, d0, depth1, camera0, camera1, T_0to1, valid0, ccth=cc_th)
    kp1_0, visible1 = project(kp1, d1, depth0, camera1, camera0, T_1to0, valid1, ccth=cc_th)
    return d0, d1, valid0, valid1, kp0_1, kp1_0, visible0, visible1


@torch.no_grad()
def gt_matches_from_pose_depth(kp0, kp1, data, pos_th=3, neg_th=5, epi_th=None, cc_th=None, **kw):
    """Ground truth from depth maps and the relative pose (gluefactory/geometry/gt_generation.py:13-106): reproject
    both ways through the sampled depth, positives = mutual nearest neighbours under max(d_0->1, d_1->0) among
    co-visible points closer than pos_th, negatives = points with valid depth whose reprojection is farther
    than neg_th from every keypoint (optionally extended by the epipolar test), everything else ignored (-2)."""
    from .geometry import skew_symmetric, sym_epipolar_distance_all
    b, m = kp0.shape[:2]
    n = kp1.shape[1]
    if m == 0 or n == 0:
        return {"assignment": torch.zeros(b, m, n, dtype=torch.bool, device=kp0.device),
                "matches0": -torch.ones(b, m, dtype=torch.long, device=kp0.device),
                "matches1": -torch.ones(b, n, dtype=torch.long, device=kp0.device)}
    d0, d1, valid0, valid1, kp0_1, kp1_0, visible0, visible1 = _depth_projections(kp0, kp1, data, cc_th, kw)
    mask_visible = visible0.unsqueeze(-1) & visible1.unsqueeze(-2)
    dist0 = torch.sum((kp0_1.unsqueeze(-2) - kp1.unsqueeze(-3)) ** 2, -1)
    dist1 = torch.sum((kp0.unsqueeze(-2) - kp1_0.unsqueeze(-3)) ** 2, -1)
    dist = torch.max(dist0, dist1)
    inf = dist.new_tensor(float("inf"))
    dist = torch.where(mask_visible, dist, inf)
    min0 = dist.min(-1).indices
    min1 = dist.min(-2).indices
    ismin0 = torch.zeros(dist.shape, dtype=torch.bool, device=dist.device)
    ismin1 = ismin0.clone()
    ismin0.scatter_(-1, min0.unsqueeze(-1), value=1)
    ismin1.scatter_(-2, min1.unsqueeze(-2), value=1)
    positive = ismin0 & ismin1 & (dist < pos_th ** 2)
    negative0 = (dist0.min(-1).values > neg_th ** 2) & valid0
    negative1 = (dist1.min(-2).values > neg_th ** 2) & valid1
    unmatched, ignore = min0.new_tensor(UNMATCHED_FEATURE), min0.new_tensor(IGNORE_FEATURE)
    m0 = torch.where(positive.any(-1), min0, ignore)
    m1 = torch.where(positive.any(-2), min1, ignore)
    m0 = torch.where(negative0, unmatched, m0)
    m1 = torch.where(negative1, unmatched, m1)
    camera0, camera1, T_0to1 = data["view0"]["camera"], data["view1"]["camera"], data["T_0to1"]
    Fm = (camera1.calibration_matrix().inverse().transpose(-1, -2) @ (skew_symmetric(T_0to1.t) @ T_0to1.R)
          @ camera0.calibration_matrix().inverse())
    epi_dist = sym_epipolar_distance_all(kp0, kp1, Fm)
    if epi_th is not None:
        mask_ignore = (m0.unsqueeze(-1) == ignore) & (m1.unsqueeze(-2) == ignore)
        epi_dist = torch.where(mask_ignore, epi_dist, inf)
        exclude0 = epi_dist.min(-1).values > neg_th
        exclude1 = epi_dist.min(-2).values > neg_th
        m0 = torch.where((~valid0) & exclude0, ignore.new_tensor(-1), m0)
        m1 = torch.where((~valid1) & exclude1, ignore.new_tensor(-1), m1)
    return {"assignment": positive,
            "assignment_col0": torch.where(positive.any(-1), min0, torch.full_like(min0, -1)),
            "reward": (dist < pos_th ** 2).float() - (epi_dist > neg_th).float(),
            "matches0": m0, "matches1": m1,
            "matching_scores0": (m0 > -1).float(), "matching_scores1": (m1 > -1).float(),
            "depth_keypoints0": d0, "depth_keypoints1": d1,
            "proj_0to1": kp0_1, "proj_1to0": kp1_0, "visible0": visible0, "visible1": visible1}


@torch.no_grad()
def gt_matches_from_pose_depth_fused(kp0, kp1, data, pos_th=3, neg_th=5, cc_th=None, **kw):
    """Same labels without any [B,M,N] fp32 tensor (HIP nearest-neighbour kernel gf_gt_nn); for the
    configuration the matchers train with (no epipolar extension, no dense ``reward``).  Points that are not
    co-visible are moved far away for the mutual-NN search (their rows / columns would be +inf in the dense
    form); the negative test runs on the true reprojections, exactly as in the dense form."""
    from . import lib as _lib
    b, m = kp0.shape[:2]
    n = kp1.shape[1]
    d0, d1, valid0, valid1, kp0_1, kp1_0, visible0, visible1 = _depth_projections(kp0, kp1, data, cc_th, kw)
    dev = kp0.device
    kp0f, kp1f = kp0.float().contiguous(), kp1.float().contiguous()
    # reprojections of points without depth are NaN: they are invisible (moved away below) and never "valid"
    p01 = torch.nan_to_num(kp0_1.float(), nan=0.0, posinf=0.0, neginf=0.0).contiguous()
    p10 = torch.nan_to_num(kp1_0.float(), nan=0.0, posinf=0.0, neginf=0.0).contiguous()
    L = _lib.load()
    st = torch.cuda.current_stream().cuda_stream

    def nn(own, own_w, oth, oth_w):
        no = own.shape[1]
        assert all(t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda for t in (own, own_w, oth, oth_w))
        arg = torch.empty((b, no), dtype=torch.int64, device=dev)
        dmin = torch.empty((b, no), dtype=torch.float32, device=dev)
        omin = torch.empty((b, no), dtype=torch.float32, device=dev)
        _lib.check(L.gf_gt_nn(own.data_ptr(), own_w.data_ptr(), oth.data_ptr(), oth_w.data_ptr(), arg.data_ptr(),
                              dmin.data_ptr(), omin.data_ptr(), b, no, oth.shape[1], st), "gf_gt_nn")
        return arg, dmin, omin

    # invisible points: reprojection AND position pushed out, so max(d0, d1) is huge against every partner
    far = 1.0e8
    off0 = torch.where(visible0[..., None], torch.zeros_like(kp0f), torch.full_like(kp0f, far))
    off1 = torch.where(visible1[..., None], torch.zeros_like(kp1f), torch.full_like(kp1f, -far))
    a0, a0w = (kp0f + off0).contiguous(), (p01 + off0).contiguous()
    a1, a1w = (kp1f + off1).contiguous(), (p10 + off1).contiguous()
    min0, dd0, _ = nn(a0, a0w, a1, a1w)
    min1, dd1, _ = nn(a1, a1w, a0, a0w)
    _, _, own0 = nn(kp0f, p01, kp1f, p10)          # true reprojections: min_j |kp0_1 - kp1|^2
    _, _, own1 = nn(kp1f, p10, kp0f, p01)
    ar0 = torch.arange(m, device=dev)[None]
    ar1 = torch.arange(n, device=dev)[None]
    pos0 = (min1.gather(1, min0) == ar0) & (dd0 < pos_th ** 2) & visible0 & visible1.gather(1, min0)
    pos1 = (min0.gather(1, min1) == ar1) & (dd1 < pos_th ** 2) & visible1 & visible0.gather(1, min1)
    positive = torch.zeros(b, m, n, dtype=torch.bool, device=dev)
    positive.scatter_(2, min0[..., None], pos0[..., None])
    m0 = torch.where(pos0, min0, torch.full_like(min0, IGNORE_FEATURE))
    m1 = torch.where(pos1, min1, torch.full_like(min1, IGNORE_FEATURE))
    m0 = torch.where((own0 > neg_th ** 2) & valid0, torch.full_like(m0, UNMATCHED_FEATURE), m0)
    m1 = torch.where((own1 > neg_th ** 2) & valid1, torch.full_like(m1, UNMATCHED_FEATURE), m1)
    return {"assignment": positive, "assignment_col0": torch.where(pos0, min0, torch.full_like(min0, -1)),
            "matches0": m0, "matches1": m1,
            "matching_scores0": (m0 > -1).float(), "matching_scores1": (m1 > -1).float(),
            "depth_keypoints0": d0, "depth_keypoints1": d1,
            "proj_0to1": kp0_1, "proj_1to0": kp1_0, "visible0": visible0, "visible1": visible1}


# ------------------------------------------------------------------------------------------------ lines
def _segments(lines):
    """[B,L,2,2] / [B,L,P,2] / [B,L,4] -> [B,L,4] = (x0, y0, x1, y1)."""
    if lines.shape[-2:] == (2, 2):
        return lines.flatten(-2)
    if lines.dim() == 4:
        return torch.cat([lines[:, :, 0], lines[:, :, -1]], dim=2)
    return lines


def _close_point_counts(seg, pts, dist_th, keep=None):
    """count[b, a, c] = number of the sampled points pts[b, c, :, :] (of those flagged in keep[b, c, :], if given) that lie
    within dist_th of segment seg[b, a] and project onto it (the test of gt_generation.py:173-206, incl. its fp16-rounded segment length: the
    direction is normalised by the half-precision norm, which the thresholds are sensitive to)."""
    d = seg[..., 2:] - seg[..., :2]
    length = torch.norm(d, dim=-1).half()                    # reference quirk, kept for identical labels
    u = d / length.unsqueeze(-1)                             # [B,A,2] (promotes back to fp32)
    rel = pts[:, None] - seg[..., None, None, 2:]            # [B,A,C,P,2], relative to the segment END point
    if rel.is_cuda:                                          # the reference halves the big tensor on GPUs
        rel, u = rel.half(), u.half()
    along = rel[..., 0] * u[..., None, None, 0] + rel[..., 1] * u[..., None, None, 1]
    perp = rel[..., 1] * u[..., None, None, 0] - rel[..., 0] * u[..., None, None, 1]
    inside = (along <= 0) & (along.abs() <= length[..., None, None])
    close = (perp.abs() < dist_th) & inside
    if keep is not None:
        close = close & keep[:, None]
    return close.sum(-1)


def _line_samples(seg, npts):
    """npts points on every segment, end points included: [B,L,4] -> [B,L,npts,2] (gt_generation.py:164-170)."""
    step = (seg[..., 2:4] - seg[..., :2]) / (npts - 1)
    t = torch.arange(npts).to(seg)
    return seg[..., None, :2] + t[:, None] * step[..., None, :]


def _mostly_outside(p, w, h, min_visibility_th):
    out = (p < 0).any(-1) | (p >= torch.tensor([w, h]).to(p)).any(-1)
    return out.float().mean(-1) >= (1 - min_visibility_th)


def _assign_lines(both, mask_close, unmatched0, unmatched1, ignore0, ignore1):
    """The labelling both line ground truths end in (gt_generation.py:343-407, 502-558): one-to-one assignment that maximises
    the product of the two close-point counts (Hungarian method on the CPU, scipy -- as the reference), positives = assigned
    pairs that also pass `mask_close`; the rest unmatched (-1) or ignored (-2)."""
    from scipy.optimize import linear_sum_assignment
    b, n0, n1 = both.shape
    cost = -both.clone()
    cost[unmatched0] = 1e6
    cost[ignore0] = 1e6
    cost = cost.transpose(1, 2)
    cost[unmatched1] = 1e6
    cost[ignore1] = 1e6
    cost = cost.transpose(1, 2)
    if both.numel() == 0:
        none = torch.zeros(b, 0).to(mask_close)
        return both.new_zeros(both.shape, dtype=torch.bool), none, none
    pairs = torch.tensor(np.array([linear_sum_assignment(c) for c in cost.detach().cpu().numpy()])).to(both)
    positive = both.new_zeros(both.shape, dtype=torch.bool)
    positive[torch.arange(b)[:, None].repeat(1, pairs.shape[-1]).flatten(), pairs[:, 0].flatten(), pairs[:, 1].flatten()] = True
    m0 = pairs.new_full((b, n0), UNMATCHED_FEATURE, dtype=torch.long)
    m0.scatter_(-1, pairs[:, 0], pairs[:, 1])
    m1 = pairs.new_full((b, n1), UNMATCHED_FEATURE, dtype=torch.long)
    m1.scatter_(-1, pairs[:, 1], pairs[:, 0])
    positive = positive & mask_close
    positive[unmatched0] = False
    positive[ignore0] = False
    positive = positive.transpose(1, 2)
    positive[unmatched1] = False
    positive[ignore1] = False
    positive = positive.transpose(1, 2)
    m0[~positive.any(-1)] = UNMATCHED_FEATURE
    m0[unmatched0] = UNMATCHED_FEATURE
    m0[ignore0] = IGNORE_FEATURE
    m1[~positive.any(-2)] = UNMATCHED_FEATURE
    m1[unmatched1] = UNMATCHED_FEATURE
    m1[ignore1] = IGNORE_FEATURE
    return positive, m0, m1


@torch.no_grad()
def gt_line_matches_from_homography(pred_lines0, pred_lines1, valid_lines0, valid_lines1, shape0, shape1, H,
                                    npts=50, dist_th=5, overlap_th=0.2, min_visibility_th=0.2):
    """Line ground truth under a homography (gluefactory/geometry/gt_generation.py:409-558): sample npts points
    on every segment, warp them, count per segment pair how many warped samples fall on the other segment (both
    ways), keep pairs whose two overlaps exceed overlap_th, solve the one-to-one assignment with the Hungarian
    method on CPU (scipy, as the reference) and label the rest unmatched (-1) / ignored (-2, invalid lines).
    Returns (assignment [B,L0,L1] bool, matches0 [B,L0], matches1 [B,L1])."""
    h0, w0 = shape0[-2:]
    h1, w1 = shape1[-2:]
    l0, l1 = _segments(pred_lines0.clone()), _segments(pred_lines1.clone())
    b, n0, _ = l0.shape
    n1 = l1.shape[1]
    l0 = torch.min(torch.max(l0, torch.zeros_like(l0)), l0.new_tensor([w0 - 1, h0 - 1, w0 - 1, h0 - 1], dtype=torch.float))
    l1 = torch.min(torch.max(l1, torch.zeros_like(l1)), l1.new_tensor([w1 - 1, h1 - 1, w1 - 1, h1 - 1], dtype=torch.float))

    p0_in1 = warp_points(_line_samples(l0, npts).reshape(b, n0 * npts, 2), H, inverse=False).reshape(b, n0, npts, 2)
    p1_in0 = warp_points(_line_samples(l1, npts).reshape(b, n1 * npts, 2), H, inverse=True).reshape(b, n1, npts, 2)
    out_of0 = _mostly_outside(p1_in0, w0, h0, min_visibility_th)                         # [B,L1]
    out_of1 = _mostly_outside(p0_in1, w1, h1, min_visibility_th)                         # [B,L0]
    c0 = _close_point_counts(l0, p1_in0, dist_th)            # [B,L0,L1]
    c1t = _close_point_counts(l1, p0_in1, dist_th).transpose(-1, -2)
    both = c0 * c1t
    mask_close = (c1t > npts * overlap_th) & (c0 > npts * overlap_th) & ~out_of0.unsqueeze(1) & ~out_of1.unsqueeze(-1)
    unmatched0 = torch.all(~mask_close, dim=2) | out_of1
    unmatched1 = torch.all(~mask_close, dim=1) | out_of0
    ignore0, ignore1 = ~valid_lines0, ~valid_lines1

    return _assign_lines(both, mask_close, unmatched0, unmatched1, ignore0, ignore1)


@torch.no_grad()
def gt_line_matches_from_pose_depth(pred_lines0, pred_lines1, valid_lines0, valid_lines1, data, npts=50, dist_th=5,
                                    overlap_th=0.2, min_visibility_th=0.5):
    """Line ground truth from depth maps and the relative pose (gluefactory/geometry/gt_generation.py:207-407; the line
    branch of depth_matcher.py:70-87): sample npts points on every segment (clamped into its depth map), lift them through
    the sampled depth and reproject into the other view; a pair of segments is close when enough VISIBLE reprojected samples
    of each fall on the other (more than overlap_th of that segment's visible samples); a segment whose reprojection is
    mostly outside the other image, or that is close to nothing, is unmatched (-1); one with too few valid depth samples, or
    flagged invalid, is ignored (-2).  Returns (assignment [B,L0,L1] bool, matches0 [B,L0], matches1 [B,L1]).
    (The reference reads the inverse pose as `data.get(data["T_1to0"], data["T_0to1"].inv())`, i.e. always the inverse of
    T_0to1: so does this.)
    PARITY IS WITH THE REFERENCE'S CPU PATH (what tests/golden/gt_lines_depth.npz pins bit for bit).  On CUDA the reference
    additionally rounds the centred samples and the rotation of its perpendicular-distance test to fp16
    (gt_generation.py:193-195: `.half()` when `is_cuda`), so labels of segment pairs within fp16 rounding (~1e-3 relative) of
    `dist_th` / of a segment end can differ from a reference run ON A GPU; only the segment length is kept fp16-rounded here, as
    the reference's CPU path keeps it.  The reference's early return for inputs without elements is covered by the n0 == 0 /
    n1 == 0 branch below."""
    from .geometry import project, sample_depth
    b, n0, n1 = pred_lines0.shape[0], pred_lines0.shape[1], pred_lines1.shape[1]
    if n0 == 0 or n1 == 0:
        dev = pred_lines0.device
        return (torch.zeros((b, n0, n1), dtype=torch.bool, device=dev), torch.full((b, n0), -1, device=dev),
                torch.full((b, n1), -1, device=dev))
    l0, l1 = _segments(pred_lines0.clone()), _segments(pred_lines1.clone())
    depth0, depth1 = data["view0"]["depth"], data["view1"]["depth"]
    h0, w0 = depth0[0].shape
    h1, w1 = depth1[0].shape
    l0 = torch.min(torch.max(l0, torch.zeros_like(l0)), l0.new_tensor([w0 - 1, h0 - 1, w0 - 1, h0 - 1], dtype=torch.float))
    l1 = torch.min(torch.max(l1, torch.zeros_like(l1)), l1.new_tensor([w1 - 1, h1 - 1, w1 - 1, h1 - 1], dtype=torch.float))
    pts0 = _line_samples(l0, npts).reshape(b, n0 * npts, 2)
    pts1 = _line_samples(l1, npts).reshape(b, n1 * npts, 2)
    d0, valid0 = sample_depth(pts0, depth0)
    d1, valid1 = sample_depth(pts1, depth1)
    cam0, cam1, T_0to1 = data["view0"]["camera"], data["view1"]["camera"], data["T_0to1"]
    p0_in1, visible0 = project(pts0, d0, depth1, cam0, cam1, T_0to1, valid0)
    p1_in0, visible1 = project(pts1, d1, depth0, cam1, cam0, T_0to1.inv(), valid1)
    h0, w0 = data["view0"]["image"].shape[-2:]
    h1, w1 = data["view1"]["image"].shape[-2:]
    p0_in1, p1_in0 = p0_in1.reshape(b, n0, npts, 2), p1_in0.reshape(b, n1, npts, 2)
    out_of0 = _mostly_outside(p1_in0, w0, h0, min_visibility_th)                         # [B,L1]
    out_of1 = _mostly_outside(p0_in1, w1, h1, min_visibility_th)                         # [B,L0]
    vis0, vis1 = visible0.reshape(b, n0, npts), visible1.reshape(b, n1, npts)
    c0 = _close_point_counts(l0, p1_in0, dist_th, vis1)                                  # [B,L0,L1]
    c1t = _close_point_counts(l1, p0_in1, dist_th, vis0).transpose(-1, -2)
    both = c0 * c1t
    mask_close = (c1t > vis0.float().sum(-1)[:, :, None] * overlap_th) & (c0 > vis1.float().sum(-1)[:, None] * overlap_th)
    unmatched0 = torch.all(~mask_close, dim=2) | out_of1
    unmatched1 = torch.all(~mask_close, dim=1) | out_of0
    ignore0 = (valid0.reshape(b, n0, npts).float().mean(-1) < min_visibility_th) | ~valid_lines0
    ignore1 = (valid1.reshape(b, n1, npts).float().mean(-1) < min_visibility_th) | ~valid_lines1
    return _assign_lines(both, mask_close, unmatched0, unmatched1, ignore0, ignore1)
