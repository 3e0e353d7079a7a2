"""Ground-truth assignment from a homography (no-grad), the step that feeds the loss.

Mirrors the behaviour of gluefactory/geometry/gt_generation.py:109-161
(gt_matches_from_homography) and gluefactory/geometry/homography.py:161-180
(warp_points_torch): warp both ways, symmetric squared distance, mutual nearest
neighbour within ``pos_th`` px -> positive; nearest warped neighbour farther than
``neg_th`` px -> unmatched (-1); everything else ignored (-2).

Runs on whatever device the keypoints live on (stock torch ops; the fused HIP
nearest-neighbour kernel is the "next" row of SURVEY.md §8f).
"""
import torch

IGNORE_FEATURE = -2
UNMATCHED_FEATURE = -1


def warp_points(points, H, inverse=False, eps=1e-5):
    """points [B,N,2], H [B,3,3] (or [3,3]) -> H (or H^-1) applied in homogeneous coords."""
    Hm = torch.linalg.inv(H) if inverse else H
    if Hm.dim() == 2:
        Hm = Hm[None]
    ones = torch.ones_like(points[..., :1])
    ph = torch.cat([points, ones], -1)
    w = torch.einsum("bnj,bij->bni", ph, Hm)
    return w[..., :2] / (w[..., 2:] + eps)


@torch.no_grad()
def gt_matches_from_homography_fused(kp0, kp1, H, pos_th=3.0, neg_th=3.0, with_reward=False):
    """Same labels through the HIP nearest-neighbour kernel (gf_gt_nn): no [B,M,N] fp32 tensor is
    built; the dense boolean ``assignment`` the plugin contract asks for is a zero-fill + scatter.
    ``reward`` (dense, unused by the matcher losses) is only produced on request (stock torch)."""
    from . import lib as _lib
    b, m = kp0.shape[:2]
    n = kp1.shape[1]
    kp0, kp1 = kp0.float().contiguous(), kp1.float().contiguous()
    kp0_1 = warp_points(kp0, H.float(), inverse=False).contiguous()
    kp1_0 = warp_points(kp1, H.float(), inverse=True).contiguous()
    dev = kp0.device
    L = _lib.load()
    st = torch.cuda.current_stream().cuda_stream

    def nn(own, own_w, oth, oth_w):
        no = own.shape[1]
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
def gt_matches_from_homography(kp0, kp1, H, pos_th=3.0, neg_th=3.0):
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
