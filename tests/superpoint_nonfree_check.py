"""Shared checker: glue_factory_amd.extractors.superpoint against the reference-generated golden
(tests/golden/superpoint_nonfree.npz, oracle/gen_golden.py gen_superpoint_nonfree)."""
import numpy as np
import torch

from conftest import load_golden

CASES = {"legacy": {"nms_radius": 3}, "fixed": {"nms_radius": 4, "legacy_sampling": False},
         "refine": {"nms_radius": 3, "refinement_radius": 2}}


def check(device, autocast=False, tol=1e-4):
    from glue_factory_amd.extractors.superpoint import SuperPoint
    z = load_golden("superpoint_nonfree")
    seed = int(z["seed"])
    image = torch.from_numpy(z["image"]).to(device)
    size = torch.from_numpy(z["image_size"]).to(device)
    worst = 0.0
    for cname, extra in CASES.items():
        conf = {"max_num_keypoints": 100, "force_num_keypoints": True, "detection_threshold": 0.0, **extra}
        torch.manual_seed(seed)
        model = SuperPoint(conf)
        model.convPb.weight.data.mul_(40.0)      # (as in oracle/gen_golden.py: spreads the detector logits)
        model = model.to(device).eval()
        for tag, data in (("plain", {"image": image}), ("sized", {"image": image, "image_size": size})):
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                pred = model(dict(data))
            for b in range(image.shape[0]):
                rk, rs, rd = (z[f"{cname}.{tag}.{k}"][b] for k in ("keypoints", "keypoint_scores", "descriptors"))
                gk, gs, gd = (pred[k][b].float().cpu().numpy() for k in ("keypoints", "keypoint_scores", "descriptors"))
                valid = rs > 0                           # (padding keypoints are random in both)
                # a top-k that cuts through a group of (nearly) tied scores may keep any of them: compared above the cut
                valid &= rs > rs[valid].min() * (1 + 10 * tol)
                assert valid.sum() >= 16, valid.sum()
                if not autocast:
                    # rows are matched by POSITION (the order inside a group of tied scores is unspecified in any top-k)
                    d2 = ((rk[valid][:, None, :] - gk[None, :, :]) ** 2).sum(-1)
                    j = d2.argmin(1)
                    assert float(np.sqrt(d2[np.arange(len(j)), j].max())) <= (2e-3 if cname == "refine" else 0.0), f"{cname}.{tag}: keypoints"
                    assert len(set(j.tolist())) == len(j)
                    np.testing.assert_allclose(gs[j], rs[valid], rtol=tol, atol=tol * 1e-2, err_msg=f"{cname}.{tag} scores")
                    np.testing.assert_allclose(gd[j], rd[valid], rtol=tol, atol=5 * tol, err_msg=f"{cname}.{tag} descriptors")
                else:           # bf16 convolutions: the detections move; report the overlap of the keypoint sets
                    a = {tuple(np.round(k).astype(int)) for k in gk[gs > 0]}
                    r = {tuple(np.round(k).astype(int)) for k in rk[valid]}
                    worst = max(worst, 1.0 - len(a & r) / max(len(r), 1))
    return worst
