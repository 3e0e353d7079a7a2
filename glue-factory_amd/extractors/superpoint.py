"""SuperPoint (original MagicLeap layout) extractor -- drop-in for ``gluefactory_nonfree.superpoint``.

The extractor the N=2048 LightGlue training yaml names (configs/superpoint+lightglue_megadepth.yaml:25-31).  Same
``default_conf`` keys, ``required_data_keys`` and ``state_dict`` names / shapes as gluefactory_nonfree/superpoint.py:152-350
(``conv1a`` ... ``conv4b``, ``convPa/convPb``, ``convDa/convDb``; no BatchNorm), so ``superpoint_v1.pth`` loads unchanged --
from a LOCAL file given as ``weights`` (the reference downloads it; there is no network here: ``weights: null`` keeps the
seeded random initialisation).

The network is the same VGG topology as the open re-implementation without its BatchNorm layers, so everything runs on the
kernels of ``superpoint_open.py`` with an identity scale / shift: the first block and the three 64 -> 64 blocks as fused HIP
kernels (``gf_conv1_bias_act_bn``, ``gf_conv3x3_c64``), library convolutions + one HIP tail pass for the 128 / 256-channel
blocks, ``gf_detector_scores`` (softmax over the 65 channels + unfolding), register-resident NMS into candidate lists, the
own top-k (``gf_topk_candidates``) and the descriptor sampler.  What differs from the open variant is post-processing:
  * ``max_num_keypoints`` = -1 means "all" (only possible for a batch of one), ``max_num_keypoints_val`` replaces it in eval
    mode (superpoint.py:262-266);
  * borders are removed relative to ``data["image_size"]`` when the images are smaller than the batch tensor (:236-244);
  * ``refinement_radius`` > 0: soft-argmax refinement of the keypoints on the pre-NMS score map (:92-108, :290-293);
  * ``legacy_sampling`` (the default): the original, slightly shifted descriptor sampling with ``align_corners=True``
    (:112-127) -- stock ``grid_sample``; ``legacy_sampling: false`` is the open variant's sampler without its +0.5 (:132-143);
  * ``randomize_keypoints_training`` (:268-277): in training mode the keypoints are drawn with probabilities proportional to
    their scores (one batched ``torch.multinomial`` over the NMS candidate lists) instead of the k best.
"""
from pathlib import Path

import torch
import torch.nn.functional as F
from torch import nn

from .superpoint_open import SuperPoint as _SuperPointOpen


class _Block:
    """conv (+ ReLU) seen through the block interface of superpoint_open.py (.conv, .activation, .bn, callable)."""
    bn = None

    def __init__(self, conv, relu):
        self.conv = conv
        self.activation = nn.ReLU() if relu else nn.Identity()

    def __call__(self, x):
        return self.activation(self.conv(x))


class SuperPoint(_SuperPointOpen):
    default_conf = {
        "has_detector": True,
        "has_descriptor": True,
        "descriptor_dim": 256,
        "sparse_outputs": True,
        "dense_outputs": False,
        "nms_radius": 4,
        "refinement_radius": 0,
        "detection_threshold": 0.005,
        "max_num_keypoints": -1,
        "max_num_keypoints_val": None,
        "force_num_keypoints": False,
        "randomize_keypoints_training": False,
        "remove_borders": 4,
        "legacy_sampling": True,
        "weights": None,            # ours: local path of superpoint_v1.pth (the reference fetches it from the network)
    }
    required_data_keys = ["image"]

    def _init(self, conf):
        if not (conf.has_detector and conf.has_descriptor and conf.sparse_outputs):
            raise NotImplementedError("glue_factory_amd.extractors.superpoint: sparse outputs of detector + descriptor only")
        self.stride = 8
        self.register_buffer("_gray", torch.tensor([0.299, 0.587, 0.114]).view(1, 3, 1, 1), persistent=False)
        c1, c2, c3, c4, c5 = 64, 64, 128, 128, 256
        mk = lambda ci, co, k=3: nn.Conv2d(ci, co, kernel_size=k, stride=1, padding=k // 2)       # noqa: E731
        self.conv1a, self.conv1b = mk(1, c1), mk(c1, c1)
        self.conv2a, self.conv2b = mk(c1, c2), mk(c2, c2)
        self.conv3a, self.conv3b = mk(c2, c3), mk(c3, c3)
        self.conv4a, self.conv4b = mk(c3, c4), mk(c4, c4)
        self.convPa, self.convPb = mk(c4, c5), mk(c5, 65, 1)
        self.convDa, self.convDb = mk(c4, c5), mk(c5, conf.descriptor_dim, 1)
        if conf.weights is not None:
            path = Path(conf.weights)
            if not path.exists():
                raise FileNotFoundError(f"SuperPoint weights '{conf.weights}' not found locally (no network on this target)")
            # strict=False like the reference (superpoint.py:198-200) -- but a file that fits none of the layers must not
            # silently leave the random initialisation in place
            res = self.load_state_dict(torch.load(str(path), map_location="cpu"), strict=False)
            if res.missing_keys or res.unexpected_keys:
                import logging
                logging.getLogger(__name__).warning("SuperPoint weights '%s': missing keys %s, unexpected keys %s",
                                                    conf.weights, list(res.missing_keys), list(res.unexpected_keys))
            if len(res.missing_keys) == len(self.state_dict()):
                raise RuntimeError(f"SuperPoint weights '{conf.weights}' match none of the model's parameters")

    # ------------------------------------------------------------------ layout: the open variant's names, no BatchNorm
    def _layout(self):
        lay = self.__dict__.get("_lay")
        if lay is None:
            b = lambda conv, relu=True: _Block(conv, relu)                                          # noqa: E731
            lay = self.__dict__["_lay"] = {
                "stages": [([("backbone.0.0", b(self.conv1a)), ("backbone.0.1", b(self.conv1b))], True),
                           ([("backbone.1.0", b(self.conv2a)), ("backbone.1.1", b(self.conv2b))], True),
                           ([("backbone.2.0", b(self.conv3a)), ("backbone.2.1", b(self.conv3b))], True),
                           ([("backbone.3.0", b(self.conv4a)), ("backbone.3.1", b(self.conv4b))], False)],
                "det": [("detector.0", b(self.convPa)), ("detector.1", b(self.convPb, False))],
                "desc": [("descriptor.0", b(self.convDa)), ("descriptor.1", b(self.convDb, False))]}
        return lay

    def _named_blocks(self):
        lay = self._layout()
        return [nb for blocks, _ in lay["stages"] for nb in blocks] + lay["det"] + lay["desc"]

    def _stages(self):
        return self._layout()["stages"]

    def _heads(self):
        lay = self._layout()
        return lay["det"], lay["desc"]

    def _dense_unfused(self, image):
        x = image
        for blocks, pool in self._stages():
            for _, blk in blocks:
                x = blk(x)
            if pool:
                x = F.max_pool2d(x, 2, 2)
        (_, pa), (_, pb) = self._layout()["det"]
        (_, da), (_, db) = self._layout()["desc"]
        return pb(pa(x)), db(da(x))

    # ------------------------------------------------------------------ post-processing that differs from the open variant
    def _max_keypoints(self):
        k = self.conf.max_num_keypoints
        if not self.training and self.conf.max_num_keypoints_val is not None:
            k = self.conf.max_num_keypoints_val
        return None if k is None or k <= 0 else int(k)

    def _sample_keypoints(self, cand, scores, k):
        """``randomize_keypoints_training`` (superpoint.py:84-89, 273-282): in training mode the k keypoints are DRAWN without
        replacement with probabilities proportional to their scores instead of taking the k best; an image with fewer than k
        detections keeps them all.  Batched: one ``torch.multinomial`` over the NMS candidate lists; entries that are not
        detections (score <= detection_threshold, empty slots) carry a weight of 1e-20, so they are only drawn once an image's
        detections are exhausted -- and are then not `valid` and get padded / dropped exactly like the surplus of a top-k."""
        if not (self.conf.randomize_keypoints_training and self.training):
            return None
        if cand is not None:
            s, idx = cand[0], cand[1].long()
        else:
            s, idx = scores.reshape(scores.shape[0], -1), None
        w = torch.where(s > self.conf.detection_threshold, s, torch.full_like(s, 1e-20))
        j = torch.multinomial(w, min(k, w.shape[1]), replacement=False)
        return s.gather(1, j), (j if idx is None else idx.gather(1, j))

    def _border_limits(self, data):
        """superpoint.py:236-244 removes the right / bottom border relative to `image_size` (images smaller than the batch
        tensor): per-image first dropped column / row, as device tensors [B, 1] -- compared elementwise, never read back."""
        size = data.get("image_size")
        pad = int(self.conf.remove_borders or 0)
        if size is None or not pad:
            return None
        size = size.long()
        return (size[:, 0:1] - pad, size[:, 1:2] - pad)

    def _refine(self, keypoints, dense_scores):
        """Soft-argmax refinement (superpoint.py:92-108): offset = sum of (dx, dy) weighted by the scores of the
        (2r+1)^2 window / their sum, on the PRE-NMS score map, for every selected keypoint."""
        r = int(self.conf.refinement_radius or 0)
        if r <= 0:
            return keypoints
        sc = dense_scores.float()[:, None]
        width = 2 * r + 1
        total = F.avg_pool2d(sc, width, 1, r, divisor_override=1)
        ar = torch.arange(-r, r + 1, device=sc.device, dtype=sc.dtype)
        kx = ar[None].expand(width, -1)[None, None].contiguous()
        dx = F.conv2d(sc, kx, padding=r)
        dy = F.conv2d(sc, kx.transpose(2, 3).contiguous(), padding=r)
        b, _, H, W = sc.shape
        flat = (keypoints[..., 1].long().clamp(0, H - 1) * W + keypoints[..., 0].long().clamp(0, W - 1))      # y * W + x
        pick = lambda t: t.reshape(b, -1).gather(1, flat)                                                      # noqa: E731
        den = pick(total)
        return keypoints + torch.stack([pick(dx) / den, pick(dy) / den], -1)

    def _sample(self, keypoints, desc_map, dense, fused, s, shift=0.0):
        if not self.conf.legacy_sampling:          # corrected sampling = the open variant's without its +0.5 (:132-143)
            return super()._sample(keypoints, desc_map, dense, fused, s, shift=-0.5)
        # legacy sampling (:112-127): keypoints - s/2 + 0.5 over (w s - s/2 - 0.5), align_corners=True, on the normalised map
        d = dense()
        b, c, h, w = d.shape
        kp = keypoints - s / 2 + 0.5
        kp = torch.stack([kp[..., 0] / (w * s - s / 2 - 0.5), kp[..., 1] / (h * s - s / 2 - 0.5)], -1)      # (no host tensor: capturable)
        kp = kp * 2 - 1
        out = F.grid_sample(d, kp.view(b, 1, -1, 2), mode="bilinear", align_corners=True)
        return F.normalize(out.reshape(b, c, -1), p=2, dim=1).transpose(-1, -2)

    def loss(self, pred, data):
        raise NotImplementedError

    def metrics(self, pred, data):
        raise NotImplementedError


__main_model__ = SuperPoint
