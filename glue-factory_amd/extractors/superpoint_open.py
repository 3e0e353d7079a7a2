"""SuperPoint (open re-implementation) extractor — stock PyTorch-ROCm convolutions by design.

north_star keeps the VGG backbone and the detector / descriptor heads on stock PyTorch conv
(MIOpen); this module mirrors the interface, defaults and ``state_dict`` layout of
gluefactory/models/extractors/superpoint_open.py:78-216 so checkpoints interchange:
``{"image"} -> {"keypoints" (+0.5), "keypoint_scores", "descriptors" [B,N,256]}``.

On a HIP device, in eval mode with frozen weights, the memory-bound tails run on two HIP kernels
(csrc/extractor.hip): every VGGBlock's bias + ReLU + BatchNorm(eval) [+ 2x2 max-pool] is ONE pass over a
channels-last activation (the convolution itself is the stock library call, bias-free), and simple_nms +
border removal is one LDS-tiled kernel instead of five max-pools and a dozen elementwise passes.

Differences in HOW (not what): the per-image python loop over ``torch.where`` / ``topk``
(superpoint_open.py:154-176) is a single batched top-k over the flattened score map, so the
forward has no host synchronisation; images with fewer than ``max_num_keypoints`` detections are
padded like the reference's ``pad_and_stack(mode="random_c")`` (uniform inside the detected
keypoints' bounding box, score 0).  ``weights=None`` gives a seeded random initialisation
(pretrained files cannot be downloaded on this target); a local path is loaded as usual.
"""
from collections import OrderedDict
from pathlib import Path

# (MIOpen's find mode for the stock convolutions is chosen in the package __init__, before the first convolution.)

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..base_model import BaseModel, BatchedExtractionUnsupported


WIDE64 = True       # the 64 -> 128 block on gf_conv3x3_c64_ld (two launches) instead of the library convolution + tail pass (A/B switch)


def sample_descriptors(keypoints, descriptors, s=8):
    """Bilinear sampling of the dense descriptor map at keypoint locations, then L2 norm."""
    b, c, h, w = descriptors.shape
    kp = (keypoints + 0.5) / (keypoints.new_tensor([w, h]) * s) * 2 - 1
    d = F.grid_sample(descriptors, kp.view(b, 1, -1, 2), mode="bilinear", align_corners=False)
    return F.normalize(d.reshape(b, c, -1), p=2, dim=1)


def batched_nms(scores, nms_radius):
    assert nms_radius >= 0

    def max_pool(x):
        return F.max_pool2d(x, kernel_size=nms_radius * 2 + 1, stride=1, padding=nms_radius)

    zeros = torch.zeros_like(scores)
    max_mask = scores == max_pool(scores)
    for _ in range(2):
        supp_mask = max_pool(max_mask.float()) > 0
        supp_scores = torch.where(supp_mask, zeros, scores)
        new_max_mask = supp_scores == max_pool(supp_scores)
        max_mask = max_mask | (new_max_mask & (~supp_mask))
    return torch.where(max_mask, scores, zeros)


class VGGBlock(nn.Sequential):
    def __init__(self, c_in, c_out, kernel_size, relu=True):
        super().__init__(OrderedDict([
            ("conv", nn.Conv2d(c_in, c_out, kernel_size=kernel_size, stride=1, padding=(kernel_size - 1) // 2)),
            ("activation", nn.ReLU(inplace=True) if relu else nn.Identity()),
            ("bn", nn.BatchNorm2d(c_out, eps=0.001)),
        ]))


class SuperPoint(BaseModel):
    default_conf = {
        "descriptor_dim": 256,
        "nms_radius": 4,
        "max_num_keypoints": None,
        "force_num_keypoints": False,
        "detection_threshold": 0.005,
        "remove_borders": 4,
        "channels": [64, 64, 128, 128, 256],
        "dense_outputs": None,
        "weights": None,
    }
    required_data_keys = ["image"]
    batchable_views = True      # forward works image by image on batched keys only (`image`, `image_size`): a frozen instance may see both views at once

    def _init(self, conf):
        self.stride = 2 ** (len(conf.channels) - 2)
        self.register_buffer("_gray", torch.tensor([0.299, 0.587, 0.114]).view(1, 3, 1, 1), persistent=False)
        channels = [1, *conf.channels[:-1]]
        backbone = []
        for i, c in enumerate(channels[1:], 1):
            layers = [VGGBlock(channels[i - 1], c, 3), VGGBlock(c, c, 3)]
            if i < len(channels) - 1:
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            backbone.append(nn.Sequential(*layers))
        self.backbone = nn.Sequential(*backbone)
        c = conf.channels[-1]
        self.detector = nn.Sequential(VGGBlock(channels[-1], c, 3), VGGBlock(c, self.stride ** 2 + 1, 1, relu=False))
        self.descriptor = nn.Sequential(VGGBlock(channels[-1], c, 3), VGGBlock(c, conf.descriptor_dim, 1, relu=False))
        if conf.weights is not None:
            path = Path(conf.weights)
            if not path.exists():
                raise FileNotFoundError(f"SuperPoint weights '{conf.weights}' not found locally")
            self.load_state_dict(torch.load(str(path), map_location="cpu"))

    # ------------------------------------------------------------------ network layout (overridden by the non-free variant)
    def _named_blocks(self):
        """(name, block) of every conv block; a block has .conv, .activation, .bn (or None) and is callable."""
        return [(n, m) for n, m in self.named_modules() if isinstance(m, VGGBlock)]

    def _stages(self):
        """Backbone as [( [(name, block), ...], followed_by_a_2x2_max_pool ), ...]."""
        out = []
        for si, stage in enumerate(self.backbone):
            blocks = [m for m in stage if isinstance(m, VGGBlock)]
            out.append(([(f"backbone.{si}.{bi}", m) for bi, m in enumerate(blocks)],
                        any(isinstance(m, nn.MaxPool2d) for m in stage)))
        return out

    def _heads(self):
        """((name, block) x 2 of the detector head, (name, block) x 2 of the descriptor head)."""
        return ([("detector.0", self.detector[0]), ("detector.1", self.detector[1])],
                [("descriptor.0", self.descriptor[0]), ("descriptor.1", self.descriptor[1])])

    def _dense_unfused(self, image):
        features = self.backbone(image)
        return self.detector(features), self.descriptor(features)

    # ------------------------------------------------------------------ fused inference path (HIP)
    def _use_fused(self, image):
        if not image.is_cuda or not 1 <= self.conf.nms_radius <= 4:
            return False
        # the fused kernels apply BatchNorm with its running statistics: every BatchNorm layer must be in eval mode -- the
        # whole module (`.eval()`), or a frozen extractor inside a training pipeline (`freeze_batch_normalization: true`)
        if any(m.training for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)):
            return False
        return not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))

    def _fused_params(self, dtype):
        """Per-block (channels-last weight in `dtype`, bias, BN scale, BN shift), rebuilt when a tensor changed."""
        tensors = list(self.parameters()) + list(self.buffers())
        key = (dtype, tuple(t._version for t in tensors), tuple(t.data_ptr() for t in tensors))
        # one entry PER DTYPE: a captured graph (TrainStep(graph=True) around a pipeline) points at the bf16 tensors of its
        # entry, and an fp32 call in between (an eval pass outside autocast) must not evict -- and free -- them
        caches = self.__dict__.setdefault("_fused_cache", {})
        cache = caches.get(dtype)
        if cache is not None and cache[0] == key:
            return cache[1]
        out = {}
        for name, blk in self._named_blocks():
            if True:
                bn = blk.bn
                if bn is None:        # (the non-free variant has no BatchNorm: identity scale / shift)
                    scale = torch.ones(blk.conv.out_channels, dtype=torch.float32, device=blk.conv.weight.device)
                    shift = torch.zeros_like(scale)
                else:
                    scale = (bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)).contiguous()
                    shift = (bn.bias.float() - bn.running_mean.float() * scale).contiguous()
                w = blk.conv.weight.detach().to(dtype).contiguous(memory_format=torch.channels_last)
                out[name] = (w, blk.conv.bias.detach().float().contiguous(), scale, shift)
                if (dtype == torch.bfloat16 and blk.conv.kernel_size == (1, 1) and not isinstance(blk.activation, nn.ReLU)
                        and blk.conv.in_channels in (256, 512) and blk.conv.out_channels % 256 == 0):
                    # 1 x 1 convolution without ReLU in front of the eval BatchNorm = ONE GEMM over the pixels:
                    # (W x + b) s + h = (s W) x + (s b + h), folded in fp32 before the single rounding of the weights
                    w2 = blk.conv.weight.detach().float().reshape(blk.conv.out_channels, -1)
                    out[name + "/gemm"] = ((w2 * scale[:, None]).to(dtype).contiguous(),
                                           (blk.conv.bias.detach().float() * scale + shift).contiguous())
                if dtype == torch.bfloat16 and tuple(blk.conv.weight.shape) == (64, 64, 3, 3):
                    # [tap][c_out][c_in] for the register-resident weights of gf_conv3x3_c64
                    out[name + "/taps"] = blk.conv.weight.detach().permute(2, 3, 0, 1).to(dtype).contiguous()
                if dtype == torch.bfloat16 and tuple(blk.conv.weight.shape) == (128, 64, 3, 3):
                    # 64 -> 128 (backbone.2.0): the same kernel once per half of the output channels (gf_conv3x3_c64_ld)
                    t = blk.conv.weight.detach().permute(2, 3, 0, 1).to(dtype)                  # [3, 3, 128, 64]
                    out[name + "/taps2"] = [(t[:, :, 64 * h:64 * h + 64].contiguous(), out[name][1][64 * h:64 * h + 64].contiguous(),
                                             scale[64 * h:64 * h + 64].contiguous(), shift[64 * h:64 * h + 64].contiguous())
                                            for h in range(2)]
        caches[dtype] = (key, out)
        return out

    def _fused_block(self, name, blk, x, params, pool=False):
        """Conv2d (library, bias-free) + one HIP pass for bias, ReLU, BatchNorm(eval) and the optional pool."""
        from .. import lib as _lib
        w, bias, scale, shift = params[name]
        c_out = w.shape[0]
        vec = 8 if x.dtype == torch.bfloat16 else 4
        if c_out % vec:                         # e.g. the 65-channel detector head: stock modules
            y = blk(x)
            return F.max_pool2d(y, 2, 2) if pool else y
        with torch.autocast(device_type="cuda", enabled=False):
            y = F.conv2d(x, w, None, 1, blk.conv.padding)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        b, _, h, wd = y.shape
        out = y if not pool else torch.empty((b, c_out, h // 2, wd // 2), dtype=y.dtype, device=y.device,
                                             memory_format=torch.channels_last)
        relu = isinstance(blk.activation, nn.ReLU)
        _lib.check(_lib.load().gf_bias_act_bn_nhwc(
            y.data_ptr(), out.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(), b, h, wd, c_out,
            int(relu), int(pool), 1 if y.dtype == torch.bfloat16 else 0, torch.cuda.current_stream().cuda_stream),
            "gf_bias_act_bn_nhwc")
        return out

    def _conv64_block(self, name, blk, x, params, pool):
        """64 -> 64 channel 3x3 block in bf16 (backbone.0.1, 1.0, 1.1): convolution, bias, ReLU, BatchNorm(eval) and
        the 2x2 max-pool in ONE HIP kernel (gf_conv3x3_c64, implicit GEMM with register-resident weights)."""
        from .. import lib as _lib
        _, bias, scale, shift = params[name]
        b, _, h, wd = x.shape
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        out = torch.empty((b, 64, h // 2, wd // 2) if pool else (b, 64, h, wd), dtype=x.dtype, device=x.device,
                          memory_format=torch.channels_last)
        _lib.check(_lib.load().gf_conv3x3_c64(
            x.data_ptr(), params[name + "/taps"].data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(),
            out.data_ptr(), b, h, wd, int(isinstance(blk.activation, nn.ReLU)), int(pool), 1,
            torch.cuda.current_stream().cuda_stream), "gf_conv3x3_c64")
        return out

    def _conv64_wide_block(self, name, blk, x, params, pool):
        """64 -> 128 channel 3x3 block in bf16 (backbone.2.0): gf_conv3x3_c64's kernel once per half of the output channels,
        each writing its 64-channel slice of the [B, h, w, 128] output with the tail (bias, ReLU, BatchNorm(eval), pool)
        fused -- instead of the library convolution + a tail pass over the output."""
        from .. import lib as _lib
        b, _, h, wd = x.shape
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        out = torch.empty((b, 128, h // 2, wd // 2) if pool else (b, 128, h, wd), dtype=x.dtype, device=x.device,
                          memory_format=torch.channels_last)
        relu = int(isinstance(blk.activation, nn.ReLU))
        for half, (taps, bias, scale, shift) in enumerate(params[name + "/taps2"]):
            _lib.check(_lib.load().gf_conv3x3_c64_ld(
                x.data_ptr(), taps.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                out.data_ptr() + 64 * half * out.element_size(), 128, b, h, wd, relu, int(pool), 1,
                torch.cuda.current_stream().cuda_stream), "gf_conv3x3_c64_ld")
        return out

    def _first_block(self, name, blk, x, params):
        """backbone.0.0 on a 1-channel image: 3x3 convolution + bias + ReLU + BatchNorm(eval) -> channels-last, ONE
        HIP kernel (gf_conv1_bias_act_bn): the extractor's largest activation is written once instead of being
        written by the library convolution and rewritten by the tail pass."""
        from .. import lib as _lib
        w, bias, scale, shift = params[name]
        b, _, h, wd = x.shape
        xin = x.reshape(b, h, wd).contiguous()
        out = torch.empty((b, 64, h, wd), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        _lib.check(_lib.load().gf_conv1_bias_act_bn(
            xin.data_ptr(), w.reshape(64, 9).contiguous().data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(),
            out.data_ptr(), b, h, wd, 64, int(isinstance(blk.activation, nn.ReLU)),
            1 if x.dtype == torch.bfloat16 else 0, torch.cuda.current_stream().cuda_stream), "gf_conv1_bias_act_bn")
        return out

    def _detector_scores(self, name, blk, x, params):
        """detector.1 (library 1x1 convolution, bias-free) + ONE HIP pass for its bias / BatchNorm(eval), the softmax over
        the 65 channels and the unfolding of the 64 cell probabilities into the full-resolution score map
        (gf_detector_scores; superpoint_open.py:105-108, 141-147)."""
        from .. import lib as _lib
        w, bias, scale, shift = params[name]
        with torch.autocast(device_type="cuda", enabled=False):
            y = F.conv2d(x, w, None, 1, blk.conv.padding)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        b, _, h, wd = y.shape
        scores = torch.empty((b, h * 8, wd * 8), dtype=torch.float32, device=y.device)
        _lib.check(_lib.load().gf_detector_scores(
            y.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(), scores.data_ptr(), b, h, wd,
            int(isinstance(blk.activation, nn.ReLU)), 1 if y.dtype == torch.bfloat16 else 0,
            torch.cuda.current_stream().cuda_stream), "gf_detector_scores")
        return scores

    def _fused_features(self, image):
        dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
        if dtype not in (torch.bfloat16, torch.float32):
            dtype = torch.float32
        params = self._fused_params(dtype)
        x = image.to(dtype).contiguous()
        if x.shape[1] == 1:     # one channel: NCHW and NHWC are the same bytes; give it the NHWC strides so the
            b_, _, h_, w_ = x.shape    # first convolution already answers channels-last
            x = x.as_strided(x.shape, (h_ * w_, 1, w_, 1))
        else:
            x = x.contiguous(memory_format=torch.channels_last)
        for si, (blocks, has_pool) in enumerate(self._stages()):
            for bi, (name, blk) in enumerate(blocks):
                pool = has_pool and bi == len(blocks) - 1
                if si == 0 and bi == 0 and not pool and x.shape[1] == 1 and blk.conv.out_channels == 64 \
                        and blk.conv.kernel_size == (3, 3):
                    x = self._first_block(name, blk, x, params)      # conv + tail in one kernel
                elif name + "/taps" in params and x.shape[2] % 8 == 0 and x.shape[3] % 32 == 0 \
                        and x.shape[2] * x.shape[3] * 128 < 2 ** 31 and blk.conv.padding == (1, 1) and blk.conv.stride == (1, 1):
                    x = self._conv64_block(name, blk, x, params, pool)
                elif WIDE64 and name + "/taps2" in params and x.shape[1] == 64 and x.shape[2] % 8 == 0 and x.shape[3] % 32 == 0 \
                        and x.shape[2] * x.shape[3] * 256 < 2 ** 31 and blk.conv.padding == (1, 1) and blk.conv.stride == (1, 1):
                    x = self._conv64_wide_block(name, blk, x, params, pool)
                else:
                    x = self._fused_block(name, blk, x, params, pool=pool)
        (d0n, d0), (d1n, d1) = self._heads()[0]
        (e0n, e0), (e1n, e1) = self._heads()[1]
        det = self._fused_block(d0n, d0, x, params)
        if d1.conv.out_channels == self.stride ** 2 + 1 == 65:
            det = self._detector_scores(d1n, d1, det, params)     # -> the [B, 8h, 8w] score map
        else:
            det = self._fused_block(d1n, d1, det, params)
        desc = self._fused_block(e0n, e0, x, params)
        if e1n + "/gemm" in params and desc.is_contiguous(memory_format=torch.channels_last):
            from .. import ops
            w2, b2 = params[e1n + "/gemm"]                    # descriptor.1 (256 -> 256, 1 x 1, no ReLU) + its BatchNorm
            bb, cc, hh, ww = desc.shape
            y = ops.gemm(desc.permute(0, 2, 3, 1).reshape(-1, cc), w2, b2)
            desc = y.view(bb, hh, ww, w2.shape[0]).permute(0, 3, 1, 2)          # channels-last [B, C, h, w]
        else:
            desc = self._fused_block(e1n, e1, desc, params)
        return det, desc

    def _forward(self, data):
        conf = self.conf
        image = data["image"]
        if image.shape[1] == 3:       # (the weights live in a buffer: a host -> device copy per call cannot be captured in a graph)
            image = (image * self._gray.to(image.dtype)).sum(1, keepdim=True)
        fused = self._use_fused(image)
        if fused:
            det, desc_map = self._fused_features(image)
        else:
            det, desc_map = self._dense_unfused(image)
        def dense():          # per-pixel normalised map; the fused sampler normalises the corners itself
            return F.normalize(desc_map.float(), p=2, dim=1)

        s = self.stride
        if det.dim() == 3:                  # the fused detector tail already produced the score map
            scores = det
            b = scores.shape[0]
            h, w = scores.shape[1] // s, scores.shape[2] // s
        else:
            scores = F.softmax(det.float(), 1)[:, :-1]
            b, _, h, w = scores.shape
            scores = scores.permute(0, 2, 3, 1).reshape(b, h, w, s, s).permute(0, 1, 3, 2, 4).reshape(b, h * s, w * s)
        k = self._max_keypoints()
        H, W = h * s, w * s
        cand = None
        dense_scores = scores                     # (pre-NMS map: the non-free variant's soft-argmax refinement reads it)
        limits = self._border_limits(data)        # None, or per-image (x_max, y_max) [B,1] beyond which keypoints are dropped
        if fused:
            from .. import lib as _lib
            scores = scores.contiguous()
            r = int(conf.nms_radius)
            cap = _lib.load().gf_nms_candidates_cap(H, W, r)        # a fixed segment per kernel tile
            if k is not None and conf.detection_threshold >= 0 and k <= cap and H * W < 2 ** 31:
                # NMS straight into per-image candidate lists (positive maxima outside the border): the top-k below then
                # sorts ~H W / 49 entries instead of H W, and the dense map is never written
                cand_s = torch.full((b, cap), -1.0, dtype=torch.float32, device=scores.device)
                cand_i = torch.zeros((b, cap), dtype=torch.int32, device=scores.device)
                _lib.check(_lib.load().gf_nms_candidates(scores.data_ptr(), cand_s.data_ptr(), cand_i.data_ptr(), b, H, W, r,
                                                         int(conf.remove_borders or 0),
                                                         torch.cuda.current_stream().cuda_stream), "gf_nms_candidates")
                if limits is not None:      # (after the NMS, like the reference's border writes; elementwise, no host read)
                    ci = cand_i.long()
                    cand_s = torch.where((ci % W >= limits[0]) | (ci // W >= limits[1]), torch.full_like(cand_s, -1.0), cand_s)
                cand = (cand_s, cand_i)
            else:
                nms = torch.empty_like(scores)
                _lib.check(_lib.load().gf_nms_scores(scores.data_ptr(), nms.data_ptr(), b, H, W, r,
                                                     int(conf.remove_borders or 0), torch.cuda.current_stream().cuda_stream),
                           "gf_nms_scores")
                scores = nms
                if limits is not None:
                    scores = self._apply_limits(scores, limits)
        else:
            scores = batched_nms(scores.float(), conf.nms_radius)
            if conf.remove_borders:
                pad = conf.remove_borders
                scores[:, :pad] = -1
                scores[:, :, :pad] = -1
                scores[:, -pad:] = -1
                scores[:, :, -pad:] = -1
            if limits is not None:
                scores = self._apply_limits(scores, limits)
        if k is None:
            if b != 1:
                raise BatchedExtractionUnsupported("max_num_keypoints is required for batched extraction")
            idx = torch.where(scores[0] > conf.detection_threshold)
            keypoints = torch.stack(idx[::-1], -1).float()[None]
            kscores = scores[0][idx][None]
            keypoints = self._refine(keypoints, dense_scores)      # (the non-free variant refines in every case, superpoint.py:290-293)
        else:
            picked = self._sample_keypoints(cand, scores, k)       # hook: None = the k highest scores
            if picked is not None:
                kscores, ind = picked
            elif cand is not None and k <= min(4096, cand[0].shape[1]):
                # own top-k over the candidate lists (csrc/topk.hip): sorted scores + pixel indices in one launch, and
                # -- unlike torch.topk here, which faults on the second replay of a captured extractor tail -- safely capturable
                from .. import lib as _lib
                kscores = torch.empty((b, k), dtype=torch.float32, device=scores.device)
                ind = torch.empty((b, k), dtype=torch.int64, device=scores.device)
                _lib.check(_lib.load().gf_topk_candidates(cand[0].data_ptr(), cand[1].data_ptr(), kscores.data_ptr(),
                                                          ind.data_ptr(), b, cand[0].shape[1], k,
                                                          torch.cuda.current_stream().cuda_stream), "gf_topk_candidates")
            elif cand is not None:
                kscores, j = torch.topk(cand[0], min(k, cand[0].shape[1]), dim=1, sorted=True)
                ind = cand[1].gather(1, j).long()       # (unfilled entries: score -1, index 0 -- never valid below)
            else:
                flat = scores.reshape(b, -1)
                kscores, ind = torch.topk(flat, min(k, flat.shape[1]), dim=1, sorted=True)
            keypoints = torch.stack([ind % W, ind // W], -1).float()
            valid = kscores > conf.detection_threshold
            keypoints = self._refine(keypoints, dense_scores)
            if conf.force_num_keypoints:
                # pad like pad_and_stack(mode="random_c"): uniform in the detections' bounding box
                big = torch.full_like(keypoints, float("inf"))
                lo = torch.where(valid[..., None], keypoints, big).amin(1, keepdim=True)
                hi = torch.where(valid[..., None], keypoints, -big).amax(1, keepdim=True)
                bound = float(min(image.shape[-2:]))
                lo = torch.where(torch.isfinite(lo), lo, torch.zeros_like(lo))
                hi = torch.where(torch.isfinite(hi), hi, torch.full_like(hi, bound))
                rnd = lo + torch.rand_like(keypoints) * (hi - lo)
                keypoints = torch.where(valid[..., None], keypoints, rnd)
                kscores = torch.where(valid, kscores, torch.zeros_like(kscores))
                if keypoints.shape[1] < k:
                    extra = k - keypoints.shape[1]
                    keypoints = torch.cat([keypoints, lo + torch.rand(b, extra, 2, device=lo.device) * (hi - lo)], 1)
                    kscores = torch.cat([kscores, kscores.new_zeros(b, extra)], 1)
            elif b == 1:
                keypoints, kscores = keypoints[:, valid[0]], kscores[:, valid[0]]
            elif not bool(valid.all()):
                raise BatchedExtractionUnsupported("images yield different keypoint counts: set force_num_keypoints")
        descriptors = self._sample(keypoints, desc_map, dense, fused, s)
        pred = {"keypoints": keypoints + 0.5, "keypoint_scores": kscores, "descriptors": descriptors}
        if conf.dense_outputs:
            pred["dense_descriptors"] = dense()
        return pred

    # ------------------------------------------------------------------ hooks of the post-processing (see superpoint.py)
    def _max_keypoints(self):
        return self.conf.max_num_keypoints

    def _sample_keypoints(self, cand, scores, k):
        return None

    def _border_limits(self, data):
        return None

    @staticmethod
    def _apply_limits(scores, limits):
        H, W = scores.shape[-2:]
        xs = torch.arange(W, device=scores.device)[None, None, :]
        ys = torch.arange(H, device=scores.device)[None, :, None]
        return torch.where((xs >= limits[0][..., None]) | (ys >= limits[1][..., None]), torch.full_like(scores, -1.0), scores)

    def _refine(self, keypoints, dense_scores):
        return keypoints

    def _sample(self, keypoints, desc_map, dense, fused, s, shift=0.0):
        """Descriptors at the keypoints (superpoint_open.py:10-16): bilinear sample of the per-pixel-normalised map, then L2
        normalisation; `shift` is added to the keypoints first (the non-free variant's corrected sampling omits the +0.5)."""
        b = keypoints.shape[0]
        if fused and desc_map.shape[1] % 64 == 0 and desc_map.shape[1] <= 512:
            from .. import lib as _lib
            dm = desc_map if desc_map.is_contiguous(memory_format=torch.channels_last) else \
                desc_map.contiguous(memory_format=torch.channels_last)
            kp = (keypoints.float() + shift).contiguous()
            descriptors = torch.empty((b, kp.shape[1], dm.shape[1]), dtype=torch.float32, device=dm.device)
            _lib.check(_lib.load().gf_sample_descriptors(
                dm.data_ptr(), kp.data_ptr(), descriptors.data_ptr(), b, kp.shape[1], dm.shape[2], dm.shape[3],
                dm.shape[1], s, 1 if dm.dtype == torch.bfloat16 else 0, torch.cuda.current_stream().cuda_stream),
                "gf_sample_descriptors")
            return descriptors
        return sample_descriptors(keypoints + shift, dense(), s).transpose(-1, -2)

    def loss(self, pred, data):
        raise NotImplementedError


__main_model__ = SuperPoint
