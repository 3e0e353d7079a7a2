"""Fused tails of the frozen SuperPoint extractor (csrc/extractor.hip) against the stock-torch path and
the reference-generated golden vectors."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("radius", [1, 3, 4])
@pytest.mark.parametrize("shape", [(2, 97, 130), (1, 256, 256)])
def test_nms_kernel_equals_max_pool_chain(radius, shape):
    from glue_factory_amd import lib as L_
    from glue_factory_amd.extractors.superpoint_open import batched_nms
    g = torch.Generator(device="cuda").manual_seed(radius)
    s = torch.rand(*shape, device="cuda", generator=g)
    s[:, 10:30, 10:30] = 0.25                      # a plateau: ties everywhere inside it
    s = (s * 64).round() / 64                      # and many exact ties elsewhere
    ref = batched_nms(s, radius)
    pad = 4
    ref2 = ref.clone()
    ref2[:, :pad] = -1; ref2[:, :, :pad] = -1; ref2[:, -pad:] = -1; ref2[:, :, -pad:] = -1
    out = torch.empty_like(s)
    lib = L_.load()
    st = torch.cuda.current_stream().cuda_stream
    L_.check(lib.gf_nms_scores(s.data_ptr(), out.data_ptr(), *shape, radius, 0, st), "gf_nms_scores")
    assert torch.equal(out, ref)
    L_.check(lib.gf_nms_scores(s.data_ptr(), out.data_ptr(), *shape, radius, pad, st), "gf_nms_scores")
    assert torch.equal(out, ref2)


@pytest.mark.parametrize("radius", [1, 3, 4])
@pytest.mark.parametrize("shape", [(2, 97, 130), (3, 256, 256)])
def test_nms_candidate_lists_equal_dense_positives(radius, shape):
    """gf_nms_candidates (the NMS kernel appending its surviving maxima to per-image lists) against the dense kernel:
    the same set of (pixel, score) for every positive entry outside the border, the exact count, nothing else."""
    from glue_factory_amd import lib as L_
    B, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(radius + H)
    s = torch.rand(*shape, device="cuda", generator=g)
    s[:, 40:60, 40:60] = 0.25                      # a plateau of exact ties
    s[:, 5:9, 70:90] = 0.0                         # maxima-free zeros (never candidates)
    pad = 4
    dense = torch.empty_like(s)
    lib = L_.load()
    st = torch.cuda.current_stream().cuda_stream
    L_.check(lib.gf_nms_scores(s.data_ptr(), dense.data_ptr(), B, H, W, radius, pad, st), "gf_nms_scores")
    cap = lib.gf_nms_candidates_cap(H, W, radius)
    outs = []
    for _ in range(2):
        cs = torch.full((B, cap), -1.0, device="cuda")
        ci = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
        L_.check(lib.gf_nms_candidates(s.data_ptr(), cs.data_ptr(), ci.data_ptr(), B, H, W, radius, pad, st), "gf_nms_candidates")
        outs.append((cs, ci))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])       # deterministic layout
    dropped = 0
    for b in range(B):
        ref_idx = torch.nonzero(dense[b].flatten() > 0).flatten()
        used = cs[b] > -1
        got_idx, order = ci[b][used].long().sort()
        got_s = cs[b][used][order]
        assert got_idx.unique().numel() == got_idx.numel()
        pos = torch.searchsorted(ref_idx, got_idx)
        assert bool((ref_idx[pos.clamp(max=ref_idx.numel() - 1)] == got_idx).all())     # nothing but dense positives
        assert torch.equal(got_s, dense[b].flatten()[got_idx])
        # per kernel tile (64 - 10 r columns x 48 rows, one fixed segment each): every positive is there unless the tile
        # holds more than a segment's worth (only the plateau of exact ties can do that) -- then exactly a segment's worth
        wout, rt = 64 - 10 * radius, 48
        strips, tys = -(-W // wout), -(-H // rt)
        seg = cap // (strips * tys)
        tile_of = lambda idx: (idx // W // rt) * strips + (idx % W) // wout      # noqa: E731
        n_ref = torch.bincount(tile_of(ref_idx), minlength=strips * tys)
        n_got = torch.bincount(tile_of(got_idx), minlength=strips * tys)
        assert torch.equal(n_got, n_ref.clamp(max=seg))
        assert int((n_ref > seg).sum()) <= 4                # the 20 x 20 plateau touches at most four tiles
        dropped += ref_idx.numel() - got_idx.numel()
    print(f"radius {radius} {shape}: {dropped} tied plateau maxima beyond their tile's segment")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 17, 23), (1, 16, 16), (3, 40, 64)])
def test_detector_scores_kernel(dtype, shape):
    """gf_detector_scores (bias + BatchNorm(eval) of detector.1, softmax over the 65 channels, dustbin dropped, 8 x 8 cells
    unfolded into the score map; superpoint_open.py:105-108, 141-147) vs the stock ops on the same convolution output;
    cell counts that are / are not multiples of the 256-cell workgroup."""
    from glue_factory_amd import lib as L_
    B, h, w = shape
    g = torch.Generator(device="cuda").manual_seed(h * w)
    y = (torch.randn(B, 65, h, w, device="cuda", generator=g) * 3).to(dtype).contiguous(memory_format=torch.channels_last)
    bias, scale, shift = (torch.randn(65, device="cuda", generator=g) for _ in range(3))
    det = (y.float() + bias.view(1, 65, 1, 1)) * scale.view(1, 65, 1, 1) + shift.view(1, 65, 1, 1)
    ref = torch.softmax(det, 1)[:, :-1]
    ref = ref.permute(0, 2, 3, 1).reshape(B, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(B, h * 8, w * 8)
    out = torch.full((B, h * 8, w * 8), float("nan"), device="cuda")
    L_.check(L_.load().gf_detector_scores(y.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(),
                                          B, h, w, 0, 1 if dtype == torch.bfloat16 else 0,
                                          torch.cuda.current_stream().cuda_stream), "gf_detector_scores")
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-9)      # fp32 operation order (fma in the affine, exp)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pool", [0, 1])
def test_bias_act_bn_kernel(dtype, pool):
    from glue_factory_amd import lib as L_
    g = torch.Generator(device="cuda").manual_seed(3)
    B, C, H, W = 2, 64, 20, 36
    x = torch.randn(B, C, H, W, device="cuda", generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    bias, scale, shift = (torch.randn(C, device="cuda", generator=g) for _ in range(3))
    ref = torch.relu(x.float() + bias.view(1, C, 1, 1)) * scale.view(1, C, 1, 1) + shift.view(1, C, 1, 1)
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2, 2)
    y = torch.empty((B, C, H // 2, W // 2) if pool else (B, C, H, W), dtype=dtype, device="cuda",
                    memory_format=torch.channels_last)
    L_.check(L_.load().gf_bias_act_bn_nhwc(x.data_ptr(), y.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                           B, H, W, C, 1, pool, 1 if dtype == torch.bfloat16 else 0,
                                           torch.cuda.current_stream().cuda_stream), "gf_bias_act_bn_nhwc")
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    torch.testing.assert_close(y.float(), ref, rtol=tol, atol=tol)


def test_fused_extractor_matches_golden_and_stock_path():
    from glue_factory_amd.extractors.superpoint_open import SuperPoint
    z = load_golden("superpoint_open")
    torch.manual_seed(int(z["seed"]))
    model = SuperPoint({"max_num_keypoints": 100, "force_num_keypoints": True, "detection_threshold": 0.0,
                        "nms_radius": 3}).cuda().eval()
    image = torch.from_numpy(z["image"]).cuda()
    with torch.no_grad():
        assert model._use_fused(image)
        pred = model({"image": image})
    np.testing.assert_allclose(pred["keypoint_scores"].cpu().numpy(), z["eval.keypoint_scores"], rtol=1e-4, atol=1e-6)
    for b in range(image.shape[0]):
        ours = {tuple(k): i for i, k in enumerate(pred["keypoints"][b].cpu().tolist())}
        ref = z["eval.keypoints"][b].tolist()
        common = [(ours[tuple(k)], j) for j, k in enumerate(ref) if tuple(k) in ours]
        assert len(common) >= 0.95 * len(ref)
        io, ir = zip(*common)
        np.testing.assert_allclose(pred["descriptors"][b][list(io)].cpu().numpy(),
                                   z["eval.descriptors"][b][list(ir)], rtol=1e-3, atol=1e-4)
    # the stock path (grad enabled on a trainable copy) gives the same detections
    for p in model.parameters():
        p.requires_grad_(True)
    assert not model._use_fused(image)
    stock = model({"image": image})
    np.testing.assert_allclose(stock["keypoint_scores"].detach().cpu().numpy(), pred["keypoint_scores"].cpu().numpy(),
                               rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sample_descriptors_kernel(dtype):
    from glue_factory_amd import lib as L_
    from glue_factory_amd.extractors.superpoint_open import sample_descriptors
    g = torch.Generator(device="cuda").manual_seed(5)
    B, C, h, w, N, s = 2, 256, 17, 23, 300, 8
    m = torch.randn(B, C, h, w, device="cuda", generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    kp = torch.rand(B, N, 2, device="cuda", generator=g) * torch.tensor([w * s - 1.0, h * s - 1.0], device="cuda")
    kp[:, :4] = torch.tensor([[0.0, 0.0], [w * s - 1.0, h * s - 1.0], [0.0, h * s - 1.0], [3.5, 3.5]], device="cuda")
    ref = sample_descriptors(kp, torch.nn.functional.normalize(m.float(), p=2, dim=1), s).transpose(-1, -2)
    out = torch.empty(B, N, C, device="cuda")
    L_.check(L_.load().gf_sample_descriptors(m.data_ptr(), kp.contiguous().data_ptr(), out.data_ptr(), B, N, h, w, C, s,
                                             1 if dtype == torch.bfloat16 else 0, torch.cuda.current_stream().cuda_stream),
             "gf_sample_descriptors")
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 70, 45), (1, 64, 96)])
def test_first_block_kernel(dtype, shape):
    """gf_conv1_bias_act_bn (Conv2d(1,64,3,pad 1) + bias + ReLU + BatchNorm(eval), superpoint_open.py:98-100) vs the
    stock modules; ragged tile sizes, borders (zero padding)."""
    from glue_factory_amd import lib as L_
    B, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(H)
    img = torch.rand(B, 1, H, W, device="cuda", generator=g).to(dtype)
    w = (torch.randn(64, 1, 3, 3, device="cuda", generator=g) * 0.3).to(dtype)
    bias, scale, shift = (torch.randn(64, device="cuda", generator=g) for _ in range(3))
    ref = torch.nn.functional.conv2d(img.float(), w.float(), bias, padding=1)
    ref = torch.relu(ref) * scale.view(1, 64, 1, 1) + shift.view(1, 64, 1, 1)
    out = torch.empty((B, 64, H, W), dtype=dtype, device="cuda", memory_format=torch.channels_last)
    L_.check(L_.load().gf_conv1_bias_act_bn(img.data_ptr(), w.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                            out.data_ptr(), B, H, W, 64, 1, 1 if dtype == torch.bfloat16 else 0,
                                            torch.cuda.current_stream().cuda_stream), "gf_conv1_bias_act_bn")
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    torch.testing.assert_close(out.float(), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("shape", [(1, 8, 32), (2, 16, 96), (3, 64, 64), (1, 256, 288)])
def test_conv3x3_c64_kernel(shape, pool):
    """gf_conv3x3_c64 (Conv2d(64,64,3,pad 1) + bias + ReLU + BatchNorm(eval) [+ MaxPool 2x2], superpoint_open.py:37-75)
    vs the stock fp32 ops on the same bf16 inputs: single tile (all borders), several tiles per image / images per
    launch (persistent loop, double-buffered window), more tiles than workgroups.  bf16 output: tolerance = bf16
    rounding of the result (2^-8 relative) + fp32 accumulation order."""
    from glue_factory_amd import lib as L_
    B, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(H * 7 + W)
    x = torch.randn(B, 64, H, W, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias, scale, shift = (torch.randn(64, device="cuda", generator=g) for _ in range(3))
    ref = torch.nn.functional.conv2d(x.float(), w.float(), bias, padding=1)
    ref = torch.relu(ref) * scale.view(1, 64, 1, 1) + shift.view(1, 64, 1, 1)
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2, 2)
    taps = w.permute(2, 3, 0, 1).contiguous()
    out = torch.full(ref.shape, float("nan"), dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
    L_.check(L_.load().gf_conv3x3_c64(x.data_ptr(), taps.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                      out.data_ptr(), B, H, W, 1, int(pool), 1, torch.cuda.current_stream().cuda_stream),
             "gf_conv3x3_c64")
    torch.testing.assert_close(out.float(), ref, rtol=8e-3, atol=8e-3)


@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("shape", [(2, 16, 96), (1, 256, 288)])
def test_conv3x3_c64_into_a_wider_output(shape, pool):
    """gf_conv3x3_c64_ld: a 64 -> 128 block (backbone.2.0, superpoint_open.py:101-103) as two launches of the 64 -> 64 kernel,
    each writing its 64-channel slice of the [B, h, w, 128] channels-last output with the tail fused -- vs the stock fp32 ops
    on the same bf16 inputs; the other half of every pixel must stay untouched by each launch."""
    from glue_factory_amd import lib as L_
    B, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(H * 5 + W)
    x = torch.randn(B, 64, H, W, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 64, 3, 3, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias, scale, shift = (torch.randn(128, device="cuda", generator=g) for _ in range(3))
    ref = torch.nn.functional.conv2d(x.float(), w.float(), bias, padding=1)
    ref = torch.relu(ref) * scale.view(1, 128, 1, 1) + shift.view(1, 128, 1, 1)
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2, 2)
    out = torch.full(ref.shape, 777.0, dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
    st = torch.cuda.current_stream().cuda_stream
    for half in range(2):
        sl = slice(64 * half, 64 * half + 64)
        taps = w[sl].permute(2, 3, 0, 1).contiguous()
        L_.check(L_.load().gf_conv3x3_c64_ld(x.data_ptr(), taps.data_ptr(), bias[sl].contiguous().data_ptr(),
                                             scale[sl].contiguous().data_ptr(), shift[sl].contiguous().data_ptr(),
                                             out.data_ptr() + 128 * half, 128, B, H, W, 1, int(pool), 1, st), "gf_conv3x3_c64_ld")
        if half == 0:
            assert bool((out[:, 64:] == 777.0).all())              # the other half of every pixel: not written
    torch.testing.assert_close(out.float(), ref, rtol=8e-3, atol=8e-3)
    z = torch.zeros(8, device="cuda")
    assert L_.load().gf_conv3x3_c64_ld(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(),
                                       60, 1, 8, 32, 1, 0, 1, st) == -3            # GF_ERR_ALIGN: a pixel stride below 64 channels


def test_conv3x3_c64_rejects():
    from glue_factory_amd import lib as L_
    z = torch.zeros(64, device="cuda")
    lib = L_.load()
    assert lib.gf_conv3x3_c64(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(),
                              1, 8, 32, 1, 0, 0, None) == -4          # fp32: GF_ERR_DTYPE
    assert lib.gf_conv3x3_c64(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(),
                              1, 12, 32, 1, 0, 1, None) == -1         # H % 8: GF_ERR_UNSUPPORTED


@pytest.mark.parametrize("B,n,K,valid_frac", [(3, 87040, 2048, 0.25), (2, 5000, 2048, 0.1), (4, 4096, 4096, 0.6),
                                              (1, 300, 7, 1.0), (2, 9000, 1000, 0.0), (2, 20000, 2048, 0.5)])
def test_topk_candidates_equals_torch_topk(B, n, K, valid_frac):
    """gf_topk_candidates (csrc/topk.hip) vs torch.topk(sorted=True) + gather on NMS-style candidate lists: positive scores
    in scattered slots, -1 in the unfilled ones; incl. lists with fewer than K real entries, none at all, exact ties at
    the threshold (quantised scores) and K == n."""
    import ctypes
    from glue_factory_amd import lib
    g = torch.Generator(device="cuda").manual_seed(n + K)
    s = torch.rand(B, n, device="cuda", generator=g)
    if n == 20000:
        s = (s * 64).floor() / 64 + 1.0 / 128          # heavy ties: 64 distinct values
    s = torch.where(torch.rand(B, n, device="cuda", generator=g) < valid_frac, s, torch.full_like(s, -1.0)).contiguous()
    payload = torch.randint(0, 2 ** 20, (B, n), device="cuda", generator=g, dtype=torch.int32)
    out_s = torch.empty(B, K, device="cuda")
    out_p = torch.empty(B, K, device="cuda", dtype=torch.int64)
    lib.check(lib.load().gf_topk_candidates(s.data_ptr(), payload.data_ptr(), out_s.data_ptr(), out_p.data_ptr(), B, n, K,
                                            torch.cuda.current_stream().cuda_stream), "gf_topk_candidates")
    ref_s, ref_j = torch.topk(s, K, dim=1, sorted=True)
    assert torch.equal(out_s, ref_s)                                   # the sorted score vectors are identical
    # payloads: identical wherever the score is unique; inside a group of tied scores ours are in list order
    order = torch.sort(s, dim=1, descending=True, stable=True).indices[:, :K]      # (score descending, position ascending)
    assert torch.equal(out_p, payload.gather(1, order).long())
    uniq = torch.ones_like(ref_s, dtype=torch.bool)
    uniq[:, 1:] &= ref_s[:, 1:] != ref_s[:, :-1]
    uniq[:, :-1] &= ref_s[:, :-1] != ref_s[:, 1:]
    assert torch.equal(out_p[uniq], payload.gather(1, ref_j).long()[uniq])
    with pytest.raises(RuntimeError):
        lib.check(lib.load().gf_topk_candidates(s.data_ptr(), payload.data_ptr(), out_s.data_ptr(), out_p.data_ptr(), B, n, n + 1,
                                                torch.cuda.current_stream().cuda_stream), "gf_topk_candidates")


def test_nonfree_superpoint_fused_path_matches_reference_golden():
    """glue_factory_amd.extractors.superpoint (gluefactory_nonfree/superpoint.py:152-350: the extractor the N=2048 LightGlue
    yaml names) on the fused HIP path -- identity scale / shift in place of the open variant's BatchNorm -- against vectors the
    reference module itself produced (tests/golden/superpoint_nonfree.npz): fp32 at 1e-4, bf16 autocast by keypoint overlap."""
    from superpoint_nonfree_check import check
    check("cuda")
    miss = check("cuda", autocast=True)
    print(f"non-free SuperPoint under bf16 autocast: {100 * miss:.1f} % of the reference keypoints not re-detected at the same pixel")
    assert miss < 0.25


def test_nonfree_superpoint_randomized_keypoints_in_training_mode():
    """``randomize_keypoints_training`` (gluefactory_nonfree/superpoint.py:84-89, 273-282): in training mode k keypoints are drawn
    without replacement, proportionally to their scores, from the detections the top-k would choose from; an image with fewer
    than k detections keeps all of them; eval mode stays the deterministic top-k."""
    from conftest import load_golden
    from glue_factory_amd.extractors.superpoint import SuperPoint
    z = load_golden("superpoint_nonfree")
    image = torch.from_numpy(z["image"]).cuda()
    conf = {"force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3, "randomize_keypoints_training": True}
    torch.manual_seed(int(z["seed"]))
    pool_model = SuperPoint({**conf, "max_num_keypoints": 4000, "randomize_keypoints_training": False})
    pool_model.convPb.weight.data.mul_(40.0)
    pool_model = pool_model.cuda().eval()
    torch.manual_seed(int(z["seed"]))
    model = SuperPoint({**conf, "max_num_keypoints": 64})
    model.convPb.weight.data.mul_(40.0)
    model = model.cuda()
    with torch.no_grad():
        pool = pool_model({"image": image})                      # every detection of every image (padded beyond their count)
        model.train()
        torch.manual_seed(1)
        a = model({"image": image})
        torch.manual_seed(2)
        b = model({"image": image})
        model.eval()
        e1, e2 = model({"image": image}), model({"image": image})
    # eval: the top-k, not a draw.  Two calls agree up to near-ties at the k-th score: the stock library convolutions of the 128 /
    # 256-channel blocks (MIOpen, as in the reference) are not bit-reproducible from call to call -- fp32 outputs move by ~5e-7,
    # the score map by ~1e-8 (tools/probe/conv_determinism.py, sp_eval_determinism.py) -- which swaps the 64th and 65th candidate of
    # this sharpened detector once in a hundred calls (the exact comparison that stood here failed one suite run in ~10)
    torch.testing.assert_close(e1["keypoint_scores"], e2["keypoint_scores"], rtol=0, atol=1e-6)
    for i in range(image.shape[0]):
        k1 = {tuple(k) for k in e1["keypoints"][i].round().long().tolist()}
        k2 = {tuple(k) for k in e2["keypoints"][i].round().long().tolist()}
        assert len(k1 & k2) >= 62, (i, len(k1 & k2))
    assert not torch.equal(a["keypoints"], b["keypoints"])                           # training: a draw per call
    for i in range(image.shape[0]):
        dets = {tuple(k) for k, s in zip(pool["keypoints"][i].round().long().tolist(), pool["keypoint_scores"][i].tolist()) if s > 0}
        assert len(dets) > 64
        got = [tuple(k) for k in a["keypoints"][i].round().long().tolist()]
        assert all(s > 0 for s in a["keypoint_scores"][i].tolist())                   # enough detections: no padding
        assert len(set(got)) == 64 and set(got) <= dets                               # distinct, all of them detections
        top = {tuple(k) for k in e1["keypoints"][i].round().long().tolist()}
        assert set(got) != top                                                        # not simply the 64 best
    # fewer detections than k: all of them are kept (the surplus is padding with score 0)
    torch.manual_seed(int(z["seed"]))
    big = SuperPoint({**conf, "max_num_keypoints": 4000})
    big.convPb.weight.data.mul_(40.0)
    big = big.cuda().train()
    with torch.no_grad():
        c = big({"image": image})
    for i in range(image.shape[0]):
        dets = {tuple(k) for k, s in zip(pool["keypoints"][i].round().long().tolist(), pool["keypoint_scores"][i].tolist()) if s > 0}
        got = {tuple(k) for k, s in zip(c["keypoints"][i].round().long().tolist(), c["keypoint_scores"][i].tolist()) if s > 0}
        assert got == dets


@pytest.mark.parametrize("cname", ["uncapped", "capped", "noborder"])
def test_open_superpoint_options_match_reference_golden(cname):
    """superpoint_open configurations beyond the benchmark's (tests/golden/superpoint_options.npz, reference-generated from
    oracle/option_cases.superpoint_option_cases): no keypoint cap + detection threshold + `dense_outputs` on one image
    (variable count, row-major order of torch.where), a cap without padding (top-k of the detections above the threshold),
    `remove_borders: 0` with `nms_radius: 2` (superpoint_open.py:79-90, 142-207).  Keypoint SETS equal the reference's except
    for detections whose score sits within 1e-5 of the deciding edge (threshold / k-th score); scores 1e-4, descriptors 1e-3."""
    from glue_factory_amd.extractors.superpoint_open import SuperPoint
    from oracle.option_cases import superpoint_option_cases
    z = load_golden("superpoint_options")
    conf, shape = superpoint_option_cases()[cname]
    torch.manual_seed(int(z["seed"]))
    model = SuperPoint(conf)
    for prm in model.detector[1].parameters():
        if prm.ndim == 4:
            prm.data.mul_(40.0)
    model = model.cuda().eval()
    image = torch.from_numpy(z[f"{cname}.image"]).cuda()
    assert tuple(image.shape) == shape
    with torch.no_grad():
        assert model._use_fused(image)
        pred = model({"image": image})
    ref_kp, ref_sc, ref_desc = z[f"{cname}.keypoints"], z[f"{cname}.keypoint_scores"], z[f"{cname}.descriptors"]
    assert pred["keypoints"].shape[0] == ref_kp.shape[0] and pred["descriptors"].shape[-1] == 256
    cap = conf.get("max_num_keypoints")
    for b in range(ref_kp.shape[0]):
        ours = {tuple(k): i for i, k in enumerate(pred["keypoints"][b].cpu().tolist())}
        ref = {tuple(k): j for j, k in enumerate(ref_kp[b].tolist())}
        edge = float(ref_sc[b].min()) if cap is not None and len(ref) == cap else conf["detection_threshold"]
        sc = pred["keypoint_scores"][b].cpu().numpy()
        for k in set(ours) ^ set(ref):            # only detections sitting on the deciding edge may differ
            s_k = sc[ours[k]] if k in ours else ref_sc[b][ref[k]]
            assert abs(float(s_k) - edge) < 1e-5, (k, float(s_k), edge)
        common = sorted(set(ours) & set(ref))
        assert len(common) >= 0.98 * len(ref) and len(ours) <= (cap or 10 ** 9)
        io, ir = [ours[k] for k in common], [ref[k] for k in common]
        np.testing.assert_allclose(sc[io], ref_sc[b][ir], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(pred["descriptors"][b][io].float().cpu().numpy(), ref_desc[b][ir], rtol=1e-3, atol=1e-4)
        if cap is None:                           # same ORDER too (torch.where: row-major over the score map)
            assert sorted(common, key=ours.get) == sorted(common, key=ref.get)
    if conf.get("dense_outputs"):
        np.testing.assert_allclose(pred["dense_descriptors"].float().cpu().numpy(), z[f"{cname}.dense_descriptors"],
                                   rtol=1e-3, atol=1e-4)
    else:
        assert "dense_descriptors" not in pred
