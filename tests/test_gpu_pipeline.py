"""Config-1 plumbing on the GPU: frozen SuperPoint (stock torch) -> HIP LightGlue -> homography GT,
built through the plugin registry like the shipped yaml, one train step + a short overfit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pipeline(n_kpts=128, layers=2):
    from glue_factory_amd.base_model import get_model
    P = get_model("glue_factory_amd.pipeline")
    return P({
        "extractor": {"name": "extractors.superpoint_open", "max_num_keypoints": n_kpts,
                      "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3,
                      "trainable": False},
        "ground_truth": {"name": "matchers.homography_matcher", "th_positive": 3, "th_negative": 3},
        "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1, "flash": False,
                    "checkpointed": True, "n_layers": layers},
    })


def _batch(b=2, h=240, w=320, seed=0):
    from glue_factory_amd.synthetic import similarity_homography
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(b, 3, h, w, generator=g)
    size = torch.tensor([[w, h]], dtype=torch.float32).repeat(b, 1)
    return {"view0": {"image": img, "image_size": size}, "view1": {"image": img.roll(8, -1), "image_size": size},   # one 8-px SuperPoint cell: shift-equivariant
            "H_0to1": torch.tensor([[1.0, 0, 8], [0, 1, 0], [0, 0, 1]])[None].repeat(b, 1, 1)}


def test_pipeline_train_step_and_overfit():
    from glue_factory_amd.synthetic import to_device
    from glue_factory_amd.train_step import TrainStep
    torch.manual_seed(0)
    pipe = _pipeline().cuda()
    assert sum(p.requires_grad for p in pipe.extractor.parameters()) == 0
    data = to_device(_batch(), "cuda")
    pipe.eval()     # frozen extractor with fixed statistics -> deterministic keypoints
    pipe.matcher.train()
    pred = pipe(data)
    losses, _ = pipe.loss(pred, data)
    assert pred["gt_matches0"].shape == (2, 128) and (pred["gt_matches0"] >= 0).sum() > 50
    assert losses["total"].shape == (2,) and torch.isfinite(losses["total"]).all()
    # matcher output equals a direct matcher call on the extracted features (plumbing only)
    direct = pipe.matcher({**data, **{k: pred[k] for k in ("keypoints0", "keypoints1", "descriptors0", "descriptors1")}})
    torch.testing.assert_close(direct["log_assignment"], pred["log_assignment"])

    opt = torch.optim.Adam([p for p in pipe.parameters() if p.requires_grad], lr=1e-3)

    class Wrapper(torch.nn.Module):          # TrainStep toggles .train(): keep the extractor in eval
        def __init__(self, m):
            super().__init__()
            self.m = m

        def train(self, mode=True):
            self.m.matcher.train(mode)
            return self

        def forward(self, d):
            return self.m(d)

        def loss(self, pred, d):
            return self.m.loss(pred, d)

    step = TrainStep(Wrapper(pipe), opt)
    first = step(data)["total"].mean().item()
    for _ in range(15):
        last = step(data)["total"].mean().item()
    assert last < 0.95 * first, (first, last)


def test_triplet_pipeline_with_hip_lightglue():
    """TripletPipeline (triplet_pipeline.py:23-99) around the real HIP matcher: the three pairs stacked into one
    matcher call (3B pairs per launch) give the same log-assignment as pair-by-pair and as plain two-view calls."""
    from glue_factory_amd.base_model import get_model
    from glue_factory_amd.synthetic import to_device
    P3 = get_model("glue_factory_amd.triplet_pipeline")
    conf = {"extractor": {"name": "extractors.superpoint_open", "max_num_keypoints": 96, "force_num_keypoints": True,
                          "detection_threshold": 0.0, "nms_radius": 3, "trainable": False},
            "matcher": {"name": "matchers.lightglue", "n_layers": 2, "filter_threshold": 0.0}}
    torch.manual_seed(0)
    batched = P3({**conf, "batch_triplets": True}).cuda().eval()
    pairwise = P3({**conf, "batch_triplets": False}).cuda().eval()
    pairwise.load_state_dict(batched.state_dict())
    g = torch.Generator().manual_seed(3)
    size = torch.tensor([[160.0, 120.0]]).repeat(2, 1)
    data = to_device({f"view{i}": {"image": torch.rand(2, 3, 120, 160, generator=g), "image_size": size}
                      for i in range(3)}, "cuda")
    with torch.no_grad():
        pb, pp = batched(data), pairwise(data)
    for idx in ("0to1", "0to2", "1to2"):
        assert pb[idx]["log_assignment"].shape == (2, 97, 97)
        torch.testing.assert_close(pb[idx]["log_assignment"], pp[idx]["log_assignment"], rtol=1e-4, atol=1e-4)
        assert torch.equal(pb[idx]["matches0"], pp[idx]["matches0"])
    two = {k: v for k, v in data.items() if k != "view2"}
    with torch.no_grad():
        p2 = batched(two)
    torch.testing.assert_close(p2["log_assignment"], pb["0to1"]["log_assignment"], rtol=1e-4, atol=1e-4)


def test_triplet_pipeline_trains_lightglue_batched_equals_pairwise():
    """TripletPipeline (batch_triplets, the default) around the HIP LightGlue in TRAINING mode: the fused loss reads the
    matcher's private per-layer / image-stacked state, so the stacked prediction must reach it untouched.  Batched
    triplets == pair by pair: per-pair losses and every parameter gradient."""
    from glue_factory_amd.base_model import get_model
    from glue_factory_amd.synthetic import to_device
    P3 = get_model("glue_factory_amd.triplet_pipeline")

    def conf(batched):
        return {"extractor": {"name": "extractors.superpoint_open", "max_num_keypoints": 96, "force_num_keypoints": True,
                              "detection_threshold": 0.0, "nms_radius": 3, "trainable": False},
                "ground_truth": {"name": "matchers.homography_matcher", "th_positive": 3, "th_negative": 3},
                "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1, "n_layers": 3},
                "batch_triplets": batched}

    torch.manual_seed(0)
    pb = P3(conf(True)).cuda()
    pp = P3(conf(False)).cuda()
    pp.load_state_dict(pb.state_dict())
    d2 = _batch(b=2)
    g = torch.Generator().manual_seed(5)
    data = {"view0": d2["view0"], "view1": d2["view1"],
            "view2": {"image": d2["view0"]["image"].roll(16, -2), "image_size": d2["view0"]["image_size"]},
            "H_0to1": d2["H_0to1"],
            "H_0to2": torch.tensor([[1.0, 0, 0], [0, 1, 16], [0, 0, 1]])[None].repeat(2, 1, 1),
            "H_1to2": torch.tensor([[1.0, 0, -8], [0, 1, 16], [0, 0, 1]])[None].repeat(2, 1, 1)}
    data = to_device(data, "cuda")
    grads = []
    totals = []
    for pipe in (pb, pp):
        pipe.eval()
        pipe.matcher.train()
        pred = pipe(data)
        losses, _ = pipe.loss(pred, data)
        totals.append(losses["total"])
        losses["total"].sum().backward()
        grads.append({k: p.grad.clone() for k, p in pipe.matcher.named_parameters()})
    assert totals[0].shape == (6,) and totals[1].shape == (2,)
    # pair-by-pair losses are summed over the three pairs, the batched ones come stacked pair after pair
    torch.testing.assert_close(totals[0][0:2] + totals[0][2:4] + totals[0][4:6], totals[1], rtol=1e-4, atol=1e-4)
    for k in grads[0]:
        sc = grads[1][k].abs().max().clamp(min=1e-6)
        torch.testing.assert_close(grads[0][k] / sc, grads[1][k] / sc, rtol=5e-3, atol=5e-3, msg=lambda m: f"{k}: {m}")   # (6-pair vs 3 x 2-pair fp32 summation order: measured up to 2.1e-3)


def test_scope_p_replays_as_one_hipgraph_for_40_steps():
    """extractor + ground truth + matcher step (forward, loss, backward, fused Adam) captured as ONE hipGraph
    through the product API (TwoViewPipeline inside TrainStep(graph=True)) and replayed 40 times, with eager GPU work
    between the replays and no host synchronisation, equals the eager run.  The graph faulted on its second replay while
    the extractor's top-k was torch.topk (hipMemsetAsync nodes; tools/probe/capture_scope_p.py bisected it): the top-k is
    now csrc/topk.hip."""
    from glue_factory_amd.optim import FusedAdam
    from glue_factory_amd.pipeline import TwoViewPipeline
    from glue_factory_amd.synthetic import to_device
    from glue_factory_amd.train_step import TrainStep

    def build():
        torch.manual_seed(0)
        return TwoViewPipeline({
            "extractor": {"name": "extractors.superpoint_open", "max_num_keypoints": 256, "force_num_keypoints": True,
                          "detection_threshold": 0.0, "nms_radius": 3, "trainable": False, "freeze_batch_normalization": True},
            "ground_truth": {"name": "matchers.homography_matcher", "th_positive": 3, "th_negative": 3, "with_reward": False},
            "matcher": {"name": "matchers.lightglue", "n_layers": 2, "filter_threshold": 0.1},
        }).cuda()

    data = to_device(_batch(b=2, h=256, w=320, seed=3), "cuda")
    runs = []
    for graph in (False, True):
        pipe = build()
        step = TrainStep(pipe, FusedAdam([p for p in pipe.parameters() if p.requires_grad], lr=1e-3), amp_dtype=torch.bfloat16,
                         device_ids=[0], graph=graph)
        losses = []
        junk = torch.zeros(1 << 20, device="cuda")
        for it in range(43):                       # 2 eager warm-up calls + capture + 40 replays
            losses.append(step(data)["total"].mean().clone())
            junk.add_(1.0)                         # eager work queued between the replays
            if it % 7 == 0:
                pipe.extractor.eval()({"image": data["view0"]["image"]})      # ... incl. the library convolutions
        assert step.graph == graph and (not graph or step.static_inputs() is not None)
        runs.append(torch.stack(losses).cpu())
    le, lg = runs
    assert torch.isfinite(lg).all() and float(lg[-1]) < float(lg[0])                 # it trains
    torch.testing.assert_close(lg, le, rtol=2e-2, atol=2e-2)                          # bf16 steps, 43 updates apart at most
    # (parameters are not compared after 43 Adam updates: an entry whose gradient is rounding noise random-walks by +-lr per
    # step in either run; tests/test_gpu_optim.py::test_train_step_graph_with_fused_adam_matches_eager compares them after 6)


def _nonfree_pipeline(**ext):
    from glue_factory_amd.base_model import get_model
    P = get_model("glue_factory_amd.pipeline")
    conf = {"name": "extractors.superpoint", "detection_threshold": 0.0, "nms_radius": 3, "trainable": False,
            "remove_borders": 4, **ext}
    torch.manual_seed(7)
    pipe = P({"extractor": conf})
    pipe.extractor.convPb.weight.data.mul_(40.0)        # spread the detector logits of the random weights
    return pipe.cuda().eval()


def test_batched_extraction_carries_image_size_to_the_nonfree_extractor():
    """Square-padded views (superpoint+lightglue_megadepth style): `image_size` is smaller than the tensor and the non-free
    SuperPoint removes the border relative to it (gluefactory_nonfree/superpoint.py:236-244).  The pipeline's batched
    two-view call must hand `image_size` through: equal to the per-view calls, no keypoint in the padding."""
    from glue_factory_amd.synthetic import to_device
    pipe = _nonfree_pipeline(max_num_keypoints=256, force_num_keypoints=True)
    g = torch.Generator().manual_seed(3)
    b, h, w = 2, 256, 256
    img0, img1 = torch.rand(b, 1, h, w, generator=g), torch.rand(b, 1, h, w, generator=g)
    size0 = torch.tensor([[200.0, 160.0], [256.0, 176.0]])
    size1 = torch.tensor([[176.0, 256.0], [144.0, 208.0]])
    data = to_device({"view0": {"image": img0, "image_size": size0}, "view1": {"image": img1, "image_size": size1}}, "cuda")
    with torch.no_grad():
        pred = pipe(data)
        p0, p1 = pipe.extract_view(data, "0"), pipe.extract_view(data, "1")
    for i, (pv, size) in enumerate(((p0, size0), (p1, size1))):
        kp, sc = pred[f"keypoints{i}"], pred[f"keypoint_scores{i}"]
        det = sc > 0                                    # (padding keypoints of force_num_keypoints are random)
        assert int(det.sum()) > 100
        lim = (size - 4).cuda()[:, None]
        assert bool(((kp < lim) | ~det[..., None]).all()), "a detection in the border / padding beyond image_size"
        # equal to the per-view call: as SETS of detections (a batch of 4 and a batch of 2 may take different library
        # convolution kernels; near-equal scores then swap places in the sorted list or at the top-k cut)
        for j in range(b):
            mine = {tuple(v) for v in kp[j][det[j]].round().long().tolist()}
            per_view = {tuple(v) for v in pv["keypoints"][j][pv["keypoint_scores"][j] > 0].round().long().tolist()}
            assert len(mine ^ per_view) <= 0.04 * len(per_view), (i, j, len(mine ^ per_view), len(per_view))
            lookup = {tuple(v): n for n, v in enumerate(pv["keypoints"][j].round().long().tolist())}
            rows = [(n, lookup[tuple(v)]) for n, v in enumerate(kp[j].round().long().tolist())
                    if bool(det[j][n]) and tuple(v) in lookup]
            a, c = (torch.tensor(t, device="cuda") for t in zip(*rows))
            torch.testing.assert_close(pred[f"descriptors{i}"][j][a], pv["descriptors"][j][c], rtol=2e-2, atol=2e-2)
            torch.testing.assert_close(sc[j][a], pv["keypoint_scores"][j][c], rtol=2e-2, atol=1e-4)


def test_batch_of_one_with_a_variable_keypoint_count_extracts_view_by_view():
    """b == 1 eval of a frozen extractor with `max_num_keypoints: -1` / no force_num_keypoints (the non-free default):
    variable-length outputs exist for b == 1 only, so the two views must not be glued into a batch of two."""
    from glue_factory_amd.synthetic import to_device
    g = torch.Generator().manual_seed(4)
    data = to_device({"view0": {"image": torch.rand(1, 1, 128, 160, generator=g)},
                      "view1": {"image": torch.rand(1, 1, 128, 160, generator=g)}}, "cuda")
    for ext in ({"max_num_keypoints": -1, "detection_threshold": 0.02},
                {"max_num_keypoints": 64, "force_num_keypoints": False, "detection_threshold": 0.02, "refinement_radius": 2}):
        pipe = _nonfree_pipeline(**ext)
        with torch.no_grad():
            pred = pipe(data)
            p0 = pipe.extract_view(data, "0")
        assert pred["keypoints0"].shape[0] == 1 and pred["keypoints0"].shape[1] > 0
        torch.testing.assert_close(pred["keypoints0"], p0["keypoints"])
        torch.testing.assert_close(pred["descriptors0"], p0["descriptors"])
