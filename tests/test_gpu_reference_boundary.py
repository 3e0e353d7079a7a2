"""The drop-in boundary exercised FROM THE REFERENCE SIDE, on the GPU: the reference's OWN pipeline code
(gluefactory/models/two_view_pipeline.py:70-113, triplet_pipeline.py:23-99 -- byte-compiled into oracle/_ref by
oracle/build_ref.py, so it exists on the GPU box) constructs ``glue_factory_amd.matchers.lightglue`` and
``glue_factory_amd.matchers.homography_matcher`` through its own ``get_model`` from a yaml-style OmegaConf and runs
forward + loss + backward on cuda; results equal ``glue_factory_amd.pipeline`` on the same weights and batch.
Also here: a pred dict that was sliced / concatenated / rebuilt by the caller (what the reference's batched
TripletPipeline does to every entry, gluefactory/utils/misc.py:31-46) still trains the HIP matcher -- loss() then takes
the reference's dense formulation on the differentiable ``ref_descriptors`` instead of the fused path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    from oracle import build_ref
    if not build_ref.import_reference():
        pytest.skip("oracle/_ref not built (python oracle/build_ref.py in the build container)")
    import importlib
    for mod in ("gluefactory.models.two_view_pipeline", "gluefactory.models.triplet_pipeline"):
        try:
            importlib.import_module(mod)
        except ImportError:
            pytest.skip(f"oracle/_ref lacks {mod} (rebuild it)")
    return True


CONF = {"matcher": {"name": "glue_factory_amd.matchers.lightglue", "n_layers": 3, "filter_threshold": 0.1},
        "ground_truth": {"name": "glue_factory_amd.matchers.homography_matcher", "th_positive": 3.0, "th_negative": 3.0},
        "extractor": {"name": None}, "allow_no_extract": True}


def _views(batch, n, seed, names=("0", "1")):
    """Cached-feature batch (the reference's cached-feature training mode: view["cache"] stands in for the extractor)."""
    from glue_factory_amd.synthetic import make_pairs
    base = make_pairs(batch, n, dim=256, size=(640, 480), seed=seed, with_gt=False)
    data = {"H_0to1": base["H_0to1"]}
    for i in names[:2]:
        data["view" + i] = {"image_size": base["view" + i]["image_size"],
                            "cache": {"keypoints": base["keypoints" + i], "descriptors": base["descriptors" + i],
                                      "keypoint_scores": base["keypoint_scores" + i]}}
    return data, base


def _grads(module):
    return {k: p.grad.detach().clone() for k, p in module.named_parameters() if p.grad is not None}


def _train(pipe, data, bf16=False):
    pipe.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        pred = pipe(data)
        losses, metrics = pipe.loss(pred, data)
    torch.mean(losses["total"]).backward()
    return pred, losses, _grads(pipe)


def test_reference_two_view_pipeline_drives_the_hip_matcher(ref):
    from omegaconf import OmegaConf
    from gluefactory.models.two_view_pipeline import TwoViewPipeline as RefPipeline
    from glue_factory_amd.pipeline import TwoViewPipeline as OurPipeline
    from glue_factory_amd.matchers.lightglue import LightGlue
    from glue_factory_amd.synthetic import to_device
    torch.manual_seed(0)
    rp = RefPipeline(OmegaConf.create(CONF)).cuda().train()
    assert isinstance(rp.matcher, LightGlue) and type(rp).__module__.startswith("gluefactory.")
    ours = {**CONF, "matcher": {**CONF["matcher"], "name": "matchers.lightglue"},
            "ground_truth": {**CONF["ground_truth"], "name": "matchers.homography_matcher"}}
    op = OurPipeline(ours).cuda().train()
    op.load_state_dict(rp.state_dict(), strict=True)
    data, _ = _views(3, 256, seed=7)
    data = to_device(data, "cuda")
    for bf16 in (False, True):
        pr, lr, gr = _train(rp, data, bf16)
        po, lo, go = _train(op, data, bf16)
        assert set(k for k in pr if not k.startswith("_")) == set(k for k in po if not k.startswith("_"))
        for k in ("log_assignment", "matches0", "matches1", "matching_scores0", "gt_matches0", "gt_assignment"):
            assert torch.equal(pr[k], po[k]), k
        assert set(lr) == set(lo)
        for k, v in lr.items():
            if torch.is_tensor(v):
                torch.testing.assert_close(v, lo[k], rtol=1e-6, atol=1e-6, msg=lambda m: f"{k}: {m}")
        assert set(gr) == set(go) and len(gr) > 40
        for k in gr:
            torch.testing.assert_close(gr[k], go[k], rtol=1e-5, atol=1e-7, msg=lambda m: f"{k}: {m}")
    # eval mode: the reference pipeline returns the matcher metrics next to the losses
    rp.eval()
    with torch.no_grad():
        pe = rp(data)
        le, me = rp.loss(pe, data)
    assert {"match_recall", "match_precision", "accuracy", "average_precision"} <= set(me)
    assert torch.isfinite(le["total"]).all()


@pytest.mark.parametrize("bf16", [False, True])
def test_loss_on_a_rebuilt_pred_equals_the_fused_loss(bf16):
    """What the reference's batched TripletPipeline does to a matcher's output (unstack_twoviews slices every entry on the
    batch axis, stack_twoviews concatenates them again, gluefactory/utils/misc.py:31-46): the tensors are new objects, so
    loss() cannot use the fused path's private state and differentiates through ``ref_descriptors`` -- same loss, same
    gradients (the two formulations round differently in bf16: stated tolerance)."""
    from glue_factory_amd.matchers.lightglue import LightGlue
    from glue_factory_amd.synthetic import make_pairs, to_device
    torch.manual_seed(1)
    model = LightGlue({"n_layers": 3}).cuda().train()
    data = to_device(make_pairs(4, 192, dim=256, size=(640, 480), seed=3), "cuda")

    def run(rebuild):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            pred = model(data)
            assert pred["ref_descriptors0"].requires_grad            # the public descriptors are differentiable
            if rebuild:
                assert all(torch.is_tensor(v) for v in pred.values())        # nothing the reference's helpers cannot slice
                pred = {k: torch.cat([v[:2], v[2:]], 0) for k, v in pred.items()}
            losses, _ = model.loss(pred, {**pred, **data})
        losses["total"].mean().backward()
        return {k: v.detach() for k, v in losses.items() if torch.is_tensor(v)}, _grads(model)

    lf, gf = run(False)
    lr, gr = run(True)
    tol = 2e-2 if bf16 else 1e-4
    for k in lf:
        torch.testing.assert_close(lr[k], lf[k], rtol=tol, atol=tol, msg=lambda m: f"{k}: {m}")
    assert set(gr) == set(gf)
    worst = max(float((gr[k] - gf[k]).norm() / gf[k].norm().clamp(min=1e-20)) for k in gf)
    print(f"rebuilt-pred loss vs fused loss ({'bf16' if bf16 else 'fp32'}): worst relative gradient difference {worst:.2e}")
    assert worst < (0.05 if bf16 else 2e-4)


def test_reference_triplet_pipeline_drives_the_hip_matcher(ref):
    """The reference's TripletPipeline (pair-by-pair mode: its batched mode concatenates nested view dicts with torch.cat,
    which the reference itself cannot do on the nested batch layout) on three cached views: losses summed over the pairs
    and gradients equal our TripletPipeline's."""
    from omegaconf import OmegaConf
    from gluefactory.models.triplet_pipeline import TripletPipeline as RefTriplet
    from glue_factory_amd.triplet_pipeline import TripletPipeline as OurTriplet
    from glue_factory_amd.synthetic import make_pairs, similarity_homography, to_device
    torch.manual_seed(2)
    conf = {**CONF, "batch_triplets": False}
    rp = RefTriplet(OmegaConf.create(conf)).cuda().train()
    ours = {**conf, "matcher": {**CONF["matcher"], "name": "matchers.lightglue"},
            "ground_truth": {**CONF["ground_truth"], "name": "matchers.homography_matcher"}}
    op = OurTriplet(ours).cuda().train()
    op.load_state_dict(rp.state_dict(), strict=True)
    b, n = 2, 160
    base = make_pairs(b, n, dim=256, size=(640, 480), seed=11, with_gt=False)
    extra = make_pairs(b, n, dim=256, size=(640, 480), seed=12, with_gt=False)
    H = similarity_homography(640, 480)[None].repeat(b, 1, 1)
    view = lambda src, i: {"image_size": src["view" + i]["image_size"],            # noqa: E731
                           "cache": {"keypoints": src["keypoints" + i], "descriptors": src["descriptors" + i]}}
    data = {"view0": view(base, "0"), "view1": view(base, "1"), "view2": view(extra, "1"),
            "H_0to1": H, "H_0to2": H, "H_1to2": torch.eye(3)[None].repeat(b, 1, 1)}
    data = to_device(data, "cuda")
    pr, lr, gr = _train(rp, data)
    po, lo, go = _train(op, data)
    for idx in ("0to1", "0to2", "1to2"):
        assert torch.equal(pr[idx]["matches0"], po[idx]["matches0"]) and torch.equal(pr[idx]["log_assignment"], po[idx]["log_assignment"])
    for k in ("total", "last", "confidence", "nll_pos", "nll_neg"):
        torch.testing.assert_close(lr[k], lo[k], rtol=1e-5, atol=1e-5, msg=lambda m: f"{k}: {m}")
    assert set(gr) == set(go)
    for k in gr:
        torch.testing.assert_close(gr[k], go[k], rtol=1e-4, atol=1e-6, msg=lambda m: f"{k}: {m}")
