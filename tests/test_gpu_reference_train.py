"""The REFERENCE's own `training()` (gluefactory/train.py:216-683; byte-compiled into oracle/_ref, so it exists on the GPU
box) driving the HIP matcher on the GPU: `python -m gluefactory.train ... --mp bfloat16` in miniature -- its dataset /
DataLoader machinery on a synthetic-pairs dataset plugin, `get_model("glue_factory_amd.matchers.lightglue")`, autocast,
`torch.amp.GradScaler` at 65 536, gradient clipping, torch.optim.Adam, the exp lr schedule, validation through the eval
forward + loss + metrics, checkpoints.  The parameters it saves must be the ones `glue_factory_amd.train_step.TrainStep`
(our mirror of the loop: no scaler, device-side skip flag) reaches on the same batches, and the checkpoint must load back
into the HIP module."""
import pathlib

import pytest
import torch

import ref_train_harness as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tr():
    mod = H.import_reference_train()
    if mod is None:
        pytest.skip("oracle/_ref lacks gluefactory.train (python oracle/build_ref.py in the build container)")
    return mod


def _conf(model):
    return {"data": {"name": "synthetic_pairs_dataset", "batch_size": 4, "num_workers": 0, "prefetch_factor": None,
                     "n_train": 16, "n_val": 4, "n_kpts": 256, "dim": 256, "seed": 3},
            "model": model,
            "train": {"seed": 7, "epochs": 1, "lr": 1e-3, "log_every_iter": 1, "eval_every_iter": 1000,
                      "save_every_iter": 1000, "clip_grad": 1.0,
                      "lr_schedule": {"type": "exp", "start": 1, "exp_div_10": 10}}}


@pytest.mark.parametrize("compile_mode", [None, "default"])
def test_reference_training_drives_the_hip_lightglue(tr, tmp_path, compile_mode):
    from gluefactory.datasets import get_dataset
    from gluefactory.models import get_model
    from gluefactory.utils.tensor import batch_to_device
    from gluefactory.utils.tools import set_seed
    from omegaconf import OmegaConf
    from glue_factory_amd.train_step import TrainStep
    import torch._dynamo
    torch._dynamo.reset()
    CONF = _conf({"name": "glue_factory_amd.matchers.lightglue", "n_layers": 3, "filter_threshold": 0.1})
    out = pathlib.Path(tmp_path)
    writer = H.run_training(tr, CONF, out, H.train_args("gpu_lightglue", mixed_precision="bfloat16", compile_mode=compile_mode))
    totals = [v for k, v, _ in writer.scalars if k == "training//total"]
    assert len(totals) == 4 and all(0.0 < t < 50.0 for t in totals), totals
    val = {k: v for k, v, _ in writer.scalars if k.startswith("val/")}
    assert "val/loss/total" in val and "val/match_recall" in val and 0.0 <= val["val/match_recall"] <= 1.0
    cp = torch.load(out / "checkpoint_0_3.tar", map_location="cpu", weights_only=False)
    assert len(cp["optimizer"]["state"]) > 0 and cp["optimizer"]["state"][0]["step"] == 4        # no step was skipped
    # (torch.compile wraps the module: the reference then saves `_orig_mod.`-prefixed keys, train.py:332-333 + experiments.py)
    cp["model"] = {k.removeprefix("_orig_mod."): v for k, v in cp["model"].items()}
    # ---- the mirror: TrainStep on the same batches
    conf = OmegaConf.create(CONF)
    conf.train = OmegaConf.merge(tr.default_train_conf, conf.train)
    set_seed(conf.train.seed)
    dataset = get_dataset(conf.data.name)(conf.data)
    loader = dataset.get_data_loader("train")
    model = get_model(conf.model.name)(conf.model).to("cuda")
    opt = torch.optim.Adam(model.parameters(), lr=conf.train.lr)
    sched = tr.get_lr_scheduler(optimizer=opt, conf=conf.train.lr_schedule)
    step = TrainStep(model, opt, amp_dtype=torch.bfloat16, clip_grad=conf.train.clip_grad)
    set_seed(conf.train.seed)
    ours = []
    for data in loader:
        data = batch_to_device(data, "cuda", non_blocking=True)
        ours.append(float(step(data)["total"].mean()))
        sched.step()
    assert step.skipped == 0
    for a, b in zip(ours, totals):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (ours, totals)
    moved = 0.0
    init = get_model(conf.model.name)(conf.model)
    for (k, v), p0 in zip(model.state_dict().items(), init.state_dict().values()):
        torch.testing.assert_close(v.cpu(), cp["model"][k], rtol=2e-5, atol=2e-6, msg=lambda m: f"{k}: {m}")
        moved = max(moved, float((v.cpu() - p0).abs().max()))
    assert moved > 1e-4
    # ---- and the reference's checkpoint loads back into the HIP module (train.py:335-336 load_state_dict(strict=False))
    fresh = get_model(conf.model.name)(conf.model)
    res = fresh.load_state_dict(cp["model"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
