"""Runs the REFERENCE's own `training()` (gluefactory/train.py:216-683) around a plugin model and a synthetic dataset.

`gluefactory.train` imports tensorboard at module level (absent here): a recording stand-in for
`torch.utils.tensorboard.SummaryWriter` is put into sys.modules first; omegaconf / kornia / h5py come from oracle/stubs.
The reference modules are the byte-compiled ones of oracle/_ref (GPU box) or /root/reference itself (build container).
Test infrastructure only."""
import argparse
import os
import sys
import types


class RecordingWriter:
    """What train.py:216-683 calls on its SummaryWriter; keeps the scalars for the test to read."""
    last = None

    def __init__(self, log_dir=None, **kw):
        self.log_dir, self.scalars, self.texts = log_dir, [], []
        RecordingWriter.last = self

    def add_scalar(self, key, value, step=None):
        self.scalars.append((key, float(value), step))

    def add_scalars(self, key, values, step=None):
        for k, v in values.items():
            self.scalars.append((f"{key}/{k}", float(v), step))

    def add_text(self, key, text, step=None):
        self.texts.append((key, text, step))

    def add_pr_curve(self, *a, **k):
        pass

    def add_figure(self, *a, **k):
        pass

    def add_histogram(self, *a, **k):
        pass

    def close(self):
        pass


def import_reference_train():
    """-> the reference's `gluefactory.train` module, or None when neither oracle/_ref nor /root/reference has it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import build_ref
    ok = build_ref.import_reference()
    if not ok or not os.path.exists(os.path.join(build_ref.OUT, "gluefactory", "train.pyc")):
        if not os.path.isdir("/root/reference/gluefactory"):
            return None
        for p in (build_ref.STUBS,):
            if p not in sys.path:
                sys.path.insert(0, p)
        if "/root/reference" not in sys.path:
            sys.path.append("/root/reference")
    if "torch.utils.tensorboard" not in sys.modules:
        tb = types.ModuleType("torch.utils.tensorboard")
        tb.SummaryWriter = RecordingWriter
        sys.modules["torch.utils.tensorboard"] = tb
    try:
        import gluefactory.train as tr
    except ImportError:
        return None
    return tr


def train_args(experiment, mixed_precision=None, compile_mode=None):
    """The argparse namespace of train.py:690-728 for a single-process run."""
    return argparse.Namespace(experiment=experiment, conf=None, mixed_precision=mixed_precision, compile=compile_mode,
                              cleanup_interval=120, overfit=False, restore=False, distributed=False, profile=False,
                              print_arch=False, detect_anomaly=False, log_it=True, no_eval_0=True, run_benchmarks=False,
                              dotlist=[], n_gpus=1)


def run_training(tr, conf_dict, output_dir, args):
    from omegaconf import OmegaConf
    conf = OmegaConf.create(conf_dict)
    tr.training(0, conf, output_dir, args)
    return RecordingWriter.last
