"""CPU-only checks: the C-ABI library builds/loads and exports every symbol declared in
include/gf_amd.h; config object, plugin registry and BaseModel semantics; the product path
fails loudly without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gf_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from glue_factory_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    so = ctypes.CDLL(lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(so, name), f"{name} declared in gf_amd.h but not exported"
    assert set(declared) == set(lib.SIGNATURES), set(declared) ^ set(lib.SIGNATURES)
    assert lib.load().gf_abi_version() == lib.ABI_VERSION


def test_conf_semantics():
    from glue_factory_amd.conf import Conf, ConfError
    base = Conf.create({"a": 1, "loss": {"gamma": 1.0, "fn": "nll"}, "layers": ["self", "cross"]})
    c = Conf.merge(base, {"a": 2, "loss": {"gamma": 0.5}})
    assert c.a == 2 and c.loss.gamma == 0.5 and c.loss.fn == "nll" and c["layers"] == ["self", "cross"]
    assert base.a == 1  # merge does not alias
    c.set_struct(True)
    with pytest.raises(ConfError):
        c.unknown
    with pytest.raises(ConfError):
        c["new"] = 1
    with pytest.raises(ConfError):
        Conf.merge(c, {"nope": 1})
    c.set_readonly(True)
    with pytest.raises(ConfError):
        c.a = 5
    assert c.to_container() == {"a": 2, "loss": {"gamma": 0.5, "fn": "nll"}, "layers": ["self", "cross"]}
    d = Conf.from_dotlist(["model.matcher.n_layers=4", "train.lr=1e-4", "x=true"])
    assert d.model.matcher.n_layers == 4 and d.x is True


def test_registry_and_base_model():
    from glue_factory_amd.base_model import BaseModel, get_model

    LG = get_model("glue_factory_amd.matchers.lightglue")
    assert LG.__name__ == "LightGlue"
    assert get_model("matchers.lightglue") is LG and get_model("lightglue") is LG
    with pytest.raises(RuntimeError):
        get_model("does.not.exist")

    class Child(BaseModel):
        default_conf = {"k": 3, "nested": {"x": 1}}
        required_data_keys = ["view0"]

        def _init(self, conf):
            self.lin = torch.nn.Linear(2, 2)
            self.bn = torch.nn.BatchNorm1d(2)

        def _forward(self, data):
            return {"y": data["view0"]}

        def loss(self, pred, data):
            raise NotImplementedError

    m = Child({"nested": {"x": 5}, "trainable": False, "freeze_batch_normalization": True, "extra": 1})
    assert m.conf.k == 3 and m.conf.nested.x == 5 and m.conf.name is None and m.conf.extra == 1
    assert all(not p.requires_grad for p in m.parameters())
    m.train()
    assert m.training and not m.bn.training
    with pytest.raises(AssertionError):
        m({})
    assert m({"view0": 1}) == {"y": 1}
    assert not m.is_initialized()
    m.load_state_dict(m.state_dict())
    assert m.is_initialized()


def test_lightglue_conf_state_dict_and_no_cpu_fallback():
    from glue_factory_amd.matchers.lightglue import LightGlue
    from oracle import lightglue_oracle as lgo
    m = LightGlue({"n_layers": 2, "checkpointed": True, "flash": False})
    ref = lgo.init_params(2, 256, 4, seed=0)
    sd = m.state_dict()
    assert set(sd) == set(ref) and all(sd[k].shape == ref[k].shape for k in ref)
    torch.testing.assert_close(sd["confidence_thresholds"], ref["confidence_thresholds"])
    data = {"keypoints0": torch.rand(1, 8, 2), "keypoints1": torch.rand(1, 8, 2),
            "descriptors0": torch.rand(1, 8, 256), "descriptors1": torch.rand(1, 8, 256)}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(data)
    assert LightGlue({"descriptor_dim": 128, "num_heads": 4, "n_layers": 1}).posenc.Wr.weight.shape == (16, 2)   # head_dim 32: generic kernels
    with pytest.raises(NotImplementedError):
        LightGlue({"descriptor_dim": 192, "num_heads": 4})                                                       # head_dim 48: no kernel


def test_fixed_length_positive_list_equals_nonzero():
    """LightGlue._gt_sparse: the fixed-length list built from gt_assignment_col0 carries the same positives
    and counts as nonzero() on the dense matrix (host logic, CPU)."""
    import torch
    from glue_factory_amd.matchers.lightglue import LightGlue
    from glue_factory_amd.synthetic import make_pairs
    data = make_pairs(3, 50, 60, dim=8, size=(640, 480), seed=4)
    dense = LightGlue._gt_sparse(data)
    fixed = LightGlue._gt_sparse(data, fixed=True)
    assert fixed["pos"][2].shape[0] == 3 * 50
    keep = fixed["pos"][2] >= 0
    got = torch.stack([t[keep] for t in fixed["pos"]], 1)
    ref = torch.stack(dense["pos"], 1)
    assert torch.equal(got, ref)                       # same (b, i, j) triples in the same (sorted) order
    for k in ("num_pos", "n0", "n1", "neg0", "neg1"):
        assert torch.equal(fixed[k], dense[k]), k
    data.pop("gt_assignment_col0")
    again = LightGlue._gt_sparse(data, fixed=True)     # producer without the key: dense path
    assert torch.equal(torch.stack(again["pos"], 1), ref)


def test_precast_cache_follows_parameter_versions():
    import torch
    from glue_factory_amd import ops
    if not torch.cuda.is_available():
        # CPU tensors are never cached (the product path is HIP-only): _lp is a plain cast
        w = torch.nn.Parameter(torch.randn(4, 4))
        ops.precast([w], torch.bfloat16, key="t")
        assert ops._lp(w, torch.bfloat16).dtype == torch.bfloat16
        assert id(w) not in ops._LP_CACHE


def test_sinkhorn_resident_plan_is_consistent():
    """gf_sinkhorn_plan (host-only): the distribution of the chip-resident Sinkhorn sweeps (csrc/sinkhorn_resident.h) over a
    256-CU device -- every pair gets whole workgroups, every row a wave, the rows of a wave fit its registers + LDS share, the
    column phase covers every float4 column, the LDS request stays inside the CU -- for the benchmarked geometry and around it."""
    import ctypes
    from glue_factory_amd import lib
    L = lib.load()
    out = (ctypes.c_int64 * 8)()
    seen = 0
    for B in (1, 4, 5, 7, 8, 9, 16, 20, 32, 33, 64):
        for M in (255, 1000, 2048):
            for N in (256, 512, 1024, 1280, 2048):
                for bwd in (0, 1):
                    ok = L.gf_sinkhorn_plan(B, M, N, 256, bwd, 1, out)
                    assert ok in (0, 1)
                    if B < 5:
                        assert ok == 0                                    # few pairs stay on the streaming kernels
                    if not ok:
                        continue
                    seen += 1
                    bc, wpp, nw, base, extra, cs, nsm, lds = (int(v) for v in out)
                    R, nvec = M + 1, N // 4 + 1
                    assert 5 <= bc <= min(B, 16) and wpp * bc <= 256 and nw == 4 * wpp and nsm == N // 256
                    assert base * nw + extra == R and 0 <= extra < nw               # every row has exactly one wave
                    assert base + (1 if extra else 0) <= 64                          # per-row scalars live one per lane
                    assert cs * wpp >= nvec and cs <= 32                             # the column phase covers every column
                    lds_rows = sum(max(0, base + (1 if w < extra else 0) - 12) for w in range(4))
                    assert lds == ((2 if bwd else 1) * (N // 4) + 256) * 16 + 32 + lds_rows * (N // 4) * 16
                    assert lds <= 160 * 1024 - 512
                    # launches cover the batch with balanced chunks
                    nch = -(-B // bc)
                    assert (nch - 1) * bc < B <= nch * bc
    assert seen > 50
    assert L.gf_sinkhorn_plan(32, 2048, 2048, 256, 0, 1, out) == 1 and tuple(out)[:5] == (8, 32, 128, 16, 1)
    assert L.gf_sinkhorn_plan(32, 2048, 2050, 256, 0, 1, out) == 0                      # N % 256 != 0: streaming
    assert L.gf_sinkhorn_plan(32, 2048, 2304, 256, 0, 1, out) == 0                      # N / 256 > 8
