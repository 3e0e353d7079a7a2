import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from glue_factory_amd.matchers.lightglue import LightGlue
from glue_factory_amd.synthetic import make_pairs, to_device
from glue_factory_amd.train_step import TrainStep
from oracle import lightglue_oracle as lgo
L = 2
params = lgo.init_params(L, 256, 4, seed=21)
batches = [to_device(make_pairs(2, 256, dim=256, size=(640, 480), seed=30 + i), "cuda") for i in range(6)]
def ev(model, tag):
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return model(batches[0])["log_assignment"].clone()
for use_graph in (False, True):
    model = LightGlue({"n_layers": L}).cuda(); model.load_state_dict(params); model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)
    step = TrainStep(model, opt, amp_dtype=torch.bfloat16, graph=use_graph, graph_warmup=2)
    for b in batches[:4]:
        step(b)
    e1 = ev(model, "a")
    e2 = ev(model, "b")
    fresh = LightGlue({"n_layers": L}).cuda(); fresh.load_state_dict(model.state_dict())
    e3 = ev(fresh, "c")
    with torch.no_grad():
        model.eval(); e4 = model(batches[0])["log_assignment"]; fresh.eval(); e5 = fresh(batches[0])["log_assignment"]
    print(use_graph, "same model twice", float((e1 - e2).abs().max()), "vs fresh copy", float((e1 - e3).abs().max()), "fp32 model vs fresh", float((e4 - e5).abs().max()))
