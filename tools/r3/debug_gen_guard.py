"""Debug: per-iteration range flags of the GeneralSolver guard (heat_d6, |x| ~ 3e5)."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
from util_cases import psp
from conftest import load_golden
dev = torch.device("cuda:0")
case = load_golden("heat_d6_diffusion")["case"]
for mlp, guard in (("f16x3", True), ("f16x3", False), ("fp32", True)):
    prob = getattr(psp, case["problem"]["kind"])(device=dev, **case["problem"]["kwargs"])
    prob.boundary_distance = 3.0e5
    s = dict(case["solver"]); s.update(L=3, mlp_dtype=mlp, range_guard=guard)
    model = psp.GeneralSolver(problem=prob, name="dbg", verbose=False, device=dev, backend="native", **s)
    with torch.no_grad():
        for p in model.V.parameters():
            p.mul_(1e-1)
    torch.manual_seed(model.seed)
    plan = model._choose_plan()
    for l in range(3):
        loss, kc = plan.iteration(l)
        torch.cuda.synchronize()
        print(mlp, guard, "iter", l, "loss", float(loss), "flag", None if plan.range_flag is None else plan.range_flag.tolist(),
              "nanV", int(torch.isnan(plan.VN).sum()), "nanY", int(torch.isnan(plan.YN).sum()), "kc", int(kc), "maxabsX0?", float(plan.XN_k.abs().max()))
