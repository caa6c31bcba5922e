"""Per-trajectory comparison of the genl forward outputs (V(X_N), Y_N, X_N) with a CPU rollout of the same net and noise
(diagnostic; repo root resolved from this file)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from conftest import load_golden  # noqa: E402
from test_general_composite_golden import build as build_pkg  # noqa: E402
from util_cases import general_oracle_run  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "committor_d10_tanh2_notebook_diffusion"
case = load_golden(name)["case"]
dev = torch.device("cuda:0")
for nw in (None, "8"):
    if nw:
        os.environ["PSP_GENL_NW"] = nw
    prob, model = build_pkg(case, device=dev, backend="native", L=1)
    V0 = [p.detach().clone().cpu() for p in model.V.parameters()]
    model.train()
    plan = model._gen_plan
    oprob, ref = general_oracle_run(case, L=1, trace=True)
    tr = ref["traces"][0]
    # CPU rollout with the INITIAL parameters
    import copy
    Vc = copy.deepcopy(model.V).cpu()
    with torch.no_grad():
        for p, q in zip(Vc.parameters(), V0):
            p.copy_(q)
    X = tr["X0"].clone().requires_grad_(True)
    dt = torch.tensor(case["solver"]["delta_t"]); sq = torch.sqrt(dt)
    Y = Vc(X).squeeze().detach()
    stopped = torch.zeros(X.shape[0], dtype=torch.bool)
    Xc = X.detach().clone()
    for n, xi in enumerate(tr["xi"]):
        Xg = Xc.clone().requires_grad_(True)
        Z, = torch.autograd.grad(Vc(Xg).squeeze().sum(), Xg)
        r = Xc.norm(dim=1)
        inside = (r > 1.0) & (r < 2.0)
        act = inside & ~stopped
        Y = Y + (Z * xi).sum(1) * sq * act.float()
        Xc = torch.where(act.unsqueeze(1), Xc + xi * sq, Xc)
        stopped = stopped | ~inside
    VN = Vc(Xc).squeeze().detach()
    print("NW", plan.sizes.waves_per_tile, "loss", model.loss_log, "oracle", ref["loss_log"])
    print("  max |XN err|", float((plan.XN.cpu() - Xc).abs().max()))
    eY, eV = (plan.YN.cpu() - Y).abs(), (plan.VN.cpu() - VN).abs()
    print("  max |YN err| %.3e at %d, max |VN err| %.3e at %d; mean r^2 ref %.3e got %.3e" % (
        float(eY.max()), int(eY.argmax()), float(eV.max()), int(eV.argmax()), float(((VN - Y) ** 2).mean()),
        float(((plan.VN - plan.YN) ** 2).mean())))
    bad = torch.nonzero(eY > 1e-4).flatten().tolist()
    print("  trajectories with |YN err| > 1e-4:", bad[:40], "of", eY.numel())
