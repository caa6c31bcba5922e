"""Boundary term of the committor goldens per iteration: g on the host copy against g on the device batch (diagnostic)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from conftest import load_golden  # noqa: E402
from test_general_composite_golden import build as build_pkg  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "committor_d3_elliptic_diffusion"
rec = load_golden(name)
dev = torch.device("cuda:0")
prob, model = build_pkg(rec["case"], device=dev, backend="native")
orig = model.boundary_residual


def spy(X_b):
    host = model._Xb_host
    gh = prob.g(host)
    gd = prob.g(X_b).cpu()
    r_h = torch.sqrt(torch.sum(host ** 2, 1))
    r_d = torch.sqrt(torch.sum(X_b ** 2, 1)).cpu()
    print("  boundary: K_b %d, g host sum %.0f, g device sum %.0f, flips %d, max |X_b - host| %.2e, r_host-1 inner %s" % (
        host.shape[0], float(gh.sum()), float(gd.sum()), int((gh != gd).sum()), float((X_b.cpu() - host).abs().max()),
        ["%.1e" % float(v - 1) for v in r_h[:host.shape[0] // 2][:6]]))
    out = orig(X_b)
    print("  boundary residual %.8f" % float(out))
    return out


model.boundary_residual = spy
model.train()
print("loss", model.loss_log, "golden", rec["expected"]["loss_log"], "K_log", model.K_log, rec["expected"]["K_log"])
