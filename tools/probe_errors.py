#!/usr/bin/env python3
"""Diagnostic: learned-control probe error (max |u - u_ref| / (1e-4 max(1e-2, max|u_ref|))) and final-loss error of golden cases in
both matrix-product modes -- how close each mode sits to the probe bound of tests/test_gpu_parity.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from conftest import load_golden  # noqa: E402
from util_cases import make_pkg_solver  # noqa: E402

dev = torch.device("cuda:0")
names = sys.argv[1:] or ["llgc_d500_h64_logvar", "llgc_d200_h64_logvar", "llgc_d100_h64_logvar", "llgc_d300_h40_logvar"]
for name in names:
    rec = load_golden(name)
    exp = rec["expected"]
    for mode in ("fp32", "f16x3"):
        model = make_pkg_solver(rec["case"], dev, backend="native", mlp_dtype=mode)
        model.train()
        lerr = max(abs(g - w) / abs(w) for g, w in zip(model.loss_log, exp["loss_log"]))
        perr = 0.0
        if exp["probes"]:
            xp = torch.tensor(exp["probe_x"]).reshape(-1, model.d).to(dev)
            for pr in exp["probes"]:
                with torch.no_grad():
                    u = (-model.Z_n(xp, pr["t"])).cpu()
                want = torch.tensor(pr["minus_Z"]).reshape(u.shape)
                perr = max(perr, float((u - want).abs().max()) / (1e-4 * max(1e-2, float(want.abs().max()))))
        print("%s %s: loss rel err %.2e (bound 1e-4), probe error / bound %.2f, iterations %d, K %d" % (
            name, mode, lerr, perr, len(model.loss_log), rec["case"]["solver"]["K"]), flush=True)
