"""Helpers shared by the tests: build package objects and oracle objects from a golden case."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import path_space_pde_solver_amd as psp  # noqa: E402
from oracle import pathspace_oracle as orc  # noqa: E402  (tests are allowed to use the oracle)


def make_pkg_problem(pspec, device):
    return getattr(psp, pspec["kind"])(device=device, **pspec["kwargs"])


def make_pkg_solver(case, device, backend="auto", noise="reference", **over):
    prob = make_pkg_problem(case["problem"], device)
    kw = dict(case["solver"])
    kw.update(over)
    net = case.get("net")
    widths = tuple(net["widths"]) if net is not None and net["kind"] == "tanh_mlp" else (30, 30)
    model = psp.Solver(name=case["name"], problem=prob, verbose=False, device=device, backend=backend,
                       noise=noise, widths=widths, **kw)
    if net is not None:
        if net["kind"] == "tanh_mlp":
            model.z_n = psp.MySequential(prob.d + 1, prob.d, kw["lr"], seed=net["seed"], widths=widths)
        elif net["kind"] == "densenet":
            model.z_n = psp.DenseNet(d_in=prob.d + 1, d_out=prob.d, lr=kw["lr"], arch=net["arch"], seed=net["seed"])
        model.update_Phis()
    return model


def make_oracle(case, L=None):
    prob = orc.make_problem(case["problem"]["kind"], **case["problem"]["kwargs"])
    s = dict(case["solver"])
    cfg = orc.HJBConfig(K=s["K"], delta_t=s["delta_t"], lr=s["lr"], L=s["L"] if L is None else L, seed=s["seed"],
                        loss_method=s["loss_method"], time_approx=s["time_approx"],
                        learn_Y_0=s.get("learn_Y_0", False),
                        adaptive_forward_process=s["adaptive_forward_process"],
                        detach_forward=s["detach_forward"], random_X_0=s.get("random_X_0", False))
    models = orc.hjb_build(prob, cfg)
    net = case.get("net")
    if net is not None:
        if net["kind"] == "tanh_mlp":
            z = orc.TanhMLP(prob.d + 1, prob.d, cfg.lr, seed=net["seed"], widths=net["widths"])
        else:
            z = orc.DenseNetOracle(prob.d + 1, prob.d, cfg.lr, arch=net["arch"], seed=net["seed"])
        models = (z, models[1], models[2])
    return prob, cfg, models


def flat_params(module):
    return torch.cat([p.detach().reshape(-1).cpu() for p in module.parameters()])
