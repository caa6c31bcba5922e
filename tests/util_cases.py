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
    pb = getattr(psp, pspec["kind"])(device=device, **pspec["kwargs"])
    for call, kw in pspec.get("calls", []):              # e.g. compute_reference_solution(nx=...) before training
        getattr(pb, call)(**kw)
    return pb


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
        elif net["kind"] == "value_densenet":
            model.y_n = [psp.DenseNet(d_in=prob.d + 1, d_out=1, lr=kw["lr"], arch=net["arch"], seed=net["seed"]).to(device)]
        model.update_Phis()
    return model


def make_oracle(case, L=None):
    prob = orc.make_problem(case["problem"]["kind"], **case["problem"]["kwargs"])
    s = dict(case["solver"])
    cfg = orc.HJBConfig(K=s["K"], delta_t=s["delta_t"], lr=s["lr"], L=s["L"] if L is None else L, seed=s["seed"],
                        loss_method=s["loss_method"], time_approx=s["time_approx"],
                        learn_Y_0=s.get("learn_Y_0", False),
                        adaptive_forward_process=s["adaptive_forward_process"],
                        detach_forward=s["detach_forward"], random_X_0=s.get("random_X_0", False),
                        approx_method=s.get("approx_method", "control"))
    models = orc.hjb_build(prob, cfg)
    net = case.get("net")
    if net is not None:
        if net["kind"] == "tanh_mlp":
            z = orc.TanhMLP(prob.d + 1, prob.d, cfg.lr, seed=net["seed"], widths=net["widths"])
        elif net["kind"] == "value_densenet":
            z = orc.DenseNetOracle(prob.d + 1, 1, cfg.lr, arch=net["arch"], seed=net["seed"])
        else:
            z = orc.DenseNetOracle(prob.d + 1, prob.d, cfg.lr, arch=net["arch"], seed=net["seed"])
        models = (z, models[1], models[2])
    return prob, cfg, models


def flat_params(module):
    return torch.cat([p.detach().reshape(-1).cpu() for p in module.parameters()])


def general_oracle_run(case, L=None, trace=False):
    """Oracle run of a GeneralSolver / EllipticSolver golden case (families 'general', 'general_bounded', 'elliptic'): every
    solver switch of the case and its value net (kind 'densenet' | 'user_tanh2' | 'densenet_tanh').  Returns (problem, out)."""
    import numpy as np
    kw = dict(case["problem"]["kwargs"])
    kw.update(case["problem"].get("attrs", {}))          # attributes set on the instance -> oracle keywords
    if "numpy_seed" in case:
        np.random.seed(case["numpy_seed"])
    prob = orc.make_problem(case["problem"]["kind"], **kw)
    s = case["solver"]
    common = dict(K=s["K"], N=s["N"], delta_t=s["delta_t"], lr=s["lr"], L=s["L"] if L is None else L, seed=s["seed"],
                  K_boundary=s["K_boundary"], loss_method=s["loss_method"],
                  adaptive_forward_process=s.get("adaptive_forward_process", False),
                  uniform_square=s.get("uniform_square", False), loss_with_stopped=s.get("loss_with_stopped", False),
                  K_test_log=s.get("K_test_log"), sample_center=s.get("sample_center", False))
    net = case.get("net")
    if case["family"] == "elliptic":
        cfg = orc.EllipticConfig(alpha=tuple(s.get("alpha", (1.0, 1.0))), boundary_type=s.get("boundary_type", "Dirichlet"),
                                 **common)
        return prob, orc.elliptic_train(prob, cfg, V=orc.elliptic_build(prob, cfg, net=net), trace=trace)
    cfg = orc.GeneralConfig(alpha=tuple(s["alpha"]), **common)
    return prob, orc.general_train(prob, cfg, V=orc.general_build(prob, cfg, net=net), trace=trace)


def make_pkg_value_net(net, d_in, lr, device):
    """The package-side value net of a golden case (kinds as in tests/golden/make_golden.py)."""
    kind = net.get("kind", "densenet")
    if kind == "densenet_tanh":
        return psp.DenseNet_tanh(d_in=d_in, d_out=1, lr=lr, arch=net["arch"], seed=net["seed"]).to(device)
    cls = psp.DenseNet_tanh_2 if kind == "user_tanh2" else psp.DenseNet
    return cls(d_in=d_in, d_out=1, lr=lr, arch=net["arch"], seed=net["seed"]).to(device)
