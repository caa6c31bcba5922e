#!/usr/bin/env python3
"""Generate golden vectors by running the *reference* implementation on CPU.

Runs only in the build container (needs /root/reference, read-only). The reference
hard-codes ``device('cuda')`` (problems.py:11, solver.py:36,947); it is redirected to
CPU without touching the reference by swapping the module-local name ``pt`` for a
proxy that forwards everything to torch except ``device(...)``.

What is written under tests/golden/ is data only: the case configuration (inputs)
and the reference's outputs (loss_log, control on a probe grid, parameter
checksums, noise fingerprint).  No reference source text is stored.

Usage:  python tests/golden/make_golden.py            (rewrites tests/golden/*.json)
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get("PSP_REFERENCE_DIR", "/root/reference")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402


class _PtProxy(types.ModuleType):
    """torch with device(...) pinned to CPU."""

    def __init__(self):
        super().__init__("pt_cpu_proxy")

    def __getattr__(self, name):
        if name == "device":
            return lambda *a, **k: torch.device("cpu")
        return getattr(torch, name)


import function_space as ref_fs  # noqa: E402
import problems as ref_pb  # noqa: E402
import utilities as ref_ut  # noqa: E402
import solver as ref_sv  # noqa: E402

_proxy = _PtProxy()
for _m in (ref_fs, ref_pb, ref_ut, ref_sv):
    _m.pt = _proxy
ref_pb.device = torch.device("cpu")

OUT = os.path.dirname(os.path.abspath(__file__))


def f32list(t):
    return [float(v) for v in t.detach().reshape(-1).to(torch.float32).tolist()]


def param_fingerprint(module):
    """Order-stable checksums of a module's parameters (fp64 sums of fp32 values)."""
    out = []
    for name, p in module.named_parameters():
        q = p.detach().to(torch.float64)
        out.append({"name": name, "shape": list(p.shape), "sum": float(q.sum()),
                    "abs_sum": float(q.abs().sum()),
                    "head": f32list(p.detach().reshape(-1)[:4])})
    return out


def wide_tanh_mlp(d_in, d_out, widths, lr, seed):
    """A reference MySequential whose hidden widths are replaced.

    function_space.py:181 hard-codes [d_in,30,30,d_out]; the benchmark configs ask
    for 2x64.  The object is built by the reference class, then its layers are
    rebuilt with the same RNG recipe the reference constructor uses
    (function_space.py:180-188: manual_seed, Linear ctor per layer, then
    normal_(0,0.01) on weight and bias layer by layer) so that the init stream is
    the one a width-configurable MySequential would have produced.
    """
    net = ref_fs.MySequential(d_in=d_in, d_out=d_out, lr=lr, seed=seed)
    torch.manual_seed(seed)
    dims = [d_in] + list(widths) + [d_out]
    net.nn_dims = dims
    net.linears = torch.nn.ModuleList(
        [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
    net.activations = torch.nn.ModuleList([torch.nn.Tanh() for _ in range(len(dims) - 2)])
    net.optim = torch.optim.Adam(net.parameters(), lr=lr)
    for lin in net.linears:
        torch.nn.init.normal_(lin.weight, 0, 0.01)
        torch.nn.init.normal_(lin.bias, 0, 0.01)
    return net


class QuadraticOnBox:
    """NOT a reference class: a duck-typed user problem handed to the REFERENCE solvers to pin the
    'square' exit tests (solver.py:1125-1129, :762-767): b = 0, sigma = scale I, h = -|z|^2/2 (or 0), data |x|^2."""

    def __init__(self, d=2, T=0.5, X_l=-1.0, X_r=1.0, one_boundary=False, scale=1.0, parabolic=True, quad_h=True):
        self.name, self.d, self.T = "Quadratic on box", d, T
        self.B = scale * torch.eye(d)
        self.boundary, self.boundary_type = "square", "Dirichlet"
        self.X_l, self.X_r, self.one_boundary = X_l, X_r, one_boundary
        self.parabolic, self.quad_h = parabolic, quad_h

    def b(self, x):
        return torch.zeros(x.shape)

    def sigma(self, x):
        return self.B

    def _hq(self, z):
        return -0.5 * torch.sum(z ** 2, dim=1) if self.quad_h else torch.zeros(z.shape[0])

    def h(self, *args):            # (t, x, y, z) for GeneralSolver, (x, y, z) for EllipticSolver
        return self._hq(args[-1])

    def f(self, x, t=None):
        return torch.sum(x ** 2, 1) if self.parabolic else torch.zeros(x.shape[0])

    def g(self, x, t=None):
        return torch.sum(x ** 2, 1) + (self.T - t) if self.parabolic else torch.sum(x ** 2, 1)

    def v_true(self, x, t=None):
        return torch.sum(x ** 2, 1)


class UserTanh2Net(ref_fs.DenseNet):
    """NOT a reference class: a user-defined value net handed to the REFERENCE solvers through the `model.V = ...` extension
    point, as `Committor function.ipynb` does with the tanh(.)**2 variant of DenseNet it defines for itself.
    The reference constructor draws the parameters; only the hidden nonlinearity differs."""

    def forward(self, x):
        n = len(self.nn_dims) - 1
        for i in range(n - 1):
            x = torch.cat([x, torch.tanh(torch.matmul(x, self.W[2 * i]) + self.W[2 * i + 1]) ** 2], dim=1)
        return torch.matmul(x, self.W[2 * n - 2]) + self.W[2 * n - 1]


def make_value_net(net, d_in, lr):
    kind = net.get("kind", "densenet")
    if kind == "densenet_tanh":
        return ref_fs.DenseNet_tanh(d_in=d_in, d_out=1, lr=lr, arch=net["arch"], seed=net["seed"])
    cls = UserTanh2Net if kind == "user_tanh2" else ref_fs.DenseNet
    return cls(d_in=d_in, d_out=1, lr=lr, arch=net["arch"], seed=net["seed"])


def make_problem(spec):
    kind = spec["kind"]
    kw = dict(spec["kwargs"])
    pb = QuadraticOnBox(**kw) if kind == "QuadraticOnBox" else getattr(ref_pb, kind)(**kw)
    for k, v in spec.get("attrs", {}).items():      # attributes callers set on the instance (e.g. boundary_type)
        setattr(pb, k, v)
    for call, kw in spec.get("calls", []):          # what the notebooks call before training (compute_reference_solution[_2])
        getattr(pb, call)(**kw)
    return pb


def probe_points(d, n=5, seed=7):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, d, generator=g)


def run_solver_case(case):
    problem = make_problem(case["problem"])
    skw = dict(case["solver"])
    model = ref_sv.Solver(name=case["name"], problem=problem, verbose=False, **skw)
    net = case.get("net")
    if net is not None:
        if net["kind"] == "tanh_mlp":
            model.z_n = wide_tanh_mlp(problem.d + 1, problem.d, net["widths"], skw["lr"], net["seed"])
        elif net["kind"] == "densenet":
            model.z_n = ref_fs.DenseNet(d_in=problem.d + 1, d_out=problem.d, lr=skw["lr"],
                                        arch=net["arch"], seed=net["seed"])
        elif net["kind"] == "value_densenet":             # value-function ansatz with another value net (solver.py:97, 142-162)
            model.y_n = [ref_fs.DenseNet(d_in=problem.d + 1, d_out=1, lr=skw["lr"], arch=net["arch"], seed=net["seed"])]
        model.update_Phis()
    vf = skw.get("approx_method") == "value_function"
    init_fp = param_fingerprint(model.y_n[0]) if vf else (param_fingerprint(model.z_n) if not isinstance(model.z_n, list) else None)
    model.train()
    xp = probe_points(problem.d)
    probes = []
    for t in ([] if vf else case.get("probe_times", [0.0])):
        with torch.no_grad():
            z = model.Z_n(xp, torch.tensor(t))
        probes.append({"t": t, "minus_Z": f32list(-z)})
    ref_tables = None
    if hasattr(problem, "u") and isinstance(getattr(problem, "u"), np.ndarray):
        tabs = [problem.u] + ([problem.u_2] if hasattr(problem, "u_2") else [])
        ref_tables = [{"shape": list(t.shape), "sum": float(t.sum()), "abs_sum": float(np.abs(t).sum()),
                       "probe": [float(v) for v in t[::max(1, t.shape[0] // 3), ::max(1, t.shape[1] // 5)].reshape(-1)]} for t in tabs]
    res = {
        "ref_tables": ref_tables,
        "N": model.N, "p": int(model.p),
        "loss_log": [float(v) for v in model.loss_log],
        "u_L2_loss": [float(v) for v in model.u_L2_loss],
        "Y_0_log": [float(v) for v in model.Y_0_log],
        "init_params": init_fp,
        "final_params": param_fingerprint(model.y_n[0]) if vf else (param_fingerprint(model.z_n) if not isinstance(model.z_n, list) else None),
        "y_0_final": float(model.y_0.Y_0.detach()[0]) if hasattr(model, "y_0") else None,
        "probe_x": f32list(xp), "probes": probes,
    }
    return res


def run_is_case(case):
    """Train a few iterations, then run the reference's importance-sampling evaluation
    (utilities.py:287-359) on a fresh seed; also the in-loop variant (solver.py:521-528)."""
    problem = make_problem(case["problem"])
    skw = dict(case["solver"])
    model = ref_sv.Solver(name=case["name"], problem=problem, verbose=False, **skw)
    model.train()
    torch.manual_seed(case["is_seed"])
    mean_IS, var_IS, rel_IS = ref_ut.do_importance_sampling_me(problem, model, case["is_K"],
                                                              delta_t=case["is_delta_t"])
    return {"N": model.N, "loss_log": [float(v) for v in model.loss_log],
            "IS_rel_log": [float(v) for v in model.IS_rel_log],
            "mean_IS": float(mean_IS), "variance_IS": float(var_IS), "rel_error_IS": float(rel_IS),
            "final_params": param_fingerprint(model.z_n)}


def run_general_case(case):
    problem = make_problem(case["problem"])
    skw = dict(case["solver"])
    model = ref_sv.GeneralSolver(problem=problem, name=case["name"], verbose=False, **skw)
    net = case.get("net")
    if net is not None:
        model.V = make_value_net(net, problem.d + 1, skw["lr"])
    init_fp = param_fingerprint(model.V)
    model.train()
    xp = probe_points(problem.d)
    tp = torch.full((xp.shape[0], 1), 0.5 * problem.T)
    with torch.no_grad():
        v = model.V(torch.cat([xp, tp], 1)).squeeze()
    return {
        "loss_log": [float(v_) for v_ in model.loss_log],
        "K_log": [int(v_) for v_ in model.K_log],
        "V_test_L2": [float(v_) for v_ in model.V_test_L2],
        "init_params": init_fp, "final_params": param_fingerprint(model.V),
        "probe_x": f32list(xp), "probe_t": 0.5 * problem.T, "probe_V": f32list(v),
    }


def run_general_bounded_case(case):
    np.random.seed(case.get("numpy_seed", 0))        # GeneralSolver.train does not seed numpy (the square boundary shuffle)
    return run_general_case(case)


def run_elliptic_case(case):
    problem = make_problem(case["problem"])
    skw = dict(case["solver"])
    model = ref_sv.EllipticSolver(problem=problem, name=case["name"], verbose=False, **skw)
    net = case.get("net")
    if net is not None:
        model.V = make_value_net(net, problem.d, skw["lr"])
    init_fp = param_fingerprint(model.V)
    model.train()
    xp = 0.4 * probe_points(problem.d)
    with torch.no_grad():
        v = model.V(xp).squeeze()
    return {
        "loss_log": [float(v_) for v_ in model.loss_log],
        "K_log": [int(v_) for v_ in model.K_log],
        "V_L2_log": [float(v_) for v_ in model.V_L2_log],
        "V_test_L2": [float(v_) for v_ in model.V_test_L2], "V_test_abs": [float(v_) for v_ in model.V_test_abs],
        "init_params": init_fp, "final_params": param_fingerprint(model.V),
        "probe_x": f32list(xp), "probe_V": f32list(v),
    }


HJB = dict(loss_method="log-variance", time_approx="inner", adaptive_forward_process=True,
           detach_forward=True, early_stopping_time=None)

CASES = [
    # BASELINE.json configs[0]: d=2 HJB-LQ, K=128, N=20, log-variance
    dict(name="lqgc_d2_logvar", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=2, off_diag=0.1, T=1, seed=42, delta_t=0.05)),
         solver=dict(HJB, L=6, lr=0.01, seed=42, delta_t=0.05, K=128), probe_times=[0.0, 0.5]),
    # same without the per-step u_true logging
    dict(name="lqgc_d2_logvar_noul2", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=2, off_diag=0.1, T=1, seed=42, delta_t=0.05)),
         solver=dict(HJB, L=6, lr=0.01, seed=42, delta_t=0.05, K=128, u_l2_error_flag=False)),
    # u_L2 logging left ON (the reference's default, every HJB notebook runs with it): LLGC's u* does not depend on x
    dict(name="llgc_d8_logvar_ul2", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=8, off_diag=0.05, T=0.4, seed=42)),
         solver=dict(HJB, L=5, lr=0.003, seed=42, delta_t=0.02, K=96)),
    dict(name="llgc_d40_moment_ul2", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=40, off_diag=0.02, T=0.2, seed=42)),
         solver=dict(HJB, loss_method="moment", learn_Y_0=True, L=4, lr=0.002, seed=42, delta_t=0.01, K=80)),
    # BASELINE.json configs[1] shape with the reference's default 2x30 net
    dict(name="llgc_d100_h30_logvar", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=100, off_diag=0.01, T=0.5, seed=42)),
         solver=dict(HJB, L=4, lr=0.001, seed=42, delta_t=0.01, K=1024, u_l2_error_flag=False)),
    # BASELINE.json configs[1]: d=100, K=1024, N=50, 2x64 tanh MLP
    dict(name="llgc_d100_h64_logvar", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=100, off_diag=0.01, T=0.5, seed=42)),
         solver=dict(HJB, L=4, lr=0.001, seed=42, delta_t=0.01, K=1024, u_l2_error_flag=False),
         net=dict(kind="tanh_mlp", widths=[64, 64], seed=123), probe_times=[0.0, 0.25]),
    # BASELINE.json configs[3] / configs[4] shapes (d=200, d=500; 2x64 tanh MLP) at a K the reference runs in seconds
    dict(name="llgc_d200_h64_logvar", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=200, off_diag=0.1 / 200 ** 0.5, T=0.2, seed=42)),
         solver=dict(HJB, L=3, lr=0.001, seed=42, delta_t=0.01, K=200, u_l2_error_flag=False),
         net=dict(kind="tanh_mlp", widths=[64, 64], seed=123), probe_times=[0.0, 0.1]),
    dict(name="llgc_d500_h64_logvar", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=500, off_diag=0.1 / 500 ** 0.5, T=0.1, seed=42)),
         solver=dict(HJB, L=3, lr=0.001, seed=42, delta_t=0.01, K=72, u_l2_error_flag=False),
         net=dict(kind="tanh_mlp", widths=[64, 64], seed=123), probe_times=[0.0]),
    # shapes without an exact kernel instance: run on the next instance up after zero padding (native_shapes.py)
    dict(name="llgc_d7_default_logvar", family="solver",                       # (7, 30) -> (16, 32)
         problem=dict(kind="LLGC", kwargs=dict(d=7, off_diag=0.1, T=0.3, seed=42)),
         solver=dict(HJB, L=4, lr=0.002, seed=42, delta_t=0.01, K=100, u_l2_error_flag=False)),
    dict(name="lqgc_d33_h50_logvar", family="solver",                          # (33, 50) -> (48, 64)
         problem=dict(kind="LQGC", kwargs=dict(d=33, off_diag=0.05, T=0.5, seed=42, delta_t=0.05)),
         solver=dict(HJB, L=3, lr=0.002, seed=42, delta_t=0.05, K=130, u_l2_error_flag=False),
         net=dict(kind="tanh_mlp", widths=[50, 50], seed=123), probe_times=[0.0, 0.25]),
    dict(name="dw_d70_h64_logvar", family="solver",                            # (70, 64) -> (80, 64)
         problem=dict(kind="DoubleWell_multidim", kwargs=dict(d=70, d_1=35, d_2=35, T=0.2, eta=0.02, kappa=1.0)),  # small eta: keeps mean(D)^2 / var(D) moderate so the reference's fp32 loss is itself accurate to 1e-5
         solver=dict(HJB, L=3, lr=0.002, seed=42, delta_t=0.01, K=96, u_l2_error_flag=False),
         net=dict(kind="tanh_mlp", widths=[64, 64], seed=123)),
    dict(name="llgc_d105_h64_logvar", family="solver",                         # dense (112, 64) exceeds the LDS -> wide (128, 64)
         problem=dict(kind="LLGC", kwargs=dict(d=105, off_diag=0.01, T=0.2, seed=42)),
         solver=dict(HJB, L=3, lr=0.001, seed=42, delta_t=0.01, K=80, u_l2_error_flag=False),
         net=dict(kind="tanh_mlp", widths=[64, 64], seed=123)),
    dict(name="llgc_d300_h40_logvar", family="solver",                         # (300, 40) -> wide (320, 64)
         problem=dict(kind="LLGC", kwargs=dict(d=300, off_diag=0.1 / 300 ** 0.5, T=0.1, seed=42)),
         solver=dict(HJB, L=3, lr=0.001, seed=42, delta_t=0.01, K=48, u_l2_error_flag=False),
         net=dict(kind="tanh_mlp", widths=[40, 40], seed=123), probe_times=[0.0]),
    # gradients THROUGH the state path: adaptive_forward_process=True with detach_forward=False (the reference's default
    # flags, solver.py:451-469), and the relative-entropy loss (solver.py:179-180, 484-486)
    dict(name="lqgc_d2_attached_logvar", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=2, off_diag=0.1, T=1, seed=42, delta_t=0.05)),
         solver=dict(HJB, detach_forward=False, L=5, lr=0.01, seed=42, delta_t=0.05, K=128, u_l2_error_flag=False)),
    dict(name="llgc_d100_h64_attached_logvar", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=100, off_diag=0.01, T=0.3, seed=42)),
         solver=dict(HJB, detach_forward=False, L=3, lr=0.001, seed=42, delta_t=0.01, K=256, u_l2_error_flag=False),
         net=dict(kind="tanh_mlp", widths=[64, 64], seed=123)),
    dict(name="dw_d10_attached_moment", family="solver",
         problem=dict(kind="DoubleWell_multidim", kwargs=dict(d=10, d_1=5, d_2=5, T=0.3, eta=0.5, kappa=2.0)),
         solver=dict(HJB, detach_forward=False, loss_method="moment", learn_Y_0=True, L=4, lr=0.005, seed=42,
                     delta_t=0.01, K=160, u_l2_error_flag=False)),
    dict(name="lqgc_d2_attached_cross_entropy", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=2, off_diag=0.1, T=1, seed=42, delta_t=0.05)),
         solver=dict(HJB, detach_forward=False, loss_method="cross_entropy", L=5, lr=0.01, seed=42, delta_t=0.05, K=128,
                     u_l2_error_flag=False)),
    dict(name="llgc_d200_nonadaptive_logvar", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=200, off_diag=0.1 / 200 ** 0.5, T=0.1, seed=42)),
         solver=dict(HJB, adaptive_forward_process=False, L=3, lr=0.001, seed=42, delta_t=0.01, K=64, u_l2_error_flag=False),
         net=dict(kind="tanh_mlp", widths=[64, 64], seed=123)),
    dict(name="lqgc_d4_relative_entropy", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=4, off_diag=0.1, T=0.5, seed=42, delta_t=0.05)),
         solver=dict(HJB, detach_forward=False, loss_method="relative_entropy", L=5, lr=0.01, seed=42, delta_t=0.05,
                     K=128, u_l2_error_flag=False)),
    dict(name="llgc_d20_relative_entropy_detached", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=20, off_diag=0.05, T=0.3, seed=42)),
         solver=dict(HJB, detach_forward=True, loss_method="relative_entropy", L=4, lr=0.002, seed=42, delta_t=0.01,
                     K=200, u_l2_error_flag=False)),
    # DenseNet swapped in as the control net (notebook extension point)
    dict(name="llgc_d100_densenet64_logvar", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=100, off_diag=0.01, T=0.5, seed=42)),
         solver=dict(HJB, L=3, lr=0.001, seed=42, delta_t=0.01, K=1024, u_l2_error_flag=False),
         net=dict(kind="densenet", arch=[64, 64], seed=42)),
    # elementwise double-well drift, B = I
    dict(name="dw_d10_logvar", family="solver",
         problem=dict(kind="DoubleWell_multidim", kwargs=dict(d=10, d_1=5, d_2=5, T=0.5, eta=0.5, kappa=2.0)),
         solver=dict(HJB, L=4, lr=0.005, seed=42, delta_t=0.01, K=256, u_l2_error_flag=False)),
    # structured OU: A=-I, B=I (off_diag=0)
    dict(name="llgc_d20_diag_logvar", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=20, off_diag=0.0, T=0.3, seed=42)),
         solver=dict(HJB, L=4, lr=0.002, seed=42, delta_t=0.01, K=200, u_l2_error_flag=False)),
    # moment loss with learnable Y_0
    dict(name="lqgc_d2_moment", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=2, off_diag=0.1, T=1, seed=42, delta_t=0.05)),
         solver=dict(HJB, loss_method="moment", learn_Y_0=True, L=5, lr=0.01, seed=42, delta_t=0.05,
                     K=128, u_l2_error_flag=False)),
    # losses that only change the per-trajectory weights
    dict(name="lqgc_d2_variance", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=2, off_diag=0.1, T=1, seed=42, delta_t=0.05)),
         solver=dict(HJB, loss_method="variance", L=5, lr=0.01, seed=42, delta_t=0.05, K=128,
                     u_l2_error_flag=False)),
    # variance loss with a learnable Y_0: Y = y_0 + ... enters var(exp(-g + Y)), so dL/dY_0 = sum_k w_k != 0 (solver.py:171-172, 372-374)
    dict(name="lqgc_d2_variance_learn_y0", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=2, off_diag=0.1, T=1, seed=42, delta_t=0.05)),
         solver=dict(HJB, loss_method="variance", learn_Y_0=True, L=6, lr=0.01, seed=42, delta_t=0.05, K=128,
                     u_l2_error_flag=False)),
    dict(name="lqgc_d2_cross_entropy", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=2, off_diag=0.1, T=1, seed=42, delta_t=0.05)),
         solver=dict(HJB, loss_method="cross_entropy", L=5, lr=0.01, seed=42, delta_t=0.05, K=128,
                     u_l2_error_flag=False)),
    dict(name="llgc_d8_cross_entropy_nonadaptive", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=8, off_diag=0.05, T=0.4, seed=42)),
         solver=dict(HJB, loss_method="cross_entropy", adaptive_forward_process=False, L=4, lr=0.003, seed=42,
                     delta_t=0.02, K=160, u_l2_error_flag=False)),
    # random initial points
    dict(name="lqgc_d4_randx0", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=4, off_diag=0.1, T=0.5, seed=42, delta_t=0.05)),
         solver=dict(HJB, L=4, lr=0.01, seed=3, delta_t=0.05, K=96, random_X_0=True,
                     u_l2_error_flag=False)),
    # uncontrolled forward process (c = 0): gradient keeps the Z*dt term
    dict(name="llgc_d8_nonadaptive", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=8, off_diag=0.05, T=0.4, seed=42)),
         solver=dict(HJB, adaptive_forward_process=False, L=4, lr=0.003, seed=42, delta_t=0.02, K=160,
                     u_l2_error_flag=False)),
    # default time_approx='outer' (one DenseNet per step) -- composite plan only
    dict(name="lqgc_d2_outer", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=2, off_diag=0.1, T=0.5, seed=42, delta_t=0.05)),
         solver=dict(loss_method="log-variance", time_approx="outer", adaptive_forward_process=True,
                     detach_forward=True, early_stopping_time=None, L=3, lr=0.01, seed=42,
                     delta_t=0.05, K=64, u_l2_error_flag=False)),
    # DenseNet controls on the native plan (csrc/hjbd_kernels.h): 'outer' with a padded shape and learn_Y_0, and a
    # DenseNet(d+1 -> d) with a generic loss on the double well
    dict(name="llgc_d12_outer_moment", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=12, off_diag=0.05, T=0.2, seed=42)),
         solver=dict(loss_method="moment", learn_Y_0=True, time_approx="outer", adaptive_forward_process=True,
                     detach_forward=True, early_stopping_time=None, L=4, lr=0.003, seed=42, delta_t=0.02, K=80,
                     u_l2_error_flag=False)),
    dict(name="dw_d20_densenet_nonadaptive", family="solver",
         problem=dict(kind="DoubleWell_multidim", kwargs=dict(d=20, d_1=10, d_2=10, T=0.2, eta=0.3, kappa=1.0)),
         solver=dict(HJB, adaptive_forward_process=False, L=4, lr=0.002, seed=42, delta_t=0.01, K=96, u_l2_error_flag=False),
         net=dict(kind="densenet", arch=[40, 40], seed=7), probe_times=[0.0, 0.1]),
    dict(name="lqgc_d6_densenet_variance", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=6, off_diag=0.1, T=0.5, seed=42, delta_t=0.05)),
         solver=dict(HJB, loss_method="variance", L=4, lr=0.005, seed=42, delta_t=0.05, K=128, u_l2_error_flag=False),
         net=dict(kind="densenet", arch=[24, 24], seed=7), probe_times=[0.0]),
    # DenseNet controls with gradients THROUGH the state path (detach_forward=False: the constructor default) and the
    # relative-entropy loss: native through psp_dnet_adjoint_sweep (csrc/hjbd_kernels.h: hjbd_adj_kernel)
    dict(name="lqgc_d2_outer_attached", family="solver",                 # the reference's default flags
         problem=dict(kind="LQGC", kwargs=dict(d=2, off_diag=0.1, T=0.5, seed=42, delta_t=0.05)),
         solver=dict(loss_method="log-variance", time_approx="outer", adaptive_forward_process=True,
                     detach_forward=False, early_stopping_time=None, L=4, lr=0.01, seed=42,
                     delta_t=0.05, K=64, u_l2_error_flag=False)),
    dict(name="llgc_d12_outer_relative_entropy", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=12, off_diag=0.05, T=0.2, seed=42)),
         solver=dict(loss_method="relative_entropy", time_approx="outer", adaptive_forward_process=True,
                     detach_forward=False, early_stopping_time=None, L=4, lr=0.003, seed=42, delta_t=0.02, K=80,
                     u_l2_error_flag=False)),
    dict(name="llgc_d12_outer_relative_entropy_detached", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=12, off_diag=0.05, T=0.2, seed=42)),
         solver=dict(loss_method="relative_entropy", time_approx="outer", adaptive_forward_process=True,
                     detach_forward=True, early_stopping_time=None, L=4, lr=0.003, seed=42, delta_t=0.02, K=80,
                     u_l2_error_flag=False)),
    dict(name="dw_d20_densenet_attached_moment", family="solver",
         problem=dict(kind="DoubleWell_multidim", kwargs=dict(d=20, d_1=10, d_2=10, T=0.2, eta=0.3, kappa=1.0)),
         solver=dict(HJB, detach_forward=False, loss_method="moment", L=4, lr=0.002, seed=42, delta_t=0.01, K=96,
                     u_l2_error_flag=False),
         net=dict(kind="densenet", arch=[40, 40], seed=7), probe_times=[0.0, 0.1]),
    dict(name="lqgc_d6_densenet_attached_cross_entropy", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=6, off_diag=0.1, T=0.5, seed=42, delta_t=0.05)),
         solver=dict(HJB, detach_forward=False, loss_method="cross_entropy", L=4, lr=0.005, seed=42, delta_t=0.05, K=128,
                     u_l2_error_flag=False),
         net=dict(kind="densenet", arch=[24, 24], seed=7), probe_times=[0.0]),
    # approx_method='value_function' (solver.py:93-97, 334-339, 438-440): Z = sigma grad_x Y_n, extra loss sum_n (Y_n(X_n) - Y)^2
    dict(name="lqgc_d3_value_function", family="solver",
         problem=dict(kind="LQGC", kwargs=dict(d=3, off_diag=0.1, T=0.5, seed=42, delta_t=0.05)),
         solver=dict(approx_method="value_function", loss_method="log-variance", time_approx="inner",
                     adaptive_forward_process=True, detach_forward=False, early_stopping_time=None, L=4, lr=0.01, seed=42,
                     delta_t=0.05, K=64, u_l2_error_flag=False)),
    # the same ansatz inside the native catalogue (plan_value_native.py): sigma = I, elementwise drift, h = -|z|^2 / 2, state path detached
    dict(name="dw_d10_value_function", family="solver",
         problem=dict(kind="DoubleWell_multidim", kwargs=dict(d=10, d_1=5, d_2=5, T=0.3, eta=0.5, kappa=2.0)),
         solver=dict(approx_method="value_function", loss_method="log-variance", time_approx="inner",
                     adaptive_forward_process=True, detach_forward=True, early_stopping_time=None, L=4, lr=0.005, seed=42,
                     delta_t=0.01, K=96, u_l2_error_flag=False)),
    dict(name="dw_d10_value_function_arch3", family="solver",          # a three-layer value net: the run-time-shaped kernels (genl)
         problem=dict(kind="DoubleWell_multidim", kwargs=dict(d=10, d_1=5, d_2=5, T=0.2, eta=0.5, kappa=2.0)),
         solver=dict(approx_method="value_function", loss_method="log-variance", time_approx="inner",
                     adaptive_forward_process=True, detach_forward=True, early_stopping_time=None, L=4, lr=0.005, seed=42,
                     delta_t=0.01, K=96, u_l2_error_flag=False),
         net=dict(kind="value_densenet", arch=[20, 16, 12], seed=7)),
    dict(name="llgc_d8_diag_value_function_moment", family="solver",
         problem=dict(kind="LLGC", kwargs=dict(d=8, off_diag=0.0, T=0.4, seed=42)),
         solver=dict(approx_method="value_function", loss_method="moment", time_approx="inner",
                     adaptive_forward_process=False, detach_forward=False, early_stopping_time=None, L=4, lr=0.003, seed=42,
                     delta_t=0.02, K=80, u_l2_error_flag=False)),
    dict(name="dw_d20_value_function_randx0", family="solver",
         problem=dict(kind="DoubleWell_multidim", kwargs=dict(d=20, d_1=10, d_2=10, T=0.2, eta=1.0, kappa=1.0)),
         solver=dict(approx_method="value_function", loss_method="log-variance", time_approx="inner",
                     adaptive_forward_process=True, detach_forward=True, early_stopping_time=None, L=3, lr=0.002, seed=7,
                     delta_t=0.01, K=112, u_l2_error_flag=False, random_X_0=True)),
    # the double wells with their finite-difference reference control (problems.py:216-281, 336-476) and the u_L2 log left ON -- what
    # `Double well - 1d - high metastability.ipynb` and `Multidim. double well - mixed metastabilities.ipynb` run
    dict(name="dw1d_logvar_ul2", family="solver",
         problem=dict(kind="DoubleWell", kwargs=dict(d=1, T=0.4, eta=3.0, kappa=5.0),
                      calls=[["compute_reference_solution", dict(nx=400)]]),
         solver=dict(HJB, L=4, lr=0.005, seed=42, delta_t=0.01, K=112)),
    dict(name="dw_d6_mixed_logvar_ul2", family="solver",
         problem=dict(kind="DoubleWell_multidim", kwargs=dict(d=6, d_1=2, d_2=4, T=0.3, eta=0.5, kappa=2.0),
                      calls=[["compute_reference_solution", dict(nx=500)], ["compute_reference_solution_2", dict(nx=500)]]),
         solver=dict(HJB, L=4, lr=0.005, seed=42, delta_t=0.01, K=96),
         net=dict(kind="tanh_mlp", widths=[40, 40], seed=123)),
    # importance-sampling evaluation of the learned control, standalone and in the loop
    dict(name="llgc_d20_is_eval", family="is",
         problem=dict(kind="LLGC", kwargs=dict(d=20, off_diag=0.0, T=0.3, seed=42)),
         solver=dict(HJB, L=3, lr=0.002, seed=42, delta_t=0.01, K=200, u_l2_error_flag=False),
         is_seed=7, is_K=512, is_delta_t=0.01),
    dict(name="lqgc_d4_is_eval", family="is",
         problem=dict(kind="LQGC", kwargs=dict(d=4, off_diag=0.1, T=0.5, seed=42, delta_t=0.05)),
         solver=dict(HJB, L=3, lr=0.01, seed=3, delta_t=0.05, K=96, u_l2_error_flag=False),
         is_seed=11, is_K=300, is_delta_t=0.01),
    dict(name="dw_d10_is_in_loop", family="is",
         problem=dict(kind="DoubleWell_multidim", kwargs=dict(d=10, d_1=5, d_2=5, T=0.5, eta=0.5, kappa=2.0)),
         solver=dict(HJB, L=4, lr=0.005, seed=42, delta_t=0.01, K=256, u_l2_error_flag=False,
                     IS_variance_K=128, IS_variance_iter=2),
         is_seed=5, is_K=256, is_delta_t=0.01),
    # GeneralSolver, diffusion loss, unbounded square
    dict(name="dwgen_d10_diffusion", family="general",
         problem=dict(kind="DoubleWell_multidim_for_general_solver",
                      kwargs=dict(d=10, d_1=5, d_2=5, T=0.3, eta=1, kappa=1, modus="HJB")),
         solver=dict(seed=42, delta_t=0.01, N=12, lr=0.001, L=3, K=96, K_boundary=20,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion"),
         net=dict(arch=[32, 32], seed=42)),
    dict(name="dwgen_d10_bsde", family="general",
         problem=dict(kind="DoubleWell_multidim_for_general_solver",
                      kwargs=dict(d=10, d_1=5, d_2=5, T=0.1, eta=1, kappa=1, modus="HJB")),
         solver=dict(seed=42, delta_t=0.01, N=14, lr=0.001, L=3, K=64, K_boundary=20,
                     alpha=[1.0, 1.0, 1.0], loss_method="BSDE"),
         net=dict(arch=[32, 32], seed=42)),
    dict(name="allencahn_d10_diffusion", family="general",
         problem=dict(kind="AllenCahn", kwargs=dict(d=10, T=0.3, seed=42, modus="pt")),
         solver=dict(seed=42, delta_t=0.01, N=10, lr=0.001, L=3, K=80, K_boundary=16,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion"),
         net=dict(arch=[24, 24], seed=42)),
    # shapes without an exact GeneralSolver instance (zero padding, native_shapes.GenParamPad)
    dict(name="dwgen_d7_h20_diffusion", family="general",                        # (7, 20) -> (10, 24)
         problem=dict(kind="DoubleWell_multidim_for_general_solver",
                      kwargs=dict(d=7, d_1=3, d_2=4, T=0.3, eta=1, kappa=1, modus="HJB")),
         solver=dict(seed=42, delta_t=0.01, N=12, lr=0.001, L=3, K=90, K_boundary=20,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion"),
         net=dict(arch=[20, 20], seed=42)),
    dict(name="allencahn_d20_default_diffusion", family="general",               # default arch [30, 30]: (20, 30) -> (32, 32)
         problem=dict(kind="AllenCahn", kwargs=dict(d=20, T=0.3, seed=42, modus="pt")),
         solver=dict(seed=42, delta_t=0.01, N=10, lr=0.001, L=3, K=80, K_boundary=16,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion")),
    dict(name="dwgen_d40_h50_bsde", family="general",                            # (40, 50) -> (48, 64)
         problem=dict(kind="DoubleWell_multidim_for_general_solver",
                      kwargs=dict(d=40, d_1=20, d_2=20, T=0.1, eta=0.1, kappa=1, modus="HJB")),
         solver=dict(seed=42, delta_t=0.01, N=12, lr=0.001, L=3, K=64, K_boundary=20,
                     alpha=[1.0, 1.0, 1.0], loss_method="BSDE"),
         net=dict(arch=[50, 50], seed=42)),
    # bounded domains: exit tests, Dirichlet / Neumann boundary terms, BSDE with boundary data
    dict(name="expsphere_d4_diffusion_dirichlet", family="general_bounded",
         problem=dict(kind="ExponentialOnSphereNonlinearParabolic", kwargs=dict(d=4, T=0.6, alpha=0.5)),
         solver=dict(seed=42, delta_t=0.01, N=30, lr=0.001, L=3, K=96, K_boundary=20,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion")),
    dict(name="expsphere_d12_h40_bsde_dirichlet", family="general_bounded",
         problem=dict(kind="ExponentialOnSphereNonlinearParabolic", kwargs=dict(d=12, T=0.5, alpha=0.3)),
         solver=dict(seed=42, delta_t=0.005, N=100, lr=0.001, L=3, K=80, K_boundary=16,
                     alpha=[1.0, 1.0, 1.0], loss_method="BSDE"),
         net=dict(arch=[40, 40], seed=42)),
    dict(name="expsphere_d3_diffusion_neumann", family="general_bounded",
         problem=dict(kind="ExponentialOnSphereNonlinearParabolic", kwargs=dict(d=3, T=0.5, alpha=0.7),
                      attrs=dict(boundary_type="Neumann")),
         solver=dict(seed=42, delta_t=0.01, N=25, lr=0.001, L=3, K=64, K_boundary=18,
                     alpha=[1.0, 0.5, 2.0], loss_method="diffusion")),
    dict(name="box_d5_diffusion", family="general_bounded", numpy_seed=3,
         problem=dict(kind="QuadraticOnBox", kwargs=dict(d=5, T=0.4, X_l=-1.0, X_r=1.0, scale=1.2)),
         solver=dict(seed=42, delta_t=0.01, N=30, lr=0.001, L=3, K=96, K_boundary=20,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion", adaptive_forward_process=True)),
    dict(name="box_d3_upper_bsde", family="general_bounded", numpy_seed=4,
         problem=dict(kind="QuadraticOnBox", kwargs=dict(d=3, T=0.3, X_l=-1.0, X_r=0.5, one_boundary=True, quad_h=False)),
         solver=dict(seed=42, delta_t=0.01, N=40, lr=0.001, L=3, K=64, K_boundary=20,
                     alpha=[1.0, 1.0, 1.0], loss_method="BSDE")),
    # EllipticSolver (solver.py:560-826): same step without the time input
    dict(name="expball_sin_d5_elliptic_diffusion", family="elliptic",
         problem=dict(kind="ExponentialOnBallNonlinearSin", kwargs=dict(d=5, alpha=0.5)),
         solver=dict(seed=42, delta_t=0.01, N=20, lr=0.001, L=3, K=96, K_boundary=20, loss_method="diffusion")),
    dict(name="expball_sq_d3_elliptic_bsde", family="elliptic",
         problem=dict(kind="ExponentialOnBallNonlinear", kwargs=dict(d=3, alpha=0.5)),
         solver=dict(seed=42, delta_t=0.02, N=120, lr=0.001, L=3, K=64, K_boundary=20, loss_method="BSDE"),
         net=dict(arch=[24, 24], seed=42)),
    dict(name="expsphere_lin_d10_elliptic_diffusion", family="elliptic",
         problem=dict(kind="ExponentialOnSphere", kwargs=dict(d=10, alpha=0.3)),
         solver=dict(seed=42, delta_t=0.005, N=16, lr=0.001, L=3, K=80, K_boundary=20, loss_method="diffusion",
                     adaptive_forward_process=True)),
    dict(name="expball_sin_d4_elliptic_neumann", family="elliptic",
         problem=dict(kind="ExponentialOnBallNonlinearSin", kwargs=dict(d=4, alpha=0.5, boundary_type="Neumann")),
         solver=dict(seed=42, delta_t=0.01, N=15, lr=0.001, L=3, K=64, K_boundary=20, loss_method="diffusion",
                     boundary_type="Neumann", alpha=[1.0, 0.5])),
    dict(name="box_d4_elliptic_diffusion", family="elliptic",
         problem=dict(kind="QuadraticOnBox", kwargs=dict(d=4, X_l=-1.0, X_r=1.0, parabolic=False)),
         solver=dict(seed=42, delta_t=0.01, N=20, lr=0.001, L=3, K=96, K_boundary=20, loss_method="diffusion")),
    dict(name="box_d2_upper_elliptic_diffusion", family="elliptic",
         problem=dict(kind="QuadraticOnBox", kwargs=dict(d=2, X_l=-1.0, X_r=0.6, one_boundary=True, parabolic=False, quad_h=False)),
         solver=dict(seed=42, delta_t=0.01, N=20, lr=0.001, L=3, K=64, K_boundary=20, loss_method="diffusion")),
    # round 3: the remaining domains / boundary terms of the diffusion-loss solvers (solver.py:647-708, 750-760, 1023-1027,
    # 1048-1052, 1122-1123, 1177-1183): composite plan, never an error
    dict(name="committor_d3_elliptic_diffusion", family="elliptic",
         problem=dict(kind="Committor", kwargs=dict(d=3)),
         solver=dict(seed=42, delta_t=0.01, N=20, lr=0.001, L=3, K=90, K_boundary=20, loss_method="diffusion")),
    dict(name="committor_d4_elliptic_bsde", family="elliptic",
         problem=dict(kind="Committor", kwargs=dict(d=4)),
         solver=dict(seed=42, delta_t=0.02, N=150, lr=0.001, L=3, K=64, K_boundary=20, loss_method="BSDE"),
         net=dict(arch=[24, 24], seed=42)),
    dict(name="committor_d3_elliptic_testlog", family="elliptic",                # K_test_log + loss_with_stopped (the committor notebook's flags)
         problem=dict(kind="Committor", kwargs=dict(d=3)),
         solver=dict(seed=42, delta_t=0.01, N=20, lr=0.001, L=3, K=90, K_boundary=20, loss_method="diffusion",
                     K_test_log=200, loss_with_stopped=True, alpha=[10.0, 1.0])),
    dict(name="expsphere_d3_two_spheres_diffusion", family="general_bounded",
         problem=dict(kind="ExponentialOnSphereNonlinearParabolic", kwargs=dict(d=3, T=0.5, alpha=0.5),
                      attrs=dict(boundary="two_spheres", boundary_distance_1=0.4, boundary_distance_2=1.0)),
         solver=dict(seed=42, delta_t=0.01, N=25, lr=0.001, L=3, K=96, K_boundary=20,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion")),
    dict(name="corner_d3_elliptic_diffusion", family="elliptic",
         problem=dict(kind="QuadraticOnBox", kwargs=dict(d=3, X_l=-1.0, X_r=1.0, parabolic=False),
                      attrs=dict(boundary="square-corner", X_corner=0.2)),
         solver=dict(seed=42, delta_t=0.01, N=20, lr=0.001, L=3, K=96, K_boundary=20, loss_method="diffusion")),
    dict(name="expsphere_d3_bsde_neumann", family="general_bounded",
         problem=dict(kind="ExponentialOnSphereNonlinearParabolic", kwargs=dict(d=3, T=0.3, alpha=0.5),
                      attrs=dict(boundary_type="Neumann")),
         solver=dict(seed=42, delta_t=0.01, N=40, lr=0.001, L=3, K=64, K_boundary=18,
                     alpha=[1.0, 1.0, 1.0], loss_method="BSDE")),
    # round 3: value nets of other depths / widths -- the nets the diffusion-loss notebooks swap into model.V (function_space.py:116-140)
    dict(name="allencahn_d10_arch3_diffusion", family="general",
         problem=dict(kind="AllenCahn", kwargs=dict(d=10, T=0.3, seed=42, modus="pt")),
         solver=dict(seed=42, delta_t=0.01, N=10, lr=0.001, L=3, K=80, K_boundary=16,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion"),
         net=dict(arch=[24, 24, 12], seed=42)),
    dict(name="dwgen_d10_arch4_bsde", family="general",
         problem=dict(kind="DoubleWell_multidim_for_general_solver",
                      kwargs=dict(d=10, d_1=5, d_2=5, T=0.1, eta=0.1, kappa=1, modus="HJB")),
         solver=dict(seed=42, delta_t=0.01, N=12, lr=0.001, L=3, K=64, K_boundary=20,
                     alpha=[1.0, 1.0, 1.0], loss_method="BSDE"),
         net=dict(arch=[30, 30, 30, 30], seed=42)),
    dict(name="heat_d6_arch1_diffusion", family="general",
         problem=dict(kind="HeatEquation", kwargs=dict(d=6, T=0.5, seed=42)),
         solver=dict(seed=42, delta_t=0.01, N=10, lr=0.001, L=3, K=72, K_boundary=16,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion"),
         net=dict(arch=[40], seed=42)),
    dict(name="expsphere_d4_arch3_diffusion_dirichlet", family="general_bounded",
         problem=dict(kind="ExponentialOnSphereNonlinearParabolic", kwargs=dict(d=4, T=0.6, alpha=0.5)),
         solver=dict(seed=42, delta_t=0.01, N=30, lr=0.001, L=3, K=96, K_boundary=20,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion"),
         net=dict(arch=[20, 16, 12], seed=42)),
    dict(name="expball_sin_d5_arch3_elliptic_diffusion", family="elliptic",
         problem=dict(kind="ExponentialOnBallNonlinearSin", kwargs=dict(d=5, alpha=0.5)),
         solver=dict(seed=42, delta_t=0.01, N=20, lr=0.001, L=3, K=96, K_boundary=20, loss_method="diffusion"),
         net=dict(arch=[24, 24, 12], seed=42)),
    # ... and the one configuration the reference publishes a timing for (Allen-Cahn.ipynb:46-72, 86: 0.31 s per iteration)
    dict(name="allencahn_d100_notebook_a110", family="general",
         problem=dict(kind="AllenCahn", kwargs=dict(d=100, T=0.3, seed=42, modus="pt"), attrs=dict(boundary_distance=7.0)),
         solver=dict(seed=42, delta_t=0.001, N=25, lr=0.001, L=3, K=200, K_boundary=50,
                     alpha=[10.0, 1.0, 1.0], loss_method="diffusion", uniform_square=True),
         net=dict(arch=[110, 110, 50], seed=42)),
    # round 3: the d = 100 instance BASELINE configs[2] names (gen_*<100,64>: fp32, split-product and bf16 modes under pytest)
    dict(name="dwgen_d100_h64_diffusion", family="general",
         problem=dict(kind="DoubleWell_multidim_for_general_solver",
                      kwargs=dict(d=100, d_1=50, d_2=50, T=0.3, eta=1, kappa=1, modus="HJB")),
         solver=dict(seed=42, delta_t=0.01, N=5, lr=0.001, L=3, K=64, K_boundary=16,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion"),
         net=dict(arch=[64, 64], seed=42)),
    dict(name="heat_d6_diffusion", family="general",
         problem=dict(kind="HeatEquation", kwargs=dict(d=6, T=0.5, seed=42)),
         solver=dict(seed=42, delta_t=0.01, N=10, lr=0.001, L=3, K=72, K_boundary=16,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion")),
    # round 4: the committor notebook (`Committor function.ipynb`): EllipticSolver on 'two_spheres' with the tanh(.)**2 net the
    # notebook defines for itself, arch = [d + 10, d, d, d]; its BSDE run (cell 15: N = 5000, every trajectory runs until it leaves
    # the annulus) is the reference's second published timing (14-28 s per iteration at d = 10, K = 200)
    dict(name="committor_d4_tanh2_elliptic_bsde", family="elliptic",
         problem=dict(kind="Committor", kwargs=dict(d=4)),
         solver=dict(seed=42, delta_t=0.004, N=1500, lr=0.001, L=3, K=48, K_boundary=20, loss_method="BSDE",
                     alpha=[0.01, 1.0]),
         net=dict(kind="user_tanh2", arch=[14, 4, 4, 4], seed=42)),
    dict(name="committor_d10_tanh2_notebook_diffusion", family="elliptic",     # cell 3: K = 200, N = 50, dt = 1e-3, alpha = [10, 1]
         problem=dict(kind="Committor", kwargs=dict(d=10)),
         solver=dict(seed=42, delta_t=0.001, N=50, lr=0.001, L=3, K=200, K_boundary=50, loss_method="diffusion",
                     alpha=[10.0, 1.0], K_test_log=300),
         net=dict(kind="user_tanh2", arch=[20, 10, 10, 10], seed=42)),
    # DenseNet_tanh (function_space.py:143-158: nn.Linear layers, tanh) as the value net
    dict(name="allencahn_d10_densenet_tanh_diffusion", family="general",
         problem=dict(kind="AllenCahn", kwargs=dict(d=10, T=0.3, seed=42, modus="pt")),
         solver=dict(seed=42, delta_t=0.01, N=10, lr=0.001, L=3, K=80, K_boundary=16,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion"),
         net=dict(kind="densenet_tanh", arch=[24, 24, 12], seed=42)),
    # BASELINE configs[2] says "diffusion-loss (BSDE)": the BSDE loss on the exact (100, 64) instance
    dict(name="dwgen_d100_h64_bsde", family="general",
         problem=dict(kind="DoubleWell_multidim_for_general_solver",
                      kwargs=dict(d=100, d_1=50, d_2=50, T=0.05, eta=1, kappa=1, modus="HJB")),
         solver=dict(seed=42, delta_t=0.01, N=5, lr=0.001, L=3, K=64, K_boundary=16,
                     alpha=[1.0, 1.0, 1.0], loss_method="BSDE"),
         net=dict(arch=[64, 64], seed=42)),
    # loss_with_stopped on a parabolic sphere problem (solver.py:1185-1186), sample_center on a one-dimensional elliptic one (:643-645)
    dict(name="expsphere_d4_stopped_diffusion", family="general_bounded",
         problem=dict(kind="ExponentialOnSphereNonlinearParabolic", kwargs=dict(d=4, T=0.6, alpha=0.5)),
         solver=dict(seed=42, delta_t=0.01, N=30, lr=0.001, L=3, K=96, K_boundary=20,
                     alpha=[1.0, 1.0, 1.0], loss_method="diffusion", loss_with_stopped=True)),
    dict(name="expsphere_lin_d1_elliptic_center", family="elliptic",
         problem=dict(kind="ExponentialOnSphere", kwargs=dict(d=1, alpha=0.3)),
         solver=dict(seed=42, delta_t=0.01, N=20, lr=0.001, L=3, K=64, K_boundary=20, loss_method="diffusion",
                     sample_center=True)),
]


def main():
    torch.set_num_threads(1)
    index = {"torch": torch.__version__, "numpy": np.__version__, "cases": []}
    torch.manual_seed(42)
    index["noise_fingerprint"] = {"seed": 42, "shape": [4, 3, 2],
                                  "values": f32list(torch.randn(4, 3, 2))}
    only = set(sys.argv[1:])
    for case in CASES:
        if only and case["name"] not in only:
            continue
        print("running", case["name"], flush=True)
        res = {"solver": run_solver_case, "general": run_general_case, "is": run_is_case,
               "general_bounded": run_general_bounded_case, "elliptic": run_elliptic_case}[case["family"]](case)
        rec = {"case": case, "expected": res, "torch": torch.__version__}
        with open(os.path.join(OUT, case["name"] + ".json"), "w") as fh:
            json.dump(rec, fh, indent=1)
        index["cases"].append(case["name"])
        print("   loss_log", res["loss_log"])
    if only:                                  # a partial run: keep the index, add what is new
        with open(os.path.join(OUT, "index.json")) as fh:
            old = json.load(fh)
        old["cases"] += [c for c in index["cases"] if c not in old["cases"]]
        index = old
    with open(os.path.join(OUT, "index.json"), "w") as fh:
        json.dump(index, fh, indent=1)


if __name__ == "__main__":
    main()
