"""CPU oracle for the path-space hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain torch-CPU restatement of the reference algorithm
(lorenzrichter/path-space-PDE-solver: solver.py / problems.py / function_space.py),
written as flat functions, each citing the reference lines it follows.  It is used
ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
checker; nothing under path-space-pde-solver_amd/ imports it.

Pinning: tests/test_oracle_golden.py checks this file bit-for-bit (same torch build,
CPU) against tests/golden/*.json, which were produced by importing and running the
reference itself (tests/golden/make_golden.py).  Arithmetic is torch ATen fp32, the
same library the reference calls; the reference pins no torch version, so parity is
pinned against torch 2.10.0 as installed in the image.

Everything is fp32 on CPU; noise comes from torch's global CPU generator in exactly
the reference's draw order unless a ``noise`` tensor is supplied.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

CPU = torch.device("cpu")


# --------------------------------------------------------------------------------------
# problems  (reference problems.py)
# --------------------------------------------------------------------------------------
@dataclass
class OracleProblem:
    kind: str
    d: int
    T: float
    X_0: torch.Tensor
    B: torch.Tensor
    b: Callable
    sigma: Callable
    h: Callable
    f: Callable
    g: Optional[Callable] = None
    extra: Dict = field(default_factory=dict)


def problem_llgc(d=1, off_diag=0.0, T=5, seed=42) -> OracleProblem:
    """Ornstein-Uhlenbeck, linear terminal cost.  problems.py:14-49."""
    torch.manual_seed(seed)                                          # :20
    A = -torch.eye(d) + off_diag * torch.randn(d, d)                 # :24
    B = torch.eye(d) + off_diag * torch.randn(d, d)                  # :25
    alpha = torch.ones(d, 1)                                         # :26
    return OracleProblem(
        kind="LLGC", d=d, T=T, X_0=torch.zeros(d), B=B,
        b=lambda x: torch.mm(A, x.t()).t(),                          # :37
        sigma=lambda x: B,                                           # :40
        h=lambda t, x, y, z: -0.5 * torch.sum(z ** 2, dim=1),        # :46
        f=lambda x, t: torch.zeros(x.shape[0]),                      # :43
        g=lambda x: torch.mm(x, alpha)[:, 0],                        # :49
        extra=dict(A=A, alpha=alpha))


def problem_lqgc(d=1, off_diag=0.0, T=5, seed=42, delta_t=0.05) -> OracleProblem:
    """Linear-quadratic Gaussian control.  problems.py:118-167."""
    torch.manual_seed(seed)                                          # :124
    A = -torch.eye(d) + off_diag * torch.randn(d, d)                 # :128
    B = torch.eye(d) + off_diag * torch.randn(d, d)                  # :129
    P = 0.5 * torch.eye(d)                                           # :137
    R = torch.eye(d)                                                 # :139
    f = lambda x, t: torch.sum(x.t() * torch.mm(P, x.t()), 0)        # :161
    return OracleProblem(
        kind="LQGC", d=d, T=T, X_0=torch.zeros(d), B=B,
        b=lambda x: torch.mm(A, x.t()).t(),                          # :155
        sigma=lambda x: B,                                           # :158
        h=lambda t, x, y, z: -0.5 * torch.sum(z ** 2, dim=1) - f(x, t),   # :167
        f=f,
        g=lambda x: torch.sum(x.t() * torch.mm(R, x.t()), 0),        # :164
        extra=dict(A=A, P=P, R=R))


def _double_well_parts(d, d_1, d_2, eta, kappa):
    eta_ = torch.tensor([eta] * d_1 + [1.0] * d_2)                   # problems.py:296 / :490
    kappa_ = torch.tensor([kappa] * d_1 + [1.0] * d_2)               # :298 / :492
    ones = torch.ones(d)
    grad_V = lambda x: 4.0 * kappa_ * (x * (x ** 2 - ones))          # :312 / :507
    cost = lambda x: (torch.sum(eta_ * (x - ones) ** 2, 1)).squeeze()   # :334 / :534
    return eta_, kappa_, grad_V, cost


def problem_double_well(d=1, d_1=1, d_2=0, T=1, eta=1, kappa=1) -> OracleProblem:
    """problems.py:285-334 (DoubleWell_multidim)."""
    eta_, kappa_, grad_V, cost = _double_well_parts(d, d_1, d_2, eta, kappa)
    B = torch.eye(d)                                                 # :299
    return OracleProblem(
        kind="DoubleWell_multidim", d=d, T=T, X_0=-torch.ones(d), B=B,      # :300
        b=lambda x: -grad_V(x),                                      # :315
        sigma=lambda x: B,                                           # :318
        h=lambda t, x, y, z: -0.5 * torch.sum(z ** 2, dim=1),        # :321
        f=lambda x, t: torch.zeros(x.shape[0]),                      # :324
        g=cost, extra=dict(eta_=eta_, kappa_=kappa_))


def problem_double_well_general(d=1, d_1=1, d_2=0, T=1, eta=1, kappa=1, modus="HJB") -> OracleProblem:
    """problems.py:479-534 (DoubleWell_multidim_for_general_solver); f is the terminal value."""
    eta_, kappa_, grad_V, cost = _double_well_parts(d, d_1, d_2, eta, kappa)
    B = torch.eye(d)                                                 # :493
    if modus == "linear":
        h = lambda t, x, y, z: torch.zeros(x.shape[0])               # :517-518
        f = lambda x: torch.exp(-cost(x))                            # :531-532
    else:
        h = lambda t, x, y, z: -0.5 * torch.sum(z ** 2, dim=1)       # :519
        f = cost                                                     # :534
    return OracleProblem(
        kind="DoubleWell_multidim_for_general_solver", d=d, T=T, X_0=-torch.ones(d), B=B,
        b=lambda x: -grad_V(x), sigma=lambda x: B, h=h, f=f,
        extra=dict(boundary="unbounded_square", X_l=-2.5, X_r=2.5, eta_=eta_, kappa_=kappa_))   # :496-498


def problem_allen_cahn(d=1, T=0.3, seed=42, modus="pt") -> OracleProblem:
    """problems.py:1175-1209 (AllenCahn, modus='pt')."""
    np.random.seed(seed)                                             # :1178
    B = torch.tensor(np.eye(d) * np.sqrt(2)).float()                 # :1183-1184
    return OracleProblem(
        kind="AllenCahn", d=d, T=T, X_0=torch.zeros(d), B=B,
        b=lambda x: torch.zeros(x.shape),                            # :1194
        sigma=lambda x: B,                                           # :1200
        h=lambda t, x, y, z: y - y ** 3,                             # :1204
        f=lambda x: 1 / (2 + 2 / 5 * torch.sum(x ** 2, 1)),          # :1208
        extra=dict(boundary="unbounded", boundary_distance=2.0))     # :1190-1191


def problem_heat(d=1, T=1, seed=42) -> OracleProblem:
    """problems.py:1733-1758 (HeatEquation)."""
    torch.manual_seed(seed)                                          # :1736
    B = torch.sqrt(torch.tensor(2.0)) * torch.eye(d)                 # :1740
    return OracleProblem(
        kind="HeatEquation", d=d, T=T, X_0=torch.zeros(d), B=B,
        b=lambda x: torch.zeros(x.shape),                            # :1746
        sigma=lambda x: B,                                           # :1749
        h=lambda t, x, y, z: torch.zeros(x.shape[0]),                # :1755
        f=lambda x: torch.sum(x ** 2, 1),                            # :1758
        extra=dict(boundary="unbounded", boundary_distance=1.0))     # :1741-1743


PROBLEMS = {
    "LLGC": problem_llgc, "LQGC": problem_lqgc, "DoubleWell_multidim": problem_double_well,
    "DoubleWell_multidim_for_general_solver": problem_double_well_general,
    "AllenCahn": problem_allen_cahn, "HeatEquation": problem_heat,
}


def make_problem(kind: str, **kwargs) -> OracleProblem:
    return PROBLEMS[kind](**kwargs)


# --------------------------------------------------------------------------------------
# ansatz spaces (reference function_space.py)
# --------------------------------------------------------------------------------------
class TanhMLP(torch.nn.Module):
    """function_space.py:177-195 (MySequential) with configurable hidden widths.

    Seeding recipe :180-188: manual_seed(seed); one nn.Linear per layer (its default
    init consumes generator state); then normal_(0, 0.01) on weight and bias, layer by
    layer.  widths=[30,30] is the reference's hard-coded :181.
    """

    def __init__(self, d_in, d_out, lr, seed, widths=(30, 30)):
        super().__init__()
        torch.manual_seed(seed)
        dims = [d_in] + list(widths) + [d_out]
        self.dims = dims
        self.linears = torch.nn.ModuleList(
            [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.optim = torch.optim.Adam(self.parameters(), lr=lr)      # :185 (before the re-init)
        for lin in self.linears:
            torch.nn.init.normal_(lin.weight, 0, 0.01)
            torch.nn.init.normal_(lin.bias, 0, 0.01)

    def forward(self, x):                                            # :190-195
        last = len(self.linears) - 1
        for i, lin in enumerate(self.linears):
            x = lin(x)
            if i < last:
                x = torch.tanh(x)
        return x


class DenseNetOracle(torch.nn.Module):
    """function_space.py:116-140: dense-concat net, relu(.)**2, weights randn*0.1, zero bias."""

    def __init__(self, d_in, d_out, lr, arch=(30, 30), seed=42):
        super().__init__()
        torch.manual_seed(seed)                                      # :119
        dims = [d_in] + list(arch) + [d_out]
        self.dims = dims
        ws = []
        for i in range(len(dims) - 1):                               # :121-125
            ws.append(torch.nn.Parameter(torch.randn(sum(dims[:i + 1]), dims[i + 1]) * 0.1))
            ws.append(torch.nn.Parameter(torch.zeros(dims[i + 1])))
        self.W = ws
        for i, w in enumerate(ws):
            self.register_parameter("param %d" % i, w)               # :128-129
        self.optim = torch.optim.Adam(self.parameters(), lr=lr)      # :131

    def forward(self, x):                                            # :133-140
        nl = len(self.dims) - 1
        for i in range(nl):
            lin = torch.matmul(x, self.W[2 * i]) + self.W[2 * i + 1]
            if i == nl - 1:
                x = lin
            else:
                x = torch.cat([x, torch.nn.functional.relu(lin) ** 2], dim=1)
        return x


class ScalarY0(torch.nn.Module):
    """function_space.py:6-21 (SingleParam, initial=None)."""

    def __init__(self, lr, seed=42):
        super().__init__()
        torch.manual_seed(seed)                                      # :9
        self.Y_0 = torch.nn.Parameter(torch.tensor([0.0]))           # :11
        self.register_parameter("param", self.Y_0)                   # :17
        self.optim = torch.optim.Adam(self.parameters(), lr=lr)      # :18

    def forward(self, x):
        return self.Y_0


# --------------------------------------------------------------------------------------
# Solver.train restatement (reference solver.py:420-557), control ansatz
# --------------------------------------------------------------------------------------
@dataclass
class HJBConfig:
    K: int
    delta_t: float
    lr: float = 0.001
    L: int = 1
    seed: int = 42
    loss_method: str = "log-variance"
    time_approx: str = "inner"
    learn_Y_0: bool = False
    adaptive_forward_process: bool = True
    detach_forward: bool = False
    random_X_0: bool = False
    IS_variance_K: int = 0
    IS_variance_iter: int = 1


def hjb_build(problem: OracleProblem, cfg: HJBConfig, net: Optional[torch.nn.Module] = None):
    """Constructor side effects of Solver.__init__ that matter for parity (solver.py:84-99)."""
    torch.manual_seed(cfg.seed)                                      # :84
    y0 = ScalarY0(lr=cfg.lr)                                         # :86 (re-seeds with 42)
    N = int(np.floor(problem.T / cfg.delta_t))                       # :41 (float64)
    if cfg.time_approx == "inner":
        z = TanhMLP(problem.d + 1, problem.d, cfg.lr, seed=123)      # :91
    else:
        z = [DenseNetOracle(problem.d, problem.d, cfg.lr, seed=cfg.seed) for _ in range(N)]   # :88
    if net is not None:
        z = net
    return z, y0, N


def control_eval(z, X, n, dt32, N, time_approx="inner"):
    """solver.py:349-356 (Z_n_)."""
    if time_approx == "outer":
        return z[max(0, min(n, N - 1))](X)                           # :352-353
    t_X = torch.cat([torch.ones([X.shape[0], 1]) * n * dt32, X], 1)  # :355
    return z(t_X)                                                    # :356


def hjb_loss(kind, D, Y, gX, Z_sum=None, adaptive=True):
    """solver.py:164-192 for the losses in scope."""
    if kind == "moment":
        return D.pow(2).mean()                                       # :166
    if kind == "log-variance":
        return D.pow(2).mean() - D.mean().pow(2)                     # :168
    if kind == "variance":
        return torch.var(torch.exp(-gX + Y))                         # :172
    if kind == "cross_entropy":
        if adaptive:
            return (Y * torch.exp(-gX + Y.detach())).mean()          # :185
        return (Y * torch.exp(-gX)).mean()                           # :186
    if kind == "relative_entropy":
        return (Z_sum + gX).mean()                                   # :179-180
    raise NotImplementedError(kind)


def hjb_train(problem: OracleProblem, cfg: HJBConfig, net=None, noise: Optional[List[torch.Tensor]] = None,
              x0_noise: Optional[List[torch.Tensor]] = None, trace: bool = False, step_models=None):
    """Restates Solver.train (solver.py:420-557) for approx_method='control'.

    noise: optional list (one per iteration) of xi tensors shaped (K, d, N+1) replacing the
    CPU-generator draw of :381 (used to feed device-generated Philox noise to the oracle).
    Returns dict(loss_log, Y_0_log, z, y0, traces).
    """
    z, y0, N = step_models if step_models is not None else hjb_build(problem, cfg, net)
    phis = (list(z) if isinstance(z, list) else [z]) + ([y0] if cfg.learn_Y_0 else [])   # :142-153
    for p_ in phis:
        p_.train()
    dt32 = torch.tensor(cfg.delta_t)                                 # :39
    sq_dt32 = torch.sqrt(dt32)                                       # :40
    K, d = cfg.K, problem.d
    out = dict(loss_log=[], Y_0_log=[], traces=[], N=N, IS_rel_log=[])
    torch.manual_seed(cfg.seed)                                      # :422
    for l in range(cfg.L):
        X = problem.X_0.repeat(K, 1)                                 # :365
        if cfg.random_X_0:
            X = torch.randn(K, d) if x0_noise is None else x0_noise[l]   # :367
        Y = torch.tensor([0.0]).repeat(K)                            # :368
        if cfg.learn_Y_0:
            Y = y0(X)                                                # :373 (shape (1,), broadcasts)
            out["Y_0_log"].append(Y[0].item())                       # :374
        xi = torch.randn(K, d, N + 1) if noise is None else noise[l]     # :381
        tr = dict(X=[X.clone()], Y=[]) if trace else None
        Z_sum = torch.zeros(K)                                       # :376
        for n in range(N):
            Z = control_eval(z, X, n, dt32, N, cfg.time_approx)      # :449
            c = torch.zeros(d, K)                                    # :451
            if cfg.adaptive_forward_process:
                c = -control_eval(z, X, n, dt32, N, cfg.time_approx).t()     # :456 (2nd forward)
            if cfg.detach_forward:
                c = c.detach()                                       # :468-469
            sig = problem.sigma(X)
            X = (X + (problem.b(X) + torch.mm(sig, c).t()) * dt32
                 + torch.mm(sig, xi[:, :, n + 1].t()).t() * sq_dt32)     # :471-472
            Y = (Y + (-problem.h(dt32 * n, X, Y, Z) + torch.sum(Z * c.t(), 1)) * dt32
                 + torch.sum(Z * xi[:, :, n + 1], 1) * sq_dt32)      # :477-478 (h sees X_{n+1})
            if "relative_entropy" in cfg.loss_method:                # :484-486 (f at the UPDATED state, time n dt)
                Z_sum = Z_sum + (0.5 * torch.sum(Z ** 2, dim=1) + problem.f(X, n * dt32)) * dt32
            if trace:
                tr["X"].append(X.detach().clone())
                tr["Y"].append(Y.detach().clone())
        for p_ in phis:
            p_.optim.zero_grad()                                     # :194-196
        gX = problem.g(X)
        D = Y - gX
        loss = hjb_loss(cfg.loss_method, D, Y, gX, Z_sum=Z_sum, adaptive=cfg.adaptive_forward_process) + torch.zeros(K).mean()   # :220, :434, :499
        loss.backward()                                              # :221
        if trace:
            tr["D"] = D.detach().clone()
            tr["Zsum_g"] = (Z_sum + gX).detach().clone()
            tr["grads"] = [p.grad.detach().clone() for p in (z.parameters() if not isinstance(z, list) else z[0].parameters())]
            out["traces"].append(tr)
        for p_ in phis:
            p_.optim.step()                                          # :198-200
        out["loss_log"].append(loss.item())                          # :514
        if cfg.IS_variance_K > 0 and l % cfg.IS_variance_iter == 0:  # :521-528 (draws from the same generator)
            out["IS_rel_log"].append(is_eval(problem, z, cfg.delta_t, N, cfg.IS_variance_K,
                                             time_approx=cfg.time_approx)[2])
    out["z"], out["y0"] = z, y0
    return out


def is_eval(problem: OracleProblem, z, model_delta_t, N_model, K, delta_t=0.01, time_approx="inner"):
    """Restates utilities.do_importance_sampling_me (utilities.py:287-359), control='approx',
    simulate_naive=False: forward-only controlled rollout with its own step delta_t, Girsanov weight
    exp(-int u.dW - 0.5 int |u|^2 dt), statistics of exp(-int f - g(X_T)) * weight."""
    sq_dt = np.sqrt(delta_t)                                          # :298 (float64 numpy scalars)
    N = int(np.ceil(problem.T / delta_t))                            # :299
    dt32 = torch.tensor(model_delta_t)
    X_u = problem.X_0.repeat(K, 1)                                   # :303
    ito = torch.zeros(K)
    riemann = torch.zeros(K)
    f_int_u = torch.zeros(K)
    for n in range(N):
        xi = torch.randn(K, problem.d)                               # :310
        t = n * delta_t
        n_idx = int(torch.ceil(t / dt32))                            # solver.py:361 (Z_n)
        with torch.no_grad():
            ut = -control_eval(z, X_u, n_idx, dt32, N_model, time_approx)     # :321
        sig = problem.sigma(X_u)
        X_u = (X_u + (problem.b(X_u) + torch.mm(sig, ut.t()).t()) * delta_t
               + torch.mm(sig, xi.t()).t() * sq_dt)                  # :324-325
        ito = ito + torch.sum(ut * xi, 1) * sq_dt                    # :326
        riemann = riemann + torch.sum(ut ** 2, 1) * delta_t          # :327
        f_int_u = f_int_u + problem.f(X_u, n * delta_t) * delta_t    # :328
    girsanov = torch.exp(-ito - 0.5 * riemann)                       # :330
    w = torch.exp(-f_int_u - problem.g(X_u)) * girsanov
    mean_IS = torch.mean(w).item()                                   # :336
    var_IS = torch.var(w).item()                                     # :337
    return mean_IS, var_IS, float(np.sqrt(var_IS) / mean_IS)         # :338


def control_on_grid(z, X, t, delta_t, N, time_approx="inner"):
    """solver.py:360-362: n = ceil(t/dt) in fp32 then Z_n_; the control is u = -Z."""
    dt32 = torch.tensor(delta_t)
    n = int(torch.ceil(torch.tensor(t) / dt32))
    with torch.no_grad():
        return -control_eval(z, X, n, dt32, N, time_approx)


# --------------------------------------------------------------------------------------
# GeneralSolver.train restatement (reference solver.py:1001-1206), unbounded domains
# --------------------------------------------------------------------------------------
@dataclass
class GeneralConfig:
    K: int
    N: int
    delta_t: float
    lr: float = 0.001
    L: int = 1
    seed: int = 42
    K_boundary: int = 50
    alpha: tuple = (1.0, 1.0, 1.0)
    loss_method: str = "diffusion"
    adaptive_forward_process: bool = False
    detach_forward: bool = True
    uniform_square: bool = False


def general_build(problem: OracleProblem, cfg: GeneralConfig, arch=None):
    torch.manual_seed(cfg.seed)                                      # :978
    kw = {} if arch is None else dict(arch=arch)
    return DenseNetOracle(problem.d + 1, 1, cfg.lr, seed=cfg.seed, **kw)    # :980


def general_train(problem: OracleProblem, cfg: GeneralConfig, V=None, trace=False):
    """Restates GeneralSolver.train for loss_method in {'diffusion','BSDE'} on
    boundary in {'unbounded','unbounded_square'} (solver.py:1001-1206)."""
    if V is None:
        V = general_build(problem, cfg)
    dt32 = torch.tensor(cfg.delta_t)                                 # :950
    sq_dt32 = torch.sqrt(dt32)                                       # :951
    K, d, T = cfg.K, problem.d, problem.T
    bnd = problem.extra["boundary"]
    assert "unbounded" in bnd
    out = dict(loss_log=[], K_log=[], traces=[])
    torch.manual_seed(cfg.seed)                                      # :1003
    for l in range(cfg.L):
        loss = 0
        if bnd == "unbounded":
            if cfg.uniform_square:
                X = torch.rand(K, d) * 2 - 1                         # :1042
                X = problem.extra["boundary_distance"] * X / torch.sqrt(torch.sum(X ** 2, 1)).unsqueeze(1) \
                    * (torch.rand(K).unsqueeze(1))                   # :1043
            else:
                X = torch.randn(K, d)                                # :1045
                X = problem.extra["boundary_distance"] * X / torch.sqrt(torch.sum(X ** 2, 1)).unsqueeze(1) \
                    * (torch.rand(K).unsqueeze(1) ** (1 / d))        # :1046
        else:
            X_l, X_r = problem.extra["X_l"], problem.extra["X_r"]
            X = (X_r - X_l) * torch.rand(K, d) + X_l                 # :1056
        if cfg.loss_method != "BSDE":                                # :1062 (boundary_loss=True)
            Kb = cfg.K_boundary
            X_T = torch.cat([X[:Kb, :], T * torch.ones(Kb).unsqueeze(1)], 1)    # :1063
            loss = loss + cfg.alpha[1] * torch.mean((V(X_T).squeeze() - problem.f(X[:Kb, :])) ** 2)   # :1064
        X = X.clone().requires_grad_(True)                           # :1076
        t_n = torch.rand(K, 1) * T                                   # :1078
        X_t_n = torch.cat([X, t_n], 1)                               # :1079
        Y = V(X_t_n).squeeze()                                       # :1081
        stopped = torch.zeros(K).bool()                              # :1084
        K_count = 0
        tr = dict(X0=X.detach().clone(), t0=t_n.clone(), xi=[]) if trace else None
        for n in range(cfg.N):
            if torch.sum(~stopped) == 0:                             # :1093-1097
                break
            Y_ = V(X_t_n)                                            # :1100
            Y_eval = Y_.squeeze().sum()
            grad_V, = torch.autograd.grad(Y_eval, X, create_graph=True)      # :1103 (the extra .backward of :1102 only fills .grad, cleared at :1164)
            Z = torch.mm(problem.sigma(X).t(), grad_V.t()).t()       # :1104
            xi = torch.randn(K, d)                                   # :1106
            if trace:
                tr["xi"].append(xi.clone())
            c = torch.zeros(d, K)                                    # :1110
            if cfg.adaptive_forward_process:
                c = -Z.t()                                           # :1112
            if cfg.detach_forward:
                c = c.detach()                                       # :1114
            sel = (~stopped).float().unsqueeze(1).repeat(1, d)
            X_prop = (X + ((problem.b(X) + torch.mm(problem.sigma(X), c).t()) * dt32
                           + torch.mm(problem.sigma(X), xi.t()).t() * sq_dt32) * sel)   # :1116-1117
            new_sel = torch.ones(K).bool()                           # :1119
            new_sel = new_sel & ((t_n.squeeze() + dt32) <= T)        # :1131
            act = (new_sel & ~stopped)
            Y = (Y + ((-problem.h(n * dt32, X, Y_.squeeze(), Z) + torch.sum(Z * c.t(), 1)) * dt32
                      + torch.sum(Z * xi, 1) * sq_dt32) * act.float())       # :1141-1142
            X = (X * (~new_sel | stopped).float().unsqueeze(1).repeat(1, d)
                 + X_prop * act.float().unsqueeze(1).repeat(1, d))   # :1145-1146
            t_n = t_n + dt32 * act.float().unsqueeze(1)              # :1148
            X_t_n = torch.cat([X, t_n], 1)                           # :1149
            K_count = K_count + torch.sum(act)                       # :1152
            stopped = stopped | (~new_sel & ~stopped)                # :1154-1155
        if cfg.loss_method == "diffusion":
            loss = loss + cfg.alpha[0] * torch.mean((V(X_t_n).squeeze() - Y) ** 2)   # :1163
        V.zero_grad()                                                # :1164
        out["K_log"].append(int(K_count))                            # :1168
        if cfg.loss_method == "BSDE":
            loss = loss + torch.mean((Y - problem.f(X)) ** 2)        # :1174
        loss.backward()                                              # :1187
        if trace:
            tr["grads"] = [p.grad.detach().clone() for p in V.parameters()]
            tr["VN_minus_Y"] = None
        V.optim.step()                                               # :1188
        out["loss_log"].append(loss.item())                          # :1192
        if trace:
            out["traces"].append(tr)
    out["V"] = V
    return out


def fingerprint(module):
    res = []
    for name, p in module.named_parameters():
        q = p.detach().to(torch.float64)
        res.append(dict(name=name, shape=list(p.shape), sum=float(q.sum()), abs_sum=float(q.abs().sum()),
                        head=[float(v) for v in p.detach().reshape(-1)[:4].tolist()]))
    return res
