"""PDE / control problem definitions (API mirror of the hot-path part of the reference's
problems.py).  A problem is a duck-typed object with attributes ``d, T, X_0, B, name`` and
methods ``b(x), sigma(x), h(t, x, y, z), f(...), g(x)`` -- exactly what the reference's
solvers call.  Differences from the reference:

* the device is a constructor argument (``device=None`` -> CUDA/HIP if present, else CPU)
  instead of the module global ``device = pt.device('cuda')`` (reference problems.py:11);
* ``native_spec()`` describes the coefficient functions as a closed catalogue entry
  (dense / diagonal / double-well drift, dense / identity sigma, diagonal quadratic costs)
  so that Solver can run the hand-written HIP rollout.  Objects without ``native_spec``
  (any user-supplied problem) run through the composite torch plan.

Random matrices are drawn from the torch CPU generator in the reference's order, so equal
seeds give identical A, B.
"""
import numpy as np
import torch
from scipy.linalg import expm

try:
    from . import native as _nat
except ImportError:  # flat import (sys.path points at this directory)
    import native as _nat


def default_device():
    return torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def _resolve(device):
    return default_device() if device is None else torch.device(device)


def coefficients_overridden(problem, names=('b', 'sigma', 'h', 'f', 'g')):
    """Name of a coefficient method that is NOT this module's catalogue implementation (a subclass override or an attribute
    patched onto the instance), else None.  native_spec() / general_native_spec() describe the catalogue formulas, so a
    problem whose b / sigma / h / f / g was replaced must run on the composite torch plan, which calls the Python methods."""
    for name in names:
        if name in getattr(problem, '__dict__', {}):
            return name
        fn = getattr(type(problem), name, None)
        if fn is not None and getattr(fn, '__module__', None) != __name__:
            return name
    return None


def _classify_matrix(M):
    """'identity' | ('scaled', s) | ('diag', vec) | 'dense' for a square matrix."""
    Mc = M.detach().cpu()
    d = Mc.shape[0]
    off = Mc - torch.diag(torch.diagonal(Mc))
    if torch.count_nonzero(off) != 0:
        return 'dense', None
    diag = torch.diagonal(Mc).clone()
    if torch.equal(diag, torch.ones(d)):
        return 'identity', None
    if torch.all(diag == diag[0]):
        return 'scaled', float(diag[0])
    return 'diag', diag


class _LinearDriftMixin:
    """Shared pieces of the Ornstein-Uhlenbeck families (dense A, dense B)."""

    def b(self, x):
        return torch.mm(self.A, x.t()).t()

    def sigma(self, x):
        return self.B

    def _linear_spec(self):
        spec = {}
        kind, val = _classify_matrix(self.A)
        if kind == 'dense':
            spec['drift'] = (_nat.DRIFT_DENSE, self.A.contiguous())
        else:
            diag = torch.diagonal(self.A).contiguous()
            spec['drift'] = (_nat.DRIFT_DIAG, diag)
        kind, val = _classify_matrix(self.B)
        if kind == 'identity':
            spec['sigma'] = (_nat.SIGMA_IDENTITY, None, 1.0)
        elif kind == 'scaled':
            spec['sigma'] = (_nat.SIGMA_SCALED_IDENTITY, None, val)
        else:
            spec['sigma'] = (_nat.SIGMA_DENSE, self.B.contiguous(), 1.0)
        return spec


class LLGC(_LinearDriftMixin):
    """Ornstein-Uhlenbeck dynamics with linear terminal cost g(x) = alpha . x
    (reference problems.py:14-65)."""

    def __init__(self, name='LLGC', d=1, off_diag=0, T=5, seed=42, device=None):
        self.device = _resolve(device)
        torch.manual_seed(seed)
        self.name, self.d, self.T = name, d, T
        self.A = (-torch.eye(d) + off_diag * torch.randn(d, d)).to(self.device)
        self.B = (torch.eye(d) + off_diag * torch.randn(d, d)).to(self.device)
        self.alpha = torch.ones(d, 1).to(self.device)
        self.X_0 = torch.zeros(d).to(self.device)
        self.boundary, self.one_boundary, self.X_l, self.X_r = 'square', False, -2.0, 2.0
        if not np.all(np.linalg.eigvals(self.A.cpu().numpy()).real < 0):
            print('not all EV of A are negative')

    def f(self, x, t):
        return torch.zeros(x.shape[0]).to(x.device)

    def h(self, t, x, y, z):
        return -0.5 * torch.sum(z ** 2, dim=1)

    def g(self, x):
        return torch.mm(x, self.alpha)[:, 0]

    u_true_x_independent = True    # lets the native plan tabulate u*(t_n) once (plan_native.py)

    def u_true(self, x, t):
        # u*(x,t) = -B^T exp(A^T (T-t)) alpha, independent of x (reference problems.py:51-53)
        A, B = self.A.cpu().numpy(), self.B.cpu().numpy()
        col = B.T.dot(expm(A.T * (self.T - t)).dot(self.alpha.cpu().numpy()))
        return -(col * np.ones(x.shape).T)

    def native_spec(self):
        spec = self._linear_spec()
        spec['runcost'] = (_nat.RUNCOST_ZERO, None)
        spec['term'] = (_nat.TERM_LINEAR, self.alpha[:, 0].contiguous())
        return spec


class LQGC(_LinearDriftMixin):
    """Linear-quadratic Gaussian control: running cost x'Px, terminal cost x'Rx
    (reference problems.py:118-175)."""

    def __init__(self, name='LQGC', delta_t=0.05, d=1, off_diag=0, T=5, seed=42, device=None):
        self.device = _resolve(device)
        torch.manual_seed(seed)
        self.name, self.d, self.T = name, d, T
        self.A = (-torch.eye(d) + off_diag * torch.randn(d, d)).to(self.device)
        self.B = (torch.eye(d) + off_diag * torch.randn(d, d)).to(self.device)
        self.delta_t = delta_t
        self.N = int(np.floor(self.T / self.delta_t))
        self.X_0 = torch.zeros(d).to(self.device)
        if not np.all(np.linalg.eigvals(self.A.cpu().numpy()).real < 0):
            print('not all EV of A are negative')
        self.P = 0.5 * torch.eye(d).to(self.device)
        self.Q = 0.5 * torch.eye(d).to(self.device)
        self.R = torch.eye(d).to(self.device)
        self._riccati()

    def _riccati(self):
        """Backward Euler recursion for the value-function matrices F_n and offsets G_n
        (reference problems.py:140-152); host-side diagnostics only."""
        A, B, Q, P = (m.cpu() for m in (self.A, self.B, self.Q, self.P))
        F = torch.zeros(self.N + 1, self.d, self.d)
        F[self.N] = self.R.cpu()
        Qinv = Q.inverse()
        for n in range(self.N, 0, -1):
            Fn = F[n]
            F[n - 1] = Fn + (A.t() @ Fn + Fn @ A - Fn @ B @ Qinv @ B.t() @ Fn + P) * self.delta_t
        G = torch.zeros(self.N + 1)
        for n in range(self.N, 0, -1):
            G[n - 1] = G[n] - torch.trace(B @ F[n] @ B) * self.delta_t
        self.F, self.G = F, G

    def f(self, x, t):
        return torch.sum(x.t() * torch.mm(self.P, x.t()), 0)

    def g(self, x):
        return torch.sum(x.t() * torch.mm(self.R, x.t()), 0)

    def h(self, t, x, y, z):
        return -0.5 * torch.sum(z ** 2, dim=1) - self.f(x, t)

    u_true_linear_in_x = True      # u*(x, t) = M(t) x: the native plan tabulates M(t_n) once and logs u_L2 from its path store

    def u_true(self, x, t):
        n = int(np.ceil(t / self.delta_t))
        gain = self.Q.cpu().inverse() @ self.B.cpu().t() @ self.F[n]
        return -(gain @ x.t()).detach().numpy()

    def v_true(self, x, t):
        n = int(np.ceil(t / self.delta_t))
        return -torch.mm(x, torch.mm(self.F[n], x.t())).t() + self.G[n]

    def native_spec(self):
        kp, _ = _classify_matrix(self.P)
        kr, _ = _classify_matrix(self.R)
        if kp == 'dense' or kr == 'dense':
            return None
        spec = self._linear_spec()
        spec['runcost'] = (_nat.RUNCOST_DIAG_QUAD, torch.diagonal(self.P).contiguous())
        spec['term'] = (_nat.TERM_DIAG_QUAD, torch.diagonal(self.R).contiguous())
        return spec


class _DoubleWellBase:
    def _setup(self, d, d_1, d_2, eta, kappa):
        self.d, self.d_1, self.d_2 = d, d_1, d_2
        self.eta, self.kappa = eta, kappa
        self.eta_ = torch.tensor([eta] * d_1 + [1.0] * d_2).to(self.device)
        self.kappa_ = torch.tensor([kappa] * d_1 + [1.0] * d_2).to(self.device)
        self.B = torch.eye(d).to(self.device)
        self.X_0 = -torch.ones(d).to(self.device)
        self.ref_sol_is_defined = False

    def V(self, x):
        return self.kappa * (x ** 2 - 1) ** 2

    def grad_V(self, x):
        return 4.0 * self.kappa_ * (x * (x ** 2 - torch.ones(self.d).to(x.device)))

    def b(self, x):
        return -self.grad_V(x)

    def sigma(self, x):
        return self.B

    def _well_cost(self, x):
        return (torch.sum(self.eta_ * (x - torch.ones(self.d).to(x.device)) ** 2, 1)).squeeze()

    # ---- reference solution of the ONE-dimensional problem by finite differences (reference problems.py:216-281, 336-476) ---------
    # psi = exp(-v) solves the linear backward equation d_t psi + L psi = 0, psi(T) = exp(-g), L = generator of dX = -V'(X) dt + dW
    # (beta = 2): symmetrised as A = D^-1 L D with D = exp(beta V / 2), a tridiagonal matrix on nx cells of [-xb, xb] (reflecting
    # ends), stepped backwards by implicit Euler; the optimal control is u* = -d_x v = (1 / psi) d_x psi taken as a forward
    # difference of log psi.  Written here as array expressions (the reference fills the matrix entry by entry); the band of the
    # implicit step and the FLOAT32 rounding of the control table -- the reference multiplies by its fp32 tensor B[0, 0], which
    # makes every entry an fp32 number -- are kept, because u_L2 logs are compared with the reference's to the last digits.
    def _fd_reference(self, potential, terminal, delta_t, xb, nx):
        import numpy as np
        from scipy.linalg import solve_banded
        beta = 2
        dx = 2.0 * xb / nx
        cells = np.arange(nx)
        mid = -xb + (cells + 0.5) * dx                                   # cell midpoints
        lo_far, lo_edge = -xb + (cells - 0.5) * dx, -xb + cells * dx      # the neighbour below and the edge between
        hi_far, hi_edge = -xb + (cells + 1.5) * dx, -xb + (cells + 1) * dx
        V = potential
        diag_lo = np.exp(beta * (V(mid) - V(lo_edge))) / dx ** 2
        diag_hi = np.exp(beta * (V(mid) - V(hi_edge))) / dx ** 2
        off_hi = np.exp(beta * 0.5 * (V(hi_far) + V(mid) - 2 * V(hi_edge))) / dx ** 2
        main = np.where(cells > 0, diag_lo, 0.0) + np.where(cells < nx - 1, diag_hi, 0.0)
        A_main, A_up = -main / beta, off_hi[:nx - 1] / beta              # A = -(.) / beta; A[i, i + 1] = +off_hi[i] / beta
        n_steps = int(self.T / delta_t)
        xvec = np.linspace(-xb, xb, nx, endpoint=True)
        scale, unscale = np.exp(beta * V(xvec) / 2), np.exp(-beta * V(xvec) / 2)
        band = -delta_t * np.vstack([np.append([0], A_up), A_main - n_steps / self.T, np.append(A_up, [0])])
        psi = np.zeros([n_steps + 1, nx])
        psi[n_steps] = np.exp(-terminal(xvec))
        for n in range(n_steps - 1, -1, -1):
            psi[n] = scale * solve_banded([1, 1], band, unscale * psi[n + 1])
        diff = -np.log(psi[:, 1:]) + np.log(psi[:, :-1])
        u = ((np.float32(-2 / beta) * np.float32(self.B[0, 0].item())) * diff.astype(np.float32) / np.float32(dx)).astype(np.float64)
        return dict(xb=xb, nx=nx, dx=dx, delta_t=delta_t, xvec=xvec), psi, u

    def _table_index(self, x_col):
        """Grid cell of every entry of a (K,) column of states, as the reference computes it (fp32 arithmetic, states clamped to
        the grid) -- including its decrement of the LAST entry's index by two (problems.py:270, 276)."""
        x_col = x_col.detach().cpu().float().reshape(-1)
        idx = torch.floor((torch.clamp(x_col, -self.xb, self.xb - 2 * self.dx) + self.xb) / self.dx).long()
        idx[-1] -= 2
        return idx

    def _time_index(self, t):
        import numpy as np
        return int(np.ceil(float(t) / self.delta_t))

    def _value_index(self, x_col):
        """Grid cell for the VALUE tables (problems.py:392-394, 592-594): no clamp, same decrement of the last entry."""
        x_col = x_col.detach().cpu().float().reshape(-1)
        idx = torch.floor((x_col + self.xb) / self.dx).long()
        idx[-1] -= 2
        return idx

    # ---- the two tables of the multidimensional classes: coordinates < d_1 use (kappa, eta), the others (1, 1) ----------------------
    def V_2(self, x):
        return (x ** 2 - 1) ** 2

    def g_1(self, x_1):
        return self.eta * (x_1 - 1) ** 2

    def g_2(self, x_1):
        return (x_1 - 1) ** 2

    def _take_grid(self, grid):
        self.xb, self.nx, self.dx, self.delta_t, self.xvec = grid['xb'], grid['nx'], grid['dx'], grid['delta_t'], grid['xvec']

    def compute_reference_solution(self, delta_t=0.005, xb=2.5, nx=1000):
        grid, self.psi, self.u = self._fd_reference(self.V, self.g_1, delta_t, xb, nx)
        self._take_grid(grid)
        self._publish_u_true()

    def compute_reference_solution_2(self, delta_t=0.005, xb=2.5, nx=1000):
        grid, self.psi_2, self.u_2 = self._fd_reference(self.V_2, self.g_2, delta_t, xb, nx)
        self._take_grid(grid)
        self._publish_u_true()

    def _tables_ready(self):
        return hasattr(self, 'u') and (self.d_2 == 0 or hasattr(self, 'u_2'))

    def _publish_u_true(self):
        if self._tables_ready():
            self.ref_sol_is_defined = True
            self.u_true = self._u_true

    def _table_read(self, table, idx, t, transform=None):
        import numpy as np
        vals = table[self._time_index(t), idx.numpy()]
        return np.array(vals if transform is None else transform(vals)).reshape([1, len(idx)])

    def u_true_1(self, x, t):
        return self._table_read(self.u, self._table_index(x), t)

    def u_true_2(self, x, t):
        return self._table_read(self.u_2, self._table_index(x), t)

    def _value_transform(self):
        import numpy as np
        return None if getattr(self, 'modus', 'HJB') == 'linear' else (lambda p: -np.log(p))

    def v_true_1(self, x, t):
        return self._table_read(self.psi, self._value_index(x), t, self._value_transform())

    def v_true_2(self, x, t):
        return self._table_read(self.psi_2, self._value_index(x), t, self._value_transform())

    def _u_true(self, x, t):
        import numpy as np
        cols = [self.u_true_1(x[:, i], t).T for i in range(self.d_1)] + [self.u_true_2(x[:, i], t).T for i in range(self.d_1, self.d)]
        return np.concatenate(cols, 1).T

    def u_true_tables(self):
        """Tables and the coordinate -> table map for the native plan's device-side u_L2 log (plan_native._ul2_from_path)."""
        if not self.ref_sol_is_defined:
            return None
        tables = [self.u] + ([self.u_2] if self.d_2 > 0 else [])
        return dict(tables=tables, group_of_dim=[0] * self.d_1 + [1] * self.d_2, xb=self.xb, dx=self.dx, delta_t=self.delta_t)


class DoubleWell(_DoubleWellBase):
    """One-dimensional double-well potential (reference problems.py:178-283): dX = -V'(X) dt + dW, V = kappa (x^2 - 1)^2,
    g = eta (x - 1)^2; ``compute_reference_solution()`` tabulates the optimal control, ``u_true`` / ``v_true`` read the table."""

    def __init__(self, name='Double well', d=1, T=1, eta=1, kappa=1, device=None):
        self.device = _resolve(device)
        self.name, self.T = name, T
        self._setup(d, d, 0, eta, kappa)
        if d != 1:
            print('The double well example is only implemented for d = 1.')

    def h(self, t, x, y, z):
        return -0.5 * torch.sum(z ** 2, dim=1)

    def f(self, x, t):
        return torch.zeros(x.shape[0]).to(x.device)

    def g(self, x):
        return (self.eta * (x - 1) ** 2).squeeze()

    def grad_V(self, x):
        return 4.0 * self.kappa * x * (x ** 2 - 1)

    def compute_reference_solution(self, delta_t=0.005, xb=2.5, nx=1000):
        grid, self.psi, self.u = self._fd_reference(self.V, lambda x: self.eta * (x - 1) ** 2, delta_t, xb, nx)
        self.xb, self.nx, self.dx, self.delta_t, self.xvec = grid['xb'], grid['nx'], grid['dx'], grid['delta_t'], grid['xvec']
        self.ref_sol_is_defined = True

    def v_true(self, x, t):
        import numpy as np
        x = x.detach().cpu().float()
        idx = torch.floor((x.squeeze(0) + self.xb) / self.dx).long()
        idx[-1] -= 2
        return np.array(-np.log(self.psi[self._time_index(t), idx.numpy()])).reshape([1, len(idx)])

    def u_true(self, x, t):
        import numpy as np
        idx = self._table_index(x).reshape(x.shape) if x.dim() > 1 else self._table_index(x)
        return np.array(self.u[self._time_index(t), idx.numpy()]).reshape([1, len(idx)])

    def u_true_tables(self):
        """What the native plan needs to log u_L2 on the device (plan_native._ul2_from_path): one table for all coordinates."""
        if not self.ref_sol_is_defined:
            return None
        return dict(tables=[self.u], group_of_dim=[0] * self.d, xb=self.xb, dx=self.dx, delta_t=self.delta_t)

    def native_spec(self):
        return {
            'drift': (_nat.DRIFT_DOUBLE_WELL, self.kappa_.float().contiguous()),
            'sigma': (_nat.SIGMA_IDENTITY, None, 1.0),
            'runcost': (_nat.RUNCOST_ZERO, None),
            'term': (_nat.TERM_SHIFTED_QUAD, self.eta_.float().contiguous()),
        }


class DoubleWell_multidim(_DoubleWellBase):
    """Independent double-well potential in every coordinate, identity diffusion
    (reference problems.py:285-334)."""

    def __init__(self, name='Double well', d=1, d_1=1, d_2=0, T=1, eta=1, kappa=1, device=None):
        self.device = _resolve(device)
        self.name, self.T = name, T
        self._setup(d, d_1, d_2, eta, kappa)
        self.boundary, self.boundary_distance = 'unbounded', 2.0

    def h(self, t, x, y, z):
        return -0.5 * torch.sum(z ** 2, dim=1)

    def f(self, x, t):
        return torch.zeros(x.shape[0]).to(x.device)

    def g(self, x):
        return self._well_cost(x)

    # reference solution: the coordinates decouple -- one table for the first d_1 (kappa, eta), one for the others (1, 1)
    # (reference problems.py:336-476; methods on _DoubleWellBase).  The reference defines u_true on the class, where it fails until
    # both tables exist; here ``u_true`` appears with the tables, so a run without them logs no u_L2 instead of raising.
    def v_true(self, x, t):
        return None                                                      # problems.py:472-473

    def native_spec(self):
        return {
            'drift': (_nat.DRIFT_DOUBLE_WELL, self.kappa_.float().contiguous()),
            'sigma': (_nat.SIGMA_IDENTITY, None, 1.0),
            'runcost': (_nat.RUNCOST_ZERO, None),
            'term': (_nat.TERM_SHIFTED_QUAD, self.eta_.float().contiguous()),
        }


class DoubleWell_multidim_for_general_solver(_DoubleWellBase):
    """Same dynamics posed as a parabolic terminal-value problem for GeneralSolver
    (reference problems.py:479-534): ``f(x)`` is the terminal condition."""

    def __init__(self, name='Double well', d=1, d_1=1, d_2=0, T=1, eta=1, kappa=1, modus='HJB', device=None):
        self.device = _resolve(device)
        self.name, self.T, self.modus = name, T, modus
        self._setup(d, d_1, d_2, eta, kappa)
        self.boundary, self.X_l, self.X_r = 'unbounded_square', -2.5, 2.5

    def h(self, t, x, y, z):
        if self.modus == 'linear':
            return torch.zeros(x.shape[0]).to(x.device)
        return -0.5 * torch.sum(z ** 2, dim=1)

    def f(self, x):
        if self.modus == 'linear':
            return torch.exp(-self._well_cost(x))
        return self._well_cost(x)

    def v_true(self, x, t):
        """Reference value from the per-coordinate tables (reference problems.py:682-685): the values add; in modus 'linear' the
        tabulated psi = exp(-v) multiply."""
        import numpy as np
        cols = np.array([self.v_true_1(x[:, i], t).squeeze() for i in range(self.d_1)]
                        + [self.v_true_2(x[:, i], t).squeeze() for i in range(self.d_1, self.d)])
        return np.prod(cols, 0) if self.modus == 'linear' else np.sum(cols, 0)

    def _publish_u_true(self):                                           # GeneralSolver never asks hasattr(problem, 'u_true')
        self.ref_sol_is_defined = self._tables_ready()

    def u_true(self, x, t):
        return self._u_true(x, t)                                        # problems.py:687-688

    def general_native_spec(self):
        return {'drift': (_nat.DRIFT_DOUBLE_WELL, self.kappa_.float().contiguous()), 'sigma_scale': 1.0,
                'h': _nat.GH_ZERO if self.modus == 'linear' else _nat.GH_QUAD}


class AllenCahn:
    """Allen-Cahn reaction term h = y - y^3, sigma = sqrt(2) I (reference problems.py:1175-1217,
    torch branch ``modus='pt'``)."""

    def __init__(self, name='Allen-Cahn', d=1, T=0.3, seed=42, modus='pt', device=None):
        self.device = _resolve(device)
        np.random.seed(seed)
        self.name, self.d, self.T, self.modus = name, d, T, modus
        self.B = np.eye(d) * np.sqrt(2)
        self.B_pt = torch.tensor(self.B).float().to(self.device)
        self.X_0 = np.zeros(d)
        self.sigma_modus, self.boundary, self.boundary_distance = 'constant', 'unbounded', 2.0

    def b(self, x):
        return torch.zeros(x.shape).to(x.device)

    def sigma(self, x):
        return self.B_pt

    def h(self, t, x, y, z):
        return y - y ** 3

    def f(self, x):
        return 1 / (2 + 2 / 5 * torch.sum(x ** 2, 1))

    def general_native_spec(self):
        return {'drift': (_nat.DRIFT_ZERO, None), 'sigma_scale': float(self.B_pt[0, 0]), 'h': _nat.GH_ALLEN_CAHN}


class HeatEquation:
    """Heat equation with terminal condition |x|^2 (reference problems.py:1733-1764)."""

    def __init__(self, name='Heat equation', d=1, T=1, seed=42, device=None):
        self.device = _resolve(device)
        torch.manual_seed(seed)
        self.name, self.d, self.T = name, d, T
        self.B = torch.sqrt(torch.tensor(2.0)) * torch.eye(d).to(self.device)
        self.boundary, self.boundary_type, self.boundary_distance = 'unbounded', 'Dirichlet', 1.0

    def b(self, x):
        return torch.zeros(x.shape).to(x.device)

    def sigma(self, x):
        return self.B

    def g(self, x, t):
        return torch.zeros(x.shape[0]).to(x.device)

    def h(self, t, x, y, z):
        return torch.zeros(x.shape[0]).to(x.device)

    def f(self, x):
        return torch.sum(x ** 2, 1)

    def general_native_spec(self):
        return {'drift': (_nat.DRIFT_ZERO, None), 'sigma_scale': float(self.B[0, 0]), 'h': _nat.GH_ZERO}

    def v_true(self, x, t):
        return torch.sum(x ** 2, 1) + 2 * (self.T - t) * self.d


# ---------------------------------------------------------------------------------------------
# bounded domains: the exponential-on-the-ball family and a box problem
# ---------------------------------------------------------------------------------------------
class _ExpBall:
    """v(x[,t]) = exp(alpha |x|^2 [+ t]) on the unit ball, b = 0, sigma = sqrt(2) I (reference problems.py:962-1172).
    ``_nl`` selects the nonlinearity of h: 'none' | 'sq' | 'sin'; ``_parabolic`` adds the time argument, the extra -y
    and the 2t in the exponent (ExponentialOnSphereNonlinearParabolic, :1166)."""
    _nl, _parabolic = 'none', False

    def _setup(self, name, d, alpha, boundary_type, device):
        self.device = _resolve(device)
        self.name, self.d, self.alpha = name, d, alpha
        self.B = (torch.sqrt(torch.tensor(2.0)) * torch.eye(d)).to(self.device)
        self.X_0 = torch.zeros(d).to(self.device)
        self.Y_0 = torch.zeros(1).to(self.device)
        self.boundary, self.boundary_distance = 'sphere', 1.0
        self.boundary_type = boundary_type

    def b(self, x):
        return torch.zeros(x.shape).to(x.device)

    def sigma(self, x):
        return self.B

    def _r2(self, x):
        return torch.sum(x ** 2, 1)

    def _h(self, x, y, t):
        al, d = self.alpha, self.d
        if self._nl == 'none':
            return -al * y * (al * 4 * self._r2(x) + 2 * d)
        lin = -2 * al * y * (al * 2 * self._r2(x) + d)
        if self._parabolic:
            return lin - y + torch.sin(torch.exp(2 * al * self._r2(x) + 2 * t) - y ** 2)
        e = torch.exp(2 * al * self._r2(x))
        return lin + e - y ** 2 if self._nl == 'sq' else lin + torch.sin(e - y ** 2)

    def u_true(self, x):
        return -2 * torch.sqrt(torch.tensor(2.0)) * self.alpha * x * torch.exp(self.alpha * self._r2(x).unsqueeze(1))

    def general_native_spec(self):
        kind = {'none': _nat.GH_EXPBALL_LIN, 'sq': _nat.GH_EXPBALL_SQ, 'sin': _nat.GH_EXPBALL_SIN}[self._nl]
        par = 1.0 if self._parabolic else 0.0
        return {'drift': (_nat.DRIFT_ZERO, None), 'sigma_scale': float(self.B[0, 0]), 'h': kind,
                'h_par': (float(self.alpha), float(self.d), par, par)}


class _ExpBallElliptic(_ExpBall):
    def f(self, x, t=None):
        return torch.zeros(x.shape[0]).to(x.device)

    def g(self, x):
        if self.boundary_type == 'Neumann':
            return 2 * self.alpha * x * torch.exp(self.alpha * self._r2(x)).unsqueeze(1)
        return torch.exp(self.alpha * self._r2(x))

    def h(self, x, y, z):
        return self._h(x, y, None)

    def v_true(self, x):
        return torch.exp(self.alpha * self._r2(x))


class ExponentialOnSphere(_ExpBallElliptic):
    """Linear elliptic problem (reference problems.py:962-993)."""

    def __init__(self, name='Exponential on sphere', d=2, alpha=1.0, device=None):
        self._setup(name, d, alpha, 'Dirichlet', device)


class ExponentialOnBallNonlinear(_ExpBallElliptic):
    """h carries exp(2 alpha |x|^2) - y^2 (reference problems.py:995-1029)."""
    _nl = 'sq'

    def __init__(self, name='Exponential on ball nonlinear', d=2, alpha=1.0, boundary_type='Dirichlet', device=None):
        self._setup(name, d, alpha, boundary_type, device)


class ExponentialOnBallNonlinearSin(_ExpBallElliptic):
    """h carries sin(exp(2 alpha |x|^2) - y^2) (reference problems.py:1031-1065)."""
    _nl = 'sin'

    def __init__(self, name='Exponential on ball nonlinear', d=2, alpha=1.0, boundary_type='Dirichlet', device=None):
        self._setup(name, d, alpha, boundary_type, device)


class ExponentialOnSphereNonlinearParabolic(_ExpBall):
    """Parabolic version for GeneralSolver (reference problems.py:1137-1172); ``boundary_type`` is 'Dirichlet' and the
    Neumann notebook sets it to 'Neumann' on the instance."""
    _nl, _parabolic = 'sin', True

    def __init__(self, name='Exponential on ball', d=2, T=1.0, alpha=1.0, device=None):
        self._setup(name, d, alpha, 'Dirichlet', device)
        self.T = T

    def f(self, x):
        return torch.exp(self.alpha * self._r2(x) + self.T)

    def g(self, x, t):
        if self.boundary_type == 'Neumann':
            return 2 * self.alpha * x * torch.exp(self.alpha * self._r2(x) + t).unsqueeze(1)
        return torch.exp(self.alpha * self._r2(x) + t)

    def h(self, t, x, y, z):
        return self._h(x, y, t)

    def v_true(self, x, t):
        return torch.exp(self.alpha * self._r2(x) + t)


class Committor:
    """Committor function between two concentric spheres of radius a = 1 and c = 2 (reference problems.py:1546-1579): b = 0,
    sigma = I, h = 0, boundary data 0 on the inner and 1 on the outer sphere; ``boundary = 'two_spheres'`` (EllipticSolver:
    the batch size changes with the rejection step of every iteration; native: PSP_DOM_ANNULUS)."""

    def __init__(self, name='Committor', d=2, alpha=1.0, device=None):
        self.device = _resolve(device)
        self.name, self.d = name, d
        self.a, self.c = 1.0, 2.0
        self.B = torch.eye(d).to(self.device)
        self.X_0 = torch.zeros(d).to(self.device)
        self.Y_0 = torch.zeros(1).to(self.device)
        self.boundary = 'two_spheres'
        self.boundary_distance_1, self.boundary_distance_2 = self.a, self.c

    def b(self, x):
        return torch.zeros(x.shape).to(x.device)

    def sigma(self, x):
        return self.B

    def f(self, x):
        return torch.zeros(x.shape[0]).to(x.device)

    def g(self, x):
        return (torch.sqrt(torch.sum(x ** 2, 1)) > self.a).float()

    def h(self, x, y, z):
        return torch.zeros(x.shape[0]).to(x.device)

    def u_true(self, x):
        return torch.zeros(x.shape)

    def v_true(self, x):
        r = torch.sqrt(torch.sum(x ** 2, 1))
        return (self.a ** 2 - r ** (2 - self.d) * self.a ** self.d) / (self.a ** 2 - self.c ** (2 - self.d) * self.a ** self.d)

    def general_native_spec(self):
        return {'drift': (_nat.DRIFT_ZERO, None), 'sigma_scale': 1.0, 'h': _nat.GH_ZERO}


class QuadraticOnBox:
    """NOT a reference class: b = 0, sigma = scale I, h = -|z|^2/2 (or 0) with data |x|^2 on the box [X_l, X_r]^d.
    It exercises the 'square' exit tests of the solvers (reference solver.py:1125-1129, :762-767) with coefficients
    the kernels have; ``parabolic`` selects the GeneralSolver (h(t,x,y,z), f(x), g(x,t)) or the EllipticSolver
    (h(x,y,z), g(x)) calling convention."""

    def __init__(self, name='Quadratic on box', d=2, T=0.5, X_l=-1.0, X_r=1.0, one_boundary=False, scale=1.0,
                 parabolic=True, quad_h=True, device=None):
        self.device = _resolve(device)
        self.name, self.d, self.T = name, d, T
        self.B = (scale * torch.eye(d)).to(self.device)
        self.boundary, self.boundary_type = 'square', 'Dirichlet'
        self.X_l, self.X_r, self.one_boundary = X_l, X_r, one_boundary
        self.parabolic, self.quad_h = parabolic, quad_h

    def b(self, x):
        return torch.zeros(x.shape).to(x.device)

    def sigma(self, x):
        return self.B

    def h(self, *args):
        z = args[-1]
        return -0.5 * torch.sum(z ** 2, dim=1) if self.quad_h else torch.zeros(z.shape[0]).to(z.device)

    def f(self, x, t=None):
        return torch.sum(x ** 2, 1) if self.parabolic else torch.zeros(x.shape[0]).to(x.device)

    def g(self, x, t=None):
        return torch.sum(x ** 2, 1) + (self.T - t) if self.parabolic else torch.sum(x ** 2, 1)

    def v_true(self, x, t=None):
        return torch.sum(x ** 2, 1)

    def general_native_spec(self):
        return {'drift': (_nat.DRIFT_ZERO, None), 'sigma_scale': float(self.B[0, 0]),
                'h': _nat.GH_QUAD if self.quad_h else _nat.GH_ZERO}
