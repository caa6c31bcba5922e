"""GeneralSolver / EllipticSolver: diffusion / BSDE loss for parabolic terminal-value problems and for elliptic
exit-time problems -- API mirrors of the reference's ``solver.GeneralSolver`` (reference solver.py:934-1206) and
``solver.EllipticSolver`` (:560-826) for the hot-path part: ``loss_method in {'diffusion', 'BSDE'}`` on
``boundary in {'unbounded', 'unbounded_square', 'sphere', 'two_spheres', 'square'}`` (+ 'square-corner' for the elliptic solver).

One iteration (reference solver.py:1009-1201):
  sample X uniformly in the domain and t ~ U(0, T); Y = V(X, t);
  N Euler-Maruyama steps of the (optionally controlled) forward SDE, during which
      Z = sigma^T grad_x V(X, t),   Y += ((-h(n dt, X, V(X,t), Z) + Z.c) dt + Z.xi sqrt(dt)) * active
  (h sees the state BEFORE the move and V(X,t), not the running Y), trajectories freeze when
  t + dt > T or when they fail the exit test of a bounded domain (sphere: |X_n| < R on the state before the move;
  square: the proposal inside the box);  loss = alpha0 mean((V(X_N, t_N) - Y)^2) + alpha1 mean((V(X[:Kb], T) - f(X[:Kb]))^2)
  [+ alpha2 * Dirichlet / Neumann residual on K_boundary boundary points] ('diffusion')  or  mean((Y - f(X_N))^2)
  / mean((Y - g(X_N, t_N))^2) ('BSDE');  Adam on V.  EllipticSolver is the same step without the time input.

Execution plans, resolved once in ``train()``:
  * native (plan_general_native / plan_general_deep, hand-written HIP): V is a dense-concat net (d [+ 1] -> 1) of one to four
    hidden layers (DenseNet relu^2, DenseNet_tanh, the committor notebook's tanh^2 net, or any module of that structure),
    problem in the native catalogue, device is a GPU -- on every domain: 'two_spheres' (the committor problem: the batch size
    changes from iteration to iteration), 'square-corner', the BSDE loss with a Neumann boundary, ``loss_with_stopped``,
    ``sample_center`` and the ``K_test_log`` diagnostic included;
  * composite (this file): the reference op sequence with torch autograd on ``self.device`` (CPU runs, user-defined
    coefficients or nets: never an error).
PINN and the BSDE-2/3/4 variants are outside the scope of this build.
"""
import time
import warnings

import numpy as np
import torch

try:
    from .function_space import DenseNet, SingleParam
except ImportError:
    from function_space import DenseNet, SingleParam


def _default_device():
    return torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def sample_boundary_host(pb, Kb, d):
    """Uniform points on the boundary, drawn AND formed on the host in the reference's order (solver.py:1020-1038 == :650-668);
    the square variants shuffle with numpy's global generator like the reference.  Host arithmetic on purpose: the batch is
    then the reference's CPU batch bit for bit on every device -- the 'two_spheres' data g = [|x| > a] is evaluated exactly ON
    the inner sphere, where the last bit of the normalisation decides (GeneralSolver._g_on_boundary)."""
    if pb.boundary == 'sphere':
        Xb = torch.randn(Kb, d)
        return pb.boundary_distance * Xb / torch.sqrt(torch.sum(Xb ** 2, 1)).unsqueeze(1)
    half = int(Kb / 2)
    if pb.boundary == 'two_spheres':                              # solver.py:1023-1027 == :653-657: inner half, outer half
        Xb = torch.randn(Kb, d)
        radii = torch.tensor([pb.boundary_distance_1] * half + [pb.boundary_distance_2] * half).unsqueeze(1)
        return radii * Xb / torch.sqrt(torch.sum(Xb ** 2, 1)).unsqueeze(1)
    pick = np.concatenate([np.ones(half)[:, np.newaxis], np.zeros([half, d - 1])], 1)
    np.apply_along_axis(np.random.shuffle, 1, pick)
    lower = torch.tensor(np.concatenate([pick, np.zeros([half, d])]).astype(float)).bool()
    upper = torch.tensor(np.concatenate([np.zeros([half, d]), pick]).astype(float)).bool()
    if pb.boundary == 'square-corner':                            # solver.py:666-673: one coordinate on the corner planes
        Xb = (pb.X_r - pb.X_corner) * torch.rand(Kb, d) + pb.X_corner
        Xb[lower] = pb.X_corner
        Xb[upper] = pb.X_corner
        return Xb
    Xb = (pb.X_r - pb.X_l) * torch.rand(Kb, d) + pb.X_l
    Xb[lower] = pb.X_r if pb.one_boundary else pb.X_l
    Xb[upper] = pb.X_r
    return Xb


def sample_boundary(pb, Kb, d, dev):
    return sample_boundary_host(pb, Kb, d).to(dev)


def exit_test(pb, X, X_prop, elliptic):
    """new_selection of solver.py:1119-1129 / :758-767: True while the trajectory stays in the domain."""
    if pb.boundary == 'sphere':
        return torch.sqrt(torch.sum(X ** 2, 1)) < pb.boundary_distance          # the state BEFORE the move
    if pb.boundary == 'square':
        if pb.one_boundary:
            le = X_prop <= pb.X_r
            return torch.all(le, 1) if elliptic else torch.any(le, 1)
        return torch.all((X_prop >= pb.X_l) & (X_prop <= pb.X_r), 1)
    if pb.boundary == 'two_spheres':                              # solver.py:1122-1123 == :752-753, the state BEFORE the move
        r = torch.sqrt(torch.sum(X ** 2, 1))
        return (r > pb.boundary_distance_1) & (r < pb.boundary_distance_2)
    if pb.boundary == 'square-corner':                            # solver.py:759-760
        return torch.any(X_prop <= pb.X_r, 1)
    return torch.ones(X.shape[0], dtype=torch.bool, device=X.device)


def neumann_residual(V, Xb_in, g_val, d):
    """mean((grad_x V . x - g . x)^2) over the boundary batch (solver.py:1069-1074, :688-693)."""
    Xb_in = Xb_in.detach().clone().requires_grad_(True)
    grad_V, = torch.autograd.grad(V(Xb_in).squeeze().sum(), Xb_in, create_graph=True)
    xb = Xb_in[:, :d]
    return torch.mean((torch.sum(grad_V[:, :d] * xb, 1) - torch.sum(g_val * xb, 1)) ** 2)


_DOMAINS = ('unbounded', 'unbounded_square', 'sphere', 'two_spheres', 'square')


class GeneralSolver:
    elliptic = False

    def __init__(self, problem, name, seed=42, delta_t=0.01, N=50, lr=0.001, L=100000, K=200, K_boundary=50,
                 alpha=[1.0, 1.0, 1.0], adaptive_forward_process=False, detach_forward=True, print_every=100,
                 verbose=True, approx_method='Y', sample_center=False, loss_method='diffusion',
                 loss_with_stopped=False, K_test_log=None, PINN_log_variance=False, log_loss_parts=False,
                 boundary_loss=True, full_hessian=False, uniform_square=False, solve_linear_L2_projection=False,
                 device=None, backend='auto', noise='reference', mlp_dtype='auto', range_guard=True):
        self.problem, self.name = problem, name
        # split-product kernels ('auto' / 'f16x3'): redo an iteration on the fp32-MFMA kernels when an operand left the f16 range
        # (include/psp.h: psp_gen_config.range_flag); self.range_fallback_iterations counts them after train()
        self.range_guard, self.range_fallback_iterations = bool(range_guard), 0
        if mlp_dtype not in ('auto', 'fp32', 'f16x3', 'bf16', 'bf16_fwd'):
            raise ValueError("mlp_dtype must be 'auto', 'fp32', 'f16x3', 'bf16' or 'bf16_fwd'")
        self.mlp_dtype = mlp_dtype      # 'bf16': the matrix products of the native kernels on bf16 MFMA with fp32 accumulation
                                        # ('bf16_fwd': forward rollout only); own tolerance, see include/psp.h
        self.d = problem.d
        self.device = torch.device(device) if device is not None else getattr(problem, 'device', _default_device())
        self.seed = seed
        self.delta_t_np = delta_t
        self.delta_t = torch.tensor(self.delta_t_np).to(self.device)
        self.sq_delta_t = torch.sqrt(self.delta_t).to(self.device)
        self.N, self.lr, self.L = N, lr, L
        self.K, self.K_original, self.K_boundary = K, K, K_boundary
        self.alpha = alpha
        self.adaptive_forward_process = adaptive_forward_process
        self.detach_forward = detach_forward
        self.approx_method = approx_method
        self.sample_center = sample_center
        self.loss_method = loss_method
        self.loss_with_stopped = loss_with_stopped
        self.boundary_loss = boundary_loss
        self.PINN_log_variance = PINN_log_variance
        self.full_hessian = full_hessian
        self.uniform_square = uniform_square
        self.solve_linear_L2_projection = solve_linear_L2_projection
        self.print_every, self.verbose = print_every, verbose
        if backend not in ('auto', 'native', 'torch'):
            raise ValueError("backend must be 'auto', 'native' or 'torch'")
        self.backend, self.noise = backend, noise

        torch.manual_seed(seed)                                   # reference solver.py:978-983
        if approx_method == 'Y':
            self.V = DenseNet(d_in=self.d + 1, d_out=1, lr=lr, seed=seed).to(self.device)
        elif approx_method == 'Z':
            self.y_0 = SingleParam(lr=lr).to(self.device)
            self.Z = DenseNet(d_in=self.d + 1, d_out=self.d, lr=lr, seed=seed).to(self.device)

        self.K_test_log = K_test_log
        self.Y_0_log, self.loss_log, self.loss_log_domain, self.loss_log_boundary = [], [], [], []
        self.u_L2_log, self.V_L2_log = [], []
        self.V_test_L2, self.V_test_abs, self.V_test_rel_abs = [], [], []
        self.times, self.lambda_log, self.K_log = [], [], []
        self.log_loss_parts = log_loss_parts
        self.plan_name, self.plan_reason = None, None

    # -----------------------------------------------------------------------------------------
    def _check_scope(self):
        if self.approx_method != 'Y':
            raise NotImplementedError("approx_method='Z' is outside this build's scope")
        if self.loss_method not in ('diffusion', 'BSDE'):
            raise NotImplementedError("loss_method %r: only 'diffusion' and 'BSDE' are built" % self.loss_method)
        if self.problem.boundary not in _DOMAINS and not (self.elliptic and self.problem.boundary == 'square-corner'):
            raise NotImplementedError("boundary %r is not built (sphere / two_spheres / square / unbounded are)" % self.problem.boundary)
        if self.solve_linear_L2_projection:
            raise NotImplementedError('solve_linear_L2_projection is not built')

    def composite_only_reason(self):
        """Why this configuration runs on the composite torch plan whatever the net / problem (None: the native plan may take
        it).  Round 4: 'two_spheres', 'square-corner', the BSDE loss with a Neumann boundary, ``sample_center``,
        ``loss_with_stopped`` and ``K_test_log`` all run on the HIP kernels (plan_general_native.py); nothing of the hot path is
        left here."""
        return None

    @property
    def bounded(self):
        return 'unbounded' not in self.problem.boundary

    def sample_domain(self):
        """Initial points, drawn from the CPU generator in the reference's order (solver.py:1040-1056)."""
        K, d, dev, pb = self.K, self.d, self.device, self.problem
        if pb.boundary in ('unbounded', 'sphere'):
            if self.uniform_square:
                X = torch.rand(K, d).to(dev) * 2 - 1
                radial = torch.rand(K).unsqueeze(1).to(dev)
            else:
                X = torch.randn(K, d).to(dev)
                radial = (torch.rand(K).unsqueeze(1) ** (1 / d)).to(dev)
            return pb.boundary_distance * X / torch.sqrt(torch.sum(X ** 2, 1)).unsqueeze(1) * radial
        if pb.boundary == 'two_spheres':
            if self.elliptic and self.uniform_square:               # solver.py:702-704 (a radius per COMPONENT, as written there)
                X = torch.rand(K, d).to(dev) * 2 - 1
                return X / torch.sqrt(torch.sum(X ** 2, 1)).unsqueeze(1) * (
                    torch.rand(K, d).to(dev) * (pb.boundary_distance_2 - pb.boundary_distance_1) + pb.boundary_distance_1)
            # rejection from the outer ball: K_original draws, the points outside the inner ball stay and K follows
            # (solver.py:1048-1052 == :706-710)
            X = torch.randn(self.K_original, d).to(dev)
            X = pb.boundary_distance_2 * X / torch.sqrt(torch.sum(X ** 2, 1)).unsqueeze(1) * (
                torch.rand(self.K_original).unsqueeze(1) ** (1 / d)).to(dev)
            keep = torch.sqrt(torch.sum(X ** 2, 1)) > pb.boundary_distance_1
            self.K = int(torch.sum(keep))
            return X[keep, :]
        X = (pb.X_r - pb.X_l) * torch.rand(K, d).to(dev) + pb.X_l
        if pb.boundary == 'square-corner':                          # solver.py:708: points of the cut-out corner are mirrored
            corner = torch.all(X > pb.X_corner, 1)
            X[corner, :] = -X[corner, :]
        return X

    def _sample_boundary(self):
        """The boundary batch on ``self.device``; the host copy stays for _g_on_boundary."""
        self._Xb_host = sample_boundary_host(self.problem, self.K_boundary, self.d)
        return self._Xb_host.to(self.device)

    def _g_on_boundary(self, X_b, *t):
        """problem.g on the boundary batch.  On 'two_spheres' the data is discontinuous exactly on the sampled inner sphere
        (Committor.g = [|x| > a] at |x| = a up to rounding, problems.py:1569-1570): its value there hangs on the last bit of a
        norm, i.e. on the device's summation order.  Evaluated on the host copy of the batch it is the reference's CPU value
        on every device (the loss moves by 1 / K_boundary per flipped point otherwise).  Other domains: on the device."""
        host = getattr(self, '_Xb_host', None)
        if self.problem.boundary == 'two_spheres' and host is not None and host.shape == X_b.shape:
            try:
                return self.problem.g(host, *[ti.cpu() for ti in t]).to(X_b.device)
            except Exception:                                     # a user g tied to device tensors: evaluate it where they live
                pass
        return self.problem.g(X_b, *t)

    def boundary_residual(self, X_in_b, X_b, t_b):
        """Dirichlet / Neumann residual on the boundary batch (solver.py:1066-1074)."""
        pb = self.problem
        if pb.boundary_type == 'Dirichlet':
            return torch.mean((self.V(X_in_b).squeeze() - self._g_on_boundary(X_b, t_b.squeeze())) ** 2)
        if pb.boundary_type == 'Neumann':
            return neumann_residual(self.V, X_in_b, self._g_on_boundary(X_b, t_b.squeeze()), self.d)
        raise NotImplementedError('boundary_type %r' % pb.boundary_type)

    def train(self):
        torch.manual_seed(self.seed)
        self._check_scope()
        plan = self._choose_plan()
        if plan is not None:
            return plan.train()
        self._train_composite()

    def _plan_key(self):
        """What a native plan sizes its buffers / fixes its kernel configuration from (the value net is compared by identity:
        `model.V = DenseNet(...)` after a first train() must rebuild the plan, Allen-Cahn.ipynb:72)."""
        return (id(self.V), self.K, self.N, self.K_boundary, float(self.delta_t_np), self.loss_method, tuple(self.alpha),
                bool(self.adaptive_forward_process), bool(self.detach_forward), getattr(self, 'noise', None),
                getattr(self, 'mlp_dtype', None), bool(getattr(self, 'range_guard', True)), bool(getattr(self, 'uniform_square', False)), id(self.problem))

    def _choose_plan(self):
        if self.backend == 'torch':
            self.plan_name, self.plan_reason = 'torch', "backend='torch' requested"
            return None
        try:
            try:
                from . import plan_general_native as pgn
            except ImportError:
                import plan_general_native as pgn
        except ImportError:
            pgn = None
        reason = self.composite_only_reason()
        deep = False
        if reason is None:
            reason = 'the native diffusion-loss plan is not built yet' if pgn is None else pgn.native_eligibility(self)
            if reason is not None and pgn is not None and getattr(self, 'mlp_dtype', 'auto') in ('auto', 'fp32'):
                # value nets of other depths / widths (the notebooks' nets): the run-time-shaped kernels of csrc/genl_kernels.h
                try:
                    from . import plan_general_deep as pgd
                except ImportError:
                    import plan_general_deep as pgd
                deep_reason = pgd.deep_eligibility(self) or pgn.native_eligibility(self, deep=True)
                if deep_reason is None:
                    reason, deep = None, True
                else:
                    reason = deep_reason                          # (why NEITHER kernel family takes it)
        if reason is None:
            self.plan_name = 'native'
            plan = getattr(self, '_gen_plan', None)
            key = self._plan_key()
            if plan is None or plan.net is not self.V or plan.key != key:
                plan = pgd.GeneralDeepPlan(self) if deep else pgn.GeneralNativePlan(self)   # owns the flat parameters and Adam moments
                plan.key = key
                self._gen_plan = plan
            return plan
        if self.backend == 'native':
            raise NotImplementedError('native plan unavailable: ' + reason)
        if self.device.type == 'cuda':
            warnings.warn('path-space GeneralSolver: running the composite torch plan (%s)' % reason)
        self.plan_name, self.plan_reason = 'torch', reason
        return None

    def _log_test_error(self, modus):
        """K_test_log (solver.py:1193-1197 / :821-825): Monte-Carlo error of V on fresh points after the update."""
        try:
            from .utilities import compute_test_error
        except ImportError:
            from utilities import compute_test_error
        l2, mae, mre = compute_test_error(self, self.problem, self.K_test_log, self.device, modus)
        self.V_test_L2.append(l2)
        self.V_test_abs.append(mae)
        self.V_test_rel_abs.append(mre)

    def _train_composite(self):
        pb, dev, dt, sq = self.problem, self.device, self.delta_t, self.sq_delta_t
        d, T = self.d, pb.T
        a0, a1 = self.alpha[0], self.alpha[1]
        bounded = self.bounded
        for l in range(self.L):
            t_0 = time.time()
            loss = 0
            if self.sample_center:                               # solver.py:1015-1017 (a one-dimensional probe point, as written there)
                X_center = torch.zeros(1, 1).to(dev)
                loss = loss + torch.mean((self.V(X_center).squeeze() - pb.v_true(X_center).squeeze()) ** 2)
            if bounded:
                X_b = self._sample_boundary()
            X = self.sample_domain()
            K = self.K                                            # 'two_spheres': the rejection step sets it every iteration
            if bounded:
                t_b = torch.rand(self.K_boundary, 1).to(dev) * T
                X_t_b = torch.cat([X_b, t_b], 1)
            if self.loss_method != 'BSDE' and self.boundary_loss:
                Kb = self.K_boundary
                X_T = torch.cat([X[:Kb, :], T * torch.ones(Kb).to(dev).unsqueeze(1)], 1)
                loss = loss + a1 * torch.mean((self.V(X_T).squeeze() - pb.f(X[:Kb, :])) ** 2)
                if bounded:
                    loss = loss + self.alpha[2] * self.boundary_residual(X_t_b, X_b, t_b)
            X = X.clone().requires_grad_(True)
            t_n = torch.rand(K, 1).to(dev) * T
            X_t_n = torch.cat([X, t_n], 1)                       # time is the LAST input column here
            Y = self.V(X_t_n).squeeze()
            stopped = torch.zeros(K).bool().to(dev)
            K_count = 0
            n_done = 0
            for n in range(self.N):
                n_done = n
                if int(torch.sum(~stopped)) == 0:
                    break
                V_now = self.V(X_t_n)
                grad_V, = torch.autograd.grad(V_now.squeeze().sum(), X, create_graph=True)
                sig = pb.sigma(X)
                Z = torch.mm(sig.t(), grad_V.t()).t()
                xi = torch.randn(K, d).to(dev)
                c = torch.zeros(d, K).to(dev)
                if self.adaptive_forward_process:
                    c = -Z.t()
                if self.detach_forward:
                    c = c.detach()
                alive = (~stopped).float().unsqueeze(1).repeat(1, d)
                X_prop = X + ((pb.b(X) + torch.mm(sig, c).t()) * dt + torch.mm(sig, xi.t()).t() * sq) * alive
                in_time = exit_test(pb, X, X_prop, False) & ((t_n.squeeze() + dt) <= T)
                act = in_time & ~stopped
                actf = act.float()
                Y = Y + ((-pb.h(n * dt, X, V_now.squeeze(), Z) + torch.sum(Z * c.t(), 1)) * dt
                         + torch.sum(Z * xi, 1) * sq) * actf
                X = (X * (~in_time | stopped).float().unsqueeze(1).repeat(1, d)
                     + X_prop * actf.unsqueeze(1).repeat(1, d))
                t_n = t_n + dt * actf.unsqueeze(1)
                X_t_n = torch.cat([X, t_n], 1)
                K_count = K_count + torch.sum(act)
                stopped = stopped | (~in_time & ~stopped)
            if self.loss_method == 'diffusion':
                loss = loss + a0 * torch.mean((self.V(X_t_n).squeeze() - Y) ** 2)
            self.V.zero_grad()
            self.K_log.append(int(K_count))
            if self.loss_method == 'BSDE':
                if int(torch.sum(stopped)) != K:
                    print('Not all trajectories stopped.')
                if not bounded:
                    loss = loss + torch.mean((Y - pb.f(X)) ** 2)
                elif pb.boundary_type == 'Dirichlet':
                    loss = loss + torch.mean((Y - pb.g(X, t_n.squeeze())) ** 2)
                elif pb.boundary_type == 'Neumann':
                    # solver.py:1177-1183: trajectories that ran out of time are matched with f; the Neumann residual takes grad V of
                    # the LAST executed step (the state before its move) against the final X, over ALL trajectories, as written there
                    late = (t_n > (T - dt)).squeeze()
                    if int(torch.sum(late)) > 0:
                        loss = loss + torch.mean((Y[late] - pb.f(X[late, :])) ** 2)
                    if int(torch.sum(late)) < K:
                        loss = loss + torch.mean((torch.sum(grad_V * X, 1) - torch.sum(pb.g(X, t_n.squeeze()) * X, 1)) ** 2)
                else:
                    raise NotImplementedError('boundary_type %r' % pb.boundary_type)
            if self.loss_with_stopped:                            # solver.py:1185-1186
                loss = loss + torch.mean((Y[stopped] - pb.f(X[stopped, :])) ** 2)
            loss.backward()
            self.V.optim.step()
            self.loss_log.append(loss.item())
            self.V_L2_log.append(0.0)
            if self.K_test_log is not None:
                self._log_test_error('parabolic')
            self.times.append(time.time() - t_0)
            if self.verbose and l % self.print_every == 0:
                print('%d - loss = %.4e, v L2 error = %.4e, n = %d, active: %d/%d, %.2f'
                      % (l, self.loss_log[-1], self.V_L2_log[-1], n_done, int(torch.sum(~stopped)), K,
                         np.mean(self.times[-self.print_every:])))


class EllipticSolver(GeneralSolver):
    """Exit-time (elliptic) problems: API mirror of the reference's ``solver.EllipticSolver`` (solver.py:560-826).
    V = DenseNet(d -> 1) has no time input, ``problem.h(x, y, z)`` / ``problem.g(x)`` take no time, trajectories run
    until they leave the domain (at most N steps) and ``alpha = [domain, boundary]``.  Same two execution plans."""
    elliptic = True

    def __init__(self, problem, name, seed=42, delta_t=0.01, N=50, lr=0.001, L=100000, K=200, K_boundary=50,
                 alpha=[1.0, 1.0], adaptive_forward_process=False, detach_forward=True, print_every=100, verbose=True,
                 approx_method='Y', sample_center=False, loss_method='diffusion', loss_with_stopped=False,
                 K_test_log=None, PINN_log_variance=False, log_loss_parts=False, boundary_loss=True,
                 boundary_type='Dirichlet', variance_moment_split=False, full_hessian=False, uniform_square=False,
                 device=None, backend='auto', noise='reference', mlp_dtype='auto', v_l2_error_flag=True, range_guard=True):
        self.v_l2_error_flag = v_l2_error_flag   # False: skip the V_L2 diagnostic of solver.py:738 on the native plan (timed runs)
        super().__init__(problem, name, seed=seed, delta_t=delta_t, N=N, lr=lr, L=L, K=K, K_boundary=K_boundary,
                         alpha=alpha, adaptive_forward_process=adaptive_forward_process, detach_forward=detach_forward,
                         print_every=print_every, verbose=verbose, approx_method='skip', sample_center=sample_center,
                         loss_method=loss_method, loss_with_stopped=loss_with_stopped, K_test_log=K_test_log,
                         PINN_log_variance=PINN_log_variance, log_loss_parts=log_loss_parts, boundary_loss=boundary_loss,
                         full_hessian=full_hessian, uniform_square=uniform_square, device=device, backend=backend,
                         noise=noise, mlp_dtype=mlp_dtype, range_guard=range_guard)
        self.approx_method = approx_method
        self.boundary_type = boundary_type
        self.variance_moment_split = variance_moment_split
        torch.manual_seed(seed)                                   # reference solver.py:604-609
        if approx_method == 'Y':
            self.V = DenseNet(d_in=self.d, d_out=1, lr=lr, seed=seed).to(self.device)
        elif approx_method == 'Z':
            self.y_0 = SingleParam(lr=lr).to(self.device)
            self.Z = DenseNet(d_in=self.d, d_out=self.d, lr=lr, seed=seed).to(self.device)

    def _check_scope(self):
        if self.problem.boundary not in ('sphere', 'two_spheres', 'square', 'square-corner'):
            raise NotImplementedError("boundary %r is not built for EllipticSolver" % self.problem.boundary)
        if self.variance_moment_split or self.full_hessian:
            raise NotImplementedError('variance_moment_split / full_hessian are not built')
        super()._check_scope()

    def boundary_residual(self, X_b):
        """Dirichlet / Neumann residual on the boundary batch (solver.py:683-693); the type is the SOLVER's argument."""
        if self.boundary_type == 'Dirichlet':
            return torch.mean((self.V(X_b).squeeze() - self._g_on_boundary(X_b)) ** 2)
        if self.boundary_type == 'Neumann':
            return neumann_residual(self.V, X_b, self._g_on_boundary(X_b), self.d)
        raise NotImplementedError('boundary_type %r' % self.boundary_type)

    def train(self):
        torch.manual_seed(self.seed)
        np.random.seed(self.seed)                                 # solver.py:631
        self._check_scope()
        plan = self._choose_plan()
        if plan is not None:
            return plan.train()
        self._train_composite()

    def _train_composite(self):
        pb, dev, dt, sq = self.problem, self.device, self.delta_t, self.sq_delta_t
        d = self.d
        for l in range(self.L):
            t_0 = time.time()
            loss = 0
            if self.sample_center:                               # solver.py:643-645
                X_center = torch.zeros(1, 1).to(dev)
                loss = loss + torch.mean((self.V(X_center).squeeze() - pb.v_true(X_center).squeeze()) ** 2)
            X_b = self._sample_boundary()
            if self.loss_method != 'BSDE' and self.boundary_loss:
                loss = loss + self.alpha[1] * self.boundary_residual(X_b)
            X = self.sample_domain().clone().requires_grad_(True)
            K = self.K                                            # 'two_spheres': set by the rejection step of this iteration
            Y = self.V(X).squeeze()
            stopped = torch.zeros(K).bool().to(dev)
            V_L2 = torch.zeros(K)
            K_count, n_done = 0, 0
            for n in range(self.N):
                n_done = n
                V_now = self.V(X)
                Z, = torch.autograd.grad(V_now.squeeze().sum(), X, create_graph=True)
                sig = pb.sigma(X)
                Z = torch.mm(sig.t(), Z.t()).t()
                xi = torch.randn(K, d).to(dev)                   # drawn before the all-stopped test (solver.py:739-744)
                alive_b = ~stopped
                if int(torch.sum(alive_b)) == 0:
                    break
                if hasattr(pb, 'v_true'):
                    V_L2[alive_b.cpu()] += ((self.V(X[alive_b]).squeeze() - pb.v_true(X[alive_b].detach()).float().squeeze()) ** 2
                                            ).detach().cpu() * self.delta_t_np
                c = torch.zeros(d, K).to(dev)
                if self.adaptive_forward_process:
                    c = -Z.t()
                if self.detach_forward:
                    c = c.detach()
                alive = alive_b.float().unsqueeze(1).repeat(1, d)
                X_prop = X + ((pb.b(X) + torch.mm(sig, c).t()) * dt + torch.mm(sig, xi.t()).t() * sq) * alive
                inside = exit_test(pb, X, X_prop, True)
                act = inside & ~stopped
                actf = act.float()
                Y = Y + ((-pb.h(X, V_now.squeeze(), Z) + torch.sum(Z * c.t(), 1)) * dt + torch.sum(Z * xi, 1) * sq) * actf
                X = (X * (~inside | stopped).float().unsqueeze(1).repeat(1, d) + X_prop * actf.unsqueeze(1).repeat(1, d))
                K_count = K_count + torch.sum(act)
                stopped = stopped | (~inside & ~stopped)
            if self.loss_method == 'diffusion':
                loss = loss + self.alpha[0] * torch.mean((self.V(X).squeeze() - Y) ** 2)
            self.K_log.append(int(K_count))
            if self.loss_method == 'BSDE':
                if int(torch.sum(stopped)) != K:
                    print('Not all trajectories stopped.')
                loss = loss + torch.mean((pb.g(X) - Y) ** 2)
            if self.loss_with_stopped:                            # solver.py:803-804
                loss = loss + torch.mean((pb.g(X[stopped, :]) - Y[stopped]) ** 2)
            self.V.zero_grad()
            loss.backward()
            self.V.optim.step()
            self.loss_log.append(loss.item())
            self.V_L2_log.append(torch.mean(V_L2).item())
            if self.K_test_log is not None:
                self._log_test_error('elliptic')
            self.times.append(time.time() - t_0)
            if self.verbose and l % self.print_every == 0:
                print('%d - loss = %.4e, v L2 error = %.4e, n = %d, active: %d/%d, %.2f'
                      % (l, self.loss_log[-1], self.V_L2_log[-1], n_done, int(torch.sum(~stopped)), K,
                         np.mean(self.times[-self.print_every:])))
