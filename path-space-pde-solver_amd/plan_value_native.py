"""Native (HIP) plan for Solver.train with approx_method='value_function' (reference solver.py:93-97, 334-339, 438-440).

The ansatz is a value net Y_n(x) = V([t, x]) (DenseNet(d+1 -> 1), time_approx='inner'); the control enters through
Z_n = sigma grad_x V(X_n, n) (autograd with create_graph=True in the reference), Y starts at V(X_0, 0), and the loss is the
usual path-space loss of D = Y_N - g(X_N) plus the consistency term mean_k sum_{n=1}^{N-1} (V(X_n, n) - Y_n)^2
(`additional_loss`, solver.py:438-440, 499).  That is exactly the work of the GeneralSolver kernels (csrc/gen_kernels.h:
value net, reverse sweep for grad_x V, tangent pass for the theta-gradient of Z), with two additions made for this plan:
  * psp_gen_config.v_steps_out / y_steps_out : V(X_n, n) and the running Y at every step (the consistency term);
  * psp_gen_config.per_sample_weights        : the backward kernel takes the weight of the tangent part and the
                                               coefficient of grad_theta V PER SAMPLE (n, k) instead of per trajectory.
With the state path detached (detach_forward=True, or a non-adaptive forward process)
    dL/dtheta = sum_{m,k} WS[m,k] d(increment_m)/dtheta + sum_{n,k} AV[n,k] grad_theta V(X_n, n),
    WS[m] = w^D - (2/K) sum_{n=m+1}^{N-1} r_n,   AV[0] = WS[0],  AV[n] = (2/K) r_n  (1 <= n <= N-1),  r_n = V(X_n,n) - Y_n,
    w^D_k = dLoss/dD_k (log-variance: (2/K)(D_k - mean D), moment: (2/K) D_k).
Details that differ from GeneralSolver and are absorbed by the parameter index map (native_shapes.GenParamPad):
the net input is [t, x] with time FIRST (solver.py:343-344) and the reference feeds the STEP INDEX n as the time
(Y_n(X, n), solver.py:336, 439), while the kernel's time register holds n * dt: the time rows are moved to the kernels' last
input row and scaled by 1/dt on the way in (and the gradient on the way out).
"""
import ctypes as C

import torch

try:
    from . import native as nat
    from . import native_shapes as shapes
    from . import sharding
    from .function_space import DenseNet
    from .plan_native import PlanUnsupported, _overridden
except ImportError:
    import native as nat
    import native_shapes as shapes
    import sharding
    from function_space import DenseNet
    from plan_native import PlanUnsupported, _overridden


def value_eligibility(solver):
    """None if Solver(approx_method='value_function') can run on the GeneralSolver kernels, else the reason."""
    s = solver
    if s.device.type != 'cuda':
        return 'device is %s (the HIP rollout needs a GPU)' % s.device
    if s.approx_method != 'value_function' or s.time_approx != 'inner':
        return "only approx_method='value_function' with time_approx='inner' (with 'outer' the reference itself fails)"
    if s.loss_method not in ('log-variance', 'moment'):
        return "loss_method %r is not native for the value-function ansatz (log-variance, moment)" % s.loss_method
    if s.adaptive_forward_process and not s.detach_forward:
        return 'detach_forward=False back-propagates through the state path (not native for the value-function ansatz)'
    if s.learn_Y_0:
        return 'learn_Y_0 has no meaning for the value-function ansatz (Y_0 = V(X_0, 0))'
    if s.u_l2_error_flag or s.burgers_drift or s.compute_gradient_variance > 0 or s.log_gradient \
            or s.metastability_logs is not None or s.IS_variance_K > 0:
        return 'per-step / per-iteration diagnostics (u_L2, gradient logs, metastability, in-loop IS) are not native here'
    nets = getattr(s, 'y_n', None)
    if not isinstance(nets, list) or len(nets) != 1:
        return 'y_n is not a single value net'
    V = nets[0]
    dims = getattr(V, 'nn_dims', None)
    deep = _deep_net(s, V)
    if deep is None and (not isinstance(V, DenseNet) or dims is None or len(dims) != 4 or dims[1] != dims[2] or dims[3] != 1
                         or dims[0] != s.d + 1):
        return 'the value net is neither a DenseNet(%d -> 1) with two equal hidden widths nor a dense-concat net the ' \
               'run-time-shaped kernels take (%s)' % (s.d + 1, _deep_net(s, V, why=True))
    spec_fn = getattr(s.problem, 'native_spec', None)
    spec = spec_fn() if spec_fn is not None else None
    if spec is None:
        return 'problem has no native_spec() (coefficients outside the native catalogue)'
    over = _overridden(s.problem)
    if over is not None:
        return 'problem.%s is not the catalogue implementation native_spec() describes' % over
    if spec['sigma'][0] not in (nat.SIGMA_IDENTITY, nat.SIGMA_SCALED_IDENTITY):
        return 'a dense sigma is not built into the value-net kernels (identity / scaled identity are)'
    if spec['drift'][0] == nat.DRIFT_DENSE:
        return 'a dense drift matrix is not built into the value-net kernels (zero / diagonal / double well are)'
    if spec['runcost'][0] != nat.RUNCOST_ZERO:
        return 'a running cost f(x) is not built into the value-net kernels (h = -|z|^2 / 2 is)'
    if not nat.is_built():
        raise nat.NativeLibraryError('libpsp_hip.so is not built; run __graft_entry__.build()')
    if deep is None and not shapes.gen_candidates(s.d, dims[1]):
        return 'no compiled kernel instance covers d=%d, H=%d (see csrc/gen_instances.def)' % (s.d, dims[1])
    return None


def _deep_net(solver, V, why=False):
    """The dense-concat description of a value net that is NOT the two-equal-hidden-layer relu^2 DenseNet of the templated kernels
    (any depth 1-4, widths <= 128, relu^2 / tanh^2 / tanh: csrc/genl_kernels.h), or None (why=True: the reason instead)."""
    try:
        from . import plan_general_deep as pgd
    except ImportError:
        import plan_general_deep as pgd
    spec = pgd.value_net_spec(V, solver.d + 1)
    if isinstance(spec, str):
        return spec if why else None
    dims = spec['dims']
    templated = (isinstance(V, DenseNet) and len(dims) == 4 and dims[1] == dims[2] and spec['act'] == 'relu2'
                 and bool(shapes.gen_candidates(solver.d, dims[1])))
    if templated:
        return 'the templated kernels take it' if why else None
    L = len(dims) - 2
    if L < 1 or L > 4 or max(dims[1:-1]) > 128 or dims[0] > 112:
        return 'outside 1-4 hidden layers of <= 128 units, input <= 112' if why else None
    return 'ok' if why else spec


class ValueNativePlan:
    def __init__(self, solver, noise='reference'):
        reason = value_eligibility(solver)
        if reason is not None:
            raise PlanUnsupported(reason)
        s = solver
        self.s, self.noise, self.lib, self.dev = s, noise, nat.load(), s.device
        dev = self.dev
        self.dist, self.rank, self.world = sharding.dist_info()
        try:
            lo, hi = sharding.shard_bounds(s.K, self.rank, self.world)
        except ValueError as e:
            raise PlanUnsupported(str(e))
        self.lo, self.hi, self.K_local = lo, hi, hi - lo
        self.net = s.y_n[0]
        self.key = None
        self.H = self.net.nn_dims[1]
        self.deep = _deep_net(s, self.net)              # value nets of other depths / activations: csrc/genl_kernels.h
        self._flatten(self.net if self.deep is None else self.deep['params'])
        spec = s.problem.native_spec()
        self._keep = []
        if self.deep is not None:
            self.gcfg = nat.GenlConfig()
            cfg = self.gcfg.base
            cfg.d = s.d
        else:
            cfg = nat.GenConfig()
        cfg.K_local, cfg.N, cfg.k_offset = self.K_local, s.N, lo
        cfg.dt, cfg.sqrt_dt = float(s.delta_t.item()), float(s.sq_delta_t.item())
        cfg.T = float('inf')                              # no freezing: every trajectory takes all N steps (solver.py:440)
        cfg.d_real = s.d
        cfg.sigma_scale = float(spec['sigma'][2])
        cfg.drift_kind = spec['drift'][0]
        cfg.h_kind = nat.GH_QUAD                          # h = -|z|^2 / 2 (problems.py:46, 211, 321 with f = 0)
        cfg.adaptive = 1 if s.adaptive_forward_process else 0
        cfg.noise_mode = nat.NOISE_PHILOX if noise == 'philox' else nat.NOISE_SUPPLIED
        cfg.store_path = 1
        cfg.domain_kind = nat.DOM_NONE
        cfg.per_sample_weights = 1
        drift_vec = spec['drift'][1]
        if drift_vec is not None:
            probe = drift_vec.detach().to(device=dev, dtype=torch.float32).contiguous()
            cfg.drift = nat.ptr(probe)
        f32 = torch.float32
        self.Kpad = 16 * ((self.K_local + 15) // 16)
        if self.deep is not None:
            try:
                from .plan_general_deep import _ACT, _IdentityPad
            except ImportError:
                from plan_general_deep import _ACT, _IdentityPad
            g, dims = self.gcfg, self.deep['dims']
            g.has_time, g.n_hidden = 1, len(dims) - 2
            for i, h in enumerate(dims[1:-1]):
                g.widths[i] = int(h)
            g.activation, g.linear_layout = _ACT[self.deep['act']], 1 if self.deep['linear'] else 0
            g.time_first, g.time_scale = 1, 1.0 / cfg.dt               # input [t, x], and t is the step index (solver.py:336-338, 439)
            if drift_vec is not None:
                self._keep.append(probe)
            sz = nat.GenlSizes()
            rc = self.lib.psp_genl_query(C.byref(g), C.byref(sz))
            if rc != 0:
                raise PlanUnsupported(self.lib.psp_last_error().decode())
            assert sz.n_params == self.P, (sz.n_params, self.P)
            self.d_pad, self.H_pad = s.d, self.H
            self.pad = _IdentityPad(self.P)
            self.flat_k = self.flat
            self.tables = torch.empty(sz.table_bytes // 4, dtype=f32, device=dev)
            self.ahat_buf = torch.zeros((sz.ahat_bytes + 3) // 4, dtype=f32, device=dev)      # coefficients, then the tiles' step counts
            self.ahat = self.ahat_buf[:(s.N + 1) * self.Kpad].view(s.N + 1, self.Kpad)
        else:
            chosen, why = shapes.gen_choose(cfg, s.d, self.H)
            if chosen is None:
                raise PlanUnsupported(why)
            self.d_pad, self.H_pad, sz = chosen
            self.pad = shapes.GenParamPad(s.d, self.H, self.d_pad, self.H_pad, dev, time_input=True, time_first=True,
                                          time_scale=1.0 / cfg.dt)
            if drift_vec is not None:
                t = self.pad.vec(drift_vec.detach().to(device=dev, dtype=torch.float32)).contiguous()
                self._keep.append(t)
                cfg.drift = nat.ptr(t)
            assert sz.n_params == self.pad.Pp, (sz.n_params, self.pad.Pp)
            self.flat_k = self.pad.new_padded_params()
            self.ahat = torch.zeros(s.N + 1, self.Kpad, dtype=f32, device=dev)       # written by the forward, then overwritten by AV
        self.sizes = sz
        self.path = torch.empty(sz.path_bytes // 4, dtype=f32, device=dev)
        self.ws = torch.zeros(s.N + 1, self.Kpad, dtype=f32, device=dev)
        self.vsteps = torch.zeros(s.N, self.Kpad, dtype=f32, device=dev)
        self.ysteps = torch.zeros(s.N, self.Kpad, dtype=f32, device=dev)
        cfg.v_steps_out, cfg.y_steps_out = nat.ptr(self.vsteps), nat.ptr(self.ysteps)
        self.cfg = cfg
        self.grad_partial = torch.empty(sz.grad_partial_bytes // 4, dtype=f32, device=dev)
        self.VN = torch.empty(self.K_local, dtype=f32, device=dev)
        self.YN = torch.empty(self.K_local, dtype=f32, device=dev)
        self.tN = torch.empty(self.K_local, dtype=f32, device=dev)
        self.XN_k = torch.empty(self.K_local, self.d_pad, dtype=f32, device=dev)
        self.D = torch.empty(self.K_local, dtype=f32, device=dev)
        self.kcount = torch.zeros(1, dtype=torch.int64, device=dev)
        self.t0 = torch.zeros(self.K_local, dtype=f32, device=dev)
        self.grad = torch.empty(self.P, dtype=f32, device=dev)
        self.grad_k = self.grad if self.deep is not None else torch.empty(self.pad.Pp, dtype=f32, device=dev)
        self.m = torch.zeros(self.P, dtype=f32, device=dev)
        self.v = torch.zeros(self.P, dtype=f32, device=dev)
        self.sums = torch.zeros(2, dtype=torch.float64, device=dev)
        self.x0_row = s.X_0.detach().to(device=dev, dtype=f32).reshape(1, -1)
        self.step = 0
        self.events = None

    def _flatten(self, V):
        params = list(V) if isinstance(V, (list, tuple)) else list(V.W)      # registration order W1,b1,W2,b2,.. (include/psp.h)
        self.params = params
        self.P = sum(p.numel() for p in params)
        flat = torch.empty(self.P, dtype=torch.float32, device=self.dev)
        off = 0
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            off += n
        self.flat = flat

    def _adam_hyper(self):
        opt = getattr(self.net, 'optim', None)
        if opt is not None and len(opt.param_groups) > 0:
            g = opt.param_groups[0]
            if g.get('weight_decay', 0) or g.get('amsgrad', False):
                raise PlanUnsupported('the native Adam implements weight_decay = 0, amsgrad = False (the reference default)')
            b = g.get('betas', (0.9, 0.999))
            return float(g['lr']), float(b[0]), float(b[1]), float(g.get('eps', 1e-8))
        return float(self.s.lr), 0.9, 0.999, 1e-8

    def iteration(self, l, loss_out, ul2_out=None):
        s, lib, cfg, dev = self.s, self.lib, self.cfg, self.dev
        st = nat.stream_ptr(dev)
        K, N = float(s.K), s.N
        lo, hi = self.lo, self.hi
        # ---- initial state and noise in the reference's order (solver.py:364-382)
        xi = None
        if self.noise == 'reference':
            X0 = torch.randn(s.K, s.d)[lo:hi].to(dev) if s.random_X_0 else self.x0_row.repeat(self.K_local, 1)
            noise = torch.randn(s.K, s.d, N + 1)
            xi = self.pad.last_dim(noise[lo:hi].permute(2, 0, 1)[1:].contiguous().to(dev))   # slot n = xi[:, :, n + 1]
        elif s.random_X_0:
            g = torch.Generator(device=dev)
            g.manual_seed(int(s.seed) * 1000003 + l)
            X0 = torch.randn(s.K, s.d, generator=g, device=dev)[lo:hi]
        else:
            X0 = self.x0_row.repeat(self.K_local, 1)
        x0 = self.pad.last_dim(X0.contiguous())
        flat_k = self.flat if self.deep is not None else self.pad.scatter_params(self.flat, self.flat_k)
        self.kcount.zero_()
        ev = None
        if self.events is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        if self.deep is not None:
            nat.check(lib.psp_genl_rollout_fwd(C.byref(self.gcfg), nat.ptr(self.flat), nat.ptr(x0), nat.ptr(self.t0), nat.ptr(xi),
                                               int(s.seed) & 0xFFFFFFFFFFFFFFFF, l, nat.ptr(self.tables), nat.ptr(self.path),
                                               nat.ptr(self.ahat_buf), nat.ptr(self.VN), nat.ptr(self.YN), nat.ptr(self.XN_k),
                                               nat.ptr(self.tN), nat.ptr(self.kcount), st), 'psp_genl_rollout_fwd')
        else:
            nat.check(lib.psp_gen_rollout_fwd(C.byref(cfg), nat.ptr(flat_k), nat.ptr(x0), nat.ptr(self.t0), nat.ptr(xi),
                                              int(s.seed) & 0xFFFFFFFFFFFFFFFF, l, nat.ptr(self.path), nat.ptr(self.ahat),
                                              nat.ptr(self.VN), nat.ptr(self.YN), nat.ptr(self.XN_k), nat.ptr(self.tN),
                                              nat.ptr(self.kcount), st), 'psp_gen_rollout_fwd')
        if ev is not None:
            ev[1].record()
        # ---- loss (solver.py:164-168, 499) and the per-sample weights
        XN = self.XN_k[:, :s.d]
        torch.sub(self.YN, s.problem.g(XN).to(torch.float32), out=self.D)
        Dd = self.D.double()
        r = (self.vsteps - self.ysteps)[:, :self.K_local]
        r[0].zero_()                                              # the term starts at n = 1 (solver.py:438)
        stats = torch.stack([Dd.sum(), (Dd * Dd).sum(), (r.double() ** 2).sum()])
        sharding.allreduce_sum_(stats)                            # collective 1
        self.sums.copy_(stats[:2])
        loss = sharding.loss_from_sums(self.sums, s.K, s.loss_method) + stats[2] / K
        loss_out[l] = loss.to(torch.float32)
        wD = sharding.loss_weights(self.D, self.sums, s.K, s.loss_method)            # (K_local)
        tail = torch.flip(torch.cumsum(torch.flip(r, [0]), 0), [0])                 # tail[m] = sum_{n >= m} r_n
        ws, av = self.ws, self.ahat
        ws.zero_()
        av.zero_()
        # WS[m] = w^D - (2/K) sum_{n > m} r_n  (m = 0..N-1);  AV[0] = WS[0] + ... = w^D - (2/K) sum_{n >= 1} r_n
        ws[:N - 1, :self.K_local] = wD.unsqueeze(0) - (2.0 / K) * tail[1:]
        ws[N - 1, :self.K_local] = wD
        av[1:N, :self.K_local] = (2.0 / K) * r[1:]
        av[0, :self.K_local] = ws[0, :self.K_local]
        if ev is not None:
            ev[2].record()
        if self.deep is not None:
            nat.check(lib.psp_genl_rollout_bwd(C.byref(self.gcfg), nat.ptr(self.flat), nat.ptr(self.tables), nat.ptr(self.path),
                                               nat.ptr(self.ahat_buf), nat.ptr(ws), None, nat.ptr(self.grad_partial),
                                               nat.ptr(self.grad), st), 'psp_genl_rollout_bwd')
        else:
            nat.check(lib.psp_gen_rollout_bwd(C.byref(cfg), nat.ptr(flat_k), nat.ptr(self.path), nat.ptr(av), nat.ptr(ws), None,
                                              nat.ptr(self.grad_partial), nat.ptr(self.grad_k), st), 'psp_gen_rollout_bwd')
            self.pad.gather_grad(self.grad_k, self.grad)
        if ev is not None:
            ev[3].record()
            self.events.append(ev)
        sharding.allreduce_sum_(self.grad)                        # collective 2
        self.step += 1
        lr, b1, b2, eps = self._adam_hyper()
        nat.check(lib.psp_adam_step(nat.ptr(self.flat), nat.ptr(self.grad), nat.ptr(self.m), nat.ptr(self.v),
                                    self.P, self.step, lr, b1, b2, eps, st), 'psp_adam_step')
        return loss
