"""Evaluation utilities -- API mirror of the hot-path-adjacent part of the reference's utilities.py.

``do_importance_sampling_me`` (reference utilities.py:287-359) estimates E[exp(-int f - g(X_T))] under the
learned control with the Girsanov weight and reports mean, variance and relative error.  It is the
same controlled Euler-Maruyama rollout as the training step, forward only, at ``K`` up to 1e7
(SURVEY.md 8f rank 1), and is called from Solver.train every ``IS_variance_iter`` iterations when
``IS_variance_K > 0`` (solver.py:521-528).

Native plan: psp_hjb_rollout_eval (include/psp.h) on the solver's control net whenever the solver is
native-eligible; composite torch plan otherwise.  Plotting helpers of the reference are out of scope.
"""
import ctypes as C

import numpy as np
import torch

try:
    from . import native as nat
    from . import native_shapes as shapes
except ImportError:
    import native as nat
    import native_shapes as shapes


def _time_feature_table(model, N, delta_t):
    """Network time input the reference uses at IS step n: Z_n(X, n*delta_t) -> index ceil(t/model.delta_t)
    in fp32 (solver.py:360-362), then ones * index * model.delta_t (solver.py:355)."""
    dt32 = model.delta_t.detach().cpu()
    vals = []
    for n in range(N):
        t = torch.as_tensor(n * delta_t, dtype=torch.float32)
        idx = int(torch.ceil(t / dt32))
        vals.append(float((torch.ones(1) * idx * dt32).item()))
    return torch.tensor(vals, dtype=torch.float32)


def _native_reason(problem, model, control, simulate_naive):
    try:
        from .plan_native import native_eligibility
    except ImportError:
        from plan_native import native_eligibility
    if control != 'approx' and model.u_l2_error_flag:
        return "control='true' evaluates problem.u_true on the host"
    if simulate_naive:
        return 'simulate_naive needs the uncontrolled process too'
    if getattr(model, 'backend', 'auto') == 'torch':
        return "backend='torch' requested"
    saved = (model.IS_variance_K, model.u_l2_error_flag)
    model.IS_variance_K, model.u_l2_error_flag = 0, False          # these two do not matter for the evaluation itself
    try:
        return native_eligibility(model)
    finally:
        model.IS_variance_K, model.u_l2_error_flag = saved


def _stats(logw):
    w = torch.exp(logw)
    mean_IS = torch.mean(w).item()
    variance_IS = torch.var(w).item()
    return mean_IS, variance_IS, float(np.sqrt(variance_IS) / mean_IS)


def _is_native(problem, model, K, delta_t):
    dev = model.device
    lib = nat.load()
    N = int(np.ceil(problem.T / delta_t))
    spec = problem.native_spec()
    keep = []

    def dev_f32(t):
        t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
        keep.append(t)
        return t

    H = model.z_n.native_shape()[1]
    cfg = nat.HjbConfig()
    cfg.K_local, cfg.N = K, N
    cfg.K_global, cfg.k_offset = K, 0
    cfg.dt = float(torch.tensor(delta_t, dtype=torch.float32).item())
    cfg.sqrt_dt = float(torch.tensor(np.sqrt(delta_t), dtype=torch.float32).item())
    cfg.drift_kind = spec['drift'][0]
    cfg.sigma_kind = spec['sigma'][0]
    cfg.sigma_scale = float(spec['sigma'][2])
    cfg.runcost_kind = spec['runcost'][0]
    cfg.term_kind = spec['term'][0]
    cfg.adaptive, cfg.loss_kind, cfg.store_path = 1, nat.LOSS_LOG_VARIANCE, 0
    philox = getattr(model, 'noise', 'reference') == 'philox'
    cfg.noise_mode = nat.NOISE_PHILOX if philox else nat.NOISE_SUPPLIED
    chosen, why = shapes.choose(cfg, model.d, H)       # exact instance or the cheapest larger one (zero padding)
    if chosen is None:
        raise NotImplementedError('native IS evaluation unavailable: ' + why)
    d_pad, H_pad, _, sizes = chosen
    pad = shapes.ParamPad(model.d, H, d_pad, H_pad, dev)
    cfg.drift = nat.ptr(dev_f32(pad.drift_or_sigma(spec['drift'][1]))) if spec['drift'][1] is not None else None
    cfg.sigma = nat.ptr(dev_f32(pad.drift_or_sigma(spec['sigma'][1]))) if spec['sigma'][1] is not None else None
    cfg.runcost = nat.ptr(dev_f32(pad.vec(spec['runcost'][1]))) if spec['runcost'][1] is not None else None
    cfg.term = nat.ptr(dev_f32(pad.vec(spec['term'][1])))
    flat = torch.cat([p.detach().reshape(-1) for p in model.z_n.flat_layout()]).to(dev).contiguous()
    flat = pad.scatter_params(flat, pad.new_padded_params() if not pad.identity else None)
    xi = None
    if not philox:                                   # the reference's draws: N x randn(K, d) (utilities.py:310)
        xi_cpu = torch.zeros(N + 1, K, model.d)
        for n in range(N):
            xi_cpu[n + 1] = torch.randn(K, model.d)
        xi = pad.last_dim(xi_cpu.to(dev))
    tfeat = _time_feature_table(model, N, delta_t).to(dev)
    x0 = dev_f32(pad.vec(torch.as_tensor(problem.X_0, dtype=torch.float32).to(dev)))
    D = torch.empty(K, dtype=torch.float32, device=dev)
    Fint = torch.empty(K, dtype=torch.float32, device=dev)
    part = torch.empty(sizes.fwd_partial_bytes // 8, dtype=torch.float64, device=dev)
    model._is_calls = getattr(model, '_is_calls', 0) + 1
    nat.check(lib.psp_hjb_rollout_eval(C.byref(cfg), nat.ptr(flat), nat.ptr(x0), 0, nat.ptr(xi),
                                       (int(model.seed) + 7919) & 0xFFFFFFFFFFFFFFFF, model._is_calls,
                                       nat.ptr(tfeat), nat.ptr(D), nat.ptr(Fint), None, nat.ptr(part),
                                       nat.stream_ptr(dev)), 'psp_hjb_rollout_eval')
    return _stats(D - 2.0 * Fint)


def _dense_reason(problem, model):
    """None if the learned control is a DenseNet configuration the hjbd forward kernel covers (plan_dense_native.py)."""
    try:
        from .plan_dense_native import dense_eligibility
    except ImportError:
        from plan_dense_native import dense_eligibility
    if getattr(model, 'backend', 'auto') == 'torch':
        return "backend='torch' requested"
    saved = (model.IS_variance_K, model.u_l2_error_flag, model.loss_method, model.detach_forward)
    model.IS_variance_K, model.u_l2_error_flag, model.loss_method, model.detach_forward = 0, False, 'log-variance', True
    try:                                              # (training-only restrictions do not matter for a forward sweep)
        return dense_eligibility(model)
    finally:
        model.IS_variance_K, model.u_l2_error_flag, model.loss_method, model.detach_forward = saved


def _is_dense_native(problem, model, K, delta_t):
    """The controlled forward sweep of utilities.py:296-330 on the DenseNet-control rollout kernel.  The reference
    evaluates Z_n(X, n delta_t) through solver.py:360-362: step index ceil(t / model.delta_t) -> the time feature
    (inner) or the per-step net (outer); here that index selects the time-feature entry or the parameter set."""
    try:
        from .plan_dense_native import _instance_for, _nets
    except ImportError:
        from plan_dense_native import _instance_for, _nets
    dev = model.device
    lib = nat.load()
    N = int(np.ceil(problem.T / delta_t))
    spec = problem.native_spec()
    keep = []

    def dev_f32(t):
        t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
        keep.append(t)
        return t

    nets = _nets(model)
    outer = model.time_approx == 'outer'
    H = nets[0].nn_dims[1]
    d_pad, H_pad = _instance_for(model.d, H)
    pad = shapes.ParamPad(model.d, H, d_pad, H_pad, dev)
    cfg = nat.DnetConfig()
    b = cfg.base
    b.d, b.H, b.K_local, b.N, b.K_global, b.k_offset = d_pad, H_pad, K, N, K, 0
    b.dt = float(torch.tensor(delta_t, dtype=torch.float32).item())
    b.sqrt_dt = float(torch.tensor(np.sqrt(delta_t), dtype=torch.float32).item())
    b.drift_kind, b.sigma_kind, b.sigma_scale = spec['drift'][0], spec['sigma'][0], float(spec['sigma'][2])
    b.runcost_kind, b.term_kind = spec['runcost'][0], spec['term'][0]
    b.adaptive, b.loss_kind, b.store_path = 1, nat.LOSS_LOG_VARIANCE, 0
    philox = getattr(model, 'noise', 'reference') == 'philox'
    b.noise_mode = nat.NOISE_PHILOX if philox else nat.NOISE_SUPPLIED
    b.drift = nat.ptr(dev_f32(pad.drift_or_sigma(spec['drift'][1]))) if spec['drift'][1] is not None else None
    b.sigma = nat.ptr(dev_f32(pad.drift_or_sigma(spec['sigma'][1]))) if spec['sigma'][1] is not None else None
    b.runcost = nat.ptr(dev_f32(pad.vec(spec['runcost'][1]))) if spec['runcost'][1] is not None else None
    b.term = nat.ptr(dev_f32(pad.vec(spec['term'][1])))
    cfg.d_real, cfg.H_real = model.d, H
    cfg.time_input, cfg.per_step = (0 if outer else 1), (1 if outer else 0)
    sets = [torch.cat([p.detach().reshape(-1) for p in net.W]).to(dev) for net in nets]
    tfeat = _time_feature_table(model, N, delta_t)
    if outer:                                        # one parameter set per EVALUATION step: the net solver.py:192-193 picks
        dt32 = float(model.delta_t.detach().cpu())
        idx = [max(0, min(int(round(float(t) / dt32)), model.N - 1)) for t in tfeat.tolist()]
        flat = torch.cat([sets[i] for i in idx]).contiguous()
    else:
        flat = sets[0].contiguous()
    sizes = nat.DnetSizes()
    nat.check(lib.psp_dnet_query(C.byref(cfg), C.byref(sizes)), 'psp_dnet_query')
    xi = None
    if not philox:                                   # the reference's draws: N x randn(K, d) (utilities.py:310)
        xi_cpu = torch.zeros(N + 1, K, model.d)
        for n in range(N):
            xi_cpu[n + 1] = torch.randn(K, model.d)
        xi = pad.last_dim(xi_cpu.to(dev))
    tfeat = tfeat.to(dev)
    x0 = dev_f32(pad.vec(torch.as_tensor(problem.X_0, dtype=torch.float32).to(dev)))
    D = torch.empty(K, dtype=torch.float32, device=dev)
    Fint = torch.empty(K, dtype=torch.float32, device=dev)
    part = torch.empty(sizes.fwd_partial_bytes // 8, dtype=torch.float64, device=dev)
    tables = torch.empty(sizes.table_bytes // 4, dtype=torch.float32, device=dev)
    model._is_calls = getattr(model, '_is_calls', 0) + 1
    nat.check(lib.psp_dnet_rollout_fwd(C.byref(cfg), nat.ptr(flat), nat.ptr(x0), 0, None, nat.ptr(xi),
                                       (int(model.seed) + 7919) & 0xFFFFFFFFFFFFFFFF, model._is_calls, nat.ptr(tfeat),
                                       None, None, nat.ptr(D), nat.ptr(Fint), None, None, nat.ptr(part),
                                       nat.ptr(tables), nat.stream_ptr(dev)), 'psp_dnet_rollout_fwd')
    return _stats(D - 2.0 * Fint)


def _is_composite(problem, model, K, delta_t):
    dev = model.device
    sq_dt = np.sqrt(delta_t)
    N = int(np.ceil(problem.T / delta_t))
    X_u = torch.as_tensor(problem.X_0, dtype=torch.float32).repeat(K, 1).to(dev)
    ito = torch.zeros(K).to(dev)
    riemann = torch.zeros(K).to(dev)
    f_int_u = torch.zeros(K).to(dev)
    for n in range(N):
        xi = torch.randn(K, problem.d).to(dev)
        with torch.no_grad():
            ut = -model.Z_n(X_u, n * delta_t)
        sig = problem.sigma(X_u)
        X_u = (X_u + (problem.b(X_u) + torch.mm(sig, ut.t()).t()) * delta_t + torch.mm(sig, xi.t()).t() * sq_dt)
        ito = ito + torch.sum(ut * xi, 1) * sq_dt
        riemann = riemann + torch.sum(ut ** 2, 1) * delta_t
        f_int_u = f_int_u + model.f(X_u, n * delta_t) * delta_t
    return _stats(-f_int_u - problem.g(X_u) - ito - 0.5 * riemann)


def do_importance_sampling_me(problem, model, K, control='approx', simulate_naive=False, verbose=False,
                              delta_t=0.01, on_cpu=False, cross_statistics=None):
    """Returns (mean_IS, variance_IS, rel_error_IS) -- reference utilities.py:287-359 for control='approx'.
    ``simulate_naive``, ``control='true'`` with a reference solution, ``on_cpu`` and ``cross_statistics``
    are not built."""
    if simulate_naive or on_cpu or cross_statistics is not None or (control != 'approx' and model.u_l2_error_flag):
        raise NotImplementedError('only the controlled estimator with the learned control is built')
    reason = _native_reason(problem, model, control, simulate_naive)
    if reason is None:
        out = _is_native(problem, model, K, delta_t)
    elif _dense_reason(problem, model) is None:
        out = _is_dense_native(problem, model, K, delta_t)
    else:
        if getattr(model, 'backend', 'auto') == 'native':
            raise NotImplementedError('native IS evaluation unavailable: ' + reason)
        out = _is_composite(problem, model, K, delta_t)
    if verbose:
        print('IS mean: %.4e, IS variance: %.4e, IS RE %.4e' % out)
    return out


def compute_test_error(model, problem, K, device=None, modus='elliptic'):
    """Monte-Carlo error of the value net against ``problem.v_true`` on K fresh points of the domain -- the ``K_test_log``
    diagnostic of EllipticSolver / GeneralSolver (reference utilities.py:440-472; draws from the CPU generator in that order:
    randn(K, d) and rand(K) for the ball-shaped domains, rand(K, d) for the boxes, then rand(K, 1) for the times).
    Returns (L2 error, mean absolute error, mean relative error)."""
    device = torch.device(device) if device is not None else model.device
    d = problem.d
    if problem.boundary in ('sphere', 'unbounded', 'two_spheres'):
        R = problem.boundary_distance_2 if problem.boundary == 'two_spheres' else problem.boundary_distance
        X = torch.randn(K, d).to(device)
        X = R * X / torch.sqrt(torch.sum(X ** 2, 1)).unsqueeze(1) * (torch.rand(K).unsqueeze(1) ** (1 / d)).to(device)
        if problem.boundary == 'two_spheres':
            X = X[torch.sqrt(torch.sum(X ** 2, 1)) > problem.boundary_distance_1, :]
    else:
        X = (problem.X_r - problem.X_l) * torch.rand(K, d).to(device) + problem.X_l
    with torch.no_grad():
        if modus == 'parabolic':
            t_n = torch.rand(K, 1).to(device) * problem.T
            v_true = np.asarray(torch.as_tensor(problem.v_true(X.detach().cpu(), t_n.cpu().squeeze())).squeeze())
            v_est = model.V(torch.cat([X, t_n], 1)).squeeze().cpu().numpy()
        else:
            v_true = np.asarray(torch.as_tensor(problem.v_true(X.detach().cpu())).squeeze())
            v_est = model.V(X).squeeze().cpu().numpy()
    diff = v_true - v_est
    return float(np.mean(diff ** 2)), float(np.mean(np.abs(diff))), float(np.mean(np.abs(diff) / v_true))
