"""Native (HIP) execution plan for GeneralSolver.train and EllipticSolver.train (diffusion / BSDE loss on unbounded,
sphere, annulus ('two_spheres'), box and cut-corner ('square-corner') domains).

Per iteration (reference solver.py:1009-1201):
    host RNG in the reference's order (domain sample, t ~ U(0,T), per-step xi)      [noise='reference']
    psp_gen_rollout_fwd   -> V(X_N,t_N), Y_N, X_N, t_N, active-step count, path store
    per-trajectory loss weights (K-vectors) and the small terminal-condition term in torch
    psp_gen_rollout_bwd   -> flat gradient of the domain part of the loss
    [all-reduce gradient] -> psp_adam_step
The terminal term a1 mean((V(X[:Kb],T) - f(X[:Kb]))^2), the Dirichlet / Neumann residual on the boundary batch and the
`sample_center` probe involve K_boundary (~50) points and are differentiated by torch autograd on the same parameters (views
of the flat buffer); they are O(K_boundary) work against O(K N) in the kernels.  So is the Neumann residual of the BSDE loss
(solver.py:1177-1183): grad_x V at the K sample points of the LAST executed step, read back from the path store.

Bounded domains (reference solver.py:1119-1129, :750-767): the exit test runs inside the forward kernel
(psp_gen_config.domain_kind).  EllipticSolver (V = DenseNet(d -> 1), no time input) runs through the same kernels
with T = +inf and a parameter index map that leaves the kernels' time row zero (native_shapes.GenParamPad).
'two_spheres' (the committor problem) draws its batch by rejection, so the batch size K changes from iteration to iteration
(solver.py:1048-1052, :706-710): every buffer is sized for K_original and each launch takes that iteration's K.
With noise='reference' the host must consume exactly as many randn(K,d) draws as the reference loop executes before
its all-stopped break; on a bounded domain that count depends on the paths, so it is read back after the forward kernel
and the CPU generator is rewound and advanced by that many draws.
"""
import ctypes as C

import torch

try:
    from . import native as nat
    from . import native_shapes as shapes
    from . import sharding
    from .function_space import DenseNet
    from . import general_solver as gs
except ImportError:
    import native as nat
    import native_shapes as shapes
    import sharding
    from function_space import DenseNet
    import general_solver as gs


def native_eligibility(solver, deep=False):
    """None if the (d, H)-templated value-net kernels take this solver, else a reason.  deep=True: the checks that do not
    concern the net's shape (plan_general_deep.py has its own shape test)."""
    if solver.device.type != 'cuda':
        return 'device is %s (the HIP rollout needs a GPU)' % solver.device
    if solver.approx_method != 'Y' or solver.loss_method not in ('diffusion', 'BSDE'):
        return "only approx_method='Y' with loss_method 'diffusion' / 'BSDE' is native"
    if solver.adaptive_forward_process and not solver.detach_forward:
        return 'detach_forward=False back-propagates through the state path (not native)'
    if not solver.boundary_loss and solver.loss_method == 'diffusion':
        return 'boundary_loss=False is not native'
    V = solver.V
    dims = getattr(V, 'nn_dims', None)
    d_in = solver.d + (0 if solver.elliptic else 1)
    if not deep and (not isinstance(V, DenseNet) or dims is None or len(dims) != 4 or dims[1] != dims[2] or dims[3] != 1
                     or dims[0] != d_in or getattr(V, 'activation', 'relu2') != 'relu2'):
        return 'V is not a DenseNet(%d -> 1) with two equal hidden widths and relu^2' % d_in
    spec_fn = getattr(solver.problem, 'general_native_spec', None)
    if spec_fn is None:
        return 'problem has no general_native_spec() (coefficients outside the native catalogue)'
    try:
        from .problems import coefficients_overridden
    except ImportError:
        from problems import coefficients_overridden
    over = coefficients_overridden(solver.problem)
    if over is not None:
        return 'problem.%s is not the catalogue implementation general_native_spec() describes' % over
    if not nat.is_built():
        raise nat.NativeLibraryError('libpsp_hip.so is not built; run __graft_entry__.build()')
    if not deep and not shapes.gen_candidates(solver.d, dims[1]):
        return 'no compiled kernel instance covers d=%d, H=%d (see csrc/gen_instances.def)' % (solver.d, dims[1])
    return None


def set_domain(cfg, pb, elliptic):
    """psp_gen_config.domain_kind / dom_a / dom_b from the problem's ``boundary`` (include/psp.h PSP_DOM_*)."""
    if pb.boundary == 'sphere':
        cfg.domain_kind, cfg.dom_a = nat.DOM_SPHERE, float(pb.boundary_distance)
    elif pb.boundary == 'two_spheres':                            # solver.py:1122-1123 / :752-753
        cfg.domain_kind, cfg.dom_a, cfg.dom_b = nat.DOM_ANNULUS, float(pb.boundary_distance_1), float(pb.boundary_distance_2)
    elif pb.boundary == 'square':
        cfg.dom_a, cfg.dom_b = float(pb.X_l), float(pb.X_r)
        cfg.domain_kind = nat.DOM_BOX if not pb.one_boundary else \
            (nat.DOM_BOX_UPPER_ALL if elliptic else nat.DOM_BOX_UPPER_ANY)
    elif pb.boundary == 'square-corner':                          # solver.py:759-760: any(X_proposal <= X_r)
        cfg.domain_kind, cfg.dom_a, cfg.dom_b = nat.DOM_BOX_UPPER_ANY, float(pb.X_l), float(pb.X_r)


class GeneralNativePlan:
    def __init__(self, solver):
        s = solver
        self.s = s
        self.lib = nat.load()
        self.dev = s.device
        self._shard(s.K_original)                # capacity: the largest batch an iteration can see
        self.K_cap = self.K_local
        self.net = s.V                           # the plan is rebuilt when the caller swaps model.V (general_solver._choose_plan)
        self.key = None
        self.H = s.V.nn_dims[1]
        self._flatten(list(s.V.W))               # registration order W1,b1,W2,b2,W3,b3 (include/psp.h)
        spec = s.problem.general_native_spec()
        self._keep = []
        cfg = nat.GenConfig()
        cfg.K_local, cfg.N = self.K_local, s.N
        cfg.k_offset = self.lo
        cfg.dt, cfg.sqrt_dt = float(s.delta_t.item()), float(s.sq_delta_t.item())
        self.elliptic = bool(s.elliptic)
        cfg.T = float('inf') if self.elliptic else float(torch.tensor(s.problem.T, dtype=torch.float32).item())
        pb = s.problem
        set_domain(cfg, pb, self.elliptic)
        cfg.d_real = s.d
        # 'f16x3': fp32-grade split products on the f16 matrix pipe in the forward rollout (csrc/hjb_kernels.h gemm_Tx; same parity
        # bounds as 'fp32'); 'auto' (the default): 'f16x3' where the instance has it and its tables fit the LDS (decided below)
        self._mlp_want = getattr(s, 'mlp_dtype', 'auto')
        cfg.mlp_dtype = {'auto': nat.MLP_FP32, 'fp32': nat.MLP_FP32, 'f16x3': nat.MLP_F16X3, 'bf16_fwd': nat.MLP_BF16_FWD,
                         'bf16': nat.MLP_BF16}[self._mlp_want]
        for i, v in enumerate(spec.get('h_par', ())):
            cfg.h_par[i] = float(v)
        cfg.sigma_scale = float(spec['sigma_scale'])
        cfg.drift_kind = spec['drift'][0]
        cfg.h_kind = spec['h']
        cfg.adaptive = 1 if s.adaptive_forward_process else 0
        cfg.noise_mode = nat.NOISE_PHILOX if s.noise == 'philox' else nat.NOISE_SUPPLIED
        cfg.store_path = 1
        # kernel instance: the exact (d, H) if compiled, else the cheapest larger one (zero padding, native_shapes.py)
        if spec['drift'][1] is not None:              # psp_gen_query validates the pointer: any device tensor will do for
            t_probe = spec['drift'][1].detach().to(device=self.dev, dtype=torch.float32).contiguous()   # the size query
            cfg.drift = nat.ptr(t_probe)
        chosen, why = shapes.gen_choose(cfg, s.d, self.H)
        if chosen is None:
            raise NotImplementedError('native plan unavailable: ' + why)
        self.d_pad, self.H_pad, sz = chosen
        if self._mlp_want == 'auto':
            cfg.mlp_dtype = nat.MLP_F16X3
            rc, sz_x3, _ = nat.gen_query_rc(cfg)
            if rc == 0:
                sz = sz_x3
            else:
                cfg.mlp_dtype = nat.MLP_FP32
        self.matrix_mode = {nat.MLP_FP32: 'fp32', nat.MLP_F16X3: 'f16x3', nat.MLP_BF16_FWD: 'bf16_fwd', nat.MLP_BF16: 'bf16'}[cfg.mlp_dtype]
        # range guard of the split-product mode (include/psp.h: psp_gen_config.range_flag): a non-finite V(X_N) / Y_N raises a
        # device flag and the fp32-MFMA kernels, enqueued behind the split ones and predicated on it, redo the iteration
        self.range_flag = None
        if cfg.mlp_dtype == nat.MLP_F16X3 and getattr(s, 'range_guard', True):
            self.range_flag = torch.zeros(4, dtype=torch.int32, device=self.dev)
            cfg.range_flag = nat.ptr(self.range_flag)
        self.pad = shapes.GenParamPad(s.d, self.H, self.d_pad, self.H_pad, self.dev, time_input=not self.elliptic)
        if spec['drift'][1] is not None:
            t = self.pad.vec(spec['drift'][1].detach().to(device=self.dev, dtype=torch.float32)).contiguous()
            self._keep.append(t)
            cfg.drift = nat.ptr(t)
        self.cfg = cfg
        assert sz.n_params == self.pad.Pp, (sz.n_params, self.pad.Pp)
        self.sizes = sz
        self.flat_k = self.flat if self.pad.identity else self.pad.new_padded_params()
        dev, f32 = self.dev, torch.float32
        self.path = torch.empty(sz.path_bytes // 4, dtype=f32, device=dev)
        self.ahat = torch.zeros(sz.ahat_bytes // 4, dtype=f32, device=dev)
        self.grad_partial = torch.empty(sz.grad_partial_bytes // 4, dtype=f32, device=dev)
        self.grad_k = self.grad if self.pad.identity else torch.empty(self.pad.Pp, dtype=f32, device=dev)
        self._common_buffers(self.d_pad)

    # ---- pieces shared with plan_general_deep.GeneralDeepPlan ---------------------------------------------------------
    def _shard(self, K):
        """This rank's contiguous block of a batch of K trajectories."""
        self.dist, self.rank, self.world = sharding.dist_info()
        lo, hi = sharding.shard_bounds_ragged(K, self.rank, self.world)
        self.lo, self.hi, self.K_local = lo, hi, hi - lo

    def _common_buffers(self, d_store):
        dev, f32, cap = self.dev, torch.float32, self.K_cap
        self._VN = torch.empty(cap, dtype=f32, device=dev)
        self._YN = torch.empty(cap, dtype=f32, device=dev)
        self._tN = torch.zeros(cap, dtype=f32, device=dev)
        self._XN = torch.empty(cap * d_store, dtype=f32, device=dev)
        self.d_store = d_store
        self.kcount = torch.zeros(1, dtype=torch.int64, device=dev)
        self.grad = getattr(self, 'grad', None)
        if self.grad is None:
            self.grad = torch.empty(self.P, dtype=f32, device=dev)
        self.m = torch.zeros(self.P, dtype=f32, device=dev)
        self.v = torch.zeros(self.P, dtype=f32, device=dev)
        cap_pad = 16 * ((cap + 15) // 16)
        self.wY = torch.zeros(cap_pad, dtype=f32, device=dev)     # zero-padded (include/psp.h)
        self.wV = torch.zeros(cap_pad, dtype=f32, device=dev)
        self.step = 0
        self.last_v_l2 = None
        self._extra_grad = None
        self.events = None   # bench.py: HIP-event pairs around the two rollout kernels
        self._batch_views()

    def _batch_views(self):
        """Output views for this iteration's batch (K_local <= K_cap)."""
        Kl = self.K_local
        self.VN, self.YN, self.tN = self._VN[:Kl], self._YN[:Kl], self._tN[:Kl]
        self.XN_k = self._XN[:Kl * self.d_store].view(Kl, self.d_store)
        self.XN = self.XN_k[:, :self.s.d]
        self.Kpad = 16 * ((Kl + 15) // 16)

    def _set_batch(self, K):
        """A new batch size (the 'two_spheres' rejection step): shard it, point the kernels' config at it."""
        if K == getattr(self, '_K_now', None):
            return
        self._K_now = K
        self._shard(K)
        if self.K_local > self.K_cap or self.K_local <= 0:
            raise RuntimeError('batch of %d trajectories outside the plan capacity %d' % (self.K_local, self.K_cap))
        self.cfg.K_local, self.cfg.k_offset = self.K_local, self.lo
        self._batch_views()
        self.wY.zero_()
        self.wV.zero_()
        self._check_sizes()

    def _check_sizes(self):
        sz = nat.gen_query(self.cfg)
        assert sz.path_bytes <= self.path.numel() * 4 and sz.ahat_bytes <= self.ahat.numel() * 4
        if sz.grad_partial_bytes > self.grad_partial.numel() * 4:
            self.grad_partial = torch.empty(sz.grad_partial_bytes // 4, dtype=torch.float32, device=self.dev)
        self.sizes = sz

    def _flatten(self, params):
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
        self.grad = torch.empty(self.P, dtype=torch.float32, device=self.dev)

    def _sample_domain_device(self, l):
        """Domain sample of solver.py:1040-1056 / :695-708 drawn with a device generator (noise='philox')."""
        s, dev, pb = self.s, self.dev, self.s.problem
        if not hasattr(self, '_gen'):
            self._gen = torch.Generator(device=dev)
        self._gen.manual_seed(int(s.seed) * 1000003 + l)
        K, d, g = s.K, s.d, self._gen
        if pb.boundary in ('unbounded', 'sphere'):
            if s.uniform_square:                                 # solver.py:1042-1043
                X = torch.rand(K, d, generator=g, device=dev) * 2 - 1
                radial = torch.rand(K, generator=g, device=dev).unsqueeze(1)
            else:
                X = torch.randn(K, d, generator=g, device=dev)
                radial = torch.rand(K, generator=g, device=dev).unsqueeze(1) ** (1 / d)
            return pb.boundary_distance * X / torch.sqrt(torch.sum(X ** 2, 1)).unsqueeze(1) * radial
        if pb.boundary == 'two_spheres':
            if self.elliptic and s.uniform_square:               # solver.py:702-704
                X = torch.rand(K, d, generator=g, device=dev) * 2 - 1
                return X / torch.sqrt(torch.sum(X ** 2, 1)).unsqueeze(1) * (
                    torch.rand(K, d, generator=g, device=dev) * (pb.boundary_distance_2 - pb.boundary_distance_1) + pb.boundary_distance_1)
            Ko = s.K_original                                     # rejection from the outer ball (solver.py:1048-1052 == :706-710)
            X = torch.randn(Ko, d, generator=g, device=dev)
            X = pb.boundary_distance_2 * X / torch.sqrt(torch.sum(X ** 2, 1)).unsqueeze(1) * (
                torch.rand(Ko, generator=g, device=dev).unsqueeze(1) ** (1 / d))
            keep = torch.sqrt(torch.sum(X ** 2, 1)) > pb.boundary_distance_1
            s.K = int(torch.sum(keep))                            # (one host read: the launch geometry follows the batch size)
            return X[keep, :]
        X = (pb.X_r - pb.X_l) * torch.rand(K, d, generator=g, device=dev) + pb.X_l
        if pb.boundary == 'square-corner':                        # solver.py:708
            corner = torch.all(X > pb.X_corner, 1)
            X[corner, :] = -X[corner, :]
        return X

    def _executed_steps(self, t0_cpu):
        """Steps the reference executes before its early break (solver.py:1093-1097): it stops drawing
        xi once every trajectory is frozen.  Emulated on the host in fp32 from t0."""
        s = self.s
        dt = torch.tensor(s.delta_t_np)
        t = t0_cpu.clone().squeeze()
        stopped = torch.zeros_like(t, dtype=torch.bool)
        for n in range(s.N):
            if int((~stopped).sum()) == 0:
                return n
            in_time = (t + dt) <= s.problem.T
            act = in_time & ~stopped
            t = t + dt * act.float()
            stopped = stopped | ~in_time
        return s.N

    def _active_steps(self, t0):
        """Active steps per trajectory: round((t_N - t_0)/dt) (EllipticSolver: the kernels write t_N = m dt with one rounding)."""
        return torch.round((self.tN - t0) / self.cfg.dt)

    def _loop_steps(self, t0):
        """Steps of the time loop the reference executes before every trajectory has stopped (solver.py:1093-1097): a trajectory
        with m active steps is found stopped by the step after, so the loop runs max m + 1 steps, at most N."""
        m = self._active_steps(t0).max().reshape(1)
        dist, _, world = sharding.dist_info()
        if world > 1:
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
        return min(self.s.N, int(m.item()) + 1)

    def _draws_executed(self, t0):
        """randn(K,d) draws the reference loop makes on a bounded domain: it leaves at the first step that finds every
        trajectory stopped (solver.py:1093-1097); EllipticSolver draws xi before that test (:739-744)."""
        return min(self.s.N, self._loop_steps(t0) + (1 if self.elliptic else 0))

    def _boundary_terms(self, X, X_b, t_b):
        """The K_boundary-sized loss terms, differentiated by autograd into p.grad (solver.py:1015-1017, :1062-1074, :643-645,
        :683-693).  (Round 3 measured them behind the rollout launch and on a side stream beside it -- same-box A/B, d=100 K=65536
        and the K=200 notebook shape: both slower than here in front of the rollout, 8.65 vs 8.75 / 8.9 ms and 1.2 vs 1.33 / 1.5 ms.)"""
        s, dev = self.s, self.dev
        loss_b = None
        if s.sample_center:                                      # a one-dimensional probe point, as written there
            X_center = torch.zeros(1, 1, device=dev)
            loss_b = torch.mean((s.V(X_center).squeeze() - s.problem.v_true(X_center).squeeze()) ** 2)
        if s.loss_method != 'BSDE' and s.boundary_loss:
            if self.elliptic:
                term = s.alpha[1] * s.boundary_residual(X_b)
            else:
                Kb, T = s.K_boundary, s.problem.T
                X_T = torch.cat([X[:Kb, :], T * torch.ones(Kb, device=dev).unsqueeze(1)], 1)
                term = s.alpha[1] * torch.mean((s.V(X_T).squeeze() - s.problem.f(X[:Kb, :])) ** 2)
                if s.bounded:
                    term = term + s.alpha[2] * s.boundary_residual(torch.cat([X_b, t_b], 1), X_b, t_b)
            loss_b = term if loss_b is None else loss_b + term
        if loss_b is not None:
            loss_b.backward()                                    # K_boundary points only
        return loss_b

    def _mean_sq_masked(self, r, mask, K_global):
        """(sum over the masked entries of r^2 / count, dLoss/dr) with the count taken over ALL ranks; zero when the mask is empty."""
        cnt = mask.sum().double().reshape(1)
        sq = torch.sum(torch.where(mask, r, torch.zeros_like(r)).double() ** 2).reshape(1)
        sharding.allreduce_sum_(cnt)
        sharding.allreduce_sum_(sq)
        c = torch.clamp(cnt, min=1.0)
        w = torch.where(mask, (2.0 / c.float()) * r, torch.zeros_like(r))
        return (sq[0] / c[0]).float(), w, cnt

    def _last_step_points(self, n_last):
        """The network inputs [x, t] of loop step n_last for every trajectory of this rank, from the path store."""
        s = self.s
        nt = (self.K_local + 15) // 16
        PB = self._path_block_floats()
        nx = self._x_image_floats()
        img = self.path[:(s.N + 1) * nt * PB].view(s.N + 1, nt, PB)[n_last, :, :nx].reshape(nt, nx // 64, 4, 16)
        return img.permute(0, 3, 1, 2).reshape(nt * 16, nx // 16)[:self.K_local]

    def iteration(self, l):
        s, lib, cfg, dev = self.s, self.lib, self.cfg, self.dev
        st = nat.stream_ptr(dev)
        d, pb = s.d, s.problem
        diffusion = s.loss_method == 'diffusion'
        bounded, ell = s.bounded, self.elliptic
        reference_noise = s.noise == 'reference'
        for p in self.params:
            p.grad = None
        # ---- host RNG in the reference's order (solver.py:1020-1060, :1078, :1106; EllipticSolver :650-739)
        X_b = s._sample_boundary() if bounded else None
        if ell:
            loss_T = self._boundary_terms(None, X_b, None)      # no RNG inside: order as in the reference
        X = s.sample_domain() if reference_noise else self._sample_domain_device(l)
        K = s.K                                                  # 'two_spheres': the rejection step has just set it
        self._set_batch(K)
        lo, hi = self.lo, self.hi
        t_b = None
        if bounded and not ell:
            t_b = torch.rand(s.K_boundary, 1).to(dev) * pb.T
        if not ell:
            loss_T = self._boundary_terms(X, X_b, t_b)
        xi, rng_state, t0_all = None, None, None
        if ell:
            t0 = torch.zeros(hi - lo, device=dev)
        elif reference_noise:
            t0_all = torch.rand(K, 1) * pb.T
            t0 = t0_all[lo:hi, 0].contiguous().to(dev)
        else:
            t0 = (torch.rand(K, generator=self._gen, device=dev) * pb.T)[lo:hi].contiguous()
        if reference_noise:
            if bounded:
                rng_state = torch.get_rng_state()               # the number of draws is known after the rollout
                n_draw = s.N
            else:
                n_draw = self._executed_steps(t0_all)
            xi_cpu = torch.zeros(s.N, hi - lo, d)
            for n in range(n_draw):
                xi_cpu[n] = torch.randn(K, d)[lo:hi]
            xi = self.pad.last_dim(xi_cpu.to(dev))
        x0 = self.pad.last_dim(X[lo:hi].contiguous())
        flat_k = self.pad.scatter_params(self.flat, self.flat_k)
        self.kcount.zero_()
        ev = None
        if self.events is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        self._launch_fwd(flat_k, x0, t0, xi, l, st)
        if ev is not None:
            ev[1].record()
        if rng_state is not None:                                # rewind, then consume what the reference consumes
            draws = self._draws_executed(t0)
            torch.set_rng_state(rng_state)
            for _ in range(draws):
                torch.randn(K, d)
        # ---- per-trajectory loss weights (K-vectors)
        wV = None
        if diffusion:
            r = self.VN - self.YN
            sq = torch.sum(r.double() ** 2).reshape(1)
            sharding.allreduce_sum_(sq)
            loss = s.alpha[0] * (sq[0] / K).float()
            wV = (2.0 * s.alpha[0] / K) * r
            wY = -wV
        else:
            btype = None if ell else getattr(pb, 'boundary_type', None)
            if bounded and not ell and btype == 'Neumann':
                loss, wY = self._bsde_neumann(t0, K)             # solver.py:1177-1183
            else:
                if ell:
                    target = pb.g(self.XN)                        # solver.py:808
                elif bounded:
                    target = pb.g(self.XN, self.tN)               # :1176
                else:
                    target = pb.f(self.XN)                        # :1174
                r = self.YN - target
                sq = torch.sum(r.double() ** 2).reshape(1)
                sharding.allreduce_sum_(sq)
                loss = (sq[0] / K).float()
                wY = (2.0 / K) * r
        if s.loss_with_stopped:                                  # solver.py:1185-1186 / :803-804: Y against the data at the exit points
            stopped = self._active_steps(t0) < float(s.N)
            target = pb.g(self.XN) if ell else pb.f(self.XN)
            l_st, w_st, _ = self._mean_sq_masked(self.YN - target, stopped, K)
            loss = loss + l_st
            wY = wY + w_st
        if loss_T is not None:
            loss = loss + loss_T.detach()
        self.wY[:self.K_local].copy_(wY)
        if wV is not None:
            self.wV[:self.K_local].copy_(wV)
        if ev is not None:
            ev[2].record()
        self._launch_bwd(flat_k, st)
        self.pad.gather_grad(self.grad_k, self.grad)
        if ev is not None:
            ev[3].record()
            self.events.append(ev)
        sharding.allreduce_sum_(self.grad)
        if any(p.grad is not None for p in self.params):         # identical on every rank: add after the reduce
            self.grad += torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                    for p in self.params])       # (a pure Neumann residual never touches b3)
        if self._extra_grad is not None:                         # the BSDE Neumann residual (already summed over the ranks)
            self.grad += self._extra_grad
            self._extra_grad = None
        self.last_v_l2 = self._v_l2_from_path() if self.log_v_l2 else None     # before the update, as the reference logs it
        self.step += 1
        lr, b1, b2, eps = self._adam_hyper()
        nat.check(lib.psp_adam_step(nat.ptr(self.flat), nat.ptr(self.grad), nat.ptr(self.m), nat.ptr(self.v),
                                    self.P, self.step, lr, b1, b2, eps, st), 'psp_adam_step')
        kc = self.kcount.clone()
        sharding.allreduce_sum_(kc)
        if s.K_test_log is not None:                             # solver.py:1193-1197 / :821-825: after the update, CPU generator
            s._log_test_error('elliptic' if ell else 'parabolic')
        return loss, kc

    def _bsde_neumann(self, t0, K):
        """BSDE loss with a Neumann boundary (solver.py:1177-1183): trajectories that ran out of time are matched with f; the
        Neumann residual takes grad_x V of the LAST executed loop step (the state before its move) against the final X, over ALL
        trajectories, as written there.  Returns (loss, dLoss/dY_N); the residual's parameter gradient goes into p.grad."""
        s, pb = self.s, self.s.problem
        T, dt = pb.T, s.delta_t
        late = self.tN > (T - dt)                                # :1178 (fp32 tensor arithmetic, as there)
        loss_l, wY, n_late = self._mean_sq_masked(self.YN - pb.f(self.XN), late, K)
        loss = loss_l
        if int(n_late.item()) < K:
            n_last = self._loop_steps(t0) - 1
            img = self._last_step_points(n_last)               # [x (d_pad), t]: a zero-padded instance keeps the time behind the padding
            pts = torch.cat([img[:, :s.d], img[:, self.d_pad:self.d_pad + 1]], 1).clone().requires_grad_(True)
            grad_V, = torch.autograd.grad(s.V(pts).squeeze().sum(), pts, create_graph=True)
            XN = self.XN.detach()
            res = torch.sum(grad_V[:, :s.d] * XN, 1) - torch.sum(pb.g(XN, self.tN) * XN, 1)
            term = torch.sum(res ** 2) / K                       # mean over all K trajectories (every rank adds its share)
            gs_ = torch.autograd.grad(term, self.params, allow_unused=True)
            extra = torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(gs_, self.params)])
            tot = term.detach().double().reshape(1)
            sharding.allreduce_sum_(tot)
            sharding.allreduce_sum_(extra)
            self._extra_grad = extra
            loss = loss + tot[0].float()
        return loss, wY

    # ---- the two kernel launches of an iteration (plan_general_deep.py overrides them for value nets of other depths)
    def _launch_fwd(self, flat_k, x0, t0, xi, l, st):
        nat.check(self.lib.psp_gen_rollout_fwd(C.byref(self.cfg), nat.ptr(flat_k), nat.ptr(x0), nat.ptr(t0), nat.ptr(xi),
                                               int(self.s.seed) & 0xFFFFFFFFFFFFFFFF, l, nat.ptr(self.path),
                                               nat.ptr(self.ahat), nat.ptr(self.VN), nat.ptr(self.YN), nat.ptr(self.XN_k),
                                               nat.ptr(self.tN), nat.ptr(self.kcount), st), 'psp_gen_rollout_fwd')

    def _launch_bwd(self, flat_k, st):
        nat.check(self.lib.psp_gen_rollout_bwd(C.byref(self.cfg), nat.ptr(flat_k), nat.ptr(self.path), nat.ptr(self.ahat),
                                               nat.ptr(self.wY), nat.ptr(self.wV), nat.ptr(self.grad_partial), nat.ptr(self.grad_k),
                                               st), 'psp_gen_rollout_bwd')

    def _x_image_floats(self):
        """Floats of the X_n register image at the head of a path block (the network input incl. the time row)."""
        return 4 * ((self.d_pad + 1 + 15) // 16) * 64

    def _path_block_floats(self):
        """Floats of one (step, tile) block of the path store for the CURRENT batch."""
        nt = (self.K_local + 15) // 16
        return self.sizes.path_bytes // 4 // ((self.s.N + 1) * nt)

    def range_fallbacks(self):
        """Iterations the range guard sent to the fp32-MFMA kernels so far; one device read."""
        return int(self.range_flag[1].item()) if self.range_flag is not None else 0

    def _adam_hyper(self):
        """lr / betas / eps of V's OWN optimiser (function_space.py:131; solver.py:1188 steps V.optim)."""
        opt = getattr(self.net, 'optim', None)
        if opt is not None and len(opt.param_groups) > 0:
            g = opt.param_groups[0]
            if g.get('weight_decay', 0) or g.get('amsgrad', False):
                raise NotImplementedError('the native Adam implements weight_decay = 0, amsgrad = False (the reference default)')
            b = g.get('betas', (0.9, 0.999))
            return float(g['lr']), float(b[0]), float(b[1]), float(g.get('eps', 1e-8))
        return float(self.s.lr), 0.9, 0.999, 1e-8

    @property
    def log_v_l2(self):
        s = self.s
        return (self.elliptic and getattr(s, 'v_l2_error_flag', True) and hasattr(s.problem, 'v_true')
                and getattr(s, 'mlp_dtype', 'fp32') != 'bf16')          # (the bf16 mode keeps its X_n images as bf16 pairs)

    def _v_l2_from_path(self):
        """EllipticSolver's V_L2 log (solver.py:718, 738, 813): mean_k sum_{n alive} (V(X_n) - v_true(X_n))^2 dt, from the X_n
        register images the forward kernel left in the path store (slot n = the state BEFORE the move of step n; image
        float ks * 64 + 16 q + j holds feature 4 ks + q of sample j).  A trajectory with m active steps is alive at steps
        0..m (the step that finds it outside still counts it, :736-738), capped by the N steps of the loop; slots a tile did
        not execute (it left the loop early) are masked, never read as numbers.  Diagnostics: K N small-net evaluations in
        torch, off the timed path (v_l2_error_flag=False skips it)."""
        s, cfg = self.s, self.cfg
        N, nt = s.N, (self.K_local + 15) // 16
        PB = self._path_block_floats()
        nx = self._x_image_floats()
        m = torch.round(self.tN / cfg.dt)                                            # active steps per trajectory
        n_used = min(N, int(m.max().item()) + 1)
        img = self.path[:(N + 1) * nt * PB].view(N + 1, nt, PB)[:n_used, :, :nx].reshape(n_used, nt, nx // 64, 4, 16)
        X = img.permute(0, 1, 4, 2, 3).reshape(n_used, nt * 16, nx // 16)[:, :self.K_local, :s.d]
        alive = torch.arange(n_used, device=self.dev).unsqueeze(1) <= m.unsqueeze(0)
        with torch.no_grad():
            Xf = torch.where(alive.unsqueeze(2), X, torch.zeros_like(X)).reshape(-1, s.d)
            Xf = torch.where(alive.reshape(-1, 1), Xf, self.XN[:1].expand_as(Xf))     # (v_true may be singular at the origin)
            err = (s.V(Xf).squeeze() - torch.as_tensor(s.problem.v_true(Xf)).float().to(self.dev).squeeze()) ** 2
        tot = torch.where(alive, err.reshape(n_used, self.K_local), torch.zeros((), device=self.dev)).sum().reshape(1) * s.delta_t_np
        sharding.allreduce_sum_(tot)
        return tot[0] / float(s.K)

    def train(self):
        import time
        s = self.s
        losses, counts, vl2, Ks = [], [], [], []
        t_block = time.time()
        for l in range(s.L):
            loss, kc = self.iteration(l)
            losses.append(loss)
            counts.append(kc)
            Ks.append(s.K)
            if self.last_v_l2 is not None:
                vl2.append(self.last_v_l2)
            if (s.verbose and l % s.print_every == 0) or l == s.L - 1:
                vals = torch.stack(losses).cpu().tolist()          # one sync per block
                now = time.time()
                s.loss_log += vals
                s.K_log += [int(c.item()) for c in counts]
                s.V_L2_log += torch.stack(vl2).cpu().tolist() if vl2 else [0.0] * len(vals)
                s.times += [(now - t_block) / len(vals)] * len(vals)
                t_block = now
                if s.verbose and l % s.print_every == 0:
                    print('%d - loss = %.4e, v L2 error = %.4e, %.4f s/iter' % (l, s.loss_log[-1], s.V_L2_log[-1], s.times[-1]))
                losses, counts, vl2, Ks = [], [], [], []
        s.range_fallback_iterations = self.range_fallbacks()
