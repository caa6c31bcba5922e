"""Native (HIP) execution plan for Solver.train on MI355X.

Replaces the body of the reference's training iteration (solver.py:430-514) by five
asynchronous launches through the C ABI of libpsp_hip.so (include/psp.h):

    psp_hjb_rollout_fwd  ->  psp_hjb_terminal_reduce  -> [all-reduce (sum D, sum D^2)]
    psp_hjb_rollout_bwd  -> [all-reduce flat gradient] -> psp_adam_step

torch supplies device memory, the stream and (for world_size > 1) the RCCL all-reduces;
no arithmetic of the hot path runs in torch.  Trajectories are sharded across ranks by
contiguous blocks; noise is indexed by GLOBAL trajectory id so the result does not depend
on the number of ranks.

Path stores larger than a budget (configs[4]: d=500, K=1048576, N=200 needs 946 GB) are bounded by K-CHUNKING: the
rank's trajectories are processed in chunks that share one path-store buffer.  Because the log-variance weights
w_k = (2/K)(D_k - mean D) need the GLOBAL mean, a chunked iteration is either
  'two_gradient' (log-variance / moment, detached): per chunk forward + store, backward with weights D_k - c (c = mean of
      the first chunk, any constant is exact) AND backward with unit weights; afterwards
      grad = (2/K) [G1 - (mean D - c) G0].  No forward recompute; costs one extra backward launch per chunk.
  'recompute' (every other native mode): pass 1 = forward per chunk WITHOUT the store -> global sums / weights,
      pass 2 = forward with the store (same Philox counters: bit-identical), [adjoint sweep,] backward.
The reference keeps the whole autograd graph instead (SURVEY.md 7 "Activation memory").
"""
import ctypes as C

import numpy as np
import torch

try:
    from . import native as nat
    from . import native_shapes as shapes
    from . import sharding
except ImportError:
    import native as nat
    import native_shapes as shapes
    import sharding


def _u_tables(problem):
    """problem.u_true_tables() (the double wells' finite-difference reference control, problems.py) or None."""
    fn = getattr(problem, 'u_true_tables', None)
    return fn() if fn is not None else None


def _overridden(problem):
    try:
        from .problems import coefficients_overridden
    except ImportError:
        from problems import coefficients_overridden
    return coefficients_overridden(problem)


class PlanUnsupported(Exception):
    """The (problem, net, loss, flags) combination is outside the native catalogue."""


def native_eligibility(solver):
    """Returns None if the solver can run natively, else a human-readable reason."""
    if solver.device.type != 'cuda':
        return 'device is %s (the HIP rollout needs a GPU)' % solver.device
    if solver.approx_method != 'control' or solver.time_approx != 'inner':
        return "only approx_method='control' with time_approx='inner' is native"
    if solver.loss_method not in ('log-variance', 'moment', 'variance', 'cross_entropy', 'relative_entropy'):
        return ("loss_method %r is not native (log-variance, moment, variance, cross_entropy, relative_entropy)"
                % solver.loss_method)
    if solver.burgers_drift:
        return 'burgers_drift is not native'
    if solver.u_l2_error_flag and getattr(solver.problem, 'u_true_x_independent', False) is not True \
            and getattr(solver.problem, 'u_true_linear_in_x', False) is not True and _u_tables(solver.problem) is None:
        return ('u_l2_error_flag=True evaluates problem.u_true(X_n, t_n) on the host every step '
                '(reference solver.py:491-494); a u_true that does not depend on x (LLGC) is logged inside the kernels, one '
                'that is linear in x (LQGC) or tabulated per coordinate (the double wells) from the path store -- pass '
                'u_l2_error_flag=False for the native plan otherwise')
    if solver.compute_gradient_variance > 0 or solver.log_gradient:
        return 'per-iteration diagnostics (gradient variance / gradient log) are not native'
    if solver.metastability_logs is not None:
        return 'metastability_logs needs X_N on the host every iteration'
    net = solver.z_n
    shape = getattr(net, 'native_shape', lambda: None)()
    if shape is None or shape[0] != solver.d + 1 or shape[2] != solver.d:
        return 'control net is not a two-hidden-layer tanh MLP (MySequential)'
    spec_fn = getattr(solver.problem, 'native_spec', None)
    spec = spec_fn() if spec_fn is not None else None
    if spec is None:
        return 'problem has no native_spec() (coefficients outside the native catalogue)'
    over = _overridden(solver.problem)
    if over is not None:
        return 'problem.%s is not the catalogue implementation native_spec() describes' % over
    if not nat.is_built():
        raise nat.NativeLibraryError('libpsp_hip.so is not built; run __graft_entry__.build()')
    probe = nat.HjbConfig()
    probe.K_local, probe.N, probe.K_global = 16, 1, 16
    probe.drift_kind, probe.sigma_kind = spec['drift'][0], spec['sigma'][0]
    probe.runcost_kind, probe.term_kind = spec['runcost'][0], spec['term'][0]
    probe.adaptive = 1 if solver.adaptive_forward_process else 0
    probe.store_path = 1
    chosen, why = shapes.choose(probe, solver.d, shape[1])
    if chosen is None:
        return why
    return None


def chunk_scratch_sizes(cfg, chunk_Ks):
    """psp_hjb_sizes that serve EVERY chunk size in `chunk_Ks`: the field-wise maximum of the per-size queries."""
    sizes = None
    for k in sorted(set(int(k) for k in chunk_Ks)):
        probe = nat.HjbConfig.from_buffer_copy(cfg)
        probe.K_local = k
        q = nat.query(probe)
        if sizes is None:
            sizes = q
        else:
            for f in ('path_bytes', 'fwd_partial_bytes', 'grad_partial_bytes', 'fwd_workgroups', 'bwd_workgroups'):
                setattr(sizes, f, max(getattr(sizes, f), getattr(q, f)))
    return sizes


class HjbNativePlan:
    def __init__(self, solver, noise='reference'):
        reason = native_eligibility(solver)
        if reason is not None:
            raise PlanUnsupported(reason)
        self.s = solver
        self.lib = nat.load()
        self.noise = noise
        dev = solver.device
        self.dev = dev
        self.dist, self.rank, self.world = sharding.dist_info()
        K = solver.K
        try:
            lo, hi = sharding.shard_bounds(K, self.rank, self.world)
        except ValueError as e:
            raise PlanUnsupported(str(e))
        self.K_local, self.k_offset = hi - lo, lo
        net = solver.z_n
        self.net = net
        self.H = net.native_shape()[1]
        self._flatten(net)
        spec = solver.problem.native_spec()
        self._keep = []   # device tensors referenced by raw pointer from the config

        def dev_f32(t):
            if t is None:
                return None
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t

        cfg = nat.HjbConfig()
        cfg.K_local, cfg.N = self.K_local, solver.N
        cfg.K_global, cfg.k_offset = K, self.k_offset
        cfg.dt = float(solver.delta_t.item())
        cfg.sqrt_dt = float(solver.sq_delta_t.item())
        cfg.drift_kind = spec['drift'][0]
        cfg.sigma_kind = spec['sigma'][0]
        cfg.sigma_scale = float(spec['sigma'][2])
        cfg.runcost_kind = spec['runcost'][0]
        cfg.term_kind = spec['term'][0]
        cfg.adaptive = 1 if solver.adaptive_forward_process else 0
        cfg.loss_kind = {'log-variance': nat.LOSS_LOG_VARIANCE, 'moment': nat.LOSS_MOMENT,
                         'relative_entropy': nat.LOSS_REL_ENTROPY}.get(solver.loss_method, nat.LOSS_WEIGHTS)
        self.generic_loss = cfg.loss_kind == nat.LOSS_WEIGHTS
        self.relent = solver.loss_method == 'relative_entropy'
        # gradients through the state path (the reference's default flags): forward with the attached-mode image in
        # the xi slot, reverse-time adjoint sweep, then the ordinary backward with unit weights (include/psp.h)
        self.attached = bool(solver.adaptive_forward_process and not solver.detach_forward)
        cfg.noise_mode = nat.NOISE_PHILOX if noise == 'philox' else nat.NOISE_SUPPLIED
        cfg.store_path = 3 if self.relent else (2 if self.attached else 1)
        # matrix products: 'fp32' = v_mfma_f32_16x16x4_f32; 'f16x3' = fp32-grade split products on the f16 pipe (three
        # v_mfma_f32_16x16x32_f16 per fp32 product, csrc/hjb_kernels.h gemm_Tx, csrc/hjbx_kernels.h -- same parity bounds);
        # 'auto' (the default) = 'f16x3' where those kernels exist, fit the LDS and the tile-per-wave forward would run anyway
        # (more than two 16-trajectory tiles per CU: the small-K forward kernels are fp32 only), else 'fp32'
        want = getattr(solver, 'mlp_dtype', 'auto')
        cfg.mlp_dtype = {'auto': nat.MLP_FP32, 'fp32': nat.MLP_FP32, 'bf16': nat.MLP_BF16_FWD, 'f16x3': nat.MLP_F16X3}[want]
        # kernel instance: the exact (d, H) if compiled, else the cheapest larger one (zero padding, native_shapes.py)
        chosen, why = shapes.choose(cfg, solver.d, self.H)
        if chosen is None:
            raise PlanUnsupported(why)
        self.d_pad, self.H_pad, self.family, sizes = chosen
        if want == 'auto' and torch.device(dev).type == 'cuda':
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            # (the wide family has no small-K kernels: its split-product forward serves every K)
            if self.family == 2 or (self.K_local + 15) // 16 > 2 * cus:
                cfg.mlp_dtype = nat.MLP_F16X3
                rc, sizes_x3, _ = nat.query_rc(cfg)
                if rc == 0:
                    sizes = sizes_x3
                else:
                    cfg.mlp_dtype = nat.MLP_FP32
        self.matrix_mode = {nat.MLP_FP32: 'fp32', nat.MLP_BF16_FWD: 'bf16', nat.MLP_F16X3: 'f16x3'}[cfg.mlp_dtype]
        # range guard of the split-product mode (include/psp.h: psp_hjb_config.range_flag): the reference is fp32 end to end
        # (solver.py:39-40); an f16x3 operand beyond 65504 turns D_k into NaN, the forward call raises a device flag and the
        # fp32-MFMA kernels -- enqueued behind the split ones, predicated on that flag -- redo the iteration.  No host sync.
        self.range_flag = None
        if cfg.mlp_dtype == nat.MLP_F16X3 and getattr(solver, 'range_guard', True):
            self.range_flag = torch.zeros(4, dtype=torch.int32, device=dev)
            cfg.range_flag = nat.ptr(self.range_flag)
            if nat.query_rc(cfg)[0] != 0:                # (the fp32-MFMA tables of this instance do not fit the LDS: unguarded)
                cfg.range_flag, self.range_flag = None, None
        # store_path 4 (include/psp.h): the forward keeps X_n, h1, h2 only and the backward producers regenerate xi from the
        # Philox counters -- where the stored image would be xi itself (detached adaptive process, on-device noise), in the
        # narrow kernel family, on the eager iteration (psp_hjb_rollout_bwd_step of the hipGraph replay takes no seed; that
        # regime -- at most two tiles per CU -- is launch-bound, not store-bound).  Solver(path_noise='store') keeps xi.
        self.regen_xi = False
        if (getattr(solver, 'path_noise', 'auto') != 'store' and cfg.store_path == 1 and cfg.noise_mode == nat.NOISE_PHILOX
                and cfg.adaptive and self.family == 1 and torch.device(dev).type == 'cuda'
                and getattr(solver, 'use_graph', 'auto') is not True
                and (self.K_local + 15) // 16 > 2 * torch.cuda.get_device_properties(dev).multi_processor_count):
            cfg.store_path = 4
            if nat.query_rc(cfg)[0] == 0:
                self.regen_xi = True
            else:
                cfg.store_path = 1
        self.pad = shapes.ParamPad(solver.d, self.H, self.d_pad, self.H_pad, dev)
        pad = self.pad
        cfg.drift = nat.ptr(dev_f32(pad.drift_or_sigma(spec['drift'][1]))) if spec['drift'][1] is not None else None
        cfg.sigma = nat.ptr(dev_f32(pad.drift_or_sigma(spec['sigma'][1]))) if spec['sigma'][1] is not None else None
        cfg.runcost = nat.ptr(dev_f32(pad.vec(spec['runcost'][1]))) if spec['runcost'][1] is not None else None
        cfg.term = nat.ptr(dev_f32(pad.vec(spec['term'][1])))
        self.cfg = cfg
        assert sizes.n_params == pad.Pp, (sizes.n_params, pad.Pp)
        self._setup_chunks(solver, cfg, sizes)
        solver.path_plan = dict(n_chunks=self.n_chunks, chunk_mode=self.chunk_mode, chunk_K=self.chunk_K)   # what the budget chose
        sizes = self.sizes
        # kernel-side (padded) parameter and gradient vectors; identical to the real ones when nothing is padded
        self.flat_k = self.flat if pad.identity else pad.new_padded_params()
        self.grad_k = None
        self.path = torch.empty(sizes.path_bytes // 4, dtype=torch.float32, device=dev)
        self.fwd_partial = torch.empty(max(1, self.n_chunks) * (sizes.fwd_partial_bytes // 8), dtype=torch.float64, device=dev)
        self.grad_partial = torch.empty(sizes.grad_partial_bytes // 4, dtype=torch.float32, device=dev)
        self.D = torch.empty(self.K_local, dtype=torch.float32, device=dev)
        self.Yn = torch.empty(self.K_local, dtype=torch.float32, device=dev) if self.generic_loss else None
        self.w = torch.empty(self.K_local, dtype=torch.float32, device=dev) if self.generic_loss else None
        self.sums = torch.zeros(2, dtype=torch.float64, device=dev)
        self.grad = torch.empty(self.P, dtype=torch.float32, device=dev)
        self.grad_k = self.grad if pad.identity else torch.empty(pad.Pp, dtype=torch.float32, device=dev)
        self.m = torch.zeros(self.P, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.P, dtype=torch.float32, device=dev)
        self.x0_vec = dev_f32(pad.vec(solver.X_0.detach().to(dev)))
        self.ul2 = None
        self.ul2_gain = None
        self.ul2_tab = None
        if solver.u_l2_error_flag and _u_tables(solver.problem) is not None:
            # u*(x, t) tabulated per coordinate on a grid (the double wells: problems.py u_true_tables, reference problems.py:
            # 277-281, 399-404, 471-476): the tables go to the device once, the log is formed from the path store (_ul2_from_path)
            import numpy as np
            tb = _u_tables(solver.problem)
            self.ul2_tab = dict(tables=[torch.tensor(np.asarray(t), dtype=torch.float32, device=dev) for t in tb['tables']],
                                group=torch.tensor(tb['group_of_dim'], dtype=torch.long, device=dev),
                                xb=float(tb['xb']), dx=float(tb['dx']),
                                n_ref=[int(np.ceil(n * solver.delta_t_np / tb['delta_t'])) for n in range(solver.N)])
            self.XN_k = torch.empty(self.K_local, self.d_pad, dtype=torch.float32, device=dev)
        elif solver.u_l2_error_flag and getattr(solver.problem, 'u_true_x_independent', False) is not True:
            # u*(x, t_n) = M_n x (LQGC, problems.py:169-171): M_n once per plan by probing u_true with the unit vectors; the log
            # is then formed from the X_n and h2 images of the path store after the forward kernel (_ul2_from_path)
            import numpy as np
            eye = torch.eye(solver.d)
            gains = [torch.tensor(np.asarray(solver.problem.u_true(eye, n * solver.delta_t_np))).reshape(solver.d, solver.d).float()
                     for n in range(solver.N)]                                       # u_true returns (d, K): column i = M e_i
            self.ul2_gain = torch.stack(gains).to(dev)                             # (N, d, d) = M_n
            self.XN_k = torch.empty(self.K_local, self.d_pad, dtype=torch.float32, device=dev)
        elif solver.u_l2_error_flag:
            # u_L2 log (solver.py:491-494) for an x-independent reference control: u*(t_n) once per plan instead of
            # problem.u_true(X.cpu(), t_n) every step of every iteration; the kernels accumulate |-Z_n - u*(t_n)|^2 dt
            import numpy as np
            probe = torch.zeros(1, solver.d)
            rows = [torch.tensor(np.asarray(solver.problem.u_true(probe, n * solver.delta_t_np))).reshape(solver.d, -1)[:, 0].float()
                    for n in range(solver.N)]
            table = pad.last_dim(torch.stack(rows).to(dev)).reshape(-1)
            self.uref = dev_f32(torch.cat([table, torch.zeros(16, device=dev)]))   # 16 floats of slack: the kernels read whole blocks
            self.ul2 = torch.zeros(self.K_local, dtype=torch.float32, device=dev)
            cfg.u_ref, cfg.u_l2_out = nat.ptr(self.uref), nat.ptr(self.ul2)
        if self.attached or self.relent:
            self.cfg_w = nat.HjbConfig.from_buffer_copy(cfg)          # backward with explicit trajectory weights
            self.cfg_w.loss_kind = nat.LOSS_WEIGHTS
            self.w_bwd = torch.empty(self.K_local, dtype=torch.float32, device=dev)
            if self.attached:
                self.XN_k = torch.empty(self.K_local, self.d_pad, dtype=torch.float32, device=dev)
                self.mu = torch.zeros(self.K_local, dtype=torch.float32, device=dev)
                self.nu = torch.zeros(self.K_local, dtype=torch.float32, device=dev)
                self.wT = torch.zeros(self.K_local, dtype=torch.float32, device=dev)
        self.step = sharding.adam_state_import(net.flat_layout(), self.m, self.v, getattr(net, 'optim', None))
        self.events = None   # bench.py: list collecting HIP-event pairs around the two rollout kernels
        self.pass1_events = None
        if self.n_chunks > 1:
            self._alloc_chunks()
        # hipGraph replay of the launch-bound small-K iteration (module docstring of include/psp.h: psp_iter_state)
        self._graph = None
        self._graph_key = None
        self._graph_iter = -1
        self._eager_done = 0
        self.graph_active = False
        # learnable Y_0 (solver.py:372-374): tiny Adam in torch on a 1-element tensor
        self.learn_y0 = bool(solver.learn_Y_0)
        if self.learn_y0:
            self.y0_param = solver.y_0.Y_0
            self.y0_m = torch.zeros(1, dtype=torch.float32, device=dev)
            self.y0_v = torch.zeros(1, dtype=torch.float32, device=dev)
            self.y0_grad = torch.zeros(1, dtype=torch.float32, device=dev)
            sharding.adam_state_import([self.y0_param], self.y0_m, self.y0_v, getattr(solver.y_0, 'optim', None))

    # ------------------------------------------------------------------------------------
    def _flatten(self, net):
        """Re-home the net's parameters as views of one flat fp32 buffer (psp.h layout) so
        that state_dict / save_networks / Z_n keep working on live values."""
        params = net.flat_layout()
        self.P = sum(p.numel() for p in params)
        flat = torch.empty(self.P, dtype=torch.float32, device=self.dev)
        off = 0
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            off += n
        self.flat = flat

    def _stream(self):
        return nat.stream_ptr(self.dev)

    # ---- K-chunking (module docstring) -------------------------------------------------------------------------------
    DEFAULT_PATH_BUDGET = 180 * 2 ** 30         # five eighths of the 288 GB of HBM3E when the device cannot be asked

    def _default_budget(self):
        """Five eighths of THIS device's TOTAL memory (round 4; a third before: config 5's per-GPU share -- d = 500, K = 131 072,
        N = 200, a 120 GB store -- then ran in chunks and paid its backward twice, 179 ms, where it fits the card whole, 139 ms;
        everything else the plan allocates is a few hundred MB) -- a function of the device alone, so the same seed and configuration give the
        same chunk count (hence the same summation order of the partial sums and gradients) from run to run and on every rank of
        a job, whatever else occupies the card at the moment.  (Round 3 also capped it at 80 % of the memory free right now;
        a store that then does not fit fails in its allocation and says so -- Solver(path_budget_bytes=...) / path_chunks pick
        another split.)  The chosen n_chunks / chunk_mode are recorded on the solver (solver.path_plan) and in bench.py's line."""
        try:
            return max(1 << 28, int(torch.cuda.get_device_properties(self.dev).total_memory) * 5 // 8)
        except Exception:                                # size queries on a machine without a GPU
            return self.DEFAULT_PATH_BUDGET

    def _setup_chunks(self, solver, cfg, sizes):
        """Decides the number of trajectory chunks from the path-store budget and re-queries the scratch sizes for one
        chunk.  Sets self.n_chunks, self.chunk_K (trajectories per chunk, a multiple of 16), self.sizes, self.chunk_mode."""
        budget = getattr(solver, 'path_budget_bytes', None) or self._default_budget()
        forced = getattr(solver, 'path_chunks', None)
        n = int(forced) if forced else max(1, -(-int(sizes.path_bytes) // int(budget)))
        n = min(n, max(1, (self.K_local + 15) // 16))
        self.n_chunks, self.chunk_K, self.sizes = 1, self.K_local, sizes
        self.chunk_mode = None
        if n <= 1:
            return
        if solver.u_l2_error_flag and getattr(solver.problem, 'u_true_x_independent', False) is not True:
            raise PlanUnsupported('the u_L2 log of an x-dependent reference control reads the whole path store: not with K-chunking')
        Kc = -(-self.K_local // n)
        Kc = -(-Kc // 16) * 16
        if not forced:
            # whole launch waves: a chunk of a multiple of (16 trajectories x 4 tiles per workgroup x CUs) fills every CU the
            # same number of times (K = 1048576 under a 32 GiB budget: 32 chunks of 32768 instead of 29 ragged ones, +24 %)
            per_b = max(1, int(sizes.path_bytes) // max(1, self.K_local))          # store bytes per trajectory
            fit = int(budget) // per_b
            try:
                cus = torch.cuda.get_device_properties(self.dev).multi_processor_count
            except Exception:                                                    # (size queries on a machine without a GPU)
                cus = 256
            quantum = 16 * 4 * cus
            if fit >= quantum:
                Kc = min(Kc if Kc % quantum == 0 else (fit // quantum) * quantum, (fit // quantum) * quantum)
        n = -(-self.K_local // Kc)
        if n <= 1:
            return
        # every DISTINCT chunk size is queried and each scratch buffer takes the maximum: the forward grid (hence the
        # fp64 partials, and in the wide family the operand tables behind them) is not monotone in K_local -- a ragged
        # last chunk can pick another forward kernel with a LARGER grid than the full chunks (K_local = 2064 in two
        # chunks: 65 tiles -> feature-split kernel, grid 65; 64 tiles -> quad kernel, grid 256)
        self.sizes = chunk_scratch_sizes(cfg, {Kc, self.K_local - (n - 1) * Kc})
        self.n_chunks, self.chunk_K = n, Kc
        mode = getattr(solver, 'chunk_mode', 'auto')
        simple = (not self.attached and not self.relent and not self.generic_loss)      # log-variance / moment, detached
        if mode == 'auto':
            mode = 'two_gradient' if simple else 'recompute'
        if mode == 'two_gradient' and not simple:
            raise PlanUnsupported("chunk_mode='two_gradient' needs a detached log-variance or moment run")
        if mode not in ('two_gradient', 'recompute'):
            raise ValueError("chunk_mode must be 'auto', 'two_gradient' or 'recompute'")
        self.chunk_mode = mode

    def _alloc_chunks(self):
        """Per-chunk configs (K_local, k_offset and the per-trajectory output pointers moved to the chunk) and the small
        per-chunk result rows; the big scratch buffers (path store, partial gradients) are shared by all chunks."""
        dev, cfg, n, Kc = self.dev, self.cfg, self.n_chunks, self.chunk_K
        self.chunks = []
        for i in range(n):
            off = i * Kc
            k = min(Kc, self.K_local - off)
            c = nat.HjbConfig.from_buffer_copy(cfg)
            c.K_local, c.k_offset = k, self.k_offset + off
            if self.ul2 is not None:
                c.u_l2_out = nat.ptr(self.ul2, off)
            c0 = nat.HjbConfig.from_buffer_copy(c)       # pass 1 of 'recompute': no path store
            c0.store_path = 0
            cw = nat.HjbConfig.from_buffer_copy(c)       # backward with explicit trajectory weights
            cw.loss_kind = nat.LOSS_WEIGHTS
            self.chunks.append((off, k, c, c0, cw))
        self.sums_c = torch.zeros(n, 2, dtype=torch.float64, device=dev)
        Pp = self.pad.Pp
        self.grad_rows = torch.zeros(n, Pp, dtype=torch.float32, device=dev)
        if self.chunk_mode == 'two_gradient' and self.s.loss_method == 'log-variance':
            self.grad_rows0 = torch.zeros(n, Pp, dtype=torch.float32, device=dev)
            self.w_chunk = torch.empty(Kc, dtype=torch.float32, device=dev)
            self.ones_chunk = torch.ones(Kc, dtype=torch.float32, device=dev)
        self.fp_stride = self.sizes.fwd_partial_bytes // 8

    def _iteration_chunked(self, l, loss_out, ul2_out=None):
        s, lib, cfg = self.s, self.lib, self.cfg
        st = self._stream()
        seed = int(s.seed) & 0xFFFFFFFFFFFFFFFF
        xi_full = x0 = None
        if self.noise == 'reference':
            xi_full, x0 = self._reference_noise()
        elif s.random_X_0:
            g = torch.Generator(device=self.dev)
            g.manual_seed(int(s.seed) * 1000003 + l)
            x0 = self.pad.last_dim(torch.randn(s.K, s.d, generator=g, device=self.dev)[
                self.k_offset:self.k_offset + self.K_local].contiguous())
        flat_k = self.pad.scatter_params(self.flat, self.flat_k)
        y0_ptr = nat.ptr(self.y0_param) if self.learn_y0 else None
        K = float(s.K)

        def fwd(i, c, store):
            off, k = self.chunks[i][0], self.chunks[i][1]
            xi = xi_full[:, off:off + k].contiguous() if xi_full is not None else None
            x0_p, x0_stride = (nat.ptr(x0, off * self.d_pad), self.d_pad) if x0 is not None else (nat.ptr(self.x0_vec), 0)
            nat.check(lib.psp_hjb_rollout_fwd(C.byref(c), nat.ptr(flat_k), x0_p, x0_stride, y0_ptr, nat.ptr(xi), seed, l,
                                              nat.ptr(self.path) if store else None, nat.ptr(self.D, off),
                                              nat.ptr(self.XN_k, off * self.d_pad) if (self.attached and store) else None,
                                              nat.ptr(self.Yn, off) if self.Yn is not None else None,
                                              nat.ptr(self.fwd_partial, i * self.fp_stride), st), 'psp_hjb_rollout_fwd')
            nat.check(lib.psp_hjb_terminal_reduce(C.byref(c), nat.ptr(self.fwd_partial, i * self.fp_stride),
                                                  nat.ptr(self.sums_c, 2 * i), st), 'psp_hjb_terminal_reduce')
            return xi

        def bwd(i, c, w_t, w_off, out_rows, xi):
            nat.check(lib.psp_hjb_rollout_bwd(C.byref(c), nat.ptr(flat_k), nat.ptr(xi), seed, l, nat.ptr(self.path),
                                              nat.ptr(w_t, w_off), nat.ptr(self.sums), nat.ptr(self.grad_partial),
                                              nat.ptr(out_rows, i * self.pad.Pp), st), 'psp_hjb_rollout_bwd')

        def events():
            if self.events is None:
                return None
            return [torch.cuda.Event(enable_timing=True) for _ in range(4)]

        if self.chunk_mode == 'two_gradient':
            logvar = s.loss_method == 'log-variance'
            for i, (off, k, c, c0, cw) in enumerate(self.chunks):
                ev = events()
                if ev:
                    ev[0].record()
                xi = fwd(i, c, True)
                if ev:
                    ev[1].record()
                if logvar:
                    if i == 0:
                        shift = (self.sums_c[0, 0] / float(k)).to(torch.float32)       # device scalar, no sync
                    torch.sub(self.D[off:off + k], shift, out=self.w_chunk[:k])
                    if ev:
                        ev[2].record()
                    bwd(i, cw, self.w_chunk, 0, self.grad_rows, xi)
                    if ev:
                        ev[3].record()
                        self.events.append(ev)
                    bwd(i, cw, self.ones_chunk, 0, self.grad_rows0, xi)
                else:                                    # moment: w_k = (2/K) D_k needs nothing global
                    if ev:
                        ev[2].record()
                    bwd(i, c, self.D, off, self.grad_rows, xi)
                    if ev:
                        ev[3].record()
                        self.events.append(ev)
            torch.sum(self.sums_c, 0, out=self.sums)
            sharding.allreduce_sum_(self.sums)           # collective 1: 16 bytes
            loss = sharding.loss_from_sums(self.sums, s.K, s.loss_method)
            if logvar:
                g1, g0 = self.grad_rows.sum(0), self.grad_rows0.sum(0)
                corr = (self.sums[0] / K - shift.double()).to(torch.float32)
                torch.mul(g1 - corr * g0, 2.0 / K, out=self.grad_k)
            else:
                torch.sum(self.grad_rows, 0, out=self.grad_k)
        else:
            # pass 1: forward without the path store -> D, global sums, loss weights
            p1 = None
            if self.pass1_events is not None:
                p1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                p1[0].record()
            for i, (off, k, c, c0, cw) in enumerate(self.chunks):
                fwd(i, c0, False)
            if p1:
                p1[1].record()
                self.pass1_events.append(p1)
            torch.sum(self.sums_c, 0, out=self.sums)
            sharding.allreduce_sum_(self.sums)           # collective 1
            w_t, use_w = self.D, False
            if self.generic_loss:
                loss, w_t = self._generic_loss_weights()
                use_w = True
            else:
                loss = sharding.loss_from_sums(self.sums, s.K, s.loss_method)
            if self.attached:
                wT = None
                if self.relent:
                    self.mu.zero_()
                    self.nu.fill_(1.0 / K)
                elif self.generic_loss:
                    self.mu.copy_(w_t)
                    if s.loss_method == 'cross_entropy':
                        wT = self.wT
                        wT.copy_(-self.Yn * w_t)
                else:
                    self.mu.copy_(sharding.loss_weights(self.D, self.sums, s.K, s.loss_method))
                self.w_bwd.fill_(1.0)
                w_t, use_w = self.w_bwd, True
            elif self.relent:
                self.w_bwd.fill_(float(cfg.sqrt_dt) / K)
                w_t, use_w = self.w_bwd, True
            # pass 2: forward with the store (bit-identical rollout), [adjoint sweep,] backward
            for i, (off, k, c, c0, cw) in enumerate(self.chunks):
                ev = events()
                if ev:
                    ev[0].record()
                xi = fwd(i, c, True)
                if ev:
                    ev[1].record()
                    ev[2].record()
                if self.attached:
                    nat.check(lib.psp_hjb_adjoint_sweep(C.byref(c), nat.ptr(flat_k), nat.ptr(self.path),
                                                        nat.ptr(self.XN_k, off * self.d_pad), nat.ptr(self.mu, off),
                                                        nat.ptr(self.nu, off) if self.relent else None,
                                                        nat.ptr(wT, off) if wT is not None else None,
                                                        nat.ptr(self.fwd_partial, i * self.fp_stride), st), 'psp_hjb_adjoint_sweep')
                bwd(i, cw if use_w else c, w_t, off, self.grad_rows, xi)
                if ev:
                    ev[3].record()
                    self.events.append(ev)
            torch.sum(self.grad_rows, 0, out=self.grad_k)
        loss_out[l] = loss.to(torch.float32)
        if self.ul2 is not None and ul2_out is not None:
            m = (self.ul2.sum() / K).reshape(1)
            sharding.allreduce_sum_(m)
            ul2_out[l:l + 1] = m
        self._finish_step(st, self.w if self.generic_loss else None)
        return loss

    # ---- hipGraph replay ---------------------------------------------------------------------------------------------
    def _graph_wanted(self):
        """The iteration is captured into a hipGraph when it is launch-bound (at most two 16-trajectory tiles per CU, the
        regime of the feature-split forward kernel) and nothing in it needs a per-iteration host argument: on-device noise,
        fixed X_0, a loss the kernels form themselves (with or without the adjoint sweep of an attached forward process: its
        trajectory weights are device arithmetic on D and the sums), one rank, no u_L2 log.  Solver(use_graph=True / False) overrides
        the size rule (never the eligibility)."""
        want = getattr(self.s, 'use_graph', 'auto')
        if want is False:
            return False
        ok = (self.world == 1 and self.noise == 'philox' and not self.s.random_X_0 and not self.generic_loss
              and self.ul2 is None and self.ul2_gain is None and self.ul2_tab is None and self.n_chunks == 1)
        if not ok:
            return False
        if want is True:
            return True
        cus = torch.cuda.get_device_properties(self.dev).multi_processor_count
        return (self.K_local + 15) // 16 <= 2 * cus

    def _graph_state_upload(self, l):
        b = self._graph_hyper
        st = nat.IterState()
        nat.check(self.lib.psp_iter_state_init(C.byref(st), int(l), int(self.step) + 1, b[1], b[2]), 'psp_iter_state_init')
        host = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8)
        self._gstate.copy_(host)
        self._graph_iter = l

    def _graph_body(self, loss_out):
        """The launches of one iteration with every per-iteration quantity read from the device psp_iter_state."""
        s, lib, cfg = self.s, self.lib, self._gcfg
        st = self._stream()
        state = nat.ptr(self._gstate)
        flat_k = self.pad.scatter_params(self.flat, self.flat_k)
        y0_ptr = nat.ptr(self.y0_param) if self.learn_y0 else None
        seed = int(s.seed) & 0xFFFFFFFFFFFFFFFF
        nat.check(lib.psp_hjb_rollout_fwd(C.byref(cfg), nat.ptr(flat_k), nat.ptr(self.x0_vec), 0, y0_ptr, None, seed, 0,
                                          nat.ptr(self.path), nat.ptr(self.D), nat.ptr(self.XN_k) if self.attached else None,
                                          None, nat.ptr(self.fwd_partial), st), 'psp_hjb_rollout_fwd')
        nat.check(lib.psp_hjb_terminal_reduce_loss(C.byref(cfg), nat.ptr(self.fwd_partial), nat.ptr(self.sums),
                                                   nat.ptr(loss_out), state, st), 'psp_hjb_terminal_reduce_loss')
        d_or_w, bcfg = self.D, cfg
        if self.attached:
            # trajectory weights of the adjoint sweep (plain iteration: same lines), sweep, backward with unit weights (w_bwd = 1)
            if self.relent:
                self.mu.zero_()
                self.nu.fill_(1.0 / float(s.K))
            else:
                self.mu.copy_(sharding.loss_weights(self.D, self.sums, s.K, s.loss_method))
            nat.check(lib.psp_hjb_adjoint_sweep(C.byref(cfg), nat.ptr(flat_k), nat.ptr(self.path), nat.ptr(self.XN_k),
                                                nat.ptr(self.mu), nat.ptr(self.nu) if self.relent else None, None,
                                                nat.ptr(self.fwd_partial), st), 'psp_hjb_adjoint_sweep')
            d_or_w, bcfg = self.w_bwd, self._gcfg_w
        elif self.relent:                                # detached relative entropy: weight sqrt(dt) / K on the Z image
            d_or_w, bcfg = self.w_bwd, self._gcfg_w
        lr, b1, b2, eps = self._graph_hyper
        if self.pad.identity and not self.learn_y0:
            # backward + (gradient reduction, Adam, state advance) as two launches
            nat.check(lib.psp_hjb_rollout_bwd_step(C.byref(bcfg), nat.ptr(self.flat), nat.ptr(self.path), nat.ptr(d_or_w),
                                                   nat.ptr(self.sums), nat.ptr(self.grad_partial), nat.ptr(self.grad),
                                                   nat.ptr(self.m), nat.ptr(self.v), state, nat.ptr(self._gticket),
                                                   lr, b1, b2, eps, st), 'psp_hjb_rollout_bwd_step')
            return
        nat.check(lib.psp_hjb_rollout_bwd(C.byref(bcfg), nat.ptr(flat_k), None, seed, 0, nat.ptr(self.path),
                                          nat.ptr(d_or_w), nat.ptr(self.sums), nat.ptr(self.grad_partial),
                                          nat.ptr(self.grad_k), st), 'psp_hjb_rollout_bwd')
        self.pad.gather_grad(self.grad_k, self.grad)
        nat.check(lib.psp_adam_step_dev(nat.ptr(self.flat), nat.ptr(self.grad), nat.ptr(self.m), nat.ptr(self.v), self.P,
                                        state, lr, b1, b2, eps, st), 'psp_adam_step_dev')
        if self.learn_y0:
            self.y0_grad[0] = sharding.y0_gradient(self.sums, s.K, s.loss_method)
            ylr = self._adam_hyper(s.y_0)[0]
            nat.check(lib.psp_adam_step_dev(nat.ptr(self.y0_param), nat.ptr(self.y0_grad), nat.ptr(self.y0_m),
                                            nat.ptr(self.y0_v), 1, state, ylr, b1, b2, eps, st), 'psp_adam_step_dev(Y_0)')
        nat.check(lib.psp_iter_state_advance(state, b1, b2, st), 'psp_iter_state_advance')

    def _iteration_graph(self, l, loss_out):
        hyper = self._adam_hyper()
        if self.learn_y0 and self._adam_hyper(self.s.y_0)[1:] != hyper[1:]:
            self.s.use_graph = False                     # different betas for Y_0: one psp_iter_state cannot serve both
            return self.iteration(l, loss_out)
        if self._eager_done < 1:
            # the first iteration runs eagerly: kernels are loaded and every buffer exists before anything is captured
            self._eager_done += 1
            ev, self.events = self.events, []
            try:
                return self.iteration(l, loss_out)
            finally:
                self.events = ev
        key = (loss_out.data_ptr(), hyper)
        if self._graph is None or self._graph_key != key:
            self._graph_hyper = hyper
            self._gstate = torch.zeros(24, dtype=torch.uint8, device=self.dev)
            self._gticket = torch.zeros(1, dtype=torch.int32, device=self.dev)
            self._gcfg = nat.HjbConfig.from_buffer_copy(self.cfg)
            self._gcfg.iter_dev = nat.ptr(self._gstate)
            if self.relent or self.attached:
                self._gcfg_w = nat.HjbConfig.from_buffer_copy(self._gcfg)
                self._gcfg_w.loss_kind = nat.LOSS_WEIGHTS
                self.w_bwd.fill_(1.0 if self.attached else float(self.cfg.sqrt_dt) / float(self.s.K))
            self._graph_state_upload(l)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._graph_body(loss_out)
            self._graph, self._graph_key = g, key
        if self._graph_iter != l:                        # the caller jumped to another iteration index
            self._graph_state_upload(l)
        self._graph.replay()
        self._graph_iter += 1
        self.step += 1
        self.graph_active = True
        return loss_out[l]

    def _ul2_from_path(self):
        """u_L2 log (solver.py:491-494) for a reference control that is linear in x (LQGC) or tabulated per coordinate (the double
        wells): sum_k sum_n |-Z_n - u*(X_{n+1}, t_n)|^2 dt / K over this
        rank's trajectories, from the register images the forward kernel left in the path store -- block (n, tile): X_n image at
        float 0, h2 image behind the h1 image; image float ks * 64 + 16 q + j holds feature 4 ks + q of sample j (csrc/hjb_kernels.h
        Geo::pX / pH2).  Z_n = W3 h2 + b3 with the net's own parameters.  A diagnostic: K N small products in torch, skipped
        with u_l2_error_flag=False (as the timed runs do)."""
        s = self.s
        N, nt = s.N, (self.K_local + 15) // 16
        PB = self.sizes.path_bytes // 4 // (N * nt)
        DBp, HBp = (self.d_pad + 15) // 16, (self.H_pad + 15) // 16
        blocks = self.path.view(N, nt, PB)

        def image(off, nb):
            img = blocks[:, :, off:off + 4 * nb * 64].reshape(N, nt, 4 * nb, 4, 16)
            return img.permute(0, 1, 4, 2, 3).reshape(N, nt * 16, 16 * nb)[:, :self.K_local]
        # the reference compares -Z_n(X_n) with u*(X_{n+1}, t_n): the log line sits AFTER the Euler step (solver.py:471-472, 491-494)
        X = torch.cat([image(0, DBp)[1:, :, :s.d], self.XN_k[:, :s.d].unsqueeze(0)], 0)
        h2 = image(4 * DBp * 64 + 4 * HBp * 64, HBp)[:, :, :self.H]
        lin = self.net.linears[-1]
        with torch.no_grad():
            Z = h2 @ lin.weight.t() + lin.bias                                  # (N, K, d)
            if self.ul2_tab is not None:
                # the double wells: u*_i = table_{g(i)}[ceil(t_n / dt_ref), cell(x_i)], cell = floor((clamp(x) + xb) / dx) in fp32 as the
                # reference computes it -- which also lowers the index of the LAST trajectory of the batch by two (problems.py:270)
                tb = self.ul2_tab
                cell = torch.floor((torch.clamp(X, -tb['xb'], tb['xb'] - 2 * tb['dx']) + tb['xb']) / tb['dx']).long()
                if self.k_offset + self.K_local == s.K:
                    cell[:, -1, :] -= 2
                tot = torch.zeros((), dtype=torch.float32, device=self.dev)
                for n in range(N):                                              # (per step: bounds the index tensors)
                    u_ref = torch.empty(self.K_local, s.d, device=self.dev)
                    for gi, tab in enumerate(tb['tables']):
                        dims = torch.nonzero(tb['group'] == gi).flatten()
                        if dims.numel():
                            u_ref[:, dims] = tab[tb['n_ref'][n]][cell[n][:, dims]]
                    tot = tot + ((-Z[n] - u_ref) ** 2).sum()
                return tot * s.delta_t / float(s.K)
            u_ref = torch.bmm(X, self.ul2_gain.transpose(1, 2))                 # M_n X_n
            return ((-Z - u_ref) ** 2).sum() * s.delta_t / float(s.K)

    def _finish_step(self, st, w_generic):
        """Gather the real gradient entries, all-reduce, Adam on the net and on the learnable Y_0."""
        s, lib = self.s, self.lib
        self.pad.gather_grad(self.grad_k, self.grad)    # real entries of the (possibly padded) gradient
        sharding.allreduce_sum_(self.grad)              # collective 2: p floats
        self.step += 1
        lr, b1, b2, eps = self._adam_hyper()
        nat.check(lib.psp_adam_step(nat.ptr(self.flat), nat.ptr(self.grad), nat.ptr(self.m), nat.ptr(self.v),
                                    self.P, self.step, lr, b1, b2, eps, st), 'psp_adam_step')
        if self.learn_y0:
            # dL/dY_0 = sum_k dL/dY_k ; log-variance: exactly 0 ; moment: (2/K) sum D (global sums) ; variance: sum of the weights
            self.y0_grad[0] = sharding.y0_gradient(self.sums, s.K, s.loss_method, w_generic)
            ylr, yb1, yb2, yeps = self._adam_hyper(s.y_0)
            nat.check(lib.psp_adam_step(nat.ptr(self.y0_param), nat.ptr(self.y0_grad), nat.ptr(self.y0_m),
                                        nat.ptr(self.y0_v), 1, self.step, ylr, yb1, yb2, yeps, st),
                      'psp_adam_step(Y_0)')

    def range_fallbacks(self):
        """Iterations (chunk launches) the range guard sent to the fp32-MFMA kernels so far; one device read."""
        return int(self.range_flag[1].item()) if self.range_flag is not None else 0

    def export_optimizer_state(self):
        """Called by Solver._train_native when training returns: the nets' own optimisers see the moments this plan kept."""
        sharding.adam_state_export(self.net.flat_layout(), self.m, self.v, self.step, getattr(self.net, 'optim', None))
        if self.learn_y0:
            sharding.adam_state_export([self.y0_param], self.y0_m, self.y0_v, self.step, getattr(self.s.y_0, 'optim', None))

    def _adam_hyper(self, net=None):
        """lr, betas, eps of the net's OWN optimiser (function_space.py:185: each ansatz space builds its Adam in its
        constructor, solver.py:198-200 steps every Phi's) -- a swapped-in net keeps its learning rate."""
        opt = getattr(net if net is not None else self.net, 'optim', None)
        if opt is not None and len(opt.param_groups) > 0:
            g = opt.param_groups[0]
            if g.get('weight_decay', 0) or g.get('amsgrad', False):
                raise PlanUnsupported('the native Adam implements weight_decay = 0, amsgrad = False (the reference default)')
            b = g.get('betas', (0.9, 0.999))
            return float(g['lr']), float(b[0]), float(b[1]), float(g.get('eps', 1e-8))
        return float(self.s.lr), 0.9, 0.999, 1e-8

    def _reference_noise(self):
        """The reference's per-iteration draws from the CPU generator (solver.py:367,381),
        re-laid-out as (N+1, K_local, d) for coalesced per-step reads."""
        s = self.s
        x0 = None
        if s.random_X_0:
            x0 = torch.randn(s.K, s.d)
        xi = torch.randn(s.K, s.d, s.N + 1)
        lo, hi = self.k_offset, self.k_offset + self.K_local
        xi_dev = self.pad.last_dim(xi[lo:hi].permute(2, 0, 1).contiguous().to(self.dev))
        x0_dev = self.pad.last_dim(x0[lo:hi].contiguous().to(self.dev)) if x0 is not None else None
        return xi_dev, x0_dev

    def iteration(self, l, loss_out, ul2_out=None):
        """One training iteration; writes the fp32 loss into loss_out[l] (and mean u_L2 into ul2_out[l]) on the
        device, no sync."""
        if self.n_chunks > 1:
            return self._iteration_chunked(l, loss_out, ul2_out)
        if self._graph_wanted() and self.events is None:
            return self._iteration_graph(l, loss_out)
        s, lib, cfg = self.s, self.lib, self.cfg
        st = self._stream()
        seed = int(s.seed) & 0xFFFFFFFFFFFFFFFF
        xi = x0 = None
        if self.noise == 'reference':
            xi, x0 = self._reference_noise()
        elif s.random_X_0:
            g = torch.Generator(device=self.dev)
            g.manual_seed(int(s.seed) * 1000003 + l)
            x0 = self.pad.last_dim(torch.randn(s.K, s.d, generator=g, device=self.dev)[
                self.k_offset:self.k_offset + self.K_local].contiguous())
        x0_t = x0 if x0 is not None else self.x0_vec
        x0_stride = self.d_pad if x0 is not None else 0
        flat_k = self.pad.scatter_params(self.flat, self.flat_k)
        y0_ptr = nat.ptr(self.y0_param) if self.learn_y0 else None
        ev = None
        if self.events is not None:      # events go on torch's current stream = the launch stream
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        nat.check(lib.psp_hjb_rollout_fwd(C.byref(cfg), nat.ptr(flat_k), nat.ptr(x0_t), x0_stride, y0_ptr,
                                          nat.ptr(xi), seed, l, nat.ptr(self.path), nat.ptr(self.D),
                                          nat.ptr(self.XN_k) if (self.attached or self.ul2_gain is not None or self.ul2_tab is not None) else None,
                                          nat.ptr(self.Yn), nat.ptr(self.fwd_partial), st), 'psp_hjb_rollout_fwd')
        if ev is not None:
            ev[1].record()
        d_or_w, bcfg = self.D, cfg
        if self.world == 1 and not self.generic_loss:
            # one rank: the local sums are the global ones -- sums and the loss value in ONE launch (no element-wise torch kernels)
            nat.check(lib.psp_hjb_terminal_reduce_loss(C.byref(cfg), nat.ptr(self.fwd_partial), nat.ptr(self.sums),
                                                       nat.ptr(loss_out, l), None, st), 'psp_hjb_terminal_reduce_loss')
            loss = loss_out[l]
        else:
            nat.check(lib.psp_hjb_terminal_reduce(C.byref(cfg), nat.ptr(self.fwd_partial), nat.ptr(self.sums), st),
                      'psp_hjb_terminal_reduce')
            sharding.allreduce_sum_(self.sums)              # collective 1: 16 bytes
            if self.generic_loss:
                loss, d_or_w = self._generic_loss_weights()
            else:
                loss = sharding.loss_from_sums(self.sums, s.K, s.loss_method)
            loss_out[l] = loss.to(torch.float32)
        if self.ul2 is not None and ul2_out is not None:
            m = (self.ul2.sum() / float(s.K)).reshape(1)
            sharding.allreduce_sum_(m)
            ul2_out[l:l + 1] = m
        elif (self.ul2_gain is not None or self.ul2_tab is not None) and ul2_out is not None:
            m = self._ul2_from_path().reshape(1)
            sharding.allreduce_sum_(m)
            ul2_out[l:l + 1] = m
        if ev is not None:
            ev[2].record()
        if self.attached:
            # per-trajectory weights mu = dL/dY_N, nu = dL/dZsum_N (global K and global mean: rank-independent)
            wT = None
            if self.relent:
                self.mu.zero_()
                self.nu.fill_(1.0 / float(s.K))
            elif self.generic_loss:
                self.mu.copy_(d_or_w)
                if s.loss_method == 'cross_entropy':
                    # mean(Y exp(-g(X_N) + Y.detach())) (solver.py:183-185) also depends on X_N through exp(-g):
                    # lambda_N = -(Y_N exp(D) / K) grad g, while dL/dY_N = exp(D) / K
                    wT = self.wT
                    wT.copy_(-self.Yn * d_or_w)
            else:
                self.mu.copy_(sharding.loss_weights(self.D, self.sums, s.K, s.loss_method))
            nat.check(lib.psp_hjb_adjoint_sweep(C.byref(cfg), nat.ptr(flat_k), nat.ptr(self.path), nat.ptr(self.XN_k),
                                                nat.ptr(self.mu), nat.ptr(self.nu) if self.relent else None, nat.ptr(wT),
                                                nat.ptr(self.fwd_partial), st), 'psp_hjb_adjoint_sweep')
            self.w_bwd.fill_(1.0)                       # the sweep left dL/dZ_n / sqrt(dt) in the xi slot
            d_or_w, bcfg = self.w_bwd, self.cfg_w
        elif self.relent:
            # detached relative entropy: dL/dZ_n = Z_n dt / K, and the xi slot holds Z_n  ->  weight sqrt(dt) / K
            self.w_bwd.fill_(float(cfg.sqrt_dt) / float(s.K))
            d_or_w, bcfg = self.w_bwd, self.cfg_w
        nat.check(lib.psp_hjb_rollout_bwd(C.byref(bcfg), nat.ptr(flat_k), nat.ptr(xi), seed, l, nat.ptr(self.path),
                                          nat.ptr(d_or_w), nat.ptr(self.sums), nat.ptr(self.grad_partial),
                                          nat.ptr(self.grad_k), st), 'psp_hjb_rollout_bwd')
        if ev is not None:
            ev[3].record()
            self.events.append(ev)
        self._finish_step(st, self.w if self.generic_loss else None)
        # keep xi alive until the kernels that read it are enqueued on this stream (they are);
        # torch's caching allocator is stream-ordered, so freeing here is safe.
        return loss

    def _generic_loss_weights(self):
        """Losses whose weights w_k = dLoss/dY_k are not affine in D (K-vector arithmetic in torch, the
        rollout and the gradient stay in the kernels):
          variance       var(exp(-g + Y))            (solver.py:171-172, unbiased)  w = 2 (E - mean E) E / (K - 1)
          cross_entropy  mean(Y exp(-g + Y.detach()))  (adaptive, :185)             w = exp(D) / K
                         mean(Y exp(-g))               (else, :186)                 w = exp(-g) / K"""
        s, K = self.s, float(self.s.K)
        D = self.D
        if s.loss_method == 'variance':
            E = torch.exp(D)
            st = torch.stack([E.double().sum(), (E.double() ** 2).sum()])
            sharding.allreduce_sum_(st)
            mean = st[0] / K
            loss = (st[1] - K * mean * mean) / (K - 1.0)
            w = (2.0 / (K - 1.0)) * (E - mean.float()) * E
        else:
            Y = self.Yn
            E = torch.exp(D) if s.adaptive_forward_process else torch.exp(D - Y)     # exp(-g) = exp(D - Y)
            st = (Y.double() * E.double()).sum().reshape(1)
            sharding.allreduce_sum_(st)
            loss = st[0] / K
            w = E / K
        self.w.copy_(w)
        return loss, self.w

    def forward_only(self, l, want_XN=False):
        """Forward rollout without the path store; returns (D, X_N or None)."""
        s, lib = self.s, self.lib
        cfg = nat.HjbConfig.from_buffer_copy(self.cfg)
        cfg.store_path = 0
        xi = x0 = None
        if self.noise == 'reference':
            xi, x0 = self._reference_noise()
        x0_t = x0 if x0 is not None else self.x0_vec
        XN = torch.empty(self.K_local, self.d_pad, dtype=torch.float32, device=self.dev) if want_XN else None
        flat_k = self.pad.scatter_params(self.flat, self.flat_k)
        nat.check(lib.psp_hjb_rollout_fwd(C.byref(cfg), nat.ptr(flat_k), nat.ptr(x0_t), self.d_pad if x0 is not None else 0,
                                          nat.ptr(self.y0_param) if self.learn_y0 else None, nat.ptr(xi),
                                          int(s.seed), l, None, nat.ptr(self.D), nat.ptr(XN), None,
                                          nat.ptr(self.fwd_partial), self._stream()), 'psp_hjb_rollout_fwd')
        return self.D, (XN[:, :s.d] if XN is not None else None)
