"""Native execution plan for Solver.train with a DenseNet control (reference function_space.py:116-140):

  * time_approx='outer' -- the reference's constructor default (solver.py:88): N nets DenseNet(d -> d), one per
    time step, each with its own Adam (solver.py:142-162, 194-200);
  * time_approx='inner' with a DenseNet(d+1 -> d) swapped into z_n (notebook extension point).

Per iteration (reference solver.py:430-514):
    psp_dnet_rollout_fwd     hand-written HIP rollout (csrc/hjbd_kernels.h): D_k, the terminal sums and the two flat
                             batches the gradient needs, X_n and the xi image, each (N, K, d) row-major
    psp_dnet_terminal_reduce (sum D, sum D^2)                      [all-reduce 1]
    parameter gradient       detach_forward=True makes dL/dZ_n[k] = w_k sqrt(dt) image_n[k], so the gradient
                             of every net is that of a plain feed-forward batch: the stored relu activations and ~12 library
                             GEMMs (batched over the time steps; 'inner' shares one net across the batch), written
                             analytically below -- no autograd graph                    [all-reduce 2]
    psp_adam_step            one fused Adam over the concatenation of all parameter sets (identical to the reference's
                             per-net Adams: same lr, betas and step count)
The rollout is the sequential, launch-bound part of the reference (N steps x ~70 eager kernels) and is the hand-written
kernel; the gradient is GEMM-shaped and is left to rocBLAS.
Gradients THROUGH the state path (adaptive_forward_process=True with detach_forward=False: the reference's constructor
defaults, solver.py:23-24, 451-469) and the relative-entropy loss (:179-180, 484-486) run natively on the instances the
hand-written backward covers: the rollout leaves xi - sqrt(dt) Z (or Z) in the xi slot of the register images,
psp_dnet_adjoint_sweep (csrc/hjbd_kernels.h: hjbd_adj_kernel) walks the adjoint recursion backwards in time with the
Jacobian of the dense-concat net and overwrites the slot with dL/dZ_n / sqrt(dt), and the backward kernel runs with unit
weights -- the scheme of plan_native.py / psp_hjb_adjoint_sweep.
"""
import ctypes as C

import torch

try:
    from . import native as nat
    from . import native_shapes as shapes
    from . import sharding
    from .function_space import DenseNet
    from .plan_native import HjbNativePlan, PlanUnsupported, _overridden
except ImportError:
    import native as nat
    import native_shapes as shapes
    import sharding
    from function_space import DenseNet
    from plan_native import HjbNativePlan, PlanUnsupported, _overridden

_LOSSES = ('log-variance', 'moment', 'variance', 'cross_entropy', 'relative_entropy')


def _nets(solver):
    return list(solver.z_n) if solver.time_approx == 'outer' else [solver.z_n]


def _instance_for(d, H):
    cands = [(D, Hh) for (D, Hh) in nat.dnet_instances() if D >= d and Hh >= H]
    return min(cands, key=lambda t: (t[0] * t[0] + 3 * t[0] * t[1], t[1])) if cands else None


def dense_eligibility(solver):
    """None if the solver can run on this plan, else a human-readable reason."""
    if solver.device.type != 'cuda':
        return 'device is %s (the HIP rollout needs a GPU)' % solver.device
    if solver.approx_method != 'control':
        return "only approx_method='control' is native"
    if solver.loss_method not in _LOSSES:
        return 'loss_method %r is not native for a DenseNet control (%s are)' % (solver.loss_method, ', '.join(_LOSSES))
    if solver.loss_method == 'relative_entropy' and not solver.adaptive_forward_process:
        return 'relative_entropy with a non-adaptive forward process is not native for a DenseNet control'
    if solver.burgers_drift or solver.u_l2_error_flag or solver.compute_gradient_variance > 0 or solver.log_gradient \
            or solver.metastability_logs is not None:
        return 'per-step / per-iteration diagnostics (u_L2, gradient logs, metastability) are not native here'
    nets = _nets(solver)
    outer = solver.time_approx == 'outer'
    if outer and len(nets) != solver.N:
        return "time_approx='outer' needs one net per time step"
    d_in = solver.d + (0 if outer else 1)
    dims0 = getattr(nets[0], 'nn_dims', None)
    for net in nets:
        dims = getattr(net, 'nn_dims', None)
        if not isinstance(net, DenseNet) or dims is None or len(dims) != 4 or dims[1] != dims[2] \
                or dims[0] != d_in or dims[3] != solver.d or dims != dims0:
            return 'control is not a DenseNet(%d -> %d) with two equal hidden widths' % (d_in, solver.d)
    spec_fn = getattr(solver.problem, 'native_spec', None)
    if spec_fn is None or spec_fn() is None:
        return 'problem has no native_spec() (coefficients outside the native catalogue)'
    over = _overridden(solver.problem)
    if over is not None:
        return 'problem.%s is not the catalogue implementation native_spec() describes' % over
    if not nat.is_built():
        raise nat.NativeLibraryError('libpsp_hip.so is not built; run __graft_entry__.build()')
    if _instance_for(solver.d, dims0[1]) is None:
        return 'no compiled DenseNet-control instance covers d=%d, H=%d (csrc/dense_instances.def)' % (solver.d, dims0[1])
    return None


class DenseNativePlan:
    def __init__(self, solver, noise='reference'):
        reason = dense_eligibility(solver)
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
        self.K_local, self.k_offset = hi - lo, lo
        self.outer = s.time_approx == 'outer'
        self.nets = _nets(s)
        self.net = s.z_n                              # identity check in Solver._choose_plan
        self.B = len(self.nets)                       # parameter sets
        self.H = self.nets[0].nn_dims[1]
        self.di = s.d + (0 if self.outer else 1)
        self._flatten()
        self.d_pad, self.H_pad = _instance_for(s.d, self.H)
        self.pad = shapes.ParamPad(s.d, self.H, self.d_pad, self.H_pad, dev)      # used for the problem data / x0 / noise only
        spec = s.problem.native_spec()
        self._keep = []

        def dev_f32(t):
            if t is None:
                return None
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t

        cfg = nat.DnetConfig()
        b = cfg.base
        b.d, b.H = self.d_pad, self.H_pad
        b.K_local, b.N, b.K_global, b.k_offset = self.K_local, s.N, s.K, self.k_offset
        b.dt, b.sqrt_dt = float(s.delta_t.item()), float(s.sq_delta_t.item())
        b.drift_kind, b.sigma_kind, b.sigma_scale = spec['drift'][0], spec['sigma'][0], float(spec['sigma'][2])
        b.runcost_kind, b.term_kind = spec['runcost'][0], spec['term'][0]
        b.adaptive = 1 if s.adaptive_forward_process else 0
        b.loss_kind = {'log-variance': nat.LOSS_LOG_VARIANCE, 'moment': nat.LOSS_MOMENT,
                       'relative_entropy': nat.LOSS_REL_ENTROPY}.get(s.loss_method, nat.LOSS_WEIGHTS)
        self.generic_loss = b.loss_kind == nat.LOSS_WEIGHTS
        self.relent = s.loss_method == 'relative_entropy'
        self.attached = bool(s.adaptive_forward_process and not s.detach_forward)
        b.noise_mode = nat.NOISE_PHILOX if noise == 'philox' else nat.NOISE_SUPPLIED
        b.store_path = 3 if self.relent else (2 if self.attached else 1)
        pad = self.pad
        b.drift = nat.ptr(dev_f32(pad.drift_or_sigma(spec['drift'][1]))) if spec['drift'][1] is not None else None
        b.sigma = nat.ptr(dev_f32(pad.drift_or_sigma(spec['sigma'][1]))) if spec['sigma'][1] is not None else None
        b.runcost = nat.ptr(dev_f32(pad.vec(spec['runcost'][1]))) if spec['runcost'][1] is not None else None
        b.term = nat.ptr(dev_f32(pad.vec(spec['term'][1])))
        cfg.d_real, cfg.H_real = s.d, self.H
        cfg.time_input, cfg.per_step = (0 if self.outer else 1), (1 if self.outer else 0)
        self.cfg = cfg
        sizes = nat.DnetSizes()
        # matrix products of the forward rollout: 'f16x3' = fp32-grade split products on the f16 matrix pipe (csrc/hjbd_kernels.h
        # hjbd_fwd_kernel<.., X3>; same parity bounds); 'auto' (the default) takes it where its images fit the LDS
        want = getattr(s, 'mlp_dtype', 'auto')
        if want == 'bf16':
            raise PlanUnsupported("mlp_dtype='bf16' exists for MySequential controls only")
        b.mlp_dtype = nat.MLP_F16X3 if want in ('auto', 'f16x3') else nat.MLP_FP32
        if self.lib.psp_dnet_query(C.byref(cfg), C.byref(sizes)) != 0:
            if want == 'f16x3':
                nat.check(self.lib.psp_dnet_query(C.byref(cfg), C.byref(sizes)), 'psp_dnet_query')
            b.mlp_dtype = nat.MLP_FP32
        self.matrix_mode = 'f16x3' if b.mlp_dtype == nat.MLP_F16X3 else 'fp32'
        # range guard of the split-product rollout (include/psp.h: psp_hjb_config.range_flag)
        self.range_flag = None
        if b.mlp_dtype == nat.MLP_F16X3 and getattr(s, 'range_guard', True):
            self.range_flag = torch.zeros(4, dtype=torch.int32, device=dev)
            b.range_flag = nat.ptr(self.range_flag)
        nat.check(self.lib.psp_dnet_query(C.byref(cfg), C.byref(sizes)), 'psp_dnet_query')
        assert sizes.n_params_per_set == self.Pset, (sizes.n_params_per_set, self.Pset)
        f32 = torch.float32
        self.tables = torch.empty(sizes.table_bytes // 4, dtype=f32, device=dev)
        self.fwd_partial = torch.empty(sizes.fwd_partial_bytes // 8, dtype=torch.float64, device=dev)
        # gradient: the hand-written kernel (csrc/hjbd_kernels.h, hjbd_bwd_kernel) where the instance is covered, else the
        # library-GEMM formulation on row-major stores (PSP_DENSE_BWD=gemm forces the latter: cross-check / timing)
        import os
        self.kernel_bwd = bool(sizes.bwd_supported) and os.environ.get('PSP_DENSE_BWD', 'kernel') != 'gemm'
        self._gidx = None                                        # padded -> real gather index of a parameter set ('outer')
        if (self.attached or self.relent) and not self.kernel_bwd:
            raise PlanUnsupported('the adjoint sweep of a DenseNet control works on the register images of the hand-written '
                                  'backward, which does not cover the (%d, %d) instance' % (self.d_pad, self.H_pad))
        if self.attached or self.relent:
            self.w_bwd = torch.empty(self.K_local, dtype=f32, device=dev)
        if self.attached:
            self.XN_k = torch.empty(self.K_local, self.d_pad, dtype=f32, device=dev)
            self.mu = torch.zeros(self.K_local, dtype=f32, device=dev)
            self.nu = torch.zeros(self.K_local, dtype=f32, device=dev)
            self.wT = torch.zeros(self.K_local, dtype=f32, device=dev)
        if self.kernel_bwd:
            self.images = torch.empty(sizes.image_bytes // 4, dtype=f32, device=dev)
            self.partial = torch.zeros(s.N * sizes.slices, sizes.padded_params, dtype=f32, device=dev)
            self.wpad = torch.zeros(16 * ((self.K_local + 15) // 16), dtype=f32, device=dev)
            self.slices, self.PP = int(sizes.slices), int(sizes.padded_params)
            cfg.images_out = nat.ptr(self.images)
            self.PX = self.PXI = self.PR1 = self.PR2 = None
        else:
            self.PX = torch.empty(s.N, self.K_local, s.d, dtype=f32, device=dev)
            self.PXI = torch.empty(s.N, self.K_local, s.d, dtype=f32, device=dev)
            self.PR1 = torch.empty(s.N, self.K_local, self.H, dtype=f32, device=dev)   # relu(z1), relu(z2) from the rollout
            self.PR2 = torch.empty(s.N, self.K_local, self.H, dtype=f32, device=dev)
            cfg.r1_out, cfg.r2_out = nat.ptr(self.PR1), nat.ptr(self.PR2)
        self.D = torch.empty(self.K_local, dtype=f32, device=dev)
        self.Yn = torch.empty(self.K_local, dtype=f32, device=dev) if self.generic_loss else None
        self.w = torch.empty(self.K_local, dtype=f32, device=dev) if self.generic_loss else None
        self.sums = torch.zeros(2, dtype=torch.float64, device=dev)
        self.grad = torch.zeros(self.P, dtype=f32, device=dev)
        self.m = torch.zeros(self.P, dtype=f32, device=dev)
        self.v = torch.zeros(self.P, dtype=f32, device=dev)
        self.x0_vec = dev_f32(pad.vec(s.X_0.detach().to(dev)))
        self.tn = (torch.arange(s.N, device=dev, dtype=f32) * b.dt)        # fp32 n * dt, as the kernel forms it
        self.step = 0
        self.events = None
        self.ul2 = None
        self.learn_y0 = bool(s.learn_Y_0)
        if self.learn_y0:
            self.y0_param = s.y_0.Y_0
            self.y0_m = torch.zeros(1, dtype=f32, device=dev)
            self.y0_v = torch.zeros(1, dtype=f32, device=dev)
            self.y0_grad = torch.zeros(1, dtype=f32, device=dev)
        # sample chunk of the gradient GEMMs: bounds their temporaries (~(2H + 2d) floats per sample)
        self.chunk = max(1, (1 << 21) // max(1, self.K_local)) if not self.outer else s.N

    # ------------------------------------------------------------------------------------
    def _flatten(self):
        """All parameter sets as views of ONE flat fp32 buffer [set 0 | set 1 | ...] (include/psp.h layout per set):
        state_dict / save_networks / Z_n keep working on live values, the fused Adam sees one vector."""
        per = [list(net.W) for net in self.nets]                     # registration order W1,b1,W2,b2,W3,b3
        self.Pset = sum(p.numel() for p in per[0])
        self.P = self.Pset * len(per)
        flat = torch.empty(self.P, dtype=torch.float32, device=self.dev)
        off = 0
        for params in per:
            for p in params:
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
                off += n
        self.flat = flat
        di, H, d = self.di, self.H, self.nets[0].nn_dims[3]
        o, self.off = 0, {}
        for name, shape in (('W1', (di, H)), ('b1', (H,)), ('W2', (di + H, H)), ('b2', (H,)),
                            ('W3', (di + 2 * H, d)), ('b3', (d,))):
            n = 1
            for v in shape:
                n *= v
            self.off[name] = (o, n, shape)
            o += n

    def _set_view(self, buf, name):
        """(B, *shape) strided view of one tensor of every parameter set inside a flat (B * Pset) buffer."""
        o, n, shape = self.off[name]
        return buf.view(self.B, self.Pset)[:, o:o + n].reshape(self.B, *shape) if self.B == 1 else \
            buf.view(self.B, self.Pset)[:, o:o + n].unflatten(1, shape)

    _reference_noise = HjbNativePlan._reference_noise
    _generic_loss_weights = HjbNativePlan._generic_loss_weights

    # ------------------------------------------------------------------------------------
    def _gradient_kernel(self, w):
        """hjbd_bwd_kernel: per (step, slice) partial gradients in the instance's padded layout -> real shapes."""
        s, H, di, d, N = self.s, self.H, self.di, self.s.d, self.s.N
        D, Hp = self.d_pad, self.H_pad
        self.wpad[:self.K_local].copy_(w)
        nat.check(self.lib.psp_dnet_rollout_bwd(C.byref(self.cfg), nat.ptr(self.flat), nat.ptr(self.images), nat.ptr(self.wpad),
                                                nat.ptr(self.partial), nat.stream_ptr(self.dev)), 'psp_dnet_rollout_bwd')
        g = self.partial.view(N, self.slices, self.PP).sum(1)          # (N, PP): one padded gradient per time step
        if self.outer:
            # one parameter set per time step: the real rows / columns of every set are ONE gather through an index built once
            # (round 3 cut them out with a dozen strided slices and three torch.cat per iteration -- 35 device copies)
            if self._gidx is None:
                self._gidx = self._gather_index()
            torch.index_select(g, 1, self._gidx, out=self.grad.view(N, self.Pset))
            return self.grad
        o = 0
        W1 = g[:, o:o + D * Hp].view(N, D, Hp)[:, :d, :H]; o += D * Hp
        b1 = g[:, o:o + Hp][:, :H]; o += Hp
        W2 = g[:, o:o + (D + Hp) * Hp].view(N, D + Hp, Hp); o += (D + Hp) * Hp
        b2 = g[:, o:o + Hp][:, :H]; o += Hp
        W3 = g[:, o:o + (D + 2 * Hp) * D].view(N, D + 2 * Hp, D); o += (D + 2 * Hp) * D
        b3 = g[:, o:o + D][:, :d]
        W2r = torch.cat([W2[:, :d, :H], W2[:, D:D + H, :H]], 1)                                  # rows [x | h1]
        W3r = torch.cat([W3[:, :d, :d], W3[:, D:D + H, :d], W3[:, D + Hp:D + Hp + H, :d]], 1)    # rows [x | h1 | h2]
        grad = self.grad
        if self.outer:
            out = torch.cat([W1.reshape(N, -1), b1, W2r.reshape(N, -1), b2, W3r.reshape(N, -1), b3], 1)       # per-step sets
            grad.copy_(out.reshape(-1))
        else:                                                     # one net: sum over the steps; the time input is row 0 of
            t = self.tn.view(N, 1)                                # every layer's input block: its gradient is sum_n t_n db_n
            W1s = torch.cat([(t * b1).sum(0, keepdim=True), W1.sum(0)], 0)
            W2s = torch.cat([(t * b2).sum(0, keepdim=True), W2r.sum(0)], 0)
            W3s = torch.cat([(t * b3).sum(0, keepdim=True), W3r.sum(0)], 0)
            grad.copy_(torch.cat([W1s.reshape(-1), b1.sum(0), W2s.reshape(-1), b2.sum(0), W3s.reshape(-1), b3.sum(0)]))
        return grad

    def _gather_index(self):
        """Padded position (inside one per-step partial gradient of hjbd_bwd_kernel) of every real parameter of a set, in the
        DenseNet's registration order W1, b1, W2, b2, W3, b3 (weights (in, out); W2 rows [x | h1], W3 rows [x | h1 | h2])."""
        H, d, D, Hp = self.H, self.s.d, self.d_pad, self.H_pad
        ar = lambda n: torch.arange(n, device=self.dev)
        o, parts = 0, []
        parts.append((o + ar(d).view(-1, 1) * Hp + ar(H).view(1, -1)).reshape(-1)); o += D * Hp
        parts.append(o + ar(H)); o += Hp
        rows2 = torch.cat([ar(d), D + ar(H)])
        parts.append((o + rows2.view(-1, 1) * Hp + ar(H).view(1, -1)).reshape(-1)); o += (D + Hp) * Hp
        parts.append(o + ar(H)); o += Hp
        rows3 = torch.cat([ar(d), D + ar(H), D + Hp + ar(H)])
        parts.append((o + rows3.view(-1, 1) * D + ar(d).view(1, -1)).reshape(-1)); o += (D + 2 * Hp) * D
        parts.append(o + ar(d))
        idx = torch.cat(parts)
        assert idx.numel() == self.Pset, (idx.numel(), self.Pset)
        return idx

    def _gradient(self, w):
        if self.kernel_bwd:
            return self._gradient_kernel(w)
        return self._gradient_gemm(w)

    def _gradient_gemm(self, w):
        """grad of sum_{n,k} G_n[k] . Z_n(X_n[k]) over all parameter sets, G = w_k sqrt(dt) image (row-major stores)."""
        s, H, di, d = self.s, self.H, self.di, self.s.d
        N, K = s.N, self.K_local
        grad = self.grad
        grad.zero_()
        W2, W3 = self._set_view(self.flat, 'W2'), self._set_view(self.flat, 'W3')
        gW1, gb1, gW2, gb2, gW3, gb3 = (self._set_view(grad, k) for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3'))
        # row blocks of the dense-concat weights: [input | h1 | h2]; every product below works on one block, so the
        # concatenated activations [u, h1, h2] are never materialised (the torch.cat copies were a quarter of this function)
        W2h = W2[:, di:, :]
        W3h1T, W3h2T = W3[:, di:di + H, :].transpose(1, 2), W3[:, di + H:, :].transpose(1, 2)     # (B, d, H)
        W2hT = W2h.transpose(1, 2)
        scale = (w * float(self.cfg.base.sqrt_dt)).view(1, K, 1)
        for n0 in range(0, N, self.chunk):
            n1 = min(N, n0 + self.chunk)
            X = self.PX[n0:n1]
            G = self.PXI[n0:n1] * scale                           # (n, K, d)
            # batch = time steps in both modes: 'outer' pairs step n with its own net, 'inner' broadcasts the one net over
            # the steps and sums the per-step weight gradients (rocBLAS handles n reductions over K rows far better than
            # one tall-skinny reduction over n K rows)
            if self.outer:
                U, sl = X, slice(n0, n1)
            else:
                U, sl = torch.cat([self.tn[n0:n1].view(-1, 1, 1).expand(n1 - n0, K, 1), X], 2), slice(0, 1)
            red = (lambda t: t) if self.outer else (lambda t: t.sum(0, keepdim=True))
            UT = U.transpose(1, 2)
            r1, r2 = self.PR1[n0:n1], self.PR2[n0:n1]             # stored by the rollout kernel (no recomputed forward)
            h1, h2 = r1 * r1, r2 * r2
            gW3[sl][:, :di] += red(torch.matmul(UT, G))
            gW3[sl][:, di:di + H] += red(torch.matmul(h1.transpose(1, 2), G))
            gW3[sl][:, di + H:] += red(torch.matmul(h2.transpose(1, 2), G))
            gb3[sl] += red(G.sum(1))
            dz2 = torch.matmul(G, W3h2T[sl]) * (2.0 * r2)
            gW2[sl][:, :di] += red(torch.matmul(UT, dz2))
            gW2[sl][:, di:] += red(torch.matmul(h1.transpose(1, 2), dz2))
            gb2[sl] += red(dz2.sum(1))
            dz1 = (torch.matmul(G, W3h1T[sl]) + torch.matmul(dz2, W2hT[sl])) * (2.0 * r1)
            gW1[sl] += red(torch.matmul(UT, dz1))
            gb1[sl] += red(dz1.sum(1))
        return grad

    def iteration(self, l, loss_out, ul2_out=None):
        s, lib, cfg = self.s, self.lib, self.cfg
        st = nat.stream_ptr(self.dev)
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
        ev = None
        if self.events is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        nat.check(lib.psp_dnet_rollout_fwd(C.byref(cfg), nat.ptr(self.flat), nat.ptr(x0_t), self.d_pad if x0 is not None else 0,
                                           nat.ptr(self.y0_param) if self.learn_y0 else None, nat.ptr(xi), seed, l, None,
                                           nat.ptr(self.PX), nat.ptr(self.PXI), nat.ptr(self.D), None,
                                           nat.ptr(self.XN_k) if self.attached else None,
                                           nat.ptr(self.Yn), nat.ptr(self.fwd_partial), nat.ptr(self.tables), st),
                  'psp_dnet_rollout_fwd')
        if ev is not None:
            ev[1].record()
        nat.check(lib.psp_dnet_terminal_reduce(C.byref(cfg), nat.ptr(self.fwd_partial), nat.ptr(self.sums), st),
                  'psp_dnet_terminal_reduce')
        sharding.allreduce_sum_(self.sums)                        # collective 1: 16 bytes
        if self.generic_loss:
            loss, w = self._generic_loss_weights()
        else:
            loss = sharding.loss_from_sums(self.sums, s.K, s.loss_method)
            w = None if self.relent else sharding.loss_weights(self.D, self.sums, s.K, s.loss_method)
        loss_out[l] = loss.to(torch.float32)
        if ev is not None:
            ev[2].record()
        if self.attached:
            # per-trajectory weights mu = dL/dY_N, nu = dL/dZsum_N (global K and global mean: rank-independent), as in
            # plan_native.py; the sweep leaves dL/dZ_n / sqrt(dt) in the xi slot of the images
            wT = None
            if self.relent:
                self.mu.zero_()
                self.nu.fill_(1.0 / float(s.K))
            else:
                self.mu.copy_(w)
                if s.loss_method == 'cross_entropy':
                    # mean(Y exp(-g(X_N) + Y.detach())) (solver.py:183-185) also depends on X_N through exp(-g)
                    wT = self.wT
                    wT.copy_(-self.Yn * w)
            nat.check(lib.psp_dnet_adjoint_sweep(C.byref(cfg), nat.ptr(self.flat), nat.ptr(self.images), nat.ptr(self.XN_k),
                                                 nat.ptr(self.mu), nat.ptr(self.nu) if self.relent else None, nat.ptr(wT),
                                                 nat.ptr(self.tables), st), 'psp_dnet_adjoint_sweep')
            self.w_bwd.fill_(1.0)
            w = self.w_bwd
        elif self.relent:
            # detached relative entropy: dL/dZ_n = Z_n dt / K, and the xi slot holds Z_n  ->  weight sqrt(dt) / K
            self.w_bwd.fill_(float(cfg.base.sqrt_dt) / float(s.K))
            w = self.w_bwd
        self._gradient(w)
        if ev is not None:
            ev[3].record()
            self.events.append(ev)
        sharding.allreduce_sum_(self.grad)                        # collective 2
        self.step += 1
        lr, b1, b2, eps = self._adam_hyper(self.nets)
        nat.check(lib.psp_adam_step(nat.ptr(self.flat), nat.ptr(self.grad), nat.ptr(self.m), nat.ptr(self.v),
                                    self.P, self.step, lr, b1, b2, eps, st), 'psp_adam_step')
        if self.learn_y0:
            self.y0_grad[0] = sharding.y0_gradient(self.sums, s.K, s.loss_method, self.w if self.generic_loss else None)
            ylr, yb1, yb2, yeps = self._adam_hyper([s.y_0])
            nat.check(lib.psp_adam_step(nat.ptr(self.y0_param), nat.ptr(self.y0_grad), nat.ptr(self.y0_m),
                                        nat.ptr(self.y0_v), 1, self.step, ylr, yb1, yb2, yeps, st),
                      'psp_adam_step(Y_0)')
        return loss

    def range_fallbacks(self):
        """Iterations the range guard sent to the fp32-MFMA rollout so far; one device read."""
        return int(self.range_flag[1].item()) if self.range_flag is not None else 0

    def _adam_hyper(self, nets):
        """lr / betas / eps of the nets' OWN optimisers (function_space.py:131; solver.py:198-200 steps every Phi's Adam).  One
        fused Adam runs over the concatenation of all parameter sets, so the sets must agree."""
        hyp = None
        for net in nets:
            opt = getattr(net, 'optim', None)
            if opt is None or not opt.param_groups:
                h = (float(self.s.lr), 0.9, 0.999, 1e-8)
            else:
                g = opt.param_groups[0]
                if g.get('weight_decay', 0) or g.get('amsgrad', False):
                    raise PlanUnsupported('the native Adam implements weight_decay = 0, amsgrad = False (the reference default)')
                b = g.get('betas', (0.9, 0.999))
                h = (float(g['lr']), float(b[0]), float(b[1]), float(g.get('eps', 1e-8)))
            if hyp is not None and h != hyp:
                raise PlanUnsupported('the per-step nets carry different Adam settings; the fused native Adam needs one')
            hyp = h
        return hyp
