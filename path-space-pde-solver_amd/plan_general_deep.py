"""Native plan of GeneralSolver / EllipticSolver for value nets of ANY depth: V = dense-concat net (d [+ 1] -> 1,
arch = [H_1 .. H_L]), 1 <= L <= 4, H_i <= 128 -- the nets the reference's diffusion-loss notebooks train (Allen-Cahn.ipynb:72
arch = [110, 110, 50], the only Allen-Cahn configuration with a published timing; [30, 30, 30, 30]; `Committor function.ipynb`:
[d + 10, d, d, d] with tanh(.)**2; reference function_space.py:116-140 DenseNet, :143-158 DenseNet_tanh).

Same iteration as plan_general_native.GeneralNativePlan (host RNG in the reference's order, K-vector loss weights, the
K_boundary-sized terms by autograd, Adam through psp_adam_step); what differs are the two device steps:
    psp_genl_rollout_fwd   the rollout (csrc/genl_kernels.h: activations in per-tile LDS images, weight tables in L2,
                           rolled fp32-MFMA products; shapes and activation are run-time arguments, nothing is padded on the
                           host; a tile whose trajectories have all stopped leaves the time loop)
    psp_genl_rollout_bwd   ONE hand-written kernel over the executed sample blocks: activations and tangents recomputed, adjoint
                           sweep, every weight gradient as MFMA outer products over the samples of the block (no library GEMM)
The (d, H)-templated kernels of gen_kernels.h stay the fast path for DenseNet arch = [H, H], H <= 64.

Which nets: ``value_net_spec`` recognises the package's DenseNet (any ``activation``), DenseNet_tanh (nn.Linear weights) and --
by structure plus a numerical probe of its forward -- any module that carries ``nn_dims`` and a list ``W`` of (in, out) weights
and biases in the dense-concat layout with one of the three activations: the class a notebook defines for itself
(`Committor function.ipynb` cell 1) runs on the kernels without the package knowing its name.
"""
import ctypes as C

import torch

try:
    from . import native as nat
    from . import sharding
    from .function_space import DenseNet, DenseNet_tanh
    from .plan_general_native import GeneralNativePlan, set_domain
except ImportError:
    import native as nat
    import sharding
    from function_space import DenseNet, DenseNet_tanh
    from plan_general_native import GeneralNativePlan, set_domain

_ACT = {'relu2': nat.ACT_RELU2, 'tanh2': nat.ACT_TANH2, 'tanh': nat.ACT_TANH}


class _IdentityPad:
    """The genl kernels take the real shapes: nothing is padded on the host."""
    identity = True

    def __init__(self, P):
        self.Pp = P

    def last_dim(self, t):
        return t.contiguous()

    def vec(self, t):
        return t

    def scatter_params(self, flat, flat_pad):
        return flat

    def gather_grad(self, grad_pad, out):
        return out


def _dense_concat_forward(x, params, act):
    """The dense-concat forward on (in, out) weights: the formula the kernels implement."""
    n = len(params) // 2
    for i in range(n - 1):
        z = torch.matmul(x, params[2 * i]) + params[2 * i + 1]
        h = torch.relu(z) ** 2 if act == 'relu2' else (torch.tanh(z) ** 2 if act == 'tanh2' else torch.tanh(z))
        x = torch.cat([x, h], 1)
    return torch.matmul(x, params[2 * n - 2]) + params[2 * n - 1]


def value_net_spec(V, d_in):
    """dict(dims, act, linear, params) if V is a dense-concat net (d_in -> 1) the genl kernels implement, else a reason string."""
    dims = getattr(V, 'nn_dims', None)
    if dims is None or len(dims) < 3 or dims[0] != d_in or dims[-1] != 1:
        return 'V is not a dense-concat net (%d -> 1) (no matching nn_dims)' % d_in
    dims = [int(v) for v in dims]
    if isinstance(V, DenseNet):
        return dict(dims=dims, act=getattr(V, 'activation', 'relu2'), linear=False, params=list(V.W))
    if isinstance(V, DenseNet_tanh):
        params = []
        for layer in V.layers:
            params += [layer.weight, layer.bias]
        return dict(dims=dims, act='tanh', linear=True, params=params)
    # a user-defined module of the same structure (the notebooks define their own variants): W = [W_1, b_1, .., W_out, b_out] with
    # (in, out) weights over the growing concatenation, registered in that order and nothing else -- and a forward that IS one
    # of the three formulas, checked on a probe batch (a private generator: the global RNG stream is the reference's)
    W = getattr(V, 'W', None)
    if not isinstance(W, (list, tuple)) or len(W) != 2 * (len(dims) - 1):
        return 'V carries no dense-concat parameter list W'
    fan = 0
    for i in range(len(dims) - 1):
        fan += dims[i]
        if tuple(W[2 * i].shape) != (fan, dims[i + 1]) or tuple(W[2 * i + 1].shape) != (dims[i + 1],):
            return 'V.W does not have the dense-concat shapes'
    regs = list(V.parameters())
    if len(regs) != len(W) or any(a is not b for a, b in zip(regs, W)):
        return 'V has parameters besides V.W'
    g = torch.Generator().manual_seed(1234)
    probe = torch.randn(16, d_in, generator=g).to(W[0].device)
    with torch.no_grad():
        try:
            want = V(probe)
        except Exception as e:                                   # pragma: no cover
            return 'V could not be evaluated on a probe batch (%s)' % e
        for act in ('relu2', 'tanh2', 'tanh'):
            got = _dense_concat_forward(probe, W, act)
            if want.shape == got.shape and torch.allclose(want, got, rtol=1e-5, atol=1e-6):
                return dict(dims=dims, act=act, linear=False, params=list(W))
    return "V's forward is none of the dense-concat formulas the kernels implement (relu^2, tanh^2, tanh)"


def deep_eligibility(solver):
    """None if the value net is one the genl kernels take, else a reason."""
    d_in = solver.d + (0 if solver.elliptic else 1)
    spec = value_net_spec(solver.V, d_in)
    if isinstance(spec, str):
        return spec
    dims = spec['dims']
    L = len(dims) - 2
    if L < 1 or L > 4:
        return 'V has %d hidden layers (the native value-net kernels take 1 to 4)' % L
    if max(dims[1:-1]) > 128 or d_in > 112:
        return 'V is wider than the native value-net kernels take (hidden <= 128, input <= 112)'
    cfg = nat.GenlConfig()
    cfg.base.d, cfg.base.K_local, cfg.base.N = solver.d, 16, 1
    cfg.has_time, cfg.n_hidden = (0 if solver.elliptic else 1), L
    for i, h in enumerate(dims[1:-1]):
        cfg.widths[i] = int(h)
    sizes = nat.GenlSizes()
    lib = nat.load()
    if lib.psp_genl_query(C.byref(cfg), C.byref(sizes)) != 0:
        return lib.psp_last_error().decode()
    return None


class GeneralDeepPlan(GeneralNativePlan):
    def __init__(self, solver):
        s = solver
        self.s = s
        self.lib = nat.load()
        self.dev = s.device
        self._shard(s.K_original)
        self.K_cap = self.K_local
        self.net = s.V
        self.key = None
        self.elliptic = bool(s.elliptic)
        net = value_net_spec(s.V, s.d + (0 if self.elliptic else 1))
        assert not isinstance(net, str), net
        self.net_spec = net
        self.dims = list(net['dims'])
        self.L = len(self.dims) - 2
        self.H = self.dims[1]
        self._flatten(net['params'])
        spec = s.problem.general_native_spec()
        self._keep = []
        gcfg = nat.GenlConfig()
        cfg = gcfg.base
        cfg.d = s.d
        cfg.K_local, cfg.N, cfg.k_offset = self.K_local, s.N, self.lo
        cfg.dt, cfg.sqrt_dt = float(s.delta_t.item()), float(s.sq_delta_t.item())
        cfg.T = float('inf') if self.elliptic else float(torch.tensor(s.problem.T, dtype=torch.float32).item())
        set_domain(cfg, s.problem, self.elliptic)
        cfg.d_real = s.d
        for i, v in enumerate(spec.get('h_par', ())):
            cfg.h_par[i] = float(v)
        cfg.sigma_scale = float(spec['sigma_scale'])
        cfg.drift_kind, cfg.h_kind = spec['drift'][0], spec['h']
        cfg.adaptive = 1 if s.adaptive_forward_process else 0
        cfg.noise_mode = nat.NOISE_PHILOX if s.noise == 'philox' else nat.NOISE_SUPPLIED
        cfg.store_path = 1
        if spec['drift'][1] is not None:
            t = spec['drift'][1].detach().to(device=self.dev, dtype=torch.float32).contiguous()
            self._keep.append(t)
            cfg.drift = nat.ptr(t)
        gcfg.has_time, gcfg.n_hidden = (0 if self.elliptic else 1), self.L
        for i, h in enumerate(self.dims[1:-1]):
            gcfg.widths[i] = int(h)
        gcfg.activation = _ACT[net['act']]
        gcfg.linear_layout = 1 if net['linear'] else 0
        self.gcfg, self.cfg = gcfg, cfg
        sz = nat.GenlSizes()
        nat.check(self.lib.psp_genl_query(C.byref(gcfg), C.byref(sz)), 'psp_genl_query')
        assert sz.n_params == self.P, (sz.n_params, self.P)
        self.sizes = sz
        self.matrix_mode, self.range_flag = 'fp32', None            # fp32 MFMA only (v_mfma_f32_16x16x4_f32)
        self.d_pad, self.H_pad = s.d, self.H
        self.pad = _IdentityPad(self.P)
        self.flat_k = self.flat
        dev, f32 = self.dev, torch.float32
        self.tables = torch.empty(sz.table_bytes // 4, dtype=f32, device=dev)
        self.path = torch.empty(sz.path_bytes // 4, dtype=f32, device=dev)
        self.ahat = torch.zeros((sz.ahat_bytes + 3) // 4, dtype=f32, device=dev)
        self.grad_partial = torch.empty(sz.grad_partial_bytes // 4, dtype=f32, device=dev)
        self.grad_k = self.grad
        self._common_buffers(s.d)

    def _check_sizes(self):
        sz = nat.GenlSizes()
        nat.check(self.lib.psp_genl_query(C.byref(self.gcfg), C.byref(sz)), 'psp_genl_query')
        assert sz.path_bytes <= self.path.numel() * 4 and sz.ahat_bytes <= self.ahat.numel() * 4
        if sz.grad_partial_bytes > self.grad_partial.numel() * 4:
            self.grad_partial = torch.empty(sz.grad_partial_bytes // 4, dtype=torch.float32, device=self.dev)
        self.sizes = sz

    def _x_image_floats(self):
        return 4 * ((self.dims[0] + 15) // 16) * 64

    def _tile_steps(self):
        """Loop steps every tile of this rank executed (int32 view of the tail of `ahat`, include/psp.h)."""
        nt = (self.K_local + 15) // 16
        return self.ahat[(self.s.N + 1) * nt * 16:(self.s.N + 1) * nt * 16 + nt].view(torch.int32)

    def _loop_steps(self, t0):
        """The kernels count the executed steps of every tile themselves: the loop of the reference runs as long as the
        longest-running tile."""
        m = self._tile_steps().max().reshape(1)
        dist, _, world = sharding.dist_info()
        if world > 1:
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
        return int(m.item())

    def _launch_fwd(self, flat_k, x0, t0, xi, l, st):
        nat.check(self.lib.psp_genl_rollout_fwd(C.byref(self.gcfg), nat.ptr(self.flat), nat.ptr(x0), nat.ptr(t0), nat.ptr(xi),
                                                int(self.s.seed) & 0xFFFFFFFFFFFFFFFF, l, nat.ptr(self.tables), nat.ptr(self.path),
                                                nat.ptr(self.ahat), nat.ptr(self.VN), nat.ptr(self.YN), nat.ptr(self.XN_k),
                                                nat.ptr(self.tN), nat.ptr(self.kcount), st), 'psp_genl_rollout_fwd')

    def _launch_bwd(self, flat_k, st):
        nat.check(self.lib.psp_genl_rollout_bwd(C.byref(self.gcfg), nat.ptr(self.flat), nat.ptr(self.tables), nat.ptr(self.path),
                                                nat.ptr(self.ahat), nat.ptr(self.wY), nat.ptr(self.wV), nat.ptr(self.grad_partial),
                                                nat.ptr(self.grad), st), 'psp_genl_rollout_bwd')
