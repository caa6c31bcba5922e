"""Native plan of GeneralSolver / EllipticSolver for value nets of ANY depth: V = DenseNet(d [+ 1] -> 1, arch = [H_1 .. H_L]),
1 <= L <= 4, H_i <= 128 -- the nets the reference's diffusion-loss notebooks train (Allen-Cahn.ipynb:72 arch = [110, 110, 50],
the only configuration with a published timing; [30, 30, 30, 30], [15, 15, 15, 15], ...; reference function_space.py:116-140).

Same iteration as plan_general_native.GeneralNativePlan (host RNG in the reference's order, K-vector loss weights, the
K_boundary-sized terms by autograd, Adam through psp_adam_step); what differs are the two device steps:
    psp_genl_rollout_fwd   the rollout (csrc/genl_kernels.h: activations in per-wave LDS images, weight tables in L2,
                           rolled fp32-MFMA products; shapes are run-time arguments, nothing is padded on the host)
    psp_genl_adjoints      per sample: activations a, tangents a', adjoints zbar_i, zbar_i' as row-major matrices,
    + library GEMMs        dW_i = A[:, :in_i]^T Zbar_i + A'[:, :in_i]^T Zbar_i' (torch.matmul = hipBLASLt: plain GEMMs over the
                           sample axis), walked over the path store in slabs of a fixed memory budget.
The (d, H)-templated kernels of gen_kernels.h stay the fast path for arch = [H, H], H <= 64.
"""
import ctypes as C

import torch

try:
    from . import native as nat
    from . import sharding
    from .function_space import DenseNet
    from .plan_general_native import GeneralNativePlan
except ImportError:
    import native as nat
    import sharding
    from function_space import DenseNet
    from plan_general_native import GeneralNativePlan


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


def deep_eligibility(solver):
    """None if the value net is a DenseNet the genl kernels take (and the two-hidden-layer kernels do not), else a reason."""
    V = solver.V
    dims = getattr(V, 'nn_dims', None)
    d_in = solver.d + (0 if solver.elliptic else 1)
    if not isinstance(V, DenseNet) or dims is None or dims[0] != d_in or dims[-1] != 1:
        return 'V is not a DenseNet(%d -> 1)' % d_in
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
    ADJ_BUDGET_BYTES = 2 << 30        # activations + adjoints of one slab of samples (the GEMM operands)

    def __init__(self, solver):
        s = solver
        self.s = s
        self.lib = nat.load()
        self.dev = s.device
        self.dist, self.rank, self.world = sharding.dist_info()
        lo, hi = sharding.shard_bounds(s.K, self.rank, self.world)
        self.lo, self.hi, self.K_local = lo, hi, hi - lo
        self.net = s.V
        self.key = None
        self.dims = list(s.V.nn_dims)
        self.L = len(self.dims) - 2
        self.H = self.dims[1]
        self._flatten(s.V)
        spec = s.problem.general_native_spec()
        self._keep = []
        gcfg = nat.GenlConfig()
        cfg = gcfg.base
        cfg.d = s.d
        cfg.K_local, cfg.N, cfg.k_offset = self.K_local, s.N, lo
        cfg.dt, cfg.sqrt_dt = float(s.delta_t.item()), float(s.sq_delta_t.item())
        self.elliptic = bool(s.elliptic)
        cfg.T = float('inf') if self.elliptic else float(torch.tensor(s.problem.T, dtype=torch.float32).item())
        pb = s.problem
        if pb.boundary == 'sphere':
            cfg.domain_kind, cfg.dom_a = nat.DOM_SPHERE, float(pb.boundary_distance)
        elif pb.boundary == 'square':
            cfg.dom_a, cfg.dom_b = float(pb.X_l), float(pb.X_r)
            cfg.domain_kind = nat.DOM_BOX if not pb.one_boundary else \
                (nat.DOM_BOX_UPPER_ALL if self.elliptic else nat.DOM_BOX_UPPER_ANY)
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
        self.ahat = torch.zeros(sz.ahat_bytes // 4, dtype=f32, device=dev)
        self.VN = torch.empty(self.K_local, dtype=f32, device=dev)
        self.YN = torch.empty(self.K_local, dtype=f32, device=dev)
        self.tN = torch.zeros(self.K_local, dtype=f32, device=dev)
        self.XN_k = torch.empty(self.K_local, s.d, dtype=f32, device=dev)
        self.XN = self.XN_k
        self.kcount = torch.zeros(1, dtype=torch.int64, device=dev)
        self.grad = torch.empty(self.P, dtype=f32, device=dev)
        self.grad_k = self.grad
        self.m = torch.zeros(self.P, dtype=f32, device=dev)
        self.v = torch.zeros(self.P, dtype=f32, device=dev)
        self.Kpad = 16 * ((self.K_local + 15) // 16)
        self.wY = torch.zeros(self.Kpad, dtype=f32, device=dev)
        self.wV = torch.zeros(self.Kpad, dtype=f32, device=dev)
        self.step = 0
        self.last_v_l2 = None
        self.events = None
        # ---- slab buffers of the adjoint pass and the index maps from padded features to real parameter rows / columns
        # padded widths of the a / zbar rows, without their trailing block (ones | a, w)
        self.TBf, self.HBf = sz.act_floats_per_block // 16 - 16, sz.zbar_floats_per_block // 16 - 16
        per_block = 4 * (2 * sz.act_floats_per_block + 2 * sz.zbar_floats_per_block + 32)
        self.slab_blocks = max(1, min(int(sz.n_blocks), self.ADJ_BUDGET_BYTES // per_block))
        nb = self.slab_blocks
        self.bA = torch.empty(nb * sz.act_floats_per_block, dtype=f32, device=dev)
        self.bAd = torch.empty(nb * sz.act_floats_per_block, dtype=f32, device=dev)
        self.bZ = torch.empty(nb * sz.zbar_floats_per_block, dtype=f32, device=dev)
        self.bZd = torch.empty(nb * sz.zbar_floats_per_block, dtype=f32, device=dev)
        self.bav = torch.empty(nb * 16, dtype=f32, device=dev)
        self.bwy = torch.empty(nb * 16, dtype=f32, device=dev)
        seg = [int(sz.seg_block_offset[i]) for i in range(self.L + 1)]
        D0 = self.dims[0]
        widths = [D0] + self.dims[1:-1]
        # real feature r of the concatenation a_L  <->  padded feature 16 * seg[s] + c
        pad_of_real = []
        for sgm, w in enumerate(widths):
            pad_of_real += [16 * seg[sgm] + c for c in range(w)]
        self.pad_of_real = torch.tensor(pad_of_real, dtype=torch.long, device=dev)             # (D0 + sum H)
        self.zcol0, self.in_pad = [], []                   # per layer: first zbar column of its units; padded width of its input
        zoff = 0
        for i in range(self.L):
            self.zcol0.append(16 * zoff)
            self.in_pad.append(16 * seg[i + 1])
            zoff += (self.dims[1 + i] + 15) // 16
        # flat-gradient offsets in registration order W_1, b_1, .., W_out, b_out
        self.goff, o, n_in = [], 0, D0
        for i in range(self.L):
            Hi = self.dims[1 + i]
            self.goff.append((o, o + n_in * Hi, n_in, Hi))
            o += n_in * Hi + Hi
            n_in += Hi
        self.goff_out = (o, o + n_in, n_in)

    def _x_image_floats(self):
        return 4 * ((self.dims[0] + 15) // 16) * 64

    def _launch_fwd(self, flat_k, x0, t0, xi, l, st):
        nat.check(self.lib.psp_genl_rollout_fwd(C.byref(self.gcfg), nat.ptr(self.flat), nat.ptr(x0), nat.ptr(t0), nat.ptr(xi),
                                                int(self.s.seed) & 0xFFFFFFFFFFFFFFFF, l, nat.ptr(self.tables), nat.ptr(self.path),
                                                nat.ptr(self.ahat), nat.ptr(self.VN), nat.ptr(self.YN), nat.ptr(self.XN_k),
                                                nat.ptr(self.tN), nat.ptr(self.kcount), st), 'psp_genl_rollout_fwd')

    MAX_BATCHES = 32             # split-K factor of the weight-gradient products
    MIN_ROWS = 2048              # ... and the fewest samples a batch is worth

    @classmethod
    def _atb(cls, A, B):
        """A^T B for tall operands (samples x features): the sample axis is cut into up to MAX_BATCHES equal batches so that
        the product is a batched GEMM with many workgroups (one plain GEMM has in x H / tile^2 ~ 60 output tiles for a
        contraction over up to millions of samples and runs on 60 CUs), then summed over the batches."""
        n = A.shape[0]
        nb = max(1, min(cls.MAX_BATCHES, n // cls.MIN_ROWS))
        if nb == 1:
            return A.t() @ B
        R = n // nb
        out = torch.bmm(A[:nb * R].view(nb, R, A.shape[1]).transpose(1, 2), B[:nb * R].view(nb, R, B.shape[1])).sum(0)
        if nb * R < n:
            out = out + A[nb * R:].t() @ B[nb * R:]
        return out

    def _launch_bwd(self, flat_k, st):
        sz, g = self.sizes, self.grad
        g.zero_()
        n_blocks = int(sz.n_blocks)
        TBf, HBf = self.TBf, self.HBf
        WA, WZ = TBf + 16, HBf + 16
        for b0 in range(0, n_blocks, self.slab_blocks):
            b1 = min(n_blocks, b0 + self.slab_blocks)
            ns = 16 * (b1 - b0)
            nat.check(self.lib.psp_genl_adjoints(C.byref(self.gcfg), nat.ptr(self.flat), nat.ptr(self.tables), nat.ptr(self.path),
                                                 nat.ptr(self.ahat), nat.ptr(self.wY), nat.ptr(self.wV), b0, b1, nat.ptr(self.bA),
                                                 nat.ptr(self.bAd), nat.ptr(self.bZ), nat.ptr(self.bZd), nat.ptr(self.bav),
                                                 nat.ptr(self.bwy), st), 'psp_genl_adjoints')
            # row-major (sample, padded feature) matrices straight from the kernel, each row with one trailing block: ones
            # behind a, (a | w) behind zbar.  ONE pair of batched GEMMs forms M = A^T Zbar + A'^T Zbar' on the padded layout:
            # weight blocks, bias gradients (row TBf: the ones column) and the output layer (column HBf) at once.  Padding rows /
            # columns are exactly zero; the blocks of M no layer needs are the price of two launches instead of 2 L + 2.
            A, Ad = self.bA[:ns * WA].view(ns, WA), self.bAd[:ns * WA].view(ns, WA)
            Zb, Zd = self.bZ[:ns * WZ].view(ns, WZ), self.bZd[:ns * WZ].view(ns, WZ)
            M = self._atb(A, Zb) + self._atb(Ad, Zd)                       # (TBf + 16, HBf + 16)
            for i in range(self.L):
                oW, ob, n_in, Hi = self.goff[i]
                c0 = self.zcol0[i]
                g[oW:ob].add_(M[self.pad_of_real[:n_in], c0:c0 + Hi].reshape(-1))
                g[ob:ob + Hi].add_(M[TBf, c0:c0 + Hi])
            oW, ob, n_in = self.goff_out
            g[oW:ob].add_(M[self.pad_of_real, HBf])
            g[ob:ob + 1].add_(M[TBf, HBf:HBf + 1])
