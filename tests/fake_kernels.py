"""TEST INFRASTRUCTURE: a CPU stand-in for the launch entry points of libpsp_hip.so.

The native plan (path-space-pde-solver_amd/plan_native.py) is host logic around five C-ABI launches.  Its multi-rank path --
trajectory sharding with global k_offset, the two collectives, K-chunking, loss assembly, per-net Adam -- can run on a machine
without a GPU if something answers those launches.  `FakeKernels` wraps the REAL library (size queries, instance tables and
argument validation stay the library's own) and answers the launches psp_hjb_rollout_fwd / _terminal_reduce[_loss] /
_rollout_bwd / psp_adam_step on host tensors, reading the same psp_hjb_config and raw pointers the HIP kernels would get:
forward = the Euler-Maruyama / Y recursion of include/psp.h in torch (fp32), backward = autograd of sum_k w_k D_k with the
weights the header prescribes.  Only tests import this; the product never does, and nothing here is timed or shipped.
"""
import ctypes as C

import torch

from util_cases import psp

nat = psp.native


def _f32(ptr, n):
    return torch.frombuffer((C.c_float * n).from_address(ptr), dtype=torch.float32)


def _f64(ptr, n):
    return torch.frombuffer((C.c_double * n).from_address(ptr), dtype=torch.float64)


def _val(p):
    return p.value if isinstance(p, C.c_void_p) else p


class FakeKernels:
    def __init__(self):
        self.real = nat.load()
        self.state = {}            # path pointer (or D pointer when no path) -> (D with graph, flat parameter leaf)
        self.sums = {}             # fwd_partial pointer -> (sum D, sum D^2) of the last forward that used it
        self.calls = []

    def __getattr__(self, name):
        return getattr(self.real, name)

    # ---- the launches -------------------------------------------------------------------------------------------------
    @staticmethod
    def _cfg(cfg_ref):
        return cfg_ref._obj

    def _mlp(self, c, flat, X, t):
        d, H = c.d, c.H
        o = 0
        W1 = flat[o:o + H * (d + 1)].view(H, d + 1); o += H * (d + 1)
        b1 = flat[o:o + H]; o += H
        W2 = flat[o:o + H * H].view(H, H); o += H * H
        b2 = flat[o:o + H]; o += H
        W3 = flat[o:o + d * H].view(d, H); o += d * H
        b3 = flat[o:o + d]
        tx = torch.cat([torch.full((X.shape[0], 1), 1.0) * t, X], 1)
        return torch.tanh(torch.tanh(tx @ W1.t() + b1) @ W2.t() + b2) @ W3.t() + b3

    def psp_hjb_rollout_fwd(self, cfg_ref, params, x0, x0_stride, y0, xi, seed, it, path, D_out, XN_out, Y_out, fwd_partial,
                            stream):
        c = self._cfg(cfg_ref)
        assert c.noise_mode == nat.NOISE_SUPPLIED, "the CPU stand-in has no Philox: run the plan with noise='reference'"
        assert c.store_path in (0, 1) and c.loss_kind != nat.LOSS_REL_ENTROPY
        d, K, N = c.d, c.K_local, c.N
        n_par = (d + 1) * c.H + c.H + c.H * c.H + c.H + d * c.H + d
        flat = _f32(_val(params), n_par).clone().requires_grad_(True)
        dt, sq = torch.tensor(c.dt), torch.tensor(c.sqrt_dt)
        X = (_f32(_val(x0), K * d).view(K, d) if x0_stride else _f32(_val(x0), d).repeat(K, 1)).clone()
        noise = _f32(_val(xi), (N + 1) * K * d).view(N + 1, K, d)
        Y = torch.zeros(K) if not _val(y0) else _f32(_val(y0), 1).repeat(K).clone()
        A = _f32(c.drift, d * d).view(d, d) if c.drift_kind == nat.DRIFT_DENSE else None
        avec = _f32(c.drift, d) if c.drift_kind in (nat.DRIFT_DIAG, nat.DRIFT_DOUBLE_WELL) else None
        B = _f32(c.sigma, d * d).view(d, d) if c.sigma_kind == nat.SIGMA_DENSE else None
        s_scale = c.sigma_scale if c.sigma_kind == nat.SIGMA_SCALED_IDENTITY else 1.0
        pvec = _f32(c.runcost, d) if c.runcost_kind == nat.RUNCOST_DIAG_QUAD else None
        tvec = _f32(c.term, d)
        for n in range(N):
            Z = self._mlp(c, flat, X, float(n) * dt)
            cc = (-Z.detach()) if c.adaptive else torch.zeros_like(Z)          # detached forward process
            if A is not None:
                b = X @ A.t()
            elif c.drift_kind == nat.DRIFT_DIAG:
                b = avec * X
            elif c.drift_kind == nat.DRIFT_DOUBLE_WELL:
                b = -4.0 * avec * X * (X * X - 1.0)
            else:
                b = torch.zeros_like(X)
            v = cc * dt + noise[n + 1] * sq
            X = X + b * dt + (v @ B.t() if B is not None else s_scale * v)
            f = (pvec * X * X).sum(1) if pvec is not None else torch.zeros(K)
            Y = Y + (0.5 * (Z * Z).sum(1) + f + (Z * cc).sum(1)) * dt + (Z * noise[n + 1]).sum(1) * sq   # -h = |z|^2/2 + f
        if c.term_kind == nat.TERM_LINEAR:
            g = X @ tvec
        elif c.term_kind == nat.TERM_DIAG_QUAD:
            g = (tvec * X * X).sum(1)
        else:
            g = (tvec * (X - 1.0) ** 2).sum(1)
        D = Y - g
        _f32(_val(D_out), K).copy_(D.detach())
        if _val(Y_out):
            _f32(_val(Y_out), K).copy_(Y.detach())
        if _val(XN_out):
            _f32(_val(XN_out), K * d).copy_(X.detach().reshape(-1))
        Dd = D.detach().double()
        self.sums[_val(fwd_partial)] = (Dd.sum(), (Dd * Dd).sum())
        if c.store_path:
            self.state[_val(path)] = (D, flat)
        self.calls.append(("fwd", K, int(c.k_offset), int(c.store_path)))
        return 0

    def psp_hjb_terminal_reduce(self, cfg_ref, fwd_partial, sums_out, stream):
        s0, s1 = self.sums[_val(fwd_partial)]
        out = _f64(_val(sums_out), 2)
        out[0], out[1] = s0, s1
        return 0

    def psp_hjb_terminal_reduce_loss(self, cfg_ref, fwd_partial, sums_out, loss_log, index_dev, stream):
        self.psp_hjb_terminal_reduce(cfg_ref, fwd_partial, sums_out, stream)
        c = self._cfg(cfg_ref)
        if _val(loss_log):
            s = _f64(_val(sums_out), 2)
            K = float(c.K_global)
            m = s[0] / K
            loss = s[1] / K - m * m if c.loss_kind == nat.LOSS_LOG_VARIANCE else s[1] / K
            _f32(_val(loss_log), 1)[0] = float(loss)
        return 0

    def psp_hjb_rollout_bwd(self, cfg_ref, params, xi, seed, it, path, D_ptr, sums, grad_partial, grad_out, stream):
        c = self._cfg(cfg_ref)
        D, flat = self.state[_val(path)]
        K = c.K_local
        dvals = _f32(_val(D_ptr), K).clone()
        Kg = float(c.K_global)
        if c.loss_kind == nat.LOSS_WEIGHTS:
            w = dvals
        elif c.loss_kind == nat.LOSS_LOG_VARIANCE:
            w = (2.0 / Kg) * (dvals - float(_f64(_val(sums), 2)[0] / Kg))
        else:
            w = (2.0 / Kg) * dvals
        grad, = torch.autograd.grad((w * D).sum(), flat, retain_graph=True)
        _f32(_val(grad_out), flat.numel()).copy_(grad)
        self.calls.append(("bwd", K, int(c.k_offset), int(c.loss_kind)))
        return 0

    def psp_adam_step(self, params, grad, m, v, n, step, lr, b1, b2, eps, stream):
        p, g = _f32(_val(params), n), _f32(_val(grad), n)
        mm, vv = _f32(_val(m), n), _f32(_val(v), n)
        mm.copy_(mm + (g - mm) * (1.0 - b1))
        vv.copy_(vv * b2 + (1.0 - b2) * g * g)
        bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
        denom = vv.sqrt() / (bc2 ** 0.5) + eps
        p.copy_(p - (lr / bc1) * (mm / denom))
        return 0


def install(monkeypatch_like=None):
    """Route the native plan to the stand-in and let it accept a CPU device.  Returns the FakeKernels object."""
    fake = FakeKernels()
    import path_space_pde_solver_amd.solver as solver_mod
    accept = lambda solver: None          # noqa: E731  (the real check refuses a CPU device)
    psp.plan_native.native_eligibility = accept
    solver_mod.native_eligibility = accept
    nat.load = lambda: fake
    return fake
