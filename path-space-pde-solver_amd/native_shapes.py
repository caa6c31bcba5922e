"""Which compiled kernel instance serves a (d, H) configuration, and the zero padding that maps one onto the other.

The HIP kernels are compiled per (d, H) (csrc/instances.def, csrc/wide_instances.def).  A configuration that is not in
the list runs EXACTLY on any instance with d_pad >= d, H_pad >= H after zero padding: padded hidden units have zero
weights and biases (tanh(0) = 0 feeds nothing), padded state components have zero rows / columns in W1, W3, A, B and
zero entries in the drift / running-cost / terminal-cost vectors, so they never couple back into the real ones; the
Philox counters are indexed per 16-feature block, so the real components see the same noise.  The gradient entries
of padded parameters are simply not gathered back.

  choose(...)      cheapest instance that accepts the configuration (exact shape first, narrow family before wide)
  ParamPad         index map real flat parameter vector <-> padded one, and the tensor padding helpers
"""
import os

import torch

try:
    from . import native as nat
except ImportError:
    import native as nat


def _cost(D, H, dense):
    """MFMA work per trajectory-step of an instance (arbitrary units), used to rank candidates."""
    return (2 * D * D if dense else 0) + (D + 1) * H + H * H + D * H


def candidates(d, H, dense=True):
    force_wide = os.environ.get('PSP_FORCE_WIDE', '') == '1'
    cands = [(D, Hh, fam) for (D, Hh, fam) in nat.instances() if D >= d and Hh >= H]
    # exact shape first; then narrow before wide (unless forced); then by work
    cands.sort(key=lambda c: (0 if (c[0] == d and c[1] == H) else 1,
                              (c[2] != 2) if force_wide else (c[2] != 1), _cost(c[0], c[1], dense)))
    seen, out = set(), []
    for c in cands:
        if c[:2] not in seen or force_wide:
            out.append(c)
            seen.add(c[:2])
    return out


def choose(cfg, d, H):
    """Sets cfg.d / cfg.H to the first candidate instance psp_hjb_query accepts for this configuration (LDS budget,
    family restrictions).  Returns (d_pad, H_pad, family, sizes) or (None, reason)."""
    dense = cfg.drift_kind == nat.DRIFT_DENSE or cfg.sigma_kind == nat.SIGMA_DENSE
    last = 'no compiled kernel instance covers d=%d, H=%d (csrc/instances.def, csrc/wide_instances.def)' % (d, H)
    for D, Hh, fam in candidates(d, H, dense):
        cfg.d, cfg.H = D, Hh
        rc, sizes, msg = nat.query_rc(cfg)
        if rc == 0:
            return (D, Hh, fam, sizes), None
        last = msg
    return None, last


class ParamPad:
    """Real (d, H) <-> padded (dp, Hp) layouts of the flat MySequential parameter vector [W1,b1,W2,b2,W3,b3]."""

    def __init__(self, d, H, dp, Hp, dev):
        self.d, self.H, self.dp, self.Hp, self.dev = d, H, dp, Hp, dev
        self.identity = (d == dp and H == Hp)
        self.P = (d + 1) * H + H + H * H + H + d * H + d
        self.Pp = (dp + 1) * Hp + Hp + Hp * Hp + Hp + dp * Hp + dp
        if self.identity:
            self.idx = None
            return
        ar = torch.arange
        oW1, ob1 = 0, Hp * (dp + 1)
        oW2 = ob1 + Hp
        ob2 = oW2 + Hp * Hp
        oW3 = ob2 + Hp
        ob3 = oW3 + dp * Hp
        parts = [
            (oW1 + ar(H)[:, None] * (dp + 1) + ar(d + 1)[None, :]).reshape(-1),      # W1[h, 0..d]   (column 0 = time)
            ob1 + ar(H),
            (oW2 + ar(H)[:, None] * Hp + ar(H)[None, :]).reshape(-1),
            ob2 + ar(H),
            (oW3 + ar(d)[:, None] * Hp + ar(H)[None, :]).reshape(-1),
            ob3 + ar(d),
        ]
        self.idx = torch.cat(parts).to(dev)
        assert self.idx.numel() == self.P

    # ---- parameters / gradients
    def new_padded_params(self):
        return torch.zeros(self.Pp, dtype=torch.float32, device=self.dev)

    def scatter_params(self, flat, flat_pad):
        """flat (real) -> flat_pad (padded entries stay zero)."""
        if self.identity:
            return flat
        flat_pad.index_copy_(0, self.idx, flat)
        return flat_pad

    def gather_grad(self, grad_pad, out):
        if self.identity:
            return grad_pad
        torch.index_select(grad_pad, 0, self.idx, out=out)
        return out

    # ---- problem data
    def vec(self, v):
        """(d,) -> (dp,) zero padded."""
        if v is None or self.identity:
            return v
        out = torch.zeros(self.dp, dtype=v.dtype, device=v.device)
        out[:self.d] = v
        return out

    def mat(self, M):
        """(d, d) -> (dp, dp) zero padded."""
        if M is None or self.identity:
            return M
        out = torch.zeros(self.dp, self.dp, dtype=M.dtype, device=M.device)
        out[:self.d, :self.d] = M
        return out

    def last_dim(self, x):
        """(..., d) -> (..., dp) zero padded, contiguous."""
        if x is None or self.identity:
            return x
        return torch.nn.functional.pad(x, (0, self.dp - self.d)).contiguous()

    def drift_or_sigma(self, t):
        """dense matrices or per-component vectors of the problem spec."""
        if t is None:
            return None
        return self.mat(t) if t.dim() == 2 else self.vec(t)


# ------------------------------------------------------------------------------------------------------------------
# GeneralSolver (value net V = DenseNet(d+1 -> 1, arch [H, H]), weights stored (in, out), time is the LAST input)
# ------------------------------------------------------------------------------------------------------------------
def gen_candidates(d, H):
    c = [(D, Hh) for (D, Hh) in nat.gen_instances() if D >= d and Hh >= H]
    c.sort(key=lambda t: (0 if (t[0] == d and t[1] == H) else 1, 3 * ((t[0] + 1) * t[1] + (t[0] + 1 + t[1]) * t[1])))
    return c


def gen_choose(cfg, d, H):
    last = 'no compiled GeneralSolver kernel instance covers d=%d, H=%d (csrc/gen_instances.def)' % (d, H)
    for D, Hh in gen_candidates(d, H):
        cfg.d, cfg.H = D, Hh
        rc, sizes, msg = nat.gen_query_rc(cfg)
        if rc == 0:
            return (D, Hh, sizes), None
        last = msg
    return None, last


class GenParamPad(ParamPad):
    """Index map for the DenseNet flat vector [W1 (DI x H), b1, W2 ((DI+H) x H), b2, W3 (DI+2H), b3], DI = d + 1.
    Input rows are [x (d), t]: padding moves the t row from index d to index d_pad and shifts the h1 / h2 row blocks."""

    def __init__(self, d, H, dp, Hp, dev, time_input=True, time_first=False, time_scale=1.0):
        """time_input=False: the real net is DenseNet(d -> 1) (EllipticSolver, reference solver.py:606); the kernels'
        time row then stays zero in the padded vector, so the (finite) time register never reaches the value.
        time_first=True: the real net's input is [t, x] (Solver's value-function ansatz, solver.py:343-344) -- its row 0 goes
        to the kernels' time row; time_scale multiplies the time rows on the way in and the gradient on the way out (the
        kernels' time register holds n * dt where that ansatz feeds the step index n)."""
        self.d, self.H, self.dp, self.Hp, self.dev = d, H, dp, Hp, dev
        self.identity = (d == dp and H == Hp and time_input and not time_first and time_scale == 1.0)
        self.scale = None
        DI, DIp = d + (1 if time_input else 0), dp + 1
        self.P = DI * H + H + (DI + H) * H + H + (DI + 2 * H) + 1
        self.Pp = DIp * Hp + Hp + (DIp + Hp) * Hp + Hp + (DIp + 2 * Hp) + 1
        if self.identity:
            self.idx = None
            return
        ar = torch.arange
        if time_input and time_first:
            rows_in = torch.cat([torch.tensor([dp]), ar(d)])                             # t row first, then the x rows
        else:
            rows_in = torch.cat([ar(d), torch.tensor([dp])]) if time_input else ar(d)  # x rows, then t
        rows2 = torch.cat([rows_in, DIp + ar(H)])                                    # ... then h1
        rows3 = torch.cat([rows2, DIp + Hp + ar(H)])                                 # ... then h2
        oW1, ob1 = 0, DIp * Hp
        oW2 = ob1 + Hp
        ob2 = oW2 + (DIp + Hp) * Hp
        oW3 = ob2 + Hp
        ob3 = oW3 + DIp + 2 * Hp
        parts = [
            (oW1 + rows_in[:, None] * Hp + ar(H)[None, :]).reshape(-1),
            ob1 + ar(H),
            (oW2 + rows2[:, None] * Hp + ar(H)[None, :]).reshape(-1),
            ob2 + ar(H),
            oW3 + rows3,
            torch.tensor([ob3]),
        ]
        self.idx = torch.cat(parts).to(dev)
        assert self.idx.numel() == self.P
        if time_input and time_scale != 1.0:
            tpos = 0 if time_first else d                                                # time row of W1 / W2 / W3 in the REAL net
            sc = torch.ones(self.P)
            oW1r, oW2r = 0, DI * H + H
            oW3r = oW2r + (DI + H) * H + H
            sc[oW1r + tpos * H:oW1r + (tpos + 1) * H] = time_scale
            sc[oW2r + tpos * H:oW2r + (tpos + 1) * H] = time_scale
            sc[oW3r + tpos] = time_scale
            self.scale = sc.to(dev)

    def scatter_params(self, flat, flat_pad):
        if self.scale is None:
            return super().scatter_params(flat, flat_pad)
        flat_pad.index_copy_(0, self.idx, flat * self.scale)
        return flat_pad

    def gather_grad(self, grad_pad, out):
        if self.scale is None:
            return super().gather_grad(grad_pad, out)
        torch.index_select(grad_pad, 0, self.idx, out=out)
        out.mul_(self.scale)
        return out
