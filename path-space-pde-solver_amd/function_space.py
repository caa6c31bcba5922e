"""Ansatz spaces for the control / value function (API mirror of the reference's
function_space.py).  Every class is an ``nn.Module`` whose constructor takes ``lr`` and
owns its Adam optimiser as ``self.optim`` -- the protocol the solvers rely on
(reference function_space.py:18,131,185; solver.py:194-200).

The initial weights are bit-identical to the reference for equal seeds: each class
consumes the torch CPU generator in the same order (cited per class).  What differs:
``MySequential`` takes ``widths`` (the reference hard-codes [30, 30] at function_space.py:181)
and exposes ``flat_layout()`` which the native HIP plan uses to view its parameters as one
flat fp32 buffer.
"""
import torch
from torch import nn


def _own_adam(module, lr):
    module.optim = torch.optim.Adam(module.parameters(), lr=lr)


class SingleParam(nn.Module):
    """Learnable scalar Y_0 (reference function_space.py:6-21)."""

    def __init__(self, lr, initial=None, seed=42):
        super().__init__()
        torch.manual_seed(seed)
        if initial is None:
            start = torch.tensor([0.0])
        elif initial == 'random':
            start = torch.randn(1)
        else:
            start = torch.tensor([initial])
        self.Y_0 = nn.Parameter(start, requires_grad=True)
        self.register_parameter('param', self.Y_0)
        _own_adam(self, lr)

    def forward(self, x):
        return self.Y_0


class Constant(nn.Module):
    """State-independent control vector (reference function_space.py:24-34)."""

    def __init__(self, d, lr, seed=42):
        super().__init__()
        torch.manual_seed(seed)
        self.c = nn.Parameter(torch.randn(d), requires_grad=True)
        self.register_parameter('param', self.c)
        _own_adam(self, lr)

    def forward(self, x):
        return self.c.repeat(x.shape[0], 1)


class Linear(nn.Module):
    """u(x) = Q^-1 B^T F x with learnable F (reference function_space.py:37-48)."""

    def __init__(self, d, B, Q, lr, seed=42):
        super().__init__()
        torch.manual_seed(seed)
        self.F = nn.Parameter(torch.randn(d, d), requires_grad=True)
        self.B, self.Q = B, Q
        self.register_parameter('param', self.F)
        _own_adam(self, lr)

    def forward(self, x):
        gain = torch.mm(self.Q.inverse(), torch.mm(self.B.t(), self.F))
        return torch.mm(gain, x.t()).t()


class Affine(nn.Module):
    """u(x) = A x + b, zero-initialised (reference function_space.py:51-63)."""

    def __init__(self, d, lr, seed=42):
        super().__init__()
        torch.manual_seed(seed)
        self.A = nn.Parameter(torch.randn(d, d) * 0.0, requires_grad=True)
        self.b = nn.Parameter(torch.randn(1, d) * 0.0, requires_grad=True)
        self.register_parameter('param A', self.A)
        self.register_parameter('param b', self.b)
        _own_adam(self, lr)

    def forward(self, x):
        return torch.mm(self.A, x.t()).t() + self.b


class DenseNet(nn.Module):
    """Densely connected net with relu(.)**2 activations (reference function_space.py:116-140).

    Layer i sees the concatenation of the input and all previous hidden outputs; weights are
    stored (in, out) and drawn as randn * 0.1 with zero biases, in layer order (:121-125).
    ``activation`` (not a reference argument; default = the reference's relu(.)**2) selects the
    hidden nonlinearity: 'relu2', 'tanh2' (tanh(.)**2, the net `Committor function.ipynb` defines
    in its first cell) or 'tanh'.  The parameter draws do not depend on it.
    """
    ACTIVATIONS = ('relu2', 'tanh2', 'tanh')

    def __init__(self, d_in, d_out, lr, arch=[30, 30], seed=42, activation='relu2'):
        super().__init__()
        if activation not in self.ACTIVATIONS:
            raise ValueError('activation must be one of %s' % (self.ACTIVATIONS,))
        self.activation = activation
        torch.manual_seed(seed)
        self.nn_dims = [d_in] + list(arch) + [d_out]
        self.W = []
        fan_in = 0
        for i in range(len(self.nn_dims) - 1):
            fan_in += self.nn_dims[i]
            self.W.append(nn.Parameter(torch.randn(fan_in, self.nn_dims[i + 1], requires_grad=True) * 0.1))
            self.W.append(nn.Parameter(torch.zeros(self.nn_dims[i + 1], requires_grad=True)))
        for i, w in enumerate(self.W):
            self.register_parameter('param %d' % i, w)
        _own_adam(self, lr)

    def _hidden(self, z):
        if self.activation == 'relu2':
            return torch.relu(z) ** 2
        return torch.tanh(z) ** 2 if self.activation == 'tanh2' else torch.tanh(z)

    def forward(self, x):
        depth = len(self.nn_dims) - 1
        for i in range(depth - 1):
            hidden = self._hidden(torch.matmul(x, self.W[2 * i]) + self.W[2 * i + 1])
            x = torch.cat([x, hidden], dim=1)
        return torch.matmul(x, self.W[2 * depth - 2]) + self.W[2 * depth - 1]


class DenseNet_tanh_2(DenseNet):
    """Dense-concat net with tanh(.)**2: the class the reference's `Committor function.ipynb` defines in its first cell and
    swaps into ``model.V`` (same parameter draws as DenseNet)."""

    def __init__(self, d_in, d_out, lr, arch=[30, 30], seed=42):
        super().__init__(d_in, d_out, lr, arch=arch, seed=seed, activation='tanh2')


class DenseNet_tanh(nn.Module):
    """Dense-concat net with tanh and nn.Linear layers (reference function_space.py:143-158)."""

    def __init__(self, d_in, d_out, lr, arch=[30, 30], seed=42):
        super().__init__()
        torch.manual_seed(seed)
        self.nn_dims = [d_in] + list(arch) + [d_out]
        widths = [sum(self.nn_dims[:i + 1]) for i in range(len(self.nn_dims) - 1)]
        self.layers = nn.ModuleList([nn.Linear(w, self.nn_dims[i + 1]) for i, w in enumerate(widths)])
        _own_adam(self, lr)

    def forward(self, x):
        for layer in self.layers[:-1]:
            x = torch.cat([x, torch.tanh(layer(x))], dim=1)
        return self.layers[-1](x)


class MySequential(nn.Module):
    """Plain tanh MLP, the default control net for time_approx='inner'
    (reference function_space.py:177-195; built at solver.py:91 with seed=123).

    RNG recipe (:180-188): manual_seed(seed); one nn.Linear per layer; Adam; then
    normal_(0, 0.01) on weight and bias, layer by layer.  ``widths`` defaults to the
    reference's hard-coded [30, 30]; BASELINE.json's configs use widths=[64, 64].
    """

    def __init__(self, d_in, d_out, lr, seed, widths=(30, 30)):
        super().__init__()
        torch.manual_seed(seed)
        self.nn_dims = [d_in] + list(widths) + [d_out]
        self.linears = nn.ModuleList(
            [nn.Linear(a, b) for a, b in zip(self.nn_dims[:-1], self.nn_dims[1:])])
        self.activations = nn.ModuleList([nn.Tanh() for _ in range(len(self.nn_dims) - 2)])
        _own_adam(self, lr)
        for lin in self.linears:
            nn.init.normal_(lin.weight, 0, 0.01)
            nn.init.normal_(lin.bias, 0, 0.01)

    def forward(self, x):
        for lin, act in zip(self.linears[:-1], self.activations):
            x = act(lin(x))
        return self.linears[-1](x)

    # ---- native-plan hooks -----------------------------------------------------------------
    def native_shape(self):
        """(d_in, H, d_out) if this net has the two-equal-hidden-layer shape the HIP rollout
        kernels implement, else None."""
        dims = self.nn_dims
        if len(dims) == 4 and dims[1] == dims[2]:
            return dims[0], dims[1], dims[3]
        return None

    def flat_layout(self):
        """Parameters in the order of include/psp.h: [W1, b1, W2, b2, W3, b3]."""
        out = []
        for lin in self.linears:
            out += [lin.weight, lin.bias]
        return out
