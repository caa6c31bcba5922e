"""Host logic of the instance chooser and the zero-padding index map (native_shapes.py); no GPU needed."""
import torch

from util_cases import psp

nat, shapes = psp.native, psp.native_shapes


def mlp(flat, d, H, x_t):
    """[t, x] -> W3 tanh(W2 tanh(W1 [t,x] + b1) + b2) + b3 from a flat [W1,b1,W2,b2,W3,b3] vector."""
    o = 0
    W1 = flat[o:o + H * (d + 1)].view(H, d + 1); o += H * (d + 1)
    b1 = flat[o:o + H]; o += H
    W2 = flat[o:o + H * H].view(H, H); o += H * H
    b2 = flat[o:o + H]; o += H
    W3 = flat[o:o + d * H].view(d, H); o += d * H
    b3 = flat[o:o + d]; o += d
    assert o == flat.numel()
    return torch.tanh(torch.tanh(x_t @ W1.t() + b1) @ W2.t() + b2) @ W3.t() + b3


def test_instance_list_covers_the_padded_grid():
    inst = nat.instances()
    assert (100, 64, 1) in inst and (500, 64, 2) in inst
    for D in (16, 32, 48, 64, 80, 96, 112):
        for H in (16, 32, 48, 64):
            assert (D, H, 1) in inst
    for D in (128, 256, 512):
        assert (D, 64, 2) in inst


def test_candidate_order_exact_then_narrow_then_wide():
    c = shapes.candidates(100, 64)
    assert c[0] == (100, 64, 1)
    assert shapes.candidates(7, 30)[0] == (8, 30, 1)          # cheapest covering instance, not the grid point
    assert shapes.candidates(33, 50)[0] == (48, 64, 1)
    c = shapes.candidates(105, 64)
    assert c[0] == (112, 64, 1) and any(x == (128, 64, 2) for x in c)      # the chooser falls through on LDS overflow
    assert shapes.candidates(300, 40)[0] == (320, 64, 2)
    assert shapes.candidates(600, 64) == []


def test_padded_network_equals_real_network():
    torch.manual_seed(0)
    d, H, dp, Hp = 7, 30, 16, 32
    pad = shapes.ParamPad(d, H, dp, Hp, torch.device('cpu'))
    flat = torch.randn(pad.P)
    flat_pad = pad.scatter_params(flat, pad.new_padded_params())
    assert flat_pad.numel() == pad.Pp and int((flat_pad != 0).sum()) == pad.P
    x = torch.randn(5, d)
    t = torch.full((5, 1), 0.3)
    z = mlp(flat, d, H, torch.cat([t, x], 1))
    xp = pad.last_dim(x)
    xp[:, d:] = torch.randn(5, dp - d)                 # padded state components carry noise; they must not matter
    zp = mlp(flat_pad, dp, Hp, torch.cat([t, xp], 1))
    assert torch.allclose(zp[:, :d], z, atol=1e-6)
    assert float(zp[:, d:].abs().max()) == 0.0
    g_pad = torch.randn(pad.Pp)
    g = pad.gather_grad(g_pad, torch.empty(pad.P))
    assert torch.equal(g, g_pad[pad.idx])
    ident = shapes.ParamPad(100, 64, 100, 64, torch.device('cpu'))
    assert ident.identity and ident.scatter_params(flat, None) is flat


def densenet(flat, d, H, x_t):
    """DenseNet(d+1 -> 1, arch [H, H]), relu^2, weights (in, out), input [x, t] (time last)."""
    DI = d + 1
    o = 0
    W1 = flat[o:o + DI * H].view(DI, H); o += DI * H
    b1 = flat[o:o + H]; o += H
    W2 = flat[o:o + (DI + H) * H].view(DI + H, H); o += (DI + H) * H
    b2 = flat[o:o + H]; o += H
    W3 = flat[o:o + DI + 2 * H].view(DI + 2 * H, 1); o += DI + 2 * H
    b3 = flat[o:o + 1]; o += 1
    assert o == flat.numel()
    act = lambda z: torch.relu(z) ** 2
    h1 = act(x_t @ W1 + b1)
    c1 = torch.cat([x_t, h1], 1)
    h2 = act(c1 @ W2 + b2)
    return torch.cat([c1, h2], 1) @ W3 + b3


def test_padded_value_net_equals_real_value_net():
    torch.manual_seed(1)
    d, H, dp, Hp = 7, 20, 10, 24
    pad = shapes.GenParamPad(d, H, dp, Hp, torch.device('cpu'))
    flat = 0.3 * torch.randn(pad.P)
    flat_pad = pad.scatter_params(flat, pad.new_padded_params())
    assert flat_pad.numel() == pad.Pp and int((flat_pad != 0).sum()) == pad.P
    x, t = torch.randn(6, d), torch.rand(6, 1)
    v = densenet(flat, d, H, torch.cat([x, t], 1))
    xp = pad.last_dim(x)
    xp[:, d:] = torch.randn(6, dp - d)                  # noise in the padded state components must not matter
    vp = densenet(flat_pad, dp, Hp, torch.cat([xp, t], 1))
    assert torch.allclose(vp, v, atol=1e-5)
    cands = shapes.gen_candidates(20, 30)
    assert cands and cands[0] == (32, 32)
    assert shapes.gen_candidates(100, 64)[0] == (100, 64)
    assert shapes.gen_candidates(101, 64) == []


def test_chunk_scratch_covers_every_chunk_size():
    """the forward grid of psp_hjb_query is not monotone in K_local (K = 1040: feature-split kernel, 65
    workgroups; K = 1024: quad kernel, 256 workgroups on a 256-CU device), so the scratch shared by the chunks of a K-chunked
    plan is the field-wise maximum over the distinct chunk sizes (plan_native.chunk_scratch_sizes)."""
    from path_space_pde_solver_amd import plan_native
    cfg = nat.HjbConfig()
    cfg.d, cfg.H, cfg.K_local, cfg.N, cfg.K_global = 100, 64, 1040, 10, 2064
    cfg.drift_kind, cfg.sigma_kind, cfg.term_kind = nat.DRIFT_DENSE, nat.SIGMA_DENSE, nat.TERM_LINEAR
    cfg.adaptive, cfg.store_path, cfg.noise_mode = 1, 1, nat.NOISE_PHILOX
    big = nat.query(cfg)
    cfg.K_local = 1024
    small = nat.query(cfg)
    if small.fwd_partial_bytes <= big.fwd_partial_bytes:        # (a device with another CU count: the premise does not hold)
        return
    cfg.K_local = 1040
    both = plan_native.chunk_scratch_sizes(cfg, {1040, 1024})
    assert both.fwd_partial_bytes == small.fwd_partial_bytes > big.fwd_partial_bytes
    assert both.path_bytes == big.path_bytes >= small.path_bytes
    assert both.grad_partial_bytes == max(big.grad_partial_bytes, small.grad_partial_bytes)
