"""Trajectory sharding across ranks.

Trajectories are independent inside an iteration until the terminal reduction, so the batch
is split into contiguous blocks, one per rank.  Two collectives per iteration:

  1. SUM all-reduce of (sum_k D_k, sum_k D_k^2): the log-variance loss
     mean(D^2) - mean(D)^2 (reference solver.py:167-168) needs the GLOBAL mean -- averaging
     per-shard variances would drop the between-shard variance of the means;
  2. SUM all-reduce of the flat parameter gradient, after which every rank applies the
     identical Adam step.

The helpers below are the only place that arithmetic lives; plan_native.py uses them on the
device (RCCL) and tests/test_sharding_gloo.py exercises them on CPU with gloo, world_size 2.
"""
import torch


# bench.py's 1-rank reference run inside an N-rank job sets this: plans built meanwhile see a world of one
force_single = False


def dist_info():
    import torch.distributed as dist
    if force_single:
        return None, 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def shard_bounds(K, rank, world):
    """Contiguous block [lo, hi) of global trajectory indices owned by `rank`."""
    if K % world != 0:
        raise ValueError('K=%d is not divisible by world_size=%d' % (K, world))
    per = K // world
    return rank * per, (rank + 1) * per


def shard_bounds_ragged(K, rank, world):
    """The same split for a batch size the world size need not divide (the 'two_spheres' rejection sampler of the diffusion-loss
    solvers changes K every iteration, reference solver.py:1048-1052): equal to shard_bounds when it does."""
    return (K * rank) // world, (K * (rank + 1)) // world


# bench.py sets this to a list to collect (start_event, end_event, bytes) of every device all-reduce: HIP events on the
# launch stream, which waits for the collective (torch's synchronous all_reduce makes the current stream wait on RCCL's)
coll_events = None


def allreduce_sum_(t):
    """In-place SUM all-reduce when a process group is initialised; no-op otherwise."""
    dist, _, world = dist_info()
    if world > 1:
        if coll_events is not None and t.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(t)
            e1.record()
            coll_events.append((e0, e1, t.numel() * t.element_size()))
        else:
            dist.all_reduce(t)
    return t


def loss_from_sums(sums, K_global, loss_method):
    """Loss value from GLOBAL (sum D, sum D^2) (fp64 tensor of 2)."""
    K = float(K_global)
    if loss_method == 'log-variance':
        return sums[1] / K - (sums[0] / K) ** 2
    if loss_method == 'moment':
        return sums[1] / K
    if loss_method == 'relative_entropy':      # the forward kernel accumulates D = -(Zsum + g(X_N)) (include/psp.h)
        return -sums[0] / K
    raise ValueError(loss_method)


def loss_weights(D_local, sums, K_global, loss_method):
    """w_k = dLoss/dD_k for this rank's trajectories, from GLOBAL sums:
    log-variance (2/K)(D_k - mean D), moment (2/K) D_k.  The HIP backward kernel applies the
    same formula (csrc/hjb_kernels.h, hjb_bwd_kernel)."""
    K = float(K_global)
    if loss_method == 'log-variance':
        return (2.0 / K) * (D_local - (sums[0] / K).to(D_local.dtype))
    if loss_method == 'moment':
        return (2.0 / K) * D_local
    raise ValueError(loss_method)


def y0_gradient(sums, K_global, loss_method, w_local=None):
    """d loss / d Y_0 = sum_k dLoss/dY_k over ALL ranks (Y = y_0(X) + ..., solver.py:372-373): (2/K) sum D for moment,
    exactly 0 for log-variance and relative entropy; for the losses whose weights are formed on the host side (variance,
    cross_entropy: w_local = this rank's dLoss/dY_k) the all-reduced sum of the weights."""
    if loss_method == 'moment':
        return (2.0 / float(K_global)) * sums[0]
    if w_local is not None and loss_method not in ('log-variance', 'relative_entropy'):
        tot = w_local.double().sum().reshape(1)
        allreduce_sum_(tot)
        return tot[0]
    return torch.zeros((), dtype=sums.dtype, device=sums.device)


# ---- Adam state hand-over between a native plan and the nets' own torch optimisers -----------------------------------
def adam_state_export(params, m_flat, v_flat, step, optim):
    """Mirrors a plan's flat Adam moments into `optim.state` (torch.optim.Adam layout: step / exp_avg / exp_avg_sq per
    parameter, in `params` order), so that a later torch-side step, a state_dict() or a composite-plan run continues the same
    optimiser instead of restarting it -- the reference keeps this state in phi.optim (function_space.py:185)."""
    if optim is None or step <= 0:
        return
    off = 0
    for p in params:
        n = p.numel()
        st = optim.state[p]
        st['step'] = torch.tensor(float(step))
        st['exp_avg'] = m_flat[off:off + n].view(p.shape).clone()
        st['exp_avg_sq'] = v_flat[off:off + n].view(p.shape).clone()
        off += n


def adam_state_import(params, m_flat, v_flat, optim):
    """The reverse at plan construction: if the net's optimiser already carries Adam state (a previous torch run, a loaded
    checkpoint), the plan continues from it.  Returns the step count (0: fresh)."""
    if optim is None:
        return 0
    steps, off = [], 0
    for p in params:
        st = optim.state.get(p, {})
        n = p.numel()
        if 'exp_avg' in st and 'exp_avg_sq' in st:
            m_flat[off:off + n].copy_(st['exp_avg'].reshape(-1).to(m_flat.device))
            v_flat[off:off + n].copy_(st['exp_avg_sq'].reshape(-1).to(v_flat.device))
            steps.append(int(float(st.get('step', 0))))
        off += n
    return max(steps) if steps else 0
