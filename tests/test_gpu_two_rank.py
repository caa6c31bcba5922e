"""Multi-rank native plan on real hardware: two processes share cuda:0 (gloo carries the two
collectives through the host), trajectories sharded by contiguous blocks with global Philox /
reference-noise indexing.  The sharded run must reproduce the single-process run: SURVEY.md 8e
parity criterion (<= 1e-6 relative; differences come only from fp32/fp64 summation order).
RCCL itself cannot be rehearsed with one GPU; the code path differs only in the backend name."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CFG = dict(lr=1e-3, L=3, K=512, delta_t=0.01, loss_method="log-variance", time_approx="inner",
           adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False,
           seed=42, backend="native", widths=(64, 64))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _train(noise, out=None, rank=0, detach=True, **extra):
    sys.path.insert(0, ROOT)
    import path_space_pde_solver_amd as psp
    dev = torch.device("cuda:0")
    prob = psp.LLGC(d=100, off_diag=0.01, T=0.2, seed=42, device=dev)
    model = psp.Solver("two-rank", prob, device=dev, noise=noise, **dict(CFG, detach_forward=detach), **extra)
    model.train()
    assert model.plan_name == "native"
    res = dict(loss=model.loss_log, params=torch.cat([p.detach().reshape(-1).cpu() for p in model.z_n.parameters()]),
               K_local=model._native_plan.K_local, k_offset=model._native_plan.k_offset)
    if out is not None and rank == 0:
        torch.save(res, out)
    return res


def _worker(rank, world, port, noise, out, detach=True, extra=None):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _train(noise, out, rank, detach, **(extra or {}))
    assert res["K_local"] == CFG["K"] // world and res["k_offset"] == rank * res["K_local"]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("noise", ["philox", "reference"])
def test_two_ranks_match_one_rank(tmp_path, noise):
    out = os.path.join(str(tmp_path), "two.pt")
    mp.spawn(_worker, args=(2, _free_port(), noise, out), nprocs=2, join=True)
    two = torch.load(out)
    one = _train(noise)
    for a, b in zip(two["loss"], one["loss"]):
        assert abs(a - b) <= 1e-6 * abs(b), (two["loss"], one["loss"])
    err = float((two["params"] - one["params"]).abs().max())
    assert err <= 2e-6, err


def test_two_ranks_match_one_rank_attached(tmp_path):
    """Gradients through the state path (adjoint sweep): the per-trajectory weights use the GLOBAL mean of D."""
    out = os.path.join(str(tmp_path), "two_att.pt")
    mp.spawn(_worker, args=(2, _free_port(), "philox", out, False), nprocs=2, join=True)
    two = torch.load(out)
    one = _train("philox", detach=False)
    for a, b in zip(two["loss"], one["loss"]):
        assert abs(a - b) <= 1e-6 * abs(b), (two["loss"], one["loss"])
    err = float((two["params"] - one["params"]).abs().max())
    assert err <= 2e-6, err


@pytest.mark.parametrize("mode,detach", [("two_gradient", True), ("recompute", True), ("recompute", False)])
def test_two_ranks_with_k_chunking_match_one_rank_without(tmp_path, mode, detach):
    """Sharding AND K-chunking together: every rank processes its 256 trajectories in two chunks (global k_offset per chunk);
    the result equals the single-process, resident-store run."""
    out = os.path.join(str(tmp_path), "two_chunk.pt")
    extra = dict(path_chunks=2, chunk_mode=mode)
    mp.spawn(_worker, args=(2, _free_port(), "philox", out, detach, extra), nprocs=2, join=True)
    two = torch.load(out)
    one = _train("philox", detach=detach)
    for a, b in zip(two["loss"], one["loss"]):
        assert abs(a - b) <= 1e-6 * abs(b), (two["loss"], one["loss"])
    err = float((two["params"] - one["params"]).abs().max())
    assert err <= 5e-6, err


# ---- the other native plans: DenseNet controls (time_approx='outer') and GeneralSolver on a bounded domain -----------------
def _train_other(kind, noise, out=None, rank=0):
    sys.path.insert(0, ROOT)
    import numpy as np
    import path_space_pde_solver_amd as psp
    dev = torch.device("cuda:0")
    if kind == "outer":
        prob = psp.LLGC(d=20, off_diag=0.05, T=0.1, seed=42, device=dev)
        model = psp.Solver("two-rank-outer", prob, device=dev, noise=noise, lr=1e-3, L=3, K=256, delta_t=0.01,
                           loss_method="log-variance", time_approx="outer", adaptive_forward_process=True, detach_forward=True,
                           u_l2_error_flag=False, verbose=False, seed=42, backend="native")
        model.train()
        assert model.plan_name == "native"
        plan = model._native_plan
        params = torch.cat([p.detach().reshape(-1).cpu() for net in model.z_n for p in net.parameters()])
    elif kind == "value":
        prob = psp.DoubleWell_multidim(d=10, d_1=5, d_2=5, T=0.2, eta=0.5, kappa=2.0, device=dev)
        model = psp.Solver("two-rank-value", prob, device=dev, noise=noise, lr=2e-3, L=3, K=256, delta_t=0.01,
                           approx_method="value_function", loss_method="log-variance", time_approx="inner",
                           adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42,
                           backend="native")
        model.train()
        assert model.plan_name == "native"
        plan = model._native_plan
        # without the output bias b3: V's additive constant cancels in D = Y_N - g and in every V(X_n) - Y_n, so its gradient is
        # rounding noise and Adam moves it by +-lr per step whatever the noise's sign (in the reference as well)
        params = torch.cat([p.detach().reshape(-1).cpu() for p in model.y_n[0].parameters()])[:-1]
    else:
        prob = psp.ExponentialOnSphereNonlinearParabolic(d=6, T=0.5, alpha=0.3, device=dev)
        model = psp.GeneralSolver(prob, "two-rank-sphere", seed=42, delta_t=0.01, N=20, lr=1e-3, L=3, K=256, K_boundary=16,
                                  alpha=[1.0, 1.0, 1.0], loss_method="diffusion", verbose=False, device=dev, backend="native",
                                  noise=noise)
        np.random.seed(0)
        model.train()
        assert model.plan_name == "native"
        plan = model._gen_plan
        params = torch.cat([p.detach().reshape(-1).cpu() for p in model.V.parameters()])
    res = dict(loss=list(model.loss_log), params=params, K_local=plan.K_local, K_log=list(getattr(model, "K_log", [])))
    if out is not None and rank == 0:
        torch.save(res, out)
    return res


def _worker_other(rank, world, port, kind, noise, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _train_other(kind, noise, out, rank)
    assert res["K_local"] == 256 // world
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,noise", [("outer", "philox"), ("outer", "reference"), ("sphere", "reference"), ("sphere", "philox"),
                                        ("value", "philox"), ("value", "reference")])
def test_two_ranks_match_one_rank_other_plans(tmp_path, kind, noise):
    """Sharded DenseNet-control and bounded-domain GeneralSolver runs reproduce the single-process run (global noise
    indexing, global loss sums, summed gradients, and -- for the sphere with reference noise -- the globally agreed
    number of noise draws the host consumes)."""
    out = os.path.join(str(tmp_path), "two_%s_%s.pt" % (kind, noise))
    mp.spawn(_worker_other, args=(2, _free_port(), kind, noise, out), nprocs=2, join=True)
    two = torch.load(out)
    one = _train_other(kind, noise)
    assert two["K_log"] == one["K_log"]
    for a, b in zip(two["loss"], one["loss"]):
        assert abs(a - b) <= 2e-6 * abs(b), (two["loss"], one["loss"])
    err = float((two["params"] - one["params"]).abs().max())
    assert err <= 5e-6, err
