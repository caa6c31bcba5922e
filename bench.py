#!/usr/bin/env python3
"""Benchmark of the path-space hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one full training iteration of Solver.train (forward SDE rollout over N_t time
steps, log-variance loss, analytic backward, Adam) on synthetic d-dimensional HJB data with
random-init weights.  Metric = trajectory-timesteps/s = K_traj * N_t * steps / wall time,
inputs resident in HBM, on-device Philox noise.  For --gpus N > 1 the script is
launched by torch.distributed.run (one rank per GPU, RCCL); every rank keeps the same
per-GPU trajectory count (weak scaling) and the value is the whole-job aggregate.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- dominant kernel: algorithmic flops per launch / mean launch time (HIP events
                  recorded on the launch stream inside the timed region) against the fp32
                  MFMA peak of MI355X (157.3 TFLOP/s).
  cpu_baseline -- the CPU oracle (a port of the reference algorithm, oracle/) timed on the
                  host cores of this box on a bounded sample of the same workload; reported
                  baseline, not the optimisation target.  Rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    # BASELINE.json north-star target shape: d=100 HJB (LLGC, dense A and B), K=65536, N=100, 2x64 MLP
    "hjb_llgc_d100_K65536_N100_h64": dict(d=100, H=64, K=65536, T=1.0, dt=0.01, off_diag=0.01),
    # BASELINE.json configs[1]
    "hjb_llgc_d100_K1024_N50_h64": dict(d=100, H=64, K=1024, T=0.5, dt=0.01, off_diag=0.01),
    "hjb_llgc_d100_K4096_N50_h64": dict(d=100, H=64, K=4096, T=0.5, dt=0.01, off_diag=0.01),
    "hjb_llgc_d100_K8192_N50_h64": dict(d=100, H=64, K=8192, T=0.5, dt=0.01, off_diag=0.01),
    "hjb_llgc_d100_K16384_N50_h64": dict(d=100, H=64, K=16384, T=0.5, dt=0.01, off_diag=0.01),
    # structured variant A=-I, B=I (reported separately)
    # opt-in mode: control-net products of the forward rollout on bf16 MFMA (SURVEY 8d "bf16-MLP runs"); NOT the headline
    "hjb_llgc_d100_K65536_N100_h64_bf16mlp": dict(d=100, H=64, K=65536, T=1.0, dt=0.01, off_diag=0.01, mlp="bf16"),
    # the reference's constructor default: time_approx='outer', one DenseNet(d -> d, arch [30, 30]) per time step (solver.py:88)
    # (T = 0.5: over T = 1 the reference ALGORITHM itself blows up at this size -- relu^2 nets at their default init are
    #  expansive, single paths overflow; the CPU oracle shows the same [316, 4.0e5, nan] at K = 8192, tools/check_outer_d100.py)
    "hjb_llgc_d100_K65536_N50_outer_h30": dict(d=100, H=30, K=65536, T=0.5, dt=0.01, off_diag=0.01, outer=True),
    "hjb_llgc_d100_K65536_N100_h64_diag": dict(d=100, H=64, K=65536, T=1.0, dt=0.01, off_diag=0.0),
    # BASELINE.json configs[3]: d=200, K=262144 over 8 GPUs = 32768 per GPU, N=100 (wide kernel family)
    "hjb_llgc_d200_K32768_N100_h64": dict(d=200, H=64, K=32768, T=1.0, dt=0.01, off_diag=0.1 / 200 ** 0.5),
    # BASELINE.json configs[4] shape: d=500, N=200; K per GPU reduced from 131072 to 16384 (path store 15 GB)
    "hjb_llgc_d500_K16384_N200_h64": dict(d=500, H=64, K=16384, T=2.0, dt=0.01, off_diag=0.1 / 500 ** 0.5),
    "hjb_llgc_d500_K16384_N200_h64_diag": dict(d=500, H=64, K=16384, T=2.0, dt=0.01, off_diag=0.0),
}
GENERAL_WORKLOADS = {
    # BASELINE.json configs[2] shape in fp32: d=100 diffusion loss, K=65536, N=100, V = DenseNet(101 -> 1, [64, 64])
    "diffusion_dw_d100_K65536_N100_h64": dict(d=100, H=64, K=65536, N=100, T=0.3, dt=0.001, loss="diffusion"),
    "bsde_dw_d100_K65536_N100_h64": dict(d=100, H=64, K=65536, N=100, T=0.1, dt=0.001, loss="BSDE"),
    # BASELINE.json configs[2] as written: value-net products of the forward rollout on bf16 MFMA (fp32 state / accumulate)
    "diffusion_dw_d100_K65536_N100_h64_bf16": dict(d=100, H=64, K=65536, N=100, T=0.3, dt=0.001, loss="diffusion", mlp="bf16"),
    "diffusion_dw_d100_K65536_N100_h64_bf16fwd": dict(d=100, H=64, K=65536, N=100, T=0.3, dt=0.001, loss="diffusion", mlp="bf16_fwd"),
}
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0


def alg_flops_dense_control(d, H, dense):
    """Same accounting for a DenseNet(d -> d, [H, H]) control: forward products W1, W2 = [W2x; W2h], W3 = [W3x; W3h1; W3h2];
    backward = weight gradients of all of them + the two adjoint products through W3h, W2h (no input gradient)."""
    f_fwd = 2 * (d * H + (d + H) * H + (d + 2 * H) * d)
    f_bwd = f_fwd + 2 * (2 * H * d + H * H)
    f_sde = 2 * d * d * (3 if dense else 0)
    return dict(fwd_kernel=f_fwd + f_sde + 12 * d, bwd_kernel=f_bwd, total=f_fwd + f_bwd + f_sde + 12 * d)


def alg_flops_per_traj_step(d, H, dense):
    """algorithmic flops per trajectory-timestep of the training iteration."""
    f_fwd = 2 * ((d + 1) * H + H * H + H * d)
    f_bwd = f_fwd + 2 * (H * H + H * d)
    f_sde = 2 * d * d * (3 if dense else 0)
    return dict(fwd_kernel=f_fwd + f_sde + 12 * d, bwd_kernel=f_bwd, total=f_fwd + f_bwd + f_sde + 12 * d)


def cpu_baseline(w, seconds_budget=20.0):
    """Times oracle/pathspace_oracle.py (torch-CPU port of the reference iteration, including its
    duplicate control evaluation and dense sigma products) on K=4096 trajectories of the workload."""
    from oracle import pathspace_oracle as orc
    Kc = min(4096, w["K"])
    prob = orc.make_problem("LLGC", d=w["d"], off_diag=w["off_diag"], T=w["T"], seed=42)
    # a 1-GPU box exposes a 16-core CPU share; more torch threads than that only adds contention
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    cfg = orc.HJBConfig(K=Kc, delta_t=w["dt"], lr=1e-3, L=1, seed=42, adaptive_forward_process=True,
                        detach_forward=True)
    z = orc.TanhMLP(w["d"] + 1, w["d"], 1e-3, seed=123, widths=(w["H"], w["H"]))
    _, y0, N = orc.hjb_build(prob, cfg)
    orc.hjb_train(prob, cfg, step_models=(z, y0, N))          # warm-up iteration
    iters, t0 = 0, time.time()
    while True:
        orc.hjb_train(prob, cfg, step_models=(z, y0, N))
        iters += 1
        el = time.time() - t0
        if el > seconds_budget or iters >= 24:        # ~13 s of CPU work on the 16-core share of a GPU box
            break
    rate = Kc * N * iters / el
    return dict(value=rate, unit="trajectory-timesteps/s", cores=threads, kind="port",
                sample="oracle hjb_train, K=%d of %d trajectories, N=%d, %d iterations, %.1f s, torch %s"
                       % (Kc, w["K"], N, iters, el, torch.__version__))


def loss_rel_err_vs_cpu(psp, dev, w):
    """BASELINE.json's second metric: |L_gpu - L_ref| / |L_ref| per iteration on FIXED SEEDS -- the native plan with the
    reference's host-generated noise (torch CPU generator, seed 42) against the CPU oracle consuming the same stream, on the
    workload's problem at K = 1024 trajectories, 3 iterations.  Part of the cpu_baseline leg (the only place bench.py may
    use oracle/)."""
    from oracle import pathspace_oracle as orc
    K, L = 1024, 3
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    oprob = orc.make_problem("LLGC", d=w["d"], off_diag=w["off_diag"], T=w["T"], seed=42)
    ocfg = orc.HJBConfig(K=K, delta_t=w["dt"], lr=1e-3, L=L, seed=42, adaptive_forward_process=True, detach_forward=True)
    z = orc.TanhMLP(w["d"] + 1, w["d"], 1e-3, seed=123, widths=(w["H"], w["H"]))
    _, y0, N = orc.hjb_build(oprob, ocfg)
    ref = orc.hjb_train(oprob, ocfg, step_models=(z, y0, N))["loss_log"]
    prob = psp.LLGC(d=w["d"], off_diag=w["off_diag"], T=w["T"], seed=42, device=dev)
    model = psp.Solver("bench-parity", prob, lr=1e-3, L=L, K=K, delta_t=w["dt"], loss_method="log-variance",
                       time_approx="inner", adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False,
                       verbose=False, seed=42, device=dev, backend="native", noise="reference", widths=(w["H"], w["H"]))
    model.train()
    errs = [abs(a - b) / abs(b) for a, b in zip(model.loss_log, ref)]
    return dict(value=max(errs), per_iteration=errs, tolerance=1e-4,
                case="LLGC d=%d, K=%d, N=%d, %d iterations, seed 42, reference noise stream" % (w["d"], K, N, L))


def secondary(psp, dev, w, steps=100, warmup=10):
    prob = psp.LLGC(d=w["d"], off_diag=w["off_diag"], T=w["T"], seed=42, device=dev)
    model = psp.Solver("bench-cfg1", prob, lr=1e-3, L=steps + warmup, K=w["K"], delta_t=w["dt"],
                       loss_method="log-variance", time_approx="inner", adaptive_forward_process=True,
                       detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42, device=dev,
                       backend="native", noise="philox", widths=(w["H"], w["H"]))
    plan = model._choose_plan()
    losses = torch.zeros(steps + warmup, dtype=torch.float32, device=dev)
    for l in range(warmup):
        plan.iteration(l, losses)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for l in range(warmup, warmup + steps):
        plan.iteration(l, losses)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"value": w["K"] * model.N * steps / el, "unit": "trajectory-timesteps/s", "ms_per_step": 1e3 * el / steps,
            "steps": steps, "K": w["K"], "N": model.N}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="hjb_llgc_d100_K65536_N100_h64",
                    choices=sorted(WORKLOADS) + sorted(GENERAL_WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] side measurement")
    args = ap.parse_args()

    import path_space_pde_solver_amd as psp
    if args.workload in GENERAL_WORKLOADS:
        return main_general(args, psp)
    w = WORKLOADS[args.workload]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    # PSP_BENCH_REHEARSAL=1: all ranks share cuda:0 and talk over gloo -- a way to walk the N > 1 code path on a
    # one-GPU box (RCCL refuses two ranks on one device); numbers from such a run mean nothing
    rehearsal = os.environ.get("PSP_BENCH_REHEARSAL") == "1"
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    if args.gpus != world:
        print("note: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)"
              % (args.gpus, world), file=sys.stderr)

    K_global = w["K"] * world                           # weak scaling: fixed trajectories per GPU
    prob = psp.LLGC(d=w["d"], off_diag=w["off_diag"], T=w["T"], seed=42, device=dev)
    total = args.warmup + args.steps
    outer = bool(w.get("outer"))
    model = psp.Solver("bench", prob, lr=w.get("lr", 1e-3), L=total, K=K_global, delta_t=w["dt"], loss_method="log-variance",
                       time_approx="outer" if outer else "inner", adaptive_forward_process=True, detach_forward=True,
                       u_l2_error_flag=False, verbose=False, seed=42, device=dev, backend="native",
                       noise="philox", widths=(w["H"], w["H"]), mlp_dtype=w.get("mlp", "fp32"))
    if outer:                                            # the constructor builds arch [30, 30]; honour the workload's H
        model.z_n = [psp.DenseNet(d_in=w["d"], d_out=w["d"], lr=w.get("lr", 1e-3), arch=[w["H"], w["H"]], seed=42).to(dev)
                     for _ in range(model.N)]
        model.update_Phis()
    plan = model._choose_plan()
    assert model.plan_name == "native"
    N_t = model.N
    losses = torch.zeros(total, dtype=torch.float32, device=dev)
    for l in range(args.warmup):
        plan.iteration(l, losses)

    plan.events = []                                     # (fwd_start, fwd_end, bwd_start, bwd_end) per step
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for l in range(args.warmup, total):
        plan.iteration(l, losses)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    fwd_ms = sum(e[0].elapsed_time(e[1]) for e in plan.events) / max(1, len(plan.events))
    bwd_ms = sum(e[2].elapsed_time(e[3]) for e in plan.events) / max(1, len(plan.events))
    loss_vals = losses.cpu().tolist()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    dense = w["off_diag"] != 0.0
    fl = alg_flops_dense_control(w["d"], w["H"], dense) if outer else alg_flops_per_traj_step(w["d"], w["H"], dense)
    units_local = w["K"] * N_t                           # trajectory-timesteps per launch on one GPU
    # kernel names as they appear in rocprof: family 1 = hjb_kernels.h (hjbs_kernels.h forward when there are at most two
    # tiles per CU), family 2 = hjbw_kernels.h
    ntile = (plan.K_local + 15) // 16
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    if outer:
        fwd_name, bwd_name = "hjbd_fwd_kernel", ("hjbd_bwd_kernel" if plan.kernel_bwd else "library GEMMs")
    else:
        fwd_name = "hjbw_fwd_kernel" if plan.family == 2 else ("hjbs_fwd_kernel" if ntile <= 2 * cus else "hjb_fwd_kernel")
        bwd_name = "hjbw_bwd_kernel" if plan.family == 2 else "hjb_bwd2_kernel"
    bwd_dominant = bwd_ms >= fwd_ms
    dom = bwd_name if bwd_dominant else fwd_name
    dom_ms = max(bwd_ms, fwd_ms)
    dom_flops = (fl["bwd_kernel"] if bwd_dominant else fl["fwd_kernel"]) * units_local
    achieved = dom_flops / (dom_ms * 1e-3) / 1e12
    traffic = None                                   # HBM bytes per launch, from the committed PMC passes
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(args.workload, {}).get(dom)
    value = K_global * N_t * args.steps / elapsed
    out = {
        "metric": "trajectory-timesteps/sec (K*N/s), d=%d HJB log-variance training iteration" % w["d"],
        "value": value, "unit": "trajectory-timesteps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 control-net products in the forward rollout, f32 elsewhere" if w.get("mlp") == "bf16" else "f32",
        "data": "synthetic",
        "config": {"workload": args.workload, "problem": "LLGC", "d": w["d"], "K_per_gpu": w["K"],
                   "K_global": K_global, "N": N_t,
                   "mlp": ("%d x DenseNet %d-%d-%d-%d relu^2, one per time step (time_approx='outer')" % (N_t, w["d"], w["H"], w["H"], w["d"]))
                          if outer else "%d-%d-%d-%d tanh" % (w["d"] + 1, w["H"], w["H"], w["d"]),
                   "loss": "log-variance", "noise": "on-device Philox4x32-10",
                   "parallelism": "trajectory-sharded x%d" % world},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                     "alg_flops_per_traj_step": fl, "units_per_launch": units_local,
                     "fwd_kernel_ms": fwd_ms, "bwd_kernel_ms": bwd_ms,
                     "whole_step_tflops": fl["total"] * units_local / (1e-3 * (1e3 * elapsed / args.steps)) / 1e12},
        "loss_first_last": [loss_vals[0], loss_vals[-1]],
    }
    if world == 1 and args.workload == "hjb_llgc_d100_K65536_N100_h64" and not args.no_secondary:
        # BASELINE.json configs[1] (d=100, K=1024, N=50) measured in the same process, for readers who take
        # that as the quoted configuration (64 16-trajectory tiles: runs on the feature-split forward kernel)
        out["also_configs1_K1024_N50"] = secondary(psp, dev, WORKLOADS["hjb_llgc_d100_K1024_N50_h64"])
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w)
        out["loss_rel_err_vs_cpu_ref"] = loss_rel_err_vs_cpu(psp, dev, w)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def main_general(args, psp):
    """GeneralSolver (diffusion / BSDE loss) workloads: metric counts ACTIVE trajectory-timesteps
    (K_log, reference solver.py:1152), as SURVEY.md 8d prescribes."""
    w = GENERAL_WORKLOADS[args.workload]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    prob = psp.DoubleWell_multidim_for_general_solver(d=w["d"], d_1=w["d"] // 2, d_2=w["d"] - w["d"] // 2, T=w["T"],
                                                      eta=1, kappa=1, modus="HJB", device=dev)
    total = args.warmup + args.steps
    model = psp.GeneralSolver(problem=prob, name="bench", seed=42, delta_t=w["dt"], N=w["N"], lr=1e-3, L=total,
                              K=w["K"] * world, K_boundary=50, alpha=[1.0, 1.0, 1.0], loss_method=w["loss"],
                              verbose=False, device=dev, backend="native", noise="philox", mlp_dtype=w.get("mlp", "fp32"))
    model.V = psp.DenseNet(d_in=w["d"] + 1, d_out=1, lr=1e-3, arch=[w["H"], w["H"]], seed=42).to(dev)
    plan = model._choose_plan()
    assert model.plan_name == "native"
    for l in range(args.warmup):
        plan.iteration(l)
    plan.events = []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    counts, losses = [], []
    for l in range(args.warmup, total):
        loss, kc = plan.iteration(l)
        counts.append(kc)
        losses.append(loss)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    active = float(torch.stack(counts).sum().item())
    fwd_ms = sum(e[0].elapsed_time(e[1]) for e in plan.events) / max(1, len(plan.events))
    bwd_ms = sum(e[2].elapsed_time(e[3]) for e in plan.events) / max(1, len(plan.events))
    # algorithmic flops per LAUNCHED trajectory-timestep: value net F = 2[(d+1)H + (d+1+H)H + (d+1+2H)],
    # its input gradient ~ F (reverse sweep), and the double-backward of both ~ 2 x that: forward kernel
    # 2F (+ F for the tangent part it pre-computes), backward kernel 3F  -> 6F per unit in total
    F = 2 * ((w["d"] + 1) * w["H"] + (w["d"] + 1 + w["H"]) * w["H"] + (w["d"] + 1 + 2 * w["H"]))
    units = w["K"] * w["N"]
    dom, dom_ms, dom_fl = ("gen_bwd_kernel", bwd_ms, 3 * F) if bwd_ms >= fwd_ms else ("gen_fwd_kernel", fwd_ms, 3 * F)
    achieved = dom_fl * units / (dom_ms * 1e-3) / 1e12
    if rank == 0:
        out = {"metric": "active trajectory-timesteps/sec, d=100 %s loss training iteration" % w["loss"],
               "value": active / elapsed, "unit": "trajectory-timesteps/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None,
               "dtype": {"bf16": "bf16 MFMA operands in both rollout kernels, f32 state / accumulate / element-wise",
                         "bf16_fwd": "bf16 MFMA operands in the forward rollout, f32 state / accumulate / backward"}.get(w.get("mlp"), "f32"),
               "data": "synthetic",
               "config": {"workload": args.workload, "problem": "DoubleWell_multidim_for_general_solver",
                          "d": w["d"], "K_per_gpu": w["K"], "N": w["N"], "V": "DenseNet %d-%d-%d-1" % (w["d"] + 1, w["H"], w["H"]),
                          "loss": w["loss"], "active_fraction": active / (w["K"] * world * w["N"] * args.steps)},
               "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                            "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                            "traffic": (json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
                                        .get(args.workload, {}).get(dom)
                                        if os.path.exists(os.path.join(ROOT, "profiles", "traffic.json")) else None),
                            "alg_flops_per_launched_unit": {"value_net_F": F, "fwd_kernel": 3 * F, "bwd_kernel": 3 * F},
                            "units_per_launch": units, "fwd_kernel_ms": fwd_ms, "bwd_kernel_ms": bwd_ms},
               "launched_units_per_s": w["K"] * world * w["N"] * args.steps / elapsed,
               "loss_first_last": [float(losses[0]), float(losses[-1])]}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
