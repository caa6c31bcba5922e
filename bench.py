#!/usr/bin/env python3
"""Benchmark of the path-space hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one full training iteration of Solver.train (forward SDE rollout over N_t time
steps, log-variance loss, analytic backward, Adam) on synthetic d-dimensional HJB data with
random-init weights.  Metric = trajectory-timesteps/s = K_traj * N_t * steps / wall time,
inputs resident in HBM, on-device Philox noise.

--gpus N > 1: one rank per GPU over RCCL.  Either the caller launches the ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, WORLD_SIZE set), or
bench.py does it itself: with WORLD_SIZE unset the parent process -- which never touches the GPU --
starts that same command as a child and relays its exit code; rank 0 prints the JSON line.
Weak-scaling workloads keep the per-GPU trajectory count, strong-scaling ones (K_global in the
workload) split a fixed global batch; `value` is always the whole-job aggregate.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- dominant kernel (the longer of the two rollout kernels), timed with HIP events recorded on the launch
                  stream inside the timed region.  Two terms, both from SURVEY.md 8(d)'s ALGORITHMIC work per
                  trajectory-timestep x the units one launch processes / the kernel's mean duration:
                    hbm_term   frac_alg_bytes = (8 d + 8) B per unit against 8 TB/s (the state streamed once per step);
                               next to it frac_path_store prices the bytes this design really moves (the path store the
                               forward writes and the backward reads) -- arithmetically right, but those bytes are the
                               design's own, not the algorithm's;
                    mfma_term  minimal algorithmic flops (m = 2 dense d x d products per step) against the peak of the
                               pipe the kernel computes on: 157.3 TFLOP/s for v_mfma_f32_16x16x4_f32 kernels, 2500 / 3 =
                               833 TFLOP/s for the split-product kernels (three f16 flops per algorithmic fp32 flop), a
                               flop-weighted blend for bf16 workloads; frac_issued prices the instructions really issued.
                  `bound` / `achieved` / `peak` / `frac` are those of the LARGER term ("hbm" -> the algorithmic-bytes
                  term).  `binder` names what the kernel actually waits on when neither term exceeds 0.5 (e.g. "valu":
                  SQ_INSTS_VALU per tile-step from the committed PMC passes, profiles/r3_counters.json); `traffic` = HBM
                  bytes per launch from the same passes (2 x FETCH_SIZE + WRITE_SIZE, calibrated for gfx950).
  also_sustained -- the same iteration run back to back for >= 3 s after the timed region (the timed region of the
                  default run is only steps x 4 ms): steady-state rate, clocks and power settled.
  collectives  -- N > 1: HIP-event time of the two all-reduces per iteration and their share of the step.
  parity_vs_1rank -- N > 1: before timing, iteration 0 of a 16384-trajectories-per-rank batch on the N ranks against the
                  same global batch on rank 0 alone (Philox noise is indexed by the global trajectory: equal to summation
                  order); rccl_ranks_seen = what a SUM all-reduce of ones returns.
  cpu_baseline -- the CPU oracle (a port of the reference algorithm, oracle/) timed on the host
                  cores of this box on bounded samples (configs[0], configs[1] and a K = 4096 cut
                  of the workload; all cores and 1 thread).  Rank 0, N = 1 only.
"""
import argparse
import json
import os
import socket
import subprocess
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
    # the same shapes on the fp32-MFMA kernels (Solver(mlp_dtype='fp32')); the default workloads run the split-product kernels
    "hjb_llgc_d100_K65536_N100_h64_fp32mfma": dict(d=100, H=64, K=65536, T=1.0, dt=0.01, off_diag=0.01, mlp="fp32"),
    "hjb_llgc_d200_K32768_N100_h64_fp32mfma": dict(d=200, H=64, K=32768, T=1.0, dt=0.01, off_diag=0.1 / 200 ** 0.5, mlp="fp32"),
    "hjb_llgc_d500_K16384_N200_h64_fp32mfma": dict(d=500, H=64, K=16384, T=2.0, dt=0.01, off_diag=0.1 / 500 ** 0.5, mlp="fp32"),
    # opt-in mode: control-net products of the forward rollout on bf16 MFMA (SURVEY 8d "bf16-MLP runs"); NOT the headline
    "hjb_llgc_d100_K65536_N100_h64_bf16mlp": dict(d=100, H=64, K=65536, T=1.0, dt=0.01, off_diag=0.01, mlp="bf16"),
    # the reference's constructor default: time_approx='outer', one DenseNet(d -> d, arch [30, 30]) per time step (solver.py:88)
    # (T = 0.5: over T = 1 the reference ALGORITHM itself blows up at this size -- relu^2 nets at their default init are
    #  expansive, single paths overflow; the CPU oracle shows the same [316, 4.0e5, nan] at K = 8192, tools/check_outer_d100.py)
    "hjb_llgc_d100_K65536_N50_outer_h30": dict(d=100, H=30, K=65536, T=0.5, dt=0.01, off_diag=0.01, outer=True),
    # structured variant A=-I, B=I (reported separately)
    "hjb_llgc_d100_K65536_N100_h64_diag": dict(d=100, H=64, K=65536, T=1.0, dt=0.01, off_diag=0.0),
    # BASELINE.json configs[3]: d=200, K=262144 over 8 GPUs = 32768 per GPU, N=100 (wide kernel family) -- weak form
    "hjb_llgc_d200_K32768_N100_h64": dict(d=200, H=64, K=32768, T=1.0, dt=0.01, off_diag=0.1 / 200 ** 0.5),
    # configs[3] as written: the GLOBAL batch K=262144 is fixed and split over the ranks (strong scaling); on one GPU the
    # whole batch runs with its 57 GB path store resident
    "hjb_llgc_d200_Kglobal262144_N100_h64": dict(d=200, H=64, K_global=262144, T=1.0, dt=0.01, off_diag=0.1 / 200 ** 0.5),
    # BASELINE.json configs[4] shape: d=500, N=200 -- K per GPU cut to 16384 (path store 15 GB, no recompute)
    "hjb_llgc_d256_K32768_N100_h64": dict(d=256, H=64, K=32768, T=1.0, dt=0.01, off_diag=0.1 / 256 ** 0.5),
    "hjb_llgc_d128_K65536_N100_h64": dict(d=128, H=64, K=65536, T=1.0, dt=0.01, off_diag=0.1 / 128 ** 0.5),
    "hjb_llgc_d500_K16384_N200_h64": dict(d=500, H=64, K=16384, T=2.0, dt=0.01, off_diag=0.1 / 500 ** 0.5),
    "hjb_llgc_d500_K16384_N200_h64_diag": dict(d=500, H=64, K=16384, T=2.0, dt=0.01, off_diag=0.0),
    # configs[4] at its per-GPU share of the 8-GPU job (K = 1048576 / 8) and at its full size on ONE GPU: the path store
    # (118 GB / 946 GB) is bounded by the K-chunked two-pass plan (plan_native.py, path_budget_bytes)
    "hjb_llgc_d500_K131072_N200_h64": dict(d=500, H=64, K=131072, T=2.0, dt=0.01, off_diag=0.1 / 500 ** 0.5,
                                           path_budget_gb=32),
    # ... and with the plan's own default budget (5/8 of the device memory): the per-GPU share fits the card whole (118 GB), one backward
    "hjb_llgc_d500_K131072_N200_h64_resident": dict(d=500, H=64, K=131072, T=2.0, dt=0.01, off_diag=0.1 / 500 ** 0.5),
    "hjb_llgc_d500_K1048576_N200_h64": dict(d=500, H=64, K=1048576, T=2.0, dt=0.01, off_diag=0.1 / 500 ** 0.5,
                                            path_budget_gb=32),
    # the headline shape forced through the chunked plan (recompute overhead measured against the resident-store run)
    "hjb_llgc_d100_K65536_N100_h64_chunk4": dict(d=100, H=64, K=65536, T=1.0, dt=0.01, off_diag=0.01, chunks=4),
}
GENERAL_WORKLOADS = {
    # BASELINE.json configs[2] shape in fp32: d=100 diffusion loss, K=65536, N=100, V = DenseNet(101 -> 1, [64, 64])
    "diffusion_dw_d100_K65536_N100_h64": dict(d=100, H=64, K=65536, N=100, T=0.3, dt=0.001, loss="diffusion"),
    "bsde_dw_d100_K65536_N100_h64": dict(d=100, H=64, K=65536, N=100, T=0.1, dt=0.001, loss="BSDE"),
    # BASELINE.json configs[2] as written: value-net products of the forward rollout on bf16 MFMA (fp32 state / accumulate)
    "diffusion_dw_d100_K65536_N100_h64_bf16": dict(d=100, H=64, K=65536, N=100, T=0.3, dt=0.001, loss="diffusion", mlp="bf16"),
    "diffusion_dw_d100_K65536_N100_h64_bf16fwd": dict(d=100, H=64, K=65536, N=100, T=0.3, dt=0.001, loss="diffusion", mlp="bf16_fwd"),
    # the ONE configuration the reference publishes a timing for (Allen-Cahn.ipynb:46-72, 86, 115: 0.31-0.35 s per iteration on an
    # unnamed CUDA GPU): AllenCahn d=100, T=0.3, K=200, N=25, dt=1e-3, V = DenseNet(101 -> 1, [110, 110, 50]),
    # alpha = [10, 1, 1], uniform_square -- the run-time-shaped value-net kernels (csrc/genl_kernels.h)
    "diffusion_allencahn_d100_K200_N25_a110": dict(d=100, arch=[110, 110, 50], K=200, N=25, T=0.3, dt=0.001, loss="diffusion",
                                                   problem="AllenCahn", alpha=[10.0, 1.0, 1.0], uniform_square=True,
                                                   boundary_distance=7.0, published_s_per_iter=(0.31, 0.35),
                                                   published_units_per_s=1.4e4),
    # the same net at a batch that fills the chip
    "diffusion_allencahn_d100_K16384_N25_a110": dict(d=100, arch=[110, 110, 50], K=16384, N=25, T=0.3, dt=0.001, loss="diffusion",
                                                     problem="AllenCahn", alpha=[10.0, 1.0, 1.0], uniform_square=True,
                                                     boundary_distance=7.0),
    # the reference's OTHER published timing: EllipticSolver, BSDE loss, committor function between two
    # spheres at d = 10, K = 200, N = 5000, dt = 1e-3, the notebook's own tanh(.)**2 net arch = [d + 10, d, d, d] -- 14.07 .. 28.46 s
    # per iteration (`Committor function.ipynb` cell 15-16: every trajectory runs until it leaves the annulus, 450 .. 1700 steps)
    "elliptic_committor_d10_K200": dict(d=10, arch=[20, 10, 10, 10], act="tanh2", K=200, N=5000, dt=0.001, loss="BSDE",
                                        problem="Committor", solver="elliptic", alpha=[0.01, 1.0],
                                        published_s_per_iter=(14.07, 28.46),
                                        published_source="experiments/diffusion-loss/Committor function.ipynb:375,686 (BASELINE.md section 1)"),
    # the same problem at a batch that fills the chip
    "elliptic_committor_d10_K65536": dict(d=10, arch=[20, 10, 10, 10], act="tanh2", K=65536, N=5000, dt=0.001, loss="BSDE",
                                          problem="Committor", solver="elliptic", alpha=[0.01, 1.0]),
}
# MI355X_MICROARCH.md, chip-level parameters
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
MFMA_F32_16x16x4_FLOP = 2 * 16 * 16 * 4          # one v_mfma_f32_16x16x4_f32
MFMA_BF16_16x16x32_FLOP = 2 * 16 * 16 * 32       # one v_mfma_f32_16x16x32_bf16


def _cdiv(a, b):
    return (a + b - 1) // b


def alg_flops_dense_control(d, H, dense, m=2):
    """DenseNet(d -> d, [H, H]) control: forward products W1, W2 = [W2x; W2h], W3 = [W3x; W3h1; W3h2]; backward = weight
    gradients of all of them + the two adjoint products through W3h, W2h (no input gradient)."""
    f_fwd = 2 * (d * H + (d + H) * H + (d + 2 * H) * d)
    f_bwd = f_fwd + 2 * (2 * H * d + H * H)
    f_sde = 2 * d * d * (m if dense else 0)
    return dict(fwd_kernel=f_fwd + f_sde + 12 * d, bwd_kernel=f_bwd, total=f_fwd + f_bwd + f_sde + 12 * d,
                mlp_fwd=f_fwd, sde=f_sde + 12 * d)


def alg_flops_per_traj_step(d, H, dense, m=2):
    """Algorithmic flops per trajectory-timestep of the training iteration.  SURVEY.md 8(d) counts m = 3 dense d x d
    products (A x, B c, B xi); B c dt + B xi sqrt(dt) = B (c dt + xi sqrt(dt)) is ONE product, so the minimum is m = 2."""
    f_fwd = 2 * ((d + 1) * H + H * H + H * d)
    f_bwd = f_fwd + 2 * (H * H + H * d)
    f_sde = 2 * d * d * (m if dense else 0)
    return dict(fwd_kernel=f_fwd + f_sde + 12 * d, bwd_kernel=f_bwd, total=f_fwd + f_bwd + f_sde + 12 * d,
                mlp_fwd=f_fwd, sde=f_sde + 12 * d)


def issued_mfma_per_tile_step(d_pad, H_pad, dense, family, bf16_mlp=False):
    """MFMA instructions one 16-trajectory tile issues per time step (padded kernel instance).  Counted from the kernels'
    loop bounds and confirmed by SQ_INSTS_MFMA at the headline shape (626 forward / 452 backward per tile-step,
    profiles/r1_pmc_summary.md).  Returns (fwd_f32, fwd_bf16, bwd_f32)."""
    DB, HB = _cdiv(d_pad, 16), _cdiv(H_pad, 16)
    # k-steps of a d-deep contraction: the narrow family stops at ceil(d / 4); the wide family's rolled loops run over whole
    # 16-feature blocks (GeoW::KP = 4 * DB)
    ks_d = 4 * DB if family == 2 else _cdiv(d_pad, 4)
    ks_h = 16 if family == 2 else _cdiv(H_pad, 4)
    net = ks_d * HB + ks_h * HB + ks_h * DB
    sde = 2 * ks_d * DB if dense else 0
    # backward: the wide family's role-specialised kernel (d <= 256, hjbw_bwd2_kernel) contracts over the real k-steps like the
    # narrow one; hjbw_bwd_kernel (d > 256) over whole 16-blocks
    ks_db = _cdiv(d_pad, 4) if (family == 2 and d_pad <= 256) else ks_d
    bwd = 4 * (DB * HB + HB * HB + HB * DB) + ks_db * HB + ks_h * HB
    if bf16_mlp:                                                       # 16x16x32: one k-step spans 32 features
        net_bf = _cdiv(d_pad, 32) * HB + _cdiv(H_pad, 32) * HB + _cdiv(H_pad, 32) * DB
        return sde, net_bf, bwd
    return net + sde, 0, bwd


def issued_mfma_x3(d_pad, H_pad, dense, family=1):
    """Split-product kernels (hjb_fwd_kernel<.., 2, ..>, hjb_bwd3_kernel): MFMA instructions per tile-step from SplitGeo in
    csrc/hjb_kernels.h -- a contraction over K features runs K // 32 steps of three v_mfma_f32_16x16x32_f16 per 16-row block;
    an odd trailing 16-feature block is one exact fp32 k-step when it holds at most four features, else three
    v_mfma_f32_16x16x16_f16.  Returns dicts {f16_32, f16_16, f32} for the forward and the backward."""
    DB, HB = _cdiv(d_pad, 16), _cdiv(H_pad, 16)

    def prod(K, MB):
        inb, ks = _cdiv(K, 16), _cdiv(K, 4)
        ns, odd = inb // 2, inb % 2
        nkr = ks - 8 * ns if odd else 0
        return {"f16_32": 3 * ns * MB, "f16_16": 3 * MB if (odd and nkr > 1) else 0, "f32": MB if (odd and nkr <= 1) else 0}

    def add(*ds):
        return {k: sum(d[k] for d in ds) for k in ("f16_32", "f16_16", "f32")}

    if family == 2:
        # wide family (hjbw_fwd_kernel<.., X3>): S-steps of 32 features over whole 16-blocks (KS8 = ceil(DB / 2)), three MFMAs per
        # (S, output block)
        ks8 = _cdiv(DB, 2)
        n = 3 * (ks8 * HB + 2 * HB + 2 * DB + (2 * ks8 * DB if dense else 0))
        bwd = None                                        # (d > 256: hjbw_bwd_x3_kernel, priced on the algorithmic flops only)
        if d_pad <= 256 and HB == 4:
            # hjbw_bwd2x_kernel (csrc/hjbwx_kernels.h), per sample block: producers W3^T G (prod(d, HB)) and W2^T dz2; consumers,
            # per PAIR of blocks, four waves x (2 DB + HB) tiles x three MFMAs
            bwd = add(prod(d_pad, HB), prod(H_pad, HB), {"f16_32": 4 * 3 * (2 * DB + HB) / 2.0, "f16_16": 0, "f32": 0})
        return {"f16_32": n, "f16_16": 0, "f32": 0}, bwd
    fwd = add(prod(d_pad, HB), prod(H_pad, HB), prod(H_pad, DB), *([prod(d_pad, DB)] * (2 if dense else 0)))
    # backward: producers W3^T G and W2^T dz2 per block; consumers, per PAIR of blocks and wave, three MFMAs per owned tile
    # (dW3: DB, dW2: HB, dW1: HB x ceil(DB / 4)), four waves
    cons = 4 * 3 * (DB + HB + HB * _cdiv(DB, 4)) / 2.0
    bwd = add(prod(d_pad, HB), prod(H_pad, HB), {"f16_32": cons, "f16_16": 0, "f32": 0})
    return fwd, bwd


def x3_roofline(fl2, fl3, issued, units, tiles_steps, ms, which):
    """Split-product kernels: an algorithmic fp32 flop costs three f16 flops, so the matrix floor of the kernel is
    3 x algorithmic flops at the f16 peak (`frac`); `frac_issued` prices the instructions actually issued (padding included,
    exact fp32 k-steps at the fp32 peak)."""
    t = ms * 1e-3
    alg = fl2[which] * units
    floor_s = 3.0 * alg / (PEAK_BF16_MFMA_TFLOPS * 1e12)
    n = issued[1] if which == "bwd_kernel" else issued[0]
    iss_s = (n["f16_32"] * MFMA_BF16_16x16x32_FLOP / (PEAK_BF16_MFMA_TFLOPS * 1e12) +
             n["f16_16"] * (MFMA_BF16_16x16x32_FLOP / 2) / (PEAK_BF16_MFMA_TFLOPS * 1e12) +
             n["f32"] * MFMA_F32_16x16x4_FLOP / (PEAK_FP32_MFMA_TFLOPS * 1e12)) * tiles_steps
    achieved = alg / t / 1e12
    return dict(achieved=achieved, peak=PEAK_BF16_MFMA_TFLOPS / 3.0, frac=floor_s / t, frac_issued=iss_s / t,
                frac_survey_m3=(3.0 * fl3[which] * units / (PEAK_BF16_MFMA_TFLOPS * 1e12)) / t,
                achieved_over_fp32_mfma_peak=achieved / PEAK_FP32_MFMA_TFLOPS)


def issued_mfma_quad_kernel(d_pad, H_pad, dense):
    """hjbq_fwd_kernel (four trajectories per workgroup, K <= 4 x CUs): v_mfma_f32_4x4x1_16b_f32 (512 flop) per workgroup and
    step, from GeoQ in csrc/hjbq_kernels.h, returned in 16x16x4 equivalents (2048 flop) per 16 trajectories so that it plugs
    into the same roofline arithmetic: 4 quads x 8 waves x (KO (1 + 2 SD) + KHo (1 + SD)) / 4."""
    DB, HB = _cdiv(d_pad, 16), _cdiv(H_pad, 16)
    KO, KHo = 4 * _cdiv(4 * DB, 8), 2 * HB
    SD = _cdiv(8 * KO, 64)
    per_wave = KO * (1 + (2 * SD if dense else 0)) + KHo * (1 + SD)
    return 8 * per_wave


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks_if_needed(args):
    """--gpus N > 1 without a launcher: start the N ranks as CHILD processes (torch.distributed.run) and relay their exit
    code.  This runs before anything touches the GPU: the parent only counts devices (no HIP context) and never re-execs."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    rehearsal = os.environ.get("PSP_BENCH_REHEARSAL") == "1"
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not rehearsal:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this box (PSP_BENCH_REHEARSAL=1 walks the "
                         "N > 1 code path with all ranks on cuda:0 over gloo; its numbers mean nothing)" % (args.gpus, n_dev))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def init_ranks(args):
    """(dist or None, rank, world, device).  One process per GPU; RCCL (backend 'nccl') unless rehearsing."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
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
    return dist, rank, world, dev, rehearsal


def timed_region(dist, dev, fn, steps, first):
    """barrier + synchronize on both sides, MAX over ranks."""
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for l in range(first, first + steps):
        fn(l)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def check_c_abi_allreduce(psp, dist, rank, world, dev, rehearsal):
    """N > 1, outside the timed region: the C-ABI collective (psp_comm_* / psp_allreduce, include/psp.h) against
    torch.distributed's all-reduce on the same buffers.  Reported, never fatal: the timed path uses torch.distributed."""
    if dist is None or rehearsal:
        return None
    import ctypes as C
    nat = psp.native
    try:
        lib = nat.load()
        ident = torch.zeros(nat.COMM_ID_BYTES, dtype=torch.uint8)
        # every rank binds RCCL through the C ABI first; the ranks then AGREE whether to go on, so that a rank that cannot
        # never leaves the others waiting inside the collective communicator set-up
        buf = (C.c_ubyte * nat.COMM_ID_BYTES)()
        rc = lib.psp_comm_unique_id(buf)
        flag = torch.tensor([1.0 if rc == 0 else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) != 1.0:
            return {"ok": False, "error": "psp_comm_unique_id failed on some rank: %s" % nat.last_error()}
        if rank == 0:
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        ident = ident.to(dev)
        dist.broadcast(ident, 0)
        raw = (C.c_ubyte * nat.COMM_ID_BYTES)(*ident.cpu().tolist())
        comm = C.c_void_p()
        nat.check(lib.psp_comm_init(C.byref(comm), world, rank, raw), "psp_comm_init")
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank)
        a = torch.randn(17188, generator=g, device=dev)
        s = torch.randn(2, generator=g, device=dev, dtype=torch.float64)
        a_ref, s_ref = a.clone(), s.clone()
        dist.all_reduce(a_ref)
        dist.all_reduce(s_ref)
        st = nat.stream_ptr(dev)
        nat.check(lib.psp_allreduce(nat.ptr(s), 2, nat.DT_F64, comm, st), "psp_allreduce f64")
        nat.check(lib.psp_allreduce(nat.ptr(a), a.numel(), nat.DT_F32, comm, st), "psp_allreduce f32")
        scratch = torch.zeros(17188, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        reps = 20
        ev[0].record()
        for _ in range(reps):
            nat.check(lib.psp_allreduce(nat.ptr(scratch), scratch.numel(), nat.DT_F32, comm, st), "psp_allreduce f32")
        ev[1].record()
        torch.cuda.synchronize()
        err = float((a - a_ref).abs().max())
        ok = bool(torch.allclose(s, s_ref, rtol=1e-12, atol=0.0)) and err <= 1e-5
        nat.check(lib.psp_comm_destroy(comm), "psp_comm_destroy")
        return {"ok": ok, "max_abs_diff_vs_torch_distributed": err,
                "grad_sized_allreduce_ms": ev[0].elapsed_time(ev[1]) / reps}
    except Exception as e:                                  # noqa: BLE001 -- a diagnostic must not take the bench down
        return {"ok": False, "error": "%s: %s" % (type(e).__name__, e)}


def parity_vs_one_rank(psp, sharding, dist, rank, world, dev, w, outer):
    """N > 1, before anything is timed: iteration 0 of a batch of 16384 trajectories per rank on the N ranks (sharded plan, both
    collectives) against the SAME global batch on rank 0 alone.  Philox noise is indexed by the global trajectory id, so D is
    the same per trajectory and loss / gradient agree to summation order.  Also returns what a SUM all-reduce of ones
    returns on this backend (`rccl_ranks_seen`)."""
    if outer:
        return None
    Kc = 16384 * world
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)

    def run(single):
        sharding.force_single = single
        try:
            prob = psp.LLGC(d=w["d"], off_diag=w["off_diag"], T=w["T"], seed=42, device=dev)
            m = psp.Solver("bench-parity-%d" % single, prob, lr=1e-3, L=1, K=Kc, delta_t=w["dt"], loss_method="log-variance",
                           time_approx="inner", adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False,
                           verbose=False, seed=42, device=dev, backend="native", noise="philox", widths=(w["H"], w["H"]),
                           mlp_dtype=w.get("mlp", "auto"))
            pl = m._choose_plan()
            out = torch.zeros(1, dtype=torch.float32, device=dev)
            pl.iteration(0, out)
            torch.cuda.synchronize()
            return pl.D.clone(), float(out[0].item()), pl.grad.clone()
        finally:
            sharding.force_single = False

    D_s, loss_s, g_s = run(False)
    parts = [torch.empty_like(D_s) for _ in range(world)]
    dist.all_gather(parts, D_s)
    res = None
    try:
        if rank == 0:
            D_1, loss_1, g_1 = run(True)
            D_all = torch.cat(parts)
            res = {"K_global": Kc, "ranks": world, "rccl_ranks_seen": int(round(float(ones.item()))),
                   "D_max_abs_diff": float((D_all - D_1).abs().max()), "D_max_abs": float(D_1.abs().max()),
                   "loss_rel_diff": abs(loss_s - loss_1) / abs(loss_1),
                   "grad_max_rel_diff": float((g_s - g_1).abs().max()) / float(g_1.abs().max()),
                   "tolerance": "summation order: loss 1e-6, gradient 1e-5 x max|g|"}
            res["ok"] = bool(res["loss_rel_diff"] <= 1e-6 and res["grad_max_rel_diff"] <= 1e-5 and res["rccl_ranks_seen"] == world)
    except Exception as e:                                   # noqa: BLE001 -- a self-check must not take the measurement down
        res = {"ok": False, "error": "%s: %s" % (type(e).__name__, e), "rccl_ranks_seen": int(round(float(ones.item())))}
    finally:
        sharding.force_single = False
        dist.barrier()
    return res


def collective_summary(sharding, steps):
    """Mean HIP-event time per iteration of the all-reduces recorded by sharding.allreduce_sum_ (N > 1 only)."""
    evs = sharding.coll_events or []
    if not evs:
        return None
    by_size = {}
    for e0, e1, nbytes in evs:
        by_size.setdefault(nbytes, []).append(e0.elapsed_time(e1))
    items = [{"bytes": b, "calls_per_step": len(v) / float(steps), "mean_ms": sum(v) / len(v), "max_ms": max(v)}
             for b, v in sorted(by_size.items())]
    return {"per_step_ms": sum(sum(v) for v in by_size.values()) / float(steps), "all_reduces": items,
            "backend": "rccl" if os.environ.get("PSP_BENCH_REHEARSAL") != "1" else "gloo (rehearsal)"}


def _time_oracle(orc, prob, cfg, z, y0, N, threads, budget_s, max_iters):
    torch.set_num_threads(threads)
    orc.hjb_train(prob, cfg, step_models=(z, y0, N))          # warm-up iteration
    iters, t0 = 0, time.time()
    while True:
        orc.hjb_train(prob, cfg, step_models=(z, y0, N))
        iters += 1
        el = time.time() - t0
        if el > budget_s or iters >= max_iters:
            break
    return cfg.K * N * iters / el, iters, el


def cpu_baseline(w):
    """Times oracle/pathspace_oracle.py (torch-CPU port of the reference iteration, including its duplicate control
    evaluation and dense sigma products) on the GPU box's host cores: SURVEY 8d's legs -- configs[0] (LQGC d=2, K=128,
    N=20), configs[1] (LLGC d=100, K=1024, N=50, 2x64 MLP) and a K=4096 cut of the workload, each with all cores of the
    box's CPU share and with 1 thread.  About 20 s of CPU work in total.  The top-level value is the K=4096 cut at all cores."""
    from oracle import pathspace_oracle as orc
    # a 1-GPU box exposes a 16-core CPU share; more torch threads than that only adds contention
    allc = min(16, os.cpu_count() or 1)
    legs = []

    def leg(name, prob, K, dt, widths, d, budget, max_iters):
        for threads in (allc, 1):
            cfg = orc.HJBConfig(K=K, delta_t=dt, lr=1e-3, L=1, seed=42, adaptive_forward_process=True, detach_forward=True)
            z = orc.TanhMLP(d + 1, d, 1e-3, seed=123, widths=widths)
            _, y0, N = orc.hjb_build(prob, cfg)
            rate, iters, el = _time_oracle(orc, prob, cfg, z, y0, N, threads, budget, max_iters)
            legs.append(dict(case=name, value=rate, unit="trajectory-timesteps/s", cores=threads, s_per_iter=el / iters,
                             sample="K=%d, N=%d, %d iterations, %.1f s" % (K, N, iters, el)))

    leg("configs[0] LQGC d=2 K=128 N=20 (default 30-30 MLP)",
        orc.make_problem("LQGC", d=2, off_diag=0.1, T=1, seed=42, delta_t=0.05), 128, 0.05, (30, 30), 2, 1.0, 200)
    leg("configs[1] LLGC d=100 K=1024 N=50 (64-64 MLP)",
        orc.make_problem("LLGC", d=100, off_diag=0.01, T=0.5, seed=42), 1024, 0.01, (64, 64), 100, 2.0, 40)
    Kc = min(4096, w.get("K") or w.get("K_global"))
    leg("workload cut: LLGC d=%d K=%d of %d" % (w["d"], Kc, w.get("K") or w.get("K_global")),
        orc.make_problem("LLGC", d=w["d"], off_diag=w["off_diag"], T=w["T"], seed=42), Kc, w["dt"], (w["H"], w["H"]),
        w["d"], 6.0, 24)
    torch.set_num_threads(allc)
    main = legs[4]
    return dict(value=main["value"], unit="trajectory-timesteps/s", cores=main["cores"], kind="port",
                sample="oracle hjb_train, %s, %s, torch %s" % (main["case"], main["sample"], torch.__version__),
                legs=legs)


def loss_rel_err_vs_cpu(psp, dev, w, mode="fp32"):
    """BASELINE.json's second metric: |L_gpu - L_ref| / |L_ref| per iteration on FIXED SEEDS -- the native plan with the
    reference's host-generated noise (torch CPU generator, seed 42) against the CPU oracle consuming the same stream, on the
    workload's problem at K = 1024 trajectories, 3 iterations.  Part of the cpu_baseline leg (the only place bench.py may
    use oracle/).  `mode`: the matrix-product mode of the timed kernels (Solver(mlp_dtype=...)), forced here so that the check runs
    the same kernels at this smaller K."""
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
                       verbose=False, seed=42, device=dev, backend="native", noise="reference", widths=(w["H"], w["H"]),
                       mlp_dtype=mode)
    model.train()
    assert model._native_plan.matrix_mode == mode
    errs = [abs(a - b) / abs(b) for a, b in zip(model.loss_log, ref)]
    return dict(value=max(errs), per_iteration=errs, tolerance=1e-4,
                case="LLGC d=%d, K=%d, N=%d, %d iterations, seed 42, reference noise stream, matrix products %s" % (w["d"], K, N, L, mode))


def secondary(psp, dev, w, steps=200, warmup=20):
    prob = psp.LLGC(d=w["d"], off_diag=w["off_diag"], T=w["T"], seed=42, device=dev)
    model = psp.Solver("bench-cfg1", prob, lr=1e-3, L=steps + warmup, K=w["K"], delta_t=w["dt"],
                       loss_method="log-variance", time_approx="inner", adaptive_forward_process=True,
                       detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42, device=dev,
                       backend="native", noise="philox", widths=(w["H"], w["H"]), mlp_dtype=w.get("mlp", "auto"))
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
            "steps": steps, "K": w["K"], "N": model.N, "graph": bool(getattr(plan, "graph_active", False)),
            "matrix_products": getattr(plan, "matrix_mode", "fp32")}


def mfma_roofline(fl2, fl3, issued, units, tiles_steps, ms, bf16_mlp, which):
    """Roofline terms of one kernel.  fl2 / fl3: algorithmic flop dictionaries (m = 2 / m = 3); issued: (f32, bf16, bwd_f32)
    MFMA instructions per tile-step; which: 'fwd_kernel' | 'bwd_kernel'."""
    t = ms * 1e-3
    alg = fl2[which] * units
    if bf16_mlp and which == "fwd_kernel":
        # flop-weighted blend: control-net products at the bf16 peak, everything else at the fp32 peak
        floor_s = (fl2["mlp_fwd"] * units / (PEAK_BF16_MFMA_TFLOPS * 1e12) + fl2["sde"] * units / (PEAK_FP32_MFMA_TFLOPS * 1e12))
        peak = alg / floor_s / 1e12
        iss_floor_s = (issued[1] * MFMA_BF16_16x16x32_FLOP / (PEAK_BF16_MFMA_TFLOPS * 1e12) +
                       issued[0] * MFMA_F32_16x16x4_FLOP / (PEAK_FP32_MFMA_TFLOPS * 1e12)) * tiles_steps
        frac_issued = iss_floor_s / t
    else:
        peak = PEAK_FP32_MFMA_TFLOPS
        n = issued[2] if which == "bwd_kernel" else issued[0]
        frac_issued = n * MFMA_F32_16x16x4_FLOP * tiles_steps / t / 1e12 / PEAK_FP32_MFMA_TFLOPS
    achieved = alg / t / 1e12
    return dict(achieved=achieved, peak=peak, frac=achieved / peak, frac_issued=frac_issued,
                frac_survey_m3=(fl3[which] * units / t / 1e12) / peak)


def load_counters(workload, kernel):
    """Committed PMC means per launch of `kernel` in `workload` (profiles/r4_counters.json -- this round's kernels --, else
    profiles/r3_counters.json; written by tools/r3/make_counters_json.py from the separate --pmc passes of tools/pmc_passes.sh), or {}."""
    for name in ("r4_counters.json", "r3_counters.json"):
        cpath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(cpath):
            ent = json.load(open(cpath)).get(workload, {}).get(kernel, {})
            if ent:
                return ent
    return {}


def binder_of(mf_frac, hbm_alg_frac, hbm_store_frac, counters, tiles_steps):
    """What the dominant kernel waits on.  A term above 0.5 names itself; otherwise the committed counters decide: the share
    of wave cycles spent issuing / waiting on a busy pipe against waiting on memory, and the VALU instruction count per
    tile-step next to the MFMA count."""
    out = {"name": None}
    if counters:
        valu, mfma = counters.get("SQ_INSTS_VALU"), counters.get("SQ_INSTS_MFMA")
        wc = counters.get("SQ_WAVE_CYCLES")
        if valu and tiles_steps:
            out["valu_insts_per_tile_step"] = valu / tiles_steps
        if mfma and tiles_steps:
            out["mfma_insts_per_tile_step"] = mfma / tiles_steps
        if wc:
            for k_, n_ in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "waiting_on_busy_pipe"), ("SQ_WAIT_ANY", "waiting_on_memory_or_barrier")):
                if counters.get(k_) is not None:
                    out["wave_cycles_" + n_] = counters[k_] / wc
        bc = counters.get("SQ_BUSY_CYCLES")
        if counters.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and counters.get("GRBM_GUI_ACTIVE"):
            out["matrix_pipe_busy"] = counters["SQ_VALU_MFMA_BUSY_CYCLES"] / (counters["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if max(mf_frac, hbm_alg_frac) >= 0.5:
        out["name"] = "mfma" if mf_frac >= hbm_alg_frac else "hbm"
    elif hbm_store_frac >= 0.5:
        out["name"] = "hbm (path store: the design's own bytes, 1.7x the algorithmic ones)"
    elif out.get("valu_insts_per_tile_step") and out.get("mfma_insts_per_tile_step") and \
            out["valu_insts_per_tile_step"] > 3.0 * out["mfma_insts_per_tile_step"]:
        out["name"] = "valu"
    else:
        out["name"] = "latency / issue (no single unit above half of its peak)"
    return out


def load_traffic(workload, kernel):
    """HBM bytes per launch from the committed PMC passes (profiles/traffic.json: FETCH_SIZE already carries the gfx950
    correction for the stream widths this kernel uses, as calibrated by tools/fetch_calibrate)."""
    c = load_counters(workload, kernel)
    if c.get("FETCH_SIZE") is not None and c.get("WRITE_SIZE") is not None:
        return int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None
    ent = json.load(open(tpath)).get(workload, {}).get(kernel)
    if isinstance(ent, dict):
        return ent.get("hbm_bytes")
    return ent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="hjb_llgc_d100_K65536_N100_h64",
                    choices=sorted(WORKLOADS) + sorted(GENERAL_WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] side measurement")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 3 s steady-state leg")
    ap.add_argument("--sustain-s", type=float, default=3.0)
    args = ap.parse_args()
    spawn_ranks_if_needed(args)                          # N > 1 without a launcher: children do the work

    import path_space_pde_solver_amd as psp
    from path_space_pde_solver_amd import sharding
    if args.workload in GENERAL_WORKLOADS:
        return main_general(args, psp, sharding)
    w = WORKLOADS[args.workload]
    dist, rank, world, dev, rehearsal = init_ranks(args)

    strong = "K_global" in w
    if strong:
        K_global = w["K_global"]
        if K_global % (16 * world):
            raise SystemExit("K_global=%d does not split into whole 16-trajectory tiles over %d ranks" % (K_global, world))
        K_local = K_global // world
    else:
        K_local, K_global = w["K"], w["K"] * world       # weak scaling: fixed trajectories per GPU
    prob = psp.LLGC(d=w["d"], off_diag=w["off_diag"], T=w["T"], seed=42, device=dev)
    total = args.warmup + args.steps
    outer = bool(w.get("outer"))
    extra = {}
    if "path_budget_gb" in w:
        extra["path_budget_bytes"] = int(w["path_budget_gb"] * 2 ** 30)
    if "chunks" in w:
        extra["path_chunks"] = int(w["chunks"])
    model = psp.Solver("bench", prob, lr=w.get("lr", 1e-3), L=total, K=K_global, delta_t=w["dt"], loss_method="log-variance",
                       time_approx="outer" if outer else "inner", adaptive_forward_process=True, detach_forward=True,
                       u_l2_error_flag=False, verbose=False, seed=42, device=dev, backend="native",
                       noise="philox", widths=(w["H"], w["H"]), mlp_dtype=w.get("mlp", "auto"), **extra)
    if outer:                                            # the constructor builds arch [30, 30]; honour the workload's H
        model.z_n = [psp.DenseNet(d_in=w["d"], d_out=w["d"], lr=w.get("lr", 1e-3), arch=[w["H"], w["H"]], seed=42).to(dev)
                     for _ in range(model.N)]
        model.update_Phis()
    plan = model._choose_plan()
    assert model.plan_name == "native"
    parity = parity_vs_one_rank(psp, sharding, dist, rank, world, dev, w, outer) if world > 1 else None
    N_t = model.N
    n_eager = 6                                          # eager iterations after the timed region (graph workloads only)
    losses = torch.zeros(total + n_eager, dtype=torch.float32, device=dev)
    for l in range(args.warmup):
        plan.iteration(l, losses)

    graph = bool(getattr(plan, "_graph_wanted", lambda: False)())
    if getattr(plan, "n_chunks", 1) > 1:
        plan.pass1_events = []
    sharding.coll_events = [] if world > 1 else None
    if graph:
        # launch-bound iteration replayed as a hipGraph: `value` is timed on the replay; the per-kernel HIP events of the
        # roofline come from a few eager iterations of the same kernels AFTER the timed region
        elapsed = timed_region(dist, dev, lambda l: plan.iteration(l, losses), args.steps, args.warmup)
        plan.events = []
        for l in range(total, total + n_eager):
            plan.iteration(l, losses)
        torch.cuda.synchronize()
        plan.events = plan.events[1:]
    else:
        plan.events = []                                 # (fwd_start, fwd_end, bwd_start, bwd_end) per launch pair
        elapsed = timed_region(dist, dev, lambda l: plan.iteration(l, losses), args.steps, args.warmup)

    n_ev = max(1, len(plan.events))
    fwd_ms = sum(e[0].elapsed_time(e[1]) for e in plan.events) / n_ev
    bwd_ms = sum(e[2].elapsed_time(e[3]) for e in plan.events) / n_ev
    coll = collective_summary(sharding, args.steps)
    sharding.coll_events = None
    loss_vals = losses.cpu().tolist()[:total]
    if rank != 0:
        finish_ranks(psp, dist, rank, world, dev, rehearsal)
        return

    dense = w["off_diag"] != 0.0
    flops = alg_flops_dense_control if outer else alg_flops_per_traj_step
    fl2, fl3 = flops(w["d"], w["H"], dense, m=2), flops(w["d"], w["H"], dense, m=3)
    n_chunks = int(getattr(plan, "n_chunks", 1))
    K_launch = K_local // n_chunks                       # trajectories per kernel launch
    units_launch = K_launch * N_t                        # trajectory-timesteps per launch on one GPU
    tiles_steps = _cdiv(K_launch, 16) * N_t
    # kernel names as they appear in rocprof: family 1 = hjb_kernels.h (hjbs_kernels.h forward when there are at most two
    # tiles per CU, hjbq_kernels.h when there are at most CUs / 4), family 2 = hjbw_kernels.h
    ntile = _cdiv(K_launch, 16)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    bf16_mlp = w.get("mlp") == "bf16"
    x3 = getattr(plan, "matrix_mode", "fp32") == "f16x3"      # split-product kernels (three f16 MFMAs per fp32 product)
    if outer:
        fwd_name, bwd_name = "hjbd_fwd_kernel", ("hjbd_bwd_kernel" if plan.kernel_bwd else "library GEMMs")
        issued = None
        issued_x3 = None
    else:
        quad = plan.family != 2 and 4 * ntile <= cus and not bf16_mlp and os.environ.get("PSP_FWD_VARIANT") in (None, "3")
        coop = int(getattr(plan.sizes, "fwd_coop_tiles", 0) or 0)     # cooperative wide forward (csrc/hjbc_kernels.h): tiles per workgroup
        fwd_name = ("hjbc_fwd_kernel" if coop else "hjbw_fwd_kernel") if plan.family == 2 else (
            "hjbq_fwd_kernel" if quad else ("hjbs_fwd_kernel" if ntile <= 2 * cus else "hjb_fwd_kernel"))
        bwd_name = ("hjbw_bwd2_kernel" if plan.d_pad <= 256 else "hjbw_bwd_kernel") if plan.family == 2 else "hjb_bwd2_kernel"
        issued = issued_mfma_per_tile_step(plan.d_pad, plan.H_pad, dense, plan.family, bf16_mlp)
        if x3:
            if plan.family != 2:
                bwd_name = "hjb_bwd3_kernel"
            elif plan.d_pad > 256:
                bwd_name = "hjbw_bwd_x3_kernel"
            elif os.environ.get("PSP_WIDE_BWD_X3", "1")[:1] != "0":
                bwd_name = "hjbw_bwd2x_kernel"
            issued_x3 = issued_mfma_x3(plan.d_pad, plan.H_pad, dense, plan.family)
            if plan.family == 2 and bwd_name == "hjbw_bwd2_kernel":      # (A/B switch: the fp32-MFMA backward)
                issued_x3 = (issued_x3[0], None)
        if quad:
            issued = (issued_mfma_quad_kernel(plan.d_pad, plan.H_pad, dense), 0, issued[2])
    bwd_dominant = bwd_ms >= fwd_ms
    dom, dom_ms, which = (bwd_name, bwd_ms, "bwd_kernel") if bwd_dominant else (fwd_name, fwd_ms, "fwd_kernel")
    if issued is None:                                   # DenseNet control: algorithmic terms only
        t = dom_ms * 1e-3
        # (split-product forward: three f16 flops per algorithmic fp32 flop; its backward kernel runs fp32 MFMA)
        mf = dict(achieved=fl2[which] * units_launch / t / 1e12,
                  peak=(PEAK_BF16_MFMA_TFLOPS / 3.0) if (x3 and which == "fwd_kernel") else PEAK_FP32_MFMA_TFLOPS)
        mf["frac"] = mf["achieved"] / mf["peak"]
        mf["frac_issued"] = None
        mf["frac_survey_m3"] = fl3[which] * units_launch / t / 1e12 / mf["peak"]
    elif x3 and not (which == "bwd_kernel" and issued_x3[1] is None):
        mf = x3_roofline(fl2, fl3, issued_x3, units_launch, tiles_steps, dom_ms, which)
    else:
        mf = mfma_roofline(fl2, fl3, issued, units_launch, tiles_steps, dom_ms, bf16_mlp, which)
    # HBM term: SURVEY 8d's algorithmic bytes (state streamed once per step, 8d + 8) and this design's real ones (the path
    # store: state stays on chip, the forward writes / the backward reads one block of register images per unit)
    path_B_unit = float(plan.sizes.path_bytes) / max(1, units_launch) if hasattr(plan, "sizes") else None
    if path_B_unit and getattr(plan, "regen_xi", False):
        # store_path 4 (include/psp.h): the xi slot of a block is neither written nor read (the backward regenerates the
        # increments from the Philox counters); the allocation keeps the slot, the traffic does not
        path_B_unit -= 4.0 * 16 * ((plan.d_pad + 15) // 16)
    step_s = elapsed / args.steps
    dom_s = dom_ms * 1e-3
    alg_B_unit = 8 * w["d"] + 8
    hbm = {"bytes_per_unit_alg": alg_B_unit, "achieved_GBps_alg": alg_B_unit * units_launch / dom_s / 1e9,
           "bytes_per_unit_path_store": path_B_unit,
           "achieved_GBps_path_store": (path_B_unit * units_launch / dom_s / 1e9) if path_B_unit else None,
           "peak_GBps": PEAK_HBM_GBS, "achievable_GBps": 6300.0}
    hbm["frac_alg_bytes"] = hbm["achieved_GBps_alg"] / PEAK_HBM_GBS
    hbm["frac_path_store"] = (hbm["achieved_GBps_path_store"] or 0.0) / PEAK_HBM_GBS
    hbm["frac"] = hbm["frac_alg_bytes"]
    bound = "mfma" if mf["frac"] >= hbm["frac_alg_bytes"] else "hbm"
    value = K_global * N_t * args.steps / elapsed
    whole_tf = fl2["total"] * K_local * N_t / step_s / 1e12
    counters = load_counters(args.workload, dom)
    roof = {"bound": bound, "kernel": dom,
            "achieved": mf["achieved"] if bound == "mfma" else hbm["achieved_GBps_alg"],
            "peak": mf["peak"] if bound == "mfma" else PEAK_HBM_GBS,
            "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
            "frac": max(mf["frac"], hbm["frac_alg_bytes"]),
            "traffic": load_traffic(args.workload, dom),
            "binder": binder_of(mf["frac"], hbm["frac_alg_bytes"], hbm["frac_path_store"], counters, tiles_steps),
            "convention": ("SURVEY 8d algorithmic work per trajectory-timestep x units per launch / HIP-event kernel time / peak: "
                           "bytes 8d+8 against 8 TB/s, minimal flops (m = 2 dense d x d products) against the peak of the pipe the "
                           "kernel computes on (%s); frac = the larger of the two; hbm_term.frac_path_store prices the bytes the "
                           "design really moves" % ("2500/3 TFLOP/s: split products, three f16 flops per fp32 flop" if x3 else
                                                    ("bf16 / fp32 blend" if bf16_mlp else "157.3 TFLOP/s fp32 MFMA"))),
            "mfma_term": mf, "hbm_term": hbm,
            "alg_flops_per_traj_step": {k: fl2[k] for k in ("fwd_kernel", "bwd_kernel", "total")},
            "alg_flops_per_traj_step_survey_m3": {k: fl3[k] for k in ("fwd_kernel", "bwd_kernel", "total")},
            "issued_mfma_per_tile_step": ({"fwd": issued_x3[0], "bwd": issued_x3[1]} if (x3 and issued_x3) else
                                          {"fwd_f32_16x16x4": issued[0], "fwd_bf16_16x16x32": issued[1],
                                           "bwd_f32_16x16x4": issued[2]} if issued else None),
            "units_per_launch": units_launch, "launches_per_step": n_chunks,
            "fwd_kernel_ms": fwd_ms, "bwd_kernel_ms": bwd_ms,
            "whole_step_tflops": whole_tf, "whole_step_frac_of_fp32_peak": whole_tf / PEAK_FP32_MFMA_TFLOPS}
    # the other rollout kernel, same two terms (the iteration is the pair)
    oth, oth_ms, oth_which = (fwd_name, fwd_ms, "fwd_kernel") if bwd_dominant else (bwd_name, bwd_ms, "bwd_kernel")
    if oth_ms > 0:
        oth_peak = mf["peak"] if not (x3 and oth_which == "bwd_kernel" and issued_x3 and issued_x3[1] is None) else PEAK_FP32_MFMA_TFLOPS
        roof["other_kernel"] = {"kernel": oth, "ms": oth_ms,
                                "frac_alg_bytes": alg_B_unit * units_launch / (oth_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                "frac_path_store": ((path_B_unit or 0.0) * units_launch / (oth_ms * 1e-3) / 1e9) / PEAK_HBM_GBS,
                                "frac_mfma": fl2[oth_which] * units_launch / (oth_ms * 1e-3) / 1e12 / oth_peak,
                                "traffic": load_traffic(args.workload, oth)}
    out = {
        "metric": "trajectory-timesteps/sec (K*N/s), d=%d HJB log-variance training iteration" % w["d"],
        "value": value, "unit": "trajectory-timesteps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * step_s, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": ("bf16 control-net products in the forward rollout, f32 elsewhere" if bf16_mlp else
                  ("f32 (f16x3 split products: every matrix product as three f16 MFMAs, fp32 accumulate; state / sums fp32)" if x3 else "f32")),
        "data": "synthetic",
        "config": {"workload": args.workload, "problem": "LLGC", "d": w["d"], "K_per_gpu": K_local,
                   "K_global": K_global, "N": N_t,
                   "mlp": ("%d x DenseNet %d-%d-%d-%d relu^2, one per time step (time_approx='outer')" % (N_t, w["d"], w["H"], w["H"], w["d"]))
                          if outer else "%d-%d-%d-%d tanh" % (w["d"] + 1, w["H"], w["H"], w["d"]),
                   "loss": "log-variance", "noise": "on-device Philox4x32-7",
                   "matrix_products": ("fp32-grade split products on the f16 matrix pipe: x = hi + lo/2048 (two f16 numbers), "
                                       "a.b = hi.hi + (hi.lo + lo.hi)/2048 as three v_mfma_f32_16x16x32_f16 with fp32 accumulation "
                                       "(product error 1.07x that of v_mfma_f32_16x16x4_f32; same parity bounds, "
                                       "tests/test_gpu_split_product.py); state, sums and the path store are fp32"
                                       + ("; forward kernel only (hjbw_bwd2_kernel, d <= 256, runs v_mfma_f32_16x16x4_f32)"
                                          if (getattr(plan, "family", 1) == 2 and plan.d_pad <= 256) else "")) if x3 else
                                      ("v_mfma_f32_16x16x32_bf16 for the control net, v_mfma_f32_16x16x4_f32 elsewhere" if bf16_mlp
                                       else "v_mfma_f32_16x16x4_f32"),
                   "kernels": {"forward": fwd_name + (" (%d tiles per 512-thread workgroup)" % int(getattr(plan.sizes, "fwd_coop_tiles", 0))
                                                       if (not outer and int(getattr(plan.sizes, "fwd_coop_tiles", 0) or 0)) else ""),
                               "backward": bwd_name},
                   "launch": "hipGraph replay of the captured iteration" if graph else "eager launches",
                   "parallelism": "trajectory-sharded x%d (%s)" % (world, "gloo rehearsal on one GPU" if rehearsal else
                                                                    ("RCCL" if world > 1 else "single process")),
                   "path_store": ("K-chunked plan (%s): %d chunks of %d trajectories share a %.1f GB store"
                                  % (plan.chunk_mode, n_chunks, K_launch, plan.sizes.path_bytes / 1e9)) if n_chunks > 1 else
                                 ("resident (%.1f GB)" % (plan.sizes.path_bytes / 1e9) if hasattr(plan, "sizes") else "resident")},
        "roofline": roof,
        "loss_first_last": [loss_vals[0], loss_vals[-1]],
    }
    if n_chunks > 1:
        # recompute overhead: pass 1 (forward without store) is extra work over the resident-store plan
        p1 = getattr(plan, "pass1_events", None) or []
        out["chunking"] = {"chunks": n_chunks, "mode": plan.chunk_mode,
                           "pass1_forward_ms_per_step": (sum(a.elapsed_time(b) for a, b in p1) / len(p1)) if p1 else 0.0,
                           "note": ("two_gradient: no forward recompute; each chunk's backward kernel runs twice (weights D_k - c "
                                    "and 1), grad = (2/K)[G1 - (mean D - c) G0]" if plan.chunk_mode == "two_gradient" else
                                    "recompute: pass 1 re-runs the forward rollout without the path store to obtain the global "
                                    "(sum D, sum D^2)") + "; value counts each trajectory-timestep once"}
    if coll is not None:
        coll["fraction_of_step"] = coll["per_step_ms"] / (1e3 * step_s)
        out["collectives"] = coll
    if parity is not None:
        out["parity_vs_1rank"] = parity
        out["rccl_ranks_seen"] = parity.get("rccl_ranks_seen")
    if hasattr(plan, "regen_xi"):
        out["config"]["path_store"] = ("store_path 4: X_n, h1, h2 kept, xi regenerated by the backward from the Philox counters"
                                       if plan.regen_xi else "store_path %d" % int(plan.cfg.store_path))
    if getattr(plan, "range_flag", None) is not None:
        out["config"]["range_guard"] = ("on: device flag + predicated fp32-MFMA twins of the split kernels (include/psp.h range_flag); "
                                        "fallback iterations in this run: %d" % plan.range_fallbacks())
    if world == 1 and not args.no_sustained and not graph:
        # >= 3 s of back-to-back iterations (the timed region above is steps x ms_per_step, i.e. under 0.1 s by default)
        # (a continuation of the SAME training run: the iteration index keeps counting -- it is the Philox `iter` counter -- and the
        #  losses go to a log of their own; nothing of the timed region is replayed or overwritten)
        n_s, t_s = 0, 0.0
        plan.events = None
        scratch = torch.zeros(1 << 20, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while t_s < args.sustain_s and total + n_s + 50 <= scratch.numel():
            for l in range(total + n_s, total + n_s + 50):
                plan.iteration(l, scratch)
            n_s += 50
            torch.cuda.synchronize()
            t_s = time.perf_counter() - t0
        out["also_sustained"] = {"value": K_global * N_t * n_s / t_s, "unit": "trajectory-timesteps/s", "ms_per_step": 1e3 * t_s / n_s,
                                 "steps": n_s, "seconds": t_s, "note": "same plan, iterations back to back after the timed region"}
    if world == 1 and args.workload == "hjb_llgc_d100_K65536_N100_h64" and not args.no_secondary:
        # BASELINE.json configs[1] (d=100, K=1024, N=50) measured in the same process, for readers who take
        # that as the quoted configuration (64 16-trajectory tiles: runs on the feature-split forward kernel)
        out["also_configs1_K1024_N50"] = secondary(psp, dev, WORKLOADS["hjb_llgc_d100_K1024_N50_h64"])
    if world == 1 and x3 and not args.no_secondary:
        # the same workload on the fp32-MFMA kernels (Solver(mlp_dtype='fp32')), same process, for comparison
        w32 = dict(w)
        w32["mlp"] = "fp32"
        out["also_fp32_mfma_kernels"] = secondary(psp, dev, w32, steps=args.steps, warmup=min(args.warmup, 3))
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w)
        out["loss_rel_err_vs_cpu_ref"] = loss_rel_err_vs_cpu(psp, dev, w, getattr(plan, "matrix_mode", "fp32"))
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    print(json.dumps(out))
    sys.stdout.flush()
    finish_ranks(psp, dist, rank, world, dev, rehearsal)


def finish_ranks(psp, dist, rank, world, dev, rehearsal):
    """N > 1, AFTER rank 0 has printed the JSON line: cross-check of the C-ABI collective (psp_comm_* / psp_allreduce) against
    torch.distributed's all-reduce on every rank, reported as one line on rank 0's stderr.  A watchdog ends the process if the
    second RCCL communicator does not come up within a minute (exit status 3 and a line on stderr: a hung check must not
    read as success) -- the measurement is already out by then."""
    if dist is None:
        return
    if not rehearsal and os.environ.get("PSP_BENCH_CABI_CHECK", "1") == "1":
        import threading
        def bark():
            sys.stderr.write("bench.py: the C-ABI RCCL cross-check (second communicator) did not finish within 60 s on rank %d; "
                             "the JSON line above is complete, exiting with status 3\n" % rank)
            sys.stderr.flush()
            os._exit(3)
        dog = threading.Timer(60.0, bark)
        dog.daemon = True
        dog.start()
        res = check_c_abi_allreduce(psp, dist, rank, world, dev, rehearsal)
        dog.cancel()
        if rank == 0:
            print("psp_allreduce C-ABI check (%d ranks): %s" % (world, json.dumps(res)), file=sys.stderr)
            sys.stderr.flush()
    dist.destroy_process_group()


def make_general_model(psp, w, dev, arch, K, L, backend, noise, mlp="auto", name="bench"):
    """Problem + solver + value net of a GENERAL_WORKLOADS entry (GeneralSolver, or EllipticSolver for solver='elliptic')."""
    if w.get("problem") == "AllenCahn":
        prob = psp.AllenCahn(d=w["d"], T=w["T"], seed=42, modus="pt", device=dev)
        prob.boundary_distance = w.get("boundary_distance", 1.0)
    elif w.get("problem") == "Committor":
        prob = psp.Committor(d=w["d"], device=dev)
    else:
        prob = psp.DoubleWell_multidim_for_general_solver(d=w["d"], d_1=w["d"] // 2, d_2=w["d"] - w["d"] // 2, T=w["T"],
                                                          eta=1, kappa=1, modus="HJB", device=dev)
    elliptic = w.get("solver") == "elliptic"
    if elliptic:
        model = psp.EllipticSolver(problem=prob, name=name, seed=42, delta_t=w["dt"], N=w["N"], lr=1e-3, L=L, K=K, K_boundary=50,
                                   alpha=w.get("alpha", [1.0, 1.0]), loss_method=w["loss"], verbose=False, device=dev,
                                   backend=backend, noise=noise, v_l2_error_flag=False)
    else:
        model = psp.GeneralSolver(problem=prob, name=name, seed=42, delta_t=w["dt"], N=w["N"], lr=1e-3, L=L, K=K, K_boundary=50,
                                  alpha=w.get("alpha", [1.0, 1.0, 1.0]), loss_method=w["loss"], verbose=False, device=dev,
                                  backend=backend, noise=noise, mlp_dtype=mlp, uniform_square=bool(w.get("uniform_square", False)))
    model.V = psp.DenseNet(d_in=w["d"] + (0 if elliptic else 1), d_out=1, lr=1e-3, arch=arch, seed=42,
                           activation=w.get("act", "relu2")).to(dev)
    return prob, model


def general_composite_leg(psp, dev, w, arch, iters=6):
    """The same configuration on the package's composite torch plan (the reference's op sequence with autograd) on this GPU."""
    prob, m = make_general_model(psp, w, dev, arch, w["K"], iters, "torch", "reference", name="bench-composite")
    m.train()
    per = sorted(m.times[1:])[len(m.times[1:]) // 2]
    return {"s_per_iteration": per, "value": (sum(m.K_log[1:]) / max(1, len(m.K_log) - 1)) / per, "unit": "trajectory-timesteps/s",
            "iterations": iters, "plan": "composite (torch autograd, eager)"}


def general_cpu_baseline(w, arch, iters=8):
    """The CPU oracle (oracle/pathspace_oracle.py general_train / elliptic_train: ports of GeneralSolver.train and
    EllipticSolver.train) on this box's host cores."""
    from oracle import pathspace_oracle as orc
    allc = min(16, os.cpu_count() or 1)
    torch.set_num_threads(allc)
    net = dict(kind="user_tanh2" if w.get("act") == "tanh2" else "densenet", arch=arch, seed=42)
    if w.get("solver") == "elliptic":
        prob = orc.make_problem(w["problem"], d=w["d"])
        cfg = orc.EllipticConfig(K=w["K"], N=w["N"], delta_t=w["dt"], lr=1e-3, L=1, seed=42, K_boundary=50,
                                 alpha=tuple(w.get("alpha", (1.0, 1.0))), loss_method=w["loss"])
        V, train = orc.elliptic_build(prob, cfg, net=net), orc.elliptic_train
    else:
        prob = orc.make_problem("AllenCahn", d=w["d"], T=w["T"], seed=42, modus="pt", boundary_distance=w.get("boundary_distance", 1.0))
        cfg = orc.GeneralConfig(K=w["K"], N=w["N"], delta_t=w["dt"], lr=1e-3, L=1, seed=42, K_boundary=50, alpha=tuple(w.get("alpha", (1.0, 1.0, 1.0))),
                                loss_method=w["loss"], uniform_square=bool(w.get("uniform_square", False)))
        V, train = orc.general_build(prob, cfg, net=net), orc.general_train
    train(prob, cfg, V=V)
    t0, act = time.time(), 0
    for _ in range(iters):
        act += train(prob, cfg, V=V)["K_log"][0]
    el = time.time() - t0
    return {"value": act / el, "unit": "trajectory-timesteps/s", "cores": allc, "kind": "port", "s_per_iteration": el / iters,
            "sample": "oracle %s, the whole workload (K=%d, N=%d), %d iterations, %.1f s, torch %s"
                      % (train.__name__, w["K"], w["N"], iters, el, torch.__version__)}


def main_general(args, psp, sharding):
    """GeneralSolver (diffusion / BSDE loss) workloads: metric counts ACTIVE trajectory-timesteps
    (K_log, reference solver.py:1152), as SURVEY.md 8d prescribes."""
    w = GENERAL_WORKLOADS[args.workload]
    dist, rank, world, dev, rehearsal = init_ranks(args)
    total = args.warmup + args.steps
    arch = w.get("arch") or [w["H"], w["H"]]
    prob, model = make_general_model(psp, w, dev, arch, w["K"] * world, total, "native", "philox", mlp=w.get("mlp", "auto"))
    plan = model._choose_plan()
    assert model.plan_name == "native"
    deep = type(plan).__name__ == "GeneralDeepPlan"
    for l in range(args.warmup):
        plan.iteration(l)
    plan.events = []
    sharding.coll_events = [] if world > 1 else None
    counts, losses = [], []

    def one(l):
        loss, kc = plan.iteration(l)
        counts.append(kc)
        losses.append(loss)

    elapsed = timed_region(dist, dev, one, args.steps, args.warmup)
    coll = collective_summary(sharding, args.steps)
    sharding.coll_events = None
    active = float(torch.stack(counts).sum().item())
    fwd_ms = sum(e[0].elapsed_time(e[1]) for e in plan.events) / max(1, len(plan.events))
    bwd_ms = sum(e[2].elapsed_time(e[3]) for e in plan.events) / max(1, len(plan.events))
    # algorithmic flops per LAUNCHED trajectory-timestep: value net F = 2[(d+1)H + (d+1+H)H + (d+1+2H)],
    # its input gradient ~ F (reverse sweep), and the double-backward of both ~ 2 x that: forward kernel
    # 2F (+ F for the tangent part it pre-computes), backward kernel 3F  -> 6F per unit in total
    n_in, F = w["d"] + (0 if w.get("solver") == "elliptic" else 1), 0
    for h_ in arch:                                              # dense-concat layers: in_i = d + 1 + sum of the earlier widths
        F += 2 * n_in * h_
        n_in += h_
    F += 2 * n_in
    units = w["K"] * w["N"]
    mlp = w.get("mlp")
    bwd_dom = bwd_ms >= fwd_ms
    dom, dom_ms, dom_fl = ("gen_bwd2_kernel", bwd_ms, 3 * F) if bwd_dom else ("gen_fwd_kernel", fwd_ms, 3 * F)
    if deep:
        # run-time-shaped family: forward = value + reverse sweep (2 F); backward = ONE kernel: recompute value and tangent (2 F),
        # adjoints (2 F), weight-gradient outer products (2 F)
        dom, dom_fl = ("genl_bwd_kernel", 6 * F) if bwd_dom else ("genl_fwd_kernel", 2 * F)
    if w.get("solver") == "elliptic":
        # exit-time problem: a tile leaves the time loop once all of its trajectories have left the domain -- the launched units are
        # the executed steps, measured (not K x N)
        units = active / args.steps
    # every product of these kernels is a value-net product: on bf16 workloads the whole kernel is priced at the bf16 MFMA
    # peak (forward always; backward only with mlp == 'bf16')
    on_bf16 = (mlp in ("bf16", "bf16_fwd") and not bwd_dom) or (mlp == "bf16" and bwd_dom)
    peak = PEAK_BF16_MFMA_TFLOPS if on_bf16 else PEAK_FP32_MFMA_TFLOPS
    # split-product kernels (mlp_dtype 'auto' / 'f16x3', forward and backward): three f16 flops per algorithmic fp32 flop
    x3 = getattr(plan, "matrix_mode", "fp32") == "f16x3"
    if x3:
        peak = PEAK_BF16_MFMA_TFLOPS / 3.0
    achieved = dom_fl * units / (dom_ms * 1e-3) / 1e12
    path_B = float(plan.sizes.path_bytes) / ((w["N"] + 1) * w["K"]) if hasattr(plan, "sizes") else None
    hbm_gbps = (float(plan.sizes.path_bytes) / (dom_ms * 1e-3) / 1e9) if hasattr(plan, "sizes") else 0.0
    alg_B = 8 * w["d"] + 8                                       # state streamed once per step
    hbm_alg_gbps = alg_B * units / (dom_ms * 1e-3) / 1e9
    mf_frac, hbm_frac, hbm_store_frac = achieved / peak, hbm_alg_gbps / PEAK_HBM_GBS, hbm_gbps / PEAK_HBM_GBS
    bound = "mfma" if mf_frac >= hbm_frac else "hbm"
    counters = load_counters(args.workload, dom)
    if rank == 0:
        out = {"metric": "active trajectory-timesteps/sec, d=100 %s loss training iteration" % w["loss"],
               "value": active / elapsed, "unit": "trajectory-timesteps/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None,
               "dtype": {"bf16": "bf16 MFMA operands in both rollout kernels, f32 state / accumulate / element-wise",
                         "bf16_fwd": "bf16 MFMA operands in the forward rollout, f32 state / accumulate / backward"}.get(
                             mlp, "f32 (f16x3 split products: every matrix product as three f16 MFMAs, fp32 accumulate)" if x3 else "f32"),
               "data": "synthetic",
               "config": {"workload": args.workload, "problem": w.get("problem", "DoubleWell_multidim_for_general_solver"),
                          "solver": "EllipticSolver" if w.get("solver") == "elliptic" else "GeneralSolver",
                          "d": w["d"], "K_per_gpu": w["K"], "N": w["N"],
                          "V": "dense-concat %s net %d-%s-1" % (w.get("act", "relu2"), w["d"] + (0 if w.get("solver") == "elliptic" else 1), "-".join(str(h_) for h_ in arch)),
                          "loss": w["loss"], "active_fraction": active / (w["K"] * world * w["N"] * args.steps)},
               "roofline": {"bound": bound, "kernel": dom,
                            "achieved": achieved if bound == "mfma" else hbm_alg_gbps,
                            "peak": peak if bound == "mfma" else PEAK_HBM_GBS,
                            "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                            "frac": max(mf_frac, hbm_frac),
                            "traffic": load_traffic(args.workload, dom),
                            "binder": binder_of(mf_frac, hbm_frac, hbm_store_frac, counters, (w["K"] // 16) * (w["N"] + 1)),
                            "convention": "algorithmic work per launched trajectory-timestep x units per launch / HIP-event kernel time / "
                                          "peak of the pipe the kernel computes on (%s); hbm_term.frac_path_store prices the design's own bytes"
                                          % ("2500/3 TFLOP/s split products" if x3 else ("2500 TFLOP/s bf16" if on_bf16 else "157.3 TFLOP/s fp32 MFMA")),
                            "mfma_term": {"achieved": achieved, "peak": peak, "frac": mf_frac},
                            "hbm_term": {"bytes_per_unit_alg": alg_B, "achieved_GBps_alg": hbm_alg_gbps, "frac_alg_bytes": hbm_frac,
                                         "bytes_per_sample_slot_path_store": path_B, "achieved_GBps_path_store": hbm_gbps,
                                         "frac_path_store": hbm_store_frac, "peak_GBps": PEAK_HBM_GBS, "frac": hbm_frac},
                            "alg_flops_per_launched_unit": {"value_net_F": F, "fwd_kernel": 3 * F, "bwd_kernel": 3 * F},
                            "units_per_launch": units, "fwd_kernel_ms": fwd_ms, "bwd_kernel_ms": bwd_ms},
               "launched_units_per_s": w["K"] * world * w["N"] * args.steps / elapsed,
               "s_per_iteration": elapsed / args.steps,
               "loss_first_last": [float(losses[0]), float(losses[-1])]}
        if w.get("solver") == "elliptic":
            out["config"]["last_batch_size"] = model.K
            out["config"]["mean_active_steps_per_iteration"] = active / args.steps
            out["config"]["waves_per_tile"] = int(plan.sizes.waves_per_tile) if deep else None
        if coll is not None:
            coll["fraction_of_step"] = coll["per_step_ms"] / (1e3 * elapsed / args.steps)
            out["collectives"] = coll
        if getattr(plan, "range_flag", None) is not None:
            out["config"]["range_guard"] = "on (include/psp.h range_flag); fallback iterations in this run: %d" % plan.range_fallbacks()
        if w.get("published_s_per_iter"):
            lo_s, hi_s = w["published_s_per_iter"]
            out["reference_published"] = {"s_per_iteration": [lo_s, hi_s], "units_per_s": w.get("published_units_per_s"),
                                          "hardware": "unnamed CUDA GPU",
                                          "source": w.get("published_source", "experiments/diffusion-loss/Allen-Cahn.ipynb:86,115 (BASELINE.md section 1)"),
                                          "iteration_time_ratio": [lo_s / (elapsed / args.steps), hi_s / (elapsed / args.steps)]}
            if w.get("published_units_per_s"):
                out["vs_baseline"] = out["value"] / w["published_units_per_s"]
            else:
                # the notebook prints seconds per iteration, not the number of active steps; the same sampler and net give the same
                # distribution of exit times, so the published rate is this run's active steps per iteration over the published time
                mid = 0.5 * (lo_s + hi_s)
                out["reference_published"]["units_per_s_derived"] = (active / args.steps) / mid
                out["reference_published"]["note"] = ("units/s derived from this run's mean active steps per iteration and the midpoint of the "
                                                      "published seconds per iteration (the notebook logs no step counts)")
                out["vs_baseline"] = mid / (elapsed / args.steps)
            if world == 1 and not args.no_secondary:
                out["also_composite_torch_plan_same_gpu"] = general_composite_leg(psp, dev, w, arch)
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = general_cpu_baseline(w, arch)
        if world == 1 and not args.no_sustained:
            n_s, t_s, c_s = 0, 0.0, []
            plan.events = None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            while t_s < args.sustain_s:
                for l in range(total + n_s, total + n_s + 20):
                    c_s.append(plan.iteration(l)[1])
                n_s += 20
                torch.cuda.synchronize()
                t_s = time.perf_counter() - t0
            out["also_sustained"] = {"value": float(torch.stack(c_s).sum().item()) / t_s, "unit": "trajectory-timesteps/s",
                                     "ms_per_step": 1e3 * t_s / n_s, "steps": n_s, "seconds": t_s}
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
