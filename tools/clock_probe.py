"""Sustained-clock / power probe for the two hot kernels (diagnostic, not part of the product path).

Runs one kernel of the headline workload back to back for a few seconds while a sampler thread polls
`rocm-smi --showclocks --showpower`, and prints mean kernel time, mean shader clock and mean socket power.
The fp32 MFMA peak quoted in MI355X_MICROARCH.md assumes 2.4 GHz; this shows the clock the chip actually
sustains under each kernel, i.e. how much of the distance to that peak is clock rather than schedule.

    python tools/clock_probe.py [--seconds 4] [--mode fwd|bwd|both]
    PSP_BWD_VARIANT=1 python tools/clock_probe.py --mode bwd      # the two-workgroups-per-CU backward
"""
import argparse
import ctypes as C
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sample_smi():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True,
                             text=True, timeout=20).stdout
        j = json.loads(out)
        card = j[sorted(j.keys())[0]]
        sclk = power = None
        for k, v in card.items():
            if 'sclk' in k.lower():
                m = re.search(r'(\d+)\s*Mhz', str(v), re.I)
                if m:
                    sclk = float(m.group(1))
            if 'power' in k.lower() and 'socket' in k.lower():
                try:
                    power = float(v)
                except ValueError:
                    pass
        return sclk, power, card
    except Exception as e:                                   # the probe must never take the run down
        return None, None, {'error': repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=4.0)
    ap.add_argument('--mode', default='both')
    ap.add_argument('--workload', default='hjb_llgc_d100_K65536_N100_h64')
    args = ap.parse_args()
    import torch
    import bench
    import importlib
    psp = importlib.import_module('path_space_pde_solver_amd')
    from path_space_pde_solver_amd import native as nat       # noqa: F401
    dev = torch.device('cuda:0')
    w = bench.WORKLOADS[args.workload]
    prob = psp.LLGC(d=w["d"], off_diag=w["off_diag"], T=w["T"], seed=42, device=dev)
    model = psp.Solver("probe", prob, lr=1e-3, L=8, K=w["K"], delta_t=w["dt"], loss_method="log-variance",
                       time_approx="inner", adaptive_forward_process=True, detach_forward=True,
                       u_l2_error_flag=False, verbose=False, seed=42, device=dev, backend="native",
                       noise="philox", widths=(w["H"], w["H"]))
    plan = model._choose_plan()
    losses = torch.zeros(8, device=dev)
    plan.iteration(0, losses)                                 # fills path / D / sums once
    torch.cuda.synchronize()
    lib, cfg, st = plan.lib, plan.cfg, plan._stream()
    seed = int(model.seed)

    def fwd():
        lib.psp_hjb_rollout_fwd(C.byref(cfg), nat.ptr(plan.flat), nat.ptr(plan.x0_vec), 0, None, None, seed, 1,
                                nat.ptr(plan.path), nat.ptr(plan.D), None, nat.ptr(plan.Yn),
                                nat.ptr(plan.fwd_partial), st)

    def bwd():
        lib.psp_hjb_rollout_bwd(C.byref(cfg), nat.ptr(plan.flat), None, seed, 1, nat.ptr(plan.path),
                                nat.ptr(plan.D), nat.ptr(plan.sums), nat.ptr(plan.grad_partial),
                                nat.ptr(plan.grad), st)

    modes = {'fwd': [fwd], 'bwd': [bwd], 'both': [fwd, bwd]}
    for name in (['fwd', 'bwd', 'both'] if args.mode == 'both' else [args.mode]):
        fns = modes[name]
        samples, stop = [], threading.Event()

        def sampler():
            while not stop.is_set():
                samples.append(sample_smi()[:2])
                time.sleep(0.2)
        th = threading.Thread(target=sampler)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        th.start()
        t0 = time.time()
        e0.record()
        while time.time() - t0 < args.seconds:
            for _ in range(20):
                for f in fns:
                    f()
                n += 1
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        ms = e0.elapsed_time(e1) / n
        sc = [s for s, _ in samples[1:] if s]
        pw = [p for _, p in samples[1:] if p]
        print(json.dumps({'mode': name, 'variant': os.environ.get('PSP_BWD_VARIANT', 'default'), 'launches': n,
                          'ms_per_launch': ms, 'sclk_mhz_mean': sum(sc) / len(sc) if sc else None,
                          'sclk_mhz_min_max': [min(sc), max(sc)] if sc else None,
                          'power_w_mean': sum(pw) / len(pw) if pw else None, 'n_samples': len(samples)}))
        time.sleep(1.0)
    print(json.dumps({'idle_sample': sample_smi()[2]}))


if __name__ == '__main__':
    main()
