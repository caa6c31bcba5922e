#!/bin/bash
# Round-2 GPU session 3: 8-wave split forward -- parity, stamps, small-K bench.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r2s3
mkdir -p $OUT
step() {
    local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $OUT/progress.log
    timeout -k 10 $to "$@" > $OUT/$name.out 2> $OUT/$name.err
    local rc=$?
    echo "   rc=$rc" | tee -a $OUT/progress.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "   TIMEOUT -- stopping" | tee -a $OUT/progress.log; exit $rc; fi
    return 0
}
step pytest_parity 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py tests/test_gpu_shape_sweep.py tests/test_gpu_fuzz.py -q -x
step stamps 300 python tools/split_stamps.py 1024
step bench_k1024 300 python bench.py --steps 200 --warmup 20 --workload hjb_llgc_d100_K1024_N50_h64 --no-cpu-baseline
step bench_k4096 300 python bench.py --steps 200 --warmup 20 --workload hjb_llgc_d100_K4096_N50_h64 --no-cpu-baseline
step bench_k8192 300 python bench.py --steps 100 --warmup 20 --workload hjb_llgc_d100_K8192_N50_h64 --no-cpu-baseline
step bench_k8192_tile 300 env PSP_FWD_VARIANT=1 python bench.py --steps 100 --warmup 20 --workload hjb_llgc_d100_K8192_N50_h64 --no-cpu-baseline
for f in $OUT/pytest_parity.out $OUT/stamps.out; do echo "--- $f"; tail -c 1500 $f; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2s3/bench_*.out')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.3e'%d['value'], d['ms_per_step'], d['roofline']['fwd_kernel_ms'], d['roofline']['bwd_kernel_ms'], d['config'].get('launch'))
    except Exception as e: print(f, 'ERR', e)
PY
