#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r2wide
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
step pytest_wide 600 python -m pytest tests/test_gpu_wide_family.py tests/test_gpu_full_size.py "tests/test_gpu_parity.py::test_first_iteration_D_and_gradient_match_oracle" -q -x
step bench_d200 300 python bench.py --steps 20 --warmup 5 --workload hjb_llgc_d200_K32768_N100_h64 --no-cpu-baseline
step bench_d500 300 python bench.py --steps 10 --warmup 3 --workload hjb_llgc_d500_K16384_N200_h64 --no-cpu-baseline
tail -3 $OUT/pytest_wide.out
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2wide/bench_*.out')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.3e'%d['value'], d['ms_per_step'], d['roofline']['fwd_kernel_ms'], d['roofline']['bwd_kernel_ms'], d['roofline']['mfma_term']['frac_issued'])
    except Exception as e: print(f, 'ERR', e)
PY
