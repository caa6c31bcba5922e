#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r2s5
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
step pytest_variants 600 python -m pytest tests/test_gpu_general_variants.py tests/test_gpu_general.py tests/test_gpu_value_function.py -q -x
step bench_bf16 300 python bench.py --steps 20 --warmup 5 --workload diffusion_dw_d100_K65536_N100_h64_bf16
step bench_bf16fwd 300 python bench.py --steps 20 --warmup 5 --workload diffusion_dw_d100_K65536_N100_h64_bf16fwd
step bench_fp32 300 python bench.py --steps 20 --warmup 5 --workload diffusion_dw_d100_K65536_N100_h64
for f in $OUT/pytest_variants.out; do echo "--- $f"; tail -c 1500 $f; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2s5/bench_*.out')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.3e'%d['value'], d['ms_per_step'], d['roofline']['fwd_kernel_ms'], d['roofline']['bwd_kernel_ms'], d['loss_first_last'], d['roofline']['bound'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
