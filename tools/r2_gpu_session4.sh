#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r2s4
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
step pytest_vf 600 python -m pytest tests/test_gpu_value_function.py -q -s
step pytest_general 600 python -m pytest tests/test_gpu_general.py tests/test_gpu_bounded_elliptic.py tests/test_gpu_general_variants.py -q
for f in $OUT/*.out; do echo "--- $f"; tail -c 2500 $f; done
