#!/bin/bash
# Round-2 GPU session 1: GPU test suite, default bench line, N>1 rehearsal, chunked workloads.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r2s1
mkdir -p $OUT
step() {  # name, timeout, command...
    local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $OUT/progress.log
    timeout -k 10 $to "$@" > $OUT/$name.out 2> $OUT/$name.err
    local rc=$?
    echo "   rc=$rc" | tee -a $OUT/progress.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "   TIMEOUT -- stopping" | tee -a $OUT/progress.log; exit $rc; fi
    return 0
}
step pytest_new 600 python -m pytest tests/test_gpu_chunked.py tests/test_gpu_collective.py -x -q -s
step pytest_all 900 python -m pytest tests -m gpu -q
step bench_default 300 python bench.py --steps 10 --warmup 3
step bench_rehearsal2 300 env PSP_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 5 --warmup 2 --workload hjb_llgc_d100_K16384_N50_h64 --no-cpu-baseline
step bench_chunk4 300 python bench.py --steps 10 --warmup 3 --workload hjb_llgc_d100_K65536_N100_h64_chunk4 --no-cpu-baseline
step bench_d500_K131072 600 python bench.py --steps 3 --warmup 1 --workload hjb_llgc_d500_K131072_N200_h64 --no-cpu-baseline
step bench_d200_strong 600 python bench.py --steps 3 --warmup 1 --workload hjb_llgc_d200_Kglobal262144_N100_h64 --no-cpu-baseline
tail -5 $OUT/*.out | cut -c1-600
