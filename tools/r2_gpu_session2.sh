#!/bin/bash
# Round-2 GPU session 2: new tests (graph, full size, tightened tolerances), small-K bench.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r2s2
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
step pytest_graph 300 python -m pytest tests/test_gpu_graph.py -x -q -s
step pytest_full 600 python -m pytest tests/test_gpu_full_size.py -q -s
step pytest_general 600 python -m pytest tests/test_gpu_general.py tests/test_gpu_bounded_elliptic.py -q -s
step bench_k1024 300 python bench.py --steps 200 --warmup 20 --workload hjb_llgc_d100_K1024_N50_h64 --no-cpu-baseline
step bench_k4096 300 python bench.py --steps 200 --warmup 20 --workload hjb_llgc_d100_K4096_N50_h64 --no-cpu-baseline
step pytest_all 900 python -m pytest tests -m gpu -q
for f in $OUT/*.out; do echo "--- $f"; tail -c 1500 $f; done
