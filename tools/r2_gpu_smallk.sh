#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r2smallk
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_graph.py -q -x > $OUT/pytest.out 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.out
for W in hjb_llgc_d100_K1024_N50_h64 hjb_llgc_d100_K4096_N50_h64; do
  timeout -k 10 300 python bench.py --steps 300 --warmup 20 --workload $W --no-cpu-baseline > $OUT/$W.json 2> $OUT/$W.err
  python3 -c "
import json; d=json.loads(open('$OUT/$W.json').read().strip().splitlines()[-1]); print('$W', '%.4e'%d['value'], d['ms_per_step'], d['roofline']['fwd_kernel_ms'], d['roofline']['bwd_kernel_ms'])"
done
