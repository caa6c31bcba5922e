#!/bin/bash
# whole GPU suite + the headline family of bench lines
OUT=${1:-gpurun_out/r4_batch2}
mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -q -m gpu > $OUT/pytest_gpu_all.log 2>&1
tail -4 $OUT/pytest_gpu_all.log
B="--steps 20 --warmup 5 --no-sustained --no-cpu-baseline --no-secondary"
for w in hjb_llgc_d100_K65536_N100_h64 hjb_llgc_d100_K1024_N50_h64 hjb_llgc_d100_K8192_N50_h64 hjb_llgc_d100_K65536_N100_h64_fp32mfma; do
  timeout -k 10 150 python bench.py --workload $w $B > $OUT/$w.json 2> $OUT/$w.err
  python - $OUT/$w.json $w <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); rf = r["roofline"]
    print(sys.argv[2], "ms %.4f" % r["ms_per_step"], "fwd", rf.get("fwd_kernel_ms"), "bwd", rf.get("bwd_kernel_ms"), "value %.3e" % r["value"], "frac %.3f" % rf["frac"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
