#!/bin/bash
# Round-4 GPU batch: the whole GPU suite, then short bench lines of the kernels that changed (one gpurun call).
# usage (from the repo root, on the GPU box):  bash tools/r4/gpu_batch.sh <out-dir> [skip-tests]
OUT=${1:-gpurun_out/r4_batch}
mkdir -p $OUT
if [ "$2" != "skip-tests" ]; then
  timeout -k 10 800 python -m pytest tests -q -m gpu > $OUT/pytest_gpu_all.log 2>&1
  tail -4 $OUT/pytest_gpu_all.log
fi
B="--steps 20 --warmup 5 --no-sustained --no-cpu-baseline --no-secondary"
line() {  # workload-json -> one summary line
  python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); rf = r["roofline"]
    print(sys.argv[2], "ms %.3f" % r["ms_per_step"], "fwd", rf.get("fwd_kernel_ms"), "bwd", rf.get("bwd_kernel_ms"), "value %.3e" % r["value"], "frac %.3f" % rf["frac"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for w in hjb_llgc_d100_K65536_N100_h64 hjb_llgc_d100_K65536_N50_outer_h30 diffusion_dw_d100_K65536_N100_h64 diffusion_dw_d100_K65536_N100_h64_bf16 hjb_llgc_d200_K32768_N100_h64 hjb_llgc_d500_K16384_N200_h64; do
  timeout -k 10 150 python bench.py --workload $w $B > $OUT/$w.json 2> $OUT/$w.err
  line $OUT/$w.json $w
done
for w in hjb_llgc_d200_K32768_N100_h64 hjb_llgc_d500_K16384_N200_h64; do
  PSP_WIDE_SPEC=1 timeout -k 10 150 python bench.py --workload $w $B > $OUT/${w}_spec.json 2> $OUT/${w}_spec.err
  line $OUT/${w}_spec.json "$w WIDE_SPEC"
done
