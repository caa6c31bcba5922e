#!/bin/bash
# A/B of an environment switch on ONE box: bench.py lines with VAR=a / VAR=b interleaved twice
# usage: tools/r4/ab_env.sh VAR "a b" <workload>...
set -o pipefail
OUT=gpurun_out/ab_env; mkdir -p $OUT
VAR=$1; VALS=$2; shift 2
for w in "$@"; do
  for rep in 1 2; do
    for v in $VALS; do
      env $VAR=$v timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-secondary --no-sustained > $OUT/line.json 2>$OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
      python - "$w" "$VAR=$v" <<'PY' | tee -a $OUT/ab.txt
import json, sys
j = json.loads(open("gpurun_out/ab_env/line.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("%-48s %-22s %8.3f ms/it  fwd %7.3f  bwd %7.3f  loss_rel_err %s" % (sys.argv[1], sys.argv[2], j["ms_per_step"], r.get("fwd_kernel_ms") or 0, r.get("bwd_kernel_ms") or 0, j.get("loss_rel_err_vs_cpu")))
PY
    done
  done
done
