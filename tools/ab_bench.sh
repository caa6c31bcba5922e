#!/bin/bash
# A/B on ONE box: bench.py lines (ms per iteration, forward / backward kernel ms) for two builds of the library
# usage: tools/ab_bench.sh <libA> <libB> <workload>...
set -o pipefail
OUT=gpurun_out/ab_bench; mkdir -p $OUT
A=$1; B=$2; shift 2
for w in "$@"; do
  for L in $A $B $A $B; do
    PSP_LIB_PATH=$PWD/$L timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-secondary --no-sustained > $OUT/line.json 2>$OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
    python - "$w" "$L" <<'PY' | tee -a $OUT/ab.txt
import json, sys
j = json.loads(open("gpurun_out/ab_bench/line.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("%-48s %-22s %8.3f ms/it  fwd %7.3f  bwd %7.3f" % (sys.argv[1], sys.argv[2].split("/")[-1], j["ms_per_step"], r.get("fwd_kernel_ms") or 0, r.get("bwd_kernel_ms") or 0))
PY
  done
done
