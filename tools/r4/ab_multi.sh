#!/bin/bash
# A/B/C... on ONE box: bench.py lines for several builds of the library, interleaved twice
# usage: tools/r4/ab_multi.sh "<lib1> <lib2> ..." <workload>...
set -o pipefail
OUT=gpurun_out/ab_multi; mkdir -p $OUT
LIBS=$1; shift
for w in "$@"; do
  for rep in 1 2; do
    for L in $LIBS; do
      PSP_LIB_PATH=$PWD/path-space-pde-solver_amd/csrc/$L timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-secondary --no-sustained > $OUT/line.json 2>$OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
      python - "$w" "$L" <<'PY' | tee -a $OUT/ab.txt
import json, sys
j = json.loads(open("gpurun_out/ab_multi/line.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("%-48s %-22s %8.3f ms/it  fwd %7.3f  bwd %7.3f" % (sys.argv[1], sys.argv[2], j["ms_per_step"], r.get("fwd_kernel_ms") or 0, r.get("bwd_kernel_ms") or 0))
PY
    done
  done
done
