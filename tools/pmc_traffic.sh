#!/bin/bash
# HBM traffic of the rollout kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel trace only), mean per
# dispatch, for each workload given.  tools/make_traffic_json.py turns gpurun_out/r2traffic/*.txt into profiles/traffic.json
# (2 x FETCH_SIZE + WRITE_SIZE, the factor calibrated by tools/fetch_calibrate.hip).
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r2traffic
mkdir -p $OUT
for W in "$@"; do
  : > $OUT/$W.txt
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_MFMA; do
    rm -rf /tmp/tr_${W}_$C
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/tr_${W}_$C -o run --output-format csv -- \
        python3 "$ROOT/bench.py" --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/${W}_$C.err) || { echo "$W $C failed"; tail -2 $OUT/${W}_$C.err; continue; }
    F=$(find /tmp/tr_${W}_$C -name "*counter_collection.csv" | head -1)
    python3 - "$F" >> $OUT/$W.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void psp::", "")
    if "psp::" not in row["Kernel_Name"]:
        continue
    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("%s %s %.6g %d" % (k, c, sum(v) / len(v), len(v)))
PY
  done
  echo "== $W"; cat $OUT/$W.txt
done
