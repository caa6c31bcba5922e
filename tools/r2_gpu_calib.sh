#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r2calib
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$C
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/cal_$C -o run --output-format csv -- $ROOT/tools/fetch_calibrate > $OUT/cal_$C.out 2> $OUT/cal_$C.err) || { echo "calibration $C failed"; tail -3 $OUT/cal_$C.err; exit 1; }
  F=$(find /tmp/cal_$C -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$C" <<'PY' | tee -a $OUT/calibration.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
B = 2 << 30
for k in sorted(acc):
    v = sum(acc[k]) / len(acc[k])
    print("%-12s %-10s reported %.6g KB = %.4f x the %d bytes streamed" % (sys.argv[2], k, v, v * 1024 / B, B))
PY
done
# config 4 at its FULL size on one GPU (K = 1048576, 32 chunks)
timeout -k 10 600 python3 bench.py --workload hjb_llgc_d500_K1048576_N200_h64 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_K1M.json 2> $OUT/bench_K1M.err; echo "K1M rc=$?"; cut -c1-700 $OUT/bench_K1M.json
