#!/bin/bash
# Cache-side PMC passes for one bench workload (run through gpurun from the repo root): where the table stream of the wide
# kernels is served from.  One rocprofv3 run per counter set, --kernel-trace only.
#   TCP_TOTAL_CACHE_ACCESSES_sum / TCP_TCC_READ_REQ_sum : vector-L1 accesses and the reads it passes on to L2
#   TCC_HIT_sum / TCC_MISS_sum / TCC_REQ_sum            : L2
# Writes gpurun_out/pmc/<workload>_cache.txt: mean per dispatch and kernel.
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
W=${1:-hjb_llgc_d500_K16384_N200_h64}
OUT=$ROOT/gpurun_out/pmc
mkdir -p "$OUT"
rocprofv3 --list-avail 2>/dev/null | grep -o "TCP_[A-Z_]*\(sum\)\?\|TCC_[A-Z_]*\(sum\)\?" | sort -u > "$OUT/avail_cache_counters.txt"
: > "$OUT/${W}_cache.txt"
i=0
for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
    i=$((i + 1))
    rm -rf /tmp/pmcc_${W}_$i
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcc_${W}_$i -o run --output-format csv -- \
        python3 "$ROOT/bench.py" --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> "$OUT/${W}_cache_pass$i.err")
    rc=$?
    [ $rc -ge 124 ] && { echo "pass $i killed ($rc)"; exit 1; }
    [ $rc -ne 0 ] && { echo "pass $i ($C) failed: $(tail -2 "$OUT/${W}_cache_pass$i.err")"; continue; }
    F=$(find /tmp/pmcc_${W}_$i -name "*counter_collection.csv" | head -1)
    python3 - "$F" >> "$OUT/${W}_cache.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0]
    if "psp::" not in k:
        continue
    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("%-60s %-32s mean %.6g  (n=%d)" % (k[-60:], c, sum(v) / len(v), len(v)))
PY
done
cat "$OUT/${W}_cache.txt"
