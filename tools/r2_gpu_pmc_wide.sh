#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r2pmc
mkdir -p $OUT
(cd /tmp && rocprofv3 -L > $OUT/counters.txt 2>&1) || true
grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|TA_[A-Z_0-9]*" $OUT/counters.txt | sort -u > $OUT/counter_names.txt
wc -l $OUT/counter_names.txt
W=${1:-hjb_llgc_d500_K16384_N200_h64}
: > "$OUT/${W}_summary.txt"
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rm -rf /tmp/pmc_${W}_$i
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_${W}_$i -o run --output-format csv -- \
        python3 "$ROOT/bench.py" --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> "$OUT/${W}_pass$i.err") || { echo "pass $i ($C) failed"; tail -2 "$OUT/${W}_pass$i.err"; continue; }
    F=$(find /tmp/pmc_${W}_$i -name "*counter_collection.csv" | head -1)
    [ -z "$F" ] && { echo "pass $i: no csv"; continue; }
    python3 - "$F" >> "$OUT/${W}_summary.txt" <<'PY'
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
cat "$OUT/${W}_summary.txt"
