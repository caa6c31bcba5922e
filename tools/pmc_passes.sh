#!/bin/bash
# PMC passes for one bench workload on an MI355X box (run through gpurun from the repo root), each in its own
# rocprofv3 run with --kernel-trace only (no other trace domain), as MI355X_MICROARCH.md prescribes:
#   pass 1  SQ instruction / busy counters     pass 2  FETCH_SIZE     pass 3  WRITE_SIZE
# Writes gpurun_out/pmc/<workload>_summary.txt: mean per dispatch and kernel.
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
W=${1:-hjb_llgc_d100_K65536_N100_h64}
OUT=$ROOT/gpurun_out/pmc
mkdir -p "$OUT"
: > "$OUT/${W}_summary.txt"
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rm -rf /tmp/pmc_${W}_$i
    (cd /tmp && rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_${W}_$i -o run --output-format csv -- \
        python3 "$ROOT/bench.py" --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sustained > /dev/null 2> "$OUT/${W}_pass$i.err") || exit 1
    F=$(find /tmp/pmc_${W}_$i -name "*counter_collection.csv" | head -1)
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
        print("%-60s %-28s mean %.6g  (n=%d)" % (k[-60:], c, sum(v) / len(v), len(v)))
PY
done
cat "$OUT/${W}_summary.txt"
