#!/bin/bash
# Diagnostic PMC passes for the headline forward kernel: which unit its waves wait on (run through gpurun from the repo root).
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
W=${1:-hjb_llgc_d100_K65536_N100_h64}
OUT=$ROOT/gpurun_out/pmc_diag
mkdir -p "$OUT"
: > "$OUT/${W}_diag.txt"
i=0
for C in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SALU" \
         "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    i=$((i + 1))
    rm -rf /tmp/pmcd_${W}_$i
    (cd /tmp && rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcd_${W}_$i -o run --output-format csv -- \
        python3 "$ROOT/bench.py" --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sustained > /dev/null 2> "$OUT/${W}_pass$i.err") || { tail -5 "$OUT/${W}_pass$i.err"; exit 1; }
    F=$(find /tmp/pmcd_${W}_$i -name "*counter_collection.csv" | head -1)
    python3 - "$F" >> "$OUT/${W}_diag.txt" <<'PY'
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
cat "$OUT/${W}_diag.txt"
