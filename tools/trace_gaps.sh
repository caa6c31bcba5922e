#!/bin/bash
# Kernel timeline of one workload (rocprofv3 --kernel-trace): per-iteration busy time, idle gaps and the kernels around the
# largest gaps.  Usage: bash tools/trace_gaps.sh <workload>
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd); W=$1; OUT=$ROOT/gpurun_out/trace; mkdir -p $OUT; rm -rf /tmp/tg_$W
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace -d /tmp/tg_$W -o run --output-format csv -- \
    python3 "$ROOT/bench.py" --workload $W --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/$W.err) || { tail -3 $OUT/$W.err; exit 1; }
F=$(find /tmp/tg_$W -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY' | tee $OUT/$W.txt
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# iterations: split at the forward rollout kernels
idx = [i for i, r in enumerate(rows) if "fwd_kernel" in r[2]]
if len(idx) < 4:
    print("too few forward launches"); sys.exit(0)
a, b = idx[-3], idx[-2]                      # one full steady-state iteration: [fwd_a, fwd_b)
it = rows[a:b]
span = rows[b][0] - rows[a][0]
busy = sum(e - s for s, e, _ in it)
print("iteration span %.3f ms, kernels %d, busy %.3f ms, idle %.3f ms" % (span / 1e6, len(it), busy / 1e6, (span - busy) / 1e6))
big = sorted(it, key=lambda r: r[0] - r[1])[:3]
for s, e, n in big:
    print("  %8.1f us  %s" % ((e - s) / 1e3, n[:90]))
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(it, it[1:] + [rows[b]]):
    gaps.append((s1 - e0, n0[:60], n1[:60]))
gaps.sort(reverse=True)
print("largest gaps:")
for g, n0, n1 in gaps[:8]:
    print("  %8.1f us  after %s | before %s" % (g / 1e3, n0, n1))
small = [(e - s) for s, e, n in it if (e - s) < 200000]
print("small kernels (< 0.2 ms): %d, total %.3f ms" % (len(small), sum(small) / 1e6))
PY
