#!/bin/bash
# Round-4 PMC passes (separate rocprofv3 --pmc runs, kernel trace only) for the judged workloads; run through gpurun from the
# repo root.  Summaries land in gpurun_out/pmc/<workload>_summary.txt; `python tools/r3/make_counters_json.py r4_counters.json`
# turns them into profiles/r4_counters.json (read by bench.py for `traffic` and `binder`).
set -o pipefail
for W in "$@"; do
    echo "== $W"
    timeout -k 10 500 bash tools/pmc_passes.sh $W > gpurun_out/pmc_$W.log 2>&1 || { echo "PMC passes of $W failed"; tail -5 gpurun_out/pmc_$W.log; exit 1; }
    tail -2 gpurun_out/pmc_$W.log
done
