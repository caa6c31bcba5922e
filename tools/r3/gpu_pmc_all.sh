#!/bin/bash
# Round-3 PMC passes (separate rocprofv3 --pmc runs, kernel trace only) for the judged workloads; run through gpurun from the
# repo root.  Summaries land in gpurun_out/pmc/<workload>_summary.txt; tools/r3/make_counters_json.py turns them into
# profiles/r3_counters.json (read by bench.py for `traffic` and `binder`).
set -o pipefail
for W in hjb_llgc_d100_K65536_N100_h64 hjb_llgc_d100_K65536_N50_outer_h30 diffusion_dw_d100_K65536_N100_h64 \
         hjb_llgc_d200_K32768_N100_h64 hjb_llgc_d500_K16384_N200_h64 diffusion_allencahn_d100_K200_N25_a110 \
         diffusion_dw_d100_K65536_N100_h64_bf16 hjb_llgc_d100_K1024_N50_h64; do
    echo "== $W"
    timeout -k 10 500 bash tools/pmc_passes.sh $W > gpurun_out/pmc_$W.log 2>&1 || { echo "PMC passes of $W failed"; tail -5 gpurun_out/pmc_$W.log; exit 1; }
    tail -2 gpurun_out/pmc_$W.log
done
