#!/bin/bash
# Regenerates the judged measurement files on an MI355X box (run through gpurun from the repo root):
#   gpurun_out/final/<workload>_bench.json          bench.py JSON line (HIP-event roofline, cpu_baseline on the headline)
#   gpurun_out/final/<workload>_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary of the same command
# Copy them to profiles/r1_final_<workload>_* afterwards.  rocprofv3 gets the interpreter itself after "--".
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/final
mkdir -p "$OUT"
for W in hjb_llgc_d100_K65536_N100_h64 hjb_llgc_d100_K1024_N50_h64 hjb_llgc_d200_K32768_N100_h64 \
         hjb_llgc_d500_K16384_N200_h64 diffusion_dw_d100_K65536_N100_h64 diffusion_dw_d100_K65536_N100_h64_bf16; do
    STEPS=20; [ "$W" = hjb_llgc_d100_K1024_N50_h64 ] && STEPS=200
    EXTRA="--no-cpu-baseline --no-secondary"; [ "$W" = hjb_llgc_d100_K65536_N100_h64 ] && EXTRA="--no-secondary"
    python3 bench.py --workload $W --steps $STEPS --warmup 5 $EXTRA > "$OUT/${W}_bench.json" 2> "$OUT/${W}_bench.err" || exit 1
    echo "bench $W done: $(python3 -c "import json,sys; j=json.load(open('$OUT/${W}_bench.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'])")"
    rm -rf /tmp/prof_$W
    (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$W -o run --output-format csv -- \
        python3 "$ROOT/bench.py" --workload $W --steps $STEPS --warmup 5 --no-cpu-baseline --no-secondary > /dev/null 2> "$OUT/${W}_prof.err") || exit 1
    F=$(find /tmp/prof_$W -name "*kernel_stats.csv" | head -1)
    [ -n "$F" ] && cp "$F" "$OUT/${W}_kernel_stats.csv" && echo "profile $W: $(head -3 "$OUT/${W}_kernel_stats.csv" | tail -2 | cut -c1-150)"
done
