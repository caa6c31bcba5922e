#!/bin/bash
# Regenerates the judged measurement files on an MI355X box (run through gpurun from the repo root):
#   gpurun_out/final/<workload>_bench.json          bench.py JSON line (HIP-event roofline, cpu_baseline on the headline)
#   gpurun_out/final/<workload>_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary of the same command
# Copy them to profiles/r<round>_<workload>_* afterwards.  rocprofv3 gets the interpreter itself after "--".
# Usage: bash tools/refresh_profiles.sh [workload ...]
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/final
mkdir -p "$OUT"
WL="$@"
[ -z "$WL" ] && WL="hjb_llgc_d100_K65536_N100_h64 hjb_llgc_d100_K1024_N50_h64 hjb_llgc_d200_K32768_N100_h64 hjb_llgc_d500_K16384_N200_h64 diffusion_dw_d100_K65536_N100_h64 diffusion_dw_d100_K65536_N100_h64_bf16"
for W in $WL; do
    STEPS=20; WARM=5
    case $W in
        hjb_llgc_d100_K1024_N50_h64|hjb_llgc_d100_K4096_N50_h64) STEPS=200; WARM=20;;
        hjb_llgc_d500_K131072_N200_h64|hjb_llgc_d500_K131072_N200_h64_resident|hjb_llgc_d200_Kglobal262144_N100_h64) STEPS=4; WARM=1;;
        hjb_llgc_d500_K1048576_N200_h64) STEPS=2; WARM=1;;
        diffusion_allencahn_d100_K200_N25_a110) STEPS=100; WARM=10;;
        elliptic_committor_d10_K200) STEPS=20; WARM=3;;
        elliptic_committor_d10_K65536) STEPS=5; WARM=2;;
    esac
    EXTRA="--no-cpu-baseline --no-secondary"; [ "$W" = hjb_llgc_d100_K65536_N100_h64 ] && EXTRA=""
    [ "$W" = diffusion_allencahn_d100_K200_N25_a110 ] && EXTRA=""
    [ "$W" = elliptic_committor_d10_K200 ] && EXTRA=""
    timeout -k 10 900 python3 bench.py --workload $W --steps $STEPS --warmup $WARM $EXTRA > "$OUT/${W}_bench.json" 2> "$OUT/${W}_bench.err" || { echo "bench $W failed"; tail -3 "$OUT/${W}_bench.err"; exit 1; }
    echo "bench $W: $(python3 -c "import json,sys; j=json.loads(open('$OUT/${W}_bench.json').read().strip().splitlines()[-1]); print('%.4g units/s, %.3f ms/step, frac %.3f (issued %s)' % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('mfma_term',{}).get('frac_issued')))")"
    rm -rf /tmp/prof_$W
    PSTEPS=$STEPS; [ $PSTEPS -gt 40 ] && PSTEPS=40
    (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$W -o run --output-format csv -- \
        python3 "$ROOT/bench.py" --workload $W --steps $PSTEPS --warmup $WARM --no-cpu-baseline --no-secondary --no-sustained > /dev/null 2> "$OUT/${W}_prof.err") || { echo "profile $W failed"; tail -3 "$OUT/${W}_prof.err"; exit 1; }
    F=$(find /tmp/prof_$W -name "*kernel_stats.csv" | head -1)
    [ -n "$F" ] && cp "$F" "$OUT/${W}_kernel_stats.csv" && echo "profile $W: $(head -3 "$OUT/${W}_kernel_stats.csv" | tail -2 | cut -c1-160)"
done
