#!/bin/bash
# every bench workload once (short), to make sure each still runs and prints a sane JSON line
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r2allbench
mkdir -p $OUT
for W in bsde_dw_d100_K65536_N100_h64 diffusion_dw_d100_K65536_N100_h64_bf16fwd hjb_llgc_d100_K65536_N100_h64_diag hjb_llgc_d500_K16384_N200_h64_diag hjb_llgc_d100_K16384_N50_h64 hjb_llgc_d100_K8192_N50_h64 hjb_llgc_d100_K4096_N50_h64; do
  timeout -k 10 300 python3 bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/$W.json 2> $OUT/$W.err; rc=$?
  python3 -c "
import json
try:
    d=json.loads([l for l in open('$OUT/$W.json') if l.startswith('{')][-1]); print('$W rc=$rc', '%.3e'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'], d['roofline']['bound'], d['loss_first_last'])
except Exception as e: print('$W rc=$rc ERR', e)
"
  [ $rc -eq 124 ] && exit 124
done
