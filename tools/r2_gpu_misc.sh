#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r2misc
mkdir -p $OUT
timeout -k 10 600 python3 bench.py --workload hjb_llgc_d500_K1048576_N200_h64 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/hjb_llgc_d500_K1048576_N200_h64_bench.json 2> $OUT/K1M.err; echo "K1M rc=$?"
PSP_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/rehearsal2_headline.json 2> $OUT/rehearsal2.err; echo "rehearsal rc=$?"
PSP_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --warmup 2 --workload hjb_llgc_d200_Kglobal262144_N100_h64 --no-cpu-baseline > $OUT/rehearsal2_strong.json 2> $OUT/rehearsal2s.err; echo "rehearsal strong rc=$?"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2misc/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, '%.3e'%d['value'], d['ms_per_step'], d['n_gpus'], d['scaling'], d['config'].get('path_store'), d.get('collectives',{}).get('per_step_ms'))
    except Exception as e: print(f,'ERR',e)
PY
