#!/bin/bash
# Same-box A/B of the specialised forward instances: PSP_NO_SPEC=1 (general instances) against the default, alternating.
OUT=${1:-gpurun_out/r4_ab}
mkdir -p $OUT
B="--steps 30 --warmup 5 --no-sustained --no-cpu-baseline --no-secondary"
for w in hjb_llgc_d100_K65536_N100_h64 hjb_llgc_d100_K1024_N50_h64 hjb_llgc_d100_K8192_N50_h64 hjb_llgc_d100_K65536_N50_outer_h30 diffusion_dw_d100_K65536_N100_h64; do
  for rep in 1 2; do
    for ns in 0 1; do
      PSP_NO_SPEC=$ns timeout -k 10 150 python bench.py --workload $w $B > $OUT/${w}_nospec${ns}_$rep.json 2> $OUT/${w}_nospec${ns}_$rep.err
      python - $OUT/${w}_nospec${ns}_$rep.json "$w no_spec=$ns rep$rep" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); rf = r["roofline"]
    print(sys.argv[2], "ms %.4f" % r["ms_per_step"], "fwd %.4f" % rf.get("fwd_kernel_ms"), "bwd %.4f" % rf.get("bwd_kernel_ms"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    done
  done
done
