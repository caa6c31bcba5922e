#!/bin/bash
# Round-4 judged records on one MI355X box (run through gpurun from the repo root):
#   part bench: bench JSON line + rocprofv3 kernel stats per workload (tools/refresh_profiles.sh -> gpurun_out/final/)
#   part pmc / pmc2: PMC passes (tools/pmc_passes.sh -> gpurun_out/pmc/), the chunked / strong-scaling plans included
# Afterwards, in the build container:  cp gpurun_out/final/<w>_* profiles/r4_<w>_*;  python tools/r3/make_counters_json.py r4_counters.json
PART=${1:-all}
if [ "$PART" = "all" ] || [ "$PART" = "bench" ]; then
  bash tools/refresh_profiles.sh hjb_llgc_d100_K65536_N100_h64 hjb_llgc_d100_K1024_N50_h64 hjb_llgc_d100_K8192_N50_h64 \
      hjb_llgc_d100_K65536_N100_h64_fp32mfma hjb_llgc_d100_K65536_N100_h64_chunk4 \
      hjb_llgc_d100_K65536_N50_outer_h30 diffusion_dw_d100_K65536_N100_h64 diffusion_dw_d100_K65536_N100_h64_bf16 \
      diffusion_allencahn_d100_K200_N25_a110 diffusion_allencahn_d100_K16384_N25_a110 elliptic_committor_d10_K200 \
      elliptic_committor_d10_K65536 hjb_llgc_d200_K32768_N100_h64 hjb_llgc_d256_K32768_N100_h64 hjb_llgc_d500_K16384_N200_h64 \
      hjb_llgc_d200_Kglobal262144_N100_h64 hjb_llgc_d500_K131072_N200_h64 hjb_llgc_d500_K131072_N200_h64_resident || exit 1
fi
if [ "$PART" = "all" ] || [ "$PART" = "pmc" ]; then
  bash tools/r4/gpu_pmc_all.sh hjb_llgc_d100_K65536_N100_h64 diffusion_dw_d100_K65536_N100_h64 diffusion_allencahn_d100_K16384_N25_a110 \
      elliptic_committor_d10_K200 hjb_llgc_d100_K65536_N50_outer_h30 || exit 1
fi
if [ "$PART" = "all" ] || [ "$PART" = "pmc2" ]; then
  bash tools/r4/gpu_pmc_all.sh hjb_llgc_d500_K16384_N200_h64 hjb_llgc_d200_K32768_N100_h64 hjb_llgc_d100_K65536_N100_h64_chunk4 \
      hjb_llgc_d500_K131072_N200_h64 hjb_llgc_d200_Kglobal262144_N100_h64 || exit 1
fi
