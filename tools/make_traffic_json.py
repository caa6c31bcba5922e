#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/pmc_traffic.sh (gpurun_out/r2traffic/<workload>.txt).
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB x 1024): FETCH_SIZE reports exactly half of the bytes of 4-, 8- and
16-byte-per-lane streaming reads on gfx950 (calibrated: tools/fetch_calibrate.hip, profiles/r2_fetch_calibration.txt),
WRITE_SIZE is exact."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"_comment": __doc__.strip().replace("\n", " ")}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "r2traffic", "*.txt"))):
    w = os.path.basename(f)[:-4]
    per = {}
    for line in open(f):
        k, c, v, n = line.split()
        per.setdefault(k, {})[c] = float(v)
    ent = {}
    for k, cs in per.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs and cs["FETCH_SIZE"] + cs["WRITE_SIZE"] > 1e4:
            ent[k] = {"hbm_bytes": int((2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024), "fetch_raw_bytes": int(cs["FETCH_SIZE"] * 1024),
                      "write_bytes": int(cs["WRITE_SIZE"] * 1024)}
            if "SQ_INSTS_MFMA" in cs:
                ent[k]["mfma_instructions"] = int(cs["SQ_INSTS_MFMA"])
    if ent:
        out[w] = ent
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
